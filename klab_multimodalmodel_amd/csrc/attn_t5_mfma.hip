// T5 attention core on the matrix cores (bf16 operands, fp32 accumulation) -- the production form of
// attn_t5.hip's kernels (same C ABI; that file keeps the fp32 parity mode and the generic fallback).
//
//   forward : S^T = K Q^T is computed "swapped" (MFMA A = K rows, B = Q rows), so each lane owns ONE query
//             (column l&15) and four consecutive keys per 16x16 tile: the softmax reduction is in-register
//             plus two wave shuffles (xor 16, 32), and the probabilities are already the B operand of
//             O^T = V^T P^T (sum over the accumulator's row index: no LDS round trip, no cross-lane moves;
//             the k-order permutation this implies is applied to V's ds_read_b64_tr_b16 rows instead).
//   backward: recomputes P from the saved log-sum-exp in BOTH orientations -- swapped tiles feed
//             dQ^T = K^T dS^T (sum over keys), unswapped tiles feed dV^T = dO^T P and dK^T = Q^T dS (sum over
//             queries) -- because an accumulator can only be contracted over its row index for free.  The
//             extra S/dP MFMAs are ~2 % of the step; what they buy is a kernel with no P/dS matrix in LDS.
//   K, V, Q, dO of one (batch, head) are staged once per workgroup: a row image (16-B padded rows, plain
//   ds_read_b128 fragments) and, where a product contracts over the sequence index, a transposed-read image
//   (TrImg: 8-row groups displaced by 32 dwords, conflict-free ds_read_b64_tr_b16).
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "common.h"
#include "klab_mm.h"

namespace klab {

struct AttnMP {
  const bf16_t* q; long ldq;
  const bf16_t* k; long ldk;
  const bf16_t* v; long ldv;
  const float* bias; int causal;
  bf16_t* ctx; long ldo;
  float* lse;
  int B, H, Lq, Lk;
  float p; const uint32_t* seed; uint32_t tag;
  const bf16_t* dctx; long lddo;
  bf16_t* dq; long lddq;
  bf16_t* dkk; long lddk;
  bf16_t* dv; long lddv;
  float* dbias;
  bf16_t* ds_ws;
  int defer_reduce;
  const float* score_scale;  // backward only: per-head factor applied to Q K^T in fp32 before the bias (Swin-V2 cosine attention)
  int bias_mod;              // backward only: > 0 => bias is [bias_mod, H, Lq, Lk], slab (b % bias_mod) (window-dependent shift masks)
};

template <int COLS> struct TrImg {  // rows = contraction index, columns = COLS 16-bit elements
  static constexpr int PD = (COLS == 128) ? 72 : (COLS == 64 ? 40 : (COLS == 32 ? 24 : 8));  // dwords, >= COLS/2, PD/8 odd
  static constexpr int PITCHB = PD * 4;
  static constexpr int GROUPB = (8 * PD + 32) * 4;
  __device__ static __forceinline__ int off(int row, int col) { return (row >> 3) * GROUPB + (row & 7) * PITCHB + col * 2; }
  __host__ __device__ static constexpr size_t bytes(int rows) { return (size_t)((rows + 7) / 8) * GROUPB; }
};

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4_t;

// A-operand fragment for a product that sums over this image's ROW index, k-step of 32 rows starting at row0,
// 16 output columns starting at col0, with the k order kappa(g, j) = 16*(j>>2) + 4*g + (j&3) that matches an
// accumulator tile pair used as the B operand.
template <int COLS>
__device__ __forceinline__ bf16x8 tr_frag(const char* img, int row0, int col0, int lane) {
  const int g = lane >> 4, q4 = (lane & 15) >> 2, pp = lane & 3;
  const int r = row0 + 4 * g + q4;
  const char* a0 = img + TrImg<COLS>::off(r, col0 + 4 * pp);
  const char* a1 = img + TrImg<COLS>::off(r + 16, col0 + 4 * pp);
  bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t*)a0);
  bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t*)a1);
  return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
// plain row-image fragment: row r0 + (lane&15), 8 consecutive columns at 32*ks + 8*(lane>>4)
__device__ __forceinline__ bf16x8 row_frag(const char* img, int pitchB, int r0, int ks, int lane) {
  return *reinterpret_cast<const bf16x8*>(img + (r0 + (lane & 15)) * pitchB + ks * 64 + (lane >> 4) * 16);
}
__device__ __forceinline__ bf16x8 pack8(const f32x4& a, const f32x4& b) {
  return bf16x8{(bf16_t)a[0], (bf16_t)a[1], (bf16_t)a[2], (bf16_t)a[3], (bf16_t)b[0], (bf16_t)b[1], (bf16_t)b[2], (bf16_t)b[3]};
}

// stage rows [0, L) of a [B*L, ld] matrix slice (head h) into a row image and/or a transposed-read image
template <int DK, int DKP, bool ROWIMG, bool TRIMG>
__device__ __forceinline__ void stage(const bf16_t* __restrict__ g, long ld, int b, int h, int L, int Lp, char* rowimg, int pitchB,
                                      char* trimg) {
  constexpr int CPR = DKP / 8;
  for (int ch = threadIdx.x; ch < Lp * CPR; ch += blockDim.x) {
    const int r = ch / CPR, c = (ch % CPR) * 8;
    bf16x8 v = {};
    if (r < L && c < DK) v = *reinterpret_cast<const bf16x8*>(g + ((long)b * L + r) * ld + (long)h * DK + c);
    if (ROWIMG) *reinterpret_cast<bf16x8*>(rowimg + r * pitchB + c * 2) = v;
    if (TRIMG && c < ((DK + 15) / 16) * 16) *reinterpret_cast<bf16x8*>(trimg + TrImg<DKP>::off(r, c)) = v;
  }
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
// one wave's 16 queries (q0 .. q0+15) against the staged keys / values of (b, h): scores, softmax, dropout, P V, stores.
// qf = the wave's Q fragments (row q0 + (lane & 15), columns 32 ks + 8 (lane >> 4) .. + 7).
template <int DK, int MAXT>
__device__ __forceinline__ void attn_fwd_core(const AttnMP& p, const char* Kr, const char* Vt, const bf16x8 (&qf)[(DK + 31) / 32], int b, int h,
                                              int q0, int lane) {
  constexpr int KS = (DK + 31) / 32, DKP = KS * 32, DT = (DK + 15) / 16;
  constexpr int KPITCH = DKP * 2 + 16;
  const int Lq = p.Lq, Lk = p.Lk;
  const int Lkp = (Lk + 31) & ~31, NT = Lkp / 16;
  const int g = lane >> 4;
  const int q = q0 + (lane & 15);
  const int qc = q < Lq ? q : Lq - 1;
  f32x4 s[MAXT];
#pragma unroll
  for (int t = 0; t < MAXT; ++t) {
    s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (t < NT) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(Kr, KPITCH, t * 16, ks, lane), qf[ks], s[t], 0, 0, 0);
    }
  }
  // s[t][r] = S[q][key = 16 t + 4 g + r]
  float m = -INFINITY;
  const float* brow = p.bias ? p.bias + ((long)h * Lq + qc) * Lk : nullptr;
#pragma unroll
  for (int t = 0; t < MAXT; ++t) {
    if (t < NT) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = t * 16 + g * 4 + r;
        // branch-free: the bias read is clamped into the row, an excluded key is replaced by the select
        float x = s[t][r] + (brow ? brow[key < Lk ? key : Lk - 1] : 0.f);
        x = (key < Lk && !(p.causal && key > q)) ? x : -INFINITY;
        s[t][r] = x;
        m = fmaxf(m, x);
      }
    }
  }
  m = fmaxf(m, __shfl_xor(m, 16, 64));
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < MAXT; ++t) {
    if (t < NT) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float e = __expf(s[t][r] - m); s[t][r] = e; sum += e; }
    }
  }
  sum += __shfl_xor(sum, 16, 64);
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.f / sum;
  if (g == 0 && q < Lq && p.lse) p.lse[((long)b * p.H + h) * Lq + q] = m + __logf(sum);
  const DropCtx dc = drop_slab(make_drop(p.seed, p.tag, p.p), (uint32_t)(b * p.H + h));
  const uint32_t base = (uint32_t)qc * (uint32_t)Lk;
#pragma unroll
  for (int t = 0; t < MAXT; ++t) {
    if (t < NT) {
#pragma unroll
      for (int r = 0; r < 4; ++r) s[t][r] *= inv * drop_mult32_nb(dc, base + t * 16 + g * 4 + r);
    }
  }
  // O^T[d][q] = sum_key V[key][d] * P[q][key]
  f32x4 o[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int sidx = 0; sidx < MAXT / 2; ++sidx) {
    if (2 * sidx < NT) {
      const bf16x8 pf = pack8(s[2 * sidx], s[2 * sidx + 1]);
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag<DKP>(Vt, sidx * 32, dt * 16, lane), pf, o[dt], 0, 0, 0);
    }
  }
  if (q < Lq) {
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const int d = dt * 16 + g * 4;
      if (d < DK) *reinterpret_cast<bf16x4*>(p.ctx + ((long)b * Lq + q) * p.ldo + (long)h * DK + d) =
          bf16x4{(bf16_t)o[dt][0], (bf16_t)o[dt][1], (bf16_t)o[dt][2], (bf16_t)o[dt][3]};
    }
  }
}

template <int DK, int MAXT>
__global__ __launch_bounds__(256) void t5_attn_fwd_mfma(AttnMP p) {
  constexpr int KS = (DK + 31) / 32, DKP = KS * 32;
  constexpr int KPITCH = DKP * 2 + 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int Lq = p.Lq, Lk = p.Lk;
  const int Lkp = (Lk + 31) & ~31;
  char* Kr = smem;
  char* Vt = smem + (size_t)Lkp * KPITCH;
  const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  stage<DK, DKP, true, false>(p.k, p.ldk, b, h, Lk, Lkp, Kr, KPITCH, nullptr);
  stage<DK, DKP, false, true>(p.v, p.ldv, b, h, Lk, Lkp, nullptr, 0, Vt);
  __syncthreads();
  const int q0 = blockIdx.x * 64 + wave * 16;
  if (q0 >= Lq) return;
  const int g = lane >> 4;
  const int q = q0 + (lane & 15);
  const int qc = q < Lq ? q : Lq - 1;
  bf16x8 qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int c = ks * 32 + g * 8;
    bf16x8 v = {};
    if (c < DK) v = *reinterpret_cast<const bf16x8*>(p.q + ((long)b * Lq + qc) * p.ldq + (long)h * DK + c);
    qf[ks] = v;
  }
  attn_fwd_core<DK, MAXT>(p, Kr, Vt, qf, b, h, q0, lane);
}

// ------------------------------------------------------------------------------------------------
// fused front half of a T5 attention sub-layer (forward): T5LayerNorm -> q|k|v (self) or q (cross) projection -> attention
// (HF/t5:59-72 + :206-209, 281-369 + :144-173) in ONE launch, one workgroup per (sample, head).  For d_model = 512, head dim 64
// and at most 64 queries / keys per sample (T5-small at the caption shapes: Le = 58, Lt = 64).  Replaces three launches of the
// serial chain (klab_rmsnorm_fwd, the projection klab_gemm, klab_t5_attn_fwd); what the backward pass reads is still written:
// the normalised rows (bf16) + 1/rms by the head-0 workgroup, the projected q|k|v in the fused projection buffer's layout,
// the log-sum-exp.  The dropout masks are the attention kernel's (same indices); the unfused path and this one differ only in
// summation order (sum of squares, projection).
//   1. each lane normalises ITS 128 values of a row (row 16 wave + (lane & 15), columns 32 ks + 8 (lane >> 4) .. + 7) straight
//      into the projection's B-operand fragments: the normalised rows never pass through LDS;
//   2. the head's projection rows (192 x 512 for q|k|v, 64 x 512 for q) stream through a 3-slot LDS-DMA ring in k-tiles of 64
//      (128-byte rows, chunk c of row r at c ^ (r & 7)); per k-tile 2 x NJ MFMAs per wave, W as the A operand: a lane ends up
//      with 4 consecutive projection columns of its row; 72 KiB of LDS, two workgroups per CU;
//   3. q|k|v go to memory (bf16) and into the attention's LDS images (Q, K row images; V transposed-read image), which alias
//      the ring; cross attention stages K / V from the projected encoder output instead;
//   4. attn_fwd_core, unchanged.
struct AttnFusedP {
  const float* x; const float* gamma; float eps;
  const bf16_t* w;             // SELF: [3 * inner, 512] rows q | k | v; CROSS: [inner, 512] rows q
  bf16_t* xn; float* rstd;     // [B * Lq, 512], [B * Lq]
  bf16_t* proj; long ldproj;   // SELF: [B * Lq, 3 * inner]; CROSS: [B * Lq, inner]
  AttnMP a;
  int ablate;  // diagnostics (KLAB_AF_ABLATE): 1 = no attention core, 2 = no projection MFMAs / weight stream, 4 = no q|k|v copy-out
};
constexpr int AF_D = 512, AF_DK = 64, AF_S = 3;
template <bool CROSS>
__global__ __launch_bounds__(256, 2) void t5_attn_fused_fwd(AttnFusedP f) {
  constexpr int D = AF_D, DK = AF_DK, NROWS = CROSS ? 64 : 192, NJ = NROWS / 16, SLOT = NROWS * 128, LPW = NROWS / 32;
  constexpr int KPITCH = DK * 2 + 16, NKS = D / 32;
  constexpr int IMAGES = 3 * 64 * KPITCH + (int)TrImg<DK>::bytes(64);
  // the four 4-KiB strips of the norm prologue: ring slot 2 when it is large enough (self: 24 KiB), else behind the ring
  constexpr int STRIP_OFF = SLOT >= 16384 ? 2 * SLOT : AF_S * SLOT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ring = smem;  // AF_S slots of the weight stream; later the attention images
  const AttnMP& p = f.a;
  const int Lq = p.Lq, inner = p.H * DK;
  // Workgroup -> (sample, head).  The H workgroups of a sample all read its fp32 rows (H x 128 KiB at 64 rows): dealt round-robin
  // over the 8 XCDs as consecutive block ids, every XCD's L2 fetched every sample (64 MB through the fabric at B = 64, ~13 us of
  // the kernel).  Blocks i and i + 8 share an XCD, so the heads of one sample are given block ids that are equal mod 8: its rows
  // are fetched into ONE L2 and hit there H - 1 times.  Speed only; any mapping is correct.
  int b, h;
  {
    const int i = blockIdx.x;
    if ((p.B & 7) == 0) { const int xcd = i & 7, k = i >> 3; h = k % p.H; b = (k / p.H) * 8 + xcd; }
    else { b = i / p.H; h = i % p.H; }
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4;

  // projection rows through the ring: image row i <- weight row (i / 64) * inner + h * 64 + i % 64
  const bf16_t* wsrc[LPW];
#pragma unroll
  for (int i = 0; i < LPW; ++i) {
    const int r = (wave * LPW + i) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ (lane >> 3);
    wsrc[i] = f.w + ((long)(r >> 6) * inner + h * 64 + (r & 63)) * D + c * 8;
  }
  auto issue = [&](int kt) {
    if (kt >= D / 64) return;
    char* st = ring + (kt % AF_S) * SLOT;
#pragma unroll
    for (int i = 0; i < LPW; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[i] + kt * 64),
                                       (__attribute__((address_space(3))) void*)(st + (wave * LPW + i) * 1024), 16, 0, 0);
  };
  if (!(f.ablate & 2)) { issue(0); issue(1); }
  // 1. T5LayerNorm into MFMA fragments.  A wave reads its 16 rows as whole rows -- two fully coalesced 1-KiB loads per row, lane l
  //    holding columns 4 l .. + 3 and 256 + 4 l .. + 3 (loading each lane's fragment columns directly made every load touch
  //    half-lines of 16 rows: +5 us per launch) --, reduces the sum of squares per row across the wave, and passes the normalised
  //    bf16 rows, four rows at a time, through a 4-KiB wave-private LDS strip into the projection's B-operand layout: lane
  //    (row 16 wave + (lane & 15), g = lane >> 4) ends up with columns 32 ks + 8 g .. + 7 for ks = 0 .. 15.  The strips live
  //    in ring slot 2 (self) / behind the ring (cross), which the weight stream does not touch before the first barrier below.
  const int xr = wave * 16 + (lane & 15);
  const bool xvalid = xr < Lq;
  const long xrow = (long)b * Lq + (xvalid ? xr : Lq - 1);
  bf16x8 xf[NKS];
  {
    char* strip = smem + STRIP_OFF + wave * 4096;
    const f32x4 g0 = *reinterpret_cast<const f32x4*>(f.gamma + lane * 4), g1 = *reinterpret_cast<const f32x4*>(f.gamma + 256 + lane * 4);
    f32x4 v0[16], v1[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int r = wave * 16 + i;
      const long grow = (long)b * Lq + (r < Lq ? r : Lq - 1);  // clamped: rows past the sequence are zeroed below
      v0[i] = *reinterpret_cast<const f32x4*>(f.x + grow * D + lane * 4);
      v1[i] = *reinterpret_cast<const f32x4*>(f.x + grow * D + 256 + lane * 4);
    }
    float rsv[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float ss = v0[i][0] * v0[i][0] + v0[i][1] * v0[i][1] + v0[i][2] * v0[i][2] + v0[i][3] * v0[i][3];
      ss += v1[i][0] * v1[i][0] + v1[i][1] * v1[i][1] + v1[i][2] * v1[i][2] + v1[i][3] * v1[i][3];
      ss = wave_sum(ss);
      rsv[i] = (wave * 16 + i < Lq) ? rsqrtf(ss / (float)D + f.eps) : 0.f;  // rows past the sequence project to zero
    }
#pragma unroll
    for (int bt = 0; bt < 4; ++bt) {  // four rows per pass through the strip
#pragma unroll
      for (int ii = 0; ii < 4; ++ii) {
        const int i = bt * 4 + ii;
        const float rs = rsv[i];
        const bf16x4 o0 = bf16x4{(bf16_t)(g0[0] * (v0[i][0] * rs)), (bf16_t)(g0[1] * (v0[i][1] * rs)), (bf16_t)(g0[2] * (v0[i][2] * rs)), (bf16_t)(g0[3] * (v0[i][3] * rs))};
        const bf16x4 o1 = bf16x4{(bf16_t)(g1[0] * (v1[i][0] * rs)), (bf16_t)(g1[1] * (v1[i][1] * rs)), (bf16_t)(g1[2] * (v1[i][2] * rs)), (bf16_t)(g1[3] * (v1[i][3] * rs))};
        // strip row ii (1 KiB): 16-byte chunk c at position c ^ (ii << 2) (the four rows a read touches land in different
        // 64-byte groups of the bank row); columns 4 l .. + 3 = half (l & 1) of chunk l >> 1
        *reinterpret_cast<bf16x4*>(strip + ii * 1024 + (((lane >> 1) ^ (ii << 2)) * 16) + (lane & 1) * 8) = o0;
        *reinterpret_cast<bf16x4*>(strip + ii * 1024 + (((32 + (lane >> 1)) ^ (ii << 2)) * 16) + (lane & 1) * 8) = o1;
        if (h == 0 && wave * 16 + i < Lq) {  // what the backward pass reads: normalised rows + 1/rms, once per sample
          const long grow = (long)b * Lq + wave * 16 + i;
          *reinterpret_cast<bf16x4*>(f.xn + grow * D + lane * 4) = o0;
          *reinterpret_cast<bf16x4*>(f.xn + grow * D + 256 + lane * 4) = o1;
          if (lane == 0) f.rstd[grow] = rs;
        }
      }
      // the lanes whose row is in this pass pick up their 16 fragments (wave-private strip, LDS operations of a wave are in order)
      const int ii = (lane & 15) - bt * 4;
      if (ii >= 0 && ii < 4) {
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
          xf[ks] = *reinterpret_cast<const bf16x8*>(strip + ii * 1024 + (((4 * ks + g) ^ (ii << 2)) * 16));
      }
    }
  }

  // (the strip reads have returned before any wave can pass the first barrier below and refill ring slot 2)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

  // 2. projection: acc[j][r] = out[row 16 wave + (lane & 15)][column 16 j + 4 g + r]
  f32x4 acc[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (!(f.ablate & 2))
#pragma unroll
  for (int kt = 0; kt < D / 64; ++kt) {
    if (kt + 1 < D / 64) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // raw barrier: a __syncthreads() carries a fence that drains vmcnt, i.e. the k-tile still in flight
    __builtin_amdgcn_s_barrier();
    const char* Ws = ring + (kt % AF_S) * SLOT;
    issue(kt + 2);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int wr = j * 16 + (lane & 15);
        const bf16x8 wf = *reinterpret_cast<const bf16x8*>(Ws + wr * 128 + (((4 * ks + g) ^ (wr & 7)) * 16));
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf[2 * kt + ks], acc[j], 0, 0, 0);
      }
    }
  }
  __syncthreads();  // every wave is done with the ring: it becomes the attention's images

  // 3. q | k | v -> LDS images (Q, K, V row images for the attention and the copy-out; V also as a transposed-read image), then to
  //    memory as whole 128-byte head slices (straight from the accumulators a store instruction wrote 32-byte pieces of 16 rows)
  const int Lk = p.Lk, Lkp = (Lk + 31) & ~31;
  char* Qr = ring;
  char* Kr = Qr + 64 * KPITCH;
  char* Vr = Kr + 64 * KPITCH;  // (64 rows each whatever Lk is: the self form writes all 64 projected rows)
  char* Vt = Vr + 64 * KPITCH;
  {
    const int R = xr;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int blk = j >> 2, nn = (j & 3) * 16 + g * 4;  // 0: q, 1: k, 2: v; column inside the head
      const bf16x4 o = bf16x4{(bf16_t)acc[j][0], (bf16_t)acc[j][1], (bf16_t)acc[j][2], (bf16_t)acc[j][3]};
      if (blk == 0) *reinterpret_cast<bf16x4*>(Qr + R * KPITCH + nn * 2) = o;
      else if (blk == 1) *reinterpret_cast<bf16x4*>(Kr + R * KPITCH + nn * 2) = o;
      else {
        *reinterpret_cast<bf16x4*>(Vr + R * KPITCH + nn * 2) = o;
        *reinterpret_cast<bf16x4*>(Vt + TrImg<DK>::off(R, nn)) = o;
      }
    }
  }
  if constexpr (CROSS) {
    stage<DK, DK, true, false>(p.k, p.ldk, b, h, Lk, Lkp, Kr, KPITCH, nullptr);
    stage<DK, DK, false, true>(p.v, p.ldv, b, h, Lk, Lkp, nullptr, 0, Vt);
  }
  __syncthreads();
  if (!(f.ablate & 4)) {
    constexpr int NB = CROSS ? 1 : 3;
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
      const char* img = blk == 0 ? Qr : (blk == 1 ? Kr : Vr);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int ch = tid + u * 256, row = ch >> 3, cc = ch & 7;
        if (row < Lq)
          *reinterpret_cast<bf16x8*>(f.proj + ((long)b * Lq + row) * f.ldproj + (long)blk * inner + h * DK + cc * 8) =
              *reinterpret_cast<const bf16x8*>(img + row * KPITCH + cc * 16);
      }
    }
  }

  // 4. attention of the wave's 16 queries
  const int q0 = wave * 16;
  if (q0 >= Lq || (f.ablate & 1)) return;
  bf16x8 qf[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) qf[ks] = row_frag(Qr, KPITCH, q0, ks, lane);
  attn_fwd_core<DK, 4>(p, Kr, Vt, qf, b, h, q0, lane);
}

// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
template <int DK>
__global__ __launch_bounds__(256) void t5_attn_bwd_mfma(AttnMP p) {
  constexpr int KS = (DK + 31) / 32, DKP = KS * 32, DT = (DK + 15) / 16;
  constexpr int KPITCH = DKP * 2 + 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int Lq = p.Lq, Lk = p.Lk;
  const int Lqp = (Lq + 31) & ~31, Lkp = (Lk + 31) & ~31;
  const int NQ = Lqp / 16, NK = Lkp / 16;
  char* Qr = smem;
  char* Kr = Qr + (size_t)Lqp * KPITCH;
  char* Vr = Kr + (size_t)Lkp * KPITCH;
  char* dOr = Vr + (size_t)Lkp * KPITCH;
  char* Qt = dOr + (size_t)Lqp * KPITCH;
  char* Kt = Qt + (size_t)(Lqp / 8) * TrImg<DKP>::GROUPB;
  char* dOt = Kt + (size_t)(Lkp / 8) * TrImg<DKP>::GROUPB;
  float* delta = reinterpret_cast<float*>(dOt + (size_t)(Lqp / 8) * TrImg<DKP>::GROUPB);
  float* lses = delta + Lqp;
  const int bh = blockIdx.x, b = bh / p.H, h = bh % p.H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;

  const float* const biasp = p.bias ? p.bias + (p.bias_mod > 0 ? (long)(b % p.bias_mod) * p.H * Lq * Lk : 0L) : nullptr;
  const float sscale = p.score_scale ? p.score_scale[h] : 1.f;
  // position-bias values of each wave's FIRST tile pair in both phases: requested before anything else, so that they
  // arrive under the staging pass instead of as a separate global round trip in front of each phase
  f32x4 preA[2], preB[2];
  {
    const int qa = wave * 16 + (lane & 15), kb_ = wave * 16 + (lane & 15);
    const float* browa = (biasp && qa < Lq) ? biasp + ((long)h * Lq + qa) * Lk : nullptr;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = u * 16 + g * 4 + r, qq = u * 16 + g * 4 + r;
        preA[u][r] = browa ? browa[key < Lk ? key : Lk - 1] : 0.f;
        preB[u][r] = biasp ? biasp[((long)h * Lq + (qq < Lq ? qq : Lq - 1)) * Lk + (kb_ < Lk ? kb_ : Lk - 1)] : 0.f;
      }
  }
  bool staged_delta = false;  // the fast staging path below also produces delta / lse (no second round of loads)
  {
    // all global loads of the four operand slices go out before the first LDS store (one memory round trip instead of
    // one per chunk: the per-matrix loops were 8-12 dependent load -> store pairs)
    constexpr int CPR = DKP / 8, MAXC = 4;
    const int nq = Lqp * CPR, nk = Lkp * CPR;
    if (nq <= MAXC * 256 && nk <= MAXC * 256) {
      bf16x8 vq[MAXC], vk[MAXC], vv[MAXC], vd[MAXC], vo[MAXC];
      float lsev[MAXC];
#pragma unroll
      for (int i = 0; i < MAXC; ++i) {
        const int ch = threadIdx.x + i * 256, r = ch / CPR, c = (ch % CPR) * 8;
        vq[i] = bf16x8{}; vk[i] = bf16x8{}; vv[i] = bf16x8{}; vd[i] = bf16x8{}; vo[i] = bf16x8{};
        lsev[i] = INFINITY;  // padded queries: P = exp(-inf) = 0
        if (ch < nq && r < Lq && c < DK) {
          vq[i] = *reinterpret_cast<const bf16x8*>(p.q + ((long)b * Lq + r) * p.ldq + (long)h * DK + c);
          vd[i] = *reinterpret_cast<const bf16x8*>(p.dctx + ((long)b * Lq + r) * p.lddo + (long)h * DK + c);
          if constexpr (DK % 8 == 0) vo[i] = *reinterpret_cast<const bf16x8*>(p.ctx + ((long)b * Lq + r) * p.ldo + (long)h * DK + c);
          if (c == 0) lsev[i] = p.lse[((long)b * p.H + h) * Lq + r];
        }
        if (ch < nk && r < Lk && c < DK) {
          vk[i] = *reinterpret_cast<const bf16x8*>(p.k + ((long)b * Lk + r) * p.ldk + (long)h * DK + c);
          vv[i] = *reinterpret_cast<const bf16x8*>(p.v + ((long)b * Lk + r) * p.ldv + (long)h * DK + c);
        }
      }
#pragma unroll
      for (int i = 0; i < MAXC; ++i) {
        const int ch = threadIdx.x + i * 256, r = ch / CPR, c = (ch % CPR) * 8;
        const bool tr = c < ((DK + 15) / 16) * 16;
        if (ch < nq) {
          *reinterpret_cast<bf16x8*>(Qr + r * KPITCH + c * 2) = vq[i];
          *reinterpret_cast<bf16x8*>(dOr + r * KPITCH + c * 2) = vd[i];
          if (tr) {
            *reinterpret_cast<bf16x8*>(Qt + TrImg<DKP>::off(r, c)) = vq[i];
            *reinterpret_cast<bf16x8*>(dOt + TrImg<DKP>::off(r, c)) = vd[i];
          }
        }
        if (ch < nk) {
          *reinterpret_cast<bf16x8*>(Kr + r * KPITCH + c * 2) = vk[i];
          *reinterpret_cast<bf16x8*>(Vr + r * KPITCH + c * 2) = vv[i];
          if (tr) *reinterpret_cast<bf16x8*>(Kt + TrImg<DKP>::off(r, c)) = vk[i];
        }
        // delta[q] = dO[q,:] . O[q,:] from the chunks already in registers: the CPR lanes of a row are neighbours
        float a = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) a += (float)vd[i][u] * (float)vo[i][u];
#pragma unroll
        for (int o2 = 1; o2 < CPR; o2 <<= 1) a += __shfl_xor(a, o2, 64);
        if (ch < nq && c == 0) { delta[r] = a; lses[r] = lsev[i]; }
      }
      staged_delta = true;
    } else {
      stage<DK, DKP, true, true>(p.q, p.ldq, b, h, Lq, Lqp, Qr, KPITCH, Qt);
      stage<DK, DKP, true, true>(p.k, p.ldk, b, h, Lk, Lkp, Kr, KPITCH, Kt);
      stage<DK, DKP, true, false>(p.v, p.ldv, b, h, Lk, Lkp, Vr, KPITCH, nullptr);
      stage<DK, DKP, true, true>(p.dctx, p.lddo, b, h, Lq, Lqp, dOr, KPITCH, dOt);
    }
  }
  // delta[q] = dO[q,:] . O[q,:] and the row's log-sum-exp: 4 lanes per row with 8-byte loads, all rows of the block in
  // flight at once (one row per wave at a time made this ~20 dependent global round trips, most of the kernel's time)
  for (int r0 = 0; r0 < (staged_delta ? 0 : Lqp); r0 += 64) {
    constexpr int EPS = DK / 4;
    const int qq = r0 + (threadIdx.x >> 2), seg = threadIdx.x & 3;
    float a = 0.f;
    if (qq < Lq) {
      const bf16_t* dop = p.dctx + ((long)b * Lq + qq) * p.lddo + (long)h * DK + seg * EPS;
      const bf16_t* op = p.ctx + ((long)b * Lq + qq) * p.ldo + (long)h * DK + seg * EPS;
#pragma unroll
      for (int c = 0; c < EPS; c += 4) {
        const bf16x4 x = *reinterpret_cast<const bf16x4*>(dop + c), y = *reinterpret_cast<const bf16x4*>(op + c);
        a += (float)x[0] * (float)y[0] + (float)x[1] * (float)y[1] + (float)x[2] * (float)y[2] + (float)x[3] * (float)y[3];
      }
    }
    a += __shfl_xor(a, 1, 64);
    a += __shfl_xor(a, 2, 64);
    if (seg == 0 && qq < Lqp) {
      delta[qq] = a;
      lses[qq] = qq < Lq ? p.lse[((long)b * p.H + h) * Lq + qq] : INFINITY;  // padded queries: P = exp(-inf) = 0
    }
  }
  __syncthreads();
  const DropCtx dc = drop_slab(make_drop(p.seed, p.tag, p.p), (uint32_t)(b * p.H + h));

  // ---- phase A: swapped tiles, one query tile per wave iteration -> dQ ----
  for (int qt = wave; qt < NQ; qt += 4) {
    const int q = qt * 16 + (lane & 15);
    const float lq = lses[q], dq_ = delta[q];
    const float* brow = (biasp && q < Lq) ? biasp + ((long)h * Lq + q) * Lk : nullptr;
    float* dbrow = (p.dbias && !p.ds_ws && q < Lq) ? p.dbias + ((long)h * Lq + q) * Lk : nullptr;
    bf16_t* dsrow = (p.ds_ws && q < Lq) ? p.ds_ws + (((long)b * p.H + h) * Lq + q) * Lkp : nullptr;
    bf16x8 qf[KS], dof[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) { qf[ks] = row_frag(Qr, KPITCH, qt * 16, ks, lane); dof[ks] = row_frag(dOr, KPITCH, qt * 16, ks, lane); }
    f32x4 acc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) acc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    // position-bias values of the NEXT key pair are fetched while the current one is computed (clamped addresses)
    auto load_bias_a = [&](int sidx, f32x4 (&bb)[2]) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = (2 * sidx + u) * 16 + g * 4 + r;
          bb[u][r] = brow ? brow[key < Lk ? key : Lk - 1] : 0.f;
        }
    };
    f32x4 bcur[2], bnxt[2];
    if (qt == wave) { bcur[0] = preA[0]; bcur[1] = preA[1]; }
    else load_bias_a(0, bcur);
    for (int sidx = 0; sidx < NK / 2; ++sidx) {
      if (sidx + 1 < NK / 2) load_bias_a(sidx + 1, bnxt);
      f32x4 ds2[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int t = 2 * sidx + u;
        f32x4 st = {0.f, 0.f, 0.f, 0.f}, dpt = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          st = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(Kr, KPITCH, t * 16, ks, lane), qf[ks], st, 0, 0, 0);
          dpt = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(Vr, KPITCH, t * 16, ks, lane), dof[ks], dpt, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = t * 16 + g * 4 + r;
          const bool ok = key < Lk && q < Lq && !(p.causal && key > q);
          const float pr = ok ? __expf(st[r] * sscale + bcur[u][r] - lq) : 0.f;  // (branch-free: see the forward)
          const float dsv = pr * (dpt[r] * drop_mult32_nb(dc, (uint32_t)q * (uint32_t)Lk + key) - dq_);
          if (dbrow) { if (ok) atomicAdd(dbrow + key, dsv); }
          ds2[u][r] = dsv;
        }
      }
      const bf16x8 dsf = pack8(ds2[0], ds2[1]);
      if (dsrow) {  // dS tile pair for the batch reduction (keys 32 s + 4 g .. and + 16)
        *reinterpret_cast<bf16x4*>(dsrow + sidx * 32 + g * 4) = bf16x4{dsf[0], dsf[1], dsf[2], dsf[3]};
        *reinterpret_cast<bf16x4*>(dsrow + sidx * 32 + 16 + g * 4) = bf16x4{dsf[4], dsf[5], dsf[6], dsf[7]};
      }
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag<DKP>(Kt, sidx * 32, dt * 16, lane), dsf, acc[dt], 0, 0, 0);
      bcur[0] = bnxt[0]; bcur[1] = bnxt[1];
    }
    if (q < Lq) {
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const int d = dt * 16 + g * 4;
        if (d < DK) *reinterpret_cast<bf16x4*>(p.dq + ((long)b * Lq + q) * p.lddq + (long)h * DK + d) =
            bf16x4{(bf16_t)acc[dt][0], (bf16_t)acc[dt][1], (bf16_t)acc[dt][2], (bf16_t)acc[dt][3]};
      }
    }
  }

  // ---- phase B: unswapped tiles, one key tile per wave iteration -> dV, dK ----
  for (int kt = wave; kt < NK; kt += 4) {
    const int key = kt * 16 + (lane & 15);
    bf16x8 kf[KS], vf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) { kf[ks] = row_frag(Kr, KPITCH, kt * 16, ks, lane); vf[ks] = row_frag(Vr, KPITCH, kt * 16, ks, lane); }
    f32x4 av[DT], ak[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) { av[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; ak[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    auto load_bias_b = [&](int sidx, f32x4 (&bb)[2]) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int q = (2 * sidx + u) * 16 + g * 4 + r;
          bb[u][r] = biasp ? biasp[((long)h * Lq + (q < Lq ? q : Lq - 1)) * Lk + (key < Lk ? key : Lk - 1)] : 0.f;
        }
    };
    f32x4 bcur[2], bnxt[2];
    if (kt == wave) { bcur[0] = preB[0]; bcur[1] = preB[1]; }
    else load_bias_b(0, bcur);
    for (int sidx = 0; sidx < NQ / 2; ++sidx) {
      if (sidx + 1 < NQ / 2) load_bias_b(sidx + 1, bnxt);
      f32x4 pd2[2], ds2[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int qt = 2 * sidx + u;
        f32x4 st = {0.f, 0.f, 0.f, 0.f}, dpt = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          st = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(Qr, KPITCH, qt * 16, ks, lane), kf[ks], st, 0, 0, 0);
          dpt = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(dOr, KPITCH, qt * 16, ks, lane), vf[ks], dpt, 0, 0, 0);
        }
        // st[r] = S[q = 16 qt + 4 g + r][key]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int q = qt * 16 + g * 4 + r;
          const bool ok = key < Lk && q < Lq && !(p.causal && key > q);
          const float pr = ok ? __expf(st[r] * sscale + bcur[u][r] - lses[q]) : 0.f;
          const float mlt = drop_mult32_nb(dc, (uint32_t)q * (uint32_t)Lk + key);
          const float pdv = pr * mlt, dsv = pr * (dpt[r] * mlt - delta[q]);
          pd2[u][r] = pdv; ds2[u][r] = dsv;
        }
      }
      const bf16x8 pdf = pack8(pd2[0], pd2[1]), dsf = pack8(ds2[0], ds2[1]);
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        av[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag<DKP>(dOt, sidx * 32, dt * 16, lane), pdf, av[dt], 0, 0, 0);
        ak[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag<DKP>(Qt, sidx * 32, dt * 16, lane), dsf, ak[dt], 0, 0, 0);
      }
      bcur[0] = bnxt[0]; bcur[1] = bnxt[1];
    }
    if (key < Lk) {
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const int d = dt * 16 + g * 4;
        if (d < DK) {
          *reinterpret_cast<bf16x4*>(p.dv + ((long)b * Lk + key) * p.lddv + (long)h * DK + d) =
              bf16x4{(bf16_t)av[dt][0], (bf16_t)av[dt][1], (bf16_t)av[dt][2], (bf16_t)av[dt][3]};
          *reinterpret_cast<bf16x4*>(p.dkk + ((long)b * Lk + key) * p.lddk + (long)h * DK + d) =
              bf16x4{(bf16_t)ak[dt][0], (bf16_t)ak[dt][1], (bf16_t)ak[dt][2], (bf16_t)ak[dt][3]};
        }
      }
    }
  }
}

// dbias[h,q,k] += sum_b dS[b,h,q,k].  gridDim.y == 1: fixed order, bit-reproducible; gridDim.y > 1 (many slabs, e.g. all
// layers of a stack at once): each y reduces a contiguous chunk of slabs and adds its partial with one f32 atomic.
__global__ __launch_bounds__(256) void dbias_reduce_kernel(const bf16_t* __restrict__ ds, float* __restrict__ dbias, int B, int HLq, int Lk, int Lkp) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)HLq * Lk) return;
  const long row = idx / Lk;
  const int k = idx % Lk;
  const int per = (B + gridDim.y - 1) / gridDim.y;
  const int b0 = blockIdx.y * per, b1 = b0 + per < B ? b0 + per : B;
  float a = 0.f;
#pragma unroll 16
  for (int b = b0; b < b1; ++b) a += (float)ds[((long)b * HLq + row) * Lkp + k];  // independent loads: keep many in flight
  if (gridDim.y == 1) dbias[idx] += a;
  else if (b1 > b0) atomicAdd(dbias + idx, a);
}

// ================================================================================================
// Streaming ("flash") forms of the same three products, for sequences whose images do not fit one workgroup's LDS:
// Swin-V2 windows of 576 / 144 tokens (384 px / window 24, head dim 32) and the T5-large encoder's Le = 153 (head dim 64).
// Same swapped / unswapped MFMA tile constructions as above; what changes is the loop structure:
//   flash_fwd    : a workgroup owns 64 queries (one 16-query tile per wave, Q fragments in registers) and streams the keys
//                  through LDS in blocks of 64 (K row image + V transposed-read image) with an online softmax;
//   flash_bwd_dq : same ownership; per key block the K and V row images and the K transposed image; P from the forward's LSE,
//                  delta = dO . O from the rows already in registers;
//   flash_bwd_dkv: a workgroup owns 64 keys (K, V fragments in registers) and streams the queries (Q, dO row + transposed
//                  images, delta, LSE per block).
// Bias forms (template BIAS): 0 none; 1 dense [bias_mod | 1][H, Lq, Lk] (T5 relative-position bias, d bias through the dS
// scratch as above); 2 Swin-V2: looked up per score in the head's column of the (2w-1)^2 x H table held in LDS, plus the
// shifted-window mask from the tokens' coordinates -- index(i, j) = code(i) - code(j) + 2w(w-1), code(t) = y_t (2w-1) + x_t
// (HF/swinv2:480-490, 433-436); d(table) accumulated in LDS (ds_add_f32) and flushed with one global atomic per entry.
// ================================================================================================
struct FlashP {
  const bf16_t* q; long ldq; const bf16_t* k; long ldk; const bf16_t* v; long ldv;
  bf16_t* o; long ldo;        // forward: output rows (bt*Lq + i); backward: the forward's output (delta)
  bf16_t* otok; long ldot;    // forward, BIAS 2: also/instead written in TOKEN order (row = source token of (bt, i)); may be null
  float* lse;                 // [Bt, H, Lq]
  int Bt, H, Lq, Lk;
  const float* score_scale;   // [H] or null
  const float* bias; int bias_mod;
  const float* btab; float* dbtab; int w, R, shift, nW;  // R = side of the (padded) token grid the windows tile
  int Rreal;                                              // side of the real grid: positions beyond it are padding (no output)
  float* dbtab_part;  // [Bt * ceil(Lq/64), H, ntab]: every d-q workgroup stores its table partial here (plain stores; one
                      // reduction afterwards) -- flushing 4608 workgroups x 2209 entries with global atomics onto 8.8 k
                      // addresses cost more than the whole rest of the kernel
  float p; const uint32_t* seed; uint32_t tag;
  const bf16_t* dout; long lddo;
  bf16_t* dq; long lddq; bf16_t* dkk; long lddk; bf16_t* dv; long lddv;
  bf16_t* ds_ws;
};

__device__ __forceinline__ int flash_region(int s, int R, int w, int shift) { return (s >= R - w) + (s >= R - shift); }
// window-local index i of window `win` -> bias code and mask region (window-major sequence bt = b*nW + win)
__device__ __forceinline__ void flash_tok(const FlashP& p, int win, int i, int& code, int& reg) {
  const int w = p.w, nWr = p.R / w;
  const int iy = i / w, ix = i - iy * w;
  code = iy * (2 * w - 1) + ix;
  reg = p.shift > 0 ? flash_region((win / nWr) * w + iy, p.R, w, p.shift) * 3 + flash_region((win % nWr) * w + ix, p.R, w, p.shift) : 0;
}
__device__ __forceinline__ long flash_token_row(const FlashP& p, int bt, int i) {  // -1: a padded position
  const int w = p.w, nWr = p.R / w, win = bt % p.nW, b = bt / p.nW;
  const int ys = (win / nWr) * w + i / w, xs = (win % nWr) * w + i % w;
  const int y = (ys + p.shift) % p.R, x = (xs + p.shift) % p.R;
  return (y < p.Rreal && x < p.Rreal) ? ((long)b * p.Rreal + y) * p.Rreal + x : -1;
}

constexpr int FKB = 64;  // rows per streamed block

// rows [r0, r0 + 64) of sequence bt (zero beyond L) -> row image and / or transposed-read image, local row index
template <int DK, int DKP, bool ROWIMG, bool TRIMG>
__device__ __forceinline__ void stage_blk(const bf16_t* __restrict__ g, long ld, long seq_row0, int h, int r0, int L, char* rowimg, int pitchB,
                                          char* trimg) {
  constexpr int CPR = DKP / 8;
  for (int ch = threadIdx.x; ch < FKB * CPR; ch += 256) {
    const int r = ch / CPR, c = (ch % CPR) * 8;
    bf16x8 v = {};
    if (r0 + r < L && c < DK) v = *reinterpret_cast<const bf16x8*>(g + (seq_row0 + r0 + r) * ld + (long)h * DK + c);
    if (ROWIMG) *reinterpret_cast<bf16x8*>(rowimg + r * pitchB + c * 2) = v;
    if (TRIMG) *reinterpret_cast<bf16x8*>(trimg + TrImg<DKP>::off(r, c)) = v;
  }
}

template <int DK, int BIAS>
__global__ __launch_bounds__(256) void flash_fwd_kernel(FlashP p) {
  constexpr int KS = (DK + 31) / 32, DKP = KS * 32, DT = DK / 16, KPITCH = DKP * 2 + 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Kr = smem;
  char* Vt = Kr + FKB * KPITCH;
  int* kcode = reinterpret_cast<int*>(Vt + TrImg<DKP>::bytes(FKB));
  int* kreg = kcode + FKB;
  float* tab = reinterpret_cast<float*>(kreg + FKB);
  const int Lq = p.Lq, Lk = p.Lk;
  const int bt = blockIdx.y / p.H, h = blockIdx.y % p.H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
  const long qrow0 = (long)bt * Lq, krow0 = (long)bt * Lk;
  const int q0 = blockIdx.x * 64 + wave * 16;
  const bool active = q0 < Lq;
  const int q = q0 + (lane & 15);
  const int qc = q < Lq ? q : Lq - 1;
  const int win = BIAS == 2 ? bt % p.nW : 0;
  int qreg = 0, coff = 0;
  if constexpr (BIAS == 2) {
    const int ntab = (2 * p.w - 1) * (2 * p.w - 1);
    for (int t = threadIdx.x; t < ntab; t += 256) tab[t] = p.btab[(long)t * p.H + h];
    int qcode;
    flash_tok(p, win, qc, qcode, qreg);
    coff = qcode + 2 * p.w * (p.w - 1);
  }
  bf16x8 qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int c = ks * 32 + g * 8;
    bf16x8 v = {};
    if (c < DK) v = *reinterpret_cast<const bf16x8*>(p.q + (qrow0 + qc) * p.ldq + (long)h * DK + c);
    qf[ks] = v;
  }
  const float sscale = p.score_scale ? p.score_scale[h] : 1.f;
  const float* brow = nullptr;
  if constexpr (BIAS == 1) brow = p.bias + (p.bias_mod > 0 ? (long)(bt % p.bias_mod) * p.H * Lq * Lk : 0L) + ((long)h * Lq + qc) * Lk;
  const DropCtx dc = drop_slab(make_drop(p.seed, p.tag, p.p), (uint32_t)(bt * p.H + h));
  float m = -INFINITY, l = 0.f;
  f32x4 o[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int k0 = 0; k0 < Lk; k0 += FKB) {
    __syncthreads();
    stage_blk<DK, DKP, true, false>(p.k, p.ldk, krow0, h, k0, Lk, Kr, KPITCH, nullptr);
    stage_blk<DK, DKP, false, true>(p.v, p.ldv, krow0, h, k0, Lk, nullptr, 0, Vt);
    if constexpr (BIAS == 2) {
      if (threadIdx.x < FKB) {
        const int kk = k0 + threadIdx.x;
        int c = 0, r = -1;
        if (kk < Lk) flash_tok(p, win, kk, c, r);
        kcode[threadIdx.x] = c; kreg[threadIdx.x] = r;
      }
    }
    __syncthreads();
    if (!active) continue;
    f32x4 s[4];
    float mb = -INFINITY;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(Kr, KPITCH, t * 16, ks, lane), qf[ks], s[t], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kl = t * 16 + g * 4 + r, key = k0 + kl;
        // branch-free: every LDS read below is in range for any kl (kcode / kreg are filled for all 64 slots, code 0 for keys past
        // the end), the result of an out-of-range key is discarded by the select.  (Sixteen per-element branches split this loop
        // into a hundred basic blocks.)
        float x = s[t][r] * sscale;
        if constexpr (BIAS == 1) x += brow[key < Lk ? key : Lk - 1];
        if constexpr (BIAS == 2) x += tab[coff - kcode[kl]] + (kreg[kl] != qreg ? -200.f : 0.f);
        x = key < Lk ? x : -INFINITY;
        s[t][r] = x;
        mb = fmaxf(mb, x);
      }
    }
    mb = fmaxf(mb, __shfl_xor(mb, 16, 64));
    mb = fmaxf(mb, __shfl_xor(mb, 32, 64));
    const float mn = fmaxf(m, mb);
    const float corr = __expf(m - mn);
    float sum = 0.f;
    const uint32_t dbase = (uint32_t)qc * (uint32_t)Lk + (uint32_t)k0;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = __expf(s[t][r] - mn);
        sum += e;
        if constexpr (BIAS == 2) s[t][r] = e;  // (window attention has no probability dropout: HF/swinv2 attention_probs_dropout_prob = 0)
        else s[t][r] = e * drop_mult32(dc, dbase + t * 16 + g * 4 + r);
      }
    l = l * corr + sum;
    m = mn;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) { o[dt][0] *= corr; o[dt][1] *= corr; o[dt][2] *= corr; o[dt][3] *= corr; }
#pragma unroll
    for (int sidx = 0; sidx < 2; ++sidx) {
      const bf16x8 pf = pack8(s[2 * sidx], s[2 * sidx + 1]);
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag<DKP>(Vt, sidx * 32, dt * 16, lane), pf, o[dt], 0, 0, 0);
    }
  }
  if (!active) return;
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  const float inv = 1.f / l;
  if (q < Lq) {
    if (g == 0 && p.lse) p.lse[((long)bt * p.H + h) * Lq + q] = m + __logf(l);
    const long trow = (BIAS == 2 && p.otok) ? flash_token_row(p, bt, q) : 0;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const int d = dt * 16 + g * 4;
      const bf16x4 ov = bf16x4{(bf16_t)(o[dt][0] * inv), (bf16_t)(o[dt][1] * inv), (bf16_t)(o[dt][2] * inv), (bf16_t)(o[dt][3] * inv)};
      if (p.o) *reinterpret_cast<bf16x4*>(p.o + (qrow0 + q) * p.ldo + (long)h * DK + d) = ov;
      if (BIAS == 2 && p.otok && trow >= 0) *reinterpret_cast<bf16x4*>(p.otok + trow * p.ldot + (long)h * DK + d) = ov;
    }
  }
}

// ---- Swin-V2 forms, round 3 -------------------------------------------------------------------------------------------
// With head dim 32 a 16 x 16 score tile is two matrix instructions (S and PV) against, per lane, four scores' worth of
// vector work: these kernels are VALU-bound (131 TFLOP/s = 5 % of the MFMA peak in round 2's form), so the round-3 forms take
// vector instructions and exposed latency out of the per-score path:
//   * everything in the exp2 domain: the table is multiplied by log2(e) once when it is copied to LDS, the head's logit scale
//     too, a score is ONE fma (s * scale + table entry) and the softmax uses v_exp_f32 directly (no multiply in front);
//   * the table address is one subtraction: the keys' codes are stored as byte offsets (code * 4), the lane keeps the address of
//     its query's entry for code 0;
//   * the shifted-window mask is evaluated only in windows that HAVE more than one region (last window row / column of a shifted
//     block: 7 of 16 windows at stage 0, none in unshifted blocks) and the key-range select only in a partial last key block
//     (n = 576 has none) -- both are workgroup-uniform branches to separately compiled loop bodies;
//   * the next key block's K / V rows are loaded into registers before the current block is computed (the global-load latency
//     of every block was exposed between two barriers).
typedef __attribute__((ext_vector_type(4))) int i32x4;
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
constexpr float kLog2e = 1.4426950408889634f;

// does window `win` of a shifted block see more than one mask region?  (flash_region: only positions >= R - w differ from 0)
__device__ __forceinline__ bool flash_window_masked(const FlashP& p, int win) {
  const int nWr = p.R / p.w;
  return p.shift > 0 && (win / nWr == nWr - 1 || win % nWr == nWr - 1);
}

template <int DK>
__global__ __launch_bounds__(256) void flash_fwd_swin_kernel(FlashP p) {
  constexpr int KS = (DK + 31) / 32, DKP = KS * 32, DT = DK / 16, KPITCH = DKP * 2 + 16, CPR = DKP / 8, NCH = FKB * CPR / 256;
  static_assert(NCH >= 1 && FKB * CPR % 256 == 0, "one or more whole 16-byte chunks per thread and operand");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Kr = smem;
  char* Vt = Kr + FKB * KPITCH;
  int* kcode4 = reinterpret_cast<int*>(Vt + TrImg<DKP>::bytes(FKB));  // code * 4: a byte offset into the table
  int* kreg = kcode4 + FKB;
  float* tab = reinterpret_cast<float*>(kreg + FKB);
  const int Lq = p.Lq, Lk = p.Lk;
  const int bt = blockIdx.y / p.H, h = blockIdx.y % p.H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
  const long qrow0 = (long)bt * Lq, krow0 = (long)bt * Lk;
  const int q0 = blockIdx.x * 64 + wave * 16;
  const bool active = q0 < Lq;
  const int q = q0 + (lane & 15);
  const int qc = q < Lq ? q : Lq - 1;
  const int win = bt % p.nW;
  const int ntab = (2 * p.w - 1) * (2 * p.w - 1);
  for (int t = threadIdx.x; t < ntab; t += 256) tab[t] = p.btab[(long)t * p.H + h] * kLog2e;
  int qcode, qreg;
  flash_tok(p, win, qc, qcode, qreg);
  const char* cbase = reinterpret_cast<const char*>(tab) + (qcode + 2 * p.w * (p.w - 1)) * 4;
  const bool masked = flash_window_masked(p, win);
  bf16x8 qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int c = ks * 32 + g * 8;
    bf16x8 v = {};
    if (c < DK) v = *reinterpret_cast<const bf16x8*>(p.q + (qrow0 + qc) * p.ldq + (long)h * DK + c);
    qf[ks] = v;
  }
  const float sc2 = (p.score_scale ? p.score_scale[h] : 1.f) * kLog2e;

  // this thread's chunks of a streamed block: row ch / CPR, 8 elements from column (ch % CPR) * 8
  bf16x8 kv[NCH], vv[NCH];
  auto load_blk = [&](int k0) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int ch = threadIdx.x + i * 256, r = ch / CPR, c = (ch % CPR) * 8;
      bf16x8 a = {}, b = {};
      if (k0 + r < Lk && c < DK) {
        a = *reinterpret_cast<const bf16x8*>(p.k + (krow0 + k0 + r) * p.ldk + (long)h * DK + c);
        b = *reinterpret_cast<const bf16x8*>(p.v + (krow0 + k0 + r) * p.ldv + (long)h * DK + c);
      }
      kv[i] = a; vv[i] = b;
    }
  };
  auto store_blk = [&]() {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int ch = threadIdx.x + i * 256, r = ch / CPR, c = (ch % CPR) * 8;
      *reinterpret_cast<bf16x8*>(Kr + r * KPITCH + c * 2) = kv[i];
      *reinterpret_cast<bf16x8*>(Vt + TrImg<DKP>::off(r, c)) = vv[i];
    }
  };

  float m = -INFINITY, l = 0.f;  // running maximum (log2 domain) and sum
  f32x4 o[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto block = [&](int k0, auto MASKED, auto FULL) {
    f32x4 s[4];
    float mb = -INFINITY;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(Kr, KPITCH, t * 16, ks, lane), qf[ks], s[t], 0, 0, 0);
      const i32x4 kc = *reinterpret_cast<const i32x4*>(kcode4 + t * 16 + g * 4);
      i32x4 kr = {0, 0, 0, 0};
      if constexpr (MASKED.value) kr = *reinterpret_cast<const i32x4*>(kreg + t * 16 + g * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float x = fmaf(s[t][r], sc2, *reinterpret_cast<const float*>(cbase - kc[r]));
        if constexpr (MASKED.value) x += kr[r] != qreg ? -200.f * kLog2e : 0.f;
        if constexpr (!FULL.value) x = k0 + t * 16 + g * 4 + r < Lk ? x : -INFINITY;
        s[t][r] = x;
        mb = fmaxf(mb, x);
      }
    }
    mb = fmaxf(mb, __shfl_xor(mb, 16, 64));
    mb = fmaxf(mb, __shfl_xor(mb, 32, 64));
    const float mn = fmaxf(m, mb);
    const float corr = fast_exp2(m - mn);
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = fast_exp2(s[t][r] - mn);  // (window attention has no probability dropout: HF/swinv2 attention_probs_dropout_prob = 0)
        sum += e;
        s[t][r] = e;
      }
    l = l * corr + sum;
    m = mn;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) { o[dt][0] *= corr; o[dt][1] *= corr; o[dt][2] *= corr; o[dt][3] *= corr; }
#pragma unroll
    for (int sidx = 0; sidx < 2; ++sidx) {
      const bf16x8 pf = pack8(s[2 * sidx], s[2 * sidx + 1]);
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag<DKP>(Vt, sidx * 32, dt * 16, lane), pf, o[dt], 0, 0, 0);
    }
  };
  using T_ = std::integral_constant<bool, true>;
  using F_ = std::integral_constant<bool, false>;

  load_blk(0);
  for (int k0 = 0; k0 < Lk; k0 += FKB) {
    __syncthreads();  // every wave is past its reads of the previous block
    store_blk();
    if (threadIdx.x < FKB) {
      const int kk = k0 + threadIdx.x;
      int c = 0, r = -1;
      if (kk < Lk) flash_tok(p, win, kk, c, r);
      kcode4[threadIdx.x] = c * 4; kreg[threadIdx.x] = r;
    }
    __syncthreads();
    if (k0 + FKB < Lk) load_blk(k0 + FKB);  // in flight underneath this block's arithmetic
    if (!active) continue;
    const bool full = k0 + FKB <= Lk;
    if (masked) { if (full) block(k0, T_{}, T_{}); else block(k0, T_{}, F_{}); }
    else { if (full) block(k0, F_{}, T_{}); else block(k0, F_{}, F_{}); }
  }
  if (!active) return;
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  const float inv = 1.f / l;
  if (q < Lq) {
    if (g == 0 && p.lse) p.lse[((long)bt * p.H + h) * Lq + q] = (m + __log2f(l)) * (1.f / kLog2e);  // natural-log units, as every consumer expects
    const long trow = p.otok ? flash_token_row(p, bt, q) : 0;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const int d = dt * 16 + g * 4;
      const bf16x4 ov = bf16x4{(bf16_t)(o[dt][0] * inv), (bf16_t)(o[dt][1] * inv), (bf16_t)(o[dt][2] * inv), (bf16_t)(o[dt][3] * inv)};
      if (p.o) *reinterpret_cast<bf16x4*>(p.o + (qrow0 + q) * p.ldo + (long)h * DK + d) = ov;
      if (p.otok && trow >= 0) *reinterpret_cast<bf16x4*>(p.otok + trow * p.ldot + (long)h * DK + d) = ov;
    }
  }
}

template <int DK, int BIAS>
__global__ __launch_bounds__(256) void flash_bwd_dq_kernel(FlashP p) {
  constexpr int KS = (DK + 31) / 32, DKP = KS * 32, DT = DK / 16, KPITCH = DKP * 2 + 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Kr = smem;
  char* Vr = Kr + FKB * KPITCH;
  char* Kt = Vr + FKB * KPITCH;
  int* kcode = reinterpret_cast<int*>(Kt + TrImg<DKP>::bytes(FKB));
  int* kreg = kcode + FKB;
  float* tab = reinterpret_cast<float*>(kreg + FKB);
  const int ntab = BIAS == 2 ? (2 * p.w - 1) * (2 * p.w - 1) : 0;
  float* dtab = tab + ntab;
  const int Lq = p.Lq, Lk = p.Lk, Lkp = (Lk + 31) & ~31;
  const int bt = blockIdx.y / p.H, h = blockIdx.y % p.H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
  const long qrow0 = (long)bt * Lq, krow0 = (long)bt * Lk;
  const int q0 = blockIdx.x * 64 + wave * 16;
  const bool active = q0 < Lq;
  const int q = q0 + (lane & 15);
  const int qc = q < Lq ? q : Lq - 1;
  const int win = BIAS == 2 ? bt % p.nW : 0;
  int qreg = 0, coff = 0;
  if constexpr (BIAS == 2) {
    for (int t = threadIdx.x; t < ntab; t += 256) tab[t] = p.btab[(long)t * p.H + h];
    // d(table): one private copy per 16-lane group.  Within a group the 16 lanes are 16 consecutive queries against ONE key
    // offset: distinct table entries; across groups (keys 4 apart) entries i - j repeat -- a shared copy measured 4x slower.
    for (int t = threadIdx.x; t < 4 * ntab; t += 256) dtab[t] = 0.f;
    int qcode;
    flash_tok(p, win, qc, qcode, qreg);
    coff = qcode + 2 * p.w * (p.w - 1);
  }
  bf16x8 qf[KS], dof[KS];
  float delta = 0.f;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int c = ks * 32 + g * 8;
    bf16x8 vq = {}, vd = {}, vo = {};
    if (c < DK) {
      vq = *reinterpret_cast<const bf16x8*>(p.q + (qrow0 + qc) * p.ldq + (long)h * DK + c);
      vd = *reinterpret_cast<const bf16x8*>(p.dout + (qrow0 + qc) * p.lddo + (long)h * DK + c);
      vo = *reinterpret_cast<const bf16x8*>(p.o + (qrow0 + qc) * p.ldo + (long)h * DK + c);
    }
    qf[ks] = vq; dof[ks] = vd;
#pragma unroll
    for (int u = 0; u < 8; ++u) delta += (float)vd[u] * (float)vo[u];
  }
  delta += __shfl_xor(delta, 16, 64);
  delta += __shfl_xor(delta, 32, 64);
  const float lq = p.lse[((long)bt * p.H + h) * Lq + qc];
  const float sscale = p.score_scale ? p.score_scale[h] : 1.f;
  const bool want_dtab = (p.dbtab_part != nullptr || p.dbtab != nullptr) && p.ds_ws == nullptr;
  const float* brow = nullptr;
  if constexpr (BIAS == 1) brow = p.bias + (p.bias_mod > 0 ? (long)(bt % p.bias_mod) * p.H * Lq * Lk : 0L) + ((long)h * Lq + qc) * Lk;
  // dS rows for a second-pass reduction: T5 position-bias gradient (BIAS 1) and, for Swin (BIAS 2), the table gradient -- summing
  // dS per table entry with LDS float atomics cost 0.81 of 1.39 ms per stage-0 block (profiles/r02_swin_large_window_kernels_B8.txt)
  bf16_t* dsrow = (BIAS != 0 && p.ds_ws && q < Lq) ? p.ds_ws + (((long)bt * p.H + h) * Lq + q) * Lkp : nullptr;
  const DropCtx dc = drop_slab(make_drop(p.seed, p.tag, p.p), (uint32_t)(bt * p.H + h));
  f32x4 acc[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) acc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int k0 = 0; k0 < Lk; k0 += FKB) {
    __syncthreads();
    stage_blk<DK, DKP, true, true>(p.k, p.ldk, krow0, h, k0, Lk, Kr, KPITCH, Kt);
    stage_blk<DK, DKP, true, false>(p.v, p.ldv, krow0, h, k0, Lk, Vr, KPITCH, nullptr);
    if constexpr (BIAS == 2) {
      if (threadIdx.x < FKB) {
        const int kk = k0 + threadIdx.x;
        int c = 0, r = -1;
        if (kk < Lk) flash_tok(p, win, kk, c, r);
        kcode[threadIdx.x] = c; kreg[threadIdx.x] = r;
      }
    }
    __syncthreads();
    if (!active) continue;
#pragma unroll
    for (int sidx = 0; sidx < 2; ++sidx) {
      f32x4 ds2[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int t = 2 * sidx + u;
        f32x4 st = {0.f, 0.f, 0.f, 0.f}, dpt = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          st = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(Kr, KPITCH, t * 16, ks, lane), qf[ks], st, 0, 0, 0);
          dpt = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(Vr, KPITCH, t * 16, ks, lane), dof[ks], dpt, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int kl = t * 16 + g * 4 + r, key = k0 + kl;
          float dsv = 0.f;
          if (key < Lk && q < Lq) {
            float x = st[r] * sscale;
            if constexpr (BIAS == 1) x += brow[key];
            int ti = 0;
            if constexpr (BIAS == 2) { ti = coff - kcode[kl]; x += tab[ti]; if (kreg[kl] != qreg) x += -200.f; }
            const float pr = __expf(x - lq);
            dsv = pr * (dpt[r] * drop_mult32(dc, (uint32_t)q * (uint32_t)Lk + key) - delta);
            if constexpr (BIAS == 2) { if (want_dtab) atomicAdd(&dtab[g * ntab + ti], dsv); }
          }
          ds2[u][r] = dsv;
        }
      }
      const bf16x8 dsf = pack8(ds2[0], ds2[1]);
      if (dsrow) {
        const int kb = k0 + sidx * 32;
        if (kb + g * 4 < Lkp) *reinterpret_cast<bf16x4*>(dsrow + kb + g * 4) = bf16x4{dsf[0], dsf[1], dsf[2], dsf[3]};
        if (kb + 16 + g * 4 < Lkp) *reinterpret_cast<bf16x4*>(dsrow + kb + 16 + g * 4) = bf16x4{dsf[4], dsf[5], dsf[6], dsf[7]};
      }
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag<DKP>(Kt, sidx * 32, dt * 16, lane), dsf, acc[dt], 0, 0, 0);
    }
  }
  if (active && q < Lq) {
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const int d = dt * 16 + g * 4;
      *reinterpret_cast<bf16x4*>(p.dq + (qrow0 + q) * p.lddq + (long)h * DK + d) =
          bf16x4{(bf16_t)acc[dt][0], (bf16_t)acc[dt][1], (bf16_t)acc[dt][2], (bf16_t)acc[dt][3]};
    }
  }
  if constexpr (BIAS == 2) {
    if (p.ds_ws) {
      // the table gradient comes from the stored dS rows (swin_dtab_from_ds_kernel)
    } else if (p.dbtab_part) {
      __syncthreads();
      float* dst = p.dbtab_part + (((long)bt * gridDim.x + blockIdx.x) * p.H + h) * ntab;
      for (int t = threadIdx.x; t < ntab; t += 256) dst[t] = dtab[t] + dtab[ntab + t] + dtab[2 * ntab + t] + dtab[3 * ntab + t];
    } else if (p.dbtab) {
      __syncthreads();
      for (int t = threadIdx.x; t < ntab; t += 256) {
        const float v = dtab[t] + dtab[ntab + t] + dtab[2 * ntab + t] + dtab[3 * ntab + t];
        if (v != 0.f) atomicAdd(p.dbtab + (long)t * p.H + h, v);
      }
    }
  }
}

// dbtab[t, h] += sum over the partials [nparts, H, ntab] (fixed order: bit-reproducible)
__global__ __launch_bounds__(256) void dbtab_reduce_kernel(const float* __restrict__ part, long nparts, int H, int ntab, float* __restrict__ dbtab) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)H * ntab) return;
  const int h = (int)(idx / ntab), t = (int)(idx % ntab);
  float a = 0.f;
#pragma unroll 8
  for (long q = 0; q < nparts; ++q) a += part[(q * H + h) * ntab + t];
  dbtab[(long)t * H + h] += a;
}

// Swin d-q pass with the table gradient accumulated over SEQUENCES in registers.  The table entry of a (query, key) pair depends
// on their window-local positions only, not on the window or the image, so a workgroup that owns a query block walks `chunk`
// sequences (windows x images) and keeps the running sum of every dS element of its 16 x n strip per lane (NKB key blocks x 16
// values); the LDS float atomics that scatter dS into the table -- 0.81 of 1.39 ms per stage-0 block when done per sequence --
// run once per chunk.  grid = (query blocks, H * ceil(Bt / chunk)).
// Round 3: exp2 domain (table and logit scale carry log2(e): a score is one fma, P is one v_exp_f32 of a difference), table
// address = one subtraction (codes stored as byte offsets), the shift mask only in windows that have more than one region
// (uniform per sequence: two compiled forms of the key-block loop), the key / query range selects only in the instantiation for
// window sizes that are not a multiple of 64 (FULL = false: n = 144).
template <int DK, int NKB, bool FULL>
__global__ __launch_bounds__(256, 2) void flash_bwd_dq_swin_kernel(FlashP p, int chunk) {
  constexpr int KS = (DK + 31) / 32, DKP = KS * 32, DT = DK / 16, KPITCH = DKP * 2 + 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Kr = smem;
  char* Vr = Kr + FKB * KPITCH;
  char* Kt = Vr + FKB * KPITCH;
  int* kcode4 = reinterpret_cast<int*>(Kt + TrImg<DKP>::bytes(FKB));  // code * 4: a byte offset into the table
  int* kreg = kcode4 + FKB;
  float* tab = reinterpret_cast<float*>(kreg + FKB);
  const int ntab = (2 * p.w - 1) * (2 * p.w - 1);
  float* dtab = tab + ntab;
  const int Lq = p.Lq, Lk = p.Lk;
  const int h = blockIdx.y % p.H, btc = blockIdx.y / p.H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
  const int q0 = blockIdx.x * 64 + wave * 16;
  const bool active = q0 < Lq;
  const int q = q0 + (lane & 15);
  const int qc = q < Lq ? q : Lq - 1;
  for (int t = threadIdx.x; t < ntab; t += 256) tab[t] = p.btab[(long)t * p.H + h] * kLog2e;
  for (int t = threadIdx.x; t < 4 * ntab; t += 256) dtab[t] = 0.f;
  const int qiy = qc / p.w, qix = qc - qiy * p.w;
  const int coff = qiy * (2 * p.w - 1) + qix + 2 * p.w * (p.w - 1);
  const char* cbase = reinterpret_cast<const char*>(tab) + coff * 4;
  const float sc2 = (p.score_scale ? p.score_scale[h] : 1.f) * kLog2e;
  using T_ = std::integral_constant<bool, true>;
  using F_ = std::integral_constant<bool, false>;
  float dsacc[NKB][4][4];
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) dsacc[kb][t][r] = 0.f;

  const int bt1 = (btc + 1) * chunk < p.Bt ? (btc + 1) * chunk : p.Bt;
  for (int bt = btc * chunk; bt < bt1; ++bt) {
    const long qrow0 = (long)bt * Lq, krow0 = (long)bt * Lk;
    const int win = bt % p.nW;
    int qcode_unused, qreg;
    flash_tok(p, win, qc, qcode_unused, qreg);
    bf16x8 qf[KS], dof[KS];
    float delta = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int c = ks * 32 + g * 8;
      bf16x8 vq = {}, vd = {}, vo = {};
      if (c < DK) {
        vq = *reinterpret_cast<const bf16x8*>(p.q + (qrow0 + qc) * p.ldq + (long)h * DK + c);
        vd = *reinterpret_cast<const bf16x8*>(p.dout + (qrow0 + qc) * p.lddo + (long)h * DK + c);
        vo = *reinterpret_cast<const bf16x8*>(p.o + (qrow0 + qc) * p.ldo + (long)h * DK + c);
      }
      qf[ks] = vq; dof[ks] = vd;
#pragma unroll
      for (int u = 0; u < 8; ++u) delta += (float)vd[u] * (float)vo[u];
    }
    delta += __shfl_xor(delta, 16, 64);
    delta += __shfl_xor(delta, 32, 64);
    const float lq2 = p.lse[((long)bt * p.H + h) * Lq + qc] * kLog2e;
    f32x4 acc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) acc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto key_blocks = [&](auto MASKED) {
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      int k0 = kb * FKB;
      asm volatile("" : "+s"(k0));  // opaque: keeps the unrolled bodies' address arithmetic from being hoisted out of the sequence loop
      __syncthreads();
      stage_blk<DK, DKP, true, true>(p.k, p.ldk, krow0, h, k0, Lk, Kr, KPITCH, Kt);
      stage_blk<DK, DKP, true, false>(p.v, p.ldv, krow0, h, k0, Lk, Vr, KPITCH, nullptr);
      if (threadIdx.x < FKB) {
        const int kk = k0 + threadIdx.x;
        int c = 0, r = -1;
        if (kk < Lk) flash_tok(p, win, kk, c, r);
        kcode4[threadIdx.x] = c * 4; kreg[threadIdx.x] = r;
      }
      __syncthreads();
      if (!active) continue;
#pragma unroll
      for (int sidx = 0; sidx < 2; ++sidx) {
        f32x4 ds2[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int t = 2 * sidx + u;
          f32x4 st = {0.f, 0.f, 0.f, 0.f}, dpt = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            st = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(Kr, KPITCH, t * 16, ks, lane), qf[ks], st, 0, 0, 0);
            dpt = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(Vr, KPITCH, t * 16, ks, lane), dof[ks], dpt, 0, 0, 0);
          }
          const i32x4 kc = *reinterpret_cast<const i32x4*>(kcode4 + t * 16 + g * 4);
          i32x4 kr = {0, 0, 0, 0};
          if constexpr (MASKED.value) kr = *reinterpret_cast<const i32x4*>(kreg + t * 16 + g * 4);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float x = fmaf(st[r], sc2, *reinterpret_cast<const float*>(cbase - kc[r]));
            if constexpr (MASKED.value) x += kr[r] != qreg ? -200.f * kLog2e : 0.f;
            float dsv = fast_exp2(x - lq2) * (dpt[r] - delta);
            if constexpr (!FULL) dsv = (k0 + t * 16 + g * 4 + r < Lk && q < Lq) ? dsv : 0.f;
            ds2[u][r] = dsv;
            dsacc[kb][t][r] += dsv;
          }
        }
        const bf16x8 dsf = pack8(ds2[0], ds2[1]);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag<DKP>(Kt, sidx * 32, dt * 16, lane), dsf, acc[dt], 0, 0, 0);
      }
    }
    };
    if (NKB < 9 && !flash_window_masked(p, win)) key_blocks(F_{}); else key_blocks(T_{});  // (NKB = 9: 144 running sums per lane leave no room for a second copy of the loop)
    if (active && q < Lq) {
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const int d = dt * 16 + g * 4;
        *reinterpret_cast<bf16x4*>(p.dq + (qrow0 + q) * p.lddq + (long)h * DK + d) =
            bf16x4{(bf16_t)acc[dt][0], (bf16_t)acc[dt][1], (bf16_t)acc[dt][2], (bf16_t)acc[dt][3]};
      }
    }
  }
  // one scatter of the strip's accumulated dS into the (per 16-lane group) table copies, then the workgroup's partial table
  __syncthreads();
  if (active && q < Lq) {
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = kb * FKB + t * 16 + g * 4 + r;
          if (key < Lk) {
            const int ky = key / p.w, kx = key - ky * p.w;
            atomicAdd(&dtab[g * ntab + coff - (ky * (2 * p.w - 1) + kx)], dsacc[kb][t][r]);
          }
        }
  }
  __syncthreads();
  if (p.dbtab_part) {
    float* dst = p.dbtab_part + (((long)btc * gridDim.x + blockIdx.x) * p.H + h) * ntab;
    for (int t = threadIdx.x; t < ntab; t += 256) dst[t] = dtab[t] + dtab[ntab + t] + dtab[2 * ntab + t] + dtab[3 * ntab + t];
  }
}

template <int DK, int BIAS>
__global__ __launch_bounds__(256) void flash_bwd_dkv_kernel(FlashP p) {
  constexpr int KS = (DK + 31) / 32, DKP = KS * 32, DT = DK / 16, KPITCH = DKP * 2 + 16, CPR = DKP / 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Qr = smem;
  char* dOr = Qr + FKB * KPITCH;
  char* Qt = dOr + FKB * KPITCH;
  char* dOt = Qt + TrImg<DKP>::bytes(FKB);
  float* delta = reinterpret_cast<float*>(dOt + TrImg<DKP>::bytes(FKB));
  float* lses = delta + FKB;
  int* qcode = reinterpret_cast<int*>(lses + FKB);
  int* qreg = qcode + FKB;
  float* tab = reinterpret_cast<float*>(qreg + FKB);
  const int Lq = p.Lq, Lk = p.Lk;
  const int bt = blockIdx.y / p.H, h = blockIdx.y % p.H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
  const long qrow0 = (long)bt * Lq, krow0 = (long)bt * Lk;
  const int kt0 = blockIdx.x * 64 + wave * 16;
  const bool active = kt0 < Lk;
  const int key = kt0 + (lane & 15);
  const int kc = key < Lk ? key : Lk - 1;
  const int win = BIAS == 2 ? bt % p.nW : 0;
  int kreg_ = 0, koff = 0;
  if constexpr (BIAS == 2) {
    const int ntab = (2 * p.w - 1) * (2 * p.w - 1);
    for (int t = threadIdx.x; t < ntab; t += 256) tab[t] = p.btab[(long)t * p.H + h];
    int kcode_;
    flash_tok(p, win, kc, kcode_, kreg_);
    koff = 2 * p.w * (p.w - 1) - kcode_;
  }
  bf16x8 kf[KS], vf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int c = ks * 32 + g * 8;
    bf16x8 a = {}, b = {};
    if (c < DK) {
      a = *reinterpret_cast<const bf16x8*>(p.k + (krow0 + kc) * p.ldk + (long)h * DK + c);
      b = *reinterpret_cast<const bf16x8*>(p.v + (krow0 + kc) * p.ldv + (long)h * DK + c);
    }
    kf[ks] = a; vf[ks] = b;
  }
  const float sscale = p.score_scale ? p.score_scale[h] : 1.f;
  const float* biasp = nullptr;
  if constexpr (BIAS == 1) biasp = p.bias + (p.bias_mod > 0 ? (long)(bt % p.bias_mod) * p.H * Lq * Lk : 0L) + (long)h * Lq * Lk;
  const DropCtx dc = drop_slab(make_drop(p.seed, p.tag, p.p), (uint32_t)(bt * p.H + h));
  f32x4 av[DT], ak[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) { av[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; ak[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  for (int i0 = 0; i0 < Lq; i0 += FKB) {
    __syncthreads();
    // Q, dO of the block into row + transposed images; delta = dO . O and the LSE per row from the same pass
    for (int ch = threadIdx.x; ch < FKB * CPR; ch += 256) {
      const int r = ch / CPR, c = (ch % CPR) * 8;
      bf16x8 vq = {}, vd = {}, vo = {};
      const bool in = i0 + r < Lq && c < DK;
      if (in) {
        vq = *reinterpret_cast<const bf16x8*>(p.q + (qrow0 + i0 + r) * p.ldq + (long)h * DK + c);
        vd = *reinterpret_cast<const bf16x8*>(p.dout + (qrow0 + i0 + r) * p.lddo + (long)h * DK + c);
        vo = *reinterpret_cast<const bf16x8*>(p.o + (qrow0 + i0 + r) * p.ldo + (long)h * DK + c);
      }
      *reinterpret_cast<bf16x8*>(Qr + r * KPITCH + c * 2) = vq;
      *reinterpret_cast<bf16x8*>(dOr + r * KPITCH + c * 2) = vd;
      *reinterpret_cast<bf16x8*>(Qt + TrImg<DKP>::off(r, c)) = vq;
      *reinterpret_cast<bf16x8*>(dOt + TrImg<DKP>::off(r, c)) = vd;
      float a = 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) a += (float)vd[u] * (float)vo[u];
#pragma unroll
      for (int o2 = 1; o2 < CPR; o2 <<= 1) a += __shfl_xor(a, o2, 64);  // the CPR lanes of a row are neighbours
      if (c == 0) {
        delta[r] = a;
        lses[r] = i0 + r < Lq ? p.lse[((long)bt * p.H + h) * Lq + i0 + r] : INFINITY;  // padded queries: P = exp(-inf) = 0
        if constexpr (BIAS == 2) {
          int cc = 0, rr = -1;
          if (i0 + r < Lq) flash_tok(p, win, i0 + r, cc, rr);
          qcode[r] = cc; qreg[r] = rr;
        }
      }
    }
    __syncthreads();
    if (!active) continue;
#pragma unroll
    for (int sidx = 0; sidx < 2; ++sidx) {
      f32x4 pd2[2], ds2[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int qt = 2 * sidx + u;
        f32x4 st = {0.f, 0.f, 0.f, 0.f}, dpt = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          st = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(Qr, KPITCH, qt * 16, ks, lane), kf[ks], st, 0, 0, 0);
          dpt = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(dOr, KPITCH, qt * 16, ks, lane), vf[ks], dpt, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ql = qt * 16 + g * 4 + r, qq = i0 + ql;
          float pdv = 0.f, dsv = 0.f;
          if constexpr (BIAS == 2) {  // branch-free (no dropout in window attention); padded queries carry lse = +inf => P = 0
            const float x = st[r] * sscale + tab[qcode[ql] + koff] + (qreg[ql] != kreg_ ? -200.f : 0.f);
            const float pr = (key < Lk && qq < Lq) ? __expf(x - lses[ql]) : 0.f;
            pdv = pr;
            dsv = pr * (dpt[r] - delta[ql]);
          } else if (key < Lk && qq < Lq) {
            float x = st[r] * sscale;
            if constexpr (BIAS == 1) x += biasp[(long)qq * Lk + key];
            const float pr = __expf(x - lses[ql]);
            const float mlt = drop_mult32(dc, (uint32_t)qq * (uint32_t)Lk + key);
            pdv = pr * mlt;
            dsv = pr * (dpt[r] * mlt - delta[ql]);
          }
          pd2[u][r] = pdv; ds2[u][r] = dsv;
        }
      }
      const bf16x8 pdf = pack8(pd2[0], pd2[1]), dsf = pack8(ds2[0], ds2[1]);
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        av[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag<DKP>(dOt, sidx * 32, dt * 16, lane), pdf, av[dt], 0, 0, 0);
        ak[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag<DKP>(Qt, sidx * 32, dt * 16, lane), dsf, ak[dt], 0, 0, 0);
      }
    }
  }
  if (active && key < Lk) {
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const int d = dt * 16 + g * 4;
      *reinterpret_cast<bf16x4*>(p.dv + (krow0 + key) * p.lddv + (long)h * DK + d) =
          bf16x4{(bf16_t)av[dt][0], (bf16_t)av[dt][1], (bf16_t)av[dt][2], (bf16_t)av[dt][3]};
      *reinterpret_cast<bf16x4*>(p.dkk + (krow0 + key) * p.lddk + (long)h * DK + d) =
          bf16x4{(bf16_t)ak[dt][0], (bf16_t)ak[dt][1], (bf16_t)ak[dt][2], (bf16_t)ak[dt][3]};
    }
  }
}

// Swin d k / d v pass, round-3 form (see flash_fwd_swin_kernel): exp2 domain, table address = one addition on byte offsets, the
// shift mask only in windows with more than one region, no range selects (a padded query carries LSE = +inf, i.e. P = 0; a key
// past the end only feeds its own, never stored, output rows), and the next query block's Q / dO / O rows + LSE are loaded into
// registers underneath the current block's arithmetic.
template <int DK>
__global__ __launch_bounds__(256) void flash_bwd_dkv_swin_kernel(FlashP p) {
  constexpr int KS = (DK + 31) / 32, DKP = KS * 32, DT = DK / 16, KPITCH = DKP * 2 + 16, CPR = DKP / 8, NCH = FKB * CPR / 256;
  static_assert(NCH >= 1 && FKB * CPR % 256 == 0, "whole 16-byte chunks per thread");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Qr = smem;
  char* dOr = Qr + FKB * KPITCH;
  char* Qt = dOr + FKB * KPITCH;
  char* dOt = Qt + TrImg<DKP>::bytes(FKB);
  float* delta = reinterpret_cast<float*>(dOt + TrImg<DKP>::bytes(FKB));
  float* lse2 = delta + FKB;
  int* qcode4 = reinterpret_cast<int*>(lse2 + FKB);
  int* qreg = qcode4 + FKB;
  float* tab = reinterpret_cast<float*>(qreg + FKB);
  const int Lq = p.Lq, Lk = p.Lk;
  const int bt = blockIdx.y / p.H, h = blockIdx.y % p.H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
  const long qrow0 = (long)bt * Lq, krow0 = (long)bt * Lk;
  const int kt0 = blockIdx.x * 64 + wave * 16;
  const bool active = kt0 < Lk;
  const int key = kt0 + (lane & 15);
  const int kc = key < Lk ? key : Lk - 1;
  const int win = bt % p.nW;
  const int ntab = (2 * p.w - 1) * (2 * p.w - 1);
  for (int t = threadIdx.x; t < ntab; t += 256) tab[t] = p.btab[(long)t * p.H + h] * kLog2e;
  int kcode_, kreg_;
  flash_tok(p, win, kc, kcode_, kreg_);
  const char* kbase = reinterpret_cast<const char*>(tab) + (2 * p.w * (p.w - 1) - kcode_) * 4;
  const bool masked = flash_window_masked(p, win);
  bf16x8 kf[KS], vf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int c = ks * 32 + g * 8;
    bf16x8 a = {}, b = {};
    if (c < DK) {
      a = *reinterpret_cast<const bf16x8*>(p.k + (krow0 + kc) * p.ldk + (long)h * DK + c);
      b = *reinterpret_cast<const bf16x8*>(p.v + (krow0 + kc) * p.ldv + (long)h * DK + c);
    }
    kf[ks] = a; vf[ks] = b;
  }
  const float sc2 = (p.score_scale ? p.score_scale[h] : 1.f) * kLog2e;
  f32x4 av[DT], ak[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) { av[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; ak[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  bf16x8 vq[NCH], vd[NCH], vo[NCH];
  float lsev[NCH];
  auto load_blk = [&](int i0) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int ch = threadIdx.x + i * 256, r = ch / CPR, c = (ch % CPR) * 8;
      bf16x8 a = {}, b = {}, o = {};
      float ls = INFINITY;  // padded queries: P = exp2(-inf) = 0
      if (i0 + r < Lq) {
        if (c < DK) {
          a = *reinterpret_cast<const bf16x8*>(p.q + (qrow0 + i0 + r) * p.ldq + (long)h * DK + c);
          b = *reinterpret_cast<const bf16x8*>(p.dout + (qrow0 + i0 + r) * p.lddo + (long)h * DK + c);
          o = *reinterpret_cast<const bf16x8*>(p.o + (qrow0 + i0 + r) * p.ldo + (long)h * DK + c);
        }
        if (c == 0) ls = p.lse[((long)bt * p.H + h) * Lq + i0 + r] * kLog2e;
      }
      vq[i] = a; vd[i] = b; vo[i] = o; lsev[i] = ls;
    }
  };
  auto store_blk = [&](int i0) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int ch = threadIdx.x + i * 256, r = ch / CPR, c = (ch % CPR) * 8;
      *reinterpret_cast<bf16x8*>(Qr + r * KPITCH + c * 2) = vq[i];
      *reinterpret_cast<bf16x8*>(dOr + r * KPITCH + c * 2) = vd[i];
      *reinterpret_cast<bf16x8*>(Qt + TrImg<DKP>::off(r, c)) = vq[i];
      *reinterpret_cast<bf16x8*>(dOt + TrImg<DKP>::off(r, c)) = vd[i];
      float a = 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) a += (float)vd[i][u] * (float)vo[i][u];
#pragma unroll
      for (int o2 = 1; o2 < CPR; o2 <<= 1) a += __shfl_xor(a, o2, 64);  // the CPR lanes of a row are neighbours
      if (c == 0) {
        delta[r] = a;
        lse2[r] = lsev[i];
        int cc = 0, rr = -1;
        if (i0 + r < Lq) flash_tok(p, win, i0 + r, cc, rr);
        qcode4[r] = cc * 4; qreg[r] = rr;
      }
    }
  };
  auto block = [&](auto MASKED) {
#pragma unroll
    for (int sidx = 0; sidx < 2; ++sidx) {
      f32x4 pd2[2], ds2[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int qt = 2 * sidx + u;
        f32x4 st = {0.f, 0.f, 0.f, 0.f}, dpt = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          st = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(Qr, KPITCH, qt * 16, ks, lane), kf[ks], st, 0, 0, 0);
          dpt = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(dOr, KPITCH, qt * 16, ks, lane), vf[ks], dpt, 0, 0, 0);
        }
        const int ql0 = qt * 16 + g * 4;
        const i32x4 qc4 = *reinterpret_cast<const i32x4*>(qcode4 + ql0);
        const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse2 + ql0), d4 = *reinterpret_cast<const f32x4*>(delta + ql0);
        i32x4 qr4 = {0, 0, 0, 0};
        if constexpr (MASKED.value) qr4 = *reinterpret_cast<const i32x4*>(qreg + ql0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float x = fmaf(st[r], sc2, *reinterpret_cast<const float*>(kbase + qc4[r]));
          if constexpr (MASKED.value) x += qr4[r] != kreg_ ? -200.f * kLog2e : 0.f;
          const float pr = fast_exp2(x - l4[r]);
          pd2[u][r] = pr;
          ds2[u][r] = pr * (dpt[r] - d4[r]);
        }
      }
      const bf16x8 pdf = pack8(pd2[0], pd2[1]), dsf = pack8(ds2[0], ds2[1]);
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        av[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag<DKP>(dOt, sidx * 32, dt * 16, lane), pdf, av[dt], 0, 0, 0);
        ak[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag<DKP>(Qt, sidx * 32, dt * 16, lane), dsf, ak[dt], 0, 0, 0);
      }
    }
  };
  using T_ = std::integral_constant<bool, true>;
  using F_ = std::integral_constant<bool, false>;

  load_blk(0);
  for (int i0 = 0; i0 < Lq; i0 += FKB) {
    __syncthreads();
    store_blk(i0);
    __syncthreads();
    if (i0 + FKB < Lq) load_blk(i0 + FKB);
    if (!active) continue;
    if (masked) block(T_{}); else block(F_{});
  }
  if (active && key < Lk) {
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const int d = dt * 16 + g * 4;
      *reinterpret_cast<bf16x4*>(p.dv + (krow0 + key) * p.lddv + (long)h * DK + d) =
          bf16x4{(bf16_t)av[dt][0], (bf16_t)av[dt][1], (bf16_t)av[dt][2], (bf16_t)av[dt][3]};
      *reinterpret_cast<bf16x4*>(p.dkk + (krow0 + key) * p.lddk + (long)h * DK + d) =
          bf16x4{(bf16_t)ak[dt][0], (bf16_t)ak[dt][1], (bf16_t)ak[dt][2], (bf16_t)ak[dt][3]};
    }
  }
}

template <int DK, int BIAS>
static int flash_launch(const FlashP& p, int which, hipStream_t s) {
  constexpr int KS = (DK + 31) / 32, DKP = KS * 32, KPITCH = DKP * 2 + 16;
  const size_t ntab = BIAS == 2 ? (size_t)(2 * p.w - 1) * (2 * p.w - 1) : 0;
  const size_t tr = TrImg<DKP>::bytes(FKB);
  int rc;
  // KLAB_SWIN_FLASH_V2=0: round 2's Swin kernels (the generic streaming forms with the table look-up added)
  static const bool v2 = [] { const char* e = getenv("KLAB_SWIN_FLASH_V2"); return !e || atoi(e) != 0; }();
  if (which == 0) {
    const size_t lds = (size_t)FKB * KPITCH + tr + 2 * FKB * 4 + ntab * 4;
    if constexpr (BIAS == 2) {
      if (v2) {
        rc = ensure_dyn_lds(reinterpret_cast<const void*>(flash_fwd_swin_kernel<DK>), lds); if (rc) return rc;
        hipLaunchKernelGGL((flash_fwd_swin_kernel<DK>), dim3((p.Lq + 63) / 64, p.Bt * p.H), dim3(256), lds, s, p);
        KLAB_LAUNCH_CHECK();
        return KLAB_OK;
      }
    }
    rc = ensure_dyn_lds(reinterpret_cast<const void*>(flash_fwd_kernel<DK, BIAS>), lds); if (rc) return rc;
    hipLaunchKernelGGL((flash_fwd_kernel<DK, BIAS>), dim3((p.Lq + 63) / 64, p.Bt * p.H), dim3(256), lds, s, p);
  } else if (which == 1) {
    const size_t lds = 2 * (size_t)FKB * KPITCH + tr + 2 * FKB * 4 + 5 * ntab * 4;
    rc = ensure_dyn_lds(reinterpret_cast<const void*>(flash_bwd_dq_kernel<DK, BIAS>), lds); if (rc) return rc;
    hipLaunchKernelGGL((flash_bwd_dq_kernel<DK, BIAS>), dim3((p.Lq + 63) / 64, p.Bt * p.H), dim3(256), lds, s, p);
  } else {
    const size_t lds = 2 * (size_t)FKB * KPITCH + 2 * tr + 4 * FKB * 4 + ntab * 4;
    if constexpr (BIAS == 2) {
      if (v2) {
        rc = ensure_dyn_lds(reinterpret_cast<const void*>(flash_bwd_dkv_swin_kernel<DK>), lds); if (rc) return rc;
        hipLaunchKernelGGL((flash_bwd_dkv_swin_kernel<DK>), dim3((p.Lk + 63) / 64, p.Bt * p.H), dim3(256), lds, s, p);
        KLAB_LAUNCH_CHECK();
        return KLAB_OK;
      }
    }
    rc = ensure_dyn_lds(reinterpret_cast<const void*>(flash_bwd_dkv_kernel<DK, BIAS>), lds); if (rc) return rc;
    hipLaunchKernelGGL((flash_bwd_dkv_kernel<DK, BIAS>), dim3((p.Lk + 63) / 64, p.Bt * p.H), dim3(256), lds, s, p);
  }
  KLAB_LAUNCH_CHECK();
  return KLAB_OK;
}

// which: 0 forward, 1 d q (+ d bias), 2 d k / d v.  dk in {32, 64}; bias_form 0 / 1 / 2
int flash_attn_dispatch(const FlashP& p, int dk, int bias_form, int which, hipStream_t s) {
#define FL(D, BF) return flash_launch<D, BF>(p, which, s)
  if (dk == 32) { if (bias_form == 0) FL(32, 0); if (bias_form == 1) FL(32, 1); if (bias_form == 2) FL(32, 2); }
  if (dk == 64) { if (bias_form == 0) FL(64, 0); if (bias_form == 1) FL(64, 1); if (bias_form == 2) FL(64, 2); }
#undef FL
  return KLAB_ERR_UNSUPPORTED;
}

template <typename K>
static int set_lds_attr(K kern, size_t bytes) { return ensure_dyn_lds(reinterpret_cast<const void*>(kern), bytes); }

template <int DK>
static int launch_fwd(const AttnMP& p, hipStream_t s) {
  constexpr int KS = (DK + 31) / 32, DKP = KS * 32;
  const int Lkp = (p.Lk + 31) & ~31;
  const size_t lds = (size_t)Lkp * (DKP * 2 + 16) + TrImg<DKP>::bytes(Lkp);
  dim3 grid((p.Lq + 63) / 64, p.B * p.H);
  int rc;
#define FWD_LAUNCH(MT)                                                       \
  rc = set_lds_attr(t5_attn_fwd_mfma<DK, MT>, lds); if (rc) return rc;      \
  hipLaunchKernelGGL((t5_attn_fwd_mfma<DK, MT>), grid, dim3(256), lds, s, p)
  if (Lkp <= 64) { FWD_LAUNCH(4); }
  else if (Lkp <= 128) { FWD_LAUNCH(8); }
  else if (Lkp <= 256) { FWD_LAUNCH(16); }
  else return KLAB_ERR_UNSUPPORTED;
#undef FWD_LAUNCH
  KLAB_LAUNCH_CHECK();
  return KLAB_OK;
}

template <int DK>
static int launch_bwd(const AttnMP& p, hipStream_t s) {
  constexpr int KS = (DK + 31) / 32, DKP = KS * 32;
  const int Lqp = (p.Lq + 31) & ~31, Lkp = (p.Lk + 31) & ~31;
  const size_t pitch = DKP * 2 + 16;
  const size_t lds = (size_t)(2 * Lqp + 2 * Lkp) * pitch + 2 * TrImg<DKP>::bytes(Lqp) + TrImg<DKP>::bytes(Lkp) + 2 * (size_t)Lqp * 4;
  int rc = set_lds_attr(t5_attn_bwd_mfma<DK>, lds);
  if (rc) return rc;
  hipLaunchKernelGGL((t5_attn_bwd_mfma<DK>), dim3(p.B * p.H), dim3(256), lds, s, p);
  KLAB_LAUNCH_CHECK();
  if (p.dbias && p.ds_ws && !p.defer_reduce) {
    const long tot = (long)p.H * p.Lq * p.Lk;
    hipLaunchKernelGGL(dbias_reduce_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, p.ds_ws, p.dbias, p.B, p.H * p.Lq, p.Lk, Lkp);
    KLAB_LAUNCH_CHECK();
  }
  return KLAB_OK;
}

static AttnMP to_mp(const klab_attn_args* a) {
  AttnMP p;
  p.q = (const bf16_t*)a->q; p.ldq = a->ldq; p.k = (const bf16_t*)a->k; p.ldk = a->ldk; p.v = (const bf16_t*)a->v; p.ldv = a->ldv;
  p.bias = a->bias; p.causal = a->causal; p.ctx = (bf16_t*)a->ctx; p.ldo = a->ldo; p.lse = a->lse;
  p.B = a->B; p.H = a->H; p.Lq = a->Lq; p.Lk = a->Lk; p.p = a->drop_p; p.seed = a->seed_dev; p.tag = a->drop_tag;
  p.dctx = (const bf16_t*)a->dctx; p.lddo = a->lddo; p.dq = (bf16_t*)a->dq; p.lddq = a->lddq; p.dkk = (bf16_t*)a->dk_out; p.lddk = a->lddk;
  p.dv = (bf16_t*)a->dv; p.lddv = a->lddv; p.dbias = a->dbias; p.ds_ws = (bf16_t*)a->ds_ws;
  p.defer_reduce = a->ds_defer;
  p.score_scale = a->score_scale; p.bias_mod = a->bias_mod;
  return p;
}

static int t5_flash_fallback_fwd(const AttnMP& p, int dk, hipStream_t s);
// returns KLAB_ERR_UNSUPPORTED when the shape is outside this kernel's envelope (caller falls back)
int t5_attn_fwd_mfma_dispatch(const klab_attn_args* a, hipStream_t s) {
  if (a->dtype != KLAB_BF16 || (a->ldo & 3) || (a->ldq & 7) || (a->ldk & 7) || (a->ldv & 7)) return KLAB_ERR_UNSUPPORTED;
  AttnMP p = to_mp(a);
  int rc = KLAB_ERR_UNSUPPORTED;
  switch (a->dk) {
    case 16: rc = launch_fwd<16>(p, s); break;
    case 32: rc = launch_fwd<32>(p, s); break;
    case 64: rc = launch_fwd<64>(p, s); break;
    case 128: rc = launch_fwd<128>(p, s); break;
  }
  if (rc == KLAB_ERR_UNSUPPORTED) rc = t5_flash_fallback_fwd(p, a->dk, s);
  return rc;
}
static FlashP to_flash(const AttnMP& p) {
  FlashP f;
  memset(&f, 0, sizeof(f));
  f.q = p.q; f.ldq = p.ldq; f.k = p.k; f.ldk = p.ldk; f.v = p.v; f.ldv = p.ldv; f.o = p.ctx; f.ldo = p.ldo; f.lse = p.lse;
  f.Bt = p.B; f.H = p.H; f.Lq = p.Lq; f.Lk = p.Lk; f.score_scale = p.score_scale; f.bias = p.bias; f.bias_mod = p.bias_mod;
  f.p = p.p; f.seed = p.seed; f.tag = p.tag; f.dout = p.dctx; f.lddo = p.lddo;
  f.dq = p.dq; f.lddq = p.lddq; f.dkk = p.dkk; f.lddk = p.lddk; f.dv = p.dv; f.lddv = p.lddv; f.ds_ws = p.ds_ws;
  return f;
}
// sequences too long for the one-workgroup kernels (T5-large encoder, Le = 153 at head dim 64): the streaming forms
static int t5_flash_fallback(const AttnMP& p, int dk, bool backward, hipStream_t s) {
  static const bool on = [] { const char* e = getenv("KLAB_ATTN_FLASH"); return !e || atoi(e) != 0; }();
  if (!on || p.causal || (dk != 32 && dk != 64)) return KLAB_ERR_UNSUPPORTED;
  if (backward && p.dbias && !p.ds_ws) return KLAB_ERR_UNSUPPORTED;  // d bias only through the dS scratch
  const FlashP f = to_flash(p);
  const int bf = p.bias ? 1 : 0;
  if (!backward) return flash_attn_dispatch(f, dk, bf, 0, s);
  int rc = flash_attn_dispatch(f, dk, bf, 1, s);
  if (rc) return rc;
  rc = flash_attn_dispatch(f, dk, bf, 2, s);
  if (rc) return rc;
  if (p.dbias && p.ds_ws && !p.defer_reduce) {
    const int Lkp = (p.Lk + 31) & ~31;
    const long tot = (long)p.H * p.Lq * p.Lk;
    hipLaunchKernelGGL(dbias_reduce_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, p.ds_ws, p.dbias, p.B, p.H * p.Lq, p.Lk, Lkp);
    KLAB_LAUNCH_CHECK();
  }
  return KLAB_OK;
}

// fused front half of an attention sub-layer (see t5_attn_fused_fwd); KLAB_ERR_UNSUPPORTED outside its envelope
int t5_attn_fused_fwd_dispatch(const klab_attn_fused_args* fa, hipStream_t s) {
  const klab_attn_args* a = &fa->attn;
  if (a->dtype != KLAB_BF16 || fa->d_model != AF_D || a->dk != AF_DK || a->Lq < 1 || a->Lq > 64 || a->Lk < 1 || a->Lk > 64) return KLAB_ERR_UNSUPPORTED;
  if (!fa->cross && a->Lk != a->Lq) return KLAB_ERR_UNSUPPORTED;
  if ((a->ldo & 3) || (fa->ldproj & 7) || ((uintptr_t)fa->proj & 15) || (fa->cross && ((a->ldk & 7) || (a->ldv & 7)))) return KLAB_ERR_UNSUPPORTED;
  if (a->score_scale || a->bias_mod) return KLAB_ERR_UNSUPPORTED;
  AttnFusedP f;
  f.x = fa->x; f.gamma = fa->gamma; f.eps = fa->eps; f.w = (const bf16_t*)fa->w; f.xn = (bf16_t*)fa->xn; f.rstd = fa->rstd;
  f.proj = (bf16_t*)fa->proj; f.ldproj = fa->ldproj;
  f.a = to_mp(a);
  static const int abl = [] { const char* e = getenv("KLAB_AF_ABLATE"); return e ? atoi(e) : 0; }();
  f.ablate = abl;
  const size_t images = 3 * (size_t)64 * (AF_DK * 2 + 16) + TrImg<AF_DK>::bytes(64);
  const size_t ring = (size_t)AF_S * (fa->cross ? 64 : 192) * 128;
  const size_t strips = fa->cross ? ring + 16384 : 0;  // (cross: the norm prologue's strips sit behind the small ring)
  size_t lds = ring > images ? ring : images;
  if (strips > lds) lds = strips;
  int rc;
  if (fa->cross) {
    rc = set_lds_attr(t5_attn_fused_fwd<true>, lds); if (rc) return rc;
    hipLaunchKernelGGL(t5_attn_fused_fwd<true>, dim3(a->B * a->H), dim3(256), lds, s, f);
  } else {
    rc = set_lds_attr(t5_attn_fused_fwd<false>, lds); if (rc) return rc;
    hipLaunchKernelGGL(t5_attn_fused_fwd<false>, dim3(a->B * a->H), dim3(256), lds, s, f);
  }
  KLAB_LAUNCH_CHECK();
  return KLAB_OK;
}

int t5_attn_bwd_mfma_dispatch(const klab_attn_args* a, hipStream_t s) {
  if (a->dtype != KLAB_BF16 || (a->ldo & 7) || (a->ldq & 7) || (a->ldk & 7) || (a->ldv & 7) || (a->lddo & 7) || (a->lddq & 3) ||
      (a->lddk & 3) || (a->lddv & 3))
    return KLAB_ERR_UNSUPPORTED;
  AttnMP p = to_mp(a);
  int rc = KLAB_ERR_UNSUPPORTED;
  switch (a->dk) {
    case 16: rc = launch_bwd<16>(p, s); break;
    case 32: rc = launch_bwd<32>(p, s); break;
    case 64: rc = launch_bwd<64>(p, s); break;
    case 128: rc = launch_bwd<128>(p, s); break;
  }
  if (rc == KLAB_ERR_UNSUPPORTED) rc = t5_flash_fallback(p, a->dk, true, s);
  return rc;
}


static int t5_flash_fallback_fwd(const AttnMP& p, int dk, hipStream_t s) { return t5_flash_fallback(p, dk, false, s); }

// Swin-V2 large windows on window-major copies (attn_swin.hip): see FlashP
// d(table)[c, h] += sum over sequences bt and query tokens i of dS[bt, h, i, j(i, c)]: table entry c = (dy, dx) pairs query
// (iy, ix) with key (iy - dy, ix - dx) (HF/swinv2:480-490).  One thread per table entry, consecutive entries = consecutive dx =
// consecutive keys of one dS row (coalesced 2-byte reads); blockIdx.z splits the sequences.
__global__ __launch_bounds__(256) void swin_dtab_from_ds_kernel(const bf16_t* __restrict__ ds, float* __restrict__ dbtab, int Bt, int H, int w,
                                                                int Lkp) {
  const int tw = 2 * w - 1, ntab = tw * tw, n = w * w;
  const int c = blockIdx.x * 256 + threadIdx.x, h = blockIdx.y;
  if (c >= ntab) return;
  const int dy = c / tw - (w - 1), dx = c % tw - (w - 1);
  const int iy0 = dy > 0 ? dy : 0, iy1 = dy < 0 ? w + dy : w;  // query rows whose partner row iy - dy lies inside the window
  const int ix0 = dx > 0 ? dx : 0, ix1 = dx < 0 ? w + dx : w;
  const int per = (Bt + gridDim.z - 1) / gridDim.z;
  const int b0 = blockIdx.z * per, b1 = b0 + per < Bt ? b0 + per : Bt;
  float a = 0.f;
  for (int bt = b0; bt < b1; ++bt) {
    const bf16_t* base = ds + ((long)bt * H + h) * n * Lkp;
    for (int iy = iy0; iy < iy1; ++iy) {
      const bf16_t* row = base + (long)(iy * w) * Lkp + (iy - dy) * w - dx;  // + ix * Lkp + ix
#pragma unroll 4
      for (int ix = ix0; ix < ix1; ++ix) a += (float)row[(long)ix * Lkp + ix];
    }
  }
  if (a != 0.f) atomicAdd(dbtab + (long)c * H + h, a);
}

int swin_flash_dispatch(const void* g, long ldg, int C, void* otok, long ldot, float* lse, int Bt, int H, int n, const float* scale,
                        const float* btab, float* dbtab, float* dbtab_part, int w, int Rp, int Rreal, int shift, int nW, const void* ow,
                        const void* dow, void* dg, int which, hipStream_t s) {
  FlashP f;
  memset(&f, 0, sizeof(f));
  const bf16_t* gb = (const bf16_t*)g;
  f.q = gb; f.k = gb + C; f.v = gb + 2 * C; f.ldq = f.ldk = f.ldv = ldg;
  f.o = (bf16_t*)ow; f.ldo = C; f.otok = (bf16_t*)otok; f.ldot = ldot; f.lse = lse;
  f.Bt = Bt; f.H = H; f.Lq = f.Lk = n; f.score_scale = scale; f.btab = btab; f.dbtab = dbtab; f.w = w; f.R = Rp; f.Rreal = Rreal; f.shift = shift; f.nW = nW;
  f.dout = (const bf16_t*)dow; f.lddo = C;
  bf16_t* dgb = (bf16_t*)dg;
  f.dq = dgb; f.dkk = dgb ? dgb + C : nullptr; f.dv = dgb ? dgb + 2 * C : nullptr; f.lddq = f.lddk = f.lddv = ldg;
  // d(table): `dbtab_part` is either the per-workgroup partial-table scratch (LDS-atomics form) or, with ds_mode, the dS
  // scratch [Bt, H, n, roundup32(n)] bf16 of the store-and-reduce form
  // measured (configs[4] stages at B = 8, tools/swin_attn_bench.py): store-and-reduce 1296 / 791 / 644 / 168 us per block backward
  // against 1389 / 720 / 413 / 80 us for the LDS-atomics form (0.57 / 0.33 / 0.20 / 0.05 ms without any table gradient): the
  // one-thread-per-entry reduction of 2-byte dS elements costs what the atomics cost.  Default: atomics; KLAB_SWIN_DTAB_DS=1 selects
  // the other form.
  static const bool ds_mode = [] { const char* e = getenv("KLAB_SWIN_DTAB_DS"); return e && atoi(e) != 0; }();
  const int ntab = (2 * w - 1) * (2 * w - 1);
  const bool want = which == 1 && dbtab && dbtab_part;
  if (want && ds_mode) f.ds_ws = (bf16_t*)dbtab_part;
  else f.dbtab_part = want ? dbtab_part : nullptr;
  static const bool seq_acc = [] { const char* e = getenv("KLAB_SWIN_DTAB_SEQACC"); return !e || atoi(e) != 0; }();
  const int nkb = (n + FKB - 1) / FKB;
  if (want && !ds_mode && seq_acc && (nkb == 1 || nkb == 2 || nkb == 3 || nkb == 9)) {
    // table gradient accumulated over sequences in registers: `chunk` sequences per workgroup, about 512 workgroups or more
    const int nqb = (n + 63) / 64;
    // chunk: the scatter costs about 1.5 sequences' worth of streaming (1390 vs 570 us per stage-0 block when done per sequence);
    // pick the chunk that minimises rounds-of-resident-workgroups x (chunk + 1.5), two workgroups resident per CU
    static const int slots = [] { const char* e = getenv("KLAB_SWIN_DTAB_WGS"); return e && atoi(e) > 0 ? atoi(e) : 512; }();
    int chunk = 1;
    double best = 1e30;
    for (int c = 1; c <= 16 && c <= Bt; ++c) {
      const long nwg = (long)nqb * H * ((Bt + c - 1) / c);
      const double cost = (double)((nwg + slots - 1) / slots) * (c + 1.5);
      if (cost < best - 1e-9) { best = cost; chunk = c; }
    }
    const int nchunks = (Bt + chunk - 1) / chunk;
    const size_t lds = 2 * (size_t)FKB * 80 + TrImg<32>::bytes(FKB) + 2 * FKB * 4 + 5 * (size_t)ntab * 4;
    int rc2 = KLAB_OK;
#define SEQ_LAUNCH(NKB)                                                                                                  \
    if (n % FKB == 0) {                                                                                                     \
      rc2 = ensure_dyn_lds(reinterpret_cast<const void*>(flash_bwd_dq_swin_kernel<32, NKB, true>), lds);                    \
      if (rc2) return rc2;                                                                                                  \
      hipLaunchKernelGGL((flash_bwd_dq_swin_kernel<32, NKB, true>), dim3(nqb, H * nchunks), dim3(256), lds, s, f, chunk);   \
    } else {                                                                                                                \
      rc2 = ensure_dyn_lds(reinterpret_cast<const void*>(flash_bwd_dq_swin_kernel<32, NKB, false>), lds);                   \
      if (rc2) return rc2;                                                                                                  \
      hipLaunchKernelGGL((flash_bwd_dq_swin_kernel<32, NKB, false>), dim3(nqb, H * nchunks), dim3(256), lds, s, f, chunk);  \
    }
    if (nkb == 1) { SEQ_LAUNCH(1); } else if (nkb == 2) { SEQ_LAUNCH(2); } else if (nkb == 3) { SEQ_LAUNCH(3); } else { SEQ_LAUNCH(9); }
#undef SEQ_LAUNCH
    KLAB_LAUNCH_CHECK();
    hipLaunchKernelGGL(dbtab_reduce_kernel, dim3((unsigned)(((long)H * ntab + 255) / 256)), dim3(256), 0, s, f.dbtab_part,
                       (long)nchunks * nqb, H, ntab, dbtab);
    KLAB_LAUNCH_CHECK();
    return KLAB_OK;
  }
  const int rc = flash_attn_dispatch(f, 32, 2, which, s);
  if (rc || !want) return rc;
  if (ds_mode) {
    int gz = Bt / 8;
    gz = gz < 1 ? 1 : (gz > 64 ? 64 : gz);
    hipLaunchKernelGGL(swin_dtab_from_ds_kernel, dim3((unsigned)((ntab + 255) / 256), (unsigned)H, (unsigned)gz), dim3(256), 0, s,
                       (const bf16_t*)dbtab_part, dbtab, Bt, H, w, (n + 31) & ~31);
  } else {
    hipLaunchKernelGGL(dbtab_reduce_kernel, dim3((unsigned)(((long)H * ntab + 255) / 256)), dim3(256), 0, s, f.dbtab_part,
                       (long)Bt * ((n + 63) / 64), H, ntab, dbtab);
  }
  KLAB_LAUNCH_CHECK();
  return KLAB_OK;
}

int dbias_reduce_dispatch(const void* ds_ws, float* dbias, int nbatch, int H, int Lq, int Lk, hipStream_t s) {
  const int Lkp = (Lk + 31) & ~31;
  const long tot = (long)H * Lq * Lk;
  int gy = nbatch / 16;  // >= 16 slabs per chunk
  gy = gy < 1 ? 1 : (gy > 32 ? 32 : gy);
  if (nbatch <= 64) gy = 1;
  hipLaunchKernelGGL(dbias_reduce_kernel, dim3((unsigned)((tot + 255) / 256), gy), dim3(256), 0, s, (const bf16_t*)ds_ws, dbias, nbatch, H * Lq, Lk, Lkp);
  KLAB_LAUNCH_CHECK();
  return KLAB_OK;
}

}  // namespace klab
