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
  static size_t bytes(int rows) { return (size_t)((rows + 7) / 8) * GROUPB; }
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
template <int DK, int MAXT>
__global__ __launch_bounds__(256) void t5_attn_fwd_mfma(AttnMP p) {
  constexpr int KS = (DK + 31) / 32, DKP = KS * 32, DT = (DK + 15) / 16;
  constexpr int KPITCH = DKP * 2 + 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int Lq = p.Lq, Lk = p.Lk;
  const int Lkp = (Lk + 31) & ~31, NT = Lkp / 16;
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
        float x = s[t][r];
        if (key < Lk && !(p.causal && key > q)) { if (brow) x += brow[key]; }
        else x = -INFINITY;
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
      for (int r = 0; r < 4; ++r) s[t][r] *= inv * drop_mult32(dc, base + t * 16 + g * 4 + r);
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
          float dsv = 0.f;
          if (key < Lk && q < Lq && !(p.causal && key > q)) {
            const float x = st[r] * sscale + bcur[u][r];
            const float pr = __expf(x - lq);
            dsv = pr * (dpt[r] * drop_mult32(dc, (uint32_t)q * (uint32_t)Lk + key) - dq_);
            if (dbrow) atomicAdd(dbrow + key, dsv);
          }
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
          float pdv = 0.f, dsv = 0.f;
          if (key < Lk && q < Lq && !(p.causal && key > q)) {
            const float x = st[r] * sscale + bcur[u][r];
            const float pr = __expf(x - lses[q]);
            const float mlt = drop_mult32(dc, (uint32_t)q * (uint32_t)Lk + key);
            pdv = pr * mlt;
            dsv = pr * (dpt[r] * mlt - delta[q]);
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

// returns KLAB_ERR_UNSUPPORTED when the shape is outside this kernel's envelope (caller falls back)
int t5_attn_fwd_mfma_dispatch(const klab_attn_args* a, hipStream_t s) {
  if (a->dtype != KLAB_BF16 || (a->ldo & 3) || (a->ldq & 7) || (a->ldk & 7) || (a->ldv & 7)) return KLAB_ERR_UNSUPPORTED;
  AttnMP p = to_mp(a);
  switch (a->dk) {
    case 16: return launch_fwd<16>(p, s);
    case 32: return launch_fwd<32>(p, s);
    case 64: return launch_fwd<64>(p, s);
    case 128: return launch_fwd<128>(p, s);
  }
  return KLAB_ERR_UNSUPPORTED;
}
int t5_attn_bwd_mfma_dispatch(const klab_attn_args* a, hipStream_t s) {
  if (a->dtype != KLAB_BF16 || (a->ldo & 7) || (a->ldq & 7) || (a->ldk & 7) || (a->ldv & 7) || (a->lddo & 7) || (a->lddq & 3) ||
      (a->lddk & 3) || (a->lddv & 3))
    return KLAB_ERR_UNSUPPORTED;
  AttnMP p = to_mp(a);
  switch (a->dk) {
    case 16: return launch_bwd<16>(p, s);
    case 32: return launch_bwd<32>(p, s);
    case 64: return launch_bwd<64>(p, s);
    case 128: return launch_bwd<128>(p, s);
  }
  return KLAB_ERR_UNSUPPORTED;
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
