// mm8p: the large-tile member of the klab_gemm family -- C[M,N] = epilogue(alpha * sum_k A(m,k) B(n,k)) on 256 x 256 output
// tiles, eight waves (2 x 4, each 128 x 64 = 32 accumulator tiles of 16 x 16), BK = 64, for the products whose K is long
// enough to pay for a deep pipeline: every Linear / dgrad / wgrad of T5-base / T5-large and of the wide Swin stages
// (BASELINE configs[2..4]), the LM-head weight and input gradients and the per-layer weight gradients of configs[1].
// Replaces nothing in the reference by itself: it is another kernel behind klab_gemm (HF/t5:83-94,206-209,1047 and their
// autograd gradients), chosen by the dispatcher.
//
// Why a second main loop.  gemm.hip's ring (four waves of 64 x 64, BK = 32, one barrier per 16 MFMAs of a wave) tops out at
// 0.6-1.0 PFLOP/s: a wave's 16 MFMAs barely cover its fragment reads and there is one workgroup barrier per 32 k.  Here a
// wave owns 128 x 64 and a k-tile (64 k) is cut into FOUR PHASES of 16 MFMAs -- one 64 x 32 quadrant of the wave tile over
// the whole 64 k each -- so that consecutive phases share either their A or their B fragments:
//
//   phase 0: read A(m-half 0) [8 x b128], B(n-half 0) [4 x b128]      MFMA quadrant (m0, n0)
//   phase 1: read             B(n-half 1) [4]                          MFMA quadrant (m0, n1)
//   phase 2: read A(m-half 1) [8]                                      MFMA quadrant (m1, n1)
//   phase 3: no LDS read                                               MFMA quadrant (m1, n0)
//
// i.e. 24 fragment reads behind 64 MFMAs (the four-wave ring: 8 behind 16), and every phase also issues the LDS-DMA of ONE
// half-tile (128 rows x 64 k = 16 KiB, two global_load_lds_dwordx4 per wave).
//
// LDS: eight half-tile slots of 16 KiB (two k-tiles x {A0, B0, B1, A1}).  Half-tiles are numbered in the order they are
// consumed, h = 4 t + {0: A0, 1: B0, 2: B1, 3: A1}; half-tile h lives in slot h mod 8 and is last read in phase h (A0) or
// h - 1 (the others).  Phase g issues half-tile g + 7, whose slot was last read in phase g - 1 or earlier -- a full phase
// (two barriers) before.  One counted wait per k-tile: in phase 3, behind that phase's issue, `s_waitcnt vmcnt(6)` leaves the
// three youngest half-tiles (2 instructions each) in flight, i.e. everything the NEXT k-tile reads has landed; the barrier
// that follows makes it visible to all waves, and the reads happen one phase later.  LDS-DMA stays in flight across barriers.
//
// K-major operand (x[M,K] of a Linear forward, W[N,K]): half-tile image [128 rows][64 k] = 128-B rows; LDS-DMA writes
// linearly (lane l of an instruction -> row l>>3, 16-B position l&7), so the bank swizzle is applied to the SOURCE chunk:
// position p of row r holds global chunk p ^ (r & 7); fragments by ds_read_b128 at position c ^ (r & 7): the 16 lanes of
// every b128 lane group then touch 16 different 16-B slots of the 256-B bank row.
// m-major operand (contraction over the slow dimension: the W of a dgrad, both operands of a wgrad): image [64 k][128 m] =
// 256-B rows as stored in HBM; position p of k-row kr holds chunk p ^ 2 f(kr), f(kr) = (kr & 3) | ((kr >> 3) & 1) << 2;
// fragments by two ds_read_b64_tr_b16 (k-rows 8 g + q and + 4).
//
// Everything here is plain HIP: MFMA and LDS-DMA builtins, ordinary LDS loads, raw s_barrier and inline `s_waitcnt` only;
// hipcc schedules inside a phase and keeps the hazards (sched_barrier pins the phase boundaries).
#include <stdlib.h>

#include <type_traits>

#include "gemm_shared.h"

namespace klab {

namespace p8 {
constexpr int BM = 256, BN = 256, BK = 64, NT = 512;
constexpr int HALF = 16384;       // bytes of one half-tile image
constexpr int LDS_RING = 8 * HALF;
typedef __attribute__((address_space(3))) char lds_char;

__device__ __forceinline__ int mmajor_f(int kr) { return (kr & 3) | (((kr >> 3) & 1) << 2); }

// image row R (0..127) of half-tile H holds tile row (R / BLK) * 2 BLK + H * BLK + R % BLK: BLK = 64 for A (the first / second 64
// rows of each of the two wave rows), 32 for B (the first / second 32 columns of each of the four wave columns), so that every
// wave finds its own m-half (n-half) 0 in half-tile 0 and its half 1 in half-tile 1
template <bool KMAJOR, int BLK>
struct Operand {
  static __device__ __forceinline__ int tile_row(int R, int H) { return (R / BLK) * (2 * BLK) + H * BLK + (R % BLK); }
  const bf16_t* src[2][2];  // [half][instruction of this wave]: this lane's source address for k-tile 0
  long kstep;               // elements per k-tile
  __device__ __forceinline__ void init(const bf16_t* base, long ld, int row0, int nrows, int wave, int lane) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ins = wave * 2 + i;  // 16 instructions of 1 KiB fill one half-tile
        if constexpr (KMAJOR) {
          const int r = ins * 8 + (lane >> 3);
          const int c = (lane & 7) ^ (lane >> 3);  // (r & 7) == lane >> 3
          int gr = row0 + tile_row(r, h);
          gr = gr < nrows ? gr : nrows - 1;  // rows past the edge are computed, never stored
          src[h][i] = base + (long)gr * ld + c * 8;
        } else {
          const int kr = ins * 4 + (lane >> 4);
          const int c = (lane & 15) ^ (mmajor_f(kr) << 1);
          int m = row0 + tile_row(c * 8, h);  // (BLK is a multiple of 8: a chunk's 8 image columns are 8 consecutive tile rows)
          m = m + 8 <= nrows ? m : nrows - 8;
          src[h][i] = base + (long)kr * ld + m;
        }
      }
    kstep = KMAJOR ? 64 : 64 * ld;
  }
  // LDS-DMA of half `h` of k-tile kt into the slot at byte offset `slot`
  __device__ __forceinline__ void issue(int h, long kt, char* smem, int slot, int wave) const {
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[h][i] + kt * kstep),
                                       (__attribute__((address_space(3))) void*)(smem + slot + (wave * 2 + i) * 1024), 16, 0, 0);
  }
};

// fragment f (16 rows starting at 16 f inside the half-tile), k-step s (32 k) of the image at `img`
template <bool KMAJOR>
__device__ __forceinline__ bf16x8 frag(const char* img, int f, int s, int lane) {
  if constexpr (KMAJOR) {
    const int r = f * 16 + (lane & 15);
    const int c = (4 * s + (lane >> 4)) ^ (lane & 7);
    return *reinterpret_cast<const bf16x8*>(img + r * 128 + c * 16);
  } else {
    typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
    const int g = lane >> 4, q4 = (lane & 15) >> 2, pp = lane & 3;
    const int kr = 32 * s + 8 * g + q4;
    const int chunk = (2 * f + (pp >> 1)) ^ (mmajor_f(kr) << 1);
    const char* a = img + kr * 256 + chunk * 16 + (pp & 1) * 8;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(a));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(a + 4 * 256));
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  }
}

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// Epilogue of one [128 x 256] slab.  The waves that own the slab park their RAW accumulators (f32) in LDS -- a store per
// accumulator register, nothing else live -- and then all 512 threads walk the slab as whole row pieces (16 B of output per
// lane, consecutive lanes on consecutive addresses) and apply the element-wise tail there: alpha, bias, activation, the
// relu-mask / gelu' operand, dropout, residual, C += -- in klab_gemm's order (gemm_shared.h), with COALESCED reads of the mask
// and the residual.  (Applying the tail in the MFMA layout, as the four-wave kernels do, kept 128 accumulators live across ten
// fully unrolled variants: 546 spilled registers and 700 MB of scratch traffic per launch.)
constexpr int SLAB_PITCH = (BN + 4) * 4;  // bytes per staged f32 row (16-B pad: conflict-free f32x4 stores from the MFMA layout)

// FLAGS: compile-time feature set (EF_* of gemm_shared.h; EF_GENERIC = everything decided at run time).  A body with per-element
// run-time tests -- ten uniform branches per element -- measured 20 us per 4096 x 4096 output against 10 us for the four-wave
// kernels' epilogue; each variant here is straight-line code over one 16-byte output vector.
template <typename OutT, int FLAGS>
__device__ __forceinline__ void slab_out_v(const GemmP& p, const char* smem, int bm0, int bn0, int tid, float alpha) {
  constexpr bool GEN = (FLAGS & EF_GENERIC) != 0;
  constexpr int ESZ = sizeof(OutT), EPT = 16 / ESZ;  // elements per thread per step: one 16-byte output vector
  constexpr int CPR = BN / EPT, NCH = 128 * CPR;
  const bool vec_ok = ((p.ldc * ESZ) & 15) == 0 && (p.N % EPT) == 0;
  const DropCtx dc = make_drop(p.seed, p.tag, p.drop_p);
  OutT* Cp = reinterpret_cast<OutT*>(p.C);
  const bf16_t* auxp = reinterpret_cast<const bf16_t*>(p.aux);
  const bool aux_vec = p.aux && !(p.ldaux & 7) && !((uintptr_t)p.aux & 15);
  const bool res_vec = p.residual && p.r_f32 && !(p.ldr & 3) && !((uintptr_t)p.residual & 15);
#pragma unroll 2
  for (int ch = tid; ch < NCH; ch += NT) {
    const int row = ch / CPR, cc = ch % CPR;
    const long m = bm0 + row;
    const int n0 = bn0 + cc * EPT;
    if (m >= p.M || n0 >= p.N) continue;
    float x[EPT];
#pragma unroll
    for (int q = 0; q < EPT / 4; ++q) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(smem + row * SLAB_PITCH + (cc * EPT + q * 4) * 4);
      x[q * 4 + 0] = v[0]; x[q * 4 + 1] = v[1]; x[q * 4 + 2] = v[2]; x[q * 4 + 3] = v[3];
    }
    const int nv = (p.N - n0) < EPT ? (p.N - n0) : EPT;  // valid elements (ragged last vector)
    const bool whole = vec_ok && nv == EPT;
    // operands of the tail as whole vectors where alignment allows (interior vectors of aligned tensors: the common case)
    float av[EPT], rv[EPT];
    constexpr bool NEED_AUX = GEN || (FLAGS & (EF_AUXNZ | EF_DGELU)) != 0, NEED_RES = GEN || (FLAGS & EF_RES) != 0;
    if constexpr (NEED_AUX) {
      if (p.aux) {
        if (whole && aux_vec && EPT == 8) {
          const bf16x8 a = *reinterpret_cast<const bf16x8*>(auxp + m * p.ldaux + n0);
#pragma unroll
          for (int u = 0; u < EPT; ++u) av[u] = (float)a[u & 7];
        } else {
#pragma unroll
          for (int u = 0; u < EPT; ++u) av[u] = (float)auxp[m * p.ldaux + n0 + (u < nv ? u : 0)];
        }
      }
    }
    if constexpr (NEED_RES) {
      if (p.residual) {
        if (whole && res_vec) {
#pragma unroll
          for (int q = 0; q < EPT / 4; ++q) {
            const f32x4 r = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.residual) + m * p.ldr + n0 + q * 4);
            rv[q * 4 + 0] = r[0]; rv[q * 4 + 1] = r[1]; rv[q * 4 + 2] = r[2]; rv[q * 4 + 3] = r[3];
          }
        } else {
#pragma unroll
          for (int u = 0; u < EPT; ++u) {
            const long n = n0 + (u < nv ? u : 0);
            rv[u] = p.r_f32 ? reinterpret_cast<const float*>(p.residual)[m * p.ldr + n] : (float)reinterpret_cast<const bf16_t*>(p.residual)[m * p.ldr + n];
          }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < EPT; ++u) {
      const int n = n0 + (u < nv ? u : 0);  // clamped: the value of an invalid element is never stored
      float y = x[u] * alpha;
      if constexpr (GEN) {
        if (p.bias) y += p.bias[n];
        if (p.act == KLAB_ACT_RELU) y = fmaxf(y, 0.f);
        else if (p.act == KLAB_ACT_GELU) y = gelu_for<bf16_t>(y);
        if (p.aux) {
          if (p.aux_mode == KLAB_AUX_NONZERO) y = (av[u] != 0.f) ? y * p.aux_scale : 0.f;
          else if (p.aux_mode == KLAB_AUX_DGELU) y *= gelu_erf_grad(av[u]);
        }
        y *= drop_mult(dc, (uint64_t)m * (uint64_t)p.N + (uint64_t)n);
        if (p.residual) y += rv[u];
      } else {
        if constexpr (FLAGS & EF_BIAS) y += p.bias[n];
        if constexpr (FLAGS & EF_RELU) y = fmaxf(y, 0.f);
        if constexpr (FLAGS & EF_GELU) y = gelu_for<bf16_t>(y);
        if constexpr (FLAGS & EF_AUXNZ) y = (av[u] != 0.f) ? y * p.aux_scale : 0.f;
        if constexpr (FLAGS & EF_DGELU) y *= gelu_erf_grad(av[u]);
        if constexpr (FLAGS & EF_DROP) y *= drop_mult32_nb(dc, (uint32_t)m * (uint32_t)p.N + (uint32_t)n);  // (M * N < 2^32: fill_gemmp)
        if constexpr (FLAGS & EF_RES) y += rv[u];
      }
      x[u] = y;
    }
    OutT* dst = Cp + m * p.ldc + n0;
    if (whole) {
      if (p.accumulate) {
        if constexpr (ESZ == 4) {
          const f32x4 o = *reinterpret_cast<const f32x4*>(dst);
          x[0] += o[0]; x[1] += o[1]; x[2] += o[2]; x[3] += o[3];
        } else {
          const bf16x8 o = *reinterpret_cast<const bf16x8*>(dst);
#pragma unroll
          for (int u = 0; u < EPT; ++u) x[u] += (float)o[u & 7];
        }
      }
      if constexpr (ESZ == 4) *reinterpret_cast<f32x4*>(dst) = f32x4{x[0], x[1], x[2], x[3]};
      else *reinterpret_cast<bf16x8*>(dst) = bf16x8{(bf16_t)x[0], (bf16_t)x[1], (bf16_t)x[2], (bf16_t)x[3], (bf16_t)x[4 % EPT], (bf16_t)x[5 % EPT], (bf16_t)x[6 % EPT], (bf16_t)x[7 % EPT]};
    } else {
      for (int u = 0; u < nv; ++u) dst[u] = from_f32<OutT>(p.accumulate ? x[u] + to_f32(dst[u]) : x[u]);
    }
  }
}

template <typename OutT>
__device__ __forceinline__ void slab_out(const GemmP& p, const char* smem, int bm0, int bn0, int tid, float alpha) {
#define KLAB_SO(F) slab_out_v<OutT, F>(p, smem, bm0, bn0, tid, alpha)
  switch (p.epi) {  // workgroup-uniform: only the selected variant's instructions are fetched
    case 0: KLAB_SO(0); break;
    case EF_BIAS: KLAB_SO(EF_BIAS); break;
    case EF_BIAS | EF_GELU: KLAB_SO(EF_BIAS | EF_GELU); break;
    case EF_RELU: KLAB_SO(EF_RELU); break;
    case EF_RELU | EF_DROP: KLAB_SO(EF_RELU | EF_DROP); break;
    case EF_RES: KLAB_SO(EF_RES); break;
    case EF_DROP | EF_RES: KLAB_SO(EF_DROP | EF_RES); break;
    case EF_AUXNZ: KLAB_SO(EF_AUXNZ); break;
    case EF_DGELU: KLAB_SO(EF_DGELU); break;
    default: KLAB_SO(EF_GENERIC); break;
  }
#undef KLAB_SO
}

template <bool AK, bool BKM, bool ATOMIC>
__device__ __forceinline__ void mm8p_body(const GemmP& p, const int bid) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = (wave >> 2) * 128, wn = (wave & 3) * 64;  // 2 x 4 waves
  const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  const int tile = bid % tiles, split = bid / tiles;
  int bm0, bn0;
  tile_of_block(p, BM, BN, tile, bm0, bn0);
  const int nt_all = p.K / BK;
  const int per = (nt_all + p.splits - 1) / p.splits;
  const int kt0 = split * per, kt1 = (kt0 + per < nt_all) ? kt0 + per : nt_all;
  const int nt = kt1 - kt0;
  if (nt <= 0) return;

  Operand<AK, 64> oa;
  Operand<BKM, 32> ob;
  oa.init(reinterpret_cast<const bf16_t*>(p.A), p.lda, bm0, p.M, wave, lane);
  ob.init(reinterpret_cast<const bf16_t*>(p.B), p.ldb, bn0, p.N, wave, lane);

  // LDS-DMA of one half-tile of k-tile t (relative to kt0): which = 0 A0, 1 B0, 2 B1, 3 A1; slot index 0..7
  auto issue_half = [&](int t, auto which_c, int slot) {
    constexpr int W = decltype(which_c)::value;
    if (t >= nt) return;  // wave-uniform: past the end of the k-range
    const long kt = kt0 + t;
    if constexpr (W == 0) oa.issue(0, kt, smem, slot * HALF, wave);
    else if constexpr (W == 1) ob.issue(0, kt, smem, slot * HALF, wave);
    else if constexpr (W == 2) ob.issue(1, kt, smem, slot * HALF, wave);
    else oa.issue(1, kt, smem, slot * HALF, wave);
  };
  using W0 = std::integral_constant<int, 0>; using W1 = std::integral_constant<int, 1>;
  using W2 = std::integral_constant<int, 2>; using W3 = std::integral_constant<int, 3>;

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // prologue: half-tiles 0..6; the first k-tile (0..3) must have landed, 4..6 stay in flight
  issue_half(0, W0{}, 0); issue_half(0, W1{}, 1); issue_half(0, W2{}, 2); issue_half(0, W3{}, 3);
  issue_half(1, W0{}, 4); issue_half(1, W1{}, 5); issue_half(1, W2{}, 6);
  if (nt >= 2) wait_vm<6>(); else wait_vm<0>();
  __builtin_amdgcn_s_barrier();

  // this wave's rows inside the half-tile images (see Operand::tile_row)
  const int a_img_row0 = (wave >> 2) * 64;   // this wave's 64 rows inside an A half-tile image
  const int b_img_row0 = (wave & 3) * 32;    // this wave's 32 rows inside a B half-tile image

  auto ktile_body = [&](auto buf_c, int t) {
    constexpr int BUF = decltype(buf_c)::value;
    const char* A0 = smem + (BUF * 4 + 0) * HALF;
    const char* B0 = smem + (BUF * 4 + 1) * HALF;
    const char* B1 = smem + (BUF * 4 + 2) * HALF;
    const char* A1 = smem + (BUF * 4 + 3) * HALF;
    bf16x8 af[4][2], bf0[2][2], bf1[2][2];
    // ---- phase 0 ----
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int s = 0; s < 2; ++s) af[i][s] = frag<AK>(A0, (a_img_row0 >> 4) + i, s, lane);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int s = 0; s < 2; ++s) bf0[j][s] = frag<BKM>(B0, (b_img_row0 >> 4) + j, s, lane);
    issue_half(t + 1, W3{}, (BUF ^ 1) * 4 + 3);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    wait_lgkm0();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if constexpr (ATOMIC) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][s], bf0[j][s], acc[i][j], 0, 0, 0);
          else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf0[j][s], af[i][s], acc[i][j], 0, 0, 0);
        }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---- phase 1 ----
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int s = 0; s < 2; ++s) bf1[j][s] = frag<BKM>(B1, (b_img_row0 >> 4) + j, s, lane);
    issue_half(t + 2, W0{}, BUF * 4 + 0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    wait_lgkm0();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if constexpr (ATOMIC) acc[i][2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][s], bf1[j][s], acc[i][2 + j], 0, 0, 0);
          else acc[i][2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf1[j][s], af[i][s], acc[i][2 + j], 0, 0, 0);
        }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---- phase 2 ----
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int s = 0; s < 2; ++s) af[i][s] = frag<AK>(A1, (a_img_row0 >> 4) + i, s, lane);
    issue_half(t + 2, W1{}, BUF * 4 + 1);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    wait_lgkm0();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if constexpr (ATOMIC) acc[4 + i][2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][s], bf1[j][s], acc[4 + i][2 + j], 0, 0, 0);
          else acc[4 + i][2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf1[j][s], af[i][s], acc[4 + i][2 + j], 0, 0, 0);
        }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---- phase 3 ----
    issue_half(t + 2, W2{}, BUF * 4 + 2);
    // everything the next k-tile reads (half-tiles <= 4 t + 7) has landed once at most three younger half-tiles are in
    // flight; near the end of the k-range fewer have been issued, so the wait is for all of them
    if (t + 2 < nt) wait_vm<6>(); else wait_vm<0>();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if constexpr (ATOMIC) acc[4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][s], bf0[j][s], acc[4 + i][j], 0, 0, 0);
          else acc[4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf0[j][s], af[i][s], acc[4 + i][j], 0, 0, 0);
        }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  };

  int t = 0;
  for (; t + 1 < nt; t += 2) {
    ktile_body(std::integral_constant<int, 0>{}, t);
    ktile_body(std::integral_constant<int, 1>{}, t + 1);
  }
  if (t < nt) ktile_body(std::integral_constant<int, 0>{}, t);

  // ---- epilogue -------------------------------------------------------------------------------------------------
  // accumulator (i, j) of this wave: tile rows  wm + (i >> 2) * 64 ... see row mapping: image row -> tile row
  wait_vm<0>();
  float alpha = p.alpha;
  if (p.alpha_dev) alpha *= p.alpha_dev[0];
  // tile row of accumulator block i (16 rows): m-half (i >> 2) lives in A half-tile (i >> 2) at image rows a_img_row0 + (i & 3) * 16,
  // and image row R of A half-tile H is tile row (R / 64) * 128 + H * 64 + R % 64
  // => tile row0(i) = (wave >> 2) * 128 + (i >> 2) * 64 + (i & 3) * 16 = wm + i * 16.   Same for n: wn + j * 16.
  if constexpr (ATOMIC) {
    float* Cf = reinterpret_cast<float*>(p.C);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = bn0 + wn + j * 16 + (lane & 15);
        if (n >= p.N) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = bm0 + wm + i * 16 + (lane >> 4) * 4 + r;
          if (m < p.M) atomicAdd(Cf + (long)m * p.ldc + n, acc[i][j][r] * alpha);
        }
      }
  } else {
    // two slabs of 128 rows (the two wave rows)
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
      __syncthreads();  // ring reads / the previous slab's copy-out are done
      if ((wave >> 2) == half) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            *reinterpret_cast<f32x4*>(smem + (i * 16 + (lane & 15)) * SLAB_PITCH + (wn + j * 16 + (lane >> 4) * 4) * 4) = acc[i][j];
      }
      __syncthreads();
      if (p.c_f32) slab_out<float>(p, smem, bm0 + half * 128, bn0, tid, alpha);
      else slab_out<bf16_t>(p, smem, bm0 + half * 128, bn0, tid, alpha);
    }
  }
}

template <bool AK, bool BKM, bool ATOMIC>
__global__ __launch_bounds__(512) void mm8p_kernel(GemmP p) { mm8p_body<AK, BKM, ATOMIC>(p, blockIdx.x); }

// Up to 8 weight-gradient products dW[N_out, K_in] = dY^T X (both operands token-major, f32 C += product) in ONE grid: the
// weight gradients of a T5 layer (HF/t5 autograd of :83-94, 206-209).  One 256 x 256 tile over the WHOLE contraction per
// workgroup: no split-K, no atomics, half the operand traffic of the 128 x 128 split-K form (gemm.hip's grouped kernel), and a
// layer's gradients occupy ~50 CUs instead of all 256 -- they run on a side stream beside the activation-gradient chain.
struct Group8Entry { const void* A; long lda; const void* B; long ldb; float* C; long ldc; int M, N, K, start; float alpha; };
constexpr int GROUP_MAX = 32;  // (several layers' products in one grid: KLAB_WGRAD_GROUP_TILES)
struct Group8P { Group8Entry e[GROUP_MAX]; int n; };
__global__ __launch_bounds__(512) void mm8p_grouped_tn_kernel(Group8P g) {
  int i = 0;
#pragma unroll
  for (int k = 1; k < GROUP_MAX; ++k)
    if (k < g.n && (int)blockIdx.x >= g.e[k].start) i = k;
  const Group8Entry& e = g.e[i];
  GemmP p;
  p.M = e.M; p.N = e.N; p.K = e.K;
  p.A = e.A; p.lda = e.lda; p.a_kmajor = 0;
  p.B = e.B; p.ldb = e.ldb; p.b_kmajor = 0;
  p.C = e.C; p.ldc = e.ldc; p.c_f32 = 1; p.accumulate = 1;
  p.alpha = e.alpha; p.alpha_dev = nullptr; p.bias = nullptr; p.act = 0;
  p.aux = nullptr; p.ldaux = 0; p.aux_mode = 0; p.aux_scale = 1.f;
  p.residual = nullptr; p.ldr = 0; p.r_f32 = 1;
  p.drop_p = 0.f; p.seed = nullptr; p.tag = 0;
  p.splits = 1; p.epi = 0; p.ablate = 0;
  mm8p_body<false, false, false>(p, (int)blockIdx.x - e.start);
}

}  // namespace p8

// grouped weight gradients on the large tiles: returns KLAB_ERR_UNSUPPORTED unless EVERY member fits (the caller then uses the
// 128 x 128 split-K grouped kernel for the whole list)
int mm8p_grouped_try(const klab_gemm_args* list, int n, hipStream_t s) {
  using namespace p8;
  static const int mode = [] { const char* e = getenv("KLAB_WGRAD_P8"); return e ? atoi(e) : 0; }();  // 1: every grouped list (experiment)
  if ((mode == 0 && !tl_grouped_large_tiles) || n <= 0 || n > GROUP_MAX) return KLAB_ERR_UNSUPPORTED;
  Group8P g;
  g.n = 0;
  int blocks = 0;
  for (int i = 0; i < n; ++i) {
    const klab_gemm_args* a = &list[i];
    const bool ok = a->dtype == KLAB_BF16 && !a->a_kmajor && !a->b_kmajor && a->c_dtype == KLAB_F32 && a->accumulate && !a->bias && !a->act &&
                    !a->aux && !a->residual && a->drop_p == 0.f && !a->alpha_dev && a->M >= 128 && a->N >= 128 && (a->K % BK) == 0 &&
                    a->K >= 1024 && !(a->M & 7) && !(a->N & 7) && !(a->lda & 7) && !(a->ldb & 7) && !(a->ldc & 3) &&
                    !((uintptr_t)a->A & 15) && !((uintptr_t)a->B & 15) && !((uintptr_t)a->C & 15);
    if (!ok) return KLAB_ERR_UNSUPPORTED;
    Group8Entry& e = g.e[g.n++];
    e.A = a->A; e.lda = a->lda; e.B = a->B; e.ldb = a->ldb; e.C = (float*)a->C; e.ldc = a->ldc;
    e.M = a->M; e.N = a->N; e.K = a->K; e.start = blocks; e.alpha = a->alpha;
    blocks += ((a->M + BM - 1) / BM) * ((a->N + BN - 1) / BN);
  }
  const size_t epi_bytes = (size_t)128 * SLAB_PITCH;
  const size_t lds = LDS_RING > epi_bytes ? LDS_RING : epi_bytes;
  int rc = ensure_dyn_lds(reinterpret_cast<const void*>(mm8p_grouped_tn_kernel), lds);
  if (rc) return rc;
  probed_launch(mm8p_grouped_tn_kernel, dim3((unsigned)blocks), dim3(NT), lds, s, g);  // (carries the engine's probe events, if armed)
  KLAB_LAUNCH_CHECK();
  return KLAB_OK;
}

// Host side: is this product worth the large tiles, and the launch.  Returns KLAB_ERR_UNSUPPORTED when it is not taken.
int mm8p_try(const GemmP& pin, bool atomic_ok, int force, hipStream_t s) {
  using namespace p8;
  static const int mode = [] { const char* e = getenv("KLAB_GEMM_P8"); return e ? atoi(e) : 1; }();  // 0: off, 1: heuristic, 2: whenever legal
  if (mode == 0 && force <= 0) return KLAB_ERR_UNSUPPORTED;
  GemmP p = pin;
  if (p.K % BK || p.K < 2 * BK) return KLAB_ERR_UNSUPPORTED;
  if (!p.a_kmajor && (p.M % 8)) return KLAB_ERR_UNSUPPORTED;
  if (!p.b_kmajor && (p.N % 8)) return KLAB_ERR_UNSUPPORTED;
  if (p.M < 8 || p.N < 8) return KLAB_ERR_UNSUPPORTED;
  const long tiles = (long)((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  const int ncu = 256;
  int splits = 1;
  if (atomic_ok) {
    // split K until the grid covers the chip once; every split keeps at least 8 k-tiles
    const int nt = p.K / BK;
    while (tiles * splits * 2 <= ncu && nt / (splits * 2) >= 8) splits *= 2;
  }
  if (mode == 1 && force <= 0) {
    // Measured envelope (tools/gemm_bench.py --sq, KLAB_BENCH_AB=1): per 64 k the main loop takes 1.5 us on a 4096 x 4096 output
    // (1.44 PFLOP/s marginal) against 1.7 -> 4.0 us for the four-wave ring as K grows from 1024 to 8192, but a launch carries a
    // larger fixed cost (one workgroup per CU: nothing hides the ring fill and the two-slab epilogue).  It pays from K = 2048 per
    // split, and only when the 256 x 256 tiles (x splits) fill the chip's 256 CUs in whole rounds.
    // Same-process A/B per shape (profiles/r03_gemm_large_tile_ab.txt): wins from K = 1024 per split when the workgroups fill
    // whole rounds of the chip (4096 x 4096: 856 vs 762 TFLOP/s at K = 1024, 1247 vs 725 at K = 4096), loses when they do not
    // (T5-large M = 4896, N = 1024: 80 tiles on 256 CUs) and on long-thin outputs whose one operand every tile re-reads (the
    // LM-head weight gradient, 126 x 2 tiles).
    const long wgs = tiles * splits;
    const long rounds = (wgs + ncu - 1) / ncu;
    const long tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    if (p.K / splits < 1024) return KLAB_ERR_UNSUPPORTED;
    if (wgs * 100 < rounds * ncu * (p.K / splits >= 2048 ? 80 : 90)) return KLAB_ERR_UNSUPPORTED;
    if (splits == 1 && (tiles_m < 4 || tiles_n < 4)) return KLAB_ERR_UNSUPPORTED;
  }
  p.splits = splits;
  // the copy-out variants of the small-tile kernels do not exist here: fold them back into the accumulator-layout forms
  if (p.epi == EF_AUXNZ_CO) p.epi = EF_AUXNZ;
  else if (p.epi == EF_DGELU_CO) p.epi = EF_DGELU;
  else if (p.epi == EF_RES_CO) p.epi = EF_RES;
  else if (p.epi == (EF_DROP | EF_RES_CO)) p.epi = EF_DROP | EF_RES;
  const bool atomic = atomic_ok && splits > 1;
  const size_t epi_bytes = (size_t)128 * (BN + 4) * 4;
  const size_t lds = LDS_RING > epi_bytes ? LDS_RING : epi_bytes;
  const dim3 grid((unsigned)(tiles * splits));
  int rc;
#define KLAB_P8(AKv, BKv, ATv)                                                                           \
  {                                                                                                     \
    rc = ensure_dyn_lds(reinterpret_cast<const void*>(mm8p_kernel<AKv, BKv, ATv>), lds);                \
    if (rc) return rc;                                                                                  \
    probed_launch(mm8p_kernel<AKv, BKv, ATv>, grid, dim3(NT), lds, s, p);                               \
  }
  if (atomic) {
    if (p.a_kmajor && p.b_kmajor) KLAB_P8(true, true, true)
    else if (p.a_kmajor) KLAB_P8(true, false, true)
    else if (p.b_kmajor) KLAB_P8(false, true, true)
    else KLAB_P8(false, false, true)
  } else {
    if (p.a_kmajor && p.b_kmajor) KLAB_P8(true, true, false)
    else if (p.a_kmajor) KLAB_P8(true, false, false)
    else if (p.b_kmajor) KLAB_P8(false, true, false)
    else KLAB_P8(false, false, false)
  }
#undef KLAB_P8
  KLAB_LAUNCH_CHECK();
  return KLAB_OK;
}

}  // namespace klab
