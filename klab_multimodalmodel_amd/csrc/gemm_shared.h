// Pieces of the GEMM family shared by gemm.hip (LDS-DMA ring, 4 waves) and mm8p.hip (256x256 eight-wave tiles): the kernel
// parameter block, the XCD-aware workgroup -> tile map and the element-wise epilogue applied in the MFMA accumulator layout.
#pragma once
#include "common.h"
#include "klab_mm.h"

namespace klab {

struct GemmP {
  int M, N, K;
  const void* A; long lda; int a_kmajor;
  const void* B; long ldb; int b_kmajor;
  void* C; long ldc; int c_f32; int accumulate;
  float alpha; const float* alpha_dev;
  const float* bias;
  int act;
  const void* aux; long ldaux; int aux_mode; float aux_scale;  // aux has the input dtype
  const void* residual; long ldr; int r_f32;
  float drop_p; const uint32_t* seed; uint32_t tag;
  int splits;
  int epi;     // feature set of the epilogue (EF_* bits), chosen on the host
  int ablate;  // diagnostics only (KLAB_GEMM_ABLATE): 1 = no global loads, 2 = no MFMA, 4 = no LDS fragment reads
};

// Workgroup -> output tile.  Blocks are dealt round-robin over the 8 XCDs (b and b+8 share an L2), so the grid
// is re-linearised to give every XCD one contiguous chunk of the tile sequence (bijective for any grid size),
// and the sequence runs fastest along the dimension whose operand is SMALLER: that operand stays resident in
// the XCD's 4 MiB L2 while the other one streams through once.  Speed only, never correctness.
__device__ __forceinline__ void tile_of_block(const GemmP& p, int BM, int BN, int bid, int& bm0, int& bn0) {
  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
  const int nwg = tiles_m * tiles_n;
  const int xcd = bid & 7, idx = bid >> 3;
  const int q = nwg >> 3, r = nwg & 7;
  const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  // The smaller operand's tiles vary fastest (consecutive workgroups of an XCD share the other operand's panel), in
  // GROUPS whose panels fit in half of the XCD's 4 MiB L2: walking all 32 m-tiles of the LM head (4.2 MB of A) before
  // the next n-tile evicted A on every pass -- PMC showed ~10x the algorithmic read bytes leaving L2.
  const int kbytes = (p.K / (p.splits > 0 ? p.splits : 1)) * 2;
  if ((long)p.M <= (long)p.N) {  // A (M x K) is the smaller operand: m fastest
    int G = (2 << 20) / (BM * kbytes);
    G = (G < 8 || G > tiles_m || (p.ablate & 32)) ? tiles_m : G;  // groups narrower than 8 tiles would re-read the streamed operand too often
    const int per_group = G * tiles_n, full = tiles_m / G;
    const int g = lin / per_group;
    if (g < full) { const int rem = lin - g * per_group; bm0 = (int)(g * G + rem % G) * BM; bn0 = (int)(rem / G) * BN; }
    else { const int rem = lin - full * per_group, gm = tiles_m - full * G; bm0 = (int)(full * G + rem % gm) * BM; bn0 = (int)(rem / gm) * BN; }
  } else {
    int G = (2 << 20) / (BN * kbytes);
    G = (G < 8 || G > tiles_n || (p.ablate & 32)) ? tiles_n : G;
    const int per_group = G * tiles_m, full = tiles_n / G;
    const int g = lin / per_group;
    if (g < full) { const int rem = lin - g * per_group; bn0 = (int)(g * G + rem % G) * BN; bm0 = (int)(rem / G) * BM; }
    else { const int rem = lin - full * per_group, gn = tiles_n - full * G; bn0 = (int)(full * G + rem % gn) * BN; bm0 = (int)(rem / gn) * BM; }
  }
}

// ---- epilogue shared by all tile kernels -----------------------------------------------------------
// The element-wise tail (bias, act, aux, dropout, residual) is applied in registers, the finished tile is
// parked in LDS in the OUTPUT dtype, and the workgroup then streams it out as whole rows: 16 B per lane,
// consecutive lanes on consecutive addresses.  (Storing straight from the MFMA layout wrote 32-B pieces of
// 16 different rows per instruction: partial-line writes were the bound of every short-K GEMM.)
// Caller guarantees that all waves are past their last LDS read of the main loop (a barrier).
enum : int { EF_BIAS = 1, EF_RELU = 2, EF_GELU = 4, EF_AUXNZ = 8, EF_DGELU = 16, EF_DROP = 32, EF_RES = 64, EF_GENERIC = 128,
             // the same two features applied while the staged tile is streamed out (16 B per lane, row-contiguous: coalesced
             // reads of the mask / residual) instead of in the MFMA layout (2- and 4-byte reads of 16 rows per instruction:
             // the relu-mask dgrad ran at 38 us against 16 us for the plain GEMM); chosen on the host when dtypes/alignment allow
             EF_AUXNZ_CO = 256, EF_RES_CO = 512,
             // gelu'(z) applied while the staged bf16 tile is streamed out: z is read as whole 16-byte row pieces (in the MFMA layout each
             // lane fetched 64 separate 2-byte values from 16 different rows)
             EF_DGELU_CO = 1024 };

// FLAGS is a compile-time feature set: each variant contains only the code of its features, fully unrolled over the
// lane's 16-64 accumulator elements (~0.3-2 K instructions).  One big run-time-flagged body (every feature x every
// element, erf included) was ~15 K instructions and thrashed the instruction cache: two thirds of the LM-head
// GEMM's time went into its epilogue.  EF_GENERIC keeps that fully general body (rolled) for unusual combinations.
template <typename T, int BM, int BN, int MI, int NI, int FLAGS>
__device__ __forceinline__ void staged_epilogue_v(const GemmP& p, f32x4 (&acc)[MI][NI], float alpha, char* smem, int bm0, int bn0, int wm,
                                                  int wn, int tid, int lane) {
  constexpr bool GEN = (FLAGS & EF_GENERIC) != 0;
  const DropCtx dc = make_drop(p.seed, p.tag, p.drop_p);
  const bool f32out = p.c_f32 || sizeof(T) == 4;
  const int pitchB = f32out ? (BN + 4) * 4 : (BN + 8) * 2;  // bytes per LDS row (16-B pad)
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int ml = wm + i * 16 + (lane & 15);
    const int m = bm0 + ml;
    const int mc = m < p.M ? m : p.M - 1;  // clamped: rows past the edge are computed but never stored
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int nl = wn + j * 16 + (lane >> 4) * 4;
      const int n0 = bn0 + nl;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float x = acc[i][j][r] * alpha;
        const int n = n0 + r;
        const int nc = n < p.N ? n : p.N - 1;
        if constexpr (GEN) {
          if (p.bias) x += p.bias[nc];
          if (p.act == KLAB_ACT_RELU) x = fmaxf(x, 0.f);
          else if (p.act == KLAB_ACT_GELU) x = gelu_for<T>(x);
          if (p.aux) {
            const float a = to_f32(reinterpret_cast<const T*>(p.aux)[(long)mc * p.ldaux + nc]);
            if (p.aux_mode == KLAB_AUX_NONZERO) x = (a != 0.f) ? x * p.aux_scale : 0.f;
            else if (p.aux_mode == KLAB_AUX_DGELU) x *= gelu_erf_grad(a);
          }
          x *= drop_mult(dc, (uint64_t)mc * (uint64_t)p.N + (uint64_t)nc);
          if (p.residual) {
            x += p.r_f32 ? reinterpret_cast<const float*>(p.residual)[(long)mc * p.ldr + nc]
                         : to_f32(reinterpret_cast<const T*>(p.residual)[(long)mc * p.ldr + nc]);
          }
        } else {
          if constexpr (FLAGS & EF_BIAS) x += p.bias[nc];
          if constexpr (FLAGS & EF_RELU) x = fmaxf(x, 0.f);
          if constexpr (FLAGS & EF_GELU) x = gelu_for<T>(x);
          if constexpr (FLAGS & EF_AUXNZ) {
            const float a = to_f32(reinterpret_cast<const T*>(p.aux)[(long)mc * p.ldaux + nc]);
            x = (a != 0.f) ? x * p.aux_scale : 0.f;
          }
          if constexpr (FLAGS & EF_AUXNZ_CO) x *= p.aux_scale;  // the zero mask itself is applied at copy-out
          if constexpr (FLAGS & EF_DGELU) x *= gelu_erf_grad(to_f32(reinterpret_cast<const T*>(p.aux)[(long)mc * p.ldaux + nc]));
          // (the dedicated variants are only selected for M * N < 2^32 -- fill_gemmp -- where the one-round, branch-free hash equals
          // drop_mult's; 64 copies of its two branches otherwise chop the unrolled epilogue into basic blocks)
          if constexpr (FLAGS & EF_DROP) x *= drop_mult32_nb(dc, (uint32_t)mc * (uint32_t)p.N + (uint32_t)nc);
          if constexpr (FLAGS & EF_RES) {
            x += p.r_f32 ? reinterpret_cast<const float*>(p.residual)[(long)mc * p.ldr + nc]
                         : to_f32(reinterpret_cast<const T*>(p.residual)[(long)mc * p.ldr + nc]);
          }
        }
        v[r] = x;
      }
      char* dst = smem + ml * pitchB;
      if (f32out) *reinterpret_cast<f32x4*>(dst + nl * 4) = f32x4{v[0], v[1], v[2], v[3]};
      else *reinterpret_cast<bf16x4*>(dst + nl * 2) = bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
    }
  }
}

}  // namespace klab
