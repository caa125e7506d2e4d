// Pieces of the GEMM family shared by gemm.hip (LDS-DMA ring, 4 waves) and mm8p.hip (256x256 eight-wave tiles): the kernel
// parameter block, the XCD-aware workgroup -> tile map and the element-wise epilogue applied in the MFMA accumulator layout.
#pragma once
#include "common.h"
#include "klab_mm.h"
#include <type_traits>
#include <utility>

#include <hip/hip_ext.h>

namespace klab {

// Measurement hook of the family probe (klab_gemm_probe_*): when the calling thread has armed a pair of events, the NEXT tile-kernel
// launch carries them as its own start / stop events (hipExtLaunchKernelGGL: the dispatch's begin / end timestamps, no extra
// packets on the stream -- a hipEventRecord pair around a launch added several microseconds of barrier handling per kernel).
struct LaunchProbe { hipEvent_t a = nullptr, b = nullptr; };
inline thread_local LaunchProbe tl_launch_probe;
template <typename K, typename... Args>
inline void probed_launch(K kern, dim3 grid, dim3 block, size_t lds, hipStream_t s, Args... args) {
  if (tl_launch_probe.a) {
    hipExtLaunchKernelGGL(kern, grid, block, (unsigned)lds, s, tl_launch_probe.a, tl_launch_probe.b, 0, args...);
    tl_launch_probe.a = nullptr;
  } else {
    hipLaunchKernelGGL(kern, grid, block, lds, s, args...);
  }
}

// set by the engine around a klab_gemm_grouped call whose list spans several layers: take the 256 x 256 grouped kernel (mm8p.hip)
inline thread_local bool tl_grouped_large_tiles = false;

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

template <typename T, int BM, int BN, int CO = 0, int NT = 256>
__device__ __forceinline__ void copy_out_tile(const GemmP& p, const char* smem, int bm0, int bn0, int tid) {
  const bool f32out = p.c_f32 || sizeof(T) == 4;
  const int pitchB = f32out ? (BN + 4) * 4 : (BN + 8) * 2;
  const int esz = f32out ? 4 : 2;
  const int sh = f32out ? 2 : 3;            // log2(elements per 16-B chunk)
  const int cpr_sh = (BN == 128 ? 7 : 6) - sh;  // log2(chunks per tile row)
  const bool vec_ok = ((p.ldc * esz) & 15) == 0 && (p.N & ((1 << sh) - 1)) == 0;
  char* Cb = reinterpret_cast<char*>(p.C);
  const int nch = BM << cpr_sh;
  // Interior tiles (the common case) with whole 16-byte rows: every chunk of the thread is read from LDS -- and its residual /
  // mask operand requested from memory -- BEFORE the first store.  The general loop below does read, (load,) store per chunk; on
  // this ISA a store counts in vmcnt like a load, so each chunk's operand wait also drained the stores issued before it: 8-16
  // memory round trips in a row at the end of every tile.
  if (bm0 + BM <= p.M && bn0 + BN <= p.N && vec_ok && (!p.accumulate || (f32out && CO == 0)) && !(p.ablate & 64)) {
    auto fast = [&](auto esz_c) {
      constexpr int ESZ = decltype(esz_c)::value, EPC = 16 / ESZ, CPR = BN / EPC, PER = BM * CPR / NT;
      static_assert(BM * CPR % NT == 0, "whole chunks per thread");
      f32x4 val[PER], opnd[PER];
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        const int ch = tid + i * NT, row = ch / CPR, cc = ch % CPR;
        val[i] = *reinterpret_cast<const f32x4*>(smem + row * ((BN + 16 / ESZ) * ESZ) + cc * 16);
        const long m = bm0 + row, n = bn0 + cc * EPC;
        if constexpr (CO == 0 && ESZ == 4) { if (p.accumulate) opnd[i] = *reinterpret_cast<const f32x4*>(Cb + (m * p.ldc + n) * ESZ); }  // C += tile
        if constexpr ((CO & EF_RES_CO) != 0) opnd[i] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.residual) + m * p.ldr + n);
        if constexpr ((CO & (EF_AUXNZ_CO | EF_DGELU_CO)) != 0) opnd[i] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const bf16_t*>(p.aux) + m * p.ldaux + n);
      }
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        const int ch = tid + i * NT, row = ch / CPR, cc = ch % CPR;
        const long m = bm0 + row, n = bn0 + cc * EPC;
        f32x4 v = val[i];
        if constexpr (CO == 0 && ESZ == 4) { if (p.accumulate) { v[0] += opnd[i][0]; v[1] += opnd[i][1]; v[2] += opnd[i][2]; v[3] += opnd[i][3]; } }
        if constexpr ((CO & EF_RES_CO) != 0) { v[0] += opnd[i][0]; v[1] += opnd[i][1]; v[2] += opnd[i][2]; v[3] += opnd[i][3]; }
        if constexpr ((CO & EF_AUXNZ_CO) != 0) {
          bf16x8 nv = __builtin_bit_cast(bf16x8, v);
          const bf16x8 a = __builtin_bit_cast(bf16x8, opnd[i]);
#pragma unroll
          for (int u = 0; u < 8; ++u) nv[u] = ((float)a[u] != 0.f) ? nv[u] : (bf16_t)0.f;
          v = __builtin_bit_cast(f32x4, nv);
        }
        if constexpr ((CO & EF_DGELU_CO) != 0) {
          bf16x8 nv = __builtin_bit_cast(bf16x8, v);
          const bf16x8 a = __builtin_bit_cast(bf16x8, opnd[i]);
#pragma unroll
          for (int u = 0; u < 8; ++u) nv[u] = (bf16_t)((float)nv[u] * gelu_erf_grad((float)a[u]));
          v = __builtin_bit_cast(f32x4, nv);
        }
        *reinterpret_cast<f32x4*>(Cb + (m * p.ldc + n) * ESZ) = v;
      }
    };
    if (f32out) fast(std::integral_constant<int, 4>{});
    else fast(std::integral_constant<int, 2>{});
    return;
  }
  for (int ch = tid; ch < nch; ch += NT) {
    const int row = ch >> cpr_sh, cc = ch & ((1 << cpr_sh) - 1);
    const int m = bm0 + row, n = bn0 + (cc << sh);
    if (m >= p.M || n >= p.N) continue;
    const char* src = smem + row * pitchB + cc * 16;
    char* dst = Cb + ((long)m * p.ldc + n) * esz;
    if constexpr (CO & EF_AUXNZ_CO) {  // bf16 tile, bf16 mask source, vector path guaranteed by the host
      bf16x8 nv = *reinterpret_cast<const bf16x8*>(src);
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16_t*>(p.aux) + (long)m * p.ldaux + n);
#pragma unroll
      for (int u = 0; u < 8; ++u) nv[u] = ((float)a[u] != 0.f) ? nv[u] : (bf16_t)0.f;
      *reinterpret_cast<bf16x8*>(dst) = nv;
      continue;
    }
    if constexpr (CO & EF_DGELU_CO) {  // bf16 tile, bf16 pre-activation, vector path guaranteed by the host
      bf16x8 nv = *reinterpret_cast<const bf16x8*>(src);
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16_t*>(p.aux) + (long)m * p.ldaux + n);
#pragma unroll
      for (int u = 0; u < 8; ++u) nv[u] = (bf16_t)((float)nv[u] * gelu_erf_grad((float)a[u]));
      *reinterpret_cast<bf16x8*>(dst) = nv;
      continue;
    }
    if constexpr (CO & EF_RES_CO) {    // f32 tile + f32 residual, vector path guaranteed by the host
      f32x4 val = *reinterpret_cast<const f32x4*>(src);
      const f32x4 r = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.residual) + (long)m * p.ldr + n);
      val[0] += r[0]; val[1] += r[1]; val[2] += r[2]; val[3] += r[3];
      *reinterpret_cast<f32x4*>(dst) = val;
      continue;
    }
    if (vec_ok) {
      f32x4 val = *reinterpret_cast<const f32x4*>(src);
      if (p.accumulate) {
        if (f32out) {
          const f32x4 old = *reinterpret_cast<const f32x4*>(dst);
          val[0] += old[0]; val[1] += old[1]; val[2] += old[2]; val[3] += old[3];
        } else {
          bf16x8 nv = *reinterpret_cast<const bf16x8*>(src);
          const bf16x8 old = *reinterpret_cast<const bf16x8*>(dst);
#pragma unroll
          for (int u = 0; u < 8; ++u) nv[u] = (bf16_t)((float)nv[u] + (float)old[u]);
          *reinterpret_cast<bf16x8*>(dst) = nv;
          continue;
        }
      }
      if (p.ablate & 64) {  // experiment: write-through (sc1) stores -- nothing left dirty in L2 for the end-of-kernel release
        asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(val) : "memory");
      } else {
        *reinterpret_cast<f32x4*>(dst) = val;
      }
    } else {
      const int epc = 1 << sh;
      const int nv = (p.N - n) < epc ? (p.N - n) : epc;
      for (int u = 0; u < nv; ++u) {
        if (f32out) {
          float x = reinterpret_cast<const float*>(src)[u];
          float* d = reinterpret_cast<float*>(dst) + u;
          *d = p.accumulate ? x + *d : x;
        } else {
          float x = (float)reinterpret_cast<const bf16_t*>(src)[u];
          bf16_t* d = reinterpret_cast<bf16_t*>(dst) + u;
          *d = (bf16_t)(p.accumulate ? x + (float)*d : x);
        }
      }
    }
  }
}

template <typename T, int BM, int BN, int MI, int NI, int NT = 256>
__device__ __forceinline__ void staged_epilogue(const GemmP& p, f32x4 (&acc)[MI][NI], float alpha, char* smem, int bm0, int bn0, int wm,
                                                int wn, int tid, int lane) {
#define KLAB_EPI(F) staged_epilogue_v<T, BM, BN, MI, NI, F>(p, acc, alpha, smem, bm0, bn0, wm, wn, tid, lane)
  switch (p.epi) {  // wave-uniform: only the selected variant's instructions are ever fetched
    case 0: KLAB_EPI(0); break;
    case EF_BIAS: KLAB_EPI(EF_BIAS); break;
    case EF_BIAS | EF_GELU: KLAB_EPI(EF_BIAS | EF_GELU); break;
    case EF_RELU: KLAB_EPI(EF_RELU); break;
    case EF_RELU | EF_DROP: KLAB_EPI(EF_RELU | EF_DROP); break;
    case EF_RES: KLAB_EPI(EF_RES); break;
    case EF_DROP | EF_RES: KLAB_EPI(EF_DROP | EF_RES); break;
    case EF_AUXNZ: KLAB_EPI(EF_AUXNZ); break;
    case EF_DGELU: KLAB_EPI(EF_DGELU); break;
    case EF_AUXNZ_CO: KLAB_EPI(EF_AUXNZ_CO); break;
    case EF_DGELU_CO: KLAB_EPI(0); break;
    case EF_RES_CO: KLAB_EPI(0); break;
    case EF_DROP | EF_RES_CO: KLAB_EPI(EF_DROP); break;
    default: KLAB_EPI(EF_GENERIC); break;
  }
#undef KLAB_EPI
  __syncthreads();
  if (p.ablate & 8) return;
  if constexpr (sizeof(T) == 2) {
    if (p.epi == EF_AUXNZ_CO) { copy_out_tile<T, BM, BN, EF_AUXNZ_CO, NT>(p, smem, bm0, bn0, tid); return; }
    if (p.epi == EF_DGELU_CO) { copy_out_tile<T, BM, BN, EF_DGELU_CO, NT>(p, smem, bm0, bn0, tid); return; }
  }
  if (p.epi == EF_RES_CO || p.epi == (EF_DROP | EF_RES_CO)) { copy_out_tile<T, BM, BN, EF_RES_CO, NT>(p, smem, bm0, bn0, tid); return; }
  copy_out_tile<T, BM, BN, 0, NT>(p, smem, bm0, bn0, tid);
}
template <int BM, int BN> constexpr int epilogue_lds_bytes(bool f32out) { return f32out ? BM * (BN + 4) * 4 : BM * (BN + 8) * 2; }

}  // namespace klab
