// mmf8: the fp8 forward GEMM of BASELINE configs[4] ("fp8 MFMA path") on gfx950's BLOCK-SCALED matrix instruction.
//   C[M,N] = epilogue(alpha * sa[m] * sb[n] * sum_k A8(m,k) B8(n,k)),  A8 / B8 = OCP e4m3 bytes, both K-major (klab_gemm_fp8).
// Every forward nn.Linear of the path in fp8 mode (HF/t5:83-94,206-209,1047; HF/swinv2:322,384-386,499,539,555).
//
// Why a second fp8 kernel.  gemm.hip's gemm_glds_fp8_kernel issues v_mfma_f32_16x16x32_fp8_fp8, which takes the cycles of the
// bf16 instruction of the same shape (MI355X_MICROARCH.md, Matrix cores): fp8 then only halves the operand bytes.  The form
// that runs at 2x the bf16 rate is v_mfma_scale_f32_16x16x128_f8f6f4 -- 128 k per instruction at twice the cycles of the bf16
// 16x16x32.  Its block scales (one e8m0 exponent per 32 k) are set to 2^0 here: the quantisation scheme of the path is one
// fp32 scale per ROW of each operand (per token / per output channel; klab_quant_fp8_rows, klab_quant_fp8_arena), applied to
// the fp32 accumulator in the epilogue exactly as before, so results equal the non-scaled kernel's up to summation order.
// Operand mapping: whatever k positions the instruction assigns to (lane group, byte) are the same for A and B, and both
// fragments are filled the same way (the 32 consecutive bytes 32 g .. 32 g + 31 of the row's 128-byte k-tile, g = lane >> 4):
// every k of the tile meets its partner exactly once.
//
// Structure (plain HIP, compiler-scheduled): 128 x 128 tile, four waves of 64 x 64, BK = 128 bytes; LDS ring of 3 stages x
// (16 KiB A + 16 KiB B) filled by LDS-DMA two k-tiles ahead; per k-tile one counted wait + barrier, 16 fragment reads
// (ds_read_b128), one barrier, the refill of the slot just read, 16 scaled MFMAs.  K-major image rows are 128 B; 16-byte chunk
// c of row r sits at position c ^ (r & 7) (swizzle on the LDS-DMA SOURCE address, as in mm8p.hip): conflict-free b128 reads.
#include <stdlib.h>

#include "gemm_shared.h"

namespace klab {
namespace f8 {

constexpr int BK = 128, NT = 256, S = 3;
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int ROWS>
struct Operand {
  static constexpr int L = ROWS / 32;  // LDS-DMA instructions per wave per stage: ROWS rows x 128 B = ROWS/8 KiB over 4 waves
  const char* src[L];
  __device__ __forceinline__ void init(const char* base, long ld, int row0, int nrows, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < L; ++i) {
      const int r = (wave * L + i) * 8 + (lane >> 3);  // an instruction moves 8 rows x 128 B
      const int c = (lane & 7) ^ (lane >> 3);          // (r & 7) == lane >> 3
      int gr = row0 + r;
      gr = gr < nrows ? gr : nrows - 1;
      src[i] = base + (long)gr * ld + c * 16;
    }
  }
  __device__ __forceinline__ void issue(long kt, char* stage, int wave) const {
#pragma unroll
    for (int i = 0; i < L; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + kt * BK),
                                       (__attribute__((address_space(3))) void*)(stage + (wave * L + i) * 1024), 16, 0, 0);
  }
};

// the 32 bytes k = 32 g .. 32 g + 31 of row 16 f + (lane & 15)
__device__ __forceinline__ i32x8 frag(const char* img, int f, int lane) {
  const int r = f * 16 + (lane & 15), g = lane >> 4, x = lane & 7;
  const i32x4 lo = *reinterpret_cast<const i32x4*>(img + r * 128 + (((2 * g) ^ x) * 16));
  const i32x4 hi = *reinterpret_cast<const i32x4*>(img + r * 128 + (((2 * g + 1) ^ x) * 16));
  return i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

struct Scales { const float* sa; const float* sb; long sb_stride; };

template <int BM, int BN>
__global__ __launch_bounds__(256) void mmf8_kernel(GemmP p, Scales sc) {
  constexpr int WTM = BM / 2, WTN = BN / 2, MI = WTM / 16, NI = WTN / 16;
  constexpr int ABYTES = BM * 128, STAGE = (BM + BN) * 128;
  constexpr int LPS = Operand<BM>::L + Operand<BN>::L;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = (wave >> 1) * WTM, wn = (wave & 1) * WTN;
  int bm0, bn0;
  tile_of_block(p, BM, BN, blockIdx.x, bm0, bn0);
  const int nt = p.K / BK;

  Operand<BM> oa;
  Operand<BN> ob;
  oa.init(reinterpret_cast<const char*>(p.A), p.lda, bm0, p.M, wave, lane);
  ob.init(reinterpret_cast<const char*>(p.B), p.ldb, bn0, p.N, wave, lane);
  auto issue = [&](int t) {
    if (t >= nt) return;  // wave-uniform
    char* st = smem + (t % S) * STAGE;
    oa.issue(t, st, wave);
    ob.issue(t, st + ABYTES, wave);
  };

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  issue(0);
  issue(1);
  const int unit = 0x7F7F7F7F;  // e8m0 block scales: 2^0 in every byte
  for (int t = 0; t < nt; ++t) {
    // k-tile t has landed (this wave's part; the barrier covers the others); t + 1 may stay in flight
    if (t + 1 < nt) wait_vm<LPS>(); else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    const char* As = smem + (t % S) * STAGE;
    const char* Bs = As + ABYTES;
    i32x8 af[MI], bf[NI];
#pragma unroll
    for (int i = 0; i < MI; ++i) af[i] = frag(As, (wm >> 4) + i, lane);
#pragma unroll
    for (int j = 0; j < NI; ++j) bf[j] = frag(Bs, (wn >> 4) + j, lane);
    // slot (t + 2) % S was read during k-tile t - 1: every wave is past those reads (the barrier above), so it may be refilled
    issue(t + 2);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)  // B as the instruction's A operand: a lane ends up with 4 consecutive n of row m = lane & 15
        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(bf[j], af[i], acc[i][j], 0, 0, 0, unit, 0, unit);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  }

  float alpha = p.alpha;
  if (p.alpha_dev) alpha *= p.alpha_dev[0];
  // dequantise: lane owns m = ... + (lane & 15), n = ... + (lane >> 4) * 4 + r  (the bf16 kernels' accumulator layout)
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int m = bm0 + wm + i * 16 + (lane & 15);
    const float sam = sc.sa[m < p.M ? m : p.M - 1];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int n0 = bn0 + wn + j * 16 + (lane >> 4) * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + r < p.N ? n0 + r : p.N - 1;
        acc[i][j][r] *= sam * sc.sb[(long)n * sc.sb_stride];
      }
    }
  }
  __syncthreads();  // all LDS-DMA retired (vmcnt(0) in the last iteration) and all fragment reads done: LDS is free
  staged_epilogue<bf16_t, BM, BN, MI, NI>(p, acc, alpha, smem, bm0, bn0, wm, wn, tid, lane);
}

}  // namespace f8

// host side: KLAB_ERR_UNSUPPORTED when the shape is left to the non-scaled kernels (K % 128 != 0, tiny operands)
int mmf8_try(const GemmP& pin, const float* sa, const float* sb, long sb_stride, int force, hipStream_t s) {
  using namespace f8;
  // Measured on BASELINE configs[4]'s forward shapes (tools/fp8_bench.py, profiles/r03_fp8_scaled_vs_nonscaled.txt): this kernel
  // is correct and SLOWER than gemm_glds_fp8_kernel (qkv 40.8 vs 33.2 us, wi 50.4 vs 41.2, 4096^3 104.8 vs 94.2) although its
  // matrix instruction runs at twice the rate -- every tiled kernel of this library, bf16 or fp8, 128- or 256-wide, sits at
  // 10-11.5 TB/s of L2 -> LDS operand traffic (1 GB per 4096^3 launch in 94-110 us), so a 128 x 128 tile is bound by its operand
  // bytes, not by the MFMA rate.  Hence opt-in: KLAB_FP8_SCALED=1, or klab_gemm_args.name_tag = 2 per call (tests).
  static const bool on = [] { const char* e = getenv("KLAB_FP8_SCALED"); return e && atoi(e) != 0; }();
  if (force < 0 || (!on && force <= 0)) return KLAB_ERR_UNSUPPORTED;
  GemmP p = pin;
  if ((p.K % BK) || p.K < BK || p.M < 16 || p.N < 16) return KLAB_ERR_UNSUPPORTED;
  p.splits = 1;
  auto tiles = [&](int bm, int bn) { return (long)((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn); };
  Scales sc{sa, sb, sb_stride};
#define KLAB_F8(BM_, BN_)                                                                                         \
  {                                                                                                               \
    size_t lds = (size_t)S * (BM_ + BN_) * 128;                                                                   \
    const size_t epi = (size_t)epilogue_lds_bytes<BM_, BN_>(p.c_f32);                                             \
    if (epi > lds) lds = epi;                                                                                     \
    int rc = ensure_dyn_lds(reinterpret_cast<const void*>(mmf8_kernel<BM_, BN_>), lds);                           \
    if (rc) return rc;                                                                                            \
    hipLaunchKernelGGL((mmf8_kernel<BM_, BN_>), dim3((unsigned)tiles(BM_, BN_)), dim3(NT), lds, s, p, sc);        \
    KLAB_LAUNCH_CHECK();                                                                                          \
    return KLAB_OK;                                                                                               \
  }
  if (tiles(128, 128) >= 240) KLAB_F8(128, 128)
  if (tiles(128, 64) >= 240) KLAB_F8(128, 64)
  KLAB_F8(64, 64)
#undef KLAB_F8
}

}  // namespace klab
