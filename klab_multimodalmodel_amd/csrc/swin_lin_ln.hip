// Frozen Swin-V2 tower, wide stages: Linear + bias + LayerNorm + residual in ONE launch
//     out = shortcut + LayerNorm(x W^T + b) * gamma + beta                                  (HF/swinv2:496-506 / 555-563 + 697-702)
// for the attention output projection (K = C) and the MLP's second Linear (K = 4C) of a block whose width is C = 256 (stage 2 of
// the caption tower: 12 544 rows at B = 64).  It replaces klab_gemm + klab_layernorm_fwd (8.8 + 9.3 us and 17.5 + 9.3 us per block
// in the step, plus a launch boundary each): the Linear's output never goes to memory, the norm's row statistics come from the
// accumulators.  The narrow stages (C = 64 / 128) have their own fused kernels (swin_mlp.hip).
//
// A workgroup owns 32 rows x all 256 columns, so a row's statistics are complete inside the workgroup: four waves of 32 rows x 64
// columns (8 accumulator tiles each), row sums reduced in-lane, across the lane groups (two shuffles) and across the four waves
// through LDS.  Operands stream through a 4-slot LDS-DMA ring in k-tiles of 32 (x: 32 rows x 64 B, W: 256 rows x 64 B per slot,
// three k-tiles in flight; 16-byte chunk c of row r at position c ^ ((r >> 2) & 3) -- the swizzle on the DMA's source address, as
// in mm8p.hip -- which makes the 16 rows of a fragment read hit 16 distinct bank groups); plain HIP, compiler-scheduled, one
// counted wait + one raw barrier per k-tile.  72 KiB of LDS and < 128 VGPRs: TWO workgroups per CU, so one's epilogue (statistics,
// three barriers, the shortcut's HBM latency) hides behind the other's main loop -- the first version (64 rows, 120 KiB, one
// workgroup per CU, 196 workgroups on 256 CUs) was a single latency chain: 14.5 / 22.9 us for K = 256 / 1024 against 18.6 / 25.4
// for the two launches (tools/linln_bench.py).  The finished tile is parked in LDS (f32, normalised) and streamed out as whole
// rows: gamma, beta and the shortcut meet it there, each thread owning one 16-byte column chunk.
#include <stdlib.h>

#include "common.h"
#include "klab_mm.h"

namespace klab {
namespace {

struct LinLnP {
  const bf16_t* x; const float* shortcut; const bf16_t* w; const float* bias; const float* gamma; const float* beta;
  float* out; bf16_t* outt;
  int M, K; float eps;
};

constexpr int LL_S = 4;

template <int N> __device__ __forceinline__ void ll_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int C, int BM>
__global__ __launch_bounds__(256, 2) void swin_linear_ln_fused_kernel(LinLnP p) {
  constexpr int S = LL_S, ABYTES = BM * 64, BBYTES = C * 64, SLOT = ABYTES + BBYTES;
  constexpr int LB = C / 64;               // W: LDS-DMA instructions per wave per k-tile (16 rows x 64 B each); x: one, waves 0 .. AW - 1
  constexpr int AW = BM / 16, RPT = BM / 4;  // (rows per thread at copy-out)
  constexpr int WN = C / 4, NI = WN / 16;  // a wave's columns, its 16-column tiles
  constexpr int MI = BM / 16;
  constexpr int PITCH = (C + 4) * 4;       // staged f32 rows
  constexpr int CPR = C / 4;               // 16-byte f32 chunks per row
  static_assert(BM * PITCH + 2 * 4 * BM * 4 <= S * SLOT, "the staged tile and the row statistics alias the ring");
  static_assert(CPR == 64 && (BM == 32 || BM == 64), "copy-out: thread t owns column chunk t & 63 of rows (t >> 6) + 4 i");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave * WN;
  const long bm0 = (long)blockIdx.x * BM;
  const int nt = p.K / 32;

  const int srow = lane >> 2, schunk = (lane & 3) ^ ((lane >> 4) & 3);  // (row >> 2) & 3 == (lane >> 4) & 3: rows start at a multiple of 16
  const bf16_t* asrc;
  const bf16_t* bsrc[LB];
  {
    long gr = bm0 + (wave & (AW - 1)) * 16 + srow;
    gr = gr < p.M ? gr : p.M - 1;  // rows past the edge are computed on a copy of the last row and never stored
    asrc = p.x + gr * p.K + schunk * 8;
  }
#pragma unroll
  for (int i = 0; i < LB; ++i) bsrc[i] = p.w + (long)((wave * LB + i) * 16 + srow) * p.K + schunk * 8;
  auto issue = [&](int t) {
    if (t >= nt) return;  // wave-uniform
    char* st = smem + (t % S) * SLOT;
    if (wave < AW)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc + (long)t * 32),
                                       (__attribute__((address_space(3))) void*)(st + wave * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < LB; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bsrc[i] + (long)t * 32),
                                       (__attribute__((address_space(3))) void*)(st + ABYTES + (wave * LB + i) * 1024), 16, 0, 0);
  };
  auto frag = [&](const char* img, int f) {  // rows 16 f .. + 15, k = 8 g .. + 7 of the k-tile
    const int r = f * 16 + (lane & 15);
    return *reinterpret_cast<const bf16x8*>(img + r * 64 + ((g ^ ((r >> 2) & 3)) * 16));
  };

  // this thread's column chunk at copy-out; this lane's bias columns
  const int cc = tid & (CPR - 1);
  const f32x4 gm = *reinterpret_cast<const f32x4*>(p.gamma + cc * 4), bt = *reinterpret_cast<const f32x4*>(p.beta + cc * 4);
  f32x4 bb[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) bb[j] = *reinterpret_cast<const f32x4*>(p.bias + wn + j * 16 + g * 4);

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  issue(0);
  issue(1);
  issue(2);
  for (int t = 0; t < nt; ++t) {
    // k-tile t has landed (this wave's part; the barrier covers the others); up to two younger ones stay in flight
    const int young = nt - 1 - t;
    if (wave < AW) {
      if (young >= 2) ll_wait<2 * (LB + 1)>(); else if (young == 1) ll_wait<LB + 1>(); else ll_wait<0>();
    } else {
      if (young >= 2) ll_wait<2 * LB>(); else if (young == 1) ll_wait<LB>(); else ll_wait<0>();
    }
    __builtin_amdgcn_s_barrier();  // ... and all waves are past their reads of k-tile t - 1
    const char* As = smem + (t % S) * SLOT;
    const char* Bs = As + ABYTES;
    issue(t + 3);  // into the slot k-tile t - 1 was read from
    bf16x8 af[MI], bf[NI];
#pragma unroll
    for (int i = 0; i < MI; ++i) af[i] = frag(As, i);
#pragma unroll
    for (int j = 0; j < NI; ++j) bf[j] = frag(Bs, (wn >> 4) + j);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)  // W as the A operand: a lane ends up with 4 consecutive columns of row 16 i + (lane & 15)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j], af[i], acc[i][j], 0, 0, 0);
  }
  // the shortcut's rows start their trip now; they are needed after the statistics
  f32x4 sc[RPT];
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    long m = bm0 + (tid >> 6) + 4 * i;
    m = m < p.M ? m : p.M - 1;
    sc[i] = *reinterpret_cast<const f32x4*>(p.shortcut + m * C + cc * 4);
  }
  __syncthreads();  // the ring is free: it becomes the staged tile + the row statistics

  // ---- + bias (rounded to bf16 as the two-launch path stored it), LayerNorm over the 256 columns of each row ----
  float* red = reinterpret_cast<float*>(smem + BM * PITCH);  // [2][4 waves][BM rows]
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) { acc[i][j][r] = (float)(bf16_t)(acc[i][j][r] + bb[j][r]); s += acc[i][j][r]; }
    s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
    if (g == 0) red[wave * BM + i * 16 + (lane & 15)] = s;
  }
  __syncthreads();
  float mean[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int row = i * 16 + (lane & 15);
    mean[i] = (red[0 * BM + row] + red[1 * BM + row] + red[2 * BM + row] + red[3 * BM + row]) * (1.f / C);
    float v = 0.f;
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float d = acc[i][j][r] - mean[i]; v += d * d; }
    v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
    if (g == 0) red[(4 + wave) * BM + row] = v;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int row = i * 16 + (lane & 15);
    const float var = (red[4 * BM + row] + red[5 * BM + row] + red[6 * BM + row] + red[7 * BM + row]) * (1.f / C);
    const float rstd = rsqrtf(var + p.eps);
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      f32x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = (acc[i][j][r] - mean[i]) * rstd;
      *reinterpret_cast<f32x4*>(smem + row * PITCH + (wn + j * 16 + g * 4) * 4) = o;
    }
  }
  __syncthreads();
  // ---- whole rows out: * gamma + beta + shortcut, f32 and bf16 copies ----
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const int row = (tid >> 6) + 4 * i;
    const long m = bm0 + row;
    if (m >= p.M) continue;
    f32x4 v = *reinterpret_cast<const f32x4*>(smem + row * PITCH + cc * 16);
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = sc[i][r] + (v[r] * gm[r] + bt[r]);
    *reinterpret_cast<f32x4*>(p.out + m * C + cc * 4) = v;
    if (p.outt) *reinterpret_cast<bf16x4*>(p.outt + m * C + cc * 4) = bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
  }
}

}  // namespace
}  // namespace klab

// x [M, K] bf16 (row pitch K), w [C, K] bf16, bias / gamma / beta [C] f32, shortcut [M, C] f32; out [M, C] f32, outt bf16 (optional).
// bf16, C = 256, K % 32 == 0, K >= 128; otherwise KLAB_ERR_UNSUPPORTED (caller: klab_gemm + klab_layernorm_fwd).
extern "C" int klab_swin_linear_ln_fused(const void* x, const float* shortcut, const void* w, const float* bias, const float* gamma,
                                         const float* beta, float* out, void* outt, int dtype, int M, int K, int C, float eps, void* stream) {
  using namespace klab;
  if (!x || !shortcut || !w || !bias || !gamma || !beta || !out) return KLAB_ERR_BADARG;
  if (dtype != KLAB_BF16 || C != 256 || (K & 31) || K < 128 || ((uintptr_t)x & 15) || ((uintptr_t)w & 15)) return KLAB_ERR_UNSUPPORTED;
  if (M <= 0) return KLAB_OK;
  LinLnP p{(const bf16_t*)x, shortcut, (const bf16_t*)w, bias, gamma, beta, out, (bf16_t*)outt, M, K, eps};
  // K = C (the attention projection): 32-row workgroups, two per CU, 392 of them at M = 12 544.  K = 4C (fc2): 64 rows -- W's 256 rows
  // are re-read once per workgroup, and at 32 rows that operand traffic (231 MB per launch) alone costs what the two launches did.
#define KLAB_LL(BM_)                                                                                                              \
  {                                                                                                                               \
    const size_t lds = (size_t)LL_S * (BM_ * 64 + 256 * 64);                                                                      \
    const int rc = ensure_dyn_lds(reinterpret_cast<const void*>(swin_linear_ln_fused_kernel<256, BM_>), lds);                     \
    if (rc) return rc;                                                                                                            \
    hipLaunchKernelGGL((swin_linear_ln_fused_kernel<256, BM_>), dim3((unsigned)((M + BM_ - 1) / BM_)), dim3(256), lds, (hipStream_t)stream, p); \
  }
  if (K >= 512) KLAB_LL(64) else KLAB_LL(32)
#undef KLAB_LL
  KLAB_LAUNCH_CHECK();
  return KLAB_OK;
}
