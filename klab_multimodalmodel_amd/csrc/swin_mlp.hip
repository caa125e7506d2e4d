// Fused Swin-V2 MLP half of a block for a FROZEN tower (no activation is kept for a backward pass):
//     out = shortcut + LayerNorm( fc2( GELU( fc1(x) + b1 ) ) + b2 ) * gamma + beta        (HF/swinv2:539-563, 697-702)
// One workgroup owns 128 token rows; the hidden layer (4C wide) never leaves the chip.  In the narrow early stages
// (C = 64 / 128, 200 K / 50 K rows at B = 64) the three kernels this replaces -- fc1+GELU, fc2, LayerNorm -- are HBM-bound on
// the hidden activations they pass through memory (103 MB written and read back per block in stage 0).
//
// Layout of the work inside a wave (32 rows = 2 row tiles of 16), all with v_mfma_f32_16x16x32_bf16:
//   fc1, operands swapped:  D1[n][m] = sum_k W1[n][k] x[m][k]      -> a lane holds 4 consecutive hidden units n of row m = lane&15
//   GELU in registers; two 16-unit tiles packed = 8 bf16 per lane = the B operand of the next MFMA, in the permuted
//   k order kappa(g, j) = 16 (j >> 2) + 4 g + (j & 3)  (g = lane >> 4), so the hidden tile needs no LDS round trip
//   fc2, operands swapped:  D2[c][m] = sum_n W2[c][n] h[m][n]      -> a lane holds 4 consecutive channels c of row m
//   LayerNorm over c: in-lane + two cross-lane-group shuffles; residual add; fp32 + bf16 stores.
// The weights stream through LDS in chunks of 64 hidden units (W1 rows / W2 columns) by LDS-DMA, S slots deep.
#include <stdlib.h>

#include "common.h"
#include "klab_mm.h"

namespace klab {
namespace {

template <int N> __device__ __forceinline__ void mlp_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// LDS reads as inline asm: a compiler-visible ds_read next to in-flight LDS-DMA makes hipcc drain vmcnt to 0 first (it
// cannot tell the ring slots apart), which serialised the weight stream.  The matching wait names the loaded registers
// as read-write operands, so every consumer (a builtin MFMA) is ordered behind it by data dependence.
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
__device__ __forceinline__ u32x4 mlp_lds_b128(unsigned addr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
__device__ __forceinline__ u32x2 mlp_lds_b64(unsigned addr) {
  u32x2 v;
  asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}

struct MlpP {
  const bf16_t* x; const float* shortcut;
  const bf16_t* w1; const float* b1; const bf16_t* w2; const float* b2;
  const float* gamma; const float* beta;
  float* out; bf16_t* outt;
  int M; float eps;
};

// K-major LDS image of ROWS x 32 k (64-B rows), 16-B chunk c of row r at position c ^ 2*((r>>2)&1) (as in gemm.hip)
__device__ __forceinline__ int img_off(int r, int chunk) { return r * 64 + ((chunk ^ (((r >> 2) & 1) << 1)) * 16); }

// PROJ: the attention half's tail instead -- out = shortcut + LayerNorm(x Wp^T + bp) * gamma + beta (HF/swinv2:496-506,
// 697-700): the "hidden" layer IS the output (width C, no GELU, no second GEMM); w1/b1 carry the projection.
template <int C, int S, bool PROJ = false>
__global__ __launch_bounds__(256) void swin_mlp_fused_kernel(MlpP p) {
  constexpr int HD = PROJ ? C : 4 * C, HC = 64, NCH = HD / HC;  // hidden width, hidden units per chunk, chunks
  constexpr int KB1 = C / 32;                             // 32-deep k blocks of fc1
  constexpr int CT = C / 16;                              // 16-channel output tiles of fc2
  constexpr int W1B = KB1 * HC * 64;                      // bytes of a W1 chunk: KB1 images of [64 rows x 32 k]
  constexpr int W2B = PROJ ? 0 : 2 * C * 64;              // bytes of a W2 chunk: 2 images of [C rows x 32 k]
  constexpr int SLOT = W1B + W2B;
  constexpr int DMA1 = KB1, DMA2 = PROJ ? 0 : 2 * (C / 64), LPS = DMA1 + DMA2;  // LDS-DMA instructions per wave per chunk
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* b1s = reinterpret_cast<float*>(smem + S * SLOT);  // fc1 bias, HD floats
  const unsigned b1a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + S * SLOT;
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, lr = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long row0 = (long)blockIdx.x * 128 + wave * 32;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

  // LDS-DMA source pointers: a wave-instruction moves 16 rows x 64 B (lane -> row lane>>2, 16-B chunk lane&3, swizzled)
  const int drow = lane >> 2, dchunk = (lane & 3) ^ (((drow >> 2) & 1) << 1);
  auto issue_chunk = [&](int ch, int slot) {
    char* base = smem + slot * SLOT;
#pragma unroll
    for (int kb = 0; kb < KB1; ++kb) {  // W1 image kb: rows = hidden units ch*64 + [0,64), this wave moves rows 16*wave..
      const bf16_t* src = p.w1 + (long)(ch * HC + wave * 16 + drow) * C + kb * 32 + dchunk * 8;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(base + kb * (HC * 64) + wave * 1024), 16, 0, 0);
    }
    if constexpr (!PROJ)
#pragma unroll
    for (int pq = 0; pq < 2; ++pq)      // W2 image pq: rows = channels, k = hidden units ch*64 + 32*pq + [0,32)
#pragma unroll
      for (int i = 0; i < C / 64; ++i) {
        const int r = (wave * (C / 64) + i) * 16 + drow;
        const bf16_t* src = p.w2 + (long)r * HD + ch * HC + pq * 32 + dchunk * 8;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(base + W1B + pq * (C * 64) + (wave * (C / 64) + i) * 1024), 16, 0,
                                         0);
      }
  };

  // x fragments straight from global memory: row m = lane&15 of the tile, k = 32 kb + 8 g .. + 7
  bf16x8 xf[2][KB1];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    long r = row0 + mi * 16 + lr;
    r = r < p.M ? r : p.M - 1;  // rows past the edge are computed on a copy of the last row and never stored
#pragma unroll
    for (int kb = 0; kb < KB1; ++kb) xf[mi][kb] = *reinterpret_cast<const bf16x8*>(p.x + r * C + kb * 32 + g * 8);
  }
  for (int i = tid; i < HD; i += 256) b1s[i] = p.b1[i];
  __syncthreads();  // bias table visible (also orders it before the first LDS-DMA)
#pragma unroll
  for (int s = 0; s < S - 1; ++s)
    if (s < NCH) issue_chunk(s, s);

  f32x4 acc2[2][CT];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc2[mi][ct] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int ch = 0; ch < NCH; ++ch) {
    // chunk ch has landed (the S-2 younger chunks may stay in flight; the tail drains)
    if (ch + S - 2 < NCH) mlp_wait_vmcnt<(S - 2) * LPS>();
    else mlp_wait_vmcnt<0>();
    // raw barrier (a __syncthreads() carries a fence that drains vmcnt, i.e. the prefetched chunks): the chunk is visible
    // to every wave; every wave is done with the slot that is refilled next (its LDS reads were waited with lgkmcnt)
    __builtin_amdgcn_s_barrier();
    if (ch + S - 1 < NCH) issue_chunk(ch + S - 1, (ch + S - 1) % S);
    const unsigned w1a = lds0 + (ch % S) * SLOT, w2a = w1a + W1B;  // LDS byte addresses of this chunk's images

    // fc1 bias of this chunk, read from LDS with asm loads like the weights
    f32x4 b1r[2][2];
    {
      u32x4 t0 = mlp_lds_b128(b1a + (ch * HC + 0 * 16 + g * 4) * 4), t1 = mlp_lds_b128(b1a + (ch * HC + 1 * 16 + g * 4) * 4);
      u32x4 t2 = mlp_lds_b128(b1a + (ch * HC + 2 * 16 + g * 4) * 4), t3 = mlp_lds_b128(b1a + (ch * HC + 3 * 16 + g * 4) * 4);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3)::"memory");
      b1r[0][0] = __builtin_bit_cast(f32x4, t0); b1r[0][1] = __builtin_bit_cast(f32x4, t1);
      b1r[1][0] = __builtin_bit_cast(f32x4, t2); b1r[1][1] = __builtin_bit_cast(f32x4, t3);
    }
    // ---- fc1 on this chunk's 64 hidden units: 4 tiles of 16 ----
    f32x4 acc1[2][4];
#pragma unroll
    for (int j2 = 0; j2 < 4; j2 += 2) {  // two hidden tiles per wait: 4 independent accumulator chains for the MFMA pipe
      u32x4 wf[2][KB1];
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int kb = 0; kb < KB1; ++kb) wf[jj][kb] = mlp_lds_b128(w1a + kb * (HC * 64) + img_off((j2 + jj) * 16 + lr, g));
      if constexpr (KB1 == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wf[0][0]), "+v"(wf[0][1]), "+v"(wf[1][0]), "+v"(wf[1][1])::"memory");
      else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wf[0][0]), "+v"(wf[0][1]), "+v"(wf[0][2]), "+v"(wf[0][3]), "+v"(wf[1][0]), "+v"(wf[1][1]),
                        "+v"(wf[1][2]), "+v"(wf[1][3])::"memory");
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) { acc1[0][j2 + jj] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[1][j2 + jj] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int kb = 0; kb < KB1; ++kb)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const bf16x8 w = __builtin_bit_cast(bf16x8, wf[jj][kb]);
          acc1[0][j2 + jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, xf[0][kb], acc1[0][j2 + jj], 0, 0, 0);
          acc1[1][j2 + jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, xf[1][kb], acc1[1][j2 + jj], 0, 0, 0);
        }
    }
    if constexpr (PROJ) {  // the chunk's 64 "hidden units" are output channels 64 ch .. 64 ch + 63
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) acc2[mi][ch * 4 + j] = acc1[mi][j] + b1r[j >> 1][j & 1];
      continue;
    }
    // ---- bias + GELU, packed as the next MFMA's B operand (hidden units 32 pq + {4g..4g+3, 16+4g..16+4g+3}) ----
    bf16x8 hf[2][2];
#pragma unroll
    for (int pq = 0; pq < 2; ++pq) {
      const f32x4 ba = b1r[pq][0], bb = b1r[pq][1];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const f32x4 u = acc1[mi][2 * pq] + ba, v = acc1[mi][2 * pq + 1] + bb;
        const f32x2 g0 = gelu_poly(f32x2{u[0], u[1]}), g1 = gelu_poly(f32x2{u[2], u[3]});  // two elements per packed-fp32 op
        const f32x2 g2 = gelu_poly(f32x2{v[0], v[1]}), g3 = gelu_poly(f32x2{v[2], v[3]});
        hf[mi][pq] = bf16x8{(bf16_t)g0[0], (bf16_t)g0[1], (bf16_t)g1[0], (bf16_t)g1[1], (bf16_t)g2[0], (bf16_t)g2[1], (bf16_t)g3[0], (bf16_t)g3[1]};
      }
    }
    // ---- fc2 partial sums over this chunk: W2 fragments in the same permuted k order (two 8-byte reads) ----
#pragma unroll
    for (int ct2 = 0; ct2 < CT; ct2 += 2) {  // two channel tiles per wait: 4 independent accumulator chains
      u32x2 lo[2][2], hi[2][2];
#pragma unroll
      for (int cc = 0; cc < 2; ++cc)
#pragma unroll
        for (int pq = 0; pq < 2; ++pq) {
          const int r = (ct2 + cc) * 16 + lr;
          const unsigned rowp = w2a + pq * (C * 64);
          lo[cc][pq] = mlp_lds_b64(rowp + img_off(r, g >> 1) + (g & 1) * 8);        // k = 4g .. 4g+3
          hi[cc][pq] = mlp_lds_b64(rowp + img_off(r, (g >> 1) + 2) + (g & 1) * 8);  // k = 16+4g ..
        }
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[0][0]), "+v"(hi[0][0]), "+v"(lo[0][1]), "+v"(hi[0][1]), "+v"(lo[1][0]), "+v"(hi[1][0]),
                   "+v"(lo[1][1]), "+v"(hi[1][1])::"memory");
#pragma unroll
      for (int pq = 0; pq < 2; ++pq)
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
          const u32x4 both = {lo[cc][pq][0], lo[cc][pq][1], hi[cc][pq][0], hi[cc][pq][1]};
          const bf16x8 w = __builtin_bit_cast(bf16x8, both);
          acc2[0][ct2 + cc] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, hf[0][pq], acc2[0][ct2 + cc], 0, 0, 0);
          acc2[1][ct2 + cc] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, hf[1][pq], acc2[1][ct2 + cc], 0, 0, 0);
        }
    }
  }

  // ---- epilogue: + b2, LayerNorm over the C channels of each row, residual, stores ----
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const long r = row0 + mi * 16 + lr;
    float sum = 0.f;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      f32x4 bb = f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr (!PROJ) bb = *reinterpret_cast<const f32x4*>(p.b2 + ct * 16 + g * 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) { acc2[mi][ct][i] += bb[i]; sum += acc2[mi][ct][i]; }
    }
    sum += __shfl_xor(sum, 16, 64); sum += __shfl_xor(sum, 32, 64);
    const float mean = sum * (1.f / C);
    float var = 0.f;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int i = 0; i < 4; ++i) { const float dlt = acc2[mi][ct][i] - mean; var += dlt * dlt; }
    var += __shfl_xor(var, 16, 64); var += __shfl_xor(var, 32, 64);
    const float rstd = rsqrtf(var * (1.f / C) + p.eps);
    if (r < p.M) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const int c = ct * 16 + g * 4;
        const f32x4 gm = *reinterpret_cast<const f32x4*>(p.gamma + c), bt = *reinterpret_cast<const f32x4*>(p.beta + c);
        const f32x4 sc = *reinterpret_cast<const f32x4*>(p.shortcut + r * C + c);
        f32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = sc[i] + ((acc2[mi][ct][i] - mean) * rstd * gm[i] + bt[i]);
        *reinterpret_cast<f32x4*>(p.out + r * C + c) = o;
        if (p.outt) *reinterpret_cast<bf16x4*>(p.outt + r * C + c) = bf16x4{(bf16_t)o[0], (bf16_t)o[1], (bf16_t)o[2], (bf16_t)o[3]};
      }
    }
  }
}

template <int C, int S, bool PROJ = false>
int launch_mlp(const MlpP& p, hipStream_t s) {
  constexpr int SLOT = (C / 32) * 64 * 64 + (PROJ ? 0 : 2 * C * 64);
  const size_t lds = (size_t)S * SLOT + (PROJ ? C : 4 * C) * 4;
  const int rc = ensure_dyn_lds(reinterpret_cast<const void*>(swin_mlp_fused_kernel<C, S, PROJ>), lds);
  if (rc) return rc;
  hipLaunchKernelGGL((swin_mlp_fused_kernel<C, S, PROJ>), dim3((unsigned)((p.M + 127) / 128)), dim3(256), lds, s, p);
  KLAB_LAUNCH_CHECK();
  return KLAB_OK;
}

}  // namespace
}  // namespace klab

extern "C" int klab_swin_mlp_fused(const void* x, const float* shortcut, const void* w1, const float* b1, const void* w2, const float* b2,
                                   const float* gamma, const float* beta, float* out, void* outt, int dtype, int M, int C, float eps,
                                   void* stream) {
  using namespace klab;
  if (!x || !shortcut || !w1 || !b1 || !w2 || !b2 || !gamma || !beta || !out) return KLAB_ERR_BADARG;
  if (dtype != KLAB_BF16) return KLAB_ERR_UNSUPPORTED;
  if (M <= 0) return KLAB_OK;
  MlpP p{(const bf16_t*)x, shortcut, (const bf16_t*)w1, b1, (const bf16_t*)w2, b2, gamma, beta, out, (bf16_t*)outt, M, eps};
  if (C == 64) return launch_mlp<64, 3>(p, (hipStream_t)stream);
  if (C == 128) return launch_mlp<128, 2>(p, (hipStream_t)stream);
  return KLAB_ERR_UNSUPPORTED;  // wider stages keep the three-kernel path (their weights do not fit the streaming budget)
}

extern "C" int klab_swin_proj_ln_fused(const void* x, const float* shortcut, const void* w, const float* b, const float* gamma,
                                       const float* beta, float* out, void* outt, int dtype, int M, int C, float eps, void* stream) {
  using namespace klab;
  if (!x || !shortcut || !w || !b || !gamma || !beta || !out) return KLAB_ERR_BADARG;
  if (dtype != KLAB_BF16) return KLAB_ERR_UNSUPPORTED;
  if (M <= 0) return KLAB_OK;
  MlpP p{(const bf16_t*)x, shortcut, (const bf16_t*)w, b, nullptr, nullptr, gamma, beta, out, (bf16_t*)outt, M, eps};
  if (C == 64) return launch_mlp<64, 2, true>(p, (hipStream_t)stream);
  if (C == 128) return launch_mlp<128, 2, true>(p, (hipStream_t)stream);
  // C = 256 (stage 2 of the caption tower): opt-in experiment (KLAB_SWIN_PROJ256=1); see DESIGN for the measurement
  static const bool p256 = [] { const char* e = getenv("KLAB_SWIN_PROJ256"); return e && atoi(e) != 0; }();
  if (C == 256 && p256) return launch_mlp<256, 2, true>(p, (hipStream_t)stream);
  return KLAB_ERR_UNSUPPORTED;
}
