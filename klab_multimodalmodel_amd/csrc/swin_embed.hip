// Fused Swin-V2 patch embedding for a FROZEN tower (no activation kept for a backward pass):
//     x0 = LayerNorm( Conv2d(3 -> C, kernel 4, stride 4)(pixels) )            (HF/swinv2:234-259, 281, 293-302)
// in ONE launch instead of im2col + GEMM + LayerNorm (26 + 29 + 20 us at B = 64: the 77 MB column matrix and the bf16 GEMM output
// went through HBM for a product with K = 48).  One wave owns 16 patches: the 4 x 4 x 3 pixel block of a patch is read straight
// from the image as the MFMA's B-operand fragment (k = 16 c + 4 dy + dx: a lane's 8 consecutive k are two image rows of 4
// pixels of one channel -- two 16-byte loads; 16 neighbouring patches read 256 contiguous bytes per image row), the projection
// weight [C, 48 -> 64 zero-padded] sits in registers as the A operand, the accumulators come out as 4 consecutive channels of a
// patch per lane, and the LayerNorm over C is in-lane + two shuffles.  Memory traffic: pixels in (fp32), x0 out (fp32 + bf16).
#include <stdlib.h>

#include "common.h"
#include "klab_mm.h"

namespace klab {
namespace {

struct EmbedP {
  const float* pix; const bf16_t* w; const float* bias; const float* gamma; const float* beta;
  float* out; bf16_t* outt;
  int B, HW, R;  // image size, patches per side
  int ldw;       // weight row pitch (elements): 64
  float eps;
};

template <int C>
__global__ __launch_bounds__(256) void swin_patch_embed_fused_kernel(EmbedP p) {
  constexpr int NI = C / 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
  const long M = (long)p.B * p.R * p.R;
  const long m0 = ((long)blockIdx.x * 4 + wave) * 16;
  if (m0 >= M) return;
  // weight fragments: row n = 16 j + (lane & 15), k = 32 ks + 8 g .. + 7 (rows are zero-padded to 64)
  bf16x8 wf[NI][2];
#pragma unroll
  for (int j = 0; j < NI; ++j)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) wf[j][ks] = *reinterpret_cast<const bf16x8*>(p.w + (long)(j * 16 + (lane & 15)) * p.ldw + ks * 32 + g * 8);
  // this lane's patch and its pixel fragments
  long m = m0 + (lane & 15);
  const bool valid = m < M;
  if (!valid) m = M - 1;
  const int b = (int)(m / ((long)p.R * p.R)), rem = (int)(m % ((long)p.R * p.R)), py = rem / p.R, px = rem % p.R;
  bf16x8 xf[2];
  {
    // ks = 0: k = 8 g .. 8 g + 7 -> channel g >> 1, rows dy = 2 (g & 1), 2 (g & 1) + 1;  ks = 1: k = 32 + 8 g -> channel 2 for g < 2, padding above
    const int c0 = g >> 1, dy0 = 2 * (g & 1);
    const float* base0 = p.pix + (((long)b * 3 + c0) * p.HW + (py * 4 + dy0)) * p.HW + px * 4;
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(base0), a1 = *reinterpret_cast<const f32x4*>(base0 + p.HW);
    xf[0] = bf16x8{(bf16_t)a0[0], (bf16_t)a0[1], (bf16_t)a0[2], (bf16_t)a0[3], (bf16_t)a1[0], (bf16_t)a1[1], (bf16_t)a1[2], (bf16_t)a1[3]};
    const float* base1 = p.pix + (((long)b * 3 + 2) * p.HW + (py * 4 + dy0)) * p.HW + px * 4;  // (read by every lane: no divergent load)
    const f32x4 c0v = *reinterpret_cast<const f32x4*>(base1), c1v = *reinterpret_cast<const f32x4*>(base1 + p.HW);
    const float z = g < 2 ? 1.f : 0.f;
    xf[1] = bf16x8{(bf16_t)(c0v[0] * z), (bf16_t)(c0v[1] * z), (bf16_t)(c0v[2] * z), (bf16_t)(c0v[3] * z),
                   (bf16_t)(c1v[0] * z), (bf16_t)(c1v[1] * z), (bf16_t)(c1v[2] * z), (bf16_t)(c1v[3] * z)};
  }
  f32x4 acc[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][ks], xf[ks], acc[j], 0, 0, 0);
  }
  // acc[j][r] = conv(patch m)[channel 16 j + 4 g + r].  The three-launch path stored the GEMM output in bf16 before the norm:
  // round the same way, so that the fused and unfused towers agree to the last bit of the LayerNorm's input
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const f32x4 bb = *reinterpret_cast<const f32x4*>(p.bias + j * 16 + g * 4);
#pragma unroll
    for (int r = 0; r < 4; ++r) { acc[j][r] = (float)(bf16_t)(acc[j][r] + bb[r]); sum += acc[j][r]; }
  }
  sum += __shfl_xor(sum, 16, 64); sum += __shfl_xor(sum, 32, 64);
  const float mean = sum * (1.f / C);
  float var = 0.f;
#pragma unroll
  for (int j = 0; j < NI; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) { const float d = acc[j][r] - mean; var += d * d; }
  var += __shfl_xor(var, 16, 64); var += __shfl_xor(var, 32, 64);
  const float rstd = rsqrtf(var * (1.f / C) + p.eps);
  if (valid) {
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int c = j * 16 + g * 4;
      const f32x4 gm = *reinterpret_cast<const f32x4*>(p.gamma + c), bt = *reinterpret_cast<const f32x4*>(p.beta + c);
      f32x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = (acc[j][r] - mean) * rstd * gm[r] + bt[r];
      *reinterpret_cast<f32x4*>(p.out + m * C + c) = o;
      if (p.outt) *reinterpret_cast<bf16x4*>(p.outt + m * C + c) = bf16x4{(bf16_t)o[0], (bf16_t)o[1], (bf16_t)o[2], (bf16_t)o[3]};
    }
  }
}

}  // namespace
}  // namespace klab

// pixels [B, 3, HW, HW] f32; w [C, ldw >= 64] bf16 with columns 48..63 zero (the engine's padded arena copy of the Conv2d
// weight [C, 3, 4, 4]); bias / gamma / beta [C] f32; out [B*R*R, C] f32, outt the same in bf16 (optional).
// bf16, patch 4, 3 input channels, C in {64, 96, 128}; otherwise KLAB_ERR_UNSUPPORTED (caller: im2col + klab_gemm + klab_layernorm_fwd).
extern "C" int klab_swin_patch_embed_fused(const float* pixels, const void* w, int ldw, const float* bias, const float* gamma, const float* beta,
                                           float* out, void* outt, int dtype, int B, int in_ch, int image_size, int patch, int C, float eps,
                                           void* stream) {
  using namespace klab;
  if (!pixels || !w || !bias || !gamma || !beta || !out) return KLAB_ERR_BADARG;
  if (dtype != KLAB_BF16 || in_ch != 3 || patch != 4 || ldw < 64 || (ldw & 7) || (image_size & 3) || ((uintptr_t)pixels & 15) || ((uintptr_t)w & 15))
    return KLAB_ERR_UNSUPPORTED;
  if (B <= 0) return KLAB_OK;
  EmbedP p{pixels, (const bf16_t*)w, bias, gamma, beta, out, (bf16_t*)outt, B, image_size, image_size / 4, ldw, eps};
  const long M = (long)B * p.R * p.R;
  const unsigned grid = (unsigned)((M + 63) / 64);
  hipStream_t s = (hipStream_t)stream;
  if (C == 64) hipLaunchKernelGGL(swin_patch_embed_fused_kernel<64>, dim3(grid), dim3(256), 0, s, p);
  else if (C == 96) hipLaunchKernelGGL(swin_patch_embed_fused_kernel<96>, dim3(grid), dim3(256), 0, s, p);
  else if (C == 128) hipLaunchKernelGGL(swin_patch_embed_fused_kernel<128>, dim3(grid), dim3(256), 0, s, p);
  else return KLAB_ERR_UNSUPPORTED;
  KLAB_LAUNCH_CHECK();
  return KLAB_OK;
}
