// Shared device helpers for the gfx950 (CDNA4) kernels of the Swin-V2 -> T5 caption-training path.
// wave = 64 lanes everywhere; no CUDA compatibility paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define KLAB_OK 0
#define KLAB_ERR_UNSUPPORTED (-2)
#define KLAB_ERR_BADARG (-3)

#define KLAB_F32 0
#define KLAB_BF16 1

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define KLAB_LAUNCH_CHECK()                          \
  do {                                               \
    hipError_t e__ = hipGetLastError();              \
    if (e__ != hipSuccess) return (int)e__;          \
  } while (0)

namespace klab {

// raise a kernel's dynamic-LDS limit once per (kernel, size): hipFuncSetAttribute is not a stream operation and must
// not be issued while the stream is being captured into a hipGraph (misc.hip)
int ensure_dyn_lds(const void* kernel, size_t bytes);

template <typename T> struct TypeTag;
template <> struct TypeTag<float> { static constexpr int id = KLAB_F32; };
template <> struct TypeTag<bf16_t> { static constexpr int id = KLAB_BF16; };

__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(bf16_t x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float x) { return (bf16_t)x; }  // v_cvt_pk_bf16_f32 (RNE, NaN-safe)

// ---- counter-based dropout RNG -------------------------------------------------------------
// keep(idx) is a pure function of (seed word in device memory, per-site tag, element index), so the
// backward pass regenerates the forward mask instead of storing it (reference: nn.Dropout,
// HF/t5:80,135,168,382,411,651 -- matched statistically, not bitwise).
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
  return x;
}
struct DropCtx {
  uint32_t key;     // mix(seed, tag)
  uint32_t thresh;  // drop if hash < thresh
  float scale;      // 1/(1-p)
  bool on;
};
__device__ __forceinline__ DropCtx make_drop(const uint32_t* seed_dev, uint32_t tag, float p) {
  DropCtx d;
  d.on = (p > 0.f) && (seed_dev != nullptr);
  d.key = 0; d.thresh = 0; d.scale = 1.f;
  if (d.on) {
    d.key = mix32(seed_dev[0] * 0x9E3779B1u + tag * 0x7FEB352Du + 0x165667B1u);
    double t = (double)p * 4294967296.0;
    d.thresh = t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
    d.scale = 1.f / (1.f - p);
  }
  return d;
}
// multiplier applied to the element: 0 (dropped) or 1/(1-p)
__device__ __forceinline__ float drop_mult(const DropCtx& d, uint64_t idx) {
  if (!d.on) return 1.f;
  uint32_t h = mix32((uint32_t)idx * 0x9E3779B1u + d.key);
  const uint32_t hi = (uint32_t)(idx >> 32);
  if (hi) h = mix32(h ^ (hi * 0x85EBCA77u + 0x27D4EB2Fu));  // second round only beyond 2^32 elements
  return h < d.thresh ? 0.f : d.scale;
}
// two-part index (slab, 32-bit offset inside the slab): the slab is folded into the key once per block, so an element
// costs one hash round and no 64-bit index arithmetic (attention probabilities: slab = b*H + h, offset = q*Lk + key)
__device__ __forceinline__ DropCtx drop_slab(DropCtx d, uint32_t slab) {
  if (d.on) d.key = mix32(d.key ^ (slab * 0x85EBCA77u + 0x27D4EB2Fu));
  return d;
}
__device__ __forceinline__ float drop_mult32(const DropCtx& d, uint32_t off) {
  if (!d.on) return 1.f;
  return mix32(off * 0x9E3779B1u + d.key) < d.thresh ? 0.f : d.scale;
}
// the same value without the branch on d.on (thresh = 0 and scale = 1 when dropout is off: nothing is ever below 0): for unrolled
// per-element code, where sixteen copies of that branch cut the loop body into as many basic blocks
__device__ __forceinline__ float drop_mult32_nb(const DropCtx& d, uint32_t off) {
  return mix32(off * 0x9E3779B1u + d.key) < d.thresh ? 0.f : d.scale;
}

// ---- wave / block reductions ---------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// sub-wave group reductions (G = power of two lanes, aligned)
template <int G> __device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
template <int G> __device__ __forceinline__ float group_max(float v) {
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }
// erf by Abramowitz-Stegun 7.1.26 (|abs error| <= 1.5e-7, i.e. below bf16 AND fp32-GEMM noise): one rcp + one exp + 6
// fma instead of the ~40-instruction branchy libm erff, which doubled the time of every GELU-epilogue GEMM
__device__ __forceinline__ float erf_fast(float x) {
  const float ax = fabsf(x);
  const float t = __frcp_rn(fmaf(0.3275911f, ax, 1.f));
  float pl = fmaf(1.061405429f, t, -1.453152027f);
  pl = fmaf(pl, t, 1.421413741f);
  pl = fmaf(pl, t, -0.284496736f);
  pl = fmaf(pl, t, 0.254829592f);
  const float y = 1.f - pl * t * __expf(-ax * ax);
  return copysignf(y, x);
}
// GELU for bf16 outputs: erf(x/sqrt2) = clamp(x * Q(min(x^2, 20.48)), -1, 1) with an 8th-degree near-minimax Q (fitted on
// |x/sqrt2| <= 3.2): |gelu error| <= 1.2e-4 absolute, ~30x below the bf16 rounding of an O(1) activation.  No rcp / exp:
// 14 VALU slots per element instead of ~26 (the rational form above) -- the GELU of the Swin MLPs is VALU-bound
// (51 M elements per stage-0 block), and the form is straight fma code that also packs two lanes' worth per v_pk_fma_f32.
typedef __attribute__((ext_vector_type(2))) float f32x2;
template <typename V> __device__ __forceinline__ V gelu_poly(V x) {
  V s = x * x;
  s = __builtin_elementwise_min(s, V(20.48f));
  V q = V(7.318504830e-11f);
  q = q * s + V(-7.769299870e-09f);
  q = q * s + V(3.614930506e-07f);
  q = q * s + V(-9.785327165e-06f);
  q = q * s + V(1.732400560e-04f);
  q = q * s + V(-2.146258019e-03f);
  q = q * s + V(1.943818480e-02f);
  q = q * s + V(-1.324543953e-01f);
  q = q * s + V(7.977233529e-01f);
  V e = __builtin_elementwise_max(__builtin_elementwise_min(x * q, V(1.f)), V(-1.f));
  const V hx = x * V(0.5f);
  return hx * e + hx;
}
template <typename T> __device__ __forceinline__ float gelu_for(float x) {  // bf16 operands: polynomial; fp32 parity mode: libm erf
  if constexpr (sizeof(T) == 2) return gelu_poly<float>(x);
  else return gelu_erf(x);
}
__device__ __forceinline__ float gelu_erf_grad(float x) {
  float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752440f));
  float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// vector load/store of VEC = 16/sizeof(T) elements
template <typename T> struct Vec16;
template <> struct Vec16<float> { typedef f32x4 type; static constexpr int N = 4; };
template <> struct Vec16<bf16_t> { typedef bf16x8 type; static constexpr int N = 8; };

}  // namespace klab
