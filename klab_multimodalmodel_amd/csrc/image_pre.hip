// the reference's host-side image path on the GPU, bit-exact with Pillow's 8-bit resampler.
//   ref/modules/loader.py:15   Image.open(..).convert('RGB').resize((256,256))  (Pillow ImagingResample, default BICUBIC)
//   ref/modules/loader.py:16   ToTensor()
//   ref/train.py:55            image_processor(images)  = ViTImageProcessor (HF/vitproc:20-27): back to uint8, Pillow BILINEAR
//                              to 224x224, x 1/255 (undoing its own conversion), x 1/255 again (do_rescale), (x - 0.5) / 0.5
// Pillow (src/libImaging/Resample.c) resamples 8-bit images in 22-bit fixed point: per output index the filter weights are
// evaluated in double precision over [center - support, center + support], normalised to sum 1, rounded to integers
// (round-half-away), and every pass (horizontal first, then vertical) accumulates from 1 << 21, shifts by 22 and clamps to
// uint8.  The weights are recomputed here per workgroup with the same double-precision operations in the same order
// (contraction off: no FMA), so every intermediate uint8 image equals Pillow's.  This is HBM-bound byte work: no MFMA.
//   resample_h_kernel : [h, w, 3] u8 -> [h, mid, 3] u8      one thread per output column, weights in an LDS column, 16 rows per block
//   resample_v_kernel : [h, mid, 3] u8 -> [mid, mid, 3] u8  8 output rows per block, byte-coalesced over the row
//   resize_norm_kernel: [mid, mid, 3] u8 -> pixel_values [3, out, out] f32: both passes of the second resize through an LDS
//                       tile + rescale + normalise, written channel-major (the layout the patch-embedding im2col reads)
#include <math.h>

#include "common.h"
#include "klab_mm.h"

namespace klab {
namespace {

constexpr int PBITS = 22;  // PRECISION_BITS = 32 - 8 - 2

__device__ __forceinline__ double pil_filter(int f, double x) {
#pragma clang fp contract(off)
  if (x < 0.0) x = -x;
  if (f == 2) return x < 1.0 ? 1.0 - x : 0.0;                        // bilinear_filter
  if (x < 1.0) return ((-0.5 + 2.0) * x - (-0.5 + 3.0)) * x * x + 1;  // bicubic_filter, a = -0.5
  if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * -0.5;
  return 0.0;
}

struct Span { int lo, n; };

__host__ __device__ inline int pil_ksize(int in_size, int out_size, int f) {
  double fs = (double)in_size / out_size;
  if (fs < 1.0) fs = 1.0;
  return (int)ceil((f == 2 ? 1.0 : 2.0) * fs) * 2 + 1;
}

// precompute_coeffs + normalize_coeffs_8bpc for output index xx; weights go to k[0], k[stride], ...
__device__ inline Span pil_coeffs(int in_size, int out_size, int f, int xx, int* k, int stride) {
#pragma clang fp contract(off)
  const double scale = (double)in_size / out_size;
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = (f == 2 ? 1.0 : 2.0) * filterscale;
  const double center = 0.0 + (xx + 0.5) * scale;
  const double ss = 1.0 / filterscale;
  int xmin = (int)(center - support + 0.5);
  if (xmin < 0) xmin = 0;
  int xmax = (int)(center + support + 0.5);
  if (xmax > in_size) xmax = in_size;
  xmax -= xmin;
  double ww = 0.0;
  for (int x = 0; x < xmax; ++x) ww += pil_filter(f, (x + xmin - center + 0.5) * ss);
  for (int x = 0; x < xmax; ++x) {
    double w = pil_filter(f, (x + xmin - center + 0.5) * ss);
    if (ww != 0.0) w /= ww;
    k[x * stride] = w < 0 ? (int)(-0.5 + w * (double)(1 << PBITS)) : (int)(0.5 + w * (double)(1 << PBITS));
  }
  return Span{xmin, xmax};
}

__device__ __forceinline__ unsigned char clip8(int v) {
  v >>= PBITS;
  // keep shift and clamp apart: fused into gfx950's v_ashr_pk_u8_i32 (when two results are packed) the low bit came out
  // different from floor(v / 2^22) clamped -- measured as +1 errors against Pillow in the 32-bit vertical pass
  asm volatile("" : "+v"(v));
  return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

struct ImgP {
  const unsigned char* src; const klab_image_desc* desc; unsigned char* tmp; unsigned char* mid_img;
  int n, max_h, mid, filter_a, kmax;
};

constexpr int HROWS = 16;
__global__ __launch_bounds__(256) void resample_h_kernel(ImgP p) {
  extern __shared__ int lds_h[];  // k[kmax][mid] | lo[mid] | n[mid]
  const klab_image_desc d = p.desc[blockIdx.y];
  const int y0 = blockIdx.x * HROWS;
  if (y0 >= d.height) return;
  const int mid = p.mid;
  int* kk = lds_h;
  int* lo = kk + p.kmax * mid;
  int* nn = lo + mid;
  for (int xx = threadIdx.x; xx < mid; xx += 256) {
    const Span s = pil_coeffs(d.width, mid, p.filter_a, xx, kk + xx, mid);
    lo[xx] = s.lo; nn[xx] = s.n;
  }
  // each thread reads back only the column it wrote: no barrier needed
  const unsigned char* src = p.src + d.offset;
  unsigned char* dst = p.tmp + (size_t)blockIdx.y * p.max_h * mid * 3;
  const int y1 = y0 + HROWS < d.height ? y0 + HROWS : d.height;
  for (int xx = threadIdx.x; xx < mid; xx += 256) {
    const int l = lo[xx], n = nn[xx];
    for (int y = y0; y < y1; ++y) {
      const unsigned char* row = src + ((size_t)y * d.width + l) * 3;
      int s0 = 1 << (PBITS - 1), s1 = s0, s2 = s0;
      for (int t = 0; t < n; ++t) {
        const int k = kk[t * mid + xx];
        s0 += row[t * 3] * k; s1 += row[t * 3 + 1] * k; s2 += row[t * 3 + 2] * k;
      }
      unsigned char* o = dst + ((size_t)y * mid + xx) * 3;
      o[0] = clip8(s0); o[1] = clip8(s1); o[2] = clip8(s2);
    }
  }
}

// vertical pass: 8 output rows per block; a thread owns 4 consecutive bytes of the row (one 32-bit load per source row)
constexpr int VROWS = 8;
__global__ __launch_bounds__(256) void resample_v_kernel(ImgP p) {
  extern __shared__ __attribute__((aligned(16))) int lds_v[];  // k[VROWS][kmax] | lo[VROWS] | n[VROWS]
  const klab_image_desc d = p.desc[blockIdx.y];
  const int mid = p.mid, y0 = blockIdx.x * VROWS;
  int* kk = lds_v;
  int* lo = kk + VROWS * p.kmax;
  int* nn = lo + VROWS;
  const int nout = y0 + VROWS <= mid ? VROWS : mid - y0;
  if ((int)threadIdx.x < nout) {
    const Span s = pil_coeffs(d.height, mid, p.filter_a, y0 + threadIdx.x, kk + threadIdx.x * p.kmax, 1);
    lo[threadIdx.x] = s.lo; nn[threadIdx.x] = s.n;
  }
  __syncthreads();
  const int rowb = mid * 3;
  const unsigned char* src = p.tmp + (size_t)blockIdx.y * p.max_h * rowb;
  unsigned char* dst = p.mid_img + (size_t)blockIdx.y * mid * rowb;
  if ((rowb & 3) == 0) {  // row pitch and all bases are 4-byte aligned (workspace regions are 256-byte aligned)
    for (int e = threadIdx.x * 4; e < rowb; e += 1024) {
      for (int r = 0; r < nout; ++r) {
        const int* k = kk + r * p.kmax;
        const int lr = lo[r], nr = nn[r];
        int a0 = 1 << (PBITS - 1), a1 = a0, a2 = a0, a3 = a0;
        for (int t = 0; t < nr; ++t) {
          const unsigned int v = *reinterpret_cast<const unsigned int*>(src + (size_t)(lr + t) * rowb + e);
          const int kt = k[t];
          a0 += (int)(v & 255u) * kt; a1 += (int)((v >> 8) & 255u) * kt; a2 += (int)((v >> 16) & 255u) * kt; a3 += (int)(v >> 24) * kt;
        }
        const unsigned int o = (unsigned int)clip8(a0) | ((unsigned int)clip8(a1) << 8) | ((unsigned int)clip8(a2) << 16) | ((unsigned int)clip8(a3) << 24);
        *reinterpret_cast<unsigned int*>(dst + (size_t)(y0 + r) * rowb + e) = o;
      }
    }
    return;
  }
  for (int r = 0; r < nout; ++r) {
    const int* k = kk + r * p.kmax;
    for (int e = threadIdx.x; e < rowb; e += 256) {
      int s = 1 << (PBITS - 1);
      for (int t = 0; t < nn[r]; ++t) s += src[(size_t)(lo[r] + t) * rowb + e] * k[t];
      dst[(size_t)(y0 + r) * rowb + e] = clip8(s);
    }
  }
}

struct NormP {
  const unsigned char* mid_img; float* out;
  int n, mid, osz, filter_b, kb, tr, nr_max;
  double rescale; double mean[3], stdv[3];
};

__global__ __launch_bounds__(256) void resize_norm_kernel(NormP p) {
  extern __shared__ int lds_n[];  // kx[kb][osz] | lox[osz] | nx[osz] | ky[tr][kb] | loy[tr] | ny[tr] | lut f32 [3][256] | tile u8 [nr_max][osz*3]
  const int osz = p.osz, mid = p.mid, kb = p.kb, tr = p.tr;
  int* kx = lds_n;
  int* lox = kx + kb * osz;
  int* nx = lox + osz;
  int* ky = nx + osz;
  int* loy = ky + tr * kb;
  int* ny = loy + tr;
  float* lut = reinterpret_cast<float*>(ny + tr);
  unsigned char* tile = reinterpret_cast<unsigned char*>(lut + 768);
  const int y0 = blockIdx.x * tr;
  const int rows = y0 + tr <= osz ? tr : osz - y0;
  for (int xx = threadIdx.x; xx < osz; xx += 256) {
    const Span s = pil_coeffs(mid, osz, p.filter_b, xx, kx + xx, osz);
    lox[xx] = s.lo; nx[xx] = s.n;
  }
  if ((int)threadIdx.x < rows) {
    const Span s = pil_coeffs(mid, osz, p.filter_b, y0 + threadIdx.x, ky + threadIdx.x * kb, 1);
    loy[threadIdx.x] = s.lo; ny[threadIdx.x] = s.n;
  }
  for (int i = threadIdx.x; i < 768; i += 256) {  // (u8 * rescale - mean) / std evaluated once per value, in double
    const int c = i >> 8;
    lut[i] = (float)(((double)(i & 255) * p.rescale - p.mean[c]) / p.stdv[c]);
  }
  __syncthreads();
  const int r_lo = loy[0], r_hi = loy[rows - 1] + ny[rows - 1];  // source rows this tile needs (bounds grow with the index)
  const int nr = r_hi - r_lo;
  const int rowb = osz * 3;
  const unsigned char* src = p.mid_img + (size_t)blockIdx.y * mid * mid * 3;
  // horizontal pass of rows [r_lo, r_hi) into LDS
  for (int i = threadIdx.x; i < nr * osz; i += 256) {
    const int r = i / osz, xx = i - r * osz;
    const int l = lox[xx], n = nx[xx];
    const unsigned char* row = src + ((size_t)(r_lo + r) * mid + l) * 3;
    int s0 = 1 << (PBITS - 1), s1 = s0, s2 = s0;
    for (int t = 0; t < n; ++t) {
      const int k = kx[t * osz + xx];
      s0 += row[t * 3] * k; s1 += row[t * 3 + 1] * k; s2 += row[t * 3 + 2] * k;
    }
    unsigned char* o = tile + r * rowb + xx * 3;
    o[0] = clip8(s0); o[1] = clip8(s1); o[2] = clip8(s2);
  }
  __syncthreads();
  // vertical pass + rescale + normalise, channel-major writes
  float* out = p.out + (size_t)blockIdx.y * 3 * osz * osz;
  for (int i = threadIdx.x; i < rows * rowb; i += 256) {
    const int r = i / rowb, j = i - r * rowb;
    const int c = j / osz, xx = j - c * osz;
    const int l = loy[r] - r_lo, n = ny[r];
    const int* k = ky + r * kb;
    int s = 1 << (PBITS - 1);
    for (int t = 0; t < n; ++t) s += tile[(l + t) * rowb + xx * 3 + c] * k[t];
    out[((size_t)c * osz + (y0 + r)) * osz + xx] = lut[c * 256 + clip8(s)];
  }
}

}  // namespace
}  // namespace klab

using namespace klab;

static size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" size_t klab_image_preprocess_ws_bytes(int n_images, int max_h, int mid) {
  if (n_images <= 0 || max_h <= 0 || mid <= 0) return 0;
  return al256((size_t)n_images * max_h * mid * 3) + al256((size_t)n_images * mid * mid * 3);
}

extern "C" int klab_image_preprocess(const unsigned char* src, const klab_image_desc* desc_dev, int n_images, int max_h, int max_w, int mid,
                                     int out_size, int filter_a, int filter_b, double rescale, const float* mean3, const float* std3,
                                     float* pixel_values, void* ws, size_t ws_bytes, void* stream) {
  if (!src || !pixel_values || !mean3 || !std3 || n_images <= 0 || mid <= 0 || out_size <= 0) return KLAB_ERR_BADARG;
  if ((filter_a != 0 && filter_a != 2 && filter_a != 3) || (filter_b != 2 && filter_b != 3)) return KLAB_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const unsigned char* mid_img = src;  // filter_a == 0: src already is [n, mid, mid, 3]
  if (filter_a != 0) {
    if (!desc_dev || !ws || max_h <= 0 || max_w <= 0) return KLAB_ERR_BADARG;
    if (ws_bytes < klab_image_preprocess_ws_bytes(n_images, max_h, mid)) return KLAB_ERR_BADARG;
    ImgP p{src, desc_dev, (unsigned char*)ws, (unsigned char*)ws + al256((size_t)n_images * max_h * mid * 3), n_images, max_h, mid, filter_a, 0};
    // the widest / tallest image bounds every image's tap count
    const int kh = pil_ksize(max_w, mid, filter_a), kv = pil_ksize(max_h, mid, filter_a);
    p.kmax = kh;
    const size_t lds_h = ((size_t)kh * mid + 2 * (size_t)mid) * 4;
    if (lds_h > 128 * 1024) return KLAB_ERR_UNSUPPORTED;  // > ~30x reduction of the width
    int rc = ensure_dyn_lds(reinterpret_cast<const void*>(resample_h_kernel), lds_h);
    if (rc) return rc;
    hipLaunchKernelGGL(resample_h_kernel, dim3((max_h + HROWS - 1) / HROWS, n_images), dim3(256), lds_h, s, p);
    KLAB_LAUNCH_CHECK();
    p.kmax = kv;
    const size_t lds_v = ((size_t)VROWS * kv + 2 * VROWS) * 4;
    if (lds_v > 64 * 1024) return KLAB_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(resample_v_kernel, dim3((mid + VROWS - 1) / VROWS, n_images), dim3(256), lds_v, s, p);
    KLAB_LAUNCH_CHECK();
    mid_img = p.mid_img;
  }
  NormP q;
  q.mid_img = mid_img; q.out = pixel_values; q.n = n_images; q.mid = mid; q.osz = out_size; q.filter_b = filter_b;
  q.kb = pil_ksize(mid, out_size, filter_b);
  q.tr = 16;
  const double sc = (double)mid / out_size;
  q.nr_max = (int)ceil(q.tr * (sc < 1.0 ? 1.0 : sc)) + q.kb + 2;
  if (q.nr_max > mid) q.nr_max = mid;
  q.rescale = rescale;
  for (int c = 0; c < 3; ++c) { q.mean[c] = mean3[c]; q.stdv[c] = std3[c]; }
  const size_t lds_n = ((size_t)q.kb * out_size + 2 * (size_t)out_size + (size_t)q.tr * q.kb + 2 * q.tr + 768) * 4 + (size_t)q.nr_max * out_size * 3;
  if (lds_n > 64 * 1024) return KLAB_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(resize_norm_kernel, dim3((out_size + q.tr - 1) / q.tr, n_images), dim3(256), lds_n, s, q);
  KLAB_LAUNCH_CHECK();
  return KLAB_OK;
}
