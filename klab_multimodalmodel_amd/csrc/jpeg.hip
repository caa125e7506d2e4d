// JPEG reconstruction on the GPU: dequantisation + inverse DCT, chroma upsampling and the YCbCr -> RGB
// transform of `Image.open(path).convert('RGB')` (ref/modules/loader.py:15) for a batch of images whose Huffman decoding was
// done by csrc/jpeg_host.cpp.  Bit-exact with libjpeg-turbo (the decoder Pillow links) in its default configuration:
//   * JDCT_ISLOW: the Loeffler-Ligtenberg-Moschytz integer IDCT with 13-bit constants and a 2-bit first-pass scale;
//   * "fancy" (triangle-filter) upsampling for 2x1 and 2x2 chroma when the sub-sampled width exceeds two columns, pixel
//     replication otherwise; rows above the first / below the last real chroma row replicate that row;
//   * the 16-bit fixed-point colour transform R = Y + 1.402 Cr', G = Y - 0.34414 Cb' - 0.71414 Cr', B = Y + 1.772 Cb'.
// Restated from the published algorithm descriptions (IJG "jidctint" LL&M notes; JFIF 1.02 colour equations); the parity tests
// compare every byte with Pillow's own decode of the same files.  HBM-bound integer work: one 128-byte coefficient block per 8
// lanes in, 64 bytes out; the colour kernel reads 1-2 bytes per sample and writes 3.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "common.h"
#include "klab_mm.h"

namespace klab {

struct JpegPlane {       // one component of one image
  long long block0;      // first coefficient block (in 64-coefficient units) in the batch's coefficient buffer
  long long plane_off;   // byte offset of the component's sample plane [bh*8][bw*8] in the workspace
  long long wg0;         // first workgroup of this plane in the IDCT grid (32 blocks per workgroup)
  int bw, bh;            // blocks per row / column (padded to whole MCUs)
  int qt;                // index of the 64-entry quantisation table in qt[]
};

__device__ __forceinline__ int range_limit(int x) {  // libjpeg's table: 10-bit wrap-around, centred on 128, clamped to a sample
  const int sx = ((x & 1023) ^ 512) - 512;
  const int v = sx + 128;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// one 1-D LL&M pass over eight values (even part: 3 multiplies, odd part: 9)
__device__ __forceinline__ void llm8(const long long (&in)[8], long long (&t)[8], const bool first) {
  constexpr long long F0_298 = 2446, F0_390 = 3196, F0_541 = 4433, F0_765 = 6270, F0_899 = 7373, F1_175 = 9633, F1_501 = 12299,
                      F1_847 = 15137, F1_961 = 16069, F2_053 = 16819, F2_562 = 20995, F3_072 = 25172;
  long long z2 = in[2], z3 = in[6];
  long long z1 = (z2 + z3) * F0_541;
  long long tmp2 = z1 - z3 * F1_847;
  long long tmp3 = z1 + z2 * F0_765;
  z2 = in[0]; z3 = in[4];
  long long tmp0 = (z2 + z3) << 13;
  long long tmp1 = (z2 - z3) << 13;
  const long long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
  tmp0 = in[7]; tmp1 = in[5]; tmp2 = in[3]; tmp3 = in[1];
  z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
  long long z4 = tmp1 + tmp3;
  const long long z5 = (z3 + z4) * F1_175;
  tmp0 *= F0_298; tmp1 *= F2_053; tmp2 *= F3_072; tmp3 *= F1_501;
  z1 *= -F0_899; z2 *= -F2_562; z3 *= -F1_961; z4 *= -F0_390;
  z3 += z5; z4 += z5;
  tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
  (void)first;
  t[0] = tmp10 + tmp3; t[7] = tmp10 - tmp3;
  t[1] = tmp11 + tmp2; t[6] = tmp11 - tmp2;
  t[2] = tmp12 + tmp1; t[5] = tmp12 - tmp1;
  t[3] = tmp13 + tmp0; t[4] = tmp13 - tmp0;
}

// 256 threads = 32 blocks x 8 lanes; lane c of a block does column c in pass 1 and row c in pass 2
__global__ __launch_bounds__(256) void jpeg_idct_kernel(const short* __restrict__ coefs, const unsigned short* __restrict__ qt,
                                                        const JpegPlane* __restrict__ planes, int nplanes, unsigned char* __restrict__ ws) {
  __shared__ int wsp[32][8][9];
  // which plane: binary search over the planes' first workgroups
  int lo = 0, hi = nplanes - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (planes[mid].wg0 <= (long long)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const JpegPlane pl = planes[lo];
  const int lb = threadIdx.x >> 3, c = threadIdx.x & 7;
  const long long b = ((long long)blockIdx.x - pl.wg0) * 32 + lb;
  const long long nblk = (long long)pl.bw * pl.bh;
  const bool live = b < nblk;
  if (live) {
    const short* blk = coefs + (pl.block0 + b) * 64;
    const unsigned short* q = qt + pl.qt * 64;
    long long in[8], t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) in[k] = (long long)blk[k * 8 + c] * (long long)q[k * 8 + c];
    llm8(in, t, true);
#pragma unroll
    for (int k = 0; k < 8; ++k) wsp[lb][k][c] = (int)((t[k] + (1LL << 10)) >> 11);  // DESCALE by CONST_BITS - PASS1_BITS
  }
  __syncthreads();
  if (live) {
    long long in[8], t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) in[k] = wsp[lb][c][k];
    llm8(in, t, false);
    const int by = (int)(b / pl.bw), bx = (int)(b - (long long)by * pl.bw);
    unsigned long long out = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) out |= (unsigned long long)range_limit((int)((t[k] + (1LL << 17)) >> 18)) << (8 * k);
    *reinterpret_cast<unsigned long long*>(ws + pl.plane_off + ((long long)by * 8 + c) * ((long long)pl.bw * 8) + bx * 8) = out;
  }
}

struct JpegImage {
  long long plane_off[3];  // sample planes in the workspace
  long long rgb_off;       // byte offset of the HWC RGB image in dst
  int stride[3];           // plane row strides (bw * 8)
  int width, height;
  int hsub, vsub;          // chroma sub-sampling factors (1 or 2)
  int cw, ch;              // real (unpadded) chroma plane size: ceil(width / hsub), ceil(height / vsub)
  int colour;              // KLAB_JPEG_GRAY / _YCC / _RGB
  long long px0;           // first pixel of this image in the colour grid
};

__device__ __forceinline__ int chroma_at(const unsigned char* __restrict__ pl, int stride, int x, int y, const JpegImage& im) {
  if (im.hsub == 1 && im.vsub == 1) return pl[(long long)y * stride + x];
  const bool fancy = im.cw > 2;
  const int i = x >> 1;
  if (im.vsub == 1) {  // 2x1
    const unsigned char* row = pl + (long long)y * stride;
    if (!fancy) return row[i];
    const int s = row[i];
    if (x & 1) { const int nb = row[i + 1 < im.cw ? i + 1 : i]; return (3 * s + nb + 2) >> 2; }
    const int nb = row[i > 0 ? i - 1 : 0];
    return (3 * s + nb + 1) >> 2;
  }
  // 2x2
  const int r0 = y >> 1;
  if (!fancy) return pl[(long long)r0 * stride + i];
  int r1 = (y & 1) ? r0 + 1 : r0 - 1;
  r1 = r1 < 0 ? 0 : (r1 >= im.ch ? im.ch - 1 : r1);
  const unsigned char* a = pl + (long long)r0 * stride;
  const unsigned char* bb = pl + (long long)r1 * stride;
  const int cs = 3 * a[i] + bb[i];
  if (x & 1) {
    const int j = i + 1 < im.cw ? i + 1 : i;
    return (3 * cs + (3 * a[j] + bb[j]) + 7) >> 4;
  }
  const int j = i > 0 ? i - 1 : 0;
  return (3 * cs + (3 * a[j] + bb[j]) + 8) >> 4;
}

__device__ __forceinline__ unsigned char clamp255(int v) { return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

__global__ __launch_bounds__(256) void jpeg_colour_kernel(const unsigned char* __restrict__ ws, const JpegImage* __restrict__ imgs, int nimg,
                                                          long long total_px, unsigned char* __restrict__ dst) {
  const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
  if (g >= total_px) return;
  int lo = 0, hi = nimg - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (imgs[mid].px0 <= g) lo = mid; else hi = mid - 1;
  }
  const JpegImage im = imgs[lo];
  const long long l = g - im.px0;
  const int y = (int)(l / im.width), x = (int)(l - (long long)y * im.width);
  const int Y = ws[im.plane_off[0] + (long long)y * im.stride[0] + x];
  unsigned char* o = dst + im.rgb_off + l * 3;
  if (im.colour == KLAB_JPEG_GRAY) { o[0] = o[1] = o[2] = (unsigned char)Y; return; }
  const int c1 = chroma_at(ws + im.plane_off[1], im.stride[1], x, y, im);
  const int c2 = chroma_at(ws + im.plane_off[2], im.stride[2], x, y, im);
  if (im.colour == KLAB_JPEG_RGB) { o[0] = (unsigned char)Y; o[1] = (unsigned char)c1; o[2] = (unsigned char)c2; return; }
  const int cb = c1 - 128, cr = c2 - 128;
  const int r = Y + ((91881 * cr + 32768) >> 16);
  const int gg = Y + ((-22554 * cb + 32768 - 46802 * cr) >> 16);
  const int bl = Y + ((116130 * cb + 32768) >> 16);
  o[0] = clamp255(r); o[1] = clamp255(gg); o[2] = clamp255(bl);
}

// builds the two lookup tables in the workspace from the caller's items (one thread: n is a batch size)
__global__ void jpeg_tables_kernel(const klab_jpeg_item* __restrict__ items, int n, JpegPlane* __restrict__ planes, JpegImage* __restrict__ imgs,
                                   long long first_plane_off) {
  if (blockIdx.x || threadIdx.x) return;
  long long off = first_plane_off, wg = 0, px = 0;
  int np = 0;
  for (int i = 0; i < n; ++i) {
    const klab_jpeg_info& f = items[i].info;
    JpegImage im;
    im.plane_off[1] = im.plane_off[2] = 0; im.stride[1] = im.stride[2] = 0;
    long long b0 = items[i].coef_block0;
    for (int c = 0; c < f.ncomp && c < 3; ++c) {
      JpegPlane p;
      p.block0 = b0; p.plane_off = off; p.wg0 = wg; p.bw = f.bw[c]; p.bh = f.bh[c]; p.qt = i * 3 + c;
      planes[np++] = p;
      im.plane_off[c] = off; im.stride[c] = f.bw[c] * 8;
      const long long nb = (long long)f.bw[c] * f.bh[c];
      b0 += nb; wg += (nb + 31) / 32; off += (nb * 64 + 255) & ~255LL;
    }
    im.rgb_off = items[i].rgb_off; im.width = f.width; im.height = f.height;
    im.hsub = f.ncomp == 3 ? f.hmax / f.hs[1] : 1; im.vsub = f.ncomp == 3 ? f.vmax / f.vs[1] : 1;
    im.cw = (f.width + im.hsub - 1) / im.hsub; im.ch = (f.height + im.vsub - 1) / im.vsub;
    im.colour = f.colour; im.px0 = px;
    px += (long long)f.width * f.height;
    imgs[i] = im;
  }
}

}  // namespace klab

using namespace klab;

static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

// Workspace: [plane table | image table | sample planes]; the tables are built on the device from `items_dev`.
extern "C" size_t klab_jpeg_decode_ws_bytes(const klab_jpeg_item* items, int n) {
  if (!items || n <= 0) return 0;
  size_t bytes = al256(sizeof(JpegPlane) * 3 * (size_t)n) + al256(sizeof(JpegImage) * (size_t)n);
  for (int i = 0; i < n; ++i)
    for (int c = 0; c < items[i].info.ncomp && c < 3; ++c) bytes += al256((size_t)items[i].info.bw[c] * items[i].info.bh[c] * 64);
  return bytes;
}

extern "C" int klab_jpeg_decode_device(const short* coefs_dev, const unsigned short* qt_dev, const klab_jpeg_item* items,
                                       const klab_jpeg_item* items_dev, int n, unsigned char* rgb_dev, void* ws, size_t ws_bytes,
                                       void* stream) {
  if (!coefs_dev || !qt_dev || !items || !items_dev || !rgb_dev || !ws || n <= 0) return KLAB_ERR_BADARG;
  if (ws_bytes < klab_jpeg_decode_ws_bytes(items, n)) return KLAB_ERR_BADARG;
  hipStream_t s = (hipStream_t)stream;
  long long wg = 0, px = 0;
  int np = 0;
  for (int i = 0; i < n; ++i) {  // the host copy of the items sizes the grids and is validated; the device copy feeds the kernels
    const klab_jpeg_info& f = items[i].info;
    if (!f.supported || f.width <= 0 || f.height <= 0 || (f.ncomp != 1 && f.ncomp != 3) || f.hmax < 1 || f.vmax < 1) return KLAB_ERR_UNSUPPORTED;
    if (items[i].coef_block0 < 0 || items[i].rgb_off < 0) return KLAB_ERR_BADARG;
    for (int c = 0; c < f.ncomp; ++c) {
      if (f.hs[c] < 1 || f.vs[c] < 1 || f.bw[c] <= 0 || f.bh[c] <= 0) return KLAB_ERR_BADARG;
      // the padded block grid must cover the component: the colour kernel indexes it with image coordinates
      if ((long long)f.bw[c] * 8 * f.hmax < (long long)f.width * f.hs[c] || (long long)f.bh[c] * 8 * f.vmax < (long long)f.height * f.vs[c])
        return KLAB_ERR_BADARG;
      wg += ((long long)f.bw[c] * f.bh[c] + 31) / 32;
      ++np;
    }
    if (f.ncomp == 3 && (f.hs[0] != f.hmax || f.vs[0] != f.vmax || f.hs[1] != 1 || f.vs[1] != 1 || f.hs[2] != 1 || f.vs[2] != 1 || f.hmax > 2 ||
                         f.vmax > 2 || (f.hmax == 1 && f.vmax == 2)))
      return KLAB_ERR_UNSUPPORTED;
    px += (long long)f.width * f.height;
  }
  if (wg > 0x7fffffffLL || (px + 255) / 256 > 0x7fffffffLL) return KLAB_ERR_UNSUPPORTED;
  char* base = (char*)ws;
  JpegPlane* planes_dev = (JpegPlane*)base;
  JpegImage* imgs_dev = (JpegImage*)(base + al256(sizeof(JpegPlane) * 3 * (size_t)n));
  const long long first = (long long)(al256(sizeof(JpegPlane) * 3 * (size_t)n) + al256(sizeof(JpegImage) * (size_t)n));
  hipLaunchKernelGGL(jpeg_tables_kernel, dim3(1), dim3(1), 0, s, items_dev, n, planes_dev, imgs_dev, first);
  KLAB_LAUNCH_CHECK();
  hipLaunchKernelGGL(jpeg_idct_kernel, dim3((unsigned)wg), dim3(256), 0, s, coefs_dev, qt_dev, planes_dev, np, (unsigned char*)ws);
  KLAB_LAUNCH_CHECK();
  hipLaunchKernelGGL(jpeg_colour_kernel, dim3((unsigned)((px + 255) / 256)), dim3(256), 0, s, (const unsigned char*)ws, imgs_dev, n, px, rgb_dev);
  KLAB_LAUNCH_CHECK();
  return KLAB_OK;
}
