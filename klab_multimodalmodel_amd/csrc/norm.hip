// Row-normalisation kernels (HBM-bound): T5 RMS-norm (HF/t5:59-72) and Swin-V2 res-post-norm
// LayerNorm (HF/swinv2:697-702, 242, 354, 953), forward and backward.
// One wave (64 lanes) per row, 16-B vector accesses, fp32 statistics, wave-shuffle reductions.
// Outputs can be written through a row remap (row -> (row / grp) * grp_stride + row % grp + off)
// so that the Swin final LayerNorm and the language-encoder final RMS-norm write straight into the
// [B, N_img + Ls, d] encoder-input buffer: the torch.cat of ref/models/model.py:23 costs nothing.
#include "common.h"
#include <stdlib.h>
#include "klab_mm.h"

namespace klab {

__device__ __forceinline__ long remap_row(long row, int grp, int grp_stride, int off) {
  if (grp <= 0) return row;
  return (row / grp) * (long)grp_stride + (row % grp) + off;
}

template <typename T>
__device__ __forceinline__ void store4(T* p, float a, float b, float c, float d);
template <> __device__ __forceinline__ void store4<float>(float* p, float a, float b, float c, float d) {
  *reinterpret_cast<f32x4*>(p) = f32x4{a, b, c, d};
}
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* p, float a, float b, float c, float d) {
  *reinterpret_cast<bf16x4*>(p) = bf16x4{(bf16_t)a, (bf16_t)b, (bf16_t)c, (bf16_t)d};
}
template <typename T> __device__ __forceinline__ f32x4 load4(const T* p);
template <> __device__ __forceinline__ f32x4 load4<float>(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
template <> __device__ __forceinline__ f32x4 load4<bf16_t>(const bf16_t* p) {
  bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
  return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}

// ------------------------------------------------------------------------------------------
// RMS-norm forward: y = drop(x * rsqrt(mean(x^2)+eps) * w)
// ------------------------------------------------------------------------------------------
template <typename TY, bool DROP = true>
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          TY* __restrict__ y, float* __restrict__ y32, float* __restrict__ rstd, int rows,
                                                          int d, float eps, int grp, int grp_stride, int off, float p,
                                                          const uint32_t* seed, uint32_t tag) {
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  const DropCtx dc = make_drop(seed, tag, p);
  for (long row = (long)blockIdx.x * wpb + (threadIdx.x >> 6); row < rows; row += (long)gridDim.x * wpb) {
    const float* xr = x + row * d;
    float ss = 0.f;
    for (int c = lane * 4; c < d; c += 256) {
      f32x4 v = *reinterpret_cast<const f32x4*>(xr + c);
      ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    ss = wave_sum(ss);
    const float r = rsqrtf(ss / (float)d + eps);
    if (lane == 0 && rstd) rstd[row] = r;
    const long orow = remap_row(row, grp, grp_stride, off);
    for (int c = lane * 4; c < d; c += 256) {
      f32x4 v = *reinterpret_cast<const f32x4*>(xr + c);
      f32x4 g = *reinterpret_cast<const f32x4*>(w + c);
      float o[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        o[i] = g[i] * (v[i] * r);
        if constexpr (DROP) o[i] *= drop_mult(dc, (uint64_t)orow * d + c + i);  // (most norms have no output dropout: no hash, no branches)
      }
      if (y) store4<TY>(y + orow * d + c, o[0], o[1], o[2], o[3]);
      if (y32) store4<float>(y32 + orow * d + c, o[0], o[1], o[2], o[3]);
    }
  }
}

// RMS-norm forward that ALSO emits the row in OCP e4m3 with its dequantisation scale (fp8 mode: the per-token quantisation pass in
// front of the next Linear, klab_quant_fp8_rows, folded into the kernel that already owns the row).  Bit-identical to norm followed
// by the separate pass: the e4m3 value is taken from the bf16-ROUNDED output, amax / 448 is the scale, 1 for an all-zero row.
template <int SLOTS>
__global__ __launch_bounds__(256) void rmsnorm_fwd_q8_kernel(const float* __restrict__ x, const float* __restrict__ w, bf16_t* __restrict__ y,
                                                             float* __restrict__ rstd, unsigned char* __restrict__ y8,
                                                             float* __restrict__ yscale, int rows, int d, float eps, float p,
                                                             const uint32_t* seed, uint32_t tag) {
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  const DropCtx dc = make_drop(seed, tag, p);
  for (long row = (long)blockIdx.x * wpb + (threadIdx.x >> 6); row < rows; row += (long)gridDim.x * wpb) {
    const float* xr = x + row * d;
    f32x4 v[SLOTS];
    float ss = 0.f;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      const int c = lane * 4 + s * 256;
      v[s] = c < d ? *reinterpret_cast<const f32x4*>(xr + c) : f32x4{0.f, 0.f, 0.f, 0.f};
      ss += v[s][0] * v[s][0] + v[s][1] * v[s][1] + v[s][2] * v[s][2] + v[s][3] * v[s][3];
    }
    ss = wave_sum(ss);
    const float r = rsqrtf(ss / (float)d + eps);
    if (lane == 0 && rstd) rstd[row] = r;
    float amax = 0.f;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      const int c = lane * 4 + s * 256;
      if (c < d) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(w + c);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const bf16_t o = (bf16_t)(g[i] * (v[s][i] * r) * drop_mult(dc, (uint64_t)row * d + c + i));
          v[s][i] = (float)o;  // the value the GEMM's bf16 form would read
          amax = fmaxf(amax, fabsf(v[s][i]));
        }
        store4<bf16_t>(y + row * d + c, v[s][0], v[s][1], v[s][2], v[s][3]);
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    const float sc = amax > 0.f ? amax * (1.f / 448.f) : 1.f;
    const float inv = 1.f / sc;
    if (lane == 0) yscale[row] = sc;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      const int c = lane * 4 + s * 256;
      if (c < d) {
        float f[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] = fminf(fmaxf(v[s][i] * inv, -448.f), 448.f);
        int wd = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], 0, false);
        wd = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], wd, true);
        *reinterpret_cast<int*>(y8 + row * d + c) = wd;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// RMS-norm backward.
//   dy_eff = dy * dropmult_y            (dropout that followed the norm, if any)
//   dx     = dres + r*w*dy_eff - x * r^3/d * sum(dy_eff*w*x)
//   dw    += sum_rows dy_eff * x * r    (per-wave register partials, then one f32 atomic per column)
//   dxt    = T(dx * dropmult_prev)      (grad of the previous sub-layer's GEMM output, ready as a
//                                        bf16 GEMM operand: the residual add + dropout of HF/t5:400,141)
// ------------------------------------------------------------------------------------------
template <typename TY>
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                          const float* __restrict__ w, const float* __restrict__ rstd,
                                                          const float* __restrict__ dres, float* __restrict__ dx,
                                                          TY* __restrict__ dxt, int rows, int d,
                                                          int grp, int grp_stride, int off, float p_y, uint32_t tag_y,
                                                          float p_prev, uint32_t tag_prev, const uint32_t* seed) {
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  const DropCtx dcy = make_drop(seed, tag_y, p_y);
  const DropCtx dcp = make_drop(seed, tag_prev, p_prev);
  for (long row = (long)blockIdx.x * wpb + (threadIdx.x >> 6); row < rows; row += (long)gridDim.x * wpb) {
    const float* xr = x + row * d;
    const long yrow = remap_row(row, grp, grp_stride, off);  // dy lives in the remapped space
    const float* dyr = dy + yrow * d;
    const float r = rstd[row];
    float dot = 0.f;
    for (int c = lane * 4; c < d; c += 256) {
      f32x4 v = *reinterpret_cast<const f32x4*>(xr + c);
      f32x4 g = *reinterpret_cast<const f32x4*>(w + c);
      f32x4 e = *reinterpret_cast<const f32x4*>(dyr + c);
#pragma unroll
      for (int i = 0; i < 4; ++i) dot += e[i] * drop_mult(dcy, (uint64_t)yrow * d + c + i) * g[i] * v[i];
    }
    dot = wave_sum(dot);
    const float k = dot * r * r * r / (float)d;
    for (int c = lane * 4; c < d; c += 256) {
      f32x4 v = *reinterpret_cast<const f32x4*>(xr + c);
      f32x4 g = *reinterpret_cast<const f32x4*>(w + c);
      f32x4 e = *reinterpret_cast<const f32x4*>(dyr + c);
      float o[4], ot[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float de = e[i] * drop_mult(dcy, (uint64_t)yrow * d + c + i);
        o[i] = r * g[i] * de - v[i] * k;
      }
      if (dres) {
        f32x4 q = *reinterpret_cast<const f32x4*>(dres + row * d + c);
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] += q[i];
      }
      if (dx) store4<float>(dx + row * d + c, o[0], o[1], o[2], o[3]);
      if (dxt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) ot[i] = o[i] * drop_mult(dcp, (uint64_t)row * d + c + i);
        store4<TY>(dxt + row * d + c, ot[0], ot[1], ot[2], ot[3]);
      }
    }
  }
}

// Fused row + weight-gradient variant for d <= 256*SLOTS: the row's x / dy_eff stay in registers between the two passes,
// each lane keeps the dw partial of its own columns over the block's rows, the 4 waves fold through LDS and a block
// issues ONE f32 atomic per column (<= KLAB_RMS_BLOCKS-way contention; the un-reduced per-wave atomics were 10x slower).
// IDX32: rows * d < 2^32, so an element's dropout index fits 32 bits and the one-round hash of drop_mult (high word zero) is
// evaluated without its two branches (drop_mult32_nb gives the identical multiplier): sixteen copies of them per row otherwise cut
// the unrolled body into ~56 basic blocks
template <typename TY, int SLOTS, bool IDX32 = false>
__global__ __launch_bounds__(1024) void rmsnorm_bwd_dw_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                             const float* __restrict__ w, const float* __restrict__ rstd,
                                                             const float* __restrict__ dres, float* __restrict__ dx,
                                                             TY* __restrict__ dxt, float* __restrict__ dw, int rows, int d,
                                                             int grp, int grp_stride, int off, float p_y, uint32_t tag_y,
                                                             float p_prev, uint32_t tag_prev, const uint32_t* seed, int partial) {
  constexpr int WPB = 16;  // 16 waves per workgroup: one row per wave in flight (the kernel is latency-bound), and the
                           // weight-gradient partials of 16+ rows fold into ONE atomic per column
  __shared__ float red[WPB][256 * SLOTS];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const DropCtx dcy = make_drop(seed, tag_y, p_y);
  const DropCtx dcp = make_drop(seed, tag_prev, p_prev);
  f32x4 g[SLOTS], acc[SLOTS];
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) {
    const int c = lane * 4 + s * 256;
    g[s] = c < d ? *reinterpret_cast<const f32x4*>(w + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    acc[s] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  for (long row = (long)blockIdx.x * WPB + wv; row < rows; row += (long)gridDim.x * WPB) {
    const float* xr = x + row * d;
    const long yrow = remap_row(row, grp, grp_stride, off);
    const float* dyr = dy + yrow * d;
    const float r = rstd[row];
    f32x4 v[SLOTS], e[SLOTS], q[SLOTS];
    float dot = 0.f;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {  // every load of the row goes out before the first use (one memory round trip)
      const int c = lane * 4 + s * 256;
      q[s] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (c < d) {
        v[s] = *reinterpret_cast<const f32x4*>(xr + c);
        e[s] = *reinterpret_cast<const f32x4*>(dyr + c);
        if (dres) q[s] = *reinterpret_cast<const f32x4*>(dres + row * d + c);
      }
    }
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      const int c = lane * 4 + s * 256;
      if (c < d) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if constexpr (IDX32) e[s][i] *= drop_mult32_nb(dcy, (uint32_t)yrow * (uint32_t)d + c + i);
          else e[s][i] *= drop_mult(dcy, (uint64_t)yrow * d + c + i);
          dot += e[s][i] * g[s][i] * v[s][i];
          acc[s][i] += e[s][i] * v[s][i] * r;
        }
      }
    }
    dot = wave_sum(dot);
    const float k = dot * r * r * r / (float)d;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      const int c = lane * 4 + s * 256;
      if (c < d) {
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = r * g[s][i] * e[s][i] - v[s][i] * k + q[s][i];
        if (dx) store4<float>(dx + row * d + c, o[0], o[1], o[2], o[3]);
        if (dxt) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if constexpr (IDX32) o[i] *= drop_mult32_nb(dcp, (uint32_t)row * (uint32_t)d + c + i);
            else o[i] *= drop_mult(dcp, (uint64_t)row * d + c + i);
          }
          store4<TY>(dxt + row * d + c, o[0], o[1], o[2], o[3]);
        }
      }
    }
  }
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) *reinterpret_cast<f32x4*>(&red[wv][lane * 4 + s * 256]) = acc[s];
  __syncthreads();
  for (int c = threadIdx.x; c < d; c += 1024) {
    float a = 0.f;
#pragma unroll
    for (int w = 0; w < WPB; ++w) a += red[w][c];
    if (partial) dw[(long)blockIdx.x * d + c] = a;  // dw = per-workgroup partials [gridDim.x, d]: reduced once per stack, in a fixed order
    else atomicAdd(dw + c, a);
  }
}

// dw[c] += sum_rows dy_eff[row,c] * x[row,c] * rstd[row]: a column reduction, kept out of the row kernel
// (per-wave register partials + 512-way same-address atomics made that kernel 10x slower than its bytes).
// Block = 64 columns x 4 row lanes over a slab of rows; one 256-B atomic wave-instruction per block.
__global__ __launch_bounds__(256) void rmsnorm_dw_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ rstd,
                                                         float* __restrict__ dw, int rows, int d, int grp, int grp_stride, int off, float p_y,
                                                         uint32_t tag_y, const uint32_t* seed) {
  __shared__ float red[4][64];
  const DropCtx dcy = make_drop(seed, tag_y, p_y);
  const int col = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  float a = 0.f;
  if (col < d) {
    for (long row = (long)blockIdx.y * 4 + rl; row < rows; row += (long)gridDim.y * 4) {
      const long yrow = remap_row(row, grp, grp_stride, off);
      a += dy[yrow * d + col] * drop_mult(dcy, (uint64_t)yrow * d + col) * x[row * d + col] * rstd[row];
    }
  }
  red[rl][threadIdx.x & 63] = a;
  __syncthreads();
  if (rl == 0 && col < d) atomicAdd(dw + col, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// ------------------------------------------------------------------------------------------
// LayerNorm forward (Swin-V2): out = shortcut + (LN(y) * gamma + beta); optional T copy.
// ------------------------------------------------------------------------------------------
// LPR = lanes per row (16 for C = 64, 32 for C = 128, 64 otherwise): narrow Swin stage-0/1 rows are packed 4 / 2 per
// wave so that every lane streams 16 B (with one row per wave, C = 64 kept 16 of 64 lanes busy).
template <typename TI, typename TO, int LPR, bool Q8 = false>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const TI* __restrict__ y, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, const float* __restrict__ shortcut,
                                                            float* __restrict__ out, TO* __restrict__ outt, float* __restrict__ mean,
                                                            float* __restrict__ rstd, int rows, int C, float eps, int grp,
                                                            int grp_stride, int off, float p, const uint32_t* seed, uint32_t tag,
                                                            unsigned char* __restrict__ o8 = nullptr, float* __restrict__ oscale = nullptr) {
  constexpr int RPW = 64 / LPR;  // rows per wave
  const int lane = threadIdx.x & 63;
  const int sub = lane / LPR, l = lane % LPR;
  const int wpb = blockDim.x >> 6;
  const DropCtx dc = make_drop(seed, tag, p);
  for (long row0 = ((long)blockIdx.x * wpb + (threadIdx.x >> 6)) * RPW; row0 < rows; row0 += (long)gridDim.x * wpb * RPW) {
    const long row = row0 + sub;
    const bool live = row < rows;
    const TI* yr = y + (live ? row : 0) * C;
    float s = 0.f;
    if (live)
      for (int c = l * 4; c < C; c += LPR * 4) {
        f32x4 v = load4<TI>(yr + c);
        s += v[0] + v[1] + v[2] + v[3];
      }
    const float mu = group_sum<LPR>(s) / (float)C;
    float ss = 0.f;
    if (live)
      for (int c = l * 4; c < C; c += LPR * 4) {
        f32x4 v = load4<TI>(yr + c);
#pragma unroll
        for (int i = 0; i < 4; ++i) { float t = v[i] - mu; ss += t * t; }
      }
    const float r = rsqrtf(group_sum<LPR>(ss) / (float)C + eps);
    if (!live) continue;
    if (l == 0) { if (mean) mean[row] = mu; if (rstd) rstd[row] = r; }
    const long orow = remap_row(row, grp, grp_stride, off);
    float amax = 0.f;
    (void)amax;
    for (int c = l * 4; c < C; c += LPR * 4) {
      f32x4 v = load4<TI>(yr + c);
      f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c);
      f32x4 b = *reinterpret_cast<const f32x4*>(beta + c);
      float o[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = (v[i] - mu) * r * g[i] + b[i];
      if (shortcut) {
        f32x4 q = *reinterpret_cast<const f32x4*>(shortcut + row * C + c);
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] += q[i];
      }
      if (dc.on) {  // (one uniform test per chunk instead of one per element inside drop_mult)
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] *= drop_mult(dc, (uint64_t)orow * C + c + i);
      }
      if (out) store4<float>(out + orow * C + c, o[0], o[1], o[2], o[3]);
      if (outt) store4<TO>(outt + orow * C + c, o[0], o[1], o[2], o[3]);
      if constexpr (Q8) {
#pragma unroll
        for (int i = 0; i < 4; ++i) amax = fmaxf(amax, fabsf((float)(bf16_t)o[i]));
      }
    }
    if constexpr (Q8) {
      // fp8 mode: the same row in e4m3 with its scale (klab_quant_fp8_rows of the bf16 copy, folded in): second sweep over the row
      // (cache-hot) once the row maximum is known
#pragma unroll
      for (int o2 = LPR / 2; o2 > 0; o2 >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o2, 64));
      const float sc = amax > 0.f ? amax * (1.f / 448.f) : 1.f;
      const float inv = 1.f / sc;
      if (l == 0) oscale[orow] = sc;
      for (int c = l * 4; c < C; c += LPR * 4) {
        f32x4 v = load4<TI>(yr + c);
        f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c);
        f32x4 b = *reinterpret_cast<const f32x4*>(beta + c);
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = (v[i] - mu) * r * g[i] + b[i];
        if (shortcut) {
          f32x4 q = *reinterpret_cast<const f32x4*>(shortcut + row * C + c);
#pragma unroll
          for (int i = 0; i < 4; ++i) o[i] += q[i];
        }
        float f[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          o[i] *= drop_mult(dc, (uint64_t)orow * C + c + i);
          f[i] = fminf(fmaxf((float)(bf16_t)o[i] * inv, -448.f), 448.f);
        }
        int wd = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], 0, false);
        wd = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], wd, true);
        *reinterpret_cast<int*>(o8 + orow * C + c) = wd;
      }
    }
  }
}

// LayerNorm backward: given dout (grad of `out`, in the remapped row space, with the same output
// dropout), y, mean, rstd:  dy = LN'(dout) as T (GEMM operand), dgamma/dbeta via f32 atomics.
// The shortcut gradient is dout itself (caller keeps using it as the residual-stream gradient).
template <typename TI>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dout, const TI* __restrict__ y,
                                                            const float* __restrict__ gamma, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, TI* __restrict__ dy, int rows, int C,
                                                            int grp, int grp_stride, int off, float p, const uint32_t* seed,
                                                            uint32_t tag) {
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  const DropCtx dc = make_drop(seed, tag, p);
  for (long row = (long)blockIdx.x * wpb + (threadIdx.x >> 6); row < rows; row += (long)gridDim.x * wpb) {
    const TI* yr = y + row * C;
    const long orow = remap_row(row, grp, grp_stride, off);
    const float* dor = dout + orow * C;
    const float mu = mean[row], r = rstd[row];
    float s1 = 0.f, s2 = 0.f;  // sum(g*do), sum(g*do*xhat)
    for (int c = lane * 4; c < C; c += 256) {
      f32x4 v = load4<TI>(yr + c);
      f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c);
      f32x4 e = *reinterpret_cast<const f32x4*>(dor + c);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float de = e[i] * drop_mult(dc, (uint64_t)orow * C + c + i);
        const float xh = (v[i] - mu) * r;
        s1 += g[i] * de;
        s2 += g[i] * de * xh;
      }
    }
    s1 = wave_sum(s1) / (float)C;
    s2 = wave_sum(s2) / (float)C;
    for (int c = lane * 4; c < C; c += 256) {
      f32x4 v = load4<TI>(yr + c);
      f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c);
      f32x4 e = *reinterpret_cast<const f32x4*>(dor + c);
      float o[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float de = e[i] * drop_mult(dc, (uint64_t)orow * C + c + i);
        const float xh = (v[i] - mu) * r;
        o[i] = r * (g[i] * de - s1 - xh * s2);
      }
      if (dy) store4<TI>(dy + row * C + c, o[0], o[1], o[2], o[3]);
    }
  }
}

template <typename TI>
__global__ __launch_bounds__(256) void layernorm_dgb_kernel(const float* __restrict__ dout, const TI* __restrict__ y, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                            int rows, int C, int grp, int grp_stride, int off, float p, const uint32_t* seed,
                                                            uint32_t tag) {
  __shared__ float rg[4][64], rb[4][64];
  const DropCtx dc = make_drop(seed, tag, p);
  const int col = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  float ag = 0.f, ab = 0.f;
  if (col < C) {
    for (long row = (long)blockIdx.y * 4 + rl; row < rows; row += (long)gridDim.y * 4) {
      const long orow = remap_row(row, grp, grp_stride, off);
      const float de = dout[orow * C + col] * drop_mult(dc, (uint64_t)orow * C + col);
      ag += de * (to_f32(y[row * C + col]) - mean[row]) * rstd[row];
      ab += de;
    }
  }
  rg[rl][threadIdx.x & 63] = ag; rb[rl][threadIdx.x & 63] = ab;
  __syncthreads();
  if (rl == 0 && col < C) {
    const int t = threadIdx.x;
    if (dgamma) atomicAdd(dgamma + col, rg[0][t] + rg[1][t] + rg[2][t] + rg[3][t]);
    if (dbeta) atomicAdd(dbeta + col, rb[0][t] + rb[1][t] + rb[2][t] + rb[3][t]);
  }
}

// Fused LayerNorm backward for C <= 256*SLOTS (the trainable Swin tower, 52 calls per configs[2] step): one pass over
// (dout, y) gives the row result dy = LN'(dout) AND the three column sums that used to take two more kernels and two more reads
// of the activations: dgamma = sum dout*xhat, dbeta = sum dout, and dprev = sum dy -- the bias gradient of the Linear whose output
// this norm consumed (fc2 / the attention output projection).  One wave per row in flight, every lane keeps the partial sums of
// its own columns over the workgroup's rows, 16 waves fold through LDS and a workgroup issues one f32 atomic per column and sum.
template <typename TI, int SLOTS>
__global__ __launch_bounds__(1024) void layernorm_bwd_fused_kernel(const float* __restrict__ dout, const TI* __restrict__ y,
                                                                  const float* __restrict__ gamma, const float* __restrict__ mean,
                                                                  const float* __restrict__ rstd, TI* __restrict__ dy,
                                                                  float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                  float* __restrict__ dprev, int rows, int C, int grp, int grp_stride,
                                                                  int off, float p, const uint32_t* seed, uint32_t tag) {
  constexpr int WPB = 16;
  __shared__ float red[WPB][256 * SLOTS];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const DropCtx dc = make_drop(seed, tag, p);
  f32x4 g[SLOTS], ag[SLOTS], ab[SLOTS], ap[SLOTS];
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) {
    const int c = lane * 4 + s * 256;
    g[s] = c < C ? *reinterpret_cast<const f32x4*>(gamma + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    ag[s] = ab[s] = ap[s] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  for (long row = (long)blockIdx.x * WPB + wv; row < rows; row += (long)gridDim.x * WPB) {
    const TI* yr = y + row * C;
    const long orow = remap_row(row, grp, grp_stride, off);
    const float* dor = dout + orow * C;
    const float mu = mean[row], r = rstd[row];
    f32x4 xh[SLOTS], e[SLOTS];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {  // all loads of the row go out before the first use
      const int c = lane * 4 + s * 256;
      xh[s] = e[s] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (c < C) { xh[s] = load4<TI>(yr + c); e[s] = *reinterpret_cast<const f32x4*>(dor + c); }
    }
    float s1 = 0.f, s2 = 0.f;  // sum(g*do), sum(g*do*xhat)
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      const int c = lane * 4 + s * 256;
      if (c < C) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (dc.on) e[s][i] *= drop_mult(dc, (uint64_t)orow * C + c + i);  // (uniform; Swin has no dropout here)
          xh[s][i] = (xh[s][i] - mu) * r;
          s1 += g[s][i] * e[s][i];
          s2 += g[s][i] * e[s][i] * xh[s][i];
          ag[s][i] += e[s][i] * xh[s][i];
          ab[s][i] += e[s][i];
        }
      }
    }
    s1 = wave_sum(s1) / (float)C;
    s2 = wave_sum(s2) / (float)C;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      const int c = lane * 4 + s * 256;
      if (c < C) {
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          o[i] = r * (g[s][i] * e[s][i] - s1 - xh[s][i] * s2);
          ap[s][i] += o[i];
        }
        store4<TI>(dy + row * C + c, o[0], o[1], o[2], o[3]);
      }
    }
  }
  auto fold = [&](const f32x4 (&a)[SLOTS], float* dst) {
    if (!dst) return;  // (uniform)
    __syncthreads();
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) *reinterpret_cast<f32x4*>(&red[wv][lane * 4 + s * 256]) = a[s];
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 1024) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < WPB; ++w) t += red[w][c];
      atomicAdd(dst + c, t);
    }
  };
  fold(ag, dgamma);
  fold(ab, dbeta);
  fold(ap, dprev);
}

static inline int norm_grid(int rows) {
  int g = (rows + 3) / 4;
  return g < 1 ? 1 : (g > 2048 ? 2048 : g);
}

}  // namespace klab

using namespace klab;

extern "C" int klab_rmsnorm_fwd(const float* x, const float* w, void* y, int y_dtype, float* y_f32, float* rstd, int rows,
                                int d, float eps, int grp, int grp_stride, int off, float drop_p,
                                const uint32_t* seed_dev, uint32_t tag, void* stream) {
  if (!x || !w || rows < 0 || d <= 0 || (d & 3)) return KLAB_ERR_BADARG;
  if (rows == 0) return KLAB_OK;
  hipStream_t s = (hipStream_t)stream;
  const bool drop = drop_p > 0.f && seed_dev;
#define RF(TY, DR)                                                                                                              \
  hipLaunchKernelGGL((rmsnorm_fwd_kernel<TY, DR>), dim3(norm_grid(rows)), dim3(256), 0, s, x, w, (TY*)y, y_f32, rstd, rows, d, eps, grp, \
                     grp_stride, off, drop_p, seed_dev, tag)
  if (y_dtype == KLAB_BF16) { if (drop) RF(bf16_t, true); else RF(bf16_t, false); }
  else { if (drop) RF(float, true); else RF(float, false); }
#undef RF
  KLAB_LAUNCH_CHECK();
  return KLAB_OK;
}

// y (bf16) = norm(x) * w, rstd, and y8 / yscale = the same rows in e4m3 with one scale per row (d <= 1024, d % 4 == 0)
extern "C" int klab_rmsnorm_fwd_q8(const float* x, const float* w, void* y_bf16, float* rstd, void* y8, float* yscale, int rows, int d,
                                   float eps, float drop_p, const uint32_t* seed_dev, uint32_t tag, void* stream) {
  if (!x || !w || !y_bf16 || !y8 || !yscale || rows < 0 || d <= 0 || (d & 3)) return KLAB_ERR_BADARG;
  if (d > 1024) return KLAB_ERR_UNSUPPORTED;
  if (rows == 0) return KLAB_OK;
  hipStream_t s = (hipStream_t)stream;
#define RQ(SL)                                                                                                                         \
  hipLaunchKernelGGL((rmsnorm_fwd_q8_kernel<SL>), dim3(norm_grid(rows)), dim3(256), 0, s, x, w, (bf16_t*)y_bf16, rstd, (unsigned char*)y8, \
                     yscale, rows, d, eps, drop_p, seed_dev, tag)
  if (d <= 256) RQ(1); else if (d <= 512) RQ(2); else RQ(4);
#undef RQ
  KLAB_LAUNCH_CHECK();
  return KLAB_OK;
}

static int rms_part_rows(int rows) {
  static const int nblk = [] { const char* v = getenv("KLAB_RMS_BLOCKS"); int n = v ? atoi(v) : 512; return n < 1 ? 1 : n; }();
  const int g16 = (rows + 15) / 16;
  return g16 < nblk ? g16 : nblk;
}
static int rms_fused_launch(const float* dy, const float* x, const float* w, const float* rstd, const float* dres, float* dx, void* dxt,
                            int dxt_dtype, float* dw, int partial, int rows, int d, int grp, int grp_stride, int off, float p_y,
                            uint32_t tag_y, float p_prev, uint32_t tag_prev, const uint32_t* seed_dev, hipStream_t s) {
  const int gf = rms_part_rows(rows);
  // (the remapped output row space is at most grp_stride / grp times larger than rows; 2^31 leaves that room)
  const bool idx32 = (long)rows * d < (1L << 31) && (grp <= 0 || (long)grp_stride * ((rows + grp - 1) / grp + 1) * d < (1L << 31));
#define RB_LAUNCH(TY, SL)                                                                                                      \
  do {                                                                                                                         \
    if (idx32)                                                                                                                 \
      hipLaunchKernelGGL((rmsnorm_bwd_dw_kernel<TY, SL, true>), dim3(gf), dim3(1024), 0, s, dy, x, w, rstd, dres, dx, (TY*)dxt, dw, rows, d, grp, \
                         grp_stride, off, p_y, tag_y, p_prev, tag_prev, seed_dev, partial);                                    \
    else                                                                                                                       \
      hipLaunchKernelGGL((rmsnorm_bwd_dw_kernel<TY, SL, false>), dim3(gf), dim3(1024), 0, s, dy, x, w, rstd, dres, dx, (TY*)dxt, dw, rows, d, grp, \
                         grp_stride, off, p_y, tag_y, p_prev, tag_prev, seed_dev, partial);                                    \
  } while (0)
  if (dxt_dtype == KLAB_BF16) { if (d <= 256) RB_LAUNCH(bf16_t, 1); else if (d <= 512) RB_LAUNCH(bf16_t, 2); else RB_LAUNCH(bf16_t, 4); }
  else { if (d <= 256) RB_LAUNCH(float, 1); else if (d <= 512) RB_LAUNCH(float, 2); else RB_LAUNCH(float, 4); }
#undef RB_LAUNCH
  KLAB_LAUNCH_CHECK();
  return KLAB_OK;
}

extern "C" int klab_rmsnorm_part_rows(int rows) { return rows > 0 ? rms_part_rows(rows) : 0; }

// as klab_rmsnorm_bwd, but the weight gradient is left as per-workgroup partial sums dw_part[klab_rmsnorm_part_rows(rows), d]
extern "C" int klab_rmsnorm_bwd_part(const float* dy, const float* x, const float* w, const float* rstd, const float* dres, float* dx,
                                     void* dxt, int dxt_dtype, float* dw_part, int rows, int d, int grp, int grp_stride, int off, float p_y,
                                     uint32_t tag_y, float p_prev, uint32_t tag_prev, const uint32_t* seed_dev, void* stream) {
  if (!dy || !x || !w || !rstd || !dw_part || rows <= 0 || d <= 0 || (d & 3)) return KLAB_ERR_BADARG;
  if (d > 1024) return KLAB_ERR_UNSUPPORTED;
  return rms_fused_launch(dy, x, w, rstd, dres, dx, dxt, dxt_dtype, dw_part, 1, rows, d, grp, grp_stride, off, p_y, tag_y, p_prev, tag_prev,
                          seed_dev, (hipStream_t)stream);
}

// dst[c][col] += sum_b part[c * call_stride + b * d + col]   (b < nparts): block = 64 columns x 4 lanes over the partials
// (fixed order: bit-reproducible); one thread per (call, column) walked 256 strided loads one after the other (23 us)
__global__ __launch_bounds__(256) void colpart_reduce_kernel(const float* __restrict__ part, long call_stride, int nparts, int d,
                                                             float* const* __restrict__ dst) {
  __shared__ float red[4][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  float a = 0.f;
  if (col < d) {
    const float* src = part + (long)blockIdx.y * call_stride + col;
#pragma unroll 8
    for (int b = rl; b < nparts; b += 4) a += src[(long)b * d];
  }
  red[rl][threadIdx.x & 63] = a;
  __syncthreads();
  if (rl == 0 && col < d) dst[blockIdx.y][col] += (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
extern "C" int klab_colpart_reduce(const float* part, long call_stride, int nparts, int d, float* const* dst_dev, int ncalls, void* stream) {
  if (!part || !dst_dev || nparts <= 0 || d <= 0 || ncalls <= 0) return KLAB_ERR_BADARG;
  hipLaunchKernelGGL(colpart_reduce_kernel, dim3((d + 63) / 64, ncalls), dim3(256), 0, (hipStream_t)stream, part, call_stride, nparts, d,
                     dst_dev);
  KLAB_LAUNCH_CHECK();
  return KLAB_OK;
}

extern "C" int klab_rmsnorm_bwd(const float* dy, const float* x, const float* w, const float* rstd, const float* dres,
                                float* dx, void* dxt, int dxt_dtype, float* dw, int rows, int d, int grp, int grp_stride,
                                int off, float p_y, uint32_t tag_y, float p_prev, uint32_t tag_prev,
                                const uint32_t* seed_dev, void* stream) {
  if (!dy || !x || !w || !rstd || rows < 0 || d <= 0 || (d & 3)) return KLAB_ERR_BADARG;
  if (rows == 0) return KLAB_OK;
  hipStream_t s = (hipStream_t)stream;
  const int g = norm_grid(rows);
  if (dw && d <= 1024)  // fused row + dw kernel
    return rms_fused_launch(dy, x, w, rstd, dres, dx, dxt, dxt_dtype, dw, 0, rows, d, grp, grp_stride, off, p_y, tag_y, p_prev, tag_prev, seed_dev, s);
  if (dxt_dtype == KLAB_BF16)
    hipLaunchKernelGGL(rmsnorm_bwd_kernel<bf16_t>, dim3(g), dim3(256), 0, s, dy, x, w, rstd, dres, dx, (bf16_t*)dxt, rows,
                       d, grp, grp_stride, off, p_y, tag_y, p_prev, tag_prev, seed_dev);
  else
    hipLaunchKernelGGL(rmsnorm_bwd_kernel<float>, dim3(g), dim3(256), 0, s, dy, x, w, rstd, dres, dx, (float*)dxt, rows, d,
                       grp, grp_stride, off, p_y, tag_y, p_prev, tag_prev, seed_dev);
  KLAB_LAUNCH_CHECK();
  if (dw) {
    int gy = (rows + 63) / 64;
    gy = gy < 1 ? 1 : (gy > 64 ? 64 : gy);
    hipLaunchKernelGGL(rmsnorm_dw_kernel, dim3((d + 63) / 64, gy), dim3(256), 0, s, dy, x, rstd, dw, rows, d, grp, grp_stride, off, p_y, tag_y,
                       seed_dev);
    KLAB_LAUNCH_CHECK();
  }
  return KLAB_OK;
}

extern "C" int klab_layernorm_fwd(const void* y, int y_dtype, const float* gamma, const float* beta, const float* shortcut,
                                  float* out, void* outt, int outt_dtype, float* mean, float* rstd, int rows, int C,
                                  float eps, int grp, int grp_stride, int off, float drop_p, const uint32_t* seed_dev,
                                  uint32_t tag, void* stream) {
  if (!y || !gamma || !beta || rows < 0 || C <= 0 || (C & 3)) return KLAB_ERR_BADARG;
  if (rows == 0) return KLAB_OK;
  hipStream_t s = (hipStream_t)stream;
  const int g = norm_grid(rows);
#define LN_LAUNCH(TI, TO)                                                                                                         \
  do {                                                                                                                            \
    if (C <= 64)                                                                                                                  \
      hipLaunchKernelGGL((layernorm_fwd_kernel<TI, TO, 16>), dim3(norm_grid((rows + 3) / 4)), dim3(256), 0, s, (const TI*)y, gamma, beta, \
                         shortcut, out, (TO*)outt, mean, rstd, rows, C, eps, grp, grp_stride, off, drop_p, seed_dev, tag);        \
    else if (C <= 128)                                                                                                            \
      hipLaunchKernelGGL((layernorm_fwd_kernel<TI, TO, 32>), dim3(norm_grid((rows + 1) / 2)), dim3(256), 0, s, (const TI*)y, gamma, beta, \
                         shortcut, out, (TO*)outt, mean, rstd, rows, C, eps, grp, grp_stride, off, drop_p, seed_dev, tag);        \
    else                                                                                                                          \
      hipLaunchKernelGGL((layernorm_fwd_kernel<TI, TO, 64>), dim3(g), dim3(256), 0, s, (const TI*)y, gamma, beta, shortcut, out,  \
                         (TO*)outt, mean, rstd, rows, C, eps, grp, grp_stride, off, drop_p, seed_dev, tag);                       \
  } while (0)
  if (y_dtype == KLAB_BF16 && outt_dtype == KLAB_BF16) LN_LAUNCH(bf16_t, bf16_t);
  else if (y_dtype == KLAB_BF16) LN_LAUNCH(bf16_t, float);
  else if (outt_dtype == KLAB_BF16) LN_LAUNCH(float, bf16_t);
  else LN_LAUNCH(float, float);
#undef LN_LAUNCH
  KLAB_LAUNCH_CHECK();
  return KLAB_OK;
}

// fp8 mode: as klab_layernorm_fwd with a bf16 `outt`, plus the same rows in e4m3 (o8 [rows, C]) and one scale per row
extern "C" int klab_layernorm_fwd_q8(const void* y, const float* gamma, const float* beta, const float* shortcut, float* out, void* outt,
                                     float* mean, float* rstd, void* o8, float* oscale, int rows, int C, float eps, void* stream) {
  if (!y || !gamma || !beta || !outt || !o8 || !oscale || rows < 0 || C <= 0 || (C & 3)) return KLAB_ERR_BADARG;
  if (rows == 0) return KLAB_OK;
  hipStream_t s = (hipStream_t)stream;
  const int g = norm_grid(rows);
  if (C <= 64)
    hipLaunchKernelGGL((layernorm_fwd_kernel<bf16_t, bf16_t, 16, true>), dim3(norm_grid((rows + 3) / 4)), dim3(256), 0, s, (const bf16_t*)y, gamma,
                       beta, shortcut, out, (bf16_t*)outt, mean, rstd, rows, C, eps, 0, 0, 0, 0.f, nullptr, 0, (unsigned char*)o8, oscale);
  else if (C <= 128)
    hipLaunchKernelGGL((layernorm_fwd_kernel<bf16_t, bf16_t, 32, true>), dim3(norm_grid((rows + 1) / 2)), dim3(256), 0, s, (const bf16_t*)y, gamma,
                       beta, shortcut, out, (bf16_t*)outt, mean, rstd, rows, C, eps, 0, 0, 0, 0.f, nullptr, 0, (unsigned char*)o8, oscale);
  else
    hipLaunchKernelGGL((layernorm_fwd_kernel<bf16_t, bf16_t, 64, true>), dim3(g), dim3(256), 0, s, (const bf16_t*)y, gamma, beta, shortcut, out,
                       (bf16_t*)outt, mean, rstd, rows, C, eps, 0, 0, 0, 0.f, nullptr, 0, (unsigned char*)o8, oscale);
  KLAB_LAUNCH_CHECK();
  return KLAB_OK;
}

static int layernorm_bwd_impl(const float* dout, const void* y, int y_dtype, const float* gamma, const float* mean, const float* rstd,
                              void* dy, float* dgamma, float* dbeta, float* dprev_bias, int rows, int C, int grp, int grp_stride, int off,
                              float drop_p, const uint32_t* seed_dev, uint32_t tag, hipStream_t s) {
  if (!dout || !y || !gamma || !mean || !rstd || rows < 0 || C <= 0 || (C & 3)) return KLAB_ERR_BADARG;
  if (dprev_bias && !dy) return KLAB_ERR_BADARG;
  if (rows == 0) return KLAB_OK;
  static const bool fused_on = [] { const char* e = getenv("KLAB_LN_BWD_FUSED"); return !e || atoi(e) != 0; }();
  if (dy && (dgamma || dbeta || dprev_bias) && C <= 1024 && fused_on) {
    const int gf = rms_part_rows(rows);
#define LNB(TI, SL)                                                                                                                   \
    hipLaunchKernelGGL((layernorm_bwd_fused_kernel<TI, SL>), dim3(gf), dim3(1024), 0, s, dout, (const TI*)y, gamma, mean, rstd, (TI*)dy, \
                       dgamma, dbeta, dprev_bias, rows, C, grp, grp_stride, off, drop_p, seed_dev, tag)
    if (y_dtype == KLAB_BF16) { if (C <= 256) LNB(bf16_t, 1); else if (C <= 512) LNB(bf16_t, 2); else LNB(bf16_t, 4); }
    else { if (C <= 256) LNB(float, 1); else if (C <= 512) LNB(float, 2); else LNB(float, 4); }
#undef LNB
    KLAB_LAUNCH_CHECK();
    return KLAB_OK;
  }
  const int g = norm_grid(rows);
  int gy = (rows + 63) / 64;
  gy = gy < 1 ? 1 : (gy > 128 ? 128 : gy);
  const dim3 g2((C + 63) / 64, gy);
  if (y_dtype == KLAB_BF16) {
    if (dy) hipLaunchKernelGGL(layernorm_bwd_kernel<bf16_t>, dim3(g), dim3(256), 0, s, dout, (const bf16_t*)y, gamma, mean, rstd,
                               (bf16_t*)dy, rows, C, grp, grp_stride, off, drop_p, seed_dev, tag);
    if (dgamma || dbeta) hipLaunchKernelGGL(layernorm_dgb_kernel<bf16_t>, g2, dim3(256), 0, s, dout, (const bf16_t*)y, mean, rstd, dgamma,
                                            dbeta, rows, C, grp, grp_stride, off, drop_p, seed_dev, tag);
  } else {
    if (dy) hipLaunchKernelGGL(layernorm_bwd_kernel<float>, dim3(g), dim3(256), 0, s, dout, (const float*)y, gamma, mean, rstd,
                               (float*)dy, rows, C, grp, grp_stride, off, drop_p, seed_dev, tag);
    if (dgamma || dbeta) hipLaunchKernelGGL(layernorm_dgb_kernel<float>, g2, dim3(256), 0, s, dout, (const float*)y, mean, rstd, dgamma,
                                            dbeta, rows, C, grp, grp_stride, off, drop_p, seed_dev, tag);
  }
  KLAB_LAUNCH_CHECK();
  if (dprev_bias) return klab_colsum(dy, C, y_dtype, rows, C, dprev_bias, (void*)s);  // wide rows: the separate column sum
  return KLAB_OK;
}

extern "C" int klab_layernorm_bwd(const float* dout, const void* y, int y_dtype, const float* gamma, const float* mean,
                                  const float* rstd, void* dy, float* dgamma, float* dbeta, int rows, int C, int grp,
                                  int grp_stride, int off, float drop_p, const uint32_t* seed_dev, uint32_t tag,
                                  void* stream) {
  return layernorm_bwd_impl(dout, y, y_dtype, gamma, mean, rstd, dy, dgamma, dbeta, nullptr, rows, C, grp, grp_stride, off, drop_p, seed_dev,
                            tag, (hipStream_t)stream);
}

// as klab_layernorm_bwd, and dprev_bias[c] += sum over rows of dy[:, c]: the bias gradient of the Linear layer whose output the
// norm consumed (its input gradient IS dy), folded into the same pass
extern "C" int klab_layernorm_bwd_bias(const float* dout, const void* y, int y_dtype, const float* gamma, const float* mean,
                                       const float* rstd, void* dy, float* dgamma, float* dbeta, float* dprev_bias, int rows, int C,
                                       int grp, int grp_stride, int off, float drop_p, const uint32_t* seed_dev, uint32_t tag,
                                       void* stream) {
  return layernorm_bwd_impl(dout, y, y_dtype, gamma, mean, rstd, dy, dgamma, dbeta, dprev_bias, rows, C, grp, grp_stride, off, drop_p,
                            seed_dev, tag, (hipStream_t)stream);
}
