// HBM-bound glue kernels of the path: weight cast/pack, embedding gather / scatter-add, relative
// position bias gather / scatter, LM-head cross-entropy (+ d logits in place), patch im2col,
// patch-merge gather / scatter, bias-gradient column sums.  All are coalesced 16-byte streams with
// wave-shuffle reductions; none is reshaped into a GEMM.
#include <math.h>

#include <map>
#include <mutex>

#include "common.h"
#include "klab_mm.h"

namespace klab {

int ensure_dyn_lds(const void* kernel, size_t bytes) {
  static std::mutex mu;
  static std::map<const void*, size_t> set;
  if (bytes > 160 * 1024) return KLAB_ERR_UNSUPPORTED;
  std::lock_guard<std::mutex> g(mu);
  auto it = set.find(kernel);
  if (it != set.end() && it->second >= bytes) return KLAB_OK;
  hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) return (int)e;
  set[kernel] = bytes;
  return KLAB_OK;
}

// ---- multi-tensor f32 -> T cast / pack ------------------------------------------------------
// The master weights stay fp32 nn.Parameters (the reference trains in fp32, ref/train.py:25-28);
// once per step the GEMM weights are cast into ONE arena in compute dtype, laid out so that q|k|v
// (and all decoder layers' cross k|v) are row-concatenated: fused projections need no copies.
struct CastDesc { const float* src; long dst_off; long n4_prefix; };  // prefix in units of 4 elements

template <typename T>
__global__ __launch_bounds__(256) void cast_pack_kernel(const CastDesc* __restrict__ dglob, int nd, long total4, T* __restrict__ dst) {
  // the descriptor table in LDS: the search below is a chain of dependent loads (see adam_step_kernel)
  constexpr int MAXD = 1024;
  __shared__ CastDesc dsh[MAXD];
  if (nd <= MAXD) {
    for (int i = threadIdx.x; i < nd; i += blockDim.x) dsh[i] = dglob[i];
    __syncthreads();
  }
  const CastDesc* d = nd <= MAXD ? dsh : dglob;
  // A thread converts U vec4's, blockDim apart (so every wave-instruction is a contiguous 1 KiB read), all U loads in
  // flight at once; the descriptor is found by ONE binary search per thread and then walked forward (a search per
  // vec4 was 7 dependent loads in front of every 16-byte read: 2.4 TB/s; tensors are far longer than U*256 vec4's).
  constexpr int U = 8;
  for (long g0 = ((long)blockIdx.x * U) * blockDim.x + threadIdx.x; g0 < total4; g0 += (long)gridDim.x * U * blockDim.x) {
    int lo = 0, hi = nd - 1;
    while (lo < hi) {  // last descriptor with prefix <= g0
      const int mid = (lo + hi + 1) >> 1;
      if (d[mid].n4_prefix <= g0) lo = mid; else hi = mid - 1;
    }
    f32x4 v[U];
    T* o[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long g = g0 + (long)u * blockDim.x;
      o[u] = nullptr;
      if (g < total4) {
        while (lo + 1 < nd && d[lo + 1].n4_prefix <= g) ++lo;
        const long local = (g - d[lo].n4_prefix) * 4;
        v[u] = *reinterpret_cast<const f32x4*>(d[lo].src + local);
        o[u] = dst + d[lo].dst_off + local;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (o[u]) {
        if constexpr (sizeof(T) == 2) *reinterpret_cast<bf16x4*>(o[u]) = bf16x4{(bf16_t)v[u][0], (bf16_t)v[u][1], (bf16_t)v[u][2], (bf16_t)v[u][3]};
        else *reinterpret_cast<f32x4*>(o[u]) = v[u];
      }
  }
}

// ---- fused multi-tensor Adam with a compute-dtype shadow copy (math of torch.optim.Adam, ref/train.py:28) ----
// One pass over the trainable tensors: p, g, m, v in (16-byte vectors), p, m, v out, and -- for GEMM weights -- the
// bf16/f32 copy the next forward's GEMMs read, written straight into the engine's weight arena (the separate cast pass
// over the masters disappears).  m / v live in flat buffers laid out like the flat gradient buffer.
struct AdamDesc { float* p; long goff; long aoff; long n4_prefix; };  // aoff < 0: no arena copy
struct AdamHyper { float lr_over_bc1, beta1, beta2, eps, weight_decay, inv_sqrt_bc2; int desc_in_lds; };

template <typename T, int U = 4, bool NT = false>
__global__ __launch_bounds__(256) void adam_step_kernel(const AdamDesc* __restrict__ dglob, int nd, long begin4, long total4, const float* __restrict__ grads,
                                                        float* __restrict__ m, float* __restrict__ v, T* __restrict__ arena, AdamHyper h) {
  // The descriptor table (one entry per tensor: ~130 for T5-small) is searched once per thread and iteration -- eight DEPENDENT loads,
  // then three more per vector for the tensor's addresses.  From global memory that chain (~1.5 us of L2 round trips) ran in front
  // of every batch of streaming loads and kept the kernel at 4.5 TB/s; a copy in LDS makes it ~100 cycles per step.
  constexpr int MAXD = 1024;  // (T5-large: ~560 tensors; 32 KiB)
  __shared__ AdamDesc dsh[MAXD];
  const bool in_lds = nd <= MAXD && h.desc_in_lds;
  if (in_lds) {
    for (int i = threadIdx.x; i < nd; i += blockDim.x) dsh[i] = dglob[i];
    __syncthreads();
  }
  const AdamDesc* d = in_lds ? dsh : dglob;
  // vec4 indices [begin4, total4) of the descriptor table's prefix space (a sub-range = the tensors of one backward segment)
  for (long g0 = begin4 + ((long)blockIdx.x * U) * blockDim.x + threadIdx.x; g0 < total4; g0 += (long)gridDim.x * U * blockDim.x) {
    int lo = 0, hi = nd - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (d[mid].n4_prefix <= g0) lo = mid; else hi = mid - 1;
    }
    f32x4 pv[U], gv[U], mv[U], vv[U];
    float* pp[U]; long go[U]; long ao[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long g = g0 + (long)u * blockDim.x;
      pp[u] = nullptr;
      if (g < total4) {
        while (lo + 1 < nd && d[lo + 1].n4_prefix <= g) ++lo;
        const long local = (g - d[lo].n4_prefix) * 4;
        pp[u] = d[lo].p + local; go[u] = d[lo].goff + local; ao[u] = d[lo].aoff < 0 ? -1 : d[lo].aoff + local;
        if constexpr (NT) {
          pv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(pp[u]));
          gv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(grads + go[u]));
          mv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(m + go[u]));
          vv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(v + go[u]));
        } else {
          pv[u] = *reinterpret_cast<const f32x4*>(pp[u]);
          gv[u] = *reinterpret_cast<const f32x4*>(grads + go[u]);
          mv[u] = *reinterpret_cast<const f32x4*>(m + go[u]);
          vv[u] = *reinterpret_cast<const f32x4*>(v + go[u]);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (pp[u]) {
        f32x4 po, mo, vo;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float g = gv[u][i] + h.weight_decay * pv[u][i];
          mo[i] = mv[u][i] + (1.f - h.beta1) * (g - mv[u][i]);         // exp_avg.lerp_(grad, 1 - beta1)
          vo[i] = h.beta2 * vv[u][i] + (1.f - h.beta2) * g * g;        // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
          const float denom = sqrtf(vo[i]) * h.inv_sqrt_bc2 + h.eps;
          po[i] = pv[u][i] - h.lr_over_bc1 * (mo[i] / denom);
        }
        if constexpr (NT) {
          __builtin_nontemporal_store(po, reinterpret_cast<f32x4*>(pp[u]));
          __builtin_nontemporal_store(mo, reinterpret_cast<f32x4*>(m + go[u]));
          __builtin_nontemporal_store(vo, reinterpret_cast<f32x4*>(v + go[u]));
        } else {
          *reinterpret_cast<f32x4*>(pp[u]) = po;
          *reinterpret_cast<f32x4*>(m + go[u]) = mo;
          *reinterpret_cast<f32x4*>(v + go[u]) = vo;
        }
        if (ao[u] >= 0) {
          if constexpr (sizeof(T) == 2) *reinterpret_cast<bf16x4*>(arena + ao[u]) = bf16x4{(bf16_t)po[0], (bf16_t)po[1], (bf16_t)po[2], (bf16_t)po[3]};
          else *reinterpret_cast<f32x4*>(arena + ao[u]) = po;
        }
      }
  }
}

// ---- embedding gather (+ T5 _shift_right, HF/t5:618-637) + input dropout (HF/t5:725) -----------
__global__ __launch_bounds__(256) void embed_fwd_kernel(const long long* __restrict__ ids, int shift_right, int L, int start_id, int pad_id,
                                                        const float* __restrict__ table, int vocab, float* __restrict__ out, int rows, int d,
                                                        float p, const uint32_t* seed, uint32_t tag, int* __restrict__ err) {
  const int lane = threadIdx.x & 63;
  const DropCtx dc = make_drop(seed, tag, p);
  for (long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (long)gridDim.x * 4) {
    long long id;
    if (shift_right) {
      const int t = row % L;
      id = t == 0 ? start_id : ids[row - 1];
      if (id == -100) id = pad_id;
    } else {
      id = ids[row];
    }
    if (id < 0 || id >= vocab) { if (lane == 0 && err) atomicOr(err, 1); id = 0; }
    const float* src = table + id * d;
    for (int c = lane * 4; c < d; c += 256) {
      f32x4 v = *reinterpret_cast<const f32x4*>(src + c);
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] *= drop_mult(dc, (uint64_t)row * d + c + i);
      *reinterpret_cast<f32x4*>(out + row * d + c) = v;
    }
  }
}

// scatter-add of d(hidden_0) into the tied embedding gradient (second contributor of shared.weight,
// SURVEY §2.4 "tied-weight note"); one wave per row => each atomic wave-instruction is 256
// contiguous bytes (MI355X atomic-rate shape).
__global__ __launch_bounds__(256) void embed_bwd_kernel(const long long* __restrict__ ids, int shift_right, int L, int start_id, int pad_id,
                                                        const float* __restrict__ dh, float* __restrict__ dtable, int vocab, int rows, int d,
                                                        float p, const uint32_t* seed, uint32_t tag) {
  const int lane = threadIdx.x & 63;
  const DropCtx dc = make_drop(seed, tag, p);
  for (long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (long)gridDim.x * 4) {
    long long id;
    if (shift_right) {
      const int t = row % L;
      id = t == 0 ? start_id : ids[row - 1];
      if (id == -100) id = pad_id;
    } else {
      id = ids[row];
    }
    if (id < 0 || id >= vocab) continue;
    float* dst = dtable + id * d;
    for (int c = lane; c < d; c += 64) atomicAdd(dst + c, dh[row * d + c] * drop_mult(dc, (uint64_t)row * d + c));
  }
}

// ---- T5 relative position bias (HF/t5:264-279): bias[h,i,j] = table[bucket[i,j], h] -----------
// bucket[Lq,Lk] is computed on the host with the reference's exact float-log arithmetic (HF/t5:216-262).
__global__ void relbias_fwd_kernel(const float* __restrict__ table, const int* __restrict__ bucket, float* __restrict__ bias, int H, int LL) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)H * LL) return;
  const int h = idx / LL, ij = idx % LL;
  bias[idx] = table[bucket[ij] * H + h];
}
__global__ void relbias_bwd_kernel(const float* __restrict__ dbias, const int* __restrict__ bucket, float* __restrict__ dtable, int H, int LL, int nb) {
  // one block per head: LDS histogram over the <=64 buckets, then one atomic per bucket
  __shared__ float acc[64];
  const int h = blockIdx.x;
  for (int i = threadIdx.x; i < 64; i += blockDim.x) acc[i] = 0.f;
  __syncthreads();
  for (int ij = threadIdx.x; ij < LL; ij += blockDim.x) atomicAdd(&acc[bucket[ij]], dbias[(long)h * LL + ij]);
  __syncthreads();
  for (int i = threadIdx.x; i < nb; i += blockDim.x) atomicAdd(dtable + i * H + h, acc[i]);
}

// ---- LM-head cross-entropy (HF/t5:1050-1054: CrossEntropyLoss(ignore_index=-100), mean over the
//      non-ignored rows; pads (id 0) ARE scored) -------------------------------------
__global__ void ce_count_kernel(const long long* __restrict__ labels, int rows, float* __restrict__ inv_n) {
  __shared__ int red[4];
  int c = 0;
  for (int i = threadIdx.x; i < rows; i += blockDim.x) c += labels[i] != -100;
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int n = red[0] + red[1] + red[2] + red[3];
    inv_n[0] = n > 0 ? 1.f / (float)n : 0.f;   // (reference would produce NaN for n == 0)
  }
}

template <typename T>
__global__ __launch_bounds__(256) void ce_fwd_kernel(T* __restrict__ logits, long ld, const long long* __restrict__ labels, int V,
                                                     const float* __restrict__ inv_n, float* __restrict__ loss_row, int write_grad) {
  constexpr int VEC = Vec16<T>::N;
  using VT = typename Vec16<T>::type;
  __shared__ float red[8];
  const long row = blockIdx.x;
  const int tid = threadIdx.x;
  T* x = logits + row * ld;
  const long long lab = labels[row];
  float m = -INFINITY, s = 0.f;
  for (int c = tid * VEC; c < V; c += 256 * VEC) {
    VT v = *reinterpret_cast<const VT*>(x + c);
    float mx = to_f32(v[0]);
#pragma unroll
    for (int u = 1; u < VEC; ++u) mx = fmaxf(mx, to_f32(v[u]));
    const float mn = fmaxf(m, mx);
    float a = 0.f;
#pragma unroll
    for (int u = 0; u < VEC; ++u) a += __expf(to_f32(v[u]) - mn);
    s = s * __expf(m - mn) + a;
    m = mn;
  }
  // block-wide (max, sum) merge
  float wm = wave_max(m);
  float ws = wave_sum(m == -INFINITY ? 0.f : s * __expf(m - wm));  // lanes/waves with no column contribute 0, not NaN
  if ((tid & 63) == 0) { red[tid >> 6] = wm; red[4 + (tid >> 6)] = ws; }
  __syncthreads();
  const float bm = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float bs = 0.f;
#pragma unroll
  for (int w = 0; w < 4; ++w) bs += red[w] == -INFINITY ? 0.f : red[4 + w] * __expf(red[w] - bm);
  const float lse = bm + __logf(bs);
  const bool valid = lab != -100;
  if (tid == 0) loss_row[row] = valid ? (lse - to_f32(x[lab])) : 0.f;
  if (!write_grad) return;
  __syncthreads();  // x[lab] read above before it is overwritten below
  const float g = valid ? inv_n[0] : 0.f;
  for (int c = tid * VEC; c < V; c += 256 * VEC) {
    VT v = *reinterpret_cast<const VT*>(x + c);
    VT o;
#pragma unroll
    for (int u = 0; u < VEC; ++u) {
      float pr = __expf(to_f32(v[u]) - lse);
      if (c + u == lab) pr -= 1.f;
      o[u] = from_f32<T>(pr * g);
    }
    *reinterpret_cast<VT*>(x + c) = o;
  }
}

// One-pass form for rows that fit the workgroup's registers (V <= 256 * 16 bytes/lane * NIT: 32768 bf16 logits with NIT = 16,
// the T5 vocabulary is 32128): the row is read ONCE -- every load of the row is issued before the first use -- and d logits are
// written from the registers.  The two-pass kernel above re-read the row from HBM (4096 rows x 64 KB in flight do not fit the L2s):
// 790 MB of traffic for 526 MB of algorithmic bytes.
template <typename T, int NIT>
__global__ __launch_bounds__(256) void ce_fwd_onepass_kernel(T* __restrict__ logits, long ld, const long long* __restrict__ labels, int V,
                                                             const float* __restrict__ inv_n, float* __restrict__ loss_row, int write_grad) {
  constexpr int VEC = Vec16<T>::N;
  using VT = typename Vec16<T>::type;
  __shared__ float red[8];
  __shared__ float xlab;
  const long row = blockIdx.x;
  const int tid = threadIdx.x;
  T* x = logits + row * ld;
  const long long lab = labels[row];
  VT v[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int c = (it * 256 + tid) * VEC;
    if (c < V) v[it] = *reinterpret_cast<const VT*>(x + c);
  }
  float m = -INFINITY;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int c = (it * 256 + tid) * VEC;
    if (c < V) {
#pragma unroll
      for (int u = 0; u < VEC; ++u) m = fmaxf(m, to_f32(v[it][u]));
    }
  }
  const float wm = wave_max(m);
  if ((tid & 63) == 0) red[tid >> 6] = wm;
  __syncthreads();
  const float bm = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int c = (it * 256 + tid) * VEC;
    if (c < V) {
#pragma unroll
      for (int u = 0; u < VEC; ++u) {
        sum += __expf(to_f32(v[it][u]) - bm);
        if (c + u == lab) xlab = to_f32(v[it][u]);
      }
    }
  }
  const float ws = wave_sum(sum);
  if ((tid & 63) == 0) red[4 + (tid >> 6)] = ws;
  __syncthreads();
  const float lse = bm + __logf(red[4] + red[5] + red[6] + red[7]);
  const bool valid = lab != -100;
  if (tid == 0) loss_row[row] = valid ? (lse - xlab) : 0.f;
  if (!write_grad) return;
  const float g = valid ? inv_n[0] : 0.f;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int c = (it * 256 + tid) * VEC;
    if (c < V) {
      VT o;
#pragma unroll
      for (int u = 0; u < VEC; ++u) {
        float pr = __expf(to_f32(v[it][u]) - lse);
        if (c + u == lab) pr -= 1.f;
        o[u] = from_f32<T>(pr * g);
      }
      *reinterpret_cast<VT*>(x + c) = o;
    }
  }
}

__global__ void ce_reduce_kernel(const float* __restrict__ loss_row, int rows, const float* __restrict__ inv_n, float* __restrict__ loss) {
  __shared__ float red[4];
  float a = 0.f;
  for (int i = threadIdx.x; i < rows; i += blockDim.x) a += loss_row[i];  // fixed order => bit-reproducible
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) loss[0] = (red[0] + red[1] + red[2] + red[3]) * inv_n[0];
}

// ---- Swin patch embedding im2col (Conv2d k4 s4 == GEMM over 48-wide patches, HF/swinv2:281) ----
template <typename T>
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ pix, T* __restrict__ out, int B, int Cin, int Himg, int P) {
  const int R = Himg / P;
  const int K = Cin * P * P;
  const long total = (long)B * R * R * K;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int k = idx % K;
    const long t = idx / K;
    const int px = k % P, py = (k / P) % P, c = k / (P * P);
    const int x = t % R, y = (t / R) % R;
    const long b = t / ((long)R * R);
    out[idx] = from_f32<T>(pix[((b * Cin + c) * Himg + (y * P + py)) * Himg + (x * P + px)]);
  }
}

// P == 4 form: one thread per patch, Cin*4 independent 16-byte loads (a wave reads 64 neighbouring patches = 1 KiB
// contiguous per load), 16-byte stores; columns [K, ldo) are zero-filled so the GEMM can run with K padded to 32.
// (The element-per-thread form above gathered 4-byte pieces: 1.3 TB/s.)
template <typename T>
__global__ __launch_bounds__(256) void im2col4_kernel(const float* __restrict__ pix, T* __restrict__ out, int B, int Cin, int Himg, int ldo) {
  const int R = Himg / 4;
  const long npatch = (long)B * R * R;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < npatch; t += (long)gridDim.x * blockDim.x) {
    const int x = t % R, y = (t / R) % R;
    const long b = t / ((long)R * R);
    T* o = out + t * ldo;
    for (int c = 0; c < Cin; ++c) {
      f32x4 v[4];
#pragma unroll
      for (int py = 0; py < 4; ++py) v[py] = *reinterpret_cast<const f32x4*>(pix + ((b * Cin + c) * Himg + (y * 4 + py)) * Himg + x * 4);
      if constexpr (sizeof(T) == 2) {
        *reinterpret_cast<bf16x8*>(o + c * 16) = bf16x8{(bf16_t)v[0][0], (bf16_t)v[0][1], (bf16_t)v[0][2], (bf16_t)v[0][3],
                                                        (bf16_t)v[1][0], (bf16_t)v[1][1], (bf16_t)v[1][2], (bf16_t)v[1][3]};
        *reinterpret_cast<bf16x8*>(o + c * 16 + 8) = bf16x8{(bf16_t)v[2][0], (bf16_t)v[2][1], (bf16_t)v[2][2], (bf16_t)v[2][3],
                                                            (bf16_t)v[3][0], (bf16_t)v[3][1], (bf16_t)v[3][2], (bf16_t)v[3][3]};
      } else {
#pragma unroll
        for (int py = 0; py < 4; ++py) *reinterpret_cast<f32x4*>(o + c * 16 + py * 4) = v[py];
      }
    }
    for (int k = Cin * 16; k < ldo; k += 8) {  // ldo % 8 == 0
      if constexpr (sizeof(T) == 2) *reinterpret_cast<bf16x8*>(o + k) = bf16x8{};
      else { *reinterpret_cast<f32x4*>(o + k) = f32x4{0.f, 0.f, 0.f, 0.f}; *reinterpret_cast<f32x4*>(o + k + 4) = f32x4{0.f, 0.f, 0.f, 0.f}; }
    }
  }
}

// ---- Swin patch merging gather (HF/swinv2:342-351): [B,R,R,C] -> [B,(R/2)^2, 4C] in the order
//      (0,0),(1,0),(0,1),(1,1) (row offset, col offset) ------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void merge_gather_kernel(const float* __restrict__ x, T* __restrict__ out, int B, int R, int C) {
  const int R2 = R / 2;
  const long total4 = (long)B * R2 * R2 * C;  // in units of 4 elements: 4C per token / 4
  for (long g = (long)blockIdx.x * blockDim.x + threadIdx.x; g < total4; g += (long)gridDim.x * blockDim.x) {
    const int c4 = g % C;            // 4-element group within the 4C row
    const long t = g / C;
    const int q = (c4 * 4) / C, c = (c4 * 4) % C;
    const int dy = q & 1, dx = q >> 1;
    const int x2 = t % R2, y2 = (t / R2) % R2;
    const long b = t / ((long)R2 * R2);
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + ((b * R + (2 * y2 + dy)) * R + (2 * x2 + dx)) * C + c);
    T* o = out + t * 4L * C + c4 * 4;
    if constexpr (sizeof(T) == 2) *reinterpret_cast<bf16x4*>(o) = bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
    else *reinterpret_cast<f32x4*>(o) = v;
  }
}
// scatter of d(merged rows) [B,(R/2)^2,4C] f32 back to dx [B,R,R,C] f32 (a permutation: plain stores)
__global__ __launch_bounds__(256) void merge_scatter_kernel(const float* __restrict__ dm, float* __restrict__ dx, int B, int R, int C) {
  const int R2 = R / 2;
  const long total4 = (long)B * R2 * R2 * C;
  for (long g = (long)blockIdx.x * blockDim.x + threadIdx.x; g < total4; g += (long)gridDim.x * blockDim.x) {
    const int c4 = g % C;
    const long t = g / C;
    const int q = (c4 * 4) / C, c = (c4 * 4) % C;
    const int dy = q & 1, dxo = q >> 1;
    const int x2 = t % R2, y2 = (t / R2) % R2;
    const long b = t / ((long)R2 * R2);
    *reinterpret_cast<f32x4*>(dx + ((b * R + (2 * y2 + dy)) * R + (2 * x2 + dxo)) * C + c) =
        *reinterpret_cast<const f32x4*>(dm + t * 4L * C + c4 * 4);
  }
}

// ---- column sums: dbias[n] += sum_m dY[m,n]  (bias gradients of the Swin Linears) --------------
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ dy, long ld, int M, int N, float* __restrict__ out) {
  // block = 64 columns x 4 row-lanes; grid.x over column tiles, grid.y over row slabs
  __shared__ float red[4][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  float a = 0.f;
  if (col < N)
    for (long m = (long)blockIdx.y * 4 + rl; m < M; m += (long)gridDim.y * 4) a += to_f32(dy[m * ld + col]);
  red[rl][threadIdx.x & 63] = a;
  __syncthreads();
  if (rl == 0 && col < N) atomicAdd(out + col, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// bf16, 16-byte aligned rows: 8 columns per lane (one 16-byte load), 32 row-lanes per block
__global__ __launch_bounds__(256) void colsum8_kernel(const bf16_t* __restrict__ dy, long ld, int M, int N, float* __restrict__ out) {
  __shared__ float red[32][65];
  const int cl = threadIdx.x & 7, rl = threadIdx.x >> 3;
  const int col = blockIdx.x * 64 + cl * 8;
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (col < N)
    for (long m = (long)blockIdx.y * 32 + rl; m < M; m += (long)gridDim.y * 32) {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(dy + m * ld + col);
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] += (float)v[u];
    }
#pragma unroll
  for (int u = 0; u < 8; ++u) red[rl][cl * 8 + u] = a[u];
  __syncthreads();
  if (threadIdx.x < 64 && blockIdx.x * 64 + threadIdx.x < N) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 32; ++r) t += red[r][threadIdx.x];
    atomicAdd(out + blockIdx.x * 64 + threadIdx.x, t);
  }
}

// ---- f32 -> T convert (optionally scaled) and f32 axpy ---------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void convert_kernel(const float* __restrict__ x, T* __restrict__ y, long n4, float scale) {
  for (long g = (long)blockIdx.x * blockDim.x + threadIdx.x; g < n4; g += (long)gridDim.x * blockDim.x) {
    f32x4 v = *reinterpret_cast<const f32x4*>(x + g * 4);
    if constexpr (sizeof(T) == 2) *reinterpret_cast<bf16x4*>(y + g * 4) = bf16x4{(bf16_t)(v[0] * scale), (bf16_t)(v[1] * scale), (bf16_t)(v[2] * scale), (bf16_t)(v[3] * scale)};
    else *reinterpret_cast<f32x4*>(y + g * 4) = f32x4{v[0] * scale, v[1] * scale, v[2] * scale, v[3] * scale};
  }
}
__global__ __launch_bounds__(256) void add_f32_kernel(float* __restrict__ y, const float* __restrict__ x, long n4) {
  for (long g = (long)blockIdx.x * blockDim.x + threadIdx.x; g < n4; g += (long)gridDim.x * blockDim.x) {
    f32x4 a = *reinterpret_cast<const f32x4*>(y + g * 4);
    const f32x4 b = *reinterpret_cast<const f32x4*>(x + g * 4);
    a[0] += b[0]; a[1] += b[1]; a[2] += b[2]; a[3] += b[3];
    *reinterpret_cast<f32x4*>(y + g * 4) = a;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, long n) {
  constexpr int VEC = Vec16<T>::N;
  using VT = typename Vec16<T>::type;
  for (long g = (long)blockIdx.x * blockDim.x + threadIdx.x; g < n / VEC; g += (long)gridDim.x * blockDim.x) {
    VT v = *reinterpret_cast<const VT*>(x + g * VEC);
    VT o;
#pragma unroll
    for (int u = 0; u < VEC; ++u) o[u] = from_f32<T>(gelu_for<T>(to_f32(v[u])));
    *reinterpret_cast<VT*>(y + g * VEC) = o;
  }
}

static inline unsigned stream_grid(long work_items) {
  long g = (work_items + 255) / 256;
  if (g < 1) g = 1;
  if (g > 2048) g = 2048;  // cap + grid-stride (memory-bound grid sizing rule)
  return (unsigned)g;
}

}  // namespace klab

using namespace klab;

extern "C" int klab_cast_pack(const void* desc_dev, int ndesc, long total4, void* dst, int dtype, void* stream) {
  if (!desc_dev || !dst || ndesc <= 0) return KLAB_ERR_BADARG;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == KLAB_BF16)
    hipLaunchKernelGGL(cast_pack_kernel<bf16_t>, dim3(stream_grid((total4 + 7) / 8)), dim3(256), 0, s, (const CastDesc*)desc_dev, ndesc, total4, (bf16_t*)dst);
  else
    hipLaunchKernelGGL(cast_pack_kernel<float>, dim3(stream_grid((total4 + 7) / 8)), dim3(256), 0, s, (const CastDesc*)desc_dev, ndesc, total4, (float*)dst);
  KLAB_LAUNCH_CHECK();
  return KLAB_OK;
}

extern "C" int klab_adam_step_range(const void* desc_dev, int ndesc, long begin4, long end4, const float* grads, float* m, float* v, void* arena,
                                    int dtype, float lr, float beta1, float beta2, float eps, float weight_decay, float bias_corr1,
                                    float bias_corr2, void* stream) {
  if (!desc_dev || ndesc <= 0 || !grads || !m || !v || !arena || bias_corr1 <= 0.f || bias_corr2 <= 0.f || begin4 < 0 || end4 < begin4)
    return KLAB_ERR_BADARG;
  if (end4 == begin4) return KLAB_OK;
  static const int desc_lds = [] { const char* e = getenv("KLAB_ADAM_DESC_LDS"); return !e || atoi(e) != 0 ? 1 : 0; }();
  AdamHyper h{lr / bias_corr1, beta1, beta2, eps, weight_decay, 1.f / sqrtf(bias_corr2), desc_lds};
  hipStream_t s = (hipStream_t)stream;
  static const int cap = [] { const char* e = getenv("KLAB_ADAM_GRID"); return e ? atoi(e) : 1024; }();
  // non-temporal loads / stores: every byte is touched once per step (415 -> 397 us); an 8-deep unroll measured 1.5 ms (spills)
  static const bool nt = [] { const char* e = getenv("KLAB_ADAM_NT"); return !e || atoi(e) != 0; }();
  constexpr int U = 4;
  long gl = ((end4 - begin4 + U - 1) / U + 255) / 256;
  const unsigned grid = (unsigned)(gl < 1 ? 1 : (gl > cap ? cap : gl));
#define ADAM_LAUNCH(TT, NN)                                                                                                       \
  hipLaunchKernelGGL((adam_step_kernel<TT, U, NN>), dim3(grid), dim3(256), 0, s, (const AdamDesc*)desc_dev, ndesc, begin4, end4, grads, m, v, \
                     (TT*)arena, h)
  if (dtype == KLAB_BF16) { if (nt) ADAM_LAUNCH(bf16_t, true); else ADAM_LAUNCH(bf16_t, false); }
  else { if (nt) ADAM_LAUNCH(float, true); else ADAM_LAUNCH(float, false); }
#undef ADAM_LAUNCH
  KLAB_LAUNCH_CHECK();
  return KLAB_OK;
}
extern "C" int klab_adam_step(const void* desc_dev, int ndesc, long total4, const float* grads, float* m, float* v, void* arena, int dtype,
                              float lr, float beta1, float beta2, float eps, float weight_decay, float bias_corr1, float bias_corr2,
                              void* stream) {
  return klab_adam_step_range(desc_dev, ndesc, 0, total4, grads, m, v, arena, dtype, lr, beta1, beta2, eps, weight_decay, bias_corr1, bias_corr2,
                              stream);
}

extern "C" int klab_embed_fwd(const long long* ids, int shift_right, int L, int start_id, int pad_id, const float* table, int vocab,
                              float* out, int rows, int d, float drop_p, const uint32_t* seed_dev, uint32_t tag, int* err_flag,
                              void* stream) {
  if (!ids || !table || !out || (d & 3)) return KLAB_ERR_BADARG;
  if (rows <= 0) return KLAB_OK;
  hipLaunchKernelGGL(embed_fwd_kernel, dim3(stream_grid((long)rows * 64)), dim3(256), 0, (hipStream_t)stream, ids, shift_right, L, start_id,
                     pad_id, table, vocab, out, rows, d, drop_p, seed_dev, tag, err_flag);
  KLAB_LAUNCH_CHECK();
  return KLAB_OK;
}

extern "C" int klab_embed_bwd(const long long* ids, int shift_right, int L, int start_id, int pad_id, const float* dh, float* dtable,
                              int vocab, int rows, int d, float drop_p, const uint32_t* seed_dev, uint32_t tag, void* stream) {
  if (!ids || !dh || !dtable) return KLAB_ERR_BADARG;
  if (rows <= 0) return KLAB_OK;
  hipLaunchKernelGGL(embed_bwd_kernel, dim3(stream_grid((long)rows * 64)), dim3(256), 0, (hipStream_t)stream, ids, shift_right, L, start_id,
                     pad_id, dh, dtable, vocab, rows, d, drop_p, seed_dev, tag);
  KLAB_LAUNCH_CHECK();
  return KLAB_OK;
}

extern "C" int klab_relbias_fwd(const float* table, const int* bucket, float* bias, int heads, int Lq, int Lk, void* stream) {
  if (!table || !bucket || !bias) return KLAB_ERR_BADARG;
  const long tot = (long)heads * Lq * Lk;
  hipLaunchKernelGGL(relbias_fwd_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, table, bucket, bias, heads, Lq * Lk);
  KLAB_LAUNCH_CHECK();
  return KLAB_OK;
}

extern "C" int klab_relbias_bwd(const float* dbias, const int* bucket, float* dtable, int heads, int Lq, int Lk, int nbuckets, void* stream) {
  if (!dbias || !bucket || !dtable) return KLAB_ERR_BADARG;
  if (nbuckets > 64) return KLAB_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(relbias_bwd_kernel, dim3(heads), dim3(256), 0, (hipStream_t)stream, dbias, bucket, dtable, heads, Lq * Lk, nbuckets);
  KLAB_LAUNCH_CHECK();
  return KLAB_OK;
}

// inv_n[0] = 1 / (number of labels != -100): the first launch of klab_ce_fwd on its own, for callers that know the labels long before
// the logits exist (the engine runs it beside the Swin tower and passes write_grad | 2)
extern "C" int klab_ce_count(const long long* labels, int rows, float* inv_n, void* stream) {
  if (!labels || !inv_n || rows <= 0) return KLAB_ERR_BADARG;
  hipLaunchKernelGGL(ce_count_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, labels, rows, inv_n);
  KLAB_LAUNCH_CHECK();
  return KLAB_OK;
}

extern "C" int klab_ce_fwd(void* logits, long ld, int dtype, const long long* labels, int rows, int V, float* inv_n, float* loss_row,
                           float* loss, int write_grad, void* stream) {
  if (!logits || !labels || !inv_n || !loss_row || !loss) return KLAB_ERR_BADARG;
  const int vec = dtype == KLAB_BF16 ? 8 : 4;
  if (V % vec || ld % vec) return KLAB_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (!(write_grad & 2)) {  // bit 1: inv_n already holds 1 / n_valid of these labels (klab_ce_count ran ahead of the logits)
    hipLaunchKernelGGL(ce_count_kernel, dim3(1), dim3(256), 0, s, labels, rows, inv_n);
    KLAB_LAUNCH_CHECK();
  }
  write_grad &= 1;
  static const bool onepass = [] { const char* e = getenv("KLAB_CE_ONEPASS"); return !e || atoi(e) != 0; }();
  if (dtype == KLAB_BF16 && onepass && V <= 256 * 8 * 16 && V > 256 * 8 * 8)
    hipLaunchKernelGGL((ce_fwd_onepass_kernel<bf16_t, 16>), dim3(rows), dim3(256), 0, s, (bf16_t*)logits, ld, labels, V, inv_n, loss_row, write_grad);
  else if (dtype == KLAB_BF16 && onepass && V <= 256 * 8 * 8)
    hipLaunchKernelGGL((ce_fwd_onepass_kernel<bf16_t, 8>), dim3(rows), dim3(256), 0, s, (bf16_t*)logits, ld, labels, V, inv_n, loss_row, write_grad);
  else if (dtype == KLAB_BF16)
    hipLaunchKernelGGL(ce_fwd_kernel<bf16_t>, dim3(rows), dim3(256), 0, s, (bf16_t*)logits, ld, labels, V, inv_n, loss_row, write_grad);
  else
    hipLaunchKernelGGL(ce_fwd_kernel<float>, dim3(rows), dim3(256), 0, s, (float*)logits, ld, labels, V, inv_n, loss_row, write_grad);
  KLAB_LAUNCH_CHECK();
  hipLaunchKernelGGL(ce_reduce_kernel, dim3(1), dim3(256), 0, s, loss_row, rows, inv_n, loss);
  KLAB_LAUNCH_CHECK();
  return KLAB_OK;
}

extern "C" int klab_im2col_patch(const float* pixels, void* out, int dtype, int B, int Cin, int Himg, int P, void* stream) {
  if (!pixels || !out || Himg % P) return KLAB_ERR_BADARG;
  const long total = (long)B * (Himg / P) * (Himg / P) * Cin * P * P;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == KLAB_BF16) hipLaunchKernelGGL(im2col_kernel<bf16_t>, dim3(stream_grid(total)), dim3(256), 0, s, pixels, (bf16_t*)out, B, Cin, Himg, P);
  else hipLaunchKernelGGL(im2col_kernel<float>, dim3(stream_grid(total)), dim3(256), 0, s, pixels, (float*)out, B, Cin, Himg, P);
  KLAB_LAUNCH_CHECK();
  return KLAB_OK;
}

extern "C" int klab_im2col_patch_ld(const float* pixels, void* out, int dtype, int B, int Cin, int Himg, int P, int ldo, void* stream) {
  if (!pixels || !out || Himg % P || ldo < Cin * P * P) return KLAB_ERR_BADARG;
  if (P != 4 || (Himg & 3) || (ldo & 7) || ((uintptr_t)pixels & 15) || ((uintptr_t)out & 15)) {
    if (ldo == Cin * P * P) return klab_im2col_patch(pixels, out, dtype, B, Cin, Himg, P, stream);
    return KLAB_ERR_UNSUPPORTED;
  }
  const long npatch = (long)B * (Himg / 4) * (Himg / 4);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == KLAB_BF16) hipLaunchKernelGGL(im2col4_kernel<bf16_t>, dim3(stream_grid(npatch)), dim3(256), 0, s, pixels, (bf16_t*)out, B, Cin, Himg, ldo);
  else hipLaunchKernelGGL(im2col4_kernel<float>, dim3(stream_grid(npatch)), dim3(256), 0, s, pixels, (float*)out, B, Cin, Himg, ldo);
  KLAB_LAUNCH_CHECK();
  return KLAB_OK;
}

extern "C" int klab_merge_gather(const float* x, void* out, int dtype, int B, int R, int C, void* stream) {
  if (!x || !out || (R & 1) || (C & 3)) return KLAB_ERR_BADARG;
  const long total4 = (long)B * (R / 2) * (R / 2) * C;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == KLAB_BF16) hipLaunchKernelGGL(merge_gather_kernel<bf16_t>, dim3(stream_grid(total4)), dim3(256), 0, s, x, (bf16_t*)out, B, R, C);
  else hipLaunchKernelGGL(merge_gather_kernel<float>, dim3(stream_grid(total4)), dim3(256), 0, s, x, (float*)out, B, R, C);
  KLAB_LAUNCH_CHECK();
  return KLAB_OK;
}

extern "C" int klab_merge_scatter(const float* dmerged, float* dx, int B, int R, int C, void* stream) {
  if (!dmerged || !dx || (R & 1) || (C & 3)) return KLAB_ERR_BADARG;
  const long total4 = (long)B * (R / 2) * (R / 2) * C;
  hipLaunchKernelGGL(merge_scatter_kernel, dim3(stream_grid(total4)), dim3(256), 0, (hipStream_t)stream, dmerged, dx, B, R, C);
  KLAB_LAUNCH_CHECK();
  return KLAB_OK;
}

extern "C" int klab_colsum(const void* dy, long ld, int dtype, int M, int N, float* out, void* stream) {
  if (!dy || !out) return KLAB_ERR_BADARG;
  int gy = (M + 255) / 256;
  gy = gy < 1 ? 1 : (gy > 256 ? 256 : gy);
  dim3 grid((N + 63) / 64, gy);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == KLAB_BF16 && (N & 7) == 0 && (ld & 7) == 0 && ((uintptr_t)dy & 15) == 0) {
    int g8 = (M + 127) / 128;  // >= 4 rows per lane
    const int want = 2048 / (int)grid.x;  // ~8 blocks per CU over all column tiles
    g8 = g8 < 1 ? 1 : (g8 > want ? (want < 1 ? 1 : want) : g8);
    hipLaunchKernelGGL(colsum8_kernel, dim3(grid.x, g8), dim3(256), 0, s, (const bf16_t*)dy, ld, M, N, out);
    KLAB_LAUNCH_CHECK();
    return KLAB_OK;
  }
  if (dtype == KLAB_BF16) hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)dy, ld, M, N, out);
  else hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, s, (const float*)dy, ld, M, N, out);
  KLAB_LAUNCH_CHECK();
  return KLAB_OK;
}

extern "C" int klab_convert(const float* x, void* y, int dtype, long n, float scale, void* stream) {
  if (!x || !y || (n & 3)) return KLAB_ERR_BADARG;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == KLAB_BF16) hipLaunchKernelGGL(convert_kernel<bf16_t>, dim3(stream_grid(n / 4)), dim3(256), 0, s, x, (bf16_t*)y, n / 4, scale);
  else hipLaunchKernelGGL(convert_kernel<float>, dim3(stream_grid(n / 4)), dim3(256), 0, s, x, (float*)y, n / 4, scale);
  KLAB_LAUNCH_CHECK();
  return KLAB_OK;
}

extern "C" int klab_add_f32(float* y, const float* x, long n, void* stream) {
  if (!x || !y || (n & 3)) return KLAB_ERR_BADARG;
  hipLaunchKernelGGL(add_f32_kernel, dim3(stream_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, y, x, n / 4);
  KLAB_LAUNCH_CHECK();
  return KLAB_OK;
}

// fp8 mode: a = gelu(z) (bf16) and the same rows in e4m3 with one scale per row (the quantisation pass in front of fc2 folded in);
// one wave per row of F <= 4096 values
__global__ __launch_bounds__(256) void gelu_fwd_q8_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, unsigned char* __restrict__ y8,
                                                          float* __restrict__ yscale, int rows, int F) {
  constexpr int MAXV = 8;
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const bf16_t* xr = x + row * F;
  bf16x8 v[MAXV];
  float amax = 0.f;
#pragma unroll
  for (int u = 0; u < MAXV; ++u) {
    const int k = (u * 64 + lane) * 8;
    if (k < F) {
      const bf16x8 z = *reinterpret_cast<const bf16x8*>(xr + k);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[u][e] = from_f32<bf16_t>(gelu_for<bf16_t>(to_f32(z[e])));
        amax = fmaxf(amax, fabsf((float)v[u][e]));
      }
      *reinterpret_cast<bf16x8*>(y + row * F + k) = v[u];
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
  const float sc = amax > 0.f ? amax * (1.f / 448.f) : 1.f;
  const float inv = 1.f / sc;
  if (lane == 0) yscale[row] = sc;
#pragma unroll
  for (int u = 0; u < MAXV; ++u) {
    const int k = (u * 64 + lane) * 8;
    if (k < F) {
      float f[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = fminf(fmaxf((float)v[u][e] * inv, -448.f), 448.f);
      int w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], 0, false);
      w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], w0, true);
      int w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], 0, false);
      w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], w1, true);
      *reinterpret_cast<int2*>(y8 + row * F + k) = int2{w0, w1};
    }
  }
}

extern "C" int klab_gelu_fwd_q8(const void* x, void* y, void* y8, float* yscale, int rows, int F, void* stream) {
  if (!x || !y || !y8 || !yscale || rows < 0 || F <= 0 || (F & 7)) return KLAB_ERR_BADARG;
  if (F > 4096) return KLAB_ERR_UNSUPPORTED;
  if (rows == 0) return KLAB_OK;
  hipLaunchKernelGGL(gelu_fwd_q8_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y,
                     (unsigned char*)y8, yscale, rows, F);
  KLAB_LAUNCH_CHECK();
  return KLAB_OK;
}

extern "C" int klab_gelu_fwd(const void* x, void* y, int dtype, long n, void* stream) {
  if (!x || !y || (n & 7)) return KLAB_ERR_BADARG;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == KLAB_BF16) hipLaunchKernelGGL(gelu_fwd_kernel<bf16_t>, dim3(stream_grid(n / 8)), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, n);
  else hipLaunchKernelGGL(gelu_fwd_kernel<float>, dim3(stream_grid(n / 4)), dim3(256), 0, s, (const float*)x, (float*)y, n);
  KLAB_LAUNCH_CHECK();
  return KLAB_OK;
}

extern "C" int klab_version(void) { return 1; }
