// Swin-V2 shifted-window cosine attention (HF/swinv2:389-455 inside Swinv2Layer.forward :652-705).
// What the reference does with ~12 ATen kernels and 4 full-tensor copies per block -- torch.roll
// (:667-670), window_partition (:146-155), F.normalize(q)@F.normalize(k)^T (:413-415), * exp(clamped
// logit_scale) (:416-417), + 16*sigmoid(CPB) (:427-428), + shift mask TWICE (:433-436, the pinned
// transformers adds it on two consecutive lines => -200), softmax (:440), @V (:448), head merge,
// window_reverse + un-roll (:683-690) -- is ONE kernel here: roll/partition/reverse are index math
// on the token id, the 9-region mask (:620-643) is recomputed from coordinates, nothing is copied.
//   layout: qkv [B*R*R, 3C] (fused q|k|v projection, head h at column h*hd), ctx [B*R*R, C].
//   one wave per (image, window, head); keys/values of the window live in LDS (k pre-normalised);
//   lane i owns query row i (rows i, i+64, ... when n > 64): q-hat and the output row stay in
//   registers, softmax is online (running max / sum per lane) -- no cross-lane traffic at all.
// Round-1 form: fp32 vector-ALU dot products (n = 49, hd = 32 tiles are awkward MFMA shapes; the
// padded-to-64 MFMA form is the planned replacement, SURVEY §7 "small-tile efficiency").
#include <math.h>

#include "common.h"
#include "klab_mm.h"

namespace klab {

struct SwinAttnP {
  const void* qkv; void* ctx; const float* bias; const float* logit_scale; float* lse;
  int B, R, w, shift, H, C;
  // backward
  const void* dctx; void* dqkv; float* dbias; float* dlogit_scale;
};

__device__ __forceinline__ int swin_region(int s, int R, int w, int shift) { return (s >= R - w) + (s >= R - shift); }

template <typename T, int HD>
__global__ __launch_bounds__(64) void swin_attn_fwd_kernel(SwinAttnP p) {
  constexpr int VEC = Vec16<T>::N;
  constexpr int KST = HD + VEC;
  using V = typename Vec16<T>::type;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int w = p.w, n = w * w, R = p.R, C = p.C, H = p.H;
  const int nWr = R / w, nW = nWr * nWr;
  T* Kn = reinterpret_cast<T*>(smem);
  T* Vs = Kn + (size_t)n * KST;
  int* tok = reinterpret_cast<int*>(Vs + (size_t)n * KST);
  int* reg = tok + n;
  const int lane = threadIdx.x;
  int bid = blockIdx.x;
  const int h = bid % H; bid /= H;
  const int win = bid % nW; const int b = bid / nW;
  const int wy = win / nWr, wx = win % nWr;
  const T* qkv = reinterpret_cast<const T*>(p.qkv);
  const long ld = 3L * C;

  for (int j = lane; j < n; j += 64) {
    const int ys = wy * w + j / w, xs = wx * w + j % w;      // coordinates in the rolled image
    const int y = (ys + p.shift) % R, x = (xs + p.shift) % R; // torch.roll(-shift) source
    const int t = (b * R + y) * R + x;
    tok[j] = t;
    reg[j] = p.shift > 0 ? swin_region(ys, R, w, p.shift) * 3 + swin_region(xs, R, w, p.shift) : 0;
    const T* kr = qkv + (long)t * ld + C + h * HD;
    const T* vr = qkv + (long)t * ld + 2 * C + h * HD;
    float kf[HD];
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < HD; c += VEC) {
      V kv = *reinterpret_cast<const V*>(kr + c);
#pragma unroll
      for (int u = 0; u < VEC; ++u) { kf[c + u] = to_f32(kv[u]); ss += kf[c + u] * kf[c + u]; }
      *reinterpret_cast<V*>(Vs + j * KST + c) = *reinterpret_cast<const V*>(vr + c);
    }
    const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);  // F.normalize eps (HF/swinv2:413)
#pragma unroll
    for (int c = 0; c < HD; c += VEC) {
      V o;
#pragma unroll
      for (int u = 0; u < VEC; ++u) o[u] = from_f32<T>(kf[c + u] * inv);
      *reinterpret_cast<V*>(Kn + j * KST + c) = o;
    }
  }
  __syncthreads();

  const float scale = __expf(fminf(p.logit_scale[h], 4.6051701859880914f));  // clamp at ln(100) (HF/swinv2:416)
  T* ctx = reinterpret_cast<T*>(p.ctx);
  for (int i = lane; i < n; i += 64) {
    const int t = tok[i];
    const T* qr = qkv + (long)t * ld + h * HD;
    float qn[HD];
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < HD; c += VEC) {
      V qv = *reinterpret_cast<const V*>(qr + c);
#pragma unroll
      for (int u = 0; u < VEC; ++u) { qn[c + u] = to_f32(qv[u]); ss += qn[c + u] * qn[c + u]; }
    }
    const float inv = scale / fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
    for (int c = 0; c < HD; ++c) qn[c] *= inv;
    const int ri = reg[i];
    const float* br = p.bias + ((long)h * n + i) * n;
    float m = -INFINITY, l = 0.f;
    float o[HD];
#pragma unroll
    for (int c = 0; c < HD; ++c) o[c] = 0.f;
    for (int j = 0; j < n; ++j) {
      float s = 0.f;
      const T* kr = Kn + j * KST;
#pragma unroll
      for (int c = 0; c < HD; c += VEC) {
        V kv = *reinterpret_cast<const V*>(kr + c);
#pragma unroll
        for (int u = 0; u < VEC; ++u) s += qn[c + u] * to_f32(kv[u]);
      }
      s += br[j];
      if (reg[j] != ri) s += -200.f;  // -100 added twice by the pinned transformers (HF/swinv2:433-436)
      const float mn = fmaxf(m, s);
      const float corr = __expf(m - mn);
      const float pe = __expf(s - mn);
      l = l * corr + pe;
      const T* vr = Vs + j * KST;
#pragma unroll
      for (int c = 0; c < HD; c += VEC) {
        V vv = *reinterpret_cast<const V*>(vr + c);
#pragma unroll
        for (int u = 0; u < VEC; ++u) o[c + u] = o[c + u] * corr + pe * to_f32(vv[u]);
      }
      m = mn;
    }
    const float il = 1.f / l;
    T* orow = ctx + (long)t * C + h * HD;
#pragma unroll
    for (int c = 0; c < HD; c += VEC) {
      V ov;
#pragma unroll
      for (int u = 0; u < VEC; ++u) ov[u] = from_f32<T>(o[c + u] * il);
      *reinterpret_cast<V*>(orow + c) = ov;
    }
    if (p.lse) p.lse[(((long)b * nW + win) * H + h) * n + i] = m + __logf(l);
  }
}

// Backward of the same (used when --image_model_train, ref/models/model.py:15): lane i owns query row
// i, recomputes P from the saved lse; dK-hat / dV are accumulated in LDS with f32 atomics (ds_add_f32),
// then pushed through the L2-normalisation Jacobian.  d(logit_scale) and d(bias) use global atomics.
template <typename T, int HD>
__global__ __launch_bounds__(64) void swin_attn_bwd_kernel(SwinAttnP p) {
  constexpr int VEC = Vec16<T>::N;
  constexpr int KST = HD + VEC;
  using V = typename Vec16<T>::type;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int w = p.w, n = w * w, R = p.R, C = p.C, H = p.H;
  const int nWr = R / w, nW = nWr * nWr;
  T* Kn = reinterpret_cast<T*>(smem);
  T* Vs = Kn + (size_t)n * KST;
  float* dKn = reinterpret_cast<float*>(Vs + (size_t)n * KST);
  float* dVs = dKn + (size_t)n * HD;
  float* kinv = dVs + (size_t)n * HD;
  int* tok = reinterpret_cast<int*>(kinv + n);
  int* reg = tok + n;
  const int lane = threadIdx.x;
  int bid = blockIdx.x;
  const int h = bid % H; bid /= H;
  const int win = bid % nW; const int b = bid / nW;
  const int wy = win / nWr, wx = win % nWr;
  const T* qkv = reinterpret_cast<const T*>(p.qkv);
  const T* dctx = reinterpret_cast<const T*>(p.dctx);
  T* dqkv = reinterpret_cast<T*>(p.dqkv);
  const long ld = 3L * C;

  for (int j = lane; j < n; j += 64) {
    const int ys = wy * w + j / w, xs = wx * w + j % w;
    const int y = (ys + p.shift) % R, x = (xs + p.shift) % R;
    const int t = (b * R + y) * R + x;
    tok[j] = t;
    reg[j] = p.shift > 0 ? swin_region(ys, R, w, p.shift) * 3 + swin_region(xs, R, w, p.shift) : 0;
    const T* kr = qkv + (long)t * ld + C + h * HD;
    const T* vr = qkv + (long)t * ld + 2 * C + h * HD;
    float kf[HD];
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < HD; c += VEC) {
      V kv = *reinterpret_cast<const V*>(kr + c);
#pragma unroll
      for (int u = 0; u < VEC; ++u) { kf[c + u] = to_f32(kv[u]); ss += kf[c + u] * kf[c + u]; }
      *reinterpret_cast<V*>(Vs + j * KST + c) = *reinterpret_cast<const V*>(vr + c);
    }
    const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
    kinv[j] = inv;
#pragma unroll
    for (int c = 0; c < HD; c += VEC) {
      V o;
#pragma unroll
      for (int u = 0; u < VEC; ++u) o[u] = from_f32<T>(kf[c + u] * inv);
      *reinterpret_cast<V*>(Kn + j * KST + c) = o;
    }
#pragma unroll
    for (int c = 0; c < HD; ++c) { dKn[j * HD + c] = 0.f; dVs[j * HD + c] = 0.f; }
  }
  __syncthreads();

  const float ls = p.logit_scale[h];
  const bool clamped = ls > 4.6051701859880914f;
  const float scale = __expf(fminf(ls, 4.6051701859880914f));
  float dscale = 0.f;  // d loss / d scale, summed over this lane's rows
  for (int i = lane; i < n; i += 64) {
    const int t = tok[i];
    const T* qr = qkv + (long)t * ld + h * HD;
    const T* dor = dctx + (long)t * C + h * HD;
    float qh[HD], dO[HD];
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < HD; c += VEC) {
      V qv = *reinterpret_cast<const V*>(qr + c);
      V dv = *reinterpret_cast<const V*>(dor + c);
#pragma unroll
      for (int u = 0; u < VEC; ++u) { qh[c + u] = to_f32(qv[u]); ss += qh[c + u] * qh[c + u]; dO[c + u] = to_f32(dv[u]); }
    }
    const float qinv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
    for (int c = 0; c < HD; ++c) qh[c] *= qinv;  // unit q-hat (unscaled)
    const int ri = reg[i];
    const float* br = p.bias + ((long)h * n + i) * n;
    const float lse = p.lse[(((long)b * nW + win) * H + h) * n + i];
    // pass 1: delta = sum_j P_ij * (dO . V_j)
    float delta = 0.f;
    for (int j = 0; j < n; ++j) {
      float cs = 0.f, dp = 0.f;
#pragma unroll
      for (int c = 0; c < HD; ++c) { cs += qh[c] * to_f32(Kn[j * KST + c]); dp += dO[c] * to_f32(Vs[j * KST + c]); }
      float s = cs * scale + br[j];
      if (reg[j] != ri) s += -200.f;
      delta += __expf(s - lse) * dp;
    }
    // pass 2: dS, dq-hat, dK-hat, dV, dbias, dscale
    float dqh[HD];
#pragma unroll
    for (int c = 0; c < HD; ++c) dqh[c] = 0.f;
    for (int jj = 0; jj < n; ++jj) {
      int j = jj + i;  // staggered start: lanes of the wave hit different key rows => no same-address LDS atomics
      if (j >= n) j -= n;
      float cs = 0.f, dp = 0.f;
#pragma unroll
      for (int c = 0; c < HD; ++c) { cs += qh[c] * to_f32(Kn[j * KST + c]); dp += dO[c] * to_f32(Vs[j * KST + c]); }
      float s = cs * scale + br[j];
      if (reg[j] != ri) s += -200.f;
      const float pr = __expf(s - lse);
      const float ds = pr * (dp - delta);
      if (p.dbias) atomicAdd(p.dbias + ((long)h * n + i) * n + j, ds);
      dscale += ds * cs;
      const float dcs = ds * scale;
#pragma unroll
      for (int c = 0; c < HD; ++c) {
        dqh[c] += dcs * to_f32(Kn[j * KST + c]);
        atomicAdd(&dKn[j * HD + c], dcs * qh[c]);
        atomicAdd(&dVs[j * HD + c], pr * dO[c]);
      }
    }
    // through q-hat = q / ||q||:  dq = (dqh - qh * (qh . dqh)) / ||q||
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < HD; ++c) dot += qh[c] * dqh[c];
    T* dqr = dqkv + (long)t * ld + h * HD;
#pragma unroll
    for (int c = 0; c < HD; c += VEC) {
      V ov;
#pragma unroll
      for (int u = 0; u < VEC; ++u) ov[u] = from_f32<T>((dqh[c + u] - qh[c + u] * dot) * qinv);
      *reinterpret_cast<V*>(dqr + c) = ov;
    }
  }
  dscale = wave_sum(dscale);
  if (lane == 0 && p.dlogit_scale && !clamped) atomicAdd(p.dlogit_scale + h, dscale * scale);
  __syncthreads();
  for (int j = lane; j < n; j += 64) {
    const int t = tok[j];
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < HD; ++c) dot += to_f32(Kn[j * KST + c]) * dKn[j * HD + c];
    T* dkr = dqkv + (long)t * ld + C + h * HD;
    T* dvr = dqkv + (long)t * ld + 2 * C + h * HD;
    const float inv = kinv[j];
#pragma unroll
    for (int c = 0; c < HD; c += VEC) {
      V ok, ov;
#pragma unroll
      for (int u = 0; u < VEC; ++u) {
        ok[u] = from_f32<T>((dKn[j * HD + c + u] - to_f32(Kn[j * KST + c + u]) * dot) * inv);
        ov[u] = from_f32<T>(dVs[j * HD + c + u]);
      }
      *reinterpret_cast<V*>(dkr + c) = ok;
      *reinterpret_cast<V*>(dvr + c) = ov;
    }
  }
}

// Continuous position bias (HF/swinv2:376-378,418-428): table[t,h] = MLP(coords[t]) ; bias[h,i,j] =
// 16*sigmoid(table[index[i,j],h]).  coords / index are input-independent buffers built on the host
// exactly as HF/swinv2:457-492 does.  One block per table entry; 512 hidden units over 256 threads.
__global__ __launch_bounds__(256) void cpb_table_kernel(const float* __restrict__ coords, const float* __restrict__ w0,
                                                        const float* __restrict__ b0, const float* __restrict__ w2, float* __restrict__ table,
                                                        float* __restrict__ hidden_out, int H, int nh) {
  __shared__ float hid[512];
  __shared__ float red[4];
  const int t = blockIdx.x, tid = threadIdx.x;
  const float c0 = coords[t * 2], c1 = coords[t * 2 + 1];
  for (int j = tid; j < nh; j += 256) {
    const float v = fmaxf(w0[j * 2] * c0 + w0[j * 2 + 1] * c1 + b0[j], 0.f);
    hid[j] = v;
    if (hidden_out) hidden_out[(long)t * nh + j] = v;
  }
  __syncthreads();
  for (int h = 0; h < H; ++h) {
    float a = 0.f;
    for (int j = tid; j < nh; j += 256) a += hid[j] * w2[h * nh + j];
    a = wave_sum(a);
    if ((tid & 63) == 0) red[tid >> 6] = a;
    __syncthreads();
    if (tid == 0) table[t * H + h] = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
  }
}
__global__ void cpb_gather_kernel(const float* __restrict__ table, const int* __restrict__ index, float* __restrict__ bias, int H, int nn) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)H * nn) return;
  const int h = idx / nn, ij = idx % nn;
  const float v = table[index[ij] * H + h];
  bias[idx] = 16.f / (1.f + __expf(-v));
}


// CPB backward: dbias[H,n,n] -> d table -> MLP weight grads (accumulated into zeroed buffers).
__global__ void cpb_dtable_kernel(const float* __restrict__ dbias, const float* __restrict__ bias, const int* __restrict__ index,
                                  float* __restrict__ dtable, int H, int nn) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)H * nn) return;
  const int h = idx / nn, ij = idx % nn;
  const float bv = bias[idx];                       // 16*sigmoid(t)
  atomicAdd(dtable + index[ij] * H + h, dbias[idx] * bv * (1.f - bv * 0.0625f));
}
__global__ __launch_bounds__(256) void cpb_mlp_bwd_kernel(const float* __restrict__ dtable, const float* __restrict__ coords,
                                                          const float* __restrict__ hidden, const float* __restrict__ w2, float* __restrict__ dw0,
                                                          float* __restrict__ db0, float* __restrict__ dw2, int ntab, int H, int nh) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;  // hidden unit
  if (j >= nh) return;
  for (int h = 0; h < H; ++h) {
    float s = 0.f;
    for (int t = 0; t < ntab; ++t) s += dtable[t * H + h] * hidden[(long)t * nh + j];
    dw2[h * nh + j] += s;
  }
  float a0 = 0.f, a1 = 0.f, ab = 0.f;
  for (int t = 0; t < ntab; ++t) {
    if (hidden[(long)t * nh + j] <= 0.f) continue;  // ReLU
    float g = 0.f;
    for (int h = 0; h < H; ++h) g += dtable[t * H + h] * w2[h * nh + j];
    a0 += g * coords[t * 2]; a1 += g * coords[t * 2 + 1]; ab += g;
  }
  dw0[j * 2] += a0; dw0[j * 2 + 1] += a1; db0[j] += ab;
}

}  // namespace klab

using namespace klab;

template <typename T, int HD>
static int launch_swin_fwd(const SwinAttnP& p, hipStream_t s) {
  const int n = p.w * p.w;
  const size_t lds = 2 * (size_t)n * (HD + Vec16<T>::N) * sizeof(T) + 2 * (size_t)n * 4;
  if (lds > 64 * 1024) return KLAB_ERR_UNSUPPORTED;  // window too large for the round-1 single-tile form
  const int nW = (p.R / p.w) * (p.R / p.w);
  hipLaunchKernelGGL((swin_attn_fwd_kernel<T, HD>), dim3(p.B * nW * p.H), dim3(64), lds, s, p);
  KLAB_LAUNCH_CHECK();
  return KLAB_OK;
}
template <typename T, int HD>
static int launch_swin_bwd(const SwinAttnP& p, hipStream_t s) {
  const int n = p.w * p.w;
  const size_t lds = 2 * (size_t)n * (HD + Vec16<T>::N) * sizeof(T) + 2 * (size_t)n * HD * 4 + 3 * (size_t)n * 4;
  if (lds > 64 * 1024) return KLAB_ERR_UNSUPPORTED;
  const int nW = (p.R / p.w) * (p.R / p.w);
  hipLaunchKernelGGL((swin_attn_bwd_kernel<T, HD>), dim3(p.B * nW * p.H), dim3(64), lds, s, p);
  KLAB_LAUNCH_CHECK();
  return KLAB_OK;
}

static int swin_args_ok(const klab_swin_attn_args* a) {
  if (!a || !a->qkv || !a->ctx || !a->bias || !a->logit_scale) return KLAB_ERR_BADARG;
  if (a->w <= 0 || a->R % a->w) return KLAB_ERR_UNSUPPORTED;  // padded windows (HF/swinv2:645-650) are out of scope
  if (a->C % a->H) return KLAB_ERR_BADARG;
  if (a->shift < 0 || a->shift >= a->w) return KLAB_ERR_BADARG;
  return KLAB_OK;
}

extern "C" int klab_swin_attn_fwd(const klab_swin_attn_args* a, void* stream) {
  int rc = swin_args_ok(a);
  if (rc) return rc;
  SwinAttnP p{a->qkv, a->ctx, a->bias, a->logit_scale, a->lse, a->B, a->R, a->w, a->shift, a->H, a->C, nullptr, nullptr, nullptr, nullptr};
  const int hd = a->C / a->H;
  hipStream_t s = (hipStream_t)stream;
  if (a->dtype == KLAB_BF16) {
    if (hd == 32) return launch_swin_fwd<bf16_t, 32>(p, s);
    if (hd == 16) return launch_swin_fwd<bf16_t, 16>(p, s);
  } else if (a->dtype == KLAB_F32) {
    if (hd == 32) return launch_swin_fwd<float, 32>(p, s);
    if (hd == 16) return launch_swin_fwd<float, 16>(p, s);
  }
  return KLAB_ERR_UNSUPPORTED;
}

extern "C" int klab_swin_attn_bwd(const klab_swin_attn_args* a, void* stream) {
  int rc = swin_args_ok(a);
  if (rc) return rc;
  if (!a->dctx || !a->dqkv || !a->lse) return KLAB_ERR_BADARG;
  SwinAttnP p{a->qkv, a->ctx, a->bias, a->logit_scale, a->lse, a->B, a->R, a->w, a->shift, a->H, a->C,
              a->dctx, a->dqkv, a->dbias, a->dlogit_scale};
  const int hd = a->C / a->H;
  hipStream_t s = (hipStream_t)stream;
  if (a->dtype == KLAB_BF16) {
    if (hd == 32) return launch_swin_bwd<bf16_t, 32>(p, s);
    if (hd == 16) return launch_swin_bwd<bf16_t, 16>(p, s);
  } else if (a->dtype == KLAB_F32) {
    if (hd == 32) return launch_swin_bwd<float, 32>(p, s);
    if (hd == 16) return launch_swin_bwd<float, 16>(p, s);
  }
  return KLAB_ERR_UNSUPPORTED;
}

extern "C" int klab_swin_cpb_bias(const float* coords, const int* index, const float* w0, const float* b0, const float* w2,
                                  float* table, float* hidden, float* bias, int ntab, int n, int heads, int nhidden,
                                  void* stream) {
  if (!coords || !index || !w0 || !b0 || !w2 || !table || !bias) return KLAB_ERR_BADARG;
  if (nhidden > 512) return KLAB_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(cpb_table_kernel, dim3(ntab), dim3(256), 0, s, coords, w0, b0, w2, table, hidden, heads, nhidden);
  KLAB_LAUNCH_CHECK();
  const long tot = (long)heads * n * n;
  hipLaunchKernelGGL(cpb_gather_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, table, index, bias, heads, n * n);
  KLAB_LAUNCH_CHECK();
  return KLAB_OK;
}

extern "C" int klab_swin_cpb_bias_bwd(const float* dbias, const float* bias, const int* index, const float* coords, const float* hidden,
                                      const float* w0, const float* w2, float* dtable, float* dw0, float* db0, float* dw2, int ntab, int n,
                                      int heads, int nhidden, void* stream) {
  (void)w0;
  if (!dbias || !bias || !index || !coords || !hidden || !w2 || !dtable || !dw0 || !db0 || !dw2) return KLAB_ERR_BADARG;
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(dtable, 0, (size_t)ntab * heads * 4, s);
  if (e != hipSuccess) return (int)e;
  const long tot = (long)heads * n * n;
  hipLaunchKernelGGL(cpb_dtable_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, dbias, bias, index, dtable, heads, n * n);
  KLAB_LAUNCH_CHECK();
  hipLaunchKernelGGL(cpb_mlp_bwd_kernel, dim3((nhidden + 255) / 256), dim3(256), 0, s, dtable, coords, hidden, w2, dw0, db0, dw2, ntab, heads, nhidden);
  KLAB_LAUNCH_CHECK();
  return KLAB_OK;
}
