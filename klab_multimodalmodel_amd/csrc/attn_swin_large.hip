// Swin-V2 window attention for windows of MORE than 64 tokens (384 px / window 24: n = 576 in stages 0-2, 144 in stage 3;
// HF/swinv2:389-455, 615-618, 652-690) -- the shapes attn_swin.hip's one-tile kernels refuse.
//
// Same math, same index conventions (roll / partition / reverse as token-id arithmetic, the shift mask from coordinates and
// added twice, F.normalize eps, logit-scale clamp), but tiled:
//   * the window's keys are streamed through LDS in blocks of 64 (K-hat normalised on the way in), queries are spread over
//     ceil(n / 64) single-wave workgroups per (image, window, head): LDS use is independent of n;
//   * the continuous position bias is NOT materialised as [H, n, n] (21 MB per block at n = 576, H = 16): each score looks
//     its bias up in the head's column of the (2w-1)^2 x H table 16*sigmoid(MLP(coords)), held in LDS --
//     index(i, j) = (y_i - y_j + w-1)(2w-1) + (x_i - x_j + w-1) = code(i) - code(j) + 2w(w-1) with code(t) = y_t (2w-1) + x_t
//     (HF/swinv2:480-490).  A dense [H, n, n] bias is still accepted (bias != NULL) so that both forms can be cross-checked;
//   * backward = two passes without cross-wave reductions: d q-hat per query block (keys streamed), d k-hat / d v per key block
//     (queries streamed); P is recomputed from the forward's log-sum-exp, delta_i = dO_i . O_i.  The bias gradient is
//     accumulated per table entry (LDS atomics, one global atomic per entry and workgroup) or per dense element.
// Vector-ALU dot products (fp32 parity mode and bf16 alike).
#include <math.h>

#include "common.h"
#include "klab_mm.h"

namespace klab {

template <typename T> struct Vec16L;
template <> struct Vec16L<float> { typedef f32x4 type; static constexpr int N = 4; };
template <> struct Vec16L<bf16_t> { typedef bf16x8 type; static constexpr int N = 8; };

struct SwinLP {
  const void* qkv; void* ctx; const float* bias; const float* btab; const float* logit_scale; float* lse;
  int B, R, w, shift, H, C;
  const void* dctx; void* dqkv; float* dbias; float* dbtab; float* dlogit_scale;
  // window padding (HF/swinv2:645-650): the grid is padded to Rp = ceil(R / w) * w with ZERO input rows, which still act as
  // keys -- k = 0 (no key bias), v = the value bias -- and are cropped from the output
  int Rp; const float* vbias; float* dvbias;
};

__device__ __forceinline__ int swin_region_l(int s, int R, int w, int shift) { return (s >= R - w) + (s >= R - shift); }

__device__ __forceinline__ float wave_sum_l(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// window-local index j of window (wy, wx) of image b -> source token (-1: a padded position), mask region, bias code
struct TokInfo { int tok, reg, code; };
__device__ __forceinline__ TokInfo tok_info(const SwinLP& p, int b, int wy, int wx, int j) {
  const int w = p.w, R = p.R, Rp = p.Rp;
  const int jy = j / w, jx = j - jy * w;
  const int ys = wy * w + jy, xs = wx * w + jx;                 // coordinates in the rolled (padded) grid
  const int y = (ys + p.shift) % Rp, x = (xs + p.shift) % Rp;   // torch.roll(-shift) source (HF/swinv2:667-670)
  TokInfo t;
  t.tok = (y < R && x < R) ? (b * R + y) * R + x : -1;
  t.reg = p.shift > 0 ? swin_region_l(ys, Rp, w, p.shift) * 3 + swin_region_l(xs, Rp, w, p.shift) : 0;  // regions of the padded grid (:675)
  t.code = jy * (2 * w - 1) + jx;
  return t;
}

constexpr int KBLK = 64;

template <typename T, int HD>
struct LdsL {
  static constexpr int VEC = Vec16L<T>::N;
  static constexpr int KST = HD + VEC;  // padded row
};

// ------------------------------------------------------------------------------------------------ forward
template <typename T, int HD>
__global__ __launch_bounds__(64) void swin_attn_fwd_tiled(SwinLP p) {
  constexpr int VEC = LdsL<T, HD>::VEC, KST = LdsL<T, HD>::KST;
  using V = typename Vec16L<T>::type;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int w = p.w, n = w * w, R = p.R, C = p.C, H = p.H;
  const int nWr = p.Rp / w, nW = nWr * nWr, nqb = (n + 63) / 64, ntab = (2 * w - 1) * (2 * w - 1);
  T* Kn = reinterpret_cast<T*>(smem);
  T* Vs = Kn + (size_t)KBLK * KST;
  int* kcode = reinterpret_cast<int*>(Vs + (size_t)KBLK * KST);
  int* kreg = kcode + KBLK;
  float* tab = reinterpret_cast<float*>(kreg + KBLK);
  const int lane = threadIdx.x;
  int bid = blockIdx.x;
  const int qb = bid % nqb; bid /= nqb;
  const int h = bid % H; bid /= H;
  const int win = bid % nW; const int b = bid / nW;
  const int wy = win / nWr, wx = win % nWr;
  const T* qkv = reinterpret_cast<const T*>(p.qkv);
  const long ld = 3L * C;
  if (p.btab)
    for (int t = lane; t < ntab; t += 64) tab[t] = p.btab[(long)t * H + h];

  const int i = qb * 64 + lane;
  TokInfo qi = tok_info(p, b, wy, wx, i < n ? i : n - 1);
  const bool valid = i < n && qi.tok >= 0;  // padded positions are cropped from the output (HF/swinv2:688-690)
  if (qi.tok < 0) qi.tok = 0;
  const float scale = __expf(fminf(p.logit_scale[h], 4.6051701859880914f));  // clamp at ln(100) (HF/swinv2:416)
  float qn[HD];
  {
    const T* qr = qkv + (long)qi.tok * ld + h * HD;
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < HD; c += VEC) {
      V qv = *reinterpret_cast<const V*>(qr + c);
#pragma unroll
      for (int u = 0; u < VEC; ++u) { qn[c + u] = to_f32(qv[u]); ss += qn[c + u] * qn[c + u]; }
    }
    const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);  // F.normalize eps (HF/swinv2:413)
#pragma unroll
    for (int c = 0; c < HD; ++c) qn[c] = to_f32(from_f32<T>(qn[c] * inv)) * scale;  // operand-dtype rounding before the scale
  }
  const int coff = qi.code + 2 * w * (w - 1);
  const float* br = p.bias ? p.bias + ((long)h * n + (i < n ? i : n - 1)) * n : nullptr;
  float m = -INFINITY, l = 0.f;
  float o[HD];
#pragma unroll
  for (int c = 0; c < HD; ++c) o[c] = 0.f;

  for (int j0 = 0; j0 < n; j0 += KBLK) {
    __syncthreads();
    {
      const int j = j0 + lane;
      if (j < n) {
        const TokInfo kj = tok_info(p, b, wy, wx, j);
        kcode[lane] = kj.code; kreg[lane] = kj.reg;
        if (kj.tok < 0) {  // padded key: zero input row => k = 0 (F.normalize(0) = 0), v = value bias
#pragma unroll
          for (int c = 0; c < HD; ++c) {
            Kn[lane * KST + c] = from_f32<T>(0.f);
            Vs[lane * KST + c] = from_f32<T>(p.vbias ? p.vbias[h * HD + c] : 0.f);
          }
        } else {
        const T* kr = qkv + (long)kj.tok * ld + C + h * HD;
        const T* vr = qkv + (long)kj.tok * ld + 2 * C + h * HD;
        float kf[HD];
        float ss = 0.f;
#pragma unroll
        for (int c = 0; c < HD; c += VEC) {
          V kv = *reinterpret_cast<const V*>(kr + c);
#pragma unroll
          for (int u = 0; u < VEC; ++u) { kf[c + u] = to_f32(kv[u]); ss += kf[c + u] * kf[c + u]; }
          *reinterpret_cast<V*>(Vs + lane * KST + c) = *reinterpret_cast<const V*>(vr + c);
        }
        const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
        for (int c = 0; c < HD; c += VEC) {
          V ov;
#pragma unroll
          for (int u = 0; u < VEC; ++u) ov[u] = from_f32<T>(kf[c + u] * inv);
          *reinterpret_cast<V*>(Kn + lane * KST + c) = ov;
        }
        }
      }
    }
    __syncthreads();
    const int jn = n - j0 < KBLK ? n - j0 : KBLK;
    for (int jj = 0; jj < jn; ++jj) {
      float s = 0.f;
      const T* kr = Kn + jj * KST;
#pragma unroll
      for (int c = 0; c < HD; c += VEC) {
        V kv = *reinterpret_cast<const V*>(kr + c);
#pragma unroll
        for (int u = 0; u < VEC; ++u) s += qn[c + u] * to_f32(kv[u]);
      }
      s += br ? br[j0 + jj] : tab[coff - kcode[jj]];
      if (kreg[jj] != qi.reg) s += -200.f;  // -100 added twice by the pinned transformers (HF/swinv2:433-436)
      const float mn = fmaxf(m, s);
      const float corr = __expf(m - mn);
      const float pe = __expf(s - mn);
      l = l * corr + pe;
      const T* vr = Vs + jj * KST;
#pragma unroll
      for (int c = 0; c < HD; c += VEC) {
        V vv = *reinterpret_cast<const V*>(vr + c);
#pragma unroll
        for (int u = 0; u < VEC; ++u) o[c + u] = o[c + u] * corr + pe * to_f32(vv[u]);
      }
      m = mn;
    }
  }
  if (valid) {
    const float il = 1.f / l;
    T* orow = reinterpret_cast<T*>(p.ctx) + (long)qi.tok * C + h * HD;
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

// ------------------------------------------------------------------------------------------------ backward, pass 1: d q
// lane = one query row; keys streamed.  Also d(bias) (table entries or dense elements) and d(logit_scale).
template <typename T, int HD>
__global__ __launch_bounds__(64) void swin_attn_bwd_dq_tiled(SwinLP p) {
  constexpr int VEC = LdsL<T, HD>::VEC, KST = LdsL<T, HD>::KST;
  using V = typename Vec16L<T>::type;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int w = p.w, n = w * w, R = p.R, C = p.C, H = p.H;
  const int nWr = p.Rp / w, nW = nWr * nWr, nqb = (n + 63) / 64, ntab = (2 * w - 1) * (2 * w - 1);
  T* Kn = reinterpret_cast<T*>(smem);
  T* Vs = Kn + (size_t)KBLK * KST;
  int* kcode = reinterpret_cast<int*>(Vs + (size_t)KBLK * KST);
  int* kreg = kcode + KBLK;
  float* tab = reinterpret_cast<float*>(kreg + KBLK);
  float* dtab = tab + ntab;
  const int lane = threadIdx.x;
  int bid = blockIdx.x;
  const int qb = bid % nqb; bid /= nqb;
  const int h = bid % H; bid /= H;
  const int win = bid % nW; const int b = bid / nW;
  const int wy = win / nWr, wx = win % nWr;
  const T* qkv = reinterpret_cast<const T*>(p.qkv);
  const long ld = 3L * C;
  const bool table = p.btab != nullptr;
  if (table)
    for (int t = lane; t < ntab; t += 64) { tab[t] = p.btab[(long)t * H + h]; dtab[t] = 0.f; }

  const int i = qb * 64 + lane;
  const int ic = i < n ? i : n - 1;
  TokInfo qi = tok_info(p, b, wy, wx, ic);
  const bool valid = i < n && qi.tok >= 0;  // padded queries: cropped output, no gradient
  if (qi.tok < 0) qi.tok = 0;
  const float lsv = p.logit_scale[h];
  const bool clamped = lsv > 4.6051701859880914f;
  const float scale = __expf(fminf(lsv, 4.6051701859880914f));
  float qh[HD], dO[HD];
  float qinv, delta = 0.f;
  {
    const T* qr = qkv + (long)qi.tok * ld + h * HD;
    const T* dor = reinterpret_cast<const T*>(p.dctx) + (long)qi.tok * C + h * HD;
    const T* orow = reinterpret_cast<const T*>(p.ctx) + (long)qi.tok * C + h * HD;
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < HD; c += VEC) {
      V qv = *reinterpret_cast<const V*>(qr + c);
      V dv = *reinterpret_cast<const V*>(dor + c);
      V ov = *reinterpret_cast<const V*>(orow + c);
#pragma unroll
      for (int u = 0; u < VEC; ++u) {
        qh[c + u] = to_f32(qv[u]); ss += qh[c + u] * qh[c + u];
        dO[c + u] = to_f32(dv[u]);
        delta += dO[c + u] * to_f32(ov[u]);  // sum_j P_ij (dO_i . V_j) = dO_i . O_i
      }
    }
    qinv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
    for (int c = 0; c < HD; ++c) qh[c] = to_f32(from_f32<T>(qh[c] * qinv));  // unit q-hat, operand-dtype rounding as in forward
  }
  const float lse = p.lse[(((long)b * nW + win) * H + h) * n + ic];
  const int coff = qi.code + 2 * w * (w - 1);
  const float* br = p.bias ? p.bias + ((long)h * n + ic) * n : nullptr;
  float* dbr = p.dbias ? p.dbias + ((long)h * n + ic) * n : nullptr;
  float dqh[HD];
#pragma unroll
  for (int c = 0; c < HD; ++c) dqh[c] = 0.f;
  float dscale = 0.f;

  for (int j0 = 0; j0 < n; j0 += KBLK) {
    __syncthreads();
    {
      const int j = j0 + lane;
      if (j < n) {
        const TokInfo kj = tok_info(p, b, wy, wx, j);
        kcode[lane] = kj.code; kreg[lane] = kj.reg;
        if (kj.tok < 0) {  // padded key: zero input row => k = 0 (F.normalize(0) = 0), v = value bias
#pragma unroll
          for (int c = 0; c < HD; ++c) {
            Kn[lane * KST + c] = from_f32<T>(0.f);
            Vs[lane * KST + c] = from_f32<T>(p.vbias ? p.vbias[h * HD + c] : 0.f);
          }
        } else {
        const T* kr = qkv + (long)kj.tok * ld + C + h * HD;
        const T* vr = qkv + (long)kj.tok * ld + 2 * C + h * HD;
        float kf[HD];
        float ss = 0.f;
#pragma unroll
        for (int c = 0; c < HD; c += VEC) {
          V kv = *reinterpret_cast<const V*>(kr + c);
#pragma unroll
          for (int u = 0; u < VEC; ++u) { kf[c + u] = to_f32(kv[u]); ss += kf[c + u] * kf[c + u]; }
          *reinterpret_cast<V*>(Vs + lane * KST + c) = *reinterpret_cast<const V*>(vr + c);
        }
        const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
        for (int c = 0; c < HD; c += VEC) {
          V ov;
#pragma unroll
          for (int u = 0; u < VEC; ++u) ov[u] = from_f32<T>(kf[c + u] * inv);
          *reinterpret_cast<V*>(Kn + lane * KST + c) = ov;
        }
        }
      }
    }
    __syncthreads();
    const int jn = n - j0 < KBLK ? n - j0 : KBLK;
    for (int jj = 0; jj < jn; ++jj) {
      float cs = 0.f, dp = 0.f;
      const T* kr = Kn + jj * KST;
      const T* vr = Vs + jj * KST;
#pragma unroll
      for (int c = 0; c < HD; c += VEC) {
        V kv = *reinterpret_cast<const V*>(kr + c);
        V vv = *reinterpret_cast<const V*>(vr + c);
#pragma unroll
        for (int u = 0; u < VEC; ++u) { cs += qh[c + u] * to_f32(kv[u]); dp += dO[c + u] * to_f32(vv[u]); }
      }
      const int ti = coff - kcode[jj];
      float s = cs * scale + (br ? br[j0 + jj] : tab[ti]);
      if (kreg[jj] != qi.reg) s += -200.f;
      const float pr = __expf(s - lse);
      const float ds = valid ? pr * (dp - delta) : 0.f;
      if (table) atomicAdd(&dtab[ti], ds);
      else if (dbr && valid) atomicAdd(dbr + j0 + jj, ds);  // summed over all images and windows
      dscale += ds * cs;
      const float dcs = ds * scale;
#pragma unroll
      for (int c = 0; c < HD; c += VEC) {
        V kv = *reinterpret_cast<const V*>(kr + c);
#pragma unroll
        for (int u = 0; u < VEC; ++u) dqh[c + u] += dcs * to_f32(kv[u]);
      }
    }
  }
  if (valid) {  // through q-hat = q / ||q||:  dq = (dqh - qh (qh . dqh)) / ||q||
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < HD; ++c) dot += qh[c] * dqh[c];
    T* dqr = reinterpret_cast<T*>(p.dqkv) + (long)qi.tok * ld + h * HD;
#pragma unroll
    for (int c = 0; c < HD; c += VEC) {
      V ov;
#pragma unroll
      for (int u = 0; u < VEC; ++u) ov[u] = from_f32<T>((dqh[c + u] - qh[c + u] * dot) * qinv);
      *reinterpret_cast<V*>(dqr + c) = ov;
    }
  }
  dscale = wave_sum_l(dscale);
  if (lane == 0 && p.dlogit_scale && !clamped) atomicAdd(p.dlogit_scale + h, dscale * scale);
  if (table && p.dbtab) {
    __syncthreads();
    for (int t = lane; t < ntab; t += 64) {
      const float v = dtab[t];
      if (v != 0.f) atomicAdd(p.dbtab + (long)t * H + h, v);
    }
  }
}

// ------------------------------------------------------------------------------------------------ backward, pass 2: d k, d v
// lane = one key row (k-hat, v in registers); queries streamed through LDS with their lse and delta.
template <typename T, int HD>
__global__ __launch_bounds__(64) void swin_attn_bwd_dkv_tiled(SwinLP p) {
  constexpr int VEC = LdsL<T, HD>::VEC, KST = LdsL<T, HD>::KST;
  using V = typename Vec16L<T>::type;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int w = p.w, n = w * w, R = p.R, C = p.C, H = p.H;
  const int nWr = p.Rp / w, nW = nWr * nWr, nkb = (n + 63) / 64, ntab = (2 * w - 1) * (2 * w - 1);
  T* Qn = reinterpret_cast<T*>(smem);
  T* dOs = Qn + (size_t)KBLK * KST;
  float* qlse = reinterpret_cast<float*>(dOs + (size_t)KBLK * KST);
  float* qdel = qlse + KBLK;
  int* qcode = reinterpret_cast<int*>(qdel + KBLK);
  int* qreg = qcode + KBLK;
  float* tab = reinterpret_cast<float*>(qreg + KBLK);
  const int lane = threadIdx.x;
  int bid = blockIdx.x;
  const int kb = bid % nkb; bid /= nkb;
  const int h = bid % H; bid /= H;
  const int win = bid % nW; const int b = bid / nW;
  const int wy = win / nWr, wx = win % nWr;
  const T* qkv = reinterpret_cast<const T*>(p.qkv);
  const long ld = 3L * C;
  if (p.btab)
    for (int t = lane; t < ntab; t += 64) tab[t] = p.btab[(long)t * H + h];

  const int j = kb * 64 + lane;
  const bool valid = j < n;
  const int jc = valid ? j : n - 1;
  const TokInfo kj = tok_info(p, b, wy, wx, jc);
  const bool kpad = kj.tok < 0;  // padded key: k = 0, v = value bias; its d v belongs to the value-bias gradient
  const float scale = __expf(fminf(p.logit_scale[h], 4.6051701859880914f));
  float kh[HD], vj[HD];
  float kinv = 0.f;
  if (kpad) {
#pragma unroll
    for (int c = 0; c < HD; ++c) { kh[c] = 0.f; vj[c] = to_f32(from_f32<T>(p.vbias ? p.vbias[h * HD + c] : 0.f)); }
  } else {
    const T* kr = qkv + (long)kj.tok * ld + C + h * HD;
    const T* vr = qkv + (long)kj.tok * ld + 2 * C + h * HD;
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < HD; c += VEC) {
      V kv = *reinterpret_cast<const V*>(kr + c);
      V vv = *reinterpret_cast<const V*>(vr + c);
#pragma unroll
      for (int u = 0; u < VEC; ++u) { kh[c + u] = to_f32(kv[u]); ss += kh[c + u] * kh[c + u]; vj[c + u] = to_f32(vv[u]); }
    }
    kinv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
    for (int c = 0; c < HD; ++c) kh[c] = to_f32(from_f32<T>(kh[c] * kinv));
  }
  const int koff = 2 * w * (w - 1) - kj.code;
  float dkh[HD], dv[HD];
#pragma unroll
  for (int c = 0; c < HD; ++c) { dkh[c] = 0.f; dv[c] = 0.f; }

  for (int i0 = 0; i0 < n; i0 += KBLK) {
    __syncthreads();
    {
      const int i = i0 + lane;
      const TokInfo qi0 = tok_info(p, b, wy, wx, i < n ? i : n - 1);
      if (i < n && qi0.tok < 0) {  // padded query: P row = 0 (lse = +inf), nothing flows back through it
        qcode[lane] = qi0.code; qreg[lane] = qi0.reg;
#pragma unroll
        for (int c = 0; c < HD; ++c) { Qn[lane * KST + c] = from_f32<T>(0.f); dOs[lane * KST + c] = from_f32<T>(0.f); }
        qlse[lane] = INFINITY;
        qdel[lane] = 0.f;
      } else if (i < n) {
        const TokInfo qi = qi0;
        qcode[lane] = qi.code; qreg[lane] = qi.reg;
        const T* qr = qkv + (long)qi.tok * ld + h * HD;
        const T* dor = reinterpret_cast<const T*>(p.dctx) + (long)qi.tok * C + h * HD;
        const T* orow = reinterpret_cast<const T*>(p.ctx) + (long)qi.tok * C + h * HD;
        float qf[HD];
        float ss = 0.f, dl = 0.f;
#pragma unroll
        for (int c = 0; c < HD; c += VEC) {
          V qv = *reinterpret_cast<const V*>(qr + c);
          V dvv = *reinterpret_cast<const V*>(dor + c);
          V ov = *reinterpret_cast<const V*>(orow + c);
#pragma unroll
          for (int u = 0; u < VEC; ++u) { qf[c + u] = to_f32(qv[u]); ss += qf[c + u] * qf[c + u]; dl += to_f32(dvv[u]) * to_f32(ov[u]); }
          *reinterpret_cast<V*>(dOs + lane * KST + c) = dvv;
        }
        const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
        for (int c = 0; c < HD; c += VEC) {
          V ov;
#pragma unroll
          for (int u = 0; u < VEC; ++u) ov[u] = from_f32<T>(qf[c + u] * inv);
          *reinterpret_cast<V*>(Qn + lane * KST + c) = ov;
        }
        qlse[lane] = p.lse[(((long)b * nW + win) * H + h) * n + i];
        qdel[lane] = dl;
      }
    }
    __syncthreads();
    const int in = n - i0 < KBLK ? n - i0 : KBLK;
    for (int ii = 0; ii < in; ++ii) {
      float cs = 0.f, dp = 0.f;
      const T* qr = Qn + ii * KST;
      const T* dor = dOs + ii * KST;
#pragma unroll
      for (int c = 0; c < HD; c += VEC) {
        V qv = *reinterpret_cast<const V*>(qr + c);
        V dvv = *reinterpret_cast<const V*>(dor + c);
#pragma unroll
        for (int u = 0; u < VEC; ++u) { cs += to_f32(qv[u]) * kh[c + u]; dp += to_f32(dvv[u]) * vj[c + u]; }
      }
      float s = cs * scale + (p.bias ? p.bias[((long)h * n + (i0 + ii)) * n + jc] : tab[qcode[ii] + koff]);
      if (qreg[ii] != kj.reg) s += -200.f;
      const float pr = __expf(s - qlse[ii]);
      const float dcs = pr * (dp - qdel[ii]) * scale;
#pragma unroll
      for (int c = 0; c < HD; c += VEC) {
        V qv = *reinterpret_cast<const V*>(qr + c);
        V dvv = *reinterpret_cast<const V*>(dor + c);
#pragma unroll
        for (int u = 0; u < VEC; ++u) { dkh[c + u] += dcs * to_f32(qv[u]); dv[c + u] += pr * to_f32(dvv[u]); }
      }
    }
  }
  if (valid && kpad) {
    if (p.dvbias) {
#pragma unroll
      for (int c = 0; c < HD; ++c) atomicAdd(p.dvbias + h * HD + c, dv[c]);
    }
  } else if (valid) {  // through k-hat = k / ||k||
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < HD; ++c) dot += kh[c] * dkh[c];
    T* dkr = reinterpret_cast<T*>(p.dqkv) + (long)kj.tok * ld + C + h * HD;
    T* dvr = reinterpret_cast<T*>(p.dqkv) + (long)kj.tok * ld + 2 * C + h * HD;
#pragma unroll
    for (int c = 0; c < HD; c += VEC) {
      V ok, ov;
#pragma unroll
      for (int u = 0; u < VEC; ++u) { ok[u] = from_f32<T>((dkh[c + u] - kh[c + u] * dot) * kinv); ov[u] = from_f32<T>(dv[c + u]); }
      *reinterpret_cast<V*>(dkr + c) = ok;
      *reinterpret_cast<V*>(dvr + c) = ov;
    }
  }
}

// ------------------------------------------------------------------------------------------------ CPB table forms
__global__ void cpb_sigmoid_kernel(const float* __restrict__ table, float* __restrict__ btab, int tot) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < tot) btab[i] = 16.f / (1.f + __expf(-table[i]));  // HF/swinv2:427-428
}
__global__ void cpb_dtable_from_btab_kernel(const float* __restrict__ dbtab, const float* __restrict__ btab, float* __restrict__ dtable, int tot) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < tot) { const float bv = btab[i]; dtable[i] = dbtab[i] * bv * (1.f - bv * 0.0625f); }  // d(16 sigmoid(t)) / dt
}
// MLP backward for any table size: one workgroup per hidden unit j, threads over table rows, fixed-order reductions.
//   dW2[h, j] += sum_t dt[t, h] hid[t, j];   g[t] = [hid[t, j] > 0] sum_h dt[t, h] W2[h, j];
//   dW0[j, :] += sum_t g[t] coords[t, :];    db0[j] += sum_t g[t]
constexpr int CPB_MAXH = 64;
__global__ __launch_bounds__(256) void cpb_mlp_bwd_any_kernel(const float* __restrict__ dtable, const float* __restrict__ coords,
                                                              const float* __restrict__ hidden, const float* __restrict__ w2,
                                                              float* __restrict__ dw0, float* __restrict__ db0, float* __restrict__ dw2, int ntab,
                                                              int H, int nh) {
  __shared__ float red[4][CPB_MAXH + 3];
  const int j = blockIdx.x, tid = threadIdx.x;
  float s[CPB_MAXH];
#pragma unroll
  for (int h = 0; h < CPB_MAXH; ++h) s[h] = 0.f;
  float a0 = 0.f, a1 = 0.f, ab = 0.f;
  for (int t = tid; t < ntab; t += 256) {
    const float hv = hidden[(long)t * nh + j];
    float g = 0.f;
#pragma unroll
    for (int h = 0; h < CPB_MAXH; ++h)
      if (h < H) { const float d = dtable[t * H + h]; s[h] += d * hv; g += d * w2[h * nh + j]; }
    if (hv > 0.f) { a0 += g * coords[t * 2]; a1 += g * coords[t * 2 + 1]; ab += g; }  // ReLU gate
  }
  const int wv = tid >> 6;
#pragma unroll
  for (int h = 0; h < CPB_MAXH; ++h)
    if (h < H) { const float r = wave_sum_l(s[h]); if ((tid & 63) == 0) red[wv][h] = r; }
  a0 = wave_sum_l(a0); a1 = wave_sum_l(a1); ab = wave_sum_l(ab);
  if ((tid & 63) == 0) { red[wv][CPB_MAXH] = a0; red[wv][CPB_MAXH + 1] = a1; red[wv][CPB_MAXH + 2] = ab; }
  __syncthreads();
  if (tid < H) dw2[tid * nh + j] += red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
  if (tid == 64) dw0[j * 2] += red[0][CPB_MAXH] + red[1][CPB_MAXH] + red[2][CPB_MAXH] + red[3][CPB_MAXH];
  if (tid == 65) dw0[j * 2 + 1] += red[0][CPB_MAXH + 1] + red[1][CPB_MAXH + 1] + red[2][CPB_MAXH + 1] + red[3][CPB_MAXH + 1];
  if (tid == 66) db0[j] += red[0][CPB_MAXH + 2] + red[1][CPB_MAXH + 2] + red[2][CPB_MAXH + 2] + red[3][CPB_MAXH + 2];
}

int cpb_mlp_bwd_any_launch(const float* dtable, const float* coords, const float* hidden, const float* w2, float* dw0, float* db0,
                           float* dw2, int ntab, int heads, int nhidden, hipStream_t s) {
  if (heads > CPB_MAXH) return KLAB_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(cpb_mlp_bwd_any_kernel, dim3(nhidden), dim3(256), 0, s, dtable, coords, hidden, w2, dw0, db0, dw2, ntab, heads, nhidden);
  KLAB_LAUNCH_CHECK();
  return KLAB_OK;
}

// ------------------------------------------------------------------------------------------------ launchers
template <typename T, int HD>
static size_t tiled_lds(int ntab, bool table, int mode) {
  constexpr int KST = LdsL<T, HD>::KST;
  size_t b = 2 * (size_t)KBLK * KST * sizeof(T);
  if (mode == 2) b += 4 * (size_t)KBLK * 4;  // lse, delta, code, region
  else b += 2 * (size_t)KBLK * 4;            // code, region
  if (table) b += (size_t)ntab * 4 * (mode == 1 ? 2 : 1);
  return b;
}

template <typename T, int HD>
static int launch_tiled(const SwinLP& p, bool backward, hipStream_t s) {
  const int n = p.w * p.w, nW = (p.Rp / p.w) * (p.Rp / p.w), nb = (n + 63) / 64, ntab = (2 * p.w - 1) * (2 * p.w - 1);
  const bool table = p.btab != nullptr;
  const unsigned grid = (unsigned)((long)p.B * nW * p.H * nb);
  if (!backward) {
    const size_t lds = tiled_lds<T, HD>(ntab, table, 0);
    int rc = ensure_dyn_lds(reinterpret_cast<const void*>(swin_attn_fwd_tiled<T, HD>), lds);
    if (rc) return rc;
    hipLaunchKernelGGL((swin_attn_fwd_tiled<T, HD>), dim3(grid), dim3(64), lds, s, p);
    KLAB_LAUNCH_CHECK();
    return KLAB_OK;
  }
  size_t lds = tiled_lds<T, HD>(ntab, table, 1);
  int rc = ensure_dyn_lds(reinterpret_cast<const void*>(swin_attn_bwd_dq_tiled<T, HD>), lds);
  if (rc) return rc;
  hipLaunchKernelGGL((swin_attn_bwd_dq_tiled<T, HD>), dim3(grid), dim3(64), lds, s, p);
  KLAB_LAUNCH_CHECK();
  lds = tiled_lds<T, HD>(ntab, table, 2);
  rc = ensure_dyn_lds(reinterpret_cast<const void*>(swin_attn_bwd_dkv_tiled<T, HD>), lds);
  if (rc) return rc;
  hipLaunchKernelGGL((swin_attn_bwd_dkv_tiled<T, HD>), dim3(grid), dim3(64), lds, s, p);
  KLAB_LAUNCH_CHECK();
  return KLAB_OK;
}

// called by klab_swin_attn_fwd / klab_swin_attn_bwd (attn_swin.hip) for windows of more than 64 tokens, or whenever the
// caller supplies the bias as a table
int swin_attn_large_dispatch(const klab_swin_attn_args* a, bool backward, hipStream_t s) {
  if (!a->bias && !a->bias_table) return KLAB_ERR_BADARG;
  SwinLP p{a->qkv, a->ctx, a->bias, a->bias ? nullptr : a->bias_table, a->logit_scale, a->lse, a->B, a->R, a->w, a->shift, a->H, a->C,
           a->dctx, a->dqkv, a->bias ? a->dbias : nullptr, a->bias ? nullptr : a->dbias_table, a->dlogit_scale,
           (a->R + a->w - 1) / a->w * a->w, a->v_bias, a->dv_bias};
  const int hd = a->C / a->H;
  if (a->dtype == KLAB_BF16) {
    if (a->C & 7) return KLAB_ERR_UNSUPPORTED;
    if (hd == 32) return launch_tiled<bf16_t, 32>(p, backward, s);
    if (hd == 16) return launch_tiled<bf16_t, 16>(p, backward, s);
  } else if (a->dtype == KLAB_F32) {
    if (a->C & 3) return KLAB_ERR_UNSUPPORTED;
    if (hd == 32) return launch_tiled<float, 32>(p, backward, s);
    if (hd == 16) return launch_tiled<float, 16>(p, backward, s);
  }
  return KLAB_ERR_UNSUPPORTED;
}

}  // namespace klab

using namespace klab;

extern "C" int klab_swin_cpb_table(const float* coords, const float* w0, const float* b0, const float* w2, float* table, float* hidden,
                                   float* bias_table, int ntab, int heads, int nhidden, void* stream);
extern "C" int klab_swin_cpb_table_bwd(const float* dbias_table, const float* bias_table, const float* coords, const float* hidden,
                                       const float* w2, float* dtable, float* dw0, float* db0, float* dw2, int ntab, int heads, int nhidden,
                                       void* stream) {
  if (!dbias_table || !bias_table || !coords || !hidden || !w2 || !dtable || !dw0 || !db0 || !dw2) return KLAB_ERR_BADARG;
  if (heads > CPB_MAXH || nhidden <= 0) return KLAB_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const int tot = ntab * heads;
  hipLaunchKernelGGL(cpb_dtable_from_btab_kernel, dim3((tot + 255) / 256), dim3(256), 0, s, dbias_table, bias_table, dtable, tot);
  KLAB_LAUNCH_CHECK();
  hipLaunchKernelGGL(cpb_mlp_bwd_any_kernel, dim3(nhidden), dim3(256), 0, s, dtable, coords, hidden, w2, dw0, db0, dw2, ntab, heads, nhidden);
  KLAB_LAUNCH_CHECK();
  return KLAB_OK;
}

namespace klab { int cpb_table_launch(const float* coords, const float* w0, const float* b0, const float* w2, float* table, float* hidden,
                                      int ntab, int heads, int nhidden, hipStream_t s); }

extern "C" int klab_swin_cpb_table(const float* coords, const float* w0, const float* b0, const float* w2, float* table, float* hidden,
                                   float* bias_table, int ntab, int heads, int nhidden, void* stream) {
  if (!coords || !w0 || !b0 || !w2 || !table || !bias_table) return KLAB_ERR_BADARG;
  if (nhidden > 512) return KLAB_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const int rc = cpb_table_launch(coords, w0, b0, w2, table, hidden, ntab, heads, nhidden, s);
  if (rc) return rc;
  const int tot = ntab * heads;
  hipLaunchKernelGGL(cpb_sigmoid_kernel, dim3((tot + 255) / 256), dim3(256), 0, s, table, bias_table, tot);
  KLAB_LAUNCH_CHECK();
  return KLAB_OK;
}
