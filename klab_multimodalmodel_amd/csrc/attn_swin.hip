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
// This first kernel is the generic form (fp32 and any head dim: vector-ALU dot products); bf16 with head dim 32 and
// windows of <= 64 tokens -- every standard Swin-V2 -- runs on the matrix cores instead: swin_attn_fwd_mfma32 /
// swin_qkv_attn_fused below (49-token windows padded to 64) and, backward, the gather -> t5_attn_bwd_mfma<32> -> scatter
// sequence of swin_attn_bwd_mfma.
#include <math.h>
#include <stdlib.h>
#include <string.h>

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
    // q-hat is rounded to the operand dtype BEFORE the logit scale (as the matrix-core kernel feeds it to the MFMA), so
    // that forward and backward of either kernel family recompute bit-compatible scores
    const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
    for (int c = 0; c < HD; ++c) qn[c] = to_f32(from_f32<T>(qn[c] * inv)) * scale;
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


// ---- matrix-core forward (bf16, head dim 32, windows of <= 64 tokens: every standard Swin-V2 at 224/256 px) ----
// One wave per (image, window, head).  K-hat (L2-normalised keys) sits in LDS as a row image, V as a transposed-read
// image; S^T = K-hat Q-hat^T is computed swapped so that a lane owns one query column and 4 consecutive keys per tile:
// the softmax is in-register + two shuffles, and P^T tile pairs are directly the B operand of O^T = V^T P^T (same
// construction as t5_attn_fwd_mfma).  49-token windows are padded to 64 with -inf scores / zero V rows.
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4_sw;
__device__ __forceinline__ int swin_tr_off(int row, int col) { return (row >> 3) * 896 + (row & 7) * 96 + col * 2; }  // PD = 24 dwords

__global__ __launch_bounds__(64) void swin_attn_fwd_mfma32(SwinAttnP p) {
  constexpr int HD = 32, KP = 80;  // K row pitch: 64 B + 16 B pad
  __shared__ __attribute__((aligned(16))) char Kr[64 * KP];
  __shared__ __attribute__((aligned(16))) char Vt[8 * 896];
  __shared__ int tok[64];
  __shared__ int reg[64];
  const int w = p.w, n = w * w, R = p.R, C = p.C, H = p.H;
  const int nWr = R / w, nW = nWr * nWr;
  const int lane = threadIdx.x, g = lane >> 4;
  int bid = blockIdx.x;
  const int h = bid % H; bid /= H;
  const int win = bid % nW; const int b = bid / nW;
  const int wy = win / nWr, wx = win % nWr;
  const bf16_t* qkv = reinterpret_cast<const bf16_t*>(p.qkv);
  const long ld = 3L * C;
  {  // stage keys / values: 4 lanes per token row (16 B each), 16 rows per pass -> every load instruction reads 16 whole
     // 64-byte head slices (one lane per row touched 49 different cache lines with 16 B each)
    const int sub = lane & 3, r4 = lane >> 2;
    bf16x8 kc[4], vc[4];
    int tk[4], rgs[4];
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      const int j = ps * 16 + r4;
      kc[ps] = bf16x8{}; vc[ps] = bf16x8{};
      tk[ps] = 0; rgs[ps] = -1;
      if (j < n) {
        const int ys = wy * w + j / w, xs = wx * w + j % w;
        const int y = (ys + p.shift) % R, x = (xs + p.shift) % R;
        tk[ps] = (b * R + y) * R + x;
        rgs[ps] = p.shift > 0 ? swin_region(ys, R, w, p.shift) * 3 + swin_region(xs, R, w, p.shift) : 0;
        kc[ps] = *reinterpret_cast<const bf16x8*>(qkv + (long)tk[ps] * ld + C + h * HD + sub * 8);
        vc[ps] = *reinterpret_cast<const bf16x8*>(qkv + (long)tk[ps] * ld + 2 * C + h * HD + sub * 8);
      }
    }
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      const int j = ps * 16 + r4;
      float ss = 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) { const float f = (float)kc[ps][u]; ss += f * f; }
      ss += __shfl_xor(ss, 1, 64);
      ss += __shfl_xor(ss, 2, 64);
      const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
      for (int u = 0; u < 8; ++u) kc[ps][u] = (bf16_t)((float)kc[ps][u] * inv);
      if (sub == 0) { tok[j] = tk[ps]; reg[j] = rgs[ps]; }
      *reinterpret_cast<bf16x8*>(Kr + j * KP + sub * 16) = kc[ps];
      *reinterpret_cast<bf16x8*>(Vt + swin_tr_off(j, sub * 8)) = vc[ps];
    }
  }
  __syncthreads();
  const float scale = __expf(fminf(p.logit_scale[h], 4.6051701859880914f));
  bf16_t* ctx = reinterpret_cast<bf16_t*>(p.ctx);
  const int nqt = (n + 15) / 16;
  // Q fragment and the query's 64 bias values are fetched one q-tile ahead (they were a global round trip per tile)
  auto fetch = [&](int qt, bf16x8& qv, f32x4 (&bv)[4]) {
    const int q = qt * 16 + (lane & 15);
    const int qc = q < n ? q : n - 1;
    qv = *reinterpret_cast<const bf16x8*>(qkv + (long)tok[qc] * ld + h * HD + g * 8);
    const float* brow = p.bias + ((long)h * n + qc) * n;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const int key = t * 16 + g * 4 + r; bv[t][r] = brow[key < n ? key : n - 1]; }
  };
  bf16x8 qnext; f32x4 bnext[4];
  fetch(0, qnext, bnext);
  for (int qt = 0; qt < nqt; ++qt) {
    const int q = qt * 16 + (lane & 15);
    const int qc = q < n ? q : n - 1;
    const int tq = tok[qc], rq = reg[qc];
    bf16x8 qv = qnext;
    f32x4 bcur[4] = {bnext[0], bnext[1], bnext[2], bnext[3]};
    if (qt + 1 < nqt) fetch(qt + 1, qnext, bnext);
    // Q-hat fragment: 8 of the row's 32 values per lane; the row norm is completed across the 4 lane groups
    float ss = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) { const float f = (float)qv[u]; ss += f * f; }
    ss += __shfl_xor(ss, 16, 64);
    ss += __shfl_xor(ss, 32, 64);
    const float qs = 1.f / fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
    for (int u = 0; u < 8; ++u) qv[u] = (bf16_t)((float)qv[u] * qs);
    f32x4 st[4];
    float m = -INFINITY;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Kr + (t * 16 + (lane & 15)) * KP + g * 16);
      st[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qv, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = t * 16 + g * 4 + r;
        // branch-free (reg[] has all 64 slots, the bias fetch is clamped): -100 twice for another shift region (HF/swinv2:433-436)
        float x = st[t][r] * scale + bcur[t][r] + (reg[key] != rq ? -200.f : 0.f);
        x = key < n ? x : -INFINITY;
        st[t][r] = x;
        m = fmaxf(m, x);
      }
    }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float e = __expf(st[t][r] - m); st[t][r] = e; sum += e; }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.f / sum;
    f32x4 o[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int sidx = 0; sidx < 2; ++sidx) {
      const f32x4 a = st[2 * sidx], bb = st[2 * sidx + 1];
      const bf16x8 pf = {(bf16_t)(a[0] * inv), (bf16_t)(a[1] * inv), (bf16_t)(a[2] * inv), (bf16_t)(a[3] * inv),
                         (bf16_t)(bb[0] * inv), (bf16_t)(bb[1] * inv), (bf16_t)(bb[2] * inv), (bf16_t)(bb[3] * inv)};
      const int q4 = (lane & 15) >> 2, pp = lane & 3;
      const int r0 = sidx * 32 + 4 * g + q4;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_sw*)(Vt + swin_tr_off(r0, dt * 16 + 4 * pp)));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_sw*)(Vt + swin_tr_off(r0 + 16, dt * 16 + 4 * pp)));
        const bf16x8 vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, o[dt], 0, 0, 0);
      }
    }
    if (q < n) {
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
        *reinterpret_cast<bf16x4*>(ctx + (long)tq * C + h * HD + dt * 16 + g * 4) =
            bf16x4{(bf16_t)o[dt][0], (bf16_t)o[dt][1], (bf16_t)o[dt][2], (bf16_t)o[dt][3]};
      if (g == 0 && p.lse) p.lse[(((long)b * nW + win) * H + h) * n + q] = m + __logf(sum);
    }
  }
}

// ---- frozen tower: QKV projection fused into the window attention (bf16, head dim 32, windows <= 64 tokens, C = 64 / 128) ----
// One wave per (image, window, head) computes its head's q | k | v = x_win Wqkv_h^T + b from the window's LN'd input rows
// (swapped MFMA: a lane gets 4 consecutive features of token lane&15), normalises q and k in fp32, parks K-hat / V in its
// private LDS images and keeps Q-hat in registers: two 16-feature tiles of a token are exactly the B operand of the
// score MFMA in the permuted feature order kappa(g, j) = 16 (j >> 2) + 4 g + (j & 3), so K-hat fragments are read in that
// order too.  The [M, 3C] q|k|v tensor (77 MB written and read back per stage-0 block at B = 64) never exists.
struct SwinQkvP {
  const bf16_t* x; const bf16_t* wqkv; const float* bqkv;  // x [M, C]; wqkv [3C, C] rows q | k | v; bqkv [3C] (k part zero) or null
  bf16_t* ctx; const float* bias; const float* logit_scale;
  int B, R, w, shift, H;
};

template <int C>
__global__ __launch_bounds__(64) void swin_qkv_attn_fused(SwinQkvP p) {
  constexpr int HD = 32, KP = 80, KB = C / 32;
  __shared__ __attribute__((aligned(16))) char Kr[64 * KP];
  __shared__ __attribute__((aligned(16))) char Vt[8 * 896];
  __shared__ int tok[64];
  __shared__ int reg[64];
  const int w = p.w, n = w * w, R = p.R, H = p.H;
  const int nWr = R / w, nW = nWr * nWr;
  const int lane = threadIdx.x, g = lane >> 4, lr = lane & 15;
  int bid = blockIdx.x;
  const int h = bid % H; bid /= H;
  const int win = bid % nW; const int b = bid / nW;
  const int wy = win / nWr, wx = win % nWr;
  {  // token index + shift-mask region of window slot `lane`
    int t = 0, rg = -1;
    if (lane < n) {
      const int ys = wy * w + lane / w, xs = wx * w + lane % w;
      const int y = (ys + p.shift) % R, x = (xs + p.shift) % R;
      t = (b * R + y) * R + x;
      rg = p.shift > 0 ? swin_region(ys, R, w, p.shift) * 3 + swin_region(xs, R, w, p.shift) : 0;
    }
    tok[lane] = t; reg[lane] = rg;
  }
  __syncthreads();
  bf16x8 qh[4];  // Q-hat of the 4 query tiles, kappa feature order
  // d[j][r] = feature 16 (j&1) + 4 g + r of q (j = 0,1) / k (2,3) / v (4,5) for the token in window slot mt*16 + lane&15:
  // L2 norms over the 32 features (in-lane + lane groups), Q-hat to registers, K-hat / V to the LDS images
  auto finish_tile = [&](int mt, const f32x4 (&d)[6]) {
    const int j0 = mt * 16 + lr;
    float sq = 0.f, sk = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) { sq += d[0][r] * d[0][r] + d[1][r] * d[1][r]; sk += d[2][r] * d[2][r] + d[3][r] * d[3][r]; }
    sq += __shfl_xor(sq, 16, 64); sq += __shfl_xor(sq, 32, 64);
    sk += __shfl_xor(sk, 16, 64); sk += __shfl_xor(sk, 32, 64);
    const float iq = 1.f / fmaxf(sqrtf(sq), 1e-12f), ik = 1.f / fmaxf(sqrtf(sk), 1e-12f);
    qh[mt] = bf16x8{(bf16_t)(d[0][0] * iq), (bf16_t)(d[0][1] * iq), (bf16_t)(d[0][2] * iq), (bf16_t)(d[0][3] * iq),
                    (bf16_t)(d[1][0] * iq), (bf16_t)(d[1][1] * iq), (bf16_t)(d[1][2] * iq), (bf16_t)(d[1][3] * iq)};
    const bool valid = j0 < n;  // padded slots: zero K-hat / V rows (their scores are masked to -inf anyway)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const int col = hf * 16 + g * 4;
      bf16x4 kk = {(bf16_t)(d[2 + hf][0] * ik), (bf16_t)(d[2 + hf][1] * ik), (bf16_t)(d[2 + hf][2] * ik), (bf16_t)(d[2 + hf][3] * ik)};
      bf16x4 vv = {(bf16_t)d[4 + hf][0], (bf16_t)d[4 + hf][1], (bf16_t)d[4 + hf][2], (bf16_t)d[4 + hf][3]};
      if (!valid) { kk = bf16x4{}; vv = bf16x4{}; }
      *reinterpret_cast<bf16x4*>(Kr + j0 * KP + col * 2) = kk;
      *reinterpret_cast<bf16x4*>(Vt + swin_tr_off(j0, col)) = vv;
    }
  };
  f32x4 bq[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int row = (j >> 1) * C + h * HD + (j & 1) * 16;
    bq[j] = p.bqkv ? *reinterpret_cast<const f32x4*>(p.bqkv + row + g * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  if constexpr (C <= 128) {
    // this head's 96 weight rows as MFMA A fragments (row = lane&15 of the tile, k = 32 kb + 8 g ..), all k blocks in
    // registers for the 4 token tiles: tiles q0 q1 k0 k1 v0 v1
    bf16x8 wf[6][KB];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int row = (j >> 1) * C + h * HD + (j & 1) * 16;
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) wf[j][kb] = *reinterpret_cast<const bf16x8*>(p.wqkv + (long)(row + lr) * C + kb * 32 + g * 8);
    }
    auto load_x = [&](int mt, bf16x8 (&xf)[KB]) {
      const int j0 = mt * 16 + lr;
      const int t = tok[j0 < n ? j0 : n - 1];
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) xf[kb] = *reinterpret_cast<const bf16x8*>(p.x + (long)t * C + kb * 32 + g * 8);
    };
    bf16x8 xnext[KB];
    load_x(0, xnext);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      bf16x8 xf[KB];
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) xf[kb] = xnext[kb];
      if (mt + 1 < 4) load_x(mt + 1, xnext);             // the next tile's rows are in flight during this tile's MFMAs
      f32x4 d[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        d[j] = bq[j];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) d[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][kb], xf[kb], d[j], 0, 0, 0);
      }
      finish_tile(mt, d);
    }
  } else {
    // wide stage (C = 256): the weights do not fit in registers -- k block by k block, 6 weight + 4 token fragments in
    // flight one block ahead, all 4 x 6 accumulators live
    const long trow[4] = {(long)tok[min(lr, n - 1)] * C, (long)tok[min(16 + lr, n - 1)] * C, (long)tok[min(32 + lr, n - 1)] * C,
                          (long)tok[min(48 + lr, n - 1)] * C};
    auto load_kb = [&](int kb, bf16x8 (&wf)[6], bf16x8 (&xf)[4]) {
#pragma unroll
      for (int j = 0; j < 6; ++j)
        wf[j] = *reinterpret_cast<const bf16x8*>(p.wqkv + (long)((j >> 1) * C + h * HD + (j & 1) * 16 + lr) * C + kb * 32 + g * 8);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) xf[mt] = *reinterpret_cast<const bf16x8*>(p.x + trow[mt] + kb * 32 + g * 8);
    };
    f32x4 d[4][6];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int j = 0; j < 6; ++j) d[mt][j] = bq[j];
    bf16x8 wn[6], xn[4];
    load_kb(0, wn, xn);
    for (int kb = 0; kb < KB; ++kb) {
      bf16x8 wc[6], xc[4];
#pragma unroll
      for (int j = 0; j < 6; ++j) wc[j] = wn[j];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) xc[mt] = xn[mt];
      if (kb + 1 < KB) load_kb(kb + 1, wn, xn);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int j = 0; j < 6; ++j) d[mt][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[j], xc[mt], d[mt][j], 0, 0, 0);
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) finish_tile(mt, d[mt]);
  }
  __syncthreads();
  const float scale = __expf(fminf(p.logit_scale[h], 4.6051701859880914f));
  auto fetch_bias = [&](int qt, f32x4 (&bv)[4]) {
    const int q = qt * 16 + lr;
    const float* brow = p.bias + ((long)h * n + (q < n ? q : n - 1)) * n;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const int key = t * 16 + g * 4 + r; bv[t][r] = brow[key < n ? key : n - 1]; }
  };
  const int nqt = (n + 15) / 16;
  f32x4 bnext[4];
  fetch_bias(0, bnext);
#pragma unroll
  for (int qt = 0; qt < 4; ++qt) {
    if (qt >= nqt) break;
    const int q = qt * 16 + lr;
    const int qc = q < n ? q : n - 1;
    const int tq = tok[qc], rq = reg[qc];
    f32x4 bcur[4] = {bnext[0], bnext[1], bnext[2], bnext[3]};
    if (qt + 1 < nqt) fetch_bias(qt + 1, bnext);
    f32x4 st[4];
    float m = -INFINITY;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      // K-hat fragment in kappa order: features 4g..4g+3 and 16+4g..16+4g+3 of key row t*16 + lane&15
      const char* krow = Kr + (t * 16 + lr) * KP;
      const bf16x4 lo = *reinterpret_cast<const bf16x4*>(krow + g * 8), hi = *reinterpret_cast<const bf16x4*>(krow + 32 + g * 8);
      const bf16x8 kf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      st[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qh[qt], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = t * 16 + g * 4 + r;
        // branch-free (reg[] has all 64 slots, the bias fetch is clamped): -100 twice for another shift region (HF/swinv2:433-436)
        float x = st[t][r] * scale + bcur[t][r] + (reg[key] != rq ? -200.f : 0.f);
        x = key < n ? x : -INFINITY;
        st[t][r] = x;
        m = fmaxf(m, x);
      }
    }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float e = __expf(st[t][r] - m); st[t][r] = e; sum += e; }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.f / sum;
    f32x4 o[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int sidx = 0; sidx < 2; ++sidx) {
      const f32x4 a = st[2 * sidx], bb = st[2 * sidx + 1];
      const bf16x8 pf = {(bf16_t)(a[0] * inv), (bf16_t)(a[1] * inv), (bf16_t)(a[2] * inv), (bf16_t)(a[3] * inv),
                         (bf16_t)(bb[0] * inv), (bf16_t)(bb[1] * inv), (bf16_t)(bb[2] * inv), (bf16_t)(bb[3] * inv)};
      const int q4 = lr >> 2, pp = lane & 3;
      const int r0 = sidx * 32 + 4 * g + q4;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_sw*)(Vt + swin_tr_off(r0, dt * 16 + 4 * pp)));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_sw*)(Vt + swin_tr_off(r0 + 16, dt * 16 + 4 * pp)));
        const bf16x8 vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, o[dt], 0, 0, 0);
      }
    }
    if (q < n) {
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
        *reinterpret_cast<bf16x4*>(p.ctx + (long)tq * C + h * HD + dt * 16 + g * 4) =
            bf16x4{(bf16_t)o[dt][0], (bf16_t)o[dt][1], (bf16_t)o[dt][2], (bf16_t)o[dt][3]};
    }
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
    for (int c = 0; c < HD; ++c) qh[c] = to_f32(from_f32<T>(qh[c] * qinv));  // unit q-hat (unscaled), operand-dtype rounding as in forward
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
  // 16 hidden units per block x 16 slices of the table rows (<= 12 rows per lane, kept in registers); dtable sits in LDS;
  // one pass over the heads feeds both products, partial sums meet in LDS and are added in a fixed order (bit-reproducible)
  constexpr int TI = 12;  // ntab <= 192 (windows up to 7x7 -> 169 entries)
  extern __shared__ float cpb_lds[];
  float* dt = cpb_lds;                  // [ntab, H]
  float* red = cpb_lds + ntab * H;      // [max(H,3), 16, 17]
  const int tj = threadIdx.x & 15, tt = threadIdx.x >> 4;
  const int j = blockIdx.x * 16 + tj;
  const bool ok = j < nh;
  for (int i = threadIdx.x; i < ntab * H; i += 256) dt[i] = dtable[i];
  float hv[TI], g[TI];
#pragma unroll
  for (int i = 0; i < TI; ++i) {
    const int t = tt + i * 16;
    hv[i] = (ok && t < ntab) ? hidden[(long)t * nh + j] : 0.f;
    g[i] = 0.f;
  }
  __syncthreads();
  for (int h = 0; h < H; ++h) {
    const float wv = ok ? w2[h * nh + j] : 0.f;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < TI; ++i) {
      const int t = tt + i * 16;
      const float d = t < ntab ? dt[t * H + h] : 0.f;
      s += d * hv[i];
      g[i] += d * wv;
    }
    red[(h * 16 + tt) * 17 + tj] = s;
  }
  __syncthreads();
  for (int h = tt; h < H; h += 16) {
    float a = 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) a += red[(h * 16 + u) * 17 + tj];
    if (ok) dw2[h * nh + j] += a;
  }
  __syncthreads();
  float a0 = 0.f, a1 = 0.f, ab = 0.f;
#pragma unroll
  for (int i = 0; i < TI; ++i) {
    const int t = tt + i * 16;
    if (t < ntab && hv[i] > 0.f) { a0 += g[i] * coords[t * 2]; a1 += g[i] * coords[t * 2 + 1]; ab += g[i]; }  // ReLU gate
  }
  red[(0 * 16 + tt) * 17 + tj] = a0; red[(1 * 16 + tt) * 17 + tj] = a1; red[(2 * 16 + tt) * 17 + tj] = ab;
  __syncthreads();
  if (tt == 0 && ok) {
    float x0 = 0.f, x1 = 0.f, xb = 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) { x0 += red[(0 * 16 + u) * 17 + tj]; x1 += red[(1 * 16 + u) * 17 + tj]; xb += red[(2 * 16 + u) * 17 + tj]; }
    dw0[j * 2] += x0; dw0[j * 2 + 1] += x1; db0[j] += xb;
  }
}

// ---- matrix-core backward (bf16, head dim 32, windows of <= 64 tokens) -------------------------------------------------------
// Three passes around t5_attn_bwd_mfma<32> (attn_t5_mfma.hip), which already is the softmax-attention backward on the matrix
// cores: (1) gather the window's tokens (roll + partition as index math) into window-major unit-q | unit-k | v, dO and O
// copies and keep 1/|q|, 1/|k|; (2) the T5 kernel with score = scale_h * (q-hat . k-hat) + bias[window] (+ shift mask, folded
// into a per-window table), P recomputed from the forward's LSE, dS stored for the bias gradient; (3) back to token order
// through the Jacobian of x / |x| and the logit-scale gradient.  Every pass addresses whole 64-byte head slices with 4 lanes.
struct SwinBwdWs {
  size_t g, dow, ow, dg, invn, scale, ds, biasw, dtp, total;
};
__host__ inline SwinBwdWs swin_bwd_ws(int B, int R, int w, int H, int C) {
  // window-major rows: padded windows (R % w != 0) count in full
  const size_t n = (size_t)w * w, nW = (size_t)((R + w - 1) / w) * ((R + w - 1) / w), M = (size_t)B * nW * n;
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  SwinBwdWs o;
  size_t off = 0;
  o.g = off; off += al(M * 3 * C * 2);
  o.dow = off; off += al(M * C * 2);
  o.ow = off; off += al(M * C * 2);
  o.dg = off; off += al(M * 3 * C * 2);
  o.invn = off; off += al(M * H * 2 * 4);
  o.scale = off; off += al((size_t)H * 4);
  const bool large = n > 64 || (R % w) != 0;  // streaming kernels: no dS scratch, no per-window bias+mask tensor
  o.ds = off; off += large ? 0 : al((size_t)B * nW * H * n * 64 * 2);
  o.biasw = off; off += large ? 0 : al(nW * H * n * n * 4);
  // large windows: per-workgroup partials of d(bias table), [B*nW * ceil(n/64), H, (2w-1)^2] f32
  // (sized for both forms of the table gradient: partial tables [.., ceil(n/64), H, (2w-1)^2] f32, or dS [.., H, n, roundup32(n)] bf16)
  {
    const size_t part = (size_t)B * nW * ((n + 63) / 64) * H * (size_t)(2 * w - 1) * (2 * w - 1) * 4;
    const size_t dsb = (size_t)B * nW * H * n * ((n + 31) & ~(size_t)31) * 2;
    o.dtp = off; off += large ? al(part > dsb ? part : dsb) : 0;
  }
  o.total = off;
  return o;
}

struct SwinBwdP {
  const bf16_t* qkv; const bf16_t* ctx; const bf16_t* dctx; bf16_t* dqkv;
  bf16_t* g; bf16_t* dow; bf16_t* ow; const bf16_t* dg; float* invn; float* scale; const float* logit_scale; float* dlogit_scale;
  int B, R, w, shift, H, C;
  int Rp; const float* vbias; float* dvbias;  // window padding: padded grid size, value bias (rows of padded keys) and its gradient
};
// window-major row -> source token; -1 for a padded position of the grid (HF/swinv2:645-650)
__device__ __forceinline__ int swin_token_of_row(long rw, int n, int nW, int nWr, int w, int R, int Rp, int shift) {
  const int j = (int)(rw % n);
  const int bw = (int)(rw / n);
  const int win = bw % nW, b = bw / nW;
  const int ys = (win / nWr) * w + j / w, xs = (win % nWr) * w + j % w;
  const int y = (ys + shift) % Rp, x = (xs + shift) % Rp;
  return (y < R && x < R) ? (b * R + y) * R + x : -1;
}

__global__ __launch_bounds__(256) void swin_bwd_gather_kernel(SwinBwdP p) {
  const int C = p.C, C8 = C >> 3, n = p.w * p.w, nWr = p.Rp / p.w, nW = nWr * nWr;
  const long M = (long)p.B * nW * n;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (blockIdx.x == 0 && (int)threadIdx.x < p.H) p.scale[threadIdx.x] = __expf(fminf(p.logit_scale[threadIdx.x], 4.6051701859880914f));
  if (idx >= M * C8) return;  // C8 is a multiple of 4: the 4 lanes of a head slice leave together
  const long rw = idx / C8;
  const int c8 = (int)(idx % C8), h = c8 >> 2;
  const int t = swin_token_of_row(rw, n, nW, nWr, p.w, p.R, p.Rp, p.shift);
  if (t < 0) {  // padded position: q-hat = k-hat = 0, v = value bias, no upstream gradient
    bf16x8 z = {}, vb = {};
    if (p.vbias) {
#pragma unroll
      for (int u = 0; u < 8; ++u) vb[u] = (bf16_t)p.vbias[c8 * 8 + u];
    }
    bf16_t* dst = p.g + rw * 3 * C + c8 * 8;
    *reinterpret_cast<bf16x8*>(dst) = z;
    *reinterpret_cast<bf16x8*>(dst + C) = z;
    *reinterpret_cast<bf16x8*>(dst + 2 * C) = vb;
    *reinterpret_cast<bf16x8*>(p.dow + rw * C + c8 * 8) = z;
    *reinterpret_cast<bf16x8*>(p.ow + rw * C + c8 * 8) = z;
    if ((c8 & 3) == 0) *reinterpret_cast<float2*>(p.invn + (rw * p.H + h) * 2) = make_float2(0.f, 0.f);
    return;
  }
  const bf16_t* src = p.qkv + (long)t * 3 * C + c8 * 8;
  bf16x8 q = *reinterpret_cast<const bf16x8*>(src);
  bf16x8 k = *reinterpret_cast<const bf16x8*>(src + C);
  const bf16x8 v = *reinterpret_cast<const bf16x8*>(src + 2 * C);
  const bf16x8 d = *reinterpret_cast<const bf16x8*>(p.dctx + (long)t * C + c8 * 8);
  const bf16x8 o = *reinterpret_cast<const bf16x8*>(p.ctx + (long)t * C + c8 * 8);
  float sq = 0.f, sk = 0.f;
#pragma unroll
  for (int u = 0; u < 8; ++u) { const float a = (float)q[u], b = (float)k[u]; sq += a * a; sk += b * b; }
  sq += __shfl_xor(sq, 1, 64); sk += __shfl_xor(sk, 1, 64);
  sq += __shfl_xor(sq, 2, 64); sk += __shfl_xor(sk, 2, 64);
  const float iq = 1.f / fmaxf(sqrtf(sq), 1e-12f), ik = 1.f / fmaxf(sqrtf(sk), 1e-12f);  // F.normalize eps (HF/swinv2:413)
#pragma unroll
  for (int u = 0; u < 8; ++u) { q[u] = (bf16_t)((float)q[u] * iq); k[u] = (bf16_t)((float)k[u] * ik); }  // same rounding as forward
  bf16_t* dst = p.g + rw * 3 * C + c8 * 8;
  *reinterpret_cast<bf16x8*>(dst) = q;
  *reinterpret_cast<bf16x8*>(dst + C) = k;
  *reinterpret_cast<bf16x8*>(dst + 2 * C) = v;
  *reinterpret_cast<bf16x8*>(p.dow + rw * C + c8 * 8) = d;
  *reinterpret_cast<bf16x8*>(p.ow + rw * C + c8 * 8) = o;
  if ((c8 & 3) == 0) *reinterpret_cast<float2*>(p.invn + (rw * p.H + h) * 2) = make_float2(iq, ik);
}

// forward flavour of the gather: window-major unit-q | unit-k | v only (the streaming forward of large windows)
__global__ __launch_bounds__(256) void swin_fwd_gather_kernel(SwinBwdP p) {
  const int C = p.C, C8 = C >> 3, n = p.w * p.w, nWr = p.Rp / p.w, nW = nWr * nWr;
  const long M = (long)p.B * nW * n;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (blockIdx.x == 0 && (int)threadIdx.x < p.H) p.scale[threadIdx.x] = __expf(fminf(p.logit_scale[threadIdx.x], 4.6051701859880914f));
  if (idx >= M * C8) return;
  const long rw = idx / C8;
  const int c8 = (int)(idx % C8);
  const int t = swin_token_of_row(rw, n, nW, nWr, p.w, p.R, p.Rp, p.shift);
  if (t < 0) {  // padded position: k-hat = 0, v = value bias
    bf16x8 z = {}, vb = {};
    if (p.vbias) {
#pragma unroll
      for (int u = 0; u < 8; ++u) vb[u] = (bf16_t)p.vbias[c8 * 8 + u];
    }
    bf16_t* dst = p.g + rw * 3 * C + c8 * 8;
    *reinterpret_cast<bf16x8*>(dst) = z;
    *reinterpret_cast<bf16x8*>(dst + C) = z;
    *reinterpret_cast<bf16x8*>(dst + 2 * C) = vb;
    return;
  }
  const bf16_t* src = p.qkv + (long)t * 3 * C + c8 * 8;
  bf16x8 q = *reinterpret_cast<const bf16x8*>(src);
  bf16x8 k = *reinterpret_cast<const bf16x8*>(src + C);
  const bf16x8 v = *reinterpret_cast<const bf16x8*>(src + 2 * C);
  float sq = 0.f, sk = 0.f;
#pragma unroll
  for (int u = 0; u < 8; ++u) { const float a = (float)q[u], b = (float)k[u]; sq += a * a; sk += b * b; }
  sq += __shfl_xor(sq, 1, 64); sk += __shfl_xor(sk, 1, 64);
  sq += __shfl_xor(sq, 2, 64); sk += __shfl_xor(sk, 2, 64);
  const float iq = 1.f / fmaxf(sqrtf(sq), 1e-12f), ik = 1.f / fmaxf(sqrtf(sk), 1e-12f);  // F.normalize eps (HF/swinv2:413)
#pragma unroll
  for (int u = 0; u < 8; ++u) { q[u] = (bf16_t)((float)q[u] * iq); k[u] = (bf16_t)((float)k[u] * ik); }
  bf16_t* dst = p.g + rw * 3 * C + c8 * 8;
  *reinterpret_cast<bf16x8*>(dst) = q;
  *reinterpret_cast<bf16x8*>(dst + C) = k;
  *reinterpret_cast<bf16x8*>(dst + 2 * C) = v;
}

// bias + shift mask per window: biasw[win, h, i, j] = bias[h, i, j] - 200 * (region(i) != region(j))  (HF/swinv2:433-436, twice)
__global__ __launch_bounds__(256) void swin_bias_mask_kernel(const float* __restrict__ bias, float* __restrict__ biasw, int R, int w, int shift, int H) {
  const int n = w * w, nWr = R / w, nW = nWr * nWr;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)nW * H * n * n) return;
  const int j = (int)(idx % n), i = (int)((idx / n) % n);
  const int h = (int)((idx / ((long)n * n)) % H), win = (int)(idx / ((long)n * n * H));
  const int wy = win / nWr, wx = win % nWr;
  const int ri = swin_region(wy * w + i / w, R, w, shift) * 3 + swin_region(wx * w + i % w, R, w, shift);
  const int rj = swin_region(wy * w + j / w, R, w, shift) * 3 + swin_region(wx * w + j % w, R, w, shift);
  biasw[idx] = bias[((long)h * n + i) * n + j] + (ri != rj ? -200.f : 0.f);
}

__global__ __launch_bounds__(256) void swin_bwd_scatter_kernel(SwinBwdP p) {
  __shared__ float dsc[64];
  const int C = p.C, C8 = C >> 3, n = p.w * p.w, nWr = p.Rp / p.w, nW = nWr * nWr, H = p.H;
  const long M = (long)p.B * nW * n;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (threadIdx.x < 64) dsc[threadIdx.x] = 0.f;
  __syncthreads();
  const int tchk = idx < M * C8 ? swin_token_of_row(idx / C8, n, nW, nWr, p.w, p.R, p.Rp, p.shift) : 0;
  if (idx < M * C8 && tchk < 0) {  // padded key: its d v is part of the value-bias gradient; nothing else flows back
    if (p.dvbias) {
      const bf16x8 dv = *reinterpret_cast<const bf16x8*>(p.dg + (idx / C8) * 3 * C + 2 * C + (idx % C8) * 8);
#pragma unroll
      for (int u = 0; u < 8; ++u) atomicAdd(p.dvbias + (idx % C8) * 8 + u, (float)dv[u]);
    }
  } else if (idx < M * C8) {
    const long rw = idx / C8;
    const int c8 = (int)(idx % C8), h = c8 >> 2;
    const int t = tchk;
    const bf16_t* gs = p.g + rw * 3 * C + c8 * 8;
    const bf16_t* ds = p.dg + rw * 3 * C + c8 * 8;
    const bf16x8 qh = *reinterpret_cast<const bf16x8*>(gs), kh = *reinterpret_cast<const bf16x8*>(gs + C);
    const bf16x8 dq = *reinterpret_cast<const bf16x8*>(ds), dk = *reinterpret_cast<const bf16x8*>(ds + C);
    const bf16x8 dv = *reinterpret_cast<const bf16x8*>(ds + 2 * C);
    const float2 inv = *reinterpret_cast<const float2*>(p.invn + (rw * H + h) * 2);
    const float scale = p.scale[h];
    float dotq = 0.f, dotk = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) { dotq += (float)qh[u] * (float)dq[u]; dotk += (float)kh[u] * (float)dk[u]; }
    dotq += __shfl_xor(dotq, 1, 64); dotk += __shfl_xor(dotk, 1, 64);
    dotq += __shfl_xor(dotq, 2, 64); dotk += __shfl_xor(dotk, 2, 64);
    // x-hat = x / |x|:  dx = (dx-hat - x-hat (x-hat . dx-hat)) / |x|, with dx-hat = scale * (kernel's dQ or dK)
    bf16x8 oq, ok;
    const float fq = scale * inv.x, fk = scale * inv.y;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      oq[u] = (bf16_t)(((float)dq[u] - (float)qh[u] * dotq) * fq);
      ok[u] = (bf16_t)(((float)dk[u] - (float)kh[u] * dotk) * fk);
    }
    bf16_t* dst = p.dqkv + (long)t * 3 * C + c8 * 8;
    *reinterpret_cast<bf16x8*>(dst) = oq;
    *reinterpret_cast<bf16x8*>(dst + C) = ok;
    *reinterpret_cast<bf16x8*>(dst + 2 * C) = dv;
    if ((c8 & 3) == 0 && p.dlogit_scale) atomicAdd(&dsc[h & 63], dotq);  // d scale = sum_ij dS_ij (q-hat_i . k-hat_j) = sum_i dQ_i . q-hat_i
  }
  __syncthreads();
  if ((int)threadIdx.x < H && p.dlogit_scale) {
    const float ls = p.logit_scale[threadIdx.x];
    if (ls <= 4.6051701859880914f && dsc[threadIdx.x] != 0.f) atomicAdd(p.dlogit_scale + threadIdx.x, dsc[threadIdx.x] * p.scale[threadIdx.x]);
  }
}

int t5_attn_bwd_mfma_dispatch(const klab_attn_args* a, hipStream_t s);  // attn_t5_mfma.hip
int swin_flash_dispatch(const void* g, long ldg, int C, void* otok, long ldot, float* lse, int Bt, int H, int n, const float* scale,
                        const float* btab, float* dbtab, float* dbtab_part, int w, int Rp, int Rreal, int shift, int nW, const void* ow,
                        const void* dow, void* dg, int which, hipStream_t s);
int dbias_reduce_dispatch(const void* ds_ws, float* dbias, int nbatch, int H, int Lq, int Lk, hipStream_t s);

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

namespace klab {
int swin_attn_large_dispatch(const klab_swin_attn_args* a, bool backward, hipStream_t s);  // attn_swin_large.hip
int cpb_mlp_bwd_any_launch(const float* dtable, const float* coords, const float* hidden, const float* w2, float* dw0, float* db0,
                           float* dw2, int ntab, int heads, int nhidden, hipStream_t s);
int cpb_table_launch(const float* coords, const float* w0, const float* b0, const float* w2, float* table, float* hidden, int ntab,
                     int heads, int nhidden, hipStream_t s) {
  hipLaunchKernelGGL(cpb_table_kernel, dim3(ntab), dim3(256), 0, s, coords, w0, b0, w2, table, hidden, heads, nhidden);
  KLAB_LAUNCH_CHECK();
  return KLAB_OK;
}
}  // namespace klab
// windows of more than 64 tokens, or a bias supplied as the (2w-1)^2 x H table: the tiled kernels of attn_swin_large.hip
static bool swin_use_large(const klab_swin_attn_args* a) { return a->w * a->w > 64 || (a->R % a->w) != 0 || (!a->bias && a->bias_table); }

static int swin_args_ok(const klab_swin_attn_args* a) {
  if (!a || !a->qkv || !a->ctx || (!a->bias && !a->bias_table) || !a->logit_scale) return KLAB_ERR_BADARG;
  if (a->w <= 0) return KLAB_ERR_BADARG;  // (R % w != 0: padded windows, HF/swinv2:645-650 -- the tiled / streaming kernels)
  if (a->C % a->H) return KLAB_ERR_BADARG;
  if (a->shift < 0 || a->shift >= a->w) return KLAB_ERR_BADARG;
  return KLAB_OK;
}

static bool swin_flash_ok(int dtype, int w, int H, int C);
static int swin_attn_flash(const klab_swin_attn_args* a, bool backward, hipStream_t s);

extern "C" int klab_swin_attn_fwd(const klab_swin_attn_args* a, void* stream) {
  int rc = swin_args_ok(a);
  if (rc) return rc;
  if (swin_use_large(a)) {
    if (a->bwd_ws && a->bias_table && !a->bias && a->lse && swin_flash_ok(a->dtype, a->w, a->H, a->C)) return swin_attn_flash(a, false, (hipStream_t)stream);
    return swin_attn_large_dispatch(a, false, (hipStream_t)stream);
  }
  SwinAttnP p{a->qkv, a->ctx, a->bias, a->logit_scale, a->lse, a->B, a->R, a->w, a->shift, a->H, a->C, nullptr, nullptr, nullptr, nullptr};
  const int hd = a->C / a->H;
  hipStream_t s = (hipStream_t)stream;
  if (a->dtype == KLAB_BF16) {
    if (hd == 32 && a->w * a->w <= 64 && (a->C & 7) == 0) {
      const int nW = (a->R / a->w) * (a->R / a->w);
      hipLaunchKernelGGL(swin_attn_fwd_mfma32, dim3(a->B * nW * a->H), dim3(64), 0, s, p);
      KLAB_LAUNCH_CHECK();
      return KLAB_OK;
    }
    if (hd == 32) return launch_swin_fwd<bf16_t, 32>(p, s);
    if (hd == 16) return launch_swin_fwd<bf16_t, 16>(p, s);
  } else if (a->dtype == KLAB_F32) {
    if (hd == 32) return launch_swin_fwd<float, 32>(p, s);
    if (hd == 16) return launch_swin_fwd<float, 16>(p, s);
  }
  return KLAB_ERR_UNSUPPORTED;
}

static bool swin_bwd_mfma_ok(int dtype, int w, int H, int C) { return dtype == KLAB_BF16 && C == H * 32 && w * w <= 64 && H <= 64; }
// windows of more than 64 tokens on the matrix cores: streaming kernels over window-major copies (bf16, head dim 32, bias table)
static bool swin_flash_ok(int dtype, int w, int H, int C) {
  static const bool on = [] { const char* e = getenv("KLAB_SWIN_FLASH"); return !e || atoi(e) != 0; }();
  return on && dtype == KLAB_BF16 && C == H * 32 && H <= 64 && (C & 7) == 0 && w > 0;
}

extern "C" size_t klab_swin_attn_bwd_ws_bytes(int dtype, int B, int R, int w, int H, int C) {
  if (B <= 0 || R <= 0 || w <= 0) return 0;
  const bool large = w * w > 64 || (R % w) != 0;
  if (large ? !swin_flash_ok(dtype, w, H, C) : !swin_bwd_mfma_ok(dtype, w, H, C)) return 0;
  return swin_bwd_ws(B, R, w, H, C).total;
}

static int swin_attn_flash(const klab_swin_attn_args* a, bool backward, hipStream_t s) {
  const SwinBwdWs L = swin_bwd_ws(a->B, a->R, a->w, a->H, a->C);
  if (a->bwd_ws_bytes < L.total) return KLAB_ERR_BADARG;
  char* ws = (char*)a->bwd_ws;
  const int nWr_ = (a->R + a->w - 1) / a->w;
  const int n = a->w * a->w, nW = nWr_ * nWr_, C = a->C, H = a->H;
  SwinBwdP p{(const bf16_t*)a->qkv, (const bf16_t*)a->ctx, (const bf16_t*)a->dctx, (bf16_t*)a->dqkv,
             (bf16_t*)(ws + L.g), (bf16_t*)(ws + L.dow), (bf16_t*)(ws + L.ow), (const bf16_t*)(ws + L.dg), (float*)(ws + L.invn),
             (float*)(ws + L.scale), a->logit_scale, a->dlogit_scale, a->B, a->R, a->w, a->shift, H, C,
             nWr_ * a->w, a->v_bias, a->dv_bias};
  const long work = (long)a->B * nW * n * (C / 8);
  const unsigned nb = (unsigned)((work + 255) / 256);
  if (!backward) {
    hipLaunchKernelGGL(swin_fwd_gather_kernel, dim3(nb), dim3(256), 0, s, p);
    KLAB_LAUNCH_CHECK();
    return swin_flash_dispatch(ws + L.g, 3L * C, C, a->ctx, C, a->lse, a->B * nW, H, n, (const float*)(ws + L.scale), a->bias_table, nullptr,
                               nullptr, a->w, nWr_ * a->w, a->R, a->shift, nW, nullptr, nullptr, nullptr, 0, s);
  }
  hipLaunchKernelGGL(swin_bwd_gather_kernel, dim3(nb), dim3(256), 0, s, p);
  KLAB_LAUNCH_CHECK();
  for (int which = 1; which <= 2; ++which) {
    const int rc = swin_flash_dispatch(ws + L.g, 3L * C, C, nullptr, 0, a->lse, a->B * nW, H, n, (const float*)(ws + L.scale), a->bias_table,
                                       a->dbias_table, (float*)(ws + L.dtp), a->w, nWr_ * a->w, a->R, a->shift, nW, ws + L.ow, ws + L.dow,
                                       ws + L.dg, which, s);
    if (rc) return rc;
  }
  hipLaunchKernelGGL(swin_bwd_scatter_kernel, dim3(nb), dim3(256), 0, s, p);
  KLAB_LAUNCH_CHECK();
  return KLAB_OK;
}

static int swin_attn_bwd_mfma(const klab_swin_attn_args* a, hipStream_t s) {
  const SwinBwdWs L = swin_bwd_ws(a->B, a->R, a->w, a->H, a->C);
  if (a->bwd_ws_bytes < L.total) return KLAB_ERR_BADARG;
  char* ws = (char*)a->bwd_ws;
  const int n = a->w * a->w, nW = (a->R / a->w) * (a->R / a->w), C = a->C, H = a->H;
  SwinBwdP p{(const bf16_t*)a->qkv, (const bf16_t*)a->ctx, (const bf16_t*)a->dctx, (bf16_t*)a->dqkv,
             (bf16_t*)(ws + L.g), (bf16_t*)(ws + L.dow), (bf16_t*)(ws + L.ow), (const bf16_t*)(ws + L.dg), (float*)(ws + L.invn),
             (float*)(ws + L.scale), a->logit_scale, a->dlogit_scale, a->B, a->R, a->w, a->shift, H, C, a->R, nullptr, nullptr};
  const long work = (long)a->B * a->R * a->R * (C / 8);
  const unsigned nb = (unsigned)((work + 255) / 256);
  hipLaunchKernelGGL(swin_bwd_gather_kernel, dim3(nb), dim3(256), 0, s, p);
  KLAB_LAUNCH_CHECK();
  const float* biasw = a->bias;
  if (a->shift > 0) {
    const long tot = (long)nW * H * n * n;
    hipLaunchKernelGGL(swin_bias_mask_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, a->bias, (float*)(ws + L.biasw), a->R, a->w, a->shift, H);
    KLAB_LAUNCH_CHECK();
    biasw = (const float*)(ws + L.biasw);
  }
  klab_attn_args t;
  memset(&t, 0, sizeof(t));
  t.dtype = KLAB_BF16;
  t.q = ws + L.g; t.k = ws + L.g + (size_t)C * 2; t.v = ws + L.g + (size_t)2 * C * 2; t.ldq = t.ldk = t.ldv = 3L * C;
  t.bias = biasw; t.bias_mod = a->shift > 0 ? nW : 0; t.score_scale = (const float*)(ws + L.scale);
  t.ctx = ws + L.ow; t.ldo = C; t.lse = a->lse;
  t.B = a->B * nW; t.H = H; t.Lq = t.Lk = n; t.dk = 32;
  t.dctx = ws + L.dow; t.lddo = C;
  t.dq = ws + L.dg; t.dk_out = ws + L.dg + (size_t)C * 2; t.dv = ws + L.dg + (size_t)2 * C * 2; t.lddq = t.lddk = t.lddv = 3L * C;
  if (a->dbias) { t.ds_ws = ws + L.ds; t.ds_defer = 1; }
  int rc = t5_attn_bwd_mfma_dispatch(&t, s);
  if (rc) return rc;
  if (a->dbias) { rc = dbias_reduce_dispatch(ws + L.ds, a->dbias, a->B * nW, H, n, n, s); if (rc) return rc; }
  hipLaunchKernelGGL(swin_bwd_scatter_kernel, dim3(nb), dim3(256), 0, s, p);
  KLAB_LAUNCH_CHECK();
  return KLAB_OK;
}

extern "C" int klab_swin_attn_bwd(const klab_swin_attn_args* a, void* stream) {
  int rc = swin_args_ok(a);
  if (rc) return rc;
  if (!a->dctx || !a->dqkv || !a->lse) return KLAB_ERR_BADARG;
  if (swin_use_large(a)) {
    if (a->bwd_ws && a->bias_table && !a->bias && swin_flash_ok(a->dtype, a->w, a->H, a->C)) return swin_attn_flash(a, true, (hipStream_t)stream);
    return swin_attn_large_dispatch(a, true, (hipStream_t)stream);
  }
  SwinAttnP p{a->qkv, a->ctx, a->bias, a->logit_scale, a->lse, a->B, a->R, a->w, a->shift, a->H, a->C,
              a->dctx, a->dqkv, a->dbias, a->dlogit_scale};
  const int hd = a->C / a->H;
  hipStream_t s = (hipStream_t)stream;
  if (a->bwd_ws && swin_bwd_mfma_ok(a->dtype, a->w, a->H, a->C)) return swin_attn_bwd_mfma(a, s);
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

// dtable_zeroed != 0: the caller vouches that dtable[ntab * heads] is clear (one fill for a whole tower's blocks instead of one per call)
extern "C" int klab_swin_cpb_bias_bwd_pz(const float* dbias, const float* bias, const int* index, const float* coords, const float* hidden,
                                         const float* w0, const float* w2, float* dtable, float* dw0, float* db0, float* dw2, int ntab, int n,
                                         int heads, int nhidden, int dtable_zeroed, void* stream) {
  (void)w0;
  if (!dbias || !bias || !index || !coords || !hidden || !w2 || !dtable || !dw0 || !db0 || !dw2) return KLAB_ERR_BADARG;
  hipStream_t s = (hipStream_t)stream;
  if (!dtable_zeroed) {
    hipError_t e = hipMemsetAsync(dtable, 0, (size_t)ntab * heads * 4, s);
    if (e != hipSuccess) return (int)e;
  }
  const long tot = (long)heads * n * n;
  hipLaunchKernelGGL(cpb_dtable_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, dbias, bias, index, dtable, heads, n * n);
  KLAB_LAUNCH_CHECK();
  if (ntab > 192)  // windows above 7x7 (8x8: 225 table rows): the any-size form
    return cpb_mlp_bwd_any_launch(dtable, coords, hidden, w2, dw0, db0, dw2, ntab, heads, nhidden, s);
  const size_t cpb_lds = ((size_t)ntab * heads + (size_t)(heads > 3 ? heads : 3) * 16 * 17) * 4;
  if (cpb_lds > 64 * 1024) return KLAB_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(cpb_mlp_bwd_kernel, dim3((nhidden + 15) / 16), dim3(256), cpb_lds, s, dtable, coords, hidden, w2, dw0, db0, dw2, ntab, heads, nhidden);
  KLAB_LAUNCH_CHECK();
  return KLAB_OK;
}

extern "C" int klab_swin_cpb_bias_bwd(const float* dbias, const float* bias, const int* index, const float* coords, const float* hidden,
                                      const float* w0, const float* w2, float* dtable, float* dw0, float* db0, float* dw2, int ntab, int n,
                                      int heads, int nhidden, void* stream) {
  return klab_swin_cpb_bias_bwd_pz(dbias, bias, index, coords, hidden, w0, w2, dtable, dw0, db0, dw2, ntab, n, heads, nhidden, 0, stream);
}

extern "C" int klab_swin_qkv_attn_fused(const void* x, const void* wqkv, const float* bqkv, void* ctx, const float* bias, const float* logit_scale,
                                        int dtype, int B, int R, int w, int shift, int H, int C, void* stream) {
  if (!x || !wqkv || !ctx || !bias || !logit_scale || B <= 0 || R <= 0 || w <= 0 || H <= 0) return KLAB_ERR_BADARG;
  if (dtype != KLAB_BF16 || C != H * 32 || w * w > 64 || R % w || shift < 0 || shift >= w) return KLAB_ERR_UNSUPPORTED;
  SwinQkvP p{(const bf16_t*)x, (const bf16_t*)wqkv, bqkv, (bf16_t*)ctx, bias, logit_scale, B, R, w, shift, H};
  const int nW = (R / w) * (R / w);
  hipStream_t s = (hipStream_t)stream;
  if (C == 64) hipLaunchKernelGGL(swin_qkv_attn_fused<64>, dim3(B * nW * H), dim3(64), 0, s, p);
  else if (C == 128) hipLaunchKernelGGL(swin_qkv_attn_fused<128>, dim3(B * nW * H), dim3(64), 0, s, p);
  else if (C == 256) hipLaunchKernelGGL(swin_qkv_attn_fused<256>, dim3(B * nW * H), dim3(64), 0, s, p);
  else return KLAB_ERR_UNSUPPORTED;
  KLAB_LAUNCH_CHECK();
  return KLAB_OK;
}
