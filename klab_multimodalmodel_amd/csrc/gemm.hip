// klab_gemm: C[M,N] = epilogue(alpha * sum_k A(m,k) * B(n,k)) on gfx950 matrix cores.
//
// One LDS-tiled MFMA kernel serves every dense contraction of the path (SURVEY §2.4 K1,K5,K11-K14
// and their dgrad/wgrad): operands are addressed as logical A[M,K], B[N,K]; each may be stored
// K-major ("NT" torch Linear forward: x[M,K], W[N,K]) or row-major-over-the-other-dim ("m-major":
// element (r,k) at base[k*ld + r]), which gives the NN (dgrad) and TN (wgrad) forms without
// transposed copies -- the stager transposes 16-byte chunks in registers on the way into LDS.
//   bf16: v_mfma_f32_16x16x32_bf16, fp32 accumulate.   fp32: v_mfma_f32_16x16x4_f32 (exact f32 FMA
//   chain; used by the parity mode).  The B tile is fed as the MFMA "A" operand so that each lane's
//   four accumulator registers are four CONSECUTIVE n of one m: the epilogue stores 8/16 B per lane.
// Epilogue (all optional, in this order): *alpha(*alpha_dev) -> +bias[n] -> act (relu | erf-gelu)
//   -> *gelu'(aux) or *(aux!=0)*aux_scale (backward of gelu / of relu+dropout) -> dropout(seed,tag,p)
//   -> +residual -> (+C if accumulate) -> store as f32 or as the input dtype.
#include "common.h"
#include "klab_mm.h"

namespace klab {

template <typename T> struct MmaTraits;
template <> struct MmaTraits<bf16_t> {
  static constexpr int BK = 64;   // elements per K tile (128 B)
  static constexpr int KSTEP = 32;
};
template <> struct MmaTraits<float> {
  static constexpr int BK = 32;   // 128 B
  static constexpr int KSTEP = 4;
};

struct GemmP {
  int M, N, K;
  const void* A; long lda; int a_kmajor;
  const void* B; long ldb; int b_kmajor;
  void* C; long ldc; int c_f32; int accumulate;
  float alpha; const float* alpha_dev;
  const float* bias;
  int act;
  const void* aux; long ldaux; int aux_mode; float aux_scale;  // aux has the input dtype
  const void* residual; long ldr; int r_f32;
  float drop_p; const uint32_t* seed; uint32_t tag;
};

constexpr int ROWB = 144;  // LDS row pitch in bytes: 128 B of K + one 16-B pad (conflict-free b128 reads)

// stage one operand tile (ROWS x BK) from global into registers
template <typename T, int ROWS, bool KMAJOR>
struct Stager {
  using V = typename Vec16<T>::type;
  static constexpr int VEC = Vec16<T>::N;
  static constexpr int BK = MmaTraits<T>::BK;
  static constexpr int NCH = ROWS * 8 / 256;             // 16-B chunks per thread
  static constexpr int KPT = ROWS * BK / (256 * VEC);    // m-major: k's per thread (== NCH)
  V v[NCH];

  __device__ __forceinline__ void load(const T* __restrict__ base, long ld, int row0, int k0, int nrows, int K, int tid) {
    if constexpr (KMAJOR) {
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        int ch = tid + c * 256;
        int r = ch >> 3, kc = ch & 7;
        int gr = row0 + r, gk = k0 + kc * VEC;
        V z = {};
        if (gr < nrows && gk < K) z = *reinterpret_cast<const V*>(base + (long)gr * ld + gk);
        v[c] = z;
      }
    } else {
      constexpr int RG = ROWS / VEC;  // row groups
      int rg = tid % RG, kg = tid / RG;
      int gr = row0 + rg * VEC;
#pragma unroll
      for (int c = 0; c < KPT; ++c) {
        int gk = k0 + kg * KPT + c;
        V z = {};
        if (gr < nrows && gk < K) z = *reinterpret_cast<const V*>(base + (long)gk * ld + gr);
        v[c] = z;
      }
    }
  }
  __device__ __forceinline__ void store(char* lds, int tid) const {
    if constexpr (KMAJOR) {
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        int ch = tid + c * 256;
        int r = ch >> 3, kc = ch & 7;
        *reinterpret_cast<V*>(lds + r * ROWB + kc * 16) = v[c];
      }
    } else {
      constexpr int RG = ROWS / VEC;
      int rg = tid % RG, kg = tid / RG;
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        T* dst = reinterpret_cast<T*>(lds + (rg * VEC + i) * ROWB) + kg * KPT;
#pragma unroll
        for (int c = 0; c < KPT; ++c) dst[c] = v[c][i];
      }
    }
  }
};

// NAMETAG only gives the LM-head launch its own symbol (klab_lmhead_gemm) so that profiles and the
// in-process probe (klab_engine_probe) can be matched kernel for kernel.
template <typename T, int BM, int BN, bool AK, bool BKM>
__device__ __forceinline__ void gemm_body(const GemmP& p) {
  constexpr int BK = MmaTraits<T>::BK;
  constexpr int WTM = BM / 2, WTN = BN / 2;  // wave tile (2x2 waves)
  constexpr int MI = WTM / 16, NI = WTN / 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int STAGE = (BM + BN) * ROWB;  // one pipeline stage: A tile then B tile

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave >> 1) * WTM, wn = (wave & 1) * WTN;
  const int tiles_m = (p.M + BM - 1) / BM;
  // m-fastest block order: consecutive blocks share one B (weight) panel
  const int bm0 = (blockIdx.x % tiles_m) * BM, bn0 = (blockIdx.x / tiles_m) * BN;

  const T* A = reinterpret_cast<const T*>(p.A);
  const T* B = reinterpret_cast<const T*>(p.B);
  Stager<T, BM, AK> sa;
  Stager<T, BN, BKM> sb;

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nt = (p.K + BK - 1) / BK;
  sa.load(A, p.lda, bm0, 0, p.M, p.K, tid);
  sb.load(B, p.ldb, bn0, 0, p.N, p.K, tid);
  sa.store(smem, tid);
  sb.store(smem + BM * ROWB, tid);
  __syncthreads();

  for (int t = 0; t < nt; ++t) {
    const int cur = t & 1;
    if (t + 1 < nt) {
      sa.load(A, p.lda, bm0, (t + 1) * BK, p.M, p.K, tid);
      sb.load(B, p.ldb, bn0, (t + 1) * BK, p.N, p.K, tid);
    }
    const char* la = smem + cur * STAGE + (wm + (lane & 15)) * ROWB;
    const char* lb = smem + cur * STAGE + BM * ROWB + (wn + (lane & 15)) * ROWB;
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int ks = 0; ks < BK / 32; ++ks) {
        bf16x8 af[MI], bfr[NI];
        const int koff = ks * 64 + (lane >> 4) * 16;
#pragma unroll
        for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const bf16x8*>(la + i * 16 * ROWB + koff);
#pragma unroll
        for (int j = 0; j < NI; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(lb + j * 16 * ROWB + koff);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < BK / 4; ++ks) {
        float af[MI], bfr[NI];
        const int koff = (ks * 4 + (lane >> 4)) * 4;
#pragma unroll
        for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const float*>(la + i * 16 * ROWB + koff);
#pragma unroll
        for (int j = 0; j < NI; ++j) bfr[j] = *reinterpret_cast<const float*>(lb + j * 16 * ROWB + koff);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bfr[j], af[i], acc[i][j], 0, 0, 0);
      }
    }
    if (t + 1 < nt) {
      sa.store(smem + (cur ^ 1) * STAGE, tid);
      sb.store(smem + (cur ^ 1) * STAGE + BM * ROWB, tid);
    }
    __syncthreads();
  }

  // ---- epilogue: lane owns m = ... + (lane&15), n = ... + (lane>>4)*4 + r ----
  float alpha = p.alpha;
  if (p.alpha_dev) alpha *= p.alpha_dev[0];
  const DropCtx dc = make_drop(p.seed, p.tag, p.drop_p);
  const bool vec_ok = ((p.ldc & 3) == 0) && ((p.N & 3) == 0) && (!p.residual || (p.ldr & 3) == 0) &&
                      (!p.aux || (p.ldaux & 3) == 0);
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int m = bm0 + wm + i * 16 + (lane & 15);
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int n0 = bn0 + wn + j * 16 + (lane >> 4) * 4;
      if (n0 >= p.N) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] * alpha;
      const int nv = (p.N - n0) < 4 ? (p.N - n0) : 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (r >= nv) break;
        const int n = n0 + r;
        float x = v[r];
        if (p.bias) x += p.bias[n];
        if (p.act == KLAB_ACT_RELU) x = fmaxf(x, 0.f);
        else if (p.act == KLAB_ACT_GELU) x = gelu_erf(x);
        if (p.aux) {
          float a = to_f32(reinterpret_cast<const T*>(p.aux)[(long)m * p.ldaux + n]);
          if (p.aux_mode == KLAB_AUX_NONZERO) x = (a != 0.f) ? x * p.aux_scale : 0.f;
          else if (p.aux_mode == KLAB_AUX_DGELU) x *= gelu_erf_grad(a);
        }
        x *= drop_mult(dc, (uint64_t)m * (uint64_t)p.N + (uint64_t)n);
        if (p.residual) {
          x += p.r_f32 ? reinterpret_cast<const float*>(p.residual)[(long)m * p.ldr + n]
                       : to_f32(reinterpret_cast<const T*>(p.residual)[(long)m * p.ldr + n]);
        }
        v[r] = x;
      }
      if (p.c_f32) {
        float* c = reinterpret_cast<float*>(p.C) + (long)m * p.ldc + n0;
        if (p.accumulate) {
#pragma unroll
          for (int r = 0; r < 4; ++r) if (r < nv) v[r] += c[r];
        }
        if (vec_ok && nv == 4) *reinterpret_cast<f32x4*>(c) = f32x4{v[0], v[1], v[2], v[3]};
        else { for (int r = 0; r < nv; ++r) c[r] = v[r]; }
      } else {
        T* c = reinterpret_cast<T*>(p.C) + (long)m * p.ldc + n0;
        if (p.accumulate) {
#pragma unroll
          for (int r = 0; r < 4; ++r) if (r < nv) v[r] += to_f32(c[r]);
        }
        if constexpr (sizeof(T) == 2) {
          if (vec_ok && nv == 4) {
            bf16x4 o = {from_f32<bf16_t>(v[0]), from_f32<bf16_t>(v[1]), from_f32<bf16_t>(v[2]), from_f32<bf16_t>(v[3])};
            *reinterpret_cast<bf16x4*>(c) = o;
          } else { for (int r = 0; r < nv; ++r) c[r] = from_f32<T>(v[r]); }
        } else {
          if (vec_ok && nv == 4) *reinterpret_cast<f32x4*>(c) = f32x4{v[0], v[1], v[2], v[3]};
          else { for (int r = 0; r < nv; ++r) c[r] = v[r]; }
        }
      }
    }
  }
}

template <typename T, int BM, int BN, bool AK, bool BKM>
__global__ __launch_bounds__(256) void gemm_kernel(GemmP p) { gemm_body<T, BM, BN, AK, BKM>(p); }
template <typename T>
__global__ __launch_bounds__(256) void klab_lmhead_gemm(GemmP p) { gemm_body<T, 128, 128, true, true>(p); }

template <typename T, int BM, int BN, bool AK, bool BKM>
static int launch_gemm(const GemmP& p, hipStream_t s) {
  const size_t lds = 2 * (size_t)(BM + BN) * ROWB;
  static bool attr_set = false;
  auto kern = gemm_kernel<T, BM, BN, AK, BKM>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const int tm = (p.M + BM - 1) / BM, tn = (p.N + BN - 1) / BN;
  hipLaunchKernelGGL(kern, dim3((unsigned)(tm * tn)), dim3(256), lds, s, p);
  KLAB_LAUNCH_CHECK();
  return KLAB_OK;
}

template <typename T>
static int launch_lmhead(const GemmP& p, hipStream_t s) {
  const size_t lds = 2 * (size_t)(128 + 128) * ROWB;
  static bool attr_set = false;
  auto kern = klab_lmhead_gemm<T>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const int tm = (p.M + 127) / 128, tn = (p.N + 127) / 128;
  hipLaunchKernelGGL(kern, dim3((unsigned)(tm * tn)), dim3(256), lds, s, p);
  KLAB_LAUNCH_CHECK();
  return KLAB_OK;
}

template <typename T, int BM, int BN>
static int dispatch_layout(const GemmP& p, hipStream_t s) {
  if (p.a_kmajor && p.b_kmajor) return launch_gemm<T, BM, BN, true, true>(p, s);
  if (p.a_kmajor && !p.b_kmajor) return launch_gemm<T, BM, BN, true, false>(p, s);
  if (!p.a_kmajor && p.b_kmajor) return launch_gemm<T, BM, BN, false, true>(p, s);
  return launch_gemm<T, BM, BN, false, false>(p, s);
}

template <typename T>
static int dispatch_tile(const GemmP& p, hipStream_t s) {
  const long t128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128);
  if (t128 >= 256) return dispatch_layout<T, 128, 128>(p, s);
  return dispatch_layout<T, 64, 64>(p, s);
}

}  // namespace klab

extern "C" int klab_gemm(const klab_gemm_args* a, void* stream) {
  using namespace klab;
  if (!a || !a->A || !a->B || !a->C) return KLAB_ERR_BADARG;
  if (a->M <= 0 || a->N <= 0 || a->K <= 0) return KLAB_OK;
  const int vec = a->dtype == KLAB_BF16 ? 8 : 4;
  // 16-byte vector staging: the contiguous dimension of each operand must be a multiple of `vec`
  if (a->lda % vec || a->ldb % vec) return KLAB_ERR_UNSUPPORTED;
  if (a->a_kmajor ? (a->K % vec) : (a->M % vec)) return KLAB_ERR_UNSUPPORTED;
  if (a->b_kmajor ? (a->K % vec) : (a->N % vec)) return KLAB_ERR_UNSUPPORTED;
  if (((uintptr_t)a->A & 15) || ((uintptr_t)a->B & 15) || ((uintptr_t)a->C & 15)) return KLAB_ERR_UNSUPPORTED;
  GemmP p;
  p.M = a->M; p.N = a->N; p.K = a->K;
  p.A = a->A; p.lda = a->lda; p.a_kmajor = a->a_kmajor;
  p.B = a->B; p.ldb = a->ldb; p.b_kmajor = a->b_kmajor;
  p.C = a->C; p.ldc = a->ldc; p.c_f32 = (a->c_dtype == KLAB_F32); p.accumulate = a->accumulate;
  p.alpha = a->alpha; p.alpha_dev = a->alpha_dev; p.bias = a->bias; p.act = a->act;
  p.aux = a->aux; p.ldaux = a->ldaux; p.aux_mode = a->aux_mode; p.aux_scale = a->aux_scale;
  p.residual = a->residual; p.ldr = a->ldr; p.r_f32 = (a->r_dtype == KLAB_F32);
  p.drop_p = a->drop_p; p.seed = a->seed_dev; p.tag = a->drop_tag;
  if (a->dtype != KLAB_F32 && a->dtype != KLAB_BF16) return KLAB_ERR_BADARG;
  if (a->c_dtype != KLAB_F32 && a->c_dtype != a->dtype) return KLAB_ERR_BADARG;
  hipStream_t s = (hipStream_t)stream;
  if (a->name_tag == 1 && a->a_kmajor && a->b_kmajor)
    return a->dtype == KLAB_BF16 ? launch_lmhead<bf16_t>(p, s) : launch_lmhead<float>(p, s);
  if (a->dtype == KLAB_BF16) return dispatch_tile<bf16_t>(p, s);
  return dispatch_tile<float>(p, s);
}
