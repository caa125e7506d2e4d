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
#include <map>
#include <mutex>

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
  int splits;
};

constexpr int ROWB = 144;  // K-major LDS row pitch in bytes: 128 B of K + one 16-B pad (conflict-free b128 reads)

// ---- LDS image of an m-major bf16 operand tile (contraction dim is the SLOW global dim) ------------
// The tile is kept exactly as it is loaded -- rows = 64 k's, columns = ROWS m's (16-B chunks of 8 m go
// in with one ds_write_b128) -- and the MFMA fragments come out through ds_read_b64_tr_b16, the gfx950
// hardware transpose read: per 16-lane group a 4(k) x 16(m) block is delivered so that lane i holds
// 4 consecutive k of column m0+i, i.e. half of the 16x16x32 A/B fragment.  Bank-conflict-free placement:
// row pitch PD dwords with PD/8 odd (k&7 -> eight distinct 8-dword slots) and every group of 8 k-rows
// displaced by 32 more dwords, so the two groups a 32-lane half touches (k-rows 8g.. and 8g+8..) split
// the 64 banks between them.
template <int ROWS> struct TrLayout {
  static constexpr int PD = (ROWS == 128) ? 72 : (ROWS == 64 ? 40 : 24);  // dwords; >= ROWS/2, PD/8 odd
  static constexpr int PITCHB = PD * 4;
  static constexpr int GROUPB = (8 * PD + 32) * 4;  // bytes per group of 8 k-rows
  static constexpr int BYTES = 8 * GROUPB;          // 64 k-rows
};

template <typename T, int ROWS, bool KMAJOR> struct TileBytes { static constexpr int value = ROWS * ROWB; };
template <int ROWS> struct TileBytes<bf16_t, ROWS, false> { static constexpr int value = TrLayout<ROWS>::BYTES; };

// stage one operand tile (ROWS x BK) from global into registers, then into LDS
template <typename T, int ROWS, bool KMAJOR>
struct Stager {
  using V = typename Vec16<T>::type;
  static constexpr int VEC = Vec16<T>::N;
  static constexpr int BK = MmaTraits<T>::BK;
  static constexpr int NCH = ROWS * 8 / 256;             // 16-B chunks per thread
  static constexpr int KPT = ROWS * BK / (256 * VEC);    // fp32 m-major: k's per thread (== NCH)
  static constexpr bool TR = (!KMAJOR) && sizeof(T) == 2;
  V v[NCH];

  __device__ __forceinline__ void load(const T* __restrict__ base, long ld, int row0, int k0, int nrows, int kend, int tid) {
    if constexpr (KMAJOR) {
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        int ch = tid + c * 256;
        int r = ch >> 3, kc = ch & 7;
        int gr = row0 + r, gk = k0 + kc * VEC;
        // unconditional load from a clamped in-range address + select: no branch per load
        const bool ok = gr < nrows && gk < kend;
        const int grc = gr < nrows ? gr : nrows - 1, gkc = gk < kend ? gk : kend - VEC;
        V z = *reinterpret_cast<const V*>(base + (long)grc * ld + gkc);
        if (!ok) z = V{};
        v[c] = z;
      }
    } else if constexpr (TR) {
      constexpr int CPR = ROWS / 8;  // chunks per k-row
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        int ch = tid + c * 256;
        int kk = ch / CPR, m8 = ch % CPR;
        int gk = k0 + kk, gr = row0 + m8 * 8;
        const bool ok = gr < nrows && gk < kend;
        const int grc = gr < nrows ? gr : nrows - 8, gkc = gk < kend ? gk : kend - 1;
        V z = *reinterpret_cast<const V*>(base + (long)gkc * ld + grc);
        if (!ok) z = V{};
        v[c] = z;
      }
    } else {
      constexpr int RG = ROWS / VEC;  // row groups
      int rg = tid % RG, kg = tid / RG;
      int gr = row0 + rg * VEC;
#pragma unroll
      for (int c = 0; c < KPT; ++c) {
        int gk = k0 + kg * KPT + c;
        const bool ok = gr < nrows && gk < kend;
        const int grc = gr < nrows ? gr : nrows - VEC, gkc = gk < kend ? gk : kend - 1;
        V z = *reinterpret_cast<const V*>(base + (long)gkc * ld + grc);
        if (!ok) z = V{};
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
    } else if constexpr (TR) {
      constexpr int CPR = ROWS / 8;
      using TL = TrLayout<ROWS>;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        int ch = tid + c * 256;
        int kk = ch / CPR, m8 = ch % CPR;
        *reinterpret_cast<V*>(lds + (kk >> 3) * TL::GROUPB + (kk & 7) * TL::PITCHB + m8 * 16) = v[c];
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

// bf16 fragment (8 consecutive k of one row) for k-step ks of 32, row r0 + (lane & 15)
template <int ROWS, bool KMAJOR>
__device__ __forceinline__ bf16x8 load_frag_bf16(const char* tile, int r0, int ks, int lane) {
  if constexpr (KMAJOR) {
    return *reinterpret_cast<const bf16x8*>(tile + (r0 + (lane & 15)) * ROWB + ks * 64 + (lane >> 4) * 16);
  } else {
    using TL = TrLayout<ROWS>;
    typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
    const int q = (lane & 15) >> 2, pp = lane & 3;
    const char* base = tile + (ks * 4 + (lane >> 4)) * TL::GROUPB + q * TL::PITCHB + (r0 + 4 * pp) * 2;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(base));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(base + 4 * TL::PITCHB));
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  }
}


// Workgroup -> output tile.  Blocks are dealt round-robin over the 8 XCDs (b and b+8 share an L2), so the grid
// is re-linearised to give every XCD one contiguous chunk of the tile sequence (bijective for any grid size),
// and the sequence runs fastest along the dimension whose operand is SMALLER: that operand stays resident in
// the XCD's 4 MiB L2 while the other one streams through once.  Speed only, never correctness.
__device__ __forceinline__ void tile_of_block(const GemmP& p, int BM, int BN, int bid, int& bm0, int& bn0) {
  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
  const int nwg = tiles_m * tiles_n;
  const int xcd = bid & 7, idx = bid >> 3;
  const int q = nwg >> 3, r = nwg & 7;
  const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  if ((long)p.M <= (long)p.N) {  // A (M x K) is the smaller operand: m fastest
    bm0 = (lin % tiles_m) * BM; bn0 = (lin / tiles_m) * BN;
  } else {
    bn0 = (lin % tiles_n) * BN; bm0 = (lin / tiles_n) * BM;
  }
}

// ---- epilogue shared by all tile kernels -----------------------------------------------------------
// The element-wise tail (bias, act, aux, dropout, residual) is applied in registers, the finished tile is
// parked in LDS in the OUTPUT dtype, and the workgroup then streams it out as whole rows: 16 B per lane,
// consecutive lanes on consecutive addresses.  (Storing straight from the MFMA layout wrote 32-B pieces of
// 16 different rows per instruction: partial-line writes were the bound of every short-K GEMM.)
// Caller guarantees that all waves are past their last LDS read of the main loop (a barrier).
template <typename T, int BM, int BN, int MI, int NI>
__device__ __forceinline__ void staged_epilogue(const GemmP& p, f32x4 (&acc)[MI][NI], float alpha, char* smem, int bm0, int bn0, int wm,
                                                int wn, int tid, int lane) {
  const DropCtx dc = make_drop(p.seed, p.tag, p.drop_p);
  const bool f32out = p.c_f32 || sizeof(T) == 4;
  const int pitchB = f32out ? (BN + 4) * 4 : (BN + 8) * 2;  // bytes per LDS row (16-B pad)
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int ml = wm + i * 16 + (lane & 15);
    const int m = bm0 + ml;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int nl = wn + j * 16 + (lane >> 4) * 4;
      const int n0 = bn0 + nl;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float x = acc[i][j][r] * alpha;
        const int n = n0 + r;
        if (m < p.M && n < p.N) {
          if (p.bias) x += p.bias[n];
          if (p.act == KLAB_ACT_RELU) x = fmaxf(x, 0.f);
          else if (p.act == KLAB_ACT_GELU) x = gelu_erf(x);
          if (p.aux) {
            const float a = to_f32(reinterpret_cast<const T*>(p.aux)[(long)m * p.ldaux + n]);
            if (p.aux_mode == KLAB_AUX_NONZERO) x = (a != 0.f) ? x * p.aux_scale : 0.f;
            else if (p.aux_mode == KLAB_AUX_DGELU) x *= gelu_erf_grad(a);
          }
          x *= drop_mult(dc, (uint64_t)m * (uint64_t)p.N + (uint64_t)n);
          if (p.residual) {
            x += p.r_f32 ? reinterpret_cast<const float*>(p.residual)[(long)m * p.ldr + n]
                         : to_f32(reinterpret_cast<const T*>(p.residual)[(long)m * p.ldr + n]);
          }
        }
        v[r] = x;
      }
      char* dst = smem + ml * pitchB;
      if (f32out) *reinterpret_cast<f32x4*>(dst + nl * 4) = f32x4{v[0], v[1], v[2], v[3]};
      else *reinterpret_cast<bf16x4*>(dst + nl * 2) = bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
    }
  }
  __syncthreads();
  const int esz = f32out ? 4 : 2;
  const int epc = 16 / esz;          // elements per 16-B chunk
  const int cpr = BN / epc;          // chunks per tile row
  const bool vec_ok = ((p.ldc * esz) & 15) == 0 && (p.N % epc) == 0;
  char* Cb = reinterpret_cast<char*>(p.C);
  for (int ch = tid; ch < BM * cpr; ch += 256) {
    const int row = ch / cpr, cc = ch % cpr;
    const int m = bm0 + row, n = bn0 + cc * epc;
    if (m >= p.M || n >= p.N) continue;
    const char* src = smem + row * pitchB + cc * 16;
    char* dst = Cb + ((long)m * p.ldc + n) * esz;
    if (vec_ok) {
      f32x4 val = *reinterpret_cast<const f32x4*>(src);
      if (p.accumulate) {
        if (f32out) {
          const f32x4 old = *reinterpret_cast<const f32x4*>(dst);
          val[0] += old[0]; val[1] += old[1]; val[2] += old[2]; val[3] += old[3];
        } else {
          bf16x8 nv = *reinterpret_cast<const bf16x8*>(src);
          const bf16x8 old = *reinterpret_cast<const bf16x8*>(dst);
#pragma unroll
          for (int u = 0; u < 8; ++u) nv[u] = (bf16_t)((float)nv[u] + (float)old[u]);
          *reinterpret_cast<bf16x8*>(dst) = nv;
          continue;
        }
      }
      *reinterpret_cast<f32x4*>(dst) = val;
    } else {
      const int nv = (p.N - n) < epc ? (p.N - n) : epc;
      for (int u = 0; u < nv; ++u) {
        if (f32out) {
          float x = reinterpret_cast<const float*>(src)[u];
          float* d = reinterpret_cast<float*>(dst) + u;
          *d = p.accumulate ? x + *d : x;
        } else {
          float x = (float)reinterpret_cast<const bf16_t*>(src)[u];
          bf16_t* d = reinterpret_cast<bf16_t*>(dst) + u;
          *d = (bf16_t)(p.accumulate ? x + (float)*d : x);
        }
      }
    }
  }
}
template <int BM, int BN> constexpr int epilogue_lds_bytes(bool f32out) { return f32out ? BM * (BN + 4) * 4 : BM * (BN + 8) * 2; }

// ---- NT bf16 fast path: asynchronous global->LDS ring --------------------------------------------
// Both operands K-major (every forward Linear, the LM head).  The register-staged loop above exposes one
// full memory latency (~1 us under load) per k-tile; here S = 4 stages of BK = 32 live in LDS and are
// filled by global_load_lds_dwordx4 (LDS-DMA: no VGPR staging), three k-tiles in flight behind a COUNTED
// s_waitcnt vmcnt and one raw s_barrier per k-tile.  The LDS image is linear per wave-instruction (16 rows
// x 64 B = 1 KiB, as LDS-DMA requires); bank conflicts of the ds_read_b128 fragment reads are removed by an
// XOR swizzle applied on the SOURCE address: 16-B chunk c of tile row r sits at position c ^ (2*((r>>2)&1)).
template <int BM, int BN>
__device__ __forceinline__ void gemm_nt_glds_body(const GemmP& p) {
  typedef bf16_t T;
  constexpr int BK = 32, S = 4;
  constexpr int WTM = BM / 2, WTN = BN / 2, MI = WTM / 16, NI = WTN / 16;
  constexpr int ABYTES = BM * 64, BBYTES = BN * 64, STAGE = ABYTES + BBYTES;
  constexpr int LA = BM / 64, LB = BN / 64, LPS = LA + LB;  // LDS-DMA instructions per wave per stage
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = (wave >> 1) * WTM, wn = (wave & 1) * WTN;
  int bm0, bn0;
  tile_of_block(p, BM, BN, blockIdx.x, bm0, bn0);
  const T* A = reinterpret_cast<const T*>(p.A);
  const T* B = reinterpret_cast<const T*>(p.B);
  const int nt = p.K / BK;

  // per-lane source rows (clamped: rows past M / N are never stored) and swizzled chunk
  const int lrow = lane >> 2, lpos = lane & 3;
  const int lchunk = lpos ^ (((lrow >> 2) & 1) << 1);
  const T* asrc[LA];
  const T* bsrc[LB];
#pragma unroll
  for (int i = 0; i < LA; ++i) {
    int r = bm0 + (wave * LA + i) * 16 + lrow;
    r = r < p.M ? r : p.M - 1;
    asrc[i] = A + (long)r * p.lda + lchunk * 8;
  }
#pragma unroll
  for (int i = 0; i < LB; ++i) {
    int r = bn0 + (wave * LB + i) * 16 + lrow;
    r = r < p.N ? r : p.N - 1;
    bsrc[i] = B + (long)r * p.ldb + lchunk * 8;
  }
  auto issue = [&](int kt, int stage) {
    char* sa = smem + stage * STAGE;
#pragma unroll
    for (int i = 0; i < LA; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc[i] + kt * BK),
                                       (__attribute__((address_space(3))) void*)(sa + (wave * LA + i) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < LB; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bsrc[i] + kt * BK),
                                       (__attribute__((address_space(3))) void*)(sa + ABYTES + (wave * LB + i) * 1024), 16, 0, 0);
  };

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int t = 0; t < S - 1; ++t)
    if (t < nt) issue(t, t);

  const int fr = lane & 15, fc = lane >> 4;
  const int foff = fr * 64 + ((fc ^ (((fr >> 2) & 1) << 1)) * 16);
  for (int t = 0; t < nt; ++t) {
    const int ahead = nt - 1 - t;  // k-tiles issued after tile t that may stay in flight
    if (ahead >= S - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 2) * LPS) : "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // tile t visible to every wave; every wave is done reading tile t-1's stage
    if (t + S - 1 < nt) issue(t + S - 1, (t + S - 1) % S);
    const char* ta = smem + (t % S) * STAGE;
    const char* tb = ta + ABYTES;
    bf16x8 af[MI], bfr[NI];
#pragma unroll
    for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const bf16x8*>(ta + (wm + i * 16) * 64 + foff);
#pragma unroll
    for (int j = 0; j < NI; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(tb + (wn + j * 16) * 64 + foff);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
  }
  __syncthreads();  // all LDS-DMA retired (vmcnt(0) above) and all fragment reads done: LDS is free for the epilogue
  float alpha = p.alpha;
  if (p.alpha_dev) alpha *= p.alpha_dev[0];
  staged_epilogue<T, BM, BN, MI, NI>(p, acc, alpha, smem, bm0, bn0, wm, wn, tid, lane);
}
template <int BM, int BN>
__global__ __launch_bounds__(256) void gemm_nt_glds_kernel(GemmP p) { gemm_nt_glds_body<BM, BN>(p); }

// One workgroup = 4 waves (2x2) computing a BM x BN tile over k-tiles [kt0, kt1).
// ATOMIC: split-K partial sums are added to a pre-zeroed / accumulating f32 C with float atomics; the
// MFMA operands are then NOT swapped so that each atomic wave-instruction covers 16 consecutive n (64 B).
template <typename T, int BM, int BN, bool AK, bool BKM, bool ATOMIC>
__device__ __forceinline__ void gemm_body(const GemmP& p) {
  constexpr int BK = MmaTraits<T>::BK;
  constexpr int WTM = BM / 2, WTN = BN / 2;  // wave tile (2x2 waves)
  constexpr int MI = WTM / 16, NI = WTN / 16;
  constexpr int ABYTES = TileBytes<T, BM, AK>::value, BBYTES = TileBytes<T, BN, BKM>::value;
  constexpr int STAGE = ABYTES + BBYTES;  // one pipeline stage: A tile then B tile
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave >> 1) * WTM, wn = (wave & 1) * WTN;
  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
  const int tiles = tiles_m * tiles_n;
  const int tile = blockIdx.x % tiles, split = blockIdx.x / tiles;
  int bm0, bn0;
  tile_of_block(p, BM, BN, tile, bm0, bn0);

  const T* A = reinterpret_cast<const T*>(p.A);
  const T* B = reinterpret_cast<const T*>(p.B);
  Stager<T, BM, AK> sa;
  Stager<T, BN, BKM> sb;

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nt_all = (p.K + BK - 1) / BK;
  const int per = (nt_all + p.splits - 1) / p.splits;
  const int kt0 = split * per, kt1 = (kt0 + per < nt_all) ? kt0 + per : nt_all;
  const int nt = kt1 - kt0;
  if (nt <= 0) return;
  sa.load(A, p.lda, bm0, kt0 * BK, p.M, p.K, tid);
  sb.load(B, p.ldb, bn0, kt0 * BK, p.N, p.K, tid);
  sa.store(smem, tid);
  sb.store(smem + ABYTES, tid);
  __syncthreads();

  for (int t = 0; t < nt; ++t) {
    const int cur = t & 1;
    if (t + 1 < nt) {
      sa.load(A, p.lda, bm0, (kt0 + t + 1) * BK, p.M, p.K, tid);
      sb.load(B, p.ldb, bn0, (kt0 + t + 1) * BK, p.N, p.K, tid);
    }
    const char* ta = smem + cur * STAGE;
    const char* tb = ta + ABYTES;
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int ks = 0; ks < BK / 32; ++ks) {
        bf16x8 af[MI], bfr[NI];
#pragma unroll
        for (int i = 0; i < MI; ++i) af[i] = load_frag_bf16<BM, AK>(ta, wm + i * 16, ks, lane);
#pragma unroll
        for (int j = 0; j < NI; ++j) bfr[j] = load_frag_bf16<BN, BKM>(tb, wn + j * 16, ks, lane);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j) {
            if constexpr (ATOMIC) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
            else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
          }
      }
    } else {
      const char* la = ta + (wm + (lane & 15)) * ROWB;
      const char* lb = tb + (wn + (lane & 15)) * ROWB;
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
          for (int j = 0; j < NI; ++j) {
            if constexpr (ATOMIC) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bfr[j], acc[i][j], 0, 0, 0);
            else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bfr[j], af[i], acc[i][j], 0, 0, 0);
          }
      }
    }
    if (t + 1 < nt) {
      sa.store(smem + (cur ^ 1) * STAGE, tid);
      sb.store(smem + (cur ^ 1) * STAGE + ABYTES, tid);
    }
    __syncthreads();
  }

  float alpha = p.alpha;
  if (p.alpha_dev) alpha *= p.alpha_dev[0];
  if constexpr (ATOMIC) {
    // lane owns n = ... + (lane&15), m = ... + (lane>>4)*4 + r
    float* Cf = reinterpret_cast<float*>(p.C);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int n = bn0 + wn + j * 16 + (lane & 15);
        if (n >= p.N) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = bm0 + wm + i * 16 + (lane >> 4) * 4 + r;
          if (m < p.M) atomicAdd(Cf + (long)m * p.ldc + n, acc[i][j][r] * alpha);
        }
      }
  } else {
    staged_epilogue<T, BM, BN, MI, NI>(p, acc, alpha, smem, bm0, bn0, wm, wn, tid, lane);
  }
}

template <typename T, int BM, int BN, bool AK, bool BKM, bool ATOMIC>
__global__ __launch_bounds__(256) void gemm_kernel(GemmP p) { gemm_body<T, BM, BN, AK, BKM, ATOMIC>(p); }
// the LM-head logits GEMM under its own symbol, so that profiles and the in-process probe
// (klab_engine_probe_*) can be matched kernel for kernel
template <typename T>
__global__ __launch_bounds__(256) void klab_lmhead_gemm(GemmP p) {
  if constexpr (sizeof(T) == 2) gemm_nt_glds_body<128, 128>(p);
  else gemm_body<T, 128, 128, true, true, false>(p);
}

template <typename K>
static int launch_kernel(K kern, const GemmP& p, int BM, int BN, size_t lds, hipStream_t s) {
  static std::mutex mu;
  static std::map<const void*, size_t> attr;
  {
    std::lock_guard<std::mutex> g(mu);
    const void* key = reinterpret_cast<const void*>(kern);
    auto it = attr.find(key);
    if (it == attr.end() || it->second < lds) {
      hipError_t e = hipFuncSetAttribute(key, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return (int)e;
      attr[key] = lds;
    }
  }
  const long tm = (p.M + BM - 1) / BM, tn = (p.N + BN - 1) / BN;
  hipLaunchKernelGGL(kern, dim3((unsigned)(tm * tn * p.splits)), dim3(256), lds, s, p);
  KLAB_LAUNCH_CHECK();
  return KLAB_OK;
}

template <typename T, int BM, int BN, bool AK, bool BKM>
static int launch_gemm(const GemmP& p, bool atomic, hipStream_t s) {
  size_t lds = 2 * (size_t)(TileBytes<T, BM, AK>::value + TileBytes<T, BN, BKM>::value);
  const size_t epi = (size_t)epilogue_lds_bytes<BM, BN>(p.c_f32 || sizeof(T) == 4);
  if (!atomic && epi > lds) lds = epi;
  if (atomic) return launch_kernel(gemm_kernel<T, BM, BN, AK, BKM, true>, p, BM, BN, lds, s);
  return launch_kernel(gemm_kernel<T, BM, BN, AK, BKM, false>, p, BM, BN, lds, s);
}

template <int BM, int BN>
static int launch_nt_glds(const GemmP& p, hipStream_t s) {
  const int nt = p.K / 32;
  size_t lds = (size_t)(nt < 4 ? nt : 4) * (BM + BN) * 64;  // short K: fewer stages => more workgroups per CU
  const size_t epi = (size_t)epilogue_lds_bytes<BM, BN>(p.c_f32);
  if (epi > lds) lds = epi;
  return launch_kernel(gemm_nt_glds_kernel<BM, BN>, p, BM, BN, lds, s);
}

template <typename T, int BM, int BN>
static int dispatch_layout(const GemmP& p, bool atomic, hipStream_t s) {
  if constexpr (sizeof(T) == 2) {
    if (p.a_kmajor && p.b_kmajor && !atomic && p.splits == 1 && (p.K % 32) == 0 && p.K >= 32) return launch_nt_glds<BM, BN>(p, s);
  }
  if (p.a_kmajor && p.b_kmajor) return launch_gemm<T, BM, BN, true, true>(p, atomic, s);
  if (p.a_kmajor && !p.b_kmajor) return launch_gemm<T, BM, BN, true, false>(p, atomic, s);
  if (!p.a_kmajor && p.b_kmajor) return launch_gemm<T, BM, BN, false, true>(p, atomic, s);
  return launch_gemm<T, BM, BN, false, false>(p, atomic, s);
}

// tile / split-K choice: the largest tile that still gives about one workgroup per CU; when even the
// smallest does not and the caller allows atomic accumulation, split K until the chip is covered.
template <typename T>
static int dispatch_tile(GemmP& p, bool atomic_ok, hipStream_t s) {
  auto tiles = [&](int bm, int bn) { return (long)((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn); };
  const int BK = MmaTraits<T>::BK;
  const int nt = (p.K + BK - 1) / BK;
  p.splits = 1;
  if (tiles(128, 128) >= 240) return dispatch_layout<T, 128, 128>(p, false, s);
  if (tiles(128, 64) >= 240) return dispatch_layout<T, 128, 64>(p, false, s);
  if (atomic_ok && nt >= 16) {
    const bool big = p.M >= 128 && p.N >= 64;
    const long t = big ? tiles(128, 64) : tiles(64, 64);
    long sp = (256 + t - 1) / t;
    if (sp > nt / 8) sp = nt / 8;   // at least 8 k-tiles (512 k) per split
    if (sp > 16) sp = 16;
    if (sp >= 2) {
      p.splits = (int)sp;
      return big ? dispatch_layout<T, 128, 64>(p, true, s) : dispatch_layout<T, 64, 64>(p, true, s);
    }
  }
  return dispatch_layout<T, 64, 64>(p, false, s);
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
  if (a->dtype != KLAB_F32 && a->dtype != KLAB_BF16) return KLAB_ERR_BADARG;
  if (a->c_dtype != KLAB_F32 && a->c_dtype != a->dtype) return KLAB_ERR_BADARG;
  GemmP p;
  p.M = a->M; p.N = a->N; p.K = a->K;
  p.A = a->A; p.lda = a->lda; p.a_kmajor = a->a_kmajor;
  p.B = a->B; p.ldb = a->ldb; p.b_kmajor = a->b_kmajor;
  p.C = a->C; p.ldc = a->ldc; p.c_f32 = (a->c_dtype == KLAB_F32); p.accumulate = a->accumulate;
  p.alpha = a->alpha; p.alpha_dev = a->alpha_dev; p.bias = a->bias; p.act = a->act;
  p.aux = a->aux; p.ldaux = a->ldaux; p.aux_mode = a->aux_mode; p.aux_scale = a->aux_scale;
  p.residual = a->residual; p.ldr = a->ldr; p.r_f32 = (a->r_dtype == KLAB_F32);
  p.drop_p = a->drop_p; p.seed = a->seed_dev; p.tag = a->drop_tag;
  p.splits = 1;
  hipStream_t s = (hipStream_t)stream;
  if (a->name_tag == 1 && a->a_kmajor && a->b_kmajor) {
    if (a->dtype == KLAB_BF16) {
      if (a->K % 32) return KLAB_ERR_UNSUPPORTED;
      size_t lds = 4 * (size_t)(128 + 128) * 64;
      const size_t epi = (size_t)epilogue_lds_bytes<128, 128>(p.c_f32);
      return launch_kernel(klab_lmhead_gemm<bf16_t>, p, 128, 128, epi > lds ? epi : lds, s);
    }
    size_t lds = 2 * (size_t)(128 + 128) * ROWB;
    const size_t epi = (size_t)epilogue_lds_bytes<128, 128>(true);
    return launch_kernel(klab_lmhead_gemm<float>, p, 128, 128, epi > lds ? epi : lds, s);
  }
  // split-K with float atomics only for a plain accumulating f32 product (the wgrad form)
  const bool atomic_ok = a->atomic_ok && p.c_f32 && a->accumulate && !a->bias && !a->act && !a->aux && !a->residual && a->drop_p == 0.f;
  if (a->dtype == KLAB_BF16) return dispatch_tile<bf16_t>(p, atomic_ok, s);
  return dispatch_tile<float>(p, atomic_ok, s);
}
