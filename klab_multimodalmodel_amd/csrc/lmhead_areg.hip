// The tied LM head's logits GEMM (HF/t5:1044-1047: logits = (dec_out * d^-0.5) @ shared^T), A-stationary form for d_model = 512.
//
// Why another kernel.  Every tiled product of this library sits on the same wall: ~10.5 TB/s of operand bytes from L2 into LDS
// (DESIGN.md section 10.1).  A 128 x 128 x 512 tile moves 128 KiB of A and 128 KiB of B for 16.8 MFLOP, and the LM head has 8032
// such tiles: 2.06 GB in ~190 us, 0.29 of the MFMA peak since round 1.  But K is only 512: the 128 x 512 A panel of a row block is
// 128 KiB -- 256 bytes per lane of an eight-wave workgroup, i.e. 64 fragments of 8 bf16 = 128 VGPRs for a wave that owns 32 rows.
// So A is loaded ONCE per workgroup, straight from global memory into MFMA fragments, and stays there while the workgroup walks
// ~31 column tiles of the vocabulary: only B streams through LDS, HALF the bytes per FLOP of the tiled kernel, and since nothing
// but the B ring lives in LDS it can be deep (7 k-tiles of 16 KiB).
//
// Structure.  512 threads, eight waves (4 x 2; a wave owns 32 rows x 64 columns of the 128 x 128 tile: 8 accumulator tiles, A
// fragments for all 16 k-steps = 128 VGPRs).  B streams in k-tiles of 64 (16 KiB) through a 7-slot LDS-DMA ring, six k-tiles
// ahead.  One counted s_waitcnt vmcnt + one s_barrier per k-tile: behind barrier j k-tile j + 1 has landed and every wave is past
// its reads of k-tile j - 1, whose slot is refilled at once; the fragments of k-tile j + 1 are read (ping-pong register sets)
// while the MFMAs of k-tile j issue.  The stream never stops at a tile boundary: a finished tile is rounded into a 128 x 128
// bf16 image in LDS and written out behind the next barrier while the ring keeps filling.  Roles: waves 0-3 issue all the DMA
// (and own the counted waits), waves 4-7 do all the logits stores -- CDNA4 counts stores in vmcnt, so a wave that did both waited
// at every tile boundary for its stores to be acknowledged (41 of 199 us).
// A workgroup = (row block, one eighth of the column tiles); blockIdx & 7 (= the XCD) selects the eighth, so the 32 workgroups of
// an XCD walk the SAME 4 MB of the embedding table together and its L2 serves all of them from one HBM read.
// B image: rows of 128 B; 16-byte chunk c of row r at position c ^ (r & 7) (swizzle on the DMA source address, as in mm8p.hip).
#include <stdlib.h>

#include "gemm_shared.h"

namespace klab {
namespace la {

constexpr int BM = 128, BN = 128, KT = 8, BK = 64, K = KT * BK, R = 7, SLOT = BN * 128, NCHUNK = 8;
constexpr int STG_PITCH = 272;  // the finished 128 x 128 bf16 tile, rows of 256 B + 16 (bank spread), written by all waves, read by waves 4-7
constexpr int LDS_BYTES = R * SLOT + BM * STG_PITCH;
constexpr int NTHREADS = 512;
constexpr int INFL = 4 * (R - 3);  // DMA instructions (four per k-tile and issuing wave) that may stay in flight behind the k-tile waited for

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wg_barrier() {  // no LDS access may move across it
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// rows 16 f .. + 15 of a k-tile (rows of 128 B, chunk c of row r at c ^ (r & 7)), k = 32 ks + 8 g .. + 7
__device__ __forceinline__ bf16x8 bfrag(const char* slot, int f, int ks, int lane) {
  const int r = f * 16 + (lane & 15), g = lane >> 4;
  return *reinterpret_cast<const bf16x8*>(slot + r * 128 + (((4 * ks + g) ^ (r & 7)) * 16));
}

__global__ __launch_bounds__(NTHREADS) void klab_lmhead_areg_gemm(GemmP p, int rotate) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int chunk = blockIdx.x & (NCHUNK - 1), rb = blockIdx.x >> 3;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int t0 = (int)((long)chunk * tiles_n / NCHUNK), t1 = (int)((long)(chunk + 1) * tiles_n / NCHUNK);
  const int J = (t1 - t0) * KT;  // k-tiles this workgroup streams
  const bf16_t* Bp = reinterpret_cast<const bf16_t*>(p.B);

  // The B stream.  A k-tile is 128 rows x 128 B = 16 pieces of 8 rows.  Waves 0-3 issue them (four each: rows 32 wave + 8 i +
  // (lane >> 3), source chunk (lane & 7) ^ (row & 7)) and wait for them with counted vmcnt; waves 4-7 issue none and do all the
  // logits stores instead.  CDNA4 counts stores in vmcnt too: with DMA and stores in the same wave every counted wait behind a
  // tile boundary also waited for that tile's stores to be acknowledged (41 us of a 199 us launch, measured by skipping them).
  // The cursor (tile, k-tile, slot) advances by one k-tile per issue: no division in the loop.
  // The 32 row blocks of an XCD walk the same eighth of the vocabulary in lockstep (every B line is fetched once per XCD and serves
  // 32 workgroups); `rotate` starts each at a different tile instead (see the launch site).
  const bool dma_wave = wave < 4;
  const int srow = (wave & 3) * 32 + (lane >> 3), schunk = (lane & 7) ^ (lane >> 3);
  const int ntl = t1 - t0, ts = t0 + (rotate ? rb % (ntl > 0 ? ntl : 1) : 0);
  int is_tile = ts, is_kt = 0, is_slot = 0, issued = 0;
  const bf16_t* is_src;  // piece 0 of the current tile; pieces 1-3 are 8, 16, 24 rows further (N % 128 == 0: no row needs clamping)
  const long piece_stride = 8 * p.ldb;
  auto set_rows = [&]() { is_src = Bp + (long)(is_tile * BN + srow) * p.ldb + schunk * 8; };
  set_rows();
  auto issue = [&]() {
    if (!dma_wave || issued >= J) return;  // wave-uniform
    char* dst = smem + is_slot * SLOT + (wave & 3) * 4096;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(is_src + i * piece_stride + is_kt * BK),
                                       (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
    ++issued;
    is_slot = is_slot + 1 == R ? 0 : is_slot + 1;
    if (++is_kt == KT) { is_kt = 0; is_tile = is_tile + 1 == t1 ? t0 : is_tile + 1; set_rows(); }
  };

  const int wm = wave >> 1, wn = wave & 1;
  const long m0 = (long)rb * BM + wm * 32;
#pragma unroll
  for (int j = 0; j < R - 1; ++j) issue();
  bf16x8 af[2 * KT][2];
  {
    const bf16_t* Ap = reinterpret_cast<const bf16_t*>(p.A);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      long m = m0 + i * 16 + (lane & 15);
      m = m < p.M ? m : p.M - 1;
#pragma unroll
      for (int ks = 0; ks < 2 * KT; ++ks) af[ks][i] = *reinterpret_cast<const bf16x8*>(Ap + m * p.lda + ks * 32 + g * 8);
    }
  }
  // An s_waitcnt hipcc can SEE (the builtin, not inline asm): vmcnt(0), expcnt / lgkmcnt untouched.  Without it the compiler
  // cannot know that the A fragments have arrived -- the counted waits of the loop are opaque asm to it -- and puts its own
  // vmcnt(0) in front of the first MFMA of EVERY k-tile, which drains the whole B ring each step (measured: 194 us, no faster
  // than the tiled kernel, with the DMA latency fully exposed).
  __builtin_amdgcn_s_waitcnt(0x0F70);
  float alpha = p.alpha;
  if (p.alpha_dev) alpha *= p.alpha_dev[0];
  char* stg = smem + R * SLOT;
  bf16_t* Cp = reinterpret_cast<bf16_t*>(p.C);

  int prev_tile = ts;
  auto store_tile = [&](int tile) {  // waves 4-7: 2048 chunks of 16 B, a wave-instruction covers 4 rows x 256 B
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int id = it * 256 + (wave - 4) * 64 + lane, row = id >> 4, c = id & 15;
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(stg + row * STG_PITCH + c * 16);
      const long m = (long)rb * BM + row;
      const int n0 = tile * BN + c * 8;
      if (m < p.M && n0 < p.N) *reinterpret_cast<bf16x8*>(Cp + m * p.ldc + n0) = v;
    }
  };
  // fragments: k-step 0 of a k-tile is read one step ahead (ping-pong f0), k-step 1 while k-step 0's MFMAs issue (48 registers;
  // both k-steps ahead would be 64 and spill)
  bf16x8 f0[2][4], f1[4];
  wg_barrier();  // k-tiles 0 and 1 are in LDS (the s_waitcnt above + the other waves')
#pragma unroll
  for (int jn = 0; jn < 4; ++jn) f0[0][jn] = bfrag(smem, wn * 4 + jn, 0, lane);
  int j = 0, cur_slot = 0;  // slot of k-tile j
  for (int tt = 0, tile = ts; tt < ntl; ++tt, tile = tile + 1 == t1 ? t0 : tile + 1) {
    f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int jn = 0; jn < 4; ++jn) acc[i][jn] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < KT; ++kt, ++j) {
      // k-tile j + 1 has landed: k-tiles up to j + R - 2 are issued, R - 3 of them younger than it
      if (dma_wave) { if (j + R - 1 <= J) wait_vm<INFL>(); else wait_vm<0>(); }
      wg_barrier();
      issue();  // k-tile j + R - 1 into the slot of k-tile j - 1: every wave is past its reads
      if (kt == 0 && tt > 0 && !dma_wave) store_tile(prev_tile);  // the tile staged before this barrier, while waves 0-3 refill the ring
      const char* sl = smem + cur_slot * SLOT;
      cur_slot = cur_slot + 1 == R ? 0 : cur_slot + 1;
      const char* sn = smem + cur_slot * SLOT;  // (behind the last k-tile: a slot nobody needs -- harmless, keeps the step branch-free)
#pragma unroll
      for (int jn = 0; jn < 4; ++jn) f1[jn] = bfrag(sl, wn * 4 + jn, 1, lane);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < 4; ++jn)  // B as the A operand: a lane ends up with 4 consecutive columns of row 16 i + (lane & 15)
          acc[i][jn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f0[kt & 1][jn], af[2 * kt][i], acc[i][jn], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
#pragma unroll
      for (int jn = 0; jn < 4; ++jn) f0[(kt + 1) & 1][jn] = bfrag(sn, wn * 4 + jn, 0, lane);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < 4; ++jn)
          acc[i][jn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f1[jn], af[2 * kt + 1][i], acc[i][jn], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    }
    // ---- the finished 32 x 64 piece: scale, round, into the workgroup's tile image; waves 4-7 store it behind the next barrier ----
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int jn = 0; jn < 4; ++jn) {
        const f32x4 v = acc[i][jn];
        *reinterpret_cast<bf16x4*>(stg + (wm * 32 + i * 16 + (lane & 15)) * STG_PITCH + (wn * 64 + jn * 16 + g * 4) * 2) =
            bf16x4{(bf16_t)(v[0] * alpha), (bf16_t)(v[1] * alpha), (bf16_t)(v[2] * alpha), (bf16_t)(v[3] * alpha)};
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the image is complete before this wave reaches the next barrier
    prev_tile = tile;
  }
  wg_barrier();
  if (!dma_wave && ntl > 0) store_tile(prev_tile);
}

}  // namespace la

// KLAB_ERR_UNSUPPORTED: not this kernel's product (the caller continues with klab_lmhead_gemm's tiled form)
int lmhead_areg_try(const GemmP& p, hipStream_t s) {
  using namespace la;
  static const bool on = [] { const char* e = getenv("KLAB_LMHEAD_AREG"); return !e || atoi(e) != 0; }();
  if (!on) return KLAB_ERR_UNSUPPORTED;
  if (p.K != K || !p.a_kmajor || !p.b_kmajor || p.c_f32 || p.accumulate || p.bias || p.act || p.aux || p.residual || p.drop_p != 0.f)
    return KLAB_ERR_UNSUPPORTED;
  static const int min_n = [] { const char* e = getenv("KLAB_LMHEAD_MIN_N"); return e ? atoi(e) : 64 * BN; }();  // (tuning aid)
  if (p.M < 1024 || p.N < min_n || (p.N % BN) || (p.ldc & 7) || (p.lda & 7) || (p.ldb & 7)) return KLAB_ERR_UNSUPPORTED;
  const int rc = ensure_dyn_lds(reinterpret_cast<const void*>(klab_lmhead_areg_gemm), (size_t)LDS_BYTES);
  if (rc) return rc;
  const unsigned grid = (unsigned)(((p.M + BM - 1) / BM) * NCHUNK);
  // KLAB_LMHEAD_ROT=1: the row blocks of an XCD start at rotated tiles of their eighth of the vocabulary (32 regions in flight at once
  // instead of one shared stream).  Measured the same speed, but 1.03 GB per launch leave L2 instead of ~0.35 GB (the 4 MB working set
  // no longer survives next to the logits write stream), so the default is the lockstep walk.
  static const int rotate = [] { const char* e = getenv("KLAB_LMHEAD_ROT"); return e ? atoi(e) : 0; }();
  probed_launch(klab_lmhead_areg_gemm, dim3(grid), dim3(NTHREADS), (size_t)LDS_BYTES, s, p, rotate);
  KLAB_LAUNCH_CHECK();
  return KLAB_OK;
}

}  // namespace klab
