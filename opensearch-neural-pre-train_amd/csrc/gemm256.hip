// 256x256x64 "ping-pong" NT GEMM (bf16, fp32 MFMA accumulation) for the large encoder shapes.
//
// Why: with two independent 4-wave workgroups per CU (gemm.hip) the two waves sharing a SIMD are
// unsynchronised -- both may sit in their LDS-read phase (matrix pipe idle: measured 38 % MFMA
// busy) or both in their MFMA phase.  Here ONE 8-wave workgroup owns the CU and the two waves of
// every SIMD (wave w and w+4: the upper and lower 128 rows of the tile) run the same program one
// barrier apart, so that while one multiplies the other reads LDS and issues LDS-DMA:
//
//   slot:   s        s+1       s+2       s+3 ...
//   waves 0-3:  LOAD p   MFMA p    LOAD p+1  MFMA p+1
//   waves 4-7:  MFMA p-1 LOAD p    MFMA p    LOAD p+1          (every slot ends in one s_barrier)
//
// A K-tile (64 deep) is four phases, one 64x32 quadrant of the wave's 128x64 accumulator each
// (16 MFMAs); the operands of K-tile t+1 arrive by LDS-DMA in four 16 KiB chunks ordered by the
// phase that first needs them, one chunk issued per phase, and every LOAD segment ends with a
// COUNTED s_waitcnt vmcnt(4): everything but the two newest chunks has landed, two chunks stay in
// flight across the barrier (and across K-tile and output-tile boundaries -- the workgroup is
// persistent and the pipeline never drains).  RAW: a chunk is read one barrier after every issuing
// wave's covering wait; WAR: a chunk's LDS region was last read >= 3 slots before its refill is
// issued.  LDS: 2 stages x (256x64 A + 256x64 B) bf16 = 128 KiB, rows of 128 B, 16-B chunks
// XOR-swizzled with (row & 7) on the DMA source and on the read (conflict-free ds_read_b128).
#include "common.h"
#include "snx.h"

namespace {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;   // 64 KiB

__device__ __forceinline__ int xcd_remap256(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// tile rows covered by DMA chunk c (A: c = 0 -> m-half 0 of both wave groups, c = 3 -> m-half 1;
// B: c = 1 -> n-half 0 of the four wave columns, c = 2 -> n-half 1); ci = 0..15 eight-row groups
__device__ __forceinline__ int chunk_row(int c, int ci) {
  if (c == 0) return (ci < 8 ? 0 : 128) + (ci & 7) * 8;
  if (c == 3) return (ci < 8 ? 64 : 192) + (ci & 7) * 8;
  return (ci >> 2) * 64 + (c == 2 ? 32 : 0) + (ci & 3) * 8;
}

struct TilePos { int m0, n0; };

template <int C>
__device__ __forceinline__ void dma_chunk(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, int M, int N,
                                          int K, TilePos tp, int kt, char* stage, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ci = i * 8 + wave;
    const int r0 = chunk_row(C, ci);
    const int r = r0 + (lane >> 3);
    const int ch = (lane & 7) ^ (r & 7);
    const bool isA = (C == 0 || C == 3);
    int gr = (isA ? tp.m0 : tp.n0) + r;
    const int lim = isA ? M : N;
    gr = gr < lim ? gr : lim - 1;
    const bf16_t* src = (isA ? A : B) + (long)gr * K + kt * BK + ch * 8;
    char* dst = stage + (isA ? 0 : A_BYTES) + r0 * 128;
    __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(dst), 16, 0, 0);
  }
}

__device__ __forceinline__ bf16x8 frag256(const char* tile, int row, int chunk) {
  return *(const bf16x8*)(tile + row * 128 + ((chunk ^ (row & 7)) << 4));
}

#define WAIT_VM4() asm volatile("s_waitcnt vmcnt(4)" ::: "memory")
#define WAIT_VM2() asm volatile("s_waitcnt vmcnt(2)" ::: "memory")
#define WAIT_VM0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define BARRIER()                          \
  do {                                     \
    __builtin_amdgcn_sched_barrier(0);     \
    __builtin_amdgcn_s_barrier();          \
    __builtin_amdgcn_sched_barrier(0);     \
  } while (0)

}  // namespace

__global__ __launch_bounds__(512) void gemm_nt256_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                                         bf16_t* __restrict__ C, int M, int N, int K, int tiles_n,
                                                         int ntiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 2, wn = wave & 3;                  // wave group (row half) and column of the 2x4 grid
  const int nk = K / BK;
  const int G = gridDim.x;
  const int li = lane & 15, g = lane >> 4;

  auto tile_pos = [&](int j) {
    const int base = j * G;
    const int id = base + ((base + G <= ntiles) ? xcd_remap256(blockIdx.x, G) : (int)blockIdx.x);
    TilePos tp;
    tp.m0 = id < ntiles ? (id / tiles_n) * BM : -1;
    tp.n0 = id < ntiles ? (id % tiles_n) * BN : 0;
    return tp;
  };

  int j = 0;
  TilePos cur = tile_pos(0);
  if (cur.m0 < 0) return;
  // prologue: all four chunks of the first K-tile
  dma_chunk<0>(A, B, M, N, K, cur, 0, smem, wave, lane);
  dma_chunk<1>(A, B, M, N, K, cur, 0, smem, wave, lane);
  dma_chunk<2>(A, B, M, N, K, cur, 0, smem, wave, lane);
  dma_chunk<3>(A, B, M, N, K, cur, 0, smem, wave, lane);
  WAIT_VM0();
  BARRIER();
  if (wm == 1) BARRIER();                                   // stagger the lower-row wave group by one slot

  int stage_par = 0;
  f32x4 acc[8][4];
  while (true) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int jn = 0; jn < 4; ++jn) acc[i][jn] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const TilePos nxt_tile = tile_pos(j + 1);
    for (int kt = 0; kt < nk; ++kt, stage_par ^= 1) {
      const char* st = smem + stage_par * STAGE;
      char* nst = smem + (stage_par ^ 1) * STAGE;
      const char* ta = st + (wm * 128) * 128;               // this wave's 128 A rows
      const char* tb = st + A_BYTES + (wn * 64) * 128;      // this wave's 64 B rows (output columns)
      // where the next K-tile comes from (next k of this output tile, or k = 0 of the next one)
      const bool more_k = kt + 1 < nk;
      const TilePos ntp = more_k ? cur : nxt_tile;
      const int nkt = more_k ? kt + 1 : 0;
      const bool pf = ntp.m0 >= 0;
      bf16x8 a[2][4], b0[2][2], b1[2][2];

      // ---------------- phase 0: quadrant (rows 0..63, cols 0..31) ----------------
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int i = 0; i < 4; ++i) a[kk][i] = frag256(ta, i * 16 + li, kk * 4 + g);
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) b0[kk][jn] = frag256(tb, jn * 16 + li, kk * 4 + g);
      }
      if (pf) { dma_chunk<0>(A, B, M, N, K, ntp, nkt, nst, wave, lane); WAIT_VM4(); }
      else WAIT_VM2();                                      // nothing new issued: chunk 2 of this tile must land
      BARRIER();
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int jn = 0; jn < 2; ++jn)
            acc[i][jn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0[kk][jn], a[kk][i], acc[i][jn], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      BARRIER();
      // ---------------- phase 1: quadrant (rows 0..63, cols 32..63) ----------------
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) b1[kk][jn] = frag256(tb, 32 + jn * 16 + li, kk * 4 + g);
      if (pf) { dma_chunk<1>(A, B, M, N, K, ntp, nkt, nst, wave, lane); WAIT_VM4(); }
      else WAIT_VM0();
      BARRIER();
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int jn = 0; jn < 2; ++jn)
            acc[i][2 + jn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1[kk][jn], a[kk][i], acc[i][2 + jn], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      BARRIER();
      // ---------------- phase 2: quadrant (rows 64..127, cols 32..63) ----------------
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 4; ++i) a[kk][i] = frag256(ta, 64 + i * 16 + li, kk * 4 + g);
      if (pf) { dma_chunk<2>(A, B, M, N, K, ntp, nkt, nst, wave, lane); WAIT_VM4(); }
      else WAIT_VM0();
      BARRIER();
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int jn = 0; jn < 2; ++jn)
            acc[4 + i][2 + jn] =
                __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1[kk][jn], a[kk][i], acc[4 + i][2 + jn], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      BARRIER();
      // ---------------- phase 3: quadrant (rows 64..127, cols 0..31) ----------------
      if (pf) { dma_chunk<3>(A, B, M, N, K, ntp, nkt, nst, wave, lane); WAIT_VM4(); }
      else WAIT_VM0();
      BARRIER();
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int jn = 0; jn < 2; ++jn)
            acc[4 + i][jn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0[kk][jn], a[kk][i], acc[4 + i][jn], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      BARRIER();
    }
    // epilogue (transposed accumulators: lane owns 4 consecutive columns of one row)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = cur.m0 + wm * 128 + i * 16 + li;
      if (row >= M) continue;
#pragma unroll
      for (int jn = 0; jn < 4; ++jn) {
        const int col = cur.n0 + wn * 64 + jn * 16 + g * 4;
        if (col >= N) continue;
        const f32x4 v = acc[i][jn];
        *(bf16x4*)(C + (long)row * N + col) = (bf16x4){f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
      }
    }
    ++j;
    cur = nxt_tile;
    if (cur.m0 < 0) break;
  }
  if (wm == 0) BARRIER();                                   // balance the stagger barrier
}

extern "C" int snx_gemm_nt256_bf16(const void* A, const void* B, void* C, int32_t M, int32_t N, int32_t K,
                                   hipStream_t st) {
  if (!A || !B || !C) return SNX_E_ARG;
  if (M <= 0 || N <= 0 || K <= 0 || (K % 64) || (N % 4)) return SNX_E_SHAPE;
  const int tm = cdiv(M, BM), tn = cdiv(N, BN);
  const int ntiles = tm * tn;
  const int grid = ntiles < 256 ? ntiles : 256;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)gemm_nt256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
    attr = true;
  }
  hipLaunchKernelGGL(gemm_nt256_kernel, dim3(grid), dim3(512), 2 * STAGE, st, (const bf16_t*)A, (const bf16_t*)B,
                     (bf16_t*)C, M, N, K, tn, ntiles);
  SNX_CHECK_LAUNCH();
  return SNX_OK;
}
