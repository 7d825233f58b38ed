// bf16 MFMA GEMM main loop for C[M,N] = A[M,K] * B[N,K]^T  (both operands K-contiguous: the
// "NT" form every Linear of the encoder reduces to; weights are stored [out,in]).
//
// Design (CDNA4): BMxBNx64 tile, WAVES_M x WAVES_N waves of 64 lanes, each wave owns a
// (BM/WAVES_M)x(BN/WAVES_N) sub-tile as MI x NI accumulators of v_mfma_f32_16x16x32_bf16.
// Operand tiles go HBM/L2 -> LDS with LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave
// instruction, no VGPR round trip), double buffered: the DMA of K-tile t+1 is in flight while
// tile t is multiplied.  LDS rows are 128 B (64 bf16 of K); the eight 16-B chunks of a row are
// XOR-swizzled with (row & 7) so that the ds_read_b128 fragment reads (16 rows x one chunk per
// lane group) are bank-conflict free.  LDS-DMA writes lane-linear, so the swizzle is applied
// to the per-lane SOURCE address and again on the read (same involution).
#pragma once
#include "common.h"
#ifndef SNX_GEMM_SETPRIO
#define SNX_GEMM_SETPRIO 0
#endif
#ifndef SNX_GEMM_SCHED
#define SNX_GEMM_SCHED 1
#endif

template <int BM, int BN, int WAVES_M, int WAVES_N>
struct GemmCore {
  static constexpr int BK = 64;
  static constexpr int NW = WAVES_M * WAVES_N;
  static constexpr int NTHREADS = NW * 64;
  static constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
  static constexpr int MI = WTM / 16, NI = WTN / 16;
  static constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
  static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
  static constexpr int LDS_BYTES = 2 * STAGE_BYTES;
  static_assert((BM / 8) % NW == 0 && (BN / 8) % NW == 0, "tile rows must split evenly over waves");
  static_assert(WTM % 16 == 0 && WTN % 16 == 0, "wave tile must be a multiple of 16");

  // Stage ROWS x 64 bf16 (rows row0.. of a K-contiguous matrix with leading dim ld, K offset k0)
  // into lds.  Rows past row_max-1 are clamped (re-read of a valid row; results discarded by
  // the caller's epilogue guard).
  template <int ROWS>
  static __device__ __forceinline__ void stage(const bf16_t* __restrict__ g, long ld, int row0, int row_max,
                                               int k0, char* lds, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < ROWS / 8 / NW; ++i) {
      const int ci = i * NW + wave;               // 8-row group written by this wave instruction
      const int r = ci * 8 + (lane >> 3);
      const int c = (lane & 7) ^ (r & 7);         // logical 16-B chunk held at physical slot lane&7
      int gr = row0 + r;
      gr = gr < row_max ? gr : row_max - 1;
      const bf16_t* src = g + (long)gr * ld + k0 + c * 8;
      __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(lds + ci * 1024), 16, 0, 0);
    }
  }

  static __device__ __forceinline__ bf16x8 frag(const char* tile, int row, int chunk) {
    return *(const bf16x8*)(tile + row * 128 + ((chunk ^ (row & 7)) << 4));
  }

  // multiply-accumulate one staged 64-deep K-tile (A tile at `cur`, B tile at cur + A_BYTES)
  template <bool TRANSPOSED>
  static __device__ __forceinline__ void compute_step(const char* cur, int wm, int wn, int lane,
                                                      f32x4 (&acc)[MI][NI]) {
    const char* ta = cur;
    const char* tb = cur + A_BYTES;
    // all 2*(MI+NI) fragment reads are issued up front; the scheduler is then told to interleave
    // them with the MFMAs (1 MFMA : 1 DS read) so the k=32..63 reads fly under the k=0..31 MFMAs
    bf16x8 a[2][MI], b[2][NI];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int chunk = kk * 4 + (lane >> 4);
#pragma unroll
      for (int i = 0; i < MI; ++i) a[kk][i] = frag(ta, wm * WTM + i * 16 + (lane & 15), chunk);
#pragma unroll
      for (int j = 0; j < NI; ++j) b[kk][j] = frag(tb, wn * WTN + j * 16 + (lane & 15), chunk);
    }
#if SNX_GEMM_SETPRIO
    __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          acc[i][j] = TRANSPOSED ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[kk][j], a[kk][i], acc[i][j], 0, 0, 0)
                                 : __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kk][i], b[kk][j], acc[i][j], 0, 0, 0);
#if SNX_GEMM_SETPRIO
    __builtin_amdgcn_s_setprio(0);
#endif
#if SNX_GEMM_SCHED
    // first (MI+NI) reads must land before the first MFMA; afterwards pair each of the remaining
    // (MI+NI) reads with an MFMA
#pragma unroll
    for (int q = 0; q < MI + NI; ++q) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
    for (int q = 0; q < MI + NI; ++q) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 2 * MI * NI - (MI + NI), 0);
#endif
  }

  // acc must be zero-initialised (or hold the running sum) by the caller.
  // TRANSPOSED=false: acc[i][j][r] = C[wm*WTM + i*16 + 4g + r][wn*WTN + j*16 + li]   (g = lane>>4, li = lane&15)
  // TRANSPOSED=true : acc[i][j][r] = C[wm*WTM + i*16 + li][wn*WTN + j*16 + 4g + r]   -- each lane owns 4
  //                   CONSECUTIVE columns of one row: 8-byte bf16 / 16-byte fp32 epilogue accesses.
  // Epilogue operands (residual stream, saved activations, RoPE table rows) are fetched by `pre` -- EXACTLY 16
  // VGPR-destination vector loads per wave when npre16 is set, none otherwise (wave-uniform).  They are issued in
  // the LAST BUT ONE K-step, behind the DMA of the last K-tile: the vm counter retires in issue order, so the wait
  // of the last K-step leaves those 16 in flight (counted vmcnt(16)) and they land under its 32 MFMAs and the
  // accumulator write-back.  The K loop is split in the source (body / last-but-one / last step) so that the 64
  // registers holding them are NOT live in the body: requested before the loop they cost the main loop 20-30 %
  // (register pressure: fragments no longer prefetched; measured with the epilogue compiled out) and every tile
  // began by waiting for 64 KB of HBM reads.
  struct NoPre { __device__ __forceinline__ void operator()() const {} };

  // `early` (RoPE: the positions of this lane's rows) runs right BEHIND the prologue DMA: hipcc waits for such
  // loads where it issues them (it shifts the positions into table offsets at once), which in front of the
  // prologue cost every tile a full memory round trip before its first DMA went out; behind it the same wait
  // coincides with the wait for K-tile 0.
  template <bool TRANSPOSED = false, typename PreFn = NoPre, typename EarlyFn = NoPre>
  static __device__ __forceinline__ void mainloop(const bf16_t* __restrict__ A, long lda, int m0, int M,
                                                  const bf16_t* __restrict__ B, long ldb, int n0, int N,
                                                  int K, char* smem, f32x4 (&acc)[MI][NI], PreFn pre = PreFn(),
                                                  bool npre16 = false, EarlyFn early = EarlyFn()) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int nk = K / BK;
    stage<BM>(A, lda, m0, M, 0, smem, wave, lane);
    stage<BN>(B, ldb, n0, N, 0, smem + A_BYTES, wave, lane);
    __builtin_amdgcn_sched_barrier(0);
    early();
    __builtin_amdgcn_sched_barrier(0);
    // raw barriers: __syncthreads() would add its own vmcnt(0) (an LDS-DMA counts as a pending LDS write) and
    // drain the loads a counted wait leaves in flight.  The LDS reads of tile kt-1 were consumed by its MFMAs
    // (the compiler waited lgkmcnt for them), so only the barrier itself is needed.
    int kt = 0;
    for (; kt + 2 < nk; ++kt) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                  // tile kt landed; everyone left tile kt-1
      __builtin_amdgcn_sched_barrier(0);
      char* cur = smem + (kt & 1) * STAGE_BYTES;
      char* nxt = smem + ((kt + 1) & 1) * STAGE_BYTES;
      stage<BM>(A, lda, m0, M, (kt + 1) * BK, nxt, wave, lane);
      stage<BN>(B, ldb, n0, N, (kt + 1) * BK, nxt + A_BYTES, wave, lane);
      compute_step<TRANSPOSED>(cur, wm, wn, lane, acc);
    }
    if (kt + 1 < nk) {                               // last but one K-step: stage the last tile, then the epilogue loads
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      char* cur = smem + (kt & 1) * STAGE_BYTES;
      char* nxt = smem + ((kt + 1) & 1) * STAGE_BYTES;
      stage<BM>(A, lda, m0, M, (kt + 1) * BK, nxt, wave, lane);
      stage<BN>(B, ldb, n0, N, (kt + 1) * BK, nxt + A_BYTES, wave, lane);
      __builtin_amdgcn_sched_barrier(0);             // keep the loads behind the DMA instructions
      pre();
      __builtin_amdgcn_sched_barrier(0);
      compute_step<TRANSPOSED>(cur, wm, wn, lane, acc);
      ++kt;
      if (npre16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {                                         // K = one tile
      pre();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    compute_step<TRANSPOSED>(smem + (kt & 1) * STAGE_BYTES, wm, wn, lane, acc);
  }

  // Variant with a mid-step barrier: once every wave holds the fragments of K-tile kt in registers, the
  // stage they came from is refilled at once with tile kt+2, so a DMA has 1.5 K-steps to land instead of 1
  // (counted vmcnt: the newest tile stays in flight across the step boundary).
  template <bool TRANSPOSED = false, typename PreFn = NoPre, typename EarlyFn = NoPre>
  static __device__ __forceinline__ void mainloop_mid(const bf16_t* __restrict__ A, long lda, int m0, int M,
                                                      const bf16_t* __restrict__ B, long ldb, int n0, int N,
                                                      int K, char* smem, f32x4 (&acc)[MI][NI], PreFn pre = PreFn(),
                                                      bool npre16 = false, EarlyFn early = EarlyFn()) {
    static_assert((BM / 8 / NW) + (BN / 8 / NW) == 8, "vmcnt(8) below = the DMA instructions of one K-tile per wave");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int nk = K / BK;
    stage<BM>(A, lda, m0, M, 0, smem, wave, lane);
    stage<BN>(B, ldb, n0, N, 0, smem + A_BYTES, wave, lane);
    if (nk > 1) {
      stage<BM>(A, lda, m0, M, BK, smem + STAGE_BYTES, wave, lane);
      stage<BN>(B, ldb, n0, N, BK, smem + STAGE_BYTES + A_BYTES, wave, lane);
    }
    __builtin_amdgcn_sched_barrier(0);
    early();
    __builtin_amdgcn_sched_barrier(0);
    if (nk < 5) { pre(); npre16 = false; }
    for (int kt = 0; kt < nk; ++kt) {
      // queue at this wait: [tile kt][tile kt+1] and, for kt = 1 and 2, the 16 epilogue loads issued behind tile 2
      if ((kt == 1 || kt == 2) && npre16) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
      else if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                  // tile kt landed for every wave
      char* cur = smem + (kt & 1) * STAGE_BYTES;
      const char* ta = cur;
      const char* tb = cur + A_BYTES;
      bf16x8 a[2][MI], b[2][NI];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int chunk = kk * 4 + (lane >> 4);
#pragma unroll
        for (int i = 0; i < MI; ++i) a[kk][i] = frag(ta, wm * WTM + i * 16 + (lane & 15), chunk);
#pragma unroll
        for (int j = 0; j < NI; ++j) b[kk][j] = frag(tb, wn * WTN + j * 16 + (lane & 15), chunk);
      }
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          acc[i][j] = TRANSPOSED ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[0][j], a[0][i], acc[i][j], 0, 0, 0)
                                 : __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][i], b[0][j], acc[i][j], 0, 0, 0);
      // first (MI+NI) reads before the first MFMA, the other (MI+NI) reads paired with MFMAs
#pragma unroll
      for (int q = 0; q < MI + NI; ++q) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
      for (int q = 0; q < MI + NI; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, MI * NI - (MI + NI), 0);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                  // every wave has its fragments: the stage is free
      if (kt + 2 < nk) {
        stage<BM>(A, lda, m0, M, (kt + 2) * BK, cur, wave, lane);
        stage<BN>(B, ldb, n0, N, (kt + 2) * BK, cur + A_BYTES, wave, lane);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (kt == 0 && nk >= 5) {
        pre();
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          acc[i][j] = TRANSPOSED ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[1][j], a[1][i], acc[i][j], 0, 0, 0)
                                 : __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1][i], b[1][j], acc[i][j], 0, 0, 0);
    }
  }

  // Accumulator element (i, j, r) of this lane is C[row][col] with
  //   row = m0 + wm*WTM + i*16 + (lane>>4)*4 + r,   col = n0 + wn*WTN + j*16 + (lane&15).
  static __device__ __forceinline__ int acc_row(int i, int r) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    return (wave / WAVES_N) * WTM + i * 16 + (lane >> 4) * 4 + r;
  }
  static __device__ __forceinline__ int acc_col(int j) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    return (wave % WAVES_N) * WTN + j * 16 + (lane & 15);
  }
};

// XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch), so give each XCD
// a contiguous run of tile ids (bijective for any grid size).  Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
  const int base = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return base + (bid >> 3);
}

// Position `pos` of the XCD-contiguous tile sequence -> (row panel, column tile) of a tm x tn tile grid.
// The sequence walks SUPER-BLOCKS of `sb_rows` row panels (one per XCD when tm % 8 == 0); inside a super-block
// it walks column groups of `cg` tiles, and inside a group row panel by row panel.  The workgroups resident
// on an XCD at one time (64) then need only cg weight tiles (kept <= ~1.8 MB so they stay in the 4 MiB L2
// beside the streaming activations and outputs) and a handful of activation panels, each shared by the cg
// tiles that run back to back.  cg = tn is the plain row-panel-major order.  Bijective for any tm, tn.
struct TileOrder { int tm, tn, sb_rows, cg; };
__device__ __forceinline__ void tile_of(const TileOrder& o, int pos, int& m, int& n) {
  const int per_sb = o.sb_rows * o.tn;
  const int sb = pos / per_sb;
  const int rows = min(o.sb_rows, o.tm - sb * o.sb_rows);
  int p = pos - sb * per_sb;
  const int g = p / (rows * o.cg);
  const int cols = min(o.cg, o.tn - g * o.cg);
  p -= g * rows * o.cg;
  const int r = p / cols;
  m = sb * o.sb_rows + r;
  n = g * o.cg + (p - r * cols);
}
