// Encoder NT GEMMs, 256x256 persistent "ping-pong" form:  C[M,N] = A[M,K] * B[N,K]^T, bf16 operands, fp32 MFMA
// accumulation, the five fused epilogues of gemm.hip (replaces the cuBLAS calls behind nn.Linear under autocast(bf16),
// transformers modeling_modernbert.py:90-91,271,300,490 via ref:src/model/splade_modern.py:69-73).
//
// Why a second form.  The 128x128 kernel (two independent 4-wave workgroups per CU) moves 65 FLOP per byte from L2
// into LDS and its two waves per SIMD are unsynchronised: measured 60 % MFMA issue inside the K loop, 0.30 of the
// bf16 peak over the encoder's shapes.  Here ONE 8-wave workgroup owns the CU (256x256x64 tile: 131 FLOP per byte),
// and the two waves of every SIMD (wave w and w + 4: the upper and lower half of the tile's rows) run the same
// program ONE BARRIER APART, so that while one multiplies the other reads LDS and issues LDS-DMA:
//
//   slot:        s         s+1        s+2        s+3 ...
//   waves 0-3:   LOAD p    MFMA p     LOAD p+1   MFMA p+1
//   waves 4-7:   MFMA p-1  LOAD p     MFMA p     LOAD p+1          (every slot ends in one s_barrier)
//
// A K-tile (64 deep) is four phases, one 64x32 quadrant of the wave's 128x64 accumulator each (16 MFMAs of
// v_mfma_f32_16x16x32_bf16); the operands of K-tile t+1 arrive by LDS-DMA in four 16 KiB chunks ordered by the phase
// that first needs them, one chunk issued per phase, and every LOAD segment ends with a COUNTED s_waitcnt vmcnt(4):
// everything but the two newest chunks has landed, two chunks stay in flight across the barrier -- and across
// K-tile and output-tile boundaries: the workgroup is persistent and the pipeline never drains.  RAW: a chunk is
// read one barrier after every issuing wave's covering wait; WAR: a chunk's LDS region was last read >= 3 slots
// before its refill is issued.  LDS: 2 stages x (256x64 A + 256x64 B) bf16 = 128 KiB, rows of 128 B, 16-B chunks
// XOR-swizzled with (row & 7) on the DMA source and on the read (conflict-free ds_read_b128), + 4 KiB per wave of
// private staging for the write-back = 160 KiB.
//
// Work split without a tail and without inter-workgroup traffic.  Every XCD owns a contiguous run of the tile
// sequence (the XCD-aware column-group order of gemm_core.h); its 32 workgroups take the run's tiles interleaved,
// whole rounds of 32 at a time, and the n mod 32 tiles left over are cut into UNITS of 64 rows x 256 columns dealt
// contiguously: at most two SHORT tiles of 64, 128 or 192 rows per workgroup (round 6: the units are dealt along the
// leftover tiles' COLUMN runs, where a short tile may cross a row-panel boundary -- one short tile per workgroup instead
// of two for three workgroups in eight; see next_tile).  A short tile of 64 u rows keeps the
// LDS image and the schedule of a whole one: wave group g's rows [32 u g, 32 u (g + 1)) of the tile land in its
// usual LDS rows (the DMA source rows are remapped, rows past 32 u re-read a valid row), the MFMAs of row tiles
// that do not exist are skipped (phases of 16, 8 or 0 MFMAs), the write-back stops after u passes.  36,864 token
// rows x 2,304 columns: 162 tiles per XCD = 5 rounds of 32 + 2 tiles = 8 units dealt to 8 workgroups (whole tiles
// only: 6 rounds for 5.06).
//
// Epilogue.  The last barrier of a tile is taken BEFORE the write-back by the leading wave group and AFTER it by the
// trailing one, so both groups write back in the same slot (exposed once, not twice).  Per pass of 32 rows a wave
// packs its accumulators to bf16 into its private 4 KiB (swizzled image of gemm_epi.h) and writes them back
// row-major, 16 B per lane = full 128-byte lines.  The stores stay in flight into the next tile: its first two
// LOAD segments wait with vmcnt(4 + stores) (the vm queue retires in order).
#include "gemm_core.h"
#include "config.h"
#include "gemm_epi.h"
#include "snx.h"

// diagnostics builds (wrong results; timing only): the K loop without its B / A fragment reads from LDS (the registers
// keep whatever they held) -- how much of a LOAD segment the LDS reads are ("LDS bandwidth of the ping-pong tile")
#ifdef SNX_NT256_NOBREAD
#define NT256_BFRAG(dst, expr) asm volatile("" : "=v"(dst))
#else
#define NT256_BFRAG(dst, expr) dst = (expr)
#endif
// diagnostics build (timing only, with SNX_NT256_NODMA): phases 0+1 and 2+3 without the two barriers between them, i.e.
// slots of 32 MFMAs instead of 16
#ifdef SNX_NT256_HALF_DIAG
#define HALF_DIAG_BARRIER() do { } while (0)
#else
#define HALF_DIAG_BARRIER() BARRIER()
#endif
#ifdef SNX_NT256_NOAREAD
#define NT256_AFRAG(dst, expr) asm volatile("" : "=v"(dst))
#else
#define NT256_AFRAG(dst, expr) dst = (expr)
#endif

namespace {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;   // 64 KiB
constexpr int BOUNCE = 4096;                          // per wave: 32 rows x 64 columns bf16
constexpr int LDS_TOTAL = 2 * STAGE + 8 * BOUNCE;     // 160 KiB
constexpr int NWG = 256;

struct Work {
  TileOrder order;                   // tile position -> (row panel, column tile), gemm_core.h
  long units;                        // 4 * tm * tn
  int dbg;                           // diagnostics: 1 = no write-back, 2 = every tile reads operand tile (0, 0),
                                     // 4 = write-back without its global stores, 8 = stores drain at once (no vmcnt slack),
                                     // 16 = no deep request at tile boundaries
  int coldeal;                       // leftover units dealt along COLUMN runs (round 6, see next_tile)
  int rev;                           // row panels of every super-block walked from the last to the first ("nt256_rev")
};

struct Tile { int m0, n0, u; };      // first row, first column, 64-row units (0: no tile)

// tile rows covered by DMA chunk c (A: c = 0 -> LDS rows [0,64) + [128,192), c = 3 -> [64,128) + [192,256);
// B: c = 1 -> n-half 0 of the four wave columns, c = 2 -> n-half 1); ci = 0..15 eight-row groups
__device__ __forceinline__ int chunk_row(int c, int ci) {
  if (c == 0) return (ci < 8 ? 0 : 128) + (ci & 7) * 8;
  if (c == 3) return (ci < 8 ? 64 : 192) + (ci & 7) * 8;
  return (ci >> 2) * 64 + (c == 2 ? 32 : 0) + (ci & 3) * 8;
}

// per-lane byte offsets of this wave's two DMA instructions per chunk, for the tile being prefetched (k offset apart)
struct Off { unsigned c[4][2]; };

__device__ __forceinline__ bf16x8 frag256(const char* tile, int row, int chunk) {
  return *(const bf16x8*)(tile + row * 128 + ((chunk ^ (row & 7)) << 4));
}

#define WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define BARRIER()                          \
  do {                                     \
    __builtin_amdgcn_sched_barrier(0);     \
    __builtin_amdgcn_s_barrier();          \
    __builtin_amdgcn_sched_barrier(0);     \
  } while (0)

template <int N>
struct IC { static constexpr int value = N; };

// store instructions of one wave's write-back of a WHOLE tile (they stay in flight into the next tile): 16 plain,
// 32 with the residual / GeGLU-backward / rotated-RoPE epilogues, 24 with the GeGLU forward

}  // namespace

// NTU ("stream_nt" bit 8, GeGLU forward only): the saved u = [a | g] leaves through non-temporal stores -- its next reader
// is the backward, tens of milliseconds away; as plain stores its 170 MB per launch displace the next GEMMs' operands
// from the L2 / Infinity Cache (y, which the next GEMM reads, stays a plain store)
template <int EPI, bool NTU = false>
__global__ __launch_bounds__(512) void gemm_nt256_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                                         int M, int N, int K, Work wk, EpiArgs e) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 2, wn = wave & 3;                  // wave group (row half) and column of the 2x4 grid
  const int nk = K / BK;
  const int li = lane & 15, g = lane >> 4;
  // This workgroup's tiles.  The tile sequence (gemm_core.h tile_of: XCD super-blocks x column groups x row panels)
  // is cut into one contiguous run per XCD (blockIdx & 7 labels the workgroups that share an L2).  Inside a run the
  // 32 workgroups of the XCD take tiles INTERLEAVED -- step k: positions 32 k + 0..31 -- so that at any time they
  // work on neighbouring tiles: the column tiles of a few row panels, whose A panels and B tiles they share through
  // the L2 (with one contiguous range per workgroup every workgroup streamed an A panel of its own and re-fetched
  // it for every column tile: 756 MB of fabric traffic per Wqkv launch for 230 MB of operands and results).  The
  // n mod 32 tiles left at the end of a run are dealt as 64-row units, contiguously: short tiles.
  const int xcd = blockIdx.x & 7, jw = blockIdx.x >> 3, per = gridDim.x >> 3;
  const long ntiles = wk.units >> 2;
  const long P0 = ntiles * xcd / 8, P1 = ntiles * (xcd + 1) / 8;
  const int nrun = (int)(P1 - P0), nround = nrun / per, nleft = nrun - nround * per;
  int kround = 0;
  long x = 4 * (P0 + (long)nround * per) + 4L * nleft * jw / per;            // leftover units of this workgroup
  const long xe = 4 * (P0 + (long)nround * per) + 4L * nleft * (jw + 1) / per;
  // Column-run dealing of the leftover units (wk.coldeal: one column group, the XCD's leftover tiles inside one
  // super-block -- then position p of the super-block is (row panel p / tn, column tile p % tn)).  The nleft leftover tiles
  // of the run, positions [pa, pb), hold in column c the CONTIGUOUS row panels first_c .. last_c; their units are listed
  // column by column and every workgroup takes a contiguous piece of that list: a piece inside one column's run is ONE
  // short tile of up to 4 units whose first row is any multiple of 64 (the tile's rows need not lie in one row panel).
  // With the units dealt in tile order (above) a piece of 2.75 units straddled a tile boundary for three workgroups in
  // eight, i.e. two short tiles = two latency-bound K loops.  Same outputs bit for bit: a tile's K loop does not depend on
  // which rows share it.  The pieces of one column run side by side (workgroups jw, jw + 1, ...), the three columns at the
  // same time: the A rows of a piece are shared through the XCD's L2 by the workgroups of the other columns.
  int y = (int)(4L * nleft * jw / per);
  const int ye = (int)(4L * nleft * (jw + 1) / per);

  // "nt256_rev": the A operand of a dX GEMM (du, dqkv: 170 MB) was written row panel by row panel, in ascending order, by a
  // kernel that streamed more bytes than the 256 MiB Infinity Cache holds: what is still cached when this kernel starts are
  // the LAST rows written.  Walked in the same ascending order, the first (evicted) rows' misses push the cached tail out
  // before it is reached -- the classic LRU wipe-out; walked from the last panel to the first, the tail is read while it is
  // still there.  Which workgroup computes which tile changes no tile's arithmetic.
  auto rev_panel = [&](int pm) __attribute__((always_inline)) {
    const int sb = pm / wk.order.sb_rows;
    const int rows = wk.order.tm - sb * wk.order.sb_rows < wk.order.sb_rows ? wk.order.tm - sb * wk.order.sb_rows : wk.order.sb_rows;
    return sb * wk.order.sb_rows + (rows - 1 - (pm - sb * wk.order.sb_rows));
  };
  auto next_tile = [&]() __attribute__((always_inline)) {                                  // u = 0: none
    Tile t;
    t.m0 = 0; t.n0 = 0; t.u = 0;
    int pm, pn;
    if (kround < nround) {
      tile_of(wk.order, (int)(P0 + (long)kround * per + jw), pm, pn);
      ++kround;
      if (wk.rev) pm = rev_panel(pm);
      t.m0 = pm * BM; t.n0 = pn * BN; t.u = 4;
      return t;
    }
    if (wk.coldeal) {
      if (y >= ye) return t;
      const int tn = wk.order.tn, per_sb = wk.order.sb_rows * tn;
      const long Pl = P0 + (long)nround * per;
      const int sb = (int)(Pl / per_sb);
      const int pa = (int)(Pl - (long)sb * per_sb), pb = pa + nleft;
      int cum = 0;
      for (int c = 0; c < tn; ++c) {
        const int first = pa > c ? (pa - c + tn - 1) / tn : 0;
        const int last = pb - 1 >= c ? (pb - 1 - c) / tn : -1;
        const int cnt = last >= first ? last - first + 1 : 0;
        if (y < cum + 4 * cnt) {
          const int o = y - cum;
          int uu = ye - y < 4 ? ye - y : 4;
          uu = 4 * cnt - o < uu ? 4 * cnt - o : uu;
          int top = sb * wk.order.sb_rows + first;
          if (wk.rev) top = rev_panel(sb * wk.order.sb_rows + last);          // the run's rows, mirrored inside the super-block
          t.m0 = top * BM + 64 * o;
          t.n0 = c * BN;
          t.u = uu;
          y += uu;
          return t;
        }
        cum += 4 * cnt;
      }
      return t;
    }
    if (x >= xe) return t;
    const int q0 = (int)(x & 3);
    const long left = xe - x;
    t.u = 4 - q0 < left ? 4 - q0 : (int)left;
    tile_of(wk.order, (int)(x >> 2), pm, pn);
    if (wk.rev) pm = rev_panel(pm);
    t.m0 = pm * BM + q0 * 64;
    t.n0 = pn * BN;
    x += t.u;
    return t;
  };

  // DMA source offsets of a tile: LDS row r of the A image (wave group r >> 7, row lr = r & 127 inside it) holds
  // tile row 32 u (r >> 7) + lr for lr < 32 u and re-reads row 0 of the tile otherwise; rows past M re-read row M - 1
  auto offsets = [&](const Tile& t, Off& o) __attribute__((always_inline)) {
    int lane = threadIdx.x & 63;                            // opaque copy: recompute the lane arithmetic per call (once per
    asm volatile("" : "+v"(lane));                          // tile) instead of keeping a dozen hoisted registers live
    const int u32 = 32 * t.u;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int r = chunk_row(c, i * 8 + wave) + (lane >> 3);
        const int ch = (lane & 7) ^ (r & 7);
        int gr;
        if (c == 0 || c == 3) {
          const int lr = r & 127;
          gr = ((wk.dbg & 2) ? 0 : t.m0) + (lr < u32 ? (r >> 7) * u32 + lr : 0);
          gr = gr < M ? gr : M - 1;
        } else {
          gr = ((wk.dbg & 2) ? 0 : t.n0) + r;
          gr = gr < N ? gr : N - 1;
        }
        o.c[c][i] = ((unsigned)gr * (unsigned)K + (unsigned)(ch * 8)) * 2u;
      }
  };
  auto dma = [&](auto cc, const Off& o, int kbytes, char* stage) __attribute__((always_inline)) {
    constexpr int C = decltype(cc)::value;
    constexpr bool isA = (C == 0 || C == 3);
#ifdef SNX_NT256_NODMA                                  // diagnostics build (wrong results): the K loop without its LDS-DMA
    return;
#endif
    const char* base = (const char*)(isA ? A : B) + kbytes;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      char* dst = stage + (isA ? 0 : A_BYTES) + chunk_row(C, i * 8 + wave) * 128;
      __builtin_amdgcn_global_load_lds(GLB_PTR(base + o.c[C][i]), LDS_PTR(dst), 16, 0, 0);
    }
  };

  Tile cur = next_tile();
  if (cur.u == 0) return;
  Off off;
  offsets(cur, off);
  // prologue: all four chunks of the first K-tile
  dma(IC<0>(), off, 0, smem);
  dma(IC<1>(), off, 0, smem);
  dma(IC<2>(), off, 0, smem);
  dma(IC<3>(), off, 0, smem);
  WAIT_VM(0);
  BARRIER();
  if (wm == 1) BARRIER();                                   // stagger the lower-row wave group by one slot

  char* bounce = smem + 2 * STAGE + wave * BOUNCE;
  int stage_par = 0;
  int pend = 0;                                             // stores of the previous tile's write-back still in the vm queue
  bool deep = false;                                        // this tile's K-tile 1 was requested during the previous write-back
  f32x4 acc[8][4];

  // ---- one K-tile: four phases; N01 / N23 = row tiles (of 16 rows) of the wave in its first / second 64 rows ----
  // issue: request chunk p of the next K-tile in phase p (false: K-tile 0 of a tile whose K-tile 1 was requested during
  // the previous write-back).  hold: bit p set = the wait of phase p leaves the previous write-back's stores in
  // flight (vmcnt(4 + stores) instead of vmcnt(4)).  Two scalar branches per phase and no more: a LOAD segment runs
  // beside the partner group's 16 MFMAs = 256 cycles; with five branches per phase every launch was 15 % slower.
  auto ktile = [&](auto n01, auto n23, const char* st, char* nst, bool issue, int hold, int kbytes) __attribute__((always_inline)) {
    constexpr int N01 = decltype(n01)::value, N23 = decltype(n23)::value;
    auto wait_landed = [&](int ph) __attribute__((always_inline)) {
      if ((hold >> ph) & 1) {
        if (EPI == EPI_GEGLU_FWD) WAIT_VM(28);
        else if (EPI == EPI_RESID_F32 || EPI == EPI_GEGLU_BWD) WAIT_VM(36);
        else WAIT_VM(20);
      } else {
#if defined(SNX_NT256_LEAD_DIAG) && SNX_NT256_LEAD_DIAG == 1   // diagnostics builds: the steady-state wait one / two chunks
        WAIT_VM(2);                                             // earlier than needed (how much the loop depends on its lead)
#elif defined(SNX_NT256_LEAD_DIAG) && SNX_NT256_LEAD_DIAG == 2
        WAIT_VM(0);
#else
        WAIT_VM(4);
#endif
      }
    };
    const char* ta = st + (wm * 128) * 128;                 // this wave's 128 A rows
    const char* tb = st + A_BYTES + (wn * 64) * 128;        // this wave's 64 B rows (output columns)
    bf16x8 a[2][4], b0[2][2], b1[2][2];
    // ---------------- phase 0: quadrant (rows 0..63, cols 0..31) ----------------
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int i = 0; i < N01; ++i) NT256_AFRAG(a[kk][i], frag256(ta, i * 16 + li, kk * 4 + g));
#pragma unroll
      for (int jn = 0; jn < 2; ++jn) NT256_BFRAG(b0[kk][jn], frag256(tb, jn * 16 + li, kk * 4 + g));
    }
    if (issue) dma(IC<0>(), off, kbytes, nst);
    wait_landed(0);
    BARRIER();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < N01; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
          acc[i][jn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0[kk][jn], a[kk][i], acc[i][jn], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    HALF_DIAG_BARRIER();
    // ---------------- phase 1: quadrant (rows 0..63, cols 32..63) ----------------
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int jn = 0; jn < 2; ++jn) NT256_BFRAG(b1[kk][jn], frag256(tb, 32 + jn * 16 + li, kk * 4 + g));
    if (issue) dma(IC<1>(), off, kbytes, nst);
    wait_landed(1);
    HALF_DIAG_BARRIER();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < N01; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
          acc[i][2 + jn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1[kk][jn], a[kk][i], acc[i][2 + jn], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    BARRIER();
    // ---------------- phase 2: quadrant (rows 64..127, cols 32..63) ----------------
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < N23; ++i) NT256_AFRAG(a[kk][i], frag256(ta, 64 + i * 16 + li, kk * 4 + g));
    if (issue) dma(IC<2>(), off, kbytes, nst);
    wait_landed(2);
    BARRIER();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < N23; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
          acc[4 + i][2 + jn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1[kk][jn], a[kk][i], acc[4 + i][2 + jn], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    HALF_DIAG_BARRIER();
    // ---------------- phase 3: quadrant (rows 64..127, cols 0..31) ----------------
    if (issue) dma(IC<3>(), off, kbytes, nst);
    wait_landed(3);
    HALF_DIAG_BARRIER();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < N23; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
          acc[4 + i][jn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0[kk][jn], a[kk][i], acc[4 + i][jn], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    // (the trailing barrier of the K-tile is the caller's: the last one of an output tile frames the write-back)
  };

  // diagnostics (SNX_NT256_DBG bit 4): the write-back without its global stores (results kept alive in registers)
#define NT256_STORE(T, ptr, val)                           \
  do {                                                     \
    const T v__ = (val);                                   \
    if (wk.dbg & 4) asm volatile("" ::"v"(v__));           \
    else *(T*)(ptr) = v__;                                 \
  } while (0)
  // ---- write-back of the wave's 32 u x 64 result (transposed accumulators: lane owns 4 consecutive columns) ----
  //   acc[i][jn][r] = C[row0 + 16 i + li][col0 + 16 jn + 4 g + r]
  // Passes of 32 rows (16 for the GeGLU forward, whose second image y shares the 4 KiB): the operands a pass reads
  // from memory (residual rows, saved u, RoPE table rows) are requested one pass ahead.
  auto epilogue = [&](const Tile& t, auto deep_request) __attribute__((always_inline)) {
    const int row0 = t.m0 + wm * 32 * t.u;                  // first row of this wave
    const int col0 = t.n0 + wn * 64;
    if (col0 >= N || (wk.dbg & 1)) {                        // wave-uniform: a column span past the matrix (N % 256 != 0)
      deep_request();
      return;
    }
    // opaque copy of the lane id: none of the lane arithmetic below may be hoisted in front of the K loops, where
    // every register is taken (hipcc hoisted it and spilled 7 registers to scratch)
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));
    const int li = lane & 15, g = lane >> 4;
    const int rr = lane >> 3, rc = lane & 7;                // row-major role: rows rr + 8 k of a pass, 8 columns from 8 rc
    const int col = col0 + rc * 8;
    if (EPI == EPI_GEGLU_FWD) {
      deep_request();
      // passes of 16 rows: image u (16 x 128 B) + image y (16 x 64 B); same software pipeline as below
      char* img_y = bounce + 2048;
      const int ycol = (col0 >> 1) + (lane & 3) * 8;
      auto put_g = [&](int p) __attribute__((always_inline)) {
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) {
          const bf16x4 a4 = pack4(acc[p][jn]), g4 = pack4(acc[p][jn + 2]);
          stg_put(bounce, li, jn * 16 + g * 4, a4);
          stg_put(bounce, li, (jn + 2) * 16 + g * 4, g4);
          bf16x4 y4;
#pragma unroll
          for (int r = 0; r < 4; ++r) y4[r] = f2bf(rbf(gelu_f(bf2f(a4[r]))) * bf2f(g4[r]));
          stg32_put(img_y, li, jn * 16 + g * 4, y4);
        }
      };
      put_g(0);
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        if (p >= 2 * t.u) break;
        const bf16x8 v0 = stg_get<false>(bounce, rr, rc), v1 = stg_get<true>(bounce, rr + 8, rc);
        const bf16x8 vy = stg32_get(img_y, lane >> 2, lane & 3);
        if (p + 1 < 8 && p + 1 < 2 * t.u) put_g(p + 1);
        const int r0 = row0 + 16 * p + rr, ry = row0 + 16 * p + (lane >> 2);
        if (NTU) {
          if (r0 < M) __builtin_nontemporal_store(v0, (bf16x8*)(e.C + (long)r0 * N + col));
          if (r0 + 8 < M) __builtin_nontemporal_store(v1, (bf16x8*)(e.C + (long)(r0 + 8) * N + col));
        } else {
          if (r0 < M) NT256_STORE(bf16x8, e.C + (long)r0 * N + col, v0);
          if (r0 + 8 < M) NT256_STORE(bf16x8, e.C + (long)(r0 + 8) * N + col, v1);
        }
        if (ry < M) NT256_STORE(bf16x8, e.Y + (long)ry * (N >> 1) + ycol, vy);
      }
      return;
    }
    constexpr bool PRE = (EPI == EPI_RESID_F32 || EPI == EPI_GEGLU_BWD || EPI == EPI_ROPE);
    const bool rotate = EPI == EPI_ROPE && col0 < e.rope_cols;      // wave-uniform: a head of q or k
    f32x4 pre[2][PRE ? 8 : 1];
    // RoPE runs in the ROW-MAJOR layout of the read-back (lane = row rr + 8 k, columns 8 rc ..): lane (q = rc & 3,
    // h = rc >> 2) rotates the four pairs d = 8 q + 4 h + 0..3 of its row, (d, d + 32), read as two 8-byte pieces of
    // the staging image; its (cos, sin) are 32 contiguous bytes of the table row, 256 B per row over the 8 lanes
    // (in the accumulator layout the same values were 16 rows x 64 B per load instruction: +33 us per launch).
    // the positions of a pass's rows are loaded TWO passes ahead, its table rows one pass ahead
    // round 5: SIXTEEN-byte pieces -- lane (row = lane >> 2 of 16, q = lane & 3) rotates the EIGHT pairs d = 8 q + 0..7 of
    // its row: two 16-byte reads of the image, two 16-byte stores, i.e. the store-instruction count of a plain store (the
    // write-back is bound by the number of vector-memory instructions: with 8-byte pieces the rotated spans cost 32
    // stores per wave and tile instead of 16, +17 us per launch)
    const int xr = lane >> 2, xq = lane & 3;
    int prow[4][2];
    const bool by_row = EPI == EPI_ROPE && e.rope_rows != nullptr;   // (cos, sin) rows resolved per token beforehand
    auto request_pos = [&](int p) __attribute__((always_inline)) {
      if (EPI == EPI_ROPE && rotate && !by_row) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int row = row0 + 32 * p + xr + 16 * k;
          prow[p][k] = e.pos[row < M ? row : M - 1];
        }
      }
    };
    request_pos(0);
    if (t.u > 1) request_pos(1);
    auto request = [&](int p) __attribute__((always_inline)) {
      if (EPI == EPI_ROPE) {
        if (rotate) {
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            int row = row0 + 32 * p + xr + 16 * k;
            row = row < M ? row : M - 1;
            const f32x2* trow = by_row ? e.rope_rows + (long)row * 32 : e.rope_tab + (long)prow[p][k] * 32;
            const f32x4* cs = (const f32x4*)(trow + xq * 8);
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) pre[p & 1][4 * k + c4] = cs[c4];   // (cos, sin) of pairs 8 q + 2 c4, + 1
          }
        }
      } else if (PRE) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          int row = row0 + 32 * p + rr + 8 * k;
          row = row < M ? row : M - 1;
          if (EPI == EPI_RESID_F32) {
            const float* h = e.Hin + (long)row * N + col;
            pre[p & 1][2 * k] = *(const f32x4*)h;
            pre[p & 1][2 * k + 1] = *(const f32x4*)(h + 4);
          } else {                                          // dy columns [col, col+8) <-> a at u[64 q + 8 s], g at +32
            const bf16_t* uu = e.U + (long)row * (2 * N) + 64 * (col >> 5) + (col & 31);
            pre[p & 1][2 * k] = *(const f32x4*)uu;           // 8 bf16 of a
            pre[p & 1][2 * k + 1] = *(const f32x4*)(uu + 32);   // 8 bf16 of g
          }
        }
      }
    };
    // accumulators of pass p -> staging image (bf16)
    auto put = [&](int p) __attribute__((always_inline)) {
#pragma unroll
      for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int jn = 0; jn < 4; ++jn)
          stg_put(bounce, ii * 16 + li, jn * 16 + g * 4, pack4(acc[2 * p + ii][jn]));
    };
    if (PRE) request(0);
    deep_request();
    put(0);
    // Software pipeline over the ONE image: the row-major reads of pass p are issued, then the image is refilled
    // with pass p + 1 (LDS executes a wave's instructions in order: the refill cannot overtake the reads), then the
    // reads are consumed -- their latency runs under the packing and the writes of the next pass.
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      if (p >= t.u) break;                                  // wave-uniform: a short tile has u passes
      if (p + 2 < 4 && p + 2 < t.u) request_pos(p + 2);
      if (PRE && p + 1 < 4 && p + 1 < t.u) request(p + 1);
      bf16x8 v[4];
      bf16x8 x1[2], x2[2];
      if (EPI == EPI_ROPE && rotate) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {                       // the Linear's output is bf16: rotate what was rounded
          // rows xr + 16 k: bit 3 of the row (the image's half swap) is bit 3 of xr for both k
          x1[k] = (xr & 8) ? stg_get<true>(bounce, xr + 16 * k, xq) : stg_get<false>(bounce, xr + 16 * k, xq);
          x2[k] = (xr & 8) ? stg_get<true>(bounce, xr + 16 * k, 4 + xq) : stg_get<false>(bounce, xr + 16 * k, 4 + xq);
        }
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          v[k] = (k & 1) ? stg_get<true>(bounce, rr + 8 * k, rc) : stg_get<false>(bounce, rr + 8 * k, rc);
      }
      if (p + 1 < 4 && p + 1 < t.u) put(p + 1);
      if (EPI == EPI_ROPE && rotate) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int row = row0 + 32 * p + xr + 16 * k;
          bf16x8 lo, hi;
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            const f32x4 cs = pre[p & 1][4 * k + (r >> 1)];
            const float c = cs[(r & 1) * 2], sn = cs[(r & 1) * 2 + 1];
            const float a1 = bf2f(x1[k][r]), a2 = bf2f(x2[k][r]);
            lo[r] = f2bf(mul_rn(a1, c) - mul_rn(a2, sn));   // products rounded separately, as torch's
            hi[r] = f2bf(mul_rn(a2, c) + mul_rn(a1, sn));   // q * cos + rotate_half(q) * sin (hf:196-219)
          }
          if (row >= M) continue;
          bf16_t* o = e.C + (long)row * N + col0 + xq * 8;
          NT256_STORE(bf16x8, o, lo);
          NT256_STORE(bf16x8, o + 32, hi);
        }
        continue;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int row = row0 + 32 * p + rr + 8 * k;
        if (row >= M) continue;
        if (EPI == EPI_RESID_F32) {
          const f32x4 h0 = pre[p & 1][2 * k], h1 = pre[p & 1][2 * k + 1];
          float* o = e.Hout + (long)row * N + col;
          NT256_STORE(f32x4, o, ((f32x4){h0[0] + bf2f(v[k][0]), h0[1] + bf2f(v[k][1]), h0[2] + bf2f(v[k][2]), h0[3] + bf2f(v[k][3])}));
          NT256_STORE(f32x4, o + 4, ((f32x4){h1[0] + bf2f(v[k][4]), h1[1] + bf2f(v[k][5]), h1[2] + bf2f(v[k][6]), h1[3] + bf2f(v[k][7])}));
        } else if (EPI == EPI_GEGLU_BWD) {                  // v = dy (bf16); du = GeGLU'(u, dy)
          const bf16x8 a8 = __builtin_bit_cast(bf16x8, pre[p & 1][2 * k]), g8 = __builtin_bit_cast(bf16x8, pre[p & 1][2 * k + 1]);
          bf16x8 da, dg;
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            const float af = bf2f(a8[r]), gf = bf2f(g8[r]), df = bf2f(v[k][r]);
            dg[r] = f2bf(df * rbf(gelu_f(af)));
            da[r] = f2bf(rbf(df * gf) * gelu_grad_f(af));
          }
          bf16_t* o = e.C + (long)row * (2 * N) + 64 * (col >> 5) + (col & 31);
          NT256_STORE(bf16x8, o, da);
          NT256_STORE(bf16x8, o + 32, dg);
        } else {
          NT256_STORE(bf16x8, e.C + (long)row * N + col, v[k]);
        }
      }
    }
  };

  while (true) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int jn = 0; jn < 4; ++jn) acc[i][jn] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const Tile nxt = next_tile();
    const int u = cur.u;
    // (one K loop per tile height: with the four bodies inside one loop hipcc spilled 300 registers)
    // The previous write-back's stores may stay in flight through K-tile 0 -- and, when K-tile 1 was requested ahead
    // of them (deep), through the first two phases of K-tile 1: the vm queue retires in order.
    auto kloop = [&](auto n01, auto n23) __attribute__((always_inline)) {
      for (int kt = 0; kt < nk; ++kt, stage_par ^= 1) {
        const char* st = smem + stage_par * STAGE;
        char* nst = smem + (stage_par ^ 1) * STAGE;
        // where the next K-tile comes from (next k of this output tile, or k = 0 of the next one).  The very last
        // K-tile of the workgroup re-requests its own chunks into the free stage (nobody reads them): the waits
        // then need no third form; the requests are drained before the workgroup ends.
        const bool more_k = kt + 1 < nk;
        if (!more_k && nxt.u > 0) offsets(nxt, off);        // the prefetch stream moves on to the next tile
        const int kbytes = more_k ? (kt + 1) * (BK * 2) : 0;
        const int soft = kt == 0 ? (deep ? 15 : 3) : (kt == 1 && deep) ? 3 : 0;
        ktile(n01, n23, st, nst, !(deep && kt == 0), pend ? soft : 0, kbytes);
        if (more_k) BARRIER();
      }
    };
    if (u == 4) kloop(IC<4>(), IC<4>());
    else if (u == 3) kloop(IC<4>(), IC<2>());
    else if (u == 2) kloop(IC<4>(), IC<0>());
    else kloop(IC<2>(), IC<0>());
    // the leading group writes back after the tile's last barrier, the trailing group before it: same slot
    if (wm == 0) BARRIER();
    // K-tile 1 of the next tile is requested HERE, in front of the write-back's stores: its stage is the one just
    // consumed (last read >= 3 slots ago).  K-tile 0 of the next tile then requests nothing and its waits -- like
    // those of the first half of K-tile 1 -- leave the stores in flight: a whole K-tile for the 128 KiB burst to
    // drain behind the MFMAs instead of in front of them.
    // (behind the first operand loads of the write-back itself: the queue returns in order)
    const bool deep_next = nxt.u > 0 && nk >= 2 && !(wk.dbg & 16);
    auto deep_request = [&]() __attribute__((always_inline)) {
      if (deep_next) {
        char* s1 = smem + (stage_par ^ 1) * STAGE;
        dma(IC<0>(), off, BK * 2, s1);
        dma(IC<1>(), off, BK * 2, s1);
        dma(IC<2>(), off, BK * 2, s1);
        dma(IC<3>(), off, BK * 2, s1);
      }
    };
    epilogue(cur, deep_request);
    if (wm != 0) BARRIER();
    deep = deep_next;
    // a whole tile inside the matrix: this wave has just issued a known number of stores
    pend = 0;
    if ((u == 4) && !(wk.dbg & 5) && (cur.n0 + wn * 64 < N) && (cur.m0 + wm * 128 + 127 < M)) {
      pend = 16;
      if (EPI == EPI_RESID_F32 || EPI == EPI_GEGLU_BWD) pend = 32;
      if (EPI == EPI_GEGLU_FWD) pend = 24;
      if (wk.dbg & 8) pend = 0;
    }
    cur = nxt;
    if (cur.u == 0) break;
  }
  if (wm == 0) BARRIER();                                   // balance the stagger barrier
  WAIT_VM(0);                                               // no LDS-DMA may outlive the workgroup's LDS allocation
}

template <int EPI, bool NTU = false>
static int launch(const void* A, const void* B, int M, int N, int K, const EpiArgs& e, hipStream_t st) {
  const int tm = cdiv(M, BM), tn = cdiv(N, BN);
  // column-group width of the tile order: an XCD's share of B (cg tiles of 256 x K bf16) should stay in its 4 MiB L2
  // beside the streaming A panels and outputs; re-reading the A panels ceil(tn / cg) times is the price
  const int cg_env = SNX_DIAG_CFG(nt256_cg, -1);
  const int dbg = SNX_DIAG_CFG(nt256_dbg, 0);
  int cg = tn;
  if (cg_env > 0) cg = cg_env < tn ? cg_env : tn;
  else if (cg_env < 0) {
    const double a_bytes = 2.0 * M * K, b_bytes = 2.0 * N * K, cap = 2.2e6;
    double best = -1;
    for (int parts = 1; parts <= 4; ++parts) {
      const int c = cdiv(tn, parts);
      const double bsub = 2.0 * c * BN * K;
      const double cost = a_bytes * cdiv(tn, c) + 8.0 * b_bytes * (bsub <= cap ? 1.0 : 4.0);
      if (best < 0 || cost < best) { best = cost; cg = c; }
    }
  }
  Work wk;
  wk.order = TileOrder{tm, tn, cdiv(tm, 8), cg};
  wk.units = 4L * tm * tn;
  wk.dbg = dbg;
  // column-run dealing of the leftover units: needs position -> (p / tn, p % tn) inside a super-block, i.e. ONE column
  // group, and every XCD's leftover tiles inside one super-block ("nt256_coldeal" = 0: the round-3 dealing, for A/B)
  // reverse walk for the launches whose A operand is the big, just-written gradient tensor: the plain-store dX GEMMs with a
  // long contraction (K >= 3 N: du / dqkv against Wi^T / Wqkv^T)
  wk.rev = (g_snx_cfg.nt256_rev != 0 && EPI == EPI_STORE_BF16 && K >= 3 * N) ? 1 : 0;
  wk.coldeal = 0;
  {
    const int nwg_ = NWG - snx_get_reserved_cus(), per = nwg_ >> 3;
    const long ntiles = (long)tm * tn, per_sb = (long)wk.order.sb_rows * tn;
    bool ok = cg == tn && per > 0 && g_snx_cfg.nt256_coldeal != 0;
    for (int xcd = 0; ok && xcd < 8; ++xcd) {
      const long P0 = ntiles * xcd / 8, P1 = ntiles * (xcd + 1) / 8;
      const long nround = (P1 - P0) / per, Pl = P0 + nround * per;
      if (P1 > Pl && Pl / per_sb != (P1 - 1) / per_sb) ok = false;
    }
    wk.coldeal = ok ? 1 : 0;
  }
  static bool attr[64] = {};
  int devid = 0;
  if (hipGetDevice(&devid) != hipSuccess || devid < 0 || devid >= 64) return SNX_E_ARG;
  auto kern = gemm_nt256_kernel<EPI, NTU>;
  static bool done[64] = {};
  if (!done[devid]) {
    hipError_t err = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
    if (err != hipSuccess) return (int)err;
    done[devid] = true;
  }
  (void)attr;
  // CUs left to RCCL while a gradient bucket is exchanged (snx_set_reserved_cus, gemm_tn256.hip): this kernel takes
  // whole CUs too, and a workgroup that finds none free runs its entire share after another has finished -- twice
  // the launch.  Tiles are dealt in rounds of gridDim / 8 per XCD: any multiple of 8 workgroups works.
  const int nwg = NWG - snx_get_reserved_cus();
  hipLaunchKernelGGL(kern, dim3(nwg), dim3(512), LDS_TOTAL, st, (const bf16_t*)A, (const bf16_t*)B, M, N, K, wk, e);
  SNX_CHECK_LAUNCH();
  return SNX_OK;
}

// Shapes this form takes: enough rows to give every workgroup work, N % 64 == 0 (16-byte row-major write-back in
// whole 64-column wave spans), K % 64 == 0, operand byte offsets within 32 bits.
// A/B and test switch (= snx_configure "nt256" / "nt256_min_m"): on = 0 keeps every shape on the 128x128 kernel
extern "C" int snx_nt256_configure(int32_t on, int32_t min_m) {
  const int rc = snx_configure("nt256", on);           // 0 = off, 1 = default shape policy, 2 = every eligible shape
  if (rc != SNX_OK) return rc;
  return min_m > 0 ? snx_configure("nt256_min_m", min_m) : SNX_OK;
}

int snx_launch_nt256(int epi, const void* A, const void* B, int M, int N, int K, const EpiArgs& e, hipStream_t st) {
  const int on = g_snx_cfg.nt256, min_m = g_snx_cfg.nt256_min_m;
  // Which shapes come here by default (measured at 36,864 rows against the 128x128 kernel, tools/gpu_nt256.py):
  // wide outputs (>= 6 column tiles: ~20 units per workgroup, short tiles are a small share) with the RoPE and
  // GeGLU-forward epilogues, and the plain store from three column tiles on (N = 768: 6.75 units per workgroup, one
  // whole tile and one or two short ones that cost most of a whole tile's latency-bound K loop -- level with the
  // 128x128 kernel in the microbenchmark, 0.5 ms per micro-step faster inside the training step, where operands
  // come cold).  The two epilogues that stream a second operand (residual, GeGLU backward) gain nothing from the
  // exposed write-back (bench with SNX_NT256_FORCE=2 / 16: -0.2 / -0.4 ms the wrong way).
  // diagnostics builds, "nt256_force" = <bitmask over EPI>: take every eligible shape of those epilogues (A/B runs).
  const int force = SNX_DIAG_CFG(nt256_force, 0);
  const bool dflt = ((epi == EPI_ROPE || epi == EPI_GEGLU_FWD) && N >= 6 * BN) || (epi == EPI_STORE_BF16 && N >= 3 * BN);
  if (on != 2 && !dflt && !((force >> epi) & 1)) return SNX_E_SHAPE;   // on = 2 (snx_nt256_configure): take all
  // a plain store with less than one round of 256x256 tiles leaves workgroups idle for the whole launch: the 128x128
  // kernel is faster there (16,384 rows -- a document pass of the three-call loop --, N = 768: 192 tiles, 30.0 against
  // 25.0 us at K = 768, 75.9 against 66.0 at K = 2304; profiles/r05_experiments.txt section 4)
  if (on != 2 && epi == EPI_STORE_BF16 && (long)cdiv(M, BM) * cdiv(N, BN) < NWG) return SNX_E_SHAPE;
  if (!on || M < min_m || (N % 64) || (K % 64) || K < 64) return SNX_E_SHAPE;
  if ((long)M * K * 2 >= (1L << 32) || (long)N * K * 2 >= (1L << 32)) return SNX_E_SHAPE;
  switch (epi) {
    case EPI_STORE_BF16: return launch<EPI_STORE_BF16>(A, B, M, N, K, e, st);
    case EPI_RESID_F32: return launch<EPI_RESID_F32>(A, B, M, N, K, e, st);
    case EPI_ROPE: return launch<EPI_ROPE>(A, B, M, N, K, e, st);
    case EPI_GEGLU_FWD:
      return (g_snx_cfg.stream_nt & 8) ? launch<EPI_GEGLU_FWD, true>(A, B, M, N, K, e, st)
                                       : launch<EPI_GEGLU_FWD>(A, B, M, N, K, e, st);
    case EPI_GEGLU_BWD: return launch<EPI_GEGLU_BWD>(A, B, M, N, K, e, st);
    default: return SNX_E_SHAPE;
  }
}
