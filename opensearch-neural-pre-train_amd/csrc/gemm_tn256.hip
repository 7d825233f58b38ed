// Weight-gradient GEMM, 256x256 persistent "ping-pong" form:  dW[N,K] += dY[M,N]^T * X[M,K]  (contraction over
// the token rows; replaces the autograd dW = dY^T X of every nn.Linear on the path, the `+=` being
// ref:src/train/cli/train_v33_ddp.py:364's gradient accumulation).
//
// Why a second form: the 128x128 kernel (gemm.hip) moves 65 FLOP per byte from L2 into LDS -- at 64 B/clk/CU that
// path saturates together with the matrix pipe (measured with in-kernel stamps, tools/gpu_gemm_trace.py: two waves
// per SIMD issue MFMAs 60 % of the K loop).  A 256x256 tile halves the bytes and the LDS-DMA instructions per MFMA.
//
// One 8-wave workgroup per CU (128 KiB LDS), waves 2 (dY column halves) x 4 (X column quarters), wave tile 128 x 64
// = 8 x 4 accumulators of v_mfma_f32_16x16x32_bf16.  The two waves of every SIMD (wave w and w+4) run the same
// program one barrier apart, so while one multiplies the other reads LDS and issues LDS-DMA:
//     slot:        s         s+1       s+2       s+3
//     waves 0-3:   LOAD p    MFMA p    LOAD p+1  MFMA p+1
//     waves 4-7:   MFMA p-1  LOAD p    MFMA p    LOAD p+1            (every slot ends in one s_barrier)
// A K-step (64 tokens) is two phases, one 64x64 half of the wave tile each (32 MFMAs = 512 matrix-pipe cycles,
// long enough to cover the partner's LOAD segment).  Its operands are four 16 KiB sub-tiles (gemm_tn.h image):
// 0 = dY columns {0-63, 128-191} (the upper 64 rows of both wave rows), 3 = dY columns {64-127, 192-255},
// 1 / 2 = X columns {64w + 0-31} / {64w + 32-63} of the four wave columns.  Phase 0 reads sub-tiles 0, 1, 2,
// phase 1 reads 3.  A sub-tile's LDS region is refilled for K-step t+2 in the phase AFTER its last read (two
// stages of 64 KiB hold t, t+1 and the parts of t+2 already requested): 0, 1, 2 in phase 1 of t, 3 in phase 0 of
// t+1, 2 DMA instructions per wave each, and every LOAD segment ends with the COUNTED s_waitcnt vmcnt(8): all but
// the four newest requests have landed.  A request has one K-step to come back from HBM (the operands of a
// weight gradient are a layer old: nothing of them is in L2).
//   RAW: a sub-tile is read one barrier after the covering wait of BOTH wave groups.
//   WAR: a refill is issued one barrier after both groups' reads, and every LOAD segment ends with lgkmcnt(0).
//
// Work split (no inter-workgroup dependency, float atomics do the reduction as in gemm.hip): XCD x = blockIdx & 7
// owns the token segment x of 8 and computes ALL output tiles of the group for it, so a token row leaves HBM for
// one XCD only; its 32 workgroups take tile r*32 + j in round r over the whole segment, in step with each other
// (the dY / X rows of a K-step are shared through the XCD's L2), and the R < 32 tiles of the last round are cut
// into floor(32 / R) token pieces each.  Accumulators leave by float atomics at the end of every (tile, piece).
#include "gemm_tn.h"
#include "snx.h"

namespace {

constexpr int SUB = 64 * 256;        // one sub-tile: 64 tokens x 128 columns bf16
constexpr int STAGE = 4 * SUB;       // 64 KiB
constexpr int NXCD = 8;

__device__ uint4 zero_page256[16];   // 256 B of zeros: DMA source for tokens past M and columns past N / K

struct Item { int tile, sb, se; };   // K-steps [sb, se) of output tile `tile`

struct Tile {                        // one output tile of the group
  const bf16_t* dy;                  // dY + n0
  const bf16_t* x;                   // X + k0
  float* dw;
  int N, K, n0, k0, inter;
};

struct Sched { int ntiles, nsteps, seg, W, dbg; };

__device__ __forceinline__ bool item_at(const Sched& s, int it, int x, int j, Item& o) {
  const int full = s.ntiles / s.W, R = s.ntiles - full * s.W;
  const int L0 = x * s.seg, L1 = min(s.nsteps, L0 + s.seg);
  if (L0 >= L1) return false;
  if (it < full) {
    o.tile = it * s.W + j; o.sb = L0; o.se = L1;
    return true;
  }
  if (it == full && R > 0) {
    const int p = s.W / R;
    const int plen = (L1 - L0 + p - 1) / p;
    const int t = j / p, q = j - t * p;
    if (t >= R) return false;
    const int sb = L0 + q * plen, se = min(L1, sb + plen);
    if (sb >= se) return false;
    o.tile = full * s.W + t; o.sb = sb; o.se = se;
    return true;
  }
  return false;
}

__device__ __forceinline__ Tile decode(const TnGroup& g, int tile) {
  int p = 0;
#pragma unroll
  for (int q = 0; q < SNX_TN_MAX_GROUP - 1; ++q)
    if (q + 1 < g.nprob && tile >= g.tile_end[q]) p = q + 1;
  if (p > 0) tile -= g.tile_end[p - 1];
  Tile t;
  t.N = g.N[p]; t.K = g.K[p]; t.inter = g.inter[p];
  const int tk = (t.K + 255) >> 8;
  t.n0 = (tile / tk) * 256;
  t.k0 = (tile % tk) * 256;
  t.dy = g.dY[p] + t.n0;
  t.x = g.X[p] + t.k0;
  t.dw = g.dW[p];
  return t;
}

// Per-lane byte offsets of the two DMA instructions a wave issues per sub-tile: (token row) * ld + column, from the
// K-step's first token row and the tile's first column.  Sub-tiles 3 / 2 are sub-tiles 0 / 1 shifted by 64 / 32
// columns (added to the wave-uniform base).  Columns past the matrix edge (K = 1152 -> half a tile) are
// clamped into it: they only feed accumulators that are never written back.
struct LaneOff { unsigned a[2], b[2]; };

__device__ __forceinline__ LaneOff lane_offsets(const Tile& t, int wave, int lane) {
  LaneOff o;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (i * 8 + wave) * 4 + (lane >> 4);
    const int lc = tn_chunk(row, lane & 15) * 8;     // first of the 8 local columns held at this lane's slot
    int ca = (lc >> 6) * 128 + (lc & 63), cb = (lc >> 5) * 64 + (lc & 31);
    ca = min(ca, t.N - t.n0 - 72);                   // ca + 64 + 8 <= N - n0
    cb = min(cb, t.K - t.k0 - 40);
    o.a[i] = (unsigned)(row * t.N + ca) * 2u;
    o.b[i] = (unsigned)(row * t.K + cb) * 2u;
  }
  return o;
}

// request sub-tile C of the K-step starting at token tok0 into `stage` (tok0 + 64 <= M)
template <int C>
__device__ __forceinline__ void dma_sub(const Tile& t, const LaneOff& o, int tok0, char* stage, int wave) {
  constexpr bool isA = (C == 0 || C == 3);
  // (the instruction's immediate offset would move the LDS address too: the column shift goes into the base)
  const char* base = isA ? (const char*)(t.dy + (long)tok0 * t.N + (C == 3 ? 64 : 0))
                         : (const char*)(t.x + (long)tok0 * t.K + (C == 2 ? 32 : 0));
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const char* src = base + (isA ? o.a[i] : o.b[i]);
    __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(stage + C * SUB + (i * 8 + wave) * 1024), 16, 0, 0);
  }
}

// the K-step that holds the last, partly filled 64 token rows of the matrix: rows past M read zeros
template <int C>
__device__ __forceinline__ void dma_sub_tail(const Tile& t, const LaneOff& o, int tok0, int M, char* stage, int wave,
                                             int lane) {
  constexpr bool isA = (C == 0 || C == 3);
  const char* base = isA ? (const char*)(t.dy + (long)tok0 * t.N) : (const char*)(t.x + (long)tok0 * t.K);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (i * 8 + wave) * 4 + (lane >> 4);
    const char* src = tok0 + row < M ? base + (isA ? o.a[i] : o.b[i]) + (C == 3 ? 128 : C == 2 ? 64 : 0)
                                     : (const char*)zero_page256 + (lane & 15) * 16;
    __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(stage + C * SUB + (i * 8 + wave) * 1024), 16, 0, 0);
  }
}

#define WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define BARRIER()                          \
  do {                                     \
    __builtin_amdgcn_sched_barrier(0);     \
    __builtin_amdgcn_s_barrier();          \
    __builtin_amdgcn_sched_barrier(0);     \
  } while (0)

}  // namespace

__global__ __launch_bounds__(512) void gemm_tn256_kernel(TnGroup grp, int M, Sched sch) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int li = lane & 15, g = lane >> 4;
  const int xcd = blockIdx.x & (NXCD - 1), jw = blockIdx.x >> 3;
  const int nitems = sch.ntiles / sch.W + ((sch.ntiles % sch.W) ? 1 : 0);

  auto next_item = [&](int from, Item& o) {          // first non-empty item with index >= from, or -1
    for (int it = from; it < nitems; ++it)
      if (item_at(sch, it, xcd, jw, o)) return it;
    return -1;
  };

  // ---- the two streams: requests (ld_*) run 7 sub-tiles ahead of the reads (cp_*) ----
  Item ld_item, cp_item;
  int ld_it = next_item(0, ld_item);
  if (ld_it < 0) return;
  int cp_it = ld_it;
  cp_item = ld_item;
  Tile ld_tile = decode(grp, ld_item.tile), cp_tile = ld_tile;
  LaneOff ld_off = lane_offsets(ld_tile, wave, lane);
  int ld_s = ld_item.sb, cp_s = cp_item.sb;
  int ld_par = 0;
  bool ld_ok = true;
  auto ld_advance = [&]() {                           // the request stream moves on to its next K-step
    ld_par ^= 1;
    if (++ld_s == ld_item.se) {
      ld_it = next_item(ld_it + 1, ld_item);
      ld_ok = ld_it >= 0;
      if (ld_ok) {
        ld_tile = decode(grp, ld_item.tile);
        ld_off = lane_offsets(ld_tile, wave, lane);
        ld_s = ld_item.sb;
      }
    }
  };
#define REQ(C)                                                                                          \
  do {                                                                                                  \
    if (ld_ok && !(sch.dbg & 2)) {                                                                      \
      const int tok0 = (sch.dbg & 4) ? 0 : ld_s * 64;                                                   \
      if (tok0 + 64 <= M) dma_sub<C>(ld_tile, ld_off, tok0, smem + ld_par * STAGE, wave);               \
      else dma_sub_tail<C>(ld_tile, ld_off, tok0, M, smem + ld_par * STAGE, wave, lane);                \
    }                                                                                                   \
  } while (0)
#define LOAD_END()                       \
  do {                                   \
    if (ld_ok) WAIT_VM(8);               \
    else WAIT_VM(0);                     \
    WAIT_LGKM0();                        \
    BARRIER();                           \
  } while (0)

  // prologue: sub-tiles 0..3 of the first K-step, 0..2 of the second
  REQ(0); REQ(1); REQ(2); REQ(3);
  ld_advance();
  REQ(0); REQ(1); REQ(2);
  if (ld_ok) WAIT_VM(8);
  else WAIT_VM(0);
  BARRIER();
  if (wm == 1) BARRIER();                             // stagger the second wave group by one slot

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int jn = 0; jn < 4; ++jn) acc[i][jn] = (f32x4){0.f, 0.f, 0.f, 0.f};

  int par = 0;
  while (true) {
    const char* st = smem + par * STAGE;
    bf16x8 a[2][4], b[2][4];
    // ---------------- phase 0: rows 0..63 of the wave tile ----------------
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int i = 0; i < 4; ++i) a[kk][i] = tn_frag(st, kk * 32, wm * 64 + i * 16, lane);
#pragma unroll
      for (int jn = 0; jn < 4; ++jn)
        b[kk][jn] = tn_frag(st + (1 + (jn >> 1)) * SUB, kk * 32, wn * 32 + (jn & 1) * 16, lane);
    }
    REQ(3);
    LOAD_END();
    ld_advance();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jn = 0; jn < 4; ++jn)
          acc[i][jn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kk][i], b[kk][jn], acc[i][jn], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    BARRIER();
    // ---------------- phase 1: rows 64..127 ----------------
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 4; ++i) a[kk][i] = tn_frag(st + 3 * SUB, kk * 32, wm * 64 + i * 16, lane);
    REQ(0); REQ(1); REQ(2);
    LOAD_END();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jn = 0; jn < 4; ++jn)
          acc[4 + i][jn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kk][i], b[kk][jn], acc[4 + i][jn], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    BARRIER();
    par ^= 1;

    if (++cp_s == cp_item.se) {
      // ---- end of a (tile, token piece): add the accumulators to the gradient ----
      // acc[i][jn][r] = dW[n0 + wm*128 + i*16 + 4g + r][k0 + wn*64 + jn*16 + li]
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int n = cp_tile.n0 + wm * 128 + i * 16 + g * 4 + r;
          const bool nok = n < cp_tile.N;
          if (cp_tile.inter > 0)                      // dY columns are in the interleaved GeGLU order
            n = ((n & 63) < 32) ? 32 * (n >> 6) + (n & 31) : cp_tile.inter + 32 * (n >> 6) + (n & 31);
#pragma unroll
          for (int jn = 0; jn < 4; ++jn) {
            const int k = cp_tile.k0 + wn * 64 + jn * 16 + li;
            if (nok && k < cp_tile.K && !(sch.dbg & 1)) atomicAdd(cp_tile.dw + (long)n * cp_tile.K + k, acc[i][jn][r]);
            acc[i][jn][r] = 0.f;
          }
        }
      cp_it = next_item(cp_it + 1, cp_item);
      if (cp_it < 0) break;
      cp_tile = decode(grp, cp_item.tile);
      cp_s = cp_item.sb;
    }
  }
  if (wm == 0) BARRIER();                             // balance the stagger barrier
#undef REQ
#undef LOAD_END
}

int snx_launch_tn256(const TnGroup& g128, int M, hipStream_t st) {
  TnGroup g = g128;
  int run = 0;
  for (int p = 0; p < g.nprob; ++p) {
    run += cdiv(g.N[p], 256) * cdiv(g.K[p], 256);
    g.tile_end[p] = run;
  }
  Sched s;
  s.ntiles = run;
  s.nsteps = cdiv(M, 64);
  s.seg = cdiv(s.nsteps, NXCD);
  s.W = 32;
  static const int dbg = getenv("SNX_TN256_DBG") ? atoi(getenv("SNX_TN256_DBG")) : 0;   // diagnostics: 1 = no atomics, 2 = no DMA
  s.dbg = dbg;
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_tn256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       2 * STAGE);
    if (e != hipSuccess) return (int)e;
    attr = true;
  }
  hipLaunchKernelGGL(gemm_tn256_kernel, dim3(NXCD * s.W), dim3(512), 2 * STAGE, st, g, M, s);
  SNX_CHECK_LAUNCH();
  return SNX_OK;
}
