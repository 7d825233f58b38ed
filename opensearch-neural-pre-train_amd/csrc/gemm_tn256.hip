// Weight-gradient GEMM, 256x256 persistent form:  dW[N,K] += dY[M,N]^T * X[M,K]  (contraction over the token rows;
// replaces the autograd dW = dY^T X of every nn.Linear on the path, the `+=` being
// ref:src/train/cli/train_v33_ddp.py:364's gradient accumulation).
//
// Why a second form: the 128x128 kernel (gemm.hip) moves 65 FLOP per byte from L2 into LDS -- at 64 B/clk/CU that
// path saturates together with the matrix pipe (measured with in-kernel stamps, tools/gpu_gemm_trace.py: two waves
// per SIMD issue MFMAs 60 % of the K loop) -- and its operands are cold (a layer old).  A 256x256 tile moves 131
// FLOP per byte, and a 128x128 WAVE tile halves the LDS fragment reads per MFMA.
//
// One 4-wave workgroup per CU = ONE wave per SIMD with the whole register file: wave tile 128 x 128 = 4 x 4
// accumulator tiles of v_mfma_f32_32x32x16_bf16 (all 256 AGPRs) + two sets of 16 fragments (128 VGPRs).  The wave
// pipelines itself: while the 32 MFMAs of half-step h (32 token rows) run, it requests half-step h+5 from HBM
// (8 LDS-DMA instructions) and reads the fragments of h+1 from LDS (32 ds_read_b64_tr_b16), all issued into the
// gaps between MFMAs.  All 160 KiB of LDS form a ring of FIVE half-steps (32 x 256 of dY + 32 x 256 of X = 32 KiB,
// stored as 32-row x 128-column sub-tiles in the image of gemm_tn.h).  Per half-step there is ONE barrier:
//     s_waitcnt vmcnt(24)   -- all but the three newest half-steps requested by this wave have landed, i.e. h+1
//     s_barrier             -- ... for every wave; and every wave has the fragments of h in registers
//     request h+5 into the slot of h;  read fragments h+1;  32 MFMAs of h;  s_waitcnt lgkmcnt(0)
// A request has four half-steps (two 64-token K-steps, 128 KiB in flight per CU) to come back from HBM.
//
// What the compiler needed (each measured): the accumulators pinned to AGPRs by an asm constraint and written by
// MFMAs only (C = 0 form on the first half-step of an item; otherwise they move to VGPRs and scratch once the flush
// reads them); the LDS reads as volatile asm (otherwise s_waitcnt vmcnt(0) in front of each while LDS-DMA is in
// flight); 32x32x16, not 16x16x32 MFMAs (24 instead of 8 free issue cycles per MFMA for the 70-odd other
// instructions of a half-step); NO branch in the K loop body (one s_cbranch per MFMA gap: 64 instead of 32 cycles
// per MFMA) -- past the end of a stream requests re-fetch parked rows and reads fetch fragments nobody uses -- and
// no division in it (the item schedule is advanced once per item); three copies of the body at most (a fourth and
// the register allocator spills).
//
// Work split (no inter-workgroup dependency): with n output tiles in the group, P = floor(256 / n) token pieces;
// workgroup (piece p, tile t) runs ONE long item and flushes its accumulators once (an XCD-segment split tried first
// flushed three times per workgroup: 188 MB of atomics and 90 us per launch).  ORDERED REDUCTION (round 5): a flush is a
// plain store of the 256 x 256 fp32 partial into slab (contributor c, tile t) of a caller-owned workspace, and a second
// small kernel (tn256_reduce_kernel) adds a tile's slabs to dW in the FIXED order c = 0, 1, ... (pieces by token range,
// then the tail workgroups by id): the weight gradients are bit-reproducible from run to run, whatever order the
// workgroups finish in.  Rounds 2-4 flushed with float atomics (order = arrival order: two runs differed in the last
// bits); that form stays behind snx_configure("det_reduce", 0) for A/B.  The 256 - P*n remaining workgroups share the last `tail_len` K-steps of the token range over
// all tiles, stream-K fashion (contiguous (tile, K-step) ranges), so that every workgroup multiplies about
// n * steps / 256 K-steps.  Logical workgroup ids are XCD-contiguous (blockIdx & 7 = XCD): an XCD's 32 workgroups
// are neighbouring tiles of the same token piece and walk it in step, sharing dY / X rows through their L2.
// N and K need only be multiples of 128 (a tile may lie half outside the matrix: its DMA re-reads valid columns
// and its flush is predicated); M a multiple of 64 (the launcher hands a ragged rest to the 128x128 kernel).
//
// Measured (layer group of the 149 M model, 36,864 tokens, 369 GFLOP): 356 us in the training step = 1.0 PFLOP/s
// (128x128 kernel: 479 us); back to back in a microbenchmark 455 us at an in-kernel clock of 1.65 GHz (the chip
// lowers its clock under the sustained load), compute-only 300 us.  In-kernel stamps: tools/gpu_tnbench.py with a
// -DSNX_GEMM_TRACE build.  Forms tried before this one (8-wave ping-pong with quadrant / half phases, 16x16x32 MFMAs):
// 585 / 477 / 521 us in the same microbenchmark; 256x192 tiles with 192 accumulators: 482 us.
#include "gemm_tn.h"
#include "config.h"
#include "snx.h"

namespace {

constexpr int HS = 32;               // token rows per half-step
constexpr int SUB = HS * 256;        // one sub-tile: 32 tokens x 128 columns bf16 = 8 KiB
constexpr int PART = 2 * SUB;        // one operand slice: 32 tokens x 256 columns
constexpr int SLOT = 2 * PART;       // dY slice + X slice = 32 KiB
constexpr int RING = 5;              // 160 KiB
constexpr int NWG = 256;              // one workgroup per CU; fewer while CUs are reserved (snx_set_reserved_cus)
constexpr int TK = 256;              // X columns of an output tile (dY columns: 256)
constexpr int NJ = 4;                // 32-column accumulator tiles per wave along X (4 along dY): wave tile 128 x 128

struct Item { int tile, sb, se; };   // K-steps (64 token rows = two half-steps) [sb, se) of output tile `tile`

struct Tile {                        // one output tile of the group
  const bf16_t* dy;                  // dY + n0
  const bf16_t* x;                   // X + k0
  float* dw;
  int N, K, n0, k0, inter;
};

struct Sched {
  int ntiles, nsteps;                // output tiles of the group, K-steps (64 token rows) of the token range
  int P, nmain, main_len;            // pieces; P * ntiles one-item workgroups over K-steps [0, main_len)
  int tail_len, tail_u;              // K-steps [main_len, nsteps): tail_u (tile, K-step) units per tail workgroup
  int dbg;
  float* ws;                         // partial slabs [contributor][tile][256][256] fp32 (nullptr: float atomics into dW)
};

// contributor index of logical workgroup L for output tile `tile`: pieces 0..P-1, then the tail workgroups that touch
// the tile in ascending id (the first of them is the one whose unit range contains the tile's first tail unit)
__device__ __host__ __forceinline__ int contributor_of(const Sched& s, int L, int tile) {
  if (L < s.nmain) return L / s.ntiles;
  return s.P + (L - s.nmain) - (tile * s.tail_len) / s.tail_u;
}
// number of slabs tile `tile` owns (main pieces that are empty -- fewer K-steps than pieces -- write nothing and are
// skipped by the reduction through the same sb < se test)
__device__ __host__ __forceinline__ int tail_contributors(const Sched& s, int tile) {
  if (s.tail_len <= 0) return 0;
  return ((tile + 1) * s.tail_len - 1) / s.tail_u - (tile * s.tail_len) / s.tail_u + 1;
}

// item `it` of logical workgroup L
__device__ __forceinline__ bool item_at(const Sched& s, int L, int it, Item& o) {
  if (L < s.nmain) {
    if (it > 0) return false;
    const int p = L / s.ntiles;
    o.tile = L - p * s.ntiles;
    o.sb = p * s.main_len / s.P;                     // < 2^31: P <= 256, main_len < 2^22 (launcher)
    o.se = (p + 1) * s.main_len / s.P;
    return o.sb < o.se;
  }
  if (s.tail_len <= 0) return false;
  const int U = s.ntiles * s.tail_len;
  const int ub = (L - s.nmain) * s.tail_u, ue = min(U, ub + s.tail_u);
  if (ub >= ue) return false;
  const int tile = ub / s.tail_len + it;
  const int lo = max(ub, tile * s.tail_len), hi = min(ue, (tile + 1) * s.tail_len);
  if (lo >= hi) return false;
  o.tile = tile;
  o.sb = s.main_len + lo - tile * s.tail_len;
  o.se = s.main_len + hi - tile * s.tail_len;
  return true;
}

__device__ __forceinline__ Tile decode(const TnGroup& g, int tile) {
  int p = 0;
#pragma unroll
  for (int q = 0; q < SNX_TN_MAX_GROUP - 1; ++q)
    if (q + 1 < g.nprob && tile >= g.tile_end[q]) p = q + 1;
  if (p > 0) tile -= g.tile_end[p - 1];
  Tile t;
  t.N = g.N[p]; t.K = g.K[p]; t.inter = g.inter[p];
  const int tk = (t.K + TK - 1) / TK;
  t.n0 = (tile / tk) * 256;
  t.k0 = (tile % tk) * TK;
  t.dy = g.dY[p] + t.n0;
  t.x = g.X[p] + t.k0;
  t.dw = g.dW[p];
  return t;
}

// Per-lane byte offsets of a wave's DMA instructions into a 32 x 256 operand slice: (token row) * ld + column from
// the half-step's first token row and the tile's first column.  Wave w fills rows 8w .. 8w+7 of every sub-tile:
// instruction k (0, 1) rows 8w + 4k .. +3; sub-tile 1 is sub-tile 0 with the base 128 columns further -- or the
// SAME base where the tile's second half lies outside the matrix (K = 1152 -> 4.5 tiles): what lands there only
// feeds accumulators that are never written back.
struct LaneOff { unsigned a[2], b[2]; };

__device__ __forceinline__ LaneOff lane_offsets(const Tile& t, int wave, int lane) {
  LaneOff o;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int row = wave * 8 + k * 4 + (lane >> 4);
    const int col = tn_chunk(row, lane & 15) * 8;    // first of the 8 columns held at this lane's slot
    o.a[k] = (unsigned)(row * t.N + col) * 2u;
    o.b[k] = (unsigned)(row * t.K + col) * 2u;
  }
  return o;
}

// DMA instruction k8 (0..7; pair k8 >> 1: dY then X) of the half-step starting at token tok0 into `slot`: rows 8w + 4(k >> 1) .. +3 of
// sub-tile (k & 1) of the dY slice and of the X slice.  8 DMA instructions per wave and half-step, no branches.  The
// launcher hands this kernel whole K-steps only (M % 64 == 0; the ragged rest goes to the 128x128 kernel).
// AUX: cache-policy bits of the LDS-DMA (0 default, 2 = nt: "stream_nt" bit 16 -- every operand of a weight-gradient GEMM is
// read for the last time; alone the nt form measured level, 489 vs 482 us per layer group, the question in the step is
// what its 0.7 GB per launch displace from the caches the launch stream's kernels live on)
template <int AUX>
__device__ __forceinline__ void request_one(const Tile& t, const LaneOff& o, int tok0, char* slot, int wave, int k8) {
  const int k = k8 >> 1;                              // pair index: rows 8 wave + 4 (k >> 1) .., sub-tile k & 1
  char* d = slot + (wave * 8 + (k >> 1) * 4) * 256 + (k & 1) * SUB;
  // (default cache policy: nt loads measured no faster here, 489 vs 482 us per layer group)
  if (!(k8 & 1)) {
    const int a2 = ((k & 1) && t.n0 + 128 < t.N) ? 256 : 0;
    const char* ba = (const char*)(t.dy + (long)tok0 * t.N) + a2;
    __builtin_amdgcn_global_load_lds(GLB_PTR(ba + o.a[k >> 1]), LDS_PTR(d), 16, 0, AUX);
  } else {
    const int b2 = ((k & 1) && t.k0 + 128 < t.K) ? 256 : 0;
    const char* bb = (const char*)(t.x + (long)tok0 * t.K) + b2;
    __builtin_amdgcn_global_load_lds(GLB_PTR(bb + o.b[k >> 1]), LDS_PTR(d + PART), 16, 0, AUX);
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

typedef __attribute__((ext_vector_type(16))) float f32x16;
struct Frags { bf16x8 a[8], b[2 * NJ]; };   // [32-wide tile][16-token half] -> index 2 * tile + half

// tn_frag (gemm_tn.h) with the two transposing reads as volatile asm: hipcc's waitcnt pass puts s_waitcnt vmcnt(0)
// in front of every LDS read it can see while LDS-DMA writes are in flight (it cannot tell the ring slots apart),
// which would drain the whole request pipeline once per half-step.  The results are first used behind the
// s_waitcnt lgkmcnt(0) that ends the half-step.
// Operand fragment of v_mfma_f32_32x32x16_bf16: lane l holds, for MFMA row / column (l & 31) of the 32-wide tile at
// column `cbase` of the sub-tile, the 8 contraction elements (token rows) mb + 8 (l >> 5) .. + 7.
__device__ __forceinline__ bf16x8 tn_frag_opaque(const char* tile, int mb, int cbase, int lane) {
  const int tq = (lane & 15) >> 2, tp = lane & 3;
  const int ch = ((cbase + (lane & 16)) >> 3) + (tp >> 1);
  const int r0 = mb + 8 * (lane >> 5) + tq, r1 = r0 + 4;
  const unsigned a0 = (unsigned)(uintptr_t)LDS_PTR(tile + r0 * 256 + tn_chunk(r0, ch) * 16 + (tp & 1) * 8);
  const unsigned a1 = (unsigned)(uintptr_t)LDS_PTR(tile + r1 * 256 + tn_chunk(r1, ch) * 16 + (tp & 1) * 8);
  bf16x4 v0, v1;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v0) : "v"(a0));
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v1) : "v"(a1));
  return (bf16x8){v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
}

// fragment q of the half-step in `slot`: q = 0..7 dY (32-row tile q >> 1, token rows 16 (q & 1) ..), q = 8..8+2*NJ-1 X
__device__ __forceinline__ void read_frag(Frags& f, int q, const char* slot, int wm, int wn, int lane) {
  if (q < 8) f.a[q] = tn_frag_opaque(slot + wm * SUB, (q & 1) * 16, (q >> 1) * 32, lane);
  else f.b[q - 8] = tn_frag_opaque(slot + PART + wn * SUB, (q & 1) * 16, ((q - 8) >> 1) * 32, lane);   // X columns 128 wn + ...
}

// The accumulators are pinned to AGPRs through the asm constraint (left to itself hipcc moves them to VGPRs and
// scratch as soon as the flush reads them), and a volatile asm is a barrier for memory instructions: the LDS reads
// and DMA requests written between two of these stay there -- the interleaving in the K loop is the source order.
__device__ __forceinline__ void mfma_pinned(f32x16& c, const bf16x8& a, const bf16x8& b) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
// first half-step of a (tile, token piece): C = 0, so the accumulators are never written by anything but an MFMA
__device__ __forceinline__ void mfma_pinned_first(f32x16& c, const bf16x8& a, const bf16x8& b) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=a"(c) : "v"(a), "v"(b));
}

}  // namespace

#ifdef SNX_GEMM_TRACE
// diagnostics build (-DSNX_GEMM_TRACE): per workgroup shader-clock and constant-clock (100 MHz) ticks around the K
// loop and the number of K-steps, for the in-kernel clock and the cycles per half-step (tools/gpu_tnbench.py)
__device__ unsigned long long* g_tn256_trace = nullptr;
extern "C" int snx_tn256_trace_set(void* buf) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_tn256_trace), &buf, sizeof(buf));
}
#endif

// SLAB: the flush stores the partial tile into the workspace (ordered reduction); otherwise float atomics into dW.  A
// template parameter, not a branch: with both flush forms in one kernel hipcc spills 660 bytes per lane.
template <bool NODMA, bool SLAB, int AUX = 0>
__global__ __launch_bounds__(256) void gemm_tn256_kernel(TnGroup grp, int M, Sched sch) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int L = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);   // XCD-contiguous logical id

  // ---- two streams over the same item sequence (items count K-steps of 64 tokens = two half-steps): requests
  //      (ld_*) run RING half-steps ahead of the MFMAs (cp_*) ----
  // (item_at divides: it runs once per item, never inside the K loop)
  Item ld_item, cp_item, ld_next, cp_next;
  if (!item_at(sch, L, 0, ld_item)) return;
  int ld_it = 0, cp_it = 0;
  cp_item = ld_item;
  bool ld_has_next = item_at(sch, L, 1, ld_next), cp_has_next = ld_has_next;
  cp_next = ld_next;
  Tile ld_tile = decode(grp, ld_item.tile), cp_tile = ld_tile;
  LaneOff ld_off = lane_offsets(ld_tile, wave, lane);
  int ld_s = ld_item.sb, cp_s = cp_item.sb;
  int ld_slot = 0;
  // DMA instruction k8 (0..7) of one half-step of the request stream (see request_one); after the eighth the stream's
  // slot moves on.  ONE per MFMA gap: the issue of an LDS-DMA instruction holds the wave for 60-70 cycles, of which
  // the MFMA in flight covers 32.
  auto issue = [&](int half, int k8) {
    if (!NODMA)
      request_one<AUX>(ld_tile, ld_off, (sch.dbg & 4) ? 0 : ld_s * 64 + half * HS, smem + ld_slot * SLOT, wave, k8);
    if (k8 == 7) ld_slot = ld_slot + 1 == RING ? 0 : ld_slot + 1;
  };
  auto issue_all = [&](int half) {
#pragma unroll
    for (int k = 0; k < 8; ++k) issue(half, k);
  };
  // ... and the stream moves on to its next K-step.  At its end it PARKS on its last K-step: the K loop keeps
  // requesting (no branch in its body); a request always goes into the ring slot whose fragments are already in
  // registers, so re-requesting old rows there is harmless.
  auto ld_advance = [&]() {
    if (ld_s + 1 < ld_item.se) {
      ++ld_s;
    } else if (ld_has_next) {
      ld_item = ld_next;
      ld_has_next = item_at(sch, L, ++ld_it + 1, ld_next);
      ld_tile = decode(grp, ld_item.tile);
      ld_off = lane_offsets(ld_tile, wave, lane);
      ld_s = ld_item.sb;
    }
  };

  f32x16 acc[4][NJ];

  // prologue: half-steps 0..4 requested (K-steps 0, 1 and the first half of 2); fragments of half-step 0 in registers
  issue_all(0); issue_all(1);
  ld_advance();
  issue_all(0); issue_all(1);
  ld_advance();
  issue_all(0);
  asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
  BARRIER();
  Frags f0, f1;
#pragma unroll
  for (int q = 0; q < 8 + 2 * NJ; ++q) read_frag(f0, q, smem, wm, wn, lane);
  WAIT_LGKM0();
  int rd_slot = 1;                                    // slot of the half-step whose fragments are read next

  auto next_rd = [&]() { rd_slot = rd_slot + 1 == RING ? 0 : rd_slot + 1; };
  // One half-step: 8 * NJ MFMAs (32x32x16: 4 x NJ tiles, two 16-token halves) on `cur`; between them (source order =
  // issue order, see mfma_pinned) the 8 DMA instructions of the request stream behind MFMAs 0..7 and the 8 + 2 NJ
  // fragments of the next half-step (2 ds_read_b64_tr_b16 each) behind MFMAs 8...  A 32x32x16 MFMA occupies the
  // matrix pipe for 32 cycles and the wave's issue for 8: 24 cycles per gap for other instructions (with 16x16x32
  // MFMAs, 8 cycles per gap, a half-step took 1312 instead of 768 cycles).  The body has NO branch (one s_cbranch
  // per gap cost the single wave of a SIMD 64 cycles per MFMA instead of 32): requests and reads are unconditional
  // -- past the end of the streams they re-request parked rows and read fragments nobody uses.
  // PRE MFMAs go in front of the barrier: they need neither the new data nor a free slot, and the wait runs under them
#define PRE 4
#define HALF_STEP(cur, nxt, half, FIRST)                                                        \
  do {                                                                                          \
    const char* rs = smem + rd_slot * SLOT;                                                     \
    _Pragma("unroll") for (int m = 0; m < 8 * NJ; ++m) {                                        \
      const int h = m / (4 * NJ), i = (m / NJ) & 3, j = m % NJ;   /* token half, dY tile, X tile */ \
      if (m == PRE) {                                                                           \
        WAIT_VM(24);                                                                            \
        BARRIER();                                                                              \
      }                                                                                         \
      if (FIRST && h == 0) mfma_pinned_first(acc[i][j], cur.a[2 * i], cur.b[2 * j]);            \
      else mfma_pinned(acc[i][j], cur.a[2 * i + h], cur.b[2 * j + h]);                          \
      if (m >= PRE && m < PRE + 8) issue(half, m - PRE);                                        \
      if (m >= PRE + 8 && m < PRE + 8 + 8 + 2 * NJ) read_frag(nxt, m - PRE - 8, rs, wm, wn, lane); \
    }                                                                                           \
    WAIT_LGKM0();                                                                               \
    next_rd();                                                                                  \
  } while (0)

#ifdef SNX_GEMM_TRACE
  const unsigned long long tr_c0 = __builtin_amdgcn_s_memtime(), tr_r0 = __builtin_amdgcn_s_memrealtime();
  int tr_steps = 0;
#endif
  bool first = true;                                  // first K-step of a (tile, token piece)
  while (true) {                                      // one K-step of the MFMA stream per iteration
#ifdef SNX_GEMM_TRACE
    ++tr_steps;
#endif
    // the request stream stands at the second half of a K-step
    if (first) HALF_STEP(f0, f1, 1, true);
    else HALF_STEP(f0, f1, 1, false);
    first = false;
    ld_advance();
    HALF_STEP(f1, f0, 0, false);
    // ---- end of a K-step; at the end of a (tile, token piece) the accumulators are added to the gradient
    //   acc[i][j][v] = dW[n0 + 128 wm + 32 i + 8 (v >> 2) + 4 (lane >> 5) + (v & 3)][k0 + 128 wn + 32 j + (lane & 31)]
    if (++cp_s == cp_item.se) {
      asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");   // MFMA results -> VALU reads: the asm MFMAs are opaque to the hazard pass
      const bool whole = cp_tile.n0 + 256 <= cp_tile.N && cp_tile.k0 + TK <= cp_tile.K;
      if (!(sch.dbg & 1)) {
        if (SLAB) {
          // plain stores of the whole 256 x 256 partial (columns / rows outside the matrix included: they stay inside the
          // slab and the reduction never reads them) in the tile's natural [n][k] order
          // Addressing: ONE per-lane 32-bit offset for all 64 stores + a wave-uniform (scalar) row base per store, so that
          // the flush needs no 64-bit address registers: the fragments of the next item are already live in VGPRs here,
          // and per-store address pairs made hipcc spill them to scratch around the flush.
          float* slab = sch.ws + ((size_t)contributor_of(sch, L, cp_item.tile) * sch.ntiles + cp_item.tile) * (256 * 256) +
                        (wm * 128) * 256 + wn * 128;
          const unsigned lo = (unsigned)(4 * (lane >> 5) * 256 + (lane & 31));
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
              // The accumulator element is read out of its AGPR by hand, one volatile asm per store: left to itself hipcc
              // copies whole 16-register accumulator tuples into VGPRs ahead of the stores (several tuples at once, whatever
              // the stores' order or volatility) and spills ~300 bytes per lane around the flush.
              float* row = slab + (i * 32 + 8 * (v >> 2) + (v & 3)) * 256;   // uniform
#pragma unroll
              for (int j = 0; j < NJ; ++j) {
                float x;
                asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(acc[i][j][v]));
                row[j * 32 + lo] = x;
              }
            }
        } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int v = 0; v < 16; ++v) {
            int n = cp_tile.n0 + wm * 128 + i * 32 + 8 * (v >> 2) + 4 * (lane >> 5) + (v & 3);
            const bool nok = n < cp_tile.N;
            if (cp_tile.inter > 0)                    // dY columns are in the interleaved GeGLU order
              n = ((n & 63) < 32) ? 32 * (n >> 6) + (n & 31) : cp_tile.inter + 32 * (n >> 6) + (n & 31);
            const int k = cp_tile.k0 + wn * 128 + (lane & 31);
            float* row = cp_tile.dw + (long)n * cp_tile.K + k;
            if (whole) {                              // wave-uniform: no per-element predicates in the common case
#pragma unroll
              for (int j = 0; j < NJ; ++j) atomicAdd(row + j * 32, acc[i][j][v]);
            } else {                                  // half a tile outside the matrix (N or K = 128 mod 256)
#pragma unroll
              for (int j = 0; j < NJ; ++j)
                if (nok && k + j * 32 < cp_tile.K) atomicAdd(row + j * 32, acc[i][j][v]);
            }
          }
        }
      }
      if (!cp_has_next) break;
      first = true;
      cp_item = cp_next;
      cp_has_next = item_at(sch, L, ++cp_it + 1, cp_next);
      cp_tile = decode(grp, cp_item.tile);
      cp_s = cp_item.sb;
    }
  }
#undef HALF_STEP
#undef PRE
#ifdef SNX_GEMM_TRACE
  if (threadIdx.x == 0 && g_tn256_trace) {
    unsigned long long* o = g_tn256_trace + 4l * blockIdx.x;
    o[0] = __builtin_amdgcn_s_memtime() - tr_c0;
    o[1] = __builtin_amdgcn_s_memrealtime() - tr_r0;
    o[2] = tr_steps;
    o[3] = L;
  }
#endif
}

// ---- ordered reduction of the partial slabs -------------------------------------------------------------------------
// dW tile += slab(0) + slab(1) + ... in contributor order (see contributor_of).  One workgroup per (tile, 16-row
// chunk); a thread owns 4 consecutive k of 4 rows: every access is a whole 16-byte piece of a 1-KiB row.  The slabs were
// written a moment ago by the GEMM kernel (the group's 60-100 MB sit in the 256-MiB Infinity Cache).  (32-row chunks:
// 624 workgroups for the layer group = 2.4 per CU, three rounds of ~10 us; 16-row chunks: 1,248.)
// NT ("stream_nt" bit 256): the slabs are read for the last time and the gradient tile is not read again before the
// optimizer -- non-temporal accesses keep both out of the way of the launch stream's operands.
template <bool NT>
__global__ __launch_bounds__(256) void tn256_reduce_kernel(TnGroup grp, Sched sch) {
  const int tile = blockIdx.x, chunk = blockIdx.y;
  const Tile t = decode(grp, tile);
  const int col = (threadIdx.x & 63) * 4;
  const int k = t.k0 + col;
  if (k >= t.K) return;                               // K % 128 == 0: a 16-byte piece lies wholly inside or outside
  const int ntail = tail_contributors(sch, tile);
  const size_t tile_stride = (size_t)sch.ntiles * (256 * 256);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int nl = chunk * 16 + q * 4 + (threadIdx.x >> 6);
    int n = t.n0 + nl;
    if (n >= t.N) continue;
    if (t.inter > 0) n = ((n & 63) < 32) ? 32 * (n >> 6) + (n & 31) : t.inter + 32 * (n >> 6) + (n & 31);
    float* dst = t.dw + (long)n * t.K + k;
    auto ld = [](const float* p_) __attribute__((always_inline)) {
      return NT ? __builtin_nontemporal_load((const f32x4*)p_) : *(const f32x4*)p_;
    };
    f32x4 a = ld(dst);
    const float* src = sch.ws + (size_t)tile * (256 * 256) + nl * 256 + col;
    for (int p = 0; p < sch.P; ++p) {
      if (p * sch.main_len / sch.P < (p + 1) * sch.main_len / sch.P) a += ld(src + p * tile_stride);
    }
    for (int c = 0; c < ntail; ++c) a += ld(src + (sch.P + c) * tile_stride);
    if (NT) __builtin_nontemporal_store(a, (f32x4*)dst);
    else *(f32x4*)dst = a;
  }
}

// CUs left to other kernels (RCCL's channel workgroups while a gradient bucket is being exchanged): this kernel's
// workgroups take a whole CU each (160 KiB of LDS, one 512-register wave per SIMD), so with fewer than 256 free CUs
// a 256-workgroup launch would run its last workgroups as a second wave -- twice the time.  The schedule below
// balances any workgroup count (one long item per workgroup + a stream-K tail), so the launch simply shrinks -- and
// with it the token partition: another fp32 summation tree, i.e. last-bit differences against the 256-workgroup result.
static int g_reserved_cus = 0;
extern "C" int snx_set_reserved_cus(int32_t n) {
  if (n < 0 || n > 128) return SNX_E_ARG;
  g_reserved_cus = (n + 7) & ~7;                      // whole rounds of the 8 XCDs (the logical-id map needs nwg % 8 == 0)
  return SNX_OK;
}
extern "C" int snx_get_reserved_cus() { return g_reserved_cus; }

// schedule of a group over M token rows on `nwg` workgroups; false: the kernel does not take it
static bool tn256_schedule(TnGroup& g, int M, int nwg, Sched& s, int& max_slabs) {
  int run = 0;
  for (int p = 0; p < g.nprob; ++p) {
    run += cdiv(g.N[p], 256) * cdiv(g.K[p], TK);
    g.tile_end[p] = run;
  }
  if (run > nwg || M >= (1 << 27)) return false;
  s.ntiles = run;
  s.nsteps = cdiv(M, 2 * HS);
  s.P = nwg / run;
  s.nmain = s.P * run;
  const int wt = nwg - s.nmain;
  // every workgroup should multiply about ntiles * nsteps / 256 half-steps; the tail workgroups flush once per tile
  // they touch, which SNX_TN256_TAIL_PCT (default 95) takes off their share
  const int tail_pct = SNX_DIAG_CFG(tn256_tail_pct, 95);
  s.tail_len = wt > 0 ? (int)((long)s.nsteps * wt * tail_pct / (100L * nwg)) : 0;
  s.main_len = s.nsteps - s.tail_len;
  s.tail_u = wt > 0 ? cdiv((long)s.ntiles * s.tail_len, wt) : 0;
  if (s.tail_u <= 0) s.tail_len = 0, s.main_len = s.nsteps;
  s.dbg = 0;
  s.ws = nullptr;
  int mt = 0;
  for (int t = 0; t < s.ntiles; ++t) mt = max(mt, tail_contributors(s, t));
  max_slabs = s.P + mt;
  return true;
}

// bytes of partial slabs the ordered reduction of this group needs (0: the 128x128 kernel takes the group)
size_t snx_tn256_ws_bytes(const TnGroup& g128, int M) {
  TnGroup g = g128;
  Sched s;
  int slabs = 0;
  if (!tn256_schedule(g, M, ::NWG - g_reserved_cus, s, slabs)) return 0;
  return (size_t)slabs * s.ntiles * (256 * 256 * 4);
}

size_t snx_tn256_ws_bound(const TnGroup& g128, int M) {
  size_t need = 0;
  if (M < 64) return 0;
  for (int r = 0; r <= 128; r += 8) {
    TnGroup g = g128;
    Sched s;
    int slabs = 0;
    if (tn256_schedule(g, M, ::NWG - r, s, slabs)) need = max(need, (size_t)slabs * s.ntiles * (256 * 256 * 4));
  }
  return need;
}

int snx_launch_tn256(const TnGroup& g128, int M, void* ws, size_t ws_bytes, hipStream_t st) {
  const int NWG = ::NWG - g_reserved_cus;
  TnGroup g = g128;
  Sched s;
  int slabs = 0;
  if (!tn256_schedule(g, M, NWG, s, slabs)) return SNX_E_SHAPE;   // caller falls back to the 128x128 kernel
  // diagnostics: 1 = no flush, 2 = no DMA, 4 = L2-resident operands
  const int dbg = SNX_DIAG_CFG(tn256_dbg, 0);
  s.dbg = dbg;
  if (g_snx_cfg.det_reduce) {
    if (!ws || ws_bytes < (size_t)slabs * s.ntiles * (256 * 256 * 4)) return SNX_E_ARG;
    s.ws = (float*)ws;
  }
  static LdsOptIn optin[5];
  void (*kern)(TnGroup, int, Sched);
  int which;
  if (s.ws) { which = (dbg & 2) ? 3 : 2; kern = (dbg & 2) ? gemm_tn256_kernel<true, true> : gemm_tn256_kernel<false, true>; }
  else { which = (dbg & 2) ? 1 : 0; kern = (dbg & 2) ? gemm_tn256_kernel<true, false> : gemm_tn256_kernel<false, false>; }
  if (which == 2 && (g_snx_cfg.stream_nt & 16)) { which = 4; kern = gemm_tn256_kernel<false, true, 2>; }
  if (const int rc = optin[which].ensure((const void*)kern, RING * SLOT)) return rc;
  hipLaunchKernelGGL(kern, dim3(NWG), dim3(256), RING * SLOT, st, g, M, s);
  SNX_CHECK_LAUNCH();
  if (s.ws && !(dbg & 1)) {
    if (g_snx_cfg.stream_nt & 256) hipLaunchKernelGGL(tn256_reduce_kernel<true>, dim3(s.ntiles, 16), dim3(256), 0, st, g, s);
    else hipLaunchKernelGGL(tn256_reduce_kernel<false>, dim3(s.ntiles, 16), dim3(256), 0, st, g, s);
    SNX_CHECK_LAUNCH();
  }
  return SNX_OK;
}
