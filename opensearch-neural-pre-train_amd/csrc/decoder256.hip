// Fused MLM-decoder GEMM + SPLADE tail, 256x192 persistent form (same contract as decoder_splade_kernel in
// splade_head.hip: hf modeling_modernbert.py:550 + ref:src/model/splade_modern.py:76-86).
//
//   key[b, v]   = max over the valid rows s of sequence b of  bf16bits(relu(Hd[b,s,:] . W[v,:] + bias[v])) << 16 | 0xFFFF - s
//   sparse[b,v] = log1p(value(key));   rowpart[2 nt + wn, t] = max over the 96 columns of half tile (nt, wn) of the value bits
//
// The 128x128 kernel (one workgroup per (sequence, vocab tile), two per CU) runs at 0.27 of the MFMA peak: K = 768
// gives every 128-row chunk a pipeline start-up of its own, and the operand tiles cross the L2 -> LDS path at 65
// FLOP per byte.  This form reuses the main loop of gemm_tn256.hip (one 4-wave workgroup per CU, one wave per SIMD,
// accumulators of v_mfma_f32_32x32x16_bf16 pinned to AGPRs, a five-slot LDS-DMA ring that never drains between
// tiles, no branch inside the K loop) for operands that are both K-contiguous: a half-step is 256 rows x 32 k of Hd
// and 192 rows x 32 k of W_E (28 KiB), rows of 64 B in LDS with the 16-B chunk index XOR-swizzled by (row >> 2) & 3
// (conflict-free ds_read_b128 in the 32x32x16 operand pattern).  Wave tile 128 x 96 = 4 x 3 accumulator tiles: 192
// of the 256 AGPRs; with all 256 taken hipcc spills inside the K loop, and a spill store right behind one of the asm
// LDS reads saves the register before its data has arrived (seen as intermittently wrong rows).
//
// Rows.  A tile's 256 rows are eight SUB-TILES of 32 rows, each inside ONE sequence: a pre-pass compacts the valid
// rows of every sequence (any mask) into a list, cuts it into sub-tiles and pads the last one of a sequence by
// repeating its last valid row -- a repeated row produces the same value at a later list position and loses every
// tie, so the epilogue needs no row mask at all.  Keys are built on list positions; the finalize pass turns them
// into sequence positions (identical for all-ones and right-padded masks) while it computes sparse = log1p(value).
// Sequences meet in the key array through atomicMax (a sequence may span several tiles and both wave rows).
//
// Tiles are dealt in blocks of 4 (rows) x 8 (vocab) to the 32 workgroups of an XCD (blockIdx & 7), which walk K
// in step: the 12 operand panels of a block are shared through that XCD's L2.  Round 3: every XCD owns a contiguous
// run of the blocks in COLUMN-major order, so that its consecutive steps keep the same 8 vocabulary tiles (2.4 MB of
// W_E, resident in its 4 MiB L2) and only stream new rows of Hd: 2.27 GB of fabric reads per launch of the bench's 192
// sequences instead of 4.85 (blocks dealt round-robin in row-major order, -DSNX_DEC256_ROUND_ROBIN; the 128x128
// kernel: 5.7 GB) against 25 GB of LDS-DMA traffic; the time is the same (3.39 ms: the L2 -> LDS path never was the
// bound).
//
// Measured (64 x 64 + 128 x 256 tokens, V = 50,000, in-kernel stamps of a -DSNX_GEMM_TRACE build, tools/gpu_decbench.py):
// 3.45 ms = 822 TFLOP/s (128x128 kernel: 4.16 ms); per tile 46.9k cycles = 24 half-steps of 1,514 (768 of them MFMA; 1,042
// without the DMA instructions, whose issue holds the single wave of a SIMD 60-70 cycles each) + 10.5k of epilogue.
// A variant with 128-B rows (K = 64 per slot, two slots) was no faster: the L2 -> LDS path is not what bounds it.
#include "common.h"
#include "snx.h"

namespace {

constexpr int HS = 32;               // k per half-step
constexpr int SUBT = 128 * 64;       // LDS sub-tile of Hd: 128 rows x 64 B
constexpr int PART = 2 * SUBT;       // Hd slice: 256 rows x 32 k
#ifdef SNX_DEC256_OVERSUBSCRIBE     // compile-only variant for the build guard's self-test (snx/asmcheck.py, tests/
constexpr int NJ = 4;                // test_build_guard.py): all 256 AGPRs as accumulators -> hipcc spills fragment registers
#else                                // behind the asm LDS reads; the launcher refuses to run it
constexpr int NJ = 3;
#endif
// NJ: 32-column accumulator tiles per wave along the vocabulary: wave tile 128 x 96
constexpr int TV = 64 * NJ;          // vocabulary columns of a tile; 192, not 256: 64 AGPRs stay free, which hipcc needs
                                     // as spill space (with all 256 taken it spills to scratch inside the K loop)
constexpr int SUBW = 32 * NJ * 64;   // LDS sub-tile of W_E: 96 rows x 64 B
constexpr int SLOT = PART + 2 * SUBW;   // Hd slice + W slice = 28 KiB
constexpr int RING = 5;
constexpr int NWG = 256;

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) int i32x4;

struct Frags { bf16x8 a[8], b[2 * NJ]; };   // [32-row tile][16-k half] -> index 2 * tile + half

// ---------------------------------------------------------------------------------------------------------------
// pre-pass 1: one wave per sequence compacts its valid rows.  list[cu[s] + p] = position in the sequence of its p-th
// valid row; nvalid[s] = their number.
__global__ void dec256_rows_kernel(const int32_t* __restrict__ cu, const int64_t* __restrict__ mask,
                                   int32_t* __restrict__ list, int32_t* __restrict__ nvalid, int nseq) {
  const int s = blockIdx.x, lane = threadIdx.x;
  if (s >= nseq) return;
  const int s0 = cu[s], len = cu[s + 1] - s0;
  int base = 0;
  for (int c = 0; c < len; c += 64) {
    const int r = c + lane;
    const bool ok = r < len && mask[s0 + r] != 0;
    const unsigned long long b = __ballot(ok);
    if (ok) list[s0 + base + __popcll(b & ((1ull << lane) - 1ull))] = r;
    base += __popcll(b);
  }
  if (lane == 0) nvalid[s] = base;
}

// pre-pass 2: one workgroup cuts the lists into 32-row sub-tiles.  subtab[k] = {sequence, cu[sequence], first list
// position, rows (1..32)}; hdr[0] = number of sub-tiles.
__global__ __launch_bounds__(256) void dec256_tiles_kernel(const int32_t* __restrict__ cu,
                                                           const int32_t* __restrict__ nvalid,
                                                           i32x4* __restrict__ subtab, int32_t* __restrict__ hdr,
                                                           int nseq) {
  __shared__ int part[256];
  const int t = threadIdx.x;
  const int per = (nseq + 255) / 256;
  const int b = t * per, e = min(nseq, b + per);
  int mine = 0;
  for (int s = b; s < e; ++s) mine += (nvalid[s] + 31) >> 5;
  part[t] = mine;
  __syncthreads();
  if (t == 0) {
    int run = 0;
    for (int i = 0; i < 256; ++i) {
      const int v = part[i];
      part[i] = run;
      run += v;
    }
    hdr[0] = run;
  }
  __syncthreads();
  int k = part[t];
  for (int s = b; s < e; ++s) {
    const int nv = nvalid[s], s0 = cu[s];
    for (int p0 = 0; p0 < nv; p0 += 32) subtab[k++] = (i32x4){s, s0, p0, min(32, nv - p0)};
  }
}

// post-pass: list positions -> sequence positions in the keys, sparse = log1p(value)
__global__ void dec256_finalize_kernel(uint32_t* __restrict__ keys, float* __restrict__ sparse,
                                       const int32_t* __restrict__ cu, const int32_t* __restrict__ list, int V,
                                       long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int s = (int)(i / V);
  uint32_t k = keys[i];
  const uint32_t bits = k >> 16;
  if (bits) {
    const int p = 0xFFFF - (int)(k & 0xFFFFu);
    k = (bits << 16) | (uint32_t)(0xFFFF - list[cu[s] + p]);
  } else {
    k = 0xFFFFu;                                      // value 0 at position 0 (what the 128x128 kernel leaves)
  }
  keys[i] = k;
  sparse[i] = log1pf(bits_to_f32(bits));
}

// ---------------------------------------------------------------------------------------------------------------
struct DecArgs {
  const bf16_t* Hd;                  // [T, K]
  const bf16_t* W;                   // [V, K]
  const float* bias;                 // [V]
  const int32_t* list;               // [T] (pre-pass 1)
  const i32x4* subtab;               // sub-tiles (pre-pass 2)
  const int32_t* hdr;                // hdr[0] = number of sub-tiles
  uint32_t* keys;                    // [nseq, V], zeroed
  unsigned short* rowpart;           // [2 * ceil(V / 192), T]: one row per 96-column half tile
  int T, V, K, ntn;                  // ntn = ceil(V / 192)
};

__device__ __forceinline__ void mfma_pinned(f32x16& c, const bf16x8& a, const bf16x8& b) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_pinned_first(f32x16& c, const bf16x8& a, const bf16x8& b) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=a"(c) : "v"(a), "v"(b));
}

// operand fragment of the 32-row tile `tile` (0..3) of a 128-row sub-tile image: lane l holds row (l & 31), k
// elements 16 half + 8 (l >> 5) .. + 7  (volatile asm: see gemm_tn256.hip)
__device__ __forceinline__ bf16x8 nt_frag(const char* sub, int tile, int half, int lane) {
  const int r = tile * 32 + (lane & 31);
  const int slot = (2 * half + (lane >> 5)) ^ ((r >> 2) & 3);
  const unsigned a = (unsigned)(uintptr_t)LDS_PTR(sub + r * 64 + slot * 16);
  bf16x8 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a));
  return v;
}

__device__ __forceinline__ void read_frag(Frags& f, int q, const char* slot, int wm, int wn, int lane) {
  if (q < 8) f.a[q] = nt_frag(slot + wm * SUBT, q >> 1, q & 1, lane);
  else f.b[q - 8] = nt_frag(slot + PART + wn * SUBW, (q - 8) >> 1, q & 1, lane);
}

#define WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define BARRIER()                          \
  do {                                     \
    __builtin_amdgcn_sched_barrier(0);     \
    __builtin_amdgcn_s_barrier();          \
    __builtin_amdgcn_sched_barrier(0);     \
  } while (0)

struct TilePos { int mt, nt; };

// tile `k` of workgroup (xcd, j): blocks of 4 x 8 tiles, block b = xcd + 8 k; {-1, .} when the block or the tile does
// not exist (the workgroup sits that step out)
__device__ __forceinline__ TilePos tile_at(int k, int xcd, int j, int mtiles, int ntn) {
  const int nbn = (ntn + 7) >> 3, nbm = (mtiles + 3) >> 2;
  const int nb = nbm * nbn;
  TilePos t;
  t.mt = -1; t.nt = 0;
#ifdef SNX_DEC256_ROUND_ROBIN                         // the earlier deal: block xcd + 8 k, row-major
  const int b = xcd + 8 * k;
  if (b >= nb) return t;
  const int bm = b / nbn, bn = b - bm * nbn;
#else
  // W_E-stationary walk: every XCD owns a contiguous run of the blocks in COLUMN-major order (row blocks fastest), so
  // that its consecutive steps keep the same 8 vocabulary tiles (2.4 MB of W_E, resident in its L2) and only stream
  // new rows of Hd
  const int lo = (int)((long)nb * xcd / 8), hi = (int)((long)nb * (xcd + 1) / 8);
  const int b = lo + k;
  if (b >= hi) return t;
  const int bn = b / nbm, bm = b - bn * nbm;
#endif
  const int mt = bm * 4 + (j >> 3), nt = bn * 8 + (j & 7);
  if (mt < mtiles && nt < ntn) { t.mt = mt; t.nt = nt; }
  return t;
}

}  // namespace

#ifdef SNX_GEMM_TRACE
// diagnostics build: per workgroup shader-clock ticks in total / inside the epilogues, constant-clock ticks, tiles
__device__ unsigned long long* g_dec256_trace = nullptr;
extern "C" int snx_dec256_trace_set(void* buf) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_dec256_trace), &buf, sizeof(buf));
}
#endif

__global__ __launch_bounds__(256) void decoder256_kernel(DecArgs g) {
#ifdef SNX_GEMM_TRACE
  const unsigned long long tr_c0 = __builtin_amdgcn_s_memtime(), tr_r0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long tr_epi = 0, tr_tiles = 0;
#endif
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int xcd = blockIdx.x & 7, jw = blockIdx.x >> 3;
  const int nsub = g.hdr[0];
  const int mtiles = (nsub + 7) >> 3;
#ifdef SNX_DEC256_ROUND_ROBIN
  const int nsteps = (((mtiles + 3) >> 2) * ((g.ntn + 7) >> 3) + 7 - xcd) / 8;   // blocks xcd, xcd + 8, ...
#else
  const int nblk = ((mtiles + 3) >> 2) * ((g.ntn + 7) >> 3);
  const int nsteps = (int)((long)nblk * (xcd + 1) / 8) - (int)((long)nblk * xcd / 8);   // this XCD's run of blocks
#endif
  const int nh = g.K / HS;                            // half-steps per tile

  auto next_tile = [&](int from, TilePos& t) {        // first existing tile with step index >= from, or -1
    for (int k = from; k < nsteps; ++k) {
      t = tile_at(k, xcd, jw, mtiles, g.ntn);
      if (t.mt >= 0) return k;
    }
    return -1;
  };

  // Per-lane byte offsets of this wave's 4 + 4 DMA instructions of a half-step, from the matrix base, without the k
  // offset: instruction i fills rows 64 wave + 16 i + (lane >> 2), physical chunk (lane & 3) = logical chunk
  // (lane & 3) ^ ((lane >> 4) & 3).  Hd rows come from the sub-tile table (rows past a sub-tile's end repeat its last
  // row; sub-tiles past the end of the table repeat row 0), W rows are clamped to V - 1.
  auto offsets = [&](const TilePos& t, unsigned (&oa)[4], unsigned (&ob)[4]) {
    int lane = threadIdx.x & 63;                      // opaque copy (see the epilogue)
    asm volatile("" : "+v"(lane));
    const int chunk = ((lane & 3) ^ ((lane >> 4) & 3)) * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int sub = t.mt * 8 + wave * 2 + (i >> 1);          // wave-uniform
      int row = 0;
      if (sub < nsub) {
        const i32x4 e = g.subtab[sub];
        const int off = min(16 * (i & 1) + (lane >> 2), e[3] - 1);
        row = e[1] + g.list[e[1] + e[2] + off];
      }
      oa[i] = (unsigned)row * (unsigned)(g.K * 2) + chunk;
      const int vr = min(t.nt * TV + wave * 48 + (i < 3 ? i : 2) * 16 + (lane >> 2), g.V - 1);
      ob[i] = (unsigned)vr * (unsigned)(g.K * 2) + chunk;   // (three W instructions per wave: ob[3] unused)
    }
  };

  // ---- request stream (runs RING half-steps ahead of the MFMA stream, across tiles) ----
  TilePos ld_tile, cp_tile;
  int ld_k = next_tile(0, ld_tile);
  if (ld_k < 0) return;
  int cp_k = ld_k;
  cp_tile = ld_tile;
  unsigned ld_oa[4], ld_ob[4];
  offsets(ld_tile, ld_oa, ld_ob);
  int ld_h = 0, ld_slot = 0;                          // half-step inside the tile, ring slot
  bool ld_more = true;                                // the stream has not reached its end (then it parks)
  // DMA instruction n (0..6: Hd rows 16 (n >> 1) .. for even n < 7... see below) of the stream's half-step, ONE per
  // MFMA gap (its issue holds the wave for 60-70 cycles, of which the MFMA in flight covers 32): n = 0, 2, 4, 6 are
  // the four Hd instructions, n = 1, 3, 5 the three W_E instructions.
  auto issue = [&](int n) {
    const int i = n >> 1;
#ifdef SNX_DEC256_NODMA                               // diagnostics build: the K loop without its LDS-DMA traffic
    if (n == 6) ld_slot = ld_slot + 1 == RING ? 0 : ld_slot + 1;
    return;
#endif
    char* s0 = smem + ld_slot * SLOT;
    if (!(n & 1)) {
      const char* ba = (const char*)g.Hd + ld_h * 64;
      __builtin_amdgcn_global_load_lds(GLB_PTR(ba + ld_oa[i]),
                                       LDS_PTR(s0 + (wave >> 1) * SUBT + ((wave & 1) * 64 + i * 16) * 64), 16, 0, 0);
    } else {
      const char* bb = (const char*)g.W + ld_h * 64;
      __builtin_amdgcn_global_load_lds(GLB_PTR(bb + ld_ob[i]),
                                       LDS_PTR(s0 + PART + (wave >> 1) * SUBW + ((wave & 1) * 48 + i * 16) * 64), 16, 0, 0);
    }
    if (n == 6) ld_slot = ld_slot + 1 == RING ? 0 : ld_slot + 1;
  };
  auto ld_advance = [&]() {                           // next half-step; at the end of the stream: park on the last one
    if (ld_h + 1 < nh) {
      ++ld_h;
    } else if (ld_more) {
      TilePos t;
      const int k = next_tile(ld_k + 1, t);
      if (k < 0) {
        ld_more = false;
      } else {
        ld_k = k;
        ld_tile = t;
        offsets(ld_tile, ld_oa, ld_ob);
        ld_h = 0;
      }
    }
  };

  f32x16 acc[4][NJ];
  // prologue: half-steps 0..4 requested, fragments of half-step 0 in registers
#pragma unroll 1
  for (int h = 0; h < RING; ++h) {
#pragma unroll
    for (int n = 0; n < 7; ++n) issue(n);
    ld_advance();
  }
  asm volatile("s_waitcnt vmcnt(28)" ::: "memory");
  BARRIER();
  Frags f0, f1;
#pragma unroll
  for (int q = 0; q < 8 + 2 * NJ; ++q) read_frag(f0, q, smem, wm, wn, lane);
  WAIT_LGKM0();
  int rd_slot = 1;
  auto next_rd = [&]() { rd_slot = rd_slot + 1 == RING ? 0 : rd_slot + 1; };

  // one half-step, no branch (gemm_tn256.hip): 24 MFMAs, the barrier behind the first four, the 7 DMA instructions behind
  // MFMAs 4..10, the 14 fragment reads of the next half-step behind MFMAs 11..22
#define HALF_STEP(cur, nxt, FIRST)                                                              \
  do {                                                                                          \
    const char* rs = smem + rd_slot * SLOT;                                                     \
    _Pragma("unroll") for (int m = 0; m < 8 * NJ; ++m) {                                        \
      const int h = m / (4 * NJ), i = (m / NJ) & 3, j = m % NJ;                                 \
      if (m == 4) {              /* four MFMAs in front of the barrier: the wait runs under them */ \
        WAIT_VM(21);                                                                            \
        BARRIER();                                                                              \
      }                                                                                         \
      if (FIRST && h == 0) mfma_pinned_first(acc[i][j], cur.a[2 * i], cur.b[2 * j]);            \
      else mfma_pinned(acc[i][j], cur.a[2 * i + h], cur.b[2 * j + h]);                          \
      if (m >= 4 && m < 11) issue(m - 4);                                                       \
      if (m >= 11 && m < 23) read_frag(nxt, m - 11, rs, wm, wn, lane);                          \
      if (m == 22) {                                                                            \
        read_frag(nxt, 12, rs, wm, wn, lane);                                                   \
        read_frag(nxt, 13, rs, wm, wn, lane);                                                   \
      }                                                                                         \
    }                                                                                           \
    WAIT_LGKM0();                                                                               \
    next_rd();                                                                                  \
    ld_advance();                                                                               \
  } while (0)

  int cp_h = 0;                                       // half-step of the MFMA stream inside its tile (even here)
  // what the epilogue reads from memory (bias, token rows of the row maxima) is requested two half-steps before it:
  // the vm queue returns in order, and behind four half-steps of DMA a load issued in the epilogue itself took
  // about 3,000 cycles to come back
  float pre_bias[NJ] = {0.f, 0.f, 0.f};
  int pre_tok[4] = {-1, -1, -1, -1};
  while (true) {
    if (cp_h + 2 >= nh) {
      int lane = threadIdx.x & 63;                    // opaque copy (see the epilogue)
      asm volatile("" : "+v"(lane));
      const int col0 = cp_tile.nt * TV + wn * 96 + (lane & 31);
#pragma unroll
      for (int j = 0; j < NJ; ++j) pre_bias[j] = col0 + 32 * j < g.V ? g.bias[col0 + 32 * j] : -3.0e38f;
      const int myrr = 8 * ((lane & 15) >> 2) + 4 * (lane >> 5) + (lane & 3);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int sub = cp_tile.mt * 8 + wm * 4 + i;
        pre_tok[i] = -1;
        if (sub < nsub) {
          const i32x4 e = g.subtab[sub];
          if ((lane & 16) && myrr < e[3]) pre_tok[i] = e[1] + g.list[e[1] + e[2] + myrr];
        }
      }
    }
    if (cp_h == 0) HALF_STEP(f0, f1, true);
    else HALF_STEP(f0, f1, false);
    HALF_STEP(f1, f0, false);
    cp_h += 2;
    if (cp_h < nh) continue;
    // =================== end of a tile: SPLADE tail on the 256 x 256 logits in the accumulators ===================
    //   acc[i][j][v] = logit(row 128 wm + 32 i + 8 (v >> 2) + 4 (lane >> 5) + (v & 3), column 96 wn + 32 j + (lane & 31))
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");   // MFMA results -> VALU reads (asm MFMAs are opaque to the hazard pass)
#ifdef SNX_GEMM_TRACE
    const unsigned long long tr_e0 = __builtin_amdgcn_s_memtime();
#endif
    {
      int lane = threadIdx.x & 63;                    // opaque copy: nothing of the lane arithmetic below may be hoisted
      asm volatile("" : "+v"(lane));                  // into the K loop, where every VGPR is taken
      const int hh = lane >> 5;
      const int col0 = cp_tile.nt * TV + wn * 96 + (lane & 31);
      const int t96 = cp_tile.nt * 2 + wn;            // row of the row-maximum array: one per 96-column half tile
      // bias per column; columns past V get a bias that sends every logit to relu's zero
      float bcol[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) bcol[j] = rbf(pre_bias[j]);
      // token rows of the 4 x 32 rows this lane stores row maxima for (lanes 16-31 / 48-63, lane & 15 = q <-> row
      // 8 (q >> 2) + 4 (lane >> 5) + (q & 3) of every sub-tile): requested before the last two half-steps
      int tok[4];
      i32x4 ent[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int sub = cp_tile.mt * 8 + wm * 4 + i;
        ent[i] = sub < nsub ? g.subtab[sub] : (i32x4){-1, 0, 0, 0};
        tok[i] = pre_tok[i];
      }
      // Keys are compared as SIGNED integers from here on: a negative logit is a negative key and loses against the
      // initial 0, which is relu; among non-negative values the order is the unsigned one.  (One v_max_f32 per
      // element less than clamping first; the bias goes on two rows at a time, v_pk_add_f32.)
      int32_t best[NJ] = {0, 0, 0};
      int cur_seq = -1;
      auto flush = [&](int seq) {                     // this wave's column maxima of sequence `seq` -> key array
        if (seq < 0) return;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          int32_t k = best[j];
          const int32_t o = __shfl_xor(k, 32, 64);
          k = o > k ? o : k;
          const int col = col0 + 32 * j;
          if (hh == 0 && col < g.V && (k >> 16)) atomicMax(g.keys + (long)seq * g.V + col, (uint32_t)k);   // k >= 0
          best[j] = 0;
        }
      };
      // maximum over the 32 lanes of each half (DPP inside the rows of 16 lanes, then lane 15 of the even rows
      // into the odd rows): lanes 16-31 / 48-63 end up with it.  old = 0 is the identity of the unsigned maximum,
      // which lets the DPP moves fold into v_max_u32_dpp.
      // (row keys are non-negative: the initial 0 below is in every maximum)
      auto half_max = [](int32_t x) {
        x = max(x, __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
        x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
        x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, true));   // row_half_mirror
        x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, true));   // row_mirror
        x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false));  // row_bcast:15 into rows 1, 3
        return x;
      };
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (ent[i][0] < 0) break;                     // wave-uniform: past the last sub-tile
        if (ent[i][0] != cur_seq) {
          flush(cur_seq);
          cur_seq = ent[i][0];
        }
        const uint32_t tagbase = 0xFFFFu - (uint32_t)(ent[i][2] + 4 * hh);
        int32_t mine = 0;
#pragma unroll
        for (int v = 0; v < 16; v += 2) {             // rows v and v + 1 of this lane: one bf16 pair per column tile
          const uint32_t tag0 = tagbase - (uint32_t)(8 * (v >> 2) + (v & 3)), tag1 = tag0 - 1u;
          int32_t k0[NJ], k1[NJ];
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            const f32x2 sum = (f32x2){acc[i][j][v], acc[i][j][v + 1]} + (f32x2){bcol[j], bcol[j]};
            const bf16x2 pk = (bf16x2){f2bf(sum[0]), f2bf(sum[1])};
            const uint32_t w = __builtin_bit_cast(uint32_t, pk);
            k0[j] = (int32_t)((w << 16) | tag0);      // key = value bits << 16 | 0xFFFF - list position
            k1[j] = (int32_t)((w & 0xFFFF0000u) | tag1);
            best[j] = max(best[j], max(k0[j], k1[j]));   // tag0 > tag1: the earlier row wins a tie
          }
          // a row's keys share their tag, so the row maximum of the keys is (max value bits) << 16 | tag
          const int32_t r0 = half_max(max(max(max(k0[0], k0[1]), k0[2]), 0)), r1 = half_max(max(max(max(k1[0], k1[1]), k1[2]), 0));
          mine = (lane & 15) == v ? r0 : mine;
          mine = (lane & 15) == v + 1 ? r1 : mine;
          if ((v & 6) == 6) __builtin_amdgcn_sched_barrier(0);   // bound the window: four rows' worth of temporaries
        }
        if (tok[i] >= 0) g.rowpart[(long)t96 * g.T + tok[i]] = (unsigned short)(mine >> 16);
      }
      flush(cur_seq);
    }
#ifdef SNX_GEMM_TRACE
    tr_epi += __builtin_amdgcn_s_memtime() - tr_e0;
    ++tr_tiles;
#endif
    // next tile of the MFMA stream
    {
      TilePos t;
      const int k = next_tile(cp_k + 1, t);
      if (k < 0) break;
      cp_k = k;
      cp_tile = t;
      cp_h = 0;
    }
    // The fragments of the next tile's first half-step are read again here, so that no fragment register is live
    // across the epilogue: hipcc would spill them, and a spill store placed right behind one of the asm LDS reads
    // (whose latency it cannot see) saves the register before the data has arrived.
    {
      const char* rs = smem + (rd_slot == 0 ? RING - 1 : rd_slot - 1) * SLOT;
#pragma unroll
      for (int q = 0; q < 8 + 2 * NJ; ++q) read_frag(f0, q, rs, wm, wn, lane);
      WAIT_LGKM0();
    }
  }
#undef HALF_STEP
#ifdef SNX_GEMM_TRACE
  if (threadIdx.x == 0 && g_dec256_trace) {
    unsigned long long* o = g_dec256_trace + 4l * blockIdx.x;
    o[0] = __builtin_amdgcn_s_memtime() - tr_c0;
    o[1] = __builtin_amdgcn_s_memrealtime() - tr_r0;
    o[2] = tr_epi;
    o[3] = tr_tiles;
  }
#endif
}

// sizes of the pre-pass tables behind the row maxima in the scratch buffer (snx_splade_head_scratch_bytes adds them)
int snx_dec256_rowtiles(int32_t V) { return 2 * cdiv(V, TV); }

size_t snx_dec256_table_bytes(int32_t T) { return (size_t)T * 4 + (size_t)T * 4 + 16 + ((size_t)T / 32 + T + 8) * 16 + 256; }

int snx_launch_decoder256(const void* Hd, const void* W, const float* bias, const int32_t* cu_seqlens,
                          const int64_t* mask, float* sparse, uint32_t* keys, void* scratch, size_t rowpart_bytes,
                          int32_t T, int32_t nseq, int32_t V, int32_t K, hipStream_t st) {
  if ((K % 64) || T <= 0 || nseq <= 0 || V <= 0 || (long)T * K * 2 >= (1L << 32) || (long)V * K * 2 >= (1L << 32))
    return SNX_E_SHAPE;
  if (NJ != 3) return SNX_E_SHAPE;                    // the DMA / read schedule of the K loop is written for NJ = 3
  char* base = (char*)scratch + ((rowpart_bytes + 255) & ~(size_t)255);
  int32_t* list = (int32_t*)base;
  int32_t* nvalid = list + T;
  i32x4* subtab = (i32x4*)(((uintptr_t)(nvalid + T) + 15) & ~(uintptr_t)15);   // 16-byte aligned for any T
  int32_t* hdr = (int32_t*)(subtab + ((size_t)T / 32 + T + 8));
  hipLaunchKernelGGL(dec256_rows_kernel, dim3(nseq), dim3(64), 0, st, cu_seqlens, mask, list, nvalid, nseq);
  SNX_CHECK_LAUNCH();
  hipLaunchKernelGGL(dec256_tiles_kernel, dim3(1), dim3(256), 0, st, cu_seqlens, nvalid, subtab, hdr, nseq);
  SNX_CHECK_LAUNCH();
  hipError_t e = hipMemsetAsync(keys, 0, (size_t)nseq * V * 4, st);
  if (e != hipSuccess) return (int)e;
  static bool attr[64] = {};
  int devid = 0;
  if (hipGetDevice(&devid) != hipSuccess || devid < 0 || devid >= 64) return SNX_E_ARG;
  if (!attr[devid]) {
    e = hipFuncSetAttribute((const void*)decoder256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, RING * SLOT);
    if (e != hipSuccess) return (int)e;
    attr[devid] = true;
  }
  DecArgs g;
  g.Hd = (const bf16_t*)Hd; g.W = (const bf16_t*)W; g.bias = bias; g.list = list; g.subtab = subtab; g.hdr = hdr;
  g.keys = keys; g.rowpart = (unsigned short*)scratch; g.T = T; g.V = V; g.K = K; g.ntn = cdiv(V, TV);
  hipLaunchKernelGGL(decoder256_kernel, dim3(NWG), dim3(256), RING * SLOT, st, g);
  SNX_CHECK_LAUNCH();
  const long total = (long)nseq * V;
  hipLaunchKernelGGL(dec256_finalize_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, keys, sparse, cu_seqlens, list, V,
                     total);
  SNX_CHECK_LAUNCH();
  return SNX_OK;
}
