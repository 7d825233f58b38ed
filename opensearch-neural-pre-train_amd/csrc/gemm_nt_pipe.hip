// dX of mlp.Wo fused with the GeGLU backward, HELPER-WAVE form:  dy = A[M,K] * B[N,K]^T (bf16, fp32 MFMA accumulation),
// du[M,2N] = GeGLU'(u, dy)  -- the backward of hf:modeling_modernbert.py:89-91 (`Wo(act(input) * gate)`) that autograd runs
// as a GEMM plus four elementwise kernels.
//
// Why a third NT form (round 5; rounds 3-4 left this launch at 0.19 of the bf16 peak).  Per 128x128 tile the epilogue of
// this op moves 64 KiB of saved u in and 64 KiB of du out and evaluates 64 exact-erf GELU + GELU' pairs per lane (~1,650
// vector instructions + 128 transcendentals = ~8.7k issue cycles per wave against 6.1k cycles of MFMA).  In gemm.hip's
// 128x128 kernel a wave runs its K loop and THEN that epilogue (K loop 15.5 us + epilogue 7.0 us per tile, 130-138 us per
// launch at 36,864 rows; the MFMAs alone need 26 us at peak, the HBM traffic 66 us); the 256x256 ping-pong kernel exposes
// it on all eight waves at once.  The first form built here kept the epilogue in the MFMA waves and cut it into pieces
// inside the next tile's K loop: 146 us -- while its K loops ALONE (persistent 128x128 tiles, two workgroups per CU) ran in
// 48 us = 1.35 PFLOP/s.  A wave's vector-memory instructions retire IN ORDER: with a double-buffered operand ring every
// other load or store of the wave has to complete within ~1.5 K-steps (~1.2 us) or the wait for the next K-tile waits for
// it too, and cold HBM loads / stores do not (profiles/r05_experiments.txt, sections 2-3).
//
// Hence: the epilogue gets waves -- and vector-memory queues -- of its own, inside a PERSISTENT workgroup.
//   * 8 waves: waves 0-3 multiply (2 x 2 grid of 64x64 wave tiles; the K loop of gemm_core.h: one barrier and one
//     vmcnt(0) per K-step, LDS-DMA double buffer); waves 4-7 are HELPERS, helper h serving MFMA wave h.  Two workgroups
//     per CU (80 KiB of LDS, <= 128 registers: 4 waves per SIMD = one MFMA wave and one helper of either workgroup -- with
//     two helpers per workgroup both workgroups' helpers landed on SIMDs 0 and 1 and the launch took 161 us: the GELU
//     arithmetic is 43 us of vector issue per launch only when it is spread over all four SIMDs).
//   * The workgroup walks its tiles (XCD-contiguous run of the tile order, interleaved over the XCD's workgroups); the
//     operand stream never drains: K-tile 0 of the next tile is requested in the last K-step of the current one.
//   * Hand-over through LDS, ordered by the K loop's own barriers (a tile PERIOD = its 12 K-steps; the helpers take part in
//     every barrier): at the end of a tile an MFMA wave packs its 64x64 accumulators to bf16 (the Linear's output IS
//     bf16) -- rows 0..31 into its private 4-KiB staging image (the swizzled image of gemm_epi.h), rows 32..63 into 16
//     parked registers.  Behind barrier 0 of the next period the helper reads that pass row-major into registers; behind
//     barrier 1 the MFMA wave refills the image with rows 32..63; the helper reads those in step 6.
//   * A helper's period: 8 units (pass, row group) of 8 GELU / GELU' evaluations per lane, two 16-byte loads of the saved
//     u and two 16-byte stores of du each -- spread as 16 half units over the 12 steps (1-2 per step, less than the
//     K-step's own length) so that the shared barriers cost neither side anything.  The saved u of a unit is requested
//     FOUR steps (~3 us) ahead -- across the period boundary for the first two units: u does not depend on dy --, plain
//     C++ loads: no LDS-DMA in the helpers' instruction stream, so hipcc's counted waits are exact there.
// Same products summed in the same order and the same epilogue arithmetic as the 128x128 kernel: bit-identical results
// (tests/test_gpu_ops.py::test_geglu_bwd_pipelined_kernel_equals_the_128_kernel_bit_for_bit).
//
// Measured (profiles/r05_experiments.txt section 3; 36,864 rows, back to back): 145 us (120 with non-temporal stores)
// against 138-143 for the 128x128 kernel; in the training step 0.25 ms per micro-step faster.  NOT the <= 95 us asked for,
// and the timing-only builds say why: helpers idle 68 us (the K loops: 0.95 PFLOP/s), + GELU arithmetic without memory
// traffic 95, + the u loads 103, loads + stores without arithmetic 125-136 -- the arithmetic hides behind the MFMAs, the
// HBM stream does not: its 66 us ADD to the K loops' 68.  A 128x128 tile pulls 1 GB per launch from the L2 into LDS (65
// FLOP per byte) and waits for every K-tile one step after requesting it; 400 MB of L2 misses in the same queues lengthen
// every one of those waits.  Only a tile with less L2 traffic per FLOP (256 wide: no registers or LDS left for helpers)
// or a deeper operand ring (LDS) would separate the two.
//
// Shapes: M % 128 == 0, N % 128 == 0, K = 768 (12 K-steps: the 149 M model's hidden size); everything else stays on the
// 128x128 kernel (snx_launch_nt_pipe_geglu_bwd returns SNX_E_SHAPE).
#include "gemm_core.h"
#include "config.h"
#include "gemm_epi.h"
#include "snx.h"
#include <stdio.h>

namespace {

constexpr int BM = 128, BN = 128, BK = 64, NK = 12;
constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;   // 32 KiB
constexpr int IMG = 4096;                                 // per MFMA wave: 32 rows x 64 columns bf16
constexpr int LDS_TOTAL = 2 * STAGE + 4 * IMG;            // 80 KiB: two workgroups per CU
constexpr int NTHREADS = 8 * 64;

template <int N>
struct IC { static constexpr int value = N; };

#define WAIT_VM0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// ---- a helper's schedule over the 12 steps of a period ----
// unit q = 4 p + k: pass p (rows 32 p ..), row group k (rows rr + 8 k);
// half unit x = 2 q + h: elements 4 h .. 4 h + 3 of the lane's 8.  Step s computes half units [first_half(s), first_half(s + 1)).
constexpr int NUNITS = 8;
__host__ __device__ constexpr int first_half(int s) { return 2 * NUNITS * s / NK; }
// the step in which unit q is first needed, and the step (of this or the PREVIOUS period) in which its u is requested
__host__ __device__ constexpr int need_step(int q) {
  for (int s = 0; s < NK; ++s)
    if (first_half(s + 1) > 2 * q) return s;
  return NK - 1;
}
constexpr int LEAD = 4;
__host__ __device__ constexpr int load_step(int q) { return (need_step(q) + NK - LEAD) % NK; }
__host__ __device__ constexpr bool load_is_ahead(int q) { return need_step(q) < LEAD; }   // requested in the previous period

enum { PERIOD_FIRST = 0, PERIOD_MIDDLE = 1, PERIOD_DRAIN = 2 };

}  // namespace

// NTS ("nt_pipe" = 1): du leaves through non-temporal stores -- 120 against 145 us per launch back to back (170 MB of
// results that otherwise push the operands out of the L2 the K loops live on), but no faster inside the training step,
// where the next two GEMMs read du (44.51 against 44.46 ms per micro-step): the default is plain stores.
template <bool NTS, bool NTL = false>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_nt_geglu_bwd_pipe_kernel(const bf16_t* __restrict__ A,
                                                                            const bf16_t* __restrict__ B, int M, int N,
                                                                            TileOrder order, int ntiles,
                                                                            const bf16_t* __restrict__ U,
                                                                            bf16_t* __restrict__ dU) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int K = NK * BK;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

  // ---- this workgroup's tiles: the XCD's contiguous run of the tile order, interleaved over its workgroups ----
  const int xcd = blockIdx.x & 7, jw = blockIdx.x >> 3, per = gridDim.x >> 3;
  const int P0 = (int)((long)ntiles * xcd / 8), P1 = (int)((long)ntiles * (xcd + 1) / 8);
  int pos = P0 + jw;
  if (pos >= P1) return;
  int tm, tn;
  tile_of(order, pos, tm, tn);
  int m0 = tm * BM, n0 = tn * BN;                           // the tile in the K loop
  int nm0 = 0, nn0 = 0;                                     // the one after it
  bool has_next = false;
  auto look_ahead = [&]() __attribute__((always_inline)) {  // (scalar divisions: once per period)
    has_next = pos + per < P1;
    if (has_next) {
      tile_of(order, pos + per, tm, tn);
      nm0 = tm * BM; nn0 = tn * BN;
    }
  };

  if (wave < 4) {
    // =================================== MFMA waves ===================================
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 15, g = lane >> 4;
    char* img = smem + 2 * STAGE + wave * IMG;
    // per-lane byte offsets of this wave's 4 + 4 LDS-DMA instructions inside a (tile, K-tile): row r of the image holds
    // logical 16-byte chunk (lane & 7) ^ (r & 7) at slot lane & 7 (gemm_core.h)
    unsigned off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = (i * 4 + wave) * 8 + (lane >> 3);
      off[i] = ((unsigned)r * (unsigned)K + (unsigned)(((lane & 7) ^ (r & 7)) * 8)) * 2u;
    }
    auto dma = [&](int tm0, int tn0, int kt, char* stage) __attribute__((always_inline)) {
      const char* ba = (const char*)(A + (long)tm0 * K + kt * BK);
      const char* bb = (const char*)(B + (long)tn0 * K + kt * BK);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_global_load_lds(GLB_PTR(ba + off[i]), LDS_PTR(stage + (i * 4 + wave) * 1024), 16, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_global_load_lds(GLB_PTR(bb + off[i]), LDS_PTR(stage + A_BYTES + (i * 4 + wave) * 1024), 16, 0, 0);
    };
    auto frag = [&](const char* tile, int row, int chunk) __attribute__((always_inline)) {
      return *(const bf16x8*)(tile + row * 128 + ((chunk ^ (row & 7)) << 4));
    };
    int par = 0;                                            // stage that holds the K-tile of the current step
    dma(m0, n0, 0, smem);
    f32x4 acc[4][4];
    bf16x4 parked[2][4];                                    // rows 32..63 of the finished tile, accumulator layout, bf16
    auto put_parked = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int jn = 0; jn < 4; ++jn) stg_put(img, ii * 16 + li, jn * 16 + g * 4, parked[ii][jn]);
    };
    // one K-step: K-tile S has landed (this wave's DMA, then everyone's), request the next one, 32 MFMAs
    auto kstep = [&](auto sc, auto kind) __attribute__((always_inline)) {
      constexpr int S = decltype(sc)::value;
      constexpr int KIND = decltype(kind)::value;
      WAIT_VM0();
#if !(defined(SNX_PIPE_DIAG) && SNX_PIPE_DIAG == 4)
      WAIT_LGKM0();                                         // (this wave's image writes have executed: the helpers read them
                                                            //  behind the barrier; the fragment reads were consumed already)
#endif
      __builtin_amdgcn_s_barrier();                         // ... and everyone has left the other stage
      __builtin_amdgcn_sched_barrier(0);
      const char* cur = smem + par * STAGE;
      char* nxt = smem + (par ^ 1) * STAGE;
      par ^= 1;
      if (S + 1 < NK) dma(m0, n0, S + 1, nxt);
      else if (has_next) dma(nm0, nn0, 0, nxt);
      __builtin_amdgcn_sched_barrier(0);
      if (S == 1 && KIND == PERIOD_MIDDLE) put_parked();    // behind barrier 1: the helper has read rows 0..31 of the image
      const char* ta = cur + (wm * 64) * 128;
      const char* tb = cur + A_BYTES + (wn * 64) * 128;
      bf16x8 a[2][4], b[2][4];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int i = 0; i < 4; ++i) a[kk][i] = frag(ta, i * 16 + li, kk * 4 + g);
#pragma unroll
        for (int j = 0; j < 4; ++j) b[kk][j] = frag(tb, j * 16 + li, kk * 4 + g);
      }
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[kk][j], a[kk][i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    };
    auto period = [&](auto kind) __attribute__((always_inline)) {
      look_ahead();
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      kstep(IC<0>(), kind); kstep(IC<1>(), kind); kstep(IC<2>(), kind); kstep(IC<3>(), kind); kstep(IC<4>(), kind);
      kstep(IC<5>(), kind); kstep(IC<6>(), kind); kstep(IC<7>(), kind); kstep(IC<8>(), kind); kstep(IC<9>(), kind);
      kstep(IC<10>(), kind); kstep(IC<11>(), kind);
      // hand the tile over: rows 0..31 -> image (the helper's last read of it was in step 6), rows 32..63 -> registers
#pragma unroll
      for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int jn = 0; jn < 4; ++jn) {
          stg_put(img, ii * 16 + li, jn * 16 + g * 4, pack4(acc[ii][jn]));
          parked[ii][jn] = pack4(acc[2 + ii][jn]);
        }
    };
    period(IC<PERIOD_FIRST>());
    while (has_next) {
      m0 = nm0; n0 = nn0;
      pos += per;
      period(IC<PERIOD_MIDDLE>());
    }
    // drain period: the helpers finish the last tile; this wave only keeps the barriers' count and refills the image
    WAIT_LGKM0();
    __builtin_amdgcn_s_barrier();                           // barrier 0
    __builtin_amdgcn_s_barrier();                           // barrier 1
    put_parked();
    WAIT_LGKM0();
#pragma unroll
    for (int s = 2; s < NK; ++s) __builtin_amdgcn_s_barrier();
    return;
  }

  // ===================================== helper waves =====================================
#if defined(SNX_PIPE_DIAG) && SNX_PIPE_DIAG == 5           // timing-only: the helpers leave at once
  return;
#endif
  const int hw = wave - 4;                                  // serves MFMA wave hw: rows 64 (hw >> 1) .., columns 64 (hw & 1) ..
  const int rr = lane >> 3, rc = lane & 7;                  // row-major role: rows rr + 8 k of a pass, 8 columns from 8 rc
  const char* img = smem + 2 * STAGE + hw * IMG;
  bf16x8 dy[4];                                             // the pass in work: row group k
  f32x4 ua[NUNITS], ug[NUNITS];                             // saved u (a | g) of unit q: 8 bf16 each
  bf16x4 da_lo, dg_lo;                                      // first half of the unit in work
  // element offset of this lane's (row rr of the wave tile, columns 8 rc ..) in u / du for a tile at (tm0, tn0); unit
  // q = (p, k) lies (32 p + 8 k) rows further (dy columns [col, col + 8) <-> a at u[64 (col >> 5) + (col & 31)], g 32 further)
  auto u_base = [&](int tm0, int tn0) __attribute__((always_inline)) {
    const int row = tm0 + 64 * (hw >> 1) + rr, col = tn0 + 64 * (hw & 1) + rc * 8;
    return (long)row * (2 * N) + 64 * (col >> 5) + (col & 31);
  };
  long base_prev = 0, base_cur = u_base(m0, n0);
  auto unit_off = [&](int q) __attribute__((always_inline)) { return (long)(32 * (q >> 2) + 8 * (q & 3)) * (2 * N); };
  auto load_unit = [&](int q, long base) __attribute__((always_inline)) {
#if defined(SNX_PIPE_DIAG) && SNX_PIPE_DIAG == 6           // timing-only: GELU arithmetic without global loads / stores
    ua[q] = __builtin_bit_cast(f32x4, dy[q & 3]); ug[q] = ua[q];
    return;
#endif
    const bf16_t* uu = U + base + unit_off(q);
    if (NTL) {                                              // "stream_nt" bit 64: the saved u is read for the last time
      ua[q] = __builtin_nontemporal_load((const f32x4*)uu);
      ug[q] = __builtin_nontemporal_load((const f32x4*)(uu + 32));
    } else {
      ua[q] = *(const f32x4*)uu;
      ug[q] = *(const f32x4*)(uu + 32);
    }
  };
  auto read_pass = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      dy[k] = (k & 1) ? stg_get<true>(img, rr + 8 * k, rc) : stg_get<false>(img, rr + 8 * k, rc);
  };
  // half unit x of the parked tile: the arithmetic of gemm.hip's EPI_GEGLU_BWD, verbatim
  auto half_unit = [&](int x) __attribute__((always_inline)) {
    const int q = x >> 1, h = x & 1;
    const bf16x8 v = dy[q & 3];
    const bf16x8 a8 = __builtin_bit_cast(bf16x8, ua[q]), g8 = __builtin_bit_cast(bf16x8, ug[q]);
    bf16x4 da, dg;
#if defined(SNX_PIPE_DIAG) && (SNX_PIPE_DIAG == 3 || SNX_PIPE_DIAG == 7)   // timing-only builds (wrong results): no GELU arithmetic
#pragma unroll
    for (int r = 0; r < 4; ++r) { da[r] = v[4 * h + r]; dg[r] = a8[4 * h + r]; }
    dg[0] = g8[4 * h];
#else
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float af = bf2f(a8[4 * h + r]), gf = bf2f(g8[4 * h + r]), df = bf2f(v[4 * h + r]);
      dg[r] = f2bf(df * rbf(gelu_f(af)));
      da[r] = f2bf(rbf(df * gf) * gelu_grad_f(af));
    }
#endif
    if (h == 0) {
      da_lo = da; dg_lo = dg;
    } else {
      bf16_t* o = dU + base_prev + unit_off(q);
#if defined(SNX_PIPE_DIAG) && (SNX_PIPE_DIAG == 1 || SNX_PIPE_DIAG == 6)   // timing-only: no du stores
      asm volatile("" ::"v"(da_lo), "v"(dg_lo), "v"(da), "v"(dg), "v"(o));
#elif defined(SNX_PIPE_DIAG) && SNX_PIPE_DIAG == 7         // timing-only: the same lines written as WHOLE 128-byte lines
      {
        const int col = (int)((base_prev + unit_off(q)) % (2 * N));
        bf16_t* line = o - col + (col & ~63) + ((rc & 4) ? -64 : 0);     // first of the row's two lines
        *(bf16x8*)(line + rc * 8) = (bf16x8){da_lo[0], da_lo[1], da_lo[2], da_lo[3], da[0], da[1], da[2], da[3]};
        *(bf16x8*)(line + 64 + rc * 8) = (bf16x8){dg_lo[0], dg_lo[1], dg_lo[2], dg_lo[3], dg[0], dg[1], dg[2], dg[3]};
      }
#else
      const bf16x8 va = (bf16x8){da_lo[0], da_lo[1], da_lo[2], da_lo[3], da[0], da[1], da[2], da[3]};
      const bf16x8 vg = (bf16x8){dg_lo[0], dg_lo[1], dg_lo[2], dg_lo[3], dg[0], dg[1], dg[2], dg[3]};
      if (NTS) {
        __builtin_nontemporal_store(va, (bf16x8*)o);
        __builtin_nontemporal_store(vg, (bf16x8*)(o + 32));
      } else {
        *(bf16x8*)o = va;
        *(bf16x8*)(o + 32) = vg;
      }
#endif
    }
  };
  auto hstep = [&](auto sc, auto kind) __attribute__((always_inline)) {
    constexpr int S = decltype(sc)::value;
    constexpr int KIND = decltype(kind)::value;
    __builtin_amdgcn_s_barrier();
#if defined(SNX_PIPE_DIAG) && (SNX_PIPE_DIAG == 2 || SNX_PIPE_DIAG == 4)   // timing-only: the helpers only keep the barriers' count
    return;
#endif
    if (KIND != PERIOD_FIRST) {
      if (S == 0 || S == 6) read_pass();
#pragma unroll
      for (int x = first_half(S); x < first_half(S + 1); ++x) half_unit(x);
    }
#pragma unroll
    for (int q = 0; q < NUNITS; ++q) {
      if (load_step(q) != S) continue;
      if (load_is_ahead(q)) {
        if (KIND != PERIOD_DRAIN) load_unit(q, base_cur);   // a unit of the tile now in the K loop
      } else {
        if (KIND != PERIOD_FIRST) load_unit(q, base_prev);  // a unit of the parked tile
      }
    }
    // the image reads of steps 0 / 6 must have returned before the MFMA waves refill the image behind barrier 1 / at the
    // end of the period (the compiler waits at the first use, which may lie in a later step)
    if (KIND != PERIOD_FIRST && (S == 0 || S == 6)) WAIT_LGKM0();
  };
  auto hperiod = [&](auto kind) __attribute__((always_inline)) {
    hstep(IC<0>(), kind); hstep(IC<1>(), kind); hstep(IC<2>(), kind); hstep(IC<3>(), kind); hstep(IC<4>(), kind);
    hstep(IC<5>(), kind); hstep(IC<6>(), kind); hstep(IC<7>(), kind); hstep(IC<8>(), kind); hstep(IC<9>(), kind);
    hstep(IC<10>(), kind); hstep(IC<11>(), kind);
  };
  look_ahead();
  hperiod(IC<PERIOD_FIRST>());
  while (has_next) {
    base_prev = base_cur;
    m0 = nm0; n0 = nn0;
    pos += per;
    base_cur = u_base(m0, n0);
    look_ahead();
    hperiod(IC<PERIOD_MIDDLE>());
  }
  base_prev = base_cur;
  hperiod(IC<PERIOD_DRAIN>());
}

// SNX_OK, SNX_E_SHAPE (shape not taken: the caller falls back to the 128x128 kernel) or a HIP error code
int snx_launch_nt_pipe_geglu_bwd(const void* A, const void* B, int M, int N, int K, const EpiArgs& e, hipStream_t st) {
  if (!g_snx_cfg.nt_pipe || M < g_snx_cfg.nt_pipe_min_m) return SNX_E_SHAPE;
  if ((M % BM) || (N % BN) || K != NK * BK) return SNX_E_SHAPE;
  if ((long)M * K * 2 >= (1L << 32) || (long)N * K * 2 >= (1L << 32)) return SNX_E_SHAPE;
  const int tm = M / BM, tn = N / BN;
  // column-group width of the tile order: as in gemm.hip (an XCD's share of B within ~1.8 MB of its L2)
  int cg = tn;
  {
    const double a_bytes = 2.0 * M * K, b_bytes = 2.0 * N * K, cap = 1.8e6;
    double best = -1;
    for (int parts = 1; parts <= 4; ++parts) {
      const int c = cdiv(tn, parts);
      const double bsub = 2.0 * c * BN * K;
      const double cost = a_bytes * cdiv(tn, c) + 8.0 * b_bytes * (bsub <= cap ? 1.0 : 4.0);
      if (best < 0 || cost < best) { best = cost; cg = c; }
    }
  }
  TileOrder order{tm, tn, cdiv(tm, 8), cg};
  static LdsOptIn optin[4];
  const bool nts = g_snx_cfg.nt_pipe == 1;                  // "nt_pipe": 2 = plain du stores (default), 1 = non-temporal
  const bool ntl = (g_snx_cfg.stream_nt & 64) != 0;         // non-temporal loads of the saved u (its last read)
  auto kern = nts ? (ntl ? gemm_nt_geglu_bwd_pipe_kernel<true, true> : gemm_nt_geglu_bwd_pipe_kernel<true, false>)
                  : (ntl ? gemm_nt_geglu_bwd_pipe_kernel<false, true> : gemm_nt_geglu_bwd_pipe_kernel<false, false>);
  if (const int rc = optin[(nts ? 1 : 0) + (ntl ? 2 : 0)].ensure((const void*)kern, LDS_TOTAL)) return rc;
  int nwg = 2 * (256 - snx_get_reserved_cus());             // two workgroups per CU
  if (nwg > tm * tn) nwg = (tm * tn + 7) & ~7;
  if (nwg < 8) nwg = 8;
#ifdef SNX_PIPE_DIAG
  {
    static bool once = false;
    if (!once) {
      int nb = -1;
      (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)kern, NTHREADS, LDS_TOTAL);
      fprintf(stderr, "[nt_pipe] workgroups per CU: %d\n", nb);
      once = true;
    }
  }
#endif
  hipLaunchKernelGGL(kern, dim3(nwg), dim3(NTHREADS), LDS_TOTAL, st, (const bf16_t*)A, (const bf16_t*)B, M, N, order,
                     tm * tn, e.U, e.C);
  SNX_CHECK_LAUNCH();
  return SNX_OK;
}
