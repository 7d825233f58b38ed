// dX of mlp.Wo fused with the GeGLU backward, PIPELINED form:  dy = A[M,K] * B[N,K]^T (bf16, fp32 MFMA accumulation),
// du[M,2N] = GeGLU'(u, dy)  -- the backward of hf:modeling_modernbert.py:89-91 (`Wo(act(input) * gate)`) that autograd runs
// as a GEMM plus four elementwise kernels.
//
// Why a third NT form (round 5; rounds 3-4 left this launch at 0.19 of the bf16 peak).  In the 128x128 kernel of gemm.hip
// a wave runs its K loop and THEN its epilogue, and the epilogue of this op is heavy: per 128x128 tile 64 KiB of saved u
// in, 64 KiB of du out, and per lane 64 exact-erf GELU + GELU' evaluations -- ~1,650 vector instructions + 128
// transcendentals = ~8.7k issue cycles per wave against 6.1k cycles of MFMA.  The two serial parts overlap only through the
// CU's other workgroup, by chance; measured: K loop 15.5 us + epilogue 7.0 us per tile, 130 us per launch at 36,864 rows
// where the MFMAs alone need 26 us at peak and the HBM traffic 66 us.  The 256x256 ping-pong kernel exposes the same
// epilogue on all eight waves at once (slower still).
//
// Here the workgroup is PERSISTENT (it walks its tiles in the XCD-aware order) and the epilogue of tile i is cut into
// pieces that ride INSIDE the K loop of tile i + 1, in the same instruction stream as its MFMAs:
//   * at the end of a tile's K loop the wave packs its 64x64 accumulators to bf16 (the Linear's output IS bf16): rows
//     0..31 into its private 4-KiB staging image (the swizzled image of gemm_epi.h), rows 32..63 into 16 parked registers;
//   * K-steps 0,1 / 5,6 of the next tile request the saved u of a 32-row pass (two 16-byte loads per row and lane, in the
//     row-major role of the write-back), K-steps 2..5 / 7..10 each take ONE row group of the pass: read its bf16 dy row
//     from the image, 8 GELU / GELU' evaluations per lane, two 16-byte stores of du -- ~280 vector instructions that
//     issue in the shadow of the step's 32 MFMAs (a 16x16x32 MFMA holds the issue port 4 of its 16 cycles) and of the
//     LDS round trip in front of them;
//   * the operand stream never drains: K-tile 0 of the next tile is requested in the last K-step of the current one.
// One barrier per K-step as in gemm_core.h's main loop; its wait is COUNTED -- vmcnt(V) with V = the vector-memory
// instructions the previous K-step issued behind its LDS-DMA (the queue retires in order) -- so the u loads and du stores
// stay in flight across it.  Same products summed in the same order and the same epilogue arithmetic as the 128x128 kernel:
// bit-identical results (tests/test_gpu_ops.py).
//
// Shapes: M % 128 == 0, N % 128 == 0, K = 64 NK for the instantiated NK (12: the 149 M model's hidden size); everything
// else stays on the 128x128 kernel (snx_launch_nt_pipe returns SNX_E_SHAPE).
#include "gemm_core.h"
#include "config.h"
#include "gemm_epi.h"
#include "snx.h"
#include <stdio.h>

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;   // 32 KiB
constexpr int IMG = 4096;                                 // per wave: 32 rows x 64 columns bf16
constexpr int LDS_TOTAL = 2 * STAGE + 4 * IMG;            // 80 KiB: two workgroups per CU

template <int N>
struct IC { static constexpr int value = N; };

#define WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

// Epilogue schedule over the K-steps of the NEXT tile.  piece(s): what step s does for the parked tile --
//   L(p, k): request u of pass p (rows 32 p ..), row group k (rows rr + 8 k);  U(p, k): row group k of pass p: compute + store;
//   P: refill the image with pass 1 (from the parked registers).
// vector-memory instructions issued per step (behind the step's LDS-DMA): 2 per L, 2 per U
//   step:   0        1        2      3      4      5                   6        7      8      9      10     11..
//   does:   L00 L01  L02 L03  U00    U01    U02    U03 P L10 L11       L12 L13  U10    U11    U12    U13    -
__host__ __device__ constexpr int vm_ops(int s) {
  return s == 0 ? 4 : s == 1 ? 4 : s == 5 ? 6 : s == 6 ? 4 : (s >= 2 && s <= 10) ? 2 : 0;
}

}  // namespace

template <int NK>
__global__ __launch_bounds__(256, 2) void gemm_nt_geglu_bwd_pipe_kernel(const bf16_t* __restrict__ A,
                                                                        const bf16_t* __restrict__ B, int M, int N,
                                                                        TileOrder order, int ntiles,
                                                                        const bf16_t* __restrict__ U,
                                                                        bf16_t* __restrict__ dU) {
  static_assert(NK == 12, "the epilogue schedule is written for 12 K-steps");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int K = NK * BK;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 15, g = lane >> 4;
  const int rr = lane >> 3, rc = lane & 7;                  // row-major role: rows rr + 8 k of a pass, 8 columns from 8 rc
  char* img = smem + 2 * STAGE + wave * IMG;

  // ---- this workgroup's tiles: the XCD's contiguous run of the tile order, interleaved over its workgroups ----
  const int xcd = blockIdx.x & 7, jw = blockIdx.x >> 3, per = gridDim.x >> 3;
  const int P0 = (int)((long)ntiles * xcd / 8), P1 = (int)((long)ntiles * (xcd + 1) / 8);
  int pos = P0 + jw;
  if (pos >= P1) return;

  // per-lane byte offsets of this wave's 4 + 4 LDS-DMA instructions inside a (tile, K-tile): row r of the image holds
  // logical 16-byte chunk (lane & 7) ^ (r & 7) at slot lane & 7 (gemm_core.h)
  unsigned off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (i * 4 + wave) * 8 + (lane >> 3);
    off[i] = ((unsigned)r * (unsigned)K + (unsigned)(((lane & 7) ^ (r & 7)) * 8)) * 2u;
  }
  auto dma = [&](int m0, int n0, int kt, char* stage) __attribute__((always_inline)) {
    const char* ba = (const char*)(A + (long)m0 * K + kt * BK);
    const char* bb = (const char*)(B + (long)n0 * K + kt * BK);
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

  int tm, tn;
  tile_of(order, pos, tm, tn);
  int m0 = tm * BM, n0 = tn * BN;
  int par = 0;                                              // stage that holds the K-tile of the current step
  dma(m0, n0, 0, smem);

  // ---- state of the PARKED tile (the previous one of this workgroup) ----
  long prev_u = 0;                                          // element offset of this lane's (row rr, 8 columns) in u / du
  bf16x4 parked[2][4];                                      // rows 32..63 of the wave tile, accumulator layout, bf16
  f32x4 ua[4], ug[4];                                       // saved u of the pass in flight: a and g of row group k

  // u / du element offset of row group k of pass p: row = row0 + 32 p + rr + 8 k, dy columns [col, col + 8) <-> a at
  // u[64 (col >> 5) + (col & 31)], g 32 further
  // The loads are VOLATILE ASM: hipcc's waitcnt pass cannot count across LDS-DMA in flight -- with plain C++ loads it put
  // s_waitcnt vmcnt(0) in front of the first use of u (steps 2 and 7), i.e. a wait for the K-tile requested a moment
  // earlier, a full memory round trip inside the step.  Their results are valid by the SCHEDULE: a load issued in step s is
  // covered by the counted wait at the top of step s + 2 (it is no longer among the newest vm_ops(s + 1) instructions), and
  // no unit reads it earlier.  The price is the one of the persistent kernels' asm LDS reads: the compiler believes the
  // destination registers valid at once, so the build guard (snx/asmcheck.py, vm rule) checks that nothing touches them
  // before a vmcnt wait that retires the load.
  auto piece_load = [&](int p, int k) __attribute__((always_inline)) {
    const bf16_t* uu = U + prev_u + (long)(32 * p + 8 * k) * (2 * N);
    asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(ua[k]) : "v"(uu));
    asm volatile("global_load_dwordx4 %0, %1, off offset:64" : "=&v"(ug[k]) : "v"(uu));
  };
  // ... and an empty volatile asm that "redefines" the registers right behind the wait that makes them valid: every use of
  // the values is then ordered behind that wait (volatile asms keep their order; plain vector instructions on an asm's
  // output may otherwise be scheduled anywhere behind the load itself -- hipcc did exactly that in the drain code)
  auto landed = [&](int k0) __attribute__((always_inline)) {
    asm volatile("" : "+v"(ua[k0]), "+v"(ug[k0]), "+v"(ua[k0 + 1]), "+v"(ug[k0 + 1]));
  };
  auto piece_unit = [&](int p, int k) __attribute__((always_inline)) {
    const bf16x8 v = (k & 1) ? stg_get<true>(img, rr + 8 * k, rc) : stg_get<false>(img, rr + 8 * k, rc);
    const bf16x8 a8 = __builtin_bit_cast(bf16x8, ua[k]), g8 = __builtin_bit_cast(bf16x8, ug[k]);
    bf16x8 da, dg;
#if defined(SNX_PIPE_DIAG) && SNX_PIPE_DIAG == 3           // timing-only: no GELU arithmetic
    da = v; dg = a8;
    dg[0] = g8[0];
#else
#pragma unroll
    for (int r = 0; r < 8; ++r) {                           // the arithmetic of gemm.hip's EPI_GEGLU_BWD, verbatim
      const float af = bf2f(a8[r]), gf = bf2f(g8[r]), df = bf2f(v[r]);
      dg[r] = f2bf(df * rbf(gelu_f(af)));
      da[r] = f2bf(rbf(df * gf) * gelu_grad_f(af));
    }
#endif
    bf16_t* o = dU + prev_u + (long)(32 * p + 8 * k) * (2 * N);
#if defined(SNX_PIPE_DIAG) && SNX_PIPE_DIAG == 1           // timing-only builds (wrong results): no du stores
    asm volatile("" ::"v"(da), "v"(dg), "v"(o));
#else
    *(bf16x8*)o = da;
    *(bf16x8*)(o + 32) = dg;
#endif
  };
  auto piece_put1 = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int jn = 0; jn < 4; ++jn) stg_put(img, ii * 16 + li, jn * 16 + g * 4, parked[ii][jn]);
  };
  // the epilogue piece of K-step S
  auto piece = [&](auto sc) __attribute__((always_inline)) {
    constexpr int S = decltype(sc)::value;
#if defined(SNX_PIPE_DIAG) && SNX_PIPE_DIAG == 2           // timing-only: the K loops alone
    return;
#endif
    if (S == 0) { piece_load(0, 0); piece_load(0, 1); }
    if (S == 1) { piece_load(0, 2); piece_load(0, 3); }
    if (S >= 2 && S <= 5) piece_unit(0, S - 2);
    if (S == 5) { piece_put1(); piece_load(1, 0); piece_load(1, 1); }
    if (S == 6) { piece_load(1, 2); piece_load(1, 3); }
    if (S >= 7 && S <= 10) piece_unit(1, S - 7);
  };

  f32x4 acc[4][4];
  int nm0 = 0, nn0 = 0;
  bool has_next = false;

  // ---- one K-step: wait + barrier, request the next K-tile, 32 MFMAs with the epilogue piece of the parked tile ----
  // HP (compile time): a parked tile exists.  A run-time test would put the piece and the MFMAs into different basic
  // blocks, and hipcc schedules inside a block only: the first tile of a workgroup runs its own copy of the K loop.
  auto kstep = [&](auto sc, auto hp) __attribute__((always_inline)) {
    constexpr int S = decltype(sc)::value;
    constexpr bool HP = decltype(hp)::value != 0;
    // K-tile S of this tile has landed for this wave: everything but the vm instructions the previous step issued
    // behind its LDS-DMA (none in front of step 0: the schedule's last step is empty)
#if defined(SNX_PIPE_DIAG) && (SNX_PIPE_DIAG == 1 || SNX_PIPE_DIAG == 2)
    constexpr int V = (S == 0 || !HP || SNX_PIPE_DIAG == 2) ? 0 : (vm_ops(S - 1) >= 4 ? vm_ops(S - 1) - (S - 1 == 5 ? 2 : 0) : 0);
#else
    constexpr int V = (S == 0 || !HP) ? 0 : vm_ops(S - 1);
#endif
    if (V == 0) WAIT_VM(0);
    else if (V == 2) WAIT_VM(2);
    else if (V == 4) WAIT_VM(4);
    else WAIT_VM(6);
    __builtin_amdgcn_s_barrier();                           // ... for every wave; everyone has left the other stage
    // the same wait covers the u loads issued two steps ago (schedule above): steps 0 / 5 -> 2 / 7, steps 1 / 6 -> 3 / 8
    if (HP && (S == 2 || S == 7)) landed(0);
    if (HP && (S == 3 || S == 8)) landed(2);
    __builtin_amdgcn_sched_barrier(0);
    const char* cur = smem + par * STAGE;
    char* nxt = smem + (par ^ 1) * STAGE;
    par ^= 1;
    if (S + 1 < NK) dma(m0, n0, S + 1, nxt);
    else if (has_next) dma(nm0, nn0, 0, nxt);
    // the piece's loads / stores stay BEHIND the DMA in the queue (the counted wait of the next step relies on it): no
    // memory instruction may cross this point, in the optimizer or in the scheduler
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
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
    if (HP) piece(sc);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[kk][j], a[kk][i], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  };

  auto tile_body = [&](auto hp) __attribute__((always_inline)) {
    // the next tile of this workgroup (scalar divisions: once per tile, in front of the K loop)
    has_next = pos + per < P1;
    if (has_next) {
      tile_of(order, pos + per, tm, tn);
      nm0 = tm * BM; nn0 = tn * BN;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    kstep(IC<0>(), hp); kstep(IC<1>(), hp); kstep(IC<2>(), hp); kstep(IC<3>(), hp); kstep(IC<4>(), hp);
    kstep(IC<5>(), hp); kstep(IC<6>(), hp); kstep(IC<7>(), hp); kstep(IC<8>(), hp); kstep(IC<9>(), hp);
    kstep(IC<10>(), hp); kstep(IC<11>(), hp);
    // ---- park the finished tile: rows 0..31 -> image (the previous tile's last read of it was in step 10: LDS executes
    //      a wave's instructions in order), rows 32..63 -> registers ----
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int jn = 0; jn < 4; ++jn) {
        stg_put(img, ii * 16 + li, jn * 16 + g * 4, pack4(acc[ii][jn]));
        parked[ii][jn] = pack4(acc[2 + ii][jn]);
      }
    {
      const int row = m0 + wm * 64 + rr, col = n0 + wn * 64 + rc * 8;
      prev_u = (long)row * (2 * N) + 64 * (col >> 5) + (col & 31);
    }
  };
  tile_body(IC<0>());
  while (has_next) {
    m0 = nm0; n0 = nn0;
    pos += per;
    tile_body(IC<1>());
  }
  // ---- the last tile's epilogue, not overlapped: no K-step waits cover the asm loads here ----
  piece(IC<0>()); piece(IC<1>());
  WAIT_VM(0);
  landed(0); landed(2);
  piece(IC<2>()); piece(IC<3>()); piece(IC<4>()); piece(IC<5>()); piece(IC<6>());
  WAIT_VM(0);
  landed(0); landed(2);
  piece(IC<7>()); piece(IC<8>()); piece(IC<9>()); piece(IC<10>());
}

// SNX_OK, SNX_E_SHAPE (shape not taken: the caller falls back to the 128x128 kernel) or a HIP error code
int snx_launch_nt_pipe_geglu_bwd(const void* A, const void* B, int M, int N, int K, const EpiArgs& e, hipStream_t st) {
  if (!g_snx_cfg.nt_pipe || M < g_snx_cfg.nt_pipe_min_m) return SNX_E_SHAPE;
  if ((M % BM) || (N % BN) || K != 12 * BK) return SNX_E_SHAPE;
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
  static LdsOptIn optin;
  auto kern = gemm_nt_geglu_bwd_pipe_kernel<12>;
  if (const int rc = optin.ensure((const void*)kern, LDS_TOTAL)) return rc;
  int nwg = 2 * (256 - snx_get_reserved_cus());             // two workgroups per CU
  if (nwg > tm * tn) nwg = (tm * tn + 7) & ~7;
  if (nwg < 8) nwg = 8;
#ifdef SNX_PIPE_DIAG
  {
    static bool once = false;
    if (!once) {
      int nb = -1;
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)kern, 256, LDS_TOTAL);
      fprintf(stderr, "[nt_pipe] workgroups per CU: %d\n", nb);
      once = true;
    }
  }
#endif
  hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), LDS_TOTAL, st, (const bf16_t*)A, (const bf16_t*)B, M, N, order, tm * tn,
                     e.U, e.C);
  SNX_CHECK_LAUNCH();
  return SNX_OK;
}
