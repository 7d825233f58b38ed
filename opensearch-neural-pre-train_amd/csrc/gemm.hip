// Encoder GEMMs (K3/K6/K7/K8/K9 of SURVEY.md §2.3 and their dX counterparts):
//   C[M,N] = A[M,K] * B[N,K]^T, bf16 operands, fp32 MFMA accumulation, fused epilogues.
// Replaces the cuBLAS calls behind nn.Linear under autocast(bf16)
// (transformers modeling_modernbert.py:271,300,90-91,490 via ref:src/model/splade_modern.py:69).
//
// Epilogues (the accumulators are "transposed": a lane owns 4 consecutive columns of one row, and
// the four 16-column tiles j = 0..3 of a wave's 64-column span sit in the SAME lane, so the
// pairings RoPE (d, d+32) and GeGLU (a, g) need no data movement):
//   STORE      C = bf16(acc)
//   RESID      Hout = Hin + bf16(acc)                       (fp32 residual stream, hf:331-332)
//   ROPE       qkv = bf16(acc), q and k thirds rotated      (apply_rotary_pos_emb, hf:196-219)
//   GEGLU_FWD  u = bf16(acc) (interleaved layout), y = gelu(a) * g                (hf:90-91)
//   GEGLU_BWD  dy = bf16(acc); du = GeGLU'(u, dy)            (backward of the above)
// "Interleaved" Wi layout: per 64-column span, columns 0..31 hold a[32q..32q+31] and columns
// 32..63 hold g[32q..32q+31] (the weight cache stores Wi's rows in that order).
//
// Tried and measured slower on MI355X (kept out of the tree): 3-stage counted-vmcnt pipeline at
// one workgroup per CU (551 vs 725 TFLOP/s), 256x128 8-wave tiles (623), 32-deep K-tiles in a
// 4-stage ring (571), persistent workgroups with cross-tile prefetch (neutral), stream-K for dW.
// Two independent 4-wave workgroups per CU with one barrier per 64-deep K-step win.
#include "gemm_core.h"
#include "config.h"
#include "gemm_epi.h"
#include "gemm_tn.h"
#include "snx.h"

#ifdef SNX_GEMM_TRACE
// Diagnostics build only (-DSNX_GEMM_TRACE, tools/gpu_gemm_trace.py): wave 0 of every workgroup records the
// constant-rate clock at entry, after the K loop, after the last store is issued and after the stores have
// left the wave, plus HW_ID / XCC_ID, so that the interleaving of the two workgroups of a CU can be drawn.
__device__ unsigned long long* g_trace = nullptr;
extern "C" int snx_gemm_trace_set(void* buf) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_trace), &buf, sizeof(buf));
}
#define TRACE_T(k) if (threadIdx.x == 0 && g_trace) tr[k] = __builtin_amdgcn_s_memrealtime()
#else
#define TRACE_T(k)
#endif

template <int BM, int BN, int WAVES_M, int WAVES_N, int EPI, bool MID>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64, 2) void gemm_nt_kernel(
    const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, int M, int N, int K, TileOrder order, int ntiles,
    int dbg, EpiArgs e) {
  using Core = GemmCore<BM, BN, WAVES_M, WAVES_N>;
  static_assert(Core::WTN == 64 && Core::WTM == 64, "epilogues assume a 64x64 wave tile");
  static_assert(Core::LDS_BYTES >= WAVES_M * WAVES_N * 16384, "16 KiB of staging per wave");
  extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef SNX_GEMM_TRACE
  unsigned long long tr[4] = {0, 0, 0, 0};
  struct TraceOut {
    unsigned long long* tr;
    __device__ ~TraceOut() {
      if (threadIdx.x == 0 && g_trace) {
        __builtin_amdgcn_s_waitcnt(0);
        tr[3] = __builtin_amdgcn_s_memrealtime();
        unsigned long long* o = g_trace + 8l * blockIdx.x;
        o[0] = tr[0]; o[1] = tr[1]; o[2] = tr[2]; o[3] = tr[3];
        o[4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_REG_HW_ID
        o[5] = __builtin_amdgcn_s_getreg((31 << 11) | 20);     // HW_REG_XCC_ID
      }
    }
  } trace_out{tr};
#endif
  TRACE_T(0);
  int tile_m, tile_n;
  tile_of(order, xcd_remap(blockIdx.x, ntiles), tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  f32x4 acc[Core::MI][Core::NI];
#pragma unroll
  for (int i = 0; i < Core::MI; ++i)
#pragma unroll
    for (int j = 0; j < Core::NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int li = lane & 15, g4 = (lane >> 4) * 4;
  const int row0 = m0 + wm * 64;                     // first row / column of this wave's 64x64 tile
  const int span0 = n0 + wn * 64;
  // row-major role of this lane in the write-back: rows (lane >> 3) + 8k, 8 columns from 8 * (lane & 7)
  const int rr = lane >> 3, rc = lane & 7;
  const bool wide = (N & 7) == 0;                    // 16-B row-major stores need N % 8 == 0 (else: direct stores)

  // Epilogue operands (residual stream / saved u / RoPE table rows), already in the layout the epilogue wants,
  // are requested INSIDE the K loop behind the DMA of K-tile 1 (gemm_core.h): 16 loads per wave, counted.
  // 64 extra VGPRs are free: LDS already limits the CU to two 4-wave workgroups.
  constexpr bool PRE = (EPI == EPI_RESID_F32 || EPI == EPI_GEGLU_BWD || EPI == EPI_ROPE);
  f32x4 pre[PRE ? 8 : 1][2];
  const bool rotate = EPI == EPI_ROPE && span0 < N && span0 < e.rope_cols;   // wave-uniform: a head of q or k
  int prow[Core::MI];                                  // RoPE: positions of this lane's rows
  auto early = [&]() {
    if (EPI == EPI_ROPE && rotate) {
#pragma unroll
      for (int i = 0; i < Core::MI; ++i) {
        const int row = row0 + i * 16 + li;
        prow[i] = e.pos[row < M ? row : M - 1];
      }
    }
  };
  const bool npre16 = EPI == EPI_ROPE ? rotate : (PRE && wide);
  auto pre_issue = [&]() {
    if (EPI == EPI_ROPE) {
      // (cos, sin) of this lane's 4 rows x 8 rotation pairs, in the accumulator layout
      if (rotate) {
#pragma unroll
        for (int i = 0; i < Core::MI; ++i) {
          const f32x4* cs = (const f32x4*)(e.rope_tab + (long)prow[i] * 32 + g4);
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            pre[i * 2 + j][0] = cs[j * 8];             // pairs j*16 + g4 + {0, 1}
            pre[i * 2 + j][1] = cs[j * 8 + 1];         // pairs j*16 + g4 + {2, 3}
          }
        }
      }
    } else if (PRE && wide) {
      // every lane loads (addresses clamped into the matrix): the wait in the main loop counts 16 instructions
      int col = span0 + rc * 8;
      col = col < N ? col : N - 8;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        int row = row0 + rr + 8 * k;
        row = row < M ? row : M - 1;
        if (EPI == EPI_RESID_F32) {
          const float* h = e.Hin + (long)row * N + col;
          pre[k][0] = *(const f32x4*)h;
          pre[k][1] = *(const f32x4*)(h + 4);
        } else {                                       // dy columns [col, col+8) <-> a at u[64q + 8s], g at +32
          const bf16_t* u = e.U + (long)row * (2 * N) + 64 * (col >> 5) + (col & 31);
          pre[k][0] = *(const f32x4*)u;                // 8 bf16 of a
          pre[k][1] = *(const f32x4*)(u + 32);         // 8 bf16 of g
        }
      }
    }
  };
  {
    // dbg bit 0 (bench diagnostics only): every tile loads operand tile (0, 0) -> all operand traffic is L2-resident
    const int lm = (dbg & 1) ? 0 : m0, ln = (dbg & 1) ? 0 : n0;
    if (MID) Core::template mainloop_mid<true>(A, K, lm, M, B, K, ln, N, K, smem, acc, pre_issue, npre16, early);
    else Core::template mainloop<true>(A, K, lm, M, B, K, ln, N, K, smem, acc, pre_issue, npre16, early);
  }
  TRACE_T(1);
  if (dbg & 2) {                                     // dbg bit 1: skip the epilogue (keep the accumulators live)
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < Core::MI; ++i)
#pragma unroll
      for (int j = 0; j < Core::NI; ++j) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (sum == 12345.678f) e.C[0] = f2bf(sum);
    return;
  }

  if (!wide) {                                       // odd widths (tests, tiny heads): direct 8-byte stores
    if (EPI == EPI_STORE_BF16 || EPI == EPI_RESID_F32) {
#pragma unroll
      for (int i = 0; i < Core::MI; ++i) {
        const int row = row0 + i * 16 + li;
        if (row >= M) continue;
#pragma unroll
        for (int j = 0; j < Core::NI; ++j) {
          const int col = span0 + j * 16 + g4;
          if (col >= N) continue;                     // N % 4 == 0: the 4 columns are in or out together
          const long o = (long)row * N + col;
          if (EPI == EPI_STORE_BF16) {
            *(bf16x4*)(e.C + o) = pack4(acc[i][j]);
          } else {
            const f32x4 h = *(const f32x4*)(e.Hin + o), v = acc[i][j];
            *(f32x4*)(e.Hout + o) = (f32x4){h[0] + rbf(v[0]), h[1] + rbf(v[1]), h[2] + rbf(v[2]), h[3] + rbf(v[3])};
          }
        }
      }
    }
    return;                                          // the fused variants require N % 64 == 0 (launch_nt)
  }

  __syncthreads();                                   // every wave is done reading the operand tiles
  char* w = smem + wave * 16384;
  char* w2 = w + 8192;
  // ---- accumulator layout -> staging image(s) (all lane-local math happens here) ----
#pragma unroll
  for (int i = 0; i < Core::MI; ++i) {
    const int lrow = i * 16 + li, row = row0 + lrow;
    if (EPI == EPI_ROPE && rotate) {                 // rotate pairs (d, d+32)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x4 lo, hi;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float c = pre[i * 2 + j][r >> 1][(r & 1) * 2], sn = pre[i * 2 + j][r >> 1][(r & 1) * 2 + 1];
          const float x1 = rbf(acc[i][j][r]), x2 = rbf(acc[i][j + 2][r]);   // Linear output is bf16
          lo[r] = mul_rn(x1, c) - mul_rn(x2, sn);   // products rounded separately, as torch's
          hi[r] = mul_rn(x2, c) + mul_rn(x1, sn);   // q * cos + rotate_half(q) * sin (hf:196-219)
        }
        stg_put(w, lrow, j * 16 + g4, pack4(lo));
        stg_put(w, lrow, (j + 2) * 16 + g4, pack4(hi));
      }
    } else if (EPI == EPI_GEGLU_FWD) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const bf16x4 a4 = pack4(acc[i][j]), gg4 = pack4(acc[i][j + 2]);
        stg_put(w, lrow, j * 16 + g4, a4);
        stg_put(w, lrow, (j + 2) * 16 + g4, gg4);
        bf16x4 y4;
#pragma unroll
        for (int r = 0; r < 4; ++r) y4[r] = f2bf(rbf(gelu_f(bf2f(a4[r]))) * bf2f(gg4[r]));
        stg32_put(w2, lrow, j * 16 + g4, y4);
      }
    } else {
#pragma unroll
      for (int j = 0; j < Core::NI; ++j) stg_put(w, lrow, j * 16 + g4, pack4(acc[i][j]));
    }
  }
  // ---- staging image -> global, row-major ----
  const int col = span0 + rc * 8;
  if (col < N) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int lrow = rr + 8 * k, row = row0 + lrow;
      const bf16x8 v = (k & 1) ? stg_get<true>(w, lrow, rc) : stg_get<false>(w, lrow, rc);
      if (row >= M) continue;
      if (EPI == EPI_STORE_BF16 || EPI == EPI_ROPE || EPI == EPI_GEGLU_FWD) {
        *(bf16x8*)(e.C + (long)row * N + col) = v;
      } else if (EPI == EPI_RESID_F32) {
        const f32x4 h0 = pre[k][0], h1 = pre[k][1];
        float* o = e.Hout + (long)row * N + col;
        *(f32x4*)o = (f32x4){h0[0] + bf2f(v[0]), h0[1] + bf2f(v[1]), h0[2] + bf2f(v[2]), h0[3] + bf2f(v[3])};
        *(f32x4*)(o + 4) = (f32x4){h1[0] + bf2f(v[4]), h1[1] + bf2f(v[5]), h1[2] + bf2f(v[6]), h1[3] + bf2f(v[7])};
      } else {                                       // EPI_GEGLU_BWD: v = dy (bf16); du = GeGLU'(u, dy)
        const bf16x8 a8 = __builtin_bit_cast(bf16x8, pre[k][0]), g8 = __builtin_bit_cast(bf16x8, pre[k][1]);
        bf16x8 da, dg;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const float af = bf2f(a8[r]), gf = bf2f(g8[r]), df = bf2f(v[r]);
          dg[r] = f2bf(df * rbf(gelu_f(af)));
          da[r] = f2bf(rbf(df * gf) * gelu_grad_f(af));
        }
        bf16_t* o = e.C + (long)row * (2 * N) + 64 * (col >> 5) + (col & 31);
        *(bf16x8*)o = da;
        *(bf16x8*)(o + 32) = dg;
      }
    }
  }
  if (EPI == EPI_GEGLU_FWD) {                        // y [M, N/2]: 64 x 32 per wave, rows (lane >> 2) + 16k
    const int ycol = (span0 >> 1) + (lane & 3) * 8;
    if (span0 < N) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int lrow = (lane >> 2) + 16 * k, row = row0 + lrow;
        const bf16x8 v = stg32_get(w2, lrow, lane & 3);
        if (row < M) *(bf16x8*)(e.Y + (long)row * (N >> 1) + ycol) = v;
      }
    }
  }  TRACE_T(2);
}

template <int EPI>
static int launch_nt(const void* A, const void* B, int M, int N, int K, const EpiArgs& e, hipStream_t st) {
  if (M <= 0 || N <= 0 || K <= 0 || (K % 64) != 0 || (N % 4) != 0) return SNX_E_SHAPE;
  if ((EPI == EPI_ROPE || EPI == EPI_GEGLU_FWD) && (N % 64) != 0) return SNX_E_SHAPE;
  if (EPI == EPI_GEGLU_BWD && (N % 32) != 0) return SNX_E_SHAPE;
  if (!A || !B) return SNX_E_ARG;
  if (EPI == EPI_GEGLU_BWD) {
    // the pipelined 128x128 form (gemm_nt_pipe.hip): the epilogue of a tile inside the K loop of the next one
    const int rc = snx_launch_nt_pipe_geglu_bwd(A, B, M, N, K, e, st);
    if (rc != SNX_E_SHAPE) return rc;
  }
  {
    // many rows: the 256x256 persistent form (gemm_nt256.hip); SNX_E_SHAPE = not taken (small M, odd N, epilogue
    // not built there yet)
    const int rc = snx_launch_nt256(EPI, A, B, M, N, K, e, st);
    if (rc != SNX_E_SHAPE) return rc;
  }
  constexpr int BM = 128, BN = 128;
  using Core = GemmCore<BM, BN, 2, 2>;
  const int tm = cdiv(M, BM), tn = cdiv(N, BN);
  // column-group width of the tile order (gemm_core.h): keep an XCD's share of the weight matrix (cg tiles of
  // BN x K bf16) within ~1.8 MB of its L2 unless re-reading the activation panels ceil(tn / cg) times costs more
  // than letting the weights spill (estimate: spilled weights are re-fetched ~4x per XCD).
  const int cg_env = SNX_DIAG_CFG(gemm_cg, -1);
  const int dbg = SNX_DIAG_CFG(gemm_dbg, 0);
  int cg = tn;
  if (cg_env > 0) cg = cg_env < tn ? cg_env : tn;
  else if (cg_env < 0) {
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
  // MID: mid-step barrier main loop (gemm_core.h) -- measured faster for the epilogues that also stream a second
  // operand (residual, saved u), slower for the plain long-K shapes (diagnostics builds: "gemm_mid" = bitmask over EPI).
  const int mid_mask = SNX_DIAG_CFG(gemm_mid, (1 << EPI_RESID_F32) | (1 << EPI_GEGLU_BWD));
  if ((mid_mask >> EPI) & 1) {
    auto kern = gemm_nt_kernel<BM, BN, 2, 2, EPI, true>;
    hipLaunchKernelGGL(kern, dim3(tm * tn), dim3(Core::NTHREADS), Core::LDS_BYTES, st, (const bf16_t*)A,
                       (const bf16_t*)B, M, N, K, order, tm * tn, dbg, e);
  } else {
    auto kern = gemm_nt_kernel<BM, BN, 2, 2, EPI, false>;
    hipLaunchKernelGGL(kern, dim3(tm * tn), dim3(Core::NTHREADS), Core::LDS_BYTES, st, (const bf16_t*)A,
                       (const bf16_t*)B, M, N, K, order, tm * tn, dbg, e);
  }
  SNX_CHECK_LAUNCH();
  return SNX_OK;
}

extern "C" int snx_gemm_nt_bf16(const void* A, const void* B, void* C, int32_t M, int32_t N, int32_t K,
                                hipStream_t stream) {
  if (!C) return SNX_E_ARG;
  EpiArgs e{};
  e.C = (bf16_t*)C;
  return launch_nt<EPI_STORE_BF16>(A, B, M, N, K, e, stream);
}

extern "C" int snx_gemm_nt_resid(const void* A, const void* B, const float* Hin, float* Hout, int32_t M, int32_t N,
                                 int32_t K, hipStream_t stream) {
  if (!Hin || !Hout) return SNX_E_ARG;
  EpiArgs e{};
  e.Hin = Hin; e.Hout = Hout;
  return launch_nt<EPI_RESID_F32>(A, B, M, N, K, e, stream);
}

extern "C" int snx_gemm_nt_rope(const void* A, const void* B, void* C, const float* rope_tab, const int32_t* pos,
                                int32_t rope_cols, int32_t M, int32_t N, int32_t K, hipStream_t stream) {
  if (!C || !rope_tab || !pos) return SNX_E_ARG;
  if (rope_cols < 0 || rope_cols > N || (rope_cols % 64) != 0) return SNX_E_SHAPE;
  EpiArgs e{};
  e.C = (bf16_t*)C; e.rope_tab = (const f32x2*)rope_tab; e.pos = pos; e.rope_cols = rope_cols;
  return launch_nt<EPI_ROPE>(A, B, M, N, K, e, stream);
}

extern "C" int snx_gemm_nt_rope_rows(const void* A, const void* B, void* C, const float* rope_tab, const int32_t* pos,
                                     const float* rope_rows, int32_t rope_cols, int32_t M, int32_t N, int32_t K,
                                     hipStream_t stream) {
  if (!C || !rope_tab || !pos) return SNX_E_ARG;
  if (rope_cols < 0 || rope_cols > N || (rope_cols % 64) != 0) return SNX_E_SHAPE;
  EpiArgs e{};
  e.C = (bf16_t*)C; e.rope_tab = (const f32x2*)rope_tab; e.pos = pos; e.rope_cols = rope_cols;
  e.rope_rows = (const f32x2*)rope_rows;
  return launch_nt<EPI_ROPE>(A, B, M, N, K, e, stream);
}

extern "C" int snx_gemm_nt_geglu_fwd(const void* A, const void* B_interleaved, void* U, void* Y, int32_t M,
                                     int32_t N, int32_t K, hipStream_t stream) {
  if (!U || !Y) return SNX_E_ARG;
  EpiArgs e{};
  e.C = (bf16_t*)U; e.Y = (bf16_t*)Y;
  return launch_nt<EPI_GEGLU_FWD>(A, B_interleaved, M, N, K, e, stream);
}

extern "C" int snx_gemm_nt_geglu_bwd(const void* A, const void* B, const void* U, void* dU, int32_t M, int32_t N,
                                     int32_t K, hipStream_t stream) {
  if (!U || !dU) return SNX_E_ARG;
  EpiArgs e{};
  e.C = (bf16_t*)dU; e.U = (const bf16_t*)U;
  return launch_nt<EPI_GEGLU_BWD>(A, B, M, N, K, e, stream);
}

// ------------------------------------------------------------------------------------------
// Weight-gradient GEMM ("TN"):  dW[N,K] += dY[M,N]^T * X[M,K]   (contraction over tokens)
// Both operands are token-major, i.e. strided along the contraction index, so the 64-token x
// 128-column tiles are staged row-major with LDS-DMA (256-B rows, chunk index XOR-swizzled with
// ((row&3)<<2 | (row>>2)&3), applied on the source address and on the read) and the MFMA
// fragments are fetched with ds_read_b64_tr_b16, the hardware transposing LDS read (two per
// 8-element fragment).  The token range is split over `splits` workgroups per output tile
// (few, large output tiles would leave most CUs idle); partial sums are added to the fp32
// gradient with float atomics -- which is also the "+=" the gradient buffer needs.
// ------------------------------------------------------------------------------------------
__device__ uint4 snx_zero_page[16];   // 256 B of zeros: DMA source for token rows past M

// stage 64 token rows x 128 columns (bf16) of G (leading dim ld), rows m0.., columns c0..
__device__ __forceinline__ void tn_stage(const bf16_t* __restrict__ G, long ld, int m0, int M, int c0, char* lds,
                                         int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ci = i * 4 + wave;                    // 4-row group of this wave instruction
    const int row = ci * 4 + (lane >> 4);
    const int ch = tn_chunk(row, lane & 15);
    const int gm = m0 + row;
    const bf16_t* src = gm < M ? G + (long)gm * ld + c0 + ch * 8 : (const bf16_t*)snx_zero_page + (lane & 15) * 8;
    __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(lds + ci * 1024), 16, 0, 0);
  }
}

// flush modes of the token-split kernel below: TN_ATOMIC = float atomics into dW (arrival order; "det_reduce" = 0),
// TN_DIRECT = plain dW += acc (ONE split: a single writer per element), TN_SLAB = plain store of the partial into slab
// (split, tile) of the caller's workspace, added to dW in split order by tn128_reduce_kernel
enum { TN_ATOMIC = 0, TN_DIRECT = 1, TN_SLAB = 2 };

__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(TnGroup grp, int M, int ntiles, int rows_per_split, int mode,
                                                         float* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int TILE_BYTES = 64 * 256, STAGE_BYTES = 2 * TILE_BYTES;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-contiguous order over the (token split, output tile) sequence: the workgroups that share an
  // XCD's L2 work on the SAME token range (neighbouring tiles), so each slice of dY / X is pulled
  // from HBM by one XCD only (measured before: 4x the algorithmic bytes, ~5 TB/s of fabric traffic).
  const int work = xcd_remap(blockIdx.x, gridDim.x);
  int tile_id = work % ntiles;
  const int tile_global = tile_id;
  const int split = work / ntiles;
  int p = 0;
#pragma unroll
  for (int q = 0; q < SNX_TN_MAX_GROUP - 1; ++q)
    if (q + 1 < grp.nprob && tile_id >= grp.tile_end[q]) p = q + 1;
  if (p > 0) tile_id -= grp.tile_end[p - 1];
  const bf16_t* __restrict__ dY = grp.dY[p];
  const bf16_t* __restrict__ X = grp.X[p];
  float* __restrict__ dW = grp.dW[p];
  const int N = grp.N[p], K = grp.K[p], interleave_I = grp.inter[p];
  const int tiles_k = K / 128;
  const int n0 = (tile_id / tiles_k) * 128, k0 = (tile_id % tiles_k) * 128;
  const int m_begin = split * rows_per_split;
  const int m_end = min(M, m_begin + rows_per_split);
  if (m_begin >= m_end) return;
  const int nsteps = (m_end - m_begin + 63) / 64;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  tn_stage(dY, N, m_begin, m_end, n0, smem, wave, lane);
  tn_stage(X, K, m_begin, m_end, k0, smem + TILE_BYTES, wave, lane);
  for (int t = 0; t < nsteps; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    char* cur = smem + (t & 1) * STAGE_BYTES;
    if (t + 1 < nsteps) {
      char* nxt = smem + ((t + 1) & 1) * STAGE_BYTES;
      tn_stage(dY, N, m_begin + (t + 1) * 64, m_end, n0, nxt, wave, lane);
      tn_stage(X, K, m_begin + (t + 1) * 64, m_end, k0, nxt + TILE_BYTES, wave, lane);
    }
    // all 32 transposed fragment reads (64 ds_read_b64_tr_b16) up front, then 1 MFMA : 2 reads
    bf16x8 a[2][4], b[2][4];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int i = 0; i < 4; ++i) a[kk][i] = tn_frag(cur, kk * 32, wm * 64 + i * 16, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) b[kk][j] = tn_frag(cur + TILE_BYTES, kk * 32, wn * 64 + j * 16, lane);
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kk][i], b[kk][j], acc[i][j], 0, 0, 0);
#if SNX_GEMM_SCHED
#pragma unroll
    for (int q = 0; q < 16; ++q) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
#endif
  }
  // acc[i][j][r] = dW[n0 + wm*64 + i*16 + 4g + r][k0 + wn*64 + j*16 + (lane&15)]
  if (mode == TN_SLAB) {
    float* slab = ws + ((size_t)split * ntiles + tile_global) * (128 * 128);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int nl = wm * 64 + i * 16 + (lane >> 4) * 4 + r;
#pragma unroll
        for (int j = 0; j < 4; ++j) slab[nl * 128 + wn * 64 + j * 16 + (lane & 15)] = acc[i][j][r];
      }
    return;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int n = n0 + wm * 64 + i * 16 + (lane >> 4) * 4 + r;
      if (interleave_I > 0)                           // dY columns are in the interleaved GeGLU order
        n = ((n & 63) < 32) ? 32 * (n >> 6) + (n & 31) : interleave_I + 32 * (n >> 6) + (n & 31);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = k0 + wn * 64 + j * 16 + (lane & 15);
        if (mode == TN_ATOMIC) atomicAdd(dW + (long)n * K + k, acc[i][j][r]);
        else dW[(long)n * K + k] += acc[i][j][r];
      }
    }
}

// dW tile += slab(split 0) + slab(split 1) + ...: the FIXED order that makes the token-split weight gradient
// bit-reproducible.  One workgroup per (tile, 32-row chunk), a thread owns 4 consecutive k of 4 rows.
__global__ __launch_bounds__(256) void tn128_reduce_kernel(TnGroup grp, int ntiles, int splits, const float* __restrict__ ws) {
  int tile_id = blockIdx.x;
  const int tile_global = tile_id;
  int p = 0;
#pragma unroll
  for (int q = 0; q < SNX_TN_MAX_GROUP - 1; ++q)
    if (q + 1 < grp.nprob && tile_id >= grp.tile_end[q]) p = q + 1;
  if (p > 0) tile_id -= grp.tile_end[p - 1];
  float* __restrict__ dW = grp.dW[p];
  const int K = grp.K[p], interleave_I = grp.inter[p];
  const int tiles_k = K / 128;
  const int n0 = (tile_id / tiles_k) * 128, k0 = (tile_id % tiles_k) * 128;
  const int col = (threadIdx.x & 31) * 4;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int nl = blockIdx.y * 32 + q * 8 + (threadIdx.x >> 5);
    int n = n0 + nl;
    if (interleave_I > 0) n = ((n & 63) < 32) ? 32 * (n >> 6) + (n & 31) : interleave_I + 32 * (n >> 6) + (n & 31);
    float* dst = dW + (long)n * K + k0 + col;
    f32x4 a = *(const f32x4*)dst;
    const float* src = ws + (size_t)tile_global * (128 * 128) + nl * 128 + col;
    for (int sp = 0; sp < splits; ++sp) a += *(const f32x4*)(src + (size_t)sp * ntiles * (128 * 128));
    *(f32x4*)dst = a;
  }
}

// splits of the token range: whole rounds of the 512 resident workgroup slots (2 x 64 KiB LDS per CU on 256
// CUs) are what counts; among the split counts that keep >= 256 tokens per workgroup pick the best-filled one,
// preferring fewer splits (less float-atomic traffic) when the fill is within 2 %.
static int tn_pick_splits(int tiles, int M) {
  const int forced = SNX_DIAG_CFG(tn_splits, 0);
  const int max_splits = cdiv(M, 256) < 1 ? 1 : cdiv(M, 256);
  if (forced > 0) return forced < max_splits ? forced : max_splits;
  int best = 1;
  double best_fill = 0;
  for (int s = 1; s <= max_splits && s <= 32 && (long)tiles * s <= 8L * 512; ++s) {
    const long w = (long)tiles * s;
    const double fill = (double)w / ((double)cdiv(w, 512) * 512);
    if (fill > best_fill + 0.02) { best_fill = fill; best = s; }
  }
  return best;
}

// token splits and rows per split of the 128x128 kernel for a group of `tiles` output tiles
static void tn128_plan(int tiles, int M, int& splits, int& rows) {
  splits = tn_pick_splits(tiles, M);
  rows = cdiv(M, splits);
  rows = ((rows + 63) / 64) * 64;
  splits = cdiv(M, rows);
}

// Bytes of partial slabs the ordered reduction of this group may need: the maximum over the process switches that
// change the schedule ("tn256", "tn256_min_m", the reserved CUs), so that a workspace sized once stays large enough when a
// test or an A/B flips one of them.
static size_t tn_group_ws_bytes(const TnGroup& g, int M) {
  size_t need = snx_tn256_ws_bound(g, M & ~63);
  const int tiles = g.tile_end[g.nprob - 1];
  int splits, rows;
  tn128_plan(tiles, M, splits, rows);
  if (splits > 1) need = max(need, (size_t)splits * tiles * (128 * 128 * 4));
  return need;
}

static int launch_tn_group(const TnGroup& g, int M, void* ws, size_t ws_bytes, hipStream_t st) {
  const bool det = g_snx_cfg.det_reduce != 0;
  // long token ranges: the 256x256 persistent form (gemm_tn256.hip); "tn256" = 0 keeps the 128x128 kernel
  const int tn256 = g_snx_cfg.tn256, tn256_min_m = g_snx_cfg.tn256_min_m;
  if (tn256 && M >= tn256_min_m) {
    // whole 64-row K-steps there; a ragged rest of the token range (< 64 rows) comes back to this kernel
    const int M64 = M & ~63;
    const int rc = snx_launch_tn256(g, M64, ws, ws_bytes, st);
    if (rc == SNX_OK) {
      if (M64 == M) return SNX_OK;
      TnGroup rest = g;
      for (int p = 0; p < g.nprob; ++p) {
        rest.dY[p] = g.dY[p] + (long)M64 * g.N[p];
        rest.X[p] = g.X[p] + (long)M64 * g.K[p];
      }
      const int tiles = g.tile_end[g.nprob - 1];
      // one split: a single writer per element, behind the reduction above in stream order
      hipLaunchKernelGGL(gemm_tn_kernel, dim3(tiles), dim3(256), 2 * 2 * 64 * 256, st, rest, M - M64, tiles, 64,
                         det ? TN_DIRECT : TN_ATOMIC, (float*)nullptr);
      SNX_CHECK_LAUNCH();
      return SNX_OK;
    }
    if (rc != SNX_E_SHAPE) return rc;                 // groups it does not take (> 256 tiles): fall through
  }
  const int tiles = g.tile_end[g.nprob - 1];
  int splits, rows;
  tn128_plan(tiles, M, splits, rows);
  int mode = TN_ATOMIC;
  if (det) {
    mode = splits > 1 ? TN_SLAB : TN_DIRECT;
    if (mode == TN_SLAB && (!ws || ws_bytes < (size_t)splits * tiles * (128 * 128 * 4))) return SNX_E_ARG;
  }
  hipLaunchKernelGGL(gemm_tn_kernel, dim3(tiles * splits), dim3(256), 2 * 2 * 64 * 256, st, g, M, tiles, rows, mode,
                     (float*)ws);
  SNX_CHECK_LAUNCH();
  if (mode == TN_SLAB) {
    hipLaunchKernelGGL(tn128_reduce_kernel, dim3(tiles, 4), dim3(256), 0, st, g, tiles, splits, (const float*)ws);
    SNX_CHECK_LAUNCH();
  }
  return SNX_OK;
}

static int tn_one(TnGroup& g, const void* dY, const void* X, float* dW, int32_t M, int32_t N, int32_t K, int interleave_I) {
  if (!dY || !X || !dW) return SNX_E_ARG;
  if (M <= 0 || N <= 0 || K <= 0 || (N % 128) || (K % 128)) return SNX_E_SHAPE;
  if (interleave_I < 0 || (interleave_I > 0 && 2 * interleave_I != N)) return SNX_E_SHAPE;
  g = TnGroup{};
  g.dY[0] = (const bf16_t*)dY; g.X[0] = (const bf16_t*)X; g.dW[0] = dW;
  g.N[0] = N; g.K[0] = K; g.inter[0] = interleave_I;
  g.tile_end[0] = (N / 128) * (K / 128);
  g.nprob = 1;
  return SNX_OK;
}

static int tn_group(TnGroup& g, const snx_tn_problem* probs, int32_t nprob, int32_t M, bool need_ptrs) {
  if (!probs || nprob < 1 || nprob > SNX_TN_MAX_GROUP || M <= 0) return SNX_E_ARG;
  g = TnGroup{};
  int run = 0;
  for (int p = 0; p < nprob; ++p) {
    const snx_tn_problem& q = probs[p];
    if (need_ptrs && (!q.dY || !q.X || !q.dW)) return SNX_E_ARG;
    if (q.N <= 0 || q.K <= 0 || (q.N % 128) || (q.K % 128)) return SNX_E_SHAPE;
    g.dY[p] = (const bf16_t*)q.dY; g.X[p] = (const bf16_t*)q.X; g.dW[p] = q.dW;
    g.N[p] = q.N; g.K[p] = q.K; g.inter[p] = q.interleaved ? q.N / 2 : 0;
    run += (q.N / 128) * (q.K / 128);
    g.tile_end[p] = run;
  }
  g.nprob = nprob;
  return SNX_OK;
}

extern "C" size_t snx_gemm_tn_workspace_bytes(const snx_tn_problem* probs, int32_t nprob, int32_t M) {
  TnGroup g;
  if (tn_group(g, probs, nprob, M, false) != SNX_OK) return 0;
  return tn_group_ws_bytes(g, M);
}

extern "C" int snx_gemm_tn_accum_group(const snx_tn_problem* probs, int32_t nprob, int32_t M, void* ws, size_t ws_bytes,
                                       hipStream_t st) {
  TnGroup g;
  if (const int rc = tn_group(g, probs, nprob, M, true)) return rc;
  return launch_tn_group(g, M, ws, ws_bytes, st);
}

extern "C" int snx_gemm_tn_accum(const void* dY, const void* X, float* dW, int32_t M, int32_t N, int32_t K, void* ws,
                                 size_t ws_bytes, hipStream_t st) {
  TnGroup g;
  if (const int rc = tn_one(g, dY, X, dW, M, N, K, 0)) return rc;
  return launch_tn_group(g, M, ws, ws_bytes, st);
}

// dY [M, 2I] in the interleaved GeGLU column order -> dW rows in the natural Wi order
extern "C" int snx_gemm_tn_accum_interleaved(const void* dY, const void* X, float* dW, int32_t M, int32_t N,
                                             int32_t K, void* ws, size_t ws_bytes, hipStream_t st) {
  TnGroup g;
  if (const int rc = tn_one(g, dY, X, dW, M, N, K, N / 2)) return rc;
  return launch_tn_group(g, M, ws, ws_bytes, st);
}
