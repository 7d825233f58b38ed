// Encoder GEMMs (K3/K6/K7/K8/K9 of SURVEY.md §2.3 and their dX counterparts):
//   C[M,N] = A[M,K] * B[N,K]^T, bf16 operands, fp32 MFMA accumulation, fused epilogues.
// Replaces the cuBLAS calls behind nn.Linear under autocast(bf16)
// (transformers modeling_modernbert.py:271,300,90-91,490 via ref:src/model/splade_modern.py:69).
#include "gemm_core.h"
#include "snx.h"

enum { EPI_STORE_BF16 = 0, EPI_RESID_F32 = 1 };

template <int BM, int BN, int WAVES_M, int WAVES_N, int EPI, int STAGES>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64) void gemm_nt_kernel(
    const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, int M, int N, int K, int tiles_n, int ntiles,
    bf16_t* __restrict__ Cb, const float* __restrict__ Hin, float* __restrict__ Hout) {
  using Core = GemmCore<BM, BN, WAVES_M, WAVES_N>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tile = xcd_remap(blockIdx.x, ntiles);
  const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
  f32x4 acc[Core::MI][Core::NI];
#pragma unroll
  for (int i = 0; i < Core::MI; ++i)
#pragma unroll
    for (int j = 0; j < Core::NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (STAGES == 3) Core::template mainloop3<true>(A, K, m0, M, B, K, n0, N, K, smem, acc);
  else Core::template mainloop<true>(A, K, m0, M, B, K, n0, N, K, smem, acc);
  // transposed accumulators: this lane owns columns col..col+3 of row `row` (8-B / 16-B accesses)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
#pragma unroll
  for (int i = 0; i < Core::MI; ++i) {
    const int row = m0 + wm * Core::WTM + i * 16 + (lane & 15);
    if (row >= M) continue;
#pragma unroll
    for (int j = 0; j < Core::NI; ++j) {
      const int col = n0 + wn * Core::WTN + j * 16 + (lane >> 4) * 4;
      if (col >= N) continue;                       // N % 4 == 0: the 4 columns are in or out together
      const long o = (long)row * N + col;
      const f32x4 v = acc[i][j];
      if (EPI == EPI_STORE_BF16) {
        *(bf16x4*)(Cb + o) = (bf16x4){f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
      } else {                                      // Linear output is bf16, residual stream fp32
        const f32x4 h = *(const f32x4*)(Hin + o);
        *(f32x4*)(Hout + o) = (f32x4){h[0] + rbf(v[0]), h[1] + rbf(v[1]), h[2] + rbf(v[2]), h[3] + rbf(v[3])};
      }
    }
  }
}

static int g_nt_variant = -1;     // -1: auto; 0: 128x128 2-stage; 1: 128x128 3-stage; 2: 256x128 3-stage (tuning knob)
extern "C" int snx_debug_set_gemm_variant(int32_t v) { g_nt_variant = v; return SNX_OK; }

template <int BM, int BN, int WM, int WN, int EPI, int STAGES>
static int launch_nt_cfg(const void* A, const void* B, int M, int N, int K, void* Cb, const float* Hin, float* Hout,
                         hipStream_t st) {
  using Core = GemmCore<BM, BN, WM, WN>;
  const int tm = cdiv(M, BM), tn = cdiv(N, BN);
  auto kern = gemm_nt_kernel<BM, BN, WM, WN, EPI, STAGES>;
  const int lds = STAGES == 3 ? Core::LDS_BYTES3 : Core::LDS_BYTES;
  static bool attr_done = false;
  if (!attr_done && lds > 64 * 1024) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3(tm * tn), dim3(Core::NTHREADS), lds, st, (const bf16_t*)A, (const bf16_t*)B, M, N, K,
                     tn, tm * tn, (bf16_t*)Cb, Hin, Hout);
  SNX_CHECK_LAUNCH();
  return SNX_OK;
}

template <int EPI>
static int launch_nt(const void* A, const void* B, int M, int N, int K, void* Cb, const float* Hin, float* Hout,
                     hipStream_t st) {
  if (M <= 0 || N <= 0 || K <= 0 || (K % 64) != 0 || (N % 4) != 0) return SNX_E_SHAPE;
  if (!A || !B) return SNX_E_ARG;
  int v = g_nt_variant;
  if (v < 0) v = 0;
  if (v == 2) return launch_nt_cfg<256, 128, 4, 2, EPI, 3>(A, B, M, N, K, Cb, Hin, Hout, st);
  if (v == 1) return launch_nt_cfg<128, 128, 2, 2, EPI, 3>(A, B, M, N, K, Cb, Hin, Hout, st);
  return launch_nt_cfg<128, 128, 2, 2, EPI, 2>(A, B, M, N, K, Cb, Hin, Hout, st);
}

extern "C" int snx_gemm_nt_bf16(const void* A, const void* B, void* C, int32_t M, int32_t N, int32_t K,
                                hipStream_t stream) {
  if (!C) return SNX_E_ARG;
  return launch_nt<EPI_STORE_BF16>(A, B, M, N, K, C, nullptr, nullptr, stream);
}

extern "C" int snx_gemm_nt_resid(const void* A, const void* B, const float* Hin, float* Hout, int32_t M, int32_t N,
                                 int32_t K, hipStream_t stream) {
  if (!Hin || !Hout) return SNX_E_ARG;
  return launch_nt<EPI_RESID_F32>(A, B, M, N, K, nullptr, Hin, Hout, stream);
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

__device__ __forceinline__ int tn_chunk(int row, int ch) { return ch ^ (((row & 3) << 2) | ((row >> 2) & 3)); }

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

// fragment for MFMA row/col index (cbase + lane&15), contraction elements m = mb + 8g + 0..7
__device__ __forceinline__ bf16x8 tn_frag(const char* tile, int mb, int cbase, int lane) {
  const int g = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
  const int ch = (cbase >> 3) + (tp >> 1);
  const int r0 = mb + 8 * g + tq, r1 = r0 + 4;
  const bf16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
      (__attribute__((address_space(3))) bf16x4*)(tile + r0 * 256 + tn_chunk(r0, ch) * 16 + (tp & 1) * 8));
  const bf16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
      (__attribute__((address_space(3))) bf16x4*)(tile + r1 * 256 + tn_chunk(r1, ch) * 16 + (tp & 1) * 8));
  return (bf16x8){a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
}

__global__ __launch_bounds__(256) void gemm_tn_kernel(const bf16_t* __restrict__ dY, const bf16_t* __restrict__ X,
                                                      float* __restrict__ dW, int M, int N, int K, int tiles_k,
                                                      int rows_per_split) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int TILE_BYTES = 64 * 256, STAGE_BYTES = 2 * TILE_BYTES;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int n0 = (blockIdx.x / tiles_k) * 128, k0 = (blockIdx.x % tiles_k) * 128;
  const int m_begin = blockIdx.y * rows_per_split;
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
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = tn_frag(cur, kk * 32, wm * 64 + i * 16, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = tn_frag(cur + TILE_BYTES, kk * 32, wn * 64 + j * 16, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
  // acc[i][j][r] = dW[n0 + wm*64 + i*16 + 4g + r][k0 + wn*64 + j*16 + (lane&15)]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + wm * 64 + i * 16 + (lane >> 4) * 4 + r;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = k0 + wn * 64 + j * 16 + (lane & 15);
        atomicAdd(dW + (long)n * K + k, acc[i][j][r]);
      }
    }
}

extern "C" int snx_gemm_tn_accum(const void* dY, const void* X, float* dW, int32_t M, int32_t N, int32_t K,
                                 hipStream_t st) {
  if (!dY || !X || !dW) return SNX_E_ARG;
  if (M <= 0 || N <= 0 || K <= 0 || (N % 128) || (K % 128)) return SNX_E_SHAPE;
  const int tiles = (N / 128) * (K / 128);
  // token-range splits: fill (at most) the 512 resident workgroup slots (2 x 64 KiB LDS per CU on
  // 256 CUs); more splits than that only add float-atomic traffic
  int splits = 512 / tiles;
  const int max_splits = cdiv(M, 256);
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  int rows = cdiv(M, splits);
  rows = ((rows + 63) / 64) * 64;
  splits = cdiv(M, rows);
  hipLaunchKernelGGL(gemm_tn_kernel, dim3(tiles, splits), dim3(256), 2 * 2 * 64 * 256, st, (const bf16_t*)dY,
                     (const bf16_t*)X, dW, M, N, K, K / 128, rows);
  SNX_CHECK_LAUNCH();
  return SNX_OK;
}
