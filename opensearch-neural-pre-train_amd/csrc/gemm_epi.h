// Epilogue vocabulary shared by the encoder's NT GEMM kernels (gemm.hip: 128x128 tiles, two workgroups per CU;
// gemm_nt256.hip: 256x256 persistent tiles) and the staging images their coalesced write-backs go through.
#pragma once
#include "common.h"

enum { EPI_STORE_BF16 = 0, EPI_RESID_F32 = 1, EPI_ROPE = 2, EPI_GEGLU_FWD = 3, EPI_GEGLU_BWD = 4 };

struct EpiArgs {
  bf16_t* C;                 // STORE / ROPE: [M,N];  GEGLU_FWD: u [M,N] interleaved;  GEGLU_BWD: du [M,2N] interleaved
  const float* Hin;          // RESID
  float* Hout;               // RESID
  const f32x2* rope_tab;     // ROPE: [max_pos][32] (cos, sin)
  const int32_t* pos;        // ROPE: [M] position of each row
  const f32x2* rope_rows;    // ROPE, optional: [M][32] = rope_tab[pos[row]] resolved once per pass (snx_rope_rows): the
                             // 256x256 kernel then reads a row's (cos, sin) without the dependent position load
  int rope_cols;             // ROPE: columns < rope_cols (= 2*hidden) are rotated
  bf16_t* Y;                 // GEGLU_FWD: y [M, N/2]
  const bf16_t* U;           // GEGLU_BWD: u [M, 2N] interleaved
};

__device__ __forceinline__ bf16x4 pack4(const f32x4 v) { return (bf16x4){f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])}; }

// ---- coalesced epilogue -----------------------------------------------------------------------
// In the (transposed) accumulator layout the 16 lanes li = 0..15 of a store instruction are 16
// different ROWS, so a direct store is 64 separate 8-byte transactions per instruction and the
// texture addresser, not HBM, sets the epilogue time (measured on the 256x256 kernel: 786 TFLOP/s
// with such stores, 1047 without, no change when all stores hit one L2-resident region).  Each
// wave therefore bounces its bf16 result through a PRIVATE piece of LDS and writes it back
// row-major, 16 B per lane, 8 lanes per 128-byte line.
// Staging image: rows of 128 B (64 columns); 16-B chunk c of row r sits at chunk c ^ (r & 7) and its two
// 8-B halves are swapped when bit 3 of r is set -- ds_write_b64 from the accumulator layout and
// ds_read_b128 in the row-major layout are both bank-conflict free.
__device__ __forceinline__ void stg_put(char* w, int row, int col, bf16x4 v) {        // col % 4 == 0
  *(bf16x4*)(w + row * 128 + ((((col >> 3) ^ row) & 7) << 4) + ((((col >> 2) ^ (row >> 3)) & 1) << 3)) = v;
}
// the 8-byte piece (columns col .. col + 3, col % 4 == 0) of a row: what stg_put wrote there
__device__ __forceinline__ bf16x4 stg_get4(const char* w, int row, int col) {
  return *(const bf16x4*)(w + row * 128 + ((((col >> 3) ^ row) & 7) << 4) + ((((col >> 2) ^ (row >> 3)) & 1) << 3));
}
template <bool SWAP>
__device__ __forceinline__ bf16x8 stg_get(const char* w, int row, int chunk) {
  const bf16x8 v = *(const bf16x8*)(w + row * 128 + (((chunk ^ row) & 7) << 4));
  if (!SWAP) return v;
  return (bf16x8){v[4], v[5], v[6], v[7], v[0], v[1], v[2], v[3]};
}
// second staging image for a 32-column result (GeGLU's y): rows of 64 B
__device__ __forceinline__ void stg32_put(char* w, int row, int col, bf16x4 v) {
  *(bf16x4*)(w + row * 64 + ((((col >> 3) ^ (row >> 1)) & 3) << 4) + ((((col >> 2) ^ (row >> 3)) & 1) << 3)) = v;
}
__device__ __forceinline__ bf16x8 stg32_get(const char* w, int row, int chunk) {
  const bf16x8 v = *(const bf16x8*)(w + row * 64 + (((chunk ^ (row >> 1)) & 3) << 4));
  return ((row >> 3) & 1) ? (bf16x8){v[4], v[5], v[6], v[7], v[0], v[1], v[2], v[3]} : v;
}

// 256x256 persistent kernel (gemm_nt256.hip).  SNX_OK, SNX_E_SHAPE when it does not take the shape (the caller
// falls back to the 128x128 kernel), or a HIP error code.
int snx_launch_nt256(int epi, const void* A, const void* B, int M, int N, int K, const EpiArgs& e, hipStream_t st);
// pipelined 128x128 form of the GeGLU-backward GEMM (gemm_nt_pipe.hip): same contract
int snx_launch_nt_pipe_geglu_bwd(const void* A, const void* B, int M, int N, int K, const EpiArgs& e, hipStream_t st);
