// Shared pieces of the weight-gradient ("TN") GEMMs: dW[N,K] += dY[M,N]^T * X[M,K], contraction over tokens.
// LDS image of one operand sub-tile: 64 token rows x 128 columns (bf16), rows of 256 B; the 16-B chunk index is
// XOR-swizzled with ((row&3)<<2 | (row>>2)&3) on the DMA source address and on the read, so that the transposing
// fragment reads (ds_read_b64_tr_b16: 8 rows x 32 B per 32-lane group) touch all 64 banks once.
#pragma once
#include "common.h"

__device__ __forceinline__ int tn_chunk(int row, int ch) { return ch ^ (((row & 3) << 2) | ((row >> 2) & 3)); }

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

// Up to SNX_TN_MAX_GROUP weight-gradient problems that share the token dimension M (the four Linears of one
// encoder layer) run as ONE launch: their output tiles are concatenated, so that tiles x splits can fill whole
// rounds of the 512 resident workgroups (a single Wqkv / Wi problem has 108 tiles: 4 splits = 432 workgroups,
// 84 % of one round; the layer's 306 tiles x 5 splits = 1530 = 99.6 % of three rounds) and only the last
// round's float atomics are exposed.
#define SNX_TN_MAX_GROUP 4
struct TnGroup {
  const bf16_t* dY[SNX_TN_MAX_GROUP];
  const bf16_t* X[SNX_TN_MAX_GROUP];
  float* dW[SNX_TN_MAX_GROUP];
  int N[SNX_TN_MAX_GROUP], K[SNX_TN_MAX_GROUP], inter[SNX_TN_MAX_GROUP];
  int tile_end[SNX_TN_MAX_GROUP];       // running tile count after problem p
  int nprob;
};


// 256x256 persistent form (gemm_tn256.hip): SNX_OK, SNX_E_SHAPE when it does not take the group (more than 256
// tiles), SNX_E_ARG when the ordered reduction ("det_reduce", default) finds `ws` missing or smaller than
// snx_tn256_ws_bytes(), or a HIP error code.  M % 64 == 0.
int snx_launch_tn256(const TnGroup& g, int M, void* ws, size_t ws_bytes, hipStream_t st);
size_t snx_tn256_ws_bytes(const TnGroup& g, int M);    // with the CUs reserved right now
size_t snx_tn256_ws_bound(const TnGroup& g, int M);    // maximum over every reservation (0..128)
