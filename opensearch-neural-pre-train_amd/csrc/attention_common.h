// Device helpers shared by the streaming (attention.hip) and the sequence-resident
// (attention_unit.hip) attention kernels.
#pragma once
#include "common.h"

#define NEG_BIG (-1.0e30f)
#define LOG2E 1.4426950408889634f
#define LN2 0.6931471805599453f

// 2^x on the transcendental unit (v_exp_f32); softmax runs in the log2 domain so that each
// probability costs one FMA + one v_exp.
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }


// true when every (query, key) pair of a 16-query x 64-key (or 64-query x 16-key) block lies inside
// the band |q - k| <= window, so the per-element band test can be skipped (wave-uniform).
__device__ __forceinline__ bool band_clean(int window, int a_lo, int a_hi, int b_lo, int b_hi) {
  return window < 0 || (a_hi - b_lo <= window && b_hi - a_lo <= window);
}


__device__ __forceinline__ int k_off(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }
__device__ __forceinline__ int v_off(int row, int chunk) { return row * 128 + ((chunk ^ (((row >> 1) & 3) << 1)) << 4); }


__device__ __forceinline__ bf16x4 lds_tr16(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(p));
}


// A-operand fragment of the TRANSPOSE of a [64 rows][64 d] tile (v_off image): MFMA row index =
// d (16*dt + lane&15), contraction elements = tile rows 16*(2c + j/4) + 4g + j%4.
__device__ __forceinline__ bf16x8 frag_tr(const char* tile, int dt, int c, int lane) {
  const int g = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
  const int r0 = 16 * (2 * c) + 4 * g + tq, r1 = r0 + 16;
  const int chunk = 2 * dt + (tp >> 1);
  const bf16x4 a0 = lds_tr16(tile + v_off(r0, chunk) + (tp & 1) * 8);
  const bf16x4 a1 = lds_tr16(tile + v_off(r1, chunk) + (tp & 1) * 8);
  return (bf16x8){a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
}

// Epilogue of dQ / dK / dV: this lane holds grad[d = 16*dt + 4g + r] of one token row.  With a
// RoPE table the transposed rotation (backward of hf:196-219) is applied to the bf16-rounded
// gradient before the store: the pair (d, d+32) lives in accumulators dt and dt+2 of the same lane.
__device__ __forceinline__ void store_grad_rows(bf16_t* orow, const f32x4 (&acc)[4], float scale,
                                                const f32x2* __restrict__ rope_tab, int p, int g) {
  if (rope_tab) {
    const f32x2* cs = rope_tab + (long)p * 32 + g * 4;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      bf16x4 lo, hi;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const f32x2 t = cs[dt * 16 + r];
        const float y1 = rbf(acc[dt][r] * scale), y2 = rbf(acc[dt + 2][r] * scale);
        lo[r] = f2bf(y1 * t[0] + y2 * t[1]);
        hi[r] = f2bf(y2 * t[0] - y1 * t[1]);
      }
      *(bf16x4*)(orow + dt * 16) = lo;
      *(bf16x4*)(orow + (dt + 2) * 16) = hi;
    }
  } else {
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
      *(bf16x4*)(orow + dt * 16) = (bf16x4){f2bf(acc[dt][0] * scale), f2bf(acc[dt][1] * scale),
                                            f2bf(acc[dt][2] * scale), f2bf(acc[dt][3] * scale)};
  }
}


// Block -> (group, index inside the group) for launches that serve several sequence groups (e.g. the fused micro-step's
// 64-token queries and 256-token documents).  `bend[i]` = exclusive prefix end of group i's blocks, n groups, NG = array
// size.  interleave = 0: group by group (longest first, as the callers order them).  interleave = 1 (round 6): the groups'
// blocks interleaved in proportion to their counts -- among the first b blocks of groups 0..i exactly floor(b n_i / N_i)
// belong to group i (peeled from the last group) -- so that blocks of different length run side by side and the CUs
// drift out of phase: in group-major order every CU is in its unit prologue (a burst of global loads with nothing else
// resident to cover it) at the same time, HBM saturated for the burst and idle through the loops.  A bijection for any
// counts; which block computes which unit changes no unit's arithmetic.
template <int NG>
__device__ __forceinline__ void block_to_group(const int (&bend)[NG], int n, int interleave, int b, int& g, int& idx) {
  if (!interleave) {
    g = 0;
    int b0 = 0;
#pragma unroll
    for (int i = 0; i < NG - 1; ++i)
      if (i + 1 < n && b >= bend[i]) { g = i + 1; b0 = bend[i]; }
    idx = b - b0;
    return;
  }
  g = 0;
  idx = b;
#pragma unroll
  for (int i = NG - 1; i >= 1; --i) {
    if (i >= n) continue;
    const long ni = bend[i] - bend[i - 1], Ni = bend[i];
    const int before = (int)((long)idx * ni / Ni), upto = (int)(((long)idx + 1) * ni / Ni);
    if (upto > before) { g = i; idx = before; return; }
    idx -= before;
  }
}
