// fp32 execution of the whole path: what the reference computes when SPLADEModernBERT.forward runs OUTSIDE
// torch.autocast -- its inference encoder (ref:benchmark/encoders.py:309-345), any bare model(...) call
// (ref:src/model/splade_modern.py:50-88) and the tolerance protocol's fp32 leg (<= 1e-5 abs against the
// reference CPU path, top-k indices exact).  Same operators, same order, NO bf16 cast point anywhere: fp32 weights
// (the nn.Parameters themselves, no cache), fp32 activations, fp32 contraction on the matrix cores
// (v_mfma_f32_32x32x2_f32: bit for bit a k-ordered fmaf chain, 1/16 of the bf16 rate -- a 149 M model's inference
// does not need more).  Shapes are generic (any hidden size, head_dim <= 64, any intermediate size): the tiny parity
// configuration (H 64, four heads of 16) runs here too, which the bf16 kernels' tilings exclude.
//
// This is the precision path, not the throughput path: one LDS-tiled GEMM with strided operands (64x64x16 tiles; 128x128x16
// with register prefetch for large problems, same bits) serves every Linear (forward NT, dX NN, dW TN) and -- with a fused
// log1p(relu) + max epilogue -- the tied decoder;
// the attention forward is MFMA-tiled (head_dim % 8 == 0; one wave per (token, head) otherwise and in the backward); the
// routed SPLADE backward and the attention backward use float atomics.
// Training runs under autocast(bf16) in the reference (ref:src/train/cli/train_v33_ddp.py:337) and on the bf16
// kernels here; the fp32 backward exists for gradient parity (tests/test_gpu_f32.py: all 137 gradients of the 149 M model
// against the reference's, and the reference's own two-rank train_epoch run replayed: tests/test_gpu_dist.py).
#include <cstdlib>
#include <vector>

#include "common.h"
#include "config.h"
#include "snx.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;

inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

// ------------------------------------------------------------------------------------------------------------
// GEMM: C[m, n] (op)= sum_k A(m, k) * B(n, k),  A(m, k) = A[m * a_row + k * a_k],  B(n, k) = B[n * b_row + k * b_k].
//   EPI 0: C = (R ? R : 0) + acc      (R: the residual stream, hf:331-332)
//   EPI 1: C += acc                   (weight gradients)
//   EPI 2: no C; SPLADE tail on the logits acc + bias[n] (ref:src/model/splade_modern.py:76-86)
// ------------------------------------------------------------------------------------------------------------
struct GemmArgs {
  const float* A; long a_row, a_k;
  const float* B; long b_row, b_k;
  float* C; long ldc;
  const float* R; long ldr;
  int M, N, K;
  int a_vec, b_vec;                  // 16-byte loads allowed along the contiguous dimension
};
struct TailArgs {                    // EPI 2: rows = tokens, columns = vocabulary
  const float* bias;                 // [V]
  const int64_t* mask;               // [T]
  const int32_t* seqid;              // [T] sequence of every token
  const int32_t* pos;                // [T] position inside its sequence
  unsigned long long* keys;          // [nseq, V] value bits << 32 | 0xFFFFFFFF - pos, zero-initialised
  uint32_t* twbits;                  // [T] bits of max_v value, zero-initialised
  int V;
};

constexpr int TP = 68;               // LDS row pitch (floats): 16-byte aligned rows, conflict-free MFMA operand reads

__device__ __forceinline__ void load_tile(const float* __restrict__ P, long s_row, long s_k, int vec, int row0, int nrows,
                                          int k0, int K, float (*S)[TP], int t) {
  if (vec && s_k == 1) {                                        // k contiguous: 4 k of one row per thread
    const int r = t >> 2, kk = (t & 3) * 4;
    if (row0 + r < nrows && k0 + kk + 3 < K) {
      const f32x4 v = *(const f32x4*)(P + (long)(row0 + r) * s_row + k0 + kk);
      S[kk][r] = v[0]; S[kk + 1][r] = v[1]; S[kk + 2][r] = v[2]; S[kk + 3][r] = v[3];
      return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
      S[kk + i][r] = (row0 + r < nrows && k0 + kk + i < K) ? P[(long)(row0 + r) * s_row + k0 + kk + i] : 0.f;
  } else if (vec && s_row == 1) {                               // rows contiguous: 4 rows of one k per thread
    const int kk = t >> 4, r = (t & 15) * 4;
    if (k0 + kk < K && row0 + r + 3 < nrows) {
      *(f32x4*)&S[kk][r] = *(const f32x4*)(P + (long)(k0 + kk) * s_k + row0 + r);
      return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
      S[kk][r + i] = (k0 + kk < K && row0 + r + i < nrows) ? P[(long)(k0 + kk) * s_k + row0 + r + i] : 0.f;
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = t + i * 256, r = idx & 63, kk = idx >> 6;
      S[kk][r] = (row0 + r < nrows && k0 + kk < K) ? P[(long)(row0 + r) * s_row + (long)(k0 + kk) * s_k] : 0.f;
    }
  }
}

// epilogue of a 64x64 tile whose wave (wm, wn) holds acc (shared by the two 64x64 kernels)
template <int EPI>
__device__ __forceinline__ void gemm_f32_epilogue(const GemmArgs& g, const TailArgs& ta, const f32x16& acc, int m0, int n0, int wm,
                                                  int wn, int lane) {
  // acc[r] = C[m0 + 32 wm + (r & 3) + 8 (r >> 2) + 4 (lane >> 5)][n0 + 32 wn + (lane & 31)]
  const int col = n0 + wn * 32 + (lane & 31);
  const int rbase = m0 + wm * 32 + 4 * (lane >> 5);
  if (EPI == 0 || EPI == 1) {
    if (col >= g.N) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = rbase + (r & 3) + 8 * (r >> 2);
      if (row >= g.M) continue;
      float* c = g.C + (long)row * g.ldc + col;
      if (EPI == 1) *c += acc[r];
      else *c = (g.R ? g.R[(long)row * g.ldr + col] : 0.f) + acc[r];
    }
  } else {
    // SPLADE tail: w = log1p(relu(logit)) * mask; per (sequence, v) the maximum over the sequence's tokens with the
    // FIRST position winning ties (torch.max on the CPU), per token the maximum over v.
    const bool cok = col < ta.V;
    const float bias = cok ? ta.bias[col] : 0.f;
    unsigned long long best = 0ull;
    int cur = -1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {                     // rows ascend with r: sequences ascend too
      const int row = rbase + (r & 3) + 8 * (r >> 2);
      const bool rok = row < g.M;
      float w = 0.f;
      if (rok && cok && ta.mask[row] != 0) w = log1pf(fmaxf(acc[r] + bias, 0.f));
      const int sq = rok ? ta.seqid[row] : -1;
      if (sq != cur) {
        if (cur >= 0 && cok) atomicMax(ta.keys + (long)cur * ta.V + col, best);
        cur = sq;
        best = 0ull;
      }
      if (rok) {
        const unsigned long long key = ((unsigned long long)__float_as_uint(w) << 32) | (0xFFFFFFFFu - (uint32_t)ta.pos[row]);
        best = key > best ? key : best;
      }
      // token maximum over this wave's 32 columns (the two halves of the wave hold different rows)
      float mx = cok ? w : 0.f;
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
      if (rok && (lane & 31) == 0) atomicMax(ta.twbits + row, __float_as_uint(mx));
    }
    if (cur >= 0 && cok) atomicMax(ta.keys + (long)cur * ta.V + col, best);
  }
}

template <int EPI>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g, TailArgs ta) {
  __shared__ __attribute__((aligned(16))) float As[16][TP];
  __shared__ __attribute__((aligned(16))) float Bs[16][TP];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int k0 = 0; k0 < g.K; k0 += 16) {
    load_tile(g.A, g.a_row, g.a_k, g.a_vec, m0, g.M, k0, g.K, As, t);
    load_tile(g.B, g.b_row, g.b_k, g.b_vec, n0, g.N, k0, g.K, Bs, t);
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const float a = As[2 * s + (lane >> 5)][wm * 32 + (lane & 31)];
      const float b = Bs[2 * s + (lane >> 5)][wn * 32 + (lane & 31)];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  gemm_f32_epilogue<EPI>(g, ta, acc, m0, n0, wm, wn, lane);
}

// The same contraction on 128x128x16 tiles for the large problems (the 149 M model's Linears and its decoder at
// inference): four waves of 64x64 (2 x 2 accumulators of 32x32), operands prefetched into registers one K-tile ahead
// and stored into the other half of a double-buffered LDS image, one barrier per K-tile.  Every output element is
// still the k-ascending chain of the 64x64 kernel above: identical bits.
constexpr int TP2 = 132;             // LDS row pitch of the 128-wide tiles (floats)

// this thread's share of a 128 x 16 operand tile, from global memory into 8 registers ...
__device__ __forceinline__ void fetch_tile128(const float* __restrict__ P, long s_row, long s_k, int vec, int row0, int nrows,
                                              int k0, int K, int t, float (&v)[8]) {
  if (vec && s_k == 1) {                                        // k contiguous: rows r, r + 64; 4 k each
    const int kk = (t & 3) * 4;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = (t >> 2) + 64 * i;
      if (row0 + r < nrows && k0 + kk + 3 < K) {
        const f32x4 x = *(const f32x4*)(P + (long)(row0 + r) * s_row + k0 + kk);
        v[4 * i] = x[0]; v[4 * i + 1] = x[1]; v[4 * i + 2] = x[2]; v[4 * i + 3] = x[3];
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u)
          v[4 * i + u] = (row0 + r < nrows && k0 + kk + u < K) ? P[(long)(row0 + r) * s_row + k0 + kk + u] : 0.f;
      }
    }
  } else if (vec && s_row == 1) {                               // rows contiguous: k rows kk, kk + 8; 4 rows each
    const int r = (t & 31) * 4;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int kk = (t >> 5) + 8 * i;
      if (k0 + kk < K && row0 + r + 3 < nrows) {
        const f32x4 x = *(const f32x4*)(P + (long)(k0 + kk) * s_k + row0 + r);
        v[4 * i] = x[0]; v[4 * i + 1] = x[1]; v[4 * i + 2] = x[2]; v[4 * i + 3] = x[3];
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u)
          v[4 * i + u] = (k0 + kk < K && row0 + r + u < nrows) ? P[(long)(k0 + kk) * s_k + row0 + r + u] : 0.f;
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int idx = t + i * 256, r = idx & 127, kk = idx >> 7;
      v[i] = (row0 + r < nrows && k0 + kk < K) ? P[(long)(row0 + r) * s_row + (long)(k0 + kk) * s_k] : 0.f;
    }
  }
}
// ... and from the registers into the [k][row] image
__device__ __forceinline__ void put_tile128(long s_row, long s_k, int vec, int t, const float (&v)[8], float (*S)[TP2]) {
  if (vec && s_k == 1) {
    const int kk = (t & 3) * 4;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = (t >> 2) + 64 * i;
#pragma unroll
      for (int u = 0; u < 4; ++u) S[kk + u][r] = v[4 * i + u];
    }
  } else if (vec && s_row == 1) {
    const int r = (t & 31) * 4;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int kk = (t >> 5) + 8 * i;
      *(f32x4*)&S[kk][r] = (f32x4){v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]};
    }
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int idx = t + i * 256;
      S[idx >> 7][idx & 127] = v[i];
    }
  }
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_f32_128_kernel(GemmArgs g, TailArgs ta) {
  __shared__ __attribute__((aligned(16))) float As[2][16][TP2];
  __shared__ __attribute__((aligned(16))) float Bs[2][16][TP2];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1, c = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.y * 128, n0 = blockIdx.x * 128;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float va[8], vb[8];
  fetch_tile128(g.A, g.a_row, g.a_k, g.a_vec, m0, g.M, 0, g.K, t, va);
  fetch_tile128(g.B, g.b_row, g.b_k, g.b_vec, n0, g.N, 0, g.K, t, vb);
  put_tile128(g.a_row, g.a_k, g.a_vec, t, va, As[0]);
  put_tile128(g.b_row, g.b_k, g.b_vec, t, vb, Bs[0]);
  __syncthreads();
  const int nk = (g.K + 15) / 16;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) {
      fetch_tile128(g.A, g.a_row, g.a_k, g.a_vec, m0, g.M, (kt + 1) * 16, g.K, t, va);
      fetch_tile128(g.B, g.b_row, g.b_k, g.b_vec, n0, g.N, (kt + 1) * 16, g.K, t, vb);
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const float a0 = As[cur][2 * s + h][wm * 64 + c], a1 = As[cur][2 * s + h][wm * 64 + 32 + c];
      const float b0 = Bs[cur][2 * s + h][wn * 64 + c], b1 = Bs[cur][2 * s + h][wn * 64 + 32 + c];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (kt + 1 < nk) {
      put_tile128(g.a_row, g.a_k, g.a_vec, t, va, As[cur ^ 1]);
      put_tile128(g.b_row, g.b_k, g.b_vec, t, vb, Bs[cur ^ 1]);
    }
    __syncthreads();
  }
  // acc[i][j][r] = C[m0 + 64 wm + 32 i + (r & 3) + 8 (r >> 2) + 4 h][n0 + 64 wn + 32 j + c]
  if (EPI == 0 || EPI == 1) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + c;
        if (col >= g.N) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (row >= g.M) continue;
          float* cp = g.C + (long)row * g.ldc + col;
          if (EPI == 1) *cp += acc[i][j][r];
          else *cp = (g.R ? g.R[(long)row * g.ldr + col] : 0.f) + acc[i][j][r];
        }
      }
  } else {
    // SPLADE tail, as in the 64x64 kernel: rows ascend with (i, r), so sequences ascend too
    const int col0 = n0 + wn * 64 + c, col1 = col0 + 32;
    const bool ok0 = col0 < ta.V, ok1 = col1 < ta.V;
    const float bias0 = ok0 ? ta.bias[col0] : 0.f, bias1 = ok1 ? ta.bias[col1] : 0.f;
    unsigned long long best0 = 0ull, best1 = 0ull;
    int cur = -1;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const bool rok = row < g.M;
        const bool live = rok && ta.mask[row] != 0;
        const float w0 = live && ok0 ? log1pf(fmaxf(acc[i][0][r] + bias0, 0.f)) : 0.f;
        const float w1 = live && ok1 ? log1pf(fmaxf(acc[i][1][r] + bias1, 0.f)) : 0.f;
        const int sq = rok ? ta.seqid[row] : -1;
        if (sq != cur) {
          if (cur >= 0) {
            if (ok0) atomicMax(ta.keys + (long)cur * ta.V + col0, best0);
            if (ok1) atomicMax(ta.keys + (long)cur * ta.V + col1, best1);
          }
          cur = sq;
          best0 = best1 = 0ull;
        }
        if (rok) {
          const unsigned long long tag = 0xFFFFFFFFu - (uint32_t)ta.pos[row];
          const unsigned long long k0 = ((unsigned long long)__float_as_uint(w0) << 32) | tag;
          const unsigned long long k1 = ((unsigned long long)__float_as_uint(w1) << 32) | tag;
          best0 = k0 > best0 ? k0 : best0;
          best1 = k1 > best1 ? k1 : best1;
        }
        float mx = fmaxf(ok0 ? w0 : 0.f, ok1 ? w1 : 0.f);
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        if (rok && c == 0) atomicMax(ta.twbits + row, __float_as_uint(mx));
      }
    if (cur >= 0) {
      if (ok0) atomicMax(ta.keys + (long)cur * ta.V + col0, best0);
      if (ok1) atomicMax(ta.keys + (long)cur * ta.V + col1, best1);
    }
  }
}

// The 64x64 tile with a 64-deep K-step and the next step's operands prefetched into registers: for SMALL problems (a
// query batch of 64 tokens is one row of 36 tiles), where the kernel above is a chain of K / 16 exposed memory round
// trips on a few dozen workgroups -- 60 us per Linear, 5.3 of the 6.6 ms of a single-query forward.  Same k-ascending
// chain per output element: identical bits.
__device__ __forceinline__ void fetch_tile64x64(const float* __restrict__ P, long s_row, long s_k, int vec, int row0, int nrows,
                                                int k0, int K, int t, float (&v)[16]) {
  if (vec && s_k == 1) {                                        // k contiguous: one row, 16 consecutive k per thread
    const int r = t >> 2, kk = (t & 3) * 16;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (row0 + r < nrows && k0 + kk + 4 * q + 3 < K) {
        const f32x4 x = *(const f32x4*)(P + (long)(row0 + r) * s_row + k0 + kk + 4 * q);
        v[4 * q] = x[0]; v[4 * q + 1] = x[1]; v[4 * q + 2] = x[2]; v[4 * q + 3] = x[3];
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u)
          v[4 * q + u] = (row0 + r < nrows && k0 + kk + 4 * q + u < K) ? P[(long)(row0 + r) * s_row + k0 + kk + 4 * q + u] : 0.f;
      }
    }
  } else if (vec && s_row == 1) {                               // rows contiguous: k rows kk + 16 q, 4 rows each
    const int r = (t & 15) * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int kk = (t >> 4) + 16 * q;
      if (k0 + kk < K && row0 + r + 3 < nrows) {
        const f32x4 x = *(const f32x4*)(P + (long)(k0 + kk) * s_k + row0 + r);
        v[4 * q] = x[0]; v[4 * q + 1] = x[1]; v[4 * q + 2] = x[2]; v[4 * q + 3] = x[3];
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u)
          v[4 * q + u] = (k0 + kk < K && row0 + r + u < nrows) ? P[(long)(k0 + kk) * s_k + row0 + r + u] : 0.f;
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int idx = t + i * 256, r = idx & 63, kk = idx >> 6;
      v[i] = (row0 + r < nrows && k0 + kk < K) ? P[(long)(row0 + r) * s_row + (long)(k0 + kk) * s_k] : 0.f;
    }
  }
}
__device__ __forceinline__ void put_tile64x64(long s_row, long s_k, int vec, int t, const float (&v)[16], float (*S)[TP]) {
  if (vec && s_k == 1) {
    const int r = t >> 2, kk = (t & 3) * 16;
#pragma unroll
    for (int i = 0; i < 16; ++i) S[kk + i][r] = v[i];
  } else if (vec && s_row == 1) {
    const int r = (t & 15) * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int kk = (t >> 4) + 16 * q;
      *(f32x4*)&S[kk][r] = (f32x4){v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
    }
  } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int idx = t + i * 256;
      S[idx >> 6][idx & 63] = v[i];
    }
  }
}

template <int EPI>
__global__ __launch_bounds__(256) void gemm_f32_deepk_kernel(GemmArgs g, TailArgs ta) {
  __shared__ __attribute__((aligned(16))) float As[64][TP];
  __shared__ __attribute__((aligned(16))) float Bs[64][TP];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  float va[16], vb[16];
  fetch_tile64x64(g.A, g.a_row, g.a_k, g.a_vec, m0, g.M, 0, g.K, t, va);
  fetch_tile64x64(g.B, g.b_row, g.b_k, g.b_vec, n0, g.N, 0, g.K, t, vb);
  for (int k0 = 0; k0 < g.K; k0 += 64) {
    put_tile64x64(g.a_row, g.a_k, g.a_vec, t, va, As);
    put_tile64x64(g.b_row, g.b_k, g.b_vec, t, vb, Bs);
    __syncthreads();
    if (k0 + 64 < g.K) {                                // the next step's operands fly under this step's 32 MFMAs
      fetch_tile64x64(g.A, g.a_row, g.a_k, g.a_vec, m0, g.M, k0 + 64, g.K, t, va);
      fetch_tile64x64(g.B, g.b_row, g.b_k, g.b_vec, n0, g.N, k0 + 64, g.K, t, vb);
    }
#pragma unroll
    for (int s = 0; s < 32; ++s) {
      const float a = As[2 * s + (lane >> 5)][wm * 32 + (lane & 31)];
      const float b = Bs[2 * s + (lane >> 5)][wn * 32 + (lane & 31)];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  gemm_f32_epilogue<EPI>(g, ta, acc, m0, n0, wm, wn, lane);
}

template <int EPI>
int launch_gemm(GemmArgs g, const TailArgs& ta, hipStream_t st) {
  if (g.M <= 0 || g.N <= 0 || g.K <= 0) return SNX_E_SHAPE;
  auto vec_ok = [](const float* p, long s_row, long s_k) {
    if (((uintptr_t)p & 15) != 0) return 0;
    if (s_k == 1) return (s_row % 4) == 0 ? 1 : 0;
    if (s_row == 1) return (s_k % 4) == 0 ? 1 : 0;
    return 0;
  };
  g.a_vec = vec_ok(g.A, g.a_row, g.a_k);
  g.b_vec = vec_ok(g.B, g.b_row, g.b_k);
  const bool small_only = g_snx_cfg.f32_gemm64 != 0;             // A/B and tests: the 64x64 kernel for every shape
  if (!small_only && (long)g.M * g.N >= 128L * 128 * 256 && g.K >= 32)
    hipLaunchKernelGGL(gemm_f32_128_kernel<EPI>, dim3(cdiv(g.N, 128), cdiv(g.M, 128)), dim3(256), 0, st, g, ta);
  else if (!small_only && g.K >= 128 && (long)cdiv(g.M, 64) * cdiv(g.N, 64) <= 512)   // few tiles: latency-bound
    hipLaunchKernelGGL(gemm_f32_deepk_kernel<EPI>, dim3(cdiv(g.N, 64), cdiv(g.M, 64)), dim3(256), 0, st, g, ta);
  else
    hipLaunchKernelGGL(gemm_f32_kernel<EPI>, dim3(cdiv(g.N, 64), cdiv(g.M, 64)), dim3(256), 0, st, g, ta);
  SNX_CHECK_LAUNCH();
  return SNX_OK;
}

// y[M, N] = (R) + x[M, K] W[N, K]^T   (nn.Linear, bias=False)
int linear_fwd(const float* x, const float* W, const float* R, float* y, int M, int N, int K, hipStream_t st) {
  GemmArgs g{x, K, 1, W, K, 1, y, N, R, N, M, N, K, 0, 0};
  return launch_gemm<0>(g, TailArgs{}, st);
}
// dx[M, K] = dy[M, N] W[N, K]
int linear_dx(const float* dy, const float* W, float* dx, int M, int N, int K, hipStream_t st) {
  GemmArgs g{dy, N, 1, W, 1, K, dx, K, nullptr, 0, M, K, N, 0, 0};
  return launch_gemm<0>(g, TailArgs{}, st);
}
// dW[N, K] += dy[M, N]^T x[M, K]
int linear_dw(const float* dy, const float* x, float* dW, int M, int N, int K, hipStream_t st) {
  GemmArgs g{dy, 1, N, x, 1, K, dW, K, nullptr, 0, N, K, M, 0, 0};
  return launch_gemm<1>(g, TailArgs{}, st);
}

// ------------------------------------------------------------------------------------------------------------
// token-wise kernels: one wave per row, any width
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float gelu_exact(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_exact_grad(float x) {
  return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.39894228040143268f * expf(-0.5f * x * x);
}

// SRC 0: x = in[t];  SRC 1: x = E[ids[t]] (embedding gather);  SRC 2: x = gelu(in[t])   ->  out[t] = LN(x) * w
template <int SRC>
__global__ void ln_f32_kernel(const float* __restrict__ in, const int64_t* __restrict__ ids, const float* __restrict__ w,
                              float* __restrict__ out, int T, int H, float eps) {
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= T) return;
  const float* x = SRC == 1 ? in + ids[t] * (long)H : in + (long)t * H;
  float s = 0.f;
  for (int c = lane; c < H; c += 64) s += SRC == 2 ? gelu_exact(x[c]) : x[c];
  const float mean = wave_sum(s) / (float)H;
  float v = 0.f;
  for (int c = lane; c < H; c += 64) {
    const float d = (SRC == 2 ? gelu_exact(x[c]) : x[c]) - mean;
    v += d * d;
  }
  const float rstd = rsqrtf(wave_sum(v) / (float)H + eps);
  for (int c = lane; c < H; c += 64) out[(long)t * H + c] = ((SRC == 2 ? gelu_exact(x[c]) : x[c]) - mean) * rstd * w[c];
}

// backward of the above.  dx = rstd (g - mean(g) - xhat mean(g xhat)), g = dy w;  dw += sum_t dy xhat (atomics).
//   SRC 0: dst[t] (+)= dx   (ACC: add into the residual-stream gradient / overwrite)
//   SRC 1: gradE[ids[t]] += dx (atomics; the padding row gets none)
//   SRC 2: dst[t] = dx * gelu'(in[t])
template <int SRC, bool ACC>
__global__ void ln_f32_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ in, const int64_t* __restrict__ ids,
                                  const float* __restrict__ w, float* __restrict__ dst, float* __restrict__ dw, int T, int H,
                                  float eps, int pad_id) {
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= T) return;
  const long id = SRC == 1 ? ids[t] : 0;
  const float* x = SRC == 1 ? in + id * (long)H : in + (long)t * H;
  const float* g = dy + (long)t * H;
  float s = 0.f;
  for (int c = lane; c < H; c += 64) s += SRC == 2 ? gelu_exact(x[c]) : x[c];
  const float mean = wave_sum(s) / (float)H;
  float v = 0.f;
  for (int c = lane; c < H; c += 64) {
    const float d = (SRC == 2 ? gelu_exact(x[c]) : x[c]) - mean;
    v += d * d;
  }
  const float rstd = rsqrtf(wave_sum(v) / (float)H + eps);
  float s1 = 0.f, s2 = 0.f;
  for (int c = lane; c < H; c += 64) {
    const float xh = ((SRC == 2 ? gelu_exact(x[c]) : x[c]) - mean) * rstd;
    const float gw = g[c] * w[c];
    s1 += gw;
    s2 += gw * xh;
    atomicAdd(dw + c, g[c] * xh);
  }
  s1 = wave_sum(s1) / (float)H;
  s2 = wave_sum(s2) / (float)H;
  for (int c = lane; c < H; c += 64) {
    const float xh = ((SRC == 2 ? gelu_exact(x[c]) : x[c]) - mean) * rstd;
    const float dx = rstd * (g[c] * w[c] - s1 - xh * s2);
    if (SRC == 1) {
      if (id != pad_id) atomicAdd(dst + id * (long)H + c, dx);
    } else if (SRC == 2) {
      dst[(long)t * H + c] = dx * gelu_exact_grad(x[c]);
    } else if (ACC) {
      dst[(long)t * H + c] += dx;
    } else {
      dst[(long)t * H + c] = dx;
    }
  }
}

// apply_rotary_pos_emb (hf:196-219) in place on the q and k thirds of qkv [T, 3, heads, hd]; tab [max_pos][hd/2] (cos, sin)
__global__ void rope_f32_kernel(float* __restrict__ qkv, const f32x2* __restrict__ tab, const int32_t* __restrict__ pos,
                                long n, int heads, int hd, int inverse) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;   // (t, which, head, d < hd / 2)
  if (i >= n) return;
  const int half = hd >> 1;
  const int d = (int)(i % half);
  long rest = i / half;
  const int head = (int)(rest % heads); rest /= heads;
  const int which = (int)(rest & 1);
  const long t = rest >> 1;
  float* base = qkv + ((t * 3 + which) * heads + head) * (long)hd;
  const f32x2 cs = tab[(long)pos[t] * half + d];
  const float c = cs[0], s = inverse ? -cs[1] : cs[1];
  const float x1 = base[d], x2 = base[d + half];
  base[d] = mul_rn(x1, c) - mul_rn(x2, s);            // x cos + rotate_half(x) sin, products rounded separately
  base[d + half] = mul_rn(x2, c) + mul_rn(x1, s);
}

// GeGLU (hf:90-91): u = [a | g] along the last dimension (natural order), y = gelu(a) * g
__global__ void geglu_f32_kernel(const float* __restrict__ u, float* __restrict__ y, long n, int I) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long t = i / I;
  const int c = (int)(i - t * I);
  y[i] = gelu_exact(u[t * 2 * I + c]) * u[t * 2 * I + I + c];
}
__global__ void geglu_f32_bwd_kernel(const float* __restrict__ u, const float* __restrict__ dy, float* __restrict__ du, long n,
                                     int I) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long t = i / I;
  const int c = (int)(i - t * I);
  const float a = u[t * 2 * I + c], g = u[t * 2 * I + I + c], d = dy[i];
  du[t * 2 * I + c] = d * g * gelu_exact_grad(a);
  du[t * 2 * I + I + c] = d * gelu_exact(a);
}

__global__ void seqid_kernel(const int32_t* __restrict__ cu, int32_t* __restrict__ seqid, int T, int nseq) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  int lo = 0, hi = nseq - 1;                           // last s with cu[s] <= t
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (cu[mid] <= t) lo = mid;
    else hi = mid - 1;
  }
  seqid[t] = lo;
}

__global__ void tail_finalize_kernel(const unsigned long long* __restrict__ keys, const uint32_t* __restrict__ twbits,
                                     float* __restrict__ sparse, float* __restrict__ tw, long nv, int T) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nv) sparse[i] = __uint_as_float((uint32_t)(keys[i] >> 32));
  if (i < T) tw[i] = __uint_as_float(twbits[i]);
}

// ------------------------------------------------------------------------------------------------------------
// attention (hf:286-297 -> SDPA; masks masking_utils.py:141-150): one wave per (token, head).  Scores with lanes over
// keys (64 per chunk, online softmax), P V with lanes over the head dimension (probabilities through LDS).
// window < 0: global layer.  A query that sees no key (padding rows only) outputs zeros; nothing downstream of such a
// row reaches a result (its SPLADE weights are masked, it is visible to no other token).
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool visible(int qp, int kp, int window, const int64_t* __restrict__ mask, int s0) {
  if (mask[s0 + kp] == 0) return false;
  const int d = qp - kp;
  return window < 0 || (d <= window && -d <= window);
}

__global__ __launch_bounds__(256) void attn_f32_fwd_kernel(const float* __restrict__ qkv, const int32_t* __restrict__ cu,
                                                           const int32_t* __restrict__ seqid, const int64_t* __restrict__ mask,
                                                           float* __restrict__ out, float* __restrict__ lse, int T, int heads,
                                                           int hd, int window, float scale) {
  __shared__ float sP[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long wid = (long)blockIdx.x * 4 + wave;
  if (wid >= (long)T * heads) return;
  const int t = (int)(wid / heads), head = (int)(wid % heads);
  const int sq = seqid[t], s0 = cu[sq], slen = cu[sq + 1] - s0, qp = t - s0;
  const long rs = 3L * heads * hd;
  const float* q = qkv + (long)t * rs + head * hd;
  const float* kb = qkv + (long)s0 * rs + (long)heads * hd + head * hd;
  const float* vb = kb + (long)heads * hd;
  int lo = 0, hi = slen - 1;
  if (window >= 0) { lo = max(0, qp - window); hi = min(slen - 1, qp + window); }
  float m = -INFINITY, l = 0.f, o = 0.f;               // o: this lane's output dimension (lane < hd)
  for (int c0 = lo; c0 <= hi; c0 += 64) {
    const int kp = c0 + lane;
    float s = -INFINITY;
    if (kp <= hi && visible(qp, kp, window, mask, s0)) {
      const float* k = kb + (long)kp * rs;
      float a = 0.f;
      for (int d = 0; d < hd; ++d) a = fmaf(q[d], k[d], a);
      s = a * scale;
    }
    const float mx = wave_max(s);
    const float mn = fmaxf(m, mx);
    if (mn == -INFINITY) continue;                     // nothing visible so far
    const float p = s == -INFINITY ? 0.f : expf(s - mn);
    const float alpha = m == -INFINITY ? 0.f : expf(m - mn);
    l = l * alpha + wave_sum(p);
    m = mn;
    sP[wave][lane] = p;
    __builtin_amdgcn_wave_barrier();
    o *= alpha;
    if (lane < hd) {
      const int n = min(64, hi - c0 + 1);
      for (int j = 0; j < n; ++j) o = fmaf(sP[wave][j], vb[(long)(c0 + j) * rs + lane], o);
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (lane < hd) out[(long)t * heads * hd + head * hd + lane] = l > 0.f ? o / l : 0.f;
  if (lane == 0) lse[(long)head * T + t] = l > 0.f ? m + logf(l) : 0.f;
}

// Tiled form of the forward for head_dim % 8 == 0 (both shipped geometries: 64 and 16) -- what the inference encoder
// runs.  One workgroup per (sequence, head, 128-query block), four waves of 32 queries, key tiles of 64 rows of K and V
// in LDS, both contractions on v_mfma_f32_32x32x2_f32 in TRANSPOSED form so that a lane owns one query:
//   S^T[key, q] = sum_d K[key, d] Q[q, d]      acc element r of lane (c, h): key 32 b + (r & 3) + 8 (r >> 2) + 4 h, query c
//   O^T[d, q]   = sum_key V[key, d] P[key, q]  the probabilities are consumed from the registers S^T left them in (the
//                                              k pair of MFMA step (b, r) is keys {.., .. + 4} of the two wave halves)
// Running maximum / sum / rescale are per lane (two lane-xor-32 exchanges per tile); the contraction over d pairs
// dimension 8 m + i with 8 m + 4 + i (float4 operand reads).  The wave-per-(token, head) kernel above took 8.3 ms per
// layer at 64 x 256 tokens (70 % of the fp32 forward); this one is MFMA-shaped work of 13 GFLOP.
constexpr int AQ = 128, AKT = 64, APITCH = 68;

__global__ __launch_bounds__(256) void attn_f32_fwd_tiled_kernel(const float* __restrict__ qkv, const int32_t* __restrict__ cu,
                                                                 const int64_t* __restrict__ mask, float* __restrict__ out,
                                                                 float* __restrict__ lse, int T, int heads, int hd, int window,
                                                                 float scale) {
  __shared__ __attribute__((aligned(16))) float sm[2 * AKT * APITCH];   // K tile | V tile; the output block at the end
  float (*sK)[APITCH] = (float (*)[APITCH])sm;
  float (*sV)[APITCH] = (float (*)[APITCH])(sm + AKT * APITCH);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, c = lane & 31, h = lane >> 5;
  const int sq = blockIdx.x, head = blockIdx.y;
  const int s0 = cu[sq], slen = cu[sq + 1] - s0;
  const long rs = 3L * heads * hd, H = (long)heads * hd;
  const int mg = hd >> 3, db_n = hd > 32 ? 2 : 1, q4 = hd >> 2;
  const float* kbase = qkv + (long)s0 * rs + H + head * hd;
  const float* vbase = kbase + H;
  for (int q0 = blockIdx.z * AQ; q0 < slen; q0 += gridDim.z * AQ) {
    const int qp = q0 + wave * 32 + c;
    const bool qok = qp < slen;
    f32x4 qr[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      qr[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (m < mg && qok) qr[m] = *(const f32x4*)(qkv + (long)(s0 + qp) * rs + head * hd + 8 * m + 4 * h);
    }
    int klo = 0, khi = slen - 1;
    if (window >= 0) { klo = max(0, q0 - window); khi = min(slen - 1, q0 + AQ - 1 + window); }
    const int wq_lo = q0 + wave * 32, wq_hi = wq_lo + 31;          // this wave's queries
    float mrun = -INFINITY, lrun = 0.f;
    f32x16 o[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { o[0][i] = 0.f; o[1][i] = 0.f; }
    for (int c0 = klo; c0 <= khi; c0 += AKT) {
      __syncthreads();                                 // the previous tile has been consumed
      for (int idx = t; idx < AKT * q4; idx += 256) {
        const int key = idx / q4, j = idx - key * q4;
        f32x4 kv = (f32x4){0.f, 0.f, 0.f, 0.f}, vv = kv;
        if (c0 + key <= khi) {
          kv = *(const f32x4*)(kbase + (long)(c0 + key) * rs + 4 * j);
          vv = *(const f32x4*)(vbase + (long)(c0 + key) * rs + 4 * j);
        }
        *(f32x4*)&sK[key][4 * j] = kv;
        *(f32x4*)&sV[key][4 * j] = vv;
      }
      __syncthreads();
      if (window >= 0 && (c0 > wq_hi + window || c0 + AKT - 1 < wq_lo - window)) continue;   // wave-uniform
      const int kb = c0 + lane;
      const unsigned long long bits = __ballot(kb <= khi && mask[s0 + kb] != 0);
      f32x16 sa[2];
#pragma unroll
      for (int i = 0; i < 16; ++i) { sa[0][i] = 0.f; sa[1][i] = 0.f; }
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        if (m < mg) {
          const f32x4 k0 = *(const f32x4*)&sK[c][8 * m + 4 * h], k1 = *(const f32x4*)&sK[32 + c][8 * m + 4 * h];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            sa[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(k0[i], qr[m][i], sa[0], 0, 0, 0);
            sa[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(k1[i], qr[m][i], sa[1], 0, 0, 0);
          }
        }
      }
      float mx = -INFINITY;
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = 32 * b + (r & 3) + 8 * (r >> 2) + 4 * h;
          const int dq = qp - (c0 + key);
          const bool vis = ((bits >> key) & 1ull) && (window < 0 || (dq <= window && -dq <= window));
          const float sv = vis ? sa[b][r] * scale : -INFINITY;
          sa[b][r] = sv;
          mx = fmaxf(mx, sv);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float mn = fmaxf(mrun, mx);
      const float alpha = mrun == -INFINITY ? 0.f : expf(mrun - mn);
      float psum = 0.f;
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pv = sa[b][r] == -INFINITY ? 0.f : expf(sa[b][r] - mn);
          sa[b][r] = pv;
          psum += pv;
        }
      psum += __shfl_xor(psum, 32, 64);
      lrun = lrun * alpha + psum;
      mrun = mn;
#pragma unroll
      for (int i = 0; i < 16; ++i) { o[0][i] *= alpha; o[1][i] *= alpha; }
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = 32 * b + (r & 3) + 8 * (r >> 2) + 4 * h;
          const float v0 = c < hd ? sV[key][c] : 0.f;
          o[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, sa[b][r], o[0], 0, 0, 0);
          if (db_n > 1) {
            const float v1 = 32 + c < hd ? sV[key][32 + c] : 0.f;
            o[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, sa[b][r], o[1], 0, 0, 0);
          }
        }
    }
    // o[db][r] = O[dimension 32 db + (r & 3) + 8 (r >> 2) + 4 h][query c] (unnormalised): through LDS for row-major stores
    __syncthreads();
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int d = 32 * db + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (db < db_n) sm[(wave * 32 + c) * APITCH + d] = lrun > 0.f ? o[db][r] / lrun : 0.f;
      }
    __syncthreads();
    for (int idx = t; idx < AQ * q4; idx += 256) {
      const int row = idx / q4, j = idx - row * q4;
      if (q0 + row < slen)
        *(f32x4*)(out + (long)(s0 + q0 + row) * H + head * hd + 4 * j) = *(const f32x4*)&sm[row * APITCH + 4 * j];
    }
    if (h == 0 && qok) lse[(long)head * T + s0 + qp] = lrun > 0.f ? mrun + logf(lrun) : 0.f;
  }
}

// dq (written), dk / dv (float atomics into zero-initialised dqkv): p = exp(s - lse), ds = p (dp - delta) scale
__global__ __launch_bounds__(256) void attn_f32_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ out,
                                                           const float* __restrict__ dout, const float* __restrict__ lse,
                                                           const int32_t* __restrict__ cu, const int32_t* __restrict__ seqid,
                                                           const int64_t* __restrict__ mask, float* __restrict__ dqkv, int T,
                                                           int heads, int hd, int window, float scale) {
  __shared__ float sP[4][64], sD[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long wid = (long)blockIdx.x * 4 + wave;
  if (wid >= (long)T * heads) return;
  const int t = (int)(wid / heads), head = (int)(wid % heads);
  const int sq = seqid[t], s0 = cu[sq], slen = cu[sq + 1] - s0, qp = t - s0;
  const long rs = 3L * heads * hd;
  const long H = (long)heads * hd;
  const float* q = qkv + (long)t * rs + head * hd;
  const float* kb = qkv + (long)s0 * rs + H + head * hd;
  const float* vb = kb + H;
  const float* dO = dout + (long)t * H + head * hd;
  const float* O = out + (long)t * H + head * hd;
  float* dq = dqkv + (long)t * rs + head * hd;
  float* dkb = dqkv + (long)s0 * rs + H + head * hd;
  float* dvb = dkb + H;
  float delta = 0.f;
  for (int d = 0; d < hd; ++d) delta = fmaf(dO[d], O[d], delta);
  const float L = lse[(long)head * T + t];
  int lo = 0, hi = slen - 1;
  if (window >= 0) { lo = max(0, qp - window); hi = min(slen - 1, qp + window); }
  float dqa = 0.f;                                     // this lane's dimension of dq
  const float qd = lane < hd ? q[lane] : 0.f, dod = lane < hd ? dO[lane] : 0.f;
  for (int c0 = lo; c0 <= hi; c0 += 64) {
    const int kp = c0 + lane;
    float p = 0.f, ds = 0.f;
    if (kp <= hi && visible(qp, kp, window, mask, s0)) {
      const float* k = kb + (long)kp * rs;
      const float* v = vb + (long)kp * rs;
      float a = 0.f, dp = 0.f;
      for (int d = 0; d < hd; ++d) {
        a = fmaf(q[d], k[d], a);
        dp = fmaf(dO[d], v[d], dp);
      }
      p = expf(a * scale - L);
      ds = p * (dp - delta) * scale;
    }
    sP[wave][lane] = p;
    sD[wave][lane] = ds;
    __builtin_amdgcn_wave_barrier();
    if (lane < hd) {
      const int n = min(64, hi - c0 + 1);
      for (int j = 0; j < n; ++j) {
        const float pj = sP[wave][j], dj = sD[wave][j];
        if (pj == 0.f && dj == 0.f) continue;          // wave-uniform: masked key
        dqa = fmaf(dj, kb[(long)(c0 + j) * rs + lane], dqa);
        atomicAdd(dkb + (long)(c0 + j) * rs + lane, dj * qd);
        atomicAdd(dvb + (long)(c0 + j) * rs + lane, pj * dod);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (lane < hd) atomicAdd(dq + lane, dqa);            // (the q third is zero-initialised like the others)
}

// routed backward of the SPLADE tail + tied decoder: for every (sequence b, v) with value y > 0:
//   c = g[b, v] / (1 + relu(x)) = g exp(-y);  token = cu[b] + argmax position;
//   dHd[token] += c E[v];  gradE[v] += c Hd[token];  gradb[v] += c.        One wave per (b, v).
__global__ __launch_bounds__(256) void tail_f32_bwd_kernel(const float* __restrict__ g, const unsigned long long* __restrict__ keys,
                                                           const float* __restrict__ Hd, const float* __restrict__ E,
                                                           const int32_t* __restrict__ cu, float* __restrict__ dHd,
                                                           float* __restrict__ gradE, float* __restrict__ gradb, long nv, int V,
                                                           int H) {
  const int lane = threadIdx.x & 63;
  const long i = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= nv) return;
  const unsigned long long key = keys[i];
  const float y = __uint_as_float((uint32_t)(key >> 32));
  const float gv = g[i];
  if (!(y > 0.f) || gv == 0.f) return;
  const int b = (int)(i / V), v = (int)(i - (long)b * V);
  const int tok = cu[b] + (int)(0xFFFFFFFFu - (uint32_t)key);
  const float c = gv * expf(-y);
  for (int d = lane; d < H; d += 64) {
    atomicAdd(dHd + (long)tok * H + d, c * E[(long)v * H + d]);
    atomicAdd(gradE + (long)v * H + d, c * Hd[(long)tok * H + d]);
  }
  if (lane == 0) atomicAdd(gradb + v, c);
}

__global__ void add_f32_kernel(float* __restrict__ dst, const float* __restrict__ src, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] += src[i];
}

// ------------------------------------------------------------------------------------------------------------
// arena plans
// ------------------------------------------------------------------------------------------------------------
struct PIdx {                        // canonical parameter order (model.hip)
  int L;
  int tok_emb() const { return 0; }
  int emb_norm() const { return 1; }
  int base(int l) const { return l == 0 ? 2 : 7 + 6 * (l - 1); }
  int attn_norm(int l) const { return base(l); }
  int wqkv(int l) const { return base(l) + (l == 0 ? 0 : 1); }
  int wo(int l) const { return wqkv(l) + 1; }
  int mlp_norm(int l) const { return wqkv(l) + 2; }
  int wi(int l) const { return wqkv(l) + 3; }
  int wo_mlp(int l) const { return wqkv(l) + 4; }
  int tail() const { return 7 + 6 * (L - 1); }
  int final_norm() const { return tail(); }
  int head_dense() const { return tail() + 1; }
  int head_norm() const { return tail() + 2; }
  int dec_bias() const { return tail() + 3; }
};

struct Plan {
  size_t h[129], x_attn[64], qkv[64], attn[64], lse[64], x_mlp[64], u[64], y[64];
  size_t xf, dd, hd, keys, twbits, seqid, total;
};

bool f32_desc_ok(const snx_model_desc* d) {
  return d && d->vocab > 0 && d->hidden > 0 && d->inter > 0 && d->layers >= 1 && d->layers <= 64 && d->heads > 0 &&
         d->head_dim >= 2 && d->head_dim <= 64 && (d->head_dim % 2) == 0 && d->heads * d->head_dim == d->hidden &&
         d->global_every >= 1 && d->window >= 0;
}

void plan(const snx_model_desc* d, size_t T, size_t nseq, bool save, Plan& s) {
  const size_t H = d->hidden, I = d->inter, V = d->vocab, L = d->layers;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = al(off + bytes); return o; };
  if (save) {
    for (size_t i = 0; i <= 2 * L; ++i) s.h[i] = take(T * H * 4);
    for (size_t l = 0; l < L; ++l) {
      s.x_attn[l] = l == 0 ? s.h[0] : take(T * H * 4);          // layer 0 has no attn_norm (hf:312: Identity)
      s.qkv[l] = take(T * 3 * H * 4); s.attn[l] = take(T * H * 4); s.lse[l] = take((size_t)d->heads * T * 4);
      s.x_mlp[l] = take(T * H * 4); s.u[l] = take(T * 2 * I * 4); s.y[l] = take(T * I * 4);
    }
  } else {
    const size_t h0 = take(T * H * 4), h1 = take(T * H * 4), h2 = take(T * H * 4);
    for (size_t i = 0; i <= 2 * L; ++i) s.h[i] = (i % 3 == 0) ? h0 : (i % 3 == 1) ? h1 : h2;
    const size_t xa = take(T * H * 4), qkv = take(T * 3 * H * 4), at = take(T * H * 4), ls = take((size_t)d->heads * T * 4);
    const size_t xm = take(T * H * 4), u = take(T * 2 * I * 4), y = take(T * I * 4);
    for (size_t l = 0; l < L; ++l) {
      s.x_attn[l] = l == 0 ? s.h[0] : xa; s.qkv[l] = qkv; s.attn[l] = at; s.lse[l] = ls; s.x_mlp[l] = xm; s.u[l] = u; s.y[l] = y;
    }
  }
  s.xf = take(T * H * 4); s.dd = take(T * H * 4); s.hd = take(T * H * 4);
  s.keys = take(nseq * V * 8);
  s.twbits = take(T * 4);
  s.seqid = take(T * 4);
  s.total = off;
}

struct BPlan { size_t dh, a, b, c, total; };
void bplan(const snx_model_desc* d, size_t T, BPlan& p) {
  const size_t H = d->hidden, I = d->inter;
  const size_t wide = 3 * H > 2 * I ? 3 * H : 2 * I;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = al(off + bytes); return o; };
  p.dh = take(T * H * 4);
  p.a = take(T * wide * 4);
  p.b = take(T * wide * 4);
  p.c = take(T * H * 4);
  p.total = off;
}

#define RC(call)            \
  do {                      \
    int rc__ = (call);      \
    if (rc__) return rc__;  \
  } while (0)
#define LAUNCH1D(kern, n, ...)                                                              \
  do {                                                                                      \
    hipLaunchKernelGGL(kern, dim3(cdiv((n), 256)), dim3(256), 0, st, __VA_ARGS__);           \
    SNX_CHECK_LAUNCH();                                                                     \
  } while (0)
#define LAUNCH_ROWS(kern, T, ...)                                                           \
  do {                                                                                      \
    hipLaunchKernelGGL(kern, dim3(cdiv((T), 4)), dim3(256), 0, st, __VA_ARGS__);             \
    SNX_CHECK_LAUNCH();                                                                     \
  } while (0)

}  // namespace

extern "C" size_t snx_model_workspace_bytes_f32(const snx_model_desc* d, int32_t T, int32_t nseq, int32_t save_for_bwd) {
  if (!f32_desc_ok(d) || T <= 0 || nseq <= 0) return 0;
  Plan s;
  plan(d, T, nseq, save_for_bwd != 0, s);
  return s.total;
}

extern "C" size_t snx_model_bwd_workspace_bytes_f32(const snx_model_desc* d, int32_t T) {
  if (!f32_desc_ok(d) || T <= 0) return 0;
  BPlan p;
  bplan(d, T, p);
  return p.total;
}

extern "C" int snx_gemm_f32(const float* A, int64_t a_row, int64_t a_k, const float* B, int64_t b_row, int64_t b_k, float* C,
                            int64_t ldc, const float* R, int64_t ldr, int32_t M, int32_t N, int32_t K, int32_t accumulate,
                            hipStream_t st) {
  if (!A || !B || !C) return SNX_E_ARG;
  GemmArgs g{A, a_row, a_k, B, b_row, b_k, C, ldc, R, ldr, M, N, K, 0, 0};
  return accumulate ? launch_gemm<1>(g, TailArgs{}, st) : launch_gemm<0>(g, TailArgs{}, st);
}

extern "C" int snx_model_forward_f32(const snx_model_desc* d, const void* const* params, const int64_t* ids,
                                     const int64_t* mask, const int32_t* cu_seqlens, const int32_t* pos,
                                     const float* rope_global, const float* rope_local, void* saved, float* sparse,
                                     float* token_weights, int32_t T, int32_t nseq, int32_t flags, hipStream_t st) {
  if (!f32_desc_ok(d)) return SNX_E_SHAPE;
  if (!params || !ids || !mask || !cu_seqlens || !pos || !rope_global || !rope_local || !saved || !sparse || !token_weights ||
      T <= 0 || nseq <= 0)
    return SNX_E_ARG;
  Plan s;
  plan(d, T, nseq, (flags & SNX_FWD_SAVE_FOR_BACKWARD) != 0, s);
  PIdx p{d->layers};
  char* sv = (char*)saved;
  const int H = d->hidden, I = d->inter, V = d->vocab, L = d->layers, hd = d->head_dim;
  auto F = [&](int idx) { return (const float*)params[idx]; };
  auto hbuf = [&](int i) { return (float*)(sv + s.h[i]); };
  auto B = [&](size_t off) { return (float*)(sv + off); };
  int32_t* seqid = (int32_t*)(sv + s.seqid);
  LAUNCH1D(seqid_kernel, T, cu_seqlens, seqid, T, nseq);
  LAUNCH_ROWS(ln_f32_kernel<1>, T, F(p.tok_emb()), ids, F(p.emb_norm()), hbuf(0), T, H, d->ln_eps);
  const float scale = 1.0f / sqrtf((float)hd);
  for (int l = 0; l < L; ++l) {
    const bool global = (l % d->global_every) == 0;
    if (l > 0) LAUNCH_ROWS(ln_f32_kernel<0>, T, hbuf(2 * l), nullptr, F(p.attn_norm(l)), B(s.x_attn[l]), T, H, d->ln_eps);
    RC(linear_fwd(B(s.x_attn[l]), F(p.wqkv(l)), nullptr, B(s.qkv[l]), T, 3 * H, H, st));
    const long nrope = (long)T * 2 * d->heads * (hd / 2);
    LAUNCH1D(rope_f32_kernel, nrope, B(s.qkv[l]), (const f32x2*)(global ? rope_global : rope_local), pos, nrope, d->heads, hd, 0);
    if (hd % 8 == 0 && !g_snx_cfg.f32_attn_rows) {
      const int zq = max(1, min(64, cdiv(cdiv(T, nseq), AQ)));
      hipLaunchKernelGGL(attn_f32_fwd_tiled_kernel, dim3(nseq, d->heads, zq), dim3(256), 0, st, B(s.qkv[l]), cu_seqlens, mask,
                         B(s.attn[l]), B(s.lse[l]), T, d->heads, hd, global ? -1 : d->window, scale);
    } else {
      hipLaunchKernelGGL(attn_f32_fwd_kernel, dim3(cdiv((long)T * d->heads, 4)), dim3(256), 0, st, B(s.qkv[l]), cu_seqlens,
                         seqid, mask, B(s.attn[l]), B(s.lse[l]), T, d->heads, hd, global ? -1 : d->window, scale);
    }
    SNX_CHECK_LAUNCH();
    RC(linear_fwd(B(s.attn[l]), F(p.wo(l)), hbuf(2 * l), hbuf(2 * l + 1), T, H, H, st));
    LAUNCH_ROWS(ln_f32_kernel<0>, T, hbuf(2 * l + 1), nullptr, F(p.mlp_norm(l)), B(s.x_mlp[l]), T, H, d->ln_eps);
    RC(linear_fwd(B(s.x_mlp[l]), F(p.wi(l)), nullptr, B(s.u[l]), T, 2 * I, H, st));
    LAUNCH1D(geglu_f32_kernel, (long)T * I, B(s.u[l]), B(s.y[l]), (long)T * I, I);
    RC(linear_fwd(B(s.y[l]), F(p.wo_mlp(l)), hbuf(2 * l + 1), hbuf(2 * l + 2), T, H, I, st));
  }
  LAUNCH_ROWS(ln_f32_kernel<0>, T, hbuf(2 * L), nullptr, F(p.final_norm()), B(s.xf), T, H, d->ln_eps);
  RC(linear_fwd(B(s.xf), F(p.head_dense()), nullptr, B(s.dd), T, H, H, st));
  LAUNCH_ROWS(ln_f32_kernel<2>, T, B(s.dd), nullptr, F(p.head_norm()), B(s.hd), T, H, d->ln_eps);
  // tied decoder + SPLADE tail
  if (hipMemsetAsync(sv + s.keys, 0, (size_t)nseq * V * 8, st) != hipSuccess) return SNX_E_ARG;
  if (hipMemsetAsync(sv + s.twbits, 0, (size_t)T * 4, st) != hipSuccess) return SNX_E_ARG;
  {
    GemmArgs g{B(s.hd), H, 1, F(p.tok_emb()), H, 1, nullptr, 0, nullptr, 0, T, V, H, 0, 0};
    TailArgs ta{F(p.dec_bias()), mask, seqid, pos, (unsigned long long*)(sv + s.keys), (uint32_t*)(sv + s.twbits), V};
    RC(launch_gemm<2>(g, ta, st));
  }
  const long nv = (long)nseq * V;
  LAUNCH1D(tail_finalize_kernel, nv > T ? nv : T, (const unsigned long long*)(sv + s.keys), (const uint32_t*)(sv + s.twbits), sparse,
           token_weights, nv, T);
  return SNX_OK;
}

extern "C" int snx_model_backward_f32(const snx_model_desc* d, const void* const* params, void* const* grads,
                                      const int64_t* ids, const int64_t* mask, const int32_t* cu_seqlens, const int32_t* pos,
                                      const float* rope_global, const float* rope_local, const void* saved,
                                      const float* g_sparse, void* scratch, int32_t T, int32_t nseq, hipStream_t st) {
  if (!f32_desc_ok(d)) return SNX_E_SHAPE;
  if (!params || !grads || !ids || !mask || !cu_seqlens || !pos || !rope_global || !rope_local || !saved || !g_sparse ||
      !scratch || T <= 0 || nseq <= 0)
    return SNX_E_ARG;
  Plan s;
  plan(d, T, nseq, true, s);
  BPlan bp;
  bplan(d, T, bp);
  PIdx p{d->layers};
  const char* sv = (const char*)saved;
  char* sc = (char*)scratch;
  const int H = d->hidden, I = d->inter, V = d->vocab, L = d->layers, hd = d->head_dim;
  auto F = [&](int idx) { return (const float*)params[idx]; };
  auto G = [&](int idx) { return (float*)grads[idx]; };
  auto hbuf = [&](int i) { return (const float*)(sv + s.h[i]); };
  auto S = [&](size_t off) { return (const float*)(sv + off); };
  float* dh = (float*)(sc + bp.dh);
  float* A = (float*)(sc + bp.a);
  float* Bb = (float*)(sc + bp.b);
  float* Cc = (float*)(sc + bp.c);
  const int32_t* seqid = (const int32_t*)(sv + s.seqid);
  const float scale = 1.0f / sqrtf((float)hd);
  const long TH = (long)T * H;
  // ---- SPLADE tail + tied decoder (routed), head
  if (hipMemsetAsync(Cc, 0, TH * 4, st) != hipSuccess) return SNX_E_ARG;                  // dHd
  {
    const long nv = (long)nseq * V;
    hipLaunchKernelGGL(tail_f32_bwd_kernel, dim3(cdiv(nv, 4)), dim3(256), 0, st, g_sparse, (const unsigned long long*)(sv + s.keys),
                       S(s.hd), F(p.tok_emb()), cu_seqlens, Cc, G(p.tok_emb()), G(p.dec_bias()), nv, V, H);
    SNX_CHECK_LAUNCH();
  }
  // hd = LN(gelu(dd)) w: A = d(dd)
  LAUNCH_ROWS((ln_f32_bwd_kernel<2, false>), T, Cc, S(s.dd), nullptr, F(p.head_norm()), A, G(p.head_norm()), T, H, d->ln_eps, -1);
  RC(linear_dw(A, S(s.xf), G(p.head_dense()), T, H, H, st));
  RC(linear_dx(A, F(p.head_dense()), Bb, T, H, H, st));                                    // d(xf)
  LAUNCH_ROWS((ln_f32_bwd_kernel<0, false>), T, Bb, hbuf(2 * L), nullptr, F(p.final_norm()), dh, G(p.final_norm()), T, H, d->ln_eps, -1);
  for (int l = L - 1; l >= 0; --l) {
    const bool global = (l % d->global_every) == 0;
    // ---- MLP: h[2l+2] = h[2l+1] + Wo(gelu(a) g), [a | g] = Wi(LN(h[2l+1]))
    RC(linear_dw(dh, S(s.y[l]), G(p.wo_mlp(l)), T, H, I, st));
    RC(linear_dx(dh, F(p.wo_mlp(l)), A, T, H, I, st));                                     // dy [T, I]
    LAUNCH1D(geglu_f32_bwd_kernel, (long)T * I, S(s.u[l]), A, Bb, (long)T * I, I);          // du [T, 2I]
    RC(linear_dw(Bb, S(s.x_mlp[l]), G(p.wi(l)), T, 2 * I, H, st));
    RC(linear_dx(Bb, F(p.wi(l)), Cc, T, 2 * I, H, st));                                    // d(x_mlp)
    LAUNCH_ROWS((ln_f32_bwd_kernel<0, true>), T, Cc, hbuf(2 * l + 1), nullptr, F(p.mlp_norm(l)), dh, G(p.mlp_norm(l)), T, H, d->ln_eps, -1);
    // ---- attention: h[2l+1] = h[2l] + Wo(attn(rope(Wqkv(LN(h[2l])))))
    RC(linear_dw(dh, S(s.attn[l]), G(p.wo(l)), T, H, H, st));
    RC(linear_dx(dh, F(p.wo(l)), Cc, T, H, H, st));                                        // d(attn out)
    if (hipMemsetAsync(A, 0, (size_t)T * 3 * H * 4, st) != hipSuccess) return SNX_E_ARG;   // dqkv
    hipLaunchKernelGGL(attn_f32_bwd_kernel, dim3(cdiv((long)T * d->heads, 4)), dim3(256), 0, st, S(s.qkv[l]), S(s.attn[l]), Cc,
                       S(s.lse[l]), cu_seqlens, seqid, mask, A, T, d->heads, hd, global ? -1 : d->window, scale);
    SNX_CHECK_LAUNCH();
    const long nrope = (long)T * 2 * d->heads * (hd / 2);
    LAUNCH1D(rope_f32_kernel, nrope, A, (const f32x2*)(global ? rope_global : rope_local), pos, nrope, d->heads, hd, 1);
    RC(linear_dw(A, S(s.x_attn[l]), G(p.wqkv(l)), T, 3 * H, H, st));
    RC(linear_dx(A, F(p.wqkv(l)), Cc, T, 3 * H, H, st));                                   // d(x_attn)
    if (l > 0) {
      LAUNCH_ROWS((ln_f32_bwd_kernel<0, true>), T, Cc, hbuf(2 * l), nullptr, F(p.attn_norm(l)), dh, G(p.attn_norm(l)), T, H, d->ln_eps, -1);
    } else {
      LAUNCH1D(add_f32_kernel, TH, dh, Cc, TH);
    }
  }
  LAUNCH_ROWS((ln_f32_bwd_kernel<1, false>), T, dh, F(p.tok_emb()), ids, F(p.emb_norm()), G(p.tok_emb()), G(p.emb_norm()), T, H, d->ln_eps,
              d->pad_id);
  return SNX_OK;
}
