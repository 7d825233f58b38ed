// Fused MLM-decoder GEMM + SPLADE tail (K10 + K11 of SURVEY.md §2.3):
//   logits = Hd @ W_E^T + b  (bf16 GEMM, tied embedding matrix; hf modeling_modernbert.py:550)
//   sparse_repr[b, v] = max_s  log1p(relu(logits[b, s, v])) * mask[b, s]
//   token_weights[b, s] = max_v log1p(relu(logits[b, s, v])) * mask[b, s]
// (ref:src/model/splade_modern.py:76-86).  The [B,S,V] logits / fp32 score tensors (1.5 + 3 GiB
// per document pass in the reference) are never written: log1p(relu(.)) and the bf16 rounding
// are monotone, so the max is taken on the raw accumulators and log1p is applied to the B*V
// (and T) reduced values only.
//
// One workgroup owns a (sequence, 128-column vocab tile) pair and walks the sequence's rows in
// BM-row chunks, keeping per-column running maxima in registers as packed keys
//   key = bf16_bits(relu(logit)) << 16 | (0xFFFF - s)
// (unsigned max = largest value, ties -> smallest s: torch's first-index argmax).  The key is
// saved per (b, v): the backward pass routes the gradient to exactly that row (row a7 of §8).
#include "gemm_core.h"
#include "snx.h"

template <int BM>
__global__ __launch_bounds__(256) void decoder_splade_kernel(
    const bf16_t* __restrict__ Hd, const bf16_t* __restrict__ W, const float* __restrict__ bias,
    const int32_t* __restrict__ cu_seqlens, const int64_t* __restrict__ mask, float* __restrict__ sparse,
    uint32_t* __restrict__ keys, unsigned short* __restrict__ rowpart, int T, int V, int K, int n_tiles,
    int total_tiles) {
  constexpr int BN = 128;
  using Core = GemmCore<BM, BN, 2, 2>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint32_t* sBest = (uint32_t*)(smem + Core::LDS_BYTES);           // [2][BN]
  uint32_t* sRow = sBest + 2 * BN;                                 // [2][BM]
  const int tile = xcd_remap(blockIdx.x, total_tiles);
  const int seq = tile / n_tiles, nt = tile % n_tiles;
  const int n0 = nt * BN;
  const int s0 = cu_seqlens[seq], slen = cu_seqlens[seq + 1] - s0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 15;

  float bcol[Core::NI];
  bool colok[Core::NI];
#pragma unroll
  for (int j = 0; j < Core::NI; ++j) {
    const int col = n0 + Core::acc_col(j);
    colok[j] = col < V;
    bcol[j] = colok[j] ? rbf(bias[col]) : 0.f;
  }
  uint32_t best[Core::NI];
#pragma unroll
  for (int j = 0; j < Core::NI; ++j) best[j] = 0u;

  for (int c0 = 0; c0 < slen; c0 += BM) {
    f32x4 acc[Core::MI][Core::NI];
#pragma unroll
    for (int i = 0; i < Core::MI; ++i)
#pragma unroll
      for (int j = 0; j < Core::NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();                                  // LDS (tiles + sRow) free for this chunk
    Core::mainloop(Hd, K, s0 + c0, s0 + slen, W, K, n0, V, K, smem, acc);
#pragma unroll
    for (int i = 0; i < Core::MI; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int lrow = Core::acc_row(i, r);         // row inside the chunk
        const int srow = c0 + lrow;                   // position inside the sequence
        const bool valid = srow < slen && mask[s0 + (srow < slen ? srow : slen - 1)] != 0;
        const uint32_t rtag = 0xFFFFu - (uint32_t)srow;
        uint32_t rb = 0u;
#pragma unroll
        for (int j = 0; j < Core::NI; ++j) {
          const float v = fmaxf(acc[i][j][r] + bcol[j], 0.f);
          const uint32_t bits = (valid && colok[j]) ? bf16_bits(v) : 0u;
          const uint32_t key = (bits << 16) | rtag;
          best[j] = (valid && key > best[j]) ? key : best[j];
          rb = bits > rb ? bits : rb;
        }
        rb = max(rb, (uint32_t)__shfl_xor((int)rb, 1, 64));
        rb = max(rb, (uint32_t)__shfl_xor((int)rb, 2, 64));
        rb = max(rb, (uint32_t)__shfl_xor((int)rb, 4, 64));
        rb = max(rb, (uint32_t)__shfl_xor((int)rb, 8, 64));
        if (li == 0) sRow[wn * BM + lrow] = rb;
      }
    }
    __syncthreads();
    for (int rr = threadIdx.x; rr < BM; rr += 256) {
      const int srow = c0 + rr;
      if (srow < slen) {
        const uint32_t a = sRow[rr], b = sRow[BM + rr];
        rowpart[(long)nt * T + s0 + srow] = (unsigned short)(a > b ? a : b);
      }
    }
  }

#pragma unroll
  for (int j = 0; j < Core::NI; ++j) {
    uint32_t b = best[j];
    b = max(b, (uint32_t)__shfl_xor((int)b, 16, 64));
    b = max(b, (uint32_t)__shfl_xor((int)b, 32, 64));
    if (lane < 16) sBest[wm * BN + wn * Core::WTN + j * 16 + li] = b;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < BN; c += 256) {
    const int col = n0 + c;
    if (col < V) {
      const uint32_t a = sBest[c], b = sBest[BN + c];
      const uint32_t k = a > b ? a : b;
      keys[(long)seq * V + col] = k;
      sparse[(long)seq * V + col] = log1pf(bits_to_f32(k >> 16));
    }
  }
}

// token_weights[t] = mask[t] ? log1p(max over vocab tiles of rowpart[nt][t]) : 0
__global__ void token_weights_kernel(const unsigned short* __restrict__ rowpart, const int64_t* __restrict__ mask,
                                     float* __restrict__ tw, int T, int n_tiles) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  uint32_t m = 0;
  for (int nt = 0; nt < n_tiles; ++nt) {
    const uint32_t v = rowpart[(long)nt * T + t];
    m = v > m ? v : m;
  }
  tw[t] = mask[t] != 0 ? log1pf(bits_to_f32(m)) : 0.f;
}

extern "C" size_t snx_splade_head_scratch_bytes(int32_t T, int32_t V) { return (size_t)cdiv(V, 128) * T * 2; }

extern "C" int snx_decoder_splade_fwd(const void* Hd, const void* W, const float* bias, const int32_t* cu_seqlens,
                                      const int64_t* mask, float* sparse, uint32_t* keys, float* token_weights,
                                      void* scratch, int32_t T, int32_t nseq, int32_t max_seqlen, int32_t V,
                                      int32_t K, hipStream_t st) {
  if (!Hd || !W || !bias || !cu_seqlens || !mask || !sparse || !keys || !token_weights || !scratch) return SNX_E_ARG;
  if (T <= 0 || nseq <= 0 || V <= 0 || K <= 0 || (K % 64) || max_seqlen > 65535) return SNX_E_SHAPE;
  const int n_tiles = cdiv(V, 128);
  const int total = n_tiles * nseq;
  unsigned short* rowpart = (unsigned short*)scratch;
  if (max_seqlen <= 64) {
    using Core = GemmCore<64, 128, 2, 2>;
    const size_t lds = Core::LDS_BYTES + (2 * 128 + 2 * 64) * 4;
    hipLaunchKernelGGL(decoder_splade_kernel<64>, dim3(total), dim3(256), lds, st, (const bf16_t*)Hd,
                       (const bf16_t*)W, bias, cu_seqlens, mask, sparse, keys, rowpart, T, V, K, n_tiles, total);
  } else {
    using Core = GemmCore<128, 128, 2, 2>;
    const size_t lds = Core::LDS_BYTES + (2 * 128 + 2 * 128) * 4;
    hipLaunchKernelGGL(decoder_splade_kernel<128>, dim3(total), dim3(256), lds, st, (const bf16_t*)Hd,
                       (const bf16_t*)W, bias, cu_seqlens, mask, sparse, keys, rowpart, T, V, K, n_tiles, total);
  }
  SNX_CHECK_LAUNCH();
  hipLaunchKernelGGL(token_weights_kernel, dim3(cdiv(T, 256)), dim3(256), 0, st, rowpart, mask, token_weights, T,
                     n_tiles);
  SNX_CHECK_LAUNCH();
  return SNX_OK;
}
