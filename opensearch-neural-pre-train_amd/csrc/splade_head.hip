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
#include <cstdlib>

#include "gemm_core.h"
#include "config.h"
#include "snx.h"

#ifndef SNX_DEC_GROUP
#define SNX_DEC_GROUP 8
#endif

template <int BM>
__global__ __launch_bounds__(256, 2) void decoder_splade_kernel(
    const bf16_t* __restrict__ Hd, const bf16_t* __restrict__ W, const float* __restrict__ bias,
    const int32_t* __restrict__ cu_seqlens, const int64_t* __restrict__ mask, float* __restrict__ sparse,
    uint32_t* __restrict__ keys, unsigned short* __restrict__ rowpart, int T, int V, int K, int n_tiles,
    int total_tiles, int nseq) {
  constexpr int BN = 128;
  using Core = GemmCore<BM, BN, 2, 2>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint32_t* sBest = (uint32_t*)(smem + Core::LDS_BYTES);           // [2][BN]
  uint32_t* sRow = sBest + 2 * BN;                                 // [2][BM]
  // Work order (speed only).  Blocks b and b+8 share an XCD (and its L2).  Sequences are taken in
  // GROUPS of SNX_DEC_GROUP; group g is placed on XCD g % 8 and, inside it, ids walk vocab tiles with
  // the group's sequences innermost.  The ~64 workgroups resident on one XCD therefore cover
  // (group sequences) x (a few vocab tiles): a W_E tile fetched into that L2 serves the whole group
  // instead of one sequence (W_E was re-streamed once per sequence before: ~9.7 GB of fabric traffic
  // per 192-sequence launch) while the group's Hd rows (G x S x 1.5 KB) stay resident; dealing the
  // groups round-robin keeps the XCDs balanced when short (query) and long (document) sequences
  // share a launch.
  const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
  const int per_group = SNX_DEC_GROUP * n_tiles;
  const int grp = (k / per_group) * 8 + xcd, within = k % per_group;
  const int gsz = min(SNX_DEC_GROUP, nseq - grp * SNX_DEC_GROUP);
  if (gsz <= 0 || within >= gsz * n_tiles) return;
  const int nt = within / gsz, seq = grp * SNX_DEC_GROUP + (within - nt * gsz);
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
  // 8 independent loads in flight per thread: one dependent load per tile made this pass latency-bound (130 us for
  // 522 tiles at T = 36,864)
  uint32_t m = 0;
  int nt = 0;
  for (; nt + 8 <= n_tiles; nt += 8) {
    uint32_t v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = rowpart[(long)(nt + q) * T + t];
#pragma unroll
    for (int q = 0; q < 8; ++q) m = v[q] > m ? v[q] : m;
  }
  for (; nt < n_tiles; ++nt) {
    const uint32_t v = rowpart[(long)nt * T + t];
    m = v > m ? v : m;
  }
  tw[t] = mask[t] != 0 ? log1pf(bits_to_f32(m)) : 0.f;
}

// 256x256 persistent form (decoder256.hip)
size_t snx_dec256_table_bytes(int32_t T);
int snx_dec256_rowtiles(int32_t V);                   // rows of the row-maximum array it writes (96-column half tiles)
int snx_launch_decoder256(const void* Hd, const void* W, const float* bias, const int32_t* cu_seqlens,
                          const int64_t* mask, float* sparse, uint32_t* keys, void* scratch, size_t rowpart_bytes,
                          int32_t T, int32_t nseq, int32_t V, int32_t K, hipStream_t st);

// SNX_DEC256=0 keeps the 128x128 kernel; the 256x192 form wants enough rows to fill its tiles.  (model.hip asks: the
// 256x192 form takes all sequence groups of a pass in ONE call, the 128x128 kernel one call per group.)
bool snx_dec256_takes(int32_t T) {
  return g_snx_cfg.dec256 && T >= g_snx_cfg.dec256_min_t;
}

// row maxima [tiles, T] ushort (tiles = ceil(V / 128) for the 128x128 kernel, 2 ceil(V / 192) for the 256x192 form)
// + the pre-pass tables of the latter
extern "C" size_t snx_splade_head_scratch_bytes(int32_t T, int32_t V) {
  const size_t tiles = (size_t)(cdiv(V, 128) > snx_dec256_rowtiles(V) ? cdiv(V, 128) : snx_dec256_rowtiles(V));
  return ((tiles * T * 2 + 255) & ~(size_t)255) + snx_dec256_table_bytes(T);
}

// `finalize` = 0 skips the token_weights pass (used when several sequence groups of one token
// buffer are processed by separate calls; the last call finalises all T rows).
extern "C" int snx_decoder_splade_fwd_ex(const void* Hd, const void* W, const float* bias,
                                         const int32_t* cu_seqlens, const int64_t* mask, float* sparse,
                                         uint32_t* keys, float* token_weights, void* scratch, int32_t T,
                                         int32_t nseq, int32_t max_seqlen, int32_t V, int32_t K, int32_t finalize,
                                         hipStream_t st) {
  if (!Hd || !W || !bias || !cu_seqlens || !mask || !sparse || !keys || !token_weights || !scratch) return SNX_E_ARG;
  if (T <= 0 || nseq <= 0 || V <= 0 || K <= 0 || (K % 64) || max_seqlen > 65535) return SNX_E_SHAPE;
  const int n_tiles = cdiv(V, 128);
  if (snx_dec256_takes(T)) {
    const int rowtiles = snx_dec256_rowtiles(V);
    const size_t tiles = (size_t)(n_tiles > rowtiles ? n_tiles : rowtiles);
    const int rc = snx_launch_decoder256(Hd, W, bias, cu_seqlens, mask, sparse, keys, scratch, tiles * T * 2, T, nseq,
                                         V, K, st);
    if (rc != SNX_E_SHAPE) {
      if (rc != SNX_OK) return rc;
      if (finalize) {
        hipLaunchKernelGGL(token_weights_kernel, dim3(cdiv(T, 256)), dim3(256), 0, st, (const unsigned short*)scratch,
                           mask, token_weights, T, rowtiles);
        SNX_CHECK_LAUNCH();
      }
      return SNX_OK;
    }
  }
  const long total_l = 8L * cdiv(cdiv(nseq, SNX_DEC_GROUP), 8) * SNX_DEC_GROUP * n_tiles;
  if (total_l > 0x7fffffffL) return SNX_E_SHAPE;
  const int total = (int)total_l;
  unsigned short* rowpart = (unsigned short*)scratch;
  if (max_seqlen <= 64) {
    using Core = GemmCore<64, 128, 2, 2>;
    const size_t lds = Core::LDS_BYTES + (2 * 128 + 2 * 64) * 4;
    hipLaunchKernelGGL(decoder_splade_kernel<64>, dim3(total), dim3(256), lds, st, (const bf16_t*)Hd,
                       (const bf16_t*)W, bias, cu_seqlens, mask, sparse, keys, rowpart, T, V, K, n_tiles, total,
                       nseq);
  } else {
    using Core = GemmCore<128, 128, 2, 2>;
    const size_t lds = Core::LDS_BYTES + (2 * 128 + 2 * 128) * 4;
    hipLaunchKernelGGL(decoder_splade_kernel<128>, dim3(total), dim3(256), lds, st, (const bf16_t*)Hd,
                       (const bf16_t*)W, bias, cu_seqlens, mask, sparse, keys, rowpart, T, V, K, n_tiles, total,
                       nseq);
  }
  SNX_CHECK_LAUNCH();
  if (finalize) {
    hipLaunchKernelGGL(token_weights_kernel, dim3(cdiv(T, 256)), dim3(256), 0, st, rowpart, mask, token_weights, T,
                       n_tiles);
    SNX_CHECK_LAUNCH();
  }
  return SNX_OK;
}

extern "C" int snx_decoder_splade_fwd(const void* Hd, const void* W, const float* bias, const int32_t* cu_seqlens,
                                      const int64_t* mask, float* sparse, uint32_t* keys, float* token_weights,
                                      void* scratch, int32_t T, int32_t nseq, int32_t max_seqlen, int32_t V,
                                      int32_t K, hipStream_t st) {
  return snx_decoder_splade_fwd_ex(Hd, W, bias, cu_seqlens, mask, sparse, keys, token_weights, scratch, T, nseq,
                                   max_seqlen, V, K, 1, st);
}

// ==========================================================================================
// Backward of the fused decoder + SPLADE tail (row a7 of SURVEY.md §8).
// Autograd in the reference materialises a dense [T, V] dlogits tensor that is zero except at one
// sequence position per (b, v) (the arg-max row), then runs two dense 2*T*V*H GEMMs.  Here the
// gradient is routed through the saved packed keys instead:
//   c[b,v]   = bf16( g[b,v] * [x > 0] / (1 + x) ),  x = bf16 logit at the arg-max row s*(b,v)
//   dW[v,:] += sum_b c[b,v] * Hd[row(b, s*), :]          (one wave per vocab row: gather + FMA)
//   db[v]   += sum_b c[b,v]
//   dHd[row(b,s), :] = sum_{v: s*(b,v) = s} c[b,v] * W[v,:]   (per-sequence LDS accumulation)
// Entries with c == 0 (inactive vocabulary terms -- the vast majority once the model is
// trained) are skipped, so the cost follows the activation sparsity.
// ==========================================================================================
__device__ __forceinline__ float splade_coef(float g, uint32_t key) {
  const float x = bits_to_f32(key >> 16);
  return x > 0.f ? rbf(g / (1.0f + x)) : 0.f;
}

// Gathers are issued in batches of GB rows so that GB independent loads per wave are in flight (a
// one-at-a-time loop is latency-bound: ~1 us per dependent row fetch).
#define GB 8

// dE[v] += sum_b c[b,v] * Hd[row(b,v)],  db[v] += sum_b c[b,v].
// A wave owns 8 consecutive vocab rows (accumulators in registers) and ALL waves sweep the
// sequences in the same order, 8 at a time (lane = 8 * (b - b0) + (v - v0): the key/gradient reads
// are 32-byte runs).  The sweep order is what matters: the workgroups resident on an XCD are then
// gathering from the same few sequences at any moment and the Hd rows come out of that XCD's L2
// (with one wave per vocab row running its own sweep the gathers touched all of Hd at once and
// every 1.5 KB row crossed the fabric: 13.6 GB per 192-sequence step, 7.6 TB/s, 1.8 ms).
// Zero coefficients (most entries once the model is trained) are skipped wave-uniformly.
// NT ("splade_dw_last"): the read-modify-write of the gradient rows (2 x 153 MB, no reader before the optimizer / the
// embedding unit at the end of the backward) through non-temporal accesses
template <int NV, bool NT = false>
__global__ __launch_bounds__(256) void splade_bwd_dw_kernel(const float* __restrict__ g,
                                                            const uint32_t* __restrict__ keys,
                                                            const bf16_t* __restrict__ Hd,
                                                            const int32_t* __restrict__ cu_seqlens,
                                                            float* __restrict__ gradE, float* __restrict__ gradb,
                                                            int nseq, int V, int H) {
  const int lane = threadIdx.x & 63;
  const int v0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 8;
  if (v0 >= V) return;
  const int vl = lane & 7, bl = lane >> 3;
  const int v = v0 + vl;
  f32x4 acc[8][NV];
#pragma unroll
  for (int k = 0; k < 8; ++k)
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[k][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  for (int b0 = 0; b0 < nseq; b0 += 8) {
    const int b = b0 + bl;
    float c = 0.f;
    int trow = 0;
    if (b < nseq && v < V) {
      const uint32_t key = keys[(long)b * V + v];
      c = splade_coef(g[(long)b * V + v], key);
      trow = cu_seqlens[b] + (int)(0xFFFFu - (key & 0xFFFFu));
    }
    bsum += c;
    const unsigned long long m = __ballot(c != 0.f);
    if (!m) continue;
#pragma unroll
    for (int k = 0; k < 8; ++k) {                     // vocab row v0 + k: lanes k, k + 8, ..., k + 56
      if (!(m & (0x0101010101010101ull << k))) continue;
      float cb[GB];
      bf16x4 hv[GB][NV];
#pragma unroll
      for (int u = 0; u < GB; ++u) {
        cb[u] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, c), u * 8 + k));
        const int tr = __builtin_amdgcn_readlane(trow, u * 8 + k);
        if (cb[u] != 0.f) {                           // wave-uniform
          const bf16_t* hrow = Hd + (long)tr * H;
#pragma unroll
          for (int i = 0; i < NV; ++i) hv[u][i] = *(const bf16x4*)(hrow + (i * 64 + lane) * 4);
        } else {
#pragma unroll
          for (int i = 0; i < NV; ++i) hv[u][i] = (bf16x4){0, 0, 0, 0};
        }
      }
#pragma unroll
      for (int u = 0; u < GB; ++u)
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          acc[k][i][0] += cb[u] * bf2f(hv[u][i][0]);
          acc[k][i][1] += cb[u] * bf2f(hv[u][i][1]);
          acc[k][i][2] += cb[u] * bf2f(hv[u][i][2]);
          acc[k][i][3] += cb[u] * bf2f(hv[u][i][3]);
        }
    }
  }
  bsum += __shfl_xor(bsum, 8, 64);
  bsum += __shfl_xor(bsum, 16, 64);
  bsum += __shfl_xor(bsum, 32, 64);
  if (lane < 8 && v < V) gradb[v] += bsum;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if (v0 + k >= V) break;
    float* dst = gradE + (long)(v0 + k) * H;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      f32x4* p = (f32x4*)(dst + (i * 64 + lane) * 4);
      if (NT) __builtin_nontemporal_store(__builtin_nontemporal_load(p) + acc[k][i], p);
      else *p = *p + acc[k][i];
    }
  }
}

// dHd: LDS float atomics run at only ~1 lane per 2-3 cycles on this chip (measured: 16 ms per
// document pass when the [S][H] slab was accumulated with ds_add_f32), so the routed entries are
// first BUCKETED by target row (counting sort per sequence, integer LDS atomics on S counters
// only), then one wave per token row gathers its W rows and accumulates in registers -- no float
// atomics, every dHd row written exactly once (rows without entries get zeros).
// Slot order inside a row's bucket = vocabulary order, so that the fp32 accumulation order of the gather
// kernel below -- and with it dHd -- is the same in every run (two-run bit-compare).  Each of the
// 16 waves owns a contiguous vocabulary range and keeps its own per-row counters; a wave's slots for a row
// start after those of the waves before it, and inside one 64-entry step lanes that hit the same row are
// ranked by lane id (bitonic sort of (row, lane) across the wave + segment heads by ballot: no atomics
// whose arrival order could vary).  DET = false (rows > 1024: the per-wave tables no longer fit) falls back
// to first-come slots from an LDS atomic counter.
template <bool DET>
__global__ __launch_bounds__(1024) void splade_bucket_kernel(const float* __restrict__ g,
                                                             const uint32_t* __restrict__ keys,
                                                             const int32_t* __restrict__ cu_seqlens,
                                                             int32_t* __restrict__ list_v, float* __restrict__ list_c,
                                                             int32_t* __restrict__ row_off, int V, int max_rows) {
  extern __shared__ __attribute__((aligned(16))) int32_t smi[];
  int32_t* cnt = smi;                    // [max_rows]
  int32_t* off = smi + max_rows;         // [max_rows + 1]
  int32_t* fill = off + max_rows + 1;    // [max_rows]            (DET: unused)
  int32_t* wtab = fill + max_rows;       // [16][max_rows]        (DET only)
  const int seq = blockIdx.x;
  const int slen = cu_seqlens[seq + 1] - cu_seqlens[seq];
  const int rows = slen < max_rows ? slen : max_rows;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < max_rows; i += blockDim.x) { cnt[i] = 0; fill[i] = 0; }
  if (DET)
    for (int i = threadIdx.x; i < 16 * max_rows; i += blockDim.x) wtab[i] = 0;
  __syncthreads();
  const float* gs = g + (long)seq * V;
  const uint32_t* ks = keys + (long)seq * V;
  const int per_wave = ((V + 16 * 64 - 1) / (16 * 64)) * 64;          // vocabulary range of one wave
  const int vb = wave * per_wave, ve = min(V, vb + per_wave);
  if (DET) {
    int32_t* mine = wtab + wave * max_rows;
    for (int v = vb + lane; v < ve; v += 64) {
      const uint32_t key = ks[v];
      const int row = (int)(0xFFFFu - (key & 0xFFFFu));
      if (splade_coef(gs[v], key) != 0.f && row < rows) atomicAdd(&mine[row], 1);
    }
    __syncthreads();
    for (int r = threadIdx.x; r < rows; r += blockDim.x) {            // wave-exclusive prefix per row + row total
      int run = 0;
#pragma unroll
      for (int w = 0; w < 16; ++w) { const int c = wtab[w * max_rows + r]; wtab[w * max_rows + r] = run; run += c; }
      cnt[r] = run;
    }
  } else {
    for (int v = threadIdx.x; v < V; v += blockDim.x) {
      const uint32_t key = ks[v];
      const int row = (int)(0xFFFFu - (key & 0xFFFFu));
      if (splade_coef(gs[v], key) != 0.f && row < rows) atomicAdd(&cnt[row], 1);
    }
  }
  __syncthreads();
  if (threadIdx.x < 64) {                // exclusive scan of cnt[0..rows) by one wave
    const int per = (rows + 63) / 64;
    int local = 0;
    for (int i = 0; i < per; ++i) {
      const int r = lane * per + i;
      if (r < rows) local += cnt[r];
    }
    int incl = local;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(incl, o, 64);
      if (lane >= o) incl += t;
    }
    int run = incl - local;
    for (int i = 0; i < per; ++i) {
      const int r = lane * per + i;
      if (r < rows) { off[r] = run; run += cnt[r]; }
    }
    if (lane == 63) off[rows] = incl;
  }
  __syncthreads();
  int32_t* lv = list_v + (long)seq * V;
  float* lc = list_c + (long)seq * V;
  if (DET) {
    int32_t* mine = wtab + wave * max_rows;                           // running slot offset of this wave per row
    for (int v0 = vb; v0 < ve; v0 += 64) {                            // wave-uniform trip count
      const int v = v0 + lane;
      uint32_t key = 0;
      float c = 0.f;
      if (v < ve) { key = ks[v]; c = splade_coef(gs[v], key); }
      const int row = (int)(0xFFFFu - (key & 0xFFFFu));
      const bool act = c != 0.f && row < rows;
      // rank of every entry among the entries of this step that hit the same row, lower lane (= lower vocabulary
      // id) first: bitonic sort of (row << 6 | lane) across the wave, then position minus position of the
      // segment head.  No atomics whose arrival order could vary, and no LDS round trip per distinct row.
      // (idle lanes sort to the end with their own lane id in the low bits, so that the hand-back below is a bijection)
      uint32_t sk = act ? (((uint32_t)row << 6) | (uint32_t)lane) : (0xFFFFFFC0u | (uint32_t)lane);
#pragma unroll
      for (int k = 2; k <= 64; k <<= 1)
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
          const uint32_t other = (uint32_t)__shfl_xor((int)sk, j, 64);
          const bool up = ((lane & k) == 0) || k == 64;               // ascending overall
          const bool lower = (lane & j) == 0;
          const uint32_t mn = sk < other ? sk : other, mx = sk < other ? other : sk;
          sk = (lower == up) ? mn : mx;
        }
      const bool sact = sk < 0xFFFFFFC0u;                             // sorted position `lane` holds a live entry
      const int srow = (int)(sk >> 6), src = (int)(sk & 63u);
      const int prev_row = __shfl_up(srow, 1, 64);
      const bool head = sact && (lane == 0 || prev_row != srow);
      const unsigned long long heads = __ballot(head), live = __ballot(sact);
      // position of this segment's head = highest head bit at or below this lane; its end = next head bit or the live count
      const unsigned long long below = heads & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));
      const int hpos = 63 - __builtin_clzll(below | 1ull);
      const unsigned long long above = heads & ~((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));
      const int nlive = __popcll(live);
      const int epos = above ? (__ffsll((long long)above) - 1) : nlive;
      int slot_sorted = 0;
      if (sact) {
        const int base = off[srow] + mine[srow];
        slot_sorted = base + (lane - hpos);
      }
      if (head) mine[srow] += epos - hpos;                            // one writer per row and step
      // hand every slot back to the lane that owns the entry (sorted position -> source lane)
      const int slot = __builtin_amdgcn_ds_permute(src << 2, sact ? slot_sorted : 0);
      if (act) { lv[slot] = v; lc[slot] = c; }
    }
  } else {
    for (int v = threadIdx.x; v < V; v += blockDim.x) {
      const uint32_t key = ks[v];
      const int row = (int)(0xFFFFu - (key & 0xFFFFu));
      const float c = splade_coef(gs[v], key);
      if (c != 0.f && row < rows) {
        const int slot = off[row] + atomicAdd(&fill[row], 1);
        lv[slot] = v;
        lc[slot] = c;
      }
    }
  }
  for (int r = threadIdx.x; r <= rows; r += blockDim.x) row_off[(long)seq * (max_rows + 1) + r] = off[r];
}

template <int NV>
__global__ __launch_bounds__(256) void splade_bwd_dh_rows_kernel(const int32_t* __restrict__ list_v,
                                                                 const float* __restrict__ list_c,
                                                                 const int32_t* __restrict__ row_off,
                                                                 const bf16_t* __restrict__ W,
                                                                 const int32_t* __restrict__ cu_seqlens,
                                                                 bf16_t* __restrict__ dHd, int V, int H, int max_rows) {
  const int seq = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int s0 = cu_seqlens[seq], slen = cu_seqlens[seq + 1] - s0;
  if (row >= slen) return;
  f32x4 acc[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (row < max_rows) {
    const int32_t* ro = row_off + (long)seq * (max_rows + 1);
    const int beg = ro[row], end = ro[row + 1];
    const int32_t* lv = list_v + (long)seq * V;
    const float* lc = list_c + (long)seq * V;
    for (int e0 = beg; e0 < end; e0 += 64) {
      const int e = e0 + lane;
      const int vv = e < end ? lv[e] : 0;
      const float cc = e < end ? lc[e] : 0.f;
      const int n = min(64, end - e0);
      for (int u0 = 0; u0 < n; u0 += GB) {
        float cb[GB];
        bf16x4 wv[GB][NV];
#pragma unroll
        for (int u = 0; u < GB; ++u) {
          const int src = u0 + u;                               // wave-uniform
          cb[u] = src < n ? __shfl(cc, src, 64) : 0.f;
          const bf16_t* wrow = W + (long)__shfl(vv, src < n ? src : 0, 64) * H;
#pragma unroll
          for (int i = 0; i < NV; ++i) wv[u][i] = *(const bf16x4*)(wrow + (i * 64 + lane) * 4);
        }
#pragma unroll
        for (int u = 0; u < GB; ++u)
#pragma unroll
          for (int i = 0; i < NV; ++i) {
            acc[i][0] += cb[u] * bf2f(wv[u][i][0]);
            acc[i][1] += cb[u] * bf2f(wv[u][i][1]);
            acc[i][2] += cb[u] * bf2f(wv[u][i][2]);
            acc[i][3] += cb[u] * bf2f(wv[u][i][3]);
          }
      }
    }
  }
  bf16_t* out = dHd + (long)(s0 + row) * H;
#pragma unroll
  for (int i = 0; i < NV; ++i)
    *(bf16x4*)(out + (i * 64 + lane) * 4) = (bf16x4){f2bf(acc[i][0]), f2bf(acc[i][1]), f2bf(acc[i][2]), f2bf(acc[i][3])};
}

// Panel form of the gather above for vocabulary-ordered buckets (the default; SNX_SPLADE_DH_PANELS=0: the kernel above).
// A wave owns RW = 8 rows (accumulators in registers) and walks the vocabulary in panels -- for panel p: for each of its
// rows: the row's entries below the panel's end -- inside a persistent launch of 8 waves per CU that takes (sequence,
// 8-row group) items from a prefix table.  Per row the entries are taken in bucket order, as above: same accumulation
// order, same dHd bits.  A row's next 64 entries (v, c) stay in two registers per row between panels.
// Measured at the bench's worst case (192 sequences, every one of the 9.6 M (sequence, vocabulary) entries active,
// tools/gpu_splade_bwd_ab.py): 1.58 instead of 1.84 ms, 11.0 instead of 13.8 GB fetched (W_E rows are shared through
// the L2 while the waves of a launch are still inside the same panel; they drift apart like a random walk, and 8-16
// panels are the optimum).  Holding the waves together with a per-panel progress counter (start panel p when all waves
// have finished p - 2; bounded spin) cost far more than it saved: 9-45 ms per call at 16-128 panels.
constexpr int RW = 8, GP = 4, DH_WAVES_PER_CU = 8, DH_WGS = 256 * DH_WAVES_PER_CU / 4;

// items[s] = number of RW-row groups of the sequences before s (one workgroup; nseq + 1 entries + the item counter)
__global__ __launch_bounds__(256) void splade_dh_items_kernel(const int32_t* __restrict__ cu_seqlens, int32_t* __restrict__ items,
                                                              int nseq, int max_rows) {
  __shared__ int part[256];
  const int t = threadIdx.x;
  const int per = (nseq + 255) / 256;
  const int b = min(nseq, t * per), e = min(nseq, b + per);
  int mine = 0;
  for (int s = b; s < e; ++s) mine += (min(cu_seqlens[s + 1] - cu_seqlens[s], max_rows) + RW - 1) / RW;
  part[t] = mine;
  __syncthreads();
  if (t == 0) {
    int run = 0;
    for (int i = 0; i < 256; ++i) {
      const int v = part[i];
      part[i] = run;
      run += v;
    }
    items[nseq] = run;
    items[nseq + 1] = 0;                              // the gather's item counter
  }
  __syncthreads();
  int k = part[t];
  for (int s = b; s < e; ++s) {
    items[s] = k;
    k += (min(cu_seqlens[s + 1] - cu_seqlens[s], max_rows) + RW - 1) / RW;
  }
}

// NTL ("splade_dw_last" = 2): the bucket lists (77 MB, read once) through non-temporal loads -- the gather is bound by how
// fast W_E rows come out of the Infinity Cache, and everything else that passes through it competes with them
template <int NV, bool NTL = false>
__global__ __launch_bounds__(256) void splade_bwd_dh_panels_kernel(const int32_t* __restrict__ list_v,
                                                                   const float* __restrict__ list_c,
                                                                   const int32_t* __restrict__ row_off,
                                                                   const int32_t* __restrict__ items,
                                                                   const bf16_t* __restrict__ W,
                                                                   const int32_t* __restrict__ cu_seqlens,
                                                                   bf16_t* __restrict__ dHd, int V, int H, int max_rows,
                                                                   int nseq, int npanel, int* __restrict__ counter) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nitems = items[nseq];
  const int pw = (V + npanel - 1) / npanel;
  // persistent: every wave of the launch is resident from the start.  Items are handed out by a counter, first come
  // first served: rows of short sequences own more entries each (V / length on average at random init), and a fixed
  // assignment (wave, wave + all, ...) left the waves with the heavy items a round behind
#pragma unroll 1
  while (true) {
    int item = 0;
    if (lane == 0) item = atomicAdd(counter, 1);
    item = __builtin_amdgcn_readfirstlane(item);
    if (item >= nitems) break;
    int lo = 0, hi = nseq - 1;                        // the sequence of this item: last s with items[s] <= item
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (items[mid] <= item) lo = mid;
      else hi = mid - 1;
    }
    const int seq = lo;
    const int row0 = (item - items[seq]) * RW;
    const int s0 = cu_seqlens[seq], slen = cu_seqlens[seq + 1] - s0;
    const int32_t* ro = row_off + (long)seq * (max_rows + 1);
    const int32_t* lv = list_v + (long)seq * V;
    const float* lc = list_c + (long)seq * V;
    f32x4 acc[RW][NV];
    int cur[RW], end[RW], cbase[RW];                  // wave-uniform: next entry, end of the bucket, first entry of the chunk
    int cv[RW];                                       // chunk registers: entry cbase + lane
    float cc[RW];
#pragma unroll
    for (int j = 0; j < RW; ++j) {
#pragma unroll
      for (int i = 0; i < NV; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const int row = row0 + j;
      const bool live = row < slen && row < max_rows;
      cur[j] = live ? ro[row] : 0;
      end[j] = live ? ro[row + 1] : 0;
      cbase[j] = cur[j];
      const int e = cbase[j] + lane;
      cv[j] = e < end[j] ? (NTL ? __builtin_nontemporal_load(lv + e) : lv[e]) : 0x7FFFFFFF;
      cc[j] = e < end[j] ? (NTL ? __builtin_nontemporal_load(lc + e) : lc[e]) : 0.f;
    }
#pragma unroll 1
    for (int p = 0; p < npanel; ++p) {
      const int vb = p + 1 == npanel ? 0x7FFFFFFF : (p + 1) * pw;
#pragma unroll
      for (int j = 0; j < RW; ++j) {
        while (cur[j] < end[j]) {                     // wave-uniform
          const int first = cur[j] - cbase[j];
          const int n = __popcll(__ballot(lane >= first && cv[j] < vb));   // sorted: a run starting at `first`
          for (int u0 = 0; u0 < n; u0 += GP) {
            float cb[GP];
            bf16x4 wv[GP][NV];
#pragma unroll
            for (int u = 0; u < GP; ++u) {
              cb[u] = 0.f;
#pragma unroll
              for (int i = 0; i < NV; ++i) wv[u][i] = (bf16x4){0, 0, 0, 0};
              if (u0 + u < n) {                       // wave-uniform
                const int src = first + u0 + u;
                cb[u] = __shfl(cc[j], src, 64);
                const bf16_t* wrow = W + (long)__shfl(cv[j], src, 64) * H;
#pragma unroll
                for (int i = 0; i < NV; ++i) wv[u][i] = *(const bf16x4*)(wrow + (i * 64 + lane) * 4);
              }
            }
#pragma unroll
            for (int u = 0; u < GP; ++u)
#pragma unroll
              for (int i = 0; i < NV; ++i) {
                acc[j][i][0] += cb[u] * bf2f(wv[u][i][0]);
                acc[j][i][1] += cb[u] * bf2f(wv[u][i][1]);
                acc[j][i][2] += cb[u] * bf2f(wv[u][i][2]);
                acc[j][i][3] += cb[u] * bf2f(wv[u][i][3]);
              }
          }
          cur[j] += n;
          if (first + n < 64) break;                  // the chunk's next entry belongs to a later panel (or the bucket ended)
          cbase[j] = cur[j];                          // chunk used up: the next 64 entries
          const int e = cbase[j] + lane;
          cv[j] = e < end[j] ? (NTL ? __builtin_nontemporal_load(lv + e) : lv[e]) : 0x7FFFFFFF;
          cc[j] = e < end[j] ? (NTL ? __builtin_nontemporal_load(lc + e) : lc[e]) : 0.f;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < RW; ++j) {
      const int row = row0 + j;
      if (row < slen) {
        bf16_t* out = dHd + (long)(s0 + row) * H;
#pragma unroll
        for (int i = 0; i < NV; ++i)
          *(bf16x4*)(out + (i * 64 + lane) * 4) =
              (bf16x4){f2bf(acc[j][i][0]), f2bf(acc[j][i][1]), f2bf(acc[j][i][2]), f2bf(acc[j][i][3])};
      }
    }
  }
}

extern "C" size_t snx_splade_bwd_scratch_bytes(int32_t nseq, int32_t max_seqlen, int32_t V) {
  return (size_t)nseq * V * 8 + (size_t)nseq * (max_seqlen + 1) * 4 + 256 + ((size_t)nseq + 2) * 4 + 256;
}

extern "C" int snx_splade_bwd(const float* g, const uint32_t* keys, const void* Hd, const void* W,
                              const int32_t* cu_seqlens, void* dHd, float* gradE, float* gradb, void* scratch,
                              int32_t T, int32_t nseq, int32_t max_seqlen, int32_t V, int32_t H, hipStream_t st) {
  if (!g || !keys || !Hd || !W || !cu_seqlens || !dHd || !gradE || !gradb || !scratch) return SNX_E_ARG;
  if (T <= 0 || nseq <= 0 || V <= 0 || max_seqlen <= 0 || H <= 0 || (H % 256) || H > 1024) return SNX_E_SHAPE;
  const int blocks = cdiv(V, 32);               // 4 waves x 8 vocab rows per workgroup
  // "splade_dw_last" = 1 (round 6): the weight half (dE, db) AFTER the activation half.  The dHd gather below is bound by
  // how fast W_E rows (77 MB) come out of the Infinity Cache, where the decoder forward has just left them; run first, the
  // weight half's read-modify-write of the 153 MB gradient matrix pushes them out.  (Then also with non-temporal accesses
  // to the gradient rows: nobody reads them before the end of the backward.)
  const bool dw_last = g_snx_cfg.splade_dw_last != 0;
  auto launch_dw = [&]() {
    if (dw_last) {
      switch (H / 256) {
        case 1: hipLaunchKernelGGL((splade_bwd_dw_kernel<1, true>), dim3(blocks), dim3(256), 0, st, g, keys, (const bf16_t*)Hd, cu_seqlens, gradE, gradb, nseq, V, H); break;
        case 2: hipLaunchKernelGGL((splade_bwd_dw_kernel<2, true>), dim3(blocks), dim3(256), 0, st, g, keys, (const bf16_t*)Hd, cu_seqlens, gradE, gradb, nseq, V, H); break;
        case 3: hipLaunchKernelGGL((splade_bwd_dw_kernel<3, true>), dim3(blocks), dim3(256), 0, st, g, keys, (const bf16_t*)Hd, cu_seqlens, gradE, gradb, nseq, V, H); break;
        default: hipLaunchKernelGGL((splade_bwd_dw_kernel<4, true>), dim3(blocks), dim3(256), 0, st, g, keys, (const bf16_t*)Hd, cu_seqlens, gradE, gradb, nseq, V, H); break;
      }
    } else {
      switch (H / 256) {
        case 1: hipLaunchKernelGGL(splade_bwd_dw_kernel<1>, dim3(blocks), dim3(256), 0, st, g, keys, (const bf16_t*)Hd, cu_seqlens, gradE, gradb, nseq, V, H); break;
        case 2: hipLaunchKernelGGL(splade_bwd_dw_kernel<2>, dim3(blocks), dim3(256), 0, st, g, keys, (const bf16_t*)Hd, cu_seqlens, gradE, gradb, nseq, V, H); break;
        case 3: hipLaunchKernelGGL(splade_bwd_dw_kernel<3>, dim3(blocks), dim3(256), 0, st, g, keys, (const bf16_t*)Hd, cu_seqlens, gradE, gradb, nseq, V, H); break;
        default: hipLaunchKernelGGL(splade_bwd_dw_kernel<4>, dim3(blocks), dim3(256), 0, st, g, keys, (const bf16_t*)Hd, cu_seqlens, gradE, gradb, nseq, V, H); break;
      }
    }
  };
  if (!dw_last) {
    launch_dw();
    SNX_CHECK_LAUNCH();
  }
  // dHd: bucket by row, then one wave per token row
  if (max_seqlen > 8192) return SNX_E_SHAPE;
  char* sc = (char*)scratch;
  int32_t* list_v = (int32_t*)sc;
  float* list_c = (float*)(sc + (size_t)nseq * V * 4);
  int32_t* row_off = (int32_t*)(sc + (size_t)nseq * V * 8);
  const bool det = max_seqlen <= 1024;         // per-wave slot tables fit in LDS: vocabulary-ordered buckets
  const size_t lds = (size_t)(3 * max_seqlen + 1 + (det ? 16 * max_seqlen : 0)) * 4;
  if (det) {
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)splade_bucket_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(splade_bucket_kernel<true>, dim3(nseq), dim3(1024), lds, st, g, keys, cu_seqlens, list_v, list_c,
                       row_off, V, max_seqlen);
  } else {
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)splade_bucket_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(splade_bucket_kernel<false>, dim3(nseq), dim3(1024), lds, st, g, keys, cu_seqlens, list_v, list_c,
                       row_off, V, max_seqlen);
  }
  SNX_CHECK_LAUNCH();
  const int panels = g_snx_cfg.splade_dh_panels;       // number of vocabulary panels; 0: the wave-per-row gather (A/B, tests)
  if (det && panels > 0) {
    int32_t* items = (int32_t*)(sc + (((size_t)nseq * V * 8 + (size_t)nseq * (max_seqlen + 1) * 4 + 255) & ~(size_t)255));
    hipLaunchKernelGGL(splade_dh_items_kernel, dim3(1), dim3(256), 0, st, cu_seqlens, items, nseq, max_seqlen);
    SNX_CHECK_LAUNCH();
    const dim3 pgrid(DH_WGS);
    const bool ntl = g_snx_cfg.splade_dw_last >= 2;
#define SNX_DH_PANELS(NVV)                                                                                                  \
  do {                                                                                                                      \
    if (ntl)                                                                                                                \
      hipLaunchKernelGGL((splade_bwd_dh_panels_kernel<NVV, true>), pgrid, dim3(256), 0, st, list_v, list_c, row_off, items,  \
                         (const bf16_t*)W, cu_seqlens, (bf16_t*)dHd, V, H, max_seqlen, nseq, panels, items + nseq + 1);      \
    else                                                                                                                    \
      hipLaunchKernelGGL((splade_bwd_dh_panels_kernel<NVV, false>), pgrid, dim3(256), 0, st, list_v, list_c, row_off, items, \
                         (const bf16_t*)W, cu_seqlens, (bf16_t*)dHd, V, H, max_seqlen, nseq, panels, items + nseq + 1);      \
  } while (0)
    switch (H / 256) {
      case 1: SNX_DH_PANELS(1); break;
      case 2: SNX_DH_PANELS(2); break;
      case 3: SNX_DH_PANELS(3); break;
      default: SNX_DH_PANELS(4); break;
    }
#undef SNX_DH_PANELS
    SNX_CHECK_LAUNCH();
    if (dw_last) {
      launch_dw();
      SNX_CHECK_LAUNCH();
    }
    return SNX_OK;
  }
  const dim3 grid(cdiv(max_seqlen, 4), nseq);
  switch (H / 256) {
    case 1: hipLaunchKernelGGL(splade_bwd_dh_rows_kernel<1>, grid, dim3(256), 0, st, list_v, list_c, row_off, (const bf16_t*)W, cu_seqlens, (bf16_t*)dHd, V, H, max_seqlen); break;
    case 2: hipLaunchKernelGGL(splade_bwd_dh_rows_kernel<2>, grid, dim3(256), 0, st, list_v, list_c, row_off, (const bf16_t*)W, cu_seqlens, (bf16_t*)dHd, V, H, max_seqlen); break;
    case 3: hipLaunchKernelGGL(splade_bwd_dh_rows_kernel<3>, grid, dim3(256), 0, st, list_v, list_c, row_off, (const bf16_t*)W, cu_seqlens, (bf16_t*)dHd, V, H, max_seqlen); break;
    default: hipLaunchKernelGGL(splade_bwd_dh_rows_kernel<4>, grid, dim3(256), 0, st, list_v, list_c, row_off, (const bf16_t*)W, cu_seqlens, (bf16_t*)dHd, V, H, max_seqlen); break;
  }
  SNX_CHECK_LAUNCH();
  if (dw_last) {
    launch_dw();
    SNX_CHECK_LAUNCH();
  }
  return SNX_OK;
}
