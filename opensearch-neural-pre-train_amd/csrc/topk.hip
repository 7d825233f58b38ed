// Device-side post-processing of the inference encoder (ref:benchmark/encoders.py:309-345,
// NeuralSparseEncoderV33._encode_batch): per row of sparse_repr [B,V]
//   keep v with rep > 0 and allowed[v] (not a special id, token text non-empty and not "[..."/"<...");
//   if top_k is given and more than top_k survive: the top_k largest, ordered by weight descending
//   (ties: lowest vocab id first -- Python's stable sort over the id-ordered dict);
//   otherwise: all survivors in vocab-id order.
// The reference copies every row to the host and loops over its non-zeros in Python; here one
// workgroup per row does a radix select (3 LDS histogram passes over the positive-float bit patterns),
// an id-ordered ballot compaction and, for the top-k case, a bitonic sort of <= 16384 packed
// (value bits << 32 | ~id) keys in LDS.  The row (200 KB) is re-read from L2; HBM-bound and tiny next
// to the encoder forward.
#include "common.h"
#include "snx.h"

namespace {

constexpr int TK_THREADS = 1024;
constexpr int TK_KMAX = 16384;

__device__ __forceinline__ uint32_t tk_key(const float* __restrict__ rep, const uint8_t* __restrict__ allowed, int i) {
  const float x = rep[i];
  return (x > 0.f && allowed[i]) ? __builtin_bit_cast(uint32_t, x) : 0u;
}

__global__ __launch_bounds__(TK_THREADS) void sparse_topk_kernel(const float* __restrict__ rep_all,
                                                                  const uint8_t* __restrict__ allowed, int V, int k,
                                                                  int cap, float* __restrict__ out_val,
                                                                  int32_t* __restrict__ out_idx,
                                                                  int32_t* __restrict__ out_cnt,
                                                                  int32_t* __restrict__ out_sorted) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long sbuf[];   // [pow2 >= k] sort keys (top-k case)
  __shared__ uint32_t hist[2048];
  __shared__ int wcnt[2][16];
  __shared__ int sh[6];                  // 0: n_pos, 1: prefix, 2: remaining, 3: gt_base, 4: eq_base
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* rep = rep_all + (long)row * V;

  // ---- how many entries survive the filter ----
  if (tid < 6) sh[tid] = 0;
  __syncthreads();
  int local = 0;
  for (int i = tid; i < V; i += TK_THREADS) local += tk_key(rep, allowed, i) != 0u;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o, 64);
  if (lane == 0 && local) atomicAdd(&sh[0], local);
  __syncthreads();
  const int n_pos = sh[0];
  const bool topk = k > 0 && n_pos > k;
  uint32_t thr = 0u;
  int need_eq = 0, n_gt = n_pos;
  if (topk) {
    // ---- radix select of the k-th largest key: 11 + 11 + 10 bits, most significant first ----
    uint32_t prefix = 0u, known = 0u;
    int remaining = k;
    const int shifts[3] = {21, 10, 0}, widths[3] = {11, 11, 10};
    for (int p = 0; p < 3; ++p) {
      const int shift = shifts[p];
      const uint32_t bm = (1u << widths[p]) - 1u;
      for (int i = tid; i < 2048; i += TK_THREADS) hist[i] = 0u;
      __syncthreads();
      for (int i = tid; i < V; i += TK_THREADS) {
        const uint32_t key = tk_key(rep, allowed, i);
        if (key != 0u && (key & known) == prefix) atomicAdd(&hist[(key >> shift) & bm], 1u);
      }
      __syncthreads();
      if (tid == 0) {                                   // walk the bins from the top
        int rem = remaining, b = (int)bm;
        for (; b > 0; --b) {
          const int c = (int)hist[b];
          if (c >= rem) break;
          rem -= c;
        }
        sh[1] = b;
        sh[2] = rem;
      }
      __syncthreads();
      prefix |= (uint32_t)sh[1] << shift;
      known |= bm << shift;
      remaining = sh[2];
      __syncthreads();
    }
    thr = prefix;                                       // the k-th largest key; `remaining` entries equal to it are taken
    need_eq = remaining;
    n_gt = k - need_eq;
  }
  const int nsel = topk ? k : n_pos;
  int P = 1;
  while (P < nsel) P <<= 1;
  if (topk)
    for (int i = tid; i < P; i += TK_THREADS) sbuf[i] = 0ull;
  __syncthreads();

  // ---- id-ordered compaction (1024 ids per step; ballots inside a wave, LDS across waves) ----
  float* ov = out_val + (long)row * cap;
  int32_t* oi = out_idx + (long)row * cap;
  for (int base = 0; base < V; base += TK_THREADS) {
    const int i = base + tid;
    const uint32_t key = i < V ? tk_key(rep, allowed, i) : 0u;
    const bool is_gt = key > thr;
    const bool is_eq = topk && key == thr;
    const unsigned long long mg = __ballot(is_gt), me = __ballot(is_eq);
    const unsigned long long below = (1ull << lane) - 1ull;
    if (lane == 0) { wcnt[0][wave] = __popcll(mg); wcnt[1][wave] = __popcll(me); }
    __syncthreads();
    int goff = sh[3], eoff = sh[4];
    for (int w = 0; w < wave; ++w) { goff += wcnt[0][w]; eoff += wcnt[1][w]; }
    int slot = -1;
    if (is_gt) slot = goff + __popcll(mg & below);
    else if (is_eq) {
      const int e = eoff + __popcll(me & below);
      if (e < need_eq) slot = n_gt + e;
    }
    if (slot >= 0) {
      if (topk) sbuf[slot] = ((unsigned long long)key << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)i);
      else if (slot < cap) { ov[slot] = __builtin_bit_cast(float, key); oi[slot] = i; }
    }
    __syncthreads();
    if (tid == 0) {
      int a = 0, b = 0;
      for (int w = 0; w < 16; ++w) { a += wcnt[0][w]; b += wcnt[1][w]; }
      sh[3] += a;
      sh[4] += b;
    }
    __syncthreads();
  }

  if (topk) {
    // ---- bitonic sort, descending, of P packed keys ----
    for (int size = 2; size <= P; size <<= 1)
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
        for (int t = tid; t < (P >> 1); t += TK_THREADS) {
          const int lo = 2 * t - (t & (stride - 1));     // index with the `stride` bit clear
          const int hi = lo + stride;
          const bool desc = (lo & size) == 0;
          const unsigned long long a = sbuf[lo], b = sbuf[hi];
          if ((a < b) == desc) { sbuf[lo] = b; sbuf[hi] = a; }
        }
        __syncthreads();
      }
    for (int i = tid; i < nsel; i += TK_THREADS) {
      const unsigned long long e = sbuf[i];
      ov[i] = __builtin_bit_cast(float, (uint32_t)(e >> 32));
      oi[i] = (int32_t)(0xFFFFFFFFu - (uint32_t)(e & 0xFFFFFFFFull));
    }
  }
  if (tid == 0) { out_cnt[row] = nsel < cap ? nsel : cap; out_sorted[row] = topk ? 1 : 0; }
}

}  // namespace

extern "C" int snx_sparse_topk(const float* rep, const uint8_t* allowed, float* out_val, int32_t* out_idx,
                               int32_t* out_cnt, int32_t* out_sorted, int32_t B, int32_t V, int32_t k, int32_t cap,
                               hipStream_t st) {
  if (!rep || !allowed || !out_val || !out_idx || !out_cnt || !out_sorted) return SNX_E_ARG;
  if (B <= 0 || V <= 0 || cap <= 0 || k > TK_KMAX) return SNX_E_SHAPE;
  if (k > 0 && cap < (k < V ? k : V)) return SNX_E_SHAPE;      // room for every row's selection
  if (k <= 0 && cap < V) return SNX_E_SHAPE;
  int P = 1;
  while (P < (k > 0 ? k : 1)) P <<= 1;
  const size_t lds = (size_t)P * 8;
  if (lds > 48 * 1024) {
    static LdsOptIn optin;
    if (const int rc = optin.ensure((const void*)sparse_topk_kernel, TK_KMAX * 8)) return rc;
  }
  hipLaunchKernelGGL(sparse_topk_kernel, dim3(B), dim3(TK_THREADS), lds, st, rep, allowed, V, k, cap, out_val,
                     out_idx, out_cnt, out_sorted);
  SNX_CHECK_LAUNCH();
  return SNX_OK;
}
