// Bidirectional attention for ModernBERT (K5g global / K5l sliding-window):
// softmax(q k^T / sqrt(64) + mask) v with key-padding mask and, on local layers, the inclusive
// band |i - j| <= window.  Replaces torch SDPA as called from transformers
// modeling_modernbert.py:286-297 (masks: masking_utils.py:141-150).
//
// Layout: qkv [T, 3, heads, 64] bf16 (RoPE already applied to q,k), sequences are row ranges
// cu_seqlens[s] .. cu_seqlens[s+1]; out [T, heads*64] bf16; lse [heads, T] fp32.
//
// Kernel shape (CDNA4): one workgroup = 4 waves = 64 query rows of one (sequence, head); K/V
// stream through LDS in 64-key tiles.  The score tile is computed TRANSPOSED (S^T = K Q^T) so
// each lane owns one query column: the online-softmax statistics are lane-local plus two
// xor-shuffles, and the fp32 score accumulators, converted to bf16, ARE the B operand of the
// second product O^T = V^T P^T (no LDS round trip for P).  V^T fragments come from the
// row-major V tile with ds_read_b64_tr_b16 (hardware transpose).  The contraction index of the
// second product is a permutation of the key index (lane group g, element j <-> key
// 16*(2c + j/4) + 4g + j%4); both operands use the same permutation.
#include "attention_common.h"
#include "config.h"
#include "snx.h"

// Block schedule (speed only).  Blocks b and b+8 share an XCD and its L2.  The unit of placement is
// one (sequence, head) pair: all its 64-row tiles go to ONE XCD (K/V, or Q/dO in the dK/dV pass, are
// then fetched from HBM once instead of once per tile: ~2x the algorithmic bytes at ~5 TB/s of fabric
// traffic before), and consecutive units are dealt round-robin over the XCDs.  Sequence GROUPS (the
// 64-token queries and 256-token documents of a fused pass) each get the tile count their own
// max length needs: a workgroup that only finds "tile beyond my sequence" still has to be launched
// behind the resident ones, and 2304 of those cost ~15 us per backward kernel.  Longest group first.
#define SNX_ATTN_MAX_GROUPS 8
struct AttnSched {
  int n;
  int seq0[SNX_ATTN_MAX_GROUPS], units[SNX_ATTN_MAX_GROUPS], ntile[SNX_ATTN_MAX_GROUPS];
  int kend[SNX_ATTN_MAX_GROUPS];      // exclusive prefix end of the group in per-XCD block index space
};
struct AttnBlock { int tile, head, seq; bool live; };
__device__ __forceinline__ AttnBlock attn_block(const AttnSched& sc, int heads) {
  const int xcd = blockIdx.x & 7;
  int k = blockIdx.x >> 3;
  int g = 0, k0 = 0;
#pragma unroll
  for (int i = 0; i < SNX_ATTN_MAX_GROUPS - 1; ++i)
    if (i + 1 < sc.n && k >= sc.kend[i]) { g = i + 1; k0 = sc.kend[i]; }
  k -= k0;
  const int nt = sc.ntile[g];
  const int unit = (k / nt) * 8 + xcd;
  AttnBlock b;
  b.tile = k % nt;
  b.head = unit % heads;
  b.seq = sc.seq0[g] + unit / heads;
  b.live = unit < sc.units[g];
  return b;
}

// groups = {n, (seq_begin, nseq, max_len) x n} on the host, or NULL for one group of nseq sequences.
static int attn_sched_build(AttnSched& sc, long& grid, const int32_t* groups, int nseq, int max_seqlen, int heads) {
  const int32_t one[4] = {1, 0, nseq, max_seqlen};
  const int32_t* g = groups ? groups : one;
  if (g[0] < 1 || g[0] > SNX_ATTN_MAX_GROUPS) return SNX_E_ARG;
  int order[SNX_ATTN_MAX_GROUPS];
  int covered = 0;
  for (int i = 0; i < g[0]; ++i) {
    order[i] = i;
    if (g[1 + 3 * i] != covered || g[2 + 3 * i] <= 0 || g[3 + 3 * i] <= 0 || g[3 + 3 * i] > max_seqlen) return SNX_E_ARG;
    covered += g[2 + 3 * i];
  }
  if (covered != nseq) return SNX_E_ARG;
  for (int i = 1; i < g[0]; ++i)                      // insertion sort, longest max_len first
    for (int j = i; j > 0 && g[3 + 3 * order[j]] > g[3 + 3 * order[j - 1]]; --j) { int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t; }
  sc.n = g[0];
  long k = 0;
  for (int i = 0; i < SNX_ATTN_MAX_GROUPS; ++i) {
    if (i < g[0]) {
      const int32_t* e = g + 1 + 3 * order[i];
      sc.seq0[i] = e[0]; sc.units[i] = e[1] * heads; sc.ntile[i] = cdiv(e[2], 64);
      k += (long)cdiv(sc.units[i], 8) * sc.ntile[i];
    } else { sc.seq0[i] = 0; sc.units[i] = 0; sc.ntile[i] = 1; }
    if (k > 0x0fffffffL) return SNX_E_SHAPE;
    sc.kend[i] = (int)k;
  }
  grid = 8 * k;
  return SNX_OK;
}

// Register-staged tile (async-STAGE split): the global loads of tile t+1 are issued right after
// tile t has been written to LDS and stay in flight while tile t is multiplied; they are written
// to LDS after the next barrier.  Each thread carries 2 x 16 B of the 64x64 bf16 tile.
struct TileRegs { bf16x8 v[2]; };
__device__ __forceinline__ void tile_load(TileRegs& t, const bf16_t* __restrict__ base, long row_stride, int r0g,
                                          int rmax) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int id = threadIdx.x + i * 256;
    const int r = id >> 3, c = id & 7;
    int gr = r0g + r;
    gr = gr < rmax ? gr : rmax - 1;
    t.v[i] = *(const bf16x8*)(base + (long)gr * row_stride + c * 8);
  }
}
// write to a row image (k_off swizzle, ds_read_b128 rows) and/or a transposed-read image (v_off)
__device__ __forceinline__ void tile_store(const TileRegs& t, char* lds_row, char* lds_tr) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int id = threadIdx.x + i * 256;
    const int r = id >> 3, c = id & 7;
    if (lds_row) *(bf16x8*)(lds_row + k_off(r, c)) = t.v[i];
    if (lds_tr) *(bf16x8*)(lds_tr + v_off(r, c)) = t.v[i];
  }
}

// Stage one 64-row x 64-dim bf16 tile (rows key0.. of the K or V third) into LDS, swizzled.
template <bool IS_V>
__device__ __forceinline__ void stage_kv(const bf16_t* __restrict__ base, long row_stride, int key0, int slen,
                                         char* lds) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int id = threadIdx.x + i * 256;
    const int r = id >> 3, c = id & 7;
    int key = key0 + r;
    key = key < slen ? key : slen - 1;
    const bf16x8 v = *(const bf16x8*)(base + (long)key * row_stride + c * 8);
    *(bf16x8*)(lds + (IS_V ? v_off(r, c) : k_off(r, c))) = v;
  }
}

__global__ __launch_bounds__(256) void attn_fwd_kernel(const bf16_t* __restrict__ qkv,
                                                       const int32_t* __restrict__ cu_seqlens,
                                                       const int64_t* __restrict__ mask, bf16_t* __restrict__ out,
                                                       float* __restrict__ lse, int T, int heads, int window,
                                                       float scale, const AttnSched sched) {
  __shared__ __attribute__((aligned(16))) char sK[64 * 128];
  __shared__ __attribute__((aligned(16))) char sV[64 * 128];
  __shared__ __attribute__((aligned(16))) unsigned char sValid[64];
  __shared__ int sAllValid;
  const AttnBlock blk = attn_block(sched, heads);
  if (!blk.live) return;
  const int seq = blk.seq, head = blk.head;
  const int s0 = cu_seqlens[seq], slen = cu_seqlens[seq + 1] - s0;
  const int q0 = blk.tile * 64;
  if (q0 >= slen) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, li = lane & 15;
  const long rs = 3L * heads * 64;                  // token stride in qkv
  const int H = heads * 64;
  const bf16_t* qbase = qkv + (long)s0 * rs + head * 64;
  const bf16_t* kbase = qbase + H;
  const bf16_t* vbase = qbase + 2 * H;

  const int qpos = q0 + wave * 16 + li;             // this lane's query row (within the sequence)
  const int qrow = qpos < slen ? qpos : slen - 1;
  bf16x8 qf[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) qf[c] = *(const bf16x8*)(qbase + (long)qrow * rs + c * 32 + g * 8);

  int j_lo = 0, j_hi = (slen - 1) >> 6;
  if (window >= 0) {
    const int lo = q0 - window, hi = q0 + 63 + window;
    j_lo = lo > 0 ? (lo >> 6) : 0;
    const int jh = (hi < slen - 1 ? hi : slen - 1) >> 6;
    j_hi = jh;
  }

  f32x4 o[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) o[d] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m_run = NEG_BIG, l_run = 0.f;

  TileRegs rk, rv;
  tile_load(rk, kbase, rs, j_lo * 64, slen);
  tile_load(rv, vbase, rs, j_lo * 64, slen);
  for (int j = j_lo; j <= j_hi; ++j) {
    const int key0 = j * 64;
    __syncthreads();                                 // previous tile fully consumed
    tile_store(rk, sK, nullptr);
    tile_store(rv, nullptr, sV);
    if (j < j_hi) {                                  // next tile's loads fly under this tile's math
      tile_load(rk, kbase, rs, key0 + 64, slen);
      tile_load(rv, vbase, rs, key0 + 64, slen);
    }
    if (threadIdx.x < 64) {
      const int key = key0 + threadIdx.x;
      const bool v = key < slen && mask[s0 + key] != 0;
      sValid[threadIdx.x] = v ? 1 : 0;
      const unsigned long long all = __ballot(v);
      if (threadIdx.x == 0) sAllValid = (all == ~0ull) ? 1 : 0;
    }
    __syncthreads();

    // S^T tile: 64 keys x 16 queries per wave
    f32x4 s[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      s[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const bf16x8 kf = *(const bf16x8*)(sK + k_off(kt * 16 + li, 4 * c + g));
        s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[c], s[kt], 0, 0, 0);
      }
    }
    // scores in the log2 domain: t = s * (scale * log2 e); masked entries -> NEG_BIG
    const float c2 = scale * LOG2E;
    const bool clean = sAllValid && band_clean(window, q0 + wave * 16, q0 + wave * 16 + 15, key0, key0 + 63);
    float mx = NEG_BIG;
    if (clean) {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          s[kt][r] *= c2;
          mx = fmaxf(mx, s[kt][r]);
        }
    } else {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        const uint32_t vm = *(const uint32_t*)(sValid + kt * 16 + g * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = key0 + kt * 16 + g * 4 + r;
          bool ok = (vm >> (8 * r)) & 1;
          if (window >= 0) {
            const int dlt = qpos - key;
            ok = ok && (dlt <= window) && (dlt >= -window);
          }
          const float v = ok ? s[kt][r] * c2 : NEG_BIG;
          s[kt][r] = v;
          mx = fmaxf(mx, v);
        }
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = fast_exp2(m_run - m_new);
    float rsum = 0.f;
    bf16x8 pb[2];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = fast_exp2(s[kt][r] - m_new);
        rsum += p;
        pb[kt >> 1][(kt & 1) * 4 + r] = f2bf(p);
      }
    rsum += __shfl_xor(rsum, 16, 64);
    rsum += __shfl_xor(rsum, 32, 64);
    l_run = l_run * alpha + rsum;
    m_run = m_new;
#pragma unroll
    for (int d = 0; d < 4; ++d) o[d] *= alpha;

    // O^T += V^T P^T
    const int tq = li >> 2, tp = li & 3;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int r0 = 16 * (2 * c) + 4 * g + tq, r1 = r0 + 16;
        const int chunk = 2 * d + (tp >> 1);
        const bf16x4 a0 = lds_tr16(sV + v_off(r0, chunk) + (tp & 1) * 8);
        const bf16x4 a1 = lds_tr16(sV + v_off(r1, chunk) + (tp & 1) * 8);
        const bf16x8 vf = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        o[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pb[c], o[d], 0, 0, 0);
      }
    }
  }

  if (qpos < slen) {
    const float inv = 1.0f / l_run;
    bf16_t* orow = out + (long)(s0 + qpos) * H + head * 64 + g * 4;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      bf16x4 v = {f2bf(o[d][0] * inv), f2bf(o[d][1] * inv), f2bf(o[d][2] * inv), f2bf(o[d][3] * inv)};
      *(bf16x4*)(orow + d * 16) = v;
    }
    if (g == 0) lse[(long)head * T + s0 + qpos] = (m_run + __log2f(l_run)) * LN2;   // natural-log LSE
  }
}

// sequence-resident fast path (attention_unit.hip): one launch for all groups whose sequences fit 256 tokens
int attn_unit_fwd(const bf16_t* qkv, const int32_t* cu_seqlens, const int64_t* mask, bf16_t* out, float* lse, int T,
                  int heads, int window, const int32_t* groups, hipStream_t st);
int attn_unit_bwd(const bf16_t* qkv, const bf16_t* out, const bf16_t* dout, const float* lse, float* delta,
                  const int32_t* cu_seqlens, const int64_t* mask, bf16_t* dqkv, const f32x2* rope_tab,
                  const int32_t* pos, int T, int heads, int window, const int32_t* groups, hipStream_t st);

static bool attn_streaming_only() { return g_snx_cfg.attn_streaming != 0; }   // A/B switch: force the tiled kernels

// Validates the group table, then calls resident(sub-table of the groups of <= 256 tokens) once and
// streaming(seq_begin, nseq, max_len) for every longer group.
template <typename FR, typename FS>
static int route_groups(const int32_t* groups, int nseq, int max_seqlen, FR resident, FS streaming) {
  const int32_t one[4] = {1, 0, nseq, max_seqlen};
  const int32_t* g = groups ? groups : one;
  if (g[0] < 1 || g[0] > SNX_ATTN_MAX_GROUPS) return SNX_E_ARG;
  int covered = 0;
  for (int i = 0; i < g[0]; ++i) {
    if (g[1 + 3 * i] != covered || g[2 + 3 * i] <= 0 || g[3 + 3 * i] <= 0 || g[3 + 3 * i] > max_seqlen) return SNX_E_ARG;
    covered += g[2 + 3 * i];
  }
  if (covered != nseq) return SNX_E_ARG;
  int32_t sub[1 + 3 * SNX_ATTN_MAX_GROUPS];
  sub[0] = 0;
  for (int i = 0; i < g[0]; ++i) {
    const int32_t* e = g + 1 + 3 * i;
    if (e[2] <= 256 && !attn_streaming_only()) {
      int32_t* d = sub + 1 + 3 * sub[0]++;
      d[0] = e[0]; d[1] = e[1]; d[2] = e[2];
    } else {
      const int rc = streaming(e[0], e[1], e[2]);
      if (rc != SNX_OK) return rc;
    }
  }
  return sub[0] ? resident(sub) : SNX_OK;
}

extern "C" int snx_attn_fwd_ex(const void* qkv, const int32_t* cu_seqlens, const int64_t* mask, void* out, float* lse,
                               const int32_t* groups, int32_t T, int32_t nseq, int32_t max_seqlen, int32_t heads,
                               int32_t head_dim, int32_t window, hipStream_t st) {
  if (!qkv || !cu_seqlens || !mask || !out || !lse || T <= 0 || nseq <= 0 || max_seqlen <= 0) return SNX_E_ARG;
  if (head_dim != 64 || heads <= 0) return SNX_E_SHAPE;
  return route_groups(
      groups, nseq, max_seqlen,
      [&](const int32_t* sub) -> int {
        return attn_unit_fwd((const bf16_t*)qkv, cu_seqlens, mask, (bf16_t*)out, lse, T, heads, window, sub, st);
      },
      [&](int seq0, int ns, int max_len) -> int {
        const int32_t one[4] = {1, 0, ns, max_len};
        AttnSched sc;
        long grid;
        const int rc = attn_sched_build(sc, grid, one, ns, max_len, heads);
        if (rc != SNX_OK) return rc;
        sc.seq0[0] = seq0;
        hipLaunchKernelGGL(attn_fwd_kernel, dim3((unsigned)grid), dim3(256), 0, st, (const bf16_t*)qkv, cu_seqlens,
                           mask, (bf16_t*)out, lse, T, heads, window, 0.125f, sc);
        SNX_CHECK_LAUNCH();
        return SNX_OK;
      });
}

extern "C" int snx_attn_fwd(const void* qkv, const int32_t* cu_seqlens, const int64_t* mask, void* out, float* lse,
                            int32_t T, int32_t nseq, int32_t max_seqlen, int32_t heads, int32_t head_dim,
                            int32_t window, hipStream_t st) {
  return snx_attn_fwd_ex(qkv, cu_seqlens, mask, out, lse, nullptr, T, nseq, max_seqlen, heads, head_dim, window, st);
}

// ==========================================================================================
// Backward.  P is recomputed from q, k and the forward's log-sum-exp; two passes so that no
// gradient needs a cross-workgroup sum (deterministic, no atomics):
//   attn_bwd_dq_kernel : one workgroup per 64 query rows, loops over key tiles     -> dQ
//   attn_bwd_dkv_kernel: one workgroup per 64 keys,       loops over query tiles   -> dK, dV
// dS = P o (dP - delta), delta_i = sum_d dO_i O_i;  dQ = scale dS K, dK = scale dS^T Q, dV = P^T dO.
// Every tile that is read both row-wise (MFMA A operand along d) and transposed (contraction
// over its row index) is staged twice, once per swizzle (k_off image for ds_read_b128 rows,
// v_off image for ds_read_b64_tr_b16).
// ==========================================================================================
// stage a 64x64 bf16 tile twice: row image (k_off) and transposed-read image (v_off)
__device__ __forceinline__ void stage_dual(const bf16_t* __restrict__ base, long row_stride, int r0g, int rmax,
                                           char* lds_row, char* lds_tr) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int id = threadIdx.x + i * 256;
    const int r = id >> 3, c = id & 7;
    int gr = r0g + r;
    gr = gr < rmax ? gr : rmax - 1;
    const bf16x8 v = *(const bf16x8*)(base + (long)gr * row_stride + c * 8);
    if (lds_row) *(bf16x8*)(lds_row + k_off(r, c)) = v;
    if (lds_tr) *(bf16x8*)(lds_tr + v_off(r, c)) = v;
  }
}

__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(
    const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ out, const bf16_t* __restrict__ dout,
    const float* __restrict__ lse, float* __restrict__ delta, const int32_t* __restrict__ cu_seqlens,
    const int64_t* __restrict__ mask, bf16_t* __restrict__ dqkv, const f32x2* __restrict__ rope_tab,
    const int32_t* __restrict__ pos, int T, int heads, int window, float scale, const AttnSched sched) {
  __shared__ __attribute__((aligned(16))) char sKr[64 * 128];
  __shared__ __attribute__((aligned(16))) char sKt[64 * 128];
  __shared__ __attribute__((aligned(16))) char sVr[64 * 128];
  __shared__ __attribute__((aligned(16))) unsigned char sValid[64];
  __shared__ int sAllValid;
  const AttnBlock blk = attn_block(sched, heads);
  if (!blk.live) return;
  const int seq = blk.seq, head = blk.head;
  const int s0 = cu_seqlens[seq], slen = cu_seqlens[seq + 1] - s0;
  const int q0 = blk.tile * 64;
  if (q0 >= slen) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, li = lane & 15;
  const int H = heads * 64;
  const long rs = 3L * H;
  const bf16_t* qbase = qkv + (long)s0 * rs + head * 64;
  const bf16_t* kbase = qbase + H;
  const bf16_t* vbase = qbase + 2 * H;
  const int qpos = q0 + wave * 16 + li;
  const int qrow = qpos < slen ? qpos : slen - 1;
  bf16x8 qf[2], dof[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    qf[c] = *(const bf16x8*)(qbase + (long)qrow * rs + c * 32 + g * 8);
    dof[c] = *(const bf16x8*)(dout + (long)(s0 + qrow) * H + head * 64 + c * 32 + g * 8);
  }
  const float c2 = scale * LOG2E;
  const float lse2_q = lse[(long)head * T + s0 + qrow] * LOG2E;
  // delta_q = sum_d dO[q,d] * O[q,d]: this lane holds 16 of the 64 d's; the 4 lane groups sum up.
  // Written out for the dK/dV pass that follows on the same stream.
  float dl_q = 0.f;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const bf16x8 of = *(const bf16x8*)(out + (long)(s0 + qrow) * H + head * 64 + c * 32 + g * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) dl_q += bf2f(of[e]) * bf2f(dof[c][e]);
  }
  dl_q += __shfl_xor(dl_q, 16, 64);
  dl_q += __shfl_xor(dl_q, 32, 64);
  if (g == 0 && qpos < slen) delta[(long)head * T + s0 + qpos] = dl_q;

  int j_lo = 0, j_hi = (slen - 1) >> 6;
  if (window >= 0) {
    const int lo = q0 - window, hi = q0 + 63 + window;
    j_lo = lo > 0 ? (lo >> 6) : 0;
    j_hi = (hi < slen - 1 ? hi : slen - 1) >> 6;
  }
  f32x4 dq[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) dq[d] = (f32x4){0.f, 0.f, 0.f, 0.f};

  TileRegs rk, rv;
  tile_load(rk, kbase, rs, j_lo * 64, slen);
  tile_load(rv, vbase, rs, j_lo * 64, slen);
  for (int j = j_lo; j <= j_hi; ++j) {
    const int key0 = j * 64;
    __syncthreads();
    tile_store(rk, sKr, sKt);
    tile_store(rv, sVr, nullptr);
    if (j < j_hi) {
      tile_load(rk, kbase, rs, key0 + 64, slen);
      tile_load(rv, vbase, rs, key0 + 64, slen);
    }
    if (threadIdx.x < 64) {
      const int key = key0 + threadIdx.x;
      const bool v = key < slen && mask[s0 + key] != 0;
      sValid[threadIdx.x] = v ? 1 : 0;
      const unsigned long long all = __ballot(v);
      if (threadIdx.x == 0) sAllValid = (all == ~0ull) ? 1 : 0;
    }
    __syncthreads();
    const bool clean = sAllValid && band_clean(window, q0 + wave * 16, q0 + wave * 16 + 15, key0, key0 + 63);
    bf16x8 dsb[2];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f}, dp = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const bf16x8 kf = *(const bf16x8*)(sKr + k_off(kt * 16 + li, 4 * c + g));
        const bf16x8 vf = *(const bf16x8*)(sVr + k_off(kt * 16 + li, 4 * c + g));
        s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[c], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, dof[c], dp, 0, 0, 0);
      }
      if (clean) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = fast_exp2(fmaf(s[r], c2, -lse2_q));
          dsb[kt >> 1][(kt & 1) * 4 + r] = f2bf(p * (dp[r] - dl_q));
        }
      } else {
        const uint32_t vm = *(const uint32_t*)(sValid + kt * 16 + g * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = key0 + kt * 16 + g * 4 + r;
          bool ok = (vm >> (8 * r)) & 1;
          if (window >= 0) {
            const int dlt = qpos - key;
            ok = ok && (dlt <= window) && (dlt >= -window);
          }
          const float p = ok ? fast_exp2(fmaf(s[r], c2, -lse2_q)) : 0.f;
          dsb[kt >> 1][(kt & 1) * 4 + r] = f2bf(p * (dp[r] - dl_q));
        }
      }
    }
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int c = 0; c < 2; ++c)
        dq[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr(sKt, d, c, lane), dsb[c], dq[d], 0, 0, 0);
  }
  if (qpos < slen) {
    bf16_t* orow = dqkv + (long)(s0 + qpos) * rs + head * 64 + g * 4;
    store_grad_rows(orow, dq, scale, rope_tab, pos ? pos[s0 + qpos] : 0, g);
  }
}

__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(
    const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout, const float* __restrict__ lse,
    const float* __restrict__ delta, const int32_t* __restrict__ cu_seqlens, const int64_t* __restrict__ mask,
    bf16_t* __restrict__ dqkv, const f32x2* __restrict__ rope_tab, const int32_t* __restrict__ pos, int T,
    int heads, int window, float scale, const AttnSched sched) {
  __shared__ __attribute__((aligned(16))) char sQr[64 * 128];
  __shared__ __attribute__((aligned(16))) char sQt[64 * 128];
  __shared__ __attribute__((aligned(16))) char sOr[64 * 128];
  __shared__ __attribute__((aligned(16))) char sOt[64 * 128];
  __shared__ __attribute__((aligned(16))) float sLse[64];
  __shared__ __attribute__((aligned(16))) float sDel[64];
  const AttnBlock blk = attn_block(sched, heads);
  if (!blk.live) return;
  const int seq = blk.seq, head = blk.head;
  const int s0 = cu_seqlens[seq], slen = cu_seqlens[seq + 1] - s0;
  const int key0 = blk.tile * 64;
  if (key0 >= slen) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, li = lane & 15;
  const int H = heads * 64;
  const long rs = 3L * H;
  const bf16_t* qbase = qkv + (long)s0 * rs + head * 64;
  const bf16_t* kbase = qbase + H;
  const bf16_t* vbase = qbase + 2 * H;
  const bf16_t* dobase = dout + (long)s0 * H + head * 64;
  const int kpos = key0 + wave * 16 + li;
  const int krow = kpos < slen ? kpos : slen - 1;
  const bool kvalid = kpos < slen && mask[s0 + krow] != 0;
  const bool wave_keys_valid = __ballot(kvalid) == ~0ull;
  const float c2 = scale * LOG2E;
  bf16x8 kf[2], vf[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    kf[c] = *(const bf16x8*)(kbase + (long)krow * rs + c * 32 + g * 8);
    vf[c] = *(const bf16x8*)(vbase + (long)krow * rs + c * 32 + g * 8);
  }
  int i_lo = 0, i_hi = (slen - 1) >> 6;
  if (window >= 0) {
    const int lo = key0 - window, hi = key0 + 63 + window;
    i_lo = lo > 0 ? (lo >> 6) : 0;
    i_hi = (hi < slen - 1 ? hi : slen - 1) >> 6;
  }
  f32x4 dk[4], dv[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    dk[d] = (f32x4){0.f, 0.f, 0.f, 0.f};
    dv[d] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  TileRegs rq, rdo;
  float r_lse = 0.f, r_del = 0.f;
  auto load_stats = [&](int q0n) {
    if (threadIdx.x < 64) {
      const int q = q0n + threadIdx.x;
      const int qc = q < slen ? q : slen - 1;
      r_lse = lse[(long)head * T + s0 + qc] * LOG2E;                 // log2 domain
      r_del = delta[(long)head * T + s0 + qc];
    }
  };
  tile_load(rq, qbase, rs, i_lo * 64, slen);
  tile_load(rdo, dobase, H, i_lo * 64, slen);
  load_stats(i_lo * 64);
  for (int i = i_lo; i <= i_hi; ++i) {
    const int q0 = i * 64;
    __syncthreads();
    tile_store(rq, sQr, sQt);
    tile_store(rdo, sOr, sOt);
    if (threadIdx.x < 64) {
      sLse[threadIdx.x] = r_lse;
      sDel[threadIdx.x] = r_del;
    }
    if (i < i_hi) {
      tile_load(rq, qbase, rs, q0 + 64, slen);
      tile_load(rdo, dobase, H, q0 + 64, slen);
      load_stats(q0 + 64);
    }
    __syncthreads();
    // no masking needed when all 64 queries exist, this wave's 16 keys are all valid and in band
    const bool clean = (q0 + 63 < slen) && wave_keys_valid &&
                       band_clean(window, q0, q0 + 63, key0 + wave * 16, key0 + wave * 16 + 15);
    bf16x8 pb[2], dsb[2];
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
      f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f}, dp = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const bf16x8 qfr = *(const bf16x8*)(sQr + k_off(qt * 16 + li, 4 * c + g));
        const bf16x8 ofr = *(const bf16x8*)(sOr + k_off(qt * 16 + li, 4 * c + g));
        s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qfr, kf[c], s, 0, 0, 0);     // S[q][key]
        dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ofr, vf[c], dp, 0, 0, 0);   // dP[q][key]
      }
      const f32x4 l4 = *(const f32x4*)(sLse + qt * 16 + g * 4);
      const f32x4 d4 = *(const f32x4*)(sDel + qt * 16 + g * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float p = fast_exp2(fmaf(s[r], c2, -l4[r]));
        if (!clean) {
          const int q = q0 + qt * 16 + g * 4 + r;
          bool ok = kvalid && q < slen;
          if (window >= 0) {
            const int dlt = q - kpos;
            ok = ok && (dlt <= window) && (dlt >= -window);
          }
          p = ok ? p : 0.f;
        }
        pb[qt >> 1][(qt & 1) * 4 + r] = f2bf(p);
        dsb[qt >> 1][(qt & 1) * 4 + r] = f2bf(p * (dp[r] - d4[r]));
      }
    }
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        dv[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr(sOt, d, c, lane), pb[c], dv[d], 0, 0, 0);
        dk[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr(sQt, d, c, lane), dsb[c], dk[d], 0, 0, 0);
      }
  }
  if (kpos < slen) {
    bf16_t* krow_out = dqkv + (long)(s0 + kpos) * rs + H + head * 64 + g * 4;
    bf16_t* vrow_out = krow_out + H;
    store_grad_rows(krow_out, dk, scale, rope_tab, pos ? pos[s0 + kpos] : 0, g);
    store_grad_rows(vrow_out, dv, 1.0f, nullptr, 0, g);
  }
}

extern "C" int snx_attn_bwd_ex(const void* qkv, const void* out, const void* dout, const float* lse,
                               const int32_t* cu_seqlens, const int64_t* mask, float* delta_scratch, void* dqkv,
                               const float* rope_tab, const int32_t* pos, const int32_t* groups, int32_t T,
                               int32_t nseq, int32_t max_seqlen, int32_t heads, int32_t head_dim, int32_t window,
                               hipStream_t st) {
  if ((rope_tab == nullptr) != (pos == nullptr)) return SNX_E_ARG;
  if (!qkv || !out || !dout || !lse || !cu_seqlens || !mask || !delta_scratch || !dqkv) return SNX_E_ARG;
  if (T <= 0 || nseq <= 0 || max_seqlen <= 0 || heads <= 0 || head_dim != 64) return SNX_E_SHAPE;
  return route_groups(
      groups, nseq, max_seqlen,
      [&](const int32_t* sub) -> int {
        return attn_unit_bwd((const bf16_t*)qkv, (const bf16_t*)out, (const bf16_t*)dout, lse, delta_scratch,
                             cu_seqlens, mask, (bf16_t*)dqkv, (const f32x2*)rope_tab, pos, T, heads, window, sub, st);
      },
      [&](int seq0, int ns, int max_len) -> int {
        const int32_t one[4] = {1, 0, ns, max_len};
        AttnSched sc;
        long grid_l;
        const int rc = attn_sched_build(sc, grid_l, one, ns, max_len, heads);
        if (rc != SNX_OK) return rc;
        sc.seq0[0] = seq0;
        const dim3 grid((unsigned)grid_l);
        hipLaunchKernelGGL(attn_bwd_dq_kernel, grid, dim3(256), 0, st, (const bf16_t*)qkv, (const bf16_t*)out,
                           (const bf16_t*)dout, lse, delta_scratch, cu_seqlens, mask, (bf16_t*)dqkv,
                           (const f32x2*)rope_tab, pos, T, heads, window, 0.125f, sc);
        SNX_CHECK_LAUNCH();
        hipLaunchKernelGGL(attn_bwd_dkv_kernel, grid, dim3(256), 0, st, (const bf16_t*)qkv, (const bf16_t*)dout, lse,
                           delta_scratch, cu_seqlens, mask, (bf16_t*)dqkv, (const f32x2*)rope_tab, pos, T, heads,
                           window, 0.125f, sc);
        SNX_CHECK_LAUNCH();
        return SNX_OK;
      });
}

extern "C" int snx_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse,
                            const int32_t* cu_seqlens, const int64_t* mask, float* delta_scratch, void* dqkv,
                            const float* rope_tab, const int32_t* pos, int32_t T, int32_t nseq, int32_t max_seqlen,
                            int32_t heads, int32_t head_dim, int32_t window, hipStream_t st) {
  return snx_attn_bwd_ex(qkv, out, dout, lse, cu_seqlens, mask, delta_scratch, dqkv, rope_tab, pos, nullptr, T, nseq,
                         max_seqlen, heads, head_dim, window, st);
}
