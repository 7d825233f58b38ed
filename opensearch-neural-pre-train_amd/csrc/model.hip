// Native runtime of the SPLADE-ModernBERT encoder: owns the layer loop, the activation arena
// plan and the bf16 weight cache, and drives the HIP kernels of this library on one stream.
// One call = one `SPLADEModernBERT.forward` (ref:src/model/splade_modern.py:50-88 ->
// transformers modeling_modernbert.py:434-478,522-550) or its complete backward.
// No allocation, no host sync, caller-owned buffers, stream-ordered, graph-capturable.
#include <vector>
#include <cstdlib>

#include "common.h"
#include "config.h"
#include "snx.h"

bool snx_dec256_takes(int32_t T);   // splade_head.hip

namespace {

inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

struct Dims {
  int V, H, I, L, heads, T, nseq;
};

// canonical parameter order (== state-dict order with the tied decoder weight listed once)
struct PIdx {
  int L;
  int tok_emb() const { return 0; }
  int emb_norm() const { return 1; }
  int base(int l) const { return l == 0 ? 2 : 7 + 6 * (l - 1); }
  int attn_norm(int l) const { return base(l); }                 // l >= 1 only
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
  int count() const { return tail() + 4; }
};

// bf16 weight cache: every matrix as stored ([out,in], for the forward NT GEMM) and transposed
// ([in,out], for the dX NT GEMM); the embedding/decoder matrix only as stored.
struct CachePlan {
  size_t emb;
  size_t wqkv[64], wqkv_t[64], wo[64], wo_t[64], wi[64], wi_t[64], wom[64], wom_t[64];
  size_t dense, dense_t;
  size_t total;
};

bool plan_cache(const snx_model_desc* d, CachePlan& c) {
  if (d->layers > 64 || d->layers < 1) return false;
  const size_t H = d->hidden, I = d->inter, V = d->vocab;
  size_t off = 0;
  auto take = [&](size_t elems) { size_t o = off; off = al(off + elems * 2); return o; };
  c.emb = take(V * H);
  for (int l = 0; l < d->layers; ++l) {
    c.wqkv[l] = take(3 * H * H); c.wqkv_t[l] = take(3 * H * H);
    c.wo[l] = take(H * H); c.wo_t[l] = take(H * H);
    c.wi[l] = take(2 * I * H); c.wi_t[l] = take(2 * I * H);
    c.wom[l] = take(H * I); c.wom_t[l] = take(H * I);
  }
  c.dense = take(H * H); c.dense_t = take(H * H);
  c.total = off;
  return true;
}

// activation arena written by forward and read by backward
struct SavedPlan {
  size_t h[129];                 // residual stream after embeddings / each sub-layer, fp32 [T,H]
  size_t x_attn[64], x_mlp[64];  // LayerNorm outputs feeding Wqkv / Wi, bf16 [T,H]
  size_t qkv[64];                // post-RoPE q,k and v, bf16 [T,3H]
  size_t attn[64];               // attention output (input of attn.Wo), bf16 [T,H]
  size_t lse[64];                // fp32 [heads,T]
  size_t u[64];                  // Wi output [a|g], bf16 [T,2I]
  size_t y[64];                  // GeGLU output (input of mlp.Wo), bf16 [T,I]
  size_t xf, dd, hd;             // final-norm out, head.dense out (pre-GELU), head out: bf16 [T,H]
  size_t keys;                   // u32 [nseq,V]
  size_t rowpart;                // forward-only scratch of the SPLADE head
  size_t rope_rows[2];           // fp32 [T,32,2]: every token's (cos, sin) row for theta_global / theta_local
  size_t total;
};

bool plan_saved(const snx_model_desc* d, long T, long nseq, bool save, SavedPlan& s) {
  if (d->layers > 64 || d->layers < 1) return false;
  const size_t H = d->hidden, I = d->inter, V = d->vocab, L = d->layers;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = al(off + bytes); return o; };
  if (save) {
    for (size_t i = 0; i <= 2 * L; ++i) s.h[i] = take(T * H * 4);
    for (size_t l = 0; l < L; ++l) {
      s.x_attn[l] = take(T * H * 2); s.x_mlp[l] = take(T * H * 2);
      s.qkv[l] = take(T * 3 * H * 2); s.attn[l] = take(T * H * 2);
      s.lse[l] = take((size_t)d->heads * T * 4);
      s.u[l] = take(T * 2 * I * 2); s.y[l] = take(T * I * 2);
    }
  } else {                       // inference: ping-pong the stream, reuse one set of layer buffers
    const size_t h0 = take(T * H * 4), h1 = take(T * H * 4);
    for (size_t i = 0; i <= 2 * L; ++i) s.h[i] = (i & 1) ? h1 : h0;
    const size_t xa = take(T * H * 2), xm = take(T * H * 2), qkv = take(T * 3 * H * 2), at = take(T * H * 2);
    const size_t ls = take((size_t)d->heads * T * 4), u = take(T * 2 * I * 2), y = take(T * I * 2);
    for (size_t l = 0; l < L; ++l) {
      s.x_attn[l] = xa; s.x_mlp[l] = xm; s.qkv[l] = qkv; s.attn[l] = at; s.lse[l] = ls; s.u[l] = u; s.y[l] = y;
    }
  }
  s.xf = take(T * H * 2); s.dd = take(T * H * 2); s.hd = take(T * H * 2);
  s.keys = take(nseq * V * 4);
  s.rowpart = take(snx_splade_head_scratch_bytes((int)T, (int)V));
  s.rope_rows[0] = take(T * 256);
  s.rope_rows[1] = take(T * 256);
  s.total = off;
  return true;
}

struct BwdPlan {
  size_t dh;        // fp32 [T,H]   gradient of the residual stream
  // gradients of the four Linear outputs of a layer, double buffered by layer parity: the layer's weight-gradient
  // GEMMs run as ONE grouped launch on the side stream after its dX chain, while the next layer already writes
  // the other set
  size_t p[2];      // bf16 [T,H]   d(mlp.Wo out)  = bf16(dh) entering the layer
  size_t du[2];     // bf16 [T,2I]  d(Wi out)      (GeGLU backward output, interleaved)
  size_t q[2];      // bf16 [T,H]   d(attn.Wo out) = bf16(dh) after the MLP block
  size_t dqkv[2];   // bf16 [T,3H]  d(Wqkv out)
  size_t b;         // bf16 [T,H]   scratch (d head.dense out; d attention out)
  size_t delta;     // fp32 [heads,T]
  size_t splade;    // bucket lists of the routed SPLADE backward
  // workspaces of the ORDERED weight-gradient reductions ("det_reduce"): partial slabs of the layer's grouped dW launch
  // (side stream), of the head's dW launch (launch stream: may overlap the side stream's), partial dw rows of the
  // LayerNorm backward launches + the embedding gradient's token lists and dx rows (launch stream, one at a time)
  size_t tnws_side, tnws_side_bytes, tnws_main, tnws_main_bytes;
  size_t lnws, ln_slot, lnws_emb, lnws_emb_bytes;   // one slot of partial dw rows per LayerNorm (reduced in one batched launch per
                                                    // unit range), the embedding's workspace behind them
  size_t total;
};

void plan_bwd(const snx_model_desc* d, long T, long nseq, long max_seqlen, BwdPlan& p) {
  const size_t H = d->hidden, I = d->inter;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = al(off + bytes); return o; };
  p.dh = take(T * H * 4);
  for (int k = 0; k < 2; ++k) {
    p.p[k] = take(T * H * 2); p.du[k] = take(T * 2 * I * 2); p.q[k] = take(T * H * 2); p.dqkv[k] = take(T * 3 * H * 2);
  }
  p.b = take(T * H * 2);
  p.delta = take((size_t)d->heads * T * 4);
  p.splade = take(snx_splade_bwd_scratch_bytes((int)nseq, (int)max_seqlen, d->vocab));
  {
    const int Hh = d->hidden, Ii = d->inter;
    const snx_tn_problem layer[4] = {{nullptr, nullptr, nullptr, 3 * Hh, Hh, 0, 0}, {nullptr, nullptr, nullptr, 2 * Ii, Hh, 1, 0},
                                     {nullptr, nullptr, nullptr, Hh, Ii, 0, 0}, {nullptr, nullptr, nullptr, Hh, Hh, 0, 0}};
    p.tnws_side_bytes = snx_gemm_tn_workspace_bytes(layer, 4, (int32_t)T);
    p.tnws_main_bytes = snx_gemm_tn_workspace_bytes(layer + 3, 1, (int32_t)T);
    p.ln_slot = al(snx_ln_bwd_workspace_bytes((int32_t)T, Hh));
    p.lnws_emb_bytes = snx_embed_ln_bwd_workspace_bytes((int32_t)T, Hh, d->vocab);
  }
  p.tnws_side = take(p.tnws_side_bytes);
  p.tnws_main = take(p.tnws_main_bytes);
  p.lnws = take(p.ln_slot * (2 * (size_t)d->layers + 2));
  p.lnws_emb = take(p.lnws_emb_bytes);
  p.total = off;
}

bool desc_ok(const snx_model_desc* d) {
  return d && d->vocab > 0 && d->hidden > 0 && d->hidden % 256 == 0 && d->hidden <= 1024 && d->inter > 0 &&
         d->inter % 64 == 0 && d->layers >= 1 && d->layers <= 64 && d->heads > 0 && d->head_dim == 64 &&
         d->heads * d->head_dim == d->hidden && d->global_every >= 1 && d->window >= 0;
}

__global__ void add_bf16_into_f32_kernel(float* __restrict__ dst, const bf16_t* __restrict__ src, long n4) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  f32x4 a = *(f32x4*)(dst + i * 4);
  const bf16x4 b = *(const bf16x4*)(src + i * 4);
  a[0] += bf2f(b[0]); a[1] += bf2f(b[1]); a[2] += bf2f(b[2]); a[3] += bf2f(b[3]);
  *(f32x4*)(dst + i * 4) = a;
}


// ---- optional per-kernel-class timing (HIP events on the launch stream) -------------------
// Disabled by default (zero overhead: one predictable branch).  bench.py enables it for a few
// extra steps to attribute time and algorithmic FLOPs/bytes to kernel classes.
enum ProfClass { PC_GEMM_NT = 0, PC_GEMM_NT_RESID, PC_GEMM_TN, PC_ATTN_FWD, PC_ATTN_BWD, PC_LN_FWD, PC_LN_BWD,
                 PC_ROPE, PC_GEGLU, PC_DECODER_SPLADE, PC_SPLADE_BWD, PC_EMBED, PC_CAST, PC_COUNT };
const char* const kProfNames[PC_COUNT] = {"gemm_nt_bf16", "gemm_nt_resid", "gemm_tn_accum", "attn_fwd", "attn_bwd",
                                          "ln_fwd", "ln_bwd", "rope", "geglu", "decoder_splade_fwd", "splade_bwd",
                                          "embed_ln", "cast"};
struct ProfRec { int cls; double work; hipEvent_t a, b; };
struct Prof {
  bool on = false;
  std::vector<ProfRec> recs;
  std::vector<hipEvent_t> pool;
  hipEvent_t get() {
    if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
    hipEvent_t e; hipEventCreate(&e); return e;
  }
} g_prof;

struct ProfScope {
  bool live; hipStream_t st; ProfRec r;
  ProfScope(int cls, double work, hipStream_t s) : live(g_prof.on), st(s) {
    if (live) { r.cls = cls; r.work = work; r.a = g_prof.get(); r.b = g_prof.get(); hipEventRecord(r.a, st); }
  }
  ~ProfScope() { if (live) { hipEventRecord(r.b, st); g_prof.recs.push_back(r); } }
};
#define PROF(cls, work) ProfScope prof_scope__(cls, (double)(work), st)

// ---- side stream for the weight-gradient GEMMs ------------------------------------------------
// dW = dY^T X needs nothing from the dX chain after its operands exist, and is MFMA-bound while
// the LayerNorm / attention backward kernels beside it are HBM- and VALU-bound; it also fills the
// last, partly empty round of workgroups of the dX GEMMs.  The backward therefore forks every dW
// GEMM onto one internal stream (events both ways guard the three scratch buffers it reads) and
// joins before returning, so callers still see everything ordered on THEIR stream.
// One process drives one GPU, hence one side stream per process.
struct Side {
  hipStream_t s = nullptr;
  std::vector<hipEvent_t> ev;
  size_t next = 0;
  int enabled = -1;
  hipEvent_t done[2] = {nullptr, nullptr};   // grouped weight-gradient launch that last read buffer set k
  bool on() {
    if (enabled < 0) {                       // decided at the first backward (snx_configure "bwd_overlap" before it)
      enabled = g_snx_cfg.bwd_overlap ? 1 : 0;
      if (enabled) {
        // LOWEST priority: the weight-gradient workgroups fill the slots the dX chain leaves free (last partial
        // round of a GEMM, LayerNorm / attention phases) instead of competing with it.  "side_prio" = 0: default.
        int least = 0, greatest = 0;
        const bool low = g_snx_cfg.side_prio != 0;
        if (low && hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = 0;
        if (hipStreamCreateWithPriority(&s, hipStreamNonBlocking, low ? least : 0) != hipSuccess) enabled = 0;
      }
    }
    return enabled == 1;
  }
  hipEvent_t get() {
    if (next == ev.size()) {
      hipEvent_t e = nullptr;
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
      ev.push_back(e);
    }
    return ev[next++];
  }
} g_side;

// unmasked (query, key) pairs of one attention layer over the call's sequence groups (group maximum lengths:
// exact for dense batches, an upper bound for ragged ones); window < 0 = global layer
double attn_pairs(const int32_t* groups, int nseq, int max_seqlen, int window) {
  int32_t one[4] = {1, 0, nseq, max_seqlen};
  const int32_t* g = groups ? groups : one;
  double tot = 0;
  for (int i = 0; i < g[0]; ++i) {
    const double ns = g[2 + 3 * i], S = g[3 + 3 * i];
    double per = S * S;
    if (window >= 0 && 2 * window + 1 < S) {     // band |i-j| <= w clipped at both ends
      const double w = window;
      per = S * (2 * w + 1) - w * (w + 1);
    }
    tot += ns * per;
  }
  return tot;
}

#define RC(call)            \
  do {                      \
    int rc__ = (call);      \
    if (rc__) return rc__;  \
  } while (0)

}  // namespace

extern "C" int snx_version(void) { return 1; }

extern "C" int32_t snx_param_count(const snx_model_desc* d) {
  if (!desc_ok(d)) return -1;
  PIdx p{d->layers};
  return p.count();
}

extern "C" size_t snx_weight_cache_bytes(const snx_model_desc* d) {
  CachePlan c;
  if (!desc_ok(d) || !plan_cache(d, c)) return 0;
  return c.total;
}

extern "C" int snx_weight_cache_refresh(const snx_model_desc* d, const void* const* params, void* cache,
                                        hipStream_t st) {
  CachePlan c;
  if (!desc_ok(d) || !plan_cache(d, c)) return SNX_E_SHAPE;
  if (!params || !cache) return SNX_E_ARG;
  PIdx p{d->layers};
  char* base = (char*)cache;
  const int H = d->hidden, I = d->inter, V = d->vocab;
  auto both = [&](int idx, size_t o, size_t ot, int R, int C) -> int {
    RC(snx_cast_bf16((const float*)params[idx], base + o, (int64_t)R * C, st));
    return snx_cast_transpose_bf16((const float*)params[idx], base + ot, R, C, st);
  };
  RC(snx_cast_bf16((const float*)params[p.tok_emb()], base + c.emb, (int64_t)V * H, st));
  const bool per_tensor = g_snx_cfg.wcache_per_tensor != 0;   // A/B and tests: one launch per copy
  if (!per_tensor && d->layers <= SNX_CAST_BATCH_MAX) {
    // one launch per shape class (Wqkv, attn.Wo, Wi, mlp.Wo), blockIdx.z = layer: 7 launches instead of 157
    CastBatch bq, bo, bi, bm;
    for (int l = 0; l < d->layers; ++l) {
      bq.src[l] = (const float*)params[p.wqkv(l)]; bq.out[l] = (bf16_t*)(base + c.wqkv[l]); bq.out_t[l] = (bf16_t*)(base + c.wqkv_t[l]);
      bo.src[l] = (const float*)params[p.wo(l)]; bo.out[l] = (bf16_t*)(base + c.wo[l]); bo.out_t[l] = (bf16_t*)(base + c.wo_t[l]);
      bi.src[l] = (const float*)params[p.wi(l)]; bi.out[l] = (bf16_t*)(base + c.wi[l]); bi.out_t[l] = (bf16_t*)(base + c.wi_t[l]);
      bm.src[l] = (const float*)params[p.wo_mlp(l)]; bm.out[l] = (bf16_t*)(base + c.wom[l]); bm.out_t[l] = (bf16_t*)(base + c.wom_t[l]);
    }
    RC(snx_cast_both_batched(bq, d->layers, 3 * H, H, 0, st));
    RC(snx_cast_both_batched(bo, d->layers, H, H, 0, st));
    RC(snx_cast_both_batched(bi, d->layers, 2 * I, H, 1, st));
    RC(snx_cast_both_batched(bm, d->layers, H, I, 0, st));
    RC(both(p.head_dense(), c.dense, c.dense_t, H, H));
    return SNX_OK;
  }
  for (int l = 0; l < d->layers; ++l) {
    RC(both(p.wqkv(l), c.wqkv[l], c.wqkv_t[l], 3 * H, H));
    RC(both(p.wo(l), c.wo[l], c.wo_t[l], H, H));
    RC(snx_cast_geglu_interleave((const float*)params[p.wi(l)], base + c.wi[l], base + c.wi_t[l], I, H, st));
    RC(both(p.wo_mlp(l), c.wom[l], c.wom_t[l], H, I));
  }
  RC(both(p.head_dense(), c.dense, c.dense_t, H, H));
  return SNX_OK;
}

extern "C" size_t snx_model_workspace_bytes(const snx_model_desc* d, int32_t T, int32_t nseq, int32_t save_for_bwd) {
  SavedPlan s;
  if (!desc_ok(d) || T <= 0 || nseq <= 0 || !plan_saved(d, T, nseq, save_for_bwd != 0, s)) return 0;
  return s.total;
}

extern "C" size_t snx_model_keys_offset(const snx_model_desc* d, int32_t T, int32_t nseq) {
  SavedPlan s;
  if (!desc_ok(d) || T <= 0 || nseq <= 0 || !plan_saved(d, T, nseq, true, s)) return (size_t)-1;
  return s.keys;
}

extern "C" size_t snx_model_bwd_workspace_bytes(const snx_model_desc* d, int32_t T, int32_t nseq, int32_t max_seqlen) {
  if (!desc_ok(d) || T <= 0 || nseq <= 0 || max_seqlen <= 0) return 0;
  BwdPlan p;
  plan_bwd(d, T, nseq, max_seqlen, p);
  return p.total;
}

extern "C" int snx_model_forward(const snx_model_desc* d, const void* const* params, const void* wcache,
                                 const int64_t* ids, const int64_t* mask, const int32_t* cu_seqlens,
                                 const int32_t* pos, const float* rope_global, const float* rope_local, void* saved,
                                 float* sparse, float* token_weights, const int32_t* groups, int32_t T,
                                 int32_t nseq, int32_t max_seqlen, int32_t flags, hipStream_t st) {
  return snx_model_forward_range(d, params, wcache, ids, mask, cu_seqlens, pos, rope_global, rope_local, saved, sparse,
                                 token_weights, groups, T, nseq, 0, 0, T, nseq, max_seqlen, flags, st);
}

// One PASS of a micro-step into a row range of a larger arena (the reference's call pattern, ref:train_v33_ddp.py:339-343:
// query, positive and negative batches are three model(...) calls).  The arena, `sparse` and `token_weights` are laid out
// for T_plan token rows / nseq_plan sequences (the whole micro-step); this call fills rows [row0, row0 + T) and sequences
// [seq0, seq0 + nseq) from the pass's OWN ids / mask / pos / cu_seqlens (cu_seqlens[0] = 0).  Every per-token buffer is
// row-major, so a pass is the same kernels on shifted pointers; only the attention LSE is head-major ([heads, T_plan]) and
// keeps T_plan as its stride.  After the last pass the arena is exactly what one fused snx_model_forward over all rows
// leaves behind, and ONE snx_model_backward over (T_plan, nseq_plan) back-propagates the micro-step.
extern "C" int snx_model_forward_range(const snx_model_desc* d, const void* const* params, const void* wcache,
                                       const int64_t* ids, const int64_t* mask, const int32_t* cu_seqlens,
                                       const int32_t* pos, const float* rope_global, const float* rope_local,
                                       void* saved, float* sparse_all, float* token_weights_all, const int32_t* groups,
                                       int32_t T_plan, int32_t nseq_plan, int32_t row0, int32_t seq0, int32_t T,
                                       int32_t nseq, int32_t max_seqlen, int32_t flags, hipStream_t st) {
  if (!desc_ok(d)) return SNX_E_SHAPE;
  if (!params || !wcache || !ids || !mask || !cu_seqlens || !pos || !rope_global || !rope_local || !saved || !sparse_all ||
      !token_weights_all || T <= 0 || nseq <= 0 || max_seqlen <= 0)
    return SNX_E_ARG;
  if (row0 < 0 || seq0 < 0 || (long)row0 + T > T_plan || (long)seq0 + nseq > nseq_plan) return SNX_E_ARG;
  const bool ranged = T != T_plan || nseq != nseq_plan;
  if (ranged && groups) return SNX_E_ARG;                  // a pass of a larger arena is one sequence group
  const bool save = (flags & SNX_FWD_SAVE_FOR_BACKWARD) != 0;
  if (ranged && !save) return SNX_E_ARG;                   // (the no-save plan reuses buffers across layers: nothing to share)
  CachePlan c;
  SavedPlan s;
  if (!plan_cache(d, c) || !plan_saved(d, T_plan, nseq_plan, save, s)) return SNX_E_SHAPE;
  // The forward's non-temporal streams ("stream_nt" bits 1, 2, 8: residual-stream rows, the saved u) are a TRAINING policy:
  // without a backward the stream ping-pongs between two buffers and small inference batches live in the caches -- an nt
  // store would send the next LayerNorm to HBM for them.  (One thread drives the library: the switch is restored on return.)
  struct NtScope {
    int saved;
    explicit NtScope(bool keep) : saved(g_snx_cfg.stream_nt) { if (!keep) g_snx_cfg.stream_nt &= ~(1 | 2 | 8); }
    ~NtScope() { g_snx_cfg.stream_nt = saved; }
  } nt_scope(save);
  PIdx p{d->layers};
  const char* wc = (const char*)wcache;
  const int H = d->hidden, I = d->inter, V = d->vocab, L = d->layers;
  // `sv + s.x` below = the pass's first row of buffer x: SavedView shifts a buffer offset by row0 rows of that buffer
  struct SavedView {
    char* base; size_t row0;
    char* at(size_t off, size_t row_bytes) const { return base + off + row0 * row_bytes; }
  } view{(char*)saved, (size_t)row0};
  float* sparse = sparse_all + (size_t)seq0 * V;
  float* token_weights = token_weights_all + row0;
  auto F = [&](int idx) { return (const float*)params[idx]; };
  auto hbuf = [&](int i) { return (float*)view.at(s.h[i], (size_t)H * 4); };
  // shifted views of the per-token bf16 buffers (named as the plan names them)
  struct Sv {
    const SavedView& v; const SavedPlan& s; size_t H, I;
    char* x_attn(int l) const { return v.at(s.x_attn[l], H * 2); }
    char* x_mlp(int l) const { return v.at(s.x_mlp[l], H * 2); }
    char* qkv(int l) const { return v.at(s.qkv[l], 3 * H * 2); }
    char* attn(int l) const { return v.at(s.attn[l], H * 2); }
    char* lse(int l) const { return v.at(s.lse[l], 4); }          // [heads, T_plan]: first token of the pass in head 0
    char* u(int l) const { return v.at(s.u[l], 2 * I * 2); }
    char* y(int l) const { return v.at(s.y[l], I * 2); }
    char* xf() const { return v.at(s.xf, H * 2); }
    char* dd() const { return v.at(s.dd, H * 2); }
    char* hd() const { return v.at(s.hd, H * 2); }
    char* rope_rows(int k) const { return v.at(s.rope_rows[k], 256); }
  } sv{view, s, (size_t)H, (size_t)I};
  uint32_t* keys = (uint32_t*)((char*)saved + s.keys) + (size_t)seq0 * V;
  char* rowpart = (char*)saved + s.rowpart;

  const double TH = (double)T * H;
  { PROF(PC_EMBED, TH * 10); RC(snx_embed_ln_fwd(ids, F(p.tok_emb()), F(p.emb_norm()), hbuf(0), sv.x_attn(0), T, H, d->ln_eps, st)); }
  // positions -> (cos, sin) rows once per pass and theta: the 22 Wqkv write-backs then read them without the dependent load
  { PROF(PC_ROPE, 2.0 * T * 512);
    RC(snx_rope_rows(rope_global, pos, (float*)sv.rope_rows(0), T, st));
    RC(snx_rope_rows(rope_local, pos, (float*)sv.rope_rows(1), T, st)); }
  // "resid_in_ln": the Wo GEMMs store their bf16 result (ybuf: the head's dense buffer, free until the layers are done)
  // and the residual add h + float(y) happens inside the LayerNorm that follows every one of them -- the same bits as the
  // GEMM's residual epilogue, the fp32 stream's read + write moved from an MFMA-bound kernel to an HBM-bound one.
  const bool ril = g_snx_cfg.resid_in_ln != 0;
  char* ybuf = sv.dd();
  for (int l = 0; l < L; ++l) {
    const bool global = (l % d->global_every) == 0;
    if (l > 0 && !ril) { PROF(PC_LN_FWD, TH * 6); RC(snx_ln_fwd(hbuf(2 * l), F(p.attn_norm(l)), sv.x_attn(l), T, H, d->ln_eps, st)); }
    { PROF(PC_GEMM_NT, 2.0 * T * 3 * H * H);      // Wqkv + RoPE fused
      RC(snx_gemm_nt_rope_rows(sv.x_attn(l), wc + c.wqkv[l], sv.qkv(l), global ? rope_global : rope_local, pos,
                               (const float*)sv.rope_rows(global ? 0 : 1), 2 * H, T, 3 * H, H, st)); }
    { PROF(PC_ATTN_FWD, 4.0 * H * attn_pairs(groups, nseq, max_seqlen, global ? -1 : d->window));
      RC(snx_attn_fwd_ex(sv.qkv(l), cu_seqlens, mask, sv.attn(l), (float*)sv.lse(l), groups, T_plan /* LSE stride */, nseq,
                         max_seqlen, d->heads, d->head_dim, global ? -1 : d->window, st)); }
    if (ril) {
      { PROF(PC_GEMM_NT_RESID, 2.0 * TH * H); RC(snx_gemm_nt_bf16(sv.attn(l), wc + c.wo[l], ybuf, T, H, H, st)); }
      { PROF(PC_LN_FWD, TH * 12); RC(snx_ln_fwd_add(hbuf(2 * l), ybuf, F(p.mlp_norm(l)), hbuf(2 * l + 1), sv.x_mlp(l), T, H, d->ln_eps, st)); }
    } else {
      { PROF(PC_GEMM_NT_RESID, 2.0 * TH * H); RC(snx_gemm_nt_resid(sv.attn(l), wc + c.wo[l], hbuf(2 * l), hbuf(2 * l + 1), T, H, H, st)); }
      { PROF(PC_LN_FWD, TH * 6); RC(snx_ln_fwd(hbuf(2 * l + 1), F(p.mlp_norm(l)), sv.x_mlp(l), T, H, d->ln_eps, st)); }
    }
    { PROF(PC_GEMM_NT, 2.0 * T * 2 * I * H);      // Wi + GeGLU fused (u kept in the interleaved column order)
      RC(snx_gemm_nt_geglu_fwd(sv.x_mlp(l), wc + c.wi[l], sv.u(l), sv.y(l), T, 2 * I, H, st)); }
    if (ril) {
      { PROF(PC_GEMM_NT_RESID, 2.0 * TH * I); RC(snx_gemm_nt_bf16(sv.y(l), wc + c.wom[l], ybuf, T, H, I, st)); }
      const bool last = l + 1 == L;
      { PROF(PC_LN_FWD, TH * 12);
        RC(snx_ln_fwd_add(hbuf(2 * l + 1), ybuf, F(last ? p.final_norm() : p.attn_norm(l + 1)), hbuf(2 * l + 2),
                          last ? sv.xf() : sv.x_attn(l + 1), T, H, d->ln_eps, st)); }
    } else {
      { PROF(PC_GEMM_NT_RESID, 2.0 * TH * I); RC(snx_gemm_nt_resid(sv.y(l), wc + c.wom[l], hbuf(2 * l + 1), hbuf(2 * l + 2), T, H, I, st)); }
    }
  }
  if (!ril) { PROF(PC_LN_FWD, TH * 6); RC(snx_ln_fwd(hbuf(2 * L), F(p.final_norm()), sv.xf(), T, H, d->ln_eps, st)); }
  { PROF(PC_GEMM_NT, 2.0 * TH * H); RC(snx_gemm_nt_bf16(sv.xf(), wc + c.dense, sv.dd(), T, H, H, st)); }
  { PROF(PC_LN_FWD, TH * 4); RC(snx_gelu_ln_fwd(sv.dd(), F(p.head_norm()), sv.hd(), T, H, d->ln_eps, st)); }
  {
    // Sequence groups (e.g. 64-token queries and 256-token documents concatenated in one call)
    // get the decoder tile height that fits their length; groups = {n, (seq_begin, nseq, max_len)*n}.
    PROF(PC_DECODER_SPLADE, 2.0 * TH * V);
    int32_t one[4] = {1, 0, nseq, max_seqlen};
    // the 256x192 decoder handles mixed lengths by itself: one call (one pre-pass, one tile schedule) for all groups
    const int32_t* g = (groups && !snx_dec256_takes(T)) ? groups : one;
    for (int i = 0; i < g[0]; ++i) {
      const int sb = g[1 + 3 * i], ns = g[2 + 3 * i], ml = g[3 + 3 * i];
      if (sb < 0 || ns <= 0 || sb + ns > nseq || ml <= 0 || ml > max_seqlen) return SNX_E_ARG;
      RC(snx_decoder_splade_fwd_ex(sv.hd(), wc + c.emb, F(p.dec_bias()), cu_seqlens + sb, mask,
                                   sparse + (size_t)sb * V, keys + (size_t)sb * V, token_weights,
                                   rowpart, T, ns, ml, V, H, i + 1 == g[0], st));
    }
  }
  return SNX_OK;
}

// The backward is a chain of L + 2 UNITS in execution order: unit 0 = SPLADE tail + decoder + head + final norm,
// unit 1 + i = encoder layer L-1-i, unit L + 1 = embeddings.  A caller may run it in several calls over
// consecutive unit ranges (same scratch buffer): the parameter gradients of a finished range are final, so a
// data-parallel trainer can all-reduce them on `notify` while the remaining units still run.
extern "C" int snx_model_backward_units(const snx_model_desc* d, const void* const* params, void* const* grads,
                                        const void* wcache, const int64_t* ids, const int64_t* mask,
                                        const int32_t* cu_seqlens, const int32_t* pos, const float* rope_global,
                                        const float* rope_local, const void* saved, const float* g_sparse,
                                        void* scratch, const int32_t* groups, int32_t T, int32_t nseq,
                                        int32_t max_seqlen, int32_t unit_begin, int32_t unit_end, hipStream_t notify,
                                        hipStream_t st) {
  return snx_model_backward_units_range(d, params, grads, wcache, ids, mask, cu_seqlens, pos, rope_global, rope_local, saved,
                                        g_sparse, scratch, groups, T, nseq, T, nseq, max_seqlen, unit_begin, unit_end,
                                        notify, st);
}

// The backward over the first T rows / nseq sequences of an arena (and a scratch buffer) laid out for T_plan rows /
// nseq_plan sequences: what is left to back-propagate when a micro-step placed fewer passes into its arena than the
// pattern promised (snx_model_forward_range).  T = T_plan, nseq = nseq_plan is the ordinary backward.
extern "C" int snx_model_backward_units_range(const snx_model_desc* d, const void* const* params, void* const* grads,
                                              const void* wcache, const int64_t* ids, const int64_t* mask,
                                              const int32_t* cu_seqlens, const int32_t* pos, const float* rope_global,
                                              const float* rope_local, const void* saved, const float* g_sparse,
                                              void* scratch, const int32_t* groups, int32_t T_plan, int32_t nseq_plan,
                                              int32_t T, int32_t nseq, int32_t max_seqlen, int32_t unit_begin,
                                              int32_t unit_end, hipStream_t notify, hipStream_t st) {
  if (!desc_ok(d)) return SNX_E_SHAPE;
  if (!params || !grads || !wcache || !ids || !mask || !cu_seqlens || !pos || !rope_global || !rope_local || !saved ||
      !g_sparse || !scratch || T <= 0 || nseq <= 0 || max_seqlen <= 0 || T > T_plan || nseq > nseq_plan)
    return SNX_E_ARG;
  if (unit_begin < 0 || unit_end > d->layers + 2 || unit_begin >= unit_end) return SNX_E_ARG;
  CachePlan c;
  SavedPlan s;
  BwdPlan b;
  if (!plan_cache(d, c) || !plan_saved(d, T_plan, nseq_plan, true, s)) return SNX_E_SHAPE;
  plan_bwd(d, T_plan, nseq_plan, max_seqlen, b);
  PIdx p{d->layers};
  const char* wc = (const char*)wcache;
  const char* sv = (const char*)saved;
  char* sc = (char*)scratch;
  const int H = d->hidden, I = d->inter, V = d->vocab, L = d->layers;
  auto F = [&](int idx) { return (const float*)params[idx]; };
  auto G = [&](int idx) { return (float*)grads[idx]; };
  auto hbuf = [&](int i) { return (const float*)(sv + s.h[i]); };
  float* dh = (float*)(sc + b.dh);
  char* Bb = sc + b.b;
  const long n4 = (long)T * H / 4;
  // LayerNorm weight gradients: every launch leaves its partial rows in a slot of its own (slot 0 head norm, 1 final
  // norm, 2 + 2l / 3 + 2l the layer's mlp / attention norm); ONE batched kernel adds them to dw at the end of this call
  LnDwBatch lnb;
  int nlnb = 0;
#define LNWS(slot) sc + b.lnws + (size_t)(slot) * b.ln_slot, b.ln_slot, &lnb, &nlnb

  // The weight-gradient GEMMs go to the side stream unless the per-class profiler is timing kernels one by one:
  // one GROUPED launch per layer (its four Linears), issued after the layer's dX chain.
  const bool overlap = !g_prof.on && g_side.on();
  hipStream_t ss = overlap ? g_side.s : st;
  if (unit_begin == 0) {                              // a new backward: recycle the event pool
    g_side.next = 0;
    g_side.done[0] = g_side.done[1] = nullptr;
  }
  auto fork = [&]() -> int {                          // side stream: wait for everything enqueued on st so far
    if (!overlap) return SNX_OK;
    hipEvent_t e = g_side.get();
    if (!e || hipEventRecord(e, st) != hipSuccess || hipStreamWaitEvent(ss, e, 0) != hipSuccess) return SNX_E_ARG;
    return SNX_OK;
  };
  auto mark = [&](hipEvent_t& e) -> int {             // e = "side stream has finished what it was given so far"
    if (!overlap) return SNX_OK;
    e = g_side.get();
    return (e && hipEventRecord(e, ss) == hipSuccess) ? SNX_OK : SNX_E_ARG;
  };
  auto join = [&](hipEvent_t e) -> int {              // st: wait for that point of the side stream
    if (!overlap || !e) return SNX_OK;
    return hipStreamWaitEvent(st, e, 0) == hipSuccess ? SNX_OK : SNX_E_ARG;
  };

  const double TH = (double)T * H;
  if (unit_begin == 0) {
    char* A = sc + b.p[(L - 1) & 1];                  // d(mlp.Wo out) of the last layer ends up here
    // SPLADE tail + decoder (sparse routed), head
    { PROF(PC_SPLADE_BWD, 2.0 * 2.0 * nseq * V * H);
      // (round 6, measured and dropped: the weight half -- dE / db -- on the idle side stream beside the activation half:
      // 45.23 against 45.01 ms per micro-step on one box, ABAB; two gathers that share the CUs' wave slots contend more
      // than they hide)
      RC(snx_splade_bwd(g_sparse, (const uint32_t*)(sv + s.keys), sv + s.hd, wc + c.emb, cu_seqlens, A, G(p.tok_emb()),
                        G(p.dec_bias()), sc + b.splade, T, nseq, max_seqlen, V, H, st)); }
    { PROF(PC_LN_BWD, TH * 6); RC(snx_gelu_ln_bwd_x(A, sv + s.dd, F(p.head_norm()), Bb, G(p.head_norm()), T, H, d->ln_eps, LNWS(0), st)); }
    { PROF(PC_GEMM_TN, 2.0 * TH * H);
      RC(snx_gemm_tn_accum(Bb, sv + s.xf, G(p.head_dense()), T, H, H, sc + b.tnws_main, b.tnws_main_bytes, st)); }
    { PROF(PC_GEMM_NT, 2.0 * TH * H); RC(snx_gemm_nt_bf16(Bb, wc + c.dense_t, A, T, H, H, st)); }
    // every LayerNorm backward also emits bf16(dh): the gradient of the next bf16 branch output
    { PROF(PC_LN_BWD, TH * 12); RC(snx_ln_bwd_x(A, hbuf(2 * L), F(p.final_norm()), dh, A, G(p.final_norm()), T, H, d->ln_eps, 1, LNWS(1), st)); }
  }

  const int l_hi = L - 1 - (unit_begin > 1 ? unit_begin - 1 : 0);
  const int l_lo = L - 1 - ((unit_end < L + 1 ? unit_end : L + 1) - 2);
  for (int l = l_hi; l >= l_lo && l >= 0; --l) {
    const bool global = (l % d->global_every) == 0;
    const double pairs = attn_pairs(groups, nseq, max_seqlen, global ? -1 : d->window);
    const int k = l & 1;
    char *P = sc + b.p[k], *Du = sc + b.du[k], *Q = sc + b.q[k], *Dq = sc + b.dqkv[k], *Pn = sc + b.p[k ^ 1];
    // this layer overwrites the buffer set the grouped launch of layer l + 2 read (and Pn, read by layer l + 1's)
    RC(join(g_side.done[k]));
    // ---- MLP:  h[2l+2] = h[2l+1] + Wo( gelu(a) * g ),  [a|g] = Wi( LN(h[2l+1]) )
    { PROF(PC_GEMM_NT, 2.0 * TH * I);             // dy = dh Wo, GeGLU backward fused -> du [T,2I] (interleaved)
      RC(snx_gemm_nt_geglu_bwd(P, wc + c.wom_t[l], sv + s.u[l], Du, T, I, H, st)); }
    { PROF(PC_GEMM_NT, 2.0 * TH * 2 * I); RC(snx_gemm_nt_bf16(Du, wc + c.wi_t[l], Q, T, H, 2 * I, st)); }   // dx [T,H]
    { PROF(PC_LN_BWD, TH * 16); RC(snx_ln_bwd_x(Q, hbuf(2 * l + 1), F(p.mlp_norm(l)), dh, Q, G(p.mlp_norm(l)), T, H, d->ln_eps, 0, LNWS(2 + 2 * l), st)); }
    // ---- attention:  h[2l+1] = h[2l] + Wo( attn( rope( Wqkv( LN(h[2l]) ) ) ) )
    { PROF(PC_GEMM_NT, 2.0 * TH * H); RC(snx_gemm_nt_bf16(Q, wc + c.wo_t[l], Bb, T, H, H, st)); }    // d(attn out)
    { PROF(PC_ATTN_BWD, 10.0 * H * pairs);
      RC(snx_attn_bwd_ex(sv + s.qkv[l], sv + s.attn[l], Bb, (const float*)(sv + s.lse[l]), cu_seqlens, mask,
                         (float*)(sc + b.delta), Dq, global ? rope_global : rope_local, pos, groups,
                         T_plan /* LSE / delta stride */, nseq, max_seqlen, d->heads, d->head_dim, global ? -1 : d->window,
                         st)); }                                                                  // inverse RoPE fused
    RC(join(g_side.done[k ^ 1]));                  // layer l + 1's grouped launch has finished reading Pn
    { PROF(PC_GEMM_NT, 2.0 * TH * 3 * H); RC(snx_gemm_nt_bf16(Dq, wc + c.wqkv_t[l], Pn, T, H, 3 * H, st)); }   // dx [T,H]
    // the four weight gradients of the layer: dWo(mlp) = P^T y, dWi = du^T x_mlp, dWo(attn) = Q^T attn, dWqkv = dqkv^T x_attn
    RC(fork());
    {
      ProfScope ps(PC_GEMM_TN, 2.0 * TH * (I + 2 * I + H + 3 * H), ss);
      snx_tn_problem pr[4] = {
          {Dq, sv + s.x_attn[l], G(p.wqkv(l)), 3 * H, H, 0, 0},
          {Du, sv + s.x_mlp[l], G(p.wi(l)), 2 * I, H, 1, 0},
          {P, sv + s.y[l], G(p.wo_mlp(l)), H, I, 0, 0},
          {Q, sv + s.attn[l], G(p.wo(l)), H, H, 0, 0}};
      RC(snx_gemm_tn_accum_group(pr, 4, T, sc + b.tnws_side, b.tnws_side_bytes, ss));
    }
    RC(mark(g_side.done[k]));
    if (nlnb > SNX_LN_BATCH_MAX - 4) {                  // deep models: reduce what is pending before the batch is full
      RC(snx_ln_dw_reduce_batch(lnb, nlnb, H, st));
      nlnb = 0;
    }
    if (l > 0) {
      PROF(PC_LN_BWD, TH * 16);
      RC(snx_ln_bwd_x(Pn, hbuf(2 * l), F(p.attn_norm(l)), dh, Pn, G(p.attn_norm(l)), T, H, d->ln_eps, 0, LNWS(3 + 2 * l), st));
    } else {
      PROF(PC_CAST, TH * 10);
      hipLaunchKernelGGL(add_bf16_into_f32_kernel, dim3(cdiv(n4, 256)), dim3(256), 0, st, dh, (const bf16_t*)Pn, n4);
      SNX_CHECK_LAUNCH();
    }
  }
  if (unit_end == L + 2) {
    RC(join(g_side.done[0]));                      // the side stream is in order: its last two marks cover all of it
    RC(join(g_side.done[1]));
    { PROF(PC_EMBED, TH * 12);
      RC(snx_embed_ln_bwd_x(dh, ids, F(p.tok_emb()), F(p.emb_norm()), G(p.tok_emb()), G(p.emb_norm()), T, H, V, d->ln_eps,
                            d->pad_id, sc + b.lnws_emb, b.lnws_emb_bytes, &lnb, &nlnb, st)); }
  }
  if (nlnb > 0) { PROF(PC_LN_BWD, 0); RC(snx_ln_dw_reduce_batch(lnb, nlnb, H, st)); }
#undef LNWS
  if (notify) {
    // `notify` learns that every gradient written by units [unit_begin, unit_end) is complete: it waits for
    // this point of the launch stream and of the weight-gradient side stream (the caller's stream does not)
    hipEvent_t e = g_side.get();
    if (!e || hipEventRecord(e, st) != hipSuccess || hipStreamWaitEvent(notify, e, 0) != hipSuccess) return SNX_E_ARG;
    if (overlap) {
      hipEvent_t e2 = g_side.get();
      if (!e2 || hipEventRecord(e2, ss) != hipSuccess || hipStreamWaitEvent(notify, e2, 0) != hipSuccess) return SNX_E_ARG;
    }
  }
  return SNX_OK;
}

extern "C" int snx_model_backward(const snx_model_desc* d, const void* const* params, void* const* grads,
                                  const void* wcache, const int64_t* ids, const int64_t* mask,
                                  const int32_t* cu_seqlens, const int32_t* pos, const float* rope_global,
                                  const float* rope_local, const void* saved, const float* g_sparse, void* scratch,
                                  const int32_t* groups, int32_t T, int32_t nseq, int32_t max_seqlen,
                                  hipStream_t st) {
  if (!desc_ok(d)) return SNX_E_SHAPE;
  return snx_model_backward_units(d, params, grads, wcache, ids, mask, cu_seqlens, pos, rope_global, rope_local, saved,
                                  g_sparse, scratch, groups, T, nseq, max_seqlen, 0, d->layers + 2, nullptr, st);
}

// ---- profiling API (see ProfScope above) ---------------------------------------------------
extern "C" int snx_prof_enable(int32_t on) {
  g_prof.on = on != 0;
  return SNX_OK;
}

extern "C" int32_t snx_prof_num_classes(void) { return PC_COUNT; }
extern "C" const char* snx_prof_class_name(int32_t i) { return (i >= 0 && i < PC_COUNT) ? kProfNames[i] : ""; }

// Synchronises on the recorded events; fills ms[c], launches[c], work[c] (algorithmic FLOPs for
// the MFMA classes, algorithmic bytes for the HBM-bound ones) per class and clears the records.
extern "C" int snx_prof_read(double* ms, int64_t* launches, double* work) {
  if (!ms || !launches || !work) return SNX_E_ARG;
  for (int c = 0; c < PC_COUNT; ++c) { ms[c] = 0; launches[c] = 0; work[c] = 0; }
  for (auto& r : g_prof.recs) {
    hipEventSynchronize(r.b);
    float t = 0.f;
    hipEventElapsedTime(&t, r.a, r.b);
    ms[r.cls] += t; launches[r.cls] += 1; work[r.cls] += r.work;
    g_prof.pool.push_back(r.a); g_prof.pool.push_back(r.b);
  }
  g_prof.recs.clear();
  return SNX_OK;
}
