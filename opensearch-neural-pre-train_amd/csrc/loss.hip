// SPLADELossV33 on device (K12-K15 of SURVEY.md §2.3; ref:src/model/losses.py:57-297):
//   InfoNCE over [in-batch positives | k hard negatives] + FLOPS(q) + FLOPS(pos) + FLOPS(neg)
//   [+ MarginMSE] [+ KL distillation, ref:losses.py:239-253], closed-form gradients, no host sync (scalars stay on the
//   device).
// Three launches: (1) per-vocab-chunk partial dot products / column means (LDS-staged tiles,
// coalesced along V), (2) one workgroup reducing the partials, soft-max / cross-entropy and the
// gradient coefficients, (3) the gradient of the three [*, V] inputs, again chunked over V.
// `P` may hold MORE rows than the anchors (Bp >= B): the all-gathered positives of every rank
// for cross-GPU in-batch negatives (config 4); labels are label_off + i.
#include "common.h"
#include "snx.h"

struct LossDims {
  int B, Bp, k, V, CH, nchunk, label_off, bf16_mm;
  float inv_tau, lam_q, lam_d, lam_neg, lam_mm, lam_kd, inv_tkd;
};

// workspace layout (floats)
struct LossWs {
  float *part_inb, *part_hard, *part_pos, *part_sc;   // [nchunk][B*Bp], [nchunk][B*k], [nchunk][B], [nchunk][8]
  float *mean_q, *mean_p, *mean_n;                     // [V] each
  float *G, *Gh, *dpos;                                // [B*Bp], [B*k], [B]
  float *Gkd;                                          // [B*B] coefficients of the KL-distillation term (own positives)
  float *red;                                          // [B*Bp + B*k + B] partials summed over chunks
};

static inline size_t ws_floats(int B, int Bp, int k, int V, int nchunk) {
  return (size_t)nchunk * ((size_t)B * Bp + (size_t)B * k + B + 8) + 3 * (size_t)V + 2 * ((size_t)B * Bp + (size_t)B * k + B) +
         (size_t)B * B;
}
static inline LossWs carve(float* w, int B, int Bp, int k, int V, int nchunk) {
  LossWs s;
  s.part_inb = w; w += (size_t)nchunk * B * Bp;
  s.part_hard = w; w += (size_t)nchunk * B * k;
  s.part_pos = w; w += (size_t)nchunk * B;
  s.part_sc = w; w += (size_t)nchunk * 8;
  s.mean_q = w; w += V;
  s.mean_p = w; w += V;
  s.mean_n = w; w += V;
  s.G = w; w += (size_t)B * Bp;
  s.Gh = w; w += (size_t)B * k;
  s.dpos = w; w += B;
  s.Gkd = w; w += (size_t)B * B;
  s.red = w;
  return s;
}
static inline int pick_ch(int B, int Bp) {
  int ch = 128;
  while (ch > 16 && (size_t)(B + Bp) * ch * 4 > 96 * 1024) ch >>= 1;
  return ch;
}

__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void loss_partial_kernel(const float* __restrict__ q, const float* __restrict__ p,
                                                           const float* __restrict__ n, LossDims d, LossWs w) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  __shared__ float red[4];
  const int LD = d.CH + 1;               // +1 float: row-strided reads stay conflict-free
  float* sq = sm;                       // [B][LD]
  float* sp = sm + d.B * LD;            // [Bp][LD]
  const int c0 = blockIdx.x * d.CH;
  const int cw = min(d.CH, d.V - c0);
  const int tid = threadIdx.x;
  for (int i = tid; i < d.B * d.CH; i += 256) {
    const int r = i / d.CH, c = i % d.CH;
    sq[r * LD + c] = c < cw ? q[(long)r * d.V + c0 + c] : 0.f;
  }
  for (int i = tid; i < d.Bp * d.CH; i += 256) {
    const int r = i / d.CH, c = i % d.CH;
    sp[r * LD + c] = c < cw ? p[(long)r * d.V + c0 + c] : 0.f;
  }
  __syncthreads();
  // in-batch partial dots (operands rounded to bf16 when the reference's mm runs under autocast): a thread owns a
  // 4 x 4 block of (anchor, positive) pairs -- 8 LDS reads per 16 products; every pair still sums its chunk in column
  // order, so the partials are the bits of the one-pair-per-thread loop this replaces
  {
    const int nbj = (d.Bp + 3) >> 2, nblk = ((d.B + 3) >> 2) * nbj;
    for (int blk = tid; blk < nblk; blk += 256) {
      const int i0 = (blk / nbj) * 4, j0 = (blk % nbj) * 4;
      const float* a[4];
      const float* b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a[u] = sq + min(i0 + u, d.B - 1) * LD;
        b[u] = sp + min(j0 + u, d.Bp - 1) * LD;
      }
      float acc[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[u][v] = 0.f;
      for (int c = 0; c < d.CH; ++c) {
        float av[4], bv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          av[u] = d.bf16_mm ? rbf(a[u][c]) : a[u][c];
          bv[u] = d.bf16_mm ? rbf(b[u][c]) : b[u][c];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int v = 0; v < 4; ++v) acc[u][v] += av[u] * bv[v];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v)
          if (i0 + u < d.B && j0 + v < d.Bp)
            w.part_inb[(long)blockIdx.x * d.B * d.Bp + (long)(i0 + u) * d.Bp + j0 + v] = acc[u][v];
    }
  }
  // q.p (own positive, fp32) for MarginMSE: positives of this rank start at row label_off
  for (int i = tid; i < d.B; i += 256) {
    const float* a = sq + i * LD;
    const float* b = sp + (d.label_off + i) * LD;
    float s = 0.f;
    for (int c = 0; c < d.CH; ++c) s += a[c] * b[c];
    w.part_pos[(long)blockIdx.x * d.B + i] = s;
  }
  // hard-negative dots: one wave per (i, kk) pair, lanes along the chunk
  const int lane = tid & 63, wave = tid >> 6;
  for (int pr = wave; pr < d.B * d.k; pr += 4) {
    const int i = pr / d.k;
    const float* nr = n + (long)pr * d.V + c0;
    float s = 0.f;
    for (int c = lane; c < cw; c += 64) s += sq[i * LD + c] * nr[c];
    s = wave_sum(s);
    if (lane == 0) w.part_hard[(long)blockIdx.x * d.B * d.k + pr] = s;
  }
  // column means (complete: the chunk holds every row), FLOPS partials, non-zero counts
  float fq = 0.f, fp = 0.f, fn = 0.f, zq = 0.f, zp = 0.f;
  for (int c = tid; c < cw; c += 256) {
    float cq = 0.f, cp = 0.f, cn = 0.f;
    for (int r = 0; r < d.B; ++r) {
      const float a = sq[r * LD + c], b = sp[(d.label_off + r) * LD + c];
      cq += a; cp += b;
      zq += a > 0.f ? 1.f : 0.f;
      zp += b > 0.f ? 1.f : 0.f;
    }
    for (int r = 0; r < d.B * d.k; ++r) cn += n[(long)r * d.V + c0 + c];
    cq /= (float)d.B; cp /= (float)d.B; cn /= (float)(d.B * d.k);
    w.mean_q[c0 + c] = cq; w.mean_p[c0 + c] = cp; w.mean_n[c0 + c] = cn;
    fq += cq * cq; fp += cp * cp; fn += cn * cn;
  }
  fq = block_sum(fq, red); fp = block_sum(fp, red); fn = block_sum(fn, red);
  zq = block_sum(zq, red); zp = block_sum(zp, red);
  if (tid == 0) {
    float* o = w.part_sc + (long)blockIdx.x * 8;
    o[0] = fq; o[1] = fp; o[2] = fn; o[3] = zq; o[4] = zp;
  }
}

// sum the per-chunk partial dot products: one thread per output, coalesced across threads
__global__ void loss_sum_partials_kernel(LossDims d, LossWs w) {
  const int n1 = d.B * d.Bp, n2 = d.B * d.k, n3 = d.B;
  const int o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= n1 + n2 + n3) return;
  const float* src;
  long stride;
  int idx;
  if (o < n1) { src = w.part_inb; stride = n1; idx = o; }
  else if (o < n1 + n2) { src = w.part_hard; stride = n2; idx = o - n1; }
  else { src = w.part_pos; stride = n3; idx = o - n1 - n2; }
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int c = 0;
  for (; c + 3 < d.nchunk; c += 4) {
    s0 += src[(long)c * stride + idx];
    s1 += src[(long)(c + 1) * stride + idx];
    s2 += src[(long)(c + 2) * stride + idx];
    s3 += src[(long)(c + 3) * stride + idx];
  }
  for (; c < d.nchunk; ++c) s0 += src[(long)c * stride + idx];
  w.red[o] = (s0 + s1) + (s2 + s3);
}

// out[0]=loss [1]=infonce [2]=flops_q [3]=flops_d [4]=flops_neg [5]=margin_mse [6]=nonzero_q [7]=nonzero_d [8]=kd
__global__ __launch_bounds__(256) void loss_reduce_kernel(LossDims d, LossWs w, const float* __restrict__ tpos,
                                                          const float* __restrict__ tneg,
                                                          const float* __restrict__ tsc, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  __shared__ float red[4];
  const int tid = threadIdx.x;
  const int NS = d.Bp + d.k;                // columns of the score matrix
  float* sc = sm;                           // [B][NS]
  float* hard_raw = sm + d.B * NS;          // [B][k]  (un-tempered q.n)
  float* posd = hard_raw + d.B * d.k;       // [B]
  for (int pr = tid; pr < d.B * d.Bp; pr += 256) {
    float s = w.red[pr];
    if (d.bf16_mm) s = rbf(s);
    sc[(pr / d.Bp) * NS + (pr % d.Bp)] = s * d.inv_tau;
  }
  for (int pr = tid; pr < d.B * d.k; pr += 256) {
    const float s = w.red[d.B * d.Bp + pr];
    hard_raw[pr] = s;
    sc[(pr / d.k) * NS + d.Bp + (pr % d.k)] = s * d.inv_tau;
  }
  for (int i = tid; i < d.B; i += 256) posd[i] = w.red[d.B * d.Bp + d.B * d.k + i];
  float sca[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  for (int c = tid; c < d.nchunk; c += 256)
    for (int e = 0; e < 5; ++e) sca[e] += w.part_sc[(long)c * 8 + e];
  float tot[5];
  for (int e = 0; e < 5; ++e) tot[e] = block_sum(sca[e], red);
  __syncthreads();
  // cross entropy, one row per thread; soft-max gradient coefficients
  float ce = 0.f, mm = 0.f, kd = 0.f;
  const bool use_kd = d.lam_kd > 0.f && tsc;
  for (int i = tid; i < d.B; i += 256) {
    if (use_kd) {
      // KL(teacher || student), batchmean (ref:losses.py:241-253): student = (q . own positives) / T_kd -- the SAME in-batch
      // dots as InfoNCE's (bf16-rounded under autocast: the reference's second torch.mm rounds alike) --, teacher =
      // softmax(teacher_scores / T_kd).  d kd / d s_ij = (softmax(s)_ij - teacher_ij) / B.
      const float* dots = w.red + (long)i * d.Bp + d.label_off;
      const float* tr = tsc + (long)i * d.B;
      float ms = -INFINITY, mt = -INFINITY;
      for (int j = 0; j < d.B; ++j) {
        const float sv = (d.bf16_mm ? rbf(dots[j]) : dots[j]) * d.inv_tkd;
        ms = fmaxf(ms, sv);
        mt = fmaxf(mt, tr[j] * d.inv_tkd);
      }
      float es = 0.f, et = 0.f;
      for (int j = 0; j < d.B; ++j) {
        es += expf((d.bf16_mm ? rbf(dots[j]) : dots[j]) * d.inv_tkd - ms);
        et += expf(tr[j] * d.inv_tkd - mt);
      }
      const float lse_s = ms + logf(es), lse_t = mt + logf(et);
      const float gk = d.lam_kd * d.inv_tkd / (float)d.B;
      for (int j = 0; j < d.B; ++j) {
        const float ls = (d.bf16_mm ? rbf(dots[j]) : dots[j]) * d.inv_tkd - lse_s;
        const float lt = tr[j] * d.inv_tkd - lse_t;
        const float pt = expf(lt);
        if (pt > 0.f) kd += pt * (lt - ls);                  // xlogy: a zero teacher probability contributes nothing
        float gv = (expf(ls) - pt) * gk;
        if (d.bf16_mm) gv = rbf(gv);
        w.Gkd[(long)i * d.B + j] = gv;
      }
    }
    float* row = sc + i * NS;
    float mx = row[0];
    for (int j = 1; j < NS; ++j) mx = fmaxf(mx, row[j]);
    float se = 0.f;
    for (int j = 0; j < NS; ++j) se += expf(row[j] - mx);
    const float lse = mx + logf(se);
    const int lab = d.label_off + i;
    ce += lse - row[lab];
    const float gscale = d.inv_tau / (float)d.B;
    for (int j = 0; j < d.Bp; ++j) {
      float gv = (expf(row[j] - lse) - (j == lab ? 1.f : 0.f)) * gscale;
      if (d.bf16_mm) gv = rbf(gv);
      w.G[(long)i * d.Bp + j] = gv;
    }
    float dps = 0.f;
    for (int kk = 0; kk < d.k; ++kk) {
      float gh = expf(row[d.Bp + kk] - lse) * gscale;
      if (d.lam_mm > 0.f && tpos && tneg) {
        const float smg = posd[i] - hard_raw[i * d.k + kk];
        const float tmg = tpos[i] - tneg[i * d.k + kk];
        const float df = smg - tmg;
        mm += df * df;
        const float dsm = d.lam_mm * 2.f * df / (float)(d.B * d.k);
        dps += dsm;
        gh -= dsm;
      }
      w.Gh[(long)i * d.k + kk] = gh;
    }
    w.dpos[i] = dps;
  }
  ce = block_sum(ce, red);
  mm = block_sum(mm, red);
  kd = block_sum(kd, red);
  if (tid == 0) {
    const float infonce = ce / (float)d.B;
    const float mmse = mm / (float)(d.B * d.k);
    const float kdl = kd / (float)d.B;
    out[1] = infonce; out[2] = tot[0]; out[3] = tot[1]; out[4] = tot[2]; out[5] = mmse;
    out[6] = tot[3] / (float)d.B; out[7] = tot[4] / (float)d.B; out[8] = use_kd ? kdl : 0.f;
    out[0] = infonce + d.lam_q * tot[0] + d.lam_d * tot[1] + d.lam_neg * tot[2] + (use_kd ? d.lam_kd * kdl : 0.f) +
             ((d.lam_mm > 0.f && tpos && tneg) ? d.lam_mm * mmse : 0.f);
  }
}

// gradients of the three inputs; gout = dL/dloss (device scalar)
__global__ __launch_bounds__(256) void loss_bwd_kernel(const float* __restrict__ q, const float* __restrict__ p,
                                                       const float* __restrict__ n, const float* __restrict__ gout,
                                                       LossDims d, LossWs w, int use_kd, float* __restrict__ dq,
                                                       float* __restrict__ dp, float* __restrict__ dn) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int LD = d.CH + 1;
  float* sq = sm;                       // [B][LD]
  float* sp = sm + d.B * LD;            // [Bp][LD]
  const int c0 = blockIdx.x * d.CH;
  const int cw = min(d.CH, d.V - c0);
  const int tid = threadIdx.x;
  const float go = gout[0];
  for (int i = tid; i < d.B * d.CH; i += 256) {
    const int r = i / d.CH, c = i % d.CH;
    sq[r * LD + c] = c < cw ? q[(long)r * d.V + c0 + c] : 0.f;
  }
  for (int i = tid; i < d.Bp * d.CH; i += 256) {
    const int r = i / d.CH, c = i % d.CH;
    sp[r * LD + c] = c < cw ? p[(long)r * d.V + c0 + c] : 0.f;
  }
  __syncthreads();
  const float fq = 2.f * d.lam_q / (float)d.B, fd = 2.f * d.lam_d / (float)d.B;
  const float fn = 2.f * d.lam_neg / (float)(d.B * d.k);
  // A wave owns four output rows at a time (wave-uniform: the coefficients G come through the scalar cache) and its
  // lanes the columns of the chunk: one LDS read feeds four rows.  Every output sums its terms in the order of the
  // one-output-per-thread loops this replaces (same bits).
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // dq
  for (int i0 = wave * 4; i0 < d.B; i0 += 16) {
    for (int c = lane; c < cw; c += 64) {
      float s[4] = {0.f, 0.f, 0.f, 0.f};
      for (int j = 0; j < d.Bp; ++j) {
        const float pv = d.bf16_mm ? rbf(sp[j * LD + c]) : sp[j * LD + c];
#pragma unroll
        for (int u = 0; u < 4; ++u) s[u] += w.G[(long)min(i0 + u, d.B - 1) * d.Bp + j] * pv;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u;
        if (i >= d.B) break;
        float r = d.bf16_mm ? rbf(s[u]) : s[u];
        if (use_kd) {                                        // the KD term's own mm backward (a second bf16 product under autocast)
          float s2 = 0.f;
          for (int j = 0; j < d.B; ++j) {
            const float pv = d.bf16_mm ? rbf(sp[(d.label_off + j) * LD + c]) : sp[(d.label_off + j) * LD + c];
            s2 += w.Gkd[(long)i * d.B + j] * pv;
          }
          r += d.bf16_mm ? rbf(s2) : s2;
        }
        for (int kk = 0; kk < d.k; ++kk) r += w.Gh[i * d.k + kk] * n[(long)(i * d.k + kk) * d.V + c0 + c];
        r += w.dpos[i] * sp[(d.label_off + i) * LD + c] + fq * w.mean_q[c0 + c];
        dq[(long)i * d.V + c0 + c] = go * r;
      }
    }
  }
  // dp (all Bp rows: remote rows receive only the in-batch term)
  for (int j0 = wave * 4; j0 < d.Bp; j0 += 16) {
    for (int c = lane; c < cw; c += 64) {
      float s[4] = {0.f, 0.f, 0.f, 0.f};
      for (int i = 0; i < d.B; ++i) {
        const float qv = d.bf16_mm ? rbf(sq[i * LD + c]) : sq[i * LD + c];
#pragma unroll
        for (int u = 0; u < 4; ++u) s[u] += w.G[(long)i * d.Bp + min(j0 + u, d.Bp - 1)] * qv;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = j0 + u;
        if (j >= d.Bp) break;
        float r = d.bf16_mm ? rbf(s[u]) : s[u];
        const int il = j - d.label_off;
        if (use_kd && il >= 0 && il < d.B) {
          float s2 = 0.f;
          for (int i = 0; i < d.B; ++i) s2 += w.Gkd[(long)i * d.B + il] * (d.bf16_mm ? rbf(sq[i * LD + c]) : sq[i * LD + c]);
          r += d.bf16_mm ? rbf(s2) : s2;
        }
        if (il >= 0 && il < d.B) r += w.dpos[il] * sq[il * LD + c] + fd * w.mean_p[c0 + c];
        dp[(long)j * d.V + c0 + c] = go * r;
      }
    }
  }
  // dn
  for (int o = tid; o < d.B * d.k * d.CH; o += 256) {
    const int r = o / d.CH, c = o % d.CH;
    if (c >= cw) continue;
    const int i = r / d.k;
    dn[(long)r * d.V + c0 + c] = go * (w.Gh[r] * sq[i * LD + c] + fn * w.mean_n[c0 + c]);
  }
}

static int make_dims(LossDims& d, int B, int Bp, int k, int V, int label_off, int bf16_mm, const float* hp) {
  if (B <= 0 || Bp < B || k <= 0 || V <= 0 || label_off < 0 || label_off + B > Bp) return SNX_E_SHAPE;
  if (B > 256 || Bp > 1024 || k > 16) return SNX_E_SHAPE;
  d.B = B; d.Bp = Bp; d.k = k; d.V = V; d.label_off = label_off; d.bf16_mm = bf16_mm;
  d.CH = pick_ch(B, Bp);
  d.nchunk = cdiv(V, d.CH);
  d.inv_tau = 1.0f / hp[0]; d.lam_q = hp[1]; d.lam_d = hp[2]; d.lam_neg = hp[3]; d.lam_mm = hp[4];
  d.lam_kd = hp[5]; d.inv_tkd = hp[6] > 0.f ? 1.0f / hp[6] : 1.0f;
  if ((size_t)(B * (Bp + k) + B * k + B) * 4 > 150 * 1024) return SNX_E_SHAPE;
  return SNX_OK;
}

extern "C" size_t snx_loss_workspace_bytes(int32_t B, int32_t Bp, int32_t k, int32_t V) {
  if (B <= 0 || Bp < B || k <= 0 || V <= 0) return 0;
  const int ch = pick_ch(B, Bp);
  return ws_floats(B, Bp, k, V, cdiv(V, ch)) * sizeof(float);
}

// hp = {temperature, lambda_q(t), lambda_d(t), lambda_neg(t), lambda_margin_mse, lambda_kd, kd_temperature}  (host floats)
// dims = {B, Bp, k, V, label_off, bf16_mm}
extern "C" int snx_loss_fwd(const float* q, const float* p, const float* n, const float* tpos, const float* tneg,
                            const float* tscores, const float* hp, const int32_t* dims, void* workspace, float* out9,
                            hipStream_t st) {
  if (!q || !p || !n || !hp || !dims || !workspace || !out9) return SNX_E_ARG;
  float* out8 = out9;
  LossDims d;
  int rc = make_dims(d, dims[0], dims[1], dims[2], dims[3], dims[4], dims[5], hp);
  if (rc) return rc;
  LossWs w = carve((float*)workspace, d.B, d.Bp, d.k, d.V, d.nchunk);
  const size_t lds1 = (size_t)(d.B + d.Bp) * (d.CH + 1) * 4;
  hipLaunchKernelGGL(loss_partial_kernel, dim3(d.nchunk), dim3(256), lds1, st, q, p, n, d, w);
  SNX_CHECK_LAUNCH();
  hipLaunchKernelGGL(loss_sum_partials_kernel, dim3(cdiv(d.B * d.Bp + d.B * d.k + d.B, 256)), dim3(256), 0, st, d, w);
  SNX_CHECK_LAUNCH();
  const size_t lds2 = (size_t)(d.B * (d.Bp + d.k) + d.B * d.k + d.B) * 4;
  hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(256), lds2, st, d, w, tpos, tneg, tscores, out8);
  SNX_CHECK_LAUNCH();
  return SNX_OK;
}

// use_kd: the forward was given teacher scores with lambda_kd > 0 (its coefficients are in the workspace)
extern "C" int snx_loss_bwd(const float* q, const float* p, const float* n, const float* gout, const float* hp,
                            const int32_t* dims, void* workspace, int32_t use_kd, float* dq, float* dp, float* dn,
                            hipStream_t st) {
  if (!q || !p || !n || !gout || !hp || !dims || !workspace || !dq || !dp || !dn) return SNX_E_ARG;
  LossDims d;
  int rc = make_dims(d, dims[0], dims[1], dims[2], dims[3], dims[4], dims[5], hp);
  if (rc) return rc;
  LossWs w = carve((float*)workspace, d.B, d.Bp, d.k, d.V, d.nchunk);
  const size_t lds = (size_t)(d.B + d.Bp) * (d.CH + 1) * 4;
  hipLaunchKernelGGL(loss_bwd_kernel, dim3(d.nchunk), dim3(256), lds, st, q, p, n, gout, d, w,
                     (use_kd && d.lam_kd > 0.f) ? 1 : 0, dq, dp, dn);
  SNX_CHECK_LAUNCH();
  return SNX_OK;
}
