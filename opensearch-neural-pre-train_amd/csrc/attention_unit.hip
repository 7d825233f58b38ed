// Sequence-resident attention kernels: the fast path for sequences of at most 256 tokens (the
// q64 / d256 training shapes).  Same math, cast points and fragment layouts as the streaming kernels
// in attention.hip (see the header comment there); what changes is the data movement:
//
//   streaming : one workgroup per 64-row tile; K/V (or Q/dO) re-staged tile by tile through LDS by
//               every workgroup of the sequence, two __syncthreads per tile.  PMC: waves parked in
//               s_waitcnt / s_barrier 54 % of their cycles, VALU 43 % and MFMA 15 % busy.
//   resident  : a (sequence, head) unit gets ntu x 2 waves (ntu = its number of 64-row tiles); its whole K and V
//               (or Q and dO) -- ntu x 8 KiB each -- are loaded ONCE, one barrier, then every wave
//               walks the key (query) tiles of its two 16-row groups straight out of LDS with no
//               further synchronisation; the band of a sliding-window layer is resolved per 16 rows,
//               not per 64-row tile.  Two workgroups fit a CU (2 x 64 KiB LDS, <= 128 VGPRs at 4
//               waves per SIMD), so one's load phase hides behind the other's math.
// One LDS image per tensor: the v_off swizzle serves the row-fragment ds_read_b128 AND the
// ds_read_b64_tr_b16 transposed fragments without bank conflicts.
#include <type_traits>

#include "attention_common.h"
#include "config.h"
#include "snx.h"

namespace {

constexpr int TILE_BYTES = 64 * 128;
#define SNX_ATTN_UNIT_GROUPS 8

constexpr int NTMAX = 4;                       // tiles per workgroup: 256 rows of K and V (or Q and dO) = 64 KiB
constexpr int IMG_BYTES = 2 * NTMAX * TILE_BYTES;

// One launch serves every sequence group of <= 256 tokens: a workgroup holds NTMAX / ntu units
// ((sequence, head) pairs of ntu 64-row tiles each), e.g. one 256-token document or four 64-token queries,
// each unit with its own waves and its own slice of the LDS images.  Longest group first.
struct UnitSched {
  int n;
  int seq0[SNX_ATTN_UNIT_GROUPS], units[SNX_ATTN_UNIT_GROUPS], ntu[SNX_ATTN_UNIT_GROUPS];
  int bend[SNX_ATTN_UNIT_GROUPS];              // exclusive prefix end of the group's workgroups
  int interleave;                              // 1: the groups' workgroups interleaved in proportion (attention_common.h)
};

struct Slot {
  int head, s0, slen, ntu;
  int wpu, lw, lt, slot;                       // waves per unit, wave / thread index inside the unit, unit slot
  bool live;
};
// RG = 16-row groups per wave (waves per unit = ntu * 4 / RG)
template <int RG>
__device__ __forceinline__ Slot slot_of_block(const UnitSched& sc, const int32_t* __restrict__ cu_seqlens, int heads) {
  int g, bidx;
  block_to_group<SNX_ATTN_UNIT_GROUPS>(sc.bend, sc.n, sc.interleave, (int)blockIdx.x, g, bidx);
  Slot t;
  t.ntu = sc.ntu[g];
  const int upb = NTMAX / t.ntu;
  t.wpu = t.ntu * 4 / RG;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: keeps the unit bookkeeping in SGPRs
  t.slot = wave / t.wpu;
  t.lw = wave - t.slot * t.wpu;
  t.lt = (int)threadIdx.x - t.slot * t.wpu * 64;
  const int unit = bidx * upb + t.slot;
  t.live = t.slot < upb && unit < sc.units[g];
  const int seq = sc.seq0[g] + (t.live ? unit / heads : 0);
  t.head = t.live ? unit % heads : 0;
  t.s0 = cu_seqlens[seq];
  const int n = cu_seqlens[seq + 1] - t.s0;
  t.slen = n < t.ntu * 64 ? n : t.ntu * 64;    // contract: the group's max_len covers its sequences
  return t;
}

// Load rows [0, ntu*64) x 64 d of one tensor (row stride rs elements) into the unit's v_off image; rows past
// the sequence repeat its last row (their products are masked).  2*RG 16-B pieces per thread.
template <int RG, bool NT = false>
__device__ __forceinline__ void load_image(const bf16_t* __restrict__ base, long rs, const Slot& t, char* img) {
  const int nthr = t.wpu * 64;
  bf16x8 v[2 * RG];
#pragma unroll
  for (int i = 0; i < 2 * RG; ++i) {
    const int id = t.lt + i * nthr;
    const int r = id >> 3, c = id & 7;
    const int gr = r < t.slen ? r : t.slen - 1;
    v[i] = NT ? __builtin_nontemporal_load((const bf16x8*)(base + (long)gr * rs + c * 8))
              : *(const bf16x8*)(base + (long)gr * rs + c * 8);
  }
#pragma unroll
  for (int i = 0; i < 2 * RG; ++i) {
    const int id = t.lt + i * nthr;
    *(bf16x8*)(img + v_off(id >> 3, id & 7)) = v[i];
  }
}

// key validity (inside the sequence and not masked) per key + "all 64 valid" per tile, for the unit
__device__ __forceinline__ void load_valid(const int64_t* __restrict__ mask, const Slot& t, unsigned char* sValid,
                                           int* sAll) {
  if (t.lt < t.ntu * 64) {                     // whole waves: ntu * 64 and the unit's first thread are multiples of 64
    const int key = t.lt;
    const bool v = key < t.slen && mask[t.s0 + key] != 0;
    sValid[key] = v ? 1 : 0;
    const unsigned long long all = __ballot(v);
    if ((t.lt & 63) == 0) sAll[t.lt >> 6] = (all == ~0ull) ? 1 : 0;
  }
}

// tile range [lo, hi] a 16-row wave needs (whole sequence on global layers, the band on local ones)
__device__ __forceinline__ void tile_range(int window, int row_lo, int slen, int& lo, int& hi) {
  lo = 0;
  hi = (slen - 1) >> 6;
  if (window >= 0) {
    const int a = row_lo - window, b = row_lo + 15 + window;
    lo = a > 0 ? (a >> 6) : 0;
    hi = (b < slen - 1 ? b : slen - 1) >> 6;
  }
}

// ------------------------------------------------------------------------------------------ forward
// NTL ("stream_nt" bit 32 ... see config.h: here the forward's reads of q, k, v, whose next reader is the backward)
template <bool NTL>
__global__ __launch_bounds__(NTMAX * 128, 4) void attn_fwd_unit_kernel(const bf16_t* __restrict__ qkv,
                                                                       const int32_t* __restrict__ cu_seqlens,
                                                                       const int64_t* __restrict__ mask,
                                                                       bf16_t* __restrict__ out, float* __restrict__ lse,
                                                                       int T, int heads, int window, float scale,
                                                                       const UnitSched sched) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const Slot t = slot_of_block<2>(sched, cu_seqlens, heads);
  const int ntu = t.ntu;
  char* sK = smem + t.slot * 2 * ntu * TILE_BYTES;
  char* sV = sK + ntu * TILE_BYTES;
  unsigned char* sValid = (unsigned char*)(smem + IMG_BYTES) + t.slot * ntu * 64;
  int* sAll = (int*)(smem + IMG_BYTES + NTMAX * 64) + t.slot * ntu;
  const int s0 = t.s0, slen = t.slen, head = t.head;
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, li = lane & 15;
  const int H = heads * 64;
  const long rs = 3L * H;
  const bf16_t* qbase = qkv + (long)s0 * rs + head * 64;
  // Q fragments of a 16-row group (B operand: lane = query li, 8 consecutive d); rows past the sequence read its last row
  auto load_q = [&](int rg, bf16x8 (&q)[2]) __attribute__((always_inline)) {
    const int qp = rg * 16 + li;
    const int qrow = qp < slen ? qp : (slen > 0 ? slen - 1 : 0);
#pragma unroll
    for (int c = 0; c < 2; ++c)
      q[c] = NTL ? __builtin_nontemporal_load((const bf16x8*)(qbase + (long)qrow * rs + c * 32 + g * 8))
                 : *(const bf16x8*)(qbase + (long)qrow * rs + c * 32 + g * 8);
  };
  // ONE round trip to memory before the barrier: the key mask and the first row group's Q fragments are requested ahead of
  // the K / V rows (as three trips -- images, then the mask, then Q after the barrier -- each cost every workgroup a
  // memory latency that nothing else on the CU covered)
  bf16x8 qf[2];
  if (t.live) {
    const int key = t.lt < slen ? t.lt : (slen > 0 ? slen - 1 : 0);
    // (asm: as a plain load hipcc sinks it below the image writes, next to its use, and waits there)
    int64_t mv;
    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(mv) : "v"(mask + s0 + key) : "memory");
    load_q(t.lw, qf);
    load_image<2, NTL>(qbase + H, rs, t, sK);
    load_image<2, NTL>(qbase + 2 * H, rs, t, sV);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(mv));           // the oldest request: long there after the image writes
    if (t.lt < ntu * 64) {                     // whole waves: ntu * 64 and the unit's first thread are multiples of 64
      const bool v = t.lt < slen && mv != 0;
      sValid[t.lt] = v ? 1 : 0;
      const unsigned long long all = __ballot(v);
      if ((t.lt & 63) == 0) sAll[t.lt >> 6] = (all == ~0ull) ? 1 : 0;
    }
  }
  __syncthreads();
  if (!t.live) return;
#pragma unroll 1
  for (int rg = t.lw; rg < ntu * 4; rg += t.wpu) {          // this wave's 16-row groups
  const int row_lo = rg * 16;
  if (row_lo >= slen) break;
  const int qpos = row_lo + li;
  bf16x8 qn[2];
  load_q(rg + t.wpu, qn);                                   // the next group's, under this group's tiles

  int j_lo, j_hi;
  tile_range(window, row_lo, slen, j_lo, j_hi);
  f32x4 o[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) o[d] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m_run = NEG_BIG, l_run = 0.f;
  const float c2 = scale * LOG2E;
  const int weff = window >= 0 ? window : (1 << 20);    // global layers: a band that never cuts
  const unsigned w2 = 2u * (unsigned)weff;
  for (int j = j_lo; j <= j_hi; ++j) {
    const int key0 = j * 64;
    const char* tK = sK + j * TILE_BYTES;
    const char* tV = sV + j * TILE_BYTES;
    f32x4 s[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      s[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const bf16x8 kf = *(const bf16x8*)(tK + v_off(kt * 16 + li, 4 * c + g));
        s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[c], s[kt], 0, 0, 0);
      }
    }
    const bool all_valid = sAll[j] != 0;
    const int mode = !all_valid ? 2 : band_clean(window, row_lo, row_lo + 15, key0, key0 + 63) ? 0 : 1;
    const int ub = qpos - key0 - g * 4 + weff;          // (unsigned)(ub - (16 kt + r)) <= 2 weff  <=>  in band
    float mx = NEG_BIG;
    if (mode == 0) {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          s[kt][r] *= c2;
          mx = fmaxf(mx, s[kt][r]);
        }
    } else if (mode == 1) {                             // all keys are tokens: only the band cuts
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = (unsigned)(ub - (kt * 16 + r)) <= w2 ? s[kt][r] * c2 : NEG_BIG;
          s[kt][r] = v;
          mx = fmaxf(mx, v);
        }
    } else {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        const uint32_t vm = *(const uint32_t*)(sValid + key0 + kt * 16 + g * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool ok = ((vm >> (8 * r)) & 1) && (unsigned)(ub - (kt * 16 + r)) <= w2;
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
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int c = 0; c < 2; ++c)
        o[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr(tV, d, c, lane), pb[c], o[d], 0, 0, 0);
  }
  if (qpos < slen) {
    const float inv = 1.0f / l_run;
    bf16_t* orow = out + (long)(s0 + qpos) * H + head * 64 + g * 4;
#pragma unroll
    for (int d = 0; d < 4; ++d)
      *(bf16x4*)(orow + d * 16) = (bf16x4){f2bf(o[d][0] * inv), f2bf(o[d][1] * inv), f2bf(o[d][2] * inv), f2bf(o[d][3] * inv)};
    if (g == 0) lse[(long)head * T + s0 + qpos] = (m_run + __log2f(l_run)) * LN2;   // natural-log LSE
  }
  qf[0] = qn[0]; qf[1] = qn[1];
  }
}

#ifdef SNX_ATTN_TRACE
// diagnostics build (-DSNX_ATTN_TRACE, tools/gpu_attn_trace.py): shader-clock stamps of wave 0 of every workgroup of
// the dQ kernel: entry, images loaded (after the barrier), row groups done, exit -- plus the constant-rate clock
__device__ unsigned long long* g_attn_trace = nullptr;
extern "C" int snx_attn_trace_set(void* buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_attn_trace), &buf, sizeof(buf)); }
#define ATRACE(k) if (threadIdx.x == 0 && g_attn_trace) g_attn_trace[8l * blockIdx.x + (k)] = __builtin_amdgcn_s_memtime()
#define ATRACE_RT(k) if (threadIdx.x == 0 && g_attn_trace) g_attn_trace[8l * blockIdx.x + (k)] = __builtin_amdgcn_s_memrealtime()
#else
#define ATRACE(k)
#define ATRACE_RT(k)
#endif
// --------------------------------------------------------------------------------------- backward dQ
__global__ __launch_bounds__(NTMAX * 128, 4) void attn_bwd_dq_unit_kernel(
    const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ out, const bf16_t* __restrict__ dout,
    const float* __restrict__ lse, float* __restrict__ delta, const int32_t* __restrict__ cu_seqlens,
    const int64_t* __restrict__ mask, bf16_t* __restrict__ dqkv, const f32x2* __restrict__ rope_tab,
    const int32_t* __restrict__ pos, int T, int heads, int window, float scale, const UnitSched sched) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  ATRACE(0); ATRACE_RT(4);
  const Slot t = slot_of_block<2>(sched, cu_seqlens, heads);
  const int ntu = t.ntu;
  char* sK = smem + t.slot * 2 * ntu * TILE_BYTES;
  char* sV = sK + ntu * TILE_BYTES;
  unsigned char* sValid = (unsigned char*)(smem + IMG_BYTES) + t.slot * ntu * 64;
  int* sAll = (int*)(smem + IMG_BYTES + NTMAX * 64) + t.slot * ntu;
  const int s0 = t.s0, slen = t.slen, head = t.head;
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, li = lane & 15;
  const int H = heads * 64;
  const long rs = 3L * H;
  const bf16_t* qbase = qkv + (long)s0 * rs + head * 64;
  if (t.live) {
    load_image<2>(qbase + H, rs, t, sK);
    load_image<2>(qbase + 2 * H, rs, t, sV);
    load_valid(mask, t, sValid, sAll);
  }
  __syncthreads();
  ATRACE(1);
  if (!t.live) return;
#pragma unroll 1
  for (int rg = t.lw; rg < ntu * 4; rg += t.wpu) {          // this wave's 16-row groups
  const int row_lo = rg * 16;
  if (row_lo >= slen) break;
  const int qpos = row_lo + li;
  const int qrow = qpos < slen ? qpos : slen - 1;
  bf16x8 qf[2], dof[2];
  float dl_q = 0.f;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    qf[c] = *(const bf16x8*)(qbase + (long)qrow * rs + c * 32 + g * 8);
    dof[c] = *(const bf16x8*)(dout + (long)(s0 + qrow) * H + head * 64 + c * 32 + g * 8);
    const bf16x8 of = *(const bf16x8*)(out + (long)(s0 + qrow) * H + head * 64 + c * 32 + g * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) dl_q += bf2f(of[e]) * bf2f(dof[c][e]);
  }
  // delta_q = sum_d dO[q,d] * O[q,d]; also written out for the dK/dV pass that follows on the same stream
  dl_q += __shfl_xor(dl_q, 16, 64);
  dl_q += __shfl_xor(dl_q, 32, 64);
  if (g == 0 && qpos < slen) delta[(long)head * T + s0 + qpos] = dl_q;
  const float lse2_q = lse[(long)head * T + s0 + qrow] * LOG2E;
  const int rope_p = pos ? pos[s0 + qrow] : 0;

  int j_lo, j_hi;
  tile_range(window, row_lo, slen, j_lo, j_hi);
  const float c2 = scale * LOG2E;
  const int weff = window >= 0 ? window : (1 << 20);    // global layers: a band that never cuts
  const unsigned w2 = 2u * (unsigned)weff;
  f32x4 dq[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) dq[d] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int j = j_lo; j <= j_hi; ++j) {
    const int key0 = j * 64;
    const char* tK = sK + j * TILE_BYTES;
    const char* tV = sV + j * TILE_BYTES;
    // The tile body exists three times so that no copy branches inside: MODE 0 = every (query, key) pair counts,
    // MODE 1 = all 64 keys are real tokens, only the band |q - k| <= window cuts (one unsigned compare per
    // element: the edge tiles of every sliding-window layer), MODE 2 = band and key validity.
    const bool all_valid = sAll[j] != 0;
    const int mode = !all_valid ? 2 : band_clean(window, row_lo, row_lo + 15, key0, key0 + 63) ? 0 : 1;
    const int ub = qpos - key0 - g * 4 + weff;          // (unsigned)(ub - (16 kt + r)) <= 2 weff  <=>  in band
    auto tile = [&](auto mode_tag) {
      constexpr int MODE = decltype(mode_tag)::value;
      bf16x8 dsb[2];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f}, dp = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const bf16x8 kf = *(const bf16x8*)(tK + v_off(kt * 16 + li, 4 * c + g));
          const bf16x8 vf = *(const bf16x8*)(tV + v_off(kt * 16 + li, 4 * c + g));
          s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[c], s, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, dof[c], dp, 0, 0, 0);
        }
        uint32_t vm = 0x01010101u;
        if (MODE == 2) vm = *(const uint32_t*)(sValid + key0 + kt * 16 + g * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float p = fast_exp2(fmaf(s[r], c2, -lse2_q));
          if (MODE >= 1) {
            bool ok = (unsigned)(ub - (kt * 16 + r)) <= w2;
            if (MODE == 2) ok = ok && ((vm >> (8 * r)) & 1);
            p = ok ? p : 0.f;
          }
          dsb[kt >> 1][(kt & 1) * 4 + r] = f2bf(p * (dp[r] - dl_q));
        }
      }
#pragma unroll
      for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int c = 0; c < 2; ++c)
          dq[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr(tK, d, c, lane), dsb[c], dq[d], 0, 0, 0);
    };
    if (mode == 0) tile(std::integral_constant<int, 0>{});
    else if (mode == 1) tile(std::integral_constant<int, 1>{});
    else tile(std::integral_constant<int, 2>{});
  }
  if (qpos < slen) {
    bf16_t* orow = dqkv + (long)(s0 + qpos) * rs + head * 64 + g * 4;
    store_grad_rows(orow, dq, scale, rope_tab, rope_p, g);
  }
  ATRACE(2 + (rg != t.lw));
  }
  ATRACE_RT(5);
}

// ------------------------------------------------------------------------------------ backward dK, dV
__global__ __launch_bounds__(NTMAX * 128, 4) void attn_bwd_dkv_unit_kernel(
    const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout, const float* __restrict__ lse,
    const float* __restrict__ delta, const int32_t* __restrict__ cu_seqlens, const int64_t* __restrict__ mask,
    bf16_t* __restrict__ dqkv, const f32x2* __restrict__ rope_tab, const int32_t* __restrict__ pos, int T,
    int heads, int window, float scale, const UnitSched sched) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const Slot t = slot_of_block<2>(sched, cu_seqlens, heads);
  const int ntu = t.ntu;
  char* sQ = smem + t.slot * 2 * ntu * TILE_BYTES;
  char* sO = sQ + ntu * TILE_BYTES;
  float* sLse = (float*)(smem + IMG_BYTES) + t.slot * ntu * 64;
  float* sDel = (float*)(smem + IMG_BYTES) + NTMAX * 64 + t.slot * ntu * 64;
  const int s0 = t.s0, slen = t.slen, head = t.head;
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, li = lane & 15;
  const int H = heads * 64;
  const long rs = 3L * H;
  const bf16_t* qbase = qkv + (long)s0 * rs + head * 64;
  if (t.live) {
    load_image<2>(qbase, rs, t, sQ);
    load_image<2>(dout + (long)s0 * H + head * 64, H, t, sO);
    if (t.lt < ntu * 64) {
      const int qc = t.lt < slen ? t.lt : slen - 1;
      sLse[t.lt] = lse[(long)head * T + s0 + qc] * LOG2E;       // log2 domain
      sDel[t.lt] = delta[(long)head * T + s0 + qc];
    }
  }
  __syncthreads();
  if (!t.live) return;
#pragma unroll 1
  for (int rg = t.lw; rg < ntu * 4; rg += t.wpu) {          // this wave's 16-row groups
  const int row_lo = rg * 16;
  if (row_lo >= slen) break;
  const int kpos = row_lo + li;
  const int krow = kpos < slen ? kpos : slen - 1;
  const bool kvalid = kpos < slen && mask[s0 + krow] != 0;
  const bool wave_keys_valid = __ballot(kvalid) == ~0ull;
  bf16x8 kf[2], vf[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    kf[c] = *(const bf16x8*)(qbase + H + (long)krow * rs + c * 32 + g * 8);
    vf[c] = *(const bf16x8*)(qbase + 2 * H + (long)krow * rs + c * 32 + g * 8);
  }
  const int rope_p = pos ? pos[s0 + krow] : 0;

  int i_lo, i_hi;
  tile_range(window, row_lo, slen, i_lo, i_hi);
  const float c2 = scale * LOG2E;
  const int weff = window >= 0 ? window : (1 << 20);    // global layers: a band that never cuts
  const unsigned w2 = 2u * (unsigned)weff;
  f32x4 dk[4], dv[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    dk[d] = (f32x4){0.f, 0.f, 0.f, 0.f};
    dv[d] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  for (int i = i_lo; i <= i_hi; ++i) {
    const int q0 = i * 64;
    const char* tQ = sQ + i * TILE_BYTES;
    const char* tO = sO + i * TILE_BYTES;
    // no masking needed when all 64 queries exist, this wave's 16 keys are all valid and in band
    const bool simple = (q0 + 63 < slen) && wave_keys_valid;        // every (query, key) pair of the tile exists
    const bool clean = simple && band_clean(window, q0, q0 + 63, row_lo, row_lo + 15);
    const int ub = q0 + g * 4 - kpos + weff;                        // (unsigned)(ub + 16 qt + r) <= 2 weff  <=>  in band
    // two halves of 32 queries: each half's P / dS (one bf16x8 per lane) feeds its 32-deep dV / dK MFMAs at once
#pragma unroll
    for (int hq = 0; hq < 2; ++hq) {
      bf16x8 pb, dsb;
#pragma unroll
      for (int q2 = 0; q2 < 2; ++q2) {
        const int qt = 2 * hq + q2;
        f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f}, dp = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const bf16x8 qfr = *(const bf16x8*)(tQ + v_off(qt * 16 + li, 4 * c + g));
          const bf16x8 ofr = *(const bf16x8*)(tO + v_off(qt * 16 + li, 4 * c + g));
          s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qfr, kf[c], s, 0, 0, 0);     // S[q][key]
          dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ofr, vf[c], dp, 0, 0, 0);   // dP[q][key]
        }
        const f32x4 l4 = *(const f32x4*)(sLse + q0 + qt * 16 + g * 4);
        const f32x4 d4 = *(const f32x4*)(sDel + q0 + qt * 16 + g * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float p = fast_exp2(fmaf(s[r], c2, -l4[r]));
          if (!clean) {
            bool ok = (unsigned)(ub + (qt * 16 + r)) <= w2;                  // |q - key| <= window
            if (!simple) ok = ok && kvalid && (q0 + qt * 16 + g * 4 + r) < slen;
            p = ok ? p : 0.f;
          }
          pb[q2 * 4 + r] = f2bf(p);
          dsb[q2 * 4 + r] = f2bf(p * (dp[r] - d4[r]));
        }
      }
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        dv[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr(tO, d, hq, lane), pb, dv[d], 0, 0, 0);
        dk[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr(tQ, d, hq, lane), dsb, dk[d], 0, 0, 0);
      }
    }
  }
  if (kpos < slen) {
    bf16_t* krow_out = dqkv + (long)(s0 + kpos) * rs + H + head * 64 + g * 4;
    store_grad_rows(krow_out, dk, scale, rope_tab, rope_p, g);
    store_grad_rows(krow_out + H, dv, 1.0f, nullptr, 0, g);
  }
  }
}

constexpr size_t UNIT_LDS = IMG_BYTES + NTMAX * 64 * 8 + 64;

// groups = {n, (seq_begin, nseq, max_len) x n}, every max_len <= 256; longest first
int build_unit_sched(UnitSched& sc, int& blocks, const int32_t* groups, int heads) {
  if (groups[0] < 1 || groups[0] > SNX_ATTN_UNIT_GROUPS) return SNX_E_ARG;
  int order[SNX_ATTN_UNIT_GROUPS];
  for (int i = 0; i < groups[0]; ++i) order[i] = i;
  for (int i = 1; i < groups[0]; ++i)
    for (int j = i; j > 0 && groups[3 + 3 * order[j]] > groups[3 + 3 * order[j - 1]]; --j) {
      const int tmp = order[j]; order[j] = order[j - 1]; order[j - 1] = tmp;
    }
  sc.n = groups[0];
  long b = 0;
  for (int i = 0; i < SNX_ATTN_UNIT_GROUPS; ++i) {
    sc.seq0[i] = 0; sc.units[i] = 0; sc.ntu[i] = 1;
    if (i < groups[0]) {
      const int32_t* e = groups + 1 + 3 * order[i];
      const int ntu = cdiv(e[2], 64);
      if (ntu < 1 || ntu > NTMAX || e[1] <= 0) return SNX_E_SHAPE;
      sc.seq0[i] = e[0]; sc.units[i] = e[1] * heads; sc.ntu[i] = ntu;
      b += cdiv(sc.units[i], NTMAX / ntu);
    }
    if (b > 0x7fffffffL) return SNX_E_SHAPE;
    sc.bend[i] = (int)b;
  }
  blocks = (int)b;
  sc.interleave = g_snx_cfg.attn_interleave != 0 && sc.n > 1;
  return SNX_OK;
}

}  // namespace

int attn_unit_fwd(const bf16_t* qkv, const int32_t* cu_seqlens, const int64_t* mask, bf16_t* out, float* lse, int T,
                  int heads, int window, const int32_t* groups, hipStream_t st) {
  UnitSched sc;
  int blocks;
  const int rc = build_unit_sched(sc, blocks, groups, heads);
  if (rc != SNX_OK) return rc;
  static LdsOptIn optin[2];
  const bool ntl = (g_snx_cfg.stream_nt & 64) != 0;
  auto kern = ntl ? attn_fwd_unit_kernel<true> : attn_fwd_unit_kernel<false>;
  if (const int rc2 = optin[ntl ? 1 : 0].ensure((const void*)kern, (int)UNIT_LDS)) return rc2;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(NTMAX * 128), UNIT_LDS, st, qkv, cu_seqlens, mask, out,
                     lse, T, heads, window, 0.125f, sc);
  SNX_CHECK_LAUNCH();
  return SNX_OK;
}

int attn_bwd_onepass(const bf16_t* qkv, const bf16_t* out, const bf16_t* dout, const float* lse,
                     const int32_t* cu_seqlens, const int64_t* mask, bf16_t* dqkv, const f32x2* rope_tab,
                     const int32_t* pos, int T, int heads, int window, const int32_t* groups, hipStream_t st);

// 1 (default): the one-pass kernel of attention_1p.hip; 0: the dQ + dK/dV pair below (A/B and second opinion in tests)
extern "C" int snx_attn_configure(int32_t bwd_onepass) { return snx_configure("attn_bwd_onepass", bwd_onepass); }

int attn_unit_bwd(const bf16_t* qkv, const bf16_t* out, const bf16_t* dout, const float* lse, float* delta,
                  const int32_t* cu_seqlens, const int64_t* mask, bf16_t* dqkv, const f32x2* rope_tab,
                  const int32_t* pos, int T, int heads, int window, const int32_t* groups, hipStream_t st) {
  if (g_snx_cfg.attn_bwd_onepass)
    return attn_bwd_onepass(qkv, out, dout, lse, cu_seqlens, mask, dqkv, rope_tab, pos, T, heads, window, groups, st);
  UnitSched sc;
  int blocks;
  const int rc = build_unit_sched(sc, blocks, groups, heads);
  if (rc != SNX_OK) return rc;
  static LdsOptIn optin_dq, optin_dkv;
  if (const int rc2 = optin_dq.ensure((const void*)attn_bwd_dq_unit_kernel, (int)UNIT_LDS)) return rc2;
  if (const int rc2 = optin_dkv.ensure((const void*)attn_bwd_dkv_unit_kernel, (int)UNIT_LDS)) return rc2;
  hipLaunchKernelGGL(attn_bwd_dq_unit_kernel, dim3(blocks), dim3(NTMAX * 128), UNIT_LDS, st, qkv, out, dout, lse,
                     delta, cu_seqlens, mask, dqkv, rope_tab, pos, T, heads, window, 0.125f, sc);
  SNX_CHECK_LAUNCH();
  hipLaunchKernelGGL(attn_bwd_dkv_unit_kernel, dim3(blocks), dim3(NTMAX * 128), UNIT_LDS, st, qkv, dout, lse, delta,
                     cu_seqlens, mask, dqkv, rope_tab, pos, T, heads, window, 0.125f, sc);
  SNX_CHECK_LAUNCH();
  return SNX_OK;
}
