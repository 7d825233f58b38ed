// One-pass attention backward for sequences of at most 256 tokens (the q64 / d256 training shapes).
// Replaces autograd's backward of torch SDPA as called from transformers modeling_modernbert.py:286-297
// (masks: masking_utils.py:141-150); same cast points as the two-pass kernels of attention_unit.hip
// (fp32 scores, P and dS rounded to bf16 as MFMA operands, fp32 accumulation, bf16 gradients).
//
// The two-pass form (a dQ kernel and a dK/dV kernel) computes S = Q K^T and dP = dO V^T twice and reads
// q, k, v and dO twice from HBM: 7 contractions and ~680 MB per layer where 5 contractions and ~450 MB are
// needed -- at head_dim 64 and <= 256 keys the pass is bound by HBM (146 FLOP per byte), not by the MFMAs.
// Here every (query, key) score is formed ONCE:
//
//   workgroup  = 8 waves = one (sequence, head) unit; wave w owns keys 32 w .. 32 w + 31 and keeps their
//                dK^T and dV^T ([64 d] x [32 keys] each) in 64 accumulator registers for the whole unit.
//                (Four waves of 64 keys, two workgroups per CU, was the first form: 128 accumulator + 32 V
//                registers leave hipcc ~20 short of 256, and a reload from scratch inside the slice loop waits
//                for every load in flight.)
//   K          resident in LDS (one image for row reads and transposed reads); V as MFMA operand fragments in
//                16 registers; Q and dO stream through LDS in 64-row slices, the next slice's global loads in
//                flight (registers) while the current one is multiplied.
//   phase 1    per 32-row half of the slice: S and dP on v_mfma_f32_32x32x16_bf16 with the KEY on the lane and the
//                row constants (-lse / scale, -delta) as the initial accumulators, so that p = exp2(c S') and
//                dS = p dP' need no further row operand; P and dS, packed to bf16, ARE the B operands of
//                dV^T += dO^T P and dK^T += Q^T dS (accumulator rows = contraction index: no lane movement).
//                dS also goes to LDS once, as [key][query] rows of 128 bytes.
//   phase 2    dQ of the slice: the contraction over ALL keys is done inside one wave (the eight waves split
//                the 64 x 64 OUTPUT: 16 queries x two 16-wide d tiles (d, d + 32: a RoPE pair) each), so dQ
//                needs no sum across waves, no atomics and no second kernel.
//   band       sliding-window layers: a wave skips the (half slice, its keys) pairs outside |q - k| <= window
//                and phase 2 contracts over the band's 32-key blocks only.
#include <type_traits>

#include "attention_common.h"
#include "config.h"
#include "snx.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int QS = 64;                         // query rows per slice (phase 1 works on its two 32-row halves)
constexpr int MAXK = 256;                      // keys per unit
// LDS images of [rows][64 d] bf16 tiles in PLANE form: plane ks holds the 32-byte pieces d = 16 ks .. 16 ks + 15 of every
// row (the k-step of a 32x32x16 fragment is then an immediate offset, one address register serves all four), planes
// 128 bytes apart modulo 256 so that the two 16-lane groups of a transposed read use different halves of the banks.
constexpr int PS_K = MAXK * 32 + 128;          // plane stride of the K image
constexpr int PS_S = QS * 32 + 128;            // plane stride of a Q / dO slice image
constexpr int OFF_K = 0;
constexpr int OFF_Q = OFF_K + 4 * PS_K;
constexpr int OFF_O = OFF_Q + 4 * PS_S;
constexpr int OFF_DS = OFF_O + 4 * PS_S;       // [256 keys][64 q] bf16, 128-B rows
constexpr int OFF_LSE = OFF_DS + MAXK * 128;   // -lse / scale per query
constexpr int OFF_DEL = OFF_LSE + MAXK * 4;    // -delta per query
constexpr int OFF_POS = OFF_DEL + MAXK * 4;    // RoPE position per token
constexpr int LDS_1P = OFF_POS + MAXK * 4;     // 86,528 B: one 8-wave workgroup per CU
static_assert(8 * 8192 <= OFF_LSE, "epilogue staging (8 KiB per wave) overlays the K, slice and dS images only");

#define SNX_ATTN_1P_GROUPS 8
struct Sched1p {
  int n;
  int seq0[SNX_ATTN_1P_GROUPS], bend[SNX_ATTN_1P_GROUPS];   // first sequence, exclusive prefix end of the group's units
  int interleave;                                           // 1: the groups' units interleaved in proportion (round 6)
};

// Inside a plane a row is 32 bytes; rows 8..15 of every 16 swap their two 4-row groups and their two 16-byte halves,
// which makes the ds_read_b128 row fragments (lane: row l & 31, half l >> 5) AND the ds_read_b64_tr_b16 blocks (4 rows x
// 16 columns per 16-lane group, rows r and r + 8 in one instruction) conflict-free under the gfx950 bank rules.
__device__ __forceinline__ int p_row(int row) { return row ^ (((row >> 3) & 1) << 2); }
__device__ __forceinline__ int p_off(int row, int half) { return p_row(row) * 32 + ((half ^ ((row >> 3) & 1)) << 4); }
// dS image: [key] rows of 64 queries = sixteen 8-byte pieces; piece index XOR a 4-bit code of the key whose high
// half comes from key bits 1 and 3 (the transposed reads of phase 2 take keys k, k + 2, k + 8, k + 10 in one bank half)
// and which differs for 16 consecutive keys (the writes of phase 1: one key per lane).
__device__ __forceinline__ int ds_code(int key) {
  return (((key >> 1) & 1) << 2) | (((key >> 3) & 1) << 3) | (key & 1) | (((key >> 2) & 1) << 1);
}
__device__ __forceinline__ int ds_off(int key, int piece) { return key * 128 + ((piece ^ ds_code(key)) << 3); }

__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// LDS writes retired (and the compiler's memory operations kept on their side), then the workgroup barrier; global
// loads and stores stay in flight across it
#define WG_BARRIER()                                          \
  do {                                                        \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        \
    __builtin_amdgcn_sched_barrier(0);                        \
    __builtin_amdgcn_s_barrier();                             \
    __builtin_amdgcn_sched_barrier(0);                        \
    asm volatile("" ::: "memory");                            \
  } while (0)

#ifdef SNX_ATTN_TRACE
// diagnostics build (-DSNX_ATTN_TRACE, tools/gpu_attn_trace.py): shader-clock stamps of wave 0 of every workgroup:
// [0] entry, [1] slice loop reached, per slice 0 / 1: [2 + 4 s ..] first barrier passed, phase 1 done, second barrier
// passed, phase 2 done; [10] loop left, [11] exit; [12] / [13] constant-rate clock at entry / exit
__device__ unsigned long long* g_attn1p_trace = nullptr;
#define ATRACE(k) do { if (threadIdx.x == 0 && g_attn1p_trace) g_attn1p_trace[16l * blockIdx.x + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#define ATRACE_RT(k) do { if (threadIdx.x == 0 && g_attn1p_trace) g_attn1p_trace[16l * blockIdx.x + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define ATRACE(k)
#define ATRACE_RT(k)
#endif

// NTL ("stream_nt" bit 32): q, k, v and dO are read for the last time here (the attention output O is read again by the
// layer's weight-gradient GEMM): non-temporal loads keep them from displacing the following GEMMs' operands in the caches.
template <bool NTL>
__global__ __launch_bounds__(512, 2) void attn_bwd_1p_kernel(
    const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ out, const bf16_t* __restrict__ dout,
    const float* __restrict__ lse, const int32_t* __restrict__ cu_seqlens, const int64_t* __restrict__ mask,
    bf16_t* __restrict__ dqkv, const f32x2* __restrict__ rope_tab, const int32_t* __restrict__ pos, int T, int heads,
    int window, float scale, const Sched1p sched) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  auto ld8 = [](const bf16_t* p) __attribute__((always_inline)) {
    return NTL ? __builtin_nontemporal_load((const bf16x8*)p) : *(const bf16x8*)p;
  };
  ATRACE(0); ATRACE_RT(12);
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  int g, unit;
  block_to_group<SNX_ATTN_1P_GROUPS>(sched.bend, sched.n, sched.interleave, (int)blockIdx.x, g, unit);   // attention_common.h
  const int seq = sched.seq0[g] + unit / heads, head = unit % heads;
  const int s0 = cu_seqlens[seq];
  int slen = cu_seqlens[seq + 1] - s0;
  slen = slen < MAXK ? slen : MAXK;                          // contract: the group's max_len (<= 256) covers its sequences
  if (slen <= 0) return;
  const int H = heads * 64;
  const long rs = 3L * H;
  const bf16_t* qbase = qkv + (long)s0 * rs + head * 64;
  const bf16_t* dobase = dout + (long)s0 * H + head * 64;
  const bf16_t* obase = out + (long)s0 * H + head * 64;
  char* sK = smem + OFF_K;
  char* sQ = smem + OFF_Q;
  char* sO = smem + OFF_O;
  char* sDS = smem + OFF_DS;
  float* sLse = (float*)(smem + OFF_LSE);
  float* sDel = (float*)(smem + OFF_DEL);
  int* sPos = (int*)(smem + OFF_POS);
  const int nrow = ((slen + 31) >> 5) << 5;                  // rows of K / of the row constants that are ever read
  // this thread's 16-B piece of a 64-row slice (a wave: 8 rows x 128 B).  8 CONTIGUOUS lanes write 4 rows x 32 B of one plane =
  // 128 contiguous bytes: ds_write_b128 works in groups of 8 lanes over 32 banks, and chunk = lane & 7 would put the four planes of
  // a row (17 / 65 x 128 B apart) on the same banks -- a 4-way conflict on every image write
  const int srow = 8 * (tid >> 6) + 4 * ((tid >> 5) & 1) + ((tid >> 1) & 3), schunk = 2 * ((tid >> 3) & 3) + (tid & 1);
  const int sdst = (schunk >> 1) * PS_S + p_off(srow, schunk & 1);

  // ---- unit prologue.  Every global load is issued before anything waits (one round trip: with one workgroup per CU
  // nothing else covers it); rows past the sequence repeat its last row.
  const int key = 32 * w + r;                                 // this lane's key
  const int keyr = key < slen ? key : slen - 1;
  bf16x8 qreg, doreg, oreg, kv[4], vb[4];
  {
    const int gr = srow < slen ? srow : slen - 1;
    qreg = ld8(qbase + (long)gr * rs + schunk * 8);
    doreg = ld8(dobase + (long)gr * H + schunk * 8);
    oreg = *(const bf16x8*)(obase + (long)gr * H + schunk * 8);
  }
  const int nk64 = (slen + 63) >> 6;                         // 64-row blocks of K that hold keys (a 64-token query: one)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = srow + 64 * i;
    const int gr = row < slen ? row : slen - 1;
    if (i < nk64) kv[i] = ld8(qbase + H + (long)gr * rs + schunk * 8);      // workgroup-uniform
  }
  const bool has_keys = 32 * w < slen;                        // wave-uniform: this wave owns keys of the sequence
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
    if (has_keys) vb[ks] = ld8(qbase + 2 * H + (long)keyr * rs + 16 * ks + 8 * h);
  const int64_t mk = mask[s0 + keyr];
  float lse_t = 0.f;
  int pos_t = 0;
  if (tid < MAXK) {
    const int gr = tid < slen ? tid : slen - 1;
    lse_t = lse[(long)head * T + s0 + gr];
    pos_t = pos ? pos[s0 + gr] : 0;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (i < nk64) *(bf16x8*)(sK + (schunk >> 1) * PS_K + i * 64 * 32 + p_off(srow, schunk & 1)) = kv[i];
  if (tid < MAXK) {
    sLse[tid] = -lse_t / scale;                               // S' = q k - lse / scale, p = exp(scale S')
    sPos[tid] = pos_t;
  }
  const bool kval = key < slen && mk != 0;
  const bool kall = __ballot(kval) == ~0ull;

  f32x16 dv[2], dk[2];                                        // [d half]: (d = 32 dt + row) x (key = lane & 31)
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int e = 0; e < 16; ++e) { dv[b][e] = 0.f; dk[b][e] = 0.f; }

  const float c2 = scale * LOG2E;
  const int weff = window >= 0 ? window : (1 << 20);          // global layers: a band that never cuts
  const unsigned w2 = 2u * (unsigned)weff;
  const int nsl = (slen + QS - 1) / QS;
  const int G = lane >> 4, li = lane & 15;
  const int qh16 = w >> 1, wd = w & 1;                        // phase 2: this wave's 16 queries and d tiles (wd, wd + 2)
  const int kb = 32 * w;                                      // this wave's keys kb .. kb + 31
  // lane parts of the LDS addresses
  const int rbase = p_off(r, h);                              // row fragment: row (32 n +) r, half h
  const int x4 = 4 * (G >> 1) + (li >> 2);                    // transposed fragments of a half slice: rows 16 s + 8 sec + x4
  const int tbase0 = (G & 1) * PS_S + p_off(x4, (li & 3) >> 1) + (li & 1) * 8;
  const int tbase1 = (G & 1) * PS_S + p_off(x4 + 8, (li & 3) >> 1) + (li & 1) * 8;
  const int dsw = key * 128;                                  // dS row of this lane's key
  const int dsc = ds_code(key);
  // phase 2: key row 32 kk + x2 (+ 4) of this lane's transposed blocks
  const int x2 = 8 * G + (li >> 2);
  const int p2s0 = ds_off(x2, 4 * qh16 + (li & 3)), p2s1 = ds_off(x2 + 4, 4 * qh16 + (li & 3));
  const int p2k0 = wd * PS_K + p_off(x2, (li & 3) >> 1) + (li & 1) * 8;
  const int p2k1 = wd * PS_K + p_off(x2 + 4, (li & 3) >> 1) + (li & 1) * 8;
  // The V fragments arrive before the slice loop: a first use inside it would make hipcc wait for them THERE, in every
  // slice, and vmcnt waits are in order -- each would also wait for the next slice's prefetch just issued.
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(vb[ks]));
  // dQ of a slice is stored one slice LATER (after the next barrier): a store tail in front of the loop top would sit
  // between the prefetch and its consumer in the in-order vmcnt queue.
  bf16x4 pend_lo, pend_hi;
  bf16_t* pend_row = nullptr;

  ATRACE(1);
#pragma unroll 1
  for (int sl = 0; sl < nsl; ++sl) {
    const int q0 = sl * QS;
    *(bf16x8*)(sQ + sdst) = qreg;
    *(bf16x8*)(sO + sdst) = doreg;
    {
      // delta_q = sum_d dO[q, d] O[q, d] of the slice's rows: eight lanes per row, 8 elements each (the forward's output is
      // read once, with the slice, instead of 64 KiB in front of the loop)
      float dl = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) dl = fmaf(bf2f(doreg[e]), bf2f(oreg[e]), dl);
      dl += __shfl_xor(dl, 1, 64);                           // the row's eight pieces: lane bits 0, 3, 4
      dl += __shfl_xor(dl, 8, 64);
      dl += __shfl_xor(dl, 16, 64);
      if (schunk == 0) sDel[q0 + srow] = -dl;
    }
    if (sl + 1 < nsl) {                                        // next slice: in flight during this one
      const int row = q0 + QS + srow;
      const int gr = row < slen ? row : slen - 1;
      qreg = ld8(qbase + (long)gr * rs + schunk * 8);
      doreg = ld8(dobase + (long)gr * H + schunk * 8);
      oreg = *(const bf16x8*)(obase + (long)gr * H + schunk * 8);
    }
    WG_BARRIER();
    if (sl < 2) ATRACE(2 + 4 * sl);
    if (pend_row) {
      *(bf16x4*)pend_row = pend_lo;
      *(bf16x4*)(pend_row + 32) = pend_hi;
    }
    // keys any query of the slice can see: phase 2 contracts over their 32-key blocks
    const int klo = q0 - weff > 0 ? q0 - weff : 0;
    const int khi = q0 + QS - 1 + weff < slen - 1 ? q0 + QS - 1 + weff : slen - 1;
    // ------------------------------------------------------------------ phase 1: both 32-row halves in flight together
    if (!(kb > khi || kb + 31 < klo)) {                        // wave-uniform
      f32x16 sa[2], da[2];
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) {
          const f32x4 l4 = *(const f32x4*)(sLse + q0 + 32 * sub + 8 * i4 + 4 * h);
          const f32x4 d4 = *(const f32x4*)(sDel + q0 + 32 * sub + 8 * i4 + 4 * h);
#pragma unroll
          for (int j = 0; j < 4; ++j) { sa[sub][4 * i4 + j] = l4[j]; da[sub][4 * i4 + j] = d4[j]; }
        }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 kf = *(const bf16x8*)(sK + ks * PS_K + kb * 32 + rbase);
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
          const bf16x8 qa = *(const bf16x8*)(sQ + ks * PS_S + sub * 1024 + rbase);
          const bf16x8 oa = *(const bf16x8*)(sO + ks * PS_S + sub * 1024 + rbase);
          sa[sub] = mfma32(qa, kf, sa[sub]);                   // S'[q][key]
          da[sub] = mfma32(oa, vb[ks], da[sub]);               // dP'[q][key] = dO V^T - delta
        }
      }
      // element e of a lane: query qs + 4 h + (e & 3) + 8 (e >> 2), key kb + r
      bf16x8 pb[2][2], dsb[2][2];
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) {
        const int qs = q0 + 32 * sub;                          // rows qs .. qs + 31 (a half outside the band or past the
                                                               // sequence comes out as zeros through the element masks)
        const bool clean = (qs + 31 < slen) && kall && band_clean(window, qs, qs + 31, kb, kb + 31);
        if (clean) {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const float p = fast_exp2(sa[sub][e] * c2);
            pb[sub][e >> 3][e & 7] = f2bf(p);
            dsb[sub][e >> 3][e & 7] = f2bf(p * da[sub][e]);
          }
        } else {
          const int ub = qs + 4 * h - key + weff;              // (unsigned)(ub + c) <= 2 weff  <=>  |q - key| <= window
          const int qlim = slen - qs - 4 * h;                  // c < qlim  <=>  the query exists
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int c = (e & 3) + 8 * (e >> 2);
            const bool ok = kval && (unsigned)(ub + c) <= w2 && c < qlim;
            const float p = fast_exp2(sa[sub][e] * c2);
            pb[sub][e >> 3][e & 7] = f2bf(ok ? p : 0.f);
            dsb[sub][e >> 3][e & 7] = f2bf(ok ? p * da[sub][e] : 0.f);
          }
        }
        // dS to LDS: registers 4 i4 .. 4 i4 + 3 = queries 32 sub + 8 i4 + 4 h + (0..3) of this key: one 8-byte piece
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) {
          const bf16x4 v = (bf16x4){dsb[sub][i4 >> 1][(i4 & 1) * 4 + 0], dsb[sub][i4 >> 1][(i4 & 1) * 4 + 1],
                                    dsb[sub][i4 >> 1][(i4 & 1) * 4 + 2], dsb[sub][i4 >> 1][(i4 & 1) * 4 + 3]};
          *(bf16x4*)(sDS + dsw + (((8 * sub + 2 * i4 + h) ^ dsc) << 3)) = v;
        }
      }
      // transposed A fragments: MFMA row = d = 32 dt + (lane & 31) (plane 2 dt + (G & 1)); element j of lane half h is
      // half-slice row 16 s + 8 (j >> 2) + 4 h + (j & 3), the row the packed accumulator element j holds
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            const int o = 2 * dt * PS_S + (32 * sub + 16 * s) * 32;
            const bf16x4 o0 = lds_tr16(sO + o + tbase0), o1 = lds_tr16(sO + o + tbase1);
            const bf16x4 u0 = lds_tr16(sQ + o + tbase0), u1 = lds_tr16(sQ + o + tbase1);
            dv[dt] = mfma32((bf16x8){o0[0], o0[1], o0[2], o0[3], o1[0], o1[1], o1[2], o1[3]}, pb[sub][s], dv[dt]);
            dk[dt] = mfma32((bf16x8){u0[0], u0[1], u0[2], u0[3], u1[0], u1[1], u1[2], u1[3]}, dsb[sub][s], dk[dt]);
          }
      }
    }
    if (sl < 2) ATRACE(3 + 4 * sl);
    WG_BARRIER();
    if (sl < 2) ATRACE(4 + 4 * sl);
    // ------------------------------------------------------------------ phase 2: dQ^T[d][q] = sum_key K^T[d][key] dS^T[key][q]
    {
      const int qpos = q0 + 16 * qh16 + li;
      const int qrow = qpos < slen ? qpos : slen - 1;
      f32x2 cs[4];
      if (rope_tab) {
        const f32x2* ct = rope_tab + (long)sPos[qrow] * 32 + 16 * wd + 4 * G;
#pragma unroll
        for (int e = 0; e < 4; ++e) cs[e] = ct[e];
      }
      f32x4 a0 = (f32x4){0.f, 0.f, 0.f, 0.f}, a2 = (f32x4){0.f, 0.f, 0.f, 0.f};
      const int k_lo = klo >> 5, k_hi = khi >> 5;
      if (q0 + 16 * qh16 < slen) {                             // wave-uniform: this wave's 16 queries exist
        // four 32-key blocks per trip: 24 transposed reads in flight in front of 8 MFMAs (one block per trip is a chain of
        // exposed LDS round trips), single blocks for the rest
        auto blocks = [&](auto n_tag, int kk) {
          constexpr int NB = decltype(n_tag)::value;
          bf16x4 b0[NB], b1[NB], x0[NB], x1[NB], y0[NB], y1[NB];
#pragma unroll
          for (int u = 0; u < NB; ++u) {
            const char* dp = sDS + (kk + u) * 32 * 128;
            const char* kp = sK + (kk + u) * 1024;
            b0[u] = lds_tr16(dp + p2s0); b1[u] = lds_tr16(dp + p2s1);
            x0[u] = lds_tr16(kp + p2k0); x1[u] = lds_tr16(kp + p2k1);
            y0[u] = lds_tr16(kp + 2 * PS_K + p2k0); y1[u] = lds_tr16(kp + 2 * PS_K + p2k1);
          }
#pragma unroll
          for (int u = 0; u < NB; ++u) {
            const bf16x8 bfr = (bf16x8){b0[u][0], b0[u][1], b0[u][2], b0[u][3], b1[u][0], b1[u][1], b1[u][2], b1[u][3]};
            a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                (bf16x8){x0[u][0], x0[u][1], x0[u][2], x0[u][3], x1[u][0], x1[u][1], x1[u][2], x1[u][3]}, bfr, a0, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                (bf16x8){y0[u][0], y0[u][1], y0[u][2], y0[u][3], y1[u][0], y1[u][1], y1[u][2], y1[u][3]}, bfr, a2, 0, 0, 0);
          }
        };
        int kk = k_lo;
        for (; kk + 3 <= k_hi; kk += 4) blocks(std::integral_constant<int, 4>{}, kk);
        for (; kk <= k_hi; ++kk) blocks(std::integral_constant<int, 1>{}, kk);
      }
      // lane: query q0 + 16 qh16 + li, d = 16 wd + 4 G + (0..3) in a0 and d + 32 in a2
      pend_row = qpos < slen ? dqkv + (long)(s0 + qpos) * rs + head * 64 + 16 * wd + 4 * G : nullptr;
      {
        bf16x4 lo, hi;
        if (rope_tab) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float y1 = rbf(a0[e] * scale), y2 = rbf(a2[e] * scale);
            lo[e] = f2bf(y1 * cs[e][0] + y2 * cs[e][1]);
            hi[e] = f2bf(y2 * cs[e][0] - y1 * cs[e][1]);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) { lo[e] = f2bf(a0[e] * scale); hi[e] = f2bf(a2[e] * scale); }
        }
        pend_lo = lo;
        pend_hi = hi;
      }
    }
    if (sl < 2) ATRACE(5 + 4 * sl);
  }
  ATRACE(10);
  if (pend_row) {
    *(bf16x4*)pend_row = pend_lo;
    *(bf16x4*)(pend_row + 32) = pend_hi;
  }

  // ---- unit epilogue: dK (scaled, inverse RoPE) and dV of this wave's keys; element e: d = 32 dt + 4 h + (e & 3) + 8 (e >> 2).
  // Through LDS (the slice images are free now; 8 KiB per wave), so that the global stores are whole 128-byte rows in 16-byte
  // pieces: sixteen 8-byte stores per lane at a row stride are bound by store ISSUE (32-64 lines per instruction).
  WG_BARRIER();
  if (has_keys) {
    char* stg = smem + w * 8192;                               // [dK | dV][32 keys][64 d] bf16, 16-B chunk ^ (key & 7)
    f32x2 t[16];
    if (rope_tab) {
      const f32x2* cs = rope_tab + (long)sPos[keyr] * 32 + 4 * h;
#pragma unroll
      for (int i4 = 0; i4 < 4; ++i4)
#pragma unroll
        for (int j = 0; j < 4; ++j) t[4 * i4 + j] = cs[8 * i4 + j];
    }
#pragma unroll
    for (int i4 = 0; i4 < 4; ++i4) {
      bf16x4 lo, hi, v0, v1;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int e = 4 * i4 + j;
        v0[j] = f2bf(dv[0][e]);
        v1[j] = f2bf(dv[1][e]);
        if (rope_tab) {
          const float y1 = rbf(dk[0][e] * scale), y2 = rbf(dk[1][e] * scale);
          lo[j] = f2bf(y1 * t[e][0] + y2 * t[e][1]);
          hi[j] = f2bf(y2 * t[e][0] - y1 * t[e][1]);
        } else {
          lo[j] = f2bf(dk[0][e] * scale);
          hi[j] = f2bf(dk[1][e] * scale);
        }
      }
      // d = 8 i4 + 4 h (+ 32): chunk i4 (+ 4), byte 8 h inside it
      const int c_lo = ((i4 ^ (r & 7)) << 4) + 8 * h, c_hi = (((i4 + 4) ^ (r & 7)) << 4) + 8 * h;
      *(bf16x4*)(stg + r * 128 + c_lo) = lo;
      *(bf16x4*)(stg + r * 128 + c_hi) = hi;
      *(bf16x4*)(stg + 4096 + r * 128 + c_lo) = v0;
      *(bf16x4*)(stg + 4096 + r * 128 + c_hi) = v1;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // wave-private region: no barrier
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const int row = 8 * n + (lane >> 3), ch = lane & 7;
      const bf16x8 a = *(const bf16x8*)(stg + row * 128 + ((ch ^ (row & 7)) << 4));
      const bf16x8 b = *(const bf16x8*)(stg + 4096 + row * 128 + ((ch ^ (row & 7)) << 4));
      if (kb + row < slen) {
        bf16_t* dst = dqkv + (long)(s0 + kb + row) * rs + H + head * 64 + ch * 8;
        *(bf16x8*)dst = a;
        *(bf16x8*)(dst + H) = b;
      }
    }
  }
  ATRACE(11); ATRACE_RT(13);
}

}  // namespace

#ifdef SNX_ATTN_TRACE
extern "C" int snx_attn1p_trace_set(void* buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_attn1p_trace), &buf, sizeof(buf)); }
#endif

// groups = {n, (seq_begin, nseq, max_len) x n}, every max_len <= 256
int attn_bwd_onepass(const bf16_t* qkv, const bf16_t* out, const bf16_t* dout, const float* lse,
                     const int32_t* cu_seqlens, const int64_t* mask, bf16_t* dqkv, const f32x2* rope_tab,
                     const int32_t* pos, int T, int heads, int window, const int32_t* groups, hipStream_t st) {
  if (groups[0] < 1 || groups[0] > SNX_ATTN_1P_GROUPS) return SNX_E_ARG;
  int order[SNX_ATTN_1P_GROUPS];
  for (int i = 0; i < groups[0]; ++i) order[i] = i;
  for (int i = 1; i < groups[0]; ++i)                                 // longest group first
    for (int j = i; j > 0 && groups[3 + 3 * order[j]] > groups[3 + 3 * order[j - 1]]; --j) {
      const int tmp = order[j]; order[j] = order[j - 1]; order[j - 1] = tmp;
    }
  Sched1p sc;
  sc.n = groups[0];
  long b = 0;
  for (int i = 0; i < SNX_ATTN_1P_GROUPS; ++i) {
    sc.seq0[i] = 0;
    if (i < groups[0]) {
      const int32_t* e = groups + 1 + 3 * order[i];
      if (e[1] <= 0 || e[2] <= 0 || e[2] > MAXK) return SNX_E_SHAPE;
      sc.seq0[i] = e[0];
      b += (long)e[1] * heads;
    }
    if (b > 0x7fffffffL) return SNX_E_SHAPE;
    sc.bend[i] = (int)b;
  }
  sc.interleave = g_snx_cfg.attn_interleave != 0 && sc.n > 1;
  static LdsOptIn optin[2];
  const bool ntl = (g_snx_cfg.stream_nt & 32) != 0;
  auto kern = ntl ? attn_bwd_1p_kernel<true> : attn_bwd_1p_kernel<false>;
  if (const int rc = optin[ntl ? 1 : 0].ensure((const void*)kern, LDS_1P)) return rc;
  hipLaunchKernelGGL(kern, dim3((unsigned)b), dim3(512), LDS_1P, st, qkv, out, dout, lse, cu_seqlens, mask,
                     dqkv, rope_tab, pos, T, heads, window, 0.125f, sc);
  SNX_CHECK_LAUNCH();
  return SNX_OK;
}
