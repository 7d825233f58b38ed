// EXPERIMENT (round 4), not part of libsnx.so: attention forward for sequences of at most 256 tokens on
// v_mfma_f32_32x32x16_bf16, as a persistent kernel.  Correct (it passed tests/test_gpu_ops.py -k attention as the
// library's forward), but not faster than the 16x16 kernel of csrc/attention_unit.hip on the training shapes
// (fused 64 x q64 + 128 x d256, 12 heads: 62.7 / 51.0 us global / window-64 against 55-57 / 47-48 us), so it was not adopted.
// Build and time it with tools/gpu_fwd32_probe.py; the measurements and what they showed are in DESIGN.md section 4
// ("attention forward on the 32x32 MFMA") and profiles/r04_experiments.txt.
//
//   workgroup  = 8 waves = one per CU, walking items (a 256-token document, or four 64-token queries) grid apart; the item's
//                Q, K and V rows resident in LDS in PLANE form (one image serves row fragments and transposed fragments
//                without bank conflicts); the NEXT item's rows are requested into registers a few per key tile inside the
//                tile loop and written to LDS after the barrier that ends the item.
//   wave       = 32 queries on the 32 LANE columns: S^T = K Q^T per 32-key tile (keys on the accumulator rows), so the
//                running maximum, the rescale factor and the row sum are per-lane scalars, and P^T -- packed to bf16 -- IS
//                the B operand of O^T += V^T P^T (accumulator rows = contraction index: no lane movement, no LDS round
//                trip).  Lazy rescale (reference moves only when a tile's maximum exceeds it by 2^8), S of the next tile
//                requested before this tile's softmax.
//   output     = staged per wave in LDS, stored as whole 128-byte rows one item later.
// What bounds it: vector-instruction ISSUE.  A clean 32 x 32 tile costs a wave about 580 issue cycles (16 v_exp at 8, about
// 100 other vector instructions at 4, 8 MFMAs holding the issue port 8 each, 12 LDS reads) against 256 cycles of matrix
// core: the 32x32 form halves the MFMA count, which was never the limit, and leaves the per-score vector work unchanged.
#include "attention_common.h"
#include "config.h"
#include "snx.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int MAXK = 256;
constexpr int PS = MAXK * 32 + 128;            // plane stride of a [256 rows][64 d] image
constexpr int IMG = 4 * PS;
constexpr int OFF_K = 0;
constexpr int OFF_V = IMG;
constexpr int OFF_Q = 2 * IMG;
constexpr int OFF_VALID = 3 * IMG;             // key validity: one bit per image row
constexpr int OFF_O = OFF_VALID + 64;          // per wave: its 32 x 64 output tile on the way out
constexpr int LDS_F32 = OFF_O + 8 * 4096;      // 132,672 B: one workgroup of two waves per SIMD per CU
constexpr int NTMAX = 4;

#define SNX_ATTN_F32T_GROUPS 8
struct SchedF {
  int n;
  int seq0[SNX_ATTN_F32T_GROUPS], units[SNX_ATTN_F32T_GROUPS], ntu[SNX_ATTN_F32T_GROUPS];
  int bend[SNX_ATTN_F32T_GROUPS];              // exclusive prefix end of the group's workgroups
};

// Inside a plane a row is 32 bytes; rows 8..15 of every 16 swap their two 4-row groups and their two 16-byte halves:
// ds_read_b128 row fragments (lane: row l & 31, half l >> 5) and ds_read_b64_tr_b16 blocks (4 rows x 16 columns per
// 16-lane group, rows r and r + 8 in one instruction) are conflict-free (the image of attention_1p.hip).
__device__ __forceinline__ int p_row(int row) { return row ^ (((row >> 3) & 1) << 2); }
__device__ __forceinline__ int p_off(int row, int half) { return p_row(row) * 32 + ((half ^ ((row >> 3) & 1)) << 4); }

// single-instruction forms (plain -O3 packs neighbouring f32 multiplies / adds into v_pk_*_f32, which issue slower beside
// MFMAs than the scalar forms, and canonicalises MFMA outputs with an extra v_max before fmaxf)
__device__ __forceinline__ float max3(float a, float b, float c) {
  float d;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ float fma1(float a, float b, float c) {
  float d;
  asm("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

#ifdef SNX_ATTN_TRACE
// stamps stay in registers until the kernel ends (a store per stamp would put vmcnt waits into the code being timed)
__device__ long long* g_f32_trace;
#define FTRACE(k) tr[k] = __builtin_amdgcn_s_memtime()
#define FTRACE_RT(k) tr[k] = __builtin_amdgcn_s_memrealtime()
#define FTRACE_IT(k) do { if (it == 1) FTRACE(k); } while (0)
#define WTRACE(k) do { if (it == 1) wt[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define FTRACE(k)
#define FTRACE_RT(k)
#define FTRACE_IT(k)
#define WTRACE(k)
#endif

// One wave's view of an item (a workgroup's worth of units): all wave-uniform
struct Unit {
  int ntu, lw, rbase0, s0, slen, head, q0;
  bool live, wave_on;
  const bf16_t* qbase;
};

__global__ __launch_bounds__(512, 2) void attn_fwd32_kernel(const bf16_t* __restrict__ qkv,
                                                            const int32_t* __restrict__ cu_seqlens,
                                                            const int64_t* __restrict__ mask, bf16_t* __restrict__ out,
                                                            float* __restrict__ lse, int T, int heads, int window,
                                                            float scale, int nitems, const SchedF sched) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef SNX_ATTN_TRACE
  long long tr[16] = {};
  long long wt[4] = {};
#endif
  FTRACE(0); FTRACE_RT(12);
  const int H = heads * 64;
  const long rs = 3L * H;
  char* sK = smem + OFF_K;
  char* sV = smem + OFF_V;
  char* sQ = smem + OFF_Q;
  uint32_t* sBits = (uint32_t*)(smem + OFF_VALID);
  const int r = lane & 31, h = lane >> 5;
  const int G = lane >> 4, li = lane & 15;
  const int rb = p_off(r, h);                                // row fragment: row (32 n +) r, half h
  const int x4 = 4 * (G >> 1) + (li >> 2);                   // transposed fragments of a 32-key tile: rows 16 s + 8 sec + x4
  const int tb0 = (G & 1) * PS + p_off(x4, (li & 3) >> 1) + (li & 1) * 8;
  const int tb1 = (G & 1) * PS + p_off(x4 + 8, (li & 3) >> 1) + (li & 1) * 8;
  // a wave's piece of a row image: 8 rows x 128 B.  Lane -> (row, 16-byte chunk) such that 8 CONTIGUOUS lanes write 4 rows x
  // 32 B of one plane = 128 contiguous bytes: ds_write_b128 works in groups of 8 lanes over 32 banks (chunk = lane & 7 would
  // put the four planes of a row, 65 x 128 B apart, on the same banks: 4-way conflicts on every image write)
  const int prow = 4 * (lane >> 5) + ((lane >> 1) & 3), pc = 2 * ((lane >> 3) & 3) + (lane & 1);
  const float c2 = scale * LOG2E;
  const int weff = window >= 0 ? window : (1 << 20);

  // ---- item -> this wave's unit
  auto decode = [&](int item, Unit& u) __attribute__((always_inline)) {
    int g = 0, b0 = 0;
#pragma unroll
    for (int i = 0; i < SNX_ATTN_F32T_GROUPS - 1; ++i)
      if (i + 1 < sched.n && item >= sched.bend[i]) { g = i + 1; b0 = sched.bend[i]; }
    u.ntu = sched.ntu[g];
    const int upb = NTMAX / u.ntu;                           // units per item
    const int wpu = 2 * u.ntu;                               // waves per unit: one per 32 queries
    const int slot = wave / wpu;
    u.lw = wave - slot * wpu;
    const int unit = (item - b0) * upb + slot;
    u.live = slot < upb && unit < sched.units[g];
    const int seq = sched.seq0[g] + (u.live ? unit / heads : 0);
    u.head = u.live ? unit % heads : 0;
    u.s0 = u.live ? cu_seqlens[seq] : 0;
    const int sl = cu_seqlens[seq + 1] - u.s0;
    u.slen = !u.live ? 0 : (sl < u.ntu * 64 ? sl : u.ntu * 64);   // contract: the group's max_len covers its sequences
    u.qbase = qkv + (long)u.s0 * rs + u.head * 64;
    u.rbase0 = slot * u.ntu * 64;                            // first image row of this unit
    u.q0 = 32 * u.lw;
    u.wave_on = u.slen > 0 && u.q0 < u.slen;                 // this wave has queries
  };
  // ---- requests: the rows of the unit as three [rows][64] tensors Q, K, V (thread -> four 16-byte pieces of each) and the
  // key mask, as 13 numbered pieces: the next item's are issued a few per key tile INSIDE this item's tile loop.  (All 13 in
  // one burst hold every wave in the issue stage for 3,500 - 8,000 cycles -- the CU's memory path takes about 14 B/clk and the
  // second wave of each SIMD queues behind the first -- and the compute of the item cannot start under them.)
  // Branch-free: rows past the sequence re-read its last row (cache hits; their keys are masked), a dead slot row 0.
  constexpr int NPIECE = 13;
  auto piece = [&](const Unit& u, int j, bf16x8 (&qv)[4], bf16x8 (&kv)[4], bf16x8 (&vv)[4], int64_t& mk) __attribute__((always_inline)) {
    const int last = u.slen > 0 ? u.slen - 1 : 0;
    const int lt = u.lw * 64 + lane, nthr = 128 * u.ntu;     // thread index inside the unit
    auto at = [&](int i, int sec) __attribute__((always_inline)) {
      const int row0 = 8 * ((lt + i * nthr) >> 6) + prow;
      return *(const bf16x8*)(u.qbase + sec * H + (long)(row0 < last ? row0 : last) * rs + pc * 8);
    };
    switch (j) {
      case 0: mk = mask[u.s0 + (lt < last ? lt : last)]; break;
      case 1: qv[0] = at(0, 0); break;
      case 2: qv[1] = at(1, 0); break;
      case 3: qv[2] = at(2, 0); break;
      case 4: qv[3] = at(3, 0); break;
      case 5: kv[0] = at(0, 1); break;
      case 6: kv[1] = at(1, 1); break;
      case 7: kv[2] = at(2, 1); break;
      case 8: kv[3] = at(3, 1); break;
      case 9: vv[0] = at(0, 2); break;
      case 10: vv[1] = at(1, 2); break;
      case 11: vv[2] = at(2, 2); break;
      default: vv[3] = at(3, 2); break;
    }
  };
  auto deposit = [&](const Unit& u, const bf16x8 (&qv)[4], const bf16x8 (&kv)[4], const bf16x8 (&vv)[4], int64_t mk) __attribute__((always_inline)) {
    if (!u.live) return;
    const int lt = u.lw * 64 + lane, nthr = 128 * u.ntu;
    if (u.lw < u.ntu) {                                      // wave lw of the unit: keys 64 lw .. 64 lw + 63
      const uint64_t bal = __ballot(mk != 0 && lt < u.slen);
      if (lane == 0) *(uint64_t*)(sBits + ((u.rbase0 + 64 * u.lw) >> 5)) = bal;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = u.rbase0 + 8 * ((lt + i * nthr) >> 6) + prow, c = pc;
      *(bf16x8*)(sQ + (c >> 1) * PS + p_off(row, c & 1)) = qv[i];
      *(bf16x8*)(sK + (c >> 1) * PS + p_off(row, c & 1)) = kv[i];
      *(bf16x8*)(sV + (c >> 1) * PS + p_off(row, c & 1)) = vv[i];
    }
  };

  // An item's output rows leave AFTER the next item's images are written (issued right behind the compute they would sit
  // between the next item's requests and their first use in the in-order vmcnt queue), and as WHOLE rows: each wave stages
  // its 32 x 64 tile in LDS ([query][8-byte piece ^ (query & 15)]: the accumulator layout writes 16 rows x one piece per
  // 16-lane group) and eight lanes store a row's 128 bytes -- 4 stores of 8 full lines instead of 8 stores of 32 sixteen-byte
  // pieces, which held the next item's requests behind them in the address unit.
  struct Pend { int rows; bf16_t* obase; float* lbase; float lse; } pend;
  pend.rows = 0;
  char* sO = smem + OFF_O + wave * 4096;
  auto flush = [&]() __attribute__((always_inline)) {
    if (pend.rows > 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = 8 * i + (lane >> 3), c = lane & 7, f = row & 15;
        const bf16x8 v = *(const bf16x8*)(sO + row * 128 + ((c ^ (f >> 1)) << 4));
        const bf16x8 w = (f & 1) ? (bf16x8){v[4], v[5], v[6], v[7], v[0], v[1], v[2], v[3]} : v;
        if (row < pend.rows) *(bf16x8*)(pend.obase + (long)row * H + c * 8) = w;
      }
      if (h == 0 && r < pend.rows) pend.lbase[r] = pend.lse;
    }
    pend.rows = 0;
  };
  int item = blockIdx.x;
  Unit u;
  bf16x8 qv[4], kv[4], vv[4];
  int64_t mk;
  decode(item, u);
#pragma unroll
  for (int j = 0; j < NPIECE; ++j) piece(u, j, qv, kv, vv, mk);
#pragma unroll 1
  while (true) {
    const int it = (item - (int)blockIdx.x) / (int)gridDim.x;
    deposit(u, qv, kv, vv, mk);
    FTRACE_IT(3);
    __syncthreads();
    FTRACE_IT(4); WTRACE(0);
    flush();
    FTRACE_IT(5);
    // the next item's rows travel while this one is computed
    const int nitem = item + (int)gridDim.x;
    const bool more = nitem < nitems;
    Unit un;
    // Q fragments of this wave's 32 queries (B operand: lane = query, 8 consecutive d per k-step half)
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const bf16x8*)(sQ + ks * PS + (u.rbase0 + u.q0) * 32 + rb);
    if (more) decode(nitem, un);
    FTRACE_IT(6); WTRACE(1);
    if (u.wave_on) {
      const int slen = u.slen, q0 = u.q0, qpos = q0 + r, rbase0 = u.rbase0;
      const int klo = q0 - weff > 0 ? q0 - weff : 0;
      const int khi = q0 + 31 + weff < slen - 1 ? q0 + 31 + weff : slen - 1;
      f32x16 o[2];                                           // O^T: (d = 32 dt + row) x (query = lane & 31)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[dt][e] = 0.f;
      // Softmax in the log2 domain against a REFERENCE m_ref that follows the running maximum lazily: it moves (and O, l
      // are rescaled) only when a tile's maximum exceeds it by more than 2^8 -- exact in real arithmetic (O and l carry
      // the same factor), P <= 256 rounds to bf16 with the same relative error, and the 32-multiply rescale leaves the
      // steady state.
      float m_ref = NEG_BIG, l_run = 0.f;                    // l_run: this lane HALF's part of the row sum
      // S^T of the NEXT tile is requested from the matrix core before this tile's softmax: its four MFMAs run under the
      // ~100 vector instructions of the softmax (two waves per SIMD cannot hide that chain by themselves)
      auto qk = [&](int kt) __attribute__((always_inline)) {
        const char* tK = sK + (rbase0 + 32 * kt) * 32;
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) acc = mfma32(*(const bf16x8*)(tK + ks * PS + rb), qf[ks], acc);   // S^T[key][query]
        return acc;
      };
      const int kt1 = khi >> 5;
      f32x16 s = qk(klo >> 5);
      // eight STAGES, fully unrolled (a wave has at most 8 key tiles): stage st requests pieces 2 st and 2 st + 1 of the next
      // item -- at fixed places in the code, into fixed registers -- and computes the wave's tile st if it has one
#pragma unroll
      for (int st = 0; st < MAXK / 32; ++st) {
        if (more) {
          piece(un, 2 * st, qv, kv, vv, mk);
          if (2 * st + 1 < NPIECE) piece(un, 2 * st + 1, qv, kv, vv, mk);
        }
        const int kt = (klo >> 5) + st;
        if (kt > kt1) continue;
        const int key0 = 32 * kt;
        const char* tV = sV + (rbase0 + key0) * 32;
        const f32x16 sn = qk(kt < kt1 ? kt + 1 : kt1);       // (the last tile once more: branch-free)
        bf16x8 vf[2][2];                                     // transposed V fragments: ahead of the softmax as well
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            const int oo = 2 * dt * PS + 16 * s2 * 32;
            const bf16x4 a0 = lds_tr16(tV + oo + tb0), a1 = lds_tr16(tV + oo + tb1);
            vf[dt][s2] = (bf16x8){a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
          }
        // element e of a lane: key key0 + c + 4 h with c = (e & 3) + 8 (e >> 2), query qpos
        const uint32_t vword = __builtin_amdgcn_readfirstlane(sBits[(rbase0 + key0) >> 5]);      // bit k: key key0 + k valid
        const bool clean = vword == 0xffffffffu && band_clean(window, q0, q0 + 31, key0, key0 + 31);
        if (!clean) {
          const int hi = qpos - key0 - 4 * h + weff, lo = hi - 2 * weff;   // lo <= c <= hi  <=>  |q - key| <= window
          const uint32_t m_hi = hi < 0 ? 0u : (hi >= 31 ? ~0u : (2u << hi) - 1u);
          const uint32_t m_lo = lo <= 0 ? ~0u : (lo >= 32 ? 0u : ~((1u << lo) - 1u));
          const uint32_t okb = (vword >> (4 * h)) & m_hi & m_lo;
#pragma unroll
          for (int e = 0; e < 16; ++e) s[e] = ((okb >> ((e & 3) + 8 * (e >> 2))) & 1u) ? s[e] : NEG_BIG;
        }
        float mx = max3(s[0], s[1], s[2]);
#pragma unroll
        for (int e = 3; e < 15; e += 2) mx = max3(mx, s[e], s[e + 1]);
        mx = fmaxf(mx, s[15]);
        // (the other 16 keys of the tile; log2 domain.  The floor keeps the reference of a row that has seen masked keys only
        // far above NEG_BIG c2, so that its probabilities are exp2(-1.8e29) = 0 and not exp2 of a rounding difference)
        mx = fmaxf(fmaxf(mx, __shfl_xor(mx, 32, 64)), -1.0e20f) * c2;
        if (__any(mx > m_ref + 8.0f)) {
          const float m_new = mx > m_ref + 8.0f ? mx : m_ref;
          const float alpha = fast_exp2(m_ref - m_new);
          m_ref = m_new;
          l_run *= alpha;
#pragma unroll
          for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int e = 0; e < 16; ++e) o[dt][e] *= alpha;
        }
        const float negm = -m_ref;
        float rsum = 0.f;
        bf16x8 pb[2];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float p = fast_exp2(fma1(s[e], c2, negm));
          rsum += p;
          pb[e >> 3][e & 7] = f2bf(p);
        }
        l_run += rsum;
        // O^T += V^T P^T: A = transposed V fragment (MFMA row = d = 32 dt + (lane & 31), plane 2 dt + (G & 1)); element j
        // of lane half h is tile key 16 s + 8 (j >> 2) + 4 h + (j & 3), the key the packed accumulator element j holds
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) o[dt] = mfma32(vf[dt][s2], pb[s2], o[dt]);
        s = sn;
      }
      FTRACE_IT(7); WTRACE(2);
      l_run += __shfl_xor(l_run, 32, 64);
      // the stores wait for the next pass of the loop (see `Pend`)
      const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;     // a row without a visible key: zeros, LSE = NEG_BIG
      pend.rows = slen - q0 < 32 ? slen - q0 : 32;
      pend.obase = out + (long)(u.s0 + q0) * H + u.head * 64;
      pend.lbase = lse + (long)u.head * T + u.s0 + q0;
      pend.lse = l_run > 0.f ? (m_ref + __log2f(l_run)) * LN2 : NEG_BIG;   // natural-log LSE
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4)                         // element e of o[dt]: d = 32 dt + 4 h + (e & 3) + 8 (e >> 2)
          *(bf16x4*)(sO + r * 128 + (((8 * dt + 2 * i4 + h) ^ (r & 15)) << 3)) =
              (bf16x4){f2bf(o[dt][4 * i4] * inv), f2bf(o[dt][4 * i4 + 1] * inv), f2bf(o[dt][4 * i4 + 2] * inv),
                       f2bf(o[dt][4 * i4 + 3] * inv)};
    }
    if (!u.wave_on && more) {                                // (a wave without queries)
#pragma unroll
      for (int j = 0; j < NPIECE; ++j) piece(un, j, qv, kv, vv, mk);
    }
    FTRACE_IT(8); WTRACE(3);
    if (it == 0) FTRACE(1);
    if (!more) break;
    __syncthreads();                                         // every wave is done with the images
    if (it == 0) FTRACE(2);
    item = nitem;
    u = un;
  }
  flush();
  FTRACE(14); FTRACE_RT(13);
#ifdef SNX_ATTN_TRACE
  if (threadIdx.x == 0 && g_f32_trace)
    for (int k = 0; k < 16; ++k) g_f32_trace[48l * blockIdx.x + k] = tr[k];
  if (lane == 0 && g_f32_trace)
    for (int k = 0; k < 4; ++k) g_f32_trace[48l * blockIdx.x + 16 + 4 * wave + k] = wt[k];
#endif
}

}  // namespace

#ifdef SNX_ATTN_TRACE
extern "C" int snx_attn_fwd32_trace_set(void* buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_f32_trace), &buf, sizeof(buf)); }
#endif

// groups = {n, (seq_begin, nseq, max_len) x n}, every max_len <= 256; longest first
int attn_fwd_tile32(const bf16_t* qkv, const int32_t* cu_seqlens, const int64_t* mask, bf16_t* out, float* lse, int T,
                    int heads, int window, const int32_t* groups, hipStream_t st) {
  if (groups[0] < 1 || groups[0] > SNX_ATTN_F32T_GROUPS) return SNX_E_ARG;
  int order[SNX_ATTN_F32T_GROUPS];
  for (int i = 0; i < groups[0]; ++i) order[i] = i;
  for (int i = 1; i < groups[0]; ++i)
    for (int j = i; j > 0 && groups[3 + 3 * order[j]] > groups[3 + 3 * order[j - 1]]; --j) {
      const int tmp = order[j]; order[j] = order[j - 1]; order[j - 1] = tmp;
    }
  SchedF sc;
  sc.n = groups[0];
  long b = 0;
  for (int i = 0; i < SNX_ATTN_F32T_GROUPS; ++i) {
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
  static bool once = false;
  if (!once) {
    (void)hipFuncSetAttribute((const void*)attn_fwd32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_F32);
    once = true;
  }
  const int cus = 256 - snx_get_reserved_cus();            // one resident workgroup per CU walks the items, grid apart
  const int grid = b < cus ? (int)b : cus;
  hipLaunchKernelGGL(attn_fwd32_kernel, dim3((unsigned)grid), dim3(512), LDS_F32, st, qkv, cu_seqlens, mask, out, lse, T, heads,
                     window, 0.125f, (int)b, sc);
  SNX_CHECK_LAUNCH();
  return SNX_OK;
}
