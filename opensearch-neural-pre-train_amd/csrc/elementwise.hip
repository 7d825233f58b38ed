// HBM-bound token-wise kernels of the encoder (K1, K2, K4, GeGLU of K7, GELU+LN of K9 and their
// backward passes).  One wave per token row, 16-byte vector accesses, all math
// in fp32 with the reference's autocast cast points (fp32 residual stream / LayerNorm / RoPE,
// bf16 GELU and GeGLU product).  References: transformers modeling_modernbert.py:64-71
// (embeddings), :89-91 (GeGLU), :196-219 (RoPE), :312-314,420,487 (LayerNorm), :489-490 (head).
#include "common.h"
#include "config.h"
#include "snx.h"

#define ROWS_PER_BLOCK 4   // 256 threads = 4 waves = 4 token rows

// ------------------------------------------------------------------------------------------
// weight cache: fp32 master -> bf16 (and bf16 transposed for the dX GEMMs)
// ------------------------------------------------------------------------------------------
__global__ void cast_bf16_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, long n) {
  long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const long stride = (long)gridDim.x * blockDim.x * 4;
  for (; i + 3 < n; i += stride) {
    const f32x4 v = *(const f32x4*)(in + i);
    bf16x4 o = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
    *(bf16x4*)(out + i) = o;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (long j = n & ~3L; j < n; ++j) out[j] = f2bf(in[j]);
}

// in [R, C] fp32 row-major -> out [C, R] bf16 row-major (64x64 tiles through LDS)
__global__ void cast_transpose_bf16_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, int R, int C) {
  __shared__ float tile[64][65];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 256 threads: 4 rows per pass
  for (int r = ty; r < 64; r += 4) {
    const int gr = r0 + r, gc = c0 + tx;
    tile[r][tx] = (gr < R && gc < C) ? in[(long)gr * C + gc] : 0.f;
  }
  __syncthreads();
  for (int c = ty; c < 64; c += 4) {
    const int gc = c0 + c, gr = r0 + tx;
    if (gc < C && gr < R) out[(long)gc * R + gr] = f2bf(tile[tx][c]);
  }
}

// GeGLU-interleaved copies of Wi [2I, C]: output row n holds source row
//   src(n) = (n%64 < 32) ? 32*(n/64) + n%32 : I + 32*(n/64) + n%32
// so that every 64-row group is [a rows 32q..32q+31 | g rows 32q..32q+31].  out [2I, C] and/or the
// transposed out_t [C, 2I] (columns in the same interleaved order).
__device__ __forceinline__ int geglu_src_row(int n, int I) {
  return ((n & 63) < 32) ? 32 * (n >> 6) + (n & 31) : I + 32 * (n >> 6) + (n & 31);
}

__global__ void cast_interleave_kernel(const float* __restrict__ in, bf16_t* __restrict__ out,
                                       bf16_t* __restrict__ out_t, int I, int C) {
  __shared__ float tile[64][65];
  const int R = 2 * I;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) {
    const int gr = r0 + r, gc = c0 + tx;
    const float v = (gr < R && gc < C) ? in[(long)geglu_src_row(gr, I) * C + gc] : 0.f;
    tile[r][tx] = v;
    if (out && gr < R && gc < C) out[(long)gr * C + gc] = f2bf(v);
  }
  __syncthreads();
  if (out_t)
    for (int c = ty; c < 64; c += 4) {
      const int gc = c0 + c, gr = r0 + tx;
      if (gc < C && gr < R) out_t[(long)gc * R + gr] = f2bf(tile[tx][c]);
    }
}

// Batched forms for the weight-cache refresh (model.hip): one launch per shape class, blockIdx.z = layer.  The cache
// refresh of the 149 M model was 157 launches of 6-12 us each behind every optimizer step; it is 7 now.
// INTERLEAVE = false: out[z] [R, C] = bf16(in[z]) and out_t[z] [C, R] = its transpose (R rows as given);
// INTERLEAVE = true : Wi [2I, C] with the GeGLU row order of cast_interleave_kernel (R = 2 I).
template <bool INTERLEAVE>
__global__ void cast_both_batched_kernel(CastBatch b, int R, int C, int I) {
  __shared__ float tile[64][65];
  const float* __restrict__ in = b.src[blockIdx.z];
  bf16_t* __restrict__ out = b.out[blockIdx.z];
  bf16_t* __restrict__ out_t = b.out_t[blockIdx.z];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) {
    const int gr = r0 + r, gc = c0 + tx;
    const float v = (gr < R && gc < C) ? in[(long)(INTERLEAVE ? geglu_src_row(gr, I) : gr) * C + gc] : 0.f;
    tile[r][tx] = v;
    if (gr < R && gc < C) out[(long)gr * C + gc] = f2bf(v);
  }
  __syncthreads();
  for (int c = ty; c < 64; c += 4) {
    const int gc = c0 + c, gr = r0 + tx;
    if (gc < C && gr < R) out_t[(long)gc * R + gr] = f2bf(tile[tx][c]);
  }
}

// interleave != 0: the tensors are Wi [2 I, C] (R = 2 I, I % 32 == 0)
int snx_cast_both_batched(const CastBatch& b, int n, int R, int C, int interleave, hipStream_t st) {
  if (n <= 0 || n > SNX_CAST_BATCH_MAX || R <= 0 || C <= 0) return SNX_E_ARG;
  if (interleave && ((R & 1) || ((R / 2) % 32))) return SNX_E_SHAPE;
  const dim3 grid(cdiv(C, 64), cdiv(R, 64), n);
  if (interleave) hipLaunchKernelGGL(cast_both_batched_kernel<true>, grid, dim3(256), 0, st, b, R, C, R / 2);
  else hipLaunchKernelGGL(cast_both_batched_kernel<false>, grid, dim3(256), 0, st, b, R, C, 0);
  SNX_CHECK_LAUNCH();
  return SNX_OK;
}

extern "C" int snx_cast_geglu_interleave(const float* in, void* out, void* out_t, int32_t I, int32_t C,
                                         hipStream_t st) {
  if (!in || (!out && !out_t) || I <= 0 || C <= 0) return SNX_E_ARG;
  if (I % 32) return SNX_E_SHAPE;
  hipLaunchKernelGGL(cast_interleave_kernel, dim3(cdiv(C, 64), cdiv(2 * I, 64)), dim3(256), 0, st, in, (bf16_t*)out,
                     (bf16_t*)out_t, I, C);
  SNX_CHECK_LAUNCH();
  return SNX_OK;
}

extern "C" int snx_cast_bf16(const float* in, void* out, int64_t n, hipStream_t st) {
  if (!in || !out || n <= 0) return SNX_E_ARG;
  if (((uintptr_t)in & 15) || ((uintptr_t)out & 7)) return SNX_E_ARG;
  int blocks = cdiv(n, 256 * 4);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(cast_bf16_kernel, dim3(blocks), dim3(256), 0, st, in, (bf16_t*)out, (long)n);
  SNX_CHECK_LAUNCH();
  return SNX_OK;
}

extern "C" int snx_cast_transpose_bf16(const float* in, void* out, int32_t R, int32_t C, hipStream_t st) {
  if (!in || !out || R <= 0 || C <= 0) return SNX_E_ARG;
  hipLaunchKernelGGL(cast_transpose_bf16_kernel, dim3(cdiv(C, 64), cdiv(R, 64)), dim3(256), 0, st, in,
                     (bf16_t*)out, R, C);
  SNX_CHECK_LAUNCH();
  return SNX_OK;
}

// ------------------------------------------------------------------------------------------
// LayerNorm (no bias), one wave per row; NV = H / 256 float4 per lane
// ------------------------------------------------------------------------------------------
template <int NV>
struct RowVec {
  f32x4 v[NV];
  __device__ __forceinline__ void load_f32(const float* p, int lane) {
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = *(const f32x4*)(p + (i * 64 + lane) * 4);
  }
  __device__ __forceinline__ void load_bf16(const bf16_t* p, int lane) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const bf16x4 b = *(const bf16x4*)(p + (i * 64 + lane) * 4);
      v[i] = (f32x4){bf2f(b[0]), bf2f(b[1]), bf2f(b[2]), bf2f(b[3])};
    }
  }
  __device__ __forceinline__ void store_f32(float* p, int lane) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) *(f32x4*)(p + (i * 64 + lane) * 4) = v[i];
  }
  // non-temporal forms ("stream_nt", config.h): rows that are read for the last time / written for a reader tens of
  // milliseconds away should not displace the next GEMM's operands from the L2 and the Infinity Cache
  __device__ __forceinline__ void load_f32_nt(const float* p, int lane) {
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = __builtin_nontemporal_load((const f32x4*)(p + (i * 64 + lane) * 4));
  }
  __device__ __forceinline__ void load_bf16_nt(const bf16_t* p, int lane) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const bf16x4 b = __builtin_nontemporal_load((const bf16x4*)(p + (i * 64 + lane) * 4));
      v[i] = (f32x4){bf2f(b[0]), bf2f(b[1]), bf2f(b[2]), bf2f(b[3])};
    }
  }
  __device__ __forceinline__ void store_f32_nt(float* p, int lane) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) __builtin_nontemporal_store(v[i], (f32x4*)(p + (i * 64 + lane) * 4));
  }
  __device__ __forceinline__ void store_bf16(bf16_t* p, int lane) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      bf16x4 b = {f2bf(v[i][0]), f2bf(v[i][1]), f2bf(v[i][2]), f2bf(v[i][3])};
      *(bf16x4*)(p + (i * 64 + lane) * 4) = b;
    }
  }
  __device__ __forceinline__ float sum() const {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    return wave_sum(s);
  }
};

// normalise in place: x <- (x - mean) * rstd ; returns rstd
template <int NV>
__device__ __forceinline__ float ln_normalize(RowVec<NV>& x, int H, float eps) {
  const float mean = x.sum() / (float)H;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      x.v[i][e] -= mean;
      s += x.v[i][e] * x.v[i][e];
    }
  const float var = wave_sum(s) / (float)H;
  const float rstd = rsqrtf(var + eps);
#pragma unroll
  for (int i = 0; i < NV; ++i) x.v[i] *= rstd;
  return rstd;
}

// MODE 0: src = h fp32 rows.  MODE 1: src = E[ids[t]] (embedding gather), also writes h fp32.
// MODE 2: src = gelu(bf16 d) evaluated on the bf16 tensor (head: LN(gelu(dense))).
// MODE 3: src = h fp32 + bf16 d (the residual add of hf:331-332 done HERE: a Linear's bf16 output joins the fp32 stream),
//         also writes the sum to h_out -- what the residual epilogue of the Wo GEMMs computes, bit for bit.
// NT: "stream_nt" bits, compile-time (a run-time branch between a plain and a non-temporal load of the same address is
// merged back into the plain one by the compiler)
template <int NV, int MODE, int NT = 0>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ h, const int64_t* __restrict__ ids,
                                                     const float* __restrict__ E, const bf16_t* __restrict__ d,
                                                     const float* __restrict__ w, float* __restrict__ h_out,
                                                     bf16_t* __restrict__ x_out, bf16_t* __restrict__ x0_out,
                                                     int T, int H, float eps) {
  constexpr int nt = NT;
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (t >= T) return;
  RowVec<NV> x, wv;
  if (MODE == 0) {
    x.load_f32(h + (long)t * H, lane);
  } else if (MODE == 1) {
    x.load_f32(E + ids[t] * (long)H, lane);
  } else if (MODE == 3) {
    RowVec<NV> y;
    if (nt & 1) {                                 // h and y: their last read of the forward
      x.load_f32_nt(h + (long)t * H, lane);
      y.load_bf16_nt(d + (long)t * H, lane);
    } else {
      x.load_f32(h + (long)t * H, lane);
      y.load_bf16(d + (long)t * H, lane);
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) x.v[i] += y.v[i];
    if (nt & 2) x.store_f32_nt(h_out + (long)t * H, lane);   // next read: the following LayerNorm, then the backward
    else x.store_f32(h_out + (long)t * H, lane);
  } else {
    x.load_bf16(d + (long)t * H, lane);
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) x.v[i][e] = rbf(gelu_f(x.v[i][e]));
  }
  wv.load_f32(w, lane);
  ln_normalize<NV>(x, H, eps);
#pragma unroll
  for (int i = 0; i < NV; ++i) x.v[i] *= wv.v[i];
  if (MODE == 1) {
    x.store_f32(h_out + (long)t * H, lane);      // residual stream starts here (fp32)
    x.store_bf16(x0_out + (long)t * H, lane);    // layer 0 has no attn_norm: Wqkv reads bf16(h)
  } else {
    x.store_bf16(x_out + (long)t * H, lane);
  }
}

#define DISPATCH_NV(H, CALL)                 \
  switch ((H) / 256) {                       \
    case 1: { constexpr int NV = 1; CALL; } break; \
    case 2: { constexpr int NV = 2; CALL; } break; \
    case 3: { constexpr int NV = 3; CALL; } break; \
    case 4: { constexpr int NV = 4; CALL; } break; \
    default: return SNX_E_SHAPE;             \
  }

static inline bool bad_h(int H) { return H <= 0 || (H % 256) != 0 || H > 1024; }

extern "C" int snx_ln_fwd(const float* h, const float* w, void* x_out, int32_t T, int32_t H, float eps,
                          hipStream_t st) {
  if (!h || !w || !x_out || T <= 0) return SNX_E_ARG;
  if (bad_h(H)) return SNX_E_SHAPE;
  DISPATCH_NV(H, hipLaunchKernelGGL((ln_fwd_kernel<NV, 0>), dim3(cdiv(T, ROWS_PER_BLOCK)), dim3(256), 0, st, h,
                                    nullptr, nullptr, nullptr, w, nullptr, (bf16_t*)x_out, nullptr, T, H, eps));
  SNX_CHECK_LAUNCH();
  return SNX_OK;
}

extern "C" int snx_ln_fwd_add(const float* h, const void* y, const float* w, float* h_out, void* x_out, int32_t T,
                              int32_t H, float eps, hipStream_t st) {
  if (!h || !y || !w || !h_out || !x_out || T <= 0) return SNX_E_ARG;
  if (bad_h(H)) return SNX_E_SHAPE;
  switch (g_snx_cfg.stream_nt & 3) {
    case 1: DISPATCH_NV(H, hipLaunchKernelGGL((ln_fwd_kernel<NV, 3, 1>), dim3(cdiv(T, ROWS_PER_BLOCK)), dim3(256), 0, st, h,
                                    nullptr, nullptr, (const bf16_t*)y, w, h_out, (bf16_t*)x_out, nullptr, T, H, eps)); break;
    case 2: DISPATCH_NV(H, hipLaunchKernelGGL((ln_fwd_kernel<NV, 3, 2>), dim3(cdiv(T, ROWS_PER_BLOCK)), dim3(256), 0, st, h,
                                    nullptr, nullptr, (const bf16_t*)y, w, h_out, (bf16_t*)x_out, nullptr, T, H, eps)); break;
    case 3: DISPATCH_NV(H, hipLaunchKernelGGL((ln_fwd_kernel<NV, 3, 3>), dim3(cdiv(T, ROWS_PER_BLOCK)), dim3(256), 0, st, h,
                                    nullptr, nullptr, (const bf16_t*)y, w, h_out, (bf16_t*)x_out, nullptr, T, H, eps)); break;
    default: DISPATCH_NV(H, hipLaunchKernelGGL((ln_fwd_kernel<NV, 3>), dim3(cdiv(T, ROWS_PER_BLOCK)), dim3(256), 0, st, h,
                                    nullptr, nullptr, (const bf16_t*)y, w, h_out, (bf16_t*)x_out, nullptr, T, H, eps)); break;
  }
  SNX_CHECK_LAUNCH();
  return SNX_OK;
}

extern "C" int snx_embed_ln_fwd(const int64_t* ids, const float* E, const float* w, float* h_out, void* x0_out,
                                int32_t T, int32_t H, float eps, hipStream_t st) {
  if (!ids || !E || !w || !h_out || !x0_out || T <= 0) return SNX_E_ARG;
  if (bad_h(H)) return SNX_E_SHAPE;
  DISPATCH_NV(H, hipLaunchKernelGGL((ln_fwd_kernel<NV, 1>), dim3(cdiv(T, ROWS_PER_BLOCK)), dim3(256), 0, st,
                                    nullptr, ids, E, nullptr, w, h_out, nullptr, (bf16_t*)x0_out, T, H, eps));
  SNX_CHECK_LAUNCH();
  return SNX_OK;
}

extern "C" int snx_gelu_ln_fwd(const void* d, const float* w, void* x_out, int32_t T, int32_t H, float eps,
                               hipStream_t st) {
  if (!d || !w || !x_out || T <= 0) return SNX_E_ARG;
  if (bad_h(H)) return SNX_E_SHAPE;
  DISPATCH_NV(H, hipLaunchKernelGGL((ln_fwd_kernel<NV, 2>), dim3(cdiv(T, ROWS_PER_BLOCK)), dim3(256), 0, st,
                                    nullptr, nullptr, nullptr, (const bf16_t*)d, w, nullptr, (bf16_t*)x_out,
                                    nullptr, T, H, eps));
  SNX_CHECK_LAUNCH();
  return SNX_OK;
}

// ------------------------------------------------------------------------------------------
// LayerNorm backward.  dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * w,
// dw[j] += sum_t dy[t,j] * xhat[t,j].  Each block walks `rows_per_block` rows with its 4 waves,
// accumulating dw in registers (a lane owns the same columns for every row), reduces the 4
// waves through LDS and stores the block's partial dw row into the caller's workspace; ln_dw_reduce_kernel then adds the
// partial rows to dw in block order (FIXED order: bit-reproducible; "det_reduce" = 0: one float atomic per column per
// block, arrival order, as in rounds 1-4).
//   MODE 0: input h fp32;   dh[t] += dx                       (residual-stream LN)
//   MODE 1: input E[ids[t]]; dx rows stored to dh[t] (fp32 scratch) for embed_scatter_kernel -- or, with "det_reduce" = 0,
//           gradE[ids[t]] += dx by float atomics (skip pad id)                                  (embedding LN)
//   MODE 2: input gelu(bf16 d); dd[t] = bf16(bf16(dx) * gelu'(d))          (head LN)
//   MODE 3: input h fp32;   dh[t]  = dx  (overwrite: the final_norm, first op of backward)
// ------------------------------------------------------------------------------------------
template <int NV, int MODE, int NT = 0>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const bf16_t* __restrict__ dy, const float* __restrict__ dyf,
                                                     const float* __restrict__ h, const int64_t* __restrict__ ids,
                                                     const float* __restrict__ E, const bf16_t* __restrict__ d,
                                                     const float* __restrict__ w, float* __restrict__ dh,
                                                     bf16_t* __restrict__ dh_bf16,
                                                     float* __restrict__ gradE, bf16_t* __restrict__ dd,
                                                     float* __restrict__ dw, float* __restrict__ dw_part, int T,
                                                     int H, float eps, int rows_per_block, int pad_id) {
  constexpr int nt = NT;
  __shared__ float red[ROWS_PER_BLOCK][1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  RowVec<NV> wv, dwacc;
  wv.load_f32(w, lane);
#pragma unroll
  for (int i = 0; i < NV; ++i) dwacc.v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int t_begin = blockIdx.x * rows_per_block;
  const int t_end = min(T, t_begin + rows_per_block);
  for (int t = t_begin + wave; t < t_end; t += ROWS_PER_BLOCK) {
    RowVec<NV> x, g, raw;
    long id = 0;
    if (MODE == 0 || MODE == 3) {
      if (nt & 4) x.load_f32_nt(h + (long)t * H, lane);   // the saved residual row: read for the last time
      else x.load_f32(h + (long)t * H, lane);
    } else if (MODE == 1) {
      id = ids[t];
      x.load_f32(E + id * (long)H, lane);
    } else {
      raw.load_bf16(d + (long)t * H, lane);
#pragma unroll
      for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) x.v[i][e] = rbf(gelu_f(raw.v[i][e]));
    }
    const float rstd = ln_normalize<NV>(x, H, eps);     // x = xhat
    if (MODE == 1) g.load_f32(dyf + (long)t * H, lane); // grad wrt embeddings output is the fp32 dh stream
    else if (nt & 4) g.load_bf16_nt(dy + (long)t * H, lane);
    else g.load_bf16(dy + (long)t * H, lane);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        dwacc.v[i][e] += g.v[i][e] * x.v[i][e];
        g.v[i][e] *= wv.v[i][e];
        s1 += g.v[i][e];
        s2 += g.v[i][e] * x.v[i][e];
      }
    s1 = wave_sum(s1) / (float)H;
    s2 = wave_sum(s2) / (float)H;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) g.v[i][e] = rstd * (g.v[i][e] - s1 - x.v[i][e] * s2);   // g = dx
    if (MODE == 0) {
      RowVec<NV> cur;
      if (nt & 128) cur.load_f32_nt(dh + (long)t * H, lane);   // (experiment: the fp32 gradient stream, next touched by the
      else cur.load_f32(dh + (long)t * H, lane);               // next LayerNorm backward three kernels later)
#pragma unroll
      for (int i = 0; i < NV; ++i) cur.v[i] += g.v[i];
      if (nt & 128) cur.store_f32_nt(dh + (long)t * H, lane);
      else cur.store_f32(dh + (long)t * H, lane);
      if (dh_bf16) cur.store_bf16(dh_bf16 + (long)t * H, lane);   // grad of the next bf16 branch output
    } else if (MODE == 3) {
      if (nt & 128) g.store_f32_nt(dh + (long)t * H, lane);
      else g.store_f32(dh + (long)t * H, lane);

      if (dh_bf16) g.store_bf16(dh_bf16 + (long)t * H, lane);
    } else if (MODE == 1 && dw_part) {
      g.store_f32(dh + (long)t * H, lane);               // dx row; summed per vocabulary id, in token order, by embed_scatter_kernel
    } else if (MODE == 1) {
      // The row goes through this wave's slice of `red` so that every atomic instruction adds 64 CONSECUTIVE
      // floats (256 B: the shape the memory-side atomic units take at full rate) instead of 4 B at a 16-B stride.
      float* mine = red[wave];
#pragma unroll
      for (int i = 0; i < NV; ++i) *(f32x4*)(mine + (i * 64 + lane) * 4) = g.v[i];
      if (id != pad_id) {                                  // nn.Embedding(padding_idx): no grad to the pad row
        float* dst = gradE + id * (long)H;
#pragma unroll
        for (int q = 0; q < 4 * NV; ++q) atomicAdd(dst + q * 64 + lane, mine[q * 64 + lane]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) g.v[i][e] = rbf(rbf(g.v[i][e]) * gelu_grad_f(raw.v[i][e]));
      g.store_bf16(dd + (long)t * H, lane);
    }
  }
  // reduce dw over the 4 waves, then one atomic per column
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) red[wave][(i * 64 + lane) * 4 + e] = dwacc.v[i][e];
  __syncthreads();
  for (int c = threadIdx.x; c < H; c += 256) {
    const float s = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
    if (dw_part) dw_part[(long)blockIdx.x * H + c] = s;
    else atomicAdd(dw + c, s);
  }
}

// dw[c] += part[0][c] + part[1][c] + ... in block order, for a BATCH of LayerNorm launches (blockIdx.y): the model's
// backward leaves every LayerNorm's partial rows in a region of its own and reduces them all in one launch per unit
// range (45 two-kernel sequences cost 0.8 ms per micro-step; measured, profiles/r05_experiments.txt).  One 1024-thread
// workgroup per 64 columns and launch: wave g sums the contiguous run of partial rows [g nb / 16, (g + 1) nb / 16) --
// 16 independent 256-byte loads in flight per trip --, the 16 wave sums are added in wave order through LDS.
__global__ __launch_bounds__(1024) void ln_dw_reduce_kernel(LnDwBatch batch, int H) {
  __shared__ float red[16][64];
  const float* __restrict__ part = batch.part[blockIdx.y];
  float* __restrict__ dw = batch.dw[blockIdx.y];
  const int nb = batch.nb[blockIdx.y];
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;                 // H % 256 == 0: always inside
  const int b0 = (int)((long)g * nb / 16), b1 = (int)((long)(g + 1) * nb / 16);
  float s = 0.f;
  int b = b0;
  for (; b + 16 <= b1; b += 16) {
    float v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = part[(long)(b + q) * H + c];
#pragma unroll
    for (int q = 0; q < 16; ++q) s += v[q];
  }
  for (; b < b1; ++b) s += part[(long)b * H + c];
  red[g][lane] = s;
  __syncthreads();
  if (g == 0) {
    float t = red[0][lane];
#pragma unroll
    for (int q = 1; q < 16; ++q) t += red[q][lane];
    dw[c] += t;
  }
}

int snx_ln_dw_reduce_batch(const LnDwBatch& batch, int n, int H, hipStream_t st) {
  if (n <= 0) return SNX_OK;
  if (n > SNX_LN_BATCH_MAX) return SNX_E_ARG;
  hipLaunchKernelGGL(ln_dw_reduce_kernel, dim3(H / 64, n), dim3(1024), 0, st, batch, H);
  SNX_CHECK_LAUNCH();
  return SNX_OK;
}

// ---- embedding gradient in a FIXED order ---------------------------------------------------------------------------
// gradE[id] += sum over the tokens t with ids[t] = id of dx[t], added in ascending t -- nn.Embedding's backward
// (hf:64-71) as a segmented sum instead of T float atomics whose arrival order changes the last bits from run to run.
// Per call: count the tokens of every id (integer atomics: exact), exclusive scan, fill the id's list in arrival order,
// rank every token inside its id's list (its place in ascending-t order: sum over ids of n_id^2 comparisons -- tiny for
// real batches, T^2 ~ a millisecond for a batch of one repeated id), then one wave per token: the FIRST token of an id
// walks the sorted list and adds the rows.  Pad tokens take no part (padding_idx).
__global__ void embed_count_kernel(const int64_t* __restrict__ ids, int* __restrict__ cnt, int T, int V, int pad_id) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  const long id = ids[t];
  if (id != pad_id && id >= 0 && id < V) atomicAdd(cnt + id, 1);
}
// off[v] = sum of cnt[0..v), off[V] = total.  One workgroup per chunk of 1024 ids: it sums everything in front of its
// chunk by itself (coalesced, at most V loads per workgroup: 50,000 ids = 49 workgroups, 1.2 M loads in all -- no second
// pass, no inter-workgroup dependency), then scans its chunk (wave-level shuffles + the 16 wave totals through LDS).
// (One workgroup walking all chunks took 45 us, a contiguous run of ids per thread 75 us.)
__global__ __launch_bounds__(1024) void embed_scan_kernel(const int* __restrict__ cnt, int* __restrict__ off, int V) {
  __shared__ int wsum[16], wpre[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int v0 = blockIdx.x * 1024;
  int before = 0;
  for (int v = threadIdx.x; v < v0; v += 1024) before += cnt[v];
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) before += __shfl_xor(before, d, 64);
  if (lane == 0) wpre[wave] = before;
  const int v = v0 + threadIdx.x;
  const int c = v < V ? cnt[v] : 0;
  int inc = c;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int up = __shfl_up(inc, d, 64);
    if (lane >= d) inc += up;
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  int base = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) {
    base += wpre[w];
    if (w < wave) base += wsum[w];
  }
  if (v < V) off[v] = base + inc - c;
  if (v == V - 1) off[V] = base + inc;
}
__global__ void embed_fill_kernel(const int64_t* __restrict__ ids, const int* __restrict__ off, int* __restrict__ fill,
                                  int* __restrict__ list, int T, int V, int pad_id) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  const long id = ids[t];
  if (id != pad_id && id >= 0 && id < V) list[off[id] + atomicAdd(fill + id, 1)] = t;
}
// Ids that own more than EMBED_HEAVY tokens of the call ("heavy": [CLS] / [SEP] once per sequence, the frequent tokens of
// Zipfian text -- several percent of T each) leave the per-token kernels: a serial walk over such a list by ONE wave is a
// millisecond-class tail at the end of the backward, and ranking it by comparisons is n^2.  They are ranked by a bitmap
// and summed in two FIXED levels instead (embed_heavy_* below).  Light ids keep the per-token path: rank by <= 64
// comparisons, the id's first token adds its rows in ascending token order.
constexpr int EMBED_HEAVY = 64;        // an id with more tokens than this is heavy
constexpr int EMBED_CHUNK = 32;        // tokens per partial row of a heavy id
constexpr int EMBED_WINDOW = 65536;    // token indices per bitmap window (8 KiB of LDS)

__global__ void embed_rank_kernel(const int64_t* __restrict__ ids, const int* __restrict__ off, const int* __restrict__ list,
                                  int* __restrict__ sorted, int T, int V, int pad_id) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  const long id = ids[t];
  if (id == pad_id || id < 0 || id >= V) return;
  const int b = off[id], e = off[id + 1];
  if (e - b > EMBED_HEAVY) return;                     // ranked by embed_heavy_sort_kernel
  int r = 0;
  for (int j = b; j < e; ++j) r += list[j] < t;
  sorted[b + r] = t;
}
template <int NV>
__global__ __launch_bounds__(256) void embed_scatter_kernel(const float* __restrict__ dx, const int64_t* __restrict__ ids,
                                                            const int* __restrict__ off, const int* __restrict__ sorted,
                                                            float* __restrict__ gradE, int T, int H, int V, int pad_id) {
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= T) return;
  const long id = ids[t];
  if (id == pad_id || id < 0 || id >= V) return;
  const int b = off[id], e = off[id + 1];
  if (e - b > EMBED_HEAVY) return;                     // summed by embed_heavy_chunk_kernel / embed_heavy_final_kernel
  if (sorted[b] != t) return;                          // only the id's first token adds
  RowVec<NV> acc, row;
  acc.load_f32(gradE + id * (long)H, lane);
  for (int j = b; j < e; ++j) {
    row.load_f32(dx + (long)sorted[j] * H, lane);
#pragma unroll
    for (int i = 0; i < NV; ++i) acc.v[i] += row.v[i];
  }
  acc.store_f32(gradE + id * (long)H, lane);
}

// ---- heavy ids -------------------------------------------------------------------------------------------------------
// collect: one thread per id; a heavy id takes a slot (which slot: arrival order -- it only places the id's partial rows in
// the workspace, no sum depends on it) and a run of cdiv(n, EMBED_CHUNK) chunk rows.
__global__ void embed_heavy_collect_kernel(const int* __restrict__ cnt, int* __restrict__ counters, int* __restrict__ heavy_id,
                                           int* __restrict__ heavy_chunk0, int V) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  const int n = cnt[v];
  if (n <= EMBED_HEAVY) return;
  const int slot = atomicAdd(counters + 0, 1);
  heavy_id[slot] = v;
  heavy_chunk0[slot] = atomicAdd(counters + 1, (n + EMBED_CHUNK - 1) / EMBED_CHUNK);
}
// sort: one workgroup per heavy slot.  The id's token list (arrival order) becomes ascending order through a bitmap over
// token indices in LDS: set the bits, scan the words' popcounts, emit.  n entries + T / 32 words of work whatever n is;
// windows of EMBED_WINDOW indices keep the LDS footprint fixed for any T.  Also describes the id's chunks for the sum.
__global__ __launch_bounds__(256) void embed_heavy_sort_kernel(const int* __restrict__ counters, const int* __restrict__ heavy_id,
                                                               const int* __restrict__ heavy_chunk0,
                                                               const int* __restrict__ off, const int* __restrict__ list,
                                                               int* __restrict__ sorted, int* __restrict__ chunk_desc, int T) {
  constexpr int WORDS = EMBED_WINDOW / 32, WPT = WORDS / 256;
  __shared__ uint32_t bits[WORDS];
  __shared__ int wsum[4];
  __shared__ int run_base;
  const int slot = blockIdx.x;
  if (slot >= counters[0]) return;
  const int id = heavy_id[slot];
  const int b = off[id], e = off[id + 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int c = threadIdx.x; c < (e - b + EMBED_CHUNK - 1) / EMBED_CHUNK; c += 256) {
    int* dsc = chunk_desc + 3 * (long)(heavy_chunk0[slot] + c);
    dsc[0] = b + c * EMBED_CHUNK;
    dsc[1] = min(EMBED_CHUNK, e - b - c * EMBED_CHUNK);
    dsc[2] = id;
  }
  if (threadIdx.x == 0) run_base = 0;
  for (int w0 = 0; w0 < T; w0 += EMBED_WINDOW) {
    for (int i = threadIdx.x; i < WORDS; i += 256) bits[i] = 0u;
    __syncthreads();
    for (int j = b + threadIdx.x; j < e; j += 256) {
      const int t = list[j] - w0;
      if (t >= 0 && t < EMBED_WINDOW) atomicOr(bits + (t >> 5), 1u << (t & 31));
    }
    __syncthreads();
    uint32_t w[WPT];
    int c = 0;
#pragma unroll
    for (int i = 0; i < WPT; ++i) { w[i] = bits[threadIdx.x * WPT + i]; c += __popc(w[i]); }
    int inc = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int up = __shfl_up(inc, d, 64);
      if (lane >= d) inc += up;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int base = run_base;
    for (int q = 0; q < wave; ++q) base += wsum[q];
    int k = b + base + inc - c;
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      uint32_t x = w[i];
      while (x) {
        const int bit = __ffs((int)x) - 1;
        sorted[k++] = w0 + (threadIdx.x * WPT + i) * 32 + bit;
        x &= x - 1;
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) run_base += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
  }
}
// level 1: one wave per chunk sums its <= EMBED_CHUNK dx rows in ascending token order into a partial row (indices and
// rows eight at a time: independent loads in flight, the adds stay in order).
template <int NV>
__global__ __launch_bounds__(256) void embed_heavy_chunk_kernel(const float* __restrict__ dx, const int* __restrict__ counters,
                                                                const int* __restrict__ chunk_desc,
                                                                const int* __restrict__ sorted, float* __restrict__ partial,
                                                                int H) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= counters[1]) return;
  const int b = chunk_desc[3 * (long)c], n = chunk_desc[3 * (long)c + 1];
  RowVec<NV> acc;
#pragma unroll
  for (int i = 0; i < NV; ++i) acc.v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int j = 0;
  for (; j + 8 <= n; j += 8) {
    int tk[8];
    RowVec<NV> row[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) tk[q] = sorted[b + j + q];
#pragma unroll
    for (int q = 0; q < 8; ++q) row[q].load_f32(dx + (long)tk[q] * H, lane);
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
      for (int i = 0; i < NV; ++i) acc.v[i] += row[q].v[i];
  }
  for (; j < n; ++j) {
    RowVec<NV> row;
    row.load_f32(dx + (long)sorted[b + j] * H, lane);
#pragma unroll
    for (int i = 0; i < NV; ++i) acc.v[i] += row.v[i];
  }
  acc.store_f32(partial + (long)c * H, lane);
}
// level 2: one wave per heavy id adds the id's partial rows to its gradient row in chunk order.
template <int NV>
__global__ __launch_bounds__(256) void embed_heavy_final_kernel(const float* __restrict__ partial, const int* __restrict__ counters,
                                                                const int* __restrict__ heavy_id,
                                                                const int* __restrict__ heavy_chunk0, const int* __restrict__ off,
                                                                float* __restrict__ gradE, int H) {
  const int lane = threadIdx.x & 63;
  const int slot = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (slot >= counters[0]) return;
  const int id = heavy_id[slot];
  const int nch = (off[id + 1] - off[id] + EMBED_CHUNK - 1) / EMBED_CHUNK;
  const float* src = partial + (long)heavy_chunk0[slot] * H;
  RowVec<NV> acc;
  acc.load_f32(gradE + id * (long)H, lane);
  int c = 0;
  for (; c + 4 <= nch; c += 4) {
    RowVec<NV> row[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) row[q].load_f32(src + (long)(c + q) * H, lane);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < NV; ++i) acc.v[i] += row[q].v[i];
  }
  for (; c < nch; ++c) {
    RowVec<NV> row;
    row.load_f32(src + (long)c * H, lane);
#pragma unroll
    for (int i = 0; i < NV; ++i) acc.v[i] += row.v[i];
  }
  acc.store_f32(gradE + id * (long)H, lane);
}
static inline int embed_max_heavy(int T) { return T / (EMBED_HEAVY + 1) + 1; }
static inline int embed_max_chunks(int T) { return T / EMBED_CHUNK + embed_max_heavy(T) + 1; }

static inline int ln_bwd_rows_per_block(int T) {
  int rpb = cdiv(T, 1024);                 // ~1024 blocks: 4 per CU
  rpb = ((rpb + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK) * ROWS_PER_BLOCK;
  return rpb < ROWS_PER_BLOCK ? ROWS_PER_BLOCK : rpb;
}

// partial dw rows of one LayerNorm backward launch (one per block).  The launch's block count cdiv(T, rows_per_block(T))
// is NOT monotone in T (8,192 rows: 1,024 blocks; 9,000 rows: 750), and a workspace planned for T_plan rows must serve
// every backward over T <= T_plan rows (the micro-step arena, snx_model_backward_units_range): the size is therefore the
// MONOTONE bound min(cdiv(T, 4), 1024) rows -- exact up to 4,096 rows, the block-count ceiling above.
extern "C" size_t snx_ln_bwd_workspace_bytes(int32_t T, int32_t H) {
  if (T <= 0 || H <= 0) return 0;
  const int nb_bound = cdiv(T, ROWS_PER_BLOCK) < 1024 ? cdiv(T, ROWS_PER_BLOCK) : 1024;
  return (size_t)nb_bound * H * 4;
}
// + the dx rows and the per-id token lists of the embedding gradient's ordered sum
extern "C" size_t snx_embed_ln_bwd_workspace_bytes(int32_t T, int32_t H, int32_t V) {
  if (T <= 0 || H <= 0 || V <= 0) return 0;
  const size_t a = (snx_ln_bwd_workspace_bytes(T, H) + 255) & ~(size_t)255;
  const size_t rows = ((size_t)T * H * 4 + 255) & ~(size_t)255;
  // cnt[V] fill[V] counters[2] (zeroed together) off[V + 1] list[T] sorted[T] heavy_id[mh] heavy_chunk0[mh] chunk_desc[3 mc]
  const size_t mh = embed_max_heavy(T), mc = embed_max_chunks(T);
  const size_t ints = ((size_t)(3 * (size_t)V + 3 + 2 * (size_t)T + 2 * mh + 3 * mc) * 4 + 255) & ~(size_t)255;
  const size_t partial = (mc * (size_t)H * 4 + 255) & ~(size_t)255;
  return a + rows + ints + partial;
}

// `defer` (the model's backward): the partial rows stay in `part` and the caller reduces them later, batched
static int ln_dw_finish(float* part, float* dw, int nb, int H, hipStream_t st, LnDwBatch* defer = nullptr, int* ndefer = nullptr) {
  if (defer) {
    if (*ndefer >= SNX_LN_BATCH_MAX) return SNX_E_ARG;
    defer->part[*ndefer] = part; defer->dw[*ndefer] = dw; defer->nb[*ndefer] = nb;
    ++*ndefer;
    return SNX_OK;
  }
  LnDwBatch one;
  one.part[0] = part; one.dw[0] = dw; one.nb[0] = nb;
  return snx_ln_dw_reduce_batch(one, 1, H, st);
}
// the workspace of a launch: nullptr with "det_reduce" = 0 (float atomics); SNX_E_ARG if it is missing or too small
#define LN_WS(need)                                              \
  float* part = nullptr;                                         \
  if (g_snx_cfg.det_reduce) {                                    \
    if (!ws || ws_bytes < (need)) return SNX_E_ARG;              \
    part = (float*)ws;                                           \
  }

int snx_ln_bwd_x(const void* dy, const float* h, const float* w, float* dh, void* dh_bf16, float* dw, int32_t T,
                 int32_t H, float eps, int32_t overwrite, void* ws, size_t ws_bytes, LnDwBatch* defer, int* ndefer,
                 hipStream_t st) {
  if (!dy || !h || !w || !dh || !dw || T <= 0) return SNX_E_ARG;
  if (bad_h(H)) return SNX_E_SHAPE;
  const int rpb = ln_bwd_rows_per_block(T);
  const int nb = cdiv(T, rpb);
  LN_WS(snx_ln_bwd_workspace_bytes(T, H));
#define SNX_LN_BWD_LAUNCH(MODE_, NT_)                                                                              \
  DISPATCH_NV(H, hipLaunchKernelGGL((ln_bwd_kernel<NV, MODE_, NT_>), dim3(nb), dim3(256), 0, st, (const bf16_t*)dy, \
                                    nullptr, h, nullptr, nullptr, nullptr, w, dh, (bf16_t*)dh_bf16, nullptr, nullptr, \
                                    dw, part, T, H, eps, rpb, -1))
  const int nt = g_snx_cfg.stream_nt & (4 | 128);
  if (overwrite) {
    if (nt == 132) { SNX_LN_BWD_LAUNCH(3, 132); } else if (nt == 128) { SNX_LN_BWD_LAUNCH(3, 128); }
    else if (nt == 4) { SNX_LN_BWD_LAUNCH(3, 4); } else { SNX_LN_BWD_LAUNCH(3, 0); }
  } else {
    if (nt == 132) { SNX_LN_BWD_LAUNCH(0, 132); } else if (nt == 128) { SNX_LN_BWD_LAUNCH(0, 128); }
    else if (nt == 4) { SNX_LN_BWD_LAUNCH(0, 4); } else { SNX_LN_BWD_LAUNCH(0, 0); }
  }
#undef SNX_LN_BWD_LAUNCH
  SNX_CHECK_LAUNCH();
  return part ? ln_dw_finish(part, dw, nb, H, st, defer, ndefer) : SNX_OK;
}
extern "C" int snx_ln_bwd(const void* dy, const float* h, const float* w, float* dh, void* dh_bf16, float* dw,
                          int32_t T, int32_t H, float eps, int32_t overwrite, void* ws, size_t ws_bytes,
                          hipStream_t st) {
  return snx_ln_bwd_x(dy, h, w, dh, dh_bf16, dw, T, H, eps, overwrite, ws, ws_bytes, nullptr, nullptr, st);
}

extern "C" int snx_embed_ln_bwd(const float* dh, const int64_t* ids, const float* E, const float* w, float* gradE,
                                float* dw, int32_t T, int32_t H, int32_t V, float eps, int32_t pad_id, void* ws,
                                size_t ws_bytes, hipStream_t st) {
  return snx_embed_ln_bwd_x(dh, ids, E, w, gradE, dw, T, H, V, eps, pad_id, ws, ws_bytes, nullptr, nullptr, st);
}
int snx_embed_ln_bwd_x(const float* dh, const int64_t* ids, const float* E, const float* w, float* gradE, float* dw,
                       int32_t T, int32_t H, int32_t V, float eps, int32_t pad_id, void* ws, size_t ws_bytes,
                       LnDwBatch* defer, int* ndefer, hipStream_t st) {
  if (!dh || !ids || !E || !w || !gradE || !dw || T <= 0 || V <= 0) return SNX_E_ARG;
  if (bad_h(H)) return SNX_E_SHAPE;
  const int rpb = ln_bwd_rows_per_block(T);
  const int nb = cdiv(T, rpb);
  LN_WS(snx_embed_ln_bwd_workspace_bytes(T, H, V));
  float *dxrows = nullptr, *partial = nullptr;
  int *cnt = nullptr, *fill = nullptr, *counters = nullptr, *off = nullptr, *list = nullptr, *sorted = nullptr;
  int *heavy_id = nullptr, *heavy_chunk0 = nullptr, *chunk_desc = nullptr;
  const int mh = embed_max_heavy(T), mc = embed_max_chunks(T);
  if (part) {
    char* base = (char*)ws;
    size_t o = (snx_ln_bwd_workspace_bytes(T, H) + 255) & ~(size_t)255;
    dxrows = (float*)(base + o);
    o += ((size_t)T * H * 4 + 255) & ~(size_t)255;
    cnt = (int*)(base + o); fill = cnt + V; counters = fill + V; off = counters + 2; list = off + V + 1; sorted = list + T;
    heavy_id = sorted + T; heavy_chunk0 = heavy_id + mh; chunk_desc = heavy_chunk0 + mh;
    o += ((size_t)(3 * (size_t)V + 3 + 2 * (size_t)T + 2 * (size_t)mh + 3 * (size_t)mc) * 4 + 255) & ~(size_t)255;
    partial = (float*)(base + o);
    if (hipMemsetAsync(cnt, 0, ((size_t)2 * V + 2) * 4, st) != hipSuccess) return SNX_E_ARG;
    hipLaunchKernelGGL(embed_count_kernel, dim3(cdiv(T, 256)), dim3(256), 0, st, ids, cnt, T, V, pad_id);
    hipLaunchKernelGGL(embed_scan_kernel, dim3(cdiv(V, 1024)), dim3(1024), 0, st, (const int*)cnt, off, V);
    hipLaunchKernelGGL(embed_fill_kernel, dim3(cdiv(T, 256)), dim3(256), 0, st, ids, (const int*)off, fill, list, T, V, pad_id);
    hipLaunchKernelGGL(embed_rank_kernel, dim3(cdiv(T, 256)), dim3(256), 0, st, ids, (const int*)off, (const int*)list,
                       sorted, T, V, pad_id);
    hipLaunchKernelGGL(embed_heavy_collect_kernel, dim3(cdiv(V, 256)), dim3(256), 0, st, (const int*)cnt, counters, heavy_id,
                       heavy_chunk0, V);
    hipLaunchKernelGGL(embed_heavy_sort_kernel, dim3(mh), dim3(256), 0, st, (const int*)counters, (const int*)heavy_id,
                       (const int*)heavy_chunk0, (const int*)off, (const int*)list, sorted, chunk_desc, T);
    SNX_CHECK_LAUNCH();
  }
  DISPATCH_NV(H, hipLaunchKernelGGL((ln_bwd_kernel<NV, 1>), dim3(nb), dim3(256), 0, st, nullptr, dh,
                                    nullptr, ids, E, nullptr, w, dxrows, nullptr, gradE, nullptr, dw, part, T, H, eps,
                                    rpb, pad_id));
  SNX_CHECK_LAUNCH();
  if (!part) return SNX_OK;
  DISPATCH_NV(H, hipLaunchKernelGGL((embed_scatter_kernel<NV>), dim3(cdiv(T, 4)), dim3(256), 0, st, (const float*)dxrows,
                                    ids, (const int*)off, (const int*)sorted, gradE, T, H, V, pad_id));
  DISPATCH_NV(H, hipLaunchKernelGGL((embed_heavy_chunk_kernel<NV>), dim3(cdiv(mc, 4)), dim3(256), 0, st, (const float*)dxrows,
                                    (const int*)counters, (const int*)chunk_desc, (const int*)sorted, partial, H));
  DISPATCH_NV(H, hipLaunchKernelGGL((embed_heavy_final_kernel<NV>), dim3(cdiv(mh, 4)), dim3(256), 0, st, (const float*)partial,
                                    (const int*)counters, (const int*)heavy_id, (const int*)heavy_chunk0, (const int*)off,
                                    gradE, H));
  SNX_CHECK_LAUNCH();
  return ln_dw_finish(part, dw, nb, H, st, defer, ndefer);
}

extern "C" int snx_gelu_ln_bwd(const void* dy, const void* d, const float* w, void* dd, float* dw, int32_t T,
                               int32_t H, float eps, void* ws, size_t ws_bytes, hipStream_t st) {
  return snx_gelu_ln_bwd_x(dy, d, w, dd, dw, T, H, eps, ws, ws_bytes, nullptr, nullptr, st);
}
int snx_gelu_ln_bwd_x(const void* dy, const void* d, const float* w, void* dd, float* dw, int32_t T, int32_t H,
                      float eps, void* ws, size_t ws_bytes, LnDwBatch* defer, int* ndefer, hipStream_t st) {
  if (!dy || !d || !w || !dd || !dw || T <= 0) return SNX_E_ARG;
  if (bad_h(H)) return SNX_E_SHAPE;
  const int rpb = ln_bwd_rows_per_block(T);
  const int nb = cdiv(T, rpb);
  LN_WS(snx_ln_bwd_workspace_bytes(T, H));
  DISPATCH_NV(H, hipLaunchKernelGGL((ln_bwd_kernel<NV, 2>), dim3(nb), dim3(256), 0, st,
                                    (const bf16_t*)dy, nullptr, nullptr, nullptr, nullptr, (const bf16_t*)d, w,
                                    nullptr, nullptr, nullptr, (bf16_t*)dd, dw, part, T, H, eps, rpb, -1));
  SNX_CHECK_LAUNCH();
  return part ? ln_dw_finish(part, dw, nb, H, st, defer, ndefer) : SNX_OK;
}

// ------------------------------------------------------------------------------------------
// RoPE in place on the q and k thirds of qkv [T, 3, heads, 64] (half-split rotate_half).
// tab = [max_pos][32] float2 (cos, sin) for this layer type.  inverse=1 applies the transposed
// rotation (backward).  One thread = 8 dims of the low half + the matching 8 of the high half.
// ------------------------------------------------------------------------------------------
__global__ void rope_kernel(bf16_t* __restrict__ qkv, const f32x2* __restrict__ tab, const int32_t* __restrict__ pos,
                            long n_items, int heads, int inverse) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;   // item = (t, which(q|k), head, quarter)
  if (gid >= n_items) return;
  const int quarter = gid & 3;
  long rest = gid >> 2;
  const int head = rest % heads; rest /= heads;
  const int which = rest & 1;
  const long t = rest >> 1;
  bf16_t* base = qkv + (t * 3 + which) * (long)heads * 64 + head * 64 + quarter * 8;
  const bf16x8 lo = *(const bf16x8*)base, hi = *(const bf16x8*)(base + 32);
  const f32x2* cs = tab + (long)pos[t] * 32 + quarter * 8;
  bf16x8 olo, ohi;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float c = cs[e][0], s = inverse ? -cs[e][1] : cs[e][1];
    const float x1 = bf2f(lo[e]), x2 = bf2f(hi[e]);
    olo[e] = f2bf(x1 * c - x2 * s);
    ohi[e] = f2bf(x2 * c + x1 * s);
  }
  *(bf16x8*)base = olo;
  *(bf16x8*)(base + 32) = ohi;
}

// rows[t] = tab[pos[t]]: the (cos, sin) row of every token, resolved once per forward pass for all layers of one
// theta (256 B per token; one wave per 4 tokens, 16 B per lane)
__global__ void rope_rows_kernel(const f32x4* __restrict__ tab, const int32_t* __restrict__ pos, f32x4* __restrict__ rows,
                                 int T) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;   // 16-byte piece: token i >> 4, piece i & 15
  if (i >= (long)T * 16) return;
  const int t = (int)(i >> 4);
  rows[i] = tab[(long)pos[t] * 16 + (i & 15)];
}

extern "C" int snx_rope_rows(const float* cos_sin_tab, const int32_t* pos, float* rows, int32_t T, hipStream_t st) {
  if (!cos_sin_tab || !pos || !rows) return SNX_E_ARG;
  if (T <= 0) return SNX_E_SHAPE;
  const long n = (long)T * 16;
  hipLaunchKernelGGL(rope_rows_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, (const f32x4*)cos_sin_tab, pos,
                     (f32x4*)rows, T);
  SNX_CHECK_LAUNCH();
  return SNX_OK;
}

extern "C" int snx_rope_inplace(void* qkv, const float* cos_sin_tab, const int32_t* pos, int32_t T, int32_t heads,
                                int32_t inverse, hipStream_t st) {
  if (!qkv || !cos_sin_tab || !pos || T <= 0 || heads <= 0) return SNX_E_ARG;
  const long n = (long)T * 2 * heads * 4;
  hipLaunchKernelGGL(rope_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, (bf16_t*)qkv, (const f32x2*)cos_sin_tab, pos,
                     n, heads, inverse);
  SNX_CHECK_LAUNCH();
  return SNX_OK;
}

// ------------------------------------------------------------------------------------------
// GeGLU: u = [a | g] bf16 [T, 2I];  y = bf16( bf16(gelu(a)) * g )        (hf:90-91)
// backward: dact = bf16(dy*g), dg = bf16(dy*act), da = bf16(dact * gelu'(a))
// ------------------------------------------------------------------------------------------
__global__ void geglu_fwd_kernel(const bf16_t* __restrict__ u, bf16_t* __restrict__ y, long n8, int I8) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= n8) return;
  const long t = gid / I8;
  const int c = gid % I8;
  const bf16x8 a = *(const bf16x8*)(u + t * (long)I8 * 16 + c * 8);
  const bf16x8 g = *(const bf16x8*)(u + t * (long)I8 * 16 + (long)I8 * 8 + c * 8);
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = f2bf(rbf(gelu_f(bf2f(a[e]))) * bf2f(g[e]));
  *(bf16x8*)(y + gid * 8) = o;
}

__global__ void geglu_bwd_kernel(const bf16_t* __restrict__ u, const bf16_t* __restrict__ dy, bf16_t* __restrict__ du,
                                 long n8, int I8) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= n8) return;
  const long t = gid / I8;
  const int c = gid % I8;
  const long oa = t * (long)I8 * 16 + c * 8, og = oa + (long)I8 * 8;
  const bf16x8 a = *(const bf16x8*)(u + oa), g = *(const bf16x8*)(u + og), d = *(const bf16x8*)(dy + gid * 8);
  bf16x8 da, dg;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float af = bf2f(a[e]), gf = bf2f(g[e]), df = bf2f(d[e]);
    const float act = rbf(gelu_f(af));
    dg[e] = f2bf(df * act);
    da[e] = f2bf(rbf(df * gf) * gelu_grad_f(af));
  }
  *(bf16x8*)(du + oa) = da;
  *(bf16x8*)(du + og) = dg;
}

extern "C" int snx_geglu_fwd(const void* u, void* y, int32_t T, int32_t I, hipStream_t st) {
  if (!u || !y || T <= 0 || I <= 0) return SNX_E_ARG;
  if (I % 8) return SNX_E_SHAPE;
  const long n8 = (long)T * (I / 8);
  hipLaunchKernelGGL(geglu_fwd_kernel, dim3(cdiv(n8, 256)), dim3(256), 0, st, (const bf16_t*)u, (bf16_t*)y, n8, I / 8);
  SNX_CHECK_LAUNCH();
  return SNX_OK;
}

extern "C" int snx_geglu_bwd(const void* u, const void* dy, void* du, int32_t T, int32_t I, hipStream_t st) {
  if (!u || !dy || !du || T <= 0 || I <= 0) return SNX_E_ARG;
  if (I % 8) return SNX_E_SHAPE;
  const long n8 = (long)T * (I / 8);
  hipLaunchKernelGGL(geglu_bwd_kernel, dim3(cdiv(n8, 256)), dim3(256), 0, st, (const bf16_t*)u, (const bf16_t*)dy,
                     (bf16_t*)du, n8, I / 8);
  SNX_CHECK_LAUNCH();
  return SNX_OK;
}

// dh fp32 -> bf16 (grad of a bf16 branch output taken from the fp32 residual-gradient stream)
extern "C" int snx_cast_f32_to_bf16(const float* in, void* out, int64_t n, hipStream_t st) {
  return snx_cast_bf16(in, out, n, st);
}
