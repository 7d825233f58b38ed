// Shared device helpers for the snx kernels (gfx950 / CDNA4 only: wave64, MFMA 16x16x32 bf16,
// LDS-DMA global_load_lds, ds_read_b64_tr_b16).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define SNX_OK 0
#define SNX_E_SHAPE (-2)     // operand shapes do not satisfy the kernel's tiling assumptions
#define SNX_E_ARG (-3)       // null / inconsistent argument

#define SNX_CHECK_LAUNCH()                                   \
  do {                                                       \
    hipError_t e__ = hipGetLastError();                      \
    if (e__ != hipSuccess) return (int)e__;                  \
  } while (0)

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

__device__ __forceinline__ float bf2f(bf16_t x) { return (float)x; }
__device__ __forceinline__ bf16_t f2bf(float x) { return (bf16_t)x; }   // v_cvt_pk_bf16_f32, RNE
// round-trip through bf16 (the "output of a bf16 op" cast point)
__device__ __forceinline__ float rbf(float x) { return (float)((bf16_t)x); }

// a * b rounded to fp32 on its own: hipcc contracts `a * b + c` (and __fmul_rn, which is just `*`) into an fma where
// it sees fit, differently from one kernel to the next; torch's eager q * cos + rotate_half(q) * sin rounds each
// product (hf:196-219), and two kernels that must agree bit for bit need the same arithmetic.
__device__ __forceinline__ float mul_rn(float a, float b) {
  float r;
  asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

__device__ __forceinline__ uint32_t bf16_bits(float x) {
  bf16_t b = (bf16_t)x;
  return (uint32_t)__builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float bits_to_f32(uint32_t bf16bits) {
  return __builtin_bit_cast(float, bf16bits << 16);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Exact-erf GELU x*Phi(x) and its derivative Phi(x) + x*phi(x) in fp32 (torch's F.gelu default).
// Phi is evaluated as 1 - erfc/2 with erfc(z) = t*P6(t)*exp(-z^2), t = 1/(1 + 0.3275911 z), z = |x|/sqrt(2)
// (the Abramowitz-Stegun 7.1.26 form with a degree-6 minimax P: 3.5e-9 absolute error before fp32
// rounding), so ONE v_exp and ONE v_rcp serve both the cdf and the pdf.  libm's erff costs ~10x the
// VALU work, which made the fused GeGLU epilogues VALU-bound (geglu_bwd GEMM: 268 TFLOP/s).
// For every bf16 input with x > -3.14 the bf16-rounded GELU equals torch's fp32 result bit for
// bit (tests/test_gpu_ops.py checks all 65k inputs); further out torch's own 1+erf cancellation
// error dominates and this form is the more accurate one.
__device__ __forceinline__ void gelu_cdf_exp(float x, float& cdf, float& e) {
  const float t = __builtin_amdgcn_rcpf(fmaf(fabsf(x), 0.23164189f, 1.0f));
  e = __builtin_amdgcn_exp2f(x * x * -0.72134752f);              // exp(-x^2/2)
  float p = -0.29844744f;
  p = fmaf(p, t, 1.50480362f);
  p = fmaf(p, t, -2.08551838f);
  p = fmaf(p, t, 2.04007077f);
  p = fmaf(p, t, -0.74904709f);
  p = fmaf(p, t, 0.43109737f);
  p = fmaf(p, t, 0.15704115f);
  const float h = 0.5f * (p * t) * e;                            // erfc(|x|/sqrt2) / 2
  cdf = x >= 0.f ? 1.0f - h : h;
}
__device__ __forceinline__ float gelu_f(float x) {
  float cdf, e;
  gelu_cdf_exp(x, cdf, e);
  return x * cdf;
}
__device__ __forceinline__ float gelu_grad_f(float x) {
  float cdf, e;
  gelu_cdf_exp(x, cdf, e);
  return fmaf(x, 0.39894228040143268f * e, cdf);
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Opt-in to more than 64 KiB of dynamic LDS for one kernel, once PER DEVICE (hipFuncSetAttribute is per device; a process
// that moves to a second device would otherwise launch without it there).  One static instance per kernel; returns SNX_OK,
// SNX_E_ARG (no current device) or the HIP error code.
struct LdsOptIn {
  bool done[64] = {};
  int ensure(const void* kernel, int bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return SNX_E_ARG;
    if (done[dev]) return SNX_OK;
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return (int)e;
    done[dev] = true;
    return SNX_OK;
  }
};

// one launch for the bf16 copy + transposed bf16 copy of up to 64 same-shaped fp32 matrices (elementwise.hip; used by the
// weight-cache refresh in model.hip)
#define SNX_CAST_BATCH_MAX 64
struct CastBatch {
  const float* src[SNX_CAST_BATCH_MAX];
  bf16_t* out[SNX_CAST_BATCH_MAX];
  bf16_t* out_t[SNX_CAST_BATCH_MAX];
};
int snx_cast_both_batched(const CastBatch& b, int n, int R, int C, int interleave, hipStream_t st);

// LayerNorm weight gradients, ordered reduction (elementwise.hip): a backward launch leaves one partial dw row per block in
// its workspace; the rows of up to SNX_LN_BATCH_MAX launches are added to their dw vectors, in block order, by ONE kernel.
// The public entry points reduce at once; the model's backward (model.hip) defers and reduces once per unit range.
#define SNX_LN_BATCH_MAX 64
struct LnDwBatch {
  const float* part[SNX_LN_BATCH_MAX];
  float* dw[SNX_LN_BATCH_MAX];
  int nb[SNX_LN_BATCH_MAX];
};
int snx_ln_dw_reduce_batch(const LnDwBatch& batch, int n, int H, hipStream_t st);
int snx_ln_bwd_x(const void* dy, const float* h, const float* w, float* dh, void* dh_bf16, float* dw, int32_t T, int32_t H,
                 float eps, int32_t overwrite, void* ws, size_t ws_bytes, LnDwBatch* defer, int* ndefer, hipStream_t st);
int snx_gelu_ln_bwd_x(const void* dy, const void* d, const float* w, void* dd, float* dw, int32_t T, int32_t H, float eps,
                      void* ws, size_t ws_bytes, LnDwBatch* defer, int* ndefer, hipStream_t st);
int snx_embed_ln_bwd_x(const float* dh, const int64_t* ids, const float* E, const float* w, float* gradE, float* dw,
                       int32_t T, int32_t H, int32_t V, float eps, int32_t pad_id, void* ws, size_t ws_bytes,
                       LnDwBatch* defer, int* ndefer, hipStream_t st);

