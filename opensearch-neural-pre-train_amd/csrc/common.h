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

// exact-erf GELU and its derivative (fp32 math)
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
  const float pdf = 0.39894228040143268f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
