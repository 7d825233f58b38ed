// Fused global-norm clip + AdamW over FLAT fp32 buffers (K17 of SURVEY.md §2.3:
// clip_grad_norm_ + torch.optim.AdamW, ref:src/train/cli/train_v33_ddp.py:367-374,560-581).
// The trainer keeps parameters, gradients and both moments each in one contiguous buffer, so
// the whole optimizer step is two HBM-bound launches (sum of squares; update) instead of
// ~300 small ones: 28 B per parameter, 149 M parameters -> ~4.2 GB per step.
#include "common.h"
#include "snx.h"

__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, long n4, float* __restrict__ part) {
  __shared__ float red[4];
  float s = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const f32x4 v = *(const f32x4*)(g + i * 4);
    s += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* __restrict__ part, int nblocks, const float* __restrict__ g,
                                                          long n, long n4, float* __restrict__ norm_out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < nblocks; i += 256) s += part[i];
  if (threadIdx.x == 0)
    for (long j = n4 * 4; j < n; ++j) s += g[j] * g[j];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) norm_out[0] = sqrtf((red[0] + red[1]) + (red[2] + red[3]));
}

// elements in [nodecay_begin, nodecay_end) get weight_decay = 0 (the reference's no-decay group)
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, long n, const float* __restrict__ norm,
                                                    float max_norm, float lr, float beta1, float beta2, float eps,
                                                    float wd, float bc1, float bc2_sqrt, long nodecay_begin,
                                                    long nodecay_end) {
  const float coef = max_norm > 0.f ? fminf(1.0f, max_norm / (norm[0] + 1e-6f)) : 1.0f;
  const float step_size = lr / bc1;
  for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (long)gridDim.x * blockDim.x * 4) {
    if (i + 3 < n) {
      f32x4 pp = *(f32x4*)(p + i), mm = *(f32x4*)(m + i), vv = *(f32x4*)(v + i);
      const f32x4 gg = *(const f32x4*)(g + i);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float gr = gg[e] * coef;
        const float w = (i + e >= nodecay_begin && i + e < nodecay_end) ? 0.f : wd;
        pp[e] *= 1.0f - lr * w;
        mm[e] = beta1 * mm[e] + (1.0f - beta1) * gr;
        vv[e] = beta2 * vv[e] + (1.0f - beta2) * gr * gr;
        pp[e] -= step_size * mm[e] / (sqrtf(vv[e]) / bc2_sqrt + eps);
      }
      *(f32x4*)(p + i) = pp; *(f32x4*)(m + i) = mm; *(f32x4*)(v + i) = vv;
    } else {
      for (long j = i; j < n; ++j) {
        const float gr = g[j] * coef;
        const float w = (j >= nodecay_begin && j < nodecay_end) ? 0.f : wd;
        float pj = p[j] * (1.0f - lr * w);
        const float mj = beta1 * m[j] + (1.0f - beta1) * gr;
        const float vj = beta2 * v[j] + (1.0f - beta2) * gr * gr;
        pj -= step_size * mj / (sqrtf(vj) / bc2_sqrt + eps);
        p[j] = pj; m[j] = mj; v[j] = vj;
      }
    }
  }
}

extern "C" size_t snx_adamw_scratch_bytes(void) { return 2048 * sizeof(float); }

// hp [host] = {lr, beta1, beta2, eps, weight_decay, max_norm (<=0: no clipping)}; step = 1-based
// optimizer step count (bias correction); norm_out [1] receives the PRE-clip global L2 norm.
extern "C" int snx_adamw_clip_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                                   const float* hp, int64_t step, int64_t nodecay_begin, int64_t nodecay_end,
                                   float* norm_out, void* scratch, hipStream_t st) {
  if (!params || !grads || !exp_avg || !exp_avg_sq || !hp || !norm_out || !scratch) return SNX_E_ARG;
  if (n <= 0 || step <= 0) return SNX_E_SHAPE;
  if (((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) return SNX_E_ARG;
  const float lr = hp[0], b1 = hp[1], b2 = hp[2], eps = hp[3], wd = hp[4], max_norm = hp[5];
  const long n4 = n / 4;
  int blocks = cdiv(n4, 256 * 8);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  float* part = (float*)scratch;
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(blocks), dim3(256), 0, st, grads, n4, part);
  SNX_CHECK_LAUNCH();
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, st, part, blocks, grads, (long)n, n4, norm_out);
  SNX_CHECK_LAUNCH();
  const float bc1 = 1.0f - powf(b1, (float)step), bc2s = sqrtf(1.0f - powf(b2, (float)step));
  int ublocks = cdiv(n, 256 * 4 * 4);
  if (ublocks > 8192) ublocks = 8192;
  hipLaunchKernelGGL(adamw_kernel, dim3(ublocks), dim3(256), 0, st, params, grads, exp_avg, exp_avg_sq, (long)n,
                     norm_out, max_norm, lr, b1, b2, eps, wd, bc1, bc2s, (long)nodecay_begin, (long)nodecay_end);
  SNX_CHECK_LAUNCH();
  return SNX_OK;
}
