// Fused AdamW (+ optional EMA) over ALL parameter tensors of a model in one launch (SURVEY 8f "next" row 1).
// Replaces (reference): optim/adamw.py:16-46 (torch.optim.AdamW stepping 215 small tensors for MobileViT-S) and the per-tensor Python
// loop of EMA.update_parameters (cvnets/misc/averaging_utils.py:43-55).  Parameters, gradients and EMA copies stay ordinary torch
// tensors (a pointer table addresses them); the moments live in two flat fp32 buffers; learning rate / weight decay per parameter
// group and the step counter are DEVICE values, so the launch can sit inside a captured hipGraph while a scheduler changes the rate.
//
// table[n+1][8] (int64): param ptr, grad ptr, ema ptr (0 = none), state offset, numel, group index, prefix start, unused;
// row n carries only the total in its prefix field.  Thread = 4 consecutive elements of the concatenated index space.
#include "common.hpp"
#include "cvnets_hip.h"

__global__ __launch_bounds__(256) void adamw_multi_kernel(const long long* __restrict__ table, int n, long long total, float* __restrict__ m_flat,
                                                          float* __restrict__ v_flat, const float* __restrict__ group_hp, float beta1, float beta2, float eps,
                                                          const float* __restrict__ step, const float* __restrict__ inv_grad_scale, float ema_momentum) {
  __shared__ float bc[2];
  if (threadIdx.x == 0) {
    const float t = *step + 1.0f;
    bc[0] = 1.0f - powf(beta1, t);
    bc[1] = sqrtf(1.0f - powf(beta2, t));
  }
  __syncthreads();
  const float bc1 = bc[0], sbc2 = bc[1];
  const float gs = inv_grad_scale ? *inv_grad_scale : 1.0f;
  for (long long e0 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; e0 < total; e0 += (long long)gridDim.x * 1024) {
    int lo = 0, hi = n - 1;  // last tensor whose prefix start <= e0
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (table[(size_t)mid * 8 + 6] <= e0) lo = mid; else hi = mid - 1;
    }
    int ti = lo;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long long e = e0 + j;
      if (e >= total) break;
      while (e >= table[(size_t)(ti + 1) * 8 + 6]) ++ti;  // crossed into the next tensor
      const long long* row = table + (size_t)ti * 8;
      const long long k = e - row[6];
      float* p = reinterpret_cast<float*>(row[0]);
      const float* g = reinterpret_cast<const float*>(row[1]);
      float* ema = reinterpret_cast<float*>(row[2]);
      const long long so = row[3] + k;
      const float lr = group_hp[row[5] * 2], wd = group_hp[row[5] * 2 + 1];
      const float grad = g[k] * gs;
      float w = p[k] * (1.0f - lr * wd);
      float m = m_flat[so];
      m = m + (grad - m) * (1.0f - beta1);
      const float v = v_flat[so] * beta2 + grad * grad * (1.0f - beta2);
      const float denom = sqrtf(v) / sbc2 + eps;
      w -= (lr / bc1) * (m / denom);
      p[k] = w;
      m_flat[so] = m;
      v_flat[so] = v;
      if (ema) ema[k] = ema[k] * (1.0f - ema_momentum) + ema_momentum * w;
    }
  }
}
__global__ void step_advance_kernel(float* step) { *step += 1.0f; }

// EMA of non-parameter state (BatchNorm running statistics): dst = dst*(1-mom) + mom*src over a pointer table [n+1][4]:
// dst ptr, src ptr, numel, prefix start
__global__ __launch_bounds__(256) void lerp_multi_kernel(const long long* __restrict__ table, int n, long long total, float mom) {
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (table[(size_t)mid * 4 + 3] <= e) lo = mid; else hi = mid - 1;
    }
    const long long* row = table + (size_t)lo * 4;
    const long long k = e - row[3];
    float* d = reinterpret_cast<float*>(row[0]);
    const float* s = reinterpret_cast<const float*>(row[1]);
    d[k] = d[k] * (1.0f - mom) + mom * s[k];
  }
}

extern "C" int cvh_adamw_multi(const long long* table, int n_tensors, long long total, float* m_flat, float* v_flat, const float* group_hp, float beta1,
                               float beta2, float eps, float* step, const float* inv_grad_scale, float ema_momentum, void* stream) {
  if (n_tensors <= 0 || total <= 0) return 0;
  long long g = (total + 1023) / 1024;
  if (g > 8192) g = 8192;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(adamw_multi_kernel, dim3((int)g), dim3(256), 0, st, table, n_tensors, total, m_flat, v_flat, group_hp, beta1, beta2, eps, step,
                     inv_grad_scale, ema_momentum);
  hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(1), 0, st, step);
  CVH_CHECK_LAUNCH();
  return 0;
}
extern "C" int cvh_lerp_multi(const long long* table, int n_tensors, long long total, float momentum, void* stream) {
  if (n_tensors <= 0 || total <= 0) return 0;
  long long g = (total + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(lerp_multi_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, table, n_tensors, total, momentum);
  CVH_CHECK_LAUNCH();
  return 0;
}
