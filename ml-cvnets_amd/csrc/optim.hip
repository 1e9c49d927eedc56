// Fused AdamW (+ optional EMA) over ALL parameter tensors of a model in one launch (SURVEY 8f "next" row 1).
// Replaces (reference): optim/adamw.py:16-46 (torch.optim.AdamW stepping 215 small tensors for MobileViT-S) and the per-tensor Python
// loop of EMA.update_parameters (cvnets/misc/averaging_utils.py:43-55).  Parameters, gradients and EMA copies stay ordinary torch
// tensors (a pointer table addresses them); the moments live in two flat fp32 buffers; learning rate / weight decay per parameter
// group and the step counter are DEVICE values, so the launch can sit inside a captured hipGraph while a scheduler changes the rate.
//
// table[n+1][8] (int64): param ptr, grad ptr, ema ptr (0 = none), state offset, numel, group index, prefix start, unused;
// row n carries only the total in its prefix field.  Thread = 4 consecutive elements of the concatenated index space; every tensor
// starts at a multiple of 4 in that space (and in the moment buffers), so a thread's elements belong to one tensor and move as float4.
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
    const long long* row = table + (size_t)lo * 8;
    const long long k0 = e0 - row[6];  // multiple of 4: the host pads every tensor's start to 4 elements
    const long long numel = row[4];
    if (k0 >= numel) continue;         // padding between tensors
    float* p = reinterpret_cast<float*>(row[0]);
    const float* g = reinterpret_cast<const float*>(row[1]);
    float* ema = reinterpret_cast<float*>(row[2]);
    const long long so = row[3] + k0;
    const float lr = group_hp[row[5] * 2], wd = group_hp[row[5] * 2 + 1];
    const float decay = 1.0f - lr * wd, step_size = lr / bc1;
    float w[4], gr[4], m[4], v[4], ev[4] = {0.f, 0.f, 0.f, 0.f};
    const bool vec = k0 + 3 < numel && ((reinterpret_cast<uintptr_t>(p + k0) | reinterpret_cast<uintptr_t>(g + k0)) & 15) == 0 &&
                     (ema == nullptr || (reinterpret_cast<uintptr_t>(ema + k0) & 15) == 0);
    const int cnt = vec ? 4 : (int)((numel - k0) < 4 ? (numel - k0) : 4);
    if (vec) {
      const float4 a = *reinterpret_cast<const float4*>(p + k0), b = *reinterpret_cast<const float4*>(g + k0);
      const float4 c = *reinterpret_cast<const float4*>(m_flat + so), d = *reinterpret_cast<const float4*>(v_flat + so);
      w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w;  gr[0] = b.x; gr[1] = b.y; gr[2] = b.z; gr[3] = b.w;
      m[0] = c.x; m[1] = c.y; m[2] = c.z; m[3] = c.w;  v[0] = d.x; v[1] = d.y; v[2] = d.z; v[3] = d.w;
      if (ema) { const float4 q = *reinterpret_cast<const float4*>(ema + k0); ev[0] = q.x; ev[1] = q.y; ev[2] = q.z; ev[3] = q.w; }
    } else {
      for (int j = 0; j < cnt; ++j) { w[j] = p[k0 + j]; gr[j] = g[k0 + j]; m[j] = m_flat[so + j]; v[j] = v_flat[so + j]; if (ema) ev[j] = ema[k0 + j]; }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float grad = gr[j] * gs;
      w[j] *= decay;
      m[j] = m[j] + (grad - m[j]) * (1.0f - beta1);
      v[j] = v[j] * beta2 + grad * grad * (1.0f - beta2);
      w[j] -= step_size * (m[j] / (sqrtf(v[j]) / sbc2 + eps));
      ev[j] = ev[j] * (1.0f - ema_momentum) + ema_momentum * w[j];
    }
    if (vec) {
      *reinterpret_cast<float4*>(p + k0) = make_float4(w[0], w[1], w[2], w[3]);
      *reinterpret_cast<float4*>(m_flat + so) = make_float4(m[0], m[1], m[2], m[3]);
      *reinterpret_cast<float4*>(v_flat + so) = make_float4(v[0], v[1], v[2], v[3]);
      if (ema) *reinterpret_cast<float4*>(ema + k0) = make_float4(ev[0], ev[1], ev[2], ev[3]);
    } else {
      for (int j = 0; j < cnt; ++j) { p[k0 + j] = w[j]; m_flat[so + j] = m[j]; v_flat[so + j] = v[j]; if (ema) ema[k0 + j] = ev[j]; }
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
