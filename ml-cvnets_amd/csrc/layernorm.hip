// LayerNorm over the channel (last) dimension of a [rows][C] token matrix: one 64-lane wave per row,
// row held in registers (C <= 1024), two-pass mean / variance in fp32.
// Replaces nn.LayerNorm (cvnets/layers/normalization/layer_norm.py:14-72, channel-last branch) + backward.
#include "common.hpp"
#include "cvnets_hip.h"

#define LN_MAXIT 4  // 64 lanes * 4 elements * 4 iterations = 1024 channels

template <typename T>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     T* __restrict__ y, float* __restrict__ mean_out, float* __restrict__ rstd_out, size_t rows, int C,
                                                     float eps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = C / 4;
  for (size_t row = (size_t)blockIdx.x * 4 + wave; row < rows; row += (size_t)gridDim.x * 4) {
    float v[LN_MAXIT][4];
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < LN_MAXIT; ++it) {
      const int ch = lane + it * 64;
      if (ch < nch) {
        v4_unpack(v4_load<T>(x + row * C + ch * 4), v[it]);
        s += v[it][0] + v[it][1] + v[it][2] + v[it][3];
      } else {
        v[it][0] = v[it][1] = v[it][2] = v[it][3] = 0.f;
      }
    }
    const float mu = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int it = 0; it < LN_MAXIT; ++it) {
      const int ch = lane + it * 64;
      if (ch < nch) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { float d = v[it][e] - mu; q += d * d; }
      }
    }
    const float var = wave_sum(q) / (float)C;
    const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
    for (int it = 0; it < LN_MAXIT; ++it) {
      const int ch = lane + it * 64;
      if (ch < nch) {
        const float4 g = *reinterpret_cast<const float4*>(gamma + ch * 4);
        const float4 b = *reinterpret_cast<const float4*>(beta + ch * 4);
        float o[4];
        o[0] = (v[it][0] - mu) * rstd * g.x + b.x;
        o[1] = (v[it][1] - mu) * rstd * g.y + b.y;
        o[2] = (v[it][2] - mu) * rstd * g.z + b.z;
        o[3] = (v[it][3] - mu) * rstd * g.w + b.w;
        V4<T> ov;
        v4_pack(o, ov);
        v4_store<T>(y + row * C + ch * 4, ov);
      }
    }
    if (lane == 0 && mean_out) { mean_out[row] = mu; rstd_out[row] = rstd; }
  }
}

// dx = rstd * (g - mean(g) - xhat * mean(g*xhat)),  g = dy*gamma ; partials of dgamma = sum dy*xhat, dbeta = sum dy
template <typename T>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, const float* __restrict__ gamma,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd, T* __restrict__ dx,
                                                     float* __restrict__ part /*[grid][2][C]*/, size_t rows, int C) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [4 waves][2][C]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = C / 4;
  float dg[LN_MAXIT][4], db[LN_MAXIT][4], gm[LN_MAXIT][4];
#pragma unroll
  for (int it = 0; it < LN_MAXIT; ++it) {
    const int ch = lane + it * 64;
#pragma unroll
    for (int e = 0; e < 4; ++e) { dg[it][e] = 0.f; db[it][e] = 0.f; gm[it][e] = (ch < nch) ? gamma[ch * 4 + e] : 0.f; }
  }
  for (size_t row = (size_t)blockIdx.x * 4 + wave; row < rows; row += (size_t)gridDim.x * 4) {
    const float mu = mean[row], rs = rstd[row];
    float xh[LN_MAXIT][4], g[LN_MAXIT][4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int it = 0; it < LN_MAXIT; ++it) {
      const int ch = lane + it * 64;
      if (ch < nch) {
        float xv[4], dv[4];
        v4_unpack(v4_load<T>(x + row * C + ch * 4), xv);
        v4_unpack(v4_load<T>(dy + row * C + ch * 4), dv);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          xh[it][e] = (xv[e] - mu) * rs;
          g[it][e] = dv[e] * gm[it][e];
          s1 += g[it][e];
          s2 += g[it][e] * xh[it][e];
          dg[it][e] += dv[e] * xh[it][e];
          db[it][e] += dv[e];
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) { xh[it][e] = 0.f; g[it][e] = 0.f; }
      }
    }
    const float m1 = wave_sum(s1) / (float)C, m2 = wave_sum(s2) / (float)C;
#pragma unroll
    for (int it = 0; it < LN_MAXIT; ++it) {
      const int ch = lane + it * 64;
      if (ch < nch) {
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = rs * (g[it][e] - m1 - xh[it][e] * m2);
        V4<T> ov;
        v4_pack(o, ov);
        v4_store<T>(dx + row * C + ch * 4, ov);
      }
    }
  }
#pragma unroll
  for (int it = 0; it < LN_MAXIT; ++it) {
    const int ch = lane + it * 64;
    if (ch < nch) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        red[(wave * 2 + 0) * C + ch * 4 + e] = dg[it][e];
        red[(wave * 2 + 1) * C + ch * 4 + e] = db[it][e];
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) s += red[w * 2 * C + i];
    part[(size_t)blockIdx.x * 2 * C + i] = s;
  }
}

extern "C" int cvh_ln_bwd_rows(long long rows) {
  long long g = (rows + 15) / 16;
  const int cap = cvh_tune_get(CVH_TUNE_COLRED_ROWS);  // workgroups = partial rows of the dgamma / dbeta reduction
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

extern "C" int cvh_layernorm_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                                 long long rows, int C, float eps, void* stream) {
  if (C % 4 || C > 64 * 4 * LN_MAXIT || C <= 0) return -2;
  long long g = (rows + 3) / 4;
  if (g > 8192) g = 8192;
  if (g < 1) g = 1;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CVH_DT_BF16) hipLaunchKernelGGL((ln_fwd_kernel<bf16_t>), dim3((int)g), dim3(256), 0, st, (const bf16_t*)x, gamma, beta, (bf16_t*)y, mean, rstd, (size_t)rows, C, eps);
  else if (dtype == CVH_DT_F32) hipLaunchKernelGGL((ln_fwd_kernel<float>), dim3((int)g), dim3(256), 0, st, (const float*)x, gamma, beta, (float*)y, mean, rstd, (size_t)rows, C, eps);
  else return -1;
  CVH_CHECK_LAUNCH();
  return 0;
}
extern "C" int cvh_layernorm_bwd(int dtype, const void* x, const void* dy, const float* gamma, const float* mean, const float* rstd, void* dx,
                                 float* part, long long rows, int C, void* stream) {
  if (C % 4 || C > 64 * 4 * LN_MAXIT || C <= 0) return -2;
  int g = cvh_ln_bwd_rows(rows);
  size_t smem = (size_t)4 * 2 * C * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CVH_DT_BF16) hipLaunchKernelGGL((ln_bwd_kernel<bf16_t>), dim3(g), dim3(256), smem, st, (const bf16_t*)x, (const bf16_t*)dy, gamma, mean, rstd, (bf16_t*)dx, part, (size_t)rows, C);
  else if (dtype == CVH_DT_F32) hipLaunchKernelGGL((ln_bwd_kernel<float>), dim3(g), dim3(256), smem, st, (const float*)x, (const float*)dy, gamma, mean, rstd, (float*)dx, part, (size_t)rows, C);
  else return -1;
  CVH_CHECK_LAUNCH();
  return 0;
}
