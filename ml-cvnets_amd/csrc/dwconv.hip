// Depthwise KxK convolution (groups == C), NHWC, HBM-bound: 9 MAC per element.
// Thread = one 8-channel group; a block's pixel lanes walk consecutive output pixels so the 3x3 halo
// is served from L1/L2.  The weights of the thread's channel group live in registers for the whole
// grid-stride loop.  Forward optionally emits BatchNorm column statistics (sum, sumsq) as partials.
//
// Replaces nn.Conv2d(groups=C) inside InvertedResidual (cvnets/modules/mobilenetv2.py:194-207) and
// its autograd backward.
#include "common.hpp"
#include "cvnets_hip.h"

struct DwParams {
  int B, H, W, Ho, Wo, C, K, stride, pad, dil;
};

template <typename T, int K>
__global__ __launch_bounds__(256) void dwconv_fwd_kernel(const T* __restrict__ x, const T* __restrict__ wp /*[K*K][C]*/, T* __restrict__ y,
                                                         DwParams p, float* __restrict__ stats_part) {
  __shared__ float red[4096];
  const int cgs = p.C / 8;
  const int RL = 256 / cgs;
  const int ci = threadIdx.x % cgs, pl = threadIdx.x / cgs;
  float w[K * K][8];
#pragma unroll
  for (int t = 0; t < K * K; ++t) v8_unpack(v8_load<T>(wp + (size_t)t * p.C + ci * 8), w[t]);
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s1[j] = s2[j] = 0.f;
  const size_t npix = (size_t)p.B * p.Ho * p.Wo;
  if (pl < RL) {
    for (size_t pix = (size_t)blockIdx.x * RL + pl; pix < npix; pix += (size_t)gridDim.x * RL) {
      const int wo = (int)(pix % p.Wo);
      const size_t t1 = pix / p.Wo;
      const int ho = (int)(t1 % p.Ho);
      const size_t b = t1 / p.Ho;
      float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int kh = 0; kh < K; ++kh) {
        const int hi = ho * p.stride - p.pad + kh * p.dil;
        if (hi < 0 || hi >= p.H) continue;
#pragma unroll
        for (int kw = 0; kw < K; ++kw) {
          const int wi = wo * p.stride - p.pad + kw * p.dil;
          if (wi < 0 || wi >= p.W) continue;
          float f[8];
          v8_unpack(v8_load<T>(x + ((b * p.H + hi) * p.W + wi) * p.C + ci * 8), f);
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] += f[j] * w[kh * K + kw][j];
        }
      }
      V8<T> o;
      v8_pack(acc, o);
      v8_store<T>(y + pix * p.C + ci * 8, o);
      float r[8];
      v8_unpack(o, r);  // statistics of the values as stored
#pragma unroll
      for (int j = 0; j < 8; ++j) { s1[j] += r[j]; s2[j] += r[j] * r[j]; }
    }
  }
  if (stats_part) {
    if (pl < RL) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        red[(pl * 2 + 0) * p.C + ci * 8 + j] = s1[j];
        red[(pl * 2 + 1) * p.C + ci * 8 + j] = s2[j];
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * p.C; i += 256) {
      float s = 0.f;
      for (int l = 0; l < RL; ++l) s += red[l * 2 * p.C + i];
      stats_part[(size_t)blockIdx.x * 2 * p.C + i] = s;
    }
  }
}

// dX[b,hi,wi,c] = sum_taps dY[b,ho,wo,c] * w[tap][c],  ho = (hi + pad - kh*dil)/stride when integral & in range
template <typename T, int K>
__global__ __launch_bounds__(256) void dwconv_bwd_x_kernel(const T* __restrict__ dy, const T* __restrict__ wp, T* __restrict__ dx, DwParams p) {
  const int cgs = p.C / 8;
  const int RL = 256 / cgs;
  const int ci = threadIdx.x % cgs, pl = threadIdx.x / cgs;
  float w[K * K][8];
#pragma unroll
  for (int t = 0; t < K * K; ++t) v8_unpack(v8_load<T>(wp + (size_t)t * p.C + ci * 8), w[t]);
  const size_t npix = (size_t)p.B * p.H * p.W;
  if (pl >= RL) return;
  for (size_t pix = (size_t)blockIdx.x * RL + pl; pix < npix; pix += (size_t)gridDim.x * RL) {
    const int wi = (int)(pix % p.W);
    const size_t t1 = pix / p.W;
    const int hi = (int)(t1 % p.H);
    const size_t b = t1 / p.H;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int kh = 0; kh < K; ++kh) {
      const int hn = hi + p.pad - kh * p.dil;
      if (hn < 0 || (hn % p.stride) != 0) continue;
      const int ho = hn / p.stride;
      if (ho >= p.Ho) continue;
#pragma unroll
      for (int kw = 0; kw < K; ++kw) {
        const int wn = wi + p.pad - kw * p.dil;
        if (wn < 0 || (wn % p.stride) != 0) continue;
        const int wo = wn / p.stride;
        if (wo >= p.Wo) continue;
        float f[8];
        v8_unpack(v8_load<T>(dy + ((b * p.Ho + ho) * p.Wo + wo) * p.C + ci * 8), f);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += f[j] * w[kh * K + kw][j];
      }
    }
    V8<T> o;
    v8_pack(acc, o);
    v8_store<T>(dx + pix * p.C + ci * 8, o);
  }
}

// dW[c][tap] partials: part[block][c*K*K + tap] = sum over this block's output pixels of dY * x(shifted)
template <typename T, int K>
__global__ __launch_bounds__(256) void dwconv_bwd_w_kernel(const T* __restrict__ x, const T* __restrict__ dy, DwParams p, float* __restrict__ part) {
  __shared__ float red[2048];
  const int cgs = p.C / 8;
  const int RL = 256 / cgs;
  const int ci = threadIdx.x % cgs, pl = threadIdx.x / cgs;
  float acc[K * K][8];
#pragma unroll
  for (int t = 0; t < K * K; ++t)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[t][j] = 0.f;
  const size_t npix = (size_t)p.B * p.Ho * p.Wo;
  if (pl < RL) {
    for (size_t pix = (size_t)blockIdx.x * RL + pl; pix < npix; pix += (size_t)gridDim.x * RL) {
      const int wo = (int)(pix % p.Wo);
      const size_t t1 = pix / p.Wo;
      const int ho = (int)(t1 % p.Ho);
      const size_t b = t1 / p.Ho;
      float d[8];
      v8_unpack(v8_load<T>(dy + pix * p.C + ci * 8), d);
#pragma unroll
      for (int kh = 0; kh < K; ++kh) {
        const int hi = ho * p.stride - p.pad + kh * p.dil;
        if (hi < 0 || hi >= p.H) continue;
#pragma unroll
        for (int kw = 0; kw < K; ++kw) {
          const int wi = wo * p.stride - p.pad + kw * p.dil;
          if (wi < 0 || wi >= p.W) continue;
          float f[8];
          v8_unpack(v8_load<T>(x + ((b * p.H + hi) * p.W + wi) * p.C + ci * 8), f);
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[kh * K + kw][j] += f[j] * d[j];
        }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < K * K; ++t) {
    __syncthreads();
    if (pl < RL) {
#pragma unroll
      for (int j = 0; j < 8; ++j) red[pl * p.C + ci * 8 + j] = acc[t][j];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < p.C; c += 256) {
      float s = 0.f;
      for (int l = 0; l < RL; ++l) s += red[l * p.C + c];
      part[(size_t)blockIdx.x * p.C * K * K + (size_t)c * K * K + t] = s;
    }
  }
}

static int dw_grid(size_t npix, int C) {
  const int RL = 256 / (C / 8);
  size_t g = (npix + (size_t)RL * 4 - 1) / ((size_t)RL * 4);
  if (g > 1024) g = 1024;  // fwd/bwd_x grid; also the number of BN-statistics partial rows
  if (g < 1) g = 1;
  return (int)g;
}

extern "C" int cvh_dwconv_rows(int B, int Ho, int Wo, int C) {
  if (C % 8 || C > 2048 || C <= 0) return -2;
  return dw_grid((size_t)B * Ho * Wo, C);
}
extern "C" int cvh_dwconv_bwd_w_rows(int B, int Ho, int Wo, int C) {
  if (C % 8 || C > 2048 || C <= 0) return -2;
  int g = dw_grid((size_t)B * Ho * Wo, C);
  return g > 512 ? 512 : g;
}

extern "C" int cvh_dwconv_fwd(int dtype, const void* x, const void* wp, void* y, int B, int H, int W, int Ho, int Wo, int C, int K,
                              int stride, int pad, int dil, float* stats_part, void* stream) {
  if (C % 8 || C > 2048 || K != 3) return -2;
  DwParams p{B, H, W, Ho, Wo, C, K, stride, pad, dil};
  int g = dw_grid((size_t)B * Ho * Wo, C);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CVH_DT_BF16) hipLaunchKernelGGL((dwconv_fwd_kernel<bf16_t, 3>), dim3(g), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)wp, (bf16_t*)y, p, stats_part);
  else if (dtype == CVH_DT_F32) hipLaunchKernelGGL((dwconv_fwd_kernel<float, 3>), dim3(g), dim3(256), 0, st, (const float*)x, (const float*)wp, (float*)y, p, stats_part);
  else return -1;
  CVH_CHECK_LAUNCH();
  return 0;
}
extern "C" int cvh_dwconv_bwd_x(int dtype, const void* dy, const void* wp, void* dx, int B, int H, int W, int Ho, int Wo, int C, int K,
                                int stride, int pad, int dil, void* stream) {
  if (C % 8 || C > 2048 || K != 3) return -2;
  DwParams p{B, H, W, Ho, Wo, C, K, stride, pad, dil};
  int g = dw_grid((size_t)B * H * W, C);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CVH_DT_BF16) hipLaunchKernelGGL((dwconv_bwd_x_kernel<bf16_t, 3>), dim3(g), dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)wp, (bf16_t*)dx, p);
  else if (dtype == CVH_DT_F32) hipLaunchKernelGGL((dwconv_bwd_x_kernel<float, 3>), dim3(g), dim3(256), 0, st, (const float*)dy, (const float*)wp, (float*)dx, p);
  else return -1;
  CVH_CHECK_LAUNCH();
  return 0;
}
extern "C" int cvh_dwconv_bwd_w(int dtype, const void* x, const void* dy, float* part, int B, int H, int W, int Ho, int Wo, int C, int K,
                                int stride, int pad, int dil, void* stream) {
  if (C % 8 || C > 2048 || K != 3) return -2;
  DwParams p{B, H, W, Ho, Wo, C, K, stride, pad, dil};
  int g = cvh_dwconv_bwd_w_rows(B, Ho, Wo, C);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CVH_DT_BF16) hipLaunchKernelGGL((dwconv_bwd_w_kernel<bf16_t, 3>), dim3(g), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)dy, p, part);
  else if (dtype == CVH_DT_F32) hipLaunchKernelGGL((dwconv_bwd_w_kernel<float, 3>), dim3(g), dim3(256), 0, st, (const float*)x, (const float*)dy, p, part);
  else return -1;
  CVH_CHECK_LAUNCH();
  return 0;
}
