// Depthwise 3x3 convolution (groups == C), NHWC, HBM-bound: 9 MAC per element.
// Thread = one 8-channel group (its 9x8 weights live in registers) x a run of TW = 4 adjacent output pixels: the 3 x
// ((TW-1)*stride + 3) input window is loaded once per row and reused by every tap / output it feeds, so L1 sees
// 4.5 (stride 1) / 6.75 (stride 2) 16-byte loads per output instead of 9.  A block's pixel lanes walk consecutive pixel runs so
// the vertical halo is served from L1/L2.  Forward also emits the BatchNorm column statistics (sum, sumsq) as per-block partials.
// Fast paths need dilation 1 / padding 1; any other geometry takes the generic per-pixel kernels at the bottom.
//
// Replaces nn.Conv2d(groups=C) inside InvertedResidual (cvnets/modules/mobilenetv2.py:194-207) and its autograd backward.
#include "common.hpp"
#include "cvnets_hip.h"

struct DwParams {
  int B, H, W, Ho, Wo, C, K, stride, pad, dil;
  int xcd;  // XCD-contiguous block mapping
};
__device__ __forceinline__ int dw_block(const DwParams& p) { return p.xcd ? xcd_chunk_id(blockIdx.x, gridDim.x) : (int)blockIdx.x; }

#define DW_TW 4

// block-level reduction of per-thread column sums into part[block][2][C]
__device__ __forceinline__ void dw_reduce_stats(float* red, const float* s1, const float* s2, int C, int ci, int pl, int RL, float* part) {
  if (pl < RL) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      red[(pl * 2 + 0) * C + ci * 8 + j] = s1[j];
      red[(pl * 2 + 1) * C + ci * 8 + j] = s2[j];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    float s = 0.f;
    for (int l = 0; l < RL; ++l) s += red[l * 2 * C + i];
    part[(size_t)blockIdx.x * 2 * C + i] = s;
  }
}

template <typename T, int S>
__global__ __launch_bounds__(256) void dwconv3_fwd_kernel(const T* __restrict__ x, const T* __restrict__ wp /*[9][C]*/, T* __restrict__ y, DwParams p,
                                                          float* __restrict__ stats_part) {
  constexpr int TW = DW_TW, NC = (TW - 1) * S + 3;
  __shared__ float red[4096];
  const int cgs = p.C / 8;
  const int RL = 256 / cgs;
  const int ci = threadIdx.x % cgs, pl = threadIdx.x / cgs;
  float w[9][8];
#pragma unroll
  for (int t = 0; t < 9; ++t) v8_unpack(v8_load<T>(wp + (size_t)t * p.C + ci * 8), w[t]);
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s1[j] = s2[j] = 0.f;
  const int Wg = (p.Wo + TW - 1) / TW;
  const size_t ngroups = (size_t)p.B * p.Ho * Wg;
  if (pl < RL) {
    for (size_t g = (size_t)dw_block(p) * RL + pl; g < ngroups; g += (size_t)gridDim.x * RL) {
      const int wg = (int)(g % Wg);
      const size_t t1 = g / Wg;
      const int ho = (int)(t1 % p.Ho);
      const size_t b = t1 / p.Ho;
      const int wo0 = wg * TW;
      const int wi0 = wo0 * S - 1;
      float acc[TW][8];
#pragma unroll
      for (int t = 0; t < TW; ++t)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[t][j] = 0.f;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int hi = ho * S - 1 + kh;
        if (hi < 0 || hi >= p.H) continue;
        const T* rowp = x + ((b * p.H + hi) * p.W) * p.C + ci * 8;
        V8<T> v[NC];
#pragma unroll
        for (int cix = 0; cix < NC; ++cix) {
          const int wi = wi0 + cix;
          v[cix] = (wi >= 0 && wi < p.W) ? v8_load<T>(rowp + (size_t)wi * p.C) : v8_zero<T>();
        }
#pragma unroll
        for (int cix = 0; cix < NC; ++cix) {
          float f[8];
          v8_unpack(v[cix], f);
#pragma unroll
          for (int t = 0; t < TW; ++t) {
            const int kw = cix - t * S;  // compile-time after unrolling
            if (kw >= 0 && kw < 3) {
#pragma unroll
              for (int j = 0; j < 8; ++j) acc[t][j] += f[j] * w[kh * 3 + kw][j];
            }
          }
        }
      }
#pragma unroll
      for (int t = 0; t < TW; ++t) {
        if (wo0 + t < p.Wo) {
          V8<T> o;
          v8_pack(acc[t], o);
          v8_store<T>(y + (((b * p.Ho + ho) * p.Wo) + wo0 + t) * p.C + ci * 8, o);
          float r[8];
          v8_unpack(o, r);  // statistics of the values as stored
#pragma unroll
          for (int j = 0; j < 8; ++j) { s1[j] += r[j]; s2[j] += r[j] * r[j]; }
        }
      }
    }
  }
  if (stats_part) dw_reduce_stats(red, s1, s2, p.C, ci, pl, RL, stats_part);
}

// dX, stride 1: correlation of dY with the flipped kernel — same sliding window as forward.
template <typename T>
__global__ __launch_bounds__(256) void dwconv3_bwd_x_s1_kernel(const T* __restrict__ dy, const T* __restrict__ wp, T* __restrict__ dx, DwParams p) {
  constexpr int TW = DW_TW, NC = TW + 2;
  const int cgs = p.C / 8;
  const int RL = 256 / cgs;
  const int ci = threadIdx.x % cgs, pl = threadIdx.x / cgs;
  if (pl >= RL) return;
  float w[9][8];
#pragma unroll
  for (int t = 0; t < 9; ++t) v8_unpack(v8_load<T>(wp + (size_t)t * p.C + ci * 8), w[t]);
  const int Wg = (p.W + TW - 1) / TW;
  const size_t ngroups = (size_t)p.B * p.H * Wg;
  for (size_t g = (size_t)dw_block(p) * RL + pl; g < ngroups; g += (size_t)gridDim.x * RL) {
    const int wg = (int)(g % Wg);
    const size_t t1 = g / Wg;
    const int hi = (int)(t1 % p.H);
    const size_t b = t1 / p.H;
    const int wi0 = wg * TW;
    float acc[TW][8];
#pragma unroll
    for (int t = 0; t < TW; ++t)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[t][j] = 0.f;
    // dX[hi][wi] = sum_{kh,kw} dY[hi + 1 - kh][wi + 1 - kw] * w[kh][kw]  (Ho == H, Wo == W)
#pragma unroll
    for (int dh = 0; dh < 3; ++dh) {  // dY row = hi - 1 + dh, i.e. kh = 2 - dh
      const int ho = hi - 1 + dh;
      if (ho < 0 || ho >= p.Ho) continue;
      const T* rowp = dy + ((b * p.Ho + ho) * p.Wo) * p.C + ci * 8;
      V8<T> v[NC];
#pragma unroll
      for (int cix = 0; cix < NC; ++cix) {
        const int wo = wi0 - 1 + cix;
        v[cix] = (wo >= 0 && wo < p.Wo) ? v8_load<T>(rowp + (size_t)wo * p.C) : v8_zero<T>();
      }
#pragma unroll
      for (int cix = 0; cix < NC; ++cix) {
        float f[8];
        v8_unpack(v[cix], f);
#pragma unroll
        for (int t = 0; t < TW; ++t) {
          const int dw = cix - t;  // dY col = wi - 1 + dw, i.e. kw = 2 - dw
          if (dw >= 0 && dw < 3) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[t][j] += f[j] * w[(2 - dh) * 3 + (2 - dw)][j];
          }
        }
      }
    }
#pragma unroll
    for (int t = 0; t < TW; ++t) {
      if (wi0 + t < p.W) {
        V8<T> o;
        v8_pack(acc[t], o);
        v8_store<T>(dx + (((b * p.H + hi) * p.W) + wi0 + t) * p.C + ci * 8, o);
      }
    }
  }
}

// dX, stride 2 (pad 1): input pixel (hi, wi) is touched by the taps kh with (hi + 1 - kh) even: 1 or 2 rows x 1 or 2 columns.
// A thread produces the 2x2 input quad (2qh + {0,1}, 2qw + {0,1}) from the 2x2 dY quad (qh + {0,1}, qw + {0,1}).
template <typename T>
__global__ __launch_bounds__(256) void dwconv3_bwd_x_s2_kernel(const T* __restrict__ dy, const T* __restrict__ wp, T* __restrict__ dx, DwParams p) {
  const int cgs = p.C / 8;
  const int RL = 256 / cgs;
  const int ci = threadIdx.x % cgs, pl = threadIdx.x / cgs;
  if (pl >= RL) return;
  float w[9][8];
#pragma unroll
  for (int t = 0; t < 9; ++t) v8_unpack(v8_load<T>(wp + (size_t)t * p.C + ci * 8), w[t]);
  const int Qh = (p.H + 1) / 2, Qw = (p.W + 1) / 2;
  const size_t nquads = (size_t)p.B * Qh * Qw;
  for (size_t g = (size_t)dw_block(p) * RL + pl; g < nquads; g += (size_t)gridDim.x * RL) {
    const int qw = (int)(g % Qw);
    const size_t t1 = g / Qw;
    const int qh = (int)(t1 % Qh);
    const size_t b = t1 / Qh;
    float d[2][2][8];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int c2 = 0; c2 < 2; ++c2) {
        const int ho = qh + a, wo = qw + c2;
        if (ho < p.Ho && wo < p.Wo) {
          v8_unpack(v8_load<T>(dy + (((b * p.Ho + ho) * p.Wo) + wo) * p.C + ci * 8), d[a][c2]);
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) d[a][c2][j] = 0.f;
        }
      }
    // hi = 2qh (even): kh = 1 -> ho = qh.   hi = 2qh + 1 (odd): kh = 2 -> ho = qh, kh = 0 -> ho = qh + 1.   wi likewise.
#pragma unroll
    for (int ph = 0; ph < 2; ++ph)
#pragma unroll
      for (int pw = 0; pw < 2; ++pw) {
        const int hi = 2 * qh + ph, wi = 2 * qw + pw;
        if (hi >= p.H || wi >= p.W) continue;
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          if (ph == 0 && a == 1) continue;
          const int kh = ph == 0 ? 1 : (a == 0 ? 2 : 0);
#pragma unroll
          for (int c2 = 0; c2 < 2; ++c2) {
            if (pw == 0 && c2 == 1) continue;
            const int kw = pw == 0 ? 1 : (c2 == 0 ? 2 : 0);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += d[a][c2][j] * w[kh * 3 + kw][j];
          }
        }
        V8<T> o;
        v8_pack(acc, o);
        v8_store<T>(dx + (((b * p.H + hi) * p.W) + wi) * p.C + ci * 8, o);
      }
  }
}

// dW[c][tap] partials: part[block][c*9 + tap] = sum over this block's output pixels of dY * x(shifted)
template <typename T, int S>
__global__ __launch_bounds__(256) void dwconv3_bwd_w_kernel(const T* __restrict__ x, const T* __restrict__ dy, DwParams p, float* __restrict__ part) {
  constexpr int TW = DW_TW, NC = (TW - 1) * S + 3;
  __shared__ float red[2048];
  const int cgs = p.C / 8;
  const int RL = 256 / cgs;
  const int ci = threadIdx.x % cgs, pl = threadIdx.x / cgs;
  float acc[9][8];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[t][j] = 0.f;
  const int Wg = (p.Wo + TW - 1) / TW;
  const size_t ngroups = (size_t)p.B * p.Ho * Wg;
  if (pl < RL) {
    for (size_t g = (size_t)dw_block(p) * RL + pl; g < ngroups; g += (size_t)gridDim.x * RL) {
      const int wg = (int)(g % Wg);
      const size_t t1 = g / Wg;
      const int ho = (int)(t1 % p.Ho);
      const size_t b = t1 / p.Ho;
      const int wo0 = wg * TW;
      const int wi0 = wo0 * S - 1;
      float d[TW][8];
#pragma unroll
      for (int t = 0; t < TW; ++t) {
        if (wo0 + t < p.Wo) {
          v8_unpack(v8_load<T>(dy + (((b * p.Ho + ho) * p.Wo) + wo0 + t) * p.C + ci * 8), d[t]);
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) d[t][j] = 0.f;
        }
      }
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int hi = ho * S - 1 + kh;
        if (hi < 0 || hi >= p.H) continue;
        const T* rowp = x + ((b * p.H + hi) * p.W) * p.C + ci * 8;
        V8<T> v[NC];
#pragma unroll
        for (int cix = 0; cix < NC; ++cix) {
          const int wi = wi0 + cix;
          v[cix] = (wi >= 0 && wi < p.W) ? v8_load<T>(rowp + (size_t)wi * p.C) : v8_zero<T>();
        }
#pragma unroll
        for (int cix = 0; cix < NC; ++cix) {
          float f[8];
          v8_unpack(v[cix], f);
#pragma unroll
          for (int t = 0; t < TW; ++t) {
            const int kw = cix - t * S;
            if (kw >= 0 && kw < 3) {
#pragma unroll
              for (int j = 0; j < 8; ++j) acc[kh * 3 + kw][j] += f[j] * d[t][j];
            }
          }
        }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    __syncthreads();
    if (pl < RL) {
#pragma unroll
      for (int j = 0; j < 8; ++j) red[pl * p.C + ci * 8 + j] = acc[t][j];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < p.C; c += 256) {
      float s = 0.f;
      for (int l = 0; l < RL; ++l) s += red[l * p.C + c];
      part[(size_t)blockIdx.x * p.C * 9 + (size_t)c * 9 + t] = s;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// generic geometry (any stride / padding / dilation): one output (or input) pixel per thread iteration
// ---------------------------------------------------------------------------------------------
template <typename T, int K>
__global__ __launch_bounds__(256) void dwconv_fwd_generic_kernel(const T* __restrict__ x, const T* __restrict__ wp, T* __restrict__ y, DwParams p,
                                                                 float* __restrict__ stats_part) {
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
      v8_unpack(o, r);
#pragma unroll
      for (int j = 0; j < 8; ++j) { s1[j] += r[j]; s2[j] += r[j] * r[j]; }
    }
  }
  if (stats_part) dw_reduce_stats(red, s1, s2, p.C, ci, pl, RL, stats_part);
}

template <typename T, int K>
__global__ __launch_bounds__(256) void dwconv_bwd_x_generic_kernel(const T* __restrict__ dy, const T* __restrict__ wp, T* __restrict__ dx, DwParams p) {
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

template <typename T, int K>
__global__ __launch_bounds__(256) void dwconv_bwd_w_generic_kernel(const T* __restrict__ x, const T* __restrict__ dy, DwParams p, float* __restrict__ part) {
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

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static int dw_grid(size_t nunits, int C, int cap) {
  const int RL = 256 / (C / 8);
  size_t g = (nunits + (size_t)RL * 2 - 1) / ((size_t)RL * 2);
  if (g > (size_t)cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}
static bool dw_fast(int K, int stride, int pad, int dil) { return K == 3 && dil == 1 && pad == 1 && (stride == 1 || stride == 2); }
static size_t dw_out_units(int B, int Ho, int Wo, bool fast) { return fast ? (size_t)B * Ho * ((Wo + DW_TW - 1) / DW_TW) : (size_t)B * Ho * Wo; }

extern "C" int cvh_dwconv_rows(int B, int Ho, int Wo, int C, int K, int stride, int pad, int dil) {
  if (C % 8 || C > 2048 || C <= 0) return -2;
  return dw_grid(dw_out_units(B, Ho, Wo, dw_fast(K, stride, pad, dil)), C, 1024);
}
extern "C" int cvh_dwconv_bwd_w_rows(int B, int Ho, int Wo, int C, int K, int stride, int pad, int dil) {
  if (C % 8 || C > 2048 || C <= 0) return -2;
  return dw_grid(dw_out_units(B, Ho, Wo, dw_fast(K, stride, pad, dil)), C, 512);
}

extern "C" int cvh_dwconv_fwd(int dtype, const void* x, const void* wp, void* y, int B, int H, int W, int Ho, int Wo, int C, int K,
                              int stride, int pad, int dil, float* stats_part, void* stream) {
  if (C % 8 || C > 2048 || K != 3) return -2;
  DwParams p{B, H, W, Ho, Wo, C, K, stride, pad, dil, cvh_tune_get(CVH_TUNE_DW_XCD)};
  hipStream_t st = (hipStream_t)stream;
  const bool fast = dw_fast(K, stride, pad, dil);
  const int g = cvh_dwconv_rows(B, Ho, Wo, C, K, stride, pad, dil);
  if (dtype == CVH_DT_BF16) {
    if (fast && stride == 1) hipLaunchKernelGGL((dwconv3_fwd_kernel<bf16_t, 1>), dim3(g), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)wp, (bf16_t*)y, p, stats_part);
    else if (fast) hipLaunchKernelGGL((dwconv3_fwd_kernel<bf16_t, 2>), dim3(g), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)wp, (bf16_t*)y, p, stats_part);
    else hipLaunchKernelGGL((dwconv_fwd_generic_kernel<bf16_t, 3>), dim3(g), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)wp, (bf16_t*)y, p, stats_part);
  } else if (dtype == CVH_DT_F32) {
    if (fast && stride == 1) hipLaunchKernelGGL((dwconv3_fwd_kernel<float, 1>), dim3(g), dim3(256), 0, st, (const float*)x, (const float*)wp, (float*)y, p, stats_part);
    else if (fast) hipLaunchKernelGGL((dwconv3_fwd_kernel<float, 2>), dim3(g), dim3(256), 0, st, (const float*)x, (const float*)wp, (float*)y, p, stats_part);
    else hipLaunchKernelGGL((dwconv_fwd_generic_kernel<float, 3>), dim3(g), dim3(256), 0, st, (const float*)x, (const float*)wp, (float*)y, p, stats_part);
  } else return -1;
  CVH_CHECK_LAUNCH();
  return 0;
}

extern "C" int cvh_dwconv_bwd_x(int dtype, const void* dy, const void* wp, void* dx, int B, int H, int W, int Ho, int Wo, int C, int K,
                                int stride, int pad, int dil, void* stream) {
  if (C % 8 || C > 2048 || K != 3) return -2;
  DwParams p{B, H, W, Ho, Wo, C, K, stride, pad, dil, cvh_tune_get(CVH_TUNE_DW_XCD)};
  hipStream_t st = (hipStream_t)stream;
  const bool fast = dw_fast(K, stride, pad, dil);
  if (fast && stride == 1) {
    const int g = dw_grid((size_t)B * H * ((W + DW_TW - 1) / DW_TW), C, 2048);
    if (dtype == CVH_DT_BF16) hipLaunchKernelGGL((dwconv3_bwd_x_s1_kernel<bf16_t>), dim3(g), dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)wp, (bf16_t*)dx, p);
    else if (dtype == CVH_DT_F32) hipLaunchKernelGGL((dwconv3_bwd_x_s1_kernel<float>), dim3(g), dim3(256), 0, st, (const float*)dy, (const float*)wp, (float*)dx, p);
    else return -1;
  } else if (fast) {
    const int g = dw_grid((size_t)B * ((H + 1) / 2) * ((W + 1) / 2), C, 2048);
    if (dtype == CVH_DT_BF16) hipLaunchKernelGGL((dwconv3_bwd_x_s2_kernel<bf16_t>), dim3(g), dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)wp, (bf16_t*)dx, p);
    else if (dtype == CVH_DT_F32) hipLaunchKernelGGL((dwconv3_bwd_x_s2_kernel<float>), dim3(g), dim3(256), 0, st, (const float*)dy, (const float*)wp, (float*)dx, p);
    else return -1;
  } else {
    const int g = dw_grid((size_t)B * H * W, C, 2048);
    if (dtype == CVH_DT_BF16) hipLaunchKernelGGL((dwconv_bwd_x_generic_kernel<bf16_t, 3>), dim3(g), dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)wp, (bf16_t*)dx, p);
    else if (dtype == CVH_DT_F32) hipLaunchKernelGGL((dwconv_bwd_x_generic_kernel<float, 3>), dim3(g), dim3(256), 0, st, (const float*)dy, (const float*)wp, (float*)dx, p);
    else return -1;
  }
  CVH_CHECK_LAUNCH();
  return 0;
}

extern "C" int cvh_dwconv_bwd_w(int dtype, const void* x, const void* dy, float* part, int B, int H, int W, int Ho, int Wo, int C, int K,
                                int stride, int pad, int dil, void* stream) {
  if (C % 8 || C > 2048 || K != 3) return -2;
  DwParams p{B, H, W, Ho, Wo, C, K, stride, pad, dil, cvh_tune_get(CVH_TUNE_DW_XCD)};
  hipStream_t st = (hipStream_t)stream;
  const bool fast = dw_fast(K, stride, pad, dil);
  const int g = cvh_dwconv_bwd_w_rows(B, Ho, Wo, C, K, stride, pad, dil);
  if (dtype == CVH_DT_BF16) {
    if (fast && stride == 1) hipLaunchKernelGGL((dwconv3_bwd_w_kernel<bf16_t, 1>), dim3(g), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)dy, p, part);
    else if (fast) hipLaunchKernelGGL((dwconv3_bwd_w_kernel<bf16_t, 2>), dim3(g), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)dy, p, part);
    else hipLaunchKernelGGL((dwconv_bwd_w_generic_kernel<bf16_t, 3>), dim3(g), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)dy, p, part);
  } else if (dtype == CVH_DT_F32) {
    if (fast && stride == 1) hipLaunchKernelGGL((dwconv3_bwd_w_kernel<float, 1>), dim3(g), dim3(256), 0, st, (const float*)x, (const float*)dy, p, part);
    else if (fast) hipLaunchKernelGGL((dwconv3_bwd_w_kernel<float, 2>), dim3(g), dim3(256), 0, st, (const float*)x, (const float*)dy, p, part);
    else hipLaunchKernelGGL((dwconv_bwd_w_generic_kernel<float, 3>), dim3(g), dim3(256), 0, st, (const float*)x, (const float*)dy, p, part);
  } else return -1;
  CVH_CHECK_LAUNCH();
  return 0;
}
