// MobileViTv2 global representation on NHWC feature maps (all HBM-bound, 16 B per lane):
//   * GroupNorm(num_groups = 1) == the reference's "layer_norm_2d": per-SAMPLE statistics over C*H*W, per-channel affine
//     (cvnets/layers/normalization/layer_norm.py:75-108), forward and backward;
//   * LinearSelfAttention (cvnets/layers/linear_attention.py:147-162), forward and backward, evaluated directly on the
//     feature map: the reference unfolds [B,C,H,W] -> [B,C,P,N] (F.unfold, mobilevit_block.py:526-540) only so that the
//     softmax / context sum run over the N patches that share a pixel position p; here the (b, p) group is addressed by
//     stride ("row(n)") so unfold and fold cost nothing.
//
// qkv tensor layout ("kvq", produced by one 1x1-conv GEMM with permuted weight rows): [M = B*H*W][LD = 2C + 8]
//   cols [0,C) key, [C,2C) value, col 2C query, cols 2C+1..2C+7 zero padding (keeps every row 16-byte aligned).
#include "common.hpp"
#include "cvnets_hip.h"

// ---------------------------------------------------------------------------------------------
// block-level reductions (256 threads = 4 waves)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum(float v, float* scratch /* >= 8 floats */) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
  __syncthreads();
  return scratch[0] + scratch[1] + scratch[2] + scratch[3];
}
__device__ __forceinline__ float block_max(float v, float* scratch) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(scratch[0], scratch[1]), fmaxf(scratch[2], scratch[3]));
}

// =============================================================================================
// GroupNorm(1, C)
// =============================================================================================
// part[(b*chunks + chunk)*2 + {0,1}] = (sum, sum of squares) over this chunk of sample b (a contiguous NHWC span)
template <typename T>
__global__ void __launch_bounds__(256) gn_stats_kernel(const T* __restrict__ x, float* __restrict__ part, size_t per_sample, int chunks) {
  __shared__ float scr[8];
  const int b = blockIdx.y, chunk = blockIdx.x;
  const size_t n8 = per_sample / 8;
  const size_t lo = n8 * chunk / chunks, hi = n8 * (chunk + 1) / chunks;
  const T* base = x + (size_t)b * per_sample;
  float s = 0.f, q = 0.f;
  for (size_t i = lo + threadIdx.x; i < hi; i += 256) {
    float f[8];
    v8_unpack(v8_load<T>(base + i * 8), f);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      s += f[j];
      q += f[j] * f[j];
    }
  }
  s = block_sum(s, scr);
  q = block_sum(q, scr + 4);
  if (threadIdx.x == 0) {
    part[((size_t)b * chunks + chunk) * 2] = s;
    part[((size_t)b * chunks + chunk) * 2 + 1] = q;
  }
}
// stats[b*2] = mean, stats[b*2+1] = 1/sqrt(var + eps)   (biased variance, fp64 combine)
__global__ void gn_finalize_kernel(const float* __restrict__ part, float* __restrict__ stats, int B, int chunks, double n, float eps) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double s = 0, q = 0;
  for (int c = 0; c < chunks; ++c) {
    s += part[((size_t)b * chunks + c) * 2];
    q += part[((size_t)b * chunks + c) * 2 + 1];
  }
  const double mean = s / n;
  double var = q / n - mean * mean;
  if (var < 0) var = 0;
  stats[b * 2] = (float)mean;
  stats[b * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
}

// thread -> (channel group g = t % cg, row lane rl = t / cg); RL = 256 / cg row lanes, threads beyond RL*cg idle
template <typename T>
__global__ void __launch_bounds__(256) gn_apply_kernel(const T* __restrict__ x, const float* __restrict__ stats, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, T* __restrict__ y, int HW, int C, int chunks) {
  const int cg = C / 8, RL = 256 / cg;
  const int g = threadIdx.x % cg, rl = threadIdx.x / cg;
  if (rl >= RL) return;
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int r0 = (int)((size_t)HW * chunk / chunks), r1 = (int)((size_t)HW * (chunk + 1) / chunks);
  const float mean = stats[b * 2], rstd = stats[b * 2 + 1];
  float a[8], c[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    a[j] = rstd * gamma[g * 8 + j];
    c[j] = beta[g * 8 + j] - mean * a[j];
  }
  const size_t base = (size_t)b * HW * C + (size_t)g * 8;
  for (int r = r0 + rl; r < r1; r += RL) {
    float f[8];
    v8_unpack(v8_load<T>(x + base + (size_t)r * C), f);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = f[j] * a[j] + c[j];
    V8<T> o;
    v8_pack(f, o);
    v8_store<T>(y + base + (size_t)r * C, o);
  }
}

// part[((b*chunks + chunk)*2 + k)*C + c] : k = 0 -> sum_rows dy, k = 1 -> sum_rows dy * xhat
template <typename T>
__global__ void __launch_bounds__(256) gn_bwd_reduce_kernel(const T* __restrict__ x, const T* __restrict__ dy, const float* __restrict__ stats,
                                                            float* __restrict__ part, int HW, int C, int chunks) {
  extern __shared__ float red[];  // [RL][2][C]
  const int cg = C / 8, RL = 256 / cg;
  const int g = threadIdx.x % cg, rl = threadIdx.x / cg;
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int r0 = (int)((size_t)HW * chunk / chunks), r1 = (int)((size_t)HW * (chunk + 1) / chunks);
  const float mean = stats[b * 2], rstd = stats[b * 2 + 1];
  float sg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sx[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const size_t base = (size_t)b * HW * C + (size_t)g * 8;
  if (rl < RL) {
    for (int r = r0 + rl; r < r1; r += RL) {
      float f[8], d[8];
      v8_unpack(v8_load<T>(x + base + (size_t)r * C), f);
      v8_unpack(v8_load<T>(dy + base + (size_t)r * C), d);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        sg[j] += d[j];
        sx[j] += d[j] * ((f[j] - mean) * rstd);
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      red[(rl * 2 + 0) * C + g * 8 + j] = sg[j];
      red[(rl * 2 + 1) * C + g * 8 + j] = sx[j];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    float s = 0.f;
    for (int l = 0; l < RL; ++l) s += red[l * 2 * C + i];
    part[((size_t)b * chunks + chunk) * 2 * C + i] = s;
  }
}
// coeff[b*2] = mean(dy*gamma), coeff[b*2+1] = mean(dy*gamma*xhat) over the sample.   One block per sample.
__global__ void __launch_bounds__(256) gn_bwd_finalize_kernel(const float* __restrict__ part, const float* __restrict__ gamma, float* __restrict__ coeff,
                                                              int C, int chunks, float inv_n) {
  __shared__ float scr[8];
  const int b = blockIdx.x;
  float c1 = 0.f, c2 = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) {
    float s0 = 0.f, s1 = 0.f;
    for (int k = 0; k < chunks; ++k) {
      s0 += part[((size_t)b * chunks + k) * 2 * C + c];
      s1 += part[((size_t)b * chunks + k) * 2 * C + C + c];
    }
    c1 += gamma[c] * s0;
    c2 += gamma[c] * s1;
  }
  c1 = block_sum(c1, scr);
  c2 = block_sum(c2, scr + 4);
  if (threadIdx.x == 0) {
    coeff[b * 2] = c1 * inv_n;
    coeff[b * 2 + 1] = c2 * inv_n;
  }
}
// dx = rstd * (dy*gamma - c1 - xhat*c2)
template <typename T>
__global__ void __launch_bounds__(256) gn_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ dy, const float* __restrict__ stats,
                                                           const float* __restrict__ coeff, const float* __restrict__ gamma, T* __restrict__ dx,
                                                           int HW, int C, int chunks) {
  const int cg = C / 8, RL = 256 / cg;
  const int g = threadIdx.x % cg, rl = threadIdx.x / cg;
  if (rl >= RL) return;
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int r0 = (int)((size_t)HW * chunk / chunks), r1 = (int)((size_t)HW * (chunk + 1) / chunks);
  const float mean = stats[b * 2], rstd = stats[b * 2 + 1], c1 = coeff[b * 2], c2 = coeff[b * 2 + 1];
  float gm[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) gm[j] = gamma[g * 8 + j];
  const size_t base = (size_t)b * HW * C + (size_t)g * 8;
  for (int r = r0 + rl; r < r1; r += RL) {
    float f[8], d[8];
    v8_unpack(v8_load<T>(x + base + (size_t)r * C), f);
    v8_unpack(v8_load<T>(dy + base + (size_t)r * C), d);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = rstd * (d[j] * gm[j] - c1 - (f[j] - mean) * rstd * c2);
    V8<T> o;
    v8_pack(f, o);
    v8_store<T>(dx + base + (size_t)r * C, o);
  }
}

// =============================================================================================
// Linear self-attention
// =============================================================================================
struct LaGeom {
  int B, H, W, ph, pw, C, LD, N, nW;
};
// pixel row of patch n in group (b, p): the reference's unfolded index [b, :, p = py*pw + px, n = ny*nW + nx]
__device__ __forceinline__ size_t la_row(const LaGeom& g, int b, int p, int n) {
  const int py = p / g.pw, px = p - py * g.pw;
  const int ny = n / g.nW, nx = n - ny * g.nW;
  return ((size_t)b * g.H + (size_t)ny * g.ph + py) * g.W + (size_t)nx * g.pw + px;
}
template <int LPR> __device__ __forceinline__ float group_sum(float v) {  // sum over the LPR consecutive lanes that share a row
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// softmax over the N query scalars of the group -> s[n] in LDS (normalised)
template <typename T>
__device__ __forceinline__ void la_scores(const LaGeom& g, const T* __restrict__ kvq, int b, int p, float* s, float* scr) {
  float mx = -INFINITY;
  for (int n = threadIdx.x; n < g.N; n += 256) {
    const float q = to_f<T>(kvq[la_row(g, b, p, n) * g.LD + 2 * g.C]);
    s[n] = q;
    mx = fmaxf(mx, q);
  }
  mx = block_max(mx, scr);
  float sum = 0.f;
  for (int n = threadIdx.x; n < g.N; n += 256) {
    const float e = fast_exp(s[n] - mx);
    s[n] = e;
    sum += e;
  }
  sum = block_sum(sum, scr);
  const float inv = 1.0f / sum;
  for (int n = threadIdx.x; n < g.N; n += 256) s[n] *= inv;
  __syncthreads();
}

// LDS: s[N] | red[RL][LPR*8] (= 2048 floats) | cv[C] | scr[8]
template <typename T, int LPR>
__global__ void __launch_bounds__(256) linattn_fwd_kernel(LaGeom g, const T* __restrict__ kvq, T* __restrict__ out, float* __restrict__ cv_out) {
  extern __shared__ float sm[];
  float* s = sm;
  float* red = s + g.N;
  float* cv = red + 2048;
  float* scr = cv + g.C;
  constexpr int RL = 256 / LPR, CP = LPR * 8;
  const int P = g.ph * g.pw;
  const int b = blockIdx.x / P, p = blockIdx.x % P;
  const int l = threadIdx.x % LPR, rl = threadIdx.x / LPR;
  const bool active = l * 8 < g.C;
  la_scores<T>(g, kvq, b, p, s, scr);
  // context vector cv[c] = sum_n s[n] * key[n][c]
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (active) {
    for (int n = rl; n < g.N; n += RL) {
      float k[8];
      v8_unpack(v8_load<T>(kvq + la_row(g, b, p, n) * g.LD + l * 8), k);
      const float w = s[n];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += w * k[j];
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[rl * CP + l * 8 + j] = acc[j];
  __syncthreads();
  for (int c = threadIdx.x; c < g.C; c += 256) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < RL; ++r) t += red[r * CP + c];
    cv[c] = t;
    cv_out[(size_t)blockIdx.x * g.C + c] = t;
  }
  __syncthreads();
  // out = relu(value) * cv
  if (active) {
    float c8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) c8[j] = cv[l * 8 + j];
    for (int n = rl; n < g.N; n += RL) {
      const size_t row = la_row(g, b, p, n);
      float v[8];
      v8_unpack(v8_load<T>(kvq + row * g.LD + g.C + l * 8), v);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f) * c8[j];
      V8<T> o;
      v8_pack(v, o);
      v8_store<T>(out + row * g.C + l * 8, o);
    }
  }
}

// LDS: s[N] | ds[N] | red[2048] | cv[C] | dcv[C] | scr[8]
template <typename T, int LPR>
__global__ void __launch_bounds__(256) linattn_bwd_kernel(LaGeom g, const T* __restrict__ kvq, const float* __restrict__ cv_in, const T* __restrict__ dout,
                                                          T* __restrict__ dkvq) {
  extern __shared__ float sm[];
  float* s = sm;
  float* ds = s + g.N;
  float* red = ds + g.N;
  float* cv = red + 2048;
  float* dcv = cv + g.C;
  float* scr = dcv + g.C;
  constexpr int RL = 256 / LPR, CP = LPR * 8;
  const int P = g.ph * g.pw;
  const int b = blockIdx.x / P, p = blockIdx.x % P;
  const int l = threadIdx.x % LPR, rl = threadIdx.x / LPR;
  const bool active = l * 8 < g.C;
  for (int c = threadIdx.x; c < g.C; c += 256) cv[c] = cv_in[(size_t)blockIdx.x * g.C + c];
  la_scores<T>(g, kvq, b, p, s, scr);  // (ends with a barrier: cv visible too)
  // dvalue = dout * cv * [value > 0] ; dcv[c] = sum_n dout[n][c] * relu(value[n][c])
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (active) {
    float c8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) c8[j] = cv[l * 8 + j];
    for (int n = rl; n < g.N; n += RL) {
      const size_t row = la_row(g, b, p, n);
      float v[8], d[8], o[8];
      v8_unpack(v8_load<T>(kvq + row * g.LD + g.C + l * 8), v);
      v8_unpack(v8_load<T>(dout + row * g.C + l * 8), d);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        acc[j] += d[j] * fmaxf(v[j], 0.f);
        o[j] = v[j] > 0.f ? d[j] * c8[j] : 0.f;
      }
      V8<T> ov;
      v8_pack(o, ov);
      v8_store<T>(dkvq + row * g.LD + g.C + l * 8, ov);
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[rl * CP + l * 8 + j] = acc[j];
  __syncthreads();
  for (int c = threadIdx.x; c < g.C; c += 256) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < RL; ++r) t += red[r * CP + c];
    dcv[c] = t;
  }
  __syncthreads();
  // dkey[n][c] = dcv[c] * s[n] ; ds[n] = sum_c dcv[c] * key[n][c]
  {
    float d8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) d8[j] = active ? dcv[l * 8 + j] : 0.f;
    for (int n0 = 0; n0 < g.N; n0 += RL) {  // uniform trip count: the row-group shuffles need every lane of the wave
      const int n = n0 + rl;
      float part = 0.f;
      if (active && n < g.N) {
        const size_t row = la_row(g, b, p, n);
        float k[8], o[8];
        v8_unpack(v8_load<T>(kvq + row * g.LD + l * 8), k);
        const float w = s[n];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          part += d8[j] * k[j];
          o[j] = d8[j] * w;
        }
        V8<T> ov;
        v8_pack(o, ov);
        v8_store<T>(dkvq + row * g.LD + l * 8, ov);
      }
      part = group_sum<LPR>(part);
      if (l == 0 && n < g.N) ds[n] = part;
    }
  }
  __syncthreads();
  // softmax backward: dq[n] = s[n] * (ds[n] - sum_m s[m] ds[m])
  float dot = 0.f;
  for (int n = threadIdx.x; n < g.N; n += 256) dot += s[n] * ds[n];
  dot = block_sum(dot, scr);
  for (int n = threadIdx.x; n < g.N; n += 256) {
    float o[8] = {s[n] * (ds[n] - dot), 0, 0, 0, 0, 0, 0, 0};
    V8<T> ov;
    v8_pack(o, ov);
    v8_store<T>(dkvq + la_row(g, b, p, n) * g.LD + 2 * g.C, ov);
  }
}

// =============================================================================================
// C ABI
// =============================================================================================
#define LA_DISPATCH_T(dtype, ...)                                  \
  if ((dtype) == CVH_DT_BF16) { using T = bf16_t; __VA_ARGS__ }    \
  else if ((dtype) == CVH_DT_F32) { using T = float; __VA_ARGS__ } \
  else return -1;

extern "C" int cvh_gn_chunks(int B, int HW, int C) {
  // >= ~2048 workgroups overall, at least 8 rows per chunk
  int chunks = (2048 + B - 1) / B;
  if (chunks > HW / 8) chunks = HW / 8;
  if (chunks < 1) chunks = 1;
  if (chunks > 256) chunks = 256;
  return chunks;
}
static inline int gn_check(int HW, int C) { return (C % 8 || C < 8 || C > 2048 || HW < 1) ? -2 : 0; }

extern "C" int cvh_gn_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* y, float* stats, float* part, int B, int HW, int C,
                          float eps, void* stream) {
  if (gn_check(HW, C)) return -2;
  const int chunks = cvh_gn_chunks(B, HW, C);
  hipStream_t st = (hipStream_t)stream;
  LA_DISPATCH_T(dtype, hipLaunchKernelGGL((gn_stats_kernel<T>), dim3(chunks, B), dim3(256), 0, st, (const T*)x, part, (size_t)HW * C, chunks);)
  hipLaunchKernelGGL(gn_finalize_kernel, dim3((B + 255) / 256), dim3(256), 0, st, part, stats, B, chunks, (double)HW * (double)C, eps);
  LA_DISPATCH_T(dtype, hipLaunchKernelGGL((gn_apply_kernel<T>), dim3(chunks, B), dim3(256), 0, st, (const T*)x, stats, gamma, beta, (T*)y, HW, C, chunks);)
  CVH_CHECK_LAUNCH();
  return 0;
}
extern "C" int cvh_gn_bwd(int dtype, const void* x, const void* dy, const float* stats, const float* gamma, void* dx, float* part, float* coeff, int B,
                          int HW, int C, void* stream) {
  if (gn_check(HW, C)) return -2;
  const int chunks = cvh_gn_chunks(B, HW, C);
  hipStream_t st = (hipStream_t)stream;
  const int RL = 256 / (C / 8);
  const size_t lds = (size_t)RL * 2 * C * sizeof(float);
  LA_DISPATCH_T(dtype, hipLaunchKernelGGL((gn_bwd_reduce_kernel<T>), dim3(chunks, B), dim3(256), lds, st, (const T*)x, (const T*)dy, stats, part, HW, C, chunks);)
  hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(B), dim3(256), 0, st, part, gamma, coeff, C, chunks, (float)(1.0 / ((double)HW * (double)C)));
  LA_DISPATCH_T(dtype, hipLaunchKernelGGL((gn_bwd_apply_kernel<T>), dim3(chunks, B), dim3(256), 0, st, (const T*)x, (const T*)dy, stats, coeff, gamma, (T*)dx, HW, C, chunks);)
  CVH_CHECK_LAUNCH();
  return 0;
}

static inline int la_geom(LaGeom& g, int B, int H, int W, int ph, int pw, int C) {
  if (C % 8 || C < 8 || C > 512 || ph < 1 || pw < 1 || H % ph || W % pw) return -2;
  g.B = B; g.H = H; g.W = W; g.ph = ph; g.pw = pw; g.C = C; g.LD = 2 * C + 8;
  g.nW = W / pw;
  g.N = (H / ph) * g.nW;
  if (g.N > 4096) return -3;  // s[N] + ds[N] must fit the 64 KB LDS window
  return 0;
}
#define LA_LAUNCH(KERNEL, lds, ...)                                                                                     \
  {                                                                                                                     \
    const int cg = C / 8;                                                                                               \
    const dim3 grid(B * ph * pw), blk(256);                                                                             \
    if (cg <= 8) hipLaunchKernelGGL((KERNEL<T, 8>), grid, blk, lds, st, __VA_ARGS__);                                   \
    else if (cg <= 16) hipLaunchKernelGGL((KERNEL<T, 16>), grid, blk, lds, st, __VA_ARGS__);                            \
    else if (cg <= 32) hipLaunchKernelGGL((KERNEL<T, 32>), grid, blk, lds, st, __VA_ARGS__);                            \
    else hipLaunchKernelGGL((KERNEL<T, 64>), grid, blk, lds, st, __VA_ARGS__);                                          \
  }

extern "C" int cvh_linattn_fwd(int dtype, const void* kvq, void* out, float* cv, int B, int H, int W, int ph, int pw, int C, void* stream) {
  LaGeom g;
  const int rc = la_geom(g, B, H, W, ph, pw, C);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = (size_t)(g.N + 2048 + C + 8) * sizeof(float);
  LA_DISPATCH_T(dtype, LA_LAUNCH(linattn_fwd_kernel, lds, g, (const T*)kvq, (T*)out, cv))
  CVH_CHECK_LAUNCH();
  return 0;
}
extern "C" int cvh_linattn_bwd(int dtype, const void* kvq, const float* cv, const void* dout, void* dkvq, int B, int H, int W, int ph, int pw, int C,
                               void* stream) {
  LaGeom g;
  const int rc = la_geom(g, B, H, W, ph, pw, C);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = (size_t)(2 * g.N + 2048 + 2 * C + 8) * sizeof(float);
  LA_DISPATCH_T(dtype, LA_LAUNCH(linattn_bwd_kernel, lds, g, (const T*)kvq, cv, (const T*)dout, (T*)dkvq))
  CVH_CHECK_LAUNCH();
  return 0;
}
