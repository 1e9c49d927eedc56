// LayerNorm over the channel (last) dimension of a [rows][C] token matrix: one 64-lane wave per row,
// row held in registers (C <= 1024), two-pass mean / variance in fp32.
// Replaces nn.LayerNorm (cvnets/layers/normalization/layer_norm.py:14-72, channel-last branch) + backward.
#include "common.hpp"
#include "cvnets_hip.h"

#define LN_MAXIT 4  // 64 lanes * 4 elements * 4 iterations = 1024 channels

// CF: the reference's channel-first formula (x - u) / (std + eps) (layer_norm.py:53-66) instead of (x - u) / sqrt(var + eps)
template <typename T, bool CF = false>
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
    const float rstd = CF ? 1.0f / (sqrtf(var) + eps) : 1.0f / sqrtf(var + eps);
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
// cf_eps > 0: the row statistic is r = 1 / (std + eps) (channel-first formula): d r / d var = -r^2 / (2 std) instead of -r^3 / 2, i.e. the
// xhat term is divided by std * r = 1 - eps * r
template <typename T>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, const float* __restrict__ gamma,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd, T* __restrict__ dx,
                                                     float* __restrict__ part /*[grid][2][C]*/, size_t rows, int C, const T* __restrict__ dres,
                                                     float cf_eps = 0.f) {
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
    const float m1 = wave_sum(s1) / (float)C;
    float m2 = wave_sum(s2) / (float)C;
    if (cf_eps > 0.f) {
      const float k = 1.0f - cf_eps * rs;  // std / (std + eps); 0 for a constant row (the reference's gradient is not finite there)
      m2 = k > 0.f ? m2 / k : 0.f;
    }
#pragma unroll
    for (int it = 0; it < LN_MAXIT; ++it) {
      const int ch = lane + it * 64;
      if (ch < nch) {
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = rs * (g[it][e] - m1 - xh[it][e] * m2);
        if (dres != nullptr) {
          float rv[4];
          v4_unpack(v4_load<T>(dres + row * C + ch * 4), rv);
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] += rv[e];
        }
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


// ---------------------------------------------------------------------------------------------------------------------------------
// Grouped variant for C <= 512 (every LayerNorm of MobileViT / MobileViTv2-block tokens, C = 96 ... 480): a row is owned by a group of
// G = 16 / 32 / 64 lanes, each lane holding 8 consecutive channels (one 16-byte load), so a wave works on 64 / G rows at once and keeps
// U = 4 such row sets in flight before the first reduction.  The one-row-per-wave kernels above move 8 B per lane with ~half the lanes
// idle at C = 144 and have ONE 288-byte row per wave in flight: ~9 KB per CU, i.e. latency-bound at 2.1 - 2.7 TB/s.
// ---------------------------------------------------------------------------------------------------------------------------------
template <int G> __device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <typename T, int G>
__global__ __launch_bounds__(256) void ln_fwd_g_kernel(const T* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       T* __restrict__ y, float* __restrict__ mean_out, float* __restrict__ rstd_out, size_t rows, int C,
                                                       float eps) {
  constexpr int RPW = 64 / G, U = 4;          // rows per wave per step, steps in flight
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int gl = lane % G, gr = lane / G;
  const int c0 = gl * 8;
  const bool cok = c0 < C;
  float gm[8], bt[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { gm[j] = cok ? gamma[c0 + j] : 0.f; bt[j] = cok ? beta[c0 + j] : 0.f; }
  const float invC = 1.0f / (float)C;
  const size_t stride = (size_t)gridDim.x * 4 * RPW * U;
  for (size_t base = ((size_t)blockIdx.x * 4 + wave) * RPW * U; base < rows; base += stride) {
    V8<T> raw[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t row = base + u * RPW + gr;
      raw[u] = v8_load_clamped<T>(x, row * C + c0, cok && row < rows);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t row = base + u * RPW + gr;
      const bool ok = cok && row < rows;
      float v[8];
      v8_unpack(v8_mask(raw[u], ok), v);
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[j];
      const float mu = group_sum<G>(s) * invC;
      float q = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = v[j] - mu; q += d * d; }
      if (!cok) q = 0.f;  // idle lanes hold zeros, not (0 - mu)
      const float rstd = 1.0f / sqrtf(group_sum<G>(q) * invC + eps);
      if (ok) {
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (v[j] - mu) * rstd * gm[j] + bt[j];
        V8<T> ov;
        v8_pack(o, ov);
        v8_store<T>(y + row * C + c0, ov);
        if (gl == 0 && mean_out) { mean_out[row] = mu; rstd_out[row] = rstd; }
      }
    }
  }
}

// DROP: the kernel also stores dxd = dx * keep-scale — dx as the Dropout in FRONT of this LayerNorm (x = res + Dropout(linear(h)),
// cvnets/modules/transformer.py:140-155) hands it to that linear's dW / dX GEMMs: the mask is regenerated from (seed, stream id, element
// index) exactly as cvh_dropout does on the stored dx, so the standalone dropout-backward pass (one read + one write of the token
// matrix per dropout site) becomes one extra write here
struct LnDropArgs {
  void* dxd;
  const unsigned long long* seed;
  float p;
  unsigned int stream_id;
};
template <typename T, int G, bool DROP = false>
__global__ __launch_bounds__(256) void ln_bwd_g_kernel(const T* __restrict__ x, const T* __restrict__ dy, const float* __restrict__ gamma,
                                                       const float* __restrict__ mean, const float* __restrict__ rstd, T* __restrict__ dx,
                                                       float* __restrict__ part /*[grid][2][C]*/, size_t rows, int C,
                                                       const T* __restrict__ dres /* optional: dx += dres (residual fork) */,
                                                       LnDropArgs da = LnDropArgs{nullptr, nullptr, 0.f, 0u}) {
  DropKey dkey = {0u, 0u};
  float inv_keep = 1.f;
  if (DROP) {
    dkey = drop_key(*da.seed, da.stream_id, da.p);
    inv_keep = 1.0f / (1.0f - da.p);
  }
  constexpr int RPW = 64 / G, U = 2;
  extern __shared__ __attribute__((aligned(16))) float red[];  // [2][C], zeroed; every (wave, row group) adds its sums in a fixed order
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int gl = lane % G, gr = lane / G;
  const int c0 = gl * 8;
  const bool cok = c0 < C;
  for (int i = threadIdx.x; i < 2 * C; i += 256) red[i] = 0.f;
  float gm[8], dg[8], db[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { gm[j] = cok ? gamma[c0 + j] : 0.f; dg[j] = 0.f; db[j] = 0.f; }
  const float invC = 1.0f / (float)C;
  const size_t stride = (size_t)gridDim.x * 4 * RPW * U;
  for (size_t base = ((size_t)blockIdx.x * 4 + wave) * RPW * U; base < rows; base += stride) {
    V8<T> rx[U], rd[U], rr[U];
    float mu[U], rs[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t row = base + u * RPW + gr;
      const bool ok = cok && row < rows;
      rx[u] = v8_load_clamped<T>(x, row * C + c0, ok);
      rd[u] = v8_load_clamped<T>(dy, row * C + c0, ok);
      rr[u] = v8_load_clamped<T>(dres != nullptr ? dres : dy, row * C + c0, ok);
      const size_t rr = row < rows ? row : 0;
      mu[u] = mean[rr];
      rs[u] = rstd[rr];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t row = base + u * RPW + gr;
      const bool ok = cok && row < rows;
      float xv[8], dv[8], xh[8], g[8];
      v8_unpack(v8_mask(rx[u], ok), xv);
      v8_unpack(v8_mask(rd[u], ok), dv);
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        xh[j] = ok ? (xv[j] - mu[u]) * rs[u] : 0.f;
        g[j] = dv[j] * gm[j];
        s1 += g[j];
        s2 += g[j] * xh[j];
        dg[j] += dv[j] * xh[j];
        db[j] += dv[j];
      }
      const float m1 = group_sum<G>(s1) * invC, m2 = group_sum<G>(s2) * invC;
      if (ok) {
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = rs[u] * (g[j] - m1 - xh[j] * m2);
        if (dres != nullptr) {
          float rv[8];
          v8_unpack(rr[u], rv);
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] += rv[j];
        }
        V8<T> ov;
        v8_pack(o, ov);
        v8_store<T>(dx + row * C + c0, ov);
        if (DROP) {
          v8_unpack(ov, o);  // the mask applies to dx as stored
          dropout_scale8(dkey, (uint64_t)(row * C + c0), inv_keep, o);
          v8_pack(o, ov);
          v8_store<T>(reinterpret_cast<T*>(da.dxd) + row * C + c0, ov);
        }
      }
    }
  }
  __syncthreads();
  lds_ordered_accumulate(wave * RPW + gr, 4 * RPW, cok, [&]() {  // the (wave, row slot) owners of one 8-channel group, in order
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      red[c0 + j] += dg[j];
      red[C + c0 + j] += db[j];
    }
  });
  for (int i = threadIdx.x; i < 2 * C; i += 256) part[(size_t)blockIdx.x * 2 * C + i] = red[i];
}

static inline int ln_group(int C) { return C <= 128 ? 16 : (C <= 256 ? 32 : 64); }
static inline bool ln_grouped_ok(int C) { return C % 8 == 0 && C <= 512 && cvh_tune_get(CVH_TUNE_LN_PER_ROW) == 0; }  // CVH_TUNE key 8 = 1: the one-row-per-wave kernels

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
  if (ln_grouped_ok(C) && rows > 0) {
    const int G = ln_group(C);
    long long gg = (rows + 4 * (64 / G) * 4 - 1) / (4 * (64 / G) * 4);  // 4 waves x (64 / G) rows x 4 steps per workgroup pass
    if (gg > 2048) gg = 2048;
#define LN_FWD_G(TT, GG) hipLaunchKernelGGL((ln_fwd_g_kernel<TT, GG>), dim3((int)gg), dim3(256), 0, st, (const TT*)x, gamma, beta, (TT*)y, mean, rstd, (size_t)rows, C, eps)
    if (dtype == CVH_DT_BF16) { if (G == 16) LN_FWD_G(bf16_t, 16); else if (G == 32) LN_FWD_G(bf16_t, 32); else LN_FWD_G(bf16_t, 64); }
    else if (dtype == CVH_DT_F32) { if (G == 16) LN_FWD_G(float, 16); else if (G == 32) LN_FWD_G(float, 32); else LN_FWD_G(float, 64); }
    else return -1;
#undef LN_FWD_G
    CVH_CHECK_LAUNCH();
    return 0;
  }
  if (dtype == CVH_DT_BF16) hipLaunchKernelGGL((ln_fwd_kernel<bf16_t>), dim3((int)g), dim3(256), 0, st, (const bf16_t*)x, gamma, beta, (bf16_t*)y, mean, rstd, (size_t)rows, C, eps);
  else if (dtype == CVH_DT_F32) hipLaunchKernelGGL((ln_fwd_kernel<float>), dim3((int)g), dim3(256), 0, st, (const float*)x, gamma, beta, (float*)y, mean, rstd, (size_t)rows, C, eps);
  else return -1;
  CVH_CHECK_LAUNCH();
  return 0;
}
extern "C" int cvh_layernorm_bwd_res(int dtype, const void* x, const void* dy, const float* gamma, const float* mean, const float* rstd, void* dx,
                                     float* part, long long rows, int C, const void* dres, void* stream);
extern "C" int cvh_layernorm_bwd(int dtype, const void* x, const void* dy, const float* gamma, const float* mean, const float* rstd, void* dx,
                                 float* part, long long rows, int C, void* stream) {
  return cvh_layernorm_bwd_res(dtype, x, dy, gamma, mean, rstd, dx, part, rows, C, nullptr, stream);
}
extern "C" int cvh_layernorm_bwd_res(int dtype, const void* x, const void* dy, const float* gamma, const float* mean, const float* rstd, void* dx,
                                     float* part, long long rows, int C, const void* dres, void* stream) {
  if (C % 4 || C > 64 * 4 * LN_MAXIT || C <= 0) return -2;
  int g = cvh_ln_bwd_rows(rows);
  size_t smem = (size_t)4 * 2 * C * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  if (ln_grouped_ok(C) && rows > 0) {  // same partial-row contract: part[g][2][C]
    const int G = ln_group(C);
    const size_t sm = (size_t)2 * C * sizeof(float);
#define LN_BWD_G(TT, GG) hipLaunchKernelGGL((ln_bwd_g_kernel<TT, GG, false>), dim3(g), dim3(256), sm, st, (const TT*)x, (const TT*)dy, gamma, mean, rstd, (TT*)dx, part, (size_t)rows, C, (const TT*)dres, LnDropArgs{nullptr, nullptr, 0.f, 0u})
    if (dtype == CVH_DT_BF16) { if (G == 16) LN_BWD_G(bf16_t, 16); else if (G == 32) LN_BWD_G(bf16_t, 32); else LN_BWD_G(bf16_t, 64); }
    else if (dtype == CVH_DT_F32) { if (G == 16) LN_BWD_G(float, 16); else if (G == 32) LN_BWD_G(float, 32); else LN_BWD_G(float, 64); }
    else return -1;
#undef LN_BWD_G
    CVH_CHECK_LAUNCH();
    return 0;
  }
  if (dtype == CVH_DT_BF16) hipLaunchKernelGGL((ln_bwd_kernel<bf16_t>), dim3(g), dim3(256), smem, st, (const bf16_t*)x, (const bf16_t*)dy, gamma, mean, rstd, (bf16_t*)dx, part, (size_t)rows, C, (const bf16_t*)dres);
  else if (dtype == CVH_DT_F32) hipLaunchKernelGGL((ln_bwd_kernel<float>), dim3(g), dim3(256), smem, st, (const float*)x, (const float*)dy, gamma, mean, rstd, (float*)dx, part, (size_t)rows, C, (const float*)dres);
  else return -1;
  CVH_CHECK_LAUNCH();
  return 0;
}

/* cvh_layernorm_bwd_res that also stores dxd = Dropout-backward of dx (mask of (seed, stream_id, p) over the [rows][C] element index, as
 * cvh_dropout draws it).  cvh_ln_bwd_drop_ok(C) == 0: this width runs on the one-row-per-wave kernels, use cvh_dropout on dx. */
extern "C" int cvh_ln_bwd_drop_ok(int C) { return (ln_grouped_ok(C) && C % 8 == 0) ? 1 : 0; }
extern "C" int cvh_layernorm_bwd_res_drop(int dtype, const void* x, const void* dy, const float* gamma, const float* mean, const float* rstd,
                                          void* dx, float* part, long long rows, int C, const void* dres, void* dxd, float drop_p,
                                          const unsigned long long* seed, unsigned int stream_id, void* stream) {
  if (!cvh_ln_bwd_drop_ok(C) || rows <= 0 || dxd == nullptr || seed == nullptr || drop_p <= 0.f || drop_p >= 1.f) return -2;
  const int g = cvh_ln_bwd_rows(rows);
  hipStream_t st = (hipStream_t)stream;
  const int G = ln_group(C);
  const size_t sm = (size_t)2 * C * sizeof(float);
  const LnDropArgs da{dxd, seed, drop_p, stream_id};
#define LN_BWD_GD(TT, GG) hipLaunchKernelGGL((ln_bwd_g_kernel<TT, GG, true>), dim3(g), dim3(256), sm, st, (const TT*)x, (const TT*)dy, gamma, mean, rstd, (TT*)dx, part, (size_t)rows, C, (const TT*)dres, da)
  if (dtype == CVH_DT_BF16) { if (G == 16) LN_BWD_GD(bf16_t, 16); else if (G == 32) LN_BWD_GD(bf16_t, 32); else LN_BWD_GD(bf16_t, 64); }
  else if (dtype == CVH_DT_F32) { if (G == 16) LN_BWD_GD(float, 16); else if (G == 32) LN_BWD_GD(float, 32); else LN_BWD_GD(float, 64); }
  else return -1;
#undef LN_BWD_GD
  CVH_CHECK_LAUNCH();
  return 0;
}

// =============================================================================================
// The reference's channel-first LayerNorm branch applied to a [B', S, C] token tensor with S == C
// (cvnets/layers/normalization/layer_norm.py:53-66 fires whenever x.shape[1] == C and x.ndim > 2 — e.g. MobileViT-S at 192x192
// (144 patches x 144 channels) or MobileViT-XXS at 128x128 (64 x 64), SURVEY.md headline fact 5):
//     u, s = mean / biased std over dim 1 (the PATCH axis) ;  y[b,n,c] = beta[n] + gamma[n] * (x[b,n,c] - u[b,c]) / (s[b,c] + eps)
// i.e. statistics per (sequence, channel) over the tokens, affine indexed by the token position, and (std + eps) rather than
// sqrt(var + eps).  Bug-compatible by default so that results stay identical to the reference at every resolution.
// One workgroup per sequence; thread = (8-channel group, row lane); rows are gathered through the SeqMap.
// =============================================================================================
struct LnSeqParams {
  int nseq, S, C;
  SeqMap map;
  float eps;
};

template <typename T>
__global__ __launch_bounds__(256) void ln_seq_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         T* __restrict__ y, float* __restrict__ stats /* [nseq][2][C]: mean, std */, LnSeqParams p) {
  extern __shared__ float sm[];  // red[RL][2][C] | mu[C] | inv_t[C]
  const int cg = p.C / 8, RL = 256 / cg;
  const int g = threadIdx.x % cg, rl = threadIdx.x / cg;
  const int s = blockIdx.x;
  float* red = sm;
  float* mu = sm + RL * 2 * p.C;
  float* it = mu + p.C;
  float a[8] = {0, 0, 0, 0, 0, 0, 0, 0}, q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (rl < RL) {
    for (int n = rl; n < p.S; n += RL) {
      float f[8];
      v8_unpack(v8_load<T>(x + (size_t)seq_row(p.map, s, n) * p.C + g * 8), f);
#pragma unroll
      for (int j = 0; j < 8; ++j) { a[j] += f[j]; q[j] += f[j] * f[j]; }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[(rl * 2) * p.C + g * 8 + j] = a[j]; red[(rl * 2 + 1) * p.C + g * 8 + j] = q[j]; }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < p.C; c += 256) {
    double s1 = 0.0, s2 = 0.0;
    for (int l = 0; l < RL; ++l) { s1 += red[(l * 2) * p.C + c]; s2 += red[(l * 2 + 1) * p.C + c]; }
    const double m = s1 / p.S;
    double var = s2 / p.S - m * m;
    if (var < 0.0) var = 0.0;
    const float sd = (float)sqrt(var);
    mu[c] = (float)m;
    it[c] = 1.0f / (sd + p.eps);
    stats[((size_t)s * 2) * p.C + c] = (float)m;
    stats[((size_t)s * 2 + 1) * p.C + c] = sd;
  }
  __syncthreads();
  if (rl < RL) {
    float m8[8], i8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { m8[j] = mu[g * 8 + j]; i8[j] = it[g * 8 + j]; }
    for (int n = rl; n < p.S; n += RL) {
      const size_t row = (size_t)seq_row(p.map, s, n);
      const float ga = gamma[n], be = beta[n];
      float f[8];
      v8_unpack(v8_load<T>(x + row * p.C + g * 8), f);
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = be + ga * ((f[j] - m8[j]) * i8[j]);
      V8<T> o;
      v8_pack(f, o);
      v8_store<T>(y + row * p.C + g * 8, o);
    }
  }
}

// z = (x-u)/t, t = s + eps, g = dy * gamma[n]:  dx = (g - mean_n g)/t - (z/s) * mean_n(g z)
// part[s][2][S]: per-sequence contributions to dgamma[n] = sum_c dy z and dbeta[n] = sum_c dy (summed over sequences by cvh_sum_partials)
template <typename T>
__global__ __launch_bounds__(256) void ln_seq_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, const float* __restrict__ gamma,
                                                         const float* __restrict__ stats, T* __restrict__ dx, float* __restrict__ part, LnSeqParams p) {
  extern __shared__ float sm[];  // red[RL][2][C] | mg[C] | mgz[C] | rowacc[2][S]
  const int cg = p.C / 8, RL = 256 / cg;
  const int g = threadIdx.x % cg, rl = threadIdx.x / cg;
  const int s = blockIdx.x;
  float* red = sm;
  float* mg = sm + RL * 2 * p.C;
  float* mgz = mg + p.C;
  float* rowacc = mgz + p.C;
  for (int i = threadIdx.x; i < 2 * p.S; i += 256) rowacc[i] = 0.f;
  float m8[8], s8[8], i8[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    m8[j] = stats[((size_t)s * 2) * p.C + g * 8 + j];
    s8[j] = stats[((size_t)s * 2 + 1) * p.C + g * 8 + j];
    i8[j] = 1.0f / (s8[j] + p.eps);
  }
  __syncthreads();
  float a[8] = {0, 0, 0, 0, 0, 0, 0, 0}, q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (rl < RL) {
    for (int n = rl; n < p.S; n += RL) {
      const size_t row = (size_t)seq_row(p.map, s, n);
      const float ga = gamma[n];
      float f[8], d[8];
      v8_unpack(v8_load<T>(x + row * p.C + g * 8), f);
      v8_unpack(v8_load<T>(dy + row * p.C + g * 8), d);
      float dg = 0.f, db = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float z = (f[j] - m8[j]) * i8[j];
        a[j] += d[j] * ga;
        q[j] += d[j] * ga * z;
        dg += d[j] * z;
        db += d[j];
      }
      atomicAdd(&rowacc[n], dg);
      atomicAdd(&rowacc[p.S + n], db);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[(rl * 2) * p.C + g * 8 + j] = a[j]; red[(rl * 2 + 1) * p.C + g * 8 + j] = q[j]; }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < p.C; c += 256) {
    float s1 = 0.f, s2 = 0.f;
    for (int l = 0; l < RL; ++l) { s1 += red[(l * 2) * p.C + c]; s2 += red[(l * 2 + 1) * p.C + c]; }
    mg[c] = s1 / (float)p.S;
    mgz[c] = s2 / (float)p.S;
  }
  for (int i = threadIdx.x; i < 2 * p.S; i += 256) part[(size_t)s * 2 * p.S + i] = rowacc[i];
  __syncthreads();
  if (rl < RL) {
    float g8[8], z8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { g8[j] = mg[g * 8 + j]; z8[j] = s8[j] > 0.f ? mgz[g * 8 + j] / s8[j] : 0.f; }
    for (int n = rl; n < p.S; n += RL) {
      const size_t row = (size_t)seq_row(p.map, s, n);
      const float ga = gamma[n];
      float f[8], d[8];
      v8_unpack(v8_load<T>(x + row * p.C + g * 8), f);
      v8_unpack(v8_load<T>(dy + row * p.C + g * 8), d);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float z = (f[j] - m8[j]) * i8[j];
        f[j] = (d[j] * ga - g8[j]) * i8[j] - z * z8[j];
      }
      V8<T> o;
      v8_pack(f, o);
      v8_store<T>(dx + row * p.C + g * 8, o);
    }
  }
}

static int ln_seq_params(LnSeqParams& p, int nseq, int S, int C, int ph, int pw, int n_w, int H, int W, float eps) {
  if (C % 8 || C < 8 || C > 2048 || S != C || nseq <= 0) return -2;  // gamma/beta [C] are indexed by the token position: needs S == C
  p.nseq = nseq; p.S = S; p.C = C; p.eps = eps;
  p.map.ph = ph; p.map.pw = pw; p.map.n_w = n_w; p.map.H = H; p.map.W = W;
  return 0;
}
extern "C" int cvh_ln_seq_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* y, float* stats, int nseq, int S, int C, int ph,
                              int pw, int n_w, int H, int W, float eps, void* stream) {
  LnSeqParams p;
  if (ln_seq_params(p, nseq, S, C, ph, pw, n_w, H, W, eps)) return -2;
  const int RL = 256 / (C / 8);
  const size_t smem = (size_t)(RL * 2 * C + 2 * C) * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CVH_DT_BF16) hipLaunchKernelGGL((ln_seq_fwd_kernel<bf16_t>), dim3(nseq), dim3(256), smem, st, (const bf16_t*)x, gamma, beta, (bf16_t*)y, stats, p);
  else if (dtype == CVH_DT_F32) hipLaunchKernelGGL((ln_seq_fwd_kernel<float>), dim3(nseq), dim3(256), smem, st, (const float*)x, gamma, beta, (float*)y, stats, p);
  else return -1;
  CVH_CHECK_LAUNCH();
  return 0;
}
extern "C" int cvh_ln_seq_bwd(int dtype, const void* x, const void* dy, const float* gamma, const float* stats, void* dx, float* part, int nseq, int S,
                              int C, int ph, int pw, int n_w, int H, int W, float eps, void* stream) {
  LnSeqParams p;
  if (ln_seq_params(p, nseq, S, C, ph, pw, n_w, H, W, eps)) return -2;
  const int RL = 256 / (C / 8);
  const size_t smem = (size_t)(RL * 2 * C + 2 * C + 2 * S) * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CVH_DT_BF16) hipLaunchKernelGGL((ln_seq_bwd_kernel<bf16_t>), dim3(nseq), dim3(256), smem, st, (const bf16_t*)x, (const bf16_t*)dy, gamma, stats, (bf16_t*)dx, part, p);
  else if (dtype == CVH_DT_F32) hipLaunchKernelGGL((ln_seq_bwd_kernel<float>), dim3(nseq), dim3(256), smem, st, (const float*)x, (const float*)dy, gamma, stats, (float*)dx, part, p);
  else return -1;
  CVH_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The reference's channel-first LayerNorm branch on a genuine feature map (cvnets/layers/normalization/layer_norm.py:53-66:
// x.shape[1] == C, x.ndim > 2): statistics over the channels of every pixel, y = beta[c] + gamma[c] (x - u) / (std + eps).  On the NHWC
// storage of the HIP path that is a row-wise normalisation of the [pixels][C] matrix with the (std + eps) denominator: the one-row-per-wave
// kernels with the CF switch.  rstd[] holds 1 / (std + eps).
// ---------------------------------------------------------------------------------------------------------------------------------
extern "C" int cvh_layernorm_cf_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                                    long long rows, int C, float eps, void* stream) {
  if (C % 4 || C > 64 * 4 * LN_MAXIT || C <= 0 || eps <= 0.f) return -2;
  if (rows <= 0) return 0;
  long long g = (rows + 3) / 4;
  if (g > 8192) g = 8192;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CVH_DT_BF16) hipLaunchKernelGGL((ln_fwd_kernel<bf16_t, true>), dim3((int)g), dim3(256), 0, st, (const bf16_t*)x, gamma, beta, (bf16_t*)y, mean, rstd, (size_t)rows, C, eps);
  else if (dtype == CVH_DT_F32) hipLaunchKernelGGL((ln_fwd_kernel<float, true>), dim3((int)g), dim3(256), 0, st, (const float*)x, gamma, beta, (float*)y, mean, rstd, (size_t)rows, C, eps);
  else return -1;
  CVH_CHECK_LAUNCH();
  return 0;
}
/* part: cvh_ln_bwd_rows(rows) partial rows [2][C] of dgamma / dbeta, as cvh_layernorm_bwd leaves them */
extern "C" int cvh_layernorm_cf_bwd(int dtype, const void* x, const void* dy, const float* gamma, const float* mean, const float* rstd, void* dx,
                                    float* part, long long rows, int C, float eps, void* stream) {
  if (C % 4 || C > 64 * 4 * LN_MAXIT || C <= 0 || eps <= 0.f) return -2;
  if (rows <= 0) return 0;
  const int g = cvh_ln_bwd_rows(rows);
  const size_t smem = (size_t)4 * 2 * C * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CVH_DT_BF16) hipLaunchKernelGGL((ln_bwd_kernel<bf16_t>), dim3(g), dim3(256), smem, st, (const bf16_t*)x, (const bf16_t*)dy, gamma, mean, rstd, (bf16_t*)dx, part, (size_t)rows, C, (const bf16_t*)nullptr, eps);
  else if (dtype == CVH_DT_F32) hipLaunchKernelGGL((ln_bwd_kernel<float>), dim3(g), dim3(256), smem, st, (const float*)x, (const float*)dy, gamma, mean, rstd, (float*)dx, part, (size_t)rows, C, (const float*)nullptr, eps);
  else return -1;
  CVH_CHECK_LAUNCH();
  return 0;
}
