// BatchNorm links, linear side (see bnlink.hpp): when the conv in FRONT of a train-mode BatchNorm is linear in its input
// (y = x W^T, the 1x1 expansion conv of InvertedResidual, cvnets/modules/mobilenetv2.py:179-192), the BatchNorm input gradient
//     dy = ca (.) g + cb (.) y + cc            (per output channel n; g = dz * act'(bn(y)))
// never has to be formed, and y never has to be re-read, to back-propagate through the conv:
//     dX = dy W    = g (diag(ca) W) + x (W^T diag(cb) W) + 1 (cc^T W)        -> ONE plain GEMM on the channel-concat [g | x] + a bias row
//     dW = dy^T x  = diag(ca) (g^T x) + diag(cb) W (x^T x) + cc (1^T x)      -> the plain dW GEMM on g, the K x K Gram matrix of the
//                                                                               (narrow) block input and its column sums
// The two kernels below are the O(N K^2) glue: the concatenated dX weight / bias, and the dW combination.
#include "common.hpp"
#include "cvnets_hip.h"

// wcat[k][0..N) = ca[n] * w[n][k];  wcat[k][N + j] = sum_n cb[n] * w[n][j] * w[n][k];  bias[k] = sum_n cc[n] * w[n][k]
// w: [N][K] float32 (torch [N][K][1][1]); rows k >= K and columns j >= K of the padded image (Kp = pad8(K)) are zero.
template <typename T>
__global__ __launch_bounds__(256) void bn_dx_weights_kernel(const float* __restrict__ w, const float* __restrict__ coef, T* __restrict__ wcat,
                                                            float* __restrict__ bias, int N, int K, int Kp) {
  // block = output row k; the n-reductions (Q column and bias) are split over 256 / JW slices of n and combined through LDS
  __shared__ float red[256];
  __shared__ float wk[512], wc[512];   // cb[n] * w[n][k] and cc[n] * w[n][k] for the n chunk being processed
  const int k = blockIdx.x, tid = threadIdx.x;
  const float* ca = coef;
  const float* cb = coef + N;
  const float* cc = coef + 2 * N;
  T* row = wcat + (size_t)k * (N + Kp);
  const bool live = k < K;
  for (int n = tid; n < N; n += 256) row[n] = from_f<T>(live ? ca[n] * w[(size_t)n * K + k] : 0.f);
  const int JW = Kp <= 32 ? 32 : (Kp <= 64 ? 64 : 128);  // threads along j
  const int NS = 256 / JW;                               // n slices
  const int j = tid % JW, sl = tid / JW;
  for (int j0 = 0; j0 < Kp; j0 += JW) {
    float q = 0.f, b = 0.f;
    for (int n0 = 0; n0 < N; n0 += 512) {
      const int nn = min(512, N - n0);
      __syncthreads();
      for (int i = tid; i < nn; i += 256) {
        const float wv = live ? w[(size_t)(n0 + i) * K + k] : 0.f;
        wk[i] = cb[n0 + i] * wv;
        wc[i] = cc[n0 + i] * wv;
      }
      __syncthreads();
      const int jj = j0 + j;
      if (jj < K) {
#pragma unroll 4
        for (int i = sl; i < nn; i += NS) q += wk[i] * w[(size_t)(n0 + i) * K + jj];
      }
      if (j0 == 0 && j == 0) {
        for (int i = sl; i < nn; i += NS) b += wc[i];
      }
    }
    __syncthreads();
    red[tid] = q;
    __syncthreads();
    if (sl == 0) {
      float t = 0.f;
      for (int s2 = 0; s2 < NS; ++s2) t += red[s2 * JW + j];
      if (j0 + j < Kp) row[N + j0 + j] = from_f<T>(t);
    }
    if (j0 == 0) {
      __syncthreads();
      red[tid] = (j == 0) ? b : 0.f;
      __syncthreads();
      if (tid == 0) {
        float t = 0.f;
        for (int s2 = 0; s2 < NS; ++s2) t += red[s2 * JW];
        bias[k] = t;
      }
    }
  }
}

// Same result with W, cb and cc staged in LDS by ONE coalesced pass (N * (K + 1) + 2 N floats must fit): the version above walks W with
// dependent strided global loads, and on the backward critical path — between the depthwise backward kernel and the dX GEMM, while a dW
// GEMM saturates HBM from the side stream — those loads took up to 750 us for a 16 KB weight.
template <typename T>
__global__ __launch_bounds__(256) void bn_dx_weights_lds_kernel(const float* __restrict__ w, const float* __restrict__ coef, T* __restrict__ wcat,
                                                                float* __restrict__ bias, int N, int K, int Kp) {
  extern __shared__ float smem_f[];
  __shared__ float red[256];
  const int P = K + 1;
  float* wl = smem_f;            // [N][K + 1]
  float* cbl = wl + N * P;       // [N]
  float* ccl = cbl + N;          // [N]
  const int k = blockIdx.x, tid = threadIdx.x;
  for (int i = tid; i < N * K; i += 256) {
    const int n = i / K;
    wl[n * P + (i - n * K)] = w[i];
  }
  for (int n = tid; n < N; n += 256) { cbl[n] = coef[N + n]; ccl[n] = coef[2 * N + n]; }
  __syncthreads();
  T* row = wcat + (size_t)k * (N + Kp);
  const bool live = k < K;
  for (int n = tid; n < N; n += 256) row[n] = from_f<T>(live ? coef[n] * wl[n * P + k] : 0.f);
  const int JW = Kp <= 32 ? 32 : (Kp <= 64 ? 64 : 128);
  const int NS = 256 / JW;
  const int j = tid % JW, sl = tid / JW;
  for (int j0 = 0; j0 < Kp; j0 += JW) {
    const int jj = j0 + j;
    float q = 0.f, b = 0.f;
    if (live && jj < K) {
      for (int n = sl; n < N; n += NS) q += cbl[n] * wl[n * P + k] * wl[n * P + jj];
    }
    if (live && j0 == 0 && j == 0) {
      for (int n = sl; n < N; n += NS) b += ccl[n] * wl[n * P + k];
    }
    __syncthreads();
    red[tid] = q;
    __syncthreads();
    if (sl == 0) {
      float t = 0.f;
      for (int s2 = 0; s2 < NS; ++s2) t += red[s2 * JW + j];
      if (jj < Kp) row[N + jj] = from_f<T>(t);
    }
    if (j0 == 0) {
      __syncthreads();
      red[tid] = (j == 0) ? b : 0.f;
      __syncthreads();
      if (tid == 0) {
        float t = 0.f;
        for (int s2 = 0; s2 < NS; ++s2) t += red[s2 * JW];
        bias[k] = t;
      }
    }
  }
}

// dw[n][k] = (accumulate ? dw : 0) + ca[n] * P[n][k] + cb[n] * sum_j w[n][j] * G[j][k] + cc[n] * s[k]
// P = g^T x [N][K], G = x^T x [Kp][Kp], s = column sums of x [Kp]
__global__ __launch_bounds__(256) void bn_dw_combine_kernel(const float* __restrict__ P, const float* __restrict__ w, const float* __restrict__ G,
                                                            const float* __restrict__ s, const float* __restrict__ coef, float* __restrict__ dw,
                                                            int N, int K, int Kp, int accumulate) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= N * K) return;
  const int n = idx / K, k = idx - n * K;
  // four independent partial sums, eight loads of each operand in flight: as one dependent chain of K (<= 128) global loads this tiny
  // kernel took 17 - 24 us on the backward critical path of every fused block
  float wg0 = 0.f, wg1 = 0.f, wg2 = 0.f, wg3 = 0.f;
  const float* wr = w + (size_t)n * K;
  int j = 0;
  for (; j + 8 <= K; j += 8) {
    float a[8], b[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      a[u] = wr[j + u];
      b[u] = G[(size_t)(j + u) * Kp + k];
    }
    wg0 += a[0] * b[0] + a[4] * b[4];
    wg1 += a[1] * b[1] + a[5] * b[5];
    wg2 += a[2] * b[2] + a[6] * b[6];
    wg3 += a[3] * b[3] + a[7] * b[7];
  }
  for (; j < K; ++j) wg0 += wr[j] * G[(size_t)j * Kp + k];
  const float wg = (wg0 + wg1) + (wg2 + wg3);
  const float v = coef[n] * P[idx] + coef[N + n] * wg + coef[2 * N + n] * s[k];
  dw[idx] = accumulate ? dw[idx] + v : v;
}

extern "C" int cvh_bn_dx_weights(int dtype, const float* w, const float* coef, void* wcat, float* bias, int N, int K, void* stream) {
  if (N <= 0 || K <= 0 || (N % 8) != 0) return -2;
  const int Kp = (K + 7) / 8 * 8;
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = ((size_t)N * (K + 1) + 2 * (size_t)N) * sizeof(float);
  if (lds <= 96 * 1024 && (dtype == CVH_DT_BF16 || dtype == CVH_DT_F32)) {
    static DynSmemAttr attr_b, attr_f;
    hipError_t e = attr_b.ensure(reinterpret_cast<const void*>(bn_dx_weights_lds_kernel<bf16_t>), 96 * 1024);
    if (e == hipSuccess) e = attr_f.ensure(reinterpret_cast<const void*>(bn_dx_weights_lds_kernel<float>), 96 * 1024);
    if (e != hipSuccess) return (int)e;
    if (dtype == CVH_DT_BF16) hipLaunchKernelGGL((bn_dx_weights_lds_kernel<bf16_t>), dim3(Kp), dim3(256), lds, st, w, coef, (bf16_t*)wcat, bias, N, K, Kp);
    else hipLaunchKernelGGL((bn_dx_weights_lds_kernel<float>), dim3(Kp), dim3(256), lds, st, w, coef, (float*)wcat, bias, N, K, Kp);
    CVH_CHECK_LAUNCH();
    return 0;
  }
  if (dtype == CVH_DT_BF16) hipLaunchKernelGGL((bn_dx_weights_kernel<bf16_t>), dim3(Kp), dim3(256), 0, st, w, coef, (bf16_t*)wcat, bias, N, K, Kp);
  else if (dtype == CVH_DT_F32) hipLaunchKernelGGL((bn_dx_weights_kernel<float>), dim3(Kp), dim3(256), 0, st, w, coef, (float*)wcat, bias, N, K, Kp);
  else return -1;
  CVH_CHECK_LAUNCH();
  return 0;
}

extern "C" int cvh_bn_dw_combine(const float* P, const float* w, const float* G, const float* s, const float* coef, float* dw, int N, int K,
                                 int accumulate, void* stream) {
  if (N <= 0 || K <= 0) return -2;
  const int Kp = (K + 7) / 8 * 8;
  hipLaunchKernelGGL(bn_dw_combine_kernel, dim3((N * K + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, w, G, s, coef, dw, N, K, Kp, accumulate);
  CVH_CHECK_LAUNCH();
  return 0;
}
