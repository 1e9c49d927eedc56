// Token-matrix plumbing for the ViT / CLIP stacks (all HBM-bound, 16 B per lane):
//   * patch tokens + positional embedding + class token  ->  [B][1+N][E] sequence   (fwd / bwd)
//   * batch reduction of a [B][L] gradient (class-token / positional-embedding gradients)
//   * strided row copy (class-token rows in / out of a [B][S][E] tensor)
//   * embedding lookup + positional embedding for the text tower, and its scatter-add backward
//
// Replaces (reference): VisionTransformer.extract_patch_embeddings cvnets/models/classification/vit.py:480-509,
// the cls split :562-565, TextTransformer.forward_embedding cvnets/text_encoders/transformer.py:321-341.
#include "common.hpp"
#include "cvnets_hip.h"

// out[b][0][:] = cls[:] ; out[b][1+n][:] = patch[b][n][:] + pos[n][:]      (has_cls == 0: out[b][n] = patch + pos)
template <typename T>
__global__ void vit_embed_fwd_kernel(const T* __restrict__ patch, const float* __restrict__ pos, const float* __restrict__ cls, T* __restrict__ out,
                                     int B, int N, int E, int has_cls) {
  const int eg = E / 8;
  const int S = N + has_cls;
  const size_t total = (size_t)B * S * eg;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int e0 = (int)(idx % eg) * 8;
    const size_t t = idx / eg;
    const int sidx = (int)(t % S);
    const size_t b = t / S;
    float f[8];
    if (has_cls && sidx == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = cls[e0 + j];
    } else {
      const int n = sidx - has_cls;
      v8_unpack(v8_load<T>(patch + (b * N + n) * E + e0), f);
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] += pos[(size_t)n * E + e0 + j];
    }
    V8<T> o;
    v8_pack(f, o);
    v8_store<T>(out + idx * 8, o);
  }
}

// dpatch[b][n][:] = dout[b][has_cls + n][:]
template <typename T>
__global__ void vit_embed_bwd_kernel(const T* __restrict__ dout, T* __restrict__ dpatch, int B, int N, int E, int has_cls) {
  const int eg = E / 8;
  const int S = N + has_cls;
  const size_t total = (size_t)B * N * eg;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int e0 = (int)(idx % eg) * 8;
    const size_t t = idx / eg;
    const int n = (int)(t % N);
    const size_t b = t / N;
    v8_store<T>(dpatch + idx * 8, v8_load<T>(dout + (b * S + n + has_cls) * E + e0));
  }
}

// out[j] (+)= sum_b x[b*L + j]   (fp32 result; L % 8 == 0)
template <typename T>
__global__ void batch_sum_kernel(const T* __restrict__ x, float* __restrict__ out, int B, size_t L, int accumulate) {
  const size_t lg = L / 8;
  for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < lg; g += (size_t)gridDim.x * blockDim.x) {
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int b = 0; b < B; ++b) {
      float f[8];
      v8_unpack(v8_load<T>(x + (size_t)b * L + g * 8), f);
#pragma unroll
      for (int j = 0; j < 8; ++j) s[j] += f[j];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) out[g * 8 + j] = accumulate ? out[g * 8 + j] + s[j] : s[j];
  }
}

// dst[r*dst_stride + c] = src[r*src_stride + c], c < C (C % 8 == 0)
template <typename T>
__global__ void rows_copy_kernel(const T* __restrict__ src, T* __restrict__ dst, size_t rows, int C, size_t src_stride, size_t dst_stride) {
  const int cg = C / 8;
  const size_t total = rows * cg;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int c0 = (int)(idx % cg) * 8;
    const size_t r = idx / cg;
    v8_store<T>(dst + r * dst_stride + c0, v8_load<T>(src + r * src_stride + c0));
  }
}

// text tower: out[b][s][:] = table[tok[b][s]][:] + pos[s][:]
template <typename T>
__global__ void embed_lookup_fwd_kernel(const long long* __restrict__ tok, const float* __restrict__ table, const float* __restrict__ pos,
                                        T* __restrict__ out, size_t rows, int S, int E) {
  const int eg = E / 8;
  const size_t total = rows * eg;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int e0 = (int)(idx % eg) * 8;
    const size_t r = idx / eg;
    const long long t = tok[r];
    const int sp = (int)(r % S);
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = table[(size_t)t * E + e0 + j] + (pos ? pos[(size_t)sp * E + e0 + j] : 0.f);
    V8<T> o;
    v8_pack(f, o);
    v8_store<T>(out + idx * 8, o);
  }
}
// dtable[tok[r]][:] += dout[r][:]   (fp32 atomics: repeated tokens collide by design)
template <typename T>
__global__ void embed_lookup_bwd_kernel(const long long* __restrict__ tok, const T* __restrict__ dout, float* __restrict__ dtable, size_t rows, int E,
                                        long long padding_idx) {
  const int eg = E / 8;
  const size_t total = rows * eg;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int e0 = (int)(idx % eg) * 8;
    const size_t r = idx / eg;
    const long long t = tok[r];
    if (t == padding_idx) continue;
    float f[8];
    v8_unpack(v8_load<T>(dout + idx * 8), f);
#pragma unroll
    for (int j = 0; j < 8; ++j) atomicAdd(dtable + (size_t)t * E + e0 + j, f[j]);
  }
}

// dst[b][:] = src[idx[b]][:]  (EOT-token rows, text_encoders/transformer.py:413-421) ; scatter form for the backward (dst pre-zeroed)
template <typename T>
__global__ void rows_gather_idx_kernel(const T* __restrict__ src, const long long* __restrict__ idx, T* __restrict__ dst, int R, int C, int scatter) {
  const int cg = C / 8;
  const size_t total = (size_t)R * cg;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c0 = (int)(i % cg) * 8;
    const size_t r = i / cg;
    const size_t sr = (size_t)idx[r];
    if (scatter) v8_store<T>(dst + sr * C + c0, v8_load<T>(src + r * C + c0));
    else v8_store<T>(dst + r * C + c0, v8_load<T>(src + sr * C + c0));
  }
}

// F.normalize(x, dim=-1): y = x / max(||x||_2, eps) — one wave per row (simple_projection_head.py:83-84, transformer.py:424-425)
template <typename T>
__global__ void __launch_bounds__(256) l2norm_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, float* __restrict__ inv_norm, int R, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  const T* xr = x + (size_t)r * C;
  float ss = 0.f;
  for (int c = lane * 8; c < C; c += 512) {
    float f[8];
    v8_unpack(v8_load<T>(xr + c), f);
#pragma unroll
    for (int j = 0; j < 8; ++j) ss += f[j] * f[j];
  }
  ss = wave_sum(ss);
  const float inv = 1.0f / fmaxf(sqrtf(ss), eps);
  if (lane == 0) inv_norm[r] = inv;
  for (int c = lane * 8; c < C; c += 512) {
    float f[8];
    v8_unpack(v8_load<T>(xr + c), f);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] *= inv;
    V8<T> o;
    v8_pack(f, o);
    v8_store<T>(y + (size_t)r * C + c, o);
  }
}
// dx = inv * (dy - y * <dy, y>)   (rows whose norm was clamped by eps: dx = inv * dy)
template <typename T>
__global__ void __launch_bounds__(256) l2norm_bwd_kernel(const T* __restrict__ y, const T* __restrict__ dy, const float* __restrict__ inv_norm,
                                                         T* __restrict__ dx, int R, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  const float inv = inv_norm[r];
  const bool clamped = inv >= 1.0f / eps;
  float dot = 0.f;
  for (int c = lane * 8; c < C; c += 512) {
    float a[8], b[8];
    v8_unpack(v8_load<T>(y + (size_t)r * C + c), a);
    v8_unpack(v8_load<T>(dy + (size_t)r * C + c), b);
#pragma unroll
    for (int j = 0; j < 8; ++j) dot += a[j] * b[j];
  }
  dot = clamped ? 0.f : wave_sum(dot);
  for (int c = lane * 8; c < C; c += 512) {
    float a[8], b[8];
    v8_unpack(v8_load<T>(y + (size_t)r * C + c), a);
    v8_unpack(v8_load<T>(dy + (size_t)r * C + c), b);
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = inv * (b[j] - a[j] * dot);
    V8<T> o;
    v8_pack(a, o);
    v8_store<T>(dx + (size_t)r * C + c, o);
  }
}

// Contrastive cross-entropy (contrastive_loss_clip.py:77-94): z = scale * logits[i][:], label_i = i + label_offset;
// loss_rows[i] = logsumexp(z) - z[label] ; lse[i] saved.  One 256-thread block per row.
template <typename T>
__global__ void __launch_bounds__(256) scaled_ce_fwd_kernel(const T* __restrict__ logits, const float* __restrict__ scale, float* __restrict__ loss_rows,
                                                            float* __restrict__ lse, int N, int M, int label_offset) {
  __shared__ float scr[8];
  const int i = blockIdx.x;
  const float s = *scale;
  const T* row = logits + (size_t)i * M;
  float mx = -INFINITY;
  for (int j = threadIdx.x; j < M; j += 256) mx = fmaxf(mx, s * to_f<T>(row[j]));
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) scr[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(scr[0], scr[1]), fmaxf(scr[2], scr[3]));
  float sum = 0.f;
  for (int j = threadIdx.x; j < M; j += 256) sum += fast_exp(s * to_f<T>(row[j]) - mx);
  sum = wave_sum(sum);
  if ((threadIdx.x & 63) == 0) scr[4 + (threadIdx.x >> 6)] = sum;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float l = mx + __logf(scr[4] + scr[5] + scr[6] + scr[7]);
    lse[i] = l;
    loss_rows[i] = l - s * to_f<T>(row[i + label_offset]);
  }
}
// dlogits[i][j] = g * scale * (softmax_ij - [j == label_i]) ; dscale_rows[i] = g * sum_j logits_ij * (softmax_ij - [j == label_i])
// (g = upstream gradient per row, e.g. 0.5 / N)
template <typename T>
__global__ void __launch_bounds__(256) scaled_ce_bwd_kernel(const T* __restrict__ logits, const float* __restrict__ scale, const float* __restrict__ lse,
                                                            const float* __restrict__ gout, T* __restrict__ dlogits, float* __restrict__ dscale_rows,
                                                            int N, int M, int label_offset) {
  __shared__ float scr[4];
  const int i = blockIdx.x;
  const float s = *scale, l = lse[i], g = *gout;
  const T* row = logits + (size_t)i * M;
  const int label = i + label_offset;
  float acc = 0.f;
  for (int j = threadIdx.x; j < M; j += 256) {
    const float x = to_f<T>(row[j]);
    const float d = fast_exp(s * x - l) - (j == label ? 1.f : 0.f);
    acc += x * d;
    dlogits[(size_t)i * M + j] = from_f<T>(g * s * d);
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) scr[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) dscale_rows[i] = g * (scr[0] + scr[1] + scr[2] + scr[3]);
}

// Label-smoothed cross-entropy over [N][M] logits (loss_fn/classification/cross_entropy.py:65-92 -> F.cross_entropy(weight=None,
// ignore_index, label_smoothing)): loss_i = lse_i - (1-eps)*z[y_i] - eps*mean_j z_ij ; rows with y_i == ignore_index give 0.
// One 256-thread block per row; fp32 math whatever the logits dtype.
template <typename T>
__global__ __launch_bounds__(256) void ce_fwd_kernel(const T* __restrict__ logits, const long long* __restrict__ labels, float eps, long long ignore_index,
                                                     float* __restrict__ loss_rows, float* __restrict__ lse, int N, int M) {
  __shared__ float scr[12];
  const int i = blockIdx.x;
  const T* row = logits + (size_t)i * M;
  float mx = -INFINITY, sm = 0.f;
  for (int j = threadIdx.x; j < M; j += 256) {
    const float z = to_f<T>(row[j]);
    mx = fmaxf(mx, z);
    sm += z;
  }
  mx = wave_max(mx);
  sm = wave_sum(sm);
  if ((threadIdx.x & 63) == 0) { scr[threadIdx.x >> 6] = mx; scr[4 + (threadIdx.x >> 6)] = sm; }
  __syncthreads();
  mx = fmaxf(fmaxf(scr[0], scr[1]), fmaxf(scr[2], scr[3]));
  sm = (scr[4] + scr[5]) + (scr[6] + scr[7]);
  float ex = 0.f;
  for (int j = threadIdx.x; j < M; j += 256) ex += fast_exp(to_f<T>(row[j]) - mx);
  ex = wave_sum(ex);
  if ((threadIdx.x & 63) == 0) scr[8 + (threadIdx.x >> 6)] = ex;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float l = mx + __logf((scr[8] + scr[9]) + (scr[10] + scr[11]));
    lse[i] = l;
    const long long y = labels[i];
    loss_rows[i] = (y == ignore_index) ? 0.f : l - (1.f - eps) * to_f<T>(row[y]) - eps * sm / (float)M;
  }
}
// dlogits[i][j] = g * (softmax_ij - (1-eps)*[j == y_i] - eps/M),  g = *gout (upstream gradient / number of valid rows); 0 for ignored rows
template <typename T>
__global__ __launch_bounds__(256) void ce_bwd_kernel(const T* __restrict__ logits, const long long* __restrict__ labels, const float* __restrict__ lse,
                                                     const float* __restrict__ gout, float eps, long long ignore_index, T* __restrict__ dlogits, int N, int M) {
  const int i = blockIdx.x;
  const long long y = labels[i];
  const float l = lse[i], g = (y == ignore_index) ? 0.f : *gout, u = eps / (float)M;
  for (int j = threadIdx.x; j < M; j += 256) {
    const float p = fast_exp(to_f<T>(logits[(size_t)i * M + j]) - l);
    dlogits[(size_t)i * M + j] = from_f<T>(g * (p - (j == y ? 1.f - eps : 0.f) - u));
  }
}

// Probability targets (the [N][M] mixtures RandomMixup / RandomCutmix produce, data/transforms/image_torch.py:131-140): F.cross_entropy with
// class probabilities and label smoothing = sum_j t'_ij (lse_i - z_ij), t' = (1-eps) t + eps/M; mean over ALL rows (no ignore_index).
template <typename T>
__global__ __launch_bounds__(256) void ce_soft_fwd_kernel(const T* __restrict__ logits, const float* __restrict__ target, float eps, float* __restrict__ loss_rows,
                                                          float* __restrict__ lse, float* __restrict__ tsum, int N, int M) {
  __shared__ float scr[16];
  const int i = blockIdx.x;
  const T* row = logits + (size_t)i * M;
  const float* trow = target + (size_t)i * M;
  float mx = -INFINITY, sm = 0.f, ts = 0.f, tz = 0.f;
  for (int j = threadIdx.x; j < M; j += 256) {
    const float z = to_f<T>(row[j]), t = trow[j];
    mx = fmaxf(mx, z);
    sm += z;
    ts += t;
    tz += t * z;
  }
  mx = wave_max(mx); sm = wave_sum(sm); ts = wave_sum(ts); tz = wave_sum(tz);
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { scr[wv] = mx; scr[4 + wv] = sm; scr[8 + wv] = ts; scr[12 + wv] = tz; }
  __syncthreads();
  mx = fmaxf(fmaxf(scr[0], scr[1]), fmaxf(scr[2], scr[3]));
  sm = (scr[4] + scr[5]) + (scr[6] + scr[7]);
  ts = (scr[8] + scr[9]) + (scr[10] + scr[11]);
  tz = (scr[12] + scr[13]) + (scr[14] + scr[15]);
  __syncthreads();
  float ex = 0.f;
  for (int j = threadIdx.x; j < M; j += 256) ex += fast_exp(to_f<T>(row[j]) - mx);
  ex = wave_sum(ex);
  if ((threadIdx.x & 63) == 0) scr[wv] = ex;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float l = mx + __logf((scr[0] + scr[1]) + (scr[2] + scr[3]));
    const float w = (1.f - eps) * ts + eps;  // sum_j t'_ij
    lse[i] = l;
    tsum[i] = w;
    loss_rows[i] = w * l - (1.f - eps) * tz - eps * sm / (float)M;
  }
}
template <typename T>
__global__ __launch_bounds__(256) void ce_soft_bwd_kernel(const T* __restrict__ logits, const float* __restrict__ target, const float* __restrict__ lse,
                                                          const float* __restrict__ tsum, const float* __restrict__ gout, float eps, T* __restrict__ dlogits,
                                                          int N, int M) {
  const int i = blockIdx.x;
  const float l = lse[i], w = tsum[i], g = *gout, u = eps / (float)M;
  for (int j = threadIdx.x; j < M; j += 256) {
    const float p = fast_exp(to_f<T>(logits[(size_t)i * M + j]) - l);
    dlogits[(size_t)i * M + j] = from_f<T>(g * (p * w - (1.f - eps) * target[(size_t)i * M + j] - u));
  }
}

static inline int tk_grid(size_t total) {
  size_t g = (total + 255) / 256;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (int)g;
}
#define TK_DISPATCH(dtype, ...)                                   \
  if ((dtype) == CVH_DT_BF16) { using T = bf16_t; __VA_ARGS__ }   \
  else if ((dtype) == CVH_DT_F32) { using T = float; __VA_ARGS__ } \
  else return -1;

extern "C" int cvh_vit_embed_fwd(int dtype, const void* patch, const float* pos, const float* cls, void* out, int B, int N, int E, void* stream) {
  if (E % 8) return -2;
  const int has_cls = cls != nullptr;
  const size_t total = (size_t)B * (N + has_cls) * (E / 8);
  TK_DISPATCH(dtype, hipLaunchKernelGGL((vit_embed_fwd_kernel<T>), dim3(tk_grid(total)), dim3(256), 0, (hipStream_t)stream, (const T*)patch, pos, cls, (T*)out, B, N, E, has_cls);)
  CVH_CHECK_LAUNCH();
  return 0;
}
extern "C" int cvh_vit_embed_bwd(int dtype, const void* dout, void* dpatch, int B, int N, int E, int has_cls, void* stream) {
  if (E % 8) return -2;
  const size_t total = (size_t)B * N * (E / 8);
  TK_DISPATCH(dtype, hipLaunchKernelGGL((vit_embed_bwd_kernel<T>), dim3(tk_grid(total)), dim3(256), 0, (hipStream_t)stream, (const T*)dout, (T*)dpatch, B, N, E, has_cls);)
  CVH_CHECK_LAUNCH();
  return 0;
}
extern "C" int cvh_batch_sum(int dtype, const void* x, float* out, int B, long long L, int accumulate, void* stream) {
  if (L % 8) return -2;
  TK_DISPATCH(dtype, hipLaunchKernelGGL((batch_sum_kernel<T>), dim3(tk_grid((size_t)L / 8)), dim3(256), 0, (hipStream_t)stream, (const T*)x, out, B, (size_t)L, accumulate);)
  CVH_CHECK_LAUNCH();
  return 0;
}
extern "C" int cvh_rows_copy(int dtype, const void* src, void* dst, long long rows, int C, long long src_stride, long long dst_stride, void* stream) {
  if (C % 8 || src_stride % 8 || dst_stride % 8) return -2;
  const size_t total = (size_t)rows * (C / 8);
  TK_DISPATCH(dtype, hipLaunchKernelGGL((rows_copy_kernel<T>), dim3(tk_grid(total)), dim3(256), 0, (hipStream_t)stream, (const T*)src, (T*)dst, (size_t)rows, C, (size_t)src_stride, (size_t)dst_stride);)
  CVH_CHECK_LAUNCH();
  return 0;
}
extern "C" int cvh_embed_lookup_fwd(int dtype, const long long* tok, const float* table, const float* pos, void* out, long long rows, int S, int E,
                                    void* stream) {
  if (E % 8) return -2;
  const size_t total = (size_t)rows * (E / 8);
  TK_DISPATCH(dtype, hipLaunchKernelGGL((embed_lookup_fwd_kernel<T>), dim3(tk_grid(total)), dim3(256), 0, (hipStream_t)stream, tok, table, pos, (T*)out, (size_t)rows, S, E);)
  CVH_CHECK_LAUNCH();
  return 0;
}
extern "C" int cvh_embed_lookup_bwd(int dtype, const long long* tok, const void* dout, float* dtable, long long rows, int E, long long padding_idx,
                                    void* stream) {
  if (E % 8) return -2;
  const size_t total = (size_t)rows * (E / 8);
  TK_DISPATCH(dtype, hipLaunchKernelGGL((embed_lookup_bwd_kernel<T>), dim3(tk_grid(total)), dim3(256), 0, (hipStream_t)stream, tok, (const T*)dout, dtable, (size_t)rows, E, padding_idx);)
  CVH_CHECK_LAUNCH();
  return 0;
}
extern "C" int cvh_rows_gather_idx(int dtype, const void* src, const long long* idx, void* dst, int R, int C, int scatter, void* stream) {
  if (C % 8) return -2;
  const size_t total = (size_t)R * (C / 8);
  TK_DISPATCH(dtype, hipLaunchKernelGGL((rows_gather_idx_kernel<T>), dim3(tk_grid(total)), dim3(256), 0, (hipStream_t)stream, (const T*)src, idx, (T*)dst, R, C, scatter);)
  CVH_CHECK_LAUNCH();
  return 0;
}
extern "C" int cvh_l2norm_fwd(int dtype, const void* x, void* y, float* inv_norm, int R, int C, float eps, void* stream) {
  if (C % 8) return -2;
  TK_DISPATCH(dtype, hipLaunchKernelGGL((l2norm_fwd_kernel<T>), dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const T*)x, (T*)y, inv_norm, R, C, eps);)
  CVH_CHECK_LAUNCH();
  return 0;
}
extern "C" int cvh_l2norm_bwd(int dtype, const void* y, const void* dy, const float* inv_norm, void* dx, int R, int C, float eps, void* stream) {
  if (C % 8) return -2;
  TK_DISPATCH(dtype, hipLaunchKernelGGL((l2norm_bwd_kernel<T>), dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const T*)y, (const T*)dy, inv_norm, (T*)dx, R, C, eps);)
  CVH_CHECK_LAUNCH();
  return 0;
}
extern "C" int cvh_scaled_ce_fwd(int dtype, const void* logits, const float* scale, float* loss_rows, float* lse, int N, int M, int label_offset,
                                 void* stream) {
  if (N <= 0 || M <= 0 || label_offset < 0 || N + label_offset > M) return -2;
  TK_DISPATCH(dtype, hipLaunchKernelGGL((scaled_ce_fwd_kernel<T>), dim3(N), dim3(256), 0, (hipStream_t)stream, (const T*)logits, scale, loss_rows, lse, N, M, label_offset);)
  CVH_CHECK_LAUNCH();
  return 0;
}
extern "C" int cvh_scaled_ce_bwd(int dtype, const void* logits, const float* scale, const float* lse, const float* gout, void* dlogits,
                                 float* dscale_rows, int N, int M, int label_offset, void* stream) {
  if (N <= 0 || M <= 0 || label_offset < 0 || N + label_offset > M) return -2;
  TK_DISPATCH(dtype, hipLaunchKernelGGL((scaled_ce_bwd_kernel<T>), dim3(N), dim3(256), 0, (hipStream_t)stream, (const T*)logits, scale, lse, gout, (T*)dlogits, dscale_rows, N, M, label_offset);)
  CVH_CHECK_LAUNCH();
  return 0;
}
extern "C" int cvh_ce_fwd(int dtype, const void* logits, const long long* labels, float label_smoothing, long long ignore_index, float* loss_rows,
                          float* lse, int N, int M, void* stream) {
  if (N <= 0 || M <= 0) return -2;
  TK_DISPATCH(dtype, hipLaunchKernelGGL((ce_fwd_kernel<T>), dim3(N), dim3(256), 0, (hipStream_t)stream, (const T*)logits, labels, label_smoothing, ignore_index, loss_rows, lse, N, M);)
  CVH_CHECK_LAUNCH();
  return 0;
}
// loss = sum(loss_rows) / max(1, #rows whose label is not ignore_index) in ONE launch (F.cross_entropy's 'mean' reduction,
// loss_fn/classification/cross_entropy.py:65-92); out[0] = loss, out[1] = 1 / that count (the backward's scale).  One workgroup, fixed-order tree.
__global__ __launch_bounds__(1024) void ce_mean_kernel(const float* __restrict__ rows, const long long* __restrict__ labels, long long ignore_index,
                                                       float* __restrict__ out, int N) {
  __shared__ float ssum[1024];
  __shared__ int scnt[1024];
  float s = 0.f;
  int c = 0;
  for (int i = threadIdx.x; i < N; i += 1024) {
    s += rows[i];
    c += labels[i] != ignore_index ? 1 : 0;
  }
  ssum[threadIdx.x] = s;
  scnt[threadIdx.x] = c;
  __syncthreads();
  for (int st = 512; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) {
      ssum[threadIdx.x] += ssum[threadIdx.x + st];
      scnt[threadIdx.x] += scnt[threadIdx.x + st];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float n = (float)(scnt[0] > 0 ? scnt[0] : 1);
    out[0] = ssum[0] / n;
    out[1] = 1.0f / n;
  }
}
extern "C" int cvh_ce_mean(const float* loss_rows, const long long* labels, long long ignore_index, float* out2, int N, void* stream) {
  if (N <= 0) return -2;
  hipLaunchKernelGGL(ce_mean_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, loss_rows, labels, ignore_index, out2, N);
  CVH_CHECK_LAUNCH();
  return 0;
}

extern "C" int cvh_ce_bwd(int dtype, const void* logits, const long long* labels, const float* lse, const float* gout, float label_smoothing,
                          long long ignore_index, void* dlogits, int N, int M, void* stream) {
  if (N <= 0 || M <= 0) return -2;
  TK_DISPATCH(dtype, hipLaunchKernelGGL((ce_bwd_kernel<T>), dim3(N), dim3(256), 0, (hipStream_t)stream, (const T*)logits, labels, lse, gout, label_smoothing, ignore_index, (T*)dlogits, N, M);)
  CVH_CHECK_LAUNCH();
  return 0;
}
extern "C" int cvh_ce_soft_fwd(int dtype, const void* logits, const float* target, float label_smoothing, float* loss_rows, float* lse, float* tsum,
                               int N, int M, void* stream) {
  if (N <= 0 || M <= 0) return -2;
  TK_DISPATCH(dtype, hipLaunchKernelGGL((ce_soft_fwd_kernel<T>), dim3(N), dim3(256), 0, (hipStream_t)stream, (const T*)logits, target, label_smoothing, loss_rows, lse, tsum, N, M);)
  CVH_CHECK_LAUNCH();
  return 0;
}
extern "C" int cvh_ce_soft_bwd(int dtype, const void* logits, const float* target, const float* lse, const float* tsum, const float* gout,
                               float label_smoothing, void* dlogits, int N, int M, void* stream) {
  if (N <= 0 || M <= 0) return -2;
  TK_DISPATCH(dtype, hipLaunchKernelGGL((ce_soft_bwd_kernel<T>), dim3(N), dim3(256), 0, (hipStream_t)stream, (const T*)logits, target, lse, tsum, gout, label_smoothing, (T*)dlogits, N, M);)
  CVH_CHECK_LAUNCH();
  return 0;
}
