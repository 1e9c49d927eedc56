// HBM-bound NHWC kernels: layout/dtype conversion, weight packing, BatchNorm (train-mode statistics,
// apply, backward), column reductions, global pooling, dropout backward.
// All are streaming kernels: 16 B (bf16) / 32 B (f32) per lane, channel index = fastest dimension.
//
// Replaces (reference, ATen calls): nn.BatchNorm2d cvnets/layers/normalization/batch_norm.py:14-49,
// nn.SiLU cvnets/layers/activation/swish.py, GlobalPool cvnets/layers/global_pool.py:60-71,
// Dropout cvnets/layers/dropout.py:11-29 and their autograd backward.
#include "common.hpp"
#include "cvnets_hip.h"

// ---------------------------------------------------------------------------------------------
// NCHW f32  ->  NHWC T (channels zero-padded to Cp, Cp % 8 == 0)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ in, T* __restrict__ out, int B, int C, int H, int W, int Cp) {
  const size_t npix = (size_t)B * H * W;
  const int cgs = Cp / 8;
  const size_t total = npix * cgs;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const size_t pix = idx % npix;
    const int cg = (int)(idx / npix);
    const size_t hw = (size_t)H * W;
    const size_t b = pix / hw, rem = pix - b * hw;
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      int c = cg * 8 + j;
      f[j] = c < C ? in[(b * C + c) * hw + rem] : 0.f;
    }
    V8<T> v;
    v8_pack(f, v);
    v8_store<T>(out + pix * Cp + cg * 8, v);
  }
}

// NHWC T -> NCHW f32 (first C of Cs channels)
template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ in, float* __restrict__ out, int B, int C, int H, int W, int Cs) {
  const size_t hw = (size_t)H * W;
  const size_t total = (size_t)B * C * hw;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const size_t rem = idx % hw;
    const size_t bc = idx / hw;
    const size_t b = bc / C, c = bc - b * C;
    out[idx] = to_f<T>(in[(b * hw + rem) * Cs + c]);
  }
}

// ---------------------------------------------------------------------------------------------
// weight packing: torch [Cout][Cin][KH][KW] f32  ->  T
//   mode 0 (forward):   out[n][tap*Cp + c]            = w[n][c][tap]          rows = Cout, Cp = pad8(Cin)
//   mode 1 (dX, s=1):   out[c][tap*Np + n]            = w[n][c][KH*KW-1-tap]  rows = Cin (transposed, taps flipped), Np = pad8(Cout)
//   mode 2 (depthwise): out[tap*C + c]                = w[c][0][tap]
//   mode 3 (patch dX):  out[tap*Cp + c][n]            = w[n][c][tap]          rows = (tap, c), Np = pad8(Cout)  (kernel == stride convs)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void weight_pack_kernel(const float* __restrict__ w, T* __restrict__ out, int Cout, int Cin, int KHW, int Cp, int Np,
                                   int mode) {
  size_t total;
  if (mode == 0) total = (size_t)Cout * KHW * Cp;
  else if (mode == 1) total = (size_t)Cin * KHW * Np;
  else if (mode == 3) total = (size_t)KHW * Cp * Np;
  else total = (size_t)KHW * Cout;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    float v = 0.f;
    if (mode == 0) {
      int c = (int)(idx % Cp);
      size_t t = idx / Cp;
      int tap = (int)(t % KHW);
      int n = (int)(t / KHW);
      if (c < Cin) v = w[((size_t)n * Cin + c) * KHW + tap];
    } else if (mode == 1) {
      int n = (int)(idx % Np);
      size_t t = idx / Np;
      int tap = (int)(t % KHW);
      int c = (int)(t / KHW);
      if (n < Cout) v = w[((size_t)n * Cin + c) * KHW + (KHW - 1 - tap)];
    } else if (mode == 3) {
      int n = (int)(idx % Np);
      size_t t = idx / Np;
      int c = (int)(t % Cp);
      int tap = (int)(t / Cp);
      if (n < Cout && c < Cin) v = w[((size_t)n * Cin + c) * KHW + tap];
    } else {
      int c = (int)(idx % Cout);
      int tap = (int)(idx / Cout);
      v = w[(size_t)c * KHW + tap];
    }
    out[idx] = from_f<T>(v);
  }
}

// all weights of a model in ONE launch.  table[e] = {src ptr, dst element offset, Cout, Cin, KHW, mode, first global element}
// (7 x int64 per entry, entries sorted by first element; table[n][6] = total).  Same three layouts as above.
template <typename T>
__global__ void weight_pack_multi_kernel(const long long* __restrict__ table, int n, long long total, T* __restrict__ out) {
  // thread = 8 consecutive packed elements = one 16-byte (bf16) store.  Every tensor's packed size and row length are multiples
  // of 8, so a chunk never straddles tensors or rows: the 8 sources are one base address + a constant stride.
  const long long total8 = total / 8;
  for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < total8; q += (long long)gridDim.x * blockDim.x) {
    const long long idx = q * 8;
    int lo = 0, hi = n - 1;
    while (lo < hi) {  // last entry whose first element <= idx
      const int mid = (lo + hi + 1) >> 1;
      if (table[(size_t)mid * 7 + 6] <= idx) lo = mid; else hi = mid - 1;
    }
    const long long* e = table + (size_t)lo * 7;
    const float* w = reinterpret_cast<const float*>(e[0]);
    const unsigned li = (unsigned)(idx - e[6]);  // position inside this tensor's packed image (< 2^32)
    const unsigned Cout = (unsigned)e[2], Cin = (unsigned)e[3], KHW = (unsigned)e[4];
    const int mode = (int)e[5];
    const unsigned Cp = (Cin + 7) / 8 * 8, Np = (Cout + 7) / 8 * 8;
    size_t base;       // source index of element 0 of the chunk
    unsigned stride;   // source distance between consecutive packed elements
    unsigned first, limit;  // elements j with first + j >= limit are padding (zero)
    if (mode == 0) {   // [Cout][KHW][Cp]
      const unsigned c = li % Cp, t = li / Cp, tap = t % KHW, nn = t / KHW;
      base = ((size_t)nn * Cin + c) * KHW + tap; stride = KHW; first = c; limit = Cin;
    } else if (mode == 1) {  // [Cin][KHW flipped][Np]
      const unsigned nn = li % Np, t = li / Np, tap = t % KHW, c = t / KHW;
      base = ((size_t)nn * Cin + c) * KHW + (KHW - 1 - tap); stride = Cin * KHW; first = nn; limit = Cout;
    } else if (mode == 3) {  // [KHW][Cp][Np]
      const unsigned nn = li % Np, t = li / Np, c = t % Cp, tap = t / Cp;
      base = ((size_t)nn * Cin + c) * KHW + tap; stride = Cin * KHW; first = nn; limit = c < Cin ? Cout : 0;
    } else {  // depthwise [KHW][C]
      const unsigned c = li % Cout, tap = li / Cout;
      base = (size_t)c * KHW + tap; stride = KHW; first = c; limit = Cout;
    }
    float v[8];
#pragma unroll
    for (unsigned jj = 0; jj < 8; ++jj) v[jj] = (first + jj < limit) ? w[base + (size_t)jj * stride] : 0.f;
    V8<T> o;
    v8_pack(v, o);
    v8_store<T>(out + e[1] + li, o);
  }
}

// ---------------------------------------------------------------------------------------------
// Device-side input stage: RandomMixup / RandomCutmix of data/transforms/image_torch.py (applied at
// engine/training_engine.py:238) fused with the NCHW float32 -> NHWC `T` conversion of the model's first op.
//   outside the box: out[b] = lam * x[b] + (1 - lam) * x[(b - 1) mod B]      (mixup: the box is empty; cutmix: lam = 1)
//   inside  the box [y1, y2) x [x1, x2):  out[b] = x[(b - 1) mod B]           (cutmix paste from the batch rolled by one)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void mix_to_nhwc_kernel(const float* __restrict__ in, T* __restrict__ out, int B, int C, int H, int W, int Cp, float lam, int x1, int y1,
                                   int x2, int y2) {
  const size_t npix = (size_t)B * H * W;
  const int cgs = Cp / 8;
  const size_t total = npix * cgs;
  const size_t hw = (size_t)H * W;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const size_t pix = idx % npix;
    const int cg = (int)(idx / npix);
    const size_t b = pix / hw, rem = pix - b * hw;
    const size_t br = (b + B - 1) % B;
    const int h = (int)(rem / W), w = (int)(rem - (size_t)h * W);
    const bool inside = h >= y1 && h < y2 && w >= x1 && w < x2;
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = cg * 8 + j;
      float v = 0.f;
      if (c < C) {
        const float a = in[(b * C + c) * hw + rem], r = in[(br * C + c) * hw + rem];
        v = inside ? r : (lam * a + (1.f - lam) * r);
      }
      f[j] = v;
    }
    V8<T> v;
    v8_pack(f, v);
    v8_store<T>(out + pix * Cp + cg * 8, v);
  }
}
// same mix, NCHW float32 -> NCHW float32 (the reference's own output format, for callers that keep their model-side conversion)
__global__ void mix_nchw_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int C, int H, int W, float lam, int x1, int y1, int x2,
                                int y2) {
  const size_t hw = (size_t)H * W, chw = (size_t)C * hw, total = (size_t)B * chw;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const size_t b = idx / chw, rem = idx - b * chw;
    const size_t p = rem % hw;
    const int h = (int)(p / W), w = (int)(p - (size_t)h * W);
    const float a = in[idx], r = in[((b + B - 1) % B) * chw + rem];
    out[idx] = (h >= y1 && h < y2 && w >= x1 && w < x2) ? r : (lam * a + (1.f - lam) * r);
  }
}

// ---------------------------------------------------------------------------------------------
// StochasticDepth, torchvision "row" mode (cvnets/layers/stochastic_depth.py:10-18): y = res + x * keep(sample) / (1 - p), one
// Bernoulli(1 - p) draw per SAMPLE of the block input.  Rows are tokens; the sample of a row follows the unfold map of cvh_attn_*
// (nseq sequences of S tokens: a MobileViT block sees B*ph*pw samples), ph = pw = 1, H = 1, W = n_w = S is the contiguous case.
// res == nullptr: y = x * keep (also the backward: dx = dy * keep with the same seed / stream id).
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void drop_path_kernel(const T* __restrict__ x, const T* __restrict__ res, T* __restrict__ y, size_t rows, int C,
                                                        int ph, int pw, int H, int W, float p, const unsigned long long* __restrict__ seed_p,
                                                        unsigned int stream_id) {
  const unsigned long long seed = *seed_p;
  const float inv_keep = 1.0f / (1.0f - p);
  const int cgs = C / 8;
  const size_t total = rows * cgs;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const size_t r = idx / cgs;
    const int cg = (int)(idx - r * cgs);
    // row -> (b, h, w) -> sample b * ph * pw + (h % ph) * pw + (w % pw)
    const size_t hw = (size_t)H * W;
    const size_t b = r / hw, rem = r - b * hw;
    const int h = (int)(rem / W), w = (int)(rem - (size_t)h * W);
    const size_t sample = b * ph * pw + (size_t)(h % ph) * pw + (w % pw);
    const float k = dropout_scale(seed, stream_id, sample, p, inv_keep);
    float f[8];
    v8_unpack(v8_load<T>(x + r * C + cg * 8), f);
    if (res != nullptr) {
      float q[8];
      v8_unpack(v8_load<T>(res + r * C + cg * 8), q);
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = q[j] + f[j] * k;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] *= k;
    }
    V8<T> o;
    v8_pack(f, o);
    v8_store<T>(y + r * C + cg * 8, o);
  }
}

// f32 vector -> T (bias etc.), or T -> f32
template <typename T>
__global__ void cast_from_f32_kernel(const float* __restrict__ in, T* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = from_f<T>(in[i]);
}
template <typename T>
__global__ void cast_to_f32_kernel(const T* __restrict__ in, float* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = to_f<T>(in[i]);
}

// ---------------------------------------------------------------------------------------------
// generic column reduction over a [rows][C] NHWC matrix.  MODE selects what is summed:
//   0: (x, x*x)                      BatchNorm forward statistics
//   1: (dz, dz*xhat)                 BatchNorm backward: dz = dout * act'(x*scale+shift), xhat = (x-mean)*invstd
//   2: (x, 0)                        plain column sum (bias gradients)
// Block = 256 threads = (C/8 channel groups) x (row lanes).  Writes part[block][2][C].
// ---------------------------------------------------------------------------------------------
template <typename T, int MODE>
__global__ __launch_bounds__(256) void colreduce_kernel(const T* __restrict__ x, const T* __restrict__ dout, const float* __restrict__ scale,
                                                        const float* __restrict__ shift, const float* __restrict__ mean,
                                                        const float* __restrict__ invstd, int act, size_t rows, int C, float* __restrict__ part,
                                                        size_t ld) {
  __shared__ float red[4096];
  const int cgs = C / 8;
  const int RL = 256 / cgs;
  const int ci = threadIdx.x % cgs;
  const int rl = threadIdx.x / cgs;
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s1[j] = s2[j] = 0.f;
  float sc[8], sh[8], mu[8], is[8];
  if (MODE == 1) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      sc[j] = scale[ci * 8 + j]; sh[j] = shift[ci * 8 + j]; mu[j] = mean[ci * 8 + j]; is[j] = invstd[ci * 8 + j];
    }
  }
  if (rl < RL) {
    const size_t step = (size_t)gridDim.x * RL;
    for (size_t r = (size_t)blockIdx.x * RL + rl; r < rows; r += 2 * step) {  // two rows in flight per thread
      const size_t r2 = r + step;
      const bool two = r2 < rows;
      V8<T> xa = v8_load<T>(x + r * ld + ci * 8);
      V8<T> xb = two ? v8_load<T>(x + r2 * ld + ci * 8) : v8_zero<T>();
      V8<T> da = v8_zero<T>(), db = v8_zero<T>();
      if (MODE == 1) {
        da = v8_load<T>(dout + r * ld + ci * 8);
        if (two) db = v8_load<T>(dout + r2 * ld + ci * 8);
      }
      float xf[8], xg[8];
      v8_unpack(xa, xf);
      v8_unpack(xb, xg);
      if (MODE == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { s1[j] += xf[j] + xg[j]; s2[j] += xf[j] * xf[j] + xg[j] * xg[j]; }
      } else if (MODE == 2) {
#pragma unroll
        for (int j = 0; j < 8; ++j) s1[j] += xf[j] + xg[j];
      } else {
        float df[8], dg[8];
        v8_unpack(da, df);
        v8_unpack(db, dg);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float dz = df[j] * act_grad(xf[j] * sc[j] + sh[j], act);
          s1[j] += dz;
          s2[j] += dz * (xf[j] - mu[j]) * is[j];
          if (two) {
            float dz2 = dg[j] * act_grad(xg[j] * sc[j] + sh[j], act);
            s1[j] += dz2;
            s2[j] += dz2 * (xg[j] - mu[j]) * is[j];
          }
        }
      }
    }
  }
  // reduce over row lanes through LDS: red[rl][2][C]
  if (rl < RL) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      red[(rl * 2 + 0) * C + ci * 8 + j] = s1[j];
      red[(rl * 2 + 1) * C + ci * 8 + j] = s2[j];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    float s = 0.f;
    for (int l = 0; l < RL; ++l) s += red[l * 2 * C + i];
    part[(size_t)blockIdx.x * 2 * C + i] = s;
  }
}

// sum of part[r*stride + off] over r = rl, rl+64, ... < R with 8 independent loads in flight (fp32 pairwise sum of each group of
// 8 partial rows, fp64 across groups): the finalize kernels are latency chains, not bandwidth problems.
__device__ __forceinline__ double strided_partial_sum(const float* __restrict__ part, int R, size_t stride, size_t off, int rl) {
  double s = 0.0;
  int r = rl;
  for (; r + 15 * 64 < R; r += 16 * 64) {  // 16 independent loads in flight: these kernels are latency chains, not bandwidth problems
    float v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = part[(size_t)(r + j * 64) * stride + off];
    s += (double)((((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]))) +
                  (((v[8] + v[9]) + (v[10] + v[11])) + ((v[12] + v[13]) + (v[14] + v[15]))));
  }
  float v[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {  // tail: still issued together (predicated)
    const int rr = r + j * 64;
    v[j] = rr < R ? part[(size_t)rr * stride + off] : 0.f;
  }
  s += (double)((((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]))) +
                (((v[8] + v[9]) + (v[10] + v[11])) + ((v[12] + v[13]) + (v[14] + v[15]))));
  return s;
}
// two column sums (columns off_a / off_b of the same partial rows) with all their loads in flight together
__device__ __forceinline__ void strided_partial_sum2(const float* __restrict__ part, int R, size_t stride, size_t off_a, size_t off_b, int rl,
                                                     double& sa, double& sb) {
  sa = sb = 0.0;
  for (int r = rl; r < R; r += 16 * 64) {
    float a[16], b[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int rr = r + j * 64;
      a[j] = rr < R ? part[(size_t)rr * stride + off_a] : 0.f;
      b[j] = rr < R ? part[(size_t)rr * stride + off_b] : 0.f;
    }
    sa += (double)((((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]))) +
                   (((a[8] + a[9]) + (a[10] + a[11])) + ((a[12] + a[13]) + (a[14] + a[15]))));
    sb += (double)((((b[0] + b[1]) + (b[2] + b[3])) + ((b[4] + b[5]) + (b[6] + b[7]))) +
                   (((b[8] + b[9]) + (b[10] + b[11])) + ((b[12] + b[13]) + (b[14] + b[15]))));
  }
}

// sum partial rows: part[R][W] -> out[W]  (f64 accumulation); block = 64 columns x 16 row lanes
__global__ __launch_bounds__(1024) void sum_partials_kernel(const float* __restrict__ part, int R, int stride, int Wd, float* __restrict__ out, float scale, int accumulate) {
  __shared__ double red[64][16];
  const int col = blockIdx.x * 16 + (threadIdx.x & 15);
  const int rl = threadIdx.x >> 4;
  double s = 0.0;
  if (col < Wd) s = strided_partial_sum(part, R, (size_t)stride, (size_t)col, rl);
  red[rl][threadIdx.x & 15] = s;
  __syncthreads();
  if (rl == 0 && col < Wd) {
    double t = 0.0;
#pragma unroll
    for (int l = 0; l < 64; ++l) t += red[l][threadIdx.x & 15];
    const float r = (float)(t * (double)scale);
    out[col] = accumulate ? out[col] + r : r;
  }
}

// ---------------------------------------------------------------------------------------------
// Deferred multi-tensor reduction: every "sum the partial rows" tail of a backward pass whose result only the optimizer reads
// (dW split partials, LayerNorm dgamma / dbeta, bias gradients, depthwise dW) is queued by the host and executed by ONE launch per
// <= CVH_REDUCE_MAX descriptors at the end of backward, instead of ~140 latency-bound 5-20 us launches per MobileViT-S step.
// Descriptors travel in the kernel-argument block (no device table: hipGraph-capturable without a host-to-device copy).
// ---------------------------------------------------------------------------------------------
struct ReduceMultiArgs {
  cvh_reduce_desc d[CVH_REDUCE_MAX];
  int first_block[CVH_REDUCE_MAX + 1];
  int n;
};
// row lanes per output: big tensors have parallelism to spare (one thread per output, 1 KB coalesced rows); small ones split the rows
__host__ __device__ __forceinline__ int reduce_lanes(int rows, long long n_out) {
  if (n_out >= 32768) return 1;
  if (n_out >= 4096) return 4;
  return rows > 128 ? 64 : 16;
}
__global__ __launch_bounds__(256) void reduce_multi_kernel(ReduceMultiArgs a) {
  __shared__ double red[256];
  int lo = 0, hi = a.n - 1;
  while (lo < hi) {  // last descriptor whose first block <= blockIdx.x
    const int mid = (lo + hi + 1) >> 1;
    if (a.first_block[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const cvh_reduce_desc& d = a.d[lo];
  const int RL = reduce_lanes(d.rows, d.n_out), OL = 256 / RL;
  const int e = threadIdx.x % OL, rl = threadIdx.x / OL;
  const long long o = (long long)((int)blockIdx.x - a.first_block[lo]) * OL + e;
  const float* __restrict__ part = d.part;
  double s = 0.0;
  if (o < d.n_out) {
    int r = rl;
    for (; r + 7 * RL < d.rows; r += 8 * RL) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = part[(size_t)(r + j * RL) * d.row_stride + o];
      s += (double)(((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7])));
    }
    float t = 0.f;
    for (; r < d.rows; r += RL) t += part[(size_t)r * d.row_stride + o];
    s += (double)t;
  }
  red[rl * OL + e] = s;
  __syncthreads();
  for (int h = RL / 2; h > 0; h >>= 1) {  // fixed-order tree: deterministic
    if (rl < h) red[rl * OL + e] += red[(rl + h) * OL + e];
    __syncthreads();
  }
  if (rl == 0 && o < d.n_out) {
    const float v = (float)(red[e] * (double)d.scale);
    float* dst;
    if (d.kind == 1) {  // conv-weight partials [N][KH*KW*Cin] -> torch [N][Cin_real][KH*KW]
      const int k = (int)(o % d.Ktot), n = (int)(o / d.Ktot);
      const int tap = k / d.Cin, c = k - tap * d.Cin;
      if (c >= d.Cin_real) return;
      dst = d.out + ((size_t)n * d.Cin_real + c) * d.khw + tap;
    } else {
      dst = d.out + o;
    }
    *dst = d.accumulate ? *dst + v : v;
  }
}

// BatchNorm forward finalize: part[R][2][C] (sum, sumsq) -> mean, invstd, scale, shift; running-stat update.
__global__ __launch_bounds__(1024) void bn_finalize_kernel(const float* __restrict__ part, int R, int C, double count, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ running_mean,
                                                           float* __restrict__ running_var, float momentum, float eps, float* __restrict__ mean,
                                                           float* __restrict__ invstd, float* __restrict__ scale, float* __restrict__ shift) {
  __shared__ double red[64][2][16];
  const int c = blockIdx.x * 16 + (threadIdx.x & 15);
  const int rl = threadIdx.x >> 4;
  double a = 0.0, b = 0.0;
  if (c < C) strided_partial_sum2(part, R, (size_t)2 * C, (size_t)c, (size_t)C + c, rl, a, b);
  red[rl][0][threadIdx.x & 15] = a;
  red[rl][1][threadIdx.x & 15] = b;
  __syncthreads();
  for (int o = 32; o > 0; o >>= 1) {  // tree over the 64 row lanes (fixed order: deterministic)
    if (rl < o) {
      red[rl][0][threadIdx.x & 15] += red[rl + o][0][threadIdx.x & 15];
      red[rl][1][threadIdx.x & 15] += red[rl + o][1][threadIdx.x & 15];
    }
    __syncthreads();
  }
  if (rl == 0 && c < C) {
    const double s1 = red[0][0][threadIdx.x & 15], s2 = red[0][1][threadIdx.x & 15];
    double mu = s1 / count;
    double var = s2 / count - mu * mu;
    if (var < 0.0) var = 0.0;
    float is = (float)(1.0 / sqrt(var + (double)eps));
    mean[c] = (float)mu;
    invstd[c] = is;
    float g = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
    scale[c] = g * is;
    shift[c] = be - (float)mu * g * is;
    if (running_mean) {
      double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mu;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
  }
}

// eval-mode BatchNorm: scale/shift from running statistics
__global__ void bn_eval_coeff_kernel(const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ rm,
                                     const float* __restrict__ rv, float eps, int C, float* __restrict__ mean, float* __restrict__ invstd,
                                     float* __restrict__ scale, float* __restrict__ shift) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) {
    float is = 1.0f / sqrtf(rv[c] + eps);
    float g = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
    mean[c] = rm[c];
    invstd[c] = is;
    scale[c] = g * is;
    shift[c] = be - rm[c] * g * is;
  }
}

// y = act(x*scale[c] + shift[c]) (+ residual)
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift, int act,
                                                       const T* __restrict__ residual, T* __restrict__ y, size_t rows, int C) {
  // thread = fixed 8-channel group (coefficients live in registers), row lanes walk the rows
  const int cgs = C / 8;
  const int RL = 256 / cgs;
  const int ci = threadIdx.x % cgs, rl = threadIdx.x / cgs;
  if (rl >= RL) return;
  float sc[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { sc[j] = scale[ci * 8 + j]; sh[j] = shift[ci * 8 + j]; }
  const size_t step = (size_t)gridDim.x * RL;
  for (size_t r = (size_t)blockIdx.x * RL + rl; r < rows; r += 2 * step) {
    const size_t r2 = r + step;
    const bool two = r2 < rows;
    const size_t o1 = r * C + ci * 8, o2 = r2 * C + ci * 8;
    V8<T> v1 = v8_load<T>(x + o1), v2 = two ? v8_load<T>(x + o2) : v8_zero<T>();
    V8<T> q1 = v8_zero<T>(), q2 = v8_zero<T>();
    if (residual) {
      q1 = v8_load<T>(residual + o1);
      if (two) q2 = v8_load<T>(residual + o2);
    }
    float f[8], g[8], rf[8], rg[8];
    v8_unpack(v1, f); v8_unpack(v2, g); v8_unpack(q1, rf); v8_unpack(q2, rg);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      f[j] = act_fwd(f[j] * sc[j] + sh[j], act) + rf[j];
      g[j] = act_fwd(g[j] * sc[j] + sh[j], act) + rg[j];
    }
    V8<T> o;
    v8_pack(f, o);
    v8_store<T>(y + o1, o);
    if (two) {
      v8_pack(g, o);
      v8_store<T>(y + o2, o);
    }
  }
}

// BatchNorm backward finalize: part[R][2][C] = (sum dz, sum dz*xhat)  ->  dgamma, dbeta and the per-channel
// coefficients of  dx = ca*dz + cb*x + cc   (train mode; eval mode: ca = gamma*invstd, cb = cc = 0)
__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(const float* __restrict__ part, int R, int C, double count, const float* __restrict__ gamma,
                                                               const float* __restrict__ mean, const float* __restrict__ invstd, int training, int accumulate,
                                                               float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ ca,
                                                               float* __restrict__ cb, float* __restrict__ cc,
                                                               const float* __restrict__ out_basis_beta = nullptr) {
  __shared__ double red[64][2][16];
  const int c = blockIdx.x * 16 + (threadIdx.x & 15);
  const int rl = threadIdx.x >> 4;
  double a = 0.0, b = 0.0;
  if (c < C) strided_partial_sum2(part, R, (size_t)2 * C, (size_t)c, (size_t)C + c, rl, a, b);
  red[rl][0][threadIdx.x & 15] = a;
  red[rl][1][threadIdx.x & 15] = b;
  __syncthreads();
  for (int o = 32; o > 0; o >>= 1) {  // tree over the 64 row lanes (fixed order: deterministic)
    if (rl < o) {
      red[rl][0][threadIdx.x & 15] += red[rl + o][0][threadIdx.x & 15];
      red[rl][1][threadIdx.x & 15] += red[rl + o][1][threadIdx.x & 15];
    }
    __syncthreads();
  }
  if (rl == 0 && c < C) {
    const double s1 = red[0][0][threadIdx.x & 15];
    double s2 = red[0][1][threadIdx.x & 15];
    if (out_basis_beta) {  // the second sum arrived as sum dz * out with out = gamma * xhat + beta (the BatchNorm's own output, no activation)
      const double g0 = gamma ? (double)gamma[c] : 1.0;
      s2 = g0 != 0.0 ? (s2 - (double)out_basis_beta[c] * s1) / g0 : 0.0;
    }
    if (dbeta) dbeta[c] = accumulate ? dbeta[c] + (float)s1 : (float)s1;
    if (dgamma) dgamma[c] = accumulate ? dgamma[c] + (float)s2 : (float)s2;
    const double g = gamma ? (double)gamma[c] : 1.0, is = (double)invstd[c], mu = (double)mean[c];
    if (training) {
      ca[c] = (float)(g * is);
      cb[c] = (float)(-g * is * is * s2 / count);
      cc[c] = (float)(-g * is * s1 / count + g * is * is * mu * s2 / count);
    } else {
      ca[c] = (float)(g * is);
      cb[c] = 0.f;
      cc[c] = 0.f;
    }
  }
}

// dx = ca[c]*dz + cb[c]*x + cc[c],  dz = dout * act'(x*scale + shift)
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ dout, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, int act, const float* __restrict__ ca,
                                                           const float* __restrict__ cb, const float* __restrict__ cc, T* __restrict__ dx,
                                                           size_t rows, int C) {
  const int cgs = C / 8;
  const int RL = 256 / cgs;
  const int ci = threadIdx.x % cgs, rl = threadIdx.x / cgs;
  if (rl >= RL) return;
  float sc[8], sh[8], a[8], b[8], c_[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = ci * 8 + j;
    sc[j] = scale[c]; sh[j] = shift[c]; a[j] = ca[c]; b[j] = cb[c]; c_[j] = cc[c];
  }
  const size_t step = (size_t)gridDim.x * RL;
  for (size_t r = (size_t)blockIdx.x * RL + rl; r < rows; r += 2 * step) {
    const size_t r2 = r + step;
    const bool two = r2 < rows;
    const size_t o1 = r * C + ci * 8, o2 = r2 * C + ci * 8;
    V8<T> x1 = v8_load<T>(x + o1), d1 = v8_load<T>(dout + o1);
    V8<T> x2 = two ? v8_load<T>(x + o2) : v8_zero<T>(), d2 = two ? v8_load<T>(dout + o2) : v8_zero<T>();
    float xf[8], df[8], xg[8], dg[8], o[8], p[8];
    v8_unpack(x1, xf); v8_unpack(d1, df); v8_unpack(x2, xg); v8_unpack(d2, dg);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      o[j] = a[j] * (df[j] * act_grad(xf[j] * sc[j] + sh[j], act)) + b[j] * xf[j] + c_[j];
      p[j] = a[j] * (dg[j] * act_grad(xg[j] * sc[j] + sh[j], act)) + b[j] * xg[j] + c_[j];
    }
    V8<T> ov;
    v8_pack(o, ov);
    v8_store<T>(dx + o1, ov);
    if (two) {
      v8_pack(p, ov);
      v8_store<T>(dx + o2, ov);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// global mean pooling over HW:  x [B][HW][C] -> y [B][C] ; backward broadcasts dy / HW
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void pool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int HW, int C) {
  const int cgs = C / 8;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * cgs) return;
  const int b = idx / cgs, cg = idx - b * cgs;
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < HW; ++i) {
    float f[8];
    v8_unpack(v8_load<T>(x + ((size_t)b * HW + i) * C + cg * 8), f);
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] += f[j];
  }
  const float inv = 1.0f / (float)HW;
#pragma unroll
  for (int j = 0; j < 8; ++j) s[j] *= inv;
  V8<T> o;
  v8_pack(s, o);
  v8_store<T>(y + (size_t)b * C + cg * 8, o);
}
// nn.AdaptiveAvgPool2d(OS) on an NHWC map (PSPNet pyramid bins, ASPP image pooling; OS = 1 is the global pool): bin (i, j) averages rows
// [floor(i H / OS), ceil((i+1) H / OS)) x the same in W (torch's windows: they overlap when H % OS != 0).  Workgroup = one (sample, bin) x 8
// channel chunks x 32 pixel lanes; the pixel lanes stride over the window and are summed through LDS.
template <typename T>
__global__ __launch_bounds__(256) void adaptive_pool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int H, int W, int C, int OS) {
  __shared__ float red[32][8][9];
  const int bin = blockIdx.x % (OS * OS), b = blockIdx.x / (OS * OS);
  const int oi = bin / OS, oj = bin - oi * OS;
  const int h0 = (oi * H) / OS, h1 = ((oi + 1) * H + OS - 1) / OS, w0 = (oj * W) / OS, w1 = ((oj + 1) * W + OS - 1) / OS;
  const int ww = w1 - w0, n = (h1 - h0) * ww;
  const int cl = threadIdx.x & 7, pl = threadIdx.x >> 3, cg = blockIdx.y * 8 + cl;
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (cg * 8 < C) {
    for (int i = pl; i < n; i += 32) {
      const int h = h0 + i / ww, w = w0 + i % ww;
      float f[8];
      v8_unpack(v8_load<T>(x + (((size_t)b * H + h) * W + w) * C + cg * 8), f);
#pragma unroll
      for (int j = 0; j < 8; ++j) s[j] += f[j];
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[pl][cl][j] = s[j];
  __syncthreads();
  if (pl == 0 && cg * 8 < C) {
    float t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int q = 0; q < 32; ++q)
#pragma unroll
      for (int j = 0; j < 8; ++j) t[j] += red[q][cl][j];
    const float inv = 1.0f / (float)n;
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] *= inv;
    V8<T> o;
    v8_pack(t, o);
    v8_store<T>(y + ((size_t)b * OS * OS + bin) * C + cg * 8, o);
  }
}
template <typename T>
__global__ void adaptive_pool_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int B, int H, int W, int C, int OS) {
  const int cgs = C / 8;
  const size_t total = (size_t)B * H * W * cgs;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int cg = (int)(idx % cgs);
    size_t r = idx / cgs;
    const int w = (int)(r % W);
    r /= W;
    const int h = (int)(r % H);
    const size_t b = r / H;
    float g[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int oi = 0; oi < OS; ++oi) {
      const int h0 = (oi * H) / OS, h1 = ((oi + 1) * H + OS - 1) / OS;
      if (h < h0 || h >= h1) continue;
      for (int oj = 0; oj < OS; ++oj) {
        const int w0 = (oj * W) / OS, w1 = ((oj + 1) * W + OS - 1) / OS;
        if (w < w0 || w >= w1) continue;
        float f[8];
        v8_unpack(v8_load<T>(dy + ((b * OS + oi) * OS + oj) * C + cg * 8), f);
        const float inv = 1.0f / (float)((h1 - h0) * (w1 - w0));
#pragma unroll
        for (int j = 0; j < 8; ++j) g[j] += f[j] * inv;
      }
    }
    V8<T> o;
    v8_pack(g, o);
    v8_store<T>(dx + idx * 8, o);
  }
}

template <typename T>
__global__ void pool_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int B, int HW, int C) {
  const int cgs = C / 8;
  const size_t total = (size_t)B * HW * cgs;
  const float inv = 1.0f / (float)HW;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int cg = (int)(idx % cgs);
    const size_t b = idx / ((size_t)HW * cgs);
    float f[8];
    v8_unpack(v8_load<T>(dy + b * C + cg * 8), f);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] *= inv;
    V8<T> o;
    v8_pack(f, o);
    v8_store<T>(dx + idx * 8, o);
  }
}

// ---------------------------------------------------------------------------------------------
// dropout (standalone form) : y = x * mask/(1-p)  — same counter-based mask as the GEMM epilogue
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void dropout_kernel(const T* __restrict__ x, T* __restrict__ y, size_t n, float p, const unsigned long long* __restrict__ seedp,
                               unsigned int stream_id) {
  const unsigned long long seed = *seedp;
  const float inv_keep = 1.0f / (1.0f - p);
  const DropKey k = drop_key(seed, stream_id, p);
  const size_t n8 = n / 8;  // 16 B per lane per step; the (rare) tail below goes element by element
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    float f[8];
    v8_unpack(v8_load<T>(x + i * 8), f);
    dropout_scale8(k, i * 8, inv_keep, f);
    V8<T> o;
    v8_pack(f, o);
    v8_store<T>(y + i * 8, o);
  }
  for (size_t i = n8 * 8 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    y[i] = from_f<T>(to_f<T>(x[i]) * dropout_scale(seed, stream_id, i, p, inv_keep));
}

// Dropout2d (cvnets/layers/dropout.py:32-50 -> nn.Dropout2d): whole channels of a sample are dropped; keep(b, c) comes from the same
// counter-based generator (element index = b*C + c), so the backward pass (the same kernel on dY) regenerates the mask.  NHWC, C % 8 == 0.
template <typename T>
__global__ void dropout2d_kernel(const T* __restrict__ x, T* __restrict__ y, size_t n8, int HW, int C, float p,
                                 const unsigned long long* __restrict__ seedp, unsigned int stream_id) {
  const unsigned long long seed = *seedp;
  const float inv_keep = 1.0f / (1.0f - p);
  const int c8n = C / 8;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    const int c8 = (int)(i % c8n);
    const size_t b = i / ((size_t)c8n * HW);
    float f[8];
    v8_unpack(v8_load<T>(x + i * 8), f);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] *= dropout_scale(seed, stream_id, b * C + c8 * 8 + j, p, inv_keep);
    V8<T> o;
    v8_pack(f, o);
    v8_store<T>(y + i * 8, o);
  }
}

// channel concat of up to 8 NHWC tensors (torch.cat(dim=1) of the ASPP branches, cvnets/modules/aspp_block.py:118-121) and its inverse
// (the backward pass: the gradient of the concat is split back into the branches)
struct CatParams {
  void* ptr[8];
  int c8_end[8];  // exclusive prefix sums of C_i / 8
  int n;
};
template <typename T, bool SPLIT>
__global__ void cat_channels_kernel(CatParams cp, T* __restrict__ whole, size_t rows, int c8_total) {
  const size_t n8 = rows * c8_total;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    const int c8 = (int)(i % c8_total);
    const size_t r = i / c8_total;
    int s = 0;
#pragma unroll
    for (int k = 0; k < 7; ++k) s += (k < cp.n - 1 && c8 >= cp.c8_end[k]) ? 1 : 0;
    const int c8_begin = s ? cp.c8_end[s - 1] : 0;
    const int cs8 = cp.c8_end[s] - c8_begin;
    T* part = reinterpret_cast<T*>(cp.ptr[s]) + (r * cs8 + (c8 - c8_begin)) * 8;
    if (SPLIT) v8_store<T>(part, v8_load<T>(whole + i * 8));
    else v8_store<T>(whole + i * 8, v8_load<T>(part));
  }
}

// elementwise a + b (residual adds outside GEMM epilogues)
template <typename T>
__global__ void add_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ y, size_t n8) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    float fa[8], fb[8];
    v8_unpack(v8_load<T>(a + i * 8), fa);
    v8_unpack(v8_load<T>(b + i * 8), fb);
#pragma unroll
    for (int j = 0; j < 8; ++j) fa[j] += fb[j];
    V8<T> o;
    v8_pack(fa, o);
    v8_store<T>(y + i * 8, o);
  }
}


// ---------------------------------------------------------------------------------------------
// bilinear resize, align_corners=False (F.interpolate in MobileViTBlock.unfolding/folding when the map is not a
// multiple of the patch: cvnets/modules/mobilevit_block.py:191-200, 260-266).  Same arithmetic as ATen's
// upsample_bilinear2d: src = scale*(dst+0.5)-0.5 clamped at 0, i1 = min(i0+1, in-1).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float resize_scale(int in_size, int out_size, int align) {
  if (align) return out_size > 1 ? (float)(in_size - 1) / (float)(out_size - 1) : 0.f;
  return (float)in_size / (float)out_size;
}
// align_corners: src = dst * (in-1)/(out-1)  (MobileViTBlockv2.resize_input_if_needed, mobilevit_block.py:595-603).
__device__ __forceinline__ void bilin_src(int o, float scale, int in_size, int align, int& i0, int& i1, float& l0, float& l1) {
  float s = align ? scale * (float)o : scale * ((float)o + 0.5f) - 0.5f;
  if (s < 0.f) s = 0.f;
  i0 = (int)s;
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + ((i0 < in_size - 1) ? 1 : 0);
  l1 = s - (float)i0;
  l0 = 1.0f - l1;
}

template <typename T>
__global__ void resize_bilinear_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int Ho, int Wo, int C, int align) {
  const int cgs = C / 8;
  const size_t total = (size_t)B * Ho * Wo * cgs;
  const float sh = resize_scale(H, Ho, align), sw = resize_scale(W, Wo, align);
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int cg = (int)(idx % cgs);
    size_t t = idx / cgs;
    const int ox = (int)(t % Wo);
    t /= Wo;
    const int oy = (int)(t % Ho);
    const size_t b = t / Ho;
    int y0, y1, x0, x1;
    float ly0, ly1, lx0, lx1;
    bilin_src(oy, sh, H, align, y0, y1, ly0, ly1);
    bilin_src(ox, sw, W, align, x0, x1, lx0, lx1);
    float p00[8], p01[8], p10[8], p11[8], o[8];
    v8_unpack(v8_load<T>(x + ((b * H + y0) * W + x0) * C + cg * 8), p00);
    v8_unpack(v8_load<T>(x + ((b * H + y0) * W + x1) * C + cg * 8), p01);
    v8_unpack(v8_load<T>(x + ((b * H + y1) * W + x0) * C + cg * 8), p10);
    v8_unpack(v8_load<T>(x + ((b * H + y1) * W + x1) * C + cg * 8), p11);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = ly0 * (lx0 * p00[j] + lx1 * p01[j]) + ly1 * (lx0 * p10[j] + lx1 * p11[j]);
    V8<T> ov;
    v8_pack(o, ov);
    v8_store<T>(y + idx * 8, ov);
  }
}

// gather-form backward: every input pixel sums the output pixels that sampled it (deterministic, no atomics)
template <typename T>
__global__ void resize_bilinear_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int B, int H, int W, int Ho, int Wo, int C, int align) {
  const int cgs = C / 8;
  const size_t total = (size_t)B * H * W * cgs;
  const float sh = resize_scale(H, Ho, align), sw = resize_scale(W, Wo, align);
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int cg = (int)(idx % cgs);
    size_t t = idx / cgs;
    const int ix = (int)(t % W);
    t /= W;
    const int iy = (int)(t % H);
    const size_t b = t / H;
    int oy_lo, oy_hi, ox_lo, ox_hi;  // conservative window of output pixels whose two taps can include (iy, ix)
    if (align) {
      oy_lo = sh > 0.f ? (int)floorf(((float)iy - 1.f) / sh) - 1 : 0; oy_hi = sh > 0.f ? (int)ceilf(((float)iy + 1.f) / sh) + 1 : Ho - 1;
      ox_lo = sw > 0.f ? (int)floorf(((float)ix - 1.f) / sw) - 1 : 0; ox_hi = sw > 0.f ? (int)ceilf(((float)ix + 1.f) / sw) + 1 : Wo - 1;
    } else {
      oy_lo = (int)floorf(((float)iy - 0.5f) / sh - 0.5f) - 1; oy_hi = (int)ceilf(((float)iy + 1.5f) / sh - 0.5f) + 1;
      ox_lo = (int)floorf(((float)ix - 0.5f) / sw - 0.5f) - 1; ox_hi = (int)ceilf(((float)ix + 1.5f) / sw - 0.5f) + 1;
    }
    oy_lo = max(oy_lo, 0); oy_hi = min(oy_hi, Ho - 1);
    ox_lo = max(ox_lo, 0); ox_hi = min(ox_hi, Wo - 1);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
      int y0, y1;
      float ly0, ly1;
      bilin_src(oy, sh, H, align, y0, y1, ly0, ly1);
      const float wy = (y0 == iy ? ly0 : 0.f) + (y1 == iy ? ly1 : 0.f);
      if (wy == 0.f) continue;
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        int x0, x1;
        float lx0, lx1;
        bilin_src(ox, sw, W, align, x0, x1, lx0, lx1);
        const float wx = (x0 == ix ? lx0 : 0.f) + (x1 == ix ? lx1 : 0.f);
        if (wx == 0.f) continue;
        float d[8];
        v8_unpack(v8_load<T>(dy + ((b * Ho + oy) * Wo + ox) * C + cg * 8), d);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += wy * wx * d[j];
      }
    }
    V8<T> ov;
    v8_pack(acc, ov);
    v8_store<T>(dx + idx * 8, ov);
  }
}

__global__ void seed_advance_kernel(unsigned long long* seed) { *seed = *seed * 6364136223846793005ull + 1442695040888963407ull; }

// =============================================================================================
// C ABI
// =============================================================================================
static int g_tune[CVH_TUNE_MAX] = {0, /*TN_PITCH*/ 0, /*TN_WGS*/ 512, /*GEMM_GRID*/ 512, /*DW_XCD*/ 1, /*BIG_GEMM*/ 1, /*COLRED_ROWS*/ 1024, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, /*GEMM_FILL*/ 256, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
int cvh_tune_get(int key) { return (key > 0 && key < CVH_TUNE_MAX) ? g_tune[key] : 0; }
extern "C" int cvh_set_tuning(int key, int value) {
  if (key <= 0 || key >= CVH_TUNE_MAX) return -2;
  g_tune[key] = value;
  return 0;
}

static inline int grid_for(size_t total, int block, int cap = 4096) {
  size_t g = (total + block - 1) / block;
  if (g > (size_t)cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}
#define DISPATCH_T(dtype, ...)                         \
  if ((dtype) == CVH_DT_BF16) { using T = bf16_t; __VA_ARGS__ } \
  else if ((dtype) == CVH_DT_F32) { using T = float; __VA_ARGS__ } \
  else return -1;

extern "C" int cvh_nchw_to_nhwc(int dtype, const float* in, void* out, int B, int C, int H, int W, int Cp, void* stream) {
  if (Cp % 8 || Cp < C) return -2;
  size_t total = (size_t)B * H * W * (Cp / 8);
  DISPATCH_T(dtype, hipLaunchKernelGGL((nchw_to_nhwc_kernel<T>), dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, in, (T*)out, B, C, H, W, Cp);)
  CVH_CHECK_LAUNCH();
  return 0;
}
extern "C" int cvh_nhwc_to_nchw(int dtype, const void* in, float* out, int B, int C, int H, int W, int Cs, void* stream) {
  size_t total = (size_t)B * C * H * W;
  DISPATCH_T(dtype, hipLaunchKernelGGL((nhwc_to_nchw_kernel<T>), dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)in, out, B, C, H, W, Cs);)
  CVH_CHECK_LAUNCH();
  return 0;
}
extern "C" int cvh_mix_batch(int dtype, const float* in, void* out, int B, int C, int H, int W, int Cp, float lam, int x1, int y1, int x2, int y2,
                             void* stream) {
  if (B <= 0) return 0;
  if (Cp == 0) {  // NCHW float32 output
    size_t total = (size_t)B * C * H * W;
    hipLaunchKernelGGL(mix_nchw_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, in, (float*)out, B, C, H, W, lam, x1, y1, x2, y2);
    CVH_CHECK_LAUNCH();
    return 0;
  }
  if (Cp % 8 || Cp < C) return -2;
  size_t total = (size_t)B * H * W * (Cp / 8);
  DISPATCH_T(dtype, hipLaunchKernelGGL((mix_to_nhwc_kernel<T>), dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, in, (T*)out, B, C, H, W, Cp, lam, x1, y1, x2, y2);)
  CVH_CHECK_LAUNCH();
  return 0;
}
extern "C" int cvh_drop_path(int dtype, const void* x, const void* res, void* y, long long rows, int C, int ph, int pw, int H, int W, float p,
                             const unsigned long long* seed, unsigned int stream_id, void* stream) {
  if (C % 8 || C <= 0 || p < 0.f || p >= 1.f || ph <= 0 || pw <= 0 || H <= 0 || W <= 0) return -2;
  if (rows <= 0) return 0;
  size_t total = (size_t)rows * (C / 8);
  DISPATCH_T(dtype, hipLaunchKernelGGL((drop_path_kernel<T>), dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)x, (const T*)res, (T*)y, (size_t)rows, C, ph, pw, H, W, p, seed, stream_id);)
  CVH_CHECK_LAUNCH();
  return 0;
}
extern "C" int cvh_weight_pack(int dtype, const float* w, void* out, int Cout, int Cin, int KHW, int mode, void* stream) {
  const int Cp = (Cin + 7) / 8 * 8, Np = (Cout + 7) / 8 * 8;
  size_t total = mode == 0 ? (size_t)Cout * KHW * Cp : (mode == 1 ? (size_t)Cin * KHW * Np : (mode == 3 ? (size_t)KHW * Cp * Np : (size_t)KHW * Cout));
  DISPATCH_T(dtype, hipLaunchKernelGGL((weight_pack_kernel<T>), dim3(grid_for(total, 256, 1024)), dim3(256), 0, (hipStream_t)stream, w, (T*)out, Cout, Cin, KHW, Cp, Np, mode);)
  CVH_CHECK_LAUNCH();
  return 0;
}
extern "C" int cvh_weight_pack_multi(int dtype, const long long* table, int n_entries, long long total, void* out, void* stream) {
  if (n_entries <= 0 || total <= 0) return 0;
  DISPATCH_T(dtype, hipLaunchKernelGGL((weight_pack_multi_kernel<T>), dim3(grid_for((size_t)total / 8, 256, 2048)), dim3(256), 0, (hipStream_t)stream, table, n_entries, total, (T*)out);)
  CVH_CHECK_LAUNCH();
  return 0;
}
extern "C" int cvh_cast_from_f32(int dtype, const float* in, void* out, long long n, void* stream) {
  DISPATCH_T(dtype, hipLaunchKernelGGL((cast_from_f32_kernel<T>), dim3(grid_for((size_t)n, 256)), dim3(256), 0, (hipStream_t)stream, in, (T*)out, (size_t)n);)
  CVH_CHECK_LAUNCH();
  return 0;
}
extern "C" int cvh_cast_to_f32(int dtype, const void* in, float* out, long long n, void* stream) {
  DISPATCH_T(dtype, hipLaunchKernelGGL((cast_to_f32_kernel<T>), dim3(grid_for((size_t)n, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)in, out, (size_t)n);)
  CVH_CHECK_LAUNCH();
  return 0;
}

// widest column block (multiple of 8, divides C, <= 2048) the 256-thread column kernels handle in one launch
static inline int col_block(int C) {
  if (C <= 2048) return C;
  for (int cb = 2048; cb >= 8; cb -= 8)
    if (C % cb == 0) return cb;
  return -1;
}
extern "C" int cvh_colreduce_rows(long long rows, int C_full) {
  // number of partial rows (gridDim.x) the column-reduction kernels write
  if (C_full % 8 || C_full <= 0) return -2;
  const int C = col_block(C_full);
  const int RL = 256 / (C / 8);
  long long g = (rows + (long long)RL * 8 - 1) / ((long long)RL * 8);
  const int cap = g_tune[CVH_TUNE_COLRED_ROWS];
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

extern "C" int cvh_bn_stats(int dtype, const void* x, long long rows, int C, float* part, void* stream) {
  if (C > 2048) return -2;
  int g = cvh_colreduce_rows(rows, C);
  if (g < 0) return g;
  DISPATCH_T(dtype, hipLaunchKernelGGL((colreduce_kernel<T, 0>), dim3(g), dim3(256), 0, (hipStream_t)stream, (const T*)x, (const T*)nullptr, nullptr, nullptr, nullptr, nullptr, 0, (size_t)rows, C, part, (size_t)C);)
  CVH_CHECK_LAUNCH();
  return 0;
}
extern "C" int cvh_colsum(int dtype, const void* x, long long rows, int C, float* part, float* out, float scale, int accumulate, void* stream) {
  int g = cvh_colreduce_rows(rows, C);
  if (g < 0) return g;
  const int Cb = col_block(C);  // wide matrices (ViT-B FFN: 3072 columns) go block of columns by block of columns
  for (int c0 = 0; c0 < C; c0 += Cb) {
    float* pb = part + (size_t)(c0 / Cb) * g * 2 * Cb;
    DISPATCH_T(dtype, hipLaunchKernelGGL((colreduce_kernel<T, 2>), dim3(g), dim3(256), 0, (hipStream_t)stream, (const T*)x + c0, (const T*)nullptr, nullptr, nullptr, nullptr, nullptr, 0, (size_t)rows, Cb, pb, (size_t)C);)
    // the plain sums live in the first Cb entries of each 2Cb-wide partial row (out == NULL: the caller reduces the rows itself, e.g.
    // through cvh_reduce_multi)
    if (out != nullptr)
      hipLaunchKernelGGL(sum_partials_kernel, dim3((Cb + 15) / 16), dim3(1024), 0, (hipStream_t)stream, pb, g, 2 * Cb, Cb, out + c0, scale, accumulate);
  }
  CVH_CHECK_LAUNCH();
  return 0;
}
extern "C" int cvh_reduce_multi(const cvh_reduce_desc* descs, int n, void* stream) {
  for (int base = 0; base < n; base += CVH_REDUCE_MAX) {
    ReduceMultiArgs a;
    a.n = n - base < CVH_REDUCE_MAX ? n - base : CVH_REDUCE_MAX;
    int blocks = 0;
    for (int i = 0; i < a.n; ++i) {
      a.d[i] = descs[base + i];
      if (a.d[i].rows <= 0 || a.d[i].n_out <= 0 || a.d[i].part == nullptr || a.d[i].out == nullptr) return -2;
      a.first_block[i] = blocks;
      const int OL = 256 / reduce_lanes(a.d[i].rows, a.d[i].n_out);
      blocks += (int)((a.d[i].n_out + OL - 1) / OL);
    }
    a.first_block[a.n] = blocks;
    for (int i = a.n; i < CVH_REDUCE_MAX; ++i) { a.d[i] = a.d[0]; a.first_block[i + 1] = blocks; }
    hipLaunchKernelGGL(reduce_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    CVH_CHECK_LAUNCH();
  }
  return 0;
}
extern "C" int cvh_sum_partials(const float* part, int R, int stride, int Wd, float* out, float scale, int accumulate, void* stream) {
  hipLaunchKernelGGL(sum_partials_kernel, dim3((Wd + 15) / 16), dim3(1024), 0, (hipStream_t)stream, part, R, stride, Wd, out, scale, accumulate);
  CVH_CHECK_LAUNCH();
  return 0;
}
extern "C" int cvh_bn_finalize(const float* part, int R, int C, double count, const float* gamma, const float* beta, float* running_mean,
                               float* running_var, float momentum, float eps, float* mean, float* invstd, float* scale, float* shift,
                               void* stream) {
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 15) / 16), dim3(1024), 0, (hipStream_t)stream, part, R, C, count, gamma, beta, running_mean,
                     running_var, momentum, eps, mean, invstd, scale, shift);
  CVH_CHECK_LAUNCH();
  return 0;
}
extern "C" int cvh_bn_eval_coeff(const float* gamma, const float* beta, const float* rm, const float* rv, float eps, int C, float* mean,
                                 float* invstd, float* scale, float* shift, void* stream) {
  hipLaunchKernelGGL(bn_eval_coeff_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, gamma, beta, rm, rv, eps, C, mean, invstd, scale, shift);
  CVH_CHECK_LAUNCH();
  return 0;
}
extern "C" int cvh_bn_apply(int dtype, const void* x, const float* scale, const float* shift, int act, const void* residual, void* y,
                            long long rows, int C, void* stream) {
  if (C % 8 || C > 2048 || C <= 0) return -2;
  size_t total = ((size_t)rows + (size_t)(256 / (C / 8)) * 2 - 1) / ((size_t)(256 / (C / 8)) * 2);  // blocks: RL rows x 2 per pass
  DISPATCH_T(dtype, hipLaunchKernelGGL((bn_apply_kernel<T>), dim3(grid_for(total, 1)), dim3(256), 0, (hipStream_t)stream, (const T*)x, scale, shift, act, (const T*)residual, (T*)y, (size_t)rows, C);)
  CVH_CHECK_LAUNCH();
  return 0;
}
extern "C" int cvh_bn_bwd_reduce(int dtype, const void* x, const void* dout, const float* scale, const float* shift, const float* mean,
                                 const float* invstd, int act, long long rows, int C, float* part, void* stream) {
  if (C > 2048) return -2;
  int g = cvh_colreduce_rows(rows, C);
  if (g < 0) return g;
  DISPATCH_T(dtype, hipLaunchKernelGGL((colreduce_kernel<T, 1>), dim3(g), dim3(256), 0, (hipStream_t)stream, (const T*)x, (const T*)dout, scale, shift, mean, invstd, act, (size_t)rows, C, part, (size_t)C);)
  CVH_CHECK_LAUNCH();
  return 0;
}
extern "C" int cvh_bn_bwd_finalize(const float* part, int R, int C, double count, const float* gamma, const float* mean, const float* invstd,
                                   int training, int accumulate, float* dgamma, float* dbeta, float* ca, float* cb, float* cc, void* stream) {
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 15) / 16), dim3(1024), 0, (hipStream_t)stream, part, R, C, count, gamma, mean, invstd, training,
                     accumulate, dgamma, dbeta, ca, cb, cc);
  CVH_CHECK_LAUNCH();
  return 0;
}
/* cvh_bn_bwd_finalize for partial rows (sum dz, sum dz * OUT) in which the second sum was taken against the BatchNorm's own output
 * out = gamma * xhat + beta (as stored) instead of xhat — what the kernel that PRODUCES dz can form when `out` is its input
 * (cvh_ir_exp_bwd_s): sum dz * xhat = (sum dz * out - beta * sum dz) / gamma, applied after the reduction in double. */
extern "C" int cvh_bn_bwd_finalize_out(const float* part, int R, int C, double count, const float* gamma, const float* beta, const float* mean,
                                       const float* invstd, int training, int accumulate, float* dgamma, float* dbeta, float* ca, float* cb,
                                       float* cc, void* stream) {
  if (beta == nullptr || gamma == nullptr) return -2;
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 15) / 16), dim3(1024), 0, (hipStream_t)stream, part, R, C, count, gamma, mean, invstd, training,
                     accumulate, dgamma, dbeta, ca, cb, cc, beta);
  CVH_CHECK_LAUNCH();
  return 0;
}
extern "C" int cvh_bn_bwd_apply(int dtype, const void* x, const void* dout, const float* scale, const float* shift, int act, const float* ca,
                                const float* cb, const float* cc, void* dx, long long rows, int C, void* stream) {
  if (C % 8 || C > 2048 || C <= 0) return -2;
  size_t total = ((size_t)rows + (size_t)(256 / (C / 8)) * 2 - 1) / ((size_t)(256 / (C / 8)) * 2);
  DISPATCH_T(dtype, hipLaunchKernelGGL((bn_bwd_apply_kernel<T>), dim3(grid_for(total, 1)), dim3(256), 0, (hipStream_t)stream, (const T*)x, (const T*)dout, scale, shift, act, ca, cb, cc, (T*)dx, (size_t)rows, C);)
  CVH_CHECK_LAUNCH();
  return 0;
}
extern "C" int cvh_pool_fwd(int dtype, const void* x, void* y, int B, int HW, int C, void* stream) {
  if (C % 8) return -2;
  if (HW >= 64) {  // one workgroup per (sample, 64 channels) with 32 pixel lanes: the thread-per-channel-group loop below took 1.8 ms on 64 x 64 maps
    DISPATCH_T(dtype, hipLaunchKernelGGL((adaptive_pool_fwd_kernel<T>), dim3(B, (C / 8 + 7) / 8), dim3(256), 0, (hipStream_t)stream, (const T*)x, (T*)y, 1, HW, C, 1);)
    CVH_CHECK_LAUNCH();
    return 0;
  }
  int total = B * (C / 8);
  DISPATCH_T(dtype, hipLaunchKernelGGL((pool_fwd_kernel<T>), dim3((total + 63) / 64), dim3(64), 0, (hipStream_t)stream, (const T*)x, (T*)y, B, HW, C);)
  CVH_CHECK_LAUNCH();
  return 0;
}
extern "C" int cvh_adaptive_pool_fwd(int dtype, const void* x, void* y, int B, int H, int W, int C, int OS, void* stream) {
  if ((C % 8) != 0 || OS < 1 || OS > H || OS > W || B <= 0) return -2;
  DISPATCH_T(dtype, hipLaunchKernelGGL((adaptive_pool_fwd_kernel<T>), dim3(B * OS * OS, (C / 8 + 7) / 8), dim3(256), 0, (hipStream_t)stream, (const T*)x, (T*)y, H, W, C, OS);)
  CVH_CHECK_LAUNCH();
  return 0;
}
extern "C" int cvh_adaptive_pool_bwd(int dtype, const void* dy, void* dx, int B, int H, int W, int C, int OS, void* stream) {
  if ((C % 8) != 0 || OS < 1 || OS > H || OS > W || B <= 0) return -2;
  const size_t total = (size_t)B * H * W * (C / 8);
  DISPATCH_T(dtype, hipLaunchKernelGGL((adaptive_pool_bwd_kernel<T>), dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)dy, (T*)dx, B, H, W, C, OS);)
  CVH_CHECK_LAUNCH();
  return 0;
}
extern "C" int cvh_pool_bwd(int dtype, const void* dy, void* dx, int B, int HW, int C, void* stream) {
  if (C % 8) return -2;
  size_t total = (size_t)B * HW * (C / 8);
  DISPATCH_T(dtype, hipLaunchKernelGGL((pool_bwd_kernel<T>), dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)dy, (T*)dx, B, HW, C);)
  CVH_CHECK_LAUNCH();
  return 0;
}
extern "C" int cvh_dropout(int dtype, const void* x, void* y, long long n, float p, const unsigned long long* seed, unsigned int stream_id,
                           void* stream) {
  DISPATCH_T(dtype, hipLaunchKernelGGL((dropout_kernel<T>), dim3(grid_for(((size_t)n + 7) / 8, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)x, (T*)y, (size_t)n, p, seed, stream_id);)
  CVH_CHECK_LAUNCH();
  return 0;
}
extern "C" int cvh_dropout2d(int dtype, const void* x, void* y, int B, int HW, int C, float p, const unsigned long long* seed,
                             unsigned int stream_id, void* stream) {
  if ((C % 8) != 0 || p < 0.f || p >= 1.f || seed == nullptr) return -2;
  const size_t n8 = (size_t)B * HW * (C / 8);
  if (n8 == 0) return 0;
  DISPATCH_T(dtype, hipLaunchKernelGGL((dropout2d_kernel<T>), dim3(grid_for(n8, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)x, (T*)y, n8, HW, C, p, seed, stream_id);)
  CVH_CHECK_LAUNCH();
  return 0;
}
extern "C" int cvh_cat_channels(int dtype, void* const* parts, const int* channels, int n, void* whole, long long rows, int split, void* stream) {
  if (n < 1 || n > 8 || rows < 0) return -2;
  CatParams cp;
  cp.n = n;
  int acc = 0;
  for (int i = 0; i < 8; ++i) {
    if (i < n) {
      if ((channels[i] % 8) != 0 || channels[i] <= 0 || parts[i] == nullptr) return -2;
      acc += channels[i] / 8;
      cp.ptr[i] = parts[i];
    } else {
      cp.ptr[i] = nullptr;
    }
    cp.c8_end[i] = acc;
  }
  const size_t n8 = (size_t)rows * acc;
  if (n8 == 0) return 0;
  if (split) { DISPATCH_T(dtype, hipLaunchKernelGGL((cat_channels_kernel<T, true>), dim3(grid_for(n8, 256)), dim3(256), 0, (hipStream_t)stream, cp, (T*)whole, (size_t)rows, acc);) }
  else { DISPATCH_T(dtype, hipLaunchKernelGGL((cat_channels_kernel<T, false>), dim3(grid_for(n8, 256)), dim3(256), 0, (hipStream_t)stream, cp, (T*)whole, (size_t)rows, acc);) }
  CVH_CHECK_LAUNCH();
  return 0;
}
extern "C" int cvh_add(int dtype, const void* a, const void* b, void* y, long long n, void* stream) {
  if (n % 8) return -2;
  DISPATCH_T(dtype, hipLaunchKernelGGL((add_kernel<T>), dim3(grid_for((size_t)n / 8, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)a, (const T*)b, (T*)y, (size_t)n / 8);)
  CVH_CHECK_LAUNCH();
  return 0;
}
extern "C" int cvh_seed_advance(unsigned long long* seed, void* stream) {
  hipLaunchKernelGGL(seed_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, seed);
  CVH_CHECK_LAUNCH();
  return 0;
}

extern "C" int cvh_resize_bilinear_fwd(int dtype, const void* x, void* y, int B, int H, int W, int Ho, int Wo, int C, int align_corners, void* stream) {
  if (C % 8) return -2;
  size_t total = (size_t)B * Ho * Wo * (C / 8);
  DISPATCH_T(dtype, hipLaunchKernelGGL((resize_bilinear_fwd_kernel<T>), dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)x, (T*)y, B, H, W, Ho, Wo, C, align_corners);)
  CVH_CHECK_LAUNCH();
  return 0;
}
/* dx[B][H][W][C] = adjoint of the (H,W)->(Ho,Wo) resize applied to dy[B][Ho][Wo][C] */
extern "C" int cvh_resize_bilinear_bwd(int dtype, const void* dy, void* dx, int B, int H, int W, int Ho, int Wo, int C, int align_corners, void* stream) {
  if (C % 8) return -2;
  size_t total = (size_t)B * H * W * (C / 8);
  DISPATCH_T(dtype, hipLaunchKernelGGL((resize_bilinear_bwd_kernel<T>), dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)dy, (T*)dx, B, H, W, Ho, Wo, C, align_corners);)
  CVH_CHECK_LAUNCH();
  return 0;
}
