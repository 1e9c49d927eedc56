// The stem: 3x3 stride-2 pad-1 convolution of the 3-channel NCHW image batch into 16 (or 32) NHWC channels, forward and weight gradient
// (MobileViT conv_1: cvnets/models/classification/mobilevit.py:62-72, a ConvLayer2d whose forward is cvnets/layers/conv_layer.py:254-255).
//
// The generic path converts the image to NHWC with channels padded 3 -> 8 (one pass: read 0.8 MB, write 1.0 MB per 256x256 image) and then
// runs an implicit GEMM with K = 72 over it (another 1.0 MB read) — 2.9 MB moved to produce 0.5 MB, and the same again for dW.  Here the
// image planes are read ONCE, straight from NCHW (fp32 or bf16), into an LDS tile laid out [row][col][4 channels] (channel 3 = 0), so that
// for an output pixel the 3 taps of one kernel row plus one spare column are 16 contiguous values.  With the contraction ordered
// k = (kh, column slot 0..3, channel 0..3) — 48 values, slot 3 / channel 3 carry zero weights — an MFMA operand fragment (8 consecutive
// k of one pixel) is ONE aligned 16-byte LDS read and the convolution of 16 pixels x 16 channels is two v_mfma_f32_16x16x32_bf16 on the
// transposed problem (D^T = W P^T): a lane ends up with 4 consecutive channels of one pixel, i.e. an 8-byte NHWC store.
//
// dW contracts over pixels instead: per kernel row kh, D[n][(slot, c)] += sum_p dY[p][n] * P_kh[p][(slot, c)], 32 pixels per MFMA.  Both
// operands are "8 consecutive pixels of one column" — gathered with the gfx950 LDS transpose read (ds_read_b64_tr_b16), every lane
// supplying the address of its own pixel row, from the dY tile [pixel][n] and from the same x tile as above.
//
// Both kernels are HBM streams: forward 0.8 (fp32 image) + 0.5 MB per image, dW the same; algorithmic bytes = image + output map.
#include <hip/hip_runtime.h>

#include "common.hpp"
#include "cvnets_hip.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef short tr_v4s __attribute__((ext_vector_type(4)));

constexpr int ST_OH = 8, ST_OW = 64;           // output tile
constexpr int ST_IH = 2 * ST_OH + 1;           // 17 input rows
constexpr int ST_IW = 132;                     // LDS columns j = 0..131 <-> iw = 2*ow0 - 1 + j (needed: 0..129)
constexpr int ST_NCHUNK = 34;                  // 4-column global chunks per row: iw = 2*ow0 - 4 + 4m
constexpr int ST_XT = ST_IH * ST_IW * 4;       // elements of the x tile

struct StemParams {
  const void* x;      // [B][3][H][W] fp32 or bf16
  const void* w;      // [NOUT][3][3][3] fp32 or bf16 (w_f32)
  void* y;            // fwd: bf16 [B][Ho][Wo][NOUT]
  const void* dy;     // dW : bf16 [B][Ho][Wo][NOUT]
  float* stats_part;  // fwd: [grid][2][NOUT] or nullptr
  float* part;        // dW : [grid][NOUT][9][8]
  int B, H, W, Ho, Wo, tiles_h, tiles_w, ntiles, w_f32;
};

template <typename TI> __device__ __forceinline__ void ld4(const TI* p, float* v);
template <> __device__ __forceinline__ void ld4<float>(const float* p, float* v) {
  const float4 t = *reinterpret_cast<const float4*>(p);
  v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
template <> __device__ __forceinline__ void ld4<bf16_t>(const bf16_t* p, float* v) {
  const uint2 t = *reinterpret_cast<const uint2*>(p);
  v[0] = bf2f((uint16_t)(t.x & 0xffff)); v[1] = bf2f((uint16_t)(t.x >> 16));
  v[2] = bf2f((uint16_t)(t.y & 0xffff)); v[3] = bf2f((uint16_t)(t.y >> 16));
}

// x tile of output tile (b, oh0, ow0): xt[(r * ST_IW + j) * 4 + c] = x[b][c][2*oh0 - 1 + r][2*ow0 - 1 + j], zero outside the image.
// Channel slot 3 is zeroed once by the caller and never written.
template <typename TI>
__device__ __forceinline__ void stage_x(bf16_t* xt, const TI* __restrict__ x, int b, int oh0, int ow0, int H, int W, int tid) {
  const int ih0 = 2 * oh0 - 1, iwc = 2 * ow0 - 4;
  for (int idx = tid; idx < ST_IH * 3 * ST_NCHUNK; idx += 256) {
    const int m = idx % ST_NCHUNK, rc = idx / ST_NCHUNK;
    const int c = rc % 3, r = rc / 3;
    const int ih = ih0 + r, iw = iwc + 4 * m;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (ih >= 0 && ih < H && iw >= 0 && iw < W) ld4<TI>(x + (((size_t)b * 3 + c) * H + ih) * W + iw, v);  // W % 4 == 0: a chunk is all in or all out
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int j = 4 * m - 3 + e;
      if (j >= 0 && j < ST_IW) xt[(r * ST_IW + j) * 4 + c] = from_f<bf16_t>(v[e]);
    }
  }
}

__device__ __forceinline__ float wval(const StemParams& p, int idx) {
  return p.w_f32 ? reinterpret_cast<const float*>(p.w)[idx] : to_f<bf16_t>(reinterpret_cast<const bf16_t*>(p.w)[idx]);
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------------------------------
template <typename TI, int NB>
__global__ __launch_bounds__(256) void stem_fwd_kernel(StemParams p) {
  constexpr int NOUT = 16 * NB;
  __shared__ __attribute__((aligned(16))) bf16_t xt[ST_XT];
  __shared__ float red[4][2 * NOUT];  // per-wave column sums, added in wave order: the statistics are bit-reproducible run to run
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, l15 = lane & 15;
  for (int i = tid; i < ST_XT / 8; i += 256) reinterpret_cast<uint4*>(xt)[i] = make_uint4(0, 0, 0, 0);
  // weight fragments: lane (channel l15, k group g); k = 8g + j <-> (kh = g >> 1, slot = 2*(g & 1) + (j >> 2), c = j & 3); second k-step: kh = 2
  bf16x8_t wa[NB][2];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      uint32_t u[4];
#pragma unroll
      for (int jp = 0; jp < 4; ++jp) {
        float f[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int j = 2 * jp + h;
          const int kh = ks == 0 ? (g >> 1) : 2, slot = 2 * (g & 1) + (j >> 2), c = j & 3;
          const bool live = slot < 3 && c < 3 && (ks == 0 || g < 2);
          f[h] = live ? wval(p, (((nb * 16 + l15) * 3 + c) * 3 + kh) * 3 + slot) : 0.f;
        }
        u[jp] = f2bf_pk(f[0], f[1]);
      }
      wa[nb][ks] = __builtin_bit_cast(bf16x8_t, make_uint4(u[0], u[1], u[2], u[3]));
    }

  const TI* __restrict__ x = reinterpret_cast<const TI*>(p.x);
  bf16_t* __restrict__ y = reinterpret_cast<bf16_t*>(p.y);
  float s1[NB][4], s2[NB][4];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int r = 0; r < 4; ++r) s1[nb][r] = s2[nb][r] = 0.f;
  const bf16x8_t zero8 = __builtin_bit_cast(bf16x8_t, make_uint4(0, 0, 0, 0));

  for (int t = blockIdx.x; t < p.ntiles; t += gridDim.x) {
    const int tw = t % p.tiles_w, t1 = t / p.tiles_w;
    const int th = t1 % p.tiles_h, b = t1 / p.tiles_h;
    const int oh0 = th * ST_OH, ow0 = tw * ST_OW;
    __syncthreads();  // previous tile consumed (first iteration: zero fill done)
    stage_x<TI>(xt, x, b, oh0, ow0, p.H, p.W, tid);
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int r = 2 * wave + rr, oh = oh0 + r;
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {
        const int owl = 16 * cb + l15, ow = ow0 + owl;
        const bf16_t* base = xt + (2 * r * ST_IW + 2 * owl + 2 * (g & 1)) * 4;
        const bf16x8_t b0 = *reinterpret_cast<const bf16x8_t*>(base + (g >> 1) * ST_IW * 4);
        const bf16x8_t b1 = g < 2 ? *reinterpret_cast<const bf16x8_t*>(base + 2 * ST_IW * 4) : zero8;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[nb][0], b0, acc, 0, 0, 0);  // D^T[n][pixel] += W[n][k] P[pixel][k]
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[nb][1], b1, acc, 0, 0, 0);
          if (oh < p.Ho && ow < p.Wo) {
            const uint2 pk = make_uint2(f2bf_pk(acc[0], acc[1]), f2bf_pk(acc[2], acc[3]));
            *reinterpret_cast<uint2*>(y + (((size_t)b * p.Ho + oh) * p.Wo + ow) * NOUT + nb * 16 + 4 * g) = pk;
            const float q[4] = {bf2f((uint16_t)(pk.x & 0xffff)), bf2f((uint16_t)(pk.x >> 16)), bf2f((uint16_t)(pk.y & 0xffff)),
                                bf2f((uint16_t)(pk.y >> 16))};  // statistics of the stored values
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              s1[nb][e] += q[e];
              s2[nb][e] += q[e] * q[e];
            }
          }
        }
      }
    }
  }

  if (p.stats_part) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float a = s1[nb][e], q = s2[nb][e];
#pragma unroll
        for (int m = 8; m >= 1; m >>= 1) {
          a += __shfl_xor(a, m, 64);
          q += __shfl_xor(q, m, 64);
        }
        if (l15 == 0) {
          red[wave][nb * 16 + 4 * g + e] = a;
          red[wave][NOUT + nb * 16 + 4 * g + e] = q;
        }
      }
    __syncthreads();
    if (tid < 2 * NOUT) p.stats_part[(size_t)blockIdx.x * 2 * NOUT + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// weight gradient
// ---------------------------------------------------------------------------------------------------------------------------------------
// 8 consecutive tile pixels (px0 + 8g .. + 7) of column lane & 15, from a row-major LDS matrix whose pixel rows start at rowaddr(pixel):
// two transpose reads; inside a 16-lane group lane i = 4r + q supplies the address of columns 4q..4q+3 of pixel row r.
template <typename F>
__device__ __forceinline__ bf16x8_t tr_frag(F rowaddr, int px0, int lane) {
  const int i = lane & 15, g = lane >> 4;
  const bf16_t* lo = rowaddr(px0 + 8 * g + (i >> 2)) + 4 * (i & 3);
  const bf16_t* hi = rowaddr(px0 + 8 * g + 4 + (i >> 2)) + 4 * (i & 3);
  const tr_v4s a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tr_v4s*)(lo));
  const tr_v4s b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tr_v4s*)(hi));
  typedef short v8s __attribute__((ext_vector_type(8)));
  return __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}

template <typename TI, int NB>
__global__ __launch_bounds__(256) void stem_dw_kernel(StemParams p) {
  constexpr int NOUT = 16 * NB;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  bf16_t* xt = reinterpret_cast<bf16_t*>(smem_raw);          // x tile
  bf16_t* dyt = xt + ST_XT;                                   // [ST_OH * ST_OW pixels][NOUT]
  float* red = reinterpret_cast<float*>(dyt + ST_OH * ST_OW * NOUT);  // [NOUT][9][8]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, l15 = lane & 15;
  for (int i = tid; i < ST_XT / 8; i += 256) reinterpret_cast<uint4*>(xt)[i] = make_uint4(0, 0, 0, 0);
  for (int i = tid; i < NOUT * 72; i += 256) red[i] = 0.f;

  const TI* __restrict__ x = reinterpret_cast<const TI*>(p.x);
  const bf16_t* __restrict__ dy = reinterpret_cast<const bf16_t*>(p.dy);
  f32x4_t acc[NB][3];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) acc[nb][kh] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  for (int t = blockIdx.x; t < p.ntiles; t += gridDim.x) {
    const int tw = t % p.tiles_w, t1 = t / p.tiles_w;
    const int th = t1 % p.tiles_h, b = t1 / p.tiles_h;
    const int oh0 = th * ST_OH, ow0 = tw * ST_OW;
    __syncthreads();
    stage_x<TI>(xt, x, b, oh0, ow0, p.H, p.W, tid);
    constexpr int CPP = NOUT / 8;  // 16-byte chunks per pixel
    for (int idx = tid; idx < ST_OH * ST_OW * CPP; idx += 256) {
      const int px = idx / CPP, ck = idx % CPP;
      const int oh = oh0 + px / ST_OW, ow = ow0 + px % ST_OW;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (oh < p.Ho && ow < p.Wo) v = *reinterpret_cast<const uint4*>(dy + (((size_t)b * p.Ho + oh) * p.Wo + ow) * NOUT + ck * 8);
      *reinterpret_cast<uint4*>(dyt + px * NOUT + ck * 8) = v;
    }
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int r = 2 * wave + rr;
#pragma unroll
      for (int hb = 0; hb < 2; ++hb) {
        const int px0 = 32 * hb;  // 32 consecutive pixels of tile row r
        bf16x8_t bx[3];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
          bx[kh] = tr_frag([&](int px) { return xt + ((2 * r + kh) * ST_IW + 2 * px) * 4; }, px0, lane);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const bf16x8_t ad = tr_frag([&](int px) { return dyt + (r * ST_OW + px) * NOUT + nb * 16; }, px0, lane);
#pragma unroll
          for (int kh = 0; kh < 3; ++kh)
            acc[nb][kh] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ad, bx[kh], acc[nb][kh], 0, 0, 0);  // D[n][(slot,c)] += dY[p][n] P[p][(slot,c)]
        }
      }
    }
  }

  // acc[nb][kh][e]: n = nb*16 + 4g + e, column l15 = (slot, c)
  // the four waves add their accumulators one after the other (fixed order: the partial row is bit-reproducible)
  const int slot = l15 >> 2, c = l15 & 3;
  for (int w = 0; w < 4; ++w) {
    __syncthreads();
    if (wave == w && slot < 3 && c < 3) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int e = 0; e < 4; ++e) red[((nb * 16 + 4 * g + e) * 9 + kh * 3 + slot) * 8 + c] += acc[nb][kh][e];
    }
  }
  __syncthreads();
  for (int i = tid; i < NOUT * 72; i += 256) p.part[(size_t)blockIdx.x * NOUT * 72 + i] = red[i];
}

template <int NB> size_t stem_dw_smem() { return (size_t)ST_XT * 2 + (size_t)ST_OH * ST_OW * 16 * NB * 2 + (size_t)16 * NB * 72 * 4; }

int stem_plan(StemParams& p, int B, int H, int W, int Cout) {
  if (!(Cout == 16 || Cout == 32) || H < 2 || W < 4 || (W % 4)) return -2;
  p.B = B; p.H = H; p.W = W;
  p.Ho = (H - 1) / 2 + 1; p.Wo = (W - 1) / 2 + 1;  // (H + 2 - 3) / 2 + 1
  p.tiles_h = (p.Ho + ST_OH - 1) / ST_OH;
  p.tiles_w = (p.Wo + ST_OW - 1) / ST_OW;
  const long long nt = (long long)B * p.tiles_h * p.tiles_w;
  if (nt <= 0 || nt > 0x7fffffff) return -2;
  p.ntiles = (int)nt;
  return 0;
}
int stem_grid(int ntiles) { return ntiles < 2048 ? ntiles : 2048; }

}  // namespace

extern "C" int cvh_stem_rows(int B, int H, int W, int Cout) {
  StemParams p;
  if (stem_plan(p, B, H, W, Cout)) return -2;
  return stem_grid(p.ntiles);
}

extern "C" int cvh_stem_conv_fwd(int in_dtype, const void* x_nchw, int w_dtype, const void* weight, void* y_bf16, float* stats_part, int B, int H,
                                 int W, int Cout, void* stream) {
  StemParams p;
  if (stem_plan(p, B, H, W, Cout)) return -2;
  if ((in_dtype != CVH_DT_F32 && in_dtype != CVH_DT_BF16) || (w_dtype != CVH_DT_F32 && w_dtype != CVH_DT_BF16)) return -1;
  p.x = x_nchw; p.w = weight; p.y = y_bf16; p.dy = nullptr; p.stats_part = stats_part; p.part = nullptr; p.w_f32 = w_dtype == CVH_DT_F32;
  const dim3 grid(stem_grid(p.ntiles));
  hipStream_t st = (hipStream_t)stream;
  if (in_dtype == CVH_DT_F32) {
    if (Cout == 16) hipLaunchKernelGGL((stem_fwd_kernel<float, 1>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((stem_fwd_kernel<float, 2>), grid, dim3(256), 0, st, p);
  } else {
    if (Cout == 16) hipLaunchKernelGGL((stem_fwd_kernel<bf16_t, 1>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((stem_fwd_kernel<bf16_t, 2>), grid, dim3(256), 0, st, p);
  }
  CVH_CHECK_LAUNCH();
  return 0;
}

extern "C" int cvh_stem_conv_dw(int in_dtype, const void* x_nchw, const void* dy_bf16, float* part, int B, int H, int W, int Cout, void* stream) {
  StemParams p;
  if (stem_plan(p, B, H, W, Cout)) return -2;
  if (in_dtype != CVH_DT_F32 && in_dtype != CVH_DT_BF16) return -1;
  p.x = x_nchw; p.w = nullptr; p.y = nullptr; p.dy = dy_bf16; p.stats_part = nullptr; p.part = part; p.w_f32 = 0;
  const dim3 grid(stem_grid(p.ntiles));
  hipStream_t st = (hipStream_t)stream;
#define STEM_DW(TI, NB)                                                                                                              \
  do {                                                                                                                               \
    const size_t smem = stem_dw_smem<NB>();                                                                                          \
    static DynSmemAttr attr;                                                                                                         \
    if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(stem_dw_kernel<TI, NB>), smem); e != hipSuccess) return (int)e;    \
    hipLaunchKernelGGL((stem_dw_kernel<TI, NB>), grid, dim3(256), smem, st, p);                                                      \
  } while (0)
  if (in_dtype == CVH_DT_F32) {
    if (Cout == 16) STEM_DW(float, 1); else STEM_DW(float, 2);
  } else {
    if (Cout == 16) STEM_DW(bf16_t, 1); else STEM_DW(bf16_t, 2);
  }
#undef STEM_DW
  CVH_CHECK_LAUNCH();
  return 0;
}
