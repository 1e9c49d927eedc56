// Depthwise 3x3 convolution (pad 1, stride 1 / 2) with BatchNorm links (bnlink.hpp), NHWC, on LDS-staged tiles.
//
// Forward:   y = dwconv(act(scale*x_raw + shift))        the BatchNorm+SiLU of the producer is applied ONCE per element while the
//            tile (with halo) is staged into LDS; the statistics of y go to a forward link.
// Backward:  one pass over (g_out, y_out, x_raw) produces
//              dy   = ca*g_out + cb*y_out + cc                         (BatchNorm-after-the-conv input gradient, formed on load)
//              g_in = dwconv^T(dy) * act'(scale*x_raw + shift)         (stored; sum g_in, sum g_in*xhat -> backward link)
//              dW[c][tap] += sum dy * act(scale*x_raw + shift)(shifted)
//            i.e. bn_bwd_apply + dwconv_bwd_x + dwconv_bwd_w + bn_bwd_reduce of the unfused path in ONE kernel, with every operand
//            read once from HBM (halo re-reads are served by L2: a workgroup walks a contiguous range of tiles).
//
// Workgroup = 256 threads = 8 channel lanes (8 channels = 16 B each: a 64-channel chunk) x 32 pixel lanes.
// Replaces nn.Conv2d(groups=C) + BatchNorm2d + SiLU in InvertedResidual (cvnets/modules/mobilenetv2.py:194-207,231-235) and their
// autograd backward.
#include "common.hpp"
#include "cvnets_hip.h"
#include "bnlink.hpp"

#define DWF_CC 64

template <int S> struct DwfTile;
// stride 1: 8 x 16 tile (output == input grid), a pixel lane owns 4 adjacent pixels of one row
template <> struct DwfTile<1> {
  static constexpr int OH = 8, OW = 16;          // forward output tile / backward owned-output tile
  static constexpr int IH = 10, IW = 18;         // forward input tile (halo 1); backward z tile
  static constexpr int DH = 10, DW = 18;         // backward dy tile
  static constexpr int XH = 8, XW = 16;          // backward input-gradient tile
};
// stride 2: 8 x 8 outputs <-> 16 x 16 inputs
template <> struct DwfTile<2> {
  static constexpr int OH = 8, OW = 8;
  static constexpr int IH = 17, IW = 17;
  static constexpr int DH = 9, DW = 9;
  static constexpr int XH = 16, XW = 16;
};

struct DwfParams {
  const void* x;        // fwd: input (raw producer output or plain activation);  bwd: x_raw
  OperandXf xf;         // fwd: transform of x;  bwd: transform of g_out (dy)
  const void* wp;       // [9][C]
  void* y;              // fwd: output;  bwd: g_in
  const void* g_out;    // bwd
  const float* in_stats;  // bwd: [4][C] of the BatchNorm in front of the conv
  int in_act;
  float* stats_part;    // [rows][2][C] column statistics of the tensor written (nullptr: skip); rows = gridDim.x / chunks
  float* dw_part;       // bwd: [rows][C*9] this workgroup's share of dW
  int B, H, W, Ho, Wo, C;
  int tiles_h, tiles_w;
  int ntiles, chunks;
  int dbg;  // developer knob (CVH_TUNE key 7): skip phases of the backward kernel to time them (results are WRONG when non-zero)
};

// LDS pixel pitch: 128 B of channels + 32 B.  A pixel lane's 4-pixel stride is then 640 B = 128 B mod 256, so the four pixel lanes
// of a ds_read_b128 service group (two channel halves each) land on four disjoint 64-byte bank quarters; with a 16-byte pad two
// of them collided (2-way conflicts on every stencil read).
template <typename T> __host__ __device__ constexpr int dwf_pitch() { return DWF_CC + 32 / (int)sizeof(T); }

// ---------------------------------------------------------------------------------------------------------------------------
// Per-channel vectors (coefficients, weights) live in LDS and are read where they are used: held in registers for the whole kernel
// they cost 100+ VGPRs and push these kernels below two workgroups per CU.
// ---------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void lds_f8(const float* p, float* o) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}
// stage n per-channel vectors (src[v][C], channels c0 .. c0+CC) into LDS as float [n][CC]; nullptr / out-of-range -> 0
__device__ __forceinline__ void stage_vec(float* dst, const float* src, int C, int c0) {
  for (int i = threadIdx.x; i < DWF_CC; i += 256) dst[i] = (src != nullptr && c0 + i < C) ? src[c0 + i] : 0.f;
}
template <typename T>
__device__ __forceinline__ void stage_weights(float* dst /*[9][CC]*/, const T* wp, int C, int c0) {
  for (int i = threadIdx.x; i < 9 * DWF_CC; i += 256) {
    const int t = i / DWF_CC, c = c0 + (i - t * DWF_CC);
    dst[i] = c < C ? to_f<T>(wp[(size_t)t * C + c]) : 0.f;
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------------------
template <typename T, int S>
__global__ __launch_bounds__(256, 2) void dwf_fwd_kernel(DwfParams p) {
  using TL = DwfTile<S>;
  constexpr int PITCH = dwf_pitch<T>();
  constexpr int NPIX = TL::IH * TL::IW;
  constexpr int NLOAD = (NPIX * 8 + 255) / 256;
  constexpr int PPL = TL::OW / 4;               // pixels per lane (4 lanes per tile row): 4 (stride 1) / 2 (stride 2)
  constexpr int NC = (PPL - 1) * S + 3;         // input columns of a lane's sliding window
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* tile = reinterpret_cast<T*>(smem_raw);                       // [NPIX][PITCH]
  float* red = reinterpret_cast<float*>(tile + NPIX * PITCH);     // [2][CC] statistics
  float* wl = red + 2 * DWF_CC;                                   // [9][CC] weights
  float* cst = wl + 9 * DWF_CC;                                   // [2][CC] scale, shift of the input transform

  const int tid = threadIdx.x, cl = tid & 7, pl = tid >> 3;
  // Work items = (spatial tile, 64-channel chunk), chunk fastest, handed out ROUND-ROBIN: item = k * gridDim.x + lb.  At any moment the
  // resident workgroups cover a contiguous run of tiles with all their channel chunks, so DRAM pages are used completely while they are
  // open (a contiguous private tile range per workgroup — 512 x 30 concurrent 2 KB streams — measured 2.0-2.8 TB/s on loads alone);
  // lb is XCD-contiguous, so a tile's chunks and its neighbours share one L2.  gridDim.x % chunks == 0: a workgroup's chunk is fixed.
  const int lb = xcd_chunk_id(blockIdx.x, gridDim.x);
  const int chunk = lb % p.chunks, row_id = lb / p.chunks;
  const int c0 = chunk * DWF_CC, ch = c0 + cl * 8;
  const bool ch_ok = ch < p.C;
  const T* __restrict__ x = reinterpret_cast<const T*>(p.x);
  T* __restrict__ y = reinterpret_cast<T*>(p.y);
  const int mode = p.xf.mode == 1 ? 1 : 0;

  if (tid < 2 * DWF_CC) red[tid] = 0.f;
  stage_weights<T>(wl, reinterpret_cast<const T*>(p.wp), p.C, c0);
  stage_vec(cst, mode ? p.xf.c0 : nullptr, p.C, c0);
  stage_vec(cst + DWF_CC, mode ? p.xf.c1 : nullptr, p.C, c0);
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s1[j] = s2[j] = 0.f;
  const bool want_stats = p.stats_part != nullptr;

  const int t_begin = row_id, t_end = p.ntiles, t_step = gridDim.x / p.chunks;
  const int r = pl >> 2, q = pl & 3;

  for (int tix = t_begin; tix < t_end; tix += t_step) {
    const int tw = tix % p.tiles_w;
    const int t1 = tix / p.tiles_w;
    const int th = t1 % p.tiles_h;
    const int b = t1 / p.tiles_h;
    const int ho0 = th * TL::OH, wo0 = tw * TL::OW;
    const int hi0 = ho0 * S - 1, wi0 = wo0 * S - 1;

    V8<T> rv[NLOAD];
    bool rok[NLOAD];
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
      const int px = (tid + i * 256) >> 3;
      const int pr = px / TL::IW, pc = px - pr * TL::IW;
      const int hi = hi0 + pr, wi = wi0 + pc;
      rok[i] = px < NPIX && hi >= 0 && hi < p.H && wi >= 0 && wi < p.W && ch_ok;
      rv[i] = v8_load_clamped<T>(x, ((size_t)(b * p.H + hi) * p.W + wi) * p.C + ch, rok[i]);
    }
    __syncthreads();  // previous tile fully consumed (first iteration: staged vectors visible)
    {
      Coef8 kx;
      lds_f8(cst + cl * 8, kx.a);
      lds_f8(cst + DWF_CC + cl * 8, kx.b);
#pragma unroll
      for (int i = 0; i < NLOAD; ++i) {
        const int px = (tid + i * 256) >> 3;
        if (px < NPIX) {
          V8<T> v = v8_mask(rv[i], rok[i]);
          if (mode == 1) v = xf_apply<T>(rv[i], rv[i], kx, 1, p.xf.act, rok[i]);  // zero padding stays zero AFTER the transform
          v8_store<T>(tile + px * PITCH + cl * 8, v);
        }
      }
    }
    __syncthreads();

    float acc[PPL][8];
#pragma unroll
    for (int t = 0; t < PPL; ++t)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[t][j] = 0.f;
#pragma unroll 1
    for (int kh = 0; kh < 3; ++kh) {
      float wr[3][8];
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) lds_f8(wl + (kh * 3 + kw) * DWF_CC + cl * 8, wr[kw]);
      const T* rowp = tile + ((r * S + kh) * TL::IW + q * PPL * S) * PITCH + cl * 8;
#pragma unroll
      for (int cix = 0; cix < NC; ++cix) {
        float f[8];
        v8_unpack(v8_load<T>(rowp + cix * PITCH), f);
#pragma unroll
        for (int t = 0; t < PPL; ++t) {
          const int kw = cix - t * S;  // compile-time after unrolling
          if (kw >= 0 && kw < 3) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[t][j] += f[j] * wr[kw][j];
          }
        }
      }
    }
    const int ho = ho0 + r;
#pragma unroll
    for (int t = 0; t < PPL; ++t) {
      const int wo = wo0 + q * PPL + t;
      if (ho < p.Ho && wo < p.Wo && ch_ok) {
        V8<T> o;
        v8_pack(acc[t], o);
        v8_store<T>(y + ((size_t)(b * p.Ho + ho) * p.Wo + wo) * p.C + ch, o);
        float vr[8];
        v8_unpack(o, vr);  // statistics of the values as stored
#pragma unroll
        for (int j = 0; j < 8; ++j) { s1[j] += vr[j]; s2[j] += vr[j] * vr[j]; }
      }
    }
  }
  if (want_stats) {
    __syncthreads();
    lds_ordered_accumulate(pl, 32, true, [&]() {  // the 32 pixel lanes of a channel group, in order
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        red[cl * 8 + j] += s1[j];
        red[DWF_CC + cl * 8 + j] += s2[j];
      }
    });
    for (int i = tid; i < 2 * DWF_CC; i += 256) {
      const int which = i / DWF_CC, c = c0 + (i - which * DWF_CC);
      if (c < p.C) p.stats_part[((size_t)row_id * 2 + which) * p.C + c] = red[i];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// backward (dX + dW + BatchNorm-backward statistics in one pass)
// ---------------------------------------------------------------------------------------------------------------------------
// z = act(y), gp = act'(y) of eight channels from ONE sigmoid per element (SiLU: z = y s, act' = s (1 + y (1 - s)))
__device__ __forceinline__ void act_fwd_grad8(const float* y, int act, float* z, float* gp) {
  if (act == CVH_ACT_SILU) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float s = sigmoidf_(y[j]);
      z[j] = y[j] * s;
      gp[j] = s * (1.0f + y[j] * (1.0f - s));
    }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) { z[j] = act_fwd(y[j], act); gp[j] = act_grad(y[j], act); }
  }
}

// Both products of the backward pass run over the SAME (input pixel p, tap d) pairs:
//     dz[p]  = sum_d dy[p - d] * w[d]          (input gradient)
//     dW[d] += sum_p dy[p - d] * z[p]          (weight gradient, re-indexed from sum_o dy[o] * z[o + d] by p = o + d)
// so every dy value fetched from the LDS tile feeds two FMAs, z = act(bn(x_raw)) is needed only at the lane's OWN pixels — computed in
// registers from the x_raw values the epilogue needs anyway (act' shares its sigmoid) — and the z tile with its halo (a second LDS tile,
// 1.4x the activation work, a second pass of 18 LDS reads + unpacks per lane) does not exist.
template <typename T, int S>
__global__ __launch_bounds__(256, 2) void dwf_bwd_kernel(DwfParams p) {
  using TL = DwfTile<S>;
  constexpr int PITCH = dwf_pitch<T>();
  constexpr int ND = TL::DH * TL::DW;
  constexpr int NLD = (ND * 8 + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* dt = reinterpret_cast<T*>(smem_raw);                  // [ND][PITCH]  dy, zero outside the image
  float* red = reinterpret_cast<float*>(dt + ND * PITCH);  // [2][CC] statistics
  float* dwl = red + 2 * DWF_CC;                          // [9][CC] dW of this workgroup
  float* wl = dwl + 9 * DWF_CC;                           // [9][CC] weights
  float* cst = wl + 9 * DWF_CC;                           // [7][CC] mean, invstd, scale, shift (BatchNorm in front); ca, cb, cc (behind)

  const int tid = threadIdx.x, cl = tid & 7, pl = tid >> 3;
  // Work items = (spatial tile, 64-channel chunk), chunk fastest, handed out ROUND-ROBIN: item = k * gridDim.x + lb.  At any moment the
  // resident workgroups cover a contiguous run of tiles with all their channel chunks, so DRAM pages are used completely while they are
  // open (a contiguous private tile range per workgroup — 512 x 30 concurrent 2 KB streams — measured 2.0-2.8 TB/s on loads alone);
  // lb is XCD-contiguous, so a tile's chunks and its neighbours share one L2.  gridDim.x % chunks == 0: a workgroup's chunk is fixed.
  const int lb = xcd_chunk_id(blockIdx.x, gridDim.x);
  const int chunk = lb % p.chunks, row_id = lb / p.chunks;
  const int c0 = chunk * DWF_CC, ch = c0 + cl * 8;
  const bool ch_ok = ch < p.C;
  const T* __restrict__ xr = reinterpret_cast<const T*>(p.x);
  const T* __restrict__ go = reinterpret_cast<const T*>(p.g_out);
  const T* __restrict__ yo = reinterpret_cast<const T*>(p.xf.src2);
  T* __restrict__ gi = reinterpret_cast<T*>(p.y);
  const bool dy2src = p.xf.mode == 2;

  for (int i = tid; i < 11 * DWF_CC; i += 256) red[i] = 0.f;
  stage_weights<T>(wl, reinterpret_cast<const T*>(p.wp), p.C, c0);
#pragma unroll
  for (int v = 0; v < 4; ++v) stage_vec(cst + v * DWF_CC, p.in_stats + (size_t)v * p.C, p.C, c0);
  stage_vec(cst + 4 * DWF_CC, dy2src ? p.xf.c0 : nullptr, p.C, c0);
  stage_vec(cst + 5 * DWF_CC, dy2src ? p.xf.c1 : nullptr, p.C, c0);
  stage_vec(cst + 6 * DWF_CC, dy2src ? p.xf.c2 : nullptr, p.C, c0);

  float dwa[9][8];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < 8; ++j) dwa[t][j] = 0.f;
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s1[j] = s2[j] = 0.f;

  const int t_begin = row_id, t_end = p.ntiles, t_step = gridDim.x / p.chunks;

  // z, act' and xhat of one own pixel from its raw value (zeros for a pixel outside the image: it then contributes nothing to dW)
  auto prep = [&](const V8<T>& xraw, bool ok, float* z, float* gp) __attribute__((always_inline)) {
    float xv[8], sc[8], sh[8], yh[8];
    v8_unpack(xraw, xv);
    lds_f8(cst + 2 * DWF_CC + cl * 8, sc);
    lds_f8(cst + 3 * DWF_CC + cl * 8, sh);
#pragma unroll
    for (int j = 0; j < 8; ++j) yh[j] = xv[j] * sc[j] + sh[j];
    if (p.dbg & 8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { z[j] = yh[j]; gp[j] = 1.f; }
    } else {
      act_fwd_grad8(yh, p.in_act, z, gp);
    }
    if (!ok) {
#pragma unroll
      for (int j = 0; j < 8; ++j) z[j] = 0.f;
    }
  };
  // g_in = dz * act' stored; statistics of the stored value
  auto emit = [&](const float* dz, const float* gp, const V8<T>& xraw, bool ok, int b, int hi, int wi) __attribute__((always_inline)) {
    if (ok) {
      float g[8], xv[8], mu[8], is[8];
      v8_unpack(xraw, xv);
      lds_f8(cst + cl * 8, mu);
      lds_f8(cst + DWF_CC + cl * 8, is);
#pragma unroll
      for (int j = 0; j < 8; ++j) g[j] = (p.dbg & 4) ? dz[j] : dz[j] * gp[j];
      V8<T> ov;
      v8_pack(g, ov);
      v8_store<T>(gi + ((size_t)(b * p.H + hi) * p.W + wi) * p.C + ch, ov);
      float gr[8];
      v8_unpack(ov, gr);
#pragma unroll
      for (int j = 0; j < 8; ++j) { s1[j] += gr[j]; s2[j] += gr[j] * (xv[j] - mu[j]) * is[j]; }
    }
  };

  for (int tix = t_begin; tix < t_end; tix += t_step) {
    const int tw = tix % p.tiles_w;
    const int t1 = tix / p.tiles_w;
    const int th = t1 % p.tiles_h;
    const int b = t1 / p.tiles_h;
    const int ho0 = th * TL::OH, wo0 = tw * TL::OW;   // first owned output pixel
    const int hi0 = ho0 * S, wi0 = wo0 * S;           // first input pixel of the tile
    // dy tile origin: stride 1: (ho0 - 1, wo0 - 1), stride 2: (ho0, wo0)
    const int dh0 = S == 1 ? ho0 - 1 : ho0, dw0 = S == 1 ? wo0 - 1 : wo0;

    {
      V8<T> rg[NLD], ry[NLD];
      bool dok[NLD];
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        const int px = (tid + i * 256) >> 3;
        const int pr = px / TL::DW, pc = px - pr * TL::DW;
        const int ho = dh0 + pr, wo = dw0 + pc;
        dok[i] = px < ND && ho >= 0 && ho < p.Ho && wo >= 0 && wo < p.Wo && ch_ok;
        const size_t o = ((size_t)(b * p.Ho + ho) * p.Wo + wo) * p.C + ch;
        rg[i] = v8_load_clamped<T>(go, o, dok[i]);
        ry[i] = v8_load_clamped<T>(dy2src ? yo : go, o, dok[i]);
      }
      __syncthreads();  // previous tile fully consumed (first iteration: staged vectors visible)
      Coef8 kd;
      lds_f8(cst + 4 * DWF_CC + cl * 8, kd.a);
      lds_f8(cst + 5 * DWF_CC + cl * 8, kd.b);
      lds_f8(cst + 6 * DWF_CC + cl * 8, kd.c);
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        const int px = (tid + i * 256) >> 3;
        if (px < ND) {
          V8<T> v = v8_mask(rg[i], dok[i]);
          if (dy2src) v = xf_apply<T>(rg[i], ry[i], kd, 2, 0, dok[i]);
          v8_store<T>(dt + px * PITCH + cl * 8, v);
        }
      }
    }
    __syncthreads();

    if (S == 1) {
      const int r = pl >> 2, q = pl & 3;  // row r, columns 4q .. 4q+3 (tile coordinates); the dy tile is offset by (-1, -1)
      const int hi = hi0 + r;
      V8<T> xc[4];
      bool pok[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int wi = wi0 + q * 4 + t;
        pok[t] = hi < p.H && wi < p.W && ch_ok;
        xc[t] = v8_load_clamped<T>(xr, ((size_t)(b * p.H + hi) * p.W + wi) * p.C + ch, pok[t]);
      }
      // two halves of two adjacent pixels each (z, act', xhat and the accumulators of all four pixels at once do not fit the register
      // file next to the 72 dW accumulators): a half's window is 4 dy columns per row
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        float z[2][8], gp[2][8], acc[2][8];
        __builtin_amdgcn_sched_barrier(0);  // keep the halves apart: interleaved, their live ranges spill
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          prep(xc[half * 2 + t], pok[half * 2 + t], z[t], gp[t]);
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[t][j] = 0.f;
        }
        // pixel (r, c): dy[r + 1 - kh][c + 1 - kw] = dy-tile row r + dh (kh = 2 - dh), column c + dwc (kw = 2 - dwc)
#pragma unroll
        for (int dh = 0; dh < 3; ++dh) {
          float wr[3][8];
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) lds_f8(wl + ((2 - dh) * 3 + kw) * DWF_CC + cl * 8, wr[kw]);
          const T* rowp = dt + ((r + dh) * TL::DW + q * 4 + half * 2) * PITCH + cl * 8;
#pragma unroll
          for (int cix = 0; cix < 4; ++cix) {
            float f[8];
            v8_unpack(v8_load<T>(rowp + cix * PITCH), f);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              const int dwc = cix - t;
              if (dwc >= 0 && dwc < 3) {
                if (!(p.dbg & 2)) {
#pragma unroll
                  for (int j = 0; j < 8; ++j) acc[t][j] += f[j] * wr[2 - dwc][j];
                }
                if (!(p.dbg & 1)) {
#pragma unroll
                  for (int j = 0; j < 8; ++j) dwa[(2 - dh) * 3 + (2 - dwc)][j] += f[j] * z[t][j];
                }
              }
            }
          }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) emit(acc[t], gp[t], xc[half * 2 + t], pok[half * 2 + t], b, hi, wi0 + q * 4 + half * 2 + t);
      }
    } else {
      // stride 2: lane = quad row qr (0..7), quads 2qc, 2qc+1.  Quad (qh, qw) = inputs (2qh + {0,1}, 2qw + {0,1}) from dy[qh + {0,1}][qw + {0,1}]
      const int qr = pl >> 2, qc = pl & 3;
      float d[2][3][8];  // dy rows qr, qr+1; cols 2qc .. 2qc+2
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c2 = 0; c2 < 3; ++c2) v8_unpack(v8_load<T>(dt + ((qr + a) * TL::DW + 2 * qc + c2) * PITCH + cl * 8), d[a][c2]);
#pragma unroll
      for (int ph = 0; ph < 2; ++ph) {
        const int hi = hi0 + 2 * qr + ph;
        V8<T> xc[4];   // the four pixels of this input row: (quad qq, parity pw)
        bool pok[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int wi = wi0 + 2 * (2 * qc + (e >> 1)) + (e & 1);
          pok[e] = hi < p.H && wi < p.W && ch_ok;
          xc[e] = v8_load_clamped<T>(xr, ((size_t)(b * p.H + hi) * p.W + wi) * p.C + ch, pok[e]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int qq = e >> 1, pw = e & 1;
          float z[8], gp[8], acc[8];
          prep(xc[e], pok[e], z, gp);
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] = 0.f;
          // hi even: kh = 1 -> ho = qh.   hi odd: kh = 2 -> ho = qh, kh = 0 -> ho = qh + 1.   wi likewise.
#pragma unroll
          for (int a = 0; a < 2; ++a) {
            if (ph == 0 && a == 1) continue;
            const int kh = ph == 0 ? 1 : (a == 0 ? 2 : 0);
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) {
              if (pw == 0 && c2 == 1) continue;
              const int kw = pw == 0 ? 1 : (c2 == 0 ? 2 : 0);
              float wv[8];
              lds_f8(wl + (kh * 3 + kw) * DWF_CC + cl * 8, wv);
              if (!(p.dbg & 2)) {
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += d[a][qq + c2][j] * wv[j];
              }
              if (!(p.dbg & 1)) {
#pragma unroll
                for (int j = 0; j < 8; ++j) dwa[kh * 3 + kw][j] += d[a][qq + c2][j] * z[j];
              }
            }
          }
          emit(acc, gp, xc[e], pok[e], b, hi, wi0 + 2 * (2 * qc + qq) + pw);
        }
      }
    }
  }

  // ---- workgroup totals: statistics and dW ----
  __syncthreads();
  lds_ordered_accumulate(pl, 32, true, [&]() {  // the 32 pixel lanes of a channel group, in order
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      red[cl * 8 + j] += s1[j];
      red[DWF_CC + cl * 8 + j] += s2[j];
    }
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int j = 0; j < 8; ++j) dwl[t * DWF_CC + cl * 8 + j] += dwa[t][j];
  });
  for (int i = tid; i < 2 * DWF_CC; i += 256) {
    const int which = i / DWF_CC, c = c0 + (i - which * DWF_CC);
    if (c < p.C) p.stats_part[((size_t)row_id * 2 + which) * p.C + c] = red[i];
  }
  for (int i = tid; i < 9 * DWF_CC; i += 256) {
    const int t = i / DWF_CC, c = c0 + (i - t * DWF_CC);
    if (c < p.C) p.dw_part[(size_t)row_id * p.C * 9 + (size_t)c * 9 + t] = dwl[i];
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------------
template <typename T, int S> static size_t dwf_fwd_smem() {
  using TL = DwfTile<S>;
  return (size_t)TL::IH * TL::IW * dwf_pitch<T>() * sizeof(T) + 13 * DWF_CC * sizeof(float);
}
template <typename T, int S> static size_t dwf_bwd_smem() {
  using TL = DwfTile<S>;
  return (size_t)(TL::DH * TL::DW) * dwf_pitch<T>() * sizeof(T) + 27 * DWF_CC * sizeof(float);
}

template <typename K> static int dwf_launch(K kern, size_t smem, dim3 grid, hipStream_t st, const DwfParams& p, DynSmemAttr* attr) {
  if (hipError_t e = attr->ensure(reinterpret_cast<const void*>(kern), smem); e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(kern, grid, dim3(256), smem, st, p);
  CVH_CHECK_LAUNCH();
  return 0;
}

static int dwf_plan(DwfParams& p, int stride, dim3* grid) {  // returns the number of partial rows
  const int OH = 8, OW = stride == 1 ? 16 : 8;
  p.tiles_h = (p.Ho + OH - 1) / OH;
  p.tiles_w = (p.Wo + OW - 1) / OW;
  p.ntiles = p.B * p.tiles_h * p.tiles_w;
  p.chunks = (p.C + DWF_CC - 1) / DWF_CC;
  int rows = 2048 / p.chunks;   // ~2048 workgroups; rows of the partial-statistics buffers <= 512
  if (rows > 512) rows = 512;
  if (rows < 32) rows = 32;
  if (rows > p.ntiles) rows = p.ntiles;
  *grid = dim3(rows * p.chunks, 1);
  return rows;
}

extern "C" int cvh_dwconv_bn_rows(int B, int Ho, int Wo, int C, int stride) {
  if (C % 8 || C <= 0 || (stride != 1 && stride != 2) || B <= 0) return -2;
  DwfParams p;
  p.B = B; p.Ho = Ho; p.Wo = Wo; p.C = C;
  dim3 grid;
  return dwf_plan(p, stride, &grid);
}

extern "C" int cvh_dwconv_bn_fwd(int dtype, const void* x, const cvh_operand_xf* x_xf, const void* wp, void* y, int B, int H, int W, int Ho,
                                 int Wo, int C, int stride, float* stats_part, void* stream) {
  if (C % 8 || C <= 0 || (stride != 1 && stride != 2)) return -2;
  if (Ho != (H + 2 - 3) / stride + 1 || Wo != (W + 2 - 3) / stride + 1) return -2;
  DwfParams p;
  p.x = x; p.xf = make_xf(x_xf); p.wp = wp; p.y = y; p.g_out = nullptr; p.in_stats = nullptr; p.in_act = 0;
  p.stats_part = stats_part; p.dw_part = nullptr;
  p.B = B; p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo; p.C = C;
  p.dbg = 0;
  if (p.xf.mode == 2) return -2;
  if (B <= 0) return 0;
  dim3 grid;
  dwf_plan(p, stride, &grid);
  hipStream_t st = (hipStream_t)stream;
  static DynSmemAttr a0, a1, a2, a3;
  if (dtype == CVH_DT_BF16) {
    if (stride == 1) return dwf_launch(dwf_fwd_kernel<bf16_t, 1>, dwf_fwd_smem<bf16_t, 1>(), grid, st, p, &a0);
    return dwf_launch(dwf_fwd_kernel<bf16_t, 2>, dwf_fwd_smem<bf16_t, 2>(), grid, st, p, &a1);
  } else if (dtype == CVH_DT_F32) {
    if (stride == 1) return dwf_launch(dwf_fwd_kernel<float, 1>, dwf_fwd_smem<float, 1>(), grid, st, p, &a2);
    return dwf_launch(dwf_fwd_kernel<float, 2>, dwf_fwd_smem<float, 2>(), grid, st, p, &a3);
  }
  return -1;
}

extern "C" int cvh_dwconv_bn_bwd(int dtype, const void* g_out, const cvh_operand_xf* dy_xf, const void* x_raw, const float* in_stats,
                                 int in_act, const void* wp, void* g_in, float* stats_part, float* dw_part, int B, int H, int W, int Ho,
                                 int Wo, int C, int stride, void* stream) {
  if (C % 8 || C <= 0 || (stride != 1 && stride != 2)) return -2;
  if (Ho != (H + 2 - 3) / stride + 1 || Wo != (W + 2 - 3) / stride + 1) return -2;
  if (in_stats == nullptr || stats_part == nullptr || dw_part == nullptr) return -2;
  DwfParams p;
  p.x = x_raw; p.xf = make_xf(dy_xf); p.wp = wp; p.y = g_in; p.g_out = g_out; p.in_stats = in_stats; p.in_act = in_act;
  p.stats_part = stats_part; p.dw_part = dw_part;
  p.B = B; p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo; p.C = C;
  p.dbg = cvh_tune_get(7);
  if (p.xf.mode == 1 || (p.xf.mode == 2 && p.xf.src2 == nullptr)) return -2;
  if (B <= 0) return 0;
  dim3 grid;
  dwf_plan(p, stride, &grid);
  hipStream_t st = (hipStream_t)stream;
  static DynSmemAttr a0, a1, a2, a3;
  if (dtype == CVH_DT_BF16) {
    if (stride == 1) return dwf_launch(dwf_bwd_kernel<bf16_t, 1>, dwf_bwd_smem<bf16_t, 1>(), grid, st, p, &a0);
    return dwf_launch(dwf_bwd_kernel<bf16_t, 2>, dwf_bwd_smem<bf16_t, 2>(), grid, st, p, &a1);
  } else if (dtype == CVH_DT_F32) {
    if (stride == 1) return dwf_launch(dwf_bwd_kernel<float, 1>, dwf_bwd_smem<float, 1>(), grid, st, p, &a2);
    return dwf_launch(dwf_bwd_kernel<float, 2>, dwf_bwd_smem<float, 2>(), grid, st, p, &a3);
  }
  return -1;
}
