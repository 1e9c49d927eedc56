// gemm_tn_kernel: dW = dY^T x im2col(X) (see gemm.hip for the overview).  Template shared by gemm.hip (FX = 0) and gemm_fx.hip
// (FX = 1: both operands transformed on load, bnlink.hpp).
#pragma once
#include "common.hpp"
#include "cvnets_hip.h"
#include "gemm_params.hpp"

// =============================================================================================
// dW kernel:  dW[n, k] += sum_{m in split} dY[m, n] * A(m, k)         (both operands M-major in HBM)
// Tile 128(n) x 128(k) per workgroup, 4 waves as 2x2, each wave 64x64 = 2x2 MFMA 32x32 tiles.
// Both operand tiles are transposed on their way into LDS (so fragments are contiguous in m):
// lanes run along m in row PAIRS and write packed {row 2i, row 2i+1} words.
// =============================================================================================

__device__ __forceinline__ void store_transposed_pair(bf16_t* dst, int pitch, const V8<bf16_t>& r0, const V8<bf16_t>& r1) {
  const uint32_t a[4] = {r0.d.x, r0.d.y, r0.d.z, r0.d.w};
  const uint32_t b[4] = {r1.d.x, r1.d.y, r1.d.z, r1.d.w};
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    uint32_t w = (j & 1) ? ((a[j >> 1] >> 16) | (b[j >> 1] & 0xffff0000u)) : ((a[j >> 1] & 0xffffu) | (b[j >> 1] << 16));
    *reinterpret_cast<uint32_t*>(dst + j * pitch) = w;
  }
}
__device__ __forceinline__ void store_transposed_pair(float* dst, int pitch, const V8<float>& r0, const V8<float>& r1) {
  float a[8], b[8];
  v8_unpack(r0, a);
  v8_unpack(r1, b);
#pragma unroll
  for (int j = 0; j < 8; ++j) *reinterpret_cast<float2*>(dst + j * pitch) = make_float2(a[j], b[j]);
}

template <typename T, int PV, int FX>
__global__ __launch_bounds__(256) void gemm_tn_kernel(GemmTNParams p) {
  constexpr int BMR = 32;
  // bf16: 72-byte rows put the 8-row-apart column chunks of a 32-lane write group on disjoint bank halves (the transposed
  // row-pair stores become conflict-free); fragments are then read as two 8-byte halves.
  constexpr int PITCH = (sizeof(T) == 2 && PV == 1) ? 36 : lds_pitch<T>(BMR);
  __shared__ __attribute__((aligned(16))) T Dt[128 * PITCH];
  __shared__ __attribute__((aligned(16))) T Xt[128 * PITCH];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_n = wave >> 1, wave_k = wave & 1;
  // XCD-contiguous work order, output tile fastest: the workgroups that reduce the SAME rows m (one per output tile) run on the same XCD
  // and share the dY / X rows in its L2 (hardware: linear workgroup id b -> XCD b % 8)
  const int bid_ = xcd_chunk_id((int)(blockIdx.y * gridDim.x + blockIdx.x), (int)(gridDim.x * gridDim.y));
  const int bx = bid_ % (int)gridDim.x, by = bid_ / (int)gridDim.x;
  const int tile_n = bx / p.k_tiles, tile_k = bx % p.k_tiles;
  const int n0 = tile_n * 128, k0 = tile_k * 128;
  const int Cin = p.C1 + p.C2;
  const T* __restrict__ dy = reinterpret_cast<const T*>(p.dy);
  const bool pointwise = (p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0);

  const int mp = tid & 15;   // row pair inside the 32-row stage
  const int nc = tid >> 4;   // 8-wide column chunk (0..15) of the 128-wide tiles
  // dY column / A(m,k) column handled by this thread
  const int n_col = n0 + nc * 8;
  const bool n_ok = n_col < p.N;
  const int k_col = k0 + nc * 8;
  const bool k_ok = k_col < p.Ktot;
  int tap = 0, c = k_col;
  if (!pointwise && k_ok) { tap = k_col / Cin; c = k_col - tap * Cin; }
  const int kh = tap / p.KW, kw = tap - kh * p.KW;
  const T* s = reinterpret_cast<const T*>(p.src1);
  int cs = p.C1, cc = c;
  if (!FX && c >= p.C1) { s = reinterpret_cast<const T*>(p.src2); cs = p.C2; cc = c - p.C1; }  // FX: single source; predicated loads need a valid base

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = acc_zero();

  const int m_begin = by * p.m_per_split;
  const int m_end = min(p.M, m_begin + p.m_per_split);

  // PF register stages in flight per thread (each = 2 rows of dY + 2 rows of A): the ring is indexed statically by unrolling
  constexpr int PF = 4;
  V8<T> rd0[PF], rd1[PF], rx0[PF], rx1[PF];
  // FX: operand transforms on load (bnlink.hpp): dY = ca*g + cb*y + cc needs a second source; X = act(scale*x + shift)
  // FX = 1: dY plain, X plain or act(c0*x + c1);  FX = 2: additionally dY = c0*g + c1*y + c2 from two sources (compile-time: a
  // run-time "maybe a second load" defeats SROA of the register ring)
  V8<T> ry0[FX == 2 ? PF : 1], ry1[FX == 2 ? PF : 1];
  const int dy_mode = FX == 2 ? 2 : 0, x_mode = FX ? (p.x_xf.mode == 1 ? 1 : 0) : 0;
  const T* __restrict__ dy2 = FX ? reinterpret_cast<const T*>(p.dy_xf.src2) : nullptr;
  Coef8 kd, kx;
  if (FX && dy_mode) coef8_load(kd, p.dy_xf, n_col, n_ok);
  if (FX && x_mode) coef8_load(kx, p.x_xf, k_col, k_ok);
  auto load_stage = [&](V8<T>& d0, V8<T>& d1, V8<T>& x0, V8<T>& x1, V8<T>& y0, V8<T>& y1, int ms) __attribute__((always_inline)) {
    const int ma = ms + 2 * mp, mb = ma + 1;
    if (FX) {  // pointwise by construction; clamped loads, masked where they are consumed (store to LDS)
      const bool va = ma < m_end, vb = mb < m_end;
      d0 = v8_load_clamped<T>(dy, (size_t)ma * p.N + n_col, va && n_ok);
      d1 = v8_load_clamped<T>(dy, (size_t)mb * p.N + n_col, vb && n_ok);
      x0 = v8_load_clamped<T>(s, (size_t)ma * cs + cc, va && k_ok);
      x1 = v8_load_clamped<T>(s, (size_t)mb * cs + cc, vb && k_ok);
      if (FX == 2) {
        y0 = v8_load_clamped<T>(dy2, (size_t)ma * p.N + n_col, va && n_ok);
        y1 = v8_load_clamped<T>(dy2, (size_t)mb * p.N + n_col, vb && n_ok);
      }
      return;
    }
    d0 = d1 = x0 = x1 = v8_zero<T>();
    if (n_ok) {
      if (ma < m_end) d0 = v8_load<T>(dy + (size_t)ma * p.N + n_col);
      if (mb < m_end) d1 = v8_load<T>(dy + (size_t)mb * p.N + n_col);
    }
    if (k_ok) {
      if (pointwise) {
        if (ma < m_end) x0 = v8_load<T>(s + (size_t)ma * cs + cc);
        if (mb < m_end) x1 = v8_load<T>(s + (size_t)mb * cs + cc);
      } else {
        const int hw = p.Ho * p.Wo;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int m = e ? mb : ma;
          if (m < m_end) {
            int b = m / hw;
            int rem = m - b * hw;
            int ho = rem / p.Wo;
            int wo = rem - ho * p.Wo;
            int hi = ho * p.stride - p.pad + kh * p.dil, wi = wo * p.stride - p.pad + kw * p.dil;
            if (hi >= 0 && hi < p.H && wi >= 0 && wi < p.W) {
              V8<T> v = v8_load<T>(s + ((size_t)(b * p.H + hi) * p.W + wi) * cs + cc);
              if (e) x1 = v; else x0 = v;
            }
          }
        }
      }
    }
  };

#pragma unroll
  for (int u = 0; u < PF; ++u)
    if (m_begin + u * BMR < m_end) load_stage(rd0[u], rd1[u], rx0[u], rx1[u], ry0[FX == 2 ? u : 0], ry1[FX == 2 ? u : 0], m_begin + u * BMR);
  for (int ms0 = m_begin; ms0 < m_end; ms0 += PF * BMR) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int ms = ms0 + u * BMR;
      if (ms < m_end) {  // uniform across the workgroup
        __syncthreads();
        if (FX) {  // rows beyond m_end (and columns beyond N / K) must contribute zero AFTER the transform: the reduction runs over m
          const bool va = ms + 2 * mp < m_end, vb = ms + 2 * mp + 1 < m_end;
          if (dy_mode) {
            rd0[u] = xf_apply<T>(rd0[u], ry0[FX == 2 ? u : 0], kd, dy_mode, p.dy_xf.act, va && n_ok);
            rd1[u] = xf_apply<T>(rd1[u], ry1[FX == 2 ? u : 0], kd, dy_mode, p.dy_xf.act, vb && n_ok);
          } else {
            rd0[u] = v8_mask(rd0[u], va && n_ok);
            rd1[u] = v8_mask(rd1[u], vb && n_ok);
          }
          if (x_mode) {
            rx0[u] = xf_apply<T>(rx0[u], rx0[u], kx, x_mode, p.x_xf.act, va && k_ok);
            rx1[u] = xf_apply<T>(rx1[u], rx1[u], kx, x_mode, p.x_xf.act, vb && k_ok);
          } else {
            rx0[u] = v8_mask(rx0[u], va && k_ok);
            rx1[u] = v8_mask(rx1[u], vb && k_ok);
          }
        }
        store_transposed_pair(Dt + (nc * 8) * PITCH + 2 * mp, PITCH, rd0[u], rd1[u]);
        store_transposed_pair(Xt + (nc * 8) * PITCH + 2 * mp, PITCH, rx0[u], rx1[u]);
        __syncthreads();
        if (ms + PF * BMR < m_end) load_stage(rd0[u], rd1[u], rx0[u], rx1[u], ry0[FX == 2 ? u : 0], ry1[FX == 2 ? u : 0], ms + PF * BMR);
#pragma unroll
        for (int kk = 0; kk < BMR; kk += 16) {
          Frag<T> a0 = PV == 1 ? lds_frag_a8(Dt, PITCH, wave_n * 64, kk, lane) : lds_frag(Dt, PITCH, wave_n * 64, kk, lane);
          Frag<T> a1 = PV == 1 ? lds_frag_a8(Dt, PITCH, wave_n * 64 + 32, kk, lane) : lds_frag(Dt, PITCH, wave_n * 64 + 32, kk, lane);
          Frag<T> b0 = PV == 1 ? lds_frag_a8(Xt, PITCH, wave_k * 64, kk, lane) : lds_frag(Xt, PITCH, wave_k * 64, kk, lane);
          Frag<T> b1 = PV == 1 ? lds_frag_a8(Xt, PITCH, wave_k * 64 + 32, kk, lane) : lds_frag(Xt, PITCH, wave_k * 64 + 32, kk, lane);
          mma32(acc[0][0], a0, b0);
          mma32(acc[0][1], a0, b1);
          mma32(acc[1][0], a1, b0);
          mma32(acc[1][1], a1, b1);
        }
      }
    }
  }

  const int khw = p.KH * p.KW;
#pragma unroll
  for (int fn = 0; fn < 2; ++fn)
#pragma unroll
    for (int fk = 0; fk < 2; ++fk) {
      const int k = k0 + wave_k * 64 + fk * 32 + (lane & 31);
      if (k >= p.Ktot) continue;
      if (p.part) {  // plain coalesced stores of this split's partial tile; gemm_dw_reduce_kernel sums the splits
        float* dst = p.part + (size_t)by * p.N * p.Ktot + k;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = n0 + wave_n * 64 + fn * 32 + acc_row(r, lane);
          if (n < p.N) dst[(size_t)n * p.Ktot] = acc[fn][fk][r];
        }
        continue;
      }
      int t2 = 0, c2 = k;
      if (!pointwise) { t2 = k / Cin; c2 = k - t2 * Cin; }
      if (c2 >= p.Cin_real) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wave_n * 64 + fn * 32 + acc_row(r, lane);
        if (n < p.N) atomicAdd(p.dw + ((size_t)n * p.Cin_real + c2) * khw + t2, acc[fn][fk][r]);
      }
    }
}
