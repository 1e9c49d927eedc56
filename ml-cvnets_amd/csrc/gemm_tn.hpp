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
  for (int j = 0; j < 8; ++j) {  // {row 2i element j, row 2i+1 element j}: one v_perm_b32 per packed word
    const uint32_t w = __builtin_amdgcn_perm(b[j >> 1], a[j >> 1], (j & 1) ? 0x07060302u : 0x05040100u);
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

template <typename T, int PW, int FX>  // PW = 1: pointwise problems only (no im2col address arithmetic in the instruction stream)
__global__ __launch_bounds__(256) void gemm_tn_kernel(GemmTNParams p) {
  constexpr int PV = 0;
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
  const bool pointwise = PW || FX || (p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0);

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
  if (!FX && k_ok && c >= p.C1) { s = reinterpret_cast<const T*>(p.src2); cs = p.C2; cc = c - p.C1; }  // k_ok: lanes without a column keep the (always valid) first source as the base of their clamped loads

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = acc_zero();

  const int m_begin = by * p.m_per_split;
  const int m_end = min(p.M, m_begin + p.m_per_split);
  const bool do_bias = p.bias_part != nullptr && tile_k == 0;
  float bsum[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) bsum[j] = 0.0f;

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
  unsigned okm = 0;  // non-FX: validity bits of the ring slots (d0, d1, x0, x1 per slot)
  auto load_stage = [&](V8<T>& d0, V8<T>& d1, V8<T>& x0, V8<T>& x1, V8<T>& y0, V8<T>& y1, int ms, int slot) __attribute__((always_inline)) {
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
    // Branch-free: clamped addresses, validity kept as 4 bits per ring slot and applied where the values are consumed.  (Per-lane
    // `if (ok) v = load` puts the loads behind exec-mask branches; the compiler can then no longer COUNT the loads in flight, every wait
    // becomes s_waitcnt vmcnt(0) and the 4-deep ring degenerates to one memory latency per stage — tools/waitcnt_survey.py.)
    {
      const bool va = ma < m_end && n_ok, vb = mb < m_end && n_ok;
      d0 = v8_load_clamped<T>(dy, (size_t)ma * p.N + n_col, va);
      d1 = v8_load_clamped<T>(dy, (size_t)mb * p.N + n_col, vb);
      bool xa, xb;
      if (pointwise) {
        xa = ma < m_end && k_ok;
        xb = mb < m_end && k_ok;
        x0 = v8_load_clamped<T>(s, (size_t)ma * cs + cc, xa);
        x1 = v8_load_clamped<T>(s, (size_t)mb * cs + cc, xb);
      } else {
        const int hw = p.Ho * p.Wo;
        size_t off[2];
        bool okx[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int m = e ? mb : ma;
          const int b = m / hw;
          const int rem = m - b * hw;
          const int ho = rem / p.Wo;
          const int wo = rem - ho * p.Wo;
          const int hi = ho * p.stride - p.pad + kh * p.dil, wi = wo * p.stride - p.pad + kw * p.dil;
          okx[e] = m < m_end && k_ok && hi >= 0 && hi < p.H && wi >= 0 && wi < p.W;
          off[e] = ((size_t)(b * p.H + hi) * p.W + wi) * cs + cc;
        }
        xa = okx[0];
        xb = okx[1];
        x0 = v8_load_clamped<T>(s, off[0], xa);
        x1 = v8_load_clamped<T>(s, off[1], xb);
      }
      const unsigned bits = (va ? 1u : 0u) | (vb ? 2u : 0u) | (xa ? 4u : 0u) | (xb ? 8u : 0u);
      okm = (okm & ~(15u << (4 * slot))) | (bits << (4 * slot));
    }
  };

  // No control flow inside the ring: every group of PF stages runs completely — stages past m_end load zeros (all lanes predicated
  // off) and add nothing — so that the compiler can count the loads in flight at every wait.
#pragma unroll
  for (int u = 0; u < PF; ++u) load_stage(rd0[u], rd1[u], rx0[u], rx1[u], ry0[FX == 2 ? u : 0], ry1[FX == 2 ? u : 0], m_begin + u * BMR, u);
  for (int ms0 = m_begin; ms0 < m_end; ms0 += PF * BMR) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int ms = ms0 + u * BMR;
      {
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
        if (!FX) {
          const unsigned bits = okm >> (4 * u);
          rd0[u] = v8_mask(rd0[u], (bits & 1u) != 0);
          rd1[u] = v8_mask(rd1[u], (bits & 2u) != 0);
          rx0[u] = v8_mask(rx0[u], (bits & 4u) != 0);
          rx1[u] = v8_mask(rx1[u], (bits & 8u) != 0);
        }
        if (do_bias) {  // column sums of dY (bias gradient) ride along in the workgroups of the first k tile
          float f0[8], f1[8];
          v8_unpack(rd0[u], f0);
          v8_unpack(rd1[u], f1);
#pragma unroll
          for (int j = 0; j < 8; ++j) bsum[j] += f0[j] + f1[j];
        }
        store_transposed_pair(Dt + (nc * 8) * PITCH + 2 * mp, PITCH, rd0[u], rd1[u]);
        store_transposed_pair(Xt + (nc * 8) * PITCH + 2 * mp, PITCH, rx0[u], rx1[u]);
        __syncthreads();
        load_stage(rd0[u], rd1[u], rx0[u], rx1[u], ry0[FX == 2 ? u : 0], ry1[FX == 2 ? u : 0], ms + PF * BMR, u);
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

  if (do_bias) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = bsum[j];
      v += __shfl_xor(v, 1);
      v += __shfl_xor(v, 2);
      v += __shfl_xor(v, 4);
      v += __shfl_xor(v, 8);
      if (mp == 0 && n_ok) p.bias_part[(size_t)by * p.N + n_col + j] = v;
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

// =============================================================================================
// Skinny pointwise dW:  N <= 128, K <= 64 (the 1x1 convolutions of the early, high-resolution stages: M = B*H*W is millions of rows,
// the output is one small tile).  gemm_tn_kernel spends such launches waiting: 3/4 of its lanes have no column to load and every
// 32-row stage costs two workgroup barriers.  Here every wave streams its OWN 32-row stages (wave w of split s takes stages w, w+4,
// ...) through a wave-private LDS region — no workgroup barrier anywhere — with all lanes loading: a "unit" is 32 columns x 32 rows
// (lane = row pair x 8-column chunk, the conflict-free transposed-store mapping of gemm_tn_kernel), NT units of dY and KT of X per
// stage, PF stages in flight per wave.  Every wave writes its own partial tile (rows of the scratch = 4 x splits); with BIAS the
// column sums of dY (the bias gradient of the same layer) ride along.
// =============================================================================================
// XF = 1: the X operand is act(c0 * x + c1) (x_xf mode 1, bnlink.hpp) — the projection dW of the fused InvertedResidual, whose X is the raw
// depthwise output: the transform is applied to the registers on their way into LDS (a lane's 8 columns are fixed: coefficients in registers).
template <int NT, int KT, int PF, int BIAS, int XF = 0>
__global__ __launch_bounds__(256) void gemm_tn_skinny_kernel(GemmTNParams p) {
  using T = bf16_t;
  constexpr int PITCH = 36;
  constexpr int WAVE_ELEMS = (NT + KT) * 32 * PITCH;
  __shared__ __attribute__((aligned(16))) T lds[4 * WAVE_ELEMS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int by = xcd_chunk_id((int)blockIdx.x, (int)gridDim.x);
  T* Dt = lds + wave * WAVE_ELEMS;
  T* Xt = Dt + NT * 32 * PITCH;
  const int mp = lane & 15, cq = lane >> 4;
  const T* __restrict__ dy = reinterpret_cast<const T*>(p.dy);
  const T* __restrict__ xs = reinterpret_cast<const T*>(p.src1);
  const int N = p.N, K = p.Ktot;

  f32x16_t acc[NT][KT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < KT; ++j) acc[i][j] = acc_zero();
  float bs[BIAS ? NT : 1][8];
#pragma unroll
  for (int i = 0; i < (BIAS ? NT : 1); ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) bs[i][j] = 0.0f;

  const int m_begin = by * p.m_per_split;
  const int m_end = min(p.M, m_begin + p.m_per_split);
  constexpr int STEP = 4 * 32;  // the four waves interleave 32-row stages: one workgroup reads 128 consecutive rows at a time

  V8<T> rd[PF][NT][2], rx[PF][KT][2];
  auto load_stage = [&](V8<T> (&d)[NT][2], V8<T> (&x)[KT][2], int ms) __attribute__((always_inline)) {
    const int ma = ms + 2 * mp, mb = ma + 1;
    const bool va = ma < m_end, vb = mb < m_end;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int col = t * 32 + cq * 8;
      d[t][0] = v8_load_clamped<T>(dy, (size_t)ma * N + col, va && col < N);
      d[t][1] = v8_load_clamped<T>(dy, (size_t)mb * N + col, vb && col < N);
    }
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      const int col = t * 32 + cq * 8;
      x[t][0] = v8_load_clamped<T>(xs, (size_t)ma * K + col, va && col < K);
      x[t][1] = v8_load_clamped<T>(xs, (size_t)mb * K + col, vb && col < K);
    }
  };

  Coef8 kx[XF ? KT : 1];
  if (XF) {
#pragma unroll
    for (int t = 0; t < KT; ++t) coef8_load(kx[t], p.x_xf, t * 32 + cq * 8, t * 32 + cq * 8 < K);
  }
  const int ms_w = m_begin + wave * 32;
  // no control flow inside the ring (see gemm_tn_kernel): stages past m_end are fully predicated off and add nothing
#pragma unroll
  for (int u = 0; u < PF; ++u) load_stage(rd[u], rx[u], ms_w + u * STEP);
  for (int ms0 = ms_w; ms0 < m_end; ms0 += PF * STEP) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int ms = ms0 + u * STEP;
      {
        const bool va = ms + 2 * mp < m_end, vb = ms + 2 * mp + 1 < m_end;
        wave_lds_sync();  // the fragment reads of the previous stage are behind us
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const bool ok = t * 32 + cq * 8 < N;
          const V8<T> d0 = v8_mask(rd[u][t][0], va && ok), d1 = v8_mask(rd[u][t][1], vb && ok);
          if (BIAS) {
            float f0[8], f1[8];
            v8_unpack(d0, f0);
            v8_unpack(d1, f1);
#pragma unroll
            for (int j = 0; j < 8; ++j) bs[t][j] += f0[j] + f1[j];
          }
          store_transposed_pair(Dt + (t * 32 + cq * 8) * PITCH + 2 * mp, PITCH, d0, d1);
        }
#pragma unroll
        for (int t = 0; t < KT; ++t) {
          const bool ok = t * 32 + cq * 8 < K;
          if (XF)
            store_transposed_pair(Xt + (t * 32 + cq * 8) * PITCH + 2 * mp, PITCH, xf_apply<T>(rx[u][t][0], rx[u][t][0], kx[t], 1, p.x_xf.act, va && ok),
                                  xf_apply<T>(rx[u][t][1], rx[u][t][1], kx[t], 1, p.x_xf.act, vb && ok));
          else
            store_transposed_pair(Xt + (t * 32 + cq * 8) * PITCH + 2 * mp, PITCH, v8_mask(rx[u][t][0], va && ok), v8_mask(rx[u][t][1], vb && ok));
        }
        wave_lds_sync();
        load_stage(rd[u], rx[u], ms + PF * STEP);
#pragma unroll
        for (int kk = 0; kk < 32; kk += 16) {
          Frag<T> a[NT], b[KT];
#pragma unroll
          for (int t = 0; t < NT; ++t) a[t] = lds_frag_a8(Dt, PITCH, t * 32, kk, lane);
#pragma unroll
          for (int t = 0; t < KT; ++t) b[t] = lds_frag_a8(Xt, PITCH, t * 32, kk, lane);
#pragma unroll
          for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int j = 0; j < KT; ++j) mma32(acc[i][j], a[i], b[j]);
        }
      }
    }
  }

  const int row = by * 4 + wave;
  float* dst = p.part + (size_t)row * N * K;
#pragma unroll
  for (int fn = 0; fn < NT; ++fn)
#pragma unroll
    for (int fk = 0; fk < KT; ++fk) {
      const int k = fk * 32 + (lane & 31);
      if (k >= K) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = fn * 32 + acc_row(r, lane);
        if (n < N) dst[(size_t)n * K + k] = acc[fn][fk][r];
      }
    }
  if (BIAS) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float v = bs[t][j];
        v += __shfl_xor(v, 1);
        v += __shfl_xor(v, 2);
        v += __shfl_xor(v, 4);
        v += __shfl_xor(v, 8);
        const int n = t * 32 + cq * 8 + j;
        if (mp == 0 && n < N) p.bias_part[(size_t)row * N + n] = v;
      }
  }
}
