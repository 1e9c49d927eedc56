// Implicit-GEMM convolution / linear kernels on MFMA (gfx950), NHWC activations.
//
//   conv_gemm  (NT):  out[m, n] = epilogue( sum_k A(m, k) * Wp[n, k] )
//        m = (b, ho, wo) output pixel, k = (kh, kw, c) with c contiguous (NHWC), Wp = packed weights
//        [N][KH*KW*Cin].  Pointwise conv / nn.Linear are the KH = KW = 1 special case (A = x[m, :]).
//        A can be the channel-concat of two tensors (MobileViT fusion conv reads cat(res, fm) without
//        materialising it).  Used for: forward of every dense conv / linear, and for dX of stride-1
//        convs / linears with the flipped-transposed weight pack.
//   gemm_tn  (dW):    dW[n, k] += sum_m dY[m, n] * A(m, k)      (reduction over pixels, split over m)
//
// Replaces (reference, ATen calls): nn.Conv2d.forward  cvnets/layers/conv_layer.py:18-66,254-255;
// F.linear cvnets/layers/linear_layer.py:90; and their autograd backward.
#include "common.hpp"
#include "cvnets_hip.h"

#include "gemm_params.hpp"

// staging-group decomposition of the NF accumulator fragments of a wave: groups of 4 / 2 / 1 fragments so that the
// number of 8-wide column chunks per staged row (16 / 8 / 4) divides the wave size
template <int NF> struct StageGroups;
template <> struct StageGroups<1> { static constexpr int n = 1; static constexpr int start[2] = {0, 0}; static constexpr int width[2] = {1, 0}; };
template <> struct StageGroups<2> { static constexpr int n = 1; static constexpr int start[2] = {0, 0}; static constexpr int width[2] = {2, 0}; };
template <> struct StageGroups<3> { static constexpr int n = 2; static constexpr int start[2] = {0, 2}; static constexpr int width[2] = {2, 1}; };
template <> struct StageGroups<4> { static constexpr int n = 1; static constexpr int start[2] = {0, 0}; static constexpr int width[2] = {4, 0}; };
template <> struct StageGroups<5> { static constexpr int n = 2; static constexpr int start[2] = {0, 4}; static constexpr int width[2] = {4, 1}; };

template <typename T, int NF> constexpr int stage_pitch() { return (NF >= 4 ? 128 : (NF >= 2 ? 64 : 32)) + 16 / (int)sizeof(T); }

// Workgroup = 4 waves stacked along M (BM = 128 rows), each wave owns a 32 x (32*NF) output strip.
template <typename T, int NF, int BK>
__global__ __launch_bounds__(256, 2) void conv_gemm_kernel(ConvGemmParams p) {
  constexpr int BM = 128;
  constexpr int BN = 32 * NF;
  constexpr int CPR = BK / 8;
  constexpr int PITCH = lds_pitch<T>(BK);
  constexpr int A_IT = (BM * CPR + 255) / 256;
  constexpr int B_IT = (BN * CPR + 255) / 256;
  constexpr int SP = stage_pitch<T, NF>();
  constexpr int TILE_ELEMS = (BM + BN) * PITCH;
  constexpr int STAGE_ELEMS = 4 * 32 * SP;
  constexpr int MAIN_ELEMS = TILE_ELEMS + STAGE_ELEMS;
  using SG = StageGroups<NF>;

  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* As = reinterpret_cast<T*>(smem_raw);
  T* Bs = As + BM * PITCH;
  float* red = reinterpret_cast<float*>(As + MAIN_ELEMS);  // [2][BN] column sums (sum, sumsq), LDS atomics

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  T* stg = As + TILE_ELEMS + wave * (32 * SP);  // per-wave PRIVATE output staging: the epilogue needs only wave-level ordering
  const int n0 = blockIdx.y * BN;
  const int Cin = p.C1 + p.C2;
  const T* __restrict__ src1 = reinterpret_cast<const T*>(p.src1);
  const T* __restrict__ src2 = reinterpret_cast<const T*>(p.src2);
  const T* __restrict__ wgt = reinterpret_cast<const T*>(p.wgt);
  T* __restrict__ out = reinterpret_cast<T*>(p.out);

  const int ccol = tid % CPR;  // this thread's 8-wide K chunk column inside a tile row (same for all its chunks)
  const bool pointwise = (p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0);
  const bool want_stats = p.stats_part != nullptr;

  // running column statistics (BatchNorm) of this lane's 8-column chunk(s), across all M tiles of this block
  float cs1[2][8], cs2[2][8];
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int j = 0; j < 8; ++j) cs1[g][j] = cs2[g][j] = 0.f;
  if (want_stats) {
    for (int i = tid; i < 2 * BN; i += 256) red[i] = 0.f;
  }

  unsigned long long seed = 0;
  if (p.drop_p > 0.f) seed = *p.seed;
  const float inv_keep = p.drop_p > 0.f ? 1.0f / (1.0f - p.drop_p) : 1.0f;

  // ---- software pipeline across M tiles: the A/B registers of tile t+1 are requested before tile t's epilogue ----
  int a_row[A_IT];
  int a_b[A_IT], a_h[A_IT], a_w[A_IT];
  bool a_ok[A_IT];
  V8<T> ra[A_IT], rb[B_IT];
  // single K step (pointwise convs with Cin <= BK): the weight tile is staged ONCE per workgroup, not once per M tile
  const bool b_resident = p.Ktot <= BK;
  if (b_resident) {
    const int k = ccol * 8;
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      int q = tid + i * 256;
      int n = n0 + q / CPR;
      V8<T> v = v8_zero<T>();
      if (q < BN * CPR && n < p.N && k < p.Ktot) v = v8_load<T>(wgt + (size_t)n * p.Ktot + k);
      if (q < BN * CPR) v8_store<T>(Bs + (q / CPR) * PITCH + ccol * 8, v);
    }
  }

  auto decode_rows = [&](int m0) {
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      int q = tid + i * 256;
      int r = q / CPR;
      a_row[i] = r;
      int m = m0 + r;
      a_ok[i] = (q < BM * CPR) && (m < p.M);
      if (pointwise) {
        a_b[i] = 0; a_h[i] = 0; a_w[i] = m;  // linear pixel index
      } else {
        int hw = p.Ho * p.Wo;
        int b = m / hw;
        int rem = m - b * hw;
        int ho = rem / p.Wo;
        int wo = rem - ho * p.Wo;
        a_b[i] = b; a_h[i] = ho * p.stride - p.pad; a_w[i] = wo * p.stride - p.pad;
      }
    }
  };
  auto load_tiles = [&](int k0) {
    const int k = k0 + ccol * 8;
    const bool kok = k < p.Ktot;
    int tap = 0, c = k;
    if (!pointwise) { tap = k / Cin; c = k - tap * Cin; }
    const int kh = tap / p.KW, kw = tap - kh * p.KW;
    const T* s = src1; int cs = p.C1; int cc = c;
    if (c >= p.C1) { s = src2; cs = p.C2; cc = c - p.C1; }
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      ra[i] = v8_zero<T>();
      if (a_ok[i] && kok) {
        if (pointwise) {
          ra[i] = v8_load<T>(s + (size_t)a_w[i] * cs + cc);
        } else {
          int hi = a_h[i] + kh * p.dil, wi = a_w[i] + kw * p.dil;
          if (hi >= 0 && hi < p.H && wi >= 0 && wi < p.W)
            ra[i] = v8_load<T>(s + ((size_t)(a_b[i] * p.H + hi) * p.W + wi) * cs + cc);
        }
      }
    }
    if (!b_resident) {
#pragma unroll
      for (int i = 0; i < B_IT; ++i) {
        int q = tid + i * 256;
        int r = q / CPR;
        int n = n0 + r;
        rb[i] = v8_zero<T>();
        if (q < BN * CPR && n < p.N && kok) rb[i] = v8_load<T>(wgt + (size_t)n * p.Ktot + k);
      }
    }
  };
  auto store_tiles = [&]() {
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      int q = tid + i * 256;
      if (q < BM * CPR) v8_store<T>(As + a_row[i] * PITCH + ccol * 8, ra[i]);
    }
    if (!b_resident) {
#pragma unroll
      for (int i = 0; i < B_IT; ++i) {
        int q = tid + i * 256;
        if (q < BN * CPR) v8_store<T>(Bs + (q / CPR) * PITCH + ccol * 8, rb[i]);
      }
    }
  };

  if ((int)blockIdx.x < p.m_tiles) {
    decode_rows(blockIdx.x * BM);
    load_tiles(0);
  }
  for (int tile_m = blockIdx.x; tile_m < p.m_tiles; tile_m += gridDim.x) {
    const int m0 = tile_m * BM;

    f32x16_t acc[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) acc[f] = acc_zero();

    for (int k0 = 0; k0 < p.Ktot; k0 += BK) {
      __syncthreads();  // previous tile (or previous epilogue's staging) fully consumed
      store_tiles();
      __syncthreads();
      if (k0 + BK < p.Ktot) load_tiles(k0 + BK);  // prefetch next K tile into registers under the MFMAs
#pragma unroll
      for (int kk = 0; kk < BK; kk += 16) {
        Frag<T> a = lds_frag(As, PITCH, wave * 32, kk, lane);
        static_for<0, NF>([&](auto fi) {
          constexpr int f = decltype(fi)::value;
          Frag<T> b = lds_frag(Bs, PITCH, f * 32, kk, lane);
          mma32(acc[f], a, b);
        });
      }
    }
    {  // request the first operand tiles of the NEXT M tile now: their HBM latency hides under this tile's epilogue
      const int next = tile_m + gridDim.x;
      if (next < p.m_tiles) {
        decode_rows(next * BM);
        load_tiles(0);
      }
    }

    // ---- epilogue: accumulators -> (bias) -> per-wave LDS staging -> coalesced 16 B/lane rows with the fused
    //      activation / act-grad / dropout / residual / BatchNorm statistics ----
    static_for<0, SG::n>([&](auto gi) {
      constexpr int g = decltype(gi)::value;
      constexpr int F0 = SG::start[g], GW = SG::width[g];
      constexpr int CH = GW * 4;        // 8-wide chunks per staged row: 16 / 8 / 4
      constexpr int RPP = 64 / CH;      // rows per pass
      static_for<0, GW>([&](auto fi) {
        constexpr int fl = decltype(fi)::value;
        constexpr int f = F0 + fl;
        const int n = n0 + f * 32 + (lane & 31);
        const float bias = (p.bias != nullptr && n < p.N) ? p.bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) stg[acc_row(r, lane) * SP + fl * 32 + (lane & 31)] = from_f<T>(acc[f][r] + bias);
      });
      wave_lds_sync();
      const int ch = lane % CH;
      const int n = n0 + F0 * 32 + ch * 8;
      if (n < p.N) {
#pragma unroll
        for (int pass = 0; pass < 32 / RPP; ++pass) {
          const int row = pass * RPP + lane / CH;
          const int m = m0 + wave * 32 + row;
          if (m < p.M) {
            size_t o = (size_t)m * p.N + n;
            if (p.sc_s) {
              const int hw = p.sc_Ho * p.sc_Wo;
              const int b = m / hw, rem = m - b * hw;
              const int ho = rem / p.sc_Wo, wo = rem - ho * p.sc_Wo;
              const int tap = n / p.sc_C, c = n - tap * p.sc_C;
              const int kh = tap / p.sc_KW, kw = tap - kh * p.sc_KW;
              o = (((size_t)b * p.sc_H + ho * p.sc_s + kh) * p.sc_W + wo * p.sc_s + kw) * p.sc_C + c;
            }
            V8<T> pv = v8_load<T>(stg + row * SP + ch * 8);
            if (p.save_pre) v8_store<T>(reinterpret_cast<T*>(p.save_pre) + o, pv);
            float v[8];
            v8_unpack(pv, v);
            if (p.act != CVH_ACT_NONE) {
#pragma unroll
              for (int j = 0; j < 8; ++j) v[j] = act_fwd(v[j], p.act);
            }
            if (p.actgrad_aux) {
              float a[8];
              v8_unpack(v8_load<T>(reinterpret_cast<const T*>(p.actgrad_aux) + o), a);
#pragma unroll
              for (int j = 0; j < 8; ++j) v[j] *= act_grad(a[j], p.actgrad_act);
            }
            if (p.drop_p > 0.f) {
#pragma unroll
              for (int j = 0; j < 8; ++j) v[j] *= dropout_scale(seed, p.stream_id, o + j, p.drop_p, inv_keep);
            }
            if (p.residual) {
              float rr[8];
              v8_unpack(v8_load<T>(reinterpret_cast<const T*>(p.residual) + o), rr);
#pragma unroll
              for (int j = 0; j < 8; ++j) v[j] += rr[j];
            }
            V8<T> ov;
            v8_pack(v, ov);
            v8_store<T>(out + o, ov);
            if (want_stats) {
              float vr[8];
              v8_unpack(ov, vr);  // statistics of the values as stored
#pragma unroll
              for (int j = 0; j < 8; ++j) { cs1[g][j] += vr[j]; cs2[g][j] += vr[j] * vr[j]; }
            }
          }
        }
      }
      wave_lds_sync();  // staging consumed before the next group / next tile overwrites it
    });
  }

  if (want_stats) {
    static_for<0, SG::n>([&](auto gi) {
      constexpr int g = decltype(gi)::value;
      constexpr int CH = SG::width[g] * 4;
      const int col = SG::start[g] * 32 + (lane % CH) * 8;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        atomicAdd(&red[col + j], cs1[g][j]);
        atomicAdd(&red[BN + col + j], cs2[g][j]);
      }
    });
    __syncthreads();
    for (int col = tid; col < BN; col += 256) {
      int n = n0 + col;
      if (n < p.N) {
        p.stats_part[(size_t)blockIdx.x * 2 * p.N + n] = red[col];
        p.stats_part[(size_t)blockIdx.x * 2 * p.N + p.N + n] = red[BN + col];
      }
    }
  }
}

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

template <typename T, int PV>
__global__ __launch_bounds__(256) void gemm_tn_kernel(GemmTNParams p) {
  constexpr int BMR = 32;
  // bf16: 72-byte rows put the 8-row-apart column chunks of a 32-lane write group on disjoint bank halves (the transposed
  // row-pair stores become conflict-free); fragments are then read as two 8-byte halves.
  constexpr int PITCH = (sizeof(T) == 2 && PV == 1) ? 36 : lds_pitch<T>(BMR);
  __shared__ __attribute__((aligned(16))) T Dt[128 * PITCH];
  __shared__ __attribute__((aligned(16))) T Xt[128 * PITCH];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_n = wave >> 1, wave_k = wave & 1;
  const int tile_n = blockIdx.x / p.k_tiles, tile_k = blockIdx.x % p.k_tiles;
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
  if (c >= p.C1) { s = reinterpret_cast<const T*>(p.src2); cs = p.C2; cc = c - p.C1; }

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = acc_zero();

  const int m_begin = blockIdx.y * p.m_per_split;
  const int m_end = min(p.M, m_begin + p.m_per_split);

  // PF register stages in flight per thread (each = 2 rows of dY + 2 rows of A): the ring is indexed statically by unrolling
  constexpr int PF = 4;
  V8<T> rd0[PF], rd1[PF], rx0[PF], rx1[PF];
  auto load_stage = [&](V8<T>& d0, V8<T>& d1, V8<T>& x0, V8<T>& x1, int ms) {
    d0 = d1 = x0 = x1 = v8_zero<T>();
    const int ma = ms + 2 * mp, mb = ma + 1;
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
    if (m_begin + u * BMR < m_end) load_stage(rd0[u], rd1[u], rx0[u], rx1[u], m_begin + u * BMR);
  for (int ms0 = m_begin; ms0 < m_end; ms0 += PF * BMR) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int ms = ms0 + u * BMR;
      if (ms < m_end) {  // uniform across the workgroup
        __syncthreads();
        store_transposed_pair(Dt + (nc * 8) * PITCH + 2 * mp, PITCH, rd0[u], rd1[u]);
        store_transposed_pair(Xt + (nc * 8) * PITCH + 2 * mp, PITCH, rx0[u], rx1[u]);
        __syncthreads();
        if (ms + PF * BMR < m_end) load_stage(rd0[u], rd1[u], rx0[u], rx1[u], ms + PF * BMR);
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
        float* dst = p.part + (size_t)blockIdx.y * p.N * p.Ktot + k;
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

// dw[n][c][tap] (torch layout) = (or +=) sum over splits of part[split][n][tap*Cin + c].
// Block = 16 consecutive output elements x 16 split lanes (the split loop is the long dimension for pointwise convs).
__global__ __launch_bounds__(256) void gemm_dw_reduce_kernel(const float* __restrict__ part, int splits, int N, int Ktot, int Cin, int Cin_real,
                                                             int khw, float* __restrict__ dw, int accumulate) {
  __shared__ float red[16][17];
  const size_t total = (size_t)N * Ktot;
  const int e = threadIdx.x & 15, sl = threadIdx.x >> 4;
  for (size_t base = (size_t)blockIdx.x * 16; base < total; base += (size_t)gridDim.x * 16) {
    const size_t idx = base + e;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (idx < total) {
      int sp = sl;
      for (; sp + 48 < splits; sp += 64) {
        s0 += part[(size_t)sp * total + idx];
        s1 += part[(size_t)(sp + 16) * total + idx];
        s2 += part[(size_t)(sp + 32) * total + idx];
        s3 += part[(size_t)(sp + 48) * total + idx];
      }
      for (; sp < splits; sp += 16) s0 += part[(size_t)sp * total + idx];
    }
    red[sl][e] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (sl == 0 && idx < total) {
      float s = 0.f;
#pragma unroll
      for (int l = 0; l < 16; ++l) s += red[l][e];
      const int k = (int)(idx % Ktot);
      const int n = (int)(idx / Ktot);
      const int tap = k / Cin, c = k - tap * Cin;
      if (c < Cin_real) {
        float* d = dw + ((size_t)n * Cin_real + c) * khw + tap;
        *d = accumulate ? *d + s : s;
      }
    }
    __syncthreads();
  }
}

// =============================================================================================
// host-side dispatch
// =============================================================================================
template <typename T, int NF, int BK>
static int launch_conv_gemm(const ConvGemmParams& p0, hipStream_t st) {
  constexpr int BM = 128, BN = 32 * NF;
  ConvGemmParams p = p0;
  p.m_tiles = (p.M + BM - 1) / BM;
  const int n_tiles = (p.N + BN - 1) / BN;
  const int cap = cvh_tune_get(CVH_TUNE_GEMM_GRID);
  int gx = p.m_tiles < cap ? p.m_tiles : cap;
  dim3 grid(gx, n_tiles);
  constexpr int TILE_ELEMS = (BM + BN) * lds_pitch<T>(BK);
  constexpr int STAGE_ELEMS = 4 * 32 * stage_pitch<T, NF>();
  constexpr int MAIN_ELEMS = TILE_ELEMS + STAGE_ELEMS;
  size_t smem = (size_t)MAIN_ELEMS * sizeof(T) + (size_t)2 * BN * sizeof(float);
  auto kern = conv_gemm_kernel<T, NF, BK>;
  if (smem > 64 * 1024) {
    static bool attr_set = false;  // one instantiation = one static
    if (!attr_set) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      if (e != hipSuccess) return (int)e;
      attr_set = true;
    }
  }
  hipLaunchKernelGGL(kern, grid, dim3(256), smem, st, p);
  CVH_CHECK_LAUNCH();
  return 0;
}

template <typename T, int BK>
static int dispatch_conv_gemm_nf(const ConvGemmParams& p, int nf, hipStream_t st) {
  switch (nf) {
    case 1: return launch_conv_gemm<T, 1, BK>(p, st);
    case 2: return launch_conv_gemm<T, 2, BK>(p, st);
    case 3: return launch_conv_gemm<T, 3, BK>(p, st);
    case 4: return launch_conv_gemm<T, 4, BK>(p, st);
    default: return launch_conv_gemm<T, 5, BK>(p, st);
  }
}

// choose the N tiling: fewest padded columns, then fewest tiles
static int choose_nf(int N) {
  static const int cand[5] = {1, 2, 3, 4, 5};
  int best = 4, best_cost = 1 << 30;
  for (int i = 0; i < 5; ++i) {
    int bn = 32 * cand[i];
    int tiles = (N + bn - 1) / bn;
    int cost = tiles * bn * 16 + tiles;  // padded width dominates, tile count breaks ties
    if (cost < best_cost) { best_cost = cost; best = cand[i]; }
  }
  return best;
}

extern "C" int cvh_conv_gemm_grid_rows(int M, int N) {
  // number of stats-partial rows conv_gemm writes for an (M, N) problem (== gridDim.x)
  (void)N;
  int mt = (M + 127) / 128;
  const int cap = cvh_tune_get(CVH_TUNE_GEMM_GRID);
  return mt < cap ? mt : cap;
}

extern "C" int cvh_conv_gemm(int dtype, const void* src1, const void* src2, int C1, int C2, const void* wgt, void* out,
                             int B, int H, int W, int Ho, int Wo, int KH, int KW, int stride, int pad, int dil, int N,
                             const float* bias, int act, void* save_pre, const void* actgrad_aux, int actgrad_act,
                             const void* residual, float drop_p, const unsigned long long* seed, unsigned int stream_id,
                             float* stats_part, void* stream) {
  if ((C1 % 8) != 0 || (C2 % 8) != 0 || C1 <= 0 || (N % 8) != 0) return -2;
  if (src2 == nullptr && C2 != 0) return -2;
  ConvGemmParams p;
  p.src1 = src1; p.src2 = src2; p.C1 = C1; p.C2 = C2; p.wgt = wgt; p.out = out;
  p.B = B; p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo; p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad; p.dil = dil;
  p.M = B * Ho * Wo; p.N = N; p.Ktot = KH * KW * (C1 + C2);
  p.bias = bias; p.act = act; p.save_pre = save_pre; p.actgrad_aux = actgrad_aux; p.actgrad_act = actgrad_act;
  p.residual = residual; p.drop_p = drop_p; p.seed = seed; p.stream_id = stream_id; p.stats_part = stats_part;
  p.m_tiles = 0;
  p.sc_s = 0; p.sc_KW = p.sc_C = p.sc_H = p.sc_W = p.sc_Ho = p.sc_Wo = 0;
  if (p.M <= 0 || N <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CVH_DT_BF16 && gemm_big_eligible(p)) return launch_gemm_big(p, st);  // transformer-sized linears (ViT-B / CLIP)
  const int nf = choose_nf(N);
  const bool bk64 = p.Ktot >= 64;
  if (dtype == CVH_DT_BF16) {
    return bk64 ? dispatch_conv_gemm_nf<bf16_t, 64>(p, nf, st) : dispatch_conv_gemm_nf<bf16_t, 32>(p, nf, st);
  } else if (dtype == CVH_DT_F32) {
    return dispatch_conv_gemm_nf<float, 32>(p, nf, st);
  }
  return -1;
}

// dX of a non-overlapping strided conv (kernel == stride, pad 0; the ViT patch-embedding convs, cvnets/models/classification/vit.py:89-123):
// every input pixel belongs to exactly one window, so dX = scatter( dY[M][Cout] x Wp3[(kh,kw,c)][Cout]^T ) — one GEMM whose
// epilogue writes each 8-channel chunk to its pixel.  wgt = mode-3 pack.  Pixels outside every window (H > Ho*s) are zeroed first.
extern "C" int cvh_conv_dx_patch(int dtype, const void* dy, const void* wgt, void* dx, int B, int Ho, int Wo, int Cout, int KH, int KW,
                                 int stride, int Cin, int H, int W, void* stream) {
  if ((Cout % 8) != 0 || (Cin % 8) != 0 || KH != stride || KW != stride || Ho * stride > H || Wo * stride > W) return -2;
  hipStream_t st = (hipStream_t)stream;
  const size_t esz = dtype == CVH_DT_BF16 ? 2 : 4;
  if (Ho * stride != H || Wo * stride != W) {
    hipError_t e = hipMemsetAsync(dx, 0, (size_t)B * H * W * Cin * esz, st);
    if (e != hipSuccess) return (int)e;
  }
  ConvGemmParams p;
  p.src1 = dy; p.src2 = nullptr; p.C1 = Cout; p.C2 = 0; p.wgt = wgt; p.out = dx;
  p.B = B * Ho * Wo; p.H = 1; p.W = 1; p.Ho = 1; p.Wo = 1; p.KH = 1; p.KW = 1; p.stride = 1; p.pad = 0; p.dil = 1;
  p.M = B * Ho * Wo; p.N = KH * KW * Cin; p.Ktot = Cout;
  p.bias = nullptr; p.act = 0; p.save_pre = nullptr; p.actgrad_aux = nullptr; p.actgrad_act = 0; p.residual = nullptr;
  p.drop_p = 0.f; p.seed = nullptr; p.stream_id = 0; p.stats_part = nullptr; p.m_tiles = 0;
  p.sc_s = stride; p.sc_KW = KW; p.sc_C = Cin; p.sc_H = H; p.sc_W = W; p.sc_Ho = Ho; p.sc_Wo = Wo;
  if (p.M <= 0) return 0;
  const int nf = choose_nf(p.N);
  const bool bk64 = p.Ktot >= 64;
  if (dtype == CVH_DT_BF16) return bk64 ? dispatch_conv_gemm_nf<bf16_t, 64>(p, nf, st) : dispatch_conv_gemm_nf<bf16_t, 32>(p, nf, st);
  if (dtype == CVH_DT_F32) return dispatch_conv_gemm_nf<float, 32>(p, nf, st);
  return -1;
}

static bool tn_big_shape(int M, int N, int Ktot) { return M >= 2048 && N >= 256 && (N % 128) == 0 && Ktot >= 256 && (Ktot % 128) == 0; }

static void tn_plan(int M, int N, int Ktot, int* out_tiles, int* k_tiles, int* splits, int* mps) {
  const int n_tiles = (N + 127) / 128;
  *k_tiles = (Ktot + 127) / 128;
  *out_tiles = n_tiles * *k_tiles;
  int sp;
  if (tn_big_shape(M, N, Ktot)) {
    // transformer-sized dW (gemm_tn128_kernel, 2 workgroups per CU = 512 slots): pick the split count that minimises
    // (rounds of workgroups) x (64-row steps per split) + the cost of summing the partial tiles.  Plain "enough workgroups"
    // planning lands on e.g. 144 tiles x 8 splits = 2.25 rounds, i.e. a third of the machine idle in the last round.
    const int slots = 512, total_steps = (M + 63) / 64;
    double best = 1e30;
    sp = 1;
    for (int s = 1; s <= 32 && s <= total_steps; ++s) {
      const int rounds = (*out_tiles * s + slots - 1) / slots;
      const int steps = (total_steps + s - 1) / s;
      const double t = (double)rounds * steps * 1.8 + (double)(s + 1) * (double)N * Ktot * 4.0 / 3.0e6;  // microseconds
      if (t < best) { best = t; sp = s; }
    }
  } else {
    // enough splits over M to put ~2 deep-prefetching workgroups on every CU
    // (K-heavy 3x3 problems have many output tiles and long per-split MFMA chains: they like twice as many workgroups)
    const int target_wgs = cvh_tune_get(CVH_TUNE_TN_WGS) * (*out_tiles >= 8 ? 2 : 1);
    sp = (target_wgs + *out_tiles - 1) / *out_tiles;
  }
  int max_splits = (M + 255) / 256;
  if (sp > max_splits) sp = max_splits;
  if (sp < 1) sp = 1;
  int m = (M + sp - 1) / sp;
  m = ((m + 63) / 64) * 64;  // whole 64-row reduction steps (gemm_tn128_kernel); also a multiple of the 32-row step of gemm_tn_kernel
  *splits = (M + m - 1) / m;
  *mps = m;
}

// part[split][total] -> dw[total] (linear layers: torch layout == GEMM layout), 16 bytes per lane, splits summed in registers
__global__ __launch_bounds__(256) void gemm_dw_reduce_linear_kernel(const float* __restrict__ part, int splits, size_t total4, float* __restrict__ dw,
                                                                    int accumulate) {
  const float4* __restrict__ p4 = reinterpret_cast<const float4*>(part);
  float4* __restrict__ d4 = reinterpret_cast<float4*>(dw);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256) {
    float4 s = accumulate ? d4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    int sp = 0;
    for (; sp + 3 < splits; sp += 4) {  // four independent loads in flight
      const float4 a = p4[(size_t)sp * total4 + i], b = p4[(size_t)(sp + 1) * total4 + i];
      const float4 c = p4[(size_t)(sp + 2) * total4 + i], d = p4[(size_t)(sp + 3) * total4 + i];
      s.x += (a.x + b.x) + (c.x + d.x); s.y += (a.y + b.y) + (c.y + d.y);
      s.z += (a.z + b.z) + (c.z + d.z); s.w += (a.w + b.w) + (c.w + d.w);
    }
    for (; sp < splits; ++sp) {
      const float4 a = p4[(size_t)sp * total4 + i];
      s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
    }
    d4[i] = s;
  }
}

extern "C" long long cvh_gemm_dw_scratch_elems(int M, int N, int Ktot) {
  if (M <= 0) return 0;
  int ot, kt, sp, mps;
  tn_plan(M, N, Ktot, &ot, &kt, &sp, &mps);
  return (long long)sp * N * Ktot;
}

extern "C" int cvh_gemm_dw(int dtype, const void* dy, const void* src1, const void* src2, int C1, int C2, float* dw,
                           int B, int H, int W, int Ho, int Wo, int KH, int KW, int stride, int pad, int dil, int N,
                           int Cin_real, float* scratch, long long scratch_elems, int accumulate, void* stream) {
  if ((C1 % 8) != 0 || (C2 % 8) != 0 || (N % 8) != 0) return -2;
  GemmTNParams p;
  p.dy = dy; p.src1 = src1; p.src2 = src2; p.C1 = C1; p.C2 = C2; p.dw = dw;
  p.B = B; p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo; p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad; p.dil = dil;
  p.M = B * Ho * Wo; p.N = N; p.Ktot = KH * KW * (C1 + C2); p.Cin_real = Cin_real;
  if (p.M <= 0) return 0;
  int out_tiles, splits, mps;
  tn_plan(p.M, N, p.Ktot, &out_tiles, &p.k_tiles, &splits, &mps);
  p.m_per_split = mps;
  // scratch path: every split writes its partial tile, one reduce kernel sums them (assign or accumulate into dw).
  // Without scratch: fp32 atomics into dw, which the caller must have zeroed (accumulate semantics only).
  p.part = nullptr;
  if (scratch != nullptr) {
    if (scratch_elems < (long long)splits * N * p.Ktot) return -2;
    p.part = scratch;
  } else if (!accumulate) {
    return -2;
  }
  dim3 grid(out_tiles, splits);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CVH_DT_BF16 && p.part != nullptr && gemm_tn_big_eligible(p)) {  // transformer-sized linears (ViT-B / CLIP)
    const int rc = launch_gemm_tn_big(p, splits, st);
    if (rc) return rc;
  } else if (dtype == CVH_DT_BF16) {
    if (cvh_tune_get(CVH_TUNE_TN_PITCH)) hipLaunchKernelGGL((gemm_tn_kernel<bf16_t, 1>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((gemm_tn_kernel<bf16_t, 0>), grid, dim3(256), 0, st, p);
  } else if (dtype == CVH_DT_F32) hipLaunchKernelGGL((gemm_tn_kernel<float, 0>), grid, dim3(256), 0, st, p);
  else return -1;
  CVH_CHECK_LAUNCH();
  if (p.part && KH * KW == 1 && Cin_real == p.Ktot && ((size_t)N * p.Ktot) % 4 == 0 && splits <= 64 && (size_t)N * p.Ktot >= 65536) {
    const size_t total4 = (size_t)N * p.Ktot / 4;
    int g = (int)((total4 + 255) / 256);
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(gemm_dw_reduce_linear_kernel, dim3(g), dim3(256), 0, st, p.part, splits, total4, dw, accumulate);
    CVH_CHECK_LAUNCH();
  } else if (p.part) {
    const size_t total = (size_t)N * p.Ktot;
    int g = (int)((total + 15) / 16);
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(gemm_dw_reduce_kernel, dim3(g), dim3(256), 0, st, p.part, splits, N, p.Ktot, C1 + C2, Cin_real, KH * KW, dw, accumulate);
    CVH_CHECK_LAUNCH();
  }
  return 0;
}
