// conv_gemm_kernel: implicit-GEMM convolution / linear on MFMA (see gemm.hip for the overview).  Template shared by gemm.hip
// (FX = 0: the general kernel) and gemm_fx.hip (FX = 1: pointwise GEMMs with operand transforms and BatchNorm links, bnlink.hpp).
#pragma once
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
__device__ __forceinline__ void lds_f8x(const float* p, float* o) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}

// WP = 1 ("wave-private"): single-K-step pointwise problems (Ktot <= BK: the weight tile is resident).  Every wave stages the 32 rows of
// A it multiplies itself, into its own LDS rows, and the tile loop has NO workgroup barrier: the four waves of a workgroup drift apart,
// so one wave's epilogue VALU / HBM waits overlap another wave's loads and MFMAs.  (PMC on the BatchNorm-link GEMMs with the
// cooperative staging: VALUBusy 25 %, MfmaUtil 3 %, LdsUtil 23 %, MemUnitStalled 0 at 7.5 waves per CU — latency, not throughput.)
template <typename T, int NF, int BK, int FX, int WP = 0>
__global__ __launch_bounds__(256, 2) void conv_gemm_kernel(ConvGemmParams p) {
  constexpr int BM = 128;
  constexpr int BN = 32 * NF;
  constexpr int CPR = BK / 8;
  constexpr int PITCH = lds_pitch<T>(BK);
  constexpr int A_IT = WP ? (32 * CPR) / 64 : (BM * CPR + 255) / 256;
  constexpr int B_IT = (BN * CPR + 255) / 256;
  constexpr int SP = stage_pitch<T, NF>();
  constexpr int TILE_ELEMS = (BM + BN) * PITCH;
  // WP: a wave's 32 private A rows are dead once its MFMAs are done, so its epilogue staging can live there (no extra LDS)
  constexpr bool STG_ALIAS = WP && PITCH >= SP;
  constexpr int STAGE_ELEMS = STG_ALIAS ? 0 : 4 * 32 * SP;
  constexpr int MAIN_ELEMS = TILE_ELEMS + STAGE_ELEMS;
  using SG = StageGroups<NF>;

  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* As = reinterpret_cast<T*>(smem_raw);
  T* Bs = As + BM * PITCH;
  float* red = reinterpret_cast<float*>(As + MAIN_ELEMS);  // [2][BN] column sums (sum, sumsq), LDS atomics

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  T* stg = STG_ALIAS ? As + wave * (32 * PITCH) : As + TILE_ELEMS + wave * (32 * SP);  // per-wave PRIVATE output staging: the epilogue needs only wave-level ordering
  // XCD-contiguous work order, N tile fastest: the workgroups that share an M tile (one per N tile) get consecutive logical ids and so
  // run on the same XCD — the A rows are fetched into ONE L2 instead of once per XCD (hardware: linear workgroup id b -> XCD b % 8)
  const int bid_ = xcd_chunk_id((int)(blockIdx.y * gridDim.x + blockIdx.x), (int)(gridDim.x * gridDim.y));
  const int bx = bid_ / (int)gridDim.y, by = bid_ % (int)gridDim.y;
  const int n0 = by * BN;
  const int Cin = p.C1 + p.C2;
  const T* __restrict__ src1 = reinterpret_cast<const T*>(p.src1);
  const T* __restrict__ src2 = reinterpret_cast<const T*>(p.src2);
  const T* __restrict__ wgt = reinterpret_cast<const T*>(p.wgt);
  T* __restrict__ out = reinterpret_cast<T*>(p.out);

  const int ccol = tid % CPR;  // this thread's 8-wide K chunk column inside a tile row (same for all its chunks; WP: lane % CPR == tid % CPR)
  const bool pointwise = (p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0);
  const bool want_stats = p.stats_part != nullptr;
  // FX = 1: A operand plain or act(c0*a + c1) (run-time), epilogue modes, links;  FX = 2: A operand = c0*a + c1*a2 + c2 (two sources)
  const int a_mode = FX == 2 ? 2 : (FX == 1 ? (p.a_xf.mode == 1 ? 1 : 0) : 0);
  const T* __restrict__ src_b = FX ? reinterpret_cast<const T*>(p.a_xf.src2) : nullptr;

  // running column statistics (BatchNorm) of this lane's 8-column chunk(s), across all M tiles of this block
  float cs1[2][8], cs2[2][8];
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int j = 0; j < 8; ++j) cs1[g][j] = cs2[g][j] = 0.f;
  if (want_stats) {
    for (int i = tid; i < 2 * BN; i += 256) red[i] = 0.f;
  }
  // FX e_mode 1: the four per-channel vectors of the BatchNorm being back-propagated through, staged once as
  // est[4][BN] = (invstd, -mean*invstd, scale, shift): the epilogue reads them from LDS (from global memory they cost 8 dependent
  // 16-byte loads per 8 output elements)
  float* est = red + 2 * BN;
  if (FX == 1 && p.e_mode == 1) {
    for (int i = tid; i < BN; i += 256) {
      const int n = n0 + i;
      const bool ok = n < p.N;
      const float mu = ok ? p.e_stats[n] : 0.f, is = ok ? p.e_stats[p.N + n] : 0.f;
      est[i] = is;
      est[BN + i] = -mu * is;
      est[2 * BN + i] = ok ? p.e_stats[2 * p.N + n] : 0.f;
      est[3 * BN + i] = ok ? p.e_stats[3 * p.N + n] : 0.f;
    }
  }

  // bias of this lane's column in every accumulator fragment, loaded ONCE: a (conditional) load inside the epilogue is followed by
  // s_waitcnt vmcnt(0), which also drains the next M tile's operand loads requested just before the epilogue — the prefetch then
  // overlaps nothing and every tile pays a full memory latency up front
  float bias_r[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    const int n = n0 + f * 32 + (lane & 31);
    bias_r[f] = (p.bias != nullptr && n < p.N) ? p.bias[n] : 0.f;
  }

  DropKey dkey = {0u, 0u};
  if (p.drop_p > 0.f) dkey = drop_key(*p.seed, p.stream_id, p.drop_p);
  const float inv_keep = p.drop_p > 0.f ? 1.0f / (1.0f - p.drop_p) : 1.0f;

  // ---- software pipeline across M tiles: the A/B registers of tile t+1 are requested before tile t's epilogue ----
  int a_row[A_IT];
  int a_b[A_IT], a_h[A_IT], a_w[A_IT];
  bool a_ok[A_IT];
  V8<T> ra[A_IT], rb[B_IT];
  V8<T> ra2[FX == 2 ? A_IT : 1];  // FX = 2: second source of the A operand
  Coef8 kc;                  // FX: coefficients of this thread's K chunk
  // single K step (pointwise convs with Cin <= BK): the weight tile is staged ONCE per workgroup, not once per M tile
  const bool b_resident = p.Ktot <= BK;
  if (b_resident) {
    if (FX && a_mode) coef8_load(kc, p.a_xf, ccol * 8, ccol * 8 < p.Ktot);
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      int q = tid + i * 256;
      int n = n0 + q / CPR;
      const int k = (q % CPR) * 8;  // (== ccol * 8 when 256 % CPR == 0)
      V8<T> v = v8_zero<T>();
      if (q < BN * CPR && n < p.N && k < p.Ktot) v = v8_load<T>(wgt + (size_t)n * p.Ktot + k);
      if (q < BN * CPR) v8_store<T>(Bs + (q / CPR) * PITCH + k, v);
    }
  }

  if (WP) __syncthreads();  // the resident weight tile (and the staged per-channel vectors) are visible to every wave; no barrier after this

  int wp_m0 = 0;  // WP: first row of the tile whose A rows are in flight; row / chunk of a load are recomputed from (lane, i), not stored
  auto decode_rows = [&](int m0) __attribute__((always_inline)) {
    if (WP) { wp_m0 = m0; return; }
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      int q = WP ? (tid & 63) + i * 64 : tid + i * 256;
      int r = WP ? (tid >> 6) * 32 + q / CPR : q / CPR;
      a_row[i] = r;
      int m = m0 + r;
      a_ok[i] = (WP || q < BM * CPR) && (m < p.M);
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
  auto load_tiles = [&](int k0) __attribute__((always_inline)) {
    const int k = k0 + ccol * 8;
    const bool kok = k < p.Ktot;
    int tap = 0, c = k;
    if (!pointwise) { tap = k / Cin; c = k - tap * Cin; }
    const int kh = tap / p.KW, kw = tap - kh * p.KW;
    const T* s = src1; int cs = p.C1; int cc = c;
    if (!FX && c >= p.C1) { s = src2; cs = p.C2; cc = c - p.C1; }  // FX: single source; predicated loads need a valid base
    if (FX && a_mode && !b_resident) coef8_load(kc, p.a_xf, k, kok);
    if (WP) {  // single K step, pointwise, one source; branch-free clamped loads
#pragma unroll
      for (int i = 0; i < A_IT; ++i) {
        const int q = (tid & 63) + i * 64;
        const int m = wp_m0 + (tid >> 6) * 32 + q / CPR, c8 = (q % CPR) * 8;
        ra[i] = v8_load_clamped<T>(src1, (size_t)m * p.C1 + c8, m < p.M && c8 < p.Ktot);
      }
    } else if (FX) {  // pointwise by construction; clamped loads, no mask: rows >= M / K chunks >= Ktot never reach a stored value
#pragma unroll
      for (int i = 0; i < A_IT; ++i) ra[i] = v8_load_clamped<T>(s, (size_t)a_w[i] * cs + cc, a_ok[i] && kok);
      if (FX == 2) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) ra2[i] = v8_load_clamped<T>(src_b, (size_t)a_w[i] * cs + cc, a_ok[i] && kok);
      }
    } else {
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
  auto store_tiles = [&]() __attribute__((always_inline)) {
    if (WP) {
#pragma unroll
      for (int i = 0; i < A_IT; ++i) {
        const int q = (tid & 63) + i * 64;
        const int r = (tid >> 6) * 32 + q / CPR, c8 = (q % CPR) * 8;
        T* dst = As + r * PITCH + c8;
        // plain operand: rows >= M / chunks >= Ktot are zeroed here (they met clamped addresses); transformed operand: such rows only
        // feed accumulator rows that are never stored and chunks >= Ktot meet zero weights
        if (FX && a_mode) v8_store<T>(dst, xf_apply<T>(ra[i], ra2[FX == 2 ? i : 0], kc, a_mode, p.a_xf.act, true));
        else v8_store<T>(dst, v8_mask(ra[i], wp_m0 + r < p.M && c8 < p.Ktot));
      }
    } else {
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      int q = tid + i * 256;
      if (FX && a_mode) {
        // rows beyond M / K chunks beyond Ktot only feed accumulator rows that are never stored (or meet zero weights): no masking needed
        if (q < BM * CPR) v8_store<T>(As + a_row[i] * PITCH + ccol * 8, xf_apply<T>(ra[i], ra2[FX == 2 ? i : 0], kc, a_mode, p.a_xf.act, true));
      } else {
        if (q < BM * CPR) v8_store<T>(As + a_row[i] * PITCH + ccol * 8, ra[i]);
      }
    }
    }
    if (!b_resident) {
#pragma unroll
      for (int i = 0; i < B_IT; ++i) {
        int q = tid + i * 256;
        if (q < BN * CPR) v8_store<T>(Bs + (q / CPR) * PITCH + ccol * 8, rb[i]);
      }
    }
  };

  if (bx < p.m_tiles) {
    decode_rows(bx * BM);
    load_tiles(0);
  }
  for (int tile_m = bx; tile_m < p.m_tiles; tile_m += gridDim.x) {
    const int m0 = tile_m * BM;

    f32x16_t acc[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) acc[f] = acc_zero();

    for (int k0 = 0; k0 < p.Ktot; k0 += BK) {
      if (WP) wave_lds_sync(); else __syncthreads();  // previous tile (or previous epilogue's staging) fully consumed
      store_tiles();
      if (WP) wave_lds_sync(); else __syncthreads();
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
    if (STG_ALIAS) wave_lds_sync();  // this wave's fragment reads of its A rows are done before the staging overwrites them
    static_for<0, SG::n>([&](auto gi) {
      constexpr int g = decltype(gi)::value;
      constexpr int F0 = SG::start[g], GW = SG::width[g];
      constexpr int CH = GW * 4;        // 8-wide chunks per staged row: 16 / 8 / 4
      constexpr int RPP = 64 / CH;      // rows per pass
      static_for<0, GW>([&](auto fi) {
        constexpr int fl = decltype(fi)::value;
        constexpr int f = F0 + fl;
        const float bias = bias_r[f];
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
            if (p.act != CVH_ACT_NONE) act_fwd8(v, p.act);
            float xh[8];  // FX e_mode 1: xhat of the BatchNorm being back-propagated through
            if (FX == 1 && p.e_mode == 1) {
              float a[8];
              v8_unpack(v8_load<T>(reinterpret_cast<const T*>(p.e_aux) + o), a);
              const float* ev = est + F0 * 32 + ch * 8;
              float is[8], mi[8], sc[8], sh[8];
              lds_f8x(ev, is);
              lds_f8x(ev + BN, mi);
              lds_f8x(ev + 2 * BN, sc);
              lds_f8x(ev + 3 * BN, sh);
              float yh[8];
#pragma unroll
              for (int j = 0; j < 8; ++j) {
                yh[j] = a[j] * sc[j] + sh[j];
                xh[j] = a[j] * is[j] + mi[j];
              }
              act_grad8_mul(v, yh, p.e_act);
            }
            if (p.actgrad_aux) {
              float a[8];
              v8_unpack(v8_load<T>(reinterpret_cast<const T*>(p.actgrad_aux) + o), a);
              act_grad8_mul(v, a, p.actgrad_act);
            }
            if (p.drop_p > 0.f) dropout_scale8(dkey, o, inv_keep, v);  // o % 8 == 0: N % 8 == 0, 8-column chunks
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
              if (FX == 1 && p.e_mode == 1) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { cs1[g][j] += vr[j]; cs2[g][j] += vr[j] * xh[j]; }
              } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) { cs1[g][j] += vr[j]; cs2[g][j] += vr[j] * vr[j]; }
              }
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
      // lanes l, l + CH, ... of a wave own the same 8 columns: fixed butterfly inside the wave, then the 4 waves in order
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        cs1[g][j] = wave_strided_sum(cs1[g][j], CH);
        cs2[g][j] = wave_strided_sum(cs2[g][j], CH);
      }
      lds_ordered_accumulate(tid >> 6, 4, lane < CH, [&]() {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          red[col + j] += cs1[g][j];
          red[BN + col + j] += cs2[g][j];
        }
      });
    });
    for (int col = tid; col < BN; col += 256) {
      int n = n0 + col;
      if (n < p.N) {
        p.stats_part[(size_t)bx * 2 * p.N + n] = red[col];
        p.stats_part[(size_t)bx * 2 * p.N + p.N + n] = red[BN + col];
      }
    }
  }
}

// =============================================================================================
// host-side dispatch
// =============================================================================================
// host-side tally of the conv_gemm_kernel family (bench.py, cvh_stream_counters): launches and their algorithmic bytes — the input tensor(s),
// the output and every [M][N] epilogue operand once, 2 (bf16) / 4 (f32) bytes per element; defined in gemm.hip
extern std::atomic<long long> g_cg_launches, g_cg_bytes;

template <typename T, int NF, int BK, int FX, int WP = 0>
static int launch_conv_gemm(const ConvGemmParams& p0, hipStream_t st) {
  constexpr int BM = 128, BN = 32 * NF;
  ConvGemmParams p = p0;
  {
    const int extra = (p.save_pre != nullptr) + (p.actgrad_aux != nullptr) + (p.residual != nullptr) + (FX == 1 && p.e_mode == 1 ? 1 : 0);
    const long long in_rows = (long long)p.B * p.H * p.W;
    g_cg_launches += 1;
    g_cg_bytes += (in_rows * (p.C1 + p.C2) + (long long)p.M * p.N * (1 + extra)) * (long long)sizeof(T);
  }
  p.m_tiles = (p.M + BM - 1) / BM;
  const int n_tiles = (p.N + BN - 1) / BN;
  const int cap = cvh_tune_get(CVH_TUNE_GEMM_GRID);
  int gx = p.m_tiles < cap ? p.m_tiles : cap;
  dim3 grid(gx, n_tiles);
  constexpr int TILE_ELEMS = (BM + BN) * lds_pitch<T>(BK);
  constexpr int STAGE_ELEMS = (WP && lds_pitch<T>(BK) >= stage_pitch<T, NF>()) ? 0 : 4 * 32 * stage_pitch<T, NF>();
  constexpr int MAIN_ELEMS = TILE_ELEMS + STAGE_ELEMS;
  size_t smem = (size_t)MAIN_ELEMS * sizeof(T) + (size_t)(2 + (FX ? 4 : 0)) * BN * sizeof(float);
  auto kern = conv_gemm_kernel<T, NF, BK, FX, WP>;
  static DynSmemAttr attr;  // one instantiation = one static
  if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(kern), smem); e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(kern, grid, dim3(256), smem, st, p);
  CVH_CHECK_LAUNCH();
  return 0;
}

template <typename T, int BK, int FX, int WP = 0>
static int dispatch_conv_gemm_nf(const ConvGemmParams& p, int nf, hipStream_t st) {
  if (WP) {
    switch (nf) {
      case 1: return launch_conv_gemm<T, 1, BK, FX, WP>(p, st);
      case 2: return launch_conv_gemm<T, 2, BK, FX, WP>(p, st);
      case 3: return launch_conv_gemm<T, 3, BK, FX, WP>(p, st);
      case 4: return launch_conv_gemm<T, 4, BK, FX, WP>(p, st);
      default: return launch_conv_gemm<T, 5, BK, FX, WP>(p, st);
    }
  }
  switch (nf) {
    case 1: return launch_conv_gemm<T, 1, BK, FX>(p, st);
    case 2: return launch_conv_gemm<T, 2, BK, FX>(p, st);
    case 3: return launch_conv_gemm<T, 3, BK, FX>(p, st);
    case 4: return launch_conv_gemm<T, 4, BK, FX>(p, st);
    default: return launch_conv_gemm<T, 5, BK, FX>(p, st);
  }
}

// choose the N tiling: fewest padded columns, then fewest tiles
static int choose_nf_capped(int N, int cap) {
  static const int cand[5] = {1, 2, 3, 4, 5};
  int best = cap < 4 ? cap : 4, best_cost = 1 << 30;
  for (int i = 0; i < 5 && cand[i] <= cap; ++i) {
    int bn = 32 * cand[i];
    int tiles = (N + bn - 1) / bn;
    int cost = tiles * bn * 16 + tiles;  // padded width dominates, tile count breaks ties
    if (cost < best_cost) { best_cost = cost; best = cand[i]; }
  }
  return best;
}
static int choose_nf(int N) { return choose_nf_capped(N, 5); }
