// gemm_stream_kernel: pointwise conv / linear GEMMs with a SHORT reduction (64 < K <= 320) under millions of rows — the token linears
// of the MobileViT blocks (d = 96 ... 240: qkv, out-proj, fc1 / fc2 and their dX) and the 1x1 convolutions around them.
// Replaces (reference) LinearLayer.forward / F.linear (cvnets/layers/linear_layer.py:74-91) and nn.Conv2d 1x1 (cvnets/layers/conv_layer.py:254-255)
// on these shapes, forward and input gradient, with the same fused epilogue as conv_gemm_kernel.
//
// Why another GEMM: on these shapes the work is HBM-bound streaming (AI ~ 70 FLOP/B), and conv_gemm's 128-row cooperative tiles spend
// 60-70 % of their wave cycles in s_waitcnt / s_barrier (PMC: SQ_WAIT_ANY) at two waves per SIMD — 190-256 VGPRs for the 96 ... 160-column
// accumulators and the register-staged operands, one exposed memory latency per 64-wide K step, zero-padded K (144 -> 192) and N (144 -> 160).
// Here
//   * the weight tile W[bn x K] is RESIDENT in LDS (loaded once per workgroup), padded to a conflict-free pitch;
//   * a wave owns 16 rows of A at a time and loads them STRAIGHT into MFMA operand registers (16-byte fragment loads, no LDS round trip, no
//     barrier anywhere in the loop); the next 16 rows are requested as soon as the last MFMA pass has consumed the current ones;
//   * the output is produced in 48- (or 32-) column chunks with v_mfma_f32_16x16x32_bf16 on the TRANSPOSED problem (D^T = W A^T), so a lane
//     ends up with 4 consecutive columns of one row: 12 accumulator registers per chunk, 8-byte LDS staging writes; the finished 16 x bn
//     tile leaves as 16-byte pieces, 64 consecutive pieces per store instruction (1 KB of contiguous output when bn == N);
//     N = 96 / 144 / 192 / 240 / 288 / 384 / 432 / 576 / 720 are all multiples of 48: no padded columns;
//   * everything fits 128 VGPRs: one 1024-thread workgroup per CU = 16 waves = 4 per SIMD, each with its own loads in flight.
#include "common.hpp"
#include "cvnets_hip.h"
#include "gemm_params.hpp"

typedef __attribute__((ext_vector_type(4))) float f32x4_t;

#define GS_WAVES 16
#define GS_MT 16

struct GemmStreamGeom {
  int n_tiles;  // workgroup columns
  int bn;       // output columns per workgroup (multiple of the chunk width)
  int nch;      // chunks per workgroup
  int Kp;       // K rounded up to 32
  int m_tiles;  // 16-row tiles
  int gx;       // workgroups along M
  int magic;    // ceil(2^16 / (bn / 8)): piece index -> row by multiply + shift (exact for the < 512 pieces of a tile)
  int rows;     // EM > 0: rows of the partial-statistics buffer the caller allocated (cvh_conv_gemm_grid_rows); rows >= gx are zero-filled
};

#define GS_NPMAX 8  // row-piece passes of the epilogue: 16 rows x bn / 8 pieces / 64 lanes, bn <= 256

// EM (epilogue mode, the BatchNorm links of bnlink.hpp): 0 plain; 1 + column statistics (sum, sumsq) of the stored values -> partial rows;
// 2 BatchNorm-backward epilogue: out = acc * act'(scale * aux + shift), statistics (sum g, sum g * xhat), xhat = invstd * aux - mean * invstd.
// EM > 0 needs bn / 8 to divide 64: a lane then owns the SAME 8 columns in every pass of every tile and its statistics stay in registers.
template <int FW, int NKMAX, int EM = 0>
__global__ __launch_bounds__(64 * GS_WAVES) void gemm_stream_kernel(ConvGemmParams p, GemmStreamGeom g) {
  constexpr int CW = 16 * FW;           // chunk width
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int PK = g.Kp + CVH_M16_PAD;    // weight pitch: Kp / 8 is a multiple of 4, + 2 chunks = 2 (mod 4): conflict-free fragment reads (common.hpp)
  const int SP = g.bn + 8;              // staging pitch (elements): rows stay 16-byte aligned
  bf16_t* Ws = reinterpret_cast<bf16_t*>(smem_raw);                       // [bn][PK]
  float* bias_s = reinterpret_cast<float*>(Ws + (size_t)g.bn * PK);       // [bn]
  float* red = bias_s + g.bn;                                             // EM > 0: [2][bn] column sums; EM == 2: + est[4][bn]
  float* est = red + 2 * g.bn;
  bf16_t* stg_all = reinterpret_cast<bf16_t*>(bias_s + g.bn + (EM ? 2 * g.bn : 0) + (EM == 2 ? 4 * g.bn : 0));  // [waves][16][SP]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  bf16_t* stg = stg_all + wave * (GS_MT * SP);
  // XCD-contiguous order with the column tile fastest: the workgroups that stream the same rows run on one XCD (one L2 fetch of A)
  const int lb = xcd_chunk_id((int)blockIdx.x, (int)gridDim.x);
  const int nt = lb % g.n_tiles, xb = lb / g.n_tiles;
  const int n0 = nt * g.bn;
  const int K = p.Ktot, N = p.N, M = p.M;
  const bf16_t* __restrict__ A = reinterpret_cast<const bf16_t*>(p.src1);
  const bf16_t* __restrict__ Wg = reinterpret_cast<const bf16_t*>(p.wgt);
  bf16_t* __restrict__ out = reinterpret_cast<bf16_t*>(p.out);
  const int nk = g.Kp / 32;

  {  // weight tile -> LDS (zero beyond N / K), bias -> LDS
    const int kch = g.Kp / 8;
    for (int i = tid; i < g.bn * kch; i += 64 * GS_WAVES) {
      const int r = i / kch, kc = (i - r * kch) * 8;
      V8<bf16_t> v = v8_zero<bf16_t>();
      if (n0 + r < N && kc < K) v = v8_load<bf16_t>(Wg + (size_t)(n0 + r) * K + kc);
      v8_store<bf16_t>(Ws + r * PK + kc, v);
    }
    for (int i = tid; i < g.bn; i += 64 * GS_WAVES) bias_s[i] = (p.bias != nullptr && n0 + i < N) ? p.bias[n0 + i] : 0.f;
    if (EM) {
      for (int i = tid; i < 2 * g.bn; i += 64 * GS_WAVES) red[i] = 0.f;
    }
    if (EM == 2) {  // (invstd, -mean * invstd, scale, shift) of the BatchNorm being back-propagated through
      for (int i = tid; i < g.bn; i += 64 * GS_WAVES) {
        const int n = n0 + i;
        const bool ok = n < N;
        const float mu = ok ? p.e_stats[n] : 0.f, is = ok ? p.e_stats[N + n] : 0.f;
        est[i] = is;
        est[g.bn + i] = -mu * is;
        est[2 * g.bn + i] = ok ? p.e_stats[2 * N + n] : 0.f;
        est[3 * g.bn + i] = ok ? p.e_stats[3 * N + n] : 0.f;
      }
    }
  }
  __syncthreads();  // the only workgroup barrier

  DropKey dkey = {0u, 0u};
  if (p.drop_p > 0.f) dkey = drop_key(*p.seed, p.stream_id, p.drop_p);
  const float inv_keep = p.drop_p > 0.f ? 1.0f / (1.0f - p.drop_p) : 1.0f;

  // The epilogue walks the wave's staged 16 x bn tile as 16-byte pieces, 64 consecutive pieces per pass: with bn == N the tile is ONE
  // contiguous 32 bn-byte range of the output and every store instruction writes 1 KB of consecutive bytes, whatever the row alignment.
  // piece index -> (row, piece in row) by a multiply-shift with a host-side constant: no run-time division, no per-pass registers.
  const int ppr = g.bn / 8, npieces = GS_MT * ppr;

  // EM: this lane's 8 columns (the same in every pass) and their running statistics
  float cs1[EM ? 8 : 1], cs2[EM ? 8 : 1];
  const int my_ch = lane % ppr;
  if (EM) {
#pragma unroll
    for (int j = 0; j < 8; ++j) cs1[j] = cs2[j] = 0.f;
  }

  const int l15 = lane & 15, l4 = lane >> 4;
  // fragment loads of 16 rows: lane (l15, l4) reads 16 bytes of row m0 + l15 at column 32 ks + 8 l4; chunks at or beyond K meet zero weights and
  // read column 0 instead (a valid address holding finite data)
  // DB (K <= 192): two operand register sets — the next 16 rows are requested BEFORE this tile's MFMAs and have the whole tile to arrive;
  // otherwise one set, re-requested as soon as the last MFMA pass has consumed it (they then fly under the epilogue only).
  constexpr bool DB = NKMAX <= 2 || (NKMAX <= 6 && EM == 0);  // the statistics epilogues need the registers themselves
  bf16x8_t a0[NKMAX], a1[DB ? NKMAX : 1];
  auto load_a = [&](bf16x8_t* a, int tile) __attribute__((always_inline)) {
    int row = tile * GS_MT + l15;
    row = row < M ? row : M - 1;
    const bf16_t* rp = A + (size_t)row * K;
#pragma unroll
    for (int ks = 0; ks < NKMAX; ++ks) {
      if (ks < nk) {
        const int col = 32 * ks + 8 * l4;
        a[ks] = *reinterpret_cast<const bf16x8_t*>(rp + (col < K ? col : 0));
      }
    }
  };

  const int tstride = g.gx * GS_WAVES;
  int tile = xb * GS_WAVES + wave;
  const int npass = (npieces + 63) / 64;
  // one 16-row tile: `a` holds its rows; `an` receives the next tile's (DB) — or `a` itself is refilled after the last MFMA pass
  auto do_tile = [&](bf16x8_t* a, bf16x8_t* an) __attribute__((always_inline)) {
    const int m0 = tile * GS_MT;
    const int next = tile + tstride;
    if (DB && next < g.m_tiles) load_a(an, next);
    for (int c = 0; c < g.nch; ++c) {
      f32x4_t acc[FW];
#pragma unroll
      for (int f = 0; f < FW; ++f) acc[f] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      const bf16_t* wrow = Ws + (size_t)(c * CW + l15) * PK + 8 * l4;
#pragma unroll
      for (int ks = 0; ks < NKMAX; ++ks) {
        if (ks < nk) {
#pragma unroll
          for (int f = 0; f < FW; ++f) {
            const bf16x8_t w = *reinterpret_cast<const bf16x8_t*>(wrow + (size_t)(16 * f) * PK + 32 * ks);
            acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, a[ks], acc[f], 0, 0, 0);  // D^T[n][m] += W[n][k] A[m][k]
          }
        }
      }
      if (!DB && c == g.nch - 1 && next < g.m_tiles) load_a(a, next);
      // (+bias) -> bf16 -> the wave's staging rows: a lane holds 4 consecutive columns of one row per fragment (8-byte writes)
#pragma unroll
      for (int f = 0; f < FW; ++f) {
        const float4 b = *reinterpret_cast<const float4*>(bias_s + c * CW + 16 * f + 4 * l4);
        uint2 pk;
        pk.x = f2bf_pk(acc[f][0] + b.x, acc[f][1] + b.y);
        pk.y = f2bf_pk(acc[f][2] + b.z, acc[f][3] + b.w);
        *reinterpret_cast<uint2*>(stg + l15 * SP + c * CW + 16 * f + 4 * l4) = pk;
      }
    }
    wave_lds_sync();
    // ---- epilogue of the 16 x bn tile: 16-byte pieces with the fused tail ----
#pragma unroll(EM ? 1 : 2)
    for (int pass = 0; pass < npass; ++pass) {
      const int idx = lane + 64 * pass;
      const int row = (idx * g.magic) >> 16, ch = idx - row * ppr;
      const int m = m0 + row, n = n0 + ch * 8;
      if (idx < npieces && m < M && n < N) {
        const size_t o = (size_t)m * N + n;
        V8<bf16_t> pv = v8_load<bf16_t>(stg + row * SP + ch * 8);
        if (p.save_pre) v8_store<bf16_t>(reinterpret_cast<bf16_t*>(p.save_pre) + o, pv);
        float v[8];
        v8_unpack(pv, v);
        if (p.act != CVH_ACT_NONE) act_fwd8(v, p.act);
        float xh[EM == 2 ? 8 : 1];
        if (EM == 2) {
          // the four per-channel vectors come from LDS where they are used (in registers they would cost 32 VGPRs of the 128)
          float ax[8], yh[8], ka[8], kb[8];
          v8_unpack(v8_load<bf16_t>(reinterpret_cast<const bf16_t*>(p.e_aux) + o), ax);
          const float* ev = est + ch * 8;
          *reinterpret_cast<float4*>(ka) = *reinterpret_cast<const float4*>(ev + 2 * g.bn);
          *reinterpret_cast<float4*>(ka + 4) = *reinterpret_cast<const float4*>(ev + 2 * g.bn + 4);
          *reinterpret_cast<float4*>(kb) = *reinterpret_cast<const float4*>(ev + 3 * g.bn);
          *reinterpret_cast<float4*>(kb + 4) = *reinterpret_cast<const float4*>(ev + 3 * g.bn + 4);
#pragma unroll
          for (int j = 0; j < 8; ++j) yh[j] = ax[j] * ka[j] + kb[j];
          act_grad8_mul(v, yh, p.e_act);
          *reinterpret_cast<float4*>(ka) = *reinterpret_cast<const float4*>(ev);
          *reinterpret_cast<float4*>(ka + 4) = *reinterpret_cast<const float4*>(ev + 4);
          *reinterpret_cast<float4*>(kb) = *reinterpret_cast<const float4*>(ev + g.bn);
          *reinterpret_cast<float4*>(kb + 4) = *reinterpret_cast<const float4*>(ev + g.bn + 4);
#pragma unroll
          for (int j = 0; j < 8; ++j) xh[j] = ax[j] * ka[j] + kb[j];
        }
        if (p.actgrad_aux) {
          float ax[8];
          v8_unpack(v8_load<bf16_t>(reinterpret_cast<const bf16_t*>(p.actgrad_aux) + o), ax);
          act_grad8_mul(v, ax, p.actgrad_act);
        }
        if (p.drop_p > 0.f) dropout_scale8(dkey, o, inv_keep, v);
        if (p.residual) {
          float rr[8];
          v8_unpack(v8_load<bf16_t>(reinterpret_cast<const bf16_t*>(p.residual) + o), rr);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] += rr[j];
        }
        V8<bf16_t> ov;
        v8_pack(v, ov);
        v8_store<bf16_t>(out + o, ov);
        if (EM) {
          float vr[8];
          v8_unpack(ov, vr);  // statistics of the values as stored
#pragma unroll
          for (int j = 0; j < 8; ++j) { cs1[j] += vr[j]; cs2[j] += vr[j] * (EM == 2 ? xh[j] : vr[j]); }
        }
      }
    }
    wave_lds_sync();  // staging consumed before the next tile overwrites it
    tile = next;
  };
  if (tile < g.m_tiles) load_a(a0, tile);
  while (tile < g.m_tiles) {
    do_tile(a0, DB ? a1 : a0);
    if (DB) {
      if (tile >= g.m_tiles) break;
      do_tile(a1, a0);
    }
  }
  if (EM) {  // workgroup totals -> partial row xb of this column tile; the caller's surplus rows are zeroed
    // lanes my_ch, my_ch + ppr, ... of a wave own the same 8 columns: fixed butterfly inside the wave, then the 16 waves in order
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      cs1[j] = wave_strided_sum(cs1[j], ppr);
      cs2[j] = wave_strided_sum(cs2[j], ppr);
    }
    lds_ordered_accumulate(wave, GS_WAVES, lane < ppr, [&]() {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        red[my_ch * 8 + j] += cs1[j];
        red[g.bn + my_ch * 8 + j] += cs2[j];
      }
    });
    for (int i = tid; i < 2 * g.bn; i += 64 * GS_WAVES) {
      const int which = i / g.bn, n = n0 + (i - which * g.bn);
      if (n < N) {
        p.stats_part[((size_t)xb * 2 + which) * N + n] = red[i];
        for (int r = g.gx + xb; r < g.rows; r += g.gx) p.stats_part[((size_t)r * 2 + which) * N + n] = 0.f;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------------------
static bool gemm_stream_geom(const ConvGemmParams& p, int em, GemmStreamGeom& g, int& fw, size_t& smem) {
  const int K = p.Ktot, N = p.N;
  // em > 0: power-of-two column tiles of 32-column chunks (bn / 8 must divide 64); else exact 48-column chunks where N allows
  fw = em ? ((N % 32) == 0 ? 2 : 0) : ((N % 48) == 0 ? 3 : ((N % 32) == 0 ? 2 : 0));
  if (fw == 0) return false;
  const int cw = 16 * fw;
  g.Kp = (K + 31) / 32 * 32;
  const int pk = g.Kp + CVH_M16_PAD;
  auto bytes = [&](int bn) { return (size_t)bn * pk * 2 + (size_t)bn * 4 * (1 + (em ? 2 : 0) + (em == 2 ? 4 : 0)) + (size_t)GS_WAVES * GS_MT * (bn + 8) * 2; };
  // widest column tile (a multiple of the chunk width that divides N) whose weights + staging fit the 160 KB of a CU
  int best = 0;
  for (int bn = cw; bn <= N && bn <= 256; bn += cw) {
    if (N % bn) continue;
    if (em && (bn & (bn - 1))) continue;
    if (bytes(bn) <= 156 * 1024) best = bn;
  }
  if (best == 0) return false;
  g.bn = best;
  g.nch = best / cw;
  g.n_tiles = N / best;
  g.m_tiles = (p.M + GS_MT - 1) / GS_MT;
  int gx = 256 / g.n_tiles;  // one 16-wave workgroup per CU
  if (gx < 1) gx = 1;
  const int need_x = (g.m_tiles + GS_WAVES - 1) / GS_WAVES;
  if (gx > need_x) gx = need_x;
  g.gx = gx;
  g.magic = (65536 + best / 8 - 1) / (best / 8);
  g.rows = 0;
  smem = bytes(best);
  return true;
}

bool gemm_stream_eligible(const ConvGemmParams& p) {
  if (cvh_tune_get(CVH_TUNE_NO_STREAM_GEMM)) return false;
  const bool linear = p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0 && p.C2 == 0 && p.src2 == nullptr;
  if (!linear || p.stats_part != nullptr || p.sc_s != 0) return false;
  if (p.Ktot <= 64 || p.Ktot > 320 || (p.Ktot % 8) != 0 || p.M < 32768) return false;
  GemmStreamGeom g;
  int fw;
  size_t smem;
  return gemm_stream_geom(p, 0, g, fw, smem);
}

// host-side tally of what went through this kernel family (bench.py: launches per step and their ALGORITHMIC bytes — every operand and
// result tensor once: A, out, and the epilogue's aux / residual / saved pre-activation where present; weights and statistics neglected)
static std::atomic<long long> g_gs_launches{0}, g_gs_bytes{0};  // updated from autograd worker threads
void gemm_stream_counters(int reset, long long* out2) {  // C entry point: cvh_stream_counters (gemm.hip)
  if (out2 != nullptr) { out2[0] = g_gs_launches; out2[1] = g_gs_bytes; }
  if (reset) { g_gs_launches = 0; g_gs_bytes = 0; }
}

template <int FW, int NKMAX, int EM> static int launch_gs(const ConvGemmParams& p, const GemmStreamGeom& g, size_t smem, hipStream_t st) {
  auto kern = gemm_stream_kernel<FW, NKMAX, EM>;
  {
    const int extra = (p.save_pre != nullptr) + (p.actgrad_aux != nullptr) + (p.residual != nullptr) + (EM == 2 ? 1 : 0);
    g_gs_launches += 1;
    g_gs_bytes += (long long)p.M * ((long long)p.Ktot + (long long)p.N * (1 + extra)) * 2;
  }
  static DynSmemAttr attr;  // one instantiation = one static
  if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(kern), smem); e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(kern, dim3(g.gx * g.n_tiles), dim3(64 * GS_WAVES), smem, st, p, g);
  CVH_CHECK_LAUNCH();
  return 0;
}

int launch_gemm_stream(const ConvGemmParams& p, hipStream_t st) {
  GemmStreamGeom g;
  int fw;
  size_t smem;
  if (!gemm_stream_geom(p, 0, g, fw, smem)) return -2;
  const bool small_k = g.Kp <= 192;  // operand registers: 4 per 32-wide K step (24 / 40 of the 128-register budget)
  if (fw == 3) return small_k ? launch_gs<3, 6, 0>(p, g, smem, st) : launch_gs<3, 10, 0>(p, g, smem, st);
  return small_k ? launch_gs<2, 6, 0>(p, g, smem, st) : launch_gs<2, 10, 0>(p, g, smem, st);
}

// BatchNorm-link GEMMs with a PLAIN A operand (cvh_pw_gemm_bn, gemm_fx.hip): forward expansion convs (statistics epilogue) and the
// projection dX with the BatchNorm-backward epilogue.  `rows` = rows of the caller's partial-statistics buffer.
static int gs_em(const ConvGemmParams& p) { return p.e_mode == 1 ? 2 : (p.stats_part != nullptr ? 1 : 0); }
bool gemm_stream_fx_eligible(const ConvGemmParams& p) {
  if (cvh_tune_get(CVH_TUNE_NO_STREAM_GEMM)) return false;
  if (p.a_xf.mode != 0 || p.sc_s != 0 || p.Ktot > 192 || (p.Ktot % 8) != 0 || p.M < 32768) return false;
  if (p.e_mode == 1 && p.stats_part == nullptr) return false;
  const int em = gs_em(p);
  if (em == 0) return false;
  if (em == 2 && p.Ktot > 64) return false;  // the BatchNorm-backward epilogue next to 6 K steps of operand registers does not fit 128 VGPRs
  GemmStreamGeom g;
  int fw;
  size_t smem;
  return gemm_stream_geom(p, em, g, fw, smem);
}
int launch_gemm_stream_fx(const ConvGemmParams& p, int rows, hipStream_t st) {
  GemmStreamGeom g;
  int fw;
  size_t smem;
  const int em = gs_em(p);
  if (!gemm_stream_geom(p, em, g, fw, smem) || fw != 2) return -2;
  if (rows < g.gx) {  // never more workgroup rows than the caller allocated
    g.gx = rows;
  }
  g.rows = rows;
  const bool tiny_k = g.Kp <= 64;
  if (em == 1) return tiny_k ? launch_gs<2, 2, 1>(p, g, smem, st) : launch_gs<2, 6, 1>(p, g, smem, st);
  return tiny_k ? launch_gs<2, 2, 2>(p, g, smem, st) : -2;
}
