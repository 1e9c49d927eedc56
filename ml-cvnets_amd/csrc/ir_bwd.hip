// ir_exp_bwd_kernel: both gradient products of the 1x1 EXPANSION conv of an InvertedResidual block (cvnets/modules/mobilenetv2.py:180-193;
// the backward of nn.Conv2d 1x1 behind a train-mode BatchNorm, cvnets/layers/conv_layer.py:254-255) in ONE pass over the 4x-wide gradient
// g = dz * act'(bn(y)) that the depthwise backward kernel leaves behind:
//
//     dX = [g | x] Wcat^T + bias (+ residual gradient)       (the BatchNorm-linked input gradient: cvh_bn_dx_weights, bnlink.hip)
//     P  = g^T x                                             (the raw weight-gradient product that cvh_bn_dw_combine finishes)
//
// Separately (conv_gemm over the channel concat + gemm_tn128 / gemm_tn_skinny) each product streams g from HBM: 2 x 2.1 GB per layer_2
// block at batch 1024.  Here a workgroup stages a 64-row tile of g and x in LDS once (double-buffered, prefetched through registers) and
// feeds both from it: dX on the transposed problem (D^T = Wcat P^T, Wcat resident in LDS: a lane ends with 4 consecutive channels of one
// row), P contracting over the ROWS with transpose-read operands (ds_read_b64_tr_b16), its [hid][Cin] accumulators in registers for the
// whole persistent loop; one partial P per workgroup leaves at the end (summed by cvh_sum_partials).
// HBM stream: algorithmic bytes = g + x + dX (+ residual) per row.
#include "common.hpp"
#include "cvnets_hip.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef short tr_v4s __attribute__((ext_vector_type(4)));

constexpr int IB_TM = 64, IB_THREADS = 512;

struct IrExpBwdParams {
  const bf16_t* g;      // [M][hid]
  const bf16_t* x;      // [M][Cin]
  const bf16_t* wcat;   // [Cin][hid + Cin]
  const float* bias;    // [Cin]
  const bf16_t* res;    // [M][Cin] or nullptr
  bf16_t* dx;           // [M][Cin]
  float* ppart;         // [grid][hid][Cin]
  float* spart;         // [grid][2][Cin] or nullptr: column sums of dX and of dX * x over the workgroup's rows (dX as stored)
  int M, hid, Cin, ntiles;
};

__device__ __forceinline__ bf16x8_t tr_frag8(const bf16_t* lo, const bf16_t* hi) {
  const tr_v4s a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tr_v4s*)(lo));
  const tr_v4s b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tr_v4s*)(hi));
  typedef short v8s __attribute__((ext_vector_type(8)));
  return __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}

// HBK = hid / 16 (16-row blocks of P, dealt round-robin to the 8 waves), CB = Cin / 16
template <int HBK, int CB>
__global__ __launch_bounds__(IB_THREADS) void ir_exp_bwd_kernel(IrExpBwdParams p) {
  constexpr int HID = 16 * HBK, CIN = 16 * CB, HPW = (HBK + 7) / 8;
  constexpr int XK = (CIN + 31) / 32 * 32;          // x part of the contraction, whole 32-wide steps (columns >= CIN are zero)
  constexpr int WP = HID + XK + CVH_M16_PAD;         // pitches (elements): whole 64-byte K steps + 2 chunks: conflict-free fragment reads (common.hpp)
  constexpr int GP = HID + CVH_M16_PAD, XP = XK + CVH_M16_PAD;
  constexpr int TILE = IB_TM * GP + IB_TM * XP;      // elements of one (g, x) tile pair
  constexpr int GIT = IB_TM * (HID / 8) / IB_THREADS;  // 16-byte g chunks per thread and tile
  constexpr int NBLK = (4 * CB + 7) / 8;             // dX output blocks (16 rows x 16 channels) per wave
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  bf16_t* Ws = reinterpret_cast<bf16_t*>(smem_raw);          // [CIN][WP]
  bf16_t* tiles = Ws + CIN * WP;                              // 2 x TILE
  float* bias_s = reinterpret_cast<float*>(tiles + 2 * TILE);  // [CIN]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;

  // weights, bias, zero padding of the x tiles
  for (int i = tid; i < CIN * (WP / 8); i += IB_THREADS) {
    const int r = i / (WP / 8), kc = (i - r * (WP / 8)) * 8;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (kc < HID + CIN) v = *reinterpret_cast<const uint4*>(p.wcat + (size_t)r * (HID + CIN) + kc);
    *reinterpret_cast<uint4*>(Ws + r * WP + kc) = v;
  }
  for (int i = tid; i < CIN; i += IB_THREADS) bias_s[i] = p.bias[i];
  for (int i = tid; i < 2 * TILE / 8; i += IB_THREADS) reinterpret_cast<uint4*>(tiles)[i] = make_uint4(0, 0, 0, 0);
  __syncthreads();

  uint4 gr[GIT], xr;
  auto load_tile = [&](int t) __attribute__((always_inline)) {
    const int row0 = t * IB_TM;
#pragma unroll
    for (int it = 0; it < GIT; ++it) {
      const int i = tid + it * IB_THREADS;
      const int r = i / (HID / 8), ck = i - r * (HID / 8);
      gr[it] = make_uint4(0, 0, 0, 0);
      if (row0 + r < p.M) gr[it] = *reinterpret_cast<const uint4*>(p.g + (size_t)(row0 + r) * HID + ck * 8);
    }
    xr = make_uint4(0, 0, 0, 0);
    if (tid < IB_TM * (CIN / 8)) {
      const int r = tid / (CIN / 8), ck = tid - r * (CIN / 8);
      if (row0 + r < p.M) xr = *reinterpret_cast<const uint4*>(p.x + (size_t)(row0 + r) * CIN + ck * 8);
    }
  };
  auto store_tile = [&](bf16_t* gt) __attribute__((always_inline)) {
    bf16_t* xt = gt + IB_TM * GP;
#pragma unroll
    for (int it = 0; it < GIT; ++it) {
      const int i = tid + it * IB_THREADS;
      const int r = i / (HID / 8), ck = i - r * (HID / 8);
      *reinterpret_cast<uint4*>(gt + r * GP + ck * 8) = gr[it];
    }
    if (tid < IB_TM * (CIN / 8)) {
      const int r = tid / (CIN / 8), ck = tid - r * (CIN / 8);
      *reinterpret_cast<uint4*>(xt + r * XP + ck * 8) = xr;
    }
  };

  f32x4_t pacc[HPW][CB];
#pragma unroll
  for (int i = 0; i < HPW; ++i)
#pragma unroll
    for (int c = 0; c < CB; ++c) pacc[i][c] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  // column sums of the stored dX and of dX * x (x = this block's input = the previous block's output): what the BatchNorm backward of the
  // PREVIOUS block's projection needs of its incoming gradient (sum dout, sum dout * xhat with xhat = (x - beta) / gamma there), so that block
  // does not read dout and its y3 once more for them.  A wave keeps the same channels in every tile: per-lane accumulators.
  const bool want_s = p.spart != nullptr;
  float ss1[NBLK][4], ss2[NBLK][4];
#pragma unroll
  for (int i = 0; i < NBLK; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) ss1[i][e] = ss2[i][e] = 0.f;

  int t = blockIdx.x, cur = 0;
  if (t < p.ntiles) {
    load_tile(t);
    store_tile(tiles);
  }
  __syncthreads();
  for (; t < p.ntiles; t += gridDim.x) {
    const int tn = t + gridDim.x;
    if (tn < p.ntiles) load_tile(tn);
    const bf16_t* gt = tiles + cur * TILE;
    const bf16_t* xt = gt + IB_TM * GP;
    const int row0 = t * IB_TM;

    // ---- dX^T[c][m] = Wcat[c][:] . [g | x][m][:]  (+ bias, + residual) ----
    if (wave * NBLK < 4 * CB) {
      const int j0 = wave * NBLK, mb = j0 / CB, cb0 = j0 - mb * CB;  // the wave's blocks share their 16 rows
      f32x4_t acc[NBLK];
#pragma unroll
      for (int i = 0; i < NBLK; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      const bf16_t* arow = gt + (16 * mb + l15) * GP + 8 * l4;
      const bf16_t* wrow = Ws + (16 * cb0 + l15) * WP + 8 * l4;
#pragma unroll
      for (int ks = 0; ks < HID / 32; ++ks) {
        const bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(arow + 32 * ks);
#pragma unroll
        for (int i = 0; i < NBLK; ++i) {
          const bf16x8_t w = *reinterpret_cast<const bf16x8_t*>(wrow + (size_t)(16 * i) * WP + 32 * ks);
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, a, acc[i], 0, 0, 0);
        }
      }
      const bf16_t* xrow = xt + (16 * mb + l15) * XP + 8 * l4;
#pragma unroll
      for (int ks = 0; ks < XK / 32; ++ks) {
        const bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(xrow + 32 * ks);
#pragma unroll
        for (int i = 0; i < NBLK; ++i) {
          const bf16x8_t w = *reinterpret_cast<const bf16x8_t*>(wrow + (size_t)(16 * i) * WP + HID + 32 * ks);
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, a, acc[i], 0, 0, 0);
        }
      }
      const int m = row0 + 16 * mb + l15;
      if (m < p.M) {
#pragma unroll
        for (int i = 0; i < NBLK; ++i) {
          const int c = 16 * (cb0 + i) + 4 * l4;
          const float4 b = *reinterpret_cast<const float4*>(bias_s + c);
          float v[4] = {acc[i][0] + b.x, acc[i][1] + b.y, acc[i][2] + b.z, acc[i][3] + b.w};
          if (p.res) {
            const uint2 rr = *reinterpret_cast<const uint2*>(p.res + (size_t)m * CIN + c);
            v[0] += bf2f((uint16_t)(rr.x & 0xffff)); v[1] += bf2f((uint16_t)(rr.x >> 16));
            v[2] += bf2f((uint16_t)(rr.y & 0xffff)); v[3] += bf2f((uint16_t)(rr.y >> 16));
          }
          const uint2 o = make_uint2(f2bf_pk(v[0], v[1]), f2bf_pk(v[2], v[3]));
          *reinterpret_cast<uint2*>(p.dx + (size_t)m * CIN + c) = o;
          if (want_s) {
            const uint2 xx = *reinterpret_cast<const uint2*>(xt + (16 * mb + l15) * XP + c);
            const float d0 = bf2f((uint16_t)(o.x & 0xffff)), d1 = bf2f((uint16_t)(o.x >> 16)), d2 = bf2f((uint16_t)(o.y & 0xffff)),
                        d3 = bf2f((uint16_t)(o.y >> 16));
            ss1[i][0] += d0; ss1[i][1] += d1; ss1[i][2] += d2; ss1[i][3] += d3;
            ss2[i][0] += d0 * bf2f((uint16_t)(xx.x & 0xffff)); ss2[i][1] += d1 * bf2f((uint16_t)(xx.x >> 16));
            ss2[i][2] += d2 * bf2f((uint16_t)(xx.y & 0xffff)); ss2[i][3] += d3 * bf2f((uint16_t)(xx.y >> 16));
          }
        }
      }
    }

    // ---- P[h][c] += sum_m g[m][h] x[m][c]: lane (l15, l4) supplies rows 8*l4 + (l15 >> 2) (+4) of the 32-row step, columns 4*(l15 & 3)..+3 ----
#pragma unroll
    for (int ks = 0; ks < IB_TM / 32; ++ks) {
      const int mlo = 32 * ks + 8 * l4 + (l15 >> 2), col = 4 * (l15 & 3);
      bf16x8_t bx[CB];
#pragma unroll
      for (int c = 0; c < CB; ++c) bx[c] = tr_frag8(xt + mlo * XP + 16 * c + col, xt + (mlo + 4) * XP + 16 * c + col);
#pragma unroll
      for (int i = 0; i < HPW; ++i) {
        const int hb = wave + 8 * i;
        if (hb >= HBK) continue;  // wave-uniform (hid = 64: four blocks for eight waves)
        const bf16x8_t ag = tr_frag8(gt + mlo * GP + 16 * hb + col, gt + (mlo + 4) * GP + 16 * hb + col);
#pragma unroll
        for (int c = 0; c < CB; ++c) pacc[i][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ag, bx[c], pacc[i][c], 0, 0, 0);
      }
    }

    if (tn < p.ntiles) store_tile(tiles + (cur ^ 1) * TILE);
    __syncthreads();
    cur ^= 1;
  }

  if (want_s) {  // (the loop ended with a barrier: the tiles are free) rows of a wave: lanes l15; waves of one channel block: fixed order
    float* red = reinterpret_cast<float*>(tiles);  // [8 waves][2][NBLK * 16]
#pragma unroll
    for (int i = 0; i < NBLK; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float a = ss1[i][e], b = ss2[i][e];
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) {
          a += __shfl_xor(a, m, 64);
          b += __shfl_xor(b, m, 64);
        }
        if (l15 == 0) {
          red[(wave * 2 + 0) * (NBLK * 16) + 16 * i + 4 * l4 + e] = a;
          red[(wave * 2 + 1) * (NBLK * 16) + 16 * i + 4 * l4 + e] = b;
        }
      }
    __syncthreads();
    if (tid < 2 * CIN) {
      const int k = tid / CIN, c = tid - k * CIN, cb = c >> 4;
      float sum = 0.f;
      for (int w = 0; w < 8; ++w) {
        const int j0 = w * NBLK;
        if (j0 >= 4 * CB) break;
        const int mb = j0 / CB, cb0 = j0 - mb * CB;
        if (cb >= cb0 && cb < cb0 + NBLK) sum += red[(w * 2 + k) * (NBLK * 16) + 16 * (cb - cb0) + (c & 15)];
      }
      p.spart[((size_t)blockIdx.x * 2 + k) * CIN + c] = sum;
    }
  }

  float* __restrict__ out = p.ppart + (size_t)blockIdx.x * HID * CIN;
#pragma unroll
  for (int i = 0; i < HPW; ++i) {
    if (wave + 8 * i >= HBK) continue;
#pragma unroll
    for (int c = 0; c < CB; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) out[(size_t)(16 * (wave + 8 * i) + 4 * l4 + e) * CIN + 16 * c + l15] = pacc[i][c][e];
  }
}

template <int HBK, int CB> size_t ir_exp_smem() {
  constexpr int HID = 16 * HBK, CIN = 16 * CB, XK = (CIN + 31) / 32 * 32;
  return ((size_t)CIN * (HID + XK + CVH_M16_PAD) + 2 * ((size_t)IB_TM * (HID + CVH_M16_PAD) + (size_t)IB_TM * (XK + CVH_M16_PAD))) * 2 + (size_t)CIN * 4;
}

bool ir_exp_shape(int hid, int Cin) { return (hid == 64 && Cin == 16) || (hid == 128 && Cin == 32) || (hid == 256 && Cin == 64); }
// workgroups per CU that LDS and registers leave room for (a narrow tile moves few bytes: more workgroups keep enough loads in flight)
int ir_exp_occupancy(int hid) { return hid == 64 ? 4 : (hid == 128 ? 2 : 1); }

}  // namespace

/* partial rows of cvh_ir_exp_bwd for M rows (0: shape not covered — use cvh_conv_gemm + cvh_gemm_dw) */
extern "C" int cvh_ir_exp_bwd_rows(long long M, int hid, int Cin) {
  if (M < 65536 || M > 0x7fffffffLL / 4 || !ir_exp_shape(hid, Cin)) return 0;
  const long long nt = (M + IB_TM - 1) / IB_TM;
  const int wgs = 256 * ir_exp_occupancy(hid);
  return nt < wgs ? (int)nt : wgs;
}

extern "C" int cvh_ir_exp_bwd_s(int dtype, const void* g, const void* x, const void* wcat, const float* bias, const void* residual, void* dx,
                                float* p_part, float* s_part, long long M, int hid, int Cin, void* stream);
extern "C" int cvh_ir_exp_bwd(int dtype, const void* g, const void* x, const void* wcat, const float* bias, const void* residual, void* dx,
                              float* p_part, long long M, int hid, int Cin, void* stream) {
  return cvh_ir_exp_bwd_s(dtype, g, x, wcat, bias, residual, dx, p_part, nullptr, M, hid, Cin, stream);
}
extern "C" int cvh_ir_exp_bwd_s(int dtype, const void* g, const void* x, const void* wcat, const float* bias, const void* residual, void* dx,
                                float* p_part, float* s_part, long long M, int hid, int Cin, void* stream) {
  if (dtype != CVH_DT_BF16) return -1;
  const int rows = cvh_ir_exp_bwd_rows(M, hid, Cin);
  if (rows <= 0) return -2;
  IrExpBwdParams p;
  p.g = reinterpret_cast<const bf16_t*>(g); p.x = reinterpret_cast<const bf16_t*>(x); p.wcat = reinterpret_cast<const bf16_t*>(wcat);
  p.bias = bias; p.res = reinterpret_cast<const bf16_t*>(residual); p.dx = reinterpret_cast<bf16_t*>(dx); p.ppart = p_part; p.spart = s_part;
  p.M = (int)M; p.hid = hid; p.Cin = Cin; p.ntiles = (int)((M + IB_TM - 1) / IB_TM);
  cvh_family_tally(3, (long long)M * (hid + Cin * (residual != nullptr ? 3 : 2)) * 2);
  hipStream_t st = (hipStream_t)stream;
#define IB_LAUNCH(HBK_, CB_)                                                                                                             \
  do {                                                                                                                                   \
    const size_t smem = ir_exp_smem<HBK_, CB_>();                                                                                        \
    static DynSmemAttr attr;                                                                                                             \
    if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(ir_exp_bwd_kernel<HBK_, CB_>), smem); e != hipSuccess) return (int)e;  \
    hipLaunchKernelGGL((ir_exp_bwd_kernel<HBK_, CB_>), dim3(rows), dim3(IB_THREADS), smem, st, p);                                       \
  } while (0)
  if (hid == 64) IB_LAUNCH(4, 1);
  else if (hid == 128) IB_LAUNCH(8, 2);
  else IB_LAUNCH(16, 4);
#undef IB_LAUNCH
  CVH_CHECK_LAUNCH();
  return 0;
}
