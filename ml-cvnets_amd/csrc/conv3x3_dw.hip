// conv3x3_dw_kernel: weight gradient of the 3x3 stride-1 pad-1 convolutions of the MobileViT blocks (local_rep.conv_3x3 and the fusion conv
// over cat(res, fm): cvnets/modules/mobilevit_block.py:102-148,269-288; the weight half of nn.Conv2d's backward, cvnets/layers/conv_layer.py:254-255)
//
//     dW[n][tap][c] = sum_p dY[p][n] * X[p + delta(tap)][c]
//
// The im2col dW GEMM (gemm_tn_kernel<.., 0, 0>) reads X once per tap — nine passes over the input map (measured 966 us for the layer_3
// fusion conv: 0.6 GB of operands).  Here a workgroup holds the 10 x 18 halo image of an 8 x 16 pixel tile (one channel slab) and the
// tile's dY rows in LDS and forms all nine taps from them: X and dY are read once per slab.  The contraction runs over PIXELS, so both
// MFMA operands are "8 consecutive pixels of one column" of a row-major LDS tile: gathered with the gfx950 LDS transpose read
// (ds_read_b64_tr_b16), each lane supplying the address of its own pixel row — for tap (kh, kw) simply the halo row of pixel p + (kh, kw).
// One wave per tap (9 waves): per 32-pixel step it reads N/16 dY fragments and (slab/16) X fragments and issues their outer product of
// v_mfma_f32_16x16x32_bf16; the N x 9 x slab accumulators live in registers for the whole persistent loop over tiles and leave as ONE
// partial row per workgroup ([N][9 * Cin] floats, the layout of cvh_gemm_dw's scratch) that cvh_reduce_multi / gemm_dw_reduce sums.
// HBM-bound stream: algorithmic bytes = X + dY (x number of channel slabs for dY, which the slabs of one tile share through L2).
#include "common.hpp"
#include "cvnets_hip.h"
#include "gemm_params.hpp"

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef short tr_v4s __attribute__((ext_vector_type(4)));

constexpr int D3_TH = 8, D3_TW = 16, D3_HW = D3_TW + 2, D3_HPIX = (D3_TH + 2) * D3_HW;  // 128-pixel tile, 180-pixel halo image
constexpr int D3_THREADS = 576;                                                          // 9 waves = 9 taps
constexpr int D3_MAXSLAB = 24;

struct Dw3Geom {
  int tiles_h, tiles_w, ntiles;
  int rows;    // partial rows = workgroups per slab
  int nslab;
  int s_src[D3_MAXSLAB], s_c0[D3_MAXSLAB], s_cs[D3_MAXSLAB];
  int xp;      // LDS pitch of the halo image (elements)
  int dp;      // LDS pitch of the dY tile (elements)
};

__device__ __forceinline__ bf16x8_t tr_frag8(const bf16_t* lo, const bf16_t* hi) {
  const tr_v4s a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tr_v4s*)(lo));
  const tr_v4s b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tr_v4s*)(hi));
  typedef short v8s __attribute__((ext_vector_type(8)));
  return __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}

// NB = N / 16 output-channel blocks, CBMAX = largest slab / 16
template <int NB, int CBMAX>
__global__ __launch_bounds__(D3_THREADS) void conv3x3_dw_kernel(GemmTNParams p, Dw3Geom g) {
  constexpr int N = 16 * NB;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  bf16_t* xt = reinterpret_cast<bf16_t*>(smem_raw);   // 2 x ([D3_HPIX][xp] halo image + [128][dp] dY tile)
  const int tid = threadIdx.x, lane = tid & 63, tap = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
  const int kh = tap / 3, kw = tap - 3 * kh;
  // XCD-contiguous order, slab fastest: the slabs of one tile (which all read its dY rows) run next to each other on one XCD
  const int lb = xcd_chunk_id((int)blockIdx.x, (int)gridDim.x);
  const int slab = lb % g.nslab, r0 = lb / g.nslab;
  const int cs = g.s_cs[slab], cb = cs / 16, c0 = g.s_c0[slab];
  const bf16_t* __restrict__ xs = reinterpret_cast<const bf16_t*>(g.s_src[slab] ? p.src2 : p.src1);
  const int Cs = g.s_src[slab] ? p.C2 : p.C1;
  const bf16_t* __restrict__ dy = reinterpret_cast<const bf16_t*>(p.dy);
  const int xch = cs / 8, dch = N / 8;

  f32x4_t acc[NB][CBMAX];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int c = 0; c < CBMAX; ++c) acc[nb][c] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // Tiles are double-buffered in LDS and prefetched through registers: the next tile's global loads are issued before this tile's MFMAs
  // and written to the other buffer after them — one barrier per tile.  (The 9-wave workgroup with ~120 registers per lane is alone on
  // its CU; without the prefetch every tile exposed a full HBM round trip.)
  constexpr int XIT = (D3_HPIX * (CBMAX * 2) + D3_THREADS - 1) / D3_THREADS;
  constexpr int DIT = (D3_TH * D3_TW * (N / 8) + D3_THREADS - 1) / D3_THREADS;
  const int tile_elems = D3_HPIX * g.xp + D3_TH * D3_TW * g.dp;
  uint4 xr[XIT], dr[DIT];
  auto load_tile = [&](int t) __attribute__((always_inline)) {
    const int tw = t % g.tiles_w, t1 = t / g.tiles_w;
    const int th = t1 % g.tiles_h, b = t1 / g.tiles_h;
    const int h0 = th * D3_TH, w0 = tw * D3_TW;
#pragma unroll
    for (int it = 0; it < XIT; ++it) {
      const int i = tid + it * D3_THREADS;
      const int px = i / xch, ck = i - px * xch;
      const int hr = px / D3_HW, hc = px - hr * D3_HW;
      const int h = h0 - 1 + hr, w = w0 - 1 + hc;
      xr[it] = make_uint4(0, 0, 0, 0);
      if (i < D3_HPIX * xch && h >= 0 && h < p.H && w >= 0 && w < p.W)
        xr[it] = *reinterpret_cast<const uint4*>(xs + (((size_t)b * p.H + h) * p.W + w) * Cs + c0 + ck * 8);
    }
#pragma unroll
    for (int it = 0; it < DIT; ++it) {
      const int i = tid + it * D3_THREADS;
      const int px = i / dch, ck = i - px * dch;
      const int h = h0 + px / D3_TW, w = w0 + px % D3_TW;
      dr[it] = make_uint4(0, 0, 0, 0);
      if (i < D3_TH * D3_TW * dch && h < p.H && w < p.W) dr[it] = *reinterpret_cast<const uint4*>(dy + (((size_t)b * p.H + h) * p.W + w) * N + ck * 8);
    }
  };
  auto store_tile = [&](bf16_t* xb) __attribute__((always_inline)) {
    bf16_t* db = xb + D3_HPIX * g.xp;
#pragma unroll
    for (int it = 0; it < XIT; ++it) {
      const int i = tid + it * D3_THREADS;
      const int px = i / xch, ck = i - px * xch;
      if (i < D3_HPIX * xch) *reinterpret_cast<uint4*>(xb + px * g.xp + ck * 8) = xr[it];
    }
#pragma unroll
    for (int it = 0; it < DIT; ++it) {
      const int i = tid + it * D3_THREADS;
      const int px = i / dch, ck = i - px * dch;
      if (i < D3_TH * D3_TW * dch) *reinterpret_cast<uint4*>(db + px * g.dp + ck * 8) = dr[it];
    }
  };

  int t = r0, cur = 0;
  if (t < g.ntiles) {
    load_tile(t);
    store_tile(xt);
  }
  __syncthreads();
  for (; t < g.ntiles; t += g.rows) {
    const int tn = t + g.rows;
    if (tn < g.ntiles) load_tile(tn);
    const bf16_t* xb = xt + cur * tile_elems;
    const bf16_t* db = xb + D3_HPIX * g.xp;
    // 4 steps of 32 pixels (two tile rows); lane (l15, l4) supplies pixel rows 8*l4 + (l15 >> 2) (+4) of the step, columns 4*(l15 & 3)..+3
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int plo = 32 * ks + 8 * l4 + (l15 >> 2), phi = plo + 4;
      const int col = 4 * (l15 & 3);
      const bf16_t* dlo = db + plo * g.dp + col;
      const bf16_t* dhi = db + phi * g.dp + col;
      const bf16_t* xlo = xb + (((plo >> 4) + kh) * D3_HW + (plo & 15) + kw) * g.xp + col;
      const bf16_t* xhi = xb + (((phi >> 4) + kh) * D3_HW + (phi & 15) + kw) * g.xp + col;
      bf16x8_t bx[CBMAX];
#pragma unroll
      for (int c = 0; c < CBMAX; ++c)
        if (c < cb) bx[c] = tr_frag8(xlo + 16 * c, xhi + 16 * c);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const bf16x8_t ad = tr_frag8(dlo + 16 * nb, dhi + 16 * nb);
#pragma unroll
        for (int c = 0; c < CBMAX; ++c)
          if (c < cb) acc[nb][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ad, bx[c], acc[nb][c], 0, 0, 0);  // D[n][c] += dY[p][n] X[p + tap][c]
      }
    }
    if (tn < g.ntiles) store_tile(xt + (cur ^ 1) * tile_elems);
    __syncthreads();
    cur ^= 1;
  }

  // partial row r0: part[r0][n][tap * Cin + (slab's first channel in the concatenation) + c]
  const int Cin = p.C1 + p.C2, cg0 = (g.s_src[slab] ? p.C1 : 0) + c0;
  float* __restrict__ out = p.part + (size_t)r0 * N * p.Ktot;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int c = 0; c < CBMAX; ++c)
      if (c < cb) {
#pragma unroll
        for (int e = 0; e < 4; ++e) out[(size_t)(nb * 16 + 4 * l4 + e) * p.Ktot + tap * Cin + cg0 + 16 * c + l15] = acc[nb][c][e];
      }
}

bool dw3_plan(const GemmTNParams& p, Dw3Geom& g, size_t& smem, int& nb, int& cbmax) {
  nb = p.N / 16;
  const int cap = nb <= 6 ? 48 : 32;  // accumulators: NB x (cap / 16) x 4 registers
  cbmax = cap / 16;
  g.nslab = 0;
  int cs_max = 0;
  for (int src = 0; src < 2; ++src) {
    const int C = src == 0 ? p.C1 : p.C2;
    for (int c0 = 0; c0 < C;) {
      if (g.nslab == D3_MAXSLAB) return false;
      const int cs = C - c0 < cap ? C - c0 : cap;
      g.s_src[g.nslab] = src; g.s_c0[g.nslab] = c0; g.s_cs[g.nslab] = cs;
      if (cs > cs_max) cs_max = cs;
      ++g.nslab;
      c0 += cs;
    }
  }
  g.tiles_h = (p.H + D3_TH - 1) / D3_TH;
  g.tiles_w = (p.W + D3_TW - 1) / D3_TW;
  g.ntiles = p.B * g.tiles_h * g.tiles_w;
  g.xp = cs_max + 8;
  g.dp = p.N + 8;
  smem = 2 * ((size_t)D3_HPIX * g.xp + (size_t)D3_TH * D3_TW * g.dp) * 2;  // double-buffered
  // one 9-wave workgroup per CU (register-bound): 256 workgroups shared between the slabs, twice that to even out the tail
  int rows = 512 / g.nslab;
  if (rows > g.ntiles) rows = g.ntiles;
  if (rows < 1) rows = 1;
  g.rows = rows;
  return smem <= 150 * 1024;
}

}  // namespace

bool conv3x3_dw_eligible(const GemmTNParams& p) {
  if (cvh_tune_get(CVH_TUNE_NO_CONV3X3_DW)) return false;
  if (p.KH != 3 || p.KW != 3 || p.stride != 1 || p.pad != 1 || p.dil != 1 || p.Ho != p.H || p.Wo != p.W) return false;
  if (p.N != 96 && p.N != 128 && p.N != 160) return false;
  if (p.C1 < 16 || (p.C1 % 16) || (p.C2 % 16) || (p.C2 != 0 && p.src2 == nullptr)) return false;
  if (p.bias_part != nullptr || p.dy_xf.mode != 0 || p.x_xf.mode != 0) return false;
  if (p.M < 16384) return false;
  Dw3Geom g;
  size_t smem;
  int nb, cbmax;
  return dw3_plan(p, g, smem, nb, cbmax);
}

int conv3x3_dw_rows(const GemmTNParams& p) {
  Dw3Geom g;
  size_t smem;
  int nb, cbmax;
  if (!dw3_plan(p, g, smem, nb, cbmax)) return 0;
  return g.rows;
}

int launch_conv3x3_dw(const GemmTNParams& p, hipStream_t st) {
  Dw3Geom g;
  size_t smem;
  int nb, cbmax;
  if (!dw3_plan(p, g, smem, nb, cbmax) || p.part == nullptr) return -2;
#define D3_LAUNCH(NB_, CB_)                                                                                                             \
  do {                                                                                                                                  \
    static DynSmemAttr attr;                                                                                                            \
    if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(conv3x3_dw_kernel<NB_, CB_>), smem); e != hipSuccess) return (int)e;  \
    hipLaunchKernelGGL((conv3x3_dw_kernel<NB_, CB_>), dim3(g.rows * g.nslab), dim3(D3_THREADS), smem, st, p, g);                        \
  } while (0)
  if (nb == 6) D3_LAUNCH(6, 3);
  else if (nb == 8) D3_LAUNCH(8, 2);
  else D3_LAUNCH(10, 2);
#undef D3_LAUNCH
  CVH_CHECK_LAUNCH();
  return 0;
}
