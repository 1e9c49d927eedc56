// ir_pb kernel: the backward pass of the projection conv of an InvertedResidual block (cvnets/modules/mobilenetv2.py:194-219,231-235:
// depthwise -> BatchNorm -> SiLU -> 1x1 projection -> BatchNorm) from ONE pass over the 4x-wide depthwise output y2 (bf16, SiLU):
//
//   g2[r][c]  = (sum_k dy3[r][k] W3[k][c]) * act'(bn2(y2[r][c]))      stored: the operand of the depthwise backward kernel
//   stats     = (sum_r g2, sum_r g2 * xhat2)                           BatchNorm-backward statistics of bn2
//   dW3[k][c] = sum_r dy3[r][k] * act(bn2(y2[r][c]))                   the projection weight gradient
//
// Before, the dW GEMM and the dX GEMM each read y2 and each evaluated the sigmoid; here y2 is read once and the sigmoid evaluated once.
// dy3 itself may be given as the BatchNorm-backward combination of the block's two NARROW tensors, dy3 = ca dout + cb y3 + cc
// (formed while staging: the standalone bn_bwd_apply pass over them disappears).
//
// Workgroup = 4 waves = 64 channels of y2 x a row split of 64-pixel tiles; wave w owns channels 16 w .. 16 w + 15 in both products:
//   g2 phase   D[ch][px] = W3^T[ch][:] . dy3[px][:]   A = W3 rows (registers), B = dy3 tile rows (LDS); the epilogue reads y2 from the LDS tile
//              in the accumulator layout (a lane = 1 pixel x 4 channels), writes z2 = act(bn2(y2)) to a second tile and g2 over y2 in place;
//   dW phase   D[k][ch]  = dy3^T[k][:] . z2[:][ch]     contraction over pixels, both operands by transposing LDS reads (ds_read_tr16_b64);
//   store      the g2 tile leaves as full 128-byte rows (16 bytes per thread).
// Partial results per row split: dw_part[R][Cout][hid], stats_part[R][2][hid] (summed by cvh_sum_partials / cvh_bn_bwd_finalize: fixed
// order, no atomics).
#include "common.hpp"
#include "cvnets_hip.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef short tr_v4s __attribute__((ext_vector_type(4)));

constexpr int PB_TP = 64;  // pixels per tile
constexpr int PB_YP = 72;  // LDS pitch (elements) of the y2 / z2 tiles: 64 channels + 16 B

__device__ __forceinline__ f32x4_t mfma16(bf16x8_t a, bf16x8_t b, f32x4_t c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
// transposed fragment: the 16 lanes of a group address 4 pixels (l15 >> 2) x 4 channel quads (l15 & 3); lane l15 receives channel
// (base + l15) at the 4 pixels of `lo` (k slots 0..3) and of `hi` (k slots 4..7)
__device__ __forceinline__ bf16x8_t tr_frag8(const bf16_t* lo, const bf16_t* hi) {
  const tr_v4s a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tr_v4s*)(lo));
  const tr_v4s b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tr_v4s*)(hi));
  return __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}

struct IrPbParams {
  const bf16_t* dy;     // [M][Cout]  dy3, or dout when y3 != nullptr
  const bf16_t* y3;     // [M][Cout]  nullptr, or the raw projection output: dy3 = ca dout + cb y3 + cc
  const float* c3;      // [3][Cout]  ca, cb, cc
  const bf16_t* y2;     // [M][hid]
  const float* st2;     // [4][hid]   mean, invstd, scale, shift of bn2
  const bf16_t* w3t;    // [hid][Cout]  projection weights, transposed pack
  bf16_t* g2;           // [M][hid]
  float* stats_part;    // [R][2][hid]
  float* dw_part;       // [R][Cout][hid]
  int M, hid, R, chunks, ntiles;
};

template <int NA>  // Cout = 16 NA, NA even
__global__ __launch_bounds__(256, NA <= 4 ? 3 : 2) void ir_pb_kernel(IrPbParams p) {
  constexpr int COUT = 16 * NA, KS3 = NA / 2, DP = COUT + CVH_M16_PAD, DC = COUT / 8;  // DP: NA even -> COUT / 8 = 0 (mod 4), + 2 chunks (common.hpp)
  constexpr int NYL = PB_TP * 8 / 256, NDL = (PB_TP * DC + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  bf16_t* yt = reinterpret_cast<bf16_t*>(smem_raw);      // [64][PB_YP]  y2 tile, this workgroup's 64 channels; g2 after the first phase
  bf16_t* zt = yt + PB_TP * PB_YP;                        // [64][PB_YP]  z2 = act(bn2(y2))
  bf16_t* dt = zt + PB_TP * PB_YP;                        // [64][DP]     dy3 tile
  float* c3s = reinterpret_cast<float*>(dt + PB_TP * DP);  // [3][COUT]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
  const int lb = xcd_chunk_id(blockIdx.x, gridDim.x);
  const int chunk = lb % p.chunks, split = lb / p.chunks;
  const int c0 = chunk * 64, hid = p.hid, cw = c0 + 16 * wave;
  const bool two_src = p.y3 != nullptr;
  if (two_src)
    for (int i = tid; i < 3 * COUT; i += 256) c3s[i] = p.c3[i];

  // A operand of the g2 product: W3^T rows of this wave's channels
  bf16x8_t w3f[KS3];
#pragma unroll
  for (int ks = 0; ks < KS3; ++ks) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (cw + l15 < hid) v = *reinterpret_cast<const uint4*>(p.w3t + (size_t)(cw + l15) * COUT + 32 * ks + 8 * l4);
    w3f[ks] = __builtin_bit_cast(bf16x8_t, v);
  }
  // BatchNorm constants of this lane's 4 accumulator channels cw + 4 l4 .. + 3, as packed pairs:
  // yh = sc y + sh, sigmoid exponent exp2(nsc y + nsh), xhat = is y + nmi
  f32x2_t sc[2], sh[2], nsc[2], nsh[2], is[2], nmi[2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int ch = cw + 4 * l4 + 2 * h + e;
      const bool ok = ch < hid;
      const float mu_ = ok ? p.st2[ch] : 0.f, is_ = ok ? p.st2[(size_t)hid + ch] : 0.f;
      const float sc_ = ok ? p.st2[(size_t)2 * hid + ch] : 0.f, sh_ = ok ? p.st2[(size_t)3 * hid + ch] : 0.f;
      sc[h][e] = sc_; sh[h][e] = sh_; nsc[h][e] = -1.4426950408889634f * sc_; nsh[h][e] = -1.4426950408889634f * sh_;
      is[h][e] = is_; nmi[h][e] = -mu_ * is_;
    }

  f32x4_t accw[NA];
#pragma unroll
  for (int a = 0; a < NA; ++a) accw[a] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  f32x2_t s1[2] = {{0.f, 0.f}, {0.f, 0.f}}, s2[2] = {{0.f, 0.f}, {0.f, 0.f}};

  uint4 yr[NYL], dr[NDL], d2r[NDL];
  auto load_tile = [&](int t) __attribute__((always_inline)) {
    const size_t r0 = (size_t)t * PB_TP;
#pragma unroll
    for (int it = 0; it < NYL; ++it) {
      const int i = tid + it * 256;
      const int px = i >> 3, cg = i & 7;
      const bool ok = r0 + px < (size_t)p.M && c0 + cg * 8 < hid;
      yr[it] = ok ? *reinterpret_cast<const uint4*>(p.y2 + (r0 + px) * hid + c0 + cg * 8) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int it = 0; it < NDL; ++it) {
      const int i = tid + it * 256;
      const int px = i / DC, cg = i - px * DC;
      const bool ok = i < PB_TP * DC && r0 + px < (size_t)p.M;
      dr[it] = ok ? *reinterpret_cast<const uint4*>(p.dy + (r0 + px) * COUT + cg * 8) : make_uint4(0, 0, 0, 0);
      if (two_src) d2r[it] = ok ? *reinterpret_cast<const uint4*>(p.y3 + (r0 + px) * COUT + cg * 8) : make_uint4(0, 0, 0, 0);
    }
  };

  int t = split;
  if (t < p.ntiles) load_tile(t);
  for (; t < p.ntiles; t += p.R) {
    const size_t r0 = (size_t)t * PB_TP;
    __syncthreads();  // the previous tile's g2 rows have left the y2 tile (first iteration: the coefficients are in place)
#pragma unroll
    for (int it = 0; it < NYL; ++it) {
      const int i = tid + it * 256;
      *reinterpret_cast<uint4*>(yt + (i >> 3) * PB_YP + (i & 7) * 8) = yr[it];
    }
#pragma unroll
    for (int it = 0; it < NDL; ++it) {
      const int i = tid + it * 256;
      const int px = i / DC, cg = i - px * DC;
      if (i < PB_TP * DC) {
        V8<bf16_t> o;
        o.d = dr[it];
        if (two_src) {
          V8<bf16_t> yv;
          yv.d = d2r[it];
          float g[8], y[8];
          v8_unpack(o, g);
          v8_unpack(yv, y);
          const bool ok = r0 + px < (size_t)p.M;  // rows past the end stay zero (cc alone would not)
#pragma unroll
          for (int j = 0; j < 8; ++j) g[j] = ok ? c3s[cg * 8 + j] * g[j] + (c3s[COUT + cg * 8 + j] * y[j] + c3s[2 * COUT + cg * 8 + j]) : 0.f;
          v8_pack(g, o);
        }
        *reinterpret_cast<uint4*>(dt + px * DP + cg * 8) = o.d;
      }
    }
    __syncthreads();
    if (t + p.R < p.ntiles) load_tile(t + p.R);  // next tile's operands, in flight under this tile's arithmetic

    // ---- g2 phase: four 16-pixel blocks; this wave's 16 channels ----
#pragma unroll
    for (int pb = 0; pb < PB_TP / 16; ++pb) {
      f32x4_t a = {0.f, 0.f, 0.f, 0.f};
      const bf16_t* drow = dt + (16 * pb + l15) * DP + 8 * l4;
#pragma unroll
      for (int ks = 0; ks < KS3; ++ks) a = mfma16(w3f[ks], *reinterpret_cast<const bf16x8_t*>(drow + 32 * ks), a);
      bf16_t* yp = yt + (16 * pb + l15) * PB_YP + 16 * wave + 4 * l4;
      const uint2 yv = *reinterpret_cast<const uint2*>(yp);
      const bool rok = r0 + 16 * pb + l15 < (size_t)p.M;
      uint32_t zw[2], gw[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const uint32_t w = h == 0 ? yv.x : yv.y;
        const f32x2_t y = {__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)};
        const f32x2_t yh = sc[h] * y + sh[h];
        const f32x2_t e = nsc[h] * y + nsh[h];
        f32x2_t d = {__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])};
        d = d + 1.0f;
        const f32x2_t sg = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
        const f32x2_t z = yh * sg;
        const f32x2_t gp = z * (1.0f - sg) + sg;
        const f32x2_t g = f32x2_t{a[2 * h], a[2 * h + 1]} * gp;  // rows past the end: dy3 is zero there, so is g
        zw[h] = f2bf_pk(z[0], z[1]);
        gw[h] = f2bf_pk(g[0], g[1]);
        if (rok) {
          s1[h] += g;
          s2[h] += g * (is[h] * y + nmi[h]);
        }
      }
      *reinterpret_cast<uint2*>(zt + (16 * pb + l15) * PB_YP + 16 * wave + 4 * l4) = make_uint2(zw[0], zw[1]);
      *reinterpret_cast<uint2*>(yp) = make_uint2(gw[0], gw[1]);
    }
    wave_lds_sync();  // z2 columns are wave-private

    // ---- dW phase: contraction over the tile's pixels ----
#pragma unroll
    for (int ks = 0; ks < PB_TP / 32; ++ks) {
      const int prow = 32 * ks + 4 * l4 + (l15 >> 2);  // k slots 0..3 <-> pixels 32 ks + 4 l4 + (0..3); slots 4..7: the same + 16
      const bf16_t* zlo = zt + prow * PB_YP + 16 * wave + 4 * (l15 & 3);
      const bf16x8_t zb = tr_frag8(zlo, zlo + 16 * PB_YP);
#pragma unroll
      for (int a = 0; a < NA; ++a) {
        const bf16_t* dlo = dt + prow * DP + 16 * a + 4 * (l15 & 3);
        accw[a] = mfma16(tr_frag8(dlo, dlo + 16 * DP), zb, accw[a]);
      }
    }
    __syncthreads();  // every wave's g2 columns are in the tile

    // ---- g2 tile -> HBM, full rows ----
#pragma unroll
    for (int it = 0; it < NYL; ++it) {
      const int i = tid + it * 256;
      const int px = i >> 3, cg = i & 7;
      if (r0 + px < (size_t)p.M && c0 + cg * 8 < hid)
        *reinterpret_cast<uint4*>(p.g2 + (r0 + px) * hid + c0 + cg * 8) = *reinterpret_cast<const uint4*>(yt + px * PB_YP + cg * 8);
    }
  }

  // ---- workgroup results ----
  float t1[4] = {s1[0][0], s1[0][1], s1[1][0], s1[1][1]}, t2[4] = {s2[0][0], s2[0][1], s2[1][0], s2[1][1]};
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) {  // a lane's 4 channels are shared with the 15 other pixel lanes of its group: fixed butterfly
      t1[e] += __shfl_xor(t1[e], m, 64);
      t2[e] += __shfl_xor(t2[e], m, 64);
    }
  if (l15 == 0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int ch = cw + 4 * l4 + e;
      if (ch < hid) {
        p.stats_part[((size_t)split * 2 + 0) * hid + ch] = t1[e];
        p.stats_part[((size_t)split * 2 + 1) * hid + ch] = t2[e];
      }
    }
  }
  if (cw + l15 < hid) {  // D[row = 16 a + 4 l4 + e (dy3 channel)][col = l15 (y2 channel)]
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
      for (int e = 0; e < 4; ++e) p.dw_part[((size_t)split * COUT + 16 * a + 4 * l4 + e) * hid + cw + l15] = accw[a][e];
  }
}

bool ir_pb_ok(int M, int hid, int Cout) { return M >= 4096 && hid % 64 == 0 && Cout % 32 == 0 && Cout >= 32 && Cout <= 160; }

int ir_pb_plan(int M, int hid, int* chunks, int* ntiles) {
  *chunks = hid / 64;
  *ntiles = (M + PB_TP - 1) / PB_TP;
  int R = 768 / *chunks;  // ~3 workgroups per CU
  if (R < 1) R = 1;
  if (R > *ntiles) R = *ntiles;
  return R;
}

template <int NA> size_t ir_pb_smem() { return (size_t)PB_TP * (2 * PB_YP + 16 * NA + CVH_M16_PAD) * 2 + 3 * 16 * NA * 4; }

}  // namespace

extern "C" int cvh_ir_pb_rows(int M, int hid, int Cout) {
  if (!ir_pb_ok(M, hid, Cout)) return 0;
  int chunks, ntiles;
  return ir_pb_plan(M, hid, &chunks, &ntiles);
}

extern "C" int cvh_ir_pb(int dtype, const void* dy, const void* y3, const float* c3, const void* y2, const float* st2, int act, const void* w3t,
                         void* g2, float* stats_part, float* dw_part, int M, int hid, int Cout, void* stream) {
  if (dtype != CVH_DT_BF16) return -1;
  if (act != CVH_ACT_SILU || !ir_pb_ok(M, hid, Cout)) return -2;
  if (dy == nullptr || y2 == nullptr || st2 == nullptr || w3t == nullptr || g2 == nullptr || stats_part == nullptr || dw_part == nullptr) return -2;
  if (y3 != nullptr && c3 == nullptr) return -2;
  IrPbParams p;
  p.dy = reinterpret_cast<const bf16_t*>(dy); p.y3 = reinterpret_cast<const bf16_t*>(y3); p.c3 = c3;
  p.y2 = reinterpret_cast<const bf16_t*>(y2); p.st2 = st2; p.w3t = reinterpret_cast<const bf16_t*>(w3t); p.g2 = reinterpret_cast<bf16_t*>(g2);
  p.stats_part = stats_part; p.dw_part = dw_part; p.M = M; p.hid = hid;
  p.R = ir_pb_plan(M, hid, &p.chunks, &p.ntiles);
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(p.R * p.chunks);
  cvh_family_tally(2, ((long long)M * Cout * (y3 != nullptr ? 2 : 1) + 2LL * M * hid) * 2);
#define IR_PB(NA_) hipLaunchKernelGGL((ir_pb_kernel<NA_>), grid, dim3(256), ir_pb_smem<NA_>(), st, p)
  switch (Cout / 16) {
    case 2: IR_PB(2); break;
    case 4: IR_PB(4); break;
    case 6: IR_PB(6); break;
    case 8: IR_PB(8); break;
    default: IR_PB(10); break;
  }
#undef IR_PB
  CVH_CHECK_LAUNCH();
  return 0;
}
