// dwx kernels: the wide half of an InvertedResidual block (cvnets/modules/mobilenetv2.py:180-207,231-235: 1x1 expansion conv -> BatchNorm
// -> SiLU -> depthwise 3x3 conv (pad 1, stride 1 / 2) -> BatchNorm) with the 4x-wide expansion output y1 = x W1^T NEVER in HBM, bf16.
//
//   forward   y2 = dwconv(act(bn1(x W1^T)))        reads the NARROW block input x (tile + halo), writes y2, emits the statistics of y2
//   backward  g1 = dwconv^T(dy2) * act'(bn1(y1))   y1 recomputed from x at the tile's own pixels; dW of the depthwise conv, statistics of g1
//
// Everything GEMM-shaped or stencil-shaped runs on the matrix pipe (v_mfma_f32_16x16x32_bf16), which the HBM-bound step leaves idle:
//   * y1^T[ch][px] = W1[ch][:] . x[px][:]                                    A = W1 rows (resident in registers), B = x tile rows (LDS)
//   * the depthwise stencil as a product with DIAGONAL weight blocks:         out^T[c][px] = sum_(tap, c') diag(w_tap)[c][c'] a[px + tap][c']
//     two taps share one K = 32 step (k = 16 * slot + c'), i.e. 5 MFMAs per 16 pixels x 16 channels; the B operand of a tap is simply
//     the 16-byte channel group of the shifted pixel — stride 2 and the transposed (backward) stencil are address arithmetic only
//   * the depthwise weight gradient dW[tap][c] = sum_px dy[px - tap][c] z[px][c] as the DIAGONAL of dy_shifted^T z per 16-channel block
//     (transpose-read operands, K = pixels)
// so the VALU is left with BatchNorm + SiLU (+ SiLU') per element, packing and the statistics — about half of what the LDS-stencil
// kernels of dwfused.hip issue — and the accumulators of the weight gradient no longer live 72-per-lane in the register file.
//
// A workgroup = 4 waves = one 64-channel chunk of one spatial tile; wave w owns channels 16w .. 16w+15 of the chunk in EVERY phase, so the
// activation tile it writes (its own 32-byte column group of each pixel row) is wave-private: between the phases there is no workgroup
// barrier, only the two around the refresh of the shared x tile.  Accumulator layout of the MFMA (lane = 1 pixel x 4 consecutive
// channels) is the layout of every epilogue; results leave as 8-byte stores (32 contiguous bytes per pixel and wave).
// BatchNorm of the expansion needs its batch statistics BEFORE this kernel runs: y1 is linear in x, so sum y1 = W1 (sum x) and
// sum y1^2 = diag(W1 (x^T x) W1^T) come from the K x K Gram matrix of the narrow input (gram_bn_stats_kernel) — the same Gram matrix the
// backward pass needs for the weight gradient (cvh_bn_dw_combine), computed once.
#include "common.hpp"
#include "cvnets_hip.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef short tr_v4s __attribute__((ext_vector_type(4)));
typedef short v8s __attribute__((ext_vector_type(8)));

constexpr int DX_CC = 64;  // channels per workgroup: 4 waves x 16
constexpr int DX_AP = 80;  // LDS pixel pitch (elements) of the activation / gradient tiles: 160 B — 16 consecutive pixels x 16 B cover all 64 banks

template <int S> struct DxTile;
template <> struct DxTile<1> {  // 8 x 16 outputs <- 10 x 18 inputs
  static constexpr int OH = 8, OW = 16, IH = 10, IW = 18;
};
template <> struct DxTile<2> {  // 4 x 8 outputs <- 9 x 17 inputs
  static constexpr int OH = 4, OW = 8, IH = 9, IW = 17;
};

__device__ __forceinline__ f32x4_t mfma16(bf16x8_t a, bf16x8_t b, f32x4_t c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float rbf(float f) { return bf2f(f2bf(f)); }

// A operand of the diagonal-weight product for tap pair tp: A[c = l15][k = 8 * l4 + j] = (16 * slot + c' == k ? w[2 tp + slot][c] : 0)
__device__ __forceinline__ bf16x8_t diag_frag(const bf16_t* wd /*[9][C]*/, int C, int ch, int tp, int l15, int l4, bool flip) {
  int tap = 2 * tp + (l4 >> 1);
  uint16_t wv = 0;
  if (tap < 9 && ch < C && (l15 >> 3) == (l4 & 1)) wv = wd[(size_t)(flip ? 8 - tap : tap) * C + ch].v;
  v8s f;
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] = (j == (l15 & 7)) ? (short)wv : (short)0;
  return __builtin_bit_cast(bf16x8_t, f);
}

struct DwxFwdParams {
  const bf16_t* x;      // [B*H*W][Cin]
  const bf16_t* w1;     // [hid][Cin]
  const float* scale1;  // [hid] BatchNorm of the expansion: scale, shift
  const float* shift1;
  const bf16_t* wd;     // [9][hid]
  bf16_t* y2;           // [B*Ho*Wo][hid]
  float* stats_part;    // [rows][2][hid] or nullptr
  int act1;
  int B, H, W, Ho, Wo, hid;
  int tiles_h, tiles_w, ntiles, chunks;
#ifdef CVH_DWX_DBG
  int dbg;  // developer knob (CVH_TUNE key 17, -DCVH_DWX_DBG builds only): skip phases to time them (results are WRONG when non-zero)
#endif
};
#ifdef CVH_DWX_DBG
#define DX_DBG(p_, bit_) ((p_).dbg & (bit_))
#else
#define DX_DBG(p_, bit_) 0
#endif

// LDS pitch of the x tiles: dense rows for one 32-wide K step (2-way conflicts on the operand reads, 4 workgroups per CU), + 16 B otherwise
template <int CIN> __host__ __device__ constexpr int dx_xp() { return 32 * ((CIN + 31) / 32) + (CIN <= 32 ? 0 : 8); }
// x tile buffers of the forward kernel (see the tile loop)
#ifndef DX_XBUF64
#define DX_XBUF64 2
#endif
template <int CIN> __host__ __device__ constexpr int dx_xbuf() { return CIN <= 32 ? 2 : (CIN <= 64 ? DX_XBUF64 : 1); }
// waves per SIMD the register allocator is held to (= workgroups per CU the LDS footprint allows)
#ifndef DX_OCC_A
#define DX_OCC_A 4
#endif
#ifndef DX_OCC_B
#define DX_OCC_B 3
#endif
#ifndef DX_OCC16
#define DX_OCC16 3
#endif
#ifndef DX_OCC64
#define DX_OCC64 2
#endif
template <int CIN> __host__ __device__ constexpr int dx_fwd_occ() { return CIN <= 16 ? DX_OCC16 : (CIN <= 32 ? 3 : (CIN <= 64 ? DX_OCC64 : 2)); }

template <int S, int CIN>
__global__ __launch_bounds__(256, dx_fwd_occ<CIN>()) void dwx_fwd_kernel(DwxFwdParams p) {
  using TL = DxTile<S>;
  constexpr int KS = (CIN + 31) / 32, XP = dx_xp<CIN>(), XC = CIN / 8;
  constexpr int NPIX = TL::IH * TL::IW, NPB = (NPIX + 15) / 16, NOB = TL::OH * TL::OW / 16;
  constexpr int NXL = (NPIX * XC + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  bf16_t* xs0 = reinterpret_cast<bf16_t*>(smem_raw);  // 2 x [NPB * 16][XP]  block input, tile + halo (zero outside the image / past CIN)
  bf16_t* at = xs0 + dx_xbuf<CIN>() * NPB * 16 * XP;               // [4][NPB * 16][16]   act(bn1(y1)): one dense [pixel][16 channels] image per wave

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
  const int lb = xcd_chunk_id(blockIdx.x, gridDim.x);
  const int chunk = lb % p.chunks, row_id = lb / p.chunks;
  const int cw = chunk * DX_CC + 16 * wave;  // first channel of this wave
  const int hid = p.hid;
  bf16_t* atw = at + wave * (NPB * 16 * 16);

  for (int i = tid; i < dx_xbuf<CIN>() * NPB * 16 * XP / 8; i += 256) reinterpret_cast<uint4*>(xs0)[i] = make_uint4(0, 0, 0, 0);

  bf16x8_t w1f[KS], wdf[5];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    uint4 v = make_uint4(0, 0, 0, 0);
    const int ch = cw + l15, k = 32 * ks + 8 * l4;
    if (ch < hid && k < CIN) v = *reinterpret_cast<const uint4*>(p.w1 + (size_t)ch * CIN + k);
    w1f[ks] = __builtin_bit_cast(bf16x8_t, v);
  }
#pragma unroll
  for (int tp = 0; tp < 5; ++tp) wdf[tp] = diag_frag(p.wd, hid, cw + l15, tp, l15, l4, false);
  // BatchNorm coefficients of the lane's 4 channels as two packed pairs (the VALU is what bounds this kernel — a plain wave64 fp32
  // instruction occupies it for 4 cycles, v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 do two lanes' worth in the same time); the exponent
  // of the sigmoid gets its own pre-scaled pair: 2^(-yh log2 e) = exp2(nsc * y + nsh)
  f32x2_t sc[2], sh[2], nsc[2], nsh[2];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int ch = cw + 4 * l4 + e;
    const float a = ch < hid ? p.scale1[ch] : 0.f, c = ch < hid ? p.shift1[ch] : 0.f;
    sc[e >> 1][e & 1] = a;
    sh[e >> 1][e & 1] = c;
    nsc[e >> 1][e & 1] = -1.4426950408889634f * a;
    nsh[e >> 1][e & 1] = -1.4426950408889634f * c;
  }
  f32x2_t s1[2] = {{0.f, 0.f}, {0.f, 0.f}}, s2[2] = {{0.f, 0.f}, {0.f, 0.f}};
  // tile-independent parts of the output addressing (elements): lane offset of output block ob from the tile's first output pixel
  // (output block ob of a lane: row ob (stride 1) / 2 ob + l15 / 8 (stride 2) -> offset = yoff0 + ob * ystep)
  const int yoff0 = ((S == 1 ? 0 : (l15 >> 3)) * p.Wo + (S == 1 ? l15 : (l15 & 7))) * hid + cw + 4 * l4;
  const int ystep = (S == 1 ? 1 : 2) * p.Wo * hid;
  const bool wave_full = cw + 16 <= hid;  // all 16 channels of this wave exist
  uint32_t vm_in = 0;                      // halo-tile pixels of this lane that exist at all (the last block overhangs the tile)
#pragma unroll
  for (int pb = 0; pb < (TL::IH * TL::IW + 15) / 16; ++pb) vm_in |= ((16 * pb + l15 < TL::IH * TL::IW) ? 1u : 0u) << pb;

  // LDS offsets (elements) of the lane's B-operand reads in the stencil phase: tap (2 tp + slot), clamped to a real tap for the idle slot
  int toff[5];
#pragma unroll
  for (int tp = 0; tp < 5; ++tp) {
    int tap = 2 * tp + (l4 >> 1);
    tap = tap > 8 ? 8 : tap;
    toff[tp] = ((tap / 3) * TL::IW + (tap % 3)) * 16 + 8 * (l4 & 1);
  }

  const int t_step = gridDim.x / p.chunks;
  uint4 xr[NXL];
  uint32_t xok = 0;
  // tile-independent part of the load addressing (elements from the tile's first halo pixel), so that an interior tile — all of its
  // halo inside the image: a wave-uniform test — costs one add per load instead of the div / mod / compare / 64-bit multiply chain
  int xoff[NXL];
  uint32_t xin = 0;
#pragma unroll
  for (int it = 0; it < NXL; ++it) {
    const int i = tid + it * 256;
    const int px = i / XC, ck = i - px * XC;
    const int pr = px / TL::IW, pc = px - pr * TL::IW;
    xoff[it] = (pr * p.W + pc) * CIN + ck * 8;
    xin |= (i < NPIX * XC ? 1u : 0u) << it;
  }
  auto load_x = [&](int tix) __attribute__((always_inline)) {
    const int tw = tix % p.tiles_w, t1 = tix / p.tiles_w;
    const int th = t1 % p.tiles_h, b = t1 / p.tiles_h;
    const int hi0 = th * TL::OH * S - 1, wi0 = tw * TL::OW * S - 1;
    if (hi0 >= 0 && wi0 >= 0 && hi0 + TL::IH <= p.H && wi0 + TL::IW <= p.W) {
      const bf16_t* base = p.x + (((size_t)b * p.H + hi0) * p.W + wi0) * CIN;
      xok = xin;
#pragma unroll
      for (int it = 0; it < NXL; ++it) xr[it] = *reinterpret_cast<const uint4*>(((xin >> it) & 1u) ? base + xoff[it] : p.x);
      return;
    }
    xok = 0;
#pragma unroll
    for (int it = 0; it < NXL; ++it) {
      const int i = tid + it * 256;
      const int px = i / XC, ck = i - px * XC;
      const int pr = px / TL::IW, pc = px - pr * TL::IW;
      const int hi = hi0 + pr, wi = wi0 + pc;
      const bool ok = i < NPIX * XC && hi >= 0 && hi < p.H && wi >= 0 && wi < p.W;
      xok |= (ok ? 1u : 0u) << it;
      xr[it] = *reinterpret_cast<const uint4*>(p.x + (ok ? (((size_t)b * p.H + hi) * p.W + wi) * CIN + ck * 8 : (size_t)0));
    }
  };

  auto stage_x = [&](bf16_t* xs) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < NXL; ++it) {
      const int i = tid + it * 256;
      const int px = i / XC, ck = i - px * XC;
      if (i < NPIX * XC) {
        uint4 v = xr[it];
        if (!((xok >> it) & 1u)) v = make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4*>(xs + px * XP + ck * 8) = v;
      }
    }
  };

  // The x tile is double-buffered in LDS: the next tile's input is requested at the top of a tile, and each wave moves its share into
  // the OTHER buffer right after its expansion phase, i.e. BEFORE this tile's result stores are issued.  (Consumed at the top of the
  // next tile instead, the compiler's `s_waitcnt vmcnt(0)` - vmcnt counts loads and stores alike - drained the result stores too: one
  // HBM write latency exposed per tile, 15-19 % of this kernel.)  One workgroup barrier per tile.
  // (Cin > 64 keeps one buffer and consumes the prefetch at the top of the next tile: two buffers would halve its occupancy.)
  constexpr bool DB = dx_xbuf<CIN>() == 2;
  int tix = row_id, cur = 0;
  if (tix < p.ntiles) {
    load_x(tix);
    if (DB) {
      __syncthreads();  // the zero fills are complete
      stage_x(xs0);
    }
  }
  for (; tix < p.ntiles; tix += t_step) {
    const int tw = tix % p.tiles_w, t1 = tix / p.tiles_w;
    const int th = t1 % p.tiles_h, b = t1 / p.tiles_h;
    const int ho0 = th * TL::OH, wo0 = tw * TL::OW;
    const int hi0 = ho0 * S - 1, wi0 = wo0 * S - 1;

    const bf16_t* xs = xs0 + cur * (NPB * 16 * XP);
    if (!DB) {
      __syncthreads();  // every wave is done with the previous x tile (first iteration: the zero fill is complete)
      stage_x(xs0);
    }
    __syncthreads();  // buffer `cur` is complete; every wave has left the previous tile (whose x lived in the other buffer)
    const bool has_next = tix + t_step < p.ntiles && !DX_DBG(p, 16);
    if (has_next) load_x(tix + t_step);  // next tile's input, in flight under this tile's expansion phase

    // pixels of the halo tile that lie inside the image (the conv's zero padding applies to the ACTIVATED tensor); interior tiles —
    // a wave-uniform test — skip the per-pixel arithmetic
    uint32_t vmask = vm_in;
    if (hi0 < 0 || wi0 < 0 || hi0 + TL::IH > p.H || wi0 + TL::IW > p.W) {
      vmask = 0;
#pragma unroll
      for (int pb = 0; pb < NPB; ++pb) {
        const int px = 16 * pb + l15;
        const int pr = px / TL::IW, pc = px - pr * TL::IW;
        const int hi = hi0 + pr, wi = wi0 + pc;
        vmask |= ((px < NPIX && hi >= 0 && hi < p.H && wi >= 0 && wi < p.W) ? 1u : 0u) << pb;
      }
    }

    // ---- expansion + BatchNorm + activation -> activation tile (this wave's 16 channels of every pixel) ----
    // Software pipeline over the 16-pixel blocks, written out: the operand reads of block pb + 1 are issued BEFORE the epilogue of block
    // pb - 1 stores to LDS (the compiler cannot hoist an LDS read above an LDS store it cannot disambiguate, and a wave that waits
    // for its reads, then for its MFMAs, then runs its epilogue, leaves the SIMD idle two thirds of the time at 2-4 waves per SIMD),
    // and the MFMAs of block pb run under the VALU work of block pb - 1.
    if (!DX_DBG(p, 4)) {
      bf16x8_t bq[KS];
      auto rd = [&](int pb) __attribute__((always_inline)) {
        const bf16_t* xrow = xs + (16 * pb + l15) * XP + 8 * l4;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) bq[ks] = *reinterpret_cast<const bf16x8_t*>(xrow + 32 * ks);
      };
      // SiLU(bn(y1)) of 4 channels in 2 x (2 v_pk_fma, 2 v_exp, v_pk_add, 2 v_rcp, v_pk_mul, v_cvt_pk) — SiLU is the only activation
      // these kernels are compiled for (a run-time dispatch per element cuts the unrolled epilogues into blocks the scheduler cannot
      // interleave); y1 enters in fp32 (it is never a tensor here, so nothing rounds it)
      auto epi = [&](int pb, const f32x4_t& acc) __attribute__((always_inline)) {
        uint32_t w[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const f32x2_t y = {acc[2 * h], acc[2 * h + 1]};
          const f32x2_t yh = sc[h] * y + sh[h];
          const f32x2_t t = nsc[h] * y + nsh[h];
          f32x2_t d = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
          d = d + 1.0f;
          const f32x2_t r = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
          const f32x2_t v = yh * r;
          w[h] = f2bf_pk(v[0], v[1]);
        }
        const uint32_t m = (uint32_t)__builtin_amdgcn_sbfe((int)vmask, pb, 1);  // all ones inside the image, zero outside
        *reinterpret_cast<uint2*>(atw + (16 * pb + l15) * 16 + 4 * l4) = make_uint2(w[0] & m, w[1] & m);
      };
      f32x4_t accp = {0.f, 0.f, 0.f, 0.f};
      rd(0);
#pragma unroll
      for (int pb = 0; pb < NPB; ++pb) {
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) acc = mfma16(w1f[ks], bq[ks], acc);
        if (pb + 1 < NPB) rd(pb + 1);
        if (pb > 0) epi(pb - 1, accp);
        accp = acc;
      }
      epi(NPB - 1, accp);
    }
    wave_lds_sync();
    if (DB) {
      if (has_next) stage_x(xs0 + (cur ^ 1) * (NPB * 16 * XP));  // before the result stores below are issued (see above)
      cur ^= 1;
    }

    // ---- depthwise stencil on the matrix pipe (two accumulators: the five products of a block are not one dependent chain) ----
    if (!DX_DBG(p, 8)) {
      bf16x8_t fq[5];
      auto rd = [&](int ob) __attribute__((always_inline)) {
        const int orow = S == 1 ? ob : 2 * ob + (l15 >> 3), ocol = S == 1 ? l15 : (l15 & 7);
        const bf16_t* base = atw + ((orow * S) * TL::IW + ocol * S) * 16;
#pragma unroll
        for (int tp = 0; tp < 5; ++tp) fq[tp] = *reinterpret_cast<const bf16x8_t*>(base + toff[tp]);
      };
      // results leave as 8-byte stores from the accumulator layout: tile base (wave-uniform) + the lane's precomputed offset; statistics
      // from the fp32 accumulators; tiles that overhang the image or a partial channel block take the checked path
      bf16_t* ybase = p.y2 + (((size_t)b * p.Ho + ho0) * p.Wo + wo0) * hid;
      const bool fast = wave_full && ho0 + TL::OH <= p.Ho && wo0 + TL::OW <= p.Wo && !DX_DBG(p, 2);
      auto epi = [&](int ob, const f32x2_t& lo, const f32x2_t& hi2) __attribute__((always_inline)) {
        if (fast) {  // wave-uniform fast path: no per-lane bounds arithmetic
          *reinterpret_cast<uint2*>(ybase + yoff0 + ob * ystep) = make_uint2(f2bf_pk(lo[0], lo[1]), f2bf_pk(hi2[0], hi2[1]));
          s1[0] += lo;
          s1[1] += hi2;
          s2[0] += lo * lo;
          s2[1] += hi2 * hi2;
        } else {
          const int orow = S == 1 ? ob : 2 * ob + (l15 >> 3), ocol = S == 1 ? l15 : (l15 & 7);
          if (ho0 + orow < p.Ho && wo0 + ocol < p.Wo && cw + 4 * l4 < hid && !DX_DBG(p, 2)) {
            *reinterpret_cast<uint2*>(ybase + yoff0 + ob * ystep) = make_uint2(f2bf_pk(lo[0], lo[1]), f2bf_pk(hi2[0], hi2[1]));
            s1[0] += lo;
            s1[1] += hi2;
            s2[0] += lo * lo;
            s2[1] += hi2 * hi2;
          }
        }
      };
      f32x2_t plo = {0.f, 0.f}, phi = {0.f, 0.f};
      rd(0);
#pragma unroll
      for (int ob = 0; ob < NOB; ++ob) {
        f32x4_t a = {0.f, 0.f, 0.f, 0.f}, c = {0.f, 0.f, 0.f, 0.f};
        a = mfma16(wdf[0], fq[0], a);
        c = mfma16(wdf[1], fq[1], c);
        a = mfma16(wdf[2], fq[2], a);
        c = mfma16(wdf[3], fq[3], c);
        a = mfma16(wdf[4], fq[4], a);
        if (ob + 1 < NOB) rd(ob + 1);
        if (ob > 0) epi(ob - 1, plo, phi);
        plo = f32x2_t{a[0], a[1]} + f32x2_t{c[0], c[1]};
        phi = f32x2_t{a[2], a[3]} + f32x2_t{c[2], c[3]};
      }
      epi(NOB - 1, plo, phi);
    }
  }

  if (p.stats_part != nullptr) {
    // a lane's 4 channels are shared with the 15 other pixel lanes of its group: fixed butterfly, then one lane per group writes
    float t1[4] = {s1[0][0], s1[0][1], s1[1][0], s1[1][1]}, t2[4] = {s2[0][0], s2[0][1], s2[1][0], s2[1][1]};
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int m = 1; m < 16; m <<= 1) {
        t1[e] += __shfl_xor(t1[e], m, 64);
        t2[e] += __shfl_xor(t2[e], m, 64);
      }
    if (l15 == 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int ch = cw + 4 * l4 + e;
        if (ch < hid) {
          p.stats_part[((size_t)row_id * 2 + 0) * hid + ch] = t1[e];
          p.stats_part[((size_t)row_id * 2 + 1) * hid + ch] = t2[e];
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------------------------------------------
// A operand of a diagonal-weight product whose two K slots carry the weights with indices wa / wb (kh * 3 + kw; -1: empty slot)
__device__ __forceinline__ bf16x8_t diag_frag2(const bf16_t* wd /*[9][C]*/, int C, int ch, int wa, int wb, int l15, int l4) {
  const int wi = (l4 >> 1) ? wb : wa;
  uint16_t wv = 0;
  if (wi >= 0 && ch < C && (l15 >> 3) == (l4 & 1)) wv = wd[(size_t)wi * C + ch].v;
  v8s f;
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] = (j == (l15 & 7)) ? (short)wv : (short)0;
  return __builtin_bit_cast(bf16x8_t, f);
}
__device__ __forceinline__ bf16x8_t tr_frag8(const bf16_t* lo, const bf16_t* hi) {
  const tr_v4s a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tr_v4s*)(lo));
  const tr_v4s b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tr_v4s*)(hi));
  return __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}

// Tile of the backward kernel = the OWN input pixels of a forward tile: 8 x 16 (both strides), as eight 16-pixel blocks.
//   stride 1: block pb = tile row pb; the dy tile is the 10 x 18 halo image of the output tile (origin (-1, -1)).
//   stride 2: 8 x 16 inputs <-> 4 x 8 outputs; dy tile 5 x 9 (origin (0, 0): an odd input row also reads the output row below).  A block
//             holds pixels of ONE parity class (ph, pw) — the taps that reach an input pixel depend on its parity only:
//             hi = 2 ho + kh - 1  =>  even row: kh = 1;  odd row: kh = 0 (ho = (hi + 1) / 2) and kh = 2 (ho = (hi - 1) / 2) — so the
//             16 pixels of a block share one weight operand: 1 + 1 + 1 + 2 MFMAs for the four classes (9 taps), 5 per 64 pixels.
template <int S> struct DxBwd;
template <> struct DxBwd<1> {
  static constexpr int DH = 10, DW = 18;
};
template <> struct DxBwd<2> {
  static constexpr int DH = 5, DW = 9;
};
// stride 2: weight indices of the two K slots of MFMA m of parity class cls = 2 ph + pw (-1: none)
__host__ __device__ constexpr int dx2_tap(int cls, int m, int slot) {
  return cls == 0 ? (slot == 0 ? 4 : -1)
       : cls == 1 ? (slot == 0 ? 3 : 5)
       : cls == 2 ? (slot == 0 ? 1 : 7)
       : (m == 0 ? (slot == 0 ? 0 : 2) : (slot == 0 ? 6 : 8));
}

struct DwxBwdParams {
  const bf16_t* x;         // [B*H*W][Cin] block input
  const bf16_t* w1;        // [hid][Cin]
  const float* in_stats;   // [4][hid] mean, invstd, scale, shift of the expansion BatchNorm
  const bf16_t* g_out;     // [B*Ho*Wo][hid] g2
  const bf16_t* y_out;     // [B*Ho*Wo][hid] y2 (nullptr: dy = g_out as it is)
  const float* ca;         // [hid] dy = ca * g_out + cb * y_out + cc
  const float* cb;
  const float* cc;
  const bf16_t* wd;        // [9][hid]
  bf16_t* g_in;            // [B*H*W][hid] g1 = dwconv^T(dy) * act'(bn1(y1))
  float* stats_part;       // [rows][2][hid] sum g1, sum g1 * xhat1
  float* dw_part;          // [rows][hid * 9]
  int act1;
  int B, H, W, Ho, Wo, hid;
  int tiles_h, tiles_w, ntiles, chunks;
};

#ifndef DXB_DBG
#define DXB_DBG 0  // developer builds (tools/build_variant.py ... -DDXB_DBG=bits): skip pieces of dwx_bwd_kernel to time them; results are WRONG when non-zero.
#endif             // 1 no g1 stores, 2 no tile loads after the first, 4 no exp / rcp (SiLU, SiLU'), 8 no depthwise dW product, 16 no stencil MFMAs, 32 no expansion MFMAs
template <int S, int CIN>
__global__ __launch_bounds__(256, 2) void dwx_bwd_kernel(DwxBwdParams p) {
  using TL = DxTile<S>;
  using TB = DxBwd<S>;
  constexpr int KS = (CIN + 31) / 32, XP = 32 * KS + 8, XC = CIN / 8;
  constexpr int ND = TB::DH * TB::DW, NLD = (ND * 8 + 255) / 256, NXL = (128 * XC + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  bf16_t* dt = reinterpret_cast<bf16_t*>(smem_raw);   // [ND][DX_AP]  dy of this workgroup's 64 channels (zero outside the image)
  bf16_t* zt = dt + ND * DX_AP;                        // [128][DX_AP] z = act(bn1(y1)) at the own pixels
  bf16_t* xo = zt + 128 * DX_AP;                       // [128][XP]    block input at the own pixels
  float* cst = reinterpret_cast<float*>(xo + 128 * XP);  // [9][64] ca, cb, cc; sc, sh, nsc, nsh, is, nmi

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
  const int lb = xcd_chunk_id(blockIdx.x, gridDim.x);
  const int chunk = lb % p.chunks, row_id = lb / p.chunks;
  const int c0 = chunk * DX_CC, cw = c0 + 16 * wave;
  const int hid = p.hid;
  const bool two_src = p.y_out != nullptr;

  for (int i = tid; i < 128 * XP / 8; i += 256) reinterpret_cast<uint4*>(xo)[i] = make_uint4(0, 0, 0, 0);
  for (int i = tid; i < 3 * DX_CC; i += 256) {
    const int v = i / DX_CC, c = c0 + (i - v * DX_CC);
    const float* src = v == 0 ? p.ca : (v == 1 ? p.cb : p.cc);
    cst[i] = (two_src && c < hid) ? src[c] : 0.f;
  }

  bf16x8_t w1f[KS], wdf[5];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    uint4 v = make_uint4(0, 0, 0, 0);
    const int ch = cw + l15, k = 32 * ks + 8 * l4;
    if (ch < hid && k < CIN) v = *reinterpret_cast<const uint4*>(p.w1 + (size_t)ch * CIN + k);
    w1f[ks] = __builtin_bit_cast(bf16x8_t, v);
  }
  if (S == 1) {
    // slot tap t' = halo offset (dh, dw) = (t' / 3, t' % 3) of the dy tile; its weight is w[2 - dh][2 - dw] = index 8 - t'
#pragma unroll
    for (int tp = 0; tp < 5; ++tp) wdf[tp] = diag_frag2(p.wd, hid, cw + l15, 8 - 2 * tp, 2 * tp + 1 < 9 ? 8 - (2 * tp + 1) : -1, l15, l4);
  } else {
    wdf[0] = diag_frag2(p.wd, hid, cw + l15, dx2_tap(0, 0, 0), dx2_tap(0, 0, 1), l15, l4);
    wdf[1] = diag_frag2(p.wd, hid, cw + l15, dx2_tap(1, 0, 0), dx2_tap(1, 0, 1), l15, l4);
    wdf[2] = diag_frag2(p.wd, hid, cw + l15, dx2_tap(2, 0, 0), dx2_tap(2, 0, 1), l15, l4);
    wdf[3] = diag_frag2(p.wd, hid, cw + l15, dx2_tap(3, 0, 0), dx2_tap(3, 0, 1), l15, l4);
    wdf[4] = diag_frag2(p.wd, hid, cw + l15, dx2_tap(3, 1, 0), dx2_tap(3, 1, 1), l15, l4);
  }
  // per-channel constants of the expansion BatchNorm, read from LDS where they are used (24 registers otherwise), in the form the
  // epilogue consumes: yh = sc y + sh, exponent of the sigmoid exp2(nsc y + nsh), xhat = is y + nmi
  for (int i = tid; i < DX_CC; i += 256) {
    const int c = c0 + i;
    const bool ok = c < hid;
    const float mu_ = ok ? p.in_stats[c] : 0.f, is_ = ok ? p.in_stats[(size_t)hid + c] : 0.f;
    const float sc_ = ok ? p.in_stats[(size_t)2 * hid + c] : 0.f, sh_ = ok ? p.in_stats[(size_t)3 * hid + c] : 0.f;
    float* d = cst + 3 * DX_CC + i;
    d[0] = sc_;
    d[DX_CC] = sh_;
    d[2 * DX_CC] = -1.4426950408889634f * sc_;
    d[3 * DX_CC] = -1.4426950408889634f * sh_;
    d[4 * DX_CC] = is_;
    d[5 * DX_CC] = -mu_ * is_;
  }
  const float* bn1 = cst + 3 * DX_CC + 16 * wave + 4 * l4;
  f32x2_t s1[2] = {{0.f, 0.f}, {0.f, 0.f}}, s2[2] = {{0.f, 0.f}, {0.f, 0.f}};
  f32x4_t dwa[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) dwa[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int slot = l4 >> 1;
  const int chof = 16 * wave + 8 * (l4 & 1);  // B-operand channel group of the stencil products
  const int q4 = 16 * wave + 4 * (l15 & 3), rsub = l15 >> 2;  // transpose reads: column group and pixel sub-row this lane addresses

  const int t_step = gridDim.x / p.chunks;
  uint4 gr[NLD], yr[NLD], xr[NXL];
  uint32_t dok = 0, xok = 0;
  auto load_tile = [&](int tix) __attribute__((always_inline)) {
    const int tw = tix % p.tiles_w, t1 = tix / p.tiles_w;
    const int th = t1 % p.tiles_h, b = t1 / p.tiles_h;
    const int ho0 = th * TL::OH, wo0 = tw * TL::OW;
    const int dh0 = S == 1 ? ho0 - 1 : ho0, dw0 = S == 1 ? wo0 - 1 : wo0;
    const int hi0 = ho0 * S, wi0 = wo0 * S;
    dok = 0;
    xok = 0;
#pragma unroll
    for (int it = 0; it < NLD; ++it) {
      const int i = tid + it * 256;
      const int px = i >> 3, cg = i & 7;
      const int pr = px / TB::DW, pc = px - pr * TB::DW;
      const int ho = dh0 + pr, wo = dw0 + pc;
      const bool ok = px < ND && ho >= 0 && ho < p.Ho && wo >= 0 && wo < p.Wo && c0 + cg * 8 < hid;
      dok |= (ok ? 1u : 0u) << it;
      const size_t o = ok ? (((size_t)b * p.Ho + ho) * p.Wo + wo) * hid + c0 + cg * 8 : (size_t)0;
      gr[it] = *reinterpret_cast<const uint4*>(p.g_out + o);
      yr[it] = *reinterpret_cast<const uint4*>((two_src ? p.y_out : p.g_out) + o);
    }
#pragma unroll
    for (int it = 0; it < NXL; ++it) {
      const int i = tid + it * 256;
      const int px = i / XC, ck = i - px * XC;
      const int hi = hi0 + (px >> 4), wi = wi0 + (px & 15);
      const bool ok = i < 128 * XC && hi < p.H && wi < p.W;
      xok |= (ok ? 1u : 0u) << it;
      xr[it] = *reinterpret_cast<const uint4*>(p.x + (ok ? (((size_t)b * p.H + hi) * p.W + wi) * CIN + ck * 8 : (size_t)0));
    }
  };

  int tix = row_id;
  if (tix < p.ntiles) load_tile(tix);
  const bool wave_full = cw + 16 <= hid;
  for (; tix < p.ntiles; tix += t_step) {
    const int tw = tix % p.tiles_w, t1 = tix / p.tiles_w;
    const int th = t1 % p.tiles_h, b = t1 / p.tiles_h;
    const int hi0 = th * TL::OH * S, wi0 = tw * TL::OW * S;
    bf16_t* gbase = p.g_in + (((size_t)b * p.H + hi0) * p.W + wi0) * hid;  // wave-uniform: the lanes add 32-bit offsets

    __syncthreads();  // every wave is done with the previous tile (first iteration: the zero fill and the coefficients are in place)
    const bool fast = wave_full && hi0 + 8 <= p.H && wi0 + 16 <= p.W;  // wave-uniform: every lane of every block stores
    {
      float ka[8], kb[8], kc[8];
      const int cg = tid & 7;
#pragma unroll
      for (int j = 0; j < 8; ++j) { ka[j] = 1.f; kb[j] = 0.f; kc[j] = 0.f; }
      if (two_src) {
        *reinterpret_cast<float4*>(ka) = *reinterpret_cast<const float4*>(cst + cg * 8);
        *reinterpret_cast<float4*>(ka + 4) = *reinterpret_cast<const float4*>(cst + cg * 8 + 4);
        *reinterpret_cast<float4*>(kb) = *reinterpret_cast<const float4*>(cst + DX_CC + cg * 8);
        *reinterpret_cast<float4*>(kb + 4) = *reinterpret_cast<const float4*>(cst + DX_CC + cg * 8 + 4);
        *reinterpret_cast<float4*>(kc) = *reinterpret_cast<const float4*>(cst + 2 * DX_CC + cg * 8);
        *reinterpret_cast<float4*>(kc + 4) = *reinterpret_cast<const float4*>(cst + 2 * DX_CC + cg * 8 + 4);
      }
#pragma unroll
      for (int it = 0; it < NLD; ++it) {
        const int i = tid + it * 256;
        const int px = i >> 3;
        if (px < ND) {
          V8<bf16_t> o;
          o.d = make_uint4(0, 0, 0, 0);
          if ((dok >> it) & 1u) {
            V8<bf16_t> gv, yv;
            gv.d = gr[it];
            yv.d = yr[it];
            if (two_src) {
              float g[8], y[8];
              v8_unpack(gv, g);
              v8_unpack(yv, y);
#pragma unroll
              for (int j = 0; j < 8; j += 2) {  // v_pk_fma_f32: two channels per instruction
                const f32x2_t r = f32x2_t{ka[j], ka[j + 1]} * f32x2_t{g[j], g[j + 1]} +
                                  (f32x2_t{kb[j], kb[j + 1]} * f32x2_t{y[j], y[j + 1]} + f32x2_t{kc[j], kc[j + 1]});
                g[j] = r[0];
                g[j + 1] = r[1];
              }
              v8_pack(g, o);
            } else {
              o = gv;
            }
          }
          *reinterpret_cast<uint4*>(dt + px * DX_AP + cg * 8) = o.d;
        }
      }
#pragma unroll
      for (int it = 0; it < NXL; ++it) {
        const int i = tid + it * 256;
        const int px = i / XC, ck = i - px * XC;
        if (i < 128 * XC) {
          uint4 v = xr[it];
          if (!((xok >> it) & 1u)) v = make_uint4(0, 0, 0, 0);
          *reinterpret_cast<uint4*>(xo + px * XP + ck * 8) = v;
        }
      }
    }
    __syncthreads();
    if (tix + t_step < p.ntiles && !(DXB_DBG & 2)) load_tile(tix + t_step);  // next tile's operands, in flight under this tile's arithmetic

    // ---- per 16-pixel block: y1 -> z, act', xhat;  dz = stencil^T(dy);  g1 = dz * act' ----
    // Unrolled by four: two waves per SIMD leave the scheduler little else to overlap a block's MFMAs and LDS reads with the previous block's
    // epilogue (stride-2 launches -6 ... -13 %, stride 1 -1 ... -5 % against the rolled loop).  Stride 2 with Cin <= 64 unrolls all eight
    // blocks: the parity class of a block becomes a compile-time constant and the dispatch on it disappears (another -6 ... -11 %); elsewhere
    // eight blocks' worth of hoisted operand reads push the persistent accumulators into scratch.
#pragma unroll (S == 2 && CIN <= 64 ? 8 : 4)
    for (int pb = 0; pb < 8; ++pb) {
      // tile coordinates of this lane's pixel
      int r, c;
      if (S == 1) {
        r = pb;
        c = l15;
      } else {
        const int cls = pb >> 1, half = pb & 1;
        r = (cls >> 1) + 2 * (2 * half + (l15 >> 3));
        c = (cls & 1) + 2 * (l15 & 7);
      }
      const int opx = r * 16 + c;
      const int hi = hi0 + r, wi = wi0 + c;
      const bool pok = hi < p.H && wi < p.W;

      f32x4_t a1 = {0.f, 0.f, 0.f, 0.f};
      const bf16_t* xrow = xo + opx * XP + 8 * l4;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        if (!(DXB_DBG & 32)) a1 = mfma16(w1f[ks], *reinterpret_cast<const bf16x8_t*>(xrow + 32 * ks), a1);

      f32x4_t a2 = {0.f, 0.f, 0.f, 0.f};
      if (DXB_DBG & 16) {
      } else if (S == 1) {
#pragma unroll
        for (int tp = 0; tp < 5; ++tp) {
          int t = 2 * tp + slot;
          t = t > 8 ? 8 : t;
          const int dpx = (r + t / 3) * TB::DW + c + t % 3;
          a2 = mfma16(wdf[tp], *reinterpret_cast<const bf16x8_t*>(dt + dpx * DX_AP + chof), a2);
        }
      } else {
        // the weight operand and the tap geometry depend on the parity class: wave-uniform dispatch on it
        const int cls = pb >> 1;
        const int rr = r >> 1, cc = c >> 1;
        auto tap_mma = [&](const bf16x8_t& wf, int ta, int tb) __attribute__((always_inline)) {
          int wi_ = slot ? tb : ta;
          wi_ = wi_ < 0 ? ta : wi_;  // empty slot: zero weights, any valid address
          const int kh = wi_ / 3, kw = wi_ - 3 * kh;
          const int dpx = (rr + (kh == 0 ? 1 : 0)) * TB::DW + cc + (kw == 0 ? 1 : 0);
          a2 = mfma16(wf, *reinterpret_cast<const bf16x8_t*>(dt + dpx * DX_AP + chof), a2);
        };
        if (cls == 0) tap_mma(wdf[0], dx2_tap(0, 0, 0), dx2_tap(0, 0, 1));
        else if (cls == 1) tap_mma(wdf[1], dx2_tap(1, 0, 0), dx2_tap(1, 0, 1));
        else if (cls == 2) tap_mma(wdf[2], dx2_tap(2, 0, 0), dx2_tap(2, 0, 1));
        else {
          tap_mma(wdf[3], dx2_tap(3, 0, 0), dx2_tap(3, 0, 1));
          tap_mma(wdf[4], dx2_tap(3, 1, 0), dx2_tap(3, 1, 1));
        }
      }

      // epilogue on packed pairs (the VALU bounds this kernel: v_pk_* do two channels per 4-cycle slot): SiLU and its derivative from one
      // sigmoid, xhat, g1 = dz * act'; y1 enters in fp32 exactly as in the forward kernel; statistics from the fp32 values
      uint32_t zw[2], gw[2];
      f32x2_t gq[2], xq2[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const f32x2_t cs = *reinterpret_cast<const f32x2_t*>(bn1 + 2 * h), csh = *reinterpret_cast<const f32x2_t*>(bn1 + DX_CC + 2 * h);
        const f32x2_t cn = *reinterpret_cast<const f32x2_t*>(bn1 + 2 * DX_CC + 2 * h), cnh = *reinterpret_cast<const f32x2_t*>(bn1 + 3 * DX_CC + 2 * h);
        const f32x2_t ci = *reinterpret_cast<const f32x2_t*>(bn1 + 4 * DX_CC + 2 * h), cm = *reinterpret_cast<const f32x2_t*>(bn1 + 5 * DX_CC + 2 * h);
        const f32x2_t y = {a1[2 * h], a1[2 * h + 1]};
        const f32x2_t yh = cs * y + csh;
        const f32x2_t t = cn * y + cnh;
        f32x2_t d = (DXB_DBG & 4) ? t : f32x2_t{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
        d = d + 1.0f;
        const f32x2_t sg = (DXB_DBG & 4) ? d : f32x2_t{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
        const f32x2_t z = yh * sg;
        const f32x2_t gp = sg * (yh * (1.0f - sg) + 1.0f);
        const f32x2_t g = f32x2_t{a2[2 * h], a2[2 * h + 1]} * gp;
        zw[h] = f2bf_pk(z[0], z[1]);
        gw[h] = f2bf_pk(g[0], g[1]);
        gq[h] = g;
        xq2[h] = ci * y + cm;
      }
      const uint32_t zm = pok ? 0xffffffffu : 0u;  // a pixel outside the image contributes nothing to dW
      *reinterpret_cast<uint2*>(zt + opx * DX_AP + 16 * wave + 4 * l4) = make_uint2(zw[0] & zm, zw[1] & zm);
      const int goff = (int)(__umul24(__umul24(r, p.W) + c, hid)) + cw + 4 * l4;  // elements from the tile's first pixel
      if (fast) {  // wave-uniform fast path: no per-lane bounds test
        if (!(DXB_DBG & 1)) *reinterpret_cast<uint2*>(gbase + goff) = make_uint2(gw[0], gw[1]);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          s1[h] += gq[h];
          s2[h] += gq[h] * xq2[h];
        }
      } else if (pok && cw + 4 * l4 < hid) {
        *reinterpret_cast<uint2*>(gbase + goff) = make_uint2(gw[0], gw[1]);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          s1[h] += gq[h];
          s2[h] += gq[h] * xq2[h];
        }
      }
    }
    wave_lds_sync();

    // ---- depthwise weight gradient: diagonal of dy_shifted^T z, contraction over the tile's own pixels ----
    if (DXB_DBG & 8) {
    } else if (S == 1) {
      // K step ks = tile rows 2 ks, 2 ks + 1; K slot (l4, j) <-> pixel (row 2 ks + (j >> 2), column 4 l4 + (j & 3))
#pragma unroll 1
      for (int ks = 0; ks < 4; ++ks) {
        const int col = 4 * l4 + rsub;
        const bf16_t* zlo = zt + ((2 * ks) * 16 + col) * DX_AP + q4;
        const bf16x8_t zb = tr_frag8(zlo, zlo + 16 * DX_AP);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const bf16_t* dlo = dt + ((2 * ks + t / 3) * TB::DW + col + t % 3) * DX_AP + q4;
          dwa[8 - t] = mfma16(tr_frag8(dlo, dlo + TB::DW * DX_AP), zb, dwa[8 - t]);
          if (t % 3 == 2) __builtin_amdgcn_sched_barrier(0);  // at most three taps' operand reads in flight (registers)
        }
      }
    } else {
      // one K step per parity class: K slot (l4, j) <-> class pixel (rr = l4, cc = j)
#pragma unroll
      for (int cls = 0; cls < 4; ++cls) {
        const int ph = cls >> 1, pw = cls & 1;
        const bf16_t* zlo = zt + ((ph + 2 * l4) * 16 + pw + 2 * rsub) * DX_AP + q4;
        const bf16x8_t zb = tr_frag8(zlo, zlo + 8 * DX_AP);
#pragma unroll
        for (int m = 0; m < (cls == 3 ? 2 : 1); ++m)
#pragma unroll
          for (int sl = 0; sl < 2; ++sl) {
            const int wi_ = dx2_tap(cls, m, sl);
            if (wi_ >= 0) {
              const int kh = wi_ / 3, kw = wi_ - 3 * kh;
              const bf16_t* dlo = dt + ((l4 + (kh == 0 ? 1 : 0)) * TB::DW + rsub + (kw == 0 ? 1 : 0)) * DX_AP + q4;
              dwa[wi_] = mfma16(tr_frag8(dlo, dlo + 4 * DX_AP), zb, dwa[wi_]);
            }
          }
      }
    }
  }

  // ---- workgroup results: statistics and the diagonal of the dW accumulators ----
  float t1[4] = {s1[0][0], s1[0][1], s1[1][0], s1[1][1]}, t2[4] = {s2[0][0], s2[0][1], s2[1][0], s2[1][1]};
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) {
      t1[e] += __shfl_xor(t1[e], m, 64);
      t2[e] += __shfl_xor(t2[e], m, 64);
    }
  if (l15 == 0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int ch = cw + 4 * l4 + e;
      if (ch < hid) {
        p.stats_part[((size_t)row_id * 2 + 0) * hid + ch] = t1[e];
        p.stats_part[((size_t)row_id * 2 + 1) * hid + ch] = t2[e];
      }
    }
  }
  if (l4 == (l15 >> 2) && cw + l15 < hid) {  // D[c][c'] lives at (rows 4 l4 + e, column l15): the diagonal element of channel l15
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const float v = (l15 & 3) == 0 ? dwa[t][0] : ((l15 & 3) == 1 ? dwa[t][1] : ((l15 & 3) == 2 ? dwa[t][2] : dwa[t][3]));
      p.dw_part[(size_t)row_id * hid * 9 + (size_t)(cw + l15) * 9 + t] = v;
    }
  }
}

// (sum y1, sum y1^2) per expansion channel from the Gram matrix G = x^T x [K][Gp] and the column sums s = 1^T x of the narrow input:
// y1 = x W1^T  =>  sum y1[c] = W1[c] . s,  sum y1[c]^2 = W1[c] G W1[c]^T.  One workgroup per channel; double accumulation (the
// quadratic form cancels when the terms of y1 do).  Output = one partial-statistics row [2][hid] for cvh_bn_finalize.
__global__ __launch_bounds__(128) void gram_bn_stats_kernel(const float* __restrict__ G, const float* __restrict__ s, const bf16_t* __restrict__ w1,
                                                            float* __restrict__ part, int hid, int K, int Gp) {
  __shared__ float wsh[128];
  __shared__ double red[2][128];
  const int c = blockIdx.x, i = threadIdx.x;
  wsh[i] = i < K ? bf2f(w1[(size_t)c * K + i].v) : 0.f;
  __syncthreads();
  double m = 0.0, q = 0.0;
  if (i < K) {
    double r = 0.0;
    for (int j = 0; j < K; ++j) r += (double)G[(size_t)j * Gp + i] * (double)wsh[j];  // G is symmetric: column i read along a row
    q = (double)wsh[i] * r;
    m = (double)wsh[i] * (double)s[i];
  }
  red[0][i] = m;
  red[1][i] = q;
  __syncthreads();
  for (int st = 64; st > 0; st >>= 1) {
    if (i < st) {
      red[0][i] += red[0][i + st];
      red[1][i] += red[1][i + st];
    }
    __syncthreads();
  }
  if (i == 0) {
    part[c] = (float)red[0][0];
    part[hid + c] = (float)red[1][0];
  }
}

// out[i] = alpha * a[i] + (b ? b[i] : 0): column sums of an InvertedResidual output without a pass over it — the block ends in a train-mode
// BatchNorm, so sum_rows bn3(y3)[c] = rows * beta3[c] exactly (+ the column sums of the input on the residual path)
__global__ void axpb_kernel(const float* __restrict__ a, float alpha, const float* __restrict__ b, float* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = alpha * a[i] + (b != nullptr ? b[i] : 0.f);
}

template <int S, int CIN> size_t dwx_fwd_smem() {
  using TL = DxTile<S>;
  constexpr int NPB = (TL::IH * TL::IW + 15) / 16;
  return (size_t)NPB * 16 * (dx_xbuf<CIN>() * dx_xp<CIN>() + DX_CC) * 2;
}

template <int S, int CIN> size_t dwx_bwd_smem() {
  return ((size_t)(DxBwd<S>::DH * DxBwd<S>::DW + 128) * DX_AP + (size_t)128 * (32 * ((CIN + 31) / 32) + 8)) * 2 + 9 * DX_CC * 4;
}

bool dwx_cin_ok(int Cin) { return Cin == 16 || Cin == 32 || Cin == 64 || Cin == 96 || Cin == 128; }

int dwx_plan(int B, int Ho, int Wo, int hid, int stride, int* tiles_h, int* tiles_w, int* chunks) {
  const int OH = stride == 1 ? 8 : 4, OW = stride == 1 ? 16 : 8;
  *tiles_h = (Ho + OH - 1) / OH;
  *tiles_w = (Wo + OW - 1) / OW;
  *chunks = (hid + DX_CC - 1) / DX_CC;
  const long long ntiles = (long long)B * *tiles_h * *tiles_w;
  int rows = 2048 / *chunks;  // ~2048 workgroups in XCD-contiguous round-robin order (up to 4 resident per CU); partial rows <= 1024
  if (rows > 1024) rows = 1024;
  if (rows < 32) rows = 32;
  if (rows > ntiles) rows = (int)ntiles;
  return rows;
}

}  // namespace

// strip-streaming forward kernel (csrc/dwxs.hip)
bool dwxs_fwd_plan(int B, int H, int W, int Cin, int hid, int stride, int* rows, int* nstrip, int* nseg, int* RS);
int dwxs_fwd_launch(const void* x, const void* w1, const float* scale1, const float* shift1, const void* wd, void* y2, float* stats_part, int B, int H,
                    int W, int Ho, int Wo, int Cin, int hid, int stride, hipStream_t st);

/* rows of the partial-statistics buffer of cvh_dwx_fwd */
extern "C" int cvh_dwx_fwd_rows(int B, int H, int W, int Cin, int hid, int stride) {
  if (B <= 0 || hid <= 0 || (hid % 8) || (stride != 1 && stride != 2)) return -2;
  int rows, ns, ng, rs;
  if (dwxs_fwd_plan(B, H, W, Cin, hid, stride, &rows, &ns, &ng, &rs)) return rows;
  int th, tw, ch;
  return dwx_plan(B, (H + 2 - 3) / stride + 1, (W + 2 - 3) / stride + 1, hid, stride, &th, &tw, &ch);
}

/* rows of the partial-statistics / partial-dW buffers of cvh_dwx_bwd (= workgroups per 64-channel chunk) */
extern "C" int cvh_dwx_rows(int B, int Ho, int Wo, int hid, int stride) {
  if (B <= 0 || hid <= 0 || (hid % 8) || (stride != 1 && stride != 2)) return -2;
  int th, tw, ch;
  return dwx_plan(B, Ho, Wo, hid, stride, &th, &tw, &ch);
}

extern "C" int cvh_gram_bn_stats(const float* G, const float* s, const void* w1, float* part, int hid, int K, int Gp, void* stream) {
  if (K <= 0 || K > 128 || hid <= 0) return -2;
  hipLaunchKernelGGL(gram_bn_stats_kernel, dim3(hid), dim3(128), 0, (hipStream_t)stream, G, s, reinterpret_cast<const bf16_t*>(w1), part, hid, K,
                     Gp);
  CVH_CHECK_LAUNCH();
  return 0;
}

extern "C" int cvh_axpb(const float* a, float alpha, const float* b, float* out, int n, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(axpb_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, a, alpha, b, out, n);
  CVH_CHECK_LAUNCH();
  return 0;
}

extern "C" int cvh_dwx_fwd(int dtype, const void* x, const void* w1, const float* scale1, const float* shift1, int act1, const void* wd, void* y2,
                           float* stats_part, int B, int H, int W, int Ho, int Wo, int Cin, int hid, int stride, void* stream) {
  if (dtype != CVH_DT_BF16) return -1;
  if (act1 != CVH_ACT_SILU) return -2;
  if (!dwx_cin_ok(Cin) || (stride == 1 && Cin > 64) || hid <= 0 || (hid % 8) || (stride != 1 && stride != 2)) return -2;
  if (Ho != (H + 2 - 3) / stride + 1 || Wo != (W + 2 - 3) / stride + 1) return -2;
  if (B <= 0) return 0;
  {
    int rows, ns, ng, rs;
    if (dwxs_fwd_plan(B, H, W, Cin, hid, stride, &rows, &ns, &ng, &rs)) {
      cvh_family_tally(0, ((long long)B * H * W * Cin + (long long)B * Ho * Wo * hid) * 2);
      return dwxs_fwd_launch(x, w1, scale1, shift1, wd, y2, stats_part, B, H, W, Ho, Wo, Cin, hid, stride, (hipStream_t)stream);
    }
  }
  DwxFwdParams p;
  p.x = reinterpret_cast<const bf16_t*>(x); p.w1 = reinterpret_cast<const bf16_t*>(w1); p.scale1 = scale1; p.shift1 = shift1;
  p.wd = reinterpret_cast<const bf16_t*>(wd); p.y2 = reinterpret_cast<bf16_t*>(y2); p.stats_part = stats_part; p.act1 = act1;
  p.B = B; p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo; p.hid = hid;
  const int rows = dwx_plan(B, Ho, Wo, hid, stride, &p.tiles_h, &p.tiles_w, &p.chunks);
  p.ntiles = B * p.tiles_h * p.tiles_w;
#ifdef CVH_DWX_DBG
  p.dbg = cvh_tune_get(17);
#endif
  cvh_family_tally(0, ((long long)B * H * W * Cin + (long long)B * Ho * Wo * hid) * 2);
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(rows * p.chunks);
#define DX_FWD(S_, C_)                                                                                                                    \
  do {                                                                                                                                    \
    const size_t smem = dwx_fwd_smem<S_, C_>();                                                                                           \
    static DynSmemAttr attr;                                                                                                              \
    if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(dwx_fwd_kernel<S_, C_>), smem); e != hipSuccess) return (int)e;               \
    hipLaunchKernelGGL((dwx_fwd_kernel<S_, C_>), grid, dim3(256), smem, st, p);                                                           \
  } while (0)
#define DX_FWD_S(C_)              \
  do {                            \
    if (stride == 1) DX_FWD(1, C_); \
    else DX_FWD(2, C_);           \
  } while (0)
  switch (Cin) {
    case 16: DX_FWD_S(16); break;
    case 32: DX_FWD_S(32); break;
    case 64: DX_FWD_S(64); break;
    case 96: DX_FWD_S(96); break;
    default: DX_FWD_S(128); break;
  }
#undef DX_FWD_S
#undef DX_FWD
  CVH_CHECK_LAUNCH();
  return 0;
}

extern "C" int cvh_dwx_bwd(int dtype, const void* x, const void* w1, const float* in_stats, int act1, const void* g_out, const void* y_out,
                           const float* ca, const float* cb, const float* cc, const void* wd, void* g_in, float* stats_part, float* dw_part,
                           int B, int H, int W, int Ho, int Wo, int Cin, int hid, int stride, void* stream) {
  if (dtype != CVH_DT_BF16) return -1;
  if (act1 != CVH_ACT_SILU) return -2;
  if (!dwx_cin_ok(Cin) || (stride == 1 && Cin > 64) || hid <= 0 || (hid % 8) || (stride != 1 && stride != 2)) return -2;
  if (Ho != (H + 2 - 3) / stride + 1 || Wo != (W + 2 - 3) / stride + 1) return -2;
  if (in_stats == nullptr || stats_part == nullptr || dw_part == nullptr) return -2;
  if (y_out != nullptr && (ca == nullptr || cb == nullptr || cc == nullptr)) return -2;
  if (B <= 0) return 0;
  DwxBwdParams p;
  p.x = reinterpret_cast<const bf16_t*>(x); p.w1 = reinterpret_cast<const bf16_t*>(w1); p.in_stats = in_stats; p.act1 = act1;
  p.g_out = reinterpret_cast<const bf16_t*>(g_out); p.y_out = reinterpret_cast<const bf16_t*>(y_out); p.ca = ca; p.cb = cb; p.cc = cc;
  p.wd = reinterpret_cast<const bf16_t*>(wd); p.g_in = reinterpret_cast<bf16_t*>(g_in); p.stats_part = stats_part; p.dw_part = dw_part;
  p.B = B; p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo; p.hid = hid;
  const int rows = dwx_plan(B, Ho, Wo, hid, stride, &p.tiles_h, &p.tiles_w, &p.chunks);
  p.ntiles = B * p.tiles_h * p.tiles_w;
  cvh_family_tally(1, ((long long)B * H * W * (Cin + hid) + (long long)B * Ho * Wo * hid * (y_out != nullptr ? 2 : 1)) * 2);
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(rows * p.chunks);
#define DX_BWD(S_, C_)                                                                                                                    \
  do {                                                                                                                                    \
    const size_t smem = dwx_bwd_smem<S_, C_>();                                                                                           \
    static DynSmemAttr attr;                                                                                                              \
    if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(dwx_bwd_kernel<S_, C_>), smem); e != hipSuccess) return (int)e;               \
    hipLaunchKernelGGL((dwx_bwd_kernel<S_, C_>), grid, dim3(256), smem, st, p);                                                           \
  } while (0)
#define DX_BWD_S(C_)              \
  do {                            \
    if (stride == 1) DX_BWD(1, C_); \
    else DX_BWD(2, C_);           \
  } while (0)
  switch (Cin) {
    case 16: DX_BWD_S(16); break;
    case 32: DX_BWD_S(32); break;
    case 64: DX_BWD_S(64); break;
    case 96: DX_BWD_S(96); break;
    default: DX_BWD_S(128); break;
  }
#undef DX_BWD_S
#undef DX_BWD
  CVH_CHECK_LAUNCH();
  return 0;
}
