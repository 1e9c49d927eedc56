// dwxs kernels: the strip-streaming form of the dwx FORWARD kernel (csrc/dwx.hip; cvnets/modules/mobilenetv2.py:180-207,231-235: 1x1 expansion
// conv -> BatchNorm -> SiLU -> depthwise 3x3 conv (pad 1, stride 1 / 2), y1 = x W1^T never in HBM), bf16.
//
// What the tile kernel of dwx.hip measured (profiles/r06_dwx_phases.txt): its phases — x tile load, expansion + BatchNorm + SiLU, stencil,
// stores — are ADDITIVE (1231 us = 187 + 257 + 300 + 350 + 120 for the 64 -> 256 channel launch): every wave of a CU is in the same phase
// at the same time, each phase is a latency chain bound by a different unit, and the x load (issued one expansion phase ahead) is exposed.
// This kernel changes the structure instead of the tuning:
//   * work unit = a vertical STRIP of the image (16 output columns, stride 2: 8), streamed top to bottom in chunks of 8 activation rows;
//     the activated rows a chunk shares with the next one stay in LDS (three 16-pixel blocks copied to the top of the wave's activation
//     image), so the halo is recomputed sideways only: 18 / 16 = 1.125 SiLU evaluations per pixel instead of 180 / 128 = 1.41;
//   * a dedicated LOADER wave (wave 4) brings the next chunk's x rows straight into LDS (global_load_lds, no staging registers) while the
//     four compute waves work on the current one: a whole chunk of lead instead of one expansion phase, and its vmcnt never counts a
//     result store (a compute wave that waits for its own prefetch also drains its stores: vmcnt counts both).  The dense x rows are
//     XOR-swizzled through the SOURCE address (the LDS destination of the DMA is lane-linear), conflict-free for the operand reads;
//   * expansion and stencil are ONE software pipeline per wave: the expansion of 16-pixel block k + 6 (matrix pipe, then the
//     transcendental-bound BatchNorm + SiLU epilogue on the VALU) is issued next to the stencil of output row k (five MFMAs, light VALU,
//     one result store), the operand reads of step k + 1 ahead of the epilogues of step k; result stores leave one per step instead of in a
//     burst at the end of a tile;
//   * no integer division in the chunk loop (units are decoded once per strip), one workgroup barrier per chunk.
// Geometry in "p space": a strip row is 18 pixels (16 + halo; stride 2: 17 + 1 unused), p = 18 * row + column + OFF; a chunk is 144
// pixels = nine 16-pixel blocks 3 .. 11, blocks 0 .. 2 hold the rows carried over from the previous chunk (or the strip's first rows).
#include "common.hpp"
#include "cvnets_hip.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef short xs_v8s __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(1))) const void* xs_gptr_t;
typedef __attribute__((address_space(3))) void* xs_lptr_t;

constexpr int XS_NPX = 192;   // pixels of an x buffer / of a wave's activation image: 48 carried + 144 new
constexpr int XS_PRO = 48;    // carried pixels (blocks 0 .. 2)
constexpr int XS_ATB = 32;    // bytes per pixel of a wave's activation image (its 16 channels)

__device__ __forceinline__ f32x4_t xs_mfma16(bf16x8_t a, bf16x8_t b, f32x4_t c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ void xs_glds16(const void* g, unsigned char* l) { __builtin_amdgcn_global_load_lds((xs_gptr_t)g, (xs_lptr_t)l, 16, 0, 0); }

// A operand of the diagonal-weight product for tap pair tp: A[c = l15][k = 8 * l4 + j] = (16 * slot + c' == k ? w[2 tp + slot][c] : 0)
__device__ __forceinline__ bf16x8_t xs_diag_frag(const bf16_t* wd /*[9][C]*/, int C, int ch, int tp, int l15, int l4) {
  const int tap = 2 * tp + (l4 >> 1);
  uint16_t wv = 0;
  if (tap < 9 && ch < C && (l15 >> 3) == (l4 & 1)) wv = wd[(size_t)tap * C + ch].v;
  xs_v8s f;
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] = (j == (l15 & 7)) ? (short)wv : (short)0;
  return __builtin_bit_cast(bf16x8_t, f);
}

struct DwxsParams {
  const bf16_t* x;      // [B*H*W][Cin]
  const bf16_t* w1;     // [hid][Cin]
  const float* scale1;  // [hid] BatchNorm of the expansion: scale, shift
  const float* shift1;
  const bf16_t* wd;     // [9][hid]
  bf16_t* y2;           // [B*Ho*Wo][hid]
  float* stats_part;    // [rows][2][hid] or nullptr
  int B, H, W, Ho, Wo, hid;
  int chunks, rows;     // 64-channel chunks; workgroups per chunk (= partial-statistics rows)
  int nstrip, nseg, RS, units;  // strips per image row, row segments per strip, activation rows per segment (multiple of 8), B * nstrip * nseg
};
#ifndef XS_DBG
#define XS_DBG 0  // developer builds (tools/build_variant.py ... -DXS_DBG=bits): skip pieces to time them; results are WRONG when non-zero
#endif

// position (16-byte chunk) of source chunk c of pixel px inside the dense LDS row: conflict-free ds_read_b128 in the 16x16x32 B-operand layout
// (lane = pixel l15, chunk l4 [+ 4 ks]) — checked by enumeration of the hardware's lane groups (tools/experiments/lds_swizzle.py)
template <int CIN> __device__ __forceinline__ int xs_swz(int c, int px) { return CIN == 64 ? (c ^ (px & 7)) : (CIN == 32 ? (c ^ ((px >> 1) & 3)) : c); }

template <int S, int CIN, int OCC, int NXB>
__global__ __launch_bounds__(320, (5 * OCC + 3) / 4) void dwxs_fwd_kernel(DwxsParams p) {
  constexpr int KS = (CIN + 31) / 32, XC = CIN / 8, ROWB = CIN * 2;  // K steps, 16-byte chunks and bytes per pixel
  constexpr int XB = XS_NPX * ROWB;                                   // bytes per x buffer
  constexpr int OFF = S == 1 ? 12 : 30;                               // p of (first carried row, first column)
  constexpr int NI = XS_NPX * XC / 64, NI0 = XS_PRO * XC / 64;        // loader instructions per buffer / skipped when the carried rows are not needed
  constexpr int ATW = XS_NPX * XS_ATB;                                // bytes of a wave's activation image
  extern __shared__ __attribute__((aligned(1024))) unsigned char xs_smem[];
  unsigned char* xb0 = xs_smem;            // NXB x [192][ROWB] block input, swizzled
  unsigned char* at0 = xs_smem + NXB * XB;   // 4 x [192][32 B]   act(bn1(y1)), one image per compute wave

  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lb = xcd_chunk_id(blockIdx.x, gridDim.x);
  const int chunk = lb % p.chunks, row_id = lb / p.chunks;
  const int Q = p.RS >> 3;

  if (wave == 4) {
    // ------------------------------------------------------------------------------------------------------------------------------
    // loader: x rows of the next chunk -> LDS.  Slot e = 64 i + lane of instruction i is pixel e / XC, chunk position e % XC.
    // Pixels outside the image (the conv's zero padding applies to the ACTIVATED tensor: the compute waves mask them) and the unused
    // slots read a clamped, valid address.
    // ------------------------------------------------------------------------------------------------------------------------------
    int tab[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int e = 64 * i + lane;
      const int px = e / XC, cp = e % XC;
      const int pp = px - OFF;
      const int rr = pp >= 0 ? (pp * 3641) >> 16 : 0, cc = pp >= 0 ? pp - 18 * rr : 0;
      tab[i] = (rr << 16) | (cc << 8) | xs_swz<CIN>(cp, px);  // XOR swizzle: source chunk = position ^ f(px)
    }
    // chunk g of this workgroup (all strips concatenated) lives in buffer g % NXB; the loader runs NXB - 1 chunks ahead of the compute waves:
    // memory latency under this kernel's own store traffic is several microseconds (profiles/r06_dwxs_phases.txt: with the arithmetic
    // compiled out and one chunk of lead the kernel takes as long as with it).  Every chunk issues exactly NI instructions when NXB > 2
    // (a non-first chunk re-fetches the carried rows it does not need) so that "chunk g has landed" is the immediate s_waitcnt
    // vmcnt((NXB - 2) * NI): loads return in order and this wave issues nothing else.
    constexpr bool FULL = NXB > 2;
    const int total = ((p.units - row_id + p.rows - 1) / p.rows) * Q;
    int iu = row_id, iq = 0, ibuf = 0;  // issue iterator: strip, chunk inside the strip, buffer
    auto issue = [&]() __attribute__((always_inline)) {
      const int seg = iu % p.nseg, t1 = iu / p.nseg;
      const int strip = t1 % p.nstrip, b = t1 / p.nstrip;
      const int colb = strip * 16 - 1, rowb = seg * p.RS - 1 + 8 * iq;
      const bf16_t* img = p.x + (size_t)b * p.H * p.W * CIN;
      unsigned char* dst = xb0 + ibuf * XB;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        if ((FULL || i >= NI0 || iq == 0) && !(XS_DBG & 2)) {
          const int e = tab[i];
          int r = rowb + (e >> 16), c = colb + ((e >> 8) & 0xff);
          r = r < 0 ? 0 : (r >= p.H ? p.H - 1 : r);
          c = c < 0 ? 0 : (c >= p.W ? p.W - 1 : c);
          xs_glds16(img + ((size_t)(r * p.W + c) * CIN + (e & 0xff) * 8), dst + i * 1024);
        }
      }
      if (++iq == Q) { iq = 0; iu += p.rows; }
      ibuf = ibuf + 1 == NXB ? 0 : ibuf + 1;
    };
    for (int g = 0; g < NXB - 1 && g < total; ++g) issue();
    for (int g = 0; g < total; ++g) {
      if (NXB > 2 && g + NXB - 2 < total) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NXB - 2) * NI) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();  // chunk g has landed; every compute wave is done with chunk g - 1
      if (g + NXB - 1 < total) issue();
    }
    __syncthreads();  // the compute waves' barrier behind the last chunk
    return;
  }

  // --------------------------------------------------------------------------------------------------------------------------------
  // compute waves: wave w owns channels 16 w .. 16 w + 15 of the workgroup's 64-channel chunk in every step
  // --------------------------------------------------------------------------------------------------------------------------------
  const int cw = chunk * 64 + 16 * wave;
  const int hid = p.hid;
  unsigned char* atw = at0 + wave * ATW;

  bf16x8_t w1f[KS], wdf[5];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    uint4 v = make_uint4(0, 0, 0, 0);
    const int k = 32 * ks + 8 * l4;
    if (k < CIN) v = *reinterpret_cast<const uint4*>(p.w1 + (size_t)(cw + l15) * CIN + k);
    w1f[ks] = __builtin_bit_cast(bf16x8_t, v);
  }
#pragma unroll
  for (int tp = 0; tp < 5; ++tp) wdf[tp] = xs_diag_frag(p.wd, hid, cw + l15, tp, l15, l4);
  // BatchNorm coefficients of the lane's 4 channels as packed pairs; the sigmoid's exponent gets its own pre-scaled pair (dwx.hip)
  f32x2_t sc[2], sh[2], nsc[2], nsh[2];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float a = p.scale1[cw + 4 * l4 + e], c = p.shift1[cw + 4 * l4 + e];
    sc[e >> 1][e & 1] = a;
    sh[e >> 1][e & 1] = c;
    nsc[e >> 1][e & 1] = -1.4426950408889634f * a;
    nsh[e >> 1][e & 1] = -1.4426950408889634f * c;
  }
  f32x2_t s1[2] = {{0.f, 0.f}, {0.f, 0.f}}, s2[2] = {{0.f, 0.f}, {0.f, 0.f}};

  // per-lane LDS byte offsets: x operand reads (block 0), activation writes (block 0), stencil operand reads (first row / row pair)
  int xl[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int c = CIN == 16 ? (l4 & 1) : 4 * ks + l4;  // Cin = 16: K = 32 is half padding (zero weights): any finite operand
    xl[ks] = l15 * ROWB + xs_swz<CIN>(c, l15) * 16;
  }
  const int awr = l15 * XS_ATB + 8 * l4;
  int toff[5];
#pragma unroll
  for (int tp = 0; tp < 5; ++tp) {
    int tap = 2 * tp + (l4 >> 1);
    tap = tap > 8 ? 8 : tap;  // idle slot: zero weights, any valid address
    const int kh = tap / 3, kw = tap % 3;
    const int px = S == 1 ? kh * 18 + kw + l15 + OFF : (2 * (l15 >> 3) + kh) * 18 + 2 * (l15 & 7) + kw + OFF;
    toff[tp] = px * XS_ATB + 16 * (l4 & 1);
  }
  // result addressing: lane offset (elements) from the strip's first output pixel; one running row offset
  const int yl = (S == 1 ? l15 : (l15 >> 3) * p.Wo + (l15 & 7)) * hid + 4 * l4;
  const int ystep = (S == 1 ? 1 : 2) * p.Wo * hid;

  // ---- the pipeline pieces -----------------------------------------------------------------------------------------------------------
  auto rdE = [&](bf16x8_t(&bq)[KS], const unsigned char* xs, int j) __attribute__((always_inline)) {
    if (XS_DBG & 16) return;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) bq[ks] = *reinterpret_cast<const bf16x8_t*>(xs + xl[ks] + j * 16 * ROWB);
  };
  auto mmE = [&](const bf16x8_t(&bq)[KS]) __attribute__((always_inline)) {
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    if (XS_DBG & 16) return acc;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) acc = xs_mfma16(w1f[ks], bq[ks], acc);
    return acc;
  };
  // SiLU(bn(y1)) of 4 channels in 2 x (2 v_pk_fma, 2 v_exp, v_pk_add, 2 v_rcp, v_pk_mul, v_cvt_pk); y1 enters in fp32
  auto epE = [&](int j, const f32x4_t& acc, uint32_t vm) __attribute__((always_inline)) {
    if (XS_DBG & 16) return;
    uint32_t w[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const f32x2_t y = {acc[2 * h], acc[2 * h + 1]};
      const f32x2_t yh = sc[h] * y + sh[h];
      const f32x2_t t = nsc[h] * y + nsh[h];
      f32x2_t d = t, r = t;
      if (!(XS_DBG & 4)) {
        d = f32x2_t{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
        d = d + 1.0f;
        r = f32x2_t{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
      }
      const f32x2_t v = yh * r;
      w[h] = f2bf_pk(v[0], v[1]);
    }
    const uint32_t m = (uint32_t)__builtin_amdgcn_sbfe((int)vm, j, 1);  // all ones inside the image, zero outside
    *reinterpret_cast<uint2*>(atw + awr + j * 16 * XS_ATB) = make_uint2(w[0] & m, w[1] & m);
  };
  auto rdS = [&](bf16x8_t(&fq)[5], int k) __attribute__((always_inline)) {  // k: output row of the chunk (stride 2: row pair)
    if (XS_DBG & 8) return;
#pragma unroll
    for (int tp = 0; tp < 5; ++tp) fq[tp] = *reinterpret_cast<const bf16x8_t*>(atw + toff[tp] + k * (S == 1 ? 1 : 4) * 18 * XS_ATB);
  };
  bf16_t* ybase = p.y2;
  int yo = 0;
  auto mmS = [&](const bf16x8_t(&fq)[5], f32x2_t& lo, f32x2_t& hi) __attribute__((always_inline)) {
    f32x4_t a = {0.f, 0.f, 0.f, 0.f}, c = {0.f, 0.f, 0.f, 0.f};
    if (!(XS_DBG & 8)) {
      a = xs_mfma16(wdf[0], fq[0], a);
      c = xs_mfma16(wdf[1], fq[1], c);
      a = xs_mfma16(wdf[2], fq[2], a);
      c = xs_mfma16(wdf[3], fq[3], c);
      a = xs_mfma16(wdf[4], fq[4], a);
    }
    lo = f32x2_t{a[0], a[1]} + f32x2_t{c[0], c[1]};
    hi = f32x2_t{a[2], a[3]} + f32x2_t{c[2], c[3]};
  };
  auto epS = [&](const f32x2_t& lo, const f32x2_t& hi) __attribute__((always_inline)) {
    if (!(XS_DBG & 1)) *reinterpret_cast<uint2*>(ybase + yo) = make_uint2(f2bf_pk(lo[0], lo[1]), f2bf_pk(hi[0], hi[1]));
    yo += ystep;
    s1[0] += lo;
    s1[1] += hi;
    s2[0] += lo * lo;
    s2[1] += hi * hi;
  };
  // one pipeline step: stencil of row k next to the expansion of block j (from x buffer xs, mask word vm)
  auto stepSE = [&](int k, const unsigned char* xs, int j, uint32_t vm) __attribute__((always_inline)) {
    bf16x8_t fq[5], bq[KS];
    rdS(fq, k);
    rdE(bq, xs, j);
    const f32x4_t e = mmE(bq);
    f32x2_t lo, hi;
    mmS(fq, lo, hi);
    epE(j, e, vm);
    epS(lo, hi);
  };
  auto stepE = [&](const unsigned char* xs, int j, uint32_t vm) __attribute__((always_inline)) {
    bf16x8_t bq[KS];
    rdE(bq, xs, j);
    epE(j, mmE(bq), vm);
  };
  auto stepS = [&](int k) __attribute__((always_inline)) {
    bf16x8_t fq[5];
    rdS(fq, k);
    f32x2_t lo, hi;
    mmS(fq, lo, hi);
    epS(lo, hi);
  };
  // rows carried into the next chunk: blocks 9 .. 11 -> 0 .. 2 of this wave's activation image (1536 bytes)
  auto carry = [&]() __attribute__((always_inline)) {
    const uint4 v0 = *reinterpret_cast<const uint4*>(atw + 144 * XS_ATB + lane * 16);
    const uint4 v1 = *reinterpret_cast<const uint4*>(atw + 144 * XS_ATB + 1024 + (lane & 31) * 16);
    *reinterpret_cast<uint4*>(atw + lane * 16) = v0;
    if (lane < 32) *reinterpret_cast<uint4*>(atw + 1024 + lane * 16) = v1;
  };
  // pixels of the strip that lie inside the image: columns per strip, rows per chunk (bit j = block j of this lane's pixel)
  auto col_mask = [&](int colb) __attribute__((always_inline)) -> uint32_t {
    if (colb >= 0 && colb + 18 <= p.W) return 0xfffu;  // wave-uniform
    uint32_t m = 0;
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      const int pp = 16 * j + l15 - OFF;
      const int rr = pp >= 0 ? (pp * 3641) >> 16 : 0, cc = pp - 18 * rr;
      m |= ((pp >= 0 && (unsigned)(colb + cc) < (unsigned)p.W) ? 1u : 0u) << j;
    }
    return m;
  };
  auto row_mask = [&](int rowb) __attribute__((always_inline)) -> uint32_t {
    if (rowb >= 0 && rowb + 10 <= p.H) return 0xfffu;  // wave-uniform
    uint32_t m = 0;
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      const int pp = 16 * j + l15 - OFF;
      const int rr = pp >= 0 ? (pp * 3641) >> 16 : 0;
      m |= ((pp >= 0 && (unsigned)(rowb + rr) < (unsigned)p.H) ? 1u : 0u) << j;
    }
    return m;
  };

  __syncthreads();  // the first chunk has landed
  int cur = 0;
  for (int u = row_id; u < p.units; u += p.rows) {
    const int seg = u % p.nseg, t1 = u / p.nseg;
    const int strip = t1 % p.nstrip, b = t1 / p.nstrip;
    const int colb = strip * 16 - 1, rowb0 = seg * p.RS - 1;
    const uint32_t cmask = col_mask(colb);
    {
      const int ro0 = S == 1 ? seg * p.RS : (seg * p.RS) >> 1, wo0 = S == 1 ? strip * 16 : strip * 8;
      ybase = p.y2 + ((size_t)(b * p.Ho + ro0) * p.Wo + wo0) * hid + cw;
      yo = yl;
    }
    const unsigned char* xs = xb0 + cur * XB;
    uint32_t vm = cmask & row_mask(rowb0);
    // the strip's first rows: blocks 0 .. 5 (stride 2: 1 .. 3; block 0 is never read)
    if (S == 1) {
#pragma unroll
      for (int j = 0; j < 6; ++j) stepE(xs, j, vm);
    } else {
#pragma unroll
      for (int j = 1; j < 4; ++j) stepE(xs, j, vm);
    }
    for (int q = 0; q < Q; ++q) {
      const bool more = q + 1 < Q;
      const uint32_t vmn = more ? (cmask & row_mask(rowb0 + 8 * (q + 1))) : 0u;
      if (S == 1) {
        // invariant: blocks 0 .. 5 are in place.  Steps 0 .. 5: stencil row k next to the expansion of block k + 6
#pragma unroll
        for (int k = 0; k < 6; ++k) stepSE(k, xs, k + 6, vm);
        __syncthreads();  // x of this chunk is consumed by every wave; the next chunk (or the next strip's first) has landed
        cur = cur + 1 == NXB ? 0 : cur + 1;
        xs = xb0 + cur * XB;
        if (more) {
          stepSE(6, xs, 3, vmn);
          {
            bf16x8_t fq[5], bq[KS], bq2[KS];
            rdS(fq, 7);
            rdE(bq, xs, 4);
            rdE(bq2, xs, 5);
            const f32x4_t e = mmE(bq);
            f32x2_t lo, hi;
            mmS(fq, lo, hi);
            const f32x4_t e2 = mmE(bq2);
            epE(4, e, vmn);
            epS(lo, hi);
            epE(5, e2, vmn);
          }
          carry();
        } else {
          stepS(6);
          stepS(7);
        }
      } else {
        // invariant: blocks 1 .. 3 are in place.  Row pair 0 reads blocks <= 7, row pair 1 blocks <= 11
#pragma unroll
        for (int j = 4; j < 9; ++j) stepE(xs, j, vm);
        stepSE(0, xs, 9, vm);
        stepE(xs, 10, vm);
        stepE(xs, 11, vm);
        __syncthreads();
        cur = cur + 1 == NXB ? 0 : cur + 1;
        xs = xb0 + cur * XB;
        if (more) {
          stepSE(1, xs, 3, vmn);
          carry();
        } else {
          stepS(1);
        }
      }
      vm = vmn;
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
        p.stats_part[((size_t)row_id * 2 + 0) * hid + ch] = t1[e];
        p.stats_part[((size_t)row_id * 2 + 1) * hid + ch] = t2[e];
      }
    }
  }
}

#ifndef XS_NXB16
#define XS_NXB16 2
#endif
#ifndef XS_NXB32
#define XS_NXB32 2
#endif
#ifndef XS_NXB64
#define XS_NXB64 2
#endif
template <int CIN> constexpr int xs_nxb() { return CIN == 16 ? XS_NXB16 : (CIN == 32 ? XS_NXB32 : XS_NXB64); }
template <int CIN> constexpr size_t xs_smem_bytes() { return (size_t)xs_nxb<CIN>() * XS_NPX * CIN * 2 + 4 * XS_NPX * XS_ATB; }
#ifndef XS_OCC_MAX
#define XS_OCC_MAX 3
#endif
template <int CIN> constexpr int xs_occ() { return (int)(163840 / xs_smem_bytes<CIN>()) >= XS_OCC_MAX ? XS_OCC_MAX : (int)(163840 / xs_smem_bytes<CIN>()); }

}  // namespace

// geometries the strip kernel covers; fills the plan.  (Everything else stays on dwx_fwd_kernel.)
bool dwxs_fwd_plan(int B, int H, int W, int Cin, int hid, int stride, int* rows, int* nstrip, int* nseg, int* RS) {
  if (cvh_tune_get(20)) return false;  // CVH_TUNE key 20: 1 = tile kernel of dwx.hip (A/B runs)
  if (!(Cin == 16 || Cin == 32 || Cin == 64) || hid <= 0 || (hid % 64) || B <= 0) return false;
  if ((H % 8) || (W % 16) || H < 8) return false;
  const int chunks = hid / 64;
  const int occ = Cin == 16 ? xs_occ<16>() : (Cin == 32 ? xs_occ<32>() : xs_occ<64>());
  *nstrip = W / 16;
  int rs = H;
  int r = 256 * occ / chunks;
  if (r < 1) r = 1;
  // shorter row segments when there are too few strips to balance the workgroups (each segment recomputes its first two rows)
  while (rs > 8 && (rs % 16) == 0 && (long long)B * *nstrip * (H / rs) < 4LL * r) rs >>= 1;
  *RS = rs;
  *nseg = H / rs;
  const long long units = (long long)B * *nstrip * *nseg;
  if (units > 0x7fffffffLL) return false;
  if (r > units) r = (int)units;
  *rows = r;
  (void)stride;
  return true;
}

int dwxs_fwd_launch(const void* x, const void* w1, const float* scale1, const float* shift1, const void* wd, void* y2, float* stats_part, int B, int H,
                    int W, int Ho, int Wo, int Cin, int hid, int stride, hipStream_t st) {
  DwxsParams p;
  p.x = reinterpret_cast<const bf16_t*>(x); p.w1 = reinterpret_cast<const bf16_t*>(w1); p.scale1 = scale1; p.shift1 = shift1;
  p.wd = reinterpret_cast<const bf16_t*>(wd); p.y2 = reinterpret_cast<bf16_t*>(y2); p.stats_part = stats_part;
  p.B = B; p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo; p.hid = hid;
  p.chunks = hid / 64;
  if (!dwxs_fwd_plan(B, H, W, Cin, hid, stride, &p.rows, &p.nstrip, &p.nseg, &p.RS)) return -2;
  p.units = B * p.nstrip * p.nseg;
  const dim3 grid(p.rows * p.chunks);
#define XS_FWD(S_, C_)                                                                                                        \
  do {                                                                                                                        \
    constexpr size_t smem = xs_smem_bytes<C_>();                                                                              \
    static DynSmemAttr attr;                                                                                                  \
    if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(dwxs_fwd_kernel<S_, C_, xs_occ<C_>(), xs_nxb<C_>()>), smem); e != hipSuccess) \
      return (int)e;                                                                                                          \
    hipLaunchKernelGGL((dwxs_fwd_kernel<S_, C_, xs_occ<C_>(), xs_nxb<C_>()>), grid, dim3(320), smem, st, p);                                \
  } while (0)
#define XS_FWD_S(C_)                \
  do {                              \
    if (stride == 1) XS_FWD(1, C_); \
    else XS_FWD(2, C_);             \
  } while (0)
  switch (Cin) {
    case 16: XS_FWD_S(16); break;
    case 32: XS_FWD_S(32); break;
    default: XS_FWD_S(64); break;
  }
#undef XS_FWD_S
#undef XS_FWD
  CVH_CHECK_LAUNCH();
  return 0;
}
