// conv3x3_kernel: dense 3x3 convolution (stride 1, pad 1, dilation 1), NHWC bf16, as an implicit GEMM whose A operand never leaves LDS.
// The 3x3 convolutions of the MobileViT blocks — local_rep.conv_3x3 and the fusion conv over cat(res, fm)
// (cvnets/modules/mobilevit_block.py:102-148,269-288 -> ConvLayer2d / nn.Conv2d, cvnets/layers/conv_layer.py:254-255) — forward and, with the
// transposed / flipped weight pack, their input gradients.
//
// conv_gemm_kernel runs these as a register-staged im2col GEMM: every 64-wide K step gathers a 128 x 64 slice of the (virtual) im2col matrix
// through VGPRs and two barriers, so each input element is fetched nine times and the kernel sits at 250-400 TFLOP/s (0.9 TB/s of L2 -> LDS
// traffic per CU is what it is really bound by).  Here
//   * a workgroup owns an 8 x 16 pixel tile; the 10 x 18 HALO tile of one channel slab (<= 128 channels of one source tensor) is brought into
//     LDS ONCE (global_load_lds, out-of-image pixels from a zero line) and all nine taps read their shifted windows from it: A traffic 1.4x
//     instead of 9x;
//   * only the weights stream: one [N x slab] chunk per tap, direct-to-LDS, double-buffered, one barrier per tap;
//   * two-source inputs (the fusion conv's cat(res, fm)) are two slabs accumulated into the same registers — the concat stays virtual;
//   * MFMA 32x32x16 on the transposed problem (D^T[n][pixel]) so a lane holds 4 consecutive output channels: 8-byte staging writes,
//     16-byte row stores; column statistics for the BatchNorm behind the conv (sum, sumsq of the stored values) from the same epilogue;
//   * LDS pitches are an odd number of 16-byte chunks: conflict-free ds_read_b128 fragment reads.
#include "common.hpp"
#include "cvnets_hip.h"
#include "gemm_params.hpp"

namespace {
typedef __attribute__((address_space(1))) const void* c3_gptr_t;
typedef __attribute__((address_space(3))) void* c3_lptr_t;
__device__ __attribute__((aligned(128))) unsigned char c3_zero_line[128];  // zero-initialised: out-of-image pixels, pad chunks, rows past N

__device__ __forceinline__ void c3_glds16(const bf16_t* g, unsigned char* l) {
  __builtin_amdgcn_global_load_lds((c3_gptr_t)g, (c3_lptr_t)l, 16, 0, 0);
}

constexpr int TH = 8, TW = 16, HH = TH + 2, HW = TW + 2, HPIX = HH * HW;  // 8 x 16 outputs, 10 x 18 halo
constexpr int A_IT = 12;   // halo slots per lane: ceil(180 * 17 / 256)
constexpr int W_IT = 9;    // weight-chunk slots per lane: ceil(128 * 17 / 256)
}  // namespace

#define C3_MAXSLAB 8
struct Conv3Geom {
  int tiles_h, tiles_w, ntiles;
  int pc;         // chunks (16 B) per pixel / per weight row in LDS: (largest slab) / 8 + 1, odd
  int nslab;      // channel slabs: each <= the capacity that keeps two workgroups per CU, never across the two sources
  int s_src[C3_MAXSLAB];   // 0: src1, 1: src2
  int s_c0[C3_MAXSLAB];    // first channel inside its source
  int s_cs[C3_MAXSLAB];    // channels
  int rows;       // rows of the caller's partial-statistics buffer
  int a_bytes, w_bytes;  // LDS bytes of the halo image and of ONE weight buffer (64-slot multiples)
};

// NF = N / 32 accumulator fragments per wave (3: N = 96, 4: N = 128)
template <int NF>
__global__ __launch_bounds__(256, 2) void conv3x3_kernel(ConvGemmParams p, Conv3Geom g) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  unsigned char* As = smem;                                   // [HPIX][pc] x 16 B   (later: the output staging)
  unsigned char* Wb0 = smem + g.a_bytes;                      // 2 x [N][pc] x 16 B
  float* red = reinterpret_cast<float*>(Wb0 + 2 * g.w_bytes);  // [2][N] statistics, [N] bias
  float* bias_s = red + 2 * 32 * NF;
  constexpr int N = 32 * NF;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int pc = g.pc;
  const bf16_t* zero = reinterpret_cast<const bf16_t*>(c3_zero_line);
  const bf16_t* __restrict__ wgt = reinterpret_cast<const bf16_t*>(p.wgt);
  bf16_t* __restrict__ out = reinterpret_cast<bf16_t*>(p.out);
  const int Cin = p.C1 + p.C2;
  const bool want_stats = p.stats_part != nullptr;

  if (tid < 2 * N) red[tid] = 0.f;
  if (tid < N) bias_s[tid] = p.bias != nullptr ? p.bias[tid] : 0.f;

  // slot tables (tile-invariant): halo slot -> (pixel row, pixel col, chunk); weight slot -> (n, chunk).  -1 = a slot past the image
  int a_tab[A_IT], w_tab[W_IT];
#pragma unroll
  for (int j = 0; j < A_IT; ++j) {
    const int slot = 64 * (wave + 4 * j) + lane;
    const int px = slot / pc, c = slot - px * pc;
    a_tab[j] = (px < HPIX) ? ((px / HW) << 16) | ((px % HW) << 8) | c : -1;
  }
#pragma unroll
  for (int j = 0; j < W_IT; ++j) {
    const int slot = 64 * (wave + 4 * j) + lane;
    const int n = slot / pc, c = slot - n * pc;
    w_tab[j] = (n < N) ? (n << 8) | c : -1;
  }
  const int a_n = (HPIX * pc + 63) / 64, w_n = (N * pc + 63) / 64;  // wave-instructions that hold real slots

  // halo image of (tile, slab): every lane fetches its slots' 16 bytes from the pixel they belong to (zero line outside the image / in the pad chunk)
  auto issue_halo = [&](int b, int h0, int w0, const bf16_t* src, int cs, int csrc) {  // src already points at the slab's first channel
#pragma unroll
    for (int j = 0; j < A_IT; ++j) {
      const int i = wave + 4 * j;
      if (i < a_n) {
        const int e = a_tab[j];
        const int hh = h0 - 1 + (e >> 16), ww = w0 - 1 + ((e >> 8) & 0xff), c = e & 0xff;
        const bool ok = e >= 0 && hh >= 0 && hh < p.H && ww >= 0 && ww < p.W && c * 8 < cs;
        c3_glds16(ok ? src + ((size_t)(b * p.H + hh) * p.W + ww) * csrc + c * 8 : zero, As + i * 1024);
      }
    }
  };
  // weight chunk of (tap, slab): rows n of the packed weight [N][9][Cin], columns [tap * Cin + c0, + cs)
  auto issue_w = [&](int tap, int c0, int cs, unsigned char* dst) {
#pragma unroll
    for (int j = 0; j < W_IT; ++j) {
      const int i = wave + 4 * j;
      if (i < w_n) {
        const int e = w_tab[j];
        const int n = e >> 8, c = e & 0xff;
        const bool ok = e >= 0 && n < p.N && c * 8 < cs;
        c3_glds16(ok ? wgt + (size_t)n * p.Ktot + tap * Cin + c0 + c * 8 : zero, dst + i * 1024);
      }
    }
  };

  // this wave's 32 pixels: tile rows 2 * wave, 2 * wave + 1
  const int pr = 2 * wave + ((lane & 31) >> 4), pcx = lane & 15;
  float cs1[8], cs2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) cs1[j] = cs2[j] = 0.f;
  // epilogue mapping: ppr pieces (8 channels) per pixel, ppl whole pixels per pass — a lane keeps the same piece in every pass
  constexpr int ppr = N / 8, ppl = 64 / ppr;
  const int e_piece = lane % ppr, e_px = lane / ppr;
  const bool e_active = lane < ppr * ppl;

  for (int tix = blockIdx.x; tix < g.ntiles; tix += gridDim.x) {
    const int tw = tix % g.tiles_w;
    const int t1 = tix / g.tiles_w;
    const int th = t1 % g.tiles_h;
    const int b = t1 / g.tiles_h;
    const int h0 = th * TH, w0 = tw * TW;

    f32x16_t acc[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) acc[f] = acc_zero();

    for (int s = 0; s < g.nslab; ++s) {
      const bool second = g.s_src[s] != 0;
      const bf16_t* src = reinterpret_cast<const bf16_t*>(second ? p.src2 : p.src1) + g.s_c0[s];
      const int cs = g.s_cs[s], c0 = (second ? p.C1 : 0) + g.s_c0[s];  // c0: column of the slab inside a tap of the packed weight
      __syncthreads();  // previous slab / previous tile's staging fully consumed
      issue_halo(b, h0, w0, src, cs, second ? p.C2 : p.C1);
      issue_w(0, c0, cs, Wb0);
      const int ksteps = (cs + 15) / 16;
      for (int tap = 0; tap < 9; ++tap) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // halo + this tap's weights have landed for every wave; the other weight buffer is free
        unsigned char* Wc = Wb0 + (tap & 1) * g.w_bytes;
        if (tap < 8) issue_w(tap + 1, c0, cs, Wb0 + ((tap + 1) & 1) * g.w_bytes);
        const int kh = tap / 3, kw = tap - kh * 3;
        const unsigned char* arow = As + (((pr + kh) * HW + pcx + kw) * pc + (lane >> 5)) * 16;
        const unsigned char* wrow = Wc + ((lane & 31) * pc + (lane >> 5)) * 16;
        for (int kk = 0; kk < ksteps; ++kk) {
          Frag<bf16_t> a, w;
          a.v = *reinterpret_cast<const bf16x8_t*>(arow + kk * 32);
#pragma unroll
          for (int f = 0; f < NF; ++f) {
            w.v = *reinterpret_cast<const bf16x8_t*>(wrow + (size_t)(f * 32) * pc * 16 + kk * 32);
            mma32(acc[f], w, a);  // D^T[n][pixel] += W[n][k] X[pixel][k]
          }
        }
      }
    }
    __syncthreads();  // every wave is done with the halo image: it becomes the output staging [128 pixels][N + 8]
    constexpr int SP = N + 8;
    bf16_t* stg = reinterpret_cast<bf16_t*>(As) + wave * 32 * SP;
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int n = f * 32 + 8 * i + 4 * (lane >> 5);
        const float4 bv = *reinterpret_cast<const float4*>(bias_s + n);
        uint2 pk;
        pk.x = f2bf_pk(acc[f][4 * i] + bv.x, acc[f][4 * i + 1] + bv.y);
        pk.y = f2bf_pk(acc[f][4 * i + 2] + bv.z, acc[f][4 * i + 3] + bv.w);
        *reinterpret_cast<uint2*>(stg + (lane & 31) * SP + n) = pk;
      }
    wave_lds_sync();
    if (e_active) {
#pragma unroll 2
      for (int q = e_px; q < 32; q += ppl) {
        const int hh = h0 + 2 * wave + (q >> 4), ww = w0 + (q & 15);
        const int n = e_piece * 8;
        if (hh < p.H && ww < p.W && n < p.N) {
          const V8<bf16_t> pv = v8_load<bf16_t>(stg + q * SP + n);
          v8_store<bf16_t>(out + ((size_t)(b * p.H + hh) * p.W + ww) * p.N + n, pv);
          if (want_stats) {
            float vr[8];
            v8_unpack(pv, vr);
#pragma unroll
            for (int j = 0; j < 8; ++j) { cs1[j] += vr[j]; cs2[j] += vr[j] * vr[j]; }
          }
        }
      }
    }
  }

  if (want_stats) {
    lds_ordered_accumulate(wave * ppl + e_px, 4 * ppl, e_active, [&]() {  // the lanes that own one 8-channel piece, in (wave, pixel slot) order
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        red[e_piece * 8 + j] += cs1[j];
        red[N + e_piece * 8 + j] += cs2[j];
      }
    });
    for (int i = tid; i < 2 * N; i += 256) {
      const int which = i / N, n = i - which * N;
      if (n < p.N) {
        p.stats_part[((size_t)blockIdx.x * 2 + which) * p.N + n] = red[i];
        for (int r = blockIdx.x + gridDim.x; r < g.rows; r += gridDim.x) p.stats_part[((size_t)r * 2 + which) * p.N + n] = 0.f;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------------------
static bool conv3x3_plan(const ConvGemmParams& p, Conv3Geom& g, size_t& smem) {
  const int nf = p.N / 32;
  // slab capacity: the LDS footprint (halo image + two weight buffers) has to leave room for TWO workgroups per CU — one's halo reload
  // (an exposed HBM round trip per slab) runs under the other's MFMAs
  const int cap = nf == 3 ? 96 : 64;
  g.nslab = 0;
  int cs_max = 0;
  for (int src = 0; src < 2; ++src) {
    const int C = src == 0 ? p.C1 : p.C2;
    if (C == 0) continue;
    const int n = (C + cap - 1) / cap;
    const int per = ((C + n - 1) / n + 15) / 16 * 16;  // equal slabs, whole 16-wide K steps
    for (int c0 = 0; c0 < C; c0 += per) {
      if (g.nslab == C3_MAXSLAB) return false;
      g.s_src[g.nslab] = src;
      g.s_c0[g.nslab] = c0;
      g.s_cs[g.nslab] = C - c0 < per ? C - c0 : per;
      if (g.s_cs[g.nslab] > cs_max) cs_max = g.s_cs[g.nslab];
      ++g.nslab;
    }
  }
  g.tiles_h = (p.H + TH - 1) / TH;
  g.tiles_w = (p.W + TW - 1) / TW;
  g.ntiles = p.B * g.tiles_h * g.tiles_w;
  g.pc = (cs_max + 7) / 8 + 1;
  if ((g.pc & 1) == 0) g.pc += 1;  // odd chunk pitch: the 16 pixels / weight rows of a ds_read_b128 lane group hit 16 distinct bank quads
  g.a_bytes = ((HPIX * g.pc + 63) / 64) * 1024;
  const int stage_bytes = 128 * (p.N + 8) * 2;
  if (g.a_bytes < stage_bytes) g.a_bytes = (stage_bytes + 1023) / 1024 * 1024;
  g.w_bytes = ((p.N * g.pc + 63) / 64) * 1024;
  if ((HPIX * g.pc + 63) / 64 > 4 * A_IT || (p.N * g.pc + 63) / 64 > 4 * W_IT) return false;
  smem = (size_t)g.a_bytes + 2 * (size_t)g.w_bytes + (size_t)3 * 32 * nf * sizeof(float);
  return smem <= 160 * 1024;
}

bool conv3x3_eligible(const ConvGemmParams& p) {
  if (cvh_tune_get(CVH_TUNE_NO_CONV3X3)) return false;
  if (p.KH != 3 || p.KW != 3 || p.stride != 1 || p.pad != 1 || p.dil != 1 || p.Ho != p.H || p.Wo != p.W) return false;
  if (p.N != 96 && p.N != 128) return false;
  if (p.C1 < 32 || (p.C1 % 8) || (p.C2 % 8) || (p.C2 != 0 && p.src2 == nullptr)) return false;
  if (p.act != 0 || p.save_pre != nullptr || p.actgrad_aux != nullptr || p.residual != nullptr || p.drop_p > 0.f || p.sc_s != 0) return false;
  if (p.a_xf.mode != 0 || p.e_mode != 0) return false;
  if (p.M < 16384 || p.H >= 65536 || p.W >= 65536) return false;
  Conv3Geom g;
  size_t smem;
  return conv3x3_plan(p, g, smem);
}

int launch_conv3x3(const ConvGemmParams& p, int rows, hipStream_t st) {
  Conv3Geom g;
  size_t smem;
  if (!conv3x3_plan(p, g, smem)) return -2;
  g.rows = rows;
  const int nf = p.N / 32;
  int grid = g.ntiles < 512 ? g.ntiles : 512;
  if (p.stats_part != nullptr && rows > 0 && grid > rows) grid = rows;
  static DynSmemAttr attr3, attr4;
  if (nf == 3) {
    if (hipError_t e = attr3.ensure(reinterpret_cast<const void*>(conv3x3_kernel<3>), smem); e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(conv3x3_kernel<3>, dim3(grid), dim3(256), smem, st, p, g);
  } else {
    if (hipError_t e = attr4.ensure(reinterpret_cast<const void*>(conv3x3_kernel<4>), smem); e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(conv3x3_kernel<4>, dim3(grid), dim3(256), smem, st, p, g);
  }
  CVH_CHECK_LAUNCH();
  return 0;
}
