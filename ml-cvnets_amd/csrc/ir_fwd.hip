// ir_red_fwd_kernel: the 1x1 PROJECTION conv of an InvertedResidual block in the forward pass (cvnets/modules/mobilenetv2.py:208-219;
// nn.Conv2d 1x1 behind BatchNorm + SiLU, cvnets/layers/conv_layer.py:254-255) as a read-dominated HBM stream:
//
//     y3[M][N] = act(scale * y2 + shift)[M][hid] x W3[N][hid]^T          + per-workgroup column statistics (sum, sum of squares) of y3
//
// y2 is the RAW depthwise output (4x wider than y3); its BatchNorm + activation is applied on the way into LDS (a BatchNorm link,
// bnlink.hpp), never written to HBM.  The generic implicit-GEMM kernel (conv_gemm_kernel<.., FX = 1>) runs these shapes at ~3.5 TB/s of
// operand traffic with two workgroups of 4 waves per CU and one exposed memory latency per K step; here a persistent 8-wave workgroup
// prefetches the next 64-row tile of y2 into registers while the current one is multiplied (W3 resident in LDS, tiles double-buffered,
// one barrier per tile) — the same skeleton as ir_exp_bwd_kernel (ir_bwd.hip), which streams at ~5 TB/s.
// Output on the transposed problem (D^T = W3 A^T): a lane ends with 4 consecutive channels of one row (8-byte stores); its columns are the
// same in every tile, so the statistics stay in registers and are reduced in a fixed order at the end (bit-reproducible).
// Algorithmic bytes per row: (hid + N) * 2.
#include "common.hpp"
#include "cvnets_hip.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4_t;

constexpr int IF_TM = 64, IF_THREADS = 512;

struct IrRedFwdParams {
  const bf16_t* y2;     // [M][hid] raw
  const float* scale;   // [hid]
  const float* shift;   // [hid]
  const bf16_t* w;      // [N][hid]
  bf16_t* out;          // [M][N]
  float* stats_part;    // [grid][2][N] or nullptr
  int act, M, ntiles;
};

// HBK = hid / 16, NB = N / 16
template <int HBK, int NB>
__global__ __launch_bounds__(IF_THREADS) void ir_red_fwd_kernel(IrRedFwdParams p) {
  constexpr int HID = 16 * HBK, N = 16 * NB;
  constexpr int WP = HID + CVH_M16_PAD, AP = HID + CVH_M16_PAD;  // pitches (elements): HID / 8 = 0 (mod 4), + 2 chunks: conflict-free fragment reads (common.hpp)
  constexpr int TILE = IF_TM * AP;
  constexpr int GIT = IF_TM * (HID / 8) / IF_THREADS;
  constexpr int NBLK = (4 * NB + 7) / 8;          // output blocks (16 rows x 16 channels) per wave; they share their 16 rows
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  bf16_t* Ws = reinterpret_cast<bf16_t*>(smem_raw);            // [N][WP]
  bf16_t* tiles = Ws + N * WP;                                  // 2 x TILE: act(bn(y2)) of a 64-row tile
  float* cf = reinterpret_cast<float*>(tiles + 2 * TILE);       // [2][HID] scale, shift
  float* red = cf + 2 * HID;                                    // [2][N]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;

  for (int i = tid; i < N * (WP / 8); i += IF_THREADS) {
    const int r = i / (WP / 8), kc = (i - r * (WP / 8)) * 8;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (kc < HID) v = *reinterpret_cast<const uint4*>(p.w + (size_t)r * HID + kc);
    *reinterpret_cast<uint4*>(Ws + r * WP + kc) = v;
  }
  for (int i = tid; i < HID; i += IF_THREADS) {
    cf[i] = p.scale[i];
    cf[HID + i] = p.shift[i];
  }
  for (int i = tid; i < 2 * N; i += IF_THREADS) red[i] = 0.f;
  __syncthreads();

  uint4 gr[GIT];
  auto load_tile = [&](int t) __attribute__((always_inline)) {
    const int row0 = t * IF_TM;
#pragma unroll
    for (int it = 0; it < GIT; ++it) {
      const int i = tid + it * IF_THREADS;
      const int r = i / (HID / 8), ck = i - r * (HID / 8);
      gr[it] = make_uint4(0, 0, 0, 0);
      if (row0 + r < p.M) gr[it] = *reinterpret_cast<const uint4*>(p.y2 + (size_t)(row0 + r) * HID + ck * 8);
    }
  };
  // BatchNorm + activation on the way into LDS (rows past M become act(shift): they are multiplied but never stored or counted)
  auto store_tile = [&](bf16_t* at) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < GIT; ++it) {
      const int i = tid + it * IF_THREADS;
      const int r = i / (HID / 8), ck = i - r * (HID / 8);
      V8<bf16_t> raw;
      raw.d = gr[it];
      float v[8], sc[8], sh[8];
      v8_unpack(raw, v);
      *reinterpret_cast<float4*>(sc) = *reinterpret_cast<const float4*>(cf + ck * 8);
      *reinterpret_cast<float4*>(sc + 4) = *reinterpret_cast<const float4*>(cf + ck * 8 + 4);
      *reinterpret_cast<float4*>(sh) = *reinterpret_cast<const float4*>(cf + HID + ck * 8);
      *reinterpret_cast<float4*>(sh + 4) = *reinterpret_cast<const float4*>(cf + HID + ck * 8 + 4);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = v[j] * sc[j] + sh[j];
      act_fwd8(v, p.act);
      V8<bf16_t> o;
      v8_pack(v, o);
      *reinterpret_cast<uint4*>(at + r * AP + ck * 8) = o.d;
    }
  };

  const bool has_blocks = wave * NBLK < 4 * NB;
  const int j0 = wave * NBLK, mb = j0 / NB, nb0 = j0 - mb * NB;
  float s1[NBLK][4], s2[NBLK][4];
#pragma unroll
  for (int i = 0; i < NBLK; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) s1[i][e] = s2[i][e] = 0.f;

  int t = blockIdx.x, cur = 0;
  if (t < p.ntiles) {
    load_tile(t);
    store_tile(tiles);
  }
  __syncthreads();
  for (; t < p.ntiles; t += gridDim.x) {
    const int tn = t + gridDim.x;
    if (tn < p.ntiles) load_tile(tn);
    const bf16_t* at = tiles + cur * TILE;
    if (has_blocks) {
      f32x4_t acc[NBLK];
#pragma unroll
      for (int i = 0; i < NBLK; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      const bf16_t* arow = at + (16 * mb + l15) * AP + 8 * l4;
      const bf16_t* wrow = Ws + (16 * nb0 + l15) * WP + 8 * l4;
#pragma unroll
      for (int ks = 0; ks < HID / 32; ++ks) {
        const bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(arow + 32 * ks);
#pragma unroll
        for (int i = 0; i < NBLK; ++i) {
          const bf16x8_t w = *reinterpret_cast<const bf16x8_t*>(wrow + (size_t)(16 * i) * WP + 32 * ks);
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, a, acc[i], 0, 0, 0);  // D^T[n][m] += W3[n][k] a[m][k]
        }
      }
      const int m = t * IF_TM + 16 * mb + l15;
      if (m < p.M) {
#pragma unroll
        for (int i = 0; i < NBLK; ++i) {
          const uint2 pk = make_uint2(f2bf_pk(acc[i][0], acc[i][1]), f2bf_pk(acc[i][2], acc[i][3]));
          *reinterpret_cast<uint2*>(p.out + (size_t)m * N + 16 * (nb0 + i) + 4 * l4) = pk;
          const float q[4] = {bf2f((uint16_t)(pk.x & 0xffff)), bf2f((uint16_t)(pk.x >> 16)), bf2f((uint16_t)(pk.y & 0xffff)),
                              bf2f((uint16_t)(pk.y >> 16))};  // statistics of the values as stored
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            s1[i][e] += q[e];
            s2[i][e] += q[e] * q[e];
          }
        }
      }
    }
    if (tn < p.ntiles) store_tile(tiles + (cur ^ 1) * TILE);
    __syncthreads();
    cur ^= 1;
  }

  if (p.stats_part) {
    // a lane's 4 columns per block are the same for all 16 rows (l15) it shares them with: butterfly over l15, then the waves in order
#pragma unroll
    for (int i = 0; i < NBLK; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int msk = 1; msk < 16; msk <<= 1) {
          s1[i][e] += __shfl_xor(s1[i][e], msk, 64);
          s2[i][e] += __shfl_xor(s2[i][e], msk, 64);
        }
    lds_ordered_accumulate(wave, IF_THREADS / 64, has_blocks && l15 == 0, [&]() {
#pragma unroll
      for (int i = 0; i < NBLK; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          red[16 * (nb0 + i) + 4 * l4 + e] += s1[i][e];
          red[N + 16 * (nb0 + i) + 4 * l4 + e] += s2[i][e];
        }
    });
    for (int i = tid; i < 2 * N; i += IF_THREADS) p.stats_part[(size_t)blockIdx.x * 2 * N + i] = red[i];
  }
}

template <int HBK, int NB> size_t ir_red_fwd_smem() {
  constexpr int HID = 16 * HBK, N = 16 * NB;
  return ((size_t)N * (HID + CVH_M16_PAD) + 2 * (size_t)IF_TM * (HID + CVH_M16_PAD)) * 2 + (size_t)(2 * HID + 2 * N) * 4;
}

bool ir_red_shape(int hid, int N) { return (hid == 64 && N == 32) || (hid == 128 && N == 64) || (hid == 256 && (N == 64 || N == 96)); }
int ir_red_occupancy(int hid) { return hid == 64 ? 4 : (hid == 128 ? 2 : 1); }

}  // namespace

/* partial statistics rows (= workgroups) of cvh_ir_red_fwd; 0: shape not covered (use cvh_pw_gemm_bn) */
extern "C" int cvh_ir_red_fwd_rows(long long M, int hid, int N) {
  if (M < 65536 || M > 0x7fffffffLL / 4 || !ir_red_shape(hid, N)) return 0;
  const long long nt = (M + IF_TM - 1) / IF_TM;
  const int wgs = 256 * ir_red_occupancy(hid);
  return nt < wgs ? (int)nt : wgs;
}

extern "C" int cvh_ir_red_fwd(int dtype, const void* y2, const float* scale, const float* shift, int act, const void* wgt, void* out,
                              float* stats_part, long long M, int hid, int N, void* stream) {
  if (dtype != CVH_DT_BF16) return -1;
  const int rows = cvh_ir_red_fwd_rows(M, hid, N);
  if (rows <= 0) return -2;
  IrRedFwdParams p;
  p.y2 = reinterpret_cast<const bf16_t*>(y2); p.scale = scale; p.shift = shift; p.w = reinterpret_cast<const bf16_t*>(wgt);
  p.out = reinterpret_cast<bf16_t*>(out); p.stats_part = stats_part; p.act = act; p.M = (int)M; p.ntiles = (int)((M + IF_TM - 1) / IF_TM);
  cvh_family_tally(4, (long long)M * (hid + N) * 2);
  hipStream_t st = (hipStream_t)stream;
#define IF_LAUNCH(HBK_, NB_)                                                                                                             \
  do {                                                                                                                                   \
    const size_t smem = ir_red_fwd_smem<HBK_, NB_>();                                                                                    \
    static DynSmemAttr attr;                                                                                                             \
    if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(ir_red_fwd_kernel<HBK_, NB_>), smem); e != hipSuccess) return (int)e;  \
    hipLaunchKernelGGL((ir_red_fwd_kernel<HBK_, NB_>), dim3(rows), dim3(IF_THREADS), smem, st, p);                                       \
  } while (0)
  if (hid == 64) IF_LAUNCH(4, 2);
  else if (hid == 128) IF_LAUNCH(8, 4);
  else if (N == 64) IF_LAUNCH(16, 4);
  else IF_LAUNCH(16, 6);
#undef IF_LAUNCH
  CVH_CHECK_LAUNCH();
  return 0;
}
