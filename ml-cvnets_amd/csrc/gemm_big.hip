// Large-tile bf16 NT GEMM for transformer-sized linear layers (ViT-B / CLIP token GEMMs: M = B*S >= 2048 rows, K and N in the
// hundreds to thousands):   out[m][n] = epilogue( sum_k A[m][k] * Wp[n][k] ),  A = activations [M][K], Wp = packed weights [N][K].
// (dX of a linear layer is the same product with the transposed weight pack, so it runs here too.)
//
//   * 128 x 128 output tile per workgroup, 4 waves as 2 x 2, each wave 64 x 64 = 2 x 2 MFMA 32x32x16 accumulators;
//   * K step 64; both operand tiles go HBM/L2 -> LDS directly (global_load_lds, 16 B per lane, no staging VGPRs), double
//     buffered: the loads of K step t+1 are in flight while step t is on the matrix cores, one barrier per K step;
//   * LDS image = 128-byte rows (64 bf16) with the 16-byte chunk index XOR-swizzled by (row >> 1) & 7: the direct-to-LDS write
//     is lane-linear, so the swizzle is applied by choosing WHICH global chunk each lane fetches (same 128-B line -> coalescing
//     unchanged), and the ds_read_b128 fragment reads of 16 consecutive rows hit 16 distinct bank groups;
//   * XCD-contiguous tile order: consecutive workgroups on one XCD walk the N tiles of the same A row panel (A panel and the
//     whole weight matrix stay in that XCD's L2);
//   * epilogue identical in semantics to conv_gemm's (bias -> save_pre -> activation -> dropout -> residual), staged through
//     LDS so every store is a full 128-byte row segment.
#include "common.hpp"
#include "gemm_params.hpp"

#ifndef CVH_TN128_GI
#define CVH_TN128_GI 1  // the same for gemm_tn128_kernel
#endif
#ifndef CVH_TN_GI
#define CVH_TN_GI 1  // gemm_tn256_kernel: 0 (tools/build_variant.py) = all eight pieces of the next step issued as a block after the barrier
#endif

namespace {
constexpr int BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * BK * 2;     // 16 KB per 128-row operand tile
constexpr int BUF_BYTES = 2 * TILE_BYTES;    // A + B of the 128 x 128 kernels
constexpr int STG_PITCH = 64 + 8;            // bf16 elements per staged output row (64 columns + 16 B pad)

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
__device__ __attribute__((aligned(128))) unsigned char g_zero_line[128];  // zero-initialised: source of out-of-range rows / chunks
#define ONE8 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80
__device__ __attribute__((aligned(128))) unsigned short g_ones_line[64] = {ONE8, ONE8, ONE8, ONE8, ONE8, ONE8, ONE8, ONE8};  // bf16 1.0
#undef ONE8

__device__ __forceinline__ void glds16(const bf16_t* g, unsigned char* l) {
  __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 16, 0, 0);
}
__device__ __forceinline__ Frag<bf16_t> frag_swz(const unsigned char* tile, int row, int chunk) {
  const unsigned char* p = tile + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
  Frag<bf16_t> f;
  f.v = *reinterpret_cast<const bf16x8_t*>(p);
  return f;
}
}  // namespace

// WM = wave rows: tile = (64*WM) x 128, 2*WM waves.  WM = 2: 128x128, 64 KB LDS, 2 workgroups per CU.  WM = 4: 256x128, 96 KB LDS, one
// 8-wave workgroup per CU — same waves per CU, but 1.33x the MFMA work per byte brought into LDS (the kernel is bound by the bytes
// it can keep in flight towards LDS: ~1.5 us of L2/HBM latency x 2 tile buffers).
// RAGGED: N need not be a multiple of 128 nor K of 64 (only of 8): 16-byte chunks past the end of a row of A / of the weight, and weight
// rows past N, are fetched from a zero line; output columns past N are not stored.  Opens the direct-to-LDS pipeline (no staging VGPRs,
// no VALU between HBM and the MFMA operands) to the d = 144 ... 720 transformer linears of MobileViT, which the register-staged
// conv_gemm runs at 1.5-2.4 TB/s (every 64-wide K step exposes a memory latency behind two barriers).
template <int WM, bool RAGGED>
__global__ __launch_bounds__(128 * WM, WM == 2 ? 2 : 1) void gemm_nt128_kernel(ConvGemmParams p) {
  constexpr int BM = 64 * WM;
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, BUF_BYTES = A_BYTES + B_BYTES;
  constexpr int NB = 8 / WM;  // B-tile instructions per wave
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];  // ONE LDS object: 2 x (A tile | B tile); reused by the epilogue
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int K = p.Ktot, N = p.N, M = p.M;
  const int NT = (N + BN - 1) / BN;
  const int total = p.m_tiles * NT;
  const int L = xcd_chunk_id(blockIdx.x, total);
  const int mt = L / NT, nt = L - mt * NT;
  const int m0 = mt * BM, n0 = nt * BN;
  const bf16_t* __restrict__ A = reinterpret_cast<const bf16_t*>(p.src1);
  const bf16_t* __restrict__ Wp = reinterpret_cast<const bf16_t*>(p.wgt);

  // this lane's 4 + 4 chunks per K step: wave-instruction i = wave*4 + j covers tile rows [8i, 8i+8), lane -> (row, LDS slot)
  const bf16_t* ga[4];
  const bf16_t* gb[4];
  int ka[4], kb[4];      // RAGGED: column (within a K step) of this lane's chunk; weight rows past N start beyond every K
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = (wave * 4 + j) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);  // global chunk that belongs in LDS slot (lane & 7) of this row
    int am = m0 + row;
    if (am > M - 1) am = M - 1;  // rows past M: read a valid row, the result is never stored
    ga[j] = A + (size_t)am * K + c * 8;
    ka[j] = c * 8;
  }
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int row = (wave * NB + j) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    const bool row_ok = !RAGGED || n0 + row < N;
    gb[j] = Wp + (size_t)(row_ok ? n0 + row : 0) * K + c * 8;
    kb[j] = row_ok ? c * 8 : (1 << 30);
  }
  const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_line);
  auto issue = [&](int kt, int buf) {
    unsigned char* a_dst = smem + buf * BUF_BYTES + (wave * 4) * 1024;
    unsigned char* b_dst = smem + buf * BUF_BYTES + A_BYTES + (wave * NB) * 1024;
#pragma unroll
    for (int j = 0; j < 4; ++j) glds16((!RAGGED || kt * BK + ka[j] < K) ? ga[j] + kt * BK : zero, a_dst + j * 1024);
#pragma unroll
    for (int j = 0; j < NB; ++j) glds16((!RAGGED || kt * BK + kb[j] < K) ? gb[j] + kt * BK : zero, b_dst + j * 1024);
  };

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = acc_zero();

  const int KT = (K + BK - 1) / BK;
  issue(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kt = 0; kt < KT; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < KT) issue(kt + 1, buf ^ 1);
    const unsigned char* At = smem + buf * BUF_BYTES;
    const unsigned char* Bt = At + A_BYTES;
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      const int chunk = 2 * kk + (lane >> 5);
      Frag<bf16_t> a0 = frag_swz(At, wr * 64 + (lane & 31), chunk);
      Frag<bf16_t> a1 = frag_swz(At, wr * 64 + 32 + (lane & 31), chunk);
      Frag<bf16_t> b0 = frag_swz(Bt, wc * 64 + (lane & 31), chunk);
      Frag<bf16_t> b1 = frag_swz(Bt, wc * 64 + 32 + (lane & 31), chunk);
      mma32(acc[0][0], a0, b0);
      mma32(acc[0][1], a0, b1);
      mma32(acc[1][0], a1, b0);
      mma32(acc[1][1], a1, b1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // (a) everyone is done reading `buf`, (b) tile kt+1 has landed in buf^1 for all waves
  }

  // ---- epilogue: accumulators (+bias) -> per-wave LDS staging -> coalesced rows with the fused ops ----
  bf16_t* stg = reinterpret_cast<bf16_t*>(smem) + wave * (64 * STG_PITCH);
  DropKey dkey = {0u, 0u};
  if (p.drop_p > 0.f) dkey = drop_key(*p.seed, p.stream_id, p.drop_p);
  const float inv_keep = p.drop_p > 0.f ? 1.0f / (1.0f - p.drop_p) : 1.0f;
  bf16_t* __restrict__ out = reinterpret_cast<bf16_t*>(p.out);
#pragma unroll
  for (int tj = 0; tj < 2; ++tj) {
    const int n = n0 + wc * 64 + tj * 32 + (lane & 31);
    const float bias = (p.bias != nullptr && (!RAGGED || n < N)) ? p.bias[n] : 0.f;
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
      for (int r = 0; r < 16; ++r) stg[(ti * 32 + acc_row(r, lane)) * STG_PITCH + tj * 32 + (lane & 31)] = from_f<bf16_t>(acc[ti][tj][r] + bias);
  }
  wave_lds_sync();
  const int ch = lane & 7;
  const int ncol = n0 + wc * 64 + ch * 8;
  // the epilogue's own HBM operands for all 8 passes are requested first (the accumulators are dead: registers are free), so their
  // latency is paid once per tile, not once per pass
  const bf16_t* agp = reinterpret_cast<const bf16_t*>(p.actgrad_aux);
  const bf16_t* rsp = reinterpret_cast<const bf16_t*>(p.residual);
  V8<bf16_t> agr[8], rsr[8];
#pragma unroll
  for (int pass = 0; pass < 8; ++pass) {
    const int m = m0 + wr * 64 + pass * 8 + (lane >> 3);
    const bool ok = m < M && (!RAGGED || ncol < N);
    const size_t o = (size_t)m * N + ncol;
    agr[pass] = v8_load_clamped<bf16_t>(agp ? agp : out, o, ok && agp != nullptr);
    rsr[pass] = v8_load_clamped<bf16_t>(rsp ? rsp : out, o, ok && rsp != nullptr);
  }
#pragma unroll
  for (int pass = 0; pass < 8; ++pass) {
    const int row = pass * 8 + (lane >> 3);
    const int m = m0 + wr * 64 + row;
    if (m < M && (!RAGGED || ncol < N)) {
      const size_t o = (size_t)m * N + ncol;
      V8<bf16_t> pv = v8_load<bf16_t>(stg + row * STG_PITCH + ch * 8);
      float v[8];
      v8_unpack(pv, v);
      if (p.act == CVH_ACT_GELU_D) {  // GELU, and GELU'(pre) instead of pre for the backward GEMM's epilogue (cvnets_hip.h)
        float d[8];
        gelu_fwd_deriv8(v, d);
        if (p.save_pre) {
          V8<bf16_t> dv;
          v8_pack(d, dv);
          v8_store<bf16_t>(reinterpret_cast<bf16_t*>(p.save_pre) + o, dv);
        }
      } else {
        if (p.save_pre) v8_store<bf16_t>(reinterpret_cast<bf16_t*>(p.save_pre) + o, pv);
        if (p.act != CVH_ACT_NONE) act_fwd8(v, p.act);
      }
      if (p.actgrad_aux) {  // backward through the producer's activation: v *= act'(pre-activation of the layer below)
        float a[8];
        v8_unpack(agr[pass], a);
        act_grad8_mul(v, a, p.actgrad_act);
      }
      if (p.drop_p > 0.f) dropout_scale8(dkey, o, inv_keep, v);
      if (p.residual) {
        float rr[8];
        v8_unpack(rsr[pass], rr);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += rr[j];
      }
      V8<bf16_t> ov;
      v8_pack(v, ov);
      v8_store<bf16_t>(out + o, ov);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// 256 x 256 tile, 8 waves — the MFMA-bound transformer linears (ViT-B / CLIP: N, K = 512 ... 3072).
// What bounds these GEMMs on this chip is the L2 -> LDS path of a CU, not the matrix pipe: measured with the direct-to-LDS loads of the
// kernels in this file it sustains ~10 B/clk/CU with 64-byte row segments and roughly twice that with 128-byte ones, against the 62 B/clk/CU
// the 128 x 128 kernel above would need at the matrix pipe's full rate (two workgroups x 32 KB per 128 MFMAs) — it tops out at 600-800
// TFLOP/s.  Here a workgroup owns 256 x 256 outputs (each wave 64 x 128 = 2 x 4 MFMA 32x32x16 accumulators, 128 registers): 64 KB per
// 256 MFMAs of a CU, half the bytes per flop, in the same 128-byte rows / XOR swizzle as above; two 64 KB stage buffers (a K step = 64
// is 2048 matrix-pipe cycles per SIMD: longer than the load latency, so one step of loads in flight is enough).  One barrier per K step;
// XCD-contiguous tile order with the N tile fastest.  (A first version with 32-wide K steps in four buffers — three steps of loads in
// flight — measured the same 600 TFLOP/s as the 128 x 128 kernel: its 64-byte row segments halve what the path delivers.)
// ---------------------------------------------------------------------------------------------------------------------------------
namespace {
constexpr int B2_TILE = 256 * BK * 2;      // 32 KB per operand tile
constexpr int B2_STAGE = 2 * B2_TILE;      // A | B
}  // namespace

__global__ __launch_bounds__(512, 2) void gemm_nt256_kernel(ConvGemmParams p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];  // 2 x (A tile | B tile); reused by the epilogue
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;  // 4 x 2 waves: rows 64 wr .., columns 128 wc ..
  const int K = p.Ktot, N = p.N, M = p.M;
  const int NT = N / 256;
  const int total = p.m_tiles * NT;
  const int L = xcd_chunk_id(blockIdx.x, total);
  const int mt = L / NT, nt = L - mt * NT;
  const int m0 = mt * 256, n0 = nt * 256;
  const bf16_t* __restrict__ A = reinterpret_cast<const bf16_t*>(p.src1);
  const bf16_t* __restrict__ Wp = reinterpret_cast<const bf16_t*>(p.wgt);

  // direct-to-LDS assignment: wave-instruction i (32 per operand and stage) covers tile rows [8 i, 8 i + 8); lane -> (row, LDS slot);
  // wave w issues instructions 4w .. 4w + 3 of A and of B
  const bf16_t* ga[4];
  const bf16_t* gb[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = (4 * wave + j) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);  // global chunk that belongs in LDS slot (lane & 7) of this row
    int am = m0 + row;
    if (am > M - 1) am = M - 1;  // rows past M: read a valid row, the result is never stored
    ga[j] = A + (size_t)am * K + c * 8;
    gb[j] = Wp + (size_t)(n0 + row) * K + c * 8;
  }
  auto issue = [&](int kt, int buf) __attribute__((always_inline)) {
    unsigned char* dst = smem + buf * B2_STAGE + (4 * wave) * 1024;
#pragma unroll
    for (int j = 0; j < 4; ++j) glds16(ga[j] + kt * BK, dst + j * 1024);
#pragma unroll
    for (int j = 0; j < 4; ++j) glds16(gb[j] + kt * BK, dst + B2_TILE + j * 1024);
  };

  f32x16_t acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = acc_zero();

  const int KT = K / BK;
  issue(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kt = 0; kt < KT; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < KT) issue(kt + 1, buf ^ 1);
    const unsigned char* At = smem + buf * B2_STAGE;
    const unsigned char* Bt = At + B2_TILE;
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      const int chunk = 2 * kk + (lane >> 5);
      Frag<bf16_t> a[2], b[4];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = frag_swz(At, wr * 64 + 32 * i + (lane & 31), chunk);
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = frag_swz(Bt, wc * 128 + 32 * j + (lane & 31), chunk);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) mma32(acc[i][j], a[i], b[j]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // (a) everyone is done reading `buf`, (b) tile kt+1 has landed in buf^1 for all waves
  }

  // ---- epilogue (semantics of conv_gemm's: bias -> save_pre -> activation -> act-grad -> dropout -> residual), per wave, two halves of
  //      64 rows x 64 columns through a per-wave LDS staging area so that every store is a full 128-byte row segment ----
  bf16_t* stg = reinterpret_cast<bf16_t*>(smem) + wave * (64 * STG_PITCH);
  DropKey dkey = {0u, 0u};
  if (p.drop_p > 0.f) dkey = drop_key(*p.seed, p.stream_id, p.drop_p);
  const float inv_keep = p.drop_p > 0.f ? 1.0f / (1.0f - p.drop_p) : 1.0f;
  bf16_t* __restrict__ out = reinterpret_cast<bf16_t*>(p.out);
  const bf16_t* agp = reinterpret_cast<const bf16_t*>(p.actgrad_aux);
  const bf16_t* rsp = reinterpret_cast<const bf16_t*>(p.residual);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int nw = n0 + wc * 128 + h * 64;  // first column of this half
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) {
      const int n = nw + tj * 32 + (lane & 31);
      const float bias = p.bias != nullptr ? p.bias[n] : 0.f;
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          stg[(ti * 32 + acc_row(r, lane)) * STG_PITCH + tj * 32 + (lane & 31)] = from_f<bf16_t>(acc[ti][2 * h + tj][r] + bias);
    }
    wave_lds_sync();
    const int ch = lane & 7;
    const int ncol = nw + ch * 8;
    V8<bf16_t> agr[8], rsr[8];
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
      const int m = m0 + wr * 64 + pass * 8 + (lane >> 3);
      const bool ok = m < M;
      const size_t o = (size_t)m * N + ncol;
      agr[pass] = v8_load_clamped<bf16_t>(agp ? agp : out, o, ok && agp != nullptr);
      rsr[pass] = v8_load_clamped<bf16_t>(rsp ? rsp : out, o, ok && rsp != nullptr);
    }
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
      const int row = pass * 8 + (lane >> 3);
      const int m = m0 + wr * 64 + row;
      if (m < M) {
        const size_t o = (size_t)m * N + ncol;
        V8<bf16_t> pv = v8_load<bf16_t>(stg + row * STG_PITCH + ch * 8);
        float v[8];
        v8_unpack(pv, v);
        if (p.act == CVH_ACT_GELU_D) {  // GELU, and GELU'(pre) instead of pre for the backward GEMM's epilogue (cvnets_hip.h)
          float d[8];
          gelu_fwd_deriv8(v, d);
          if (p.save_pre) {
            V8<bf16_t> dv;
            v8_pack(d, dv);
            v8_store<bf16_t>(reinterpret_cast<bf16_t*>(p.save_pre) + o, dv);
          }
        } else {
          if (p.save_pre) v8_store<bf16_t>(reinterpret_cast<bf16_t*>(p.save_pre) + o, pv);
          if (p.act != CVH_ACT_NONE) act_fwd8(v, p.act);
        }
        if (p.actgrad_aux) {
          float a[8];
          v8_unpack(agr[pass], a);
          act_grad8_mul(v, a, p.actgrad_act);
        }
        if (p.drop_p > 0.f) dropout_scale8(dkey, o, inv_keep, v);
        if (p.residual) {
          float rr[8];
          v8_unpack(rsr[pass], rr);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] += rr[j];
        }
        V8<bf16_t> ov;
        v8_pack(v, ov);
        v8_store<bf16_t>(out + o, ov);
      }
    }
    wave_lds_sync();  // the second half reuses the staging area
  }
}

bool gemm_big_eligible(const ConvGemmParams& p) {
  if (cvh_tune_get(CVH_TUNE_BIG_GEMM) == 0) return false;
  const bool linear = p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0 && p.C2 == 0 && p.src2 == nullptr;
  if (!linear || p.M < 2048 || p.stats_part != nullptr || p.sc_s != 0) return false;
  if (p.N >= 256 && (p.N % BN) == 0 && p.Ktot >= 256 && (p.Ktot % BK) == 0) return true;  // ViT-B / CLIP sized
  // ragged tiles: measured against conv_gemm on the MobileViT linears (M = 65 k ... 1 M rows): ahead for wide outputs
  // (144 -> 432: 780 -> 620 us, 240 -> 720: 125 -> 58 us).  Everything with 64 < K <= 320 has gone to gemm_stream_kernel since; what reaches
  // this test is K > 320 (fc2 forward, fc1 / qkv dX of layer_4 / layer_5 at 65 k - 262 k rows): N >= 192 here is -0.65 ms per step against
  // conv_gemm's exact 96 / 160-column tiles (same box: 76.6 -> 75.9), N = 144 level
  const int min_n = cvh_tune_get(CVH_TUNE_BIG_MIN_N) > 0 ? cvh_tune_get(CVH_TUNE_BIG_MIN_N) : 192;
  return p.N >= min_n && (p.N % 8) == 0 && p.Ktot > 64 && (p.Ktot % 8) == 0;
}

int launch_gemm_big(const ConvGemmParams& p0, hipStream_t st) {
  ConvGemmParams p = p0;
  const bool ragged = (p.N % BN) != 0 || (p.Ktot % BK) != 0;
  const int n_tiles = (p.N + BN - 1) / BN;
  if (!ragged && cvh_tune_get(CVH_TUNE_BIG_GEMM) == 1 && (p.N % 256) == 0 && (p.Ktot % BK) == 0 && (long long)((p.M + 255) / 256) * (p.N / 256) >= 1024) {
    // 256 x 256 tiles, one 8-wave workgroup per CU, at least four rounds of tiles over the 256 CUs (with fewer the tail round costs more than
    // the larger tile saves: CLIP's text tower at 19.7 k rows stays on the 128 x 128 kernel); knob 3: always the 128 x 128 kernel
    p.m_tiles = (p.M + 255) / 256;
    constexpr int smem = 2 * B2_STAGE;  // 128 KB
    static DynSmemAttr attr;
    if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(gemm_nt256_kernel), smem); e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(gemm_nt256_kernel, dim3(p.m_tiles * (p.N / 256)), dim3(512), smem, st, p);
  } else if (!ragged && cvh_tune_get(CVH_TUNE_BIG_GEMM) == 2 && p.M >= 8192) {  // 256 x 128 tiles, one 8-wave workgroup per CU
    p.m_tiles = (p.M + 255) / 256;
    constexpr int smem = 2 * (256 + 128) * BK * 2;  // 96 KB
    static DynSmemAttr attr;
    if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(gemm_nt128_kernel<4, false>), smem); e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((gemm_nt128_kernel<4, false>), dim3(p.m_tiles * n_tiles), dim3(512), smem, st, p);
  } else {
    p.m_tiles = (p.M + 127) / 128;
    constexpr int smem = 2 * BUF_BYTES;  // 64 KB: two workgroups per CU
    if (ragged) hipLaunchKernelGGL((gemm_nt128_kernel<2, true>), dim3(p.m_tiles * n_tiles), dim3(256), smem, st, p);
    else hipLaunchKernelGGL((gemm_nt128_kernel<2, false>), dim3(p.m_tiles * n_tiles), dim3(256), smem, st, p);
  }
  CVH_CHECK_LAUNCH();
  return 0;
}

// =============================================================================================
// dW of a transformer-sized linear layer:  part[split][n][k] = sum_{m in split} dY[m][n] * X[m][k]       (bf16, fp32 partials)
// Both operands are M-major in HBM while the MFMA wants 8 consecutive reduction (m) elements per lane.  Instead of transposing on
// the way into LDS (gemm_tn_kernel's register path) the tiles go HBM/L2 -> LDS untouched with global_load_lds and the fragments are
// gathered by the gfx950 LDS transpose read (ds_read_b64_tr_b16: a 16-lane group reads a [4 m][16 col] block and every lane
// receives 4 consecutive m of ONE column; two reads = one 32x32x16 operand).
//   * tile 128 (n) x 128 (k) per workgroup, 4 waves as 2 x 2, each wave 64 x 64 = 2 x 2 accumulators; reduction step 64 rows of m;
//   * LDS image per operand and step: 16 sub-tiles [16 m][32 col] (64-byte rows, 1 KB = one wave-wide direct-to-LDS write with
//     lane -> (row lane>>2, 16-byte chunk lane&3)); the 4 rows x 32 bytes a 16-lane group reads are 64 bytes apart -> 32 distinct
//     banks, the two groups of a 32-lane half take the other 32: conflict-free;
//   * rows past the end of the split read a zero line, so any M works; splits over M fill the chip, gemm_dw_reduce_kernel sums them.
// =============================================================================================
namespace {
typedef short tr_v4s __attribute__((ext_vector_type(4)));
typedef short tr_v8s __attribute__((ext_vector_type(8)));

// operand fragment for columns [col0, col0+32) and reduction rows [16*kk, 16*kk+16) of a [64 m][128 col] sub-tiled LDS image
__device__ __forceinline__ Frag<bf16_t> frag_tr(const unsigned char* img, int kk, int col0, int lane) {
  const int i = lane & 15;
  const int col = col0 + 16 * ((lane >> 4) & 1) + 4 * (i & 3);
  const int r = 8 * (lane >> 5) + (i >> 2);
  const unsigned char* p = img + (kk * 4 + (col >> 5)) * 1024 + r * 64 + (col & 31) * 2;
  Frag<bf16_t> f;
  f.v = tr_frag_raw<4 * 64>(lds_addr32(p));  // inline-asm form (common.hpp): no compiler-placed vmcnt(0) in front of it; the caller runs tr_wait()
  return f;
}
}  // namespace

// RAGGED: N and K need not be multiples of 128 (only of 8): 16-byte column chunks past the end of a row read the zero line instead, and the
// output columns they feed are not stored.  That opens the direct-to-LDS path — no VALU work between HBM and the MFMA operands — to the
// transformer / fusion linears of MobileViT (N, K in 96 ... 720), where gemm_tn_kernel's unpack / mask / pack work per loaded element
// (~115 VALU instructions per 8 MFMAs) serialises with the MFMAs and caps it at ~290 TFLOP/s-equivalent.
template <bool RAGGED>
__global__ __launch_bounds__(256, 2) void gemm_tn128_kernel(GemmTNParams p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];  // 2 x (dY image 16 KB | X image 16 KB)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_n = wave >> 1, wave_k = wave & 1;
  // XCD-contiguous work order, output tile fastest: the workgroups reducing the SAME rows (one per output tile) share one L2
  const int bid_ = xcd_chunk_id((int)(blockIdx.y * gridDim.x + blockIdx.x), (int)(gridDim.x * gridDim.y));
  const int bx = bid_ % (int)gridDim.x, by = bid_ / (int)gridDim.x;
  const int n0 = (bx / p.k_tiles) * 128, k0 = (bx % p.k_tiles) * 128;
  const int m_begin = by * p.m_per_split;
  const int m_end = min(p.M, m_begin + p.m_per_split);
  const bf16_t* __restrict__ dy = reinterpret_cast<const bf16_t*>(p.dy);
  const bf16_t* __restrict__ x = reinterpret_cast<const bf16_t*>(p.src1);
  const int N = p.N, K = p.Ktot;

  // direct-to-LDS assignment: wave w owns tile rows [16w, 16w+16) of every step; instruction j covers columns [32j, 32j+32)
  const int row_off = 16 * wave + (lane >> 2);
  const bf16_t* gy = dy + (size_t)(m_begin + row_off) * N + n0 + (lane & 3) * 8;
  const bf16_t* gx = x + (size_t)(m_begin + row_off) * K + k0 + (lane & 3) * 8;
  const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_line);
  // Bias gradient for free: with K % 128 != 0 the last k tile carries zero-padded columns anyway; feeding the FIRST padded 8-column chunk
  // (columns K .. K+7) from a line of ones makes output column K the column sums of dY over this split — sum_m dY[m][n] * 1 — which is the
  // bias gradient of the same layer (otherwise a separate pass over dY, cvh_colsum).  bf16 1.0 x dY is exact, accumulation is the MFMA's fp32.
  const bool fold_bias = RAGGED && p.bias_part != nullptr;
  const bf16_t* ones = reinterpret_cast<const bf16_t*>(g_ones_line);
  auto issue_y = [&](int step, int buf, int j) __attribute__((always_inline)) {  // piece j (of 4) of this wave's dY rows of a step
    const bool ok = m_begin + step * 64 + row_off < m_end;
    unsigned char* y_dst = smem + buf * BUF_BYTES + (wave * 4) * 1024;
    const bool oky = ok && (!RAGGED || n0 + j * 32 + (lane & 3) * 8 < N);
    glds16(oky ? gy + (size_t)step * 64 * N + j * 32 : zero, y_dst + j * 1024);
  };
  auto issue_x = [&](int step, int buf, int j) __attribute__((always_inline)) {
    const bool ok = m_begin + step * 64 + row_off < m_end;
    unsigned char* x_dst = smem + buf * BUF_BYTES + (wave * 4) * 1024 + TILE_BYTES;
    const int kcol = k0 + j * 32 + (lane & 3) * 8;
    const bool okx = ok && (!RAGGED || kcol < K);
    glds16(okx ? gx + (size_t)step * 64 * K + j * 32 : ((fold_bias && ok && kcol == K) ? ones : zero), x_dst + j * 1024);
  };
  auto issue = [&](int step, int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      issue_y(step, buf, j);
      issue_x(step, buf, j);
    }
  };

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = acc_zero();

  const int steps = (m_end - m_begin + 63) / 64;
  if (steps > 0) {
    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  for (int st = 0; st < steps; ++st) {
    const int buf = st & 1;
#if !CVH_TN128_GI
    if (st + 1 < steps) issue(st + 1, buf ^ 1);
#endif
    const int stn = st + 1 < steps ? st + 1 : st;  // CVH_TN128_GI: the last step re-requests itself into the free buffer
    const unsigned char* Yt = smem + buf * BUF_BYTES;
    const unsigned char* Xt = Yt + TILE_BYTES;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      Frag<bf16_t> a0 = frag_tr(Yt, kk, wave_n * 64, lane);
      Frag<bf16_t> a1 = frag_tr(Yt, kk, wave_n * 64 + 32, lane);
      Frag<bf16_t> b0 = frag_tr(Xt, kk, wave_k * 64, lane);
      Frag<bf16_t> b1 = frag_tr(Xt, kk, wave_k * 64 + 32, lane);
      tr_wait(a0.v, a1.v, b0.v, b1.v);
#if CVH_TN128_GI
      // the next step's eight direct-to-LDS pieces go out one at a time behind the MFMAs of the first two K slices (see gemm_tn256_kernel)
      mma32(acc[0][0], a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      if (kk == 0) issue_y(stn, buf ^ 1, 0);
      if (kk == 1) issue_x(stn, buf ^ 1, 0);
      __builtin_amdgcn_sched_barrier(0);
      mma32(acc[0][1], a0, b1);
      __builtin_amdgcn_sched_barrier(0);
      if (kk == 0) issue_y(stn, buf ^ 1, 1);
      if (kk == 1) issue_x(stn, buf ^ 1, 1);
      __builtin_amdgcn_sched_barrier(0);
      mma32(acc[1][0], a1, b0);
      __builtin_amdgcn_sched_barrier(0);
      if (kk == 0) issue_y(stn, buf ^ 1, 2);
      if (kk == 1) issue_x(stn, buf ^ 1, 2);
      __builtin_amdgcn_sched_barrier(0);
      mma32(acc[1][1], a1, b1);
      __builtin_amdgcn_sched_barrier(0);
      if (kk == 0) issue_y(stn, buf ^ 1, 3);
      if (kk == 1) issue_x(stn, buf ^ 1, 3);
      __builtin_amdgcn_sched_barrier(0);
#else
      mma32(acc[0][0], a0, b0);
      mma32(acc[0][1], a0, b1);
      mma32(acc[1][0], a1, b0);
      mma32(acc[1][1], a1, b1);
#endif
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  float* dst0 = p.part + (size_t)by * N * K;
#pragma unroll
  for (int fn = 0; fn < 2; ++fn)
#pragma unroll
    for (int fk = 0; fk < 2; ++fk) {
      const int k = k0 + wave_k * 64 + fk * 32 + (lane & 31);
      if (RAGGED && k >= K) {
        if (fold_bias && k == K) {  // the ones column: bias_part[split][n]
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int n = n0 + wave_n * 64 + fn * 32 + acc_row(r, lane);
            if (n < N) p.bias_part[(size_t)by * N + n] = acc[fn][fk][r];
          }
        }
        continue;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wave_n * 64 + fn * 32 + acc_row(r, lane);
        if (!RAGGED || n < N) dst0[(size_t)n * K + k] = acc[fn][fk][r];
      }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// gemm_tn256_kernel: the same direct-to-LDS dW product on a 256 x 256 output tile (round 5).
// Why: cut into 128 x 128 tiles every workgroup re-streams its 128 dY columns and 128 X columns of every row, so the bytes moved from L2
// into LDS are (N tiles x K tiles x 256) / (N + K) times the unique ones — 9.6x for ViT-B's 3072 x 768; a 256 x 256 tile halves that
// factor (and the number of LDS fragment reads per MFMA).  Used for the ViT-B / CLIP sized products (N, K >= 512: gemm_tn256_shape).  8 waves as 4 (N) x 2 (K), each 64 x 128
// = 2 x 4 accumulator tiles (128 registers); a 64-row step is 32 KB of dY + 32 KB of X, two steps in LDS (128 KB): one workgroup per CU.
// ---------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ Frag<bf16_t> frag_tr256(const unsigned char* img, int kk, int col0, int lane) {
  const int i = lane & 15;
  const int col = col0 + 16 * ((lane >> 4) & 1) + 4 * (i & 3);
  const int r = 8 * (lane >> 5) + (i >> 2);
  const unsigned char* p = img + (kk * 8 + (col >> 5)) * 1024 + r * 64 + (col & 31) * 2;
  Frag<bf16_t> f;
  f.v = tr_frag_raw<4 * 64>(lds_addr32(p));  // the caller runs tr_wait() before the first MFMA
  return f;
}

constexpr int TN256_IMG = 64 * 256 * 2;      // one operand image of a 64-row step: 32 KB
constexpr int TN256_BUF = 2 * TN256_IMG;     // dY image | X image

template <bool RAGGED>
__global__ __launch_bounds__(512, 1) void gemm_tn256_kernel(GemmTNParams p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];  // 2 x (dY image 32 KB | X image 32 KB)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_n = wave >> 1, wave_k = wave & 1;
  const int bid_ = xcd_chunk_id((int)(blockIdx.y * gridDim.x + blockIdx.x), (int)(gridDim.x * gridDim.y));
  const int bx = bid_ % (int)gridDim.x, by = bid_ / (int)gridDim.x;
  const int n0 = (bx / p.k_tiles) * 256, k0 = (bx % p.k_tiles) * 256;
  const int m_begin = by * p.m_per_split;
  const int m_end = min(p.M, m_begin + p.m_per_split);
  const bf16_t* __restrict__ dy = reinterpret_cast<const bf16_t*>(p.dy);
  const bf16_t* __restrict__ x = reinterpret_cast<const bf16_t*>(p.src1);
  const int N = p.N, K = p.Ktot;

  // direct-to-LDS assignment: wave w fills the 16-row group kk = w & 3 of every step, column half w >> 2 (128 columns = four 1 KB blocks
  // of 16 rows x 32 columns per operand); a lane carries 16 bytes: row lane >> 2 of the group, 8-column chunk lane & 3 of the block
  const int ld_kk = wave & 3, ld_half = wave >> 2;
  const int row_off = 16 * ld_kk + (lane >> 2);
  const int col_off = 128 * ld_half + (lane & 3) * 8;
  const bf16_t* gy = dy + (size_t)(m_begin + row_off) * N + n0 + col_off;
  const bf16_t* gx = x + (size_t)(m_begin + row_off) * K + k0 + col_off;
  const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_line);
  const bool fold_bias = RAGGED && p.bias_part != nullptr && (K % 256) != 0;  // column K of the padded last k tile = a column of ones (see gemm_tn128_kernel)
  const bf16_t* ones = reinterpret_cast<const bf16_t*>(g_ones_line);
  auto issue_y = [&](int step, int buf, int j) __attribute__((always_inline)) {  // piece j (of 4) of this wave's dY rows of a step
    const bool ok = m_begin + step * 64 + row_off < m_end;
    unsigned char* y_dst = smem + buf * TN256_BUF + (ld_kk * 8 + 4 * ld_half) * 1024;
    const bool oky = ok && (!RAGGED || n0 + col_off + j * 32 < N);
    glds16(oky ? gy + (size_t)step * 64 * N + j * 32 : zero, y_dst + j * 1024);
  };
  auto issue_x = [&](int step, int buf, int j) __attribute__((always_inline)) {
    const bool ok = m_begin + step * 64 + row_off < m_end;
    unsigned char* x_dst = smem + buf * TN256_BUF + (ld_kk * 8 + 4 * ld_half) * 1024 + TN256_IMG;
    const int kcol = k0 + col_off + j * 32;
    const bool okx = ok && (!RAGGED || kcol < K);
    glds16(okx ? gx + (size_t)step * 64 * K + j * 32 : ((fold_bias && ok && kcol == K) ? ones : zero), x_dst + j * 1024);
  };
  auto issue = [&](int step, int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      issue_y(step, buf, j);
      issue_x(step, buf, j);
    }
  };

  f32x16_t acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = acc_zero();
  // Bias gradient when K is a multiple of 256 (no padded column to carry ones: every ViT-B / CLIP linear): the k-tile-0 workgroup of each
  // (n tile, split) sums the columns of the dY image it has in LDS anyway — thread (row group tid >> 5, 8-column chunk tid & 31) reads
  // 16 bytes of 4 rows per step; 32 adds per step next to 32 MFMAs.  Replaces a separate pass over dY per layer (cvh_colsum).
  const bool colsum = p.bias_part != nullptr && !fold_bias && k0 == 0;
  float cs[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) cs[j] = 0.f;
  const int cs_off = ((tid & 31) >> 2) * 1024 + (tid >> 5) * 64 + (tid & 3) * 16;  // block of the column chunk + row in its 16-row group + chunk in the block

  const int steps = (m_end - m_begin + 63) / 64;
  if (steps > 0) {
    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  for (int st = 0; st < steps; ++st) {
    const int buf = st & 1;
#if !CVH_TN_GI
    if (st + 1 < steps) issue(st + 1, buf ^ 1);
#endif
    const int stn = st + 1 < steps ? st + 1 : st;  // CVH_TN_GI: the last step re-requests itself into the free buffer (no branch in the body)
    const unsigned char* Yt = smem + buf * TN256_BUF;
    const unsigned char* Xt = Yt + TN256_IMG;
    if (colsum) {  // rows past the end of the split were fetched from the zero line
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        V8<bf16_t> v;
        cvh_u32x4 raw = lds_read_b128_raw(lds_addr32(Yt + q * 8 * 1024 + cs_off));  // (a plain load here draws a compiler-placed vmcnt(0): common.hpp)
        tr_wait1(raw);
        v.d = __builtin_bit_cast(uint4, raw);
        float f[8];
        v8_unpack(v, f);
#pragma unroll
        for (int j = 0; j < 8; ++j) cs[j] += f[j];
      }
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      Frag<bf16_t> a0 = frag_tr256(Yt, kk, wave_n * 64, lane);
      Frag<bf16_t> a1 = frag_tr256(Yt, kk, wave_n * 64 + 32, lane);
      Frag<bf16_t> b[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = frag_tr256(Xt, kk, wave_k * 128 + 32 * j, lane);
      tr_wait(a0.v, a1.v, b[0].v, b[1].v, b[2].v, b[3].v);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        mma32(acc[0][j], a0, b[j]);
        mma32(acc[1][j], a1, b[j]);
#if CVH_TN_GI
        // the next step's eight direct-to-LDS pieces go out one at a time between the MFMA pairs of the first two K slices
        __builtin_amdgcn_sched_barrier(0);
        if (kk == 0) issue_y(stn, buf ^ 1, j);
        if (kk == 1) issue_x(stn, buf ^ 1, j);
        __builtin_amdgcn_sched_barrier(0);
#endif
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  if (colsum) {
    // 16 row groups hold partial sums of the same 8 columns: lanes l and l + 32 of a wave first (row groups 2w, 2w + 1), then the 8 waves
    // through LDS (free after the loop's last barrier) in wave order — a fixed tree, like every other reduction of the step
#pragma unroll
    for (int j = 0; j < 8; ++j) cs[j] += __shfl_xor(cs[j], 32, 64);
    float* red = reinterpret_cast<float*>(smem);  // [8 waves][256 columns]
    if (lane < 32) {
#pragma unroll
      for (int j = 0; j < 8; ++j) red[wave * 256 + (lane & 31) * 8 + j] = cs[j];
    }
    __syncthreads();
    if (tid < 256 && n0 + tid < N) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) t += red[w * 256 + tid];
      p.bias_part[(size_t)by * N + n0 + tid] = t;
    }
  }

  float* dst0 = p.part + (size_t)by * N * K;
#pragma unroll
  for (int fn = 0; fn < 2; ++fn)
#pragma unroll
    for (int fk = 0; fk < 4; ++fk) {
      const int k = k0 + wave_k * 128 + fk * 32 + (lane & 31);
      if (RAGGED && k >= K) {
        if (fold_bias && k == K) {  // the ones column: bias_part[split][n]
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int n = n0 + wave_n * 64 + fn * 32 + acc_row(r, lane);
            if (n < N) p.bias_part[(size_t)by * N + n] = acc[fn][fk][r];
          }
        }
        continue;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wave_n * 64 + fn * 32 + acc_row(r, lane);
        if (!RAGGED || n < N) dst0[(size_t)n * K + k] = acc[fn][fk][r];
      }
    }
}

// shapes the 256 x 256 tile takes: it must move fewer operand columns per row than the 128 x 128 tiling (a property of (N, K) alone, so
// that the scratch planner, cvh_gemm_dw_folds_bias and the launch agree), CVH_TUNE key 19 = 1 switches it off (A/B runs)
bool gemm_tn256_shape(int N, int Ktot) {
  if (cvh_tune_get(CVH_TUNE_NO_TN256)) return false;
  // measured (round 5, same box): ViT-B / CLIP sized products (768 ... 3072 wide) +3.7 % / +2.9 % on the whole step; the MobileViT sized ones
  // (144 ... 720 x 144 ... 240) no faster at 1024 images and slower at 128 (one workgroup per CU hides less latency than two): those stay
  // on the 128 x 128 tiles
  if (N < 512 || Ktot < 512) return false;
  const long long c128 = (long long)((N + 127) / 128) * ((Ktot + 127) / 128) * 256;
  const long long c256 = (long long)((N + 255) / 256) * ((Ktot + 255) / 256) * 512;
  return c256 < c128;
}

bool gemm_tn_big_eligible(const GemmTNParams& p) {
  if (cvh_tune_get(CVH_TUNE_BIG_GEMM) == 0) return false;
  const bool linear = p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0 && p.C2 == 0 && p.src2 == nullptr;
  return linear && p.M >= 2048 && p.N >= 64 && (p.N % 8) == 0 && p.Ktot >= 64 && (p.Ktot % 8) == 0 && p.Cin_real == p.Ktot &&
         (p.m_per_split % 64) == 0;
}

int launch_gemm_tn_big(const GemmTNParams& p, int splits, hipStream_t st) {
  if (gemm_tn256_shape(p.N, p.Ktot)) {  // (p.k_tiles was planned for 256-wide tiles by tn_plan)
    const int out_tiles = ((p.N + 255) / 256) * ((p.Ktot + 255) / 256);
    const bool ragged = (p.N % 256) != 0 || (p.Ktot % 256) != 0;
    static DynSmemAttr attr_r, attr_p;
    const void* fn = ragged ? reinterpret_cast<const void*>(gemm_tn256_kernel<true>) : reinterpret_cast<const void*>(gemm_tn256_kernel<false>);
    if (hipError_t e = (ragged ? attr_r : attr_p).ensure(fn, 2 * TN256_BUF); e != hipSuccess) return (int)e;
    if (ragged) hipLaunchKernelGGL(gemm_tn256_kernel<true>, dim3(out_tiles, splits), dim3(512), 2 * TN256_BUF, st, p);
    else hipLaunchKernelGGL(gemm_tn256_kernel<false>, dim3(out_tiles, splits), dim3(512), 2 * TN256_BUF, st, p);
    CVH_CHECK_LAUNCH();
    return 0;
  }
  const int out_tiles = ((p.N + 127) / 128) * ((p.Ktot + 127) / 128);
  if ((p.N % 128) == 0 && (p.Ktot % 128) == 0) hipLaunchKernelGGL(gemm_tn128_kernel<false>, dim3(out_tiles, splits), dim3(256), 2 * BUF_BYTES, st, p);
  else hipLaunchKernelGGL(gemm_tn128_kernel<true>, dim3(out_tiles, splits), dim3(256), 2 * BUF_BYTES, st, p);
  CVH_CHECK_LAUNCH();
  return 0;
}
