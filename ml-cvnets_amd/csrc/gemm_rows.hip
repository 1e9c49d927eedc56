// gemm_tn_rows_kernel: dW[n][k] = sum_m dY[m][n] X[m][k] for the token linears of the MobileViT blocks (N, K in 96 ... 720) under a million
// rows — the weight-gradient half of LinearLayer / 1x1 Conv2d backward (cvnets/layers/linear_layer.py:74-91, cvnets/layers/conv_layer.py:254-255).
//
// Why another dW kernel.  gemm_tn128_kernel (gemm_big.hip) cuts the output into 128 x 128 tiles; every workgroup then streams its 128 dY
// columns and 128 X columns of every row into LDS.  For N = 288, K = 144 that is 6 tiles x 512 bytes per row against 864 unique bytes, half
// of it zero padding (144 -> 256 columns): measured, the kernel moves 10.3 TB/s into LDS on EVERY one of these shapes — the rate at which the
// CUs take direct-to-LDS lines — and therefore 2.9 TB/s of unique operand bytes (profiles/r05z_step_trace.txt: 305 ... 414 us for 0.9 ... 1.2 GB).
// Here a workgroup owns WHOLE rows: one 16-wave workgroup per CU holds an [NP x KP] block of the output in its accumulators (NP = N / n_parts,
// KP = K / k_parts, parts only where N x K does not fit 16 x 72 accumulator registers) and streams the rows once:
//   * per 32-row stage the dY row pieces [32][NP] and the X row pieces [32][KP] land in LDS by global_load_lds (16 bytes per lane, dense
//     rows, pitch padded to = 2 (mod 4) chunks: the transpose reads of 8 consecutive rows then hit 8 disjoint bank octets); every operand
//     byte crosses the L2 -> LDS path once per part instead of once per 128-column tile, with no zero padding;
//   * NSTAGE (4 ... 8) stages in LDS, NSTAGE - 1 of them in flight: 55 ... 130 KB of useful bytes per CU under way (the 128 x 128 kernel: 64 KB,
//     a third of it useful); each wave waits for ITS OWN pieces of the oldest stage with a counted s_waitcnt, one barrier per stage;
//   * v_mfma_f32_16x16x32_bf16 on 16 x 16 output tiles (144 = 9 x 16, 192 = 12 x 16, 240 = 15 x 16: no padded columns), both operands read
//     "down the rows" with ds_read_b64_tr_b16; wave (wn, wk) of the WN x WK arrangement owns PN x PK tiles, PN + PK fragment reads per PN * PK MFMAs;
//   * the bias gradient (column sums of dY) comes from the dY image in LDS (one 16-byte read + 8 adds per thread and stage) in the k part 0
//     workgroups: dY is read once for dW and db.
// Output: partial [split][N][K] rows (+ bias_part[split][N]) — the contract of cvh_gemm_dw's scratch; the splits are summed by
// gemm_dw_reduce / cvh_reduce_multi in a fixed order (no atomics).
#include "common.hpp"
#include "cvnets_hip.h"
#include "gemm_params.hpp"

namespace {

typedef __attribute__((ext_vector_type(4))) float gr_f32x4;
typedef short gr_v4s __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void* gr_gptr_t;
typedef __attribute__((address_space(3))) void* gr_lptr_t;

__device__ __attribute__((aligned(128))) unsigned char gr_zero_line[128];  // source of rows past the end of a split and of the pitch padding

__device__ __forceinline__ void gr_glds16(const void* g, unsigned char* l) { __builtin_amdgcn_global_load_lds((gr_gptr_t)g, (gr_lptr_t)l, 16, 0, 0); }

// inline-asm transpose reads (common.hpp: the builtin form draws a compiler-placed s_waitcnt vmcnt(0) in front of every read while direct-to-LDS
// loads are outstanding - this kernel keeps three stages of them in flight by design); the caller runs tr_wait1() on the fragment before its first MFMA
__device__ __forceinline__ bf16x8_t gr_tr_frag(const unsigned char* lo, int hi_off) { return tr_frag_raw2<0>(lds_addr32(lo), (unsigned)hi_off); }
// the builtin form (draws the compiler's vmcnt(0)): kept for the 4 x 4 rectangle, where the asm form's address registers spill
__device__ __forceinline__ bf16x8_t gr_tr_frag_builtin(const unsigned char* lo, int hi_off) {
  const gr_v4s a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) gr_v4s*)(lo));
  const gr_v4s b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) gr_v4s*)(lo + hi_off));
  return __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}

template <int N> __device__ __forceinline__ void gr_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

constexpr int GR_WAVES = 16;
constexpr int GR_ROWS = 32;    // rows per stage = one K step of the 16x16x32 MFMA
constexpr int GR_NSTAGE = 4;   // stages in LDS, three of them in flight

// IPW: direct-to-LDS instructions per wave and stage (every wave issues exactly IPW: the waits are immediates)
template <int PN, int PK, int IPW>
__global__ __launch_bounds__(64 * GR_WAVES) void gemm_tn_rows_kernel(GemmTNParams p, TnRowsGeom g) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-contiguous order with the parts of one split adjacent: the workgroups that stream the SAME rows run on one XCD at the same time
  const int lb = xcd_chunk_id((int)blockIdx.x, (int)gridDim.x);
  const int parts = g.n_parts * g.k_parts;
  const int part_id = lb % parts, by = lb / parts;
  const int np = part_id / g.k_parts, kp = part_id % g.k_parts;
  const int n0 = np * g.NP, k0 = kp * g.KP;
  const int N = p.N, K = p.Ktot;
  // Stage-cyclic rows: workgroup `by` takes the 32-row stages by, by + splits, by + 2 splits, ... — at any moment the splits read ONE
  // contiguous window of the operands, spread over all HBM channels.  (Contiguous row ranges per workgroup put the 256 concurrent streams
  // 4096 rows x 288 bytes = 9 x 2^17 bytes apart for the million-row linears: every stream on the same channel at the same time, 3.3 TB/s.)
  const int total_stages = (p.M + GR_ROWS - 1) / GR_ROWS;
  const int nsteps = by < total_stages ? (total_stages - by + g.splits - 1) / g.splits : 0;
  const bf16_t* __restrict__ dy = reinterpret_cast<const bf16_t*>(p.dy) + (size_t)by * GR_ROWS * N + n0;
  const bf16_t* __restrict__ xs = reinterpret_cast<const bf16_t*>(p.src1) + (size_t)by * GR_ROWS * K + k0;
  const int row0 = by * GR_ROWS, rstep = g.splits * GR_ROWS;

  // ---- this lane's direct-to-LDS pieces: block b = wave + 16 j of the stage image covers chunk slots 64 b ... 64 b + 63 ----
  const int ychunks = GR_ROWS * g.py, stage_chunks = ychunks + GR_ROWS * g.px, nblk = stage_chunks / 64;
  const bf16_t* src[IPW];
  int srow[IPW], sinc[IPW];
#pragma unroll
  for (int j = 0; j < IPW; ++j) {
    const int slot = 64 * (wave + GR_WAVES * j) + lane;
    const bool isy = slot < ychunks;
    const int s2 = isy ? slot : slot - ychunks;
    const int pitch = isy ? g.py : g.px, width = isy ? g.NP / 8 : g.KP / 8, ld = isy ? N : K;
    const int r = s2 / pitch, c = s2 - r * pitch;
    const bool ok = slot < stage_chunks && c < width;
    src[j] = ok ? (isy ? dy : xs) + (size_t)r * ld + c * 8 : nullptr;
    srow[j] = ok ? r : 0x40000000;  // never below M: the zero line
    sinc[j] = ld;
  }
  auto issue = [&](int st) __attribute__((always_inline)) {
    unsigned char* dst = smem + (st & (GR_NSTAGE - 1)) * g.stage_bytes + wave * 1024;
#pragma unroll
    for (int j = 0; j < IPW; ++j) {
      if (!(g.dbg & 4)) {
        const bool ok = row0 + srow[j] + st * rstep < p.M;
        // a block past the stage image (the last round of the deal) lands in a scratch area behind the stages: every wave issues IPW instructions
        unsigned char* d = (wave + GR_WAVES * j < nblk) ? dst + j * (GR_WAVES * 1024) : smem + GR_NSTAGE * g.stage_bytes + wave * 1024;
        gr_glds16(ok ? reinterpret_cast<const void*>(src[j] + (size_t)st * rstep * sinc[j]) : reinterpret_cast<const void*>(gr_zero_line), d);
      }
    }
  };

  // ---- tiles of this wave ----
  const int wn = wave / g.WK, wk = wave - wn * g.WK;
  const int tnb = g.NP / 16, tkb = g.KP / 16;
  gr_f32x4 acc[PN][PK];
#pragma unroll
  for (int i = 0; i < PN; ++i)
#pragma unroll
    for (int j = 0; j < PK; ++j) acc[i][j] = gr_f32x4{0.f, 0.f, 0.f, 0.f};
  // transpose reads: K slot (l4, e) <-> stage row 4 l4 + e (e < 4), 16 + 4 l4 + (e - 4); lane i = 4 r + q of a 16-lane group addresses row r, columns 4 q ...
  const int rd_row = 4 * l4 + (l15 >> 2), rd_col = 8 * (l15 & 3);
  const int a_off = rd_row * g.py * 16 + rd_col, b_off = ychunks * 16 + rd_row * g.px * 16 + rd_col;
  const int a_hi = 16 * g.py * 16, b_hi = 16 * g.px * 16;

  // bias gradient: thread (row r = tid / cpr, chunk c = tid % cpr) of the k part 0 workgroups sums its chunk of the dY image
  const int cpr = g.NP / 8;
  const bool do_bias = p.bias_part != nullptr && kp == 0;
  const bool bias_thread = do_bias && tid < GR_ROWS * cpr;
  const int bias_off = bias_thread ? ((tid / cpr) * g.py + (tid % cpr)) * 16 : 0;
  float cs[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) cs[j] = 0.f;

  for (int st = 0; st < GR_NSTAGE - 1 && st < nsteps; ++st) issue(st);
  for (int st = 0; st < nsteps; ++st) {
    // this wave's pieces of stage st have landed when at most the instructions of the stages issued after it are outstanding
    const int later = nsteps - 1 - st;
    if (later >= GR_NSTAGE - 2) gr_wait_vm<(GR_NSTAGE - 2) * IPW>();
    else if (later == 1) gr_wait_vm<IPW>();
    else gr_wait_vm<0>();
    wg_barrier_lds();  // every wave's pieces of stage st have landed; every wave is done with stage st - 1, whose buffer is refilled now
    if (st + GR_NSTAGE - 1 < nsteps) issue(st + GR_NSTAGE - 1);
    const unsigned char* img = smem + (st & (GR_NSTAGE - 1)) * g.stage_bytes;
    if constexpr (PN * PK >= 16) {
      // 4 x 4 rectangle: the builtin reads, one dY fragment at a time (round-5 form; 122 registers - the asm form below spills here, and a
      // spill reload inside the loop is a vmcnt(0) of its own)
      if (bias_thread) {
        V8<bf16_t> v;
        v.d = *reinterpret_cast<const uint4*>(img + bias_off);
        float f[8];
        v8_unpack(v, f);
#pragma unroll
        for (int j = 0; j < 8; ++j) cs[j] += f[j];
      }
      bf16x8_t bf[PK];
      if (!(g.dbg & 1)) {
#pragma unroll
        for (int j = 0; j < PK; ++j) {
          const int kb = wk * PK + j;
          bf[j] = gr_tr_frag_builtin(img + b_off + (kb < tkb ? kb : 0) * 32, b_hi);
        }
      }
#pragma unroll
      for (int i = 0; i < PN; ++i) {
        const int nb = wn * PN + i;
        bf16x8_t af;
        if (!(g.dbg & 1)) af = gr_tr_frag_builtin(img + a_off + (nb < tnb ? nb : 0) * 32, a_hi);
#pragma unroll
        for (int j = 0; j < PK; ++j) {
          if (!(g.dbg & 2) && nb < tnb && wk * PK + j < tkb)  // wave-uniform
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf[j], acc[i][j], 0, 0, 0);
        }
        if (i & 1) __builtin_amdgcn_sched_barrier(0);
      }
    } else {
    // every LDS read of the stage is issued up front (inline-asm forms, common.hpp: no compiler-placed vmcnt(0) in front of them) and waited
    // for once, then the MFMAs: the LDS latency is paid once per stage
    cvh_u32x4 braw;
    if (bias_thread) braw = lds_read_b128_raw(lds_addr32(img + bias_off));
    bf16x8_t bf[PK], af[PN];
    if (!(g.dbg & 1)) {
#pragma unroll
      for (int j = 0; j < PK; ++j) {
        const int kb = wk * PK + j;
        bf[j] = gr_tr_frag(img + b_off + (kb < tkb ? kb : 0) * 32, b_hi);
      }
#pragma unroll
      for (int i = 0; i < PN; ++i) {
        const int nb = wn * PN + i;
        af[i] = gr_tr_frag(img + a_off + (nb < tnb ? nb : 0) * 32, a_hi);
      }
#pragma unroll
      for (int j = 0; j < PK; ++j) tr_wait1(bf[j]);
#pragma unroll
      for (int i = 0; i < PN; ++i) tr_wait1(af[i]);
    }
    if (bias_thread) {
      tr_wait1(braw);
      V8<bf16_t> v;
      v.d = __builtin_bit_cast(uint4, braw);
      float f[8];
      v8_unpack(v, f);
#pragma unroll
      for (int j = 0; j < 8; ++j) cs[j] += f[j];
    }
#pragma unroll
    for (int i = 0; i < PN; ++i) {
      const int nb = wn * PN + i;
#pragma unroll
      for (int j = 0; j < PK; ++j) {
        if (!(g.dbg & 2) && nb < tnb && wk * PK + j < tkb)  // wave-uniform
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);  // D[n][k] += sum_m dY[m][n] X[m][k]
      }
    }
    }
  }

  // ---- partial results ----
  float* dst = p.part + (size_t)by * N * K;
#pragma unroll
  for (int i = 0; i < PN; ++i)
#pragma unroll
    for (int j = 0; j < PK; ++j) {
      const int nb = wn * PN + i, kb = wk * PK + j;
      if (nb < tnb && kb < tkb) {
        const int k = k0 + kb * 16 + l15;
#pragma unroll
        for (int e = 0; e < 4; ++e) dst[(size_t)(n0 + nb * 16 + 4 * l4 + e) * K + k] = acc[i][j][e];
      }
    }
  if (do_bias) {
    __syncthreads();  // the stage buffers are free
    float* red = reinterpret_cast<float*>(smem);  // [32 rows][NP]
    if (bias_thread) {
#pragma unroll
      for (int j = 0; j < 8; ++j) red[(tid / cpr) * g.NP + (tid % cpr) * 8 + j] = cs[j];
    }
    __syncthreads();
    for (int n = tid; n < g.NP; n += 64 * GR_WAVES) {
      float t = 0.f;
#pragma unroll 8
      for (int r = 0; r < GR_ROWS; ++r) t += red[r * g.NP + n];
      p.bias_part[(size_t)by * N + n0 + n] = t;
    }
  }
}

// instantiations: per-wave tile rectangles (PN x PK); a plan's rectangle is rounded up to the next one
struct GrInst { int pn, pk; };
constexpr GrInst GR_INST[] = {{3, 3}, {4, 4}, {5, 3}, {3, 5}};  // (a 7 x 3 rectangle — N = 432 in one part — spills 21 registers and measured 505 us against the tiled kernel's 399)
constexpr int GR_MAXI = 3;

int pad_pitch(int chunks) {  // = 2 (mod 4): 8 consecutive rows x 8 dwords of a transpose read fall on 8 disjoint bank octets
  while ((chunks & 3) != 2) ++chunks;
  return chunks;
}

}  // namespace

// A property of the SHAPE alone (the scratch planner, the bias-fold query and the launch must agree); CVH_TUNE key 22 = 1 switches the kernel off
bool gemm_tn_rows_plan(int M, int N, int K, TnRowsGeom* out) {
  if (cvh_tune_get(22)) return false;
  if (M < 262144 || (N % 16) || (K % 16) || N < 64 || K < 64 || N > 1024 || K > 1024) return false;
  TnRowsGeom best;
  long long best_cost = -1;
  for (int n_parts = 1; n_parts <= 4; ++n_parts)
    for (int k_parts = 1; k_parts <= 2; ++k_parts) {
      if (N % (16 * n_parts) || K % (16 * k_parts)) continue;
      const int NP = N / n_parts, KP = K / k_parts, tnb = NP / 16, tkb = KP / 16;
      const int arr[5][2] = {{4, 4}, {8, 2}, {2, 8}, {16, 1}, {1, 16}};
      for (const auto& a : arr) {
        const int pn = (tnb + a[0] - 1) / a[0], pk = (tkb + a[1] - 1) / a[1];
        int inst = -1;
        for (int i = 0; i < (int)(sizeof(GR_INST) / sizeof(GR_INST[0])); ++i)
          if (pn <= GR_INST[i].pn && pk <= GR_INST[i].pk) { inst = i; break; }
        if (inst < 0) continue;
        TnRowsGeom g;
        g.n_parts = n_parts; g.k_parts = k_parts; g.NP = NP; g.KP = KP; g.WN = a[0]; g.WK = a[1];
        g.PN = GR_INST[inst].pn; g.PK = GR_INST[inst].pk;
        g.py = pad_pitch(NP / 8); g.px = pad_pitch(KP / 8);
        const int stage_chunks = GR_ROWS * (g.py + g.px);
        const int nblk = stage_chunks / 64;  // (py + px) % 4 == 0: whole 1 KB blocks
        g.ipw = (nblk + GR_WAVES - 1) / GR_WAVES;
        if (g.ipw > GR_MAXI) continue;
        g.stage_bytes = nblk * 1024;  // the blocks of a stage are dealt to the waves round-robin; the surplus of the last round goes to a scratch area
        if (GR_NSTAGE * g.stage_bytes + ((nblk % GR_WAVES) ? GR_WAVES * 1024 : 0) > 160 * 1024) continue;
        g.nstage = GR_NSTAGE;
        g.dbg = cvh_tune_get(24);
        const int ns = GR_NSTAGE;
        if ((size_t)GR_ROWS * NP * 4 > (size_t)ns * g.stage_bytes) continue;  // the bias reduction reuses the stage buffers
        g.splits = 256 / (n_parts * k_parts);
        int mps = (M + g.splits - 1) / g.splits;
        mps = (mps + 63) / 64 * 64;
        g.m_per_split = mps;
        g.splits = (M + mps - 1) / mps;
        // the partial tiles are written once and read once more by the reduction: only where the operand stream dwarfs them
        if ((long long)M * (N + K) * 2 < 8LL * g.splits * N * K * 4) continue;
        // measured (tools/bench_dw.py, 1 M rows): one part 3.6 ... 4.7 TB/s against 2.6 ... 3.1 of the 128 x 128 tiles; three parts (qkv, 432 x
        // 144: X streamed three times, 85 rows per workgroup column) 2.6 against 3.0 — parts stay on the tiled kernel unless CVH_TUNE key 23 = 1
        if (n_parts * k_parts > 1 && cvh_tune_get(23) == 0) continue;
        // operand columns that cross the L2 -> LDS path per row; fewer parts on a tie, then the squarer wave arrangement
        const long long cost = ((long long)n_parts * K + (long long)k_parts * N) * 64 + (g.PN * g.PK) + (a[0] == 4 ? 0 : 1);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = g; }
      }
    }
  if (best_cost < 0) return false;
  if (out) *out = best;
  return true;
}

bool gemm_tn_rows_eligible(const GemmTNParams& p) {
  const bool linear = p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0 && p.C2 == 0 && p.src2 == nullptr;
  return linear && p.Cin_real == p.Ktot && p.part != nullptr && p.dy_xf.mode == 0 && p.x_xf.mode == 0 && gemm_tn_rows_plan(p.M, p.N, p.Ktot, nullptr);
}

int launch_gemm_tn_rows(const GemmTNParams& p, hipStream_t st) {
  TnRowsGeom g;
  if (!gemm_tn_rows_plan(p.M, p.N, p.Ktot, &g)) return -2;
  if (g.m_per_split != p.m_per_split) return -2;  // tn_plan and this launch must agree on the partial rows
  const size_t smem = (size_t)g.nstage * g.stage_bytes + (((g.stage_bytes / 1024) % GR_WAVES) ? GR_WAVES * 1024 : 0);
  const dim3 grid(g.splits * g.n_parts * g.k_parts), block(64 * GR_WAVES);
#define GR_LAUNCH(PN_, PK_, I_)                                                                                               \
  if (g.PN == PN_ && g.PK == PK_ && g.ipw == I_) {                                                                            \
    static DynSmemAttr attr;                                                                                                 \
    if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(gemm_tn_rows_kernel<PN_, PK_, I_>), smem); e != hipSuccess) return (int)e; \
    hipLaunchKernelGGL((gemm_tn_rows_kernel<PN_, PK_, I_>), grid, block, smem, st, p, g);                                    \
    CVH_CHECK_LAUNCH();                                                                                                      \
    return 0;                                                                                                                \
  }
  GR_LAUNCH(3, 3, 1) GR_LAUNCH(3, 3, 2) GR_LAUNCH(4, 4, 2) GR_LAUNCH(5, 3, 2) GR_LAUNCH(3, 5, 2) GR_LAUNCH(5, 3, 3) GR_LAUNCH(3, 5, 3) GR_LAUNCH(4, 4, 3)
#undef GR_LAUNCH
  return -2;
}
