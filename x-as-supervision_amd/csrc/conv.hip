// fp32 MFMA implicit-GEMM convolution family for gfx950: forward, data gradient
// (== ConvTranspose2d forward), weight gradient.  NHWC activations, packed weights
// [rows][R][S][chan] so both GEMM operands are "row x K-contiguous".
//
//   fwd   : Y[m, co]  = sum_{r,s,c} X[pix(m,r,s), c] * W[co][r][s][c]          M = N*Ho*Wo
//   dgrad : dX[p, ci] = sum_{r,s,co} dY[src(p,r,s), co] * Wt[ci][r][s][co]     per stride phase
//   wgrad : dW[co, (r,s,c)] = sum_m dY[m, co] * X[pix(m,r,s), c]               split over m
//
// Block = 256 threads = 4 waves, tile BM x BN x 32, v_mfma_f32_32x32x2_f32 (exact fp32,
// 64 FLOP/clk/SIMD).  Operands are staged global -> registers -> LDS (double buffered,
// one barrier per K-step, the next K-step's global loads in flight during the MFMAs).
// LDS rows are padded to 36 dwords: a lane reads 4 consecutive k as one ds_read_b128 and
// feeds 4 MFMAs (MFMA j of a lane-half h consumes k = 4h + j on both operands), which is
// bank-conflict free for the 16-lane b128 groups.
//
// Replaces the cuDNN conv kernels behind integral_base_modules/resnet.py:16-47,
// deconv_head.py:24-35, physique_network.py:15-50 and torchvision's Bottleneck.
#include "conv_shared.h"

namespace xas {

// 64 bytes of zeros: out-of-range taps / rows load from here instead of branching around the load
__device__ __constant__ float4 g_zero16[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};

static int g_tune = 0;
int tune_flags() { return g_tune; }

// Operand roles are swapped (MFMA "A" = weight rows, "B" = pixel rows), so an accumulator holds
// D[n][m]: lane = pixel m, registers 4g..4g+3 = four CONSECUTIVE output channels -> float4 stores.
// Consecutive MFMAs go to DIFFERENT accumulators (k outer, tile inner): a chain of dependent
// v_mfma_f32_32x32x2_f32 on one accumulator does not issue back to back, and a wave with a single 32x32
// tile (64x64 block tile) keeps two partial accumulators (even / odd k) that are summed in the epilogue.
template <int BM, int BN>
__device__ __forceinline__ void mfma_tile(const float* __restrict__ As, const float* __restrict__ Bs,
                                          f32x16 (&acc)[TileCfg<BM, BN>::MI][TileCfg<BM, BN>::NI],
                                          f32x16& acc2, int wm, int wn, int lane) {
  using C = TileCfg<BM, BN>;
  constexpr bool SPLIT = (C::MI == 1 && C::NI == 1);
  const int i = lane & 31, h = lane >> 5;
  float4 a[C::MI], b[C::NI];
#pragma unroll
  for (int kk = 0; kk < BK / 8; ++kk) {
#pragma unroll
    for (int mi = 0; mi < C::MI; ++mi)
      a[mi] = *reinterpret_cast<const float4*>(As + (wm * C::WM + mi * 32 + i) * LDK + kk * 8 + h * 4);
#pragma unroll
    for (int ni = 0; ni < C::NI; ++ni)
      b[ni] = *reinterpret_cast<const float4*>(Bs + (wn * C::WN + ni * 32 + i) * LDK + kk * 8 + h * 4);
    if (SPLIT) {
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[0].x, a[0].x, acc[0][0], 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(b[0].y, a[0].y, acc2, 0, 0, 0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[0].z, a[0].z, acc[0][0], 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(b[0].w, a[0].w, acc2, 0, 0, 0);
    } else {
#pragma unroll
      for (int mi = 0; mi < C::MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < C::NI; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[ni].x, a[mi].x, acc[mi][ni], 0, 0, 0);
#pragma unroll
      for (int mi = 0; mi < C::MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < C::NI; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[ni].y, a[mi].y, acc[mi][ni], 0, 0, 0);
#pragma unroll
      for (int mi = 0; mi < C::MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < C::NI; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[ni].z, a[mi].z, acc[mi][ni], 0, 0, 0);
#pragma unroll
      for (int mi = 0; mi < C::MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < C::NI; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[ni].w, a[mi].w, acc[mi][ni], 0, 0, 0);
    }
  }
}

// ------------------------------------------------------------------------------------
// fwd (MODE 0) and dgrad (MODE 1)
// ------------------------------------------------------------------------------------
// One K-slice (8 k) of the block tile: fragment loads and the MFMAs that consume them, separately callable so the
// pipelined K-loop can order them by hand.
template <int BM, int BN>
__device__ __forceinline__ void frag_load(const float* __restrict__ As, const float* __restrict__ Bs, int kk, int wm, int wn,
                                          int lane, float4 (&a)[TileCfg<BM, BN>::MI], float4 (&b)[TileCfg<BM, BN>::NI]) {
  using C = TileCfg<BM, BN>;
  const int i = lane & 31, h = lane >> 5;
#pragma unroll
  for (int mi = 0; mi < C::MI; ++mi)
    a[mi] = *reinterpret_cast<const float4*>(As + (wm * C::WM + mi * 32 + i) * LDK + kk * 8 + h * 4);
#pragma unroll
  for (int ni = 0; ni < C::NI; ++ni)
    b[ni] = *reinterpret_cast<const float4*>(Bs + (wn * C::WN + ni * 32 + i) * LDK + kk * 8 + h * 4);
}

template <int BM, int BN>
__device__ __forceinline__ void mfma_slice(const float4 (&a)[TileCfg<BM, BN>::MI], const float4 (&b)[TileCfg<BM, BN>::NI],
                                           f32x16 (&acc)[TileCfg<BM, BN>::MI][TileCfg<BM, BN>::NI], f32x16& acc2) {
  using C = TileCfg<BM, BN>;
  if (C::MI == 1 && C::NI == 1) {
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[0].x, a[0].x, acc[0][0], 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(b[0].y, a[0].y, acc2, 0, 0, 0);
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[0].z, a[0].z, acc[0][0], 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(b[0].w, a[0].w, acc2, 0, 0, 0);
  } else {
#pragma unroll
    for (int mi = 0; mi < C::MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < C::NI; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[ni].x, a[mi].x, acc[mi][ni], 0, 0, 0);
#pragma unroll
    for (int mi = 0; mi < C::MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < C::NI; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[ni].y, a[mi].y, acc[mi][ni], 0, 0, 0);
#pragma unroll
    for (int mi = 0; mi < C::MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < C::NI; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[ni].z, a[mi].z, acc[mi][ni], 0, 0, 0);
#pragma unroll
    for (int mi = 0; mi < C::MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < C::NI; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[ni].w, a[mi].w, acc[mi][ni], 0, 0, 0);
  }
}

// Scheduling hint for one K-slice of the pipelined loop: NMF MFMAs with NOTH memory operations of kind MASK
// (0x100 LDS read, 0x200 LDS write, 0x020 global read) spread evenly between them.
template <int NMF, int NOTH, int MASK>
__device__ __forceinline__ void sched_mix() {
  constexpr int G = NOTH > 0 ? (NMF / NOTH > 0 ? NMF / NOTH : 1) : NMF;
  constexpr int USED = NOTH > 0 ? (G * NOTH < NMF ? G * NOTH : NMF) : 0;
#pragma unroll
  for (int i = 0; i < NOTH; ++i) {
    if (i * G < NMF) __builtin_amdgcn_sched_group_barrier(0x008, G, 0);
    __builtin_amdgcn_sched_group_barrier(MASK, 1, 0);
  }
  if (NMF - USED > 0) __builtin_amdgcn_sched_group_barrier(0x008, NMF - USED, 0);
}

template <int BM, int BN, int MODE, int NBUF = 2, bool PIPE = false>
__global__ __launch_bounds__(256) void igemm_kernel(IgemmParams p) {
  using C = TileCfg<BM, BN>;
  constexpr int APASS = BM / 32, BPASS = BN / 32;
  extern __shared__ __align__(16) float lds[];
  float* As = lds;                       // [NBUF][BM][LDK]
  float* Bs = lds + NBUF * BM * LDK;     // [NBUF][BN][LDK]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / C::WAVES_N, wn = wave % C::WAVES_N;
  const int kq = tid & 7, lrow = tid >> 3;          // float4 slot along k, row within a 32-row pass

  // ---- per-mode geometry -----------------------------------------------------------
  int Hrow = p.Hrow, Wrow = p.Wrow;
  int ph = 0, pw = 0, base_r = 0, base_s = 0, nr = p.R, ns = p.S, off_h = -p.pad, off_w = -p.pad, rstep = 1;
  int sa = p.stride;                                 // source step per row-grid step
  if (MODE == 1) {
    const int st = p.stride;
    ph = blockIdx.z / st; pw = blockIdx.z % st;
    Hrow = (p.Hd - ph + st - 1) / st; Wrow = (p.Wd - pw + st - 1) / st;
    base_r = (ph + p.pad) % st; base_s = (pw + p.pad) % st;
    nr = base_r < p.R ? (p.R - base_r + st - 1) / st : 0;
    ns = base_s < p.S ? (p.S - base_s + st - 1) / st : 0;
    off_h = (ph + p.pad - base_r) / st; off_w = (pw + p.pad - base_s) / st;
    rstep = st; sa = 1;
  }
  const int Mrows = p.N * Hrow * Wrow;
  // XCD-aware tile order.  Workgroups are dealt round-robin over the 8 XCDs (private 4 MiB L2 each):
  // linear id L runs on XCD L % 8.  Give every XCD a contiguous range of M-tiles and walk the N-tiles of one
  // M-tile back to back, so the gathered activation rows are fetched into that XCD's L2 once and re-used by
  // all N-tiles, while the (small) weight matrix stays L2-resident.  Placement only affects speed.
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
  const int mt = xcd * p.mt_per_xcd + q / p.nNt;
  const int nt = q - (q / p.nNt) * p.nNt;
  if (mt >= p.nMt) return;
  const int m0 = mt * BM, n0 = nt * BN;
  if (m0 >= Mrows) return;                           // uniform per block (uneven phases)
  const int HW = Hrow * Wrow;

  // rows this thread stages: lrow + 32*j
  int rn[APASS], ra[APASS], rb[APASS];               // image, source base coords (already * sa + off)
  long rbase[APASS];                                 // element offset of (rn, ra, rb, channel 0) in the source
#pragma unroll
  for (int j = 0; j < APASS; ++j) {
    const int m = m0 + lrow + 32 * j;
    if (m < Mrows) {
      int n, a, b;
      if (MODE == 0) { n = p.div_hw.div(m); const int rem = m - n * HW; a = p.div_w.div(rem); b = rem - a * Wrow; }
      else { n = m / HW; const int rem = m - n * HW; a = rem / Wrow; b = rem - a * Wrow; }
      rn[j] = n; ra[j] = a * sa + off_h; rb[j] = b * sa + off_w;
      rbase[j] = (((long)n * p.Hs + ra[j]) * p.Ws + rb[j]) * p.Cs;
    } else { rn[j] = -1; ra[j] = 0; rb[j] = 0; rbase[j] = 0; }
  }
  const int cchunks = p.Cs / BK;
  const int nk = nr * ns * cchunks;
  const size_t wrow_stride = (size_t)p.R * p.S * p.Cs;

  f32x16 acc[C::MI][C::NI];
#pragma unroll
  for (int mi = 0; mi < C::MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < C::NI; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  f32x16 acc2;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc2[e] = 0.f;

  // Loads are issued UNCONDITIONALLY from a clamped (always valid) address; rows that are out of range are
  // zeroed later, when the registers are written to LDS.  A `cond ? load : 0` makes hipcc branch around each
  // load and wait vmcnt(0) in the middle of the prefetch, exposing the full memory latency every K-step.
  // Two register sets: the global loads of K-step ks+2 are issued while step ks is computed, so a load has
  // TWO MFMA phases to arrive (in-kernel stamps showed 3-4k cycles of load latency under load against a
  // 2-4k cycle MFMA phase: with one step of look-ahead the waves sat in s_waitcnt vmcnt before every LDS
  // store).  The sets are separate named arrays and the loop is unrolled by two: a runtime-indexed register
  // array would be demoted to scratch memory.
  float4 ra4_0[APASS], rb4_0[BPASS], ra4_1[APASS], rb4_1[BPASS];
  unsigned okmask_0 = 0, okmask_1 = 0;
  auto load_step = [&](int ks, float4 (&ra4)[APASS], float4 (&rb4)[BPASS], unsigned& okmask) {
    const int tap = ks / cchunks, c0 = (ks - tap * cchunks) * BK + kq * 4;
    const int jr = tap / ns, js = tap - jr * ns;
    // fwd: source = base + tap ; dgrad: source = base - tap index (transposed walk)
    const int dh = (MODE == 0) ? jr : -jr, dw = (MODE == 0) ? js : -js;
    const int wtap = (base_r + rstep * jr) * p.S + (base_s + rstep * js);
    const long delta = ((long)dh * p.Ws + dw) * p.Cs;          // uniform over the block
    unsigned m = 0;
#pragma unroll
    for (int j = 0; j < APASS; ++j) {
      const int hs = ra[j] + dh, ws = rb[j] + dw;
      const bool ok = rn[j] >= 0 && (unsigned)hs < (unsigned)p.Hs && (unsigned)ws < (unsigned)p.Ws;
      const long off = ok ? rbase[j] + delta : 0l;
      ra4[j] = *reinterpret_cast<const float4*>(p.src + off + c0);
      m |= (ok ? 1u : 0u) << j;
    }
#pragma unroll
    for (int j = 0; j < BPASS; ++j) {
      const int n = n0 + lrow + 32 * j;
      const bool ok = n < p.Cd;
      rb4[j] = *reinterpret_cast<const float4*>(p.wgt + (ok ? (size_t)n * wrow_stride : (size_t)0) + (size_t)wtap * p.Cs + c0);
      m |= (ok ? 1u : 0u) << (16 + j);
    }
    okmask = m;
  };
  auto store_step = [&](int buf, const float4 (&ra4)[APASS], const float4 (&rb4)[BPASS], unsigned okmask) {
    float* a = As + buf * BM * LDK;
    float* b = Bs + buf * BN * LDK;
#pragma unroll
    for (int j = 0; j < APASS; ++j) {
      const bool ok = (okmask >> j) & 1u;
      float4 v = ra4[j];
      v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
      *reinterpret_cast<float4*>(a + (lrow + 32 * j) * LDK + kq * 4) = v;
    }
#pragma unroll
    for (int j = 0; j < BPASS; ++j) {
      const bool ok = (okmask >> (16 + j)) & 1u;
      float4 v = rb4[j];
      v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
      *reinterpret_cast<float4*>(b + (lrow + 32 * j) * LDK + kq * 4) = v;
    }
  };

#define XAS_KSTEP(KS, BUF, RA, RB, MASK)                                                     \
  {                                                                                          \
    store_step(BUF, RA, RB, MASK);                                                           \
    __syncthreads();                                                                         \
    if ((KS) + 2 < nk) load_step((KS) + 2, RA, RB, MASK);                                    \
    mfma_tile<BM, BN>(As + (BUF) * BM * LDK, Bs + (BUF) * BN * LDK, acc, acc2, wm, wn, lane); \
  }
  if constexpr (PIPE) {
    // Pipelined K-loop.  The LDS stores of step ks+1 and the global loads of step ks+3 are issued INSIDE the MFMA
    // sequence of step ks (the other LDS buffer is free as soon as every wave has passed this step's barrier), so a
    // wave's only exposed work per K-step is the barrier and the first fragment read.  Loads past the last step are
    // clamped to it (valid addresses, results never consumed) to keep the loop body free of branches.
    constexpr int NMF = C::MI * C::NI * 4;                 // MFMAs per K-slice
    constexpr int NRD = C::MI + C::NI;                     // fragment reads per K-slice
    constexpr int NST = APASS + BPASS;                     // LDS stores == global loads per K-step
    const int last = nk - 1;
    auto pstep = [&](int buf, float4 (&ra4)[APASS], float4 (&rb4)[BPASS], unsigned& okmask, int ks_load) {
      const float* a_s = As + buf * BM * LDK;
      const float* b_s = Bs + buf * BN * LDK;
      float4 fa0[C::MI], fb0[C::NI], fa1[C::MI], fb1[C::NI];
      __syncthreads();
      frag_load<BM, BN>(a_s, b_s, 0, wm, wn, lane, fa0, fb0);
      frag_load<BM, BN>(a_s, b_s, 1, wm, wn, lane, fa1, fb1);
      mfma_slice<BM, BN>(fa0, fb0, acc, acc2);
      store_step(buf ^ 1, ra4, rb4, okmask);
      frag_load<BM, BN>(a_s, b_s, 2, wm, wn, lane, fa0, fb0);
      mfma_slice<BM, BN>(fa1, fb1, acc, acc2);
      load_step(ks_load < last ? ks_load : last, ra4, rb4, okmask);
      frag_load<BM, BN>(a_s, b_s, 3, wm, wn, lane, fa1, fb1);
      mfma_slice<BM, BN>(fa0, fb0, acc, acc2);
      mfma_slice<BM, BN>(fa1, fb1, acc, acc2);
      // wanted issue order
      __builtin_amdgcn_sched_group_barrier(0x100, 2 * NRD, 0);            // slices 0 and 1
      sched_mix<NMF, NST, 0x200>();                                       // slice 0 + LDS stores of the next step
      sched_mix<NMF, NRD, 0x100>();                                       // slice 1 + fragments of slice 2
      sched_mix<NMF, NST, 0x020>();                                       // slice 2 + global loads, then fragments of 3
      __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, NMF, 0);                // slice 3
    };
    if (nk > 0) {
      load_step(0, ra4_0, rb4_0, okmask_0);
      load_step(1 < last ? 1 : last, ra4_1, rb4_1, okmask_1);
      store_step(0, ra4_0, rb4_0, okmask_0);
      load_step(2 < last ? 2 : last, ra4_0, rb4_0, okmask_0);
      int ks = 0;
      for (; ks + 1 < nk; ks += 2) {
        pstep(0, ra4_1, rb4_1, okmask_1, ks + 3);
        pstep(1, ra4_0, rb4_0, okmask_0, ks + 4);
      }
      if (ks < nk) {
        __syncthreads();
        mfma_tile<BM, BN>(As, Bs, acc, acc2, wm, wn, lane);
      }
    }
  } else {
  if (nk > 0) load_step(0, ra4_0, rb4_0, okmask_0);
  if (nk > 1) load_step(1, ra4_1, rb4_1, okmask_1);
  // Two K-steps per trip with NO condition between them (an `if (ks + 1 < nk)` in the body makes hipcc carry the
  // accumulators through VGPR phis: 64 v_accvgpr_read + 64 v_accvgpr_write per trip); an odd last step is peeled.
  int ks = 0;
  for (; ks + 1 < nk; ks += 2) {
    XAS_KSTEP(ks, 0, ra4_0, rb4_0, okmask_0)
    XAS_KSTEP(ks + 1, 1, ra4_1, rb4_1, okmask_1)
  }
  if (ks < nk) XAS_KSTEP(ks, 0, ra4_0, rb4_0, okmask_0)
  }   // !PIPE
#undef XAS_KSTEP

  igemm_epilogue<BM, BN, MODE>(p, acc, acc2, m0, n0, wm, wn, lane, Mrows, HW, Wrow, ph, pw, lds);
}

// ------------------------------------------------------------------------------------
// fwd / dgrad with BUFFER loads (the shipped path; igemm_kernel above stays as the fallback for tensors of 2 GiB
// and more, whose byte offsets do not fit the scheme).
//
// In-kernel stamps (tools/stamp_conv.py) showed where the K-loop of igemm_kernel loses its time: not in the MFMAs
// and not in memory latency, but in ISSUING the ~115 vector-ALU instructions per K-step that form 64-bit addresses,
// clamp them and mask the staged registers - a wave that competes with the MFMA stream of the other wave on its
// SIMD gets about one VALU issue slot per MFMA (64 cycles).  Here the address of every load is
//     buffer base (SGPRs)  +  per-lane byte offset, FIXED for the whole tile (1 VGPR per staged row)
//                          +  scalar offset of the (tap, channel chunk) of the K-step (SGPR, scalar ALU only)
// and padding / ragged edges are handled by the buffer unit itself: a lane whose tap falls outside the image gets
// the out-of-range offset 0x80000000 and the hardware returns zeros, so nothing is masked on the way to LDS.
// Per K-step a wave issues 3 VALU per staged activation row (tap-validity bit -> offset select) and none for the
// weights, against ~115 before.
template <int BM, int BN, int MODE, bool PIPE, bool BNB = false>
__global__ __launch_bounds__(256, 2) void igemm_buf_kernel(IgemmParams p) {
  using C = TileCfg<BM, BN>;
  constexpr int APASS = BM / 32, BPASS = BN / 32;
  extern __shared__ __align__(16) float lds[];
  float* As = lds;                       // [2][BM][LDK]
  float* Bs = lds + 2 * BM * LDK;        // [2][BN][LDK]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / C::WAVES_N, wn = wave % C::WAVES_N;
  const int kq = tid & 7, lrow = tid >> 3;

  int Hrow = p.Hrow, Wrow = p.Wrow;
  int ph = 0, pw = 0, base_r = 0, base_s = 0, nr = p.R, ns = p.S, off_h = -p.pad, off_w = -p.pad, rstep = 1;
  int sa = p.stride;
  if (MODE == 1) {
    const int st = p.stride;
    ph = blockIdx.z / st; pw = blockIdx.z % st;
    Hrow = (p.Hd - ph + st - 1) / st; Wrow = (p.Wd - pw + st - 1) / st;
    base_r = (ph + p.pad) % st; base_s = (pw + p.pad) % st;
    nr = base_r < p.R ? (p.R - base_r + st - 1) / st : 0;
    ns = base_s < p.S ? (p.S - base_s + st - 1) / st : 0;
    off_h = (ph + p.pad - base_r) / st; off_w = (pw + p.pad - base_s) / st;
    rstep = st; sa = 1;
  }
  const int Mrows = p.N * Hrow * Wrow;
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;      // XCD-aware tile order, as igemm_kernel
  const int mt = xcd * p.mt_per_xcd + q / p.nNt;
  const int nt = q - (q / p.nNt) * p.nNt;
  if (mt >= p.nMt) return;
  const int m0 = mt * BM, n0 = nt * BN;
  if (m0 >= Mrows) return;
  const int HW = Hrow * Wrow;
  const int cchunks = p.Cs / BK;
  const int nk = nr * ns * cchunks;

  // Offsets.  true element offset of (row j, tap, channel) = rbase[j] + delta(tap) + c, where
  //   fwd  : delta = (jr*Ws + js)*Cs >= 0,           rbase >= -(pad*Ws + pad)*Cs
  //   dgrad: delta = -(jr*Ws + js)*Cs <= 0,          rbase >= 0
  // The buffer base is moved down by `bias` elements so that the per-lane part (rbase + min delta + bias) and the
  // scalar part (delta - min delta + chunk) are both non-negative 32-bit byte offsets.
  const long dmin = MODE == 0 ? 0l : -((long)(nr - 1) * p.Ws + (ns - 1)) * p.Cs;
  const long rmin = MODE == 0 ? -((long)p.pad * p.Ws + p.pad) * p.Cs : 0l;
  const long bias = -(rmin + dmin);
  const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.src) - bias, 0, (int)((bias + p.src_elems) * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcB =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wgt), 0, (int)(p.wgt_elems * 4), 0x00020000);

  unsigned voffA[APASS], maskA[APASS], voffB[BPASS];
#pragma unroll
  for (int j = 0; j < APASS; ++j) {
    const int m = m0 + lrow + 32 * j;
    voffA[j] = kOOB; maskA[j] = 0u;
    if (m < Mrows) {
      int n, a, b;
      if (MODE == 0) { n = p.div_hw.div(m); const int rem = m - n * HW; a = p.div_w.div(rem); b = rem - a * Wrow; }
      else { n = m / HW; const int rem = m - n * HW; a = rem / Wrow; b = rem - a * Wrow; }
      const int ra = a * sa + off_h, rb = b * sa + off_w;
      const long rbase = (((long)n * p.Hs + ra) * p.Ws + rb) * p.Cs;
      voffA[j] = (unsigned)((rbase + dmin + bias + kq * 4) * 4);
      unsigned colmask = 0u, msk = 0u;                 // bit (jr*ns + js) = tap inside the image
      for (int js = 0; js < ns; ++js) {
        const int ws = rb + (MODE == 0 ? js : -js);
        colmask |= ((unsigned)ws < (unsigned)p.Ws ? 1u : 0u) << js;
      }
      for (int jr = 0; jr < nr; ++jr) {
        const int hs = ra + (MODE == 0 ? jr : -jr);
        if ((unsigned)hs < (unsigned)p.Hs) msk |= colmask << (jr * ns);
      }
      maskA[j] = msk;
    }
  }
  const unsigned wrow_bytes = (unsigned)(p.R * p.S * p.Cs) * 4u;
#pragma unroll
  for (int j = 0; j < BPASS; ++j) {
    const int n = n0 + lrow + 32 * j;
    voffB[j] = n < p.Cd ? (unsigned)n * wrow_bytes + (unsigned)kq * 16u : kOOB;
  }

  f32x16 acc[C::MI][C::NI];
#pragma unroll
  for (int mi = 0; mi < C::MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < C::NI; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;
  f32x16 acc2;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc2[e] = 0.f;

  // Load stream: K-steps are visited in order; (chunk, js, jr) advance with scalar ALU only and stop at the last
  // step (further calls re-load it: valid addresses, values never consumed).
  int ld_chunk = 0, ld_js = 0, ld_jr = 0, ld_left = nk;
  float4 ra4_0[APASS], rb4_0[BPASS], ra4_1[APASS], rb4_1[BPASS];
  auto load_next = [&](float4 (&ra4)[APASS], float4 (&rb4)[BPASS]) {
    const int tap = ld_jr * ns + ld_js;
    const int rel = MODE == 0 ? (ld_jr * p.Ws + ld_js) : ((nr - 1 - ld_jr) * p.Ws + (ns - 1 - ld_js));
    const unsigned soffA = (unsigned)(rel * p.Cs + ld_chunk * BK) * 4u;
    const int wtap = (base_r + rstep * ld_jr) * p.S + (base_s + rstep * ld_js);
    const unsigned soffB = (unsigned)(wtap * p.Cs + ld_chunk * BK) * 4u;
#pragma unroll
    for (int j = 0; j < APASS; ++j) {
      const unsigned off = ((maskA[j] >> tap) & 1u) ? voffA[j] : kOOB;
      ra4[j] = buf_load16(rsrcA, off, soffA);
    }
#pragma unroll
    for (int j = 0; j < BPASS; ++j) rb4[j] = buf_load16(rsrcB, voffB[j], soffB);
    const bool more = ld_left > 1;
    ld_left -= more ? 1 : 0;
    int c = ld_chunk + 1, s = ld_js, r = ld_jr;
    if (c == cchunks) { c = 0; ++s; }
    if (s == ns) { s = 0; ++r; }
    ld_chunk = more ? c : ld_chunk; ld_js = more ? s : ld_js; ld_jr = more ? r : ld_jr;
  };
  auto store_step = [&](int buf, const float4 (&ra4)[APASS], const float4 (&rb4)[BPASS]) {
    float* a = As + buf * BM * LDK;
    float* b = Bs + buf * BN * LDK;
#pragma unroll
    for (int j = 0; j < APASS; ++j) *reinterpret_cast<float4*>(a + (lrow + 32 * j) * LDK + kq * 4) = ra4[j];
#pragma unroll
    for (int j = 0; j < BPASS; ++j) *reinterpret_cast<float4*>(b + (lrow + 32 * j) * LDK + kq * 4) = rb4[j];
  };

  if (nk > 0) {
    if constexpr (PIPE) {
      // LDS stores of step ks+1 and loads of step ks+3 are issued inside the MFMA sequence of step ks.
      constexpr int NMF = C::MI * C::NI * 4, NRD = C::MI + C::NI, NST = APASS + BPASS;
      auto pstep = [&](int buf, float4 (&ra4)[APASS], float4 (&rb4)[BPASS]) {
        const float* a_s = As + buf * BM * LDK;
        const float* b_s = Bs + buf * BN * LDK;
        float4 fa0[C::MI], fb0[C::NI], fa1[C::MI], fb1[C::NI];
        __syncthreads();
        frag_load<BM, BN>(a_s, b_s, 0, wm, wn, lane, fa0, fb0);
        frag_load<BM, BN>(a_s, b_s, 1, wm, wn, lane, fa1, fb1);
        mfma_slice<BM, BN>(fa0, fb0, acc, acc2);
        store_step(buf ^ 1, ra4, rb4);
        frag_load<BM, BN>(a_s, b_s, 2, wm, wn, lane, fa0, fb0);
        mfma_slice<BM, BN>(fa1, fb1, acc, acc2);
        load_next(ra4, rb4);
        frag_load<BM, BN>(a_s, b_s, 3, wm, wn, lane, fa1, fb1);
        mfma_slice<BM, BN>(fa0, fb0, acc, acc2);
        mfma_slice<BM, BN>(fa1, fb1, acc, acc2);
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * NRD, 0);
        sched_mix<NMF, NST, 0x200>();
        sched_mix<NMF, NRD, 0x100>();
        sched_mix<NMF, NST, 0x020>();
        __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, NMF, 0);
      };
      load_next(ra4_0, rb4_0);                       // step 0
      load_next(ra4_1, rb4_1);                       // step 1
      store_step(0, ra4_0, rb4_0);
      load_next(ra4_0, rb4_0);                       // step 2
      int ks = 0;
      for (; ks + 1 < nk; ks += 2) {
        pstep(0, ra4_1, rb4_1);                      // computes ks, stores ks+1, loads ks+3
        pstep(1, ra4_0, rb4_0);                      // computes ks+1, stores ks+2, loads ks+4
      }
      if (ks < nk) {
        __syncthreads();
        mfma_tile<BM, BN>(As, Bs, acc, acc2, wm, wn, lane);
      }
    } else {
      // store(ks) -> barrier -> issue loads(ks+2) -> MFMAs(ks); two K-steps per trip, odd last step peeled
      load_next(ra4_0, rb4_0);
      load_next(ra4_1, rb4_1);
      int ks = 0;
      for (; ks + 1 < nk; ks += 2) {
        store_step(0, ra4_0, rb4_0);
        __syncthreads();
        load_next(ra4_0, rb4_0);
        mfma_tile<BM, BN>(As, Bs, acc, acc2, wm, wn, lane);
        store_step(1, ra4_1, rb4_1);
        __syncthreads();
        load_next(ra4_1, rb4_1);
        mfma_tile<BM, BN>(As + BM * LDK, Bs + BN * LDK, acc, acc2, wm, wn, lane);
      }
      if (ks < nk) {
        store_step(0, ra4_0, rb4_0);
        __syncthreads();
        mfma_tile<BM, BN>(As, Bs, acc, acc2, wm, wn, lane);
      }
    }
  }
  igemm_epilogue<BM, BN, MODE, BNB>(p, acc, acc2, m0, n0, wm, wn, lane, Mrows, HW, Wrow, ph, pw, lds);
}

// ------------------------------------------------------------------------------------
// wgrad: C[co][nn] = sum_m dY[m][co] * Xcol[m][nn],  nn = (r*S+s)*Cin + c
// LDS tiles are [k = pixel][row] (row-contiguous, as they come from memory); MFMA operands
// are read with ds_read_b32 (lanes walk rows: conflict free).
// ------------------------------------------------------------------------------------
template <int BM, int BN, bool VEC>
__global__ __launch_bounds__(256) void wgrad_kernel(WgradParams p) {
  using C = TileCfg<BM, BN>;
  constexpr int LDA = BM + 4, LDB = BN + 4;          // +4 keeps float4 stores aligned
  constexpr int AQ = BM / 4, BQ = BN / 4;            // float4 per pixel row
  constexpr int AROWS = 256 / AQ, BROWS = 256 / BQ;  // pixel rows per pass
  constexpr int APASS = WBK / AROWS, BPASS = WBK / BROWS;
  extern __shared__ __align__(16) float lds[];
  float* As = lds;                                   // [2][WBK][LDA]
  float* Bs = lds + 2 * WBK * LDA;                   // [2][WBK][LDB]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / C::WAVES_N, wn = wave % C::WAVES_N;
  // Block order.  Measured both ways (tools/ab_step.py, PMC): grouping all tiles of one pixel split on ONE XCD
  // cuts the kernel's HBM fetch 2-7x, yet the step is 3 % SLOWER (the tiles of a split then hammer the same L2
  // lines at the same time); the plain order below is the shipped one.
  int split, tile;
  if (!(p.tune & 8192)) {                    // default: tiles fastest, dealt round-robin over the XCDs
    tile = blockIdx.x % p.ntiles; split = blockIdx.x / p.ntiles;
  } else {                                   // experiment (bit13): one split per XCD - 2-7x less HBM traffic (PMC) but slower
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    split = xcd + 8 * (q / p.ntiles);
    tile = q - (q / p.ntiles) * p.ntiles;
  }
  if (split >= p.nsplits) return;
  const int co0 = (tile % p.nct) * BM, nn0 = (tile / p.nct) * BN;
  const int mbeg = split * p.m_per_split, mend = min(p.M, mbeg + p.m_per_split);
  const int HWo = p.Ho * p.Wo;

  const int aq = tid % AQ, arow = tid / AQ;
  const int bq = tid % BQ, brow = tid / BQ;
  // column geometry of this thread's B float4 (VEC) : fixed tap and channel
  const int nnb = nn0 + bq * 4;
  int tap_r = 0, tap_s = 0, tap_c = 0;
  bool bcol_ok = nnb < p.KK;
  if (VEC && bcol_ok) {
    const int tap = p.div_cin.div(nnb); tap_c = nnb - tap * p.Cin;
    tap_r = p.div_s.div(tap); tap_s = tap - tap_r * p.S;
  }

  f32x16 acc[C::MI][C::NI];
#pragma unroll
  for (int mi = 0; mi < C::MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < C::NI; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  float4 ra4[APASS], rb4[BPASS];
  unsigned okmask = 0;
  auto load_step = [&](int mk) {
    unsigned msk = 0;
#pragma unroll
    for (int j = 0; j < APASS; ++j) {
      const int m = mk + arow + AROWS * j, co = co0 + aq * 4;
      const bool ok = m < mend && co < p.Cout;
      ra4[j] = *reinterpret_cast<const float4*>(p.dy + (ok ? (size_t)m * p.Cout + co : (size_t)0));
      msk |= (ok ? 1u : 0u) << j;
    }
#pragma unroll
    for (int j = 0; j < BPASS; ++j) {
      const int m = mk + brow + BROWS * j;
      if (VEC) {
        const int mm = m < mend ? m : mbeg;
        const int n = p.div_hw.div(mm); const int rem = mm - n * HWo;
        const int ho = p.div_w.div(rem), wo = rem - ho * p.Wo;
        const int hi = ho * p.stride - p.pad + tap_r, wi = wo * p.stride - p.pad + tap_s;
        const bool ok = m < mend && bcol_ok && (unsigned)hi < (unsigned)p.Hi && (unsigned)wi < (unsigned)p.Wi;
        rb4[j] = *reinterpret_cast<const float4*>(p.x + (ok ? (((size_t)n * p.Hi + hi) * p.Wi + wi) * p.Cin + tap_c : (size_t)0));
        msk |= (ok ? 1u : 0u) << (16 + j);
      } else {
        float t[4] = {0.f, 0.f, 0.f, 0.f};
        if (m < mend && bcol_ok) {
          const int n = p.div_hw.div(m); const int rem = m - n * HWo;
          const int ho = p.div_w.div(rem), wo = rem - ho * p.Wo;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int nn = nnb + e;
            if (nn < p.KK) {
              const int tap = p.div_cin.div(nn); const int c = nn - tap * p.Cin;
              const int r = p.div_s.div(tap), s = tap - r * p.S;
              const int hi = ho * p.stride - p.pad + r, wi = wo * p.stride - p.pad + s;
              if ((unsigned)hi < (unsigned)p.Hi && (unsigned)wi < (unsigned)p.Wi)
                t[e] = p.x[(((size_t)n * p.Hi + hi) * p.Wi + wi) * p.Cin + c];
            }
          }
        }
        rb4[j] = make_float4(t[0], t[1], t[2], t[3]);
        msk |= 1u << (16 + j);
      }
    }
    okmask = msk;
  };
  auto store_step = [&](int buf) {
    float* a = As + buf * WBK * LDA;
    float* b = Bs + buf * WBK * LDB;
#pragma unroll
    for (int j = 0; j < APASS; ++j) {
      const bool ok = (okmask >> j) & 1u;
      float4 v = ra4[j];
      v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
      *reinterpret_cast<float4*>(a + (arow + AROWS * j) * LDA + aq * 4) = v;
    }
#pragma unroll
    for (int j = 0; j < BPASS; ++j) {
      const bool ok = (okmask >> (16 + j)) & 1u;
      float4 v = rb4[j];
      v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
      *reinterpret_cast<float4*>(b + (brow + BROWS * j) * LDB + bq * 4) = v;
    }
  };

  const int i = lane & 31, h = lane >> 5;
  const int nsteps = (mend > mbeg) ? (mend - mbeg + WBK - 1) / WBK : 0;
  if (nsteps > 0) load_step(mbeg);
  for (int st = 0; st < nsteps; ++st) {
    const int buf = st & 1;
    store_step(buf);
    __syncthreads();
    if (st + 1 < nsteps) load_step(mbeg + (st + 1) * WBK);
    const float* a = As + buf * WBK * LDA + wm * C::WM + i;
    const float* b = Bs + buf * WBK * LDB + wn * C::WN + i;
#pragma unroll
    for (int kk = 0; kk < WBK / 2; ++kk) {
      float av[C::MI], bv[C::NI];
#pragma unroll
      for (int mi = 0; mi < C::MI; ++mi) av[mi] = a[(2 * kk + h) * LDA + mi * 32];
#pragma unroll
      for (int ni = 0; ni < C::NI; ++ni) bv[ni] = b[(2 * kk + h) * LDB + ni * 32];
#pragma unroll
      for (int mi = 0; mi < C::MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < C::NI; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi], bv[ni], acc[mi][ni], 0, 0, 0);
    }
  }
  float* slab = p.out + (size_t)split * p.Cout * p.KK;
  const int col_l = lane & 31, rsub = 4 * (lane >> 5);
#pragma unroll
  for (int mi = 0; mi < C::MI; ++mi)
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int co = co0 + wm * C::WM + mi * 32 + (reg & 3) + 8 * (reg >> 2) + rsub;
      if (co >= p.Cout) continue;
#pragma unroll
      for (int ni = 0; ni < C::NI; ++ni) {
        const int nn = nn0 + wn * C::WN + ni * 32 + col_l;
        if (nn < p.KK) slab[(size_t)co * p.KK + nn] = acc[mi][ni][reg];
      }
    }
}

// ------------------------------------------------------------------------------------
// wgrad with BUFFER loads (shipped path for Cin % 4 == 0; wgrad_kernel above is the fallback).
// Same tile / split / slab scheme, but the per-K-step vector-ALU work is cut from ~110 to ~35 instructions per wave
// (see igemm_buf_kernel for why that is what bounds these kernels):
//   dy tile : per-lane offset fixed for the whole tile, the pixel rows of a K-step come from a scalar offset,
//             rows past the split end are cut off by the buffer range -> no VALU at all;
//   x tile  : a thread stages ONE pixel per K-step (8 threads x BN/32 float4 each per pixel) so the
//             pixel -> (n, ho, wo) decode is done once per K-step, from a scalar decode of the step's first pixel
//             plus a per-thread constant; each float4 adds a precomputed (tap, channel) offset and is sent to the
//             out-of-range offset when its tap falls outside the image (hardware returns zeros).
// ------------------------------------------------------------------------------------
template <int BM, int BN, int T, bool PIPE>
__global__ __launch_bounds__(256) void wgrad_buf_kernel(WgradParams p) {
  using C = TileCfg<BM, BN>;
  constexpr int LDA = BM + 4, LDB = BN + 32;         // LDB = 32 mod 64: the two 32-float rows of a b128 store group hit disjoint banks
  constexpr int AQ = BM / 4, AROWS = 256 / AQ, APASS = WBK / AROWS;
  constexpr int BPASS = BN / 32;                     // float4 per thread per K-step of the x tile
  constexpr int GQ = BPASS / T;                      // float4 per tap group
  static_assert(T >= 1 && T <= BPASS && BPASS % T == 0, "bad tap grouping");
  extern __shared__ __align__(16) float lds[];
  float* As = lds;                                   // [2][WBK][LDA]
  float* Bs = lds + 2 * WBK * LDA;                   // [2][WBK][LDB]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / C::WAVES_N, wn = wave % C::WAVES_N;
  int tile, split;
  if (p.tune & 8192) {                       // experiment (bit13): tiles fastest, dealt round-robin over the XCDs
    tile = blockIdx.x % p.ntiles; split = blockIdx.x / p.ntiles;
  } else {
    // XCD-grouped order.  A group = the KK-tiles of one (pixel split, Cout tile): they read the SAME dy tile and the
    // same x pixels (the taps of a 3x3 filter shift by one pixel).  Workgroups are dealt round-robin over the 8 XCDs,
    // so group g runs on XCD g % 8: its operands are fetched into ONE L2 (round 1 dealt the tiles of a split over
    // all eight L2s: 2.04x the algorithmic HBM bytes).  Placement only affects speed.
    const int nkt = p.ntiles / p.nct;
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int grp = xcd + 8 * (q / nkt);                      // group index = split * nct + co-tile
    const int kt = q - (q / nkt) * nkt;
    split = grp / p.nct;
    tile = (grp - split * p.nct) + kt * p.nct;                // tile = co-tile + nct * kk-tile
  }
  if (split >= p.nsplits) return;
  const int co0 = (tile % p.nct) * BM, nn0 = (tile / p.nct) * BN;
  const int mbeg = split * p.m_per_split, mend = min(p.M, mbeg + p.m_per_split);
  const int HWo = p.Ho * p.Wo;

  // ---- dy operand: per-lane offset fixed, K-step rows from a scalar offset, split end = buffer range
  const int aq = tid % AQ, arow = tid / AQ;
  const __amdgpu_buffer_rsrc_t rsrcA =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, (int)((long)mend * p.Cout * 4), 0x00020000);
  const unsigned voffA = (co0 + aq * 4 < p.Cout) ? (unsigned)(arow * p.Cout + co0 + aq * 4) * 4u : kOOB;
  const unsigned passA = (unsigned)(AROWS * p.Cout) * 4u;

  // ---- x operand: thread = (pixel brow of the K-step, float4 slot bq8); the BN columns of the tile are T groups
  // of BN/T columns, each inside ONE filter tap (host guarantees Cin % (BN/T) == 0)
  const int brow = tid >> 3, bq8 = tid & 7;
  const long biasB = ((long)p.pad * p.Wi + p.pad) * p.Cin;
  const __amdgpu_buffer_rsrc_t rsrcB = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.x) - biasB, 0, (int)((biasB + (long)p.N * p.Hi * p.Wi * p.Cin) * 4), 0x00020000);
  int tr[T], ts[T];
  unsigned offG[T];                                  // ((tr*Wi + ts)*Cin + c)*4 + bias bytes of the group's first float4 of this thread
#pragma unroll
  for (int g = 0; g < T; ++g) {
    const int nn = nn0 + g * (BN / T) + bq8 * 4;
    const int nc = nn < p.KK ? nn : 0;               // columns past KK are never stored; any in-range address will do
    const int tap = p.div_cin.div(nc), c = nc - tap * p.Cin;
    tr[g] = p.div_s.div(tap); ts[g] = tap - tr[g] * p.S;
    offG[g] = (unsigned)(((tr[g] * p.Wi + ts[g]) * p.Cin + c) * 4 + biasB * 4);
  }
  const int dh_r = brow / p.Wo, dw_r = brow - dh_r * p.Wo;     // this thread's pixel relative to the step's first pixel

  f32x16 acc[C::MI][C::NI];
#pragma unroll
  for (int mi = 0; mi < C::MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < C::NI; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  auto load_step = [&](int mk, float4 (&ra4)[APASS], float4 (&rb4)[BPASS]) {   // mk: first pixel of the K-step (uniform)
    const unsigned soffA = (unsigned)mk * (unsigned)p.Cout * 4u;
#pragma unroll
    for (int j = 0; j < APASS; ++j) ra4[j] = buf_load16(rsrcA, voffA, soffA + j * passA);
    // scalar decode of the step's first pixel, then this thread's pixel with at most one carry per coordinate
    const int n0 = p.div_hw.div(mk), rem = mk - n0 * HWo;
    const int h0 = p.div_w.div(rem), w0 = rem - h0 * p.Wo;
    int w = w0 + dw_r;
    const int c1 = w >= p.Wo ? 1 : 0;
    w -= c1 ? p.Wo : 0;
    int h = h0 + dh_r + c1;
    const int c2 = h >= p.Ho ? 1 : 0;
    h -= c2 ? p.Ho : 0;
    const int n = n0 + c2;
    const int hb = h * p.stride - p.pad, wb = w * p.stride - p.pad;
    const unsigned pix = (unsigned)(((n * p.Hi + hb) * p.Wi + wb) * p.Cin) * 4u;     // + bias stays >= 0 for valid taps
#pragma unroll
    for (int g = 0; g < T; ++g) {
      const bool ok = (unsigned)(hb + tr[g]) < (unsigned)p.Hi && (unsigned)(wb + ts[g]) < (unsigned)p.Wi;
      const unsigned off = ok ? pix + offG[g] : kOOB;
#pragma unroll
      for (int q = 0; q < GQ; ++q) rb4[g * GQ + q] = buf_load16(rsrcB, off, (unsigned)(q * 128));
    }
  };
  auto store_step = [&](int buf, const float4 (&ra4)[APASS], const float4 (&rb4)[BPASS]) {
    float* a = As + buf * WBK * LDA;
    float* b = Bs + buf * WBK * LDB;
#pragma unroll
    for (int j = 0; j < APASS; ++j) *reinterpret_cast<float4*>(a + (arow + AROWS * j) * LDA + aq * 4) = ra4[j];
#pragma unroll
    for (int j = 0; j < BPASS; ++j)
      *reinterpret_cast<float4*>(b + brow * LDB + (j / GQ) * (BN / T) + (bq8 + 8 * (j % GQ)) * 4) = rb4[j];
  };

  // MFMA operands: a wave with two 32-row tiles (MI == 2) takes tile rows 2i and 2i+1 for lane i, so both come from
  // ONE 8-byte LDS read at a compile-time offset; same for the columns.
  const int i = lane & 31, h = lane >> 5;
  const int nsteps = (mend > mbeg) ? (mend - mbeg + WBK - 1) / WBK : 0;
  // Even and odd k-pairs are addressed from two bases whose distance the compiler cannot see: otherwise it merges
  // neighbouring fragment reads into ds_read2 (8-bit offsets) and then needs a v_add per pair to move the base,
  // 16-22 VALU instructions per K-step; this way every read is one ds_read with a 16-bit immediate offset.
  int odd_a = 2 * LDA, odd_b = 2 * LDB;
  asm volatile("" : "+v"(odd_a), "+v"(odd_b));
  auto mfma_step = [&](int buf) {
    const float* a = As + buf * WBK * LDA + h * LDA + wm * C::WM + C::MI * i;
    const float* b = Bs + buf * WBK * LDB + h * LDB + wn * C::WN + C::NI * i;
    const float* a2 = a + odd_a;
    const float* b2 = b + odd_b;
#pragma unroll
    for (int kk = 0; kk < WBK / 2; ++kk) {
      float av[C::MI], bv[C::NI];
      const float* ap = (kk & 1) ? a2 + (kk - 1) * 2 * LDA : a + kk * 2 * LDA;
      const float* bp = (kk & 1) ? b2 + (kk - 1) * 2 * LDB : b + kk * 2 * LDB;
      if constexpr (C::MI == 2) {
        const float2 t = *reinterpret_cast<const float2*>(ap);
        av[0] = t.x; av[1] = t.y;
      } else {
        av[0] = ap[0];
      }
      if constexpr (C::NI == 2) {
        const float2 t = *reinterpret_cast<const float2*>(bp);
        bv[0] = t.x; bv[1] = t.y;
      } else {
        bv[0] = bp[0];
      }
#pragma unroll
      for (int mi = 0; mi < C::MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < C::NI; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi], bv[ni], acc[mi][ni], 0, 0, 0);
    }
  };
  float4 ra4_0[APASS], rb4_0[BPASS];
  if constexpr (PIPE) {
    // Two register sets, loads three K-steps ahead; the LDS stores of step st+1 and the loads of step st+3 are issued
    // inside the MFMA sequence of step st (see igemm_buf_kernel).  Steps past the end re-load the last one.
    float4 ra4_1[APASS], rb4_1[BPASS];
    const int last = nsteps - 1;
    auto mk_of = [&](int st) { return mbeg + (st < last ? st : last) * WBK; };
    auto pstep = [&](int buf, float4 (&ra4)[APASS], float4 (&rb4)[BPASS], int st_load) {
      __syncthreads();
      mfma_step(buf);
      store_step(buf ^ 1, ra4, rb4);
      load_step(mk_of(st_load), ra4, rb4);
      constexpr int NMF = C::MI * C::NI, NST = APASS + BPASS;
#pragma unroll
      for (int kk = 0; kk < WBK / 2; ++kk) {
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, NMF, 0);
        if (kk < NST) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        else if (kk - NST < NST) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
    };
    if (nsteps > 0) {
      load_step(mk_of(0), ra4_0, rb4_0);
      load_step(mk_of(1), ra4_1, rb4_1);
      store_step(0, ra4_0, rb4_0);
      load_step(mk_of(2), ra4_0, rb4_0);
      int st = 0;
      for (; st + 1 < nsteps; st += 2) {
        pstep(0, ra4_1, rb4_1, st + 3);
        pstep(1, ra4_0, rb4_0, st + 4);
      }
      if (st < nsteps) {
        __syncthreads();
        mfma_step(0);
      }
    }
  } else {
    if (nsteps > 0) load_step(mbeg, ra4_0, rb4_0);
    for (int st = 0; st < nsteps; ++st) {
      const int buf = st & 1;
      store_step(buf, ra4_0, rb4_0);
      __syncthreads();
      if (st + 1 < nsteps) load_step(mbeg + (st + 1) * WBK, ra4_0, rb4_0);
      mfma_step(buf);
    }
  }
  static_assert(C::MI <= 2 && C::NI <= 2, "interleaved fragment scheme handles at most two tiles per wave and dimension");
  float* slab = p.out + (size_t)split * p.Cout * p.KK;
  const int col_l = lane & 31, rsub = 4 * (lane >> 5);
#pragma unroll
  for (int mi = 0; mi < C::MI; ++mi)
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int co = co0 + wm * C::WM + C::MI * ((reg & 3) + 8 * (reg >> 2) + rsub) + mi;
      if (co >= p.Cout) continue;
      const int nn = nn0 + wn * C::WN + C::NI * col_l;
      float* o = slab + (size_t)co * p.KK + nn;
      if constexpr (C::NI == 2) {
        if (nn + 1 < p.KK) *reinterpret_cast<float2*>(o) = make_float2(acc[mi][0][reg], acc[mi][1][reg]);
        else if (nn < p.KK) o[0] = acc[mi][0][reg];
      } else {
        if (nn < p.KK) o[0] = acc[mi][0][reg];
      }
    }
}

// Sum the split slabs (fixed order -> deterministic).  Block = 64 elements x (blockDim.x / 64) slab lanes, so a
// reduction over hundreds of slabs is several-way parallel instead of one long dependent chain per element.
// kSlabThreads = 256 by default (4 lanes): 1024-thread blocks are hard to place while other kernels fill the CUs.
// unpack != 0: element i is a packed index [co][r][s][ci] and is written to OIHW [co][ci][r][s].
__global__ __launch_bounds__(1024) void slab_reduce_kernel2(const float* __restrict__ slabs, int splits, long n,
                                                            int unpack, int Cin, int R, int S,
                                                            float* __restrict__ out) {
  __shared__ float red[16][64];
  const int ex = threadIdx.x & 63, ly = threadIdx.x >> 6, nl = blockDim.x >> 6;
  const long i = (long)blockIdx.x * 64 + ex;
  float acc = 0.f;
  if (i < n) {
    int k = ly;
    for (; k + 7 * nl < splits; k += 8 * nl) {         // 8 independent loads in flight per lane, summed in slab order
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = slabs[(size_t)(k + u * nl) * n + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; k < splits; k += nl) acc += slabs[(size_t)k * n + i];
  }
  red[ly][ex] = acc;
  __syncthreads();
  if (ly != 0 || i >= n) return;
  for (int k = 1; k < nl; ++k) acc += red[k][ex];
  long o = i;
  if (unpack & 1) {
    long t = i;
    const int ci = t % Cin; t /= Cin;
    const int q = t % S; t /= S;
    const int r = t % R; const long co = t / R;
    o = ((co * Cin + ci) * R + r) * S + q;
  }
  if (unpack & 2) out[o] += acc; else out[o] = acc;
}

// ------------------------------------------------------------------------------------
// Stem: 7x7 stride-2 pad-3 convolution of the 3-channel image (resnet.py:16), K = 147.
// A workgroup owns an 8 x 16 output patch of one image for all 64 output channels: the
// 21 x 37 x 3 input patch and the whole [148][64] weight matrix live in LDS, and each MFMA
// operand is one ds_read_b32 at (lane base + compile-time tap offset).
// ------------------------------------------------------------------------------------
constexpr int ST_TH = 8, ST_TW = 16;                 // output tile
constexpr int ST_PH = (ST_TH - 1) * 2 + 7;           // 21
constexpr int ST_PW = (ST_TW - 1) * 2 + 7;           // 37
constexpr int ST_K = 147, ST_KP = 148, ST_CO = 64, ST_LDW = ST_CO + 1;

#ifndef XAS_ST_TPB
#define XAS_ST_TPB 4
#endif
constexpr int ST_TPB = XAS_ST_TPB;                   // output tiles a block walks with ONE copy of the weights in LDS

__global__ __launch_bounds__(256) void stem_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       float* __restrict__ y, int N, int H, int W, int Ho, int Wo) {
  __shared__ float patch[ST_PH * ST_PW * 3 + 8];
  __shared__ float ws[ST_KP * ST_LDW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_w = (Wo + ST_TW - 1) / ST_TW, tiles = tiles_w * ((Ho + ST_TH - 1) / ST_TH);
  const int n = blockIdx.y;
  // weights packed [co][r][s][c] = [co][k] -> ws[k][co]; row 147 is the zero pad.  Staged once per block: the 37.6 KB of
  // weights were re-read for every 8 x 16 tile (r03: a block walks ST_TPB tiles: 72.7 -> 77.8 TFLOP/s at 4 tiles, 76 at 8, 73 at 16)
  for (int e = tid; e < ST_CO * ST_K; e += 256) {
    const int co = e / ST_K, k = e % ST_K;
    ws[k * ST_LDW + co] = w[e];
  }
  if (tid < ST_CO) ws[ST_K * ST_LDW + tid] = 0.f;
  const int i = lane & 31, h = lane >> 5;
  const int ly = wave * 2 + (i >> 4), lx = i & 15;            // pixel of this lane inside the tile
  const int base = ((ly * 2) * ST_PW + lx * 2) * 3;
  for (int tt = 0; tt < ST_TPB; ++tt) {
    const int tile = blockIdx.x * ST_TPB + tt;
    if (tile >= tiles) break;
    const int ty = tile / tiles_w, tx = tile % tiles_w;
    const int oy0 = ty * ST_TH, ox0 = tx * ST_TW;
    const int iy0 = oy0 * 2 - 3, ix0 = ox0 * 2 - 3;
    __syncthreads();                                   // the previous tile's patch has been consumed (and ws is complete)
    // input patch [21][37][3], zero padded
    for (int e = tid; e < ST_PH * ST_PW * 3; e += 256) {
      const int c = e % 3, px = (e / 3) % ST_PW, py = e / (3 * ST_PW);
      const int iy = iy0 + py, ix = ix0 + px;
      patch[e] = ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
                     ? x[(((size_t)n * H + iy) * W + ix) * 3 + c] : 0.f;
    }
    __syncthreads();
    f32x16 acc0, acc1;
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
#pragma unroll
    for (int kk = 0; kk < ST_KP / 2; ++kk) {
      // tap offsets of k = 2kk and 2kk+1 in the patch (k = (r*7 + s)*3 + c); k = 147 reads tap 0 (weight is 0)
      const int k0 = 2 * kk, k1 = (2 * kk + 1 < ST_K) ? 2 * kk + 1 : 0;
      const int o0 = ((k0 / 21) * ST_PW + (k0 / 3) % 7) * 3 + k0 % 3;
      const int o1 = ((k1 / 21) * ST_PW + (k1 / 3) % 7) * 3 + k1 % 3;
      const float a = patch[base + (h ? o1 : o0)];
      const float b0 = ws[(2 * kk + h) * ST_LDW + i];
      const float b1 = ws[(2 * kk + h) * ST_LDW + 32 + i];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc1, 0, 0, 0);
    }
    const int col = lane & 31, rsub = 4 * (lane >> 5);
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int pi = (reg & 3) + 8 * (reg >> 2) + rsub;        // pixel index within the wave's 2 x 16 strip
      const int oy = oy0 + wave * 2 + (pi >> 4), ox = ox0 + (pi & 15);
      if (oy < Ho && ox < Wo) {
        float* o = y + (((size_t)n * Ho + oy) * Wo + ox) * ST_CO;
        o[col] = acc0[reg];
        o[32 + col] = acc1[reg];
      }
    }
  }
}

// ------------------------------------------------------------------------------------
// The same forward in the f16x3 arithmetic (XAS_PREC_F16X3; conv_x6.hip for the scheme): image patch and weights are split
// into two fp16 planes when they are staged (2^4 x, 2^10 w), the K axis is laid out as 7 filter rows x 24 (21 taps*channels
// of a row + 3 zero weights) so that the 8 consecutive k of an MFMA operand are 8 consecutive halfwords of ONE patch row:
// 11 K-steps of v_mfma_f32_32x32x16_f16 x 3 products instead of 74 K-steps of v_mfma_f32_32x32x2_f32.
// ------------------------------------------------------------------------------------
constexpr int STH_KR = 24, STH_K = 7 * STH_KR, STH_KS = 11;       // k per filter row, real k (168), K-steps of 16 (176)
constexpr int STH_WROW = STH_KS * 16 + 8;                          // halfwords per weight row (184: rows 368 B apart)
constexpr int STH_PROW = 120;                                      // halfwords per patch row (111 used)
#ifndef XAS_STH_TPB
#define XAS_STH_TPB 16
#endif
constexpr int STH_TPB = XAS_STH_TPB;                               // output tiles per block (one split of the weights)

__device__ unsigned g_f16_weight_overflow_stem = 0u;      // (the stem splits its 9 408 weights in the kernel: its own flag, read by
                                                          // xas_f16_weight_overflow together with the preparation kernels' one)
__global__ __launch_bounds__(256) void stem_fwd_f16_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           float* __restrict__ y, int N, int H, int W, int Ho, int Wo,
                                                           const float* __restrict__ x_amax) {
  __shared__ __align__(16) unsigned short wsh[2 * ST_CO * STH_WROW];        // [plane][co][k']
  float inv_sx;
  const float f16_sx = f16_grad_scale(x_amax, &inv_sx);                      // image scale from max |x| (wave-uniform)
  const float f16_desc = inv_sx * (1.f / kF16WScale);
  __shared__ __align__(16) unsigned short ph[2 * ST_PH * STH_PROW];         // [plane][patch row][px * 3 + c]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_w = (Wo + ST_TW - 1) / ST_TW, tiles = tiles_w * ((Ho + ST_TH - 1) / ST_TH);
  const int n = blockIdx.y;
  for (int e = tid; e < ST_CO * STH_WROW; e += 256) {
    const int co = e / STH_WROW, k = e - co * STH_WROW;
    const int r = k / STH_KR, j = k - r * STH_KR;
    float v = (k < STH_K && j < 21) ? w[co * ST_K + r * 21 + j] * kF16WScale : 0.f;
    if (!(fabsf(v) < 65504.f)) atomicOr(&g_f16_weight_overflow_stem, 1u);      // |w| >= 64 or NaN: flagged, like the pre-split weights
    const _Float16 h1 = (_Float16)v, h2 = (_Float16)(v - (float)h1);
    wsh[e] = __builtin_bit_cast(unsigned short, h1);
    wsh[ST_CO * STH_WROW + e] = __builtin_bit_cast(unsigned short, h2);
  }
  for (int e = tid; e < 2 * ST_PH * STH_PROW; e += 256) ph[e] = 0;           // (row tails stay zero: read against zero weights)
  const int i = lane & 31, hh = lane >> 5;
  const int ly = wave * 2 + (i >> 4), lx = i & 15;                           // pixel of this lane inside the tile
  unsigned aoff[STH_KS];                                                     // halfword offset of this lane's 8 k of every K-step
#pragma unroll
  for (int ks = 0; ks < STH_KS; ++ks) {
    const int kb = ks * 16 + hh * 8;
    const int r = kb < STH_K ? kb / STH_KR : 0, j = kb < STH_K ? kb - r * STH_KR : 0;
    aoff[ks] = (unsigned)((ly * 2 + r) * STH_PROW + lx * 6 + j);
  }
  // patch element e = tid + 256 j -> (patch row, column * 3 + channel): fixed per thread; the NEXT tile's values are fetched
  // into registers before the MFMAs of the current one (the loads of a tile were exposed between two barriers)
  constexpr int NPV = (ST_PH * ST_PW * 3 + 255) / 256;                       // 10
  int ppy[NPV], pq[NPV];
#pragma unroll
  for (int j = 0; j < NPV; ++j) {
    const int e = tid + 256 * j;
    ppy[j] = e < ST_PH * ST_PW * 3 ? e / (3 * ST_PW) : -1;
    pq[j] = e - (e / (3 * ST_PW)) * (3 * ST_PW);
  }
  float pv[NPV];
  auto load_patch = [&](int tile) {
    const int ty = tile / tiles_w, tx = tile - ty * tiles_w;
    const int iy0 = ty * ST_TH * 2 - 3, ix0 = tx * ST_TW * 2 - 3;
#pragma unroll
    for (int j = 0; j < NPV; ++j) {
      const int iy = iy0 + ppy[j], ix = ix0 + pq[j] / 3;
      const bool ok = ppy[j] >= 0 && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
      pv[j] = ok ? x[(((size_t)n * H + iy) * W + ix0) * 3 + pq[j]] : 0.f;
    }
  };
  const int tile0 = blockIdx.x * STH_TPB;
  if (tile0 < tiles) load_patch(tile0);
  for (int tt = 0; tt < STH_TPB; ++tt) {
    const int tile = tile0 + tt;
    if (tile >= tiles) break;
    const int ty = tile / tiles_w, tx = tile % tiles_w;
    const int oy0 = ty * ST_TH, ox0 = tx * ST_TW;
    __syncthreads();                                   // the previous tile's patch has been consumed (weights, zero fill complete)
#pragma unroll
    for (int j = 0; j < NPV; ++j) {
      if (ppy[j] >= 0) {
        const float v = pv[j] * f16_sx;
        const _Float16 h1 = (_Float16)v, h2 = (_Float16)(v - (float)h1);
        ph[ppy[j] * STH_PROW + pq[j]] = __builtin_bit_cast(unsigned short, h1);
        ph[ST_PH * STH_PROW + ppy[j] * STH_PROW + pq[j]] = __builtin_bit_cast(unsigned short, h2);
      }
    }
    __syncthreads();
    if (tt + 1 < STH_TPB && tile + 1 < tiles) load_patch(tile + 1);
    f32x16 acc0, acc1;
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < STH_KS; ++ks) {
      uint4 a[2], b0[2], b1[2];
#pragma unroll
      for (int pc = 0; pc < 2; ++pc) {
        const unsigned* ap = reinterpret_cast<const unsigned*>(ph + pc * ST_PH * STH_PROW + aoff[ks]);   // 4-byte aligned
        a[pc] = make_uint4(ap[0], ap[1], ap[2], ap[3]);
        const unsigned short* wp = wsh + pc * ST_CO * STH_WROW + ks * 16 + hh * 8;
        b0[pc] = *reinterpret_cast<const uint4*>(wp + i * STH_WROW);
        b1[pc] = *reinterpret_cast<const uint4*>(wp + (32 + i) * STH_WROW);
      }
      frag_regs(a); frag_regs(b0); frag_regs(b1);
#pragma unroll
      for (int t = 0; t < 3; ++t) {                    // a2 b1, a1 b2, a1 b1 (smallest first)
        const int pa = t == 0 ? 1 : 0, pb = t == 1 ? 1 : 0;
        acc0 = mfma_piece<2>(a[pa], b0[pb], acc0);
        acc1 = mfma_piece<2>(a[pa], b1[pb], acc1);
      }
    }
    const int col = lane & 31, rsub = 4 * (lane >> 5);
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int pi = (reg & 3) + 8 * (reg >> 2) + rsub;        // pixel index within the wave's 2 x 16 strip
      const int oy = oy0 + wave * 2 + (pi >> 4), ox = ox0 + (pi & 15);
      if (oy < Ho && ox < Wo) {
        float* o = y + (((size_t)n * Ho + oy) * Wo + ox) * ST_CO;
        o[col] = acc0[reg] * f16_desc;
        o[32 + col] = acc1[reg] * f16_desc;
      }
    }
  }
}

// ------------------------------------------------------------------------------------
// Stem weight gradient: dW[co][k] = sum over pixels dy[p][co] * x[tap k of p], k = (r * 7 + s) * 3 + c, K = pixels.
// It is the LAST weight gradient of a backward pass (its dy is the last gradient the pass produces), so it runs alone at the
// tail of every step: the general global-load kernel took 2.03 ms there (38.9 TFLOP/s: three-channel scalar gathers).
// Same staging as stem_fwd_kernel: a block walks 8 x 16 output patches of its share; per patch the 21 x 37 x 3 input patch
// and the 128 x 64 dy patch go to LDS, every MFMA operand is one ds_read_b32.  Output 64 x 147 = 2 row blocks x 5 column
// blocks of 32: wave w owns row block w & 1, all five column blocks and every other K-step (pixel pair): two slabs per block.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                         float* __restrict__ slabs, int N, int H, int W, int Ho, int Wo,
                                                         int patches, int ppb) {
  __shared__ float patch[ST_PH * ST_PW * 3 + 8];
  __shared__ __align__(16) float dys[ST_TH * ST_TW * ST_CO];        // [pixel][co]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cb = wave & 1, kp = wave >> 1;             // output-channel block; which half of the K-steps (pixel pairs)
  const int tiles_w = Wo / ST_TW, tiles_h = Ho / ST_TH, per_img = tiles_w * tiles_h;
  const int i = lane & 31, kh = lane >> 5;
  int toff[5];                                         // tap offset of this lane's column in each of the five column blocks
  bool tok[5];
#pragma unroll
  for (int b = 0; b < 5; ++b) {
    const int nn = b * 32 + i;
    tok[b] = nn < ST_K;
    const int k = tok[b] ? nn : 0;
    toff[b] = ((k / 21) * ST_PW + (k / 3) % 7) * 3 + k % 3;
  }
  f32x16 acc[5];
#pragma unroll
  for (int b = 0; b < 5; ++b)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[b][e] = 0.f;
  const int pbeg = blockIdx.x * ppb, pend = min(patches, pbeg + ppb);
  for (int pt = pbeg; pt < pend; ++pt) {
    const int n = pt / per_img, t = pt - n * per_img;
    const int oy0 = (t / tiles_w) * ST_TH, ox0 = (t % tiles_w) * ST_TW;
    const int iy0 = oy0 * 2 - 3, ix0 = ox0 * 2 - 3;
    __syncthreads();                                   // the previous patch has been consumed
    for (int e = tid; e < ST_PH * ST_PW * 3; e += 256) {
      const int c = e % 3, px = (e / 3) % ST_PW, py = e / (3 * ST_PW);
      const int iy = iy0 + py, ix = ix0 + px;
      patch[e] = ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) ? x[(((size_t)n * H + iy) * W + ix) * 3 + c] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < ST_TH * ST_TW * ST_CO / 4 / 256; ++j) {     // 8 float4 per thread
      const int e = tid + 256 * j, pix = e >> 4, q = e & 15;
      const int oy = oy0 + (pix >> 4), ox = ox0 + (pix & 15);
      *reinterpret_cast<float4*>(dys + pix * ST_CO + q * 4) =
          *reinterpret_cast<const float4*>(dy + (((size_t)n * Ho + oy) * Wo + ox) * ST_CO + q * 4);
    }
    __syncthreads();
#pragma unroll 2
    for (int ks = kp; ks < ST_TH * ST_TW / 2; ks += 2) {
      const int p = 2 * ks + kh;                       // the pixel this lane supplies (k index of the MFMA)
      const int pbase = (((p >> 4) * 2) * ST_PW + (p & 15) * 2) * 3;
      const float a = dys[p * ST_CO + cb * 32 + i];
#pragma unroll
      for (int b = 0; b < 5; ++b) {
        const float v = tok[b] ? patch[pbase + toff[b]] : 0.f;
        acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, v, acc[b], 0, 0, 0);
      }
    }
  }
  float* slab = slabs + ((size_t)blockIdx.x * 2 + kp) * ST_CO * ST_K;   // (the other output-channel block fills the other rows)
  const int col = lane & 31, rsub = 4 * (lane >> 5);
#pragma unroll
  for (int b = 0; b < 5; ++b) {
    const int k = b * 32 + col;
    if (k < ST_K) {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int co = cb * 32 + (reg & 3) + 8 * (reg >> 2) + rsub;
        slab[(size_t)co * ST_K + k] = acc[b][reg];
      }
    }
  }
}

static bool stem_wgrad_ok(const xas_conv_shape* s) {
  return s->Cin == 3 && s->Cout == ST_CO && s->R == 7 && s->S == 7 && s->stride == 2 && s->pad == 3 &&
         s->Ho % ST_TH == 0 && s->Wo % ST_TW == 0 && s->Ho * 2 == s->Hi && s->Wo * 2 == s->Wi;
}
constexpr int kStemWgradBlocks = 1024;

// ------------------------------------------------------------------------------------
// "Thin" 3x3 convolutions with ONE channel on one side (physique_network.py:41 first conv 1 -> 32,
// :50 last conv 32 -> 1): HBM-bound, no GEMM shape; vector side C <= 64, C % 4 == 0.
//   T1 vec->scalar : out[m]    = b + sum_tap sum_c V[src(m,tap)][c] * W[tap][c]   (fwd Cout=1, dgrad Cin=1)
//   T2 scalar->vec : out[m][c] = b[c] + sum_tap S[src(m,tap)] * W[c][tap]         (fwd Cin=1, dgrad Cout=1)
//   T3 reduction   : dW[c][tap] = sum_m S * V[c]                                  (wgrad Cout=1 / Cin=1)
// `flip` walks the taps transposed (data gradient, stride 1): src = m + pad - tap.
// ------------------------------------------------------------------------------------
struct ThinParams {
  int N, H, W, C;       // pixel grid (same for both sides: stride 1, 'same' padding) and vector width
  int pad, flip;
  FastDiv div_w, div_hw;   // pixel index -> (n, h, w) without 64-bit divisions (they cost ~100 instructions each)
};

static ThinParams thin_params(int N, int H, int W, int C, int pad, int flip) {
  ThinParams t{N, H, W, C, pad, flip, {}, {}};
  t.div_w.init((unsigned)W);
  t.div_hw.init((unsigned)(H * W));
  return t;
}

__device__ __forceinline__ void thin_decode(const ThinParams& p, unsigned m, int* n, int* h, int* w) {
  const unsigned nn = p.div_hw.div(m), rem = m - nn * (unsigned)(p.H * p.W);
  const unsigned hh = p.div_w.div(rem);
  *n = (int)nn; *h = (int)hh; *w = (int)(rem - hh * (unsigned)p.W);
}

__device__ __forceinline__ bool thin_src(const ThinParams& p, int h, int w, int r, int q, int* hs, int* ws) {
  *hs = p.flip ? h + p.pad - r : h - p.pad + r;
  *ws = p.flip ? w + p.pad - q : w - p.pad + q;
  return (unsigned)*hs < (unsigned)p.H && (unsigned)*ws < (unsigned)p.W;
}

// Thread = (pixel, 4-channel slot): the C/4 lanes of a pixel read one contiguous C*4-byte row per tap (coalesced),
// keep their 9 weight vectors in registers, and a block walks kThinPasses pixel groups.  (The first version gave a
// thread a whole pixel: 64 lanes x 16 B at a C*4-byte stride per load instruction, 0.9 TB/s.)
constexpr int kThinPasses = 8;

__global__ __launch_bounds__(256) void thin_vec2scalar_kernel(const float* __restrict__ V, const float* __restrict__ Wt,
                                                              const float* __restrict__ bias, float* __restrict__ out,
                                                              ThinParams p) {
  const int C4 = p.C / 4, ppb = 256 / C4;                      // lanes per pixel (power of two <= 16), pixels per pass
  const int cq = threadIdx.x % C4, pl = threadIdx.x / C4;
  float4 wv[9];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) wv[tap] = reinterpret_cast<const float4*>(Wt)[tap * C4 + cq];
  const unsigned M = (unsigned)(p.N * p.H * p.W);
  const float b0 = bias ? bias[0] : 0.f;
  for (int pass = 0; pass < kThinPasses; ++pass) {
    const unsigned m = ((unsigned)blockIdx.x * kThinPasses + pass) * ppb + pl;
    const bool on = m < M;
    int n, h, w;
    thin_decode(p, on ? m : 0u, &n, &h, &w);
    // all nine 16-byte loads are issued before the first is used (clamped address, zero weight outside the image): a
    // `continue` around each load made every tap wait out its own round trip
    float4 a[9];
    bool inb[9];
    const float* Vn = V + (size_t)n * p.H * p.W * p.C + cq * 4;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      int hs, ws;
      inb[tap] = thin_src(p, h, w, tap / 3, tap % 3, &hs, &ws);
      a[tap] = *reinterpret_cast<const float4*>(Vn + (size_t)(inb[tap] ? hs * p.W + ws : h * p.W + w) * p.C);
    }
    float acc = 0.f;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      float t = a[tap].x * wv[tap].x;
      t = fmaf(a[tap].y, wv[tap].y, t); t = fmaf(a[tap].z, wv[tap].z, t); t = fmaf(a[tap].w, wv[tap].w, t);
      acc += inb[tap] ? t : 0.f;
    }
    for (int o = C4 >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);     // the C4 lanes of a pixel are adjacent
    if (on && cq == 0) out[m] = acc + b0;
  }
}

// Wct: [C][9] (packed [C][R][S][1])
__global__ __launch_bounds__(256) void thin_scalar2vec_kernel(const float* __restrict__ S, const float* __restrict__ Wct,
                                                              const float* __restrict__ bias, float* __restrict__ out,
                                                              ThinParams p) {
  const int C4 = p.C / 4, ppb = 256 / C4;
  const int cq = threadIdx.x % C4, pl = threadIdx.x / C4;
  const int c = cq * 4;
  float4 wv[9];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
    wv[tap] = make_float4(Wct[(c + 0) * 9 + tap], Wct[(c + 1) * 9 + tap], Wct[(c + 2) * 9 + tap], Wct[(c + 3) * 9 + tap]);
  const float4 b4 = bias ? *reinterpret_cast<const float4*>(bias + c) : make_float4(0, 0, 0, 0);
  const unsigned M = (unsigned)(p.N * p.H * p.W);
  for (int pass = 0; pass < kThinPasses; ++pass) {
    const unsigned m = ((unsigned)blockIdx.x * kThinPasses + pass) * ppb + pl;
    if (m >= M) break;
    int n, h, w;
    thin_decode(p, m, &n, &h, &w);
    float4 acc = b4;
    const float* Sn = S + (size_t)n * p.H * p.W;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      int hs, ws;
      const bool inb = thin_src(p, h, w, tap / 3, tap % 3, &hs, &ws);
      const float sv = inb ? Sn[hs * p.W + ws] : 0.f;
      acc.x = fmaf(sv, wv[tap].x, acc.x); acc.y = fmaf(sv, wv[tap].y, acc.y);
      acc.z = fmaf(sv, wv[tap].z, acc.z); acc.w = fmaf(sv, wv[tap].w, acc.w);
    }
    *reinterpret_cast<float4*>(out + (size_t)m * p.C + c) = acc;
  }
}

// slabs[chunk][c][tap];  scalar_at_src = 0: S = dy[m], V = x[src(m,tap)]   (wgrad Cout = 1)
//                        scalar_at_src = 1: S = x[src(m,tap)], V = dy[m]   (wgrad Cin = 1)
__global__ __launch_bounds__(256) void thin_wgrad_kernel(const float* __restrict__ S, const float* __restrict__ V,
                                                         float* __restrict__ slabs, ThinParams p, int dir,
                                                         int m_per_chunk) {
  // dW[c][tap] = sum over pixels m of V[m][c] * S[m + dir * (tap - centre)]: the VECTOR side is walked once (one 16-byte
  // load per lane and pixel, four pixels in flight), the scalar side is gathered (9 taps, L1 hits).
  //   Cin == 1 : V = dy at the output pixel, S = x,  dir = +1;   Cout == 1: V = x at the source pixel, S = dy, dir = -1.
  __shared__ float4 red[4 * 16 * 9];
  const int C4 = p.C / 4, lanes = 256 / C4;
  const int cq = threadIdx.x % C4, pl = threadIdx.x / C4;
  const unsigned M = (unsigned)(p.N * p.H * p.W);
  const unsigned mbeg = (unsigned)blockIdx.x * (unsigned)m_per_chunk, mend = min(M, mbeg + (unsigned)m_per_chunk);
  float4 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = make_float4(0, 0, 0, 0);
  constexpr int U = 4;
  for (unsigned m = mbeg + pl; m < mend; m += U * lanes) {
    float4 vv[U];
    unsigned mu[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      mu[u] = m + u * lanes;
      const bool on = mu[u] < mend;
      vv[u] = *reinterpret_cast<const float4*>(V + (size_t)(on ? mu[u] : mbeg) * p.C + cq * 4);
      if (!on) vv[u] = make_float4(0, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int n, h, w;
      thin_decode(p, mu[u] < mend ? mu[u] : mbeg, &n, &h, &w);
      const float* Sn = S + (size_t)n * p.H * p.W;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int hs = h + dir * (tap / 3 - p.pad), ws = w + dir * (tap % 3 - p.pad);
        const bool inb = (unsigned)hs < (unsigned)p.H && (unsigned)ws < (unsigned)p.W;
        const float sv = inb ? Sn[hs * p.W + ws] : 0.f;
        acc[tap].x = fmaf(sv, vv[u].x, acc[tap].x); acc[tap].y = fmaf(sv, vv[u].y, acc[tap].y);
        acc[tap].z = fmaf(sv, vv[u].z, acc[tap].z); acc[tap].w = fmaf(sv, vv[u].w, acc[tap].w);
      }
    }
  }
  // reduce over the pixel lanes: inside a wave by shuffles (lanes cq, cq + C4, ... hold the same channels), then across
  // the four waves through LDS with ONE barrier (the first version ran nine barrier pairs and a serial 32-term sum)
  float* o = slabs + (size_t)blockIdx.x * p.C * 9;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    float4 a = acc[tap];
    for (int off = C4; off < 64; off <<= 1) {
      a.x += __shfl_xor(a.x, off, 64); a.y += __shfl_xor(a.y, off, 64);
      a.z += __shfl_xor(a.z, off, 64); a.w += __shfl_xor(a.w, off, 64);
    }
    acc[tap] = a;
  }
  float4* red4 = red;                                  // [4 waves][C4 <= 16][9 taps] float4 <= 576 entries: see below
  if (lane < C4) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) red4[(wave * C4 + lane) * 9 + tap] = acc[tap];
  }
  __syncthreads();
  if (threadIdx.x < C4) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      float4 s4 = red4[(0 * C4 + cq) * 9 + tap];
      for (int k = 1; k < 4; ++k) { const float4 r = red4[(k * C4 + cq) * 9 + tap]; s4.x += r.x; s4.y += r.y; s4.z += r.z; s4.w += r.w; }
      o[(cq * 4 + 0) * 9 + tap] = s4.x; o[(cq * 4 + 1) * 9 + tap] = s4.y;
      o[(cq * 4 + 2) * 9 + tap] = s4.z; o[(cq * 4 + 3) * 9 + tap] = s4.w;
    }
  }
}

static bool thin_ok(const xas_conv_shape* s, int C) {
  return s->R == 3 && s->S == 3 && s->stride == 1 && s->pad == 1 && C % 4 == 0 && C >= 4 && C <= 64 &&
         (256 % (C / 4)) == 0 && s->Ho == s->Hi && s->Wo == s->Wi;
}
constexpr int kThinChunk = 2048;
static inline unsigned thin_grid(long M, int C) { return (unsigned)cdiv(M, (long)kThinPasses * (256 / (C / 4))); }

// ------------------------------------------------------------------------------------
// direct (VALU) fallbacks for shapes the MFMA tiles do not cover (Cin = 1 / 3, Cout = 1)
// ------------------------------------------------------------------------------------
__global__ void direct_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                  const float* __restrict__ bias, float* __restrict__ y, xas_conv_shape s) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)s.N * s.Ho * s.Wo * s.Cout;
  if (idx >= total) return;
  const int co = idx % s.Cout;
  long m = idx / s.Cout;
  const int wo = m % s.Wo; m /= s.Wo;
  const int ho = m % s.Ho; const int n = m / s.Ho;
  float acc = bias ? bias[co] : 0.f;
  for (int r = 0; r < s.R; ++r) {
    const int hi = ho * s.stride - s.pad + r;
    if ((unsigned)hi >= (unsigned)s.Hi) continue;
    for (int q = 0; q < s.S; ++q) {
      const int wi = wo * s.stride - s.pad + q;
      if ((unsigned)wi >= (unsigned)s.Wi) continue;
      const float* xp = x + (((size_t)n * s.Hi + hi) * s.Wi + wi) * s.Cin;
      const float* wp = w + (((size_t)co * s.R + r) * s.S + q) * s.Cin;
      for (int c = 0; c < s.Cin; ++c) acc = fmaf(xp[c], wp[c], acc);
    }
  }
  y[idx] = acc;
}

// wt packed transposed [Cin][R][S][Cout]
__global__ void direct_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ wt,
                                    float* __restrict__ dx, xas_conv_shape s) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)s.N * s.Hi * s.Wi * s.Cin;
  if (idx >= total) return;
  const int ci = idx % s.Cin;
  long m = idx / s.Cin;
  const int wi = m % s.Wi; m /= s.Wi;
  const int hi = m % s.Hi; const int n = m / s.Hi;
  float acc = 0.f;
  for (int r = 0; r < s.R; ++r) {
    const int th = hi + s.pad - r;
    if (th < 0 || th % s.stride) continue;
    const int ho = th / s.stride;
    if (ho >= s.Ho) continue;
    for (int q = 0; q < s.S; ++q) {
      const int tw = wi + s.pad - q;
      if (tw < 0 || tw % s.stride) continue;
      const int wo = tw / s.stride;
      if (wo >= s.Wo) continue;
      const float* gp = dy + (((size_t)n * s.Ho + ho) * s.Wo + wo) * s.Cout;
      const float* wp = wt + (((size_t)ci * s.R + r) * s.S + q) * s.Cout;
      for (int c = 0; c < s.Cout; ++c) acc = fmaf(gp[c], wp[c], acc);
    }
  }
  dx[idx] = acc;
}

// Cout == 1 weight gradient (physique_network.py:50, the 32 -> 1 output conv): thread = column nn,
// block walks a pixel chunk; slabs [chunks][KK] are summed by slab_reduce_kernel.
__global__ void wgrad_cout1_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                   float* __restrict__ slabs, xas_conv_shape s, int m_per_chunk) {
  const int KK = s.R * s.S * s.Cin;
  const int nn = blockIdx.x * blockDim.x + threadIdx.x;
  const long M = (long)s.N * s.Ho * s.Wo;
  const long mbeg = (long)blockIdx.y * m_per_chunk, mend = min(M, mbeg + m_per_chunk);
  if (nn >= KK) return;
  const int tap = nn / s.Cin, c = nn % s.Cin, r = tap / s.S, q = tap % s.S;
  float acc = 0.f;
  for (long m = mbeg; m < mend; ++m) {
    const int wo = m % s.Wo; const long t = m / s.Wo; const int ho = t % s.Ho; const int n = t / s.Ho;
    const int hi = ho * s.stride - s.pad + r, wi = wo * s.stride - s.pad + q;
    if ((unsigned)hi < (unsigned)s.Hi && (unsigned)wi < (unsigned)s.Wi)
      acc = fmaf(dy[m], x[(((size_t)n * s.Hi + hi) * s.Wi + wi) * s.Cin + c], acc);
  }
  slabs[(size_t)blockIdx.y * KK + nn] = acc;
}

// OIHW <-> packed
__global__ void pack_weight_kernel(const float* __restrict__ src, float* __restrict__ dst, int Cout, int Cin, int R,
                                   int S, int transposed, int unpack) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)Cout * Cin * R * S;
  if (idx >= total) return;
  // idx enumerates the packed tensor
  long t = idx;
  int co, ci, r, q;
  if (!transposed) { ci = t % Cin; t /= Cin; q = t % S; t /= S; r = t % R; co = t / R; }
  else { co = t % Cout; t /= Cout; q = t % S; t /= S; r = t % R; ci = t / R; }
  const long o = (((long)co * Cin + ci) * R + r) * S + q;
  if (unpack) dst[o] = src[idx]; else dst[idx] = src[o];
}

static int check_shape(const xas_conv_shape* s, const char* who) {
  XAS_REQUIRE(s != nullptr, "%s: null shape", who);
  XAS_REQUIRE(s->N > 0 && s->Hi > 0 && s->Wi > 0 && s->Cin > 0 && s->Cout > 0 && s->R > 0 && s->S > 0 &&
                  s->stride > 0 && s->pad >= 0, "%s: non-positive dimension", who);
  XAS_REQUIRE((long)s->N * s->Hi * s->Wi * s->Cin < (1l << 31) && (long)s->N * s->Ho * s->Wo * s->Cout < (1l << 31),
              "%s: tensor too large for 32-bit row indices", who);
  return 0;
}

static int check_fwd_dims(const xas_conv_shape* s, const char* who) {
  XAS_REQUIRE(s->Ho == (s->Hi + 2 * s->pad - s->R) / s->stride + 1 && s->Wo == (s->Wi + 2 * s->pad - s->S) / s->stride + 1,
              "%s: Ho/Wo (%d,%d) inconsistent with Hi=%d Wi=%d R=%d S=%d stride=%d pad=%d", who, s->Ho, s->Wo, s->Hi,
              s->Wi, s->R, s->S, s->stride, s->pad);
  return 0;
}

template <int BM, int BN, int MODE, int NBUF = 2, bool PIPE = false>
static int launch_igemm(const IgemmParams& p, int Mrows_max, int phases, hipStream_t st) {
  const size_t lds = (size_t)NBUF * (BM + BN) * LDK * sizeof(float);
  static bool attr_set_dev[kMaxDevices] = {};          // per device: the LDS limit is a per-device function attribute
  bool& attr_set = attr_set_dev[current_device()];
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_kernel<BM, BN, MODE, NBUF, PIPE>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  IgemmParams q = p;
  q.nMt = (int)cdiv(Mrows_max, BM); q.nNt = (int)cdiv(p.Cd, BN); q.mt_per_xcd = (int)cdiv(q.nMt, 8);
  const unsigned nblk = (unsigned)(8 * q.mt_per_xcd * q.nNt);
  dim3 grid(nblk, 1, (unsigned)phases);
  hipLaunchKernelGGL((igemm_kernel<BM, BN, MODE, NBUF, PIPE>), grid, dim3(256), lds, st, q);
  XAS_LAUNCH_CHECK();
  return 0;
}

template <int BM, int BN, int MODE, bool PIPE, bool BNB = false>
static int launch_igemm_buf(const IgemmParams& p, int Mrows_max, int phases, hipStream_t st) {
  const size_t lds = (size_t)2 * (BM + BN) * LDK * sizeof(float);
  static bool attr_set_dev[kMaxDevices] = {};          // per device: the LDS limit is a per-device function attribute
  bool& attr_set = attr_set_dev[current_device()];
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_buf_kernel<BM, BN, MODE, PIPE, BNB>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  IgemmParams q = p;
  q.nMt = (int)cdiv(Mrows_max, BM); q.nNt = (int)cdiv(p.Cd, BN); q.mt_per_xcd = (int)cdiv(q.nMt, 8);
  const unsigned nblk = (unsigned)(8 * q.mt_per_xcd * q.nNt);
  dim3 grid(nblk, 1, (unsigned)phases);
  hipLaunchKernelGGL((igemm_buf_kernel<BM, BN, MODE, PIPE, BNB>), grid, dim3(256), lds, st, q);
  XAS_LAUNCH_CHECK();
  return 0;
}

// The 32-bit offset scheme of the buffer-load kernels (conv.hip and conv_x6.hip) addresses tensors below 2 GiB incl. the
// padding region in front of them, and tap windows of at most 32 taps.
static bool igemm_fits(const IgemmParams& p) {
  const long bias = ((long)(p.R + p.pad) * p.Ws + p.S + p.pad) * p.Cs;
  return (bias + p.src_elems) * 4 < 0x7fffff00l && p.wgt_elems * 6 < 0x7fffff00l && p.R * p.S <= 32;
}

// Exact-fp32 MFMA path: buffer-load kernel unless its offset scheme cannot address the tensor or a coverage test asks for
// the global-load kernel (tune bit6 = 64); tune bit5 (32): plain K-loop instead of the pipelined one.
template <int BM, int BN, int MODE>
static int launch_tile(const IgemmParams& p, int Mrows_max, int phases, hipStream_t st) {
  const bool fits = igemm_fits(p);
  if constexpr (MODE == 1) {
    if (p.bnb_x) {                                    // batch-norm backward epilogue: buffer-load pipelined kernel only
      XAS_REQUIRE(fits, "conv_dgrad_bn_bwd: tensor beyond the 32-bit offset range of the buffer-load kernel");
      return launch_igemm_buf<BM, BN, 1, true, true>(p, Mrows_max, phases, st);
    }
  }
  // K loops of one or two steps (1x1 layers with 32 / 64 input channels): the pipelined loop's look-ahead loads have
  // nothing to look ahead to (they re-load the last step) - the plain loop is 18 % faster there (0.608 -> 0.497 ms for
  // 64 -> 256 channels at 256 x 64 x 64, r02)
  const bool short_k = p.stride == 1 && (long)p.R * p.S * p.Cs <= 2 * BK;
  if (fits && !(p.tune & 64)) {
    if ((p.tune & 32) || short_k) return launch_igemm_buf<BM, BN, MODE, false>(p, Mrows_max, phases, st);
    return launch_igemm_buf<BM, BN, MODE, true>(p, Mrows_max, phases, st);
  }
  if (p.tune & 32) return launch_igemm<BM, BN, MODE, 2, false>(p, Mrows_max, phases, st);
  return launch_igemm<BM, BN, MODE, 2, true>(p, Mrows_max, phases, st);
}

static int g_precision = XAS_PREC_F16X3;     // process default (xas_set_precision); a call overrides it with xas_conv_shape.mode

static inline int precision_of(const xas_conv_shape* s) { return (s->mode & 0xff) > 0 ? (s->mode & 0xff) - 1 : g_precision; }
// operand planes of a pass in a precision mode (pass 0: forward-type launch, 1 data gradient, 2 weight gradient):
// XAS_PREC_F16X3 runs on two fp16 planes when the maximum of EVERY tensor operand of the launch came with the call
// (xas_conv_shape.grad_amax; a weight gradient also needs x_amax) - the scale of each split is derived from it, there is no
// fixed scale and therefore no range the operands must stay in (r04) - and as bf16x6 otherwise
static inline int planes_of(int prec, int pass, bool has_amax) {
  (void)pass;
  if (prec == XAS_PREC_F32) return 0;
  if (prec == XAS_PREC_BF16) return 1;
  return (prec == XAS_PREC_F16X3 && has_amax) ? 2 : 3;
}
// do the maxima a pass needs come with the call?
static inline bool has_amax_for(const xas_conv_shape* s, int pass) {
  return pass == 2 ? (s->grad_amax != nullptr && s->x_amax != nullptr) : s->grad_amax != nullptr;
}

// prec: XAS_PREC_*.  The bf16-split kernels (conv_x6.hip) take PRE-SPLIT weights (xas_split_weight); the exact-fp32 kernels
// take fp32 packed weights: xas_conv_weight_planes tells the caller which of the two a (shape, pass) wants.
template <int MODE>
static int dispatch_igemm(const IgemmParams& p, int prec, int Mrows_max, int phases, hipStream_t st) {
  if (prec != XAS_PREC_F32) {
    XAS_REQUIRE(igemm_fits(p), "conv: tensor beyond the 32-bit offset range of the bf16-split kernels (use XAS_PREC_F32)");
    return launch_igemm_x6(p, MODE, Mrows_max, phases, planes_of(prec, MODE, p.a_amax != nullptr && !p.bnb_x), st);
  }
  int bm, bn;
  pick_tile(p.Cd, Mrows_max, phases, &bm, &bn);
  if (bn == 128) return launch_tile<128, 128, MODE>(p, Mrows_max, phases, st);
  if (bm == 64) return launch_tile<64, 64, MODE>(p, Mrows_max, phases, st);
  if (bn == 64) return launch_tile<128, 64, MODE>(p, Mrows_max, phases, st);
  return launch_tile<128, 32, MODE>(p, Mrows_max, phases, st);
}

}  // namespace xas

using namespace xas;

static bool wgrad_on_x6(const xas_conv_shape* s, const float* x);
static int images_per_launch(int N, long elems_per_image_a, long elems_per_image_b);

// Images per launch so that a gathered tensor stays below the 2 GiB range of the 32-bit buffer offsets (camera-batched
// passes: the logits of 128+ images are 2.4 GB and more).  Images are independent in all three convolution passes, so
// a larger batch is processed as several launches over image ranges (weight gradients: the later ones accumulate).
static int images_per_launch(int N, long elems_per_image_a, long elems_per_image_b) {
  const long lim = (0x7fffff00l / 4) - (1l << 22);            // leave room for the padding region in front of the tensor
  const long per = elems_per_image_a > elems_per_image_b ? elems_per_image_a : elems_per_image_b;
  if ((long)N * per < lim || N <= 1) return N;
  long n = lim / per;
  if (n < 1) n = 1;
  const long launches = cdiv(N, n);
  return (int)cdiv(N, launches);                               // equal-sized ranges
}

extern "C" int xas_set_tuning(int flags) { g_tune = flags; return 0; }
extern "C" int xas_set_precision(int mode) {
  XAS_REQUIRE(mode >= 0 && mode <= 3, "set_precision: 0 = exact fp32 MFMA, 1 = bf16 MFMA (not fp32 accurate), 2 = bf16x6 (fp32-accurate, no "
              "operand maxima needed), 3 = f16x3 (default; fp32-accurate; launches that come without the maxima of their operands run as bf16x6)");
  g_precision = mode;
  return 0;
}
extern "C" int xas_get_precision(void) { return g_precision; }

static int conv_fwd_impl(const float* x, const float* w_packed, const float* bias, float* y, const xas_conv_shape* s,
                         void* stream, float* stat_partial, const float* stat_pivot, float* head_partial = nullptr);

namespace xas {
__global__ void stem_overflow_peek_kernel(unsigned* out) { if (g_f16_weight_overflow_stem) atomicOr(out, 1u); }
int stem_weight_overflow_peek(unsigned* device_out, void* stream) {
  hipLaunchKernelGGL(stem_overflow_peek_kernel, dim3(1), dim3(1), 0, as_stream(stream), device_out);
  XAS_LAUNCH_CHECK();
  return 0;
}
int stem_weight_overflow(int reset) {
  unsigned v = 0u;
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_f16_weight_overflow_stem), sizeof(v)) != hipSuccess) return -1;
  if (v && reset) {
    const unsigned z = 0u;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_f16_weight_overflow_stem), &z, sizeof(z));
  }
  return v ? 1 : 0;
}
}  // namespace xas

extern "C" int xas_conv_fwd(const float* x, const float* w_packed, const float* bias, float* y,
                            const xas_conv_shape* s, void* stream) {
  return conv_fwd_impl(x, w_packed, bias, y, s, stream, nullptr, nullptr);
}

// Rows per tile when the forward pass of `s` can emit batch-norm partial sums for `groups` camera groups (MFMA path, one
// launch, no tile straddles a group), else 0.
static int fwd_stats_tile_rows(const xas_conv_shape* s, int groups) {
  if (!s || groups < 1) return 0;
  if (s->Cin % BK != 0 || s->Cout < 16 || s->Cout % 4 != 0) return 0;
  if (s->Cin == 3 || (s->Cout == 1 && thin_ok(s, s->Cin)) || (s->Cin == 1 && thin_ok(s, s->Cout))) return 0;
  if (images_per_launch(s->N, (long)s->Hi * s->Wi * s->Cin, 0) < s->N || s->N % groups) return 0;
  const long M = (long)s->N * s->Ho * s->Wo, Mg = M / groups;
  int bm, bn;
  pick_tile(s->Cout, M, 1, &bm, &bn);
  // the bf16-split kernels use 64 x 256 tiles for wide layers, unless the tap re-use kernel (128-row patches) takes the shape
  const bool x6 = precision_of(s) != XAS_PREC_F32;
  const bool tap = x6 && bm == 128 && !(g_tune & (1 << 22)) &&
                   tap_tile_ok(s->R, s->S, s->stride, s->pad, s->Hi, s->Wi, s->Ho, s->Wo, s->Cin, s->N);
  if (x6 && !tap && !(g_tune & (1 << 23))) pick_tile(s->Cout, M, 1, &bm, &bn, true);
  if (Mg % bm) return 0;
  if ((M / bm) * 2 * (long)s->Cout * 4 >= 0x7fffff00l) return 0;
  return bm;
}

static int conv_fwd_impl(const float* x, const float* w_packed, const float* bias, float* y, const xas_conv_shape* s,
                         void* stream, float* stat_partial, const float* stat_pivot, float* head_partial) {
  if (check_shape(s, "conv_fwd") || check_fwd_dims(s, "conv_fwd")) return 1;
  XAS_REQUIRE(x && w_packed && y, "conv_fwd: null buffer");
  {
    const long xi = (long)s->Hi * s->Wi * s->Cin, yi = (long)s->Ho * s->Wo * s->Cout;
    const int per = images_per_launch(s->N, xi, 0);
    if (per < s->N) {
      XAS_REQUIRE(!stat_partial, "conv_fwd: the statistics epilogue needs a single launch");
      for (int n0 = 0; n0 < s->N; n0 += per) {
        xas_conv_shape part = *s;
        part.N = s->N - n0 < per ? s->N - n0 : per;
        // (head partial records: [image][Ho * Wo / 64][Cout / 64][67] - the image range of this launch)
        float* hp = head_partial ? head_partial + (size_t)n0 * (s->Ho * s->Wo / 64) * (s->Cout / 64) * 67 : nullptr;
        const int rc = conv_fwd_impl(x + (size_t)n0 * xi, w_packed, bias, y + (size_t)n0 * yi, &part, stream, nullptr, nullptr, hp);
        if (rc) return rc;
      }
      return 0;
    }
  }
  hipStream_t st = as_stream(stream);
  if (s->Cin == 3 && s->R == 7 && s->S == 7 && s->stride == 2 && s->pad == 3 && s->Cout == ST_CO && bias == nullptr) {
    const int tiles = (int)(cdiv(s->Ho, ST_TH) * cdiv(s->Wo, ST_TW));
    // f16x3 needs max |image| (the scale of the split); without it - and under tune bit 25 - the exact-fp32 stem kernel
    if (precision_of(s) == XAS_PREC_F16X3 && s->grad_amax && !(g_tune & (1 << 25)))
      hipLaunchKernelGGL(stem_fwd_f16_kernel, dim3((unsigned)cdiv(tiles, STH_TPB), s->N), dim3(256), 0, st, x, w_packed, y, s->N, s->Hi,
                         s->Wi, s->Ho, s->Wo, s->grad_amax);
    else
    hipLaunchKernelGGL(stem_fwd_kernel, dim3((unsigned)cdiv(tiles, ST_TPB), s->N), dim3(256), 0, st, x, w_packed, y, s->N, s->Hi,
                       s->Wi, s->Ho, s->Wo);
    XAS_LAUNCH_CHECK();
    return 0;
  }
  if (s->Cout == 1 && thin_ok(s, s->Cin)) {
    const ThinParams tp = thin_params(s->N, s->Hi, s->Wi, s->Cin, s->pad, 0);
    const long M = (long)s->N * s->Hi * s->Wi;
    hipLaunchKernelGGL(thin_vec2scalar_kernel, dim3(thin_grid(M, s->Cin)), dim3(256), 0, st, x, w_packed, bias, y, tp);
    XAS_LAUNCH_CHECK();
    return 0;
  }
  if (s->Cin == 1 && thin_ok(s, s->Cout)) {
    const ThinParams tp = thin_params(s->N, s->Hi, s->Wi, s->Cout, s->pad, 0);
    const long M = (long)s->N * s->Hi * s->Wi;
    hipLaunchKernelGGL(thin_scalar2vec_kernel, dim3(thin_grid(M, s->Cout)), dim3(256), 0, st, x, w_packed,
                       bias, y, tp);
    XAS_LAUNCH_CHECK();
    return 0;
  }
  if (s->Cin % BK != 0 || s->Cout < 16) {
    const long total = (long)s->N * s->Ho * s->Wo * s->Cout;
    hipLaunchKernelGGL(direct_fwd_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, st, x, w_packed, bias, y, *s);
    XAS_LAUNCH_CHECK();
    return 0;
  }
  XAS_REQUIRE((((uintptr_t)x | (uintptr_t)w_packed) & 15) == 0, "conv_fwd: operands must be 16-byte aligned");
  IgemmParams p{};
  p.src = x; p.wgt = w_packed; p.bias = bias; p.out = y; p.N = s->N;
  p.a_amax = s->grad_amax;
  p.Hs = s->Hi; p.Ws = s->Wi; p.Cs = s->Cin; p.Hd = s->Ho; p.Wd = s->Wo; p.Cd = s->Cout;
  p.R = s->R; p.S = s->S; p.stride = s->stride; p.pad = s->pad; p.Hrow = s->Ho; p.Wrow = s->Wo; p.tune = g_tune;
  p.div_hw.init((unsigned)(s->Ho * s->Wo)); p.div_w.init((unsigned)s->Wo);
  p.src_elems = (long)p.N * p.Hs * p.Ws * p.Cs; p.wgt_elems = (long)p.Cd * p.R * p.S * p.Cs;
  p.stat_partial = stat_partial; p.stat_pivot = stat_pivot;
  p.head_partial = head_partial;
  return dispatch_igemm<0>(p, precision_of(s), s->N * s->Ho * s->Wo, 1, st);
}

// Final 1x1 convolution of the detector + the soft-argmax head's pass over its result (forward
// half): deconv_head.py:34-35 -> keypoint_detector_integral_multi.py:36-62,70-74.  The logits are still written (the backward
// needs them); what disappears is head_partial_kernel's read of them (18.9 MB per image).
// -> xas_conv_fwd_head_chunks: records per image (Ho * Wo / 64) when the fused path takes this shape, else 0: a 1x1 stride-1
//    convolution on the split-arithmetic MFMA kernels whose output is [N][64 x 64][K * 64] (depth_dim = 64, 64 x 64 heat maps).
extern "C" int xas_conv_fwd_head_chunks(const xas_conv_shape* s, int K, int D) {
  if (!s || K < 1 || D != 64) return 0;
  if (s->R != 1 || s->S != 1 || s->stride != 1 || s->pad != 0 || s->Wo != 64 || (s->Ho * s->Wo) % 64 != 0) return 0;
  if (s->Cout != K * D || s->Cin % BK != 0 || precision_of(s) == XAS_PREC_F32) return 0;
  return s->Ho * s->Wo / 64;
}

extern "C" int xas_conv_fwd_head(const float* x, const float* w_packed, const float* bias, float* y, const xas_conv_shape* s,
                                 int K, int D, float* head_partial, void* stream) {
  XAS_REQUIRE(head_partial && xas_conv_fwd_head_chunks(s, K, D) > 0, "conv_fwd_head: shape not taken (xas_conv_fwd_head_chunks) or null buffer");
  return conv_fwd_impl(x, w_packed, bias, y, s, stream, nullptr, nullptr, head_partial);
}

// Which weight buffer the forward (pass 0: xas_conv_fwd*, ConvTranspose backward) or data-gradient (pass 1: xas_conv_dgrad*,
// ConvTranspose forward) entry points expect for this shape in its precision mode: 0 = fp32 packed weights
// (xas_pack_weight); 1 or 3 = that many bf16 planes (xas_split_weight of the packed weights).  Shapes outside the MFMA
// tiles (stem, one-channel "thin" layers, direct fallbacks) always take fp32 weights.
extern "C" int xas_conv_weight_planes(const xas_conv_shape* s, int pass) {
  if (!s) return 0;
  const int prec = precision_of(s);
  if (prec == XAS_PREC_F32) return 0;
  const bool thin = (s->Cout == 1 && thin_ok(s, s->Cin)) || (s->Cin == 1 && thin_ok(s, s->Cout));
  bool mfma;
  if (pass == 0) mfma = !thin && !(s->Cin == 3 && s->R == 7) && s->Cin % BK == 0 && s->Cout >= 16;
  else mfma = !thin && s->Cout % BK == 0 && s->Cin >= 16;
  return mfma ? planes_of(prec, pass, has_amax_for(s, pass)) : 0;
}

// Which kernel family a pass of this shape runs on (for measurement: bench.py prices every launch against the peak of the
// pipe it actually uses).  pass: 0 forward-type, 1 data-gradient-type, 2 weight gradient.
// -> 0 no MFMA (direct / one-channel kernels), 1 exact-fp32 MFMA, 2 bf16 MFMA, 3 bf16x6 MFMA, 4 f16x3 MFMA.
extern "C" int xas_conv_kernel_class(const xas_conv_shape* s, int pass) {
  if (!s) return 0;
  const int prec = precision_of(s);
  const int split = prec == XAS_PREC_F32 ? 1 : (prec == XAS_PREC_BF16 ? 2 : (planes_of(prec, pass, has_amax_for(s, pass)) == 2 ? 4 : 3));
  const bool thin = (s->Cout == 1 && thin_ok(s, s->Cin)) || (s->Cin == 1 && thin_ok(s, s->Cout));
  if (thin) return 0;
  if (pass == 0) {
    if (s->Cin == 3 && s->R == 7 && s->S == 7 && s->stride == 2 && s->pad == 3 && s->Cout == ST_CO)              // stem kernels
      return (prec == XAS_PREC_F16X3 && s->grad_amax && !(g_tune & (1 << 25))) ? 4 : 1;
    return (s->Cin % BK == 0 && s->Cout >= 16) ? split : 0;
  }
  if (pass == 1) return (s->Cout % BK == 0 && s->Cin >= 16) ? split : 0;
  if (s->Cout == 1) return 0;
  xas_conv_shape part = *s;          // tensors beyond the 32-bit offset range run as several launches over image ranges
  part.N = images_per_launch(s->N, (long)s->Hi * s->Wi * s->Cin, (long)s->Ho * s->Wo * s->Cout);
  return wgrad_on_x6(&part, nullptr) ? split : 1;
}

// Convolution (no bias) + the training-mode batch-norm statistics of its result.  When the tile grid lines up with the
// camera groups the statistics come out of the convolution's epilogue (per-tile partial sums, reduced by
// xas_bn_stats_from_partials); otherwise the result is read once more by xas_bn_stats.  Same outputs either way (up to
// the order of the fp32 partial sums).
extern "C" size_t xas_conv_fwd_bnstats_workspace_floats(const xas_conv_shape* s, int groups) {
  if (!s || groups < 1) return 0;
  const long M = (long)s->N * s->Ho * s->Wo;
  const int bm = fwd_stats_tile_rows(s, groups);
  if (!bm) return xas_bn_workspace_floats(M, s->Cout, groups);
  const long rows = M / bm;
  return (size_t)rows * 2 * s->Cout + xas_bn_workspace_floats(rows, 2 * s->Cout, groups);
}

extern "C" int xas_conv_fwd_bnstats(const float* x, const float* w_packed, float* y, const xas_conv_shape* s, int groups,
                                    const float* pivot, float* mean, float* var_biased, long out_stride, float* count_out,
                                    float* workspace, float* running_mean, float* running_var, float momentum,
                                    void* stream) {
  XAS_REQUIRE(s && groups >= 1 && s->N % groups == 0, "conv_fwd_bnstats: the batch does not split into %d groups", groups);
  XAS_REQUIRE(workspace && mean && var_biased, "conv_fwd_bnstats: null buffer");
  const long M = (long)s->N * s->Ho * s->Wo, Mg = M / groups;
  const int bm = fwd_stats_tile_rows(s, groups);
  if (!bm) {
    const int rc = conv_fwd_impl(x, w_packed, nullptr, y, s, stream, nullptr, nullptr);
    if (rc) return rc;
    return xas_bn_stats(y, M, s->Cout, groups, mean, var_biased, out_stride, count_out, workspace, running_mean,
                        running_var, momentum, Mg, stream);
  }
  const long rows = M / bm;
  const int rc = conv_fwd_impl(x, w_packed, nullptr, y, s, stream, workspace, pivot);
  if (rc) return rc;
  return xas_bn_stats_from_partials(workspace, rows, s->Cout, groups, Mg, pivot, mean, var_biased, out_stride, count_out,
                                    workspace + (size_t)rows * 2 * s->Cout, running_mean, running_var, momentum, stream);
}

static int conv_dgrad_impl(const float* dy, const float* w_packed_t, float* dx, const xas_conv_shape* s, void* stream,
                           int accumulate, const float* acc_src = nullptr, const unsigned char* acc_mask = nullptr,
                           const IgemmParams* bnb = nullptr);

extern "C" int xas_conv_dgrad(const float* dy, const float* w_packed_t, float* dx, const xas_conv_shape* s,
                              void* stream) {
  return conv_dgrad_impl(dy, w_packed_t, dx, s, stream, 0);
}

extern "C" int xas_conv_dgrad_acc(const float* dy, const float* w_packed_t, float* dx, const xas_conv_shape* s,
                                  void* stream) {
  return conv_dgrad_impl(dy, w_packed_t, dx, s, stream, 1);
}

extern "C" int xas_conv_dgrad_acc_masked(const float* dy, const float* w_packed_t, float* dx, const xas_conv_shape* s,
                                         const float* dprev, const uint8_t* mask, void* stream) {
  XAS_REQUIRE(dprev && mask, "conv_dgrad_acc_masked: null buffer");
  return conv_dgrad_impl(dy, w_packed_t, dx, s, stream, 2, dprev, mask);
}

static int conv_dgrad_impl(const float* dy, const float* w_packed_t, float* dx, const xas_conv_shape* s, void* stream,
                           int accumulate, const float* acc_src, const unsigned char* acc_mask, const IgemmParams* bnb) {
  if (check_shape(s, "conv_dgrad")) return 1;
  XAS_REQUIRE(!accumulate || (s->Cout % BK == 0 && s->Cin >= 16 && s->Cin % 4 == 0 && !(s->Cin == 1 || s->Cout == 1)),
              "conv_dgrad_acc: only the MFMA path accumulates (Cout %% 32 == 0, Cin %% 4 == 0, Cin >= 16)");
  XAS_REQUIRE(dy && w_packed_t && dx, "conv_dgrad: null buffer");
  // valid for the conv (Hi -> Ho) and for ConvTranspose2d forward (Ho given, Hi = (Ho-1)*stride - 2*pad + R)
  XAS_REQUIRE((s->Ho - 1) * s->stride - 2 * s->pad + s->R <= s->Hi && (s->Wo - 1) * s->stride - 2 * s->pad + s->S <= s->Wi,
              "conv_dgrad: Hi/Wi (%d,%d) too small for Ho/Wo (%d,%d)", s->Hi, s->Wi, s->Ho, s->Wo);
  {
    const long xi = (long)s->Hi * s->Wi * s->Cin, yi = (long)s->Ho * s->Wo * s->Cout;
    const int per = images_per_launch(s->N, yi, 0);
    if (per < s->N) {
      XAS_REQUIRE(!bnb, "conv_dgrad: the batch-norm epilogue needs a single launch");
      for (int n0 = 0; n0 < s->N; n0 += per) {
        xas_conv_shape part = *s;
        part.N = s->N - n0 < per ? s->N - n0 : per;
        const int rc = conv_dgrad_impl(dy + (size_t)n0 * yi, w_packed_t, dx + (size_t)n0 * xi, &part, stream, accumulate,
                                       acc_src ? acc_src + (size_t)n0 * xi : nullptr, acc_mask ? acc_mask + (size_t)n0 * xi / 4 : nullptr);
        if (rc) return rc;
      }
      return 0;
    }
  }
  hipStream_t st = as_stream(stream);
  if (s->Cin == 1 && thin_ok(s, s->Cout)) {          // dx[m] = sum dy[src] . wt[0][tap][:]
    const ThinParams tp = thin_params(s->N, s->Hi, s->Wi, s->Cout, s->pad, 1);
    const long M = (long)s->N * s->Hi * s->Wi;
    hipLaunchKernelGGL(thin_vec2scalar_kernel, dim3(thin_grid(M, s->Cout)), dim3(256), 0, st, dy, w_packed_t, nullptr, dx, tp);
    XAS_LAUNCH_CHECK();
    return 0;
  }
  if (s->Cout == 1 && thin_ok(s, s->Cin)) {          // dx[m][ci] = sum dy[src] * wt[ci][tap]
    const ThinParams tp = thin_params(s->N, s->Hi, s->Wi, s->Cin, s->pad, 1);
    const long M = (long)s->N * s->Hi * s->Wi;
    hipLaunchKernelGGL(thin_scalar2vec_kernel, dim3(thin_grid(M, s->Cin)), dim3(256), 0, st, dy,
                       w_packed_t, nullptr, dx, tp);
    XAS_LAUNCH_CHECK();
    return 0;
  }
  if (s->Cout % BK != 0 || s->Cin < 16) {
    const long total = (long)s->N * s->Hi * s->Wi * s->Cin;
    hipLaunchKernelGGL(direct_dgrad_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, st, dy, w_packed_t, dx, *s);
    XAS_LAUNCH_CHECK();
    return 0;
  }
  XAS_REQUIRE((((uintptr_t)dy | (uintptr_t)w_packed_t) & 15) == 0, "conv_dgrad: operands must be 16-byte aligned");
  IgemmParams p{};
  p.src = dy; p.wgt = w_packed_t; p.bias = nullptr; p.out = dx; p.N = s->N;
  p.a_amax = s->grad_amax;
  p.Hs = s->Ho; p.Ws = s->Wo; p.Cs = s->Cout; p.Hd = s->Hi; p.Wd = s->Wi; p.Cd = s->Cin;
  p.R = s->R; p.S = s->S; p.stride = s->stride; p.pad = s->pad; p.tune = g_tune;
  p.accumulate = accumulate; p.acc_src = acc_src; p.acc_mask = acc_mask;
  const int Hp = (s->Hi + s->stride - 1) / s->stride, Wp = (s->Wi + s->stride - 1) / s->stride;
  p.src_elems = (long)p.N * p.Hs * p.Ws * p.Cs; p.wgt_elems = (long)p.Cd * p.R * p.S * p.Cs;
  if (bnb) {
    p.bnb_x = bnb->bnb_x; p.bnb_mean = bnb->bnb_mean; p.bnb_var = bnb->bnb_var; p.bnb_gamma = bnb->bnb_gamma;
    p.bnb_beta = bnb->bnb_beta; p.bnb_eps = bnb->bnb_eps; p.bnb_rows_per_group = bnb->bnb_rows_per_group;
    p.bnb_partial = bnb->bnb_partial;
  }
  return dispatch_igemm<1>(p, precision_of(s), s->N * Hp * Wp, s->stride * s->stride, st);
}

// Rows per tile when the data gradient of `s` can carry the batch-norm backward reduction of the layer in front of the
// convolution in its epilogue (MFMA path, stride 1, one launch, full tiles that do not straddle a camera group), else 0.
static int dgrad_bnb_tile_rows(const xas_conv_shape* s, int groups) {
  if (!s || groups < 1 || s->stride != 1) return 0;
  if (s->Cout % BK != 0 || s->Cin < 16 || s->Cin % 4 != 0 || s->Cin == 1 || s->Cout == 1) return 0;
  if (images_per_launch(s->N, (long)s->Ho * s->Wo * s->Cout, 0) < s->N || s->N % groups) return 0;
  const long M = (long)s->N * s->Hi * s->Wi, Mg = M / groups;
  int bm, bn;
  pick_tile(s->Cin, M, 1, &bm, &bn);
  if (Mg % bm || s->Cin % bn) return 0;
  if ((M / bm) * 2 * (long)s->Cin * 4 >= 0x7fffff00l) return 0;
  return bm;
}

// Data gradient of a convolution whose input is h = relu(batch_norm(xb)) (rank-local statistics), followed by that batch
// norm's backward pass: -> dx = gradient wrt xb, sums [groups][2][Cin], and the local dbeta / dgamma added into the
// accumulators (may be NULL).  dz is scratch of xb's size (the gradient wrt the norm's pre-activation output).
// When the tile grid lines up with the groups, the ReLU mask and the two reductions of the norm's backward are done in
// the convolution's epilogue (dz is never re-read for them); otherwise xas_conv_dgrad + xas_bn_bwd_reduce.
extern "C" size_t xas_conv_dgrad_bn_bwd_workspace_floats(const xas_conv_shape* s, int groups) {
  if (!s || groups < 1) return 0;
  const long M = (long)s->N * s->Hi * s->Wi;
  const int bm = dgrad_bnb_tile_rows(s, groups);
  if (!bm) return xas_bn_workspace_floats(M, s->Cin, groups);
  const long rows = M / bm;
  return (size_t)rows * 2 * s->Cin + xas_bn_workspace_floats(rows, 2 * s->Cin, groups);
}

extern "C" int xas_conv_dgrad_bn_bwd(const float* dy, const float* w_packed_t, const xas_conv_shape* s, const float* xb,
                                     const float* mean, const float* var_biased, const float* gamma, const float* beta,
                                     float eps, int groups, double count, float* dz, float* dx, float* sums,
                                     float* workspace, float* dbeta_acc, float* dgamma_acc, void* stream) {
  XAS_REQUIRE(s && groups >= 1 && s->N % groups == 0, "conv_dgrad_bn_bwd: the batch does not split into %d groups", groups);
  XAS_REQUIRE(xb && mean && var_biased && gamma && beta && dz && dx && sums && workspace, "conv_dgrad_bn_bwd: null buffer");
  // this entry always takes the weight format of the shape WITHOUT operand maxima (bf16x6 planes in the split modes: the fused
  // epilogue exists for three planes only): a grad_amax that came with the shape is ignored, for both of its code paths
  xas_conv_shape local = *s;
  local.grad_amax = nullptr; local.x_amax = nullptr;
  s = &local;
  const long M = (long)s->N * s->Hi * s->Wi;
  const int C = s->Cin;
  const int bm = dgrad_bnb_tile_rows(s, groups);
  if (!bm) {
    int rc = conv_dgrad_impl(dy, w_packed_t, dz, s, stream, 0);
    if (rc) return rc;
    rc = xas_bn_bwd_reduce(xb, nullptr, dz, mean, var_biased, gamma, beta, eps, 1, M, C, groups, sums, workspace, dbeta_acc,
                           dgamma_acc, nullptr, stream);
    if (rc) return rc;
    return xas_bn_bwd_apply(xb, nullptr, dz, mean, var_biased, gamma, beta, sums, eps, 1, M, C, groups, count, dx, nullptr,
                            nullptr, stream);
  }
  const long rows = M / bm;
  IgemmParams b{};
  b.bnb_x = xb; b.bnb_mean = mean; b.bnb_var = var_biased; b.bnb_gamma = gamma; b.bnb_beta = beta; b.bnb_eps = eps;
  b.bnb_rows_per_group = (int)(M / groups); b.bnb_partial = workspace;
  int rc = conv_dgrad_impl(dy, w_packed_t, dz, s, stream, 0, nullptr, nullptr, &b);
  if (rc) return rc;
  rc = xas_bn_bwd_sums_from_partials(workspace, rows, C, groups, sums, workspace + (size_t)rows * 2 * C, dbeta_acc,
                                     dgamma_acc, stream);
  if (rc) return rc;
  // dz is already masked: the apply pass runs without an activation
  return xas_bn_bwd_apply(xb, nullptr, dz, mean, var_biased, gamma, beta, sums, eps, 0, M, C, groups, count, dx, nullptr,
                          nullptr, stream);
}

#ifndef XAS_SLAB_THREADS
#define XAS_SLAB_THREADS 256
#endif
static inline unsigned slab_threads() { return XAS_SLAB_THREADS; }

// does the weight gradient of this shape run on the bf16-split kernel (conv_x6.hip)?  32-bit byte offsets (tensors below
// 2 GiB), whole float4s inside one filter tap, at most one carry per coordinate in the per-thread pixel decode
static bool wgrad_on_x6(const xas_conv_shape* s, const float* x) {
  if (precision_of(s) == XAS_PREC_F32 || (g_tune & 128)) return false;
  const long xbytes = ((long)s->N * s->Hi * s->Wi + (long)s->pad * s->Wi + s->pad) * s->Cin * 4;
  const long dybytes = (long)s->N * s->Ho * s->Wo * s->Cout * 4;
  return s->Cin % 4 == 0 && s->Cout % 4 == 0 && s->Cout >= 16 && (x == nullptr || ((uintptr_t)x & 15) == 0) &&
         xbytes < 0x7fffff00l && dybytes < 0x7fffff00l && 31 / s->Wo + 1 <= s->Ho;
}

static void wgrad_plan(const xas_conv_shape* s, bool x6, int* bm, int* bn, int* splits, int* mps) {
  const long KK = (long)s->R * s->S * s->Cin, M = (long)s->N * s->Ho * s->Wo;
  if (x6) wgrad_x6_tile(s->Cout, KK, bm, bn);
  else {
    *bm = s->Cout >= 96 ? 128 : (s->Cout > 32 ? 64 : 32);
    *bn = KK <= 64 ? 64 : 128;
    if (*bm == 32) *bn = 128;
  }
  const long tiles = cdiv(s->Cout, *bm) * cdiv(KK, *bn);
  // ~4 blocks per CU in total, >= 8 K-steps (4 for the bf16-split kernel) per block.  These kernels run beside the main
  // chain of the step: shorter blocks hand the CUs over sooner, but every split costs a slab of dW to write and to reduce;
  // with the round-3 kernels the slab traffic weighs more (sweeps in the define below)
#ifndef XAS_WGRAD_XTARGET
#define XAS_WGRAD_XTARGET 1024          // blocks per launch the pixel splits aim for.  r03 sweeps (weight gradients on a side stream, sharing the CUs): 512 best.  r05, ONE stream (the kernel has the chip to itself): 384 +3.2 ms/step, 512 +0.7, 768 +0.6, 1024 best, 1536 / 2048 +0.7, 3072 +2.2 (in-box, interleaved, 3 rounds)
#endif
  const int xtarget = XAS_WGRAD_XTARGET;
  long sp = cdiv(x6 ? xtarget : 1024, tiles);
  const long maxsp = x6 ? (M / 128 > 0 ? M / 128 : 1) : (M / 256 > 0 ? M / 256 : 1);
  if (sp > maxsp) sp = maxsp;
  if (sp < 1) sp = 1;
  if (x6 && sp >= 8) sp -= sp % 8;                 // bf16-split kernel: whole pixel splits per XCD (wgrad_x6_kernel)
  else if (x6 || !(g_tune & 8192)) {               // XCD-grouped order: (splits x Cout tiles) groups, 8 XCDs -> keep them balanced
    const long nct = cdiv(s->Cout, *bm);
    while (sp > 1 && (sp * nct) % 8 != 0 && (sp * nct) > 8) --sp;
  }
  long per = cdiv(M, sp);
  per = cdiv(per, WBK) * WBK;
  *mps = (int)per;
  *splits = (int)cdiv(M, per);
}

constexpr int kCout1Chunk = 2048;

extern "C" size_t xas_conv_wgrad_workspace_floats(const xas_conv_shape* s) {
  if (!s) return 0;
  if ((s->Cout == 1 && thin_ok(s, s->Cin)) || (s->Cin == 1 && thin_ok(s, s->Cout))) {
    const long C = s->Cout == 1 ? s->Cin : s->Cout;
    return (size_t)(cdiv((long)s->N * s->Hi * s->Wi, kThinChunk) + 1) * C * 9;
  }
  if (s->Cout == 1) return (size_t)cdiv((long)s->N * s->Ho * s->Wo, kCout1Chunk) * s->R * s->S * s->Cin;
  if (stem_wgrad_ok(s)) return (size_t)2 * kStemWgradBlocks * ST_CO * ST_K;
  int bm, bn, sp, mps, sp2;
  wgrad_plan(s, false, &bm, &bn, &sp, &mps);       // the largest of the kernels' slab counts: the choice between them
  wgrad_plan(s, true, &bm, &bn, &sp2, &mps);       // also depends on the alignment of x, unknown here
  int sp3 = 0, pps = 0;
  if (!wgrad_x6t_plan(s->N, s->Hi, s->Wi, s->Cin, s->Cout, s->R, s->S, s->stride, s->pad, s->Ho, s->Wo, &bm, &sp3, &pps)) sp3 = 0;
  if (sp2 > sp) sp = sp2;
  if (sp3 > sp) sp = sp3;
  return (size_t)sp * s->Cout * s->R * s->S * s->Cin;
}

template <int BM, int BN, bool VEC>
static int launch_wgrad(const WgradParams& p, int splits, hipStream_t st) {
  const size_t lds = (size_t)2 * WBK * ((BM + 4) + (BN + 4)) * sizeof(float);
  static bool attr_set_dev[kMaxDevices] = {};          // per device: the LDS limit is a per-device function attribute
  bool& attr_set = attr_set_dev[current_device()];
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel<BM, BN, VEC>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  WgradParams q = p;
  q.nct = (int)cdiv(p.Cout, BM);
  q.ntiles = q.nct * (int)cdiv(p.KK, BN);
  q.nsplits = splits;
  q.tune = g_tune;
  dim3 grid((unsigned)(8 * cdiv(splits, 8) * q.ntiles));
  hipLaunchKernelGGL((wgrad_kernel<BM, BN, VEC>), grid, dim3(256), lds, st, q);
  XAS_LAUNCH_CHECK();
  return 0;
}

template <int BM, int BN, int T, bool PIPE = true>
static int launch_wgrad_buf_t(const WgradParams& p, int splits, hipStream_t st) {
  if (PIPE && (g_tune & 524288)) return launch_wgrad_buf_t<BM, BN, T, false>(p, splits, st);   // tune bit19: plain K-loop
  size_t lds = (size_t)2 * WBK * ((BM + 4) + (BN + 32)) * sizeof(float);
  static size_t attr_set_dev[kMaxDevices] = {};
  size_t& attr_set = attr_set_dev[current_device()];
  if (attr_set < lds) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_buf_kernel<BM, BN, T, PIPE>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = lds;
  }
  WgradParams q = p;
  q.nct = (int)cdiv(p.Cout, BM);
  q.ntiles = q.nct * (int)cdiv(p.KK, BN);
  q.nsplits = splits;
  q.tune = g_tune;
  dim3 grid((unsigned)((g_tune & 8192) ? splits * q.ntiles : 8 * cdiv((long)splits * q.nct, 8) * (q.ntiles / q.nct)));
  hipLaunchKernelGGL((wgrad_buf_kernel<BM, BN, T, PIPE>), grid, dim3(256), lds, st, q);
  XAS_LAUNCH_CHECK();
  return 0;
}

// T = number of filter taps a BN-column tile can span: the smallest T in {1, 2, 4} with Cin % (BN / T) == 0
template <int BM, int BN>
static int launch_wgrad_buf(const WgradParams& p, int splits, hipStream_t st) {
  if (p.Cin % BN == 0) return launch_wgrad_buf_t<BM, BN, 1>(p, splits, st);
  if (p.Cin % (BN / 2) == 0) return launch_wgrad_buf_t<BM, BN, 2>(p, splits, st);
  if constexpr (BN >= 128) {
    if (p.Cin % (BN / 4) == 0) return launch_wgrad_buf_t<BM, BN, 4>(p, splits, st);
  }
  set_error("wgrad_buf: Cin=%d does not divide into tap groups of a %d-column tile", p.Cin, BN);
  return 1;
}

static int conv_wgrad_impl(const float* x, const float* dy, float* dw_out, float* workspace, const xas_conv_shape* s,
                           void* stream, bool oihw, bool accumulate = false);

extern "C" int xas_conv_wgrad(const float* x, const float* dy, float* dw_packed, float* workspace,
                              const xas_conv_shape* s, void* stream) {
  return conv_wgrad_impl(x, dy, dw_packed, workspace, s, stream, false);
}

extern "C" int xas_conv_wgrad_oihw(const float* x, const float* dy, float* dw_oihw, float* workspace,
                                   const xas_conv_shape* s, void* stream) {
  return conv_wgrad_impl(x, dy, dw_oihw, workspace, s, stream, true);
}

extern "C" int xas_conv_wgrad_acc(const float* x, const float* dy, float* dw_oihw, float* workspace,
                                  const xas_conv_shape* s, void* stream) {
  return conv_wgrad_impl(x, dy, dw_oihw, workspace, s, stream, true, true);
}

static int conv_wgrad_impl(const float* x, const float* dy, float* dw_packed, float* workspace, const xas_conv_shape* s,
                           void* stream, bool oihw, bool accumulate) {
  const int rflag = (oihw ? 1 : 0) | (accumulate ? 2 : 0);
  if (check_shape(s, "conv_wgrad")) return 1;
  XAS_REQUIRE(x && dy && dw_packed && workspace, "conv_wgrad: null buffer");
  XAS_REQUIRE(s->Cout % 4 == 0 || s->Cout == 1, "conv_wgrad: Cout=%d must be a multiple of 4 (or 1)", s->Cout);
  {
    const long xi = (long)s->Hi * s->Wi * s->Cin, yi = (long)s->Ho * s->Wo * s->Cout;
    const int per = images_per_launch(s->N, xi, yi);
    if (per < s->N && (oihw || accumulate)) {            // the packed single-slab form (linear layers) is never this large
      for (int n0 = 0; n0 < s->N; n0 += per) {
        xas_conv_shape part = *s;
        part.N = s->N - n0 < per ? s->N - n0 : per;
        const int rc = conv_wgrad_impl(x + (size_t)n0 * xi, dy + (size_t)n0 * yi, dw_packed, workspace, &part, stream, oihw,
                                       accumulate || n0 > 0);
        if (rc) return rc;
      }
      return 0;
    }
  }
  hipStream_t st = as_stream(stream);
  if ((s->Cout == 1 && thin_ok(s, s->Cin)) || (s->Cin == 1 && thin_ok(s, s->Cout))) {
    // slabs [chunk][c][tap] == OIHW order for both cases ([1][Cin][3][3] resp. [Cout][1][3][3]); the packed
    // layouts are [1][tap][c] resp. [c][tap]
    const bool cout1 = s->Cout == 1;
    const int C = cout1 ? s->Cin : s->Cout;
    const ThinParams tp = thin_params(s->N, s->Hi, s->Wi, C, s->pad, 0);
    const long M = (long)s->N * s->Hi * s->Wi;
    // ~4 blocks per CU at most: long chunks amortise the block-level reduction (the workspace is sized for kThinChunk)
    int chunks = (int)cdiv(M, kThinChunk);
    if (chunks > 1024) chunks = 1024;
    const int per_chunk = (int)(cdiv(cdiv(M, chunks), 256) * 256);
    chunks = (int)cdiv(M, per_chunk);
    hipLaunchKernelGGL(thin_wgrad_kernel, dim3(chunks), dim3(256), 0, st, cout1 ? dy : x, cout1 ? x : dy, workspace, tp,
                       cout1 ? -1 : 1, per_chunk);
    XAS_LAUNCH_CHECK();
    const long n = (long)C * 9;
    if (oihw || !cout1) {
      hipLaunchKernelGGL(slab_reduce_kernel2, dim3((unsigned)cdiv(n, 64)), dim3(slab_threads()), 0, st, workspace, chunks, n,
                         accumulate ? 2 : 0, 1, 1, 1, dw_packed);
    } else {          // packed [1][tap][c] wanted: treat the [c][tap] sums as an "OIHW" with Cout=1 and repack
      hipLaunchKernelGGL(slab_reduce_kernel2, dim3((unsigned)cdiv(n, 64)), dim3(slab_threads()), 0, st, workspace, chunks, n, 0, 1, 1, 1,
                         workspace + (size_t)chunks * n);
      XAS_LAUNCH_CHECK();
      hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, workspace + (size_t)chunks * n,
                         dw_packed, 1, C, 3, 3, 0, 0);
    }
    XAS_LAUNCH_CHECK();
    return 0;
  }
  if (s->Cout == 1) {
    const int KK = s->R * s->S * s->Cin;
    const int chunks = (int)cdiv((long)s->N * s->Ho * s->Wo, kCout1Chunk);
    hipLaunchKernelGGL(wgrad_cout1_kernel, dim3((unsigned)cdiv(KK, 64), (unsigned)chunks), dim3(64), 0, st, x, dy,
                       workspace, *s, kCout1Chunk);
    XAS_LAUNCH_CHECK();
    hipLaunchKernelGGL(slab_reduce_kernel2, dim3((unsigned)cdiv(KK, 64)), dim3(slab_threads()), 0, st, workspace, chunks, (long)KK,
                       rflag, s->Cin, s->R, s->S, dw_packed);
    XAS_LAUNCH_CHECK();
    return 0;
  }
  if (stem_wgrad_ok(s) && !(g_tune & (1 << 24))) {    // (tune bit 24: the general kernel)
    const int patches = s->N * (s->Ho / ST_TH) * (s->Wo / ST_TW);
    const int ppb = (int)cdiv(patches, kStemWgradBlocks), blocks = (int)cdiv(patches, ppb);
    hipLaunchKernelGGL(stem_wgrad_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, dy, workspace, s->N, s->Hi, s->Wi,
                       s->Ho, s->Wo, patches, ppb);
    XAS_LAUNCH_CHECK();
    const long n = (long)ST_CO * ST_K;
    hipLaunchKernelGGL(slab_reduce_kernel2, dim3((unsigned)cdiv(n, 64)), dim3(slab_threads()), 0, st, workspace, 2 * blocks, n,
                       rflag, s->Cin, s->R, s->S, dw_packed);
    XAS_LAUNCH_CHECK();
    return 0;
  }
  const bool x6 = wgrad_on_x6(s, x);
  int bm, bn, splits, mps;
  wgrad_plan(s, x6, &bm, &bn, &splits, &mps);
  WgradParams p{};
  p.x = x; p.dy = dy; p.out = (splits == 1 && !oihw) ? dw_packed : workspace;
  // which argument is the gradient tensor: dy (a conv's weight gradient) or x (XAS_GRAD_IS_X: a ConvTranspose2d's)
  p.a_amax = (s->mode & XAS_GRAD_IS_X) ? s->x_amax : s->grad_amax;      // max |dy argument|
  p.b_amax = (s->mode & XAS_GRAD_IS_X) ? s->grad_amax : s->x_amax;      // max |x argument|
  p.N = s->N; p.Hi = s->Hi; p.Wi = s->Wi; p.Cin = s->Cin; p.Cout = s->Cout; p.R = s->R; p.S = s->S;
  p.stride = s->stride; p.pad = s->pad; p.Ho = s->Ho; p.Wo = s->Wo;
  p.KK = s->R * s->S * s->Cin; p.M = s->N * s->Ho * s->Wo; p.m_per_split = mps;
  p.div_hw.init((unsigned)(s->Ho * s->Wo)); p.div_w.init((unsigned)s->Wo);
  p.div_cin.init((unsigned)s->Cin); p.div_s.init((unsigned)s->S);
  XAS_REQUIRE(s->Cout % 4 == 0, "conv_wgrad: Cout=%d not supported by the MFMA path", s->Cout);
  const bool vec = (s->Cin % 4 == 0) && (((uintptr_t)x & 15) == 0);
  int rc;
  // buffer-load kernel: 32-bit byte offsets (tensors < 2 GiB) and at most one carry per coordinate in the per-thread
  // pixel decode (31 / Wo + 1 <= Ho); tune bit7 (128) forces the old kernel
  const long xbytes = ((long)s->N * s->Hi * s->Wi + (long)s->pad * s->Wi + s->pad) * s->Cin * 4;
  const long dybytes = (long)p.M * s->Cout * 4;
  const bool buf_ok = vec && s->Cin % 32 == 0 && xbytes < 0x7fffff00l && dybytes < 0x7fffff00l && 31 / s->Wo + 1 <= s->Ho && !(g_tune & 128);
  int tbm = 0, tsplits = 0, tpps = 0;
  if (x6 && !(g_tune & (1 << 22)) &&                  // tune bit 22: no tap re-use kernels
      wgrad_x6t_plan(s->N, s->Hi, s->Wi, s->Cin, s->Cout, s->R, s->S, s->stride, s->pad, s->Ho, s->Wo, &tbm, &tsplits, &tpps)) {
    splits = tsplits;
    p.out = (splits == 1 && !oihw) ? dw_packed : workspace;
    rc = launch_wgrad_x6t(p, tbm, tsplits, tpps, planes_of(precision_of(s), 2, has_amax_for(s, 2)), st);
  } else
  if (x6) rc = launch_wgrad_x6(p, bm, bn, splits, planes_of(precision_of(s), 2, has_amax_for(s, 2)), st);
  else if (buf_ok) {
    if (bm == 32) rc = launch_wgrad_buf<32, 128>(p, splits, st);
    else if (bn == 64) rc = bm == 128 ? launch_wgrad_buf<128, 64>(p, splits, st) : launch_wgrad_buf<64, 64>(p, splits, st);
    else if (bm == 128) rc = launch_wgrad_buf<128, 128>(p, splits, st);
    else rc = launch_wgrad_buf<64, 128>(p, splits, st);
  } else
  if (bm == 32) rc = vec ? launch_wgrad<32, 128, true>(p, splits, st) : launch_wgrad<32, 128, false>(p, splits, st);
  else if (bn == 64 && vec) rc = bm == 128 ? launch_wgrad<128, 64, true>(p, splits, st) : launch_wgrad<64, 64, true>(p, splits, st);
  else if (bm == 128) rc = vec ? launch_wgrad<128, 128, true>(p, splits, st) : launch_wgrad<128, 128, false>(p, splits, st);
  else rc = vec ? launch_wgrad<64, 128, true>(p, splits, st) : launch_wgrad<64, 128, false>(p, splits, st);
  if (rc) return rc;
  const long n = (long)s->Cout * p.KK;
  if (oihw || splits > 1) {
    hipLaunchKernelGGL(slab_reduce_kernel2, dim3((unsigned)cdiv(n, 64)), dim3(slab_threads()), 0, st, workspace, splits, n,
                       rflag, s->Cin, s->R, s->S, dw_packed);
    XAS_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int xas_pack_weight(const float* oihw, float* packed, int Cout, int Cin, int R, int S, int transposed,
                               void* stream) {
  XAS_REQUIRE(oihw && packed && Cout > 0 && Cin > 0 && R > 0 && S > 0, "pack_weight: bad arguments");
  const long total = (long)Cout * Cin * R * S;
  hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, as_stream(stream), oihw, packed,
                     Cout, Cin, R, S, transposed, 0);
  XAS_LAUNCH_CHECK();
  return 0;
}

extern "C" int xas_unpack_weight(const float* packed, float* oihw, int Cout, int Cin, int R, int S, int transposed,
                                 void* stream) {
  XAS_REQUIRE(oihw && packed && Cout > 0 && Cin > 0 && R > 0 && S > 0, "unpack_weight: bad arguments");
  const long total = (long)Cout * Cin * R * S;
  hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, as_stream(stream), packed, oihw,
                     Cout, Cin, R, S, transposed, 1);
  XAS_LAUNCH_CHECK();
  return 0;
}
