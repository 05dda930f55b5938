// fp32-ACCURATE convolutions on the 16-bit matrix pipe of gfx950: forward, data gradient (== ConvTranspose2d forward) and
// weight gradient, in two piece formats (xas_hip.h: XAS_PREC_F16X3, the library default, and XAS_PREC_BF16X6).
//
// bf16x6 (P = 3 planes).  Every fp32 operand x is split EXACTLY into three bf16 pieces x = x1 + x2 + x3 (x1 = bf16(x),
// x2 = bf16(x - x1), x3 = bf16(x - x1 - x2): 3 x 8 significant bits = the 24 of fp32).  Of the nine partial products of
// x * y the six  x3 y1, x1 y3, x2 y2, x2 y1, x1 y2, x1 y1  are kept; each is exact in fp32 (8 x 8 bits) and the MFMA
// accumulates them in fp32, smallest first; the three dropped ones are below 2^-25 |x y|.  Measured against float64 (r02,
// DESIGN 7b): 1.09e-7 relative at K = 64 (exact-fp32 MFMA: 1.06e-7), 1.0e-6 at K = 4608 (8.1e-7).  6 x 32 cycles per
// K = 16 of v_mfma_f32_32x32x16_bf16 against 8 x 64 cycles of v_mfma_f32_32x32x2_f32: peak 2.5 PFLOP/s / 6 = 419 TFLOP/s
// of fp32-equivalent work.  No assumption on operand ranges (bf16 has fp32's exponent).
//
// f16x3 (P = 2 planes, r03).  fp16 carries 11 significant bits: TWO pieces h1 = fp16(s x), h2 = fp16(s x - h1) leave
// |s x - h1 - h2| <= 2^-22 |s x|, and THREE partial products h2 g1, h1 g2, h1 g1 (each exact in fp32) per product: half
// the matrix instructions, two thirds of the LDS traffic and of the conversion work of bf16x6.  The representation
// error (7e-8 relative on a dot product) stays under the fp32 accumulation error every MFMA mode shares (1-3e-7): the
// distance to float64 is the exact-fp32 path's.  fp16's narrow exponent is handled by EXACT power-of-two scales s, undone
// on the fp32 accumulators: weights 2^10 (pre-split once per optimizer step; |w| < 64, flagged otherwise), and EVERY tensor
// operand - activations and gradients alike (r04: no fixed activation scale, hence no activation range) - by the power of
// two that puts its MAXIMUM in [2^14, 2^15): the kernel that writes a tensor (batch-norm apply / backward, soft-argmax
// backward; xas_abs_max for anything else) merges max |v| into a device float, the consumer reads it through
// xas_conv_shape.grad_amax / x_amax (IgemmParams::a_amax / WgradParams::a_amax, b_amax).  Elements 2^-15 and more below the
// maximum lose relative (not absolute) accuracy: error floor 2^-39 max|v| per element, invisible next to the fp32
// rounding of the large terms of the same sum.  A launch without the maxima of its operands runs as bf16x6.  PIECES = 1 is the
// plain bf16 variant (operands rounded once; NOT fp32 accurate; XAS_PREC_BF16), kept as the reported-separately variant.
//
// Kernel structure (both kernels): 256 threads = 4 waves, tile BM x BN, K-step of 32 (channels of a tap / pixels) loaded
// two K-steps ahead into two register sets by buffer loads (hardware zero-fill for padding and ragged edges), processed as
// two HALF-steps of 16: LDS holds two half-buffers of three bf16 planes for the operands that are split in the kernel
// (24.6 KB for 128 activation rows; the epilogue staging, 36.9 KB, is the footprint: three blocks per CU), and during the
// 24 MFMAs of half-step t the wave splits and stores half-step t + 1 into the other buffer - ONE barrier per 24 MFMAs, the
// conversion work (5.5 vector-ALU instructions per element) issues in the shadow of the matrix pipe, the other blocks
// on the CU cover the barrier.  Weights of forward / data gradient arrive PRE-SPLIT in MFMA fragment order
// (xas_split_weight, once per optimizer step) and go straight to operand registers: no LDS traffic on that operand.
// What bounds the forward kernel (occupancy-equalised timing ablations, profiles/r03_igemm_x6_v3_ablations.txt):
// activation loads +13..18 %, the plane stores ~+10 %, fragment reads ~+10 %, weight-fragment loads +7..9 %, the
// conversion work +3..7 % (doubling it: -15 %); MFMA-busy 55-70 % by counter at 1.9-2.1 GHz under load.
//
// Replaces the cuDNN kernels behind integral_base_modules/resnet.py:16-47, deconv_head.py:24-35,
// physique_network.py:15-50 and torchvision's Bottleneck, like conv.hip.
#include "conv_shared.h"

namespace xas {

typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef short s16x8_t __attribute__((ext_vector_type(8)));

#ifndef XAS_X6_ABL
#define XAS_X6_ABL 0               // timing ablations of igemm_x6_kernel (results wrong): tools/gpu/r3_abl3.sh
#endif
constexpr int XH = 16;             // k per half-step
constexpr int XLDH = XH;           // igemm LDS row in halfwords: 32 B, unpadded.  The two 16-byte halves of rows 8..15 (mod 16) are
                                   // swapped (xswz): the 8-byte stores of a half-wave (8 rows x 32 B) and the 16-byte fragment reads
                                   // of 16 lanes (16 rows x 16 B) each cover the 64 banks exactly once.  (Rows padded to 48 B read
                                   // conflict-free too but their stores collided two-way on half the banks: PMC 0.4 conflict cycles
                                   // per LDS instruction.)
__device__ __forceinline__ int xswz(int row, int half) { return (half ^ ((row >> 3) & 1)) << 3; }   // halfword offset of a 16-byte half

// (the kept partial products of a piece format, smallest first: Products<P> in conv_shared.h)

// ------------------------------------------------------------------------------------
// weights: fp32 packed [rows][K] -> pre-split bf16 in FRAGMENT order
//     [rows / 32 blocks][K / 16 half-chunks][P planes][64 lanes][8 bf16],  lane = 32 * (k half) + (row in block)
// i.e. every (row block, 16-deep k slice, plane) is the 1 KiB image of one MFMA operand register set: a wave fetches it
// with ONE fully coalesced 16-byte-per-lane load straight into the operand registers - the weights never pass through
// LDS (rows beyond the matrix are zero).
// ------------------------------------------------------------------------------------
// f16x3 weights are split as 2^10 w: a weight of magnitude 64 or more leaves fp16's range.  The preparation kernels raise
// this flag instead of failing silently into inf / NaN results; xas_f16_weight_overflow reads (and clears) it.
__device__ unsigned g_f16_weight_overflow = 0u;
__device__ __forceinline__ void f16_weight_check(float4 a, float4 b) {
  const float m = fmaxf(fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))),
                        fmaxf(fmaxf(fabsf(b.x), fabsf(b.y)), fmaxf(fabsf(b.z), fabsf(b.w))));
  if (!(m < 65504.f)) atomicOr(&g_f16_weight_overflow, 1u);        // (NaN weights raise it too)
}

template <int P>
__global__ __launch_bounds__(256) void split_weight_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst,
                                                           int rows, int K, long total /* 8-element pieces */) {
  const long id = (long)blockIdx.x * 256 + threadIdx.x;
  if (id >= total) return;
  const int lane = (int)(id & 63);
  const long blk = id >> 6;                                   // (row block, half-chunk)
  const int nhc = K / 16;
  const int hc = (int)(blk % nhc), rb = (int)(blk / nhc);
  const int row = rb * 32 + (lane & 31), k0 = hc * 16 + (lane >> 5) * 8;
  float4 r0 = make_float4(0, 0, 0, 0), r1 = r0;
  if (row < rows) {
    r0 = *reinterpret_cast<const float4*>(src + (size_t)row * K + k0);
    r1 = *reinterpret_cast<const float4*>(src + (size_t)row * K + k0 + 4);
  }
  unsigned short* o = dst + (blk * P) * 512 + lane * 8;
  if (P == 2) {                                              // fp16 pieces of 2^10 w (exact scaling)
    r0.x *= kF16WScale; r0.y *= kF16WScale; r0.z *= kF16WScale; r0.w *= kF16WScale;
    r1.x *= kF16WScale; r1.y *= kF16WScale; r1.z *= kF16WScale; r1.w *= kF16WScale;
    f16_weight_check(r0, r1);
  }
#pragma unroll
  for (int pc = 0; pc < P; ++pc) {
    const uint2 q0 = pack_piece4<P>(r0), q1 = pack_piece4<P>(r1);
    *reinterpret_cast<uint4*>(o + pc * 512) = make_uint4(q0.x, q0.y, q1.x, q1.y);
    if (pc + 1 < P) { r0 = sub_piece4<P>(r0, q0); r1 = sub_piece4<P>(r1, q1); }
  }
}

// ------------------------------------------------------------------------------------
// forward (MODE 0) and data gradient (MODE 1)
//
// Activations: buffer loads (two K-steps ahead, two register sets) -> split -> LDS half-buffers [2][P][BM][16 + 8 pad] ->
// ds_read_b128 fragments.  Weights: pre-split, fragment-ordered (split_weight_kernel): each wave loads the fragments of its
// own column blocks straight into MFMA operand registers, one half-step ahead, 1 KiB coalesced per instruction - no LDS
// traffic, no conversion work, no LDS stores for that operand (the VGPR -> LDS store path, ~80 B/clk per CU, is what
// bounded the first version of this kernel: 24 KB of stores per half-step and block).
// ------------------------------------------------------------------------------------
#ifndef XAS_X6_WAVES2
#define XAS_X6_WAVES2 2            // waves per SIMD the two-piece (f16x3) builds of igemm_x6_kernel / wgrad_x6_kernel are compiled for
#endif
#ifndef XAS_X6_WAVES2_WIDE
#define XAS_X6_WAVES2_WIDE 2       // the same for the 64 x 256 tile (132 registers: 4 would cap it at 128)
#endif
#ifndef XAS_WX6_WAVES2
#define XAS_WX6_WAVES2 2
#endif
template <int BM, int BN, int MODE, int P, bool BNB = false>
__global__ __launch_bounds__(256, (P == 2 ? (BM == 64 && BN == 256 ? XAS_X6_WAVES2_WIDE : XAS_X6_WAVES2) : 2)) void igemm_x6_kernel(IgemmParams p) {
  using C = TileCfg<BM, BN>;
  constexpr int PLANE = BM * XLDH;              // halfwords
  constexpr int HBUF = P * PLANE;               // halfwords per half-buffer
  constexpr int AP = BM / 64;                   // activation float4 per thread per half-step (4 threads x 16 B per row)
  constexpr int NT = Products<P>::N;            // partial products
  constexpr int TILES = C::MI * C::NI, NMF = NT * TILES;
  extern __shared__ __align__(16) float lds[];
  unsigned short* S = reinterpret_cast<unsigned short*>(lds);     // [2][P][BM][XLDH]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // provably wave-uniform: the weight-fragment offsets stay scalar
  const int wm = wave / C::WAVES_N, wn = wave % C::WAVES_N;
  const int kq4 = tid & 3, arow = tid >> 2;

  int Hrow = p.Hrow, Wrow = p.Wrow;
  int ph = 0, pw = 0, base_r = 0, base_s = 0, nr = p.R, ns = p.S, off_h = -p.pad, off_w = -p.pad, rstep = 1;
  int sa = p.stride;
  if (MODE == 1) {                                   // one stride phase per blockIdx.z: no MFMA on structurally-zero taps
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
  // XCD-aware tile order: consecutive block ids go round the 8 XCDs; XCD (xi, xj) of the (8 / xn) x xn grid owns a
  // contiguous range of M-tiles and of N-tiles, the N-tiles of one M-tile adjacent in its queue (they share the
  // activation rows in that XCD's L2); launch_igemm_x6_t picks xn so that the weight slice of an XCD stays L2-resident
  const int xcd = blockIdx.x & 7, qb = blockIdx.x >> 3;
  const int xi = xcd / p.xn, xj = xcd - xi * p.xn;
  const int ml = qb / p.nt_per_x;
  const int mt = xi * p.mt_per_xcd + ml;
  const int nt = xj * p.nt_per_x + (qb - ml * p.nt_per_x);
  if (mt >= p.nMt || nt >= p.nNt) return;
  const int m0 = mt * BM, n0 = nt * BN;
  if (m0 >= Mrows) return;
  // These kernels sit on the critical chain of the step; the weight-gradient kernels of the side stream co-reside on the
  // CUs (footprints are sized for it) and only exist to fill what this chain leaves idle: win the issue arbitration.
#ifndef XAS_X6_PRIO
#define XAS_X6_PRIO 3
#endif
  __builtin_amdgcn_s_setprio(XAS_X6_PRIO);
  const int HW = Hrow * Wrow;
  const int cchunks = p.Cs / BK;
  const int nk = nr * ns * cchunks;
  float f16_sa = kF16AScale, f16_desc = kF16Descale;   // P == 2: scale of the gathered operand, scale of the result
  if (P == 2 && p.a_amax) { float inv; f16_sa = f16_grad_scale(p.a_amax, &inv); f16_desc = inv * (1.f / kF16WScale); }

  // activation operand: buffer base moved down so that per-lane and scalar parts are non-negative (see igemm_buf_kernel)
  const long dmin = MODE == 0 ? 0l : -((long)(nr - 1) * p.Ws + (ns - 1)) * p.Cs;
  const long rmin = MODE == 0 ? -((long)p.pad * p.Ws + p.pad) * p.Cs : 0l;
  const long bias = -(rmin + dmin);
  const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.src) - bias, 0, (int)((bias + p.src_elems) * 4), 0x00020000);
  unsigned voffA[AP], maskA[AP];
#pragma unroll
  for (int j = 0; j < AP; ++j) {
    const int m = m0 + arow + 64 * j;
    voffA[j] = kOOB; maskA[j] = 0u;
    if (m < Mrows) {
      int n, a, b;
      if (MODE == 0) { n = p.div_hw.div(m); const int rem = m - n * HW; a = p.div_w.div(rem); b = rem - a * Wrow; }
      else { n = m / HW; const int rem = m - n * HW; a = rem / Wrow; b = rem - a * Wrow; }
      const int ra = a * sa + off_h, rb = b * sa + off_w;
      const long rbase = (((long)n * p.Hs + ra) * p.Ws + rb) * p.Cs;
      voffA[j] = (unsigned)((rbase + dmin + bias + kq4 * 4) * 4);
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
  // weight operand: [row blocks of 32][K/16][P][64 lanes][8]: the wave's NI column blocks, all address arithmetic scalar
  const int Ktot = p.R * p.S * p.Cs;
  const unsigned blk_bytes = (unsigned)(Ktot / 16) * P * 1024u;           // one 32-row block over the whole K
  const long wbytes = (long)((p.Cd + 31) / 32) * blk_bytes;
  const __amdgpu_buffer_rsrc_t rsrcB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wgt), 0, (int)wbytes, 0x00020000);
  const unsigned voffB = (unsigned)lane * 16u;
  const unsigned nblk0 = (unsigned)(n0 / 32 + wn * (C::WN / 32));

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

  // K-step streams, advanced with scalar ALU only; both stop at the last step (further calls re-load it: valid addresses,
  // values never consumed).  Stream A feeds the activation registers (two K-steps ahead), stream B the weight fragments.
  struct Cur { int chunk, js, jr, left; };
  Cur ca{0, 0, 0, nk}, cb{0, 0, 0, nk};
  auto advance = [&](Cur& c) {
    const bool more = c.left > 1;
    c.left -= more ? 1 : 0;
    // taps innermost, channel chunks outermost: the nine taps of a 3x3 filter re-read the same 128-byte lines of the
    // block's pixels in CONSECUTIVE K-steps (L2 hits); with the chunks innermost the re-use distance was a whole channel
    // sweep of every resident block, beyond the 4 MB L2 of an XCD (PMC: 4.1x -> the algorithmic bytes on 128x32x32x128)
    int s2 = c.js + 1, r = c.jr, ch = c.chunk;
    if (s2 == ns) { s2 = 0; ++r; }
    if (r == nr) { r = 0; ++ch; }
    c.chunk = more ? ch : c.chunk; c.js = more ? s2 : c.js; c.jr = more ? r : c.jr;
  };
  float4 ra_0[2 * AP], ra_1[2 * AP];
  auto load_a = [&](float4 (&ra)[2 * AP]) {
    const int tap = ca.jr * ns + ca.js;
    const int rel = MODE == 0 ? (ca.jr * p.Ws + ca.js) : ((nr - 1 - ca.jr) * p.Ws + (ns - 1 - ca.js));
    const unsigned soffA = (unsigned)(rel * p.Cs + ca.chunk * BK) * 4u;
#pragma unroll
    for (int j = 0; j < AP; ++j) {
      const unsigned off = ((maskA[j] >> tap) & 1u) ? voffA[j] : kOOB;
      ra[j] = buf_load16(rsrcA, off, soffA);
      ra[AP + j] = buf_load16(rsrcA, off, soffA + XH * 4u);
    }
    advance(ca);
  };
  // weight fragments of one half-step (h = 0: first 16 k of the K-step at the cursor; h = 1: second 16, then advance)
  auto load_b = [&](uint4 (&gb)[P][C::NI], int h) {
    const int wtap = (base_r + rstep * cb.jr) * p.S + (base_s + rstep * cb.js);
    const unsigned hc = (unsigned)((wtap * p.Cs + cb.chunk * BK) / 16 + h);
#pragma unroll
    for (int ni = 0; ni < C::NI; ++ni)
#pragma unroll
      for (int pc = 0; pc < P; ++pc)
        gb[pc][ni] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(
            rsrcB, (int)voffB, (int)((nblk0 + ni) * blk_bytes + (hc * P + pc) * 1024u), 0));
    if (h) advance(cb);
  };
  auto split_store = [&](int buf, int h, const float4 (&ra)[2 * AP]) {
    unsigned short* sb = S + buf * HBUF;
#pragma unroll
    for (int j = 0; j < AP; ++j) {
      float4 r = ra[h * AP + j];
#if !(XAS_X6_ABL & (32 | 64))
      const Pieces<P> pcs = split_pieces<P>(r, f16_sa);       // (P = 2: both fp16 pieces in eight instructions, conv_shared.h)
#pragma unroll
      for (int pc = 0; pc < P; ++pc)
        *reinterpret_cast<uint2*>(sb + pc * PLANE + (arow + 64 * j) * XLDH + xswz(arow, kq4 >> 1) + (kq4 & 1) * 4) = pcs.q[pc];
#else
      if (P == 2) { r.x *= f16_sa; r.y *= f16_sa; r.z *= f16_sa; r.w *= f16_sa; }
#pragma unroll
      for (int pc = 0; pc < P; ++pc) {
        uint2 q = (XAS_X6_ABL & 32) ? make_uint2(__float_as_uint(r.x) + pc, __float_as_uint(r.z)) : pack_piece4<P>(r);
#if XAS_X6_ABL & 64
        {                                              // twice the conversion work, same stores
          const unsigned z = (unsigned)p.tune & 0x40000000u;          // runtime zero
          float4 r2 = make_float4(r.x * 1.0000002f, r.y * 1.0000002f, r.z * 1.0000002f, r.w * 1.0000002f);
          const uint2 q2 = pack_bf16x4(r2);
          const float4 r3 = sub_bf16x4(r2, q2);
          q.x |= (q2.x ^ __float_as_uint(r3.x) ^ __float_as_uint(r3.y)) & z; q.y |= (q2.y ^ __float_as_uint(r3.z) ^ __float_as_uint(r3.w)) & z;
        }
#endif
        *reinterpret_cast<uint2*>(sb + pc * PLANE + (arow + 64 * j) * XLDH + xswz(arow, kq4 >> 1) + (kq4 & 1) * 4) = q;
        if (pc + 1 < P && !(XAS_X6_ABL & 32)) r = sub_piece4<P>(r, q);
      }
#endif
    }
  };
  const int i = lane & 31, hh = lane >> 5;
#if XAS_X6_ABL & 16
  uint4 fa[P][C::MI];
#pragma unroll
  for (int pc = P - 1; pc >= 0; --pc)
#pragma unroll
    for (int mi = 0; mi < C::MI; ++mi)
      fa[pc][mi] = *reinterpret_cast<const uint4*>(S + pc * PLANE + (wm * C::WM + mi * 32 + i) * XLDH + xswz(i, hh));
#endif
  auto compute = [&](int buf, const uint4 (&gb)[P][C::NI]) {
    const unsigned short* sb = S + buf * HBUF;
#if !(XAS_X6_ABL & 16)
    uint4 fa[P][C::MI];
#pragma unroll
    for (int pc = P - 1; pc >= 0; --pc)              // smallest pieces first: they feed the first products
#pragma unroll
      for (int mi = 0; mi < C::MI; ++mi)
        fa[pc][mi] = *reinterpret_cast<const uint4*>(sb + pc * PLANE + (wm * C::WM + mi * 32 + i) * XLDH + xswz(i, hh));
#else
    (void)sb;
#endif
#ifdef XAS_FRAG_AGPR
    uint4 gq[P][C::NI];
#pragma unroll
    for (int pc = 0; pc < P; ++pc)
#pragma unroll
      for (int ni = 0; ni < C::NI; ++ni) gq[pc][ni] = frag_reg(gb[pc][ni]);
    frag_regs(fa);
#else
    const uint4 (&gq)[P][C::NI] = gb;
#endif
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int mi = 0; mi < C::MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < C::NI; ++ni)
          acc[mi][ni] = mfma_piece<P>(gq[Products<P>::B[t]][ni], fa[Products<P>::A[t]][mi], acc[mi][ni]);
  };
  (void)NMF;
#define SYNC() do { if (!(XAS_X6_ABL & 8)) __syncthreads(); } while (0)
#define LOADB(g, h) do { if (!(XAS_X6_ABL & 1)) load_b(g, h); } while (0)
#define LOADA(r) do { if (!(XAS_X6_ABL & 2)) load_a(r); } while (0)
#define STORE(b, h, r) do { if (!(XAS_X6_ABL & 4)) split_store(b, h, r); } while (0)
  uint4 gb_0[P][C::NI], gb_1[P][C::NI];              // weight fragments of even / odd half-steps
  if (nk > 0) {
    load_a(ra_0);                                      // K-step 0
    load_a(ra_1);                                      // K-step 1 (re-loads the last step when there is none)
    load_b(gb_0, 0);                                   // half-step 0
    load_b(gb_1, 1);                                   // half-step 1
    split_store(0, 0, ra_0);
    int ks = 0;
    for (; ks + 1 < nk; ks += 2) {
      SYNC();
      compute(0, gb_0);                                // half-step 2 ks
      LOADB(gb_0, 0);                                 // weights of half-step 2 ks + 2
      STORE(1, 1, ra_0);
      SYNC();
      LOADA(ra_0);                                    // activations of K-step ks + 2: set 0 is free
      compute(1, gb_1);                                // half-step 2 ks + 1
      LOADB(gb_1, 1);                                 // weights of half-step 2 ks + 3
      STORE(0, 0, ra_1);
      SYNC();
      compute(0, gb_0);                                // half-step 2 ks + 2
      LOADB(gb_0, 0);                                 // weights of half-step 2 ks + 4
      STORE(1, 1, ra_1);
      SYNC();
      LOADA(ra_1);                                    // K-step ks + 3
      compute(1, gb_1);                                // half-step 2 ks + 3
      LOADB(gb_1, 1);                                 // weights of half-step 2 ks + 5
      STORE(0, 0, ra_0);                         // K-step ks + 2, first half (never computed when ks + 2 == nk)
    }
    if (ks < nk) {                                     // one K-step left: in set 0, its first half is in buffer 0
      __syncthreads();
      compute(0, gb_0);
      split_store(1, 1, ra_0);
      __syncthreads();
      compute(1, gb_1);
    }
  }
  if constexpr (P == 2) {                              // the operands were split as 2^10 w and 2^4 x
#pragma unroll
    for (int mi = 0; mi < C::MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < C::NI; ++ni)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[mi][ni][e] *= f16_desc;
  }
  igemm_epilogue<BM, BN, MODE, BNB, (BN >= 128 ? BN / 64 : 1)>(p, acc, acc2, m0, n0, wm, wn, lane, Mrows, HW, Wrow, ph, pw, lds);
#ifdef XAS_DRAIN_AT_END
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
}

// ------------------------------------------------------------------------------------
// Persistent variant of the 64 x 256 tile for PURE GEMMs (1x1, stride 1, no padding; f16x3): igemm_x6p_kernel.
//
// The expanding 1x1 layers of the bottlenecks (P -> 4P forward, and the accumulate-and-mask data gradient of the contracting
// ones) have K = 64 ... 512: a 64 x 256 tile is two to sixteen K-steps and a 64 KB epilogue.  With one tile per block the
// activation loads of a tile are only in flight during its short K loop - 3 blocks x 16 KB per CU at best, far from the
// ~40 KB per CU that 5 TB/s at 2 us of latency needs - and these launches ran at 3.0-3.7 TB/s.  Here a block walks over
// `tpb` consecutive M-tiles of one N-tile and its load streams simply continue into the next tile: the last two K-steps of
// tile t request the ACTIVATIONS of the first two K-steps of tile t + 1 (all of it for K = 64), which then fly during the
// epilogue of t.
// Same LDS layout, fragment reads, MFMA order and epilogue (igemm_epilogue) as igemm_x6_kernel<64, 256, MODE, 2>: results
// are bit-identical to that kernel's.
// ------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256, 3) void igemm_x6p_kernel(IgemmParams p) {
  constexpr int BM = 64, BN = 256, P = 2;
  using C = TileCfg<BM, BN>;
  static_assert(C::WAVES_M == 1 && C::MI == 2 && C::NI == 2, "written for four waves side by side");
  constexpr int PLANE = BM * XLDH, HBUF = P * PLANE, NT = Products<P>::N;
  extern __shared__ __align__(16) float lds[];
  unsigned short* S = reinterpret_cast<unsigned short*>(lds);     // [2][P][BM][XLDH]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kq4 = tid & 3, arow = tid >> 2;
  const int Hrow = MODE == 0 ? p.Hrow : p.Hd, Wrow = MODE == 0 ? p.Wrow : p.Wd;
  const int HW = Hrow * Wrow, Mrows = p.N * HW;
  // block -> (group of tpb M-tiles, N-tile): the XCD-aware order of igemm_x6_kernel with the group as the unit
  const int xcd = blockIdx.x & 7, qb = blockIdx.x >> 3;
  const int xi = xcd / p.xn, xj = xcd - xi * p.xn;
  const int ml = qb / p.nt_per_x;
  const int mt0 = (xi * p.mt_per_xcd + ml) * p.tpb;
  const int nt = xj * p.nt_per_x + (qb - ml * p.nt_per_x);
  if (mt0 >= p.nMt || nt >= p.nNt) return;
  const int ntl = min(p.tpb, p.nMt - mt0);
  const int n0 = nt * BN;
  const int nk = p.Cs / BK;                              // even (launcher)
  float f16_sa = kF16AScale, f16_desc = kF16Descale;
  if (p.a_amax) { float inv; f16_sa = f16_grad_scale(p.a_amax, &inv); f16_desc = inv * (1.f / kF16WScale); }

  const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.src), 0, (int)(p.src_elems * 4), 0x00020000);
  auto voff_of = [&](int t) -> unsigned {
    const int m = (mt0 + t) * BM + arow;
    return (t < ntl && m < Mrows) ? (unsigned)(((long)m * p.Cs + kq4 * 4) * 4) : kOOB;
  };
  const unsigned blk_bytes = (unsigned)(p.Cs / 16) * P * 1024u;           // one 32-row block of the weights over the whole K
  const long wbytes = (long)((p.Cd + 31) / 32) * blk_bytes;
  const __amdgpu_buffer_rsrc_t rsrcB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wgt), 0, (int)wbytes, 0x00020000);
  const unsigned voffB = (unsigned)lane * 16u;
  const unsigned nblk0 = (unsigned)(n0 / 32 + wn * (C::WN / 32));

  // the two load streams: they run THROUGH the tiles of the block (activations: tile after tile; weights: the same N-tile again)
  int a_t = 0, a_c = 0, b_c = 0;
  unsigned voffA = voff_of(0);
  float4 ra_0[2], ra_1[2];
  auto load_a = [&](float4 (&ra)[2]) {
    const unsigned soff = (unsigned)(a_c * BK) * 4u;
    ra[0] = buf_load16(rsrcA, voffA, soff);
    ra[1] = buf_load16(rsrcA, voffA, soff + XH * 4u);
    if (++a_c == nk) { a_c = 0; ++a_t; voffA = voff_of(a_t); }
  };
  auto load_b = [&](uint4 (&gb)[P][C::NI], int h) {
    const unsigned hc = (unsigned)(b_c * (BK / 16) + h);
#pragma unroll
    for (int ni = 0; ni < C::NI; ++ni)
#pragma unroll
      for (int pc = 0; pc < P; ++pc)
        gb[pc][ni] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(
            rsrcB, (int)voffB, (int)((nblk0 + ni) * blk_bytes + (hc * P + pc) * 1024u), 0));
    if (h) { if (++b_c == nk) b_c = 0; }
  };
  auto split_store = [&](int buf, int h, const float4 (&ra)[2]) {
    const Pieces<P> pcs = split_pieces<P>(ra[h], f16_sa);
#pragma unroll
    for (int pc = 0; pc < P; ++pc)
      *reinterpret_cast<uint2*>(S + buf * HBUF + pc * PLANE + arow * XLDH + xswz(arow, kq4 >> 1) + (kq4 & 1) * 4) = pcs.q[pc];
  };
  const int i = lane & 31, hh = lane >> 5;
  f32x16 acc[C::MI][C::NI];
  auto compute = [&](int buf, const uint4 (&gb)[P][C::NI]) {
    const unsigned short* sb = S + buf * HBUF;
    uint4 fa[P][C::MI];
#pragma unroll
    for (int pc = P - 1; pc >= 0; --pc)
#pragma unroll
      for (int mi = 0; mi < C::MI; ++mi)
        fa[pc][mi] = *reinterpret_cast<const uint4*>(sb + pc * PLANE + (mi * 32 + i) * XLDH + xswz(i, hh));
#ifdef XAS_FRAG_AGPR
    uint4 gq[P][C::NI];
#pragma unroll
    for (int pc = 0; pc < P; ++pc)
#pragma unroll
      for (int ni = 0; ni < C::NI; ++ni) gq[pc][ni] = frag_reg(gb[pc][ni]);
    frag_regs(fa);
#else
    const uint4 (&gq)[P][C::NI] = gb;
#endif
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int mi = 0; mi < C::MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < C::NI; ++ni)
          acc[mi][ni] = mfma_piece<P>(gq[Products<P>::B[t]][ni], fa[Products<P>::A[t]][mi], acc[mi][ni]);
  };
  uint4 gb_0[P][C::NI], gb_1[P][C::NI];
  load_a(ra_0);                                          // tile 0: K-steps 0 and 1
  load_a(ra_1);
  for (int t = 0; t < ntl; ++t) {
    // the weight fragments are NOT carried across the epilogue (32 registers: the third block per CU is worth more than the
    // ~0.5 us of L2 latency this exposes at the head of a tile, which the other blocks on the CU cover)
    b_c = 0;
    load_b(gb_0, 0);
    load_b(gb_1, 1);
#pragma unroll
    for (int mi = 0; mi < C::MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < C::NI; ++ni)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;
    split_store(0, 0, ra_0);                             // (the staging of the previous epilogue is done: barrier below)
    for (int ks = 0; ks < nk; ks += 2) {
      __syncthreads();
      compute(0, gb_0);                                  // half-step 2 ks
      load_b(gb_0, 0);
      split_store(1, 1, ra_0);
      __syncthreads();
      load_a(ra_0);                                      // K-step ks + 2 - of the NEXT tile when this one has none left
      compute(1, gb_1);
      load_b(gb_1, 1);
      split_store(0, 0, ra_1);
      __syncthreads();
      compute(0, gb_0);
      if (ks + 2 < nk) load_b(gb_0, 0);
      split_store(1, 1, ra_1);
      __syncthreads();
      load_a(ra_1);                                      // K-step ks + 3
      compute(1, gb_1);
      if (ks + 2 < nk) { load_b(gb_1, 1); split_store(0, 0, ra_0); }   // else ra_0 is the next tile's first K-step: stored after the epilogue
    }
#pragma unroll
    for (int mi = 0; mi < C::MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < C::NI; ++ni)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[mi][ni][e] *= f16_desc;
    f32x16 unused;
#pragma unroll
    for (int e = 0; e < 16; ++e) unused[e] = 0.f;
    // (opaque copies: everything the epilogue derives from them would otherwise be hoisted out of the tile loop and stay in
    //  registers through the K loop - 256 registers and spills instead of ~200)
    int n0v = n0, lanev = lane, wnv = wn;
    asm volatile("" : "+s"(n0v), "+v"(lanev), "+s"(wnv));
    igemm_epilogue<BM, BN, MODE, false, BN / 64, false>(p, acc, unused, (mt0 + t) * BM, n0v, 0, wnv, lanev, Mrows, HW, Wrow, 0, 0, lds);
    __syncthreads();                                     // the staging area is the operand area again
  }
}

// ------------------------------------------------------------------------------------
// forward / data gradient of STRIDE-1 3x3 convolutions with tap re-use ("igemm_x6t").  In the implicit GEMM above every
// activation element is loaded, split and stored once PER TAP - nine times for a 3x3 filter - and that conversion work,
// not the matrix pipe, bounds the kernel (and utterly so for the 32-channel layers of the physique net, whose 32-column
// tiles amortise it over six MFMAs per half-step).  Here the 128 rows of a tile are an 8 x 16 pixel patch of one image
// (8 x 8 patches of two images for 8-pixel-wide maps): per 32-channel chunk the block stages the patch WITH ITS HALO
// ((8 + 2) x (16 + 2) pixels) as three bf16 planes ONCE, and the nine taps read their fragments from the same staging at
// a constant offset per tap ((dy * halo width + dx) pixels): 1.4 instead of 9 conversions per element, two barriers per
// chunk (18 half-steps) instead of 18, 6.4 x fewer activation bytes through the vector memory path.  Pixels are 80 B apart
// in a plane (64 B of channels + 16 B pad): the 16 lanes of a fragment read cover the 64 banks once.  The next chunk's
// halo is fetched into registers at the start of the MFMA phase.  Weights: pre-split fragments straight to registers,
// as in igemm_x6_kernel.  Epilogue: the shared one, rows mapped by tile_row().
// ------------------------------------------------------------------------------------
constexpr int XT_PIXB = 80;                    // bytes per halo pixel and plane
constexpr int XT_NJ = 7;                       // float4 per thread per chunk: 200 halo pixels x 8 float4 / 256 threads
constexpr int XT_PLANE_MAX = 200 * XT_PIXB;    // two 10 x 10 halos (8-wide maps); 10 x 18 = 180 pixels for 16-wide patches

#ifndef XAS_X6T_WAVES
#define XAS_X6T_WAVES 2                // launch-bounds waves per SIMD of igemm_x6t_kernel (3: 168 VGPRs)
#endif
template <int BN, int MODE, int P>
__global__ __launch_bounds__(256, XAS_X6T_WAVES) void igemm_x6t_kernel(IgemmParams p) {
  constexpr int BM = 128;
  using C = TileCfg<BM, BN>;
  extern __shared__ __align__(16) float lds[];
  unsigned char* S = reinterpret_cast<unsigned char*>(lds);       // [P][halo pixels][80 B]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / C::WAVES_N, wn = wave % C::WAVES_N;
  const int H = p.Hd, W = p.Wd;                                   // stride 1, same-size: input and output maps coincide
  const int tw = p.t2d_tw, tws = tw == 16 ? 4 : 3;
  const int tn_cnt = 128 >> (3 + tws);                            // images per tile
  const int hw = tw + 2, npix_img = 10 * hw, npix = tn_cnt * npix_img;
  const int plane_b = npix * XT_PIXB;

  const int xcd = blockIdx.x & 7, qb = blockIdx.x >> 3;           // XCD grid: see igemm_x6_kernel
  const int xi = xcd / p.xn, xj = xcd - xi * p.xn;
  const int ml = qb / p.nt_per_x;
  const int mt = xi * p.mt_per_xcd + ml;
  const int nt = xj * p.nt_per_x + (qb - ml * p.nt_per_x);
  if (mt >= p.nMt || nt >= p.nNt) return;
  const int m0 = mt * BM, n0 = nt * BN;
  const int Mrows = p.N * H * W;
  __builtin_amdgcn_s_setprio(XAS_X6_PRIO);
  float f16_sa = kF16AScale, f16_desc = kF16Descale;   // P == 2: scale of the gathered operand, scale of the result
  if (P == 2 && p.a_amax) { float inv; f16_sa = f16_grad_scale(p.a_amax, &inv); f16_desc = inv * (1.f / kF16WScale); }
  const int tiles_x = W / tw, per_img = tiles_x * (H >> 3);
  const int img0 = (mt / per_img) * tn_cnt, tt = mt % per_img;
  const int y0 = (tt / tiles_x) * 8, x0 = (tt % tiles_x) * tw;
  const int cchunks = p.Cs / BK;
  const int ntap = p.R * p.S;

  // ---- staging: item = tid + 256 j -> (halo pixel, float4 of the chunk's 32 channels); 8 lanes fetch a pixel's 128 bytes
  const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.src), 0, (int)(p.src_elems * 4), 0x00020000);
  unsigned voffA[XT_NJ];
#pragma unroll
  for (int j = 0; j < XT_NJ; ++j) {
    const int item = tid + 256 * j, pix = item >> 3, q = item & 7;
    voffA[j] = kOOB;
    if (pix < npix) {
      const int tn = pix / npix_img, pr = pix - tn * npix_img;
      const int hy = pr / hw, hx = pr - hy * hw;
      const int gy = y0 + hy - 1, gx = x0 + hx - 1, img = img0 + tn;
      if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W && img < p.N)
        voffA[j] = (unsigned)(((((long)img * H + gy) * W + gx) * p.Cs + q * 4) * 4);
    }
  }
  float4 ra[XT_NJ];
  auto load_chunk = [&](int chunk) {
    const unsigned soff = (unsigned)(chunk * BK) * 4u;
#pragma unroll
    for (int j = 0; j < XT_NJ; ++j) ra[j] = buf_load16(rsrcA, voffA[j], soff);
  };
  auto stage = [&]() {
#pragma unroll
    for (int j = 0; j < XT_NJ; ++j) {
      const int item = tid + 256 * j, pix = item >> 3, q = item & 7;
      if (pix < npix) {
        const Pieces<P> pcs = split_pieces<P>(ra[j], f16_sa);
        unsigned char* d = S + pix * XT_PIXB + q * 8;
#pragma unroll
        for (int pc = 0; pc < P; ++pc) *reinterpret_cast<uint2*>(d + pc * plane_b) = pcs.q[pc];
      }
    }
  };
  // ---- fragments: lane (i = row of a 32-row block, hh = k half); row -> (image of the tile, patch y, patch x)
  const int i = lane & 31, hh = lane >> 5;
  int fbase[C::MI];
#pragma unroll
  for (int mi = 0; mi < C::MI; ++mi) {
    const int r = wm * C::WM + mi * 32 + i;
    const int tn = r >> (3 + tws), ty = (r >> tws) & 7, tx = r & (tw - 1);
    fbase[mi] = (tn * npix_img + (ty + 1) * hw + (tx + 1)) * XT_PIXB + hh * 16;
  }
  // ---- weights (as igemm_x6_kernel)
  const int Ktot = ntap * p.Cs;
  const unsigned blk_bytes = (unsigned)(Ktot / 16) * P * 1024u;
  const long wbytes = (long)((p.Cd + 31) / 32) * blk_bytes;
  const __amdgpu_buffer_rsrc_t rsrcB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wgt), 0, (int)wbytes, 0x00020000);
  const unsigned voffB = (unsigned)lane * 16u;
  const unsigned nblk0 = (unsigned)(n0 / 32 + wn * (C::WN / 32));
  // half-step stream of the weights: chunk outermost, taps, the two halves of a chunk innermost
  int b_chunk = 0, b_tap = 0;
  const int nhs = 2 * ntap * cchunks;
  int b_left = nhs;
  auto load_b = [&](uint4 (&gb)[P][C::NI], int h) {
    const unsigned hc = (unsigned)((b_tap * p.Cs + b_chunk * BK) / 16 + h);
#pragma unroll
    for (int ni = 0; ni < C::NI; ++ni)
#pragma unroll
      for (int pc = 0; pc < P; ++pc)
        gb[pc][ni] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(
            rsrcB, (int)voffB, (int)((nblk0 + ni) * blk_bytes + (hc * P + pc) * 1024u), 0));
    if (h) {                                           // next tap; past the end: stay (re-loads, never consumed)
      const bool more = b_left > 2;
      b_left -= more ? 2 : 0;
      int t2 = b_tap + 1, c2 = b_chunk;
      if (t2 == ntap) { t2 = 0; ++c2; }
      b_tap = more ? t2 : b_tap; b_chunk = more ? c2 : b_chunk;
    }
  };
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
  auto read_frag = [&](uint4 (&fa)[P][C::MI], int tapoff) {          // tapoff: byte offset of the tap + half inside a plane
#pragma unroll
    for (int pc = P - 1; pc >= 0; --pc)
#pragma unroll
      for (int mi = 0; mi < C::MI; ++mi)
        fa[pc][mi] = *reinterpret_cast<const uint4*>(S + pc * plane_b + fbase[mi] + tapoff);
  };
  auto mfmas = [&](const uint4 (&fa_)[P][C::MI], const uint4 (&gb_)[P][C::NI]) {
#ifdef XAS_FRAG_AGPR
    uint4 fa[P][C::MI], gb[P][C::NI];
#pragma unroll
    for (int pc = 0; pc < P; ++pc) {
#pragma unroll
      for (int mi = 0; mi < C::MI; ++mi) fa[pc][mi] = frag_reg(fa_[pc][mi]);
#pragma unroll
      for (int ni = 0; ni < C::NI; ++ni) gb[pc][ni] = frag_reg(gb_[pc][ni]);
    }
#else
    const uint4 (&fa)[P][C::MI] = fa_;
    const uint4 (&gb)[P][C::NI] = gb_;
#endif
#pragma unroll
    for (int t = 0; t < Products<P>::N; ++t)
#pragma unroll
      for (int mi = 0; mi < C::MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < C::NI; ++ni)
          acc[mi][ni] = mfma_piece<P>(gb[Products<P>::B[t]][ni], fa[Products<P>::A[t]][mi], acc[mi][ni]);
  };
  auto tap_off = [&](int t) {                          // byte offset of tap t (= jr * S + js) inside a plane
    const int jr = t / p.S, js = t - jr * p.S;
    const int dy = MODE == 0 ? jr - p.pad : p.pad - jr, dx = MODE == 0 ? js - p.pad : p.pad - js;
    return (dy * hw + dx) * XT_PIXB;
  };
  uint4 gb_0[P][C::NI], gb_1[P][C::NI];
  uint4 fa_0[P][C::MI], fa_1[P][C::MI];                // fragments of the half-step in flight and of the next one
  load_chunk(0);
  load_b(gb_0, 0);
  load_b(gb_1, 1);
  for (int chunk = 0; chunk < cchunks; ++chunk) {
    stage();                                           // (waits for the chunk's loads)
    __syncthreads();
    load_chunk(chunk + 1 < cchunks ? chunk + 1 : chunk);
    int off = tap_off(0);
    read_frag(fa_0, off);
    for (int t = 0; t < ntap; ++t) {
      read_frag(fa_1, off + 32);                       // second half of this tap, read during the MFMAs of the first
      mfmas(fa_0, gb_0);
      load_b(gb_0, 0);
      off = tap_off(t + 1 < ntap ? t + 1 : t);
      read_frag(fa_0, off);                            // first half of the next tap (a re-read after the last one)
      mfmas(fa_1, gb_1);
      load_b(gb_1, 1);
    }
    __syncthreads();                                   // every wave has read the staging before it is overwritten
  }
  if constexpr (P == 2) {                              // the operands were split as 2^10 w and 2^4 x
#pragma unroll
    for (int mi = 0; mi < C::MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < C::NI; ++ni)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[mi][ni][e] *= f16_desc;
  }
  igemm_epilogue<BM, BN, MODE, false, (BN == 128 ? 2 : 1)>(p, acc, acc2, m0, n0, wm, wn, lane, Mrows, H * W, W, 0, 0, lds);
#ifdef XAS_DRAIN_AT_END
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
}

template <int BN>
constexpr size_t igemm_x6t_lds(int P) {
  size_t a = (size_t)P * XT_PLANE_MAX;
  size_t b = ((size_t)128 * (BN / (BN == 128 ? 2 : 1) + 4) + 2 * 256) * sizeof(float);
  return a > b ? a : b;
}

// does the tap-reuse kernel take this problem?  (stride-1 3x3, same-size maps, 128-row tiles that are whole patches)
static bool x6t_takes(const IgemmParams& p, int bm, int phases) {
  if (p.tune & (1 << 22)) return false;                // tune bit 22: implicit-GEMM kernel for every shape
  if (bm != 128 || phases != 1 || p.bnb_x) return false;
  return tap_tile_ok(p.R, p.S, p.stride, p.pad, p.Hs, p.Ws, p.Hd, p.Wd, p.Cs, p.N);
}

// Tried and dropped:
// * (commit "Experiment: igemm_x6d_kernel", profiles/r03_igemm_x6_dma_vs_regstaged.txt) the fp32 activation rows by
//   LDS-DMA (buffer_load_dwordx4 ... lds into a 4-slot ring, swizzle on the source address, counted vmcnt + raw s_barrier)
//   with the split done at fragment-read time.  Parity-green, but 154.7 against 173.6 TFLOP/s on ten layer shapes: both
//   waves that share a row block split it (92 instead of 44 vector-ALU instructions per wave and half-step, beside 24
//   MFMAs), and hipcc's in-order vmcnt for the weight-fragment loads retires every older DMA with them, so the ring cannot
//   run further ahead than the register sets of this kernel do.
// * (commit "Experiment: igemm_x6s_kernel", profiles/r03_igemm_x6_wave_specialised_ab.txt) wave specialisation: 4 MFMA
//   waves that only read fragments (one phase ahead, two register sets), load weight fragments and issue MFMAs + 2 or 4
//   producer waves that load, split and store the activations into a three-slot plane ring, one barrier per half-step,
//   producers end before the epilogue.  Parity-green; 154 TFLOP/s with 4 producers (one block per CU: a lone MFMA wave
//   per SIMD has nobody to cover its barrier and weight-load waits), 138 with 2 producers and two blocks per CU, against
//   179 for this kernel at three blocks per CU.
// * (r03, f16x3) whole K-steps between barriers for the two-piece format, whose half-step is only 12 MFMAs per wave
//   (four half-buffers, one barrier per 24 MFMAs, weight fragments requested a K-step ahead): parity-green, 249 against
//   254 TFLOP/s on ten forward shapes and +0.9 ms per step - the barrier count is not what the short half-steps cost.
// * an L2 warm-up of the activation rows four K-steps ahead (one buffer_load_dword ... lds per row and K-step into a
//   scratch strip): 163 against 169.5 TFLOP/s with the warm-up switched off in the same build - the activation-load
//   stalls of the ablation are not HBM latency that a warmer L2 removes.

template <int BM, int BN>
constexpr size_t igemm_x6_lds(int P, bool bnb) {
  size_t a = (size_t)2 * P * BM * XLDH * sizeof(unsigned short);   // two half-buffers of planes
  size_t b = ((size_t)BM * (BN / (BN >= 128 ? BN / 64 : 1) + 4) + 2 * 256) * sizeof(float);   // epilogue staging (64-column passes for BN >= 128) + partial-combine area
  size_t c = bnb ? ((size_t)2 * TileCfg<BM, BN>::WAVES_M * 32 * (BN + 4) + 2 * 256) * sizeof(float) : 0;   // bn-backward epilogue staging
  a = a > b ? (a > c ? a : c) : (b > c ? b : c);
#ifdef XAS_X6_LDS_FLOOR                              // ablation builds: same blocks per CU for every variant
  a = a > XAS_X6_LDS_FLOOR ? a : XAS_X6_LDS_FLOOR;
#endif
  return a;
}

template <int BM, int BN, int MODE, int P, bool BNB>
static int launch_igemm_x6_t(const IgemmParams& p, int Mrows_max, int phases, hipStream_t st) {
  constexpr size_t lds = igemm_x6_lds<BM, BN>(P, BNB);
  static bool attr_set_dev[kMaxDevices] = {};
  bool& attr_set = attr_set_dev[current_device()];
  if (!attr_set && lds > 64 * 1024) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_x6_kernel<BM, BN, MODE, P, BNB>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  IgemmParams q = p;
  q.nMt = (int)cdiv(Mrows_max, BM); q.nNt = (int)cdiv(p.Cd, BN);
  // XCD grid (8 / xn) x xn.  Model of the bytes that miss the 4 MB L2 of an XCD: the activation rows are fetched by the
  // xn XCDs of a grid row; an XCD's weight slice is fetched once when it stays resident (<= 2.5 MB), else once per
  // round of resident M-tiles (64 blocks per XCD).  Layer4 of the detector (8 x 8 maps, 14 MB of split 3x3 weights):
  // 878 -> 260 MB of L2 misses per launch at 256 images (PMC, tools/gpu/pmc_shapes.sh)
  {
    const double a_bytes = (double)Mrows_max * phases * p.Cs * 4.0 * (MODE == 0 ? p.stride * p.stride : 1);
    const double b_bytes = (double)p.Cd * p.R * p.S * p.Cs * 2.0 * P;
    double best = 0;
    q.xn = 1;
    for (int xn = 1; xn <= 8; xn *= 2) {
      if (q.nNt % xn != 0 || q.nMt < 8 / xn) continue;
      const int xm = 8 / xn, ntx = q.nNt / xn, mtx = (int)cdiv(q.nMt, xm);
      const double slice = b_bytes / xn;
      const double rounds = slice <= 2.5e6 ? 1.0 : (double)cdiv(mtx, 64 / ntx > 0 ? 64 / ntx : 1);
      const double cost = a_bytes * xn + slice * 8 * rounds;
      if (xn == 1 || cost < best * 0.9) { best = cost; q.xn = xn; }
    }
  }
  q.nt_per_x = q.nNt / q.xn;
  q.mt_per_xcd = (int)cdiv(q.nMt, 8 / q.xn);
  dim3 grid((unsigned)(8 * q.mt_per_xcd * q.nt_per_x), 1, (unsigned)phases);
  hipLaunchKernelGGL((igemm_x6_kernel<BM, BN, MODE, P, BNB>), grid, dim3(256), lds, st, q);
  XAS_LAUNCH_CHECK();
  return 0;
}

template <int BN, int MODE, int P>
static int launch_igemm_x6t(const IgemmParams& p, int Mrows_max, hipStream_t st) {
  constexpr size_t lds = igemm_x6t_lds<BN>(P);
  IgemmParams q = p;
  q.t2d_tw = p.Wd % 16 == 0 ? 16 : 8;
  q.nMt = Mrows_max / 128; q.nNt = (int)cdiv(p.Cd, BN);
  q.xn = 1; q.nt_per_x = q.nNt; q.mt_per_xcd = (int)cdiv(q.nMt, 8);
  dim3 grid((unsigned)(8 * q.mt_per_xcd * q.nt_per_x), 1, 1);
  hipLaunchKernelGGL((igemm_x6t_kernel<BN, MODE, P>), grid, dim3(256), lds, st, q);
  XAS_LAUNCH_CHECK();
  return 0;
}

// pure GEMM with an even number of K-steps: what igemm_x6p_kernel is written for (tune bit 26: never).
// Measured alone at 384 / 256 images (r05): K = 64: 505 -> 437 us forward (4.6 TB/s), 427 -> 362 us data gradient; K = 128:
// -5 % / -9 %; K >= 256: +-2 % either way.  In the step (in-box, interleaved, 3 rounds) taking every K is 0.2-0.3 ms better
// than stopping at 128: XAS_X6P_MAX_K bounds it for experiments.
#ifndef XAS_X6P_MAX_K
#define XAS_X6P_MAX_K 4096
#endif
static bool x6p_takes(const IgemmParams& p, int mode, int phases) {
  if (p.tune & (1 << 26)) return false;
  if (p.Cs > XAS_X6P_MAX_K && !(p.tune & (1 << 27))) return false;
  const int hr = mode == 0 ? p.Hrow : p.Hd, wr = mode == 0 ? p.Wrow : p.Wd;
  return p.R == 1 && p.S == 1 && p.stride == 1 && p.pad == 0 && phases == 1 && p.Cs % (2 * BK) == 0 && p.Hs == hr && p.Ws == wr &&
         !p.t2d_tw && !p.bnb_x && p.a_amax != nullptr && (p.Cd & 3) == 0;
}

template <int MODE>
static int launch_igemm_x6p(const IgemmParams& p, int Mrows_max, hipStream_t st) {
  constexpr size_t lds = igemm_x6_lds<64, 256>(2, false);
  IgemmParams q = p;
  q.nMt = (int)cdiv(Mrows_max, 64); q.nNt = (int)cdiv(p.Cd, 256);
  // tiles per block: long enough for the stream to matter, short enough for >= 4 rounds of the ~768 resident blocks
  const long tiles = (long)q.nMt * q.nNt;
  int tpb = (int)(tiles / (768 * 4));
  tpb = tpb < 1 ? 1 : (tpb > 8 ? 8 : tpb);
  q.tpb = tpb;
  const int nMg = (int)cdiv(q.nMt, tpb);
  {                                                    // XCD grid: the model of launch_igemm_x6_t, M-groups as the unit
    const double a_bytes = (double)Mrows_max * p.Cs * 4.0;
    const double b_bytes = (double)p.Cd * p.Cs * 2.0 * 2;
    double best = 0;
    q.xn = 1;
    for (int xn = 1; xn <= 8; xn *= 2) {
      if (q.nNt % xn != 0 || nMg < 8 / xn) continue;
      const int xm = 8 / xn, ntx = q.nNt / xn, mtx = (int)cdiv(q.nMt, xm);
      const double slice = b_bytes / xn;
      const double rounds = slice <= 2.5e6 ? 1.0 : (double)cdiv(mtx, 64 / ntx > 0 ? 64 / ntx : 1);
      const double cost = a_bytes * xn + slice * 8 * rounds;
      if (xn == 1 || cost < best * 0.9) { best = cost; q.xn = xn; }
    }
  }
  q.nt_per_x = q.nNt / q.xn;
  q.mt_per_xcd = (int)cdiv(nMg, 8 / q.xn);            // in groups
  dim3 grid((unsigned)(8 * q.mt_per_xcd * q.nt_per_x), 1, 1);
  hipLaunchKernelGGL((igemm_x6p_kernel<MODE>), grid, dim3(256), lds, st, q);
  XAS_LAUNCH_CHECK();
  return 0;
}

template <int MODE, int P>
static int launch_igemm_x6_p(const IgemmParams& p, int Mrows_max, int phases, hipStream_t st) {
  int bm, bn;
  if constexpr (MODE == 0) {                           // the head epilogue lives in the 64 x 256 tile (one image row x four joints)
    if (p.head_partial) return launch_igemm_x6_t<64, 256, 0, P, false>(p, Mrows_max, phases, st);
  }
  pick_tile(p.Cd, Mrows_max, phases, &bm, &bn);
  if (x6t_takes(p, bm, phases)) {
    if (bn == 128) return launch_igemm_x6t<128, MODE, P>(p, Mrows_max, st);
    if (bn == 64) return launch_igemm_x6t<64, MODE, P>(p, Mrows_max, st);
    return launch_igemm_x6t<32, MODE, P>(p, Mrows_max, st);
  }
  if constexpr (MODE == 1 && P != 2) {
    if (p.bnb_x) {
      if (bn == 128) return launch_igemm_x6_t<128, 128, 1, P, true>(p, Mrows_max, phases, st);
      if (bm == 64) return launch_igemm_x6_t<64, 64, 1, P, true>(p, Mrows_max, phases, st);
      if (bn == 64) return launch_igemm_x6_t<128, 64, 1, P, true>(p, Mrows_max, phases, st);
      return launch_igemm_x6_t<128, 32, 1, P, true>(p, Mrows_max, phases, st);
    }
  }
  if (!(p.tune & (1 << 23))) {                         // tune bit 23: no 64 x 256 tiles
    pick_tile(p.Cd, Mrows_max, phases, &bm, &bn, true);
    if constexpr (P == 2) {
      if (bn == 256 && x6p_takes(p, MODE, phases)) return launch_igemm_x6p<MODE>(p, Mrows_max, st);
    }
    if (bn == 256) return launch_igemm_x6_t<64, 256, MODE, P, false>(p, Mrows_max, phases, st);
  }
  if (bn == 128) return launch_igemm_x6_t<128, 128, MODE, P, false>(p, Mrows_max, phases, st);
  if (bm == 64) return launch_igemm_x6_t<64, 64, MODE, P, false>(p, Mrows_max, phases, st);
  if (bn == 64) return launch_igemm_x6_t<128, 64, MODE, P, false>(p, Mrows_max, phases, st);
  return launch_igemm_x6_t<128, 32, MODE, P, false>(p, Mrows_max, phases, st);
}

// pieces: 3 = bf16x6, 1 = bf16, 2 = f16x3 (the gathered operand at a fixed scale when it is an activation, at a scale
// derived from IgemmParams::a_amax when it is a gradient)
int launch_igemm_x6(const IgemmParams& p, int mode, int Mrows_max, int phases, int pieces, hipStream_t st) {
  if (mode == 0) {
    if (pieces == 2) {
      XAS_REQUIRE(p.a_amax, "conv: an f16x3 forward launch needs the maximum of its input tensor (xas_conv_shape.grad_amax)");
      return launch_igemm_x6_p<0, 2>(p, Mrows_max, phases, st);
    }
    return pieces == 3 ? launch_igemm_x6_p<0, 3>(p, Mrows_max, phases, st) : launch_igemm_x6_p<0, 1>(p, Mrows_max, phases, st);
  }
  if (pieces == 2) {
    XAS_REQUIRE(p.a_amax && !p.bnb_x, "conv: an f16x3 data gradient needs the maximum of its gradient operand (xas_conv_shape.grad_amax)");
    return launch_igemm_x6_p<1, 2>(p, Mrows_max, phases, st);
  }
  return pieces == 3 ? launch_igemm_x6_p<1, 3>(p, Mrows_max, phases, st) : launch_igemm_x6_p<1, 1>(p, Mrows_max, phases, st);
}

// ------------------------------------------------------------------------------------
// weight gradient: dW[co][nn] = sum_m dY[m][co] * Xcol[m][nn],  nn = (r*S + s)*Cin + c;  K = pixels, split over blocks.
// Both operands are activations ([pixel][channel] in memory): both are split in the kernel.  LDS tiles stay pixel-major
// as they arrive, [plane][16 pixels][channels]; the MFMA operands (8 consecutive pixels of one channel per lane) come out
// of ds_read_b64_tr_b16, the hardware transposing read (4 pixels x 16 channels per 16-lane group).  Row strides are
// 64 or 192 mod 256 bytes: the four pixel rows of a read land on disjoint banks.
// ------------------------------------------------------------------------------------
template <int BMN>
struct XStride { static constexpr int value = BMN * 2 + (BMN == 128 ? 64 : (BMN == 64 ? 64 : 0)); };   // 320 / 192 / 64 bytes

template <int BM, int BN, int P>
__global__ __launch_bounds__(256, (P == 2 ? XAS_WX6_WAVES2 : 2)) void wgrad_x6_kernel(WgradParams p) {
  using C = TileCfg<BM, BN>;
  constexpr int SA = XStride<BM>::value, SB = XStride<BN>::value;      // bytes per pixel row of a plane
  constexpr int ASZ = P * XH * SA, BSZ = P * XH * SB, HBUF = ASZ + BSZ; // bytes per half-buffer
  constexpr int AQ = BM / 4, AROWS = 256 / AQ;                          // dy: float4 per pixel, pixels per pass of the block
  constexpr int AK = WBK / AROWS;                                       // dy float4 per thread per K-step (BM = 32: 1, 64: 2, 128: 4)
  constexpr bool AHALF = AK == 1;                                       // one pass covers both halves: a thread's float4 belongs to ONE half
  constexpr int APH = AHALF ? 1 : AK / 2;                               // dy float4 per thread stored per half-step
  constexpr int ASET = AHALF ? 1 : 2 * APH;                             // register set size
  constexpr int BPQ = BN / 64;                                          // x: float4 per thread per half (16 threads per pixel)
  static_assert(AK >= 1 && BPQ >= 1, "tile too small for the staging scheme");
  extern __shared__ __align__(16) float lds[];
  unsigned char* S = reinterpret_cast<unsigned char*>(lds);             // [2][ A planes | B planes ]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / C::WAVES_N, wn = wave % C::WAVES_N;
  // XCD-grouped order: the KK-tiles of one (pixel split, Cout tile) read the same dy tile and the same x pixels (see
  // wgrad_buf_kernel in conv.hip)
  const int nkt = p.ntiles / p.nct;
  const int xcd = blockIdx.x & 7, qb = blockIdx.x >> 3;
  int split, tile;
  if (p.tune & 1) {                                  // ALL tiles of a pixel split on one XCD (splits a multiple of 8): x and dy
    split = xcd + 8 * (qb / p.ntiles);               // of the split are fetched into ONE L2; KK-tiles of a Cout tile adjacent
    const int t = qb - (qb / p.ntiles) * p.ntiles;
    tile = (t / nkt) + (t - (t / nkt) * nkt) * p.nct;
  } else {
    const int grp = xcd + 8 * (qb / nkt);
    const int kt = qb - (qb / nkt) * nkt;
    split = grp / p.nct;
    tile = (grp - split * p.nct) + kt * p.nct;
  }
  if (split >= p.nsplits) return;
  const int co0 = (tile % p.nct) * BM, nn0 = (tile / p.nct) * BN;
  const int mbeg = split * p.m_per_split, mend = min(p.M, mbeg + p.m_per_split);
  const int HWo = p.Ho * p.Wo;
#ifdef XAS_WGRAD_PRIO
  __builtin_amdgcn_s_setprio(XAS_WGRAD_PRIO);
#endif
  float f16_sd = kF16AScale, f16_sx = kF16AScale, f16_desc = 1.f;      // P == 2: scales of dy and of x, scale of the result
  if (P == 2) {                                        // (each operand from its maximum; the launchers insist on both)
    float id = 1.f / kF16AScale, ix = 1.f / kF16AScale;
    if (p.a_amax) f16_sd = f16_grad_scale(p.a_amax, &id);
    if (p.b_amax) f16_sx = f16_grad_scale(p.b_amax, &ix);
    f16_desc = id * ix;
  }

  // ---- dy operand: per-lane offset fixed, rows of a half-step from a scalar offset, split end = buffer range
  const int aq = tid % AQ, apix = tid / AQ;
  const __amdgpu_buffer_rsrc_t rsrcA =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, (int)((long)mend * p.Cout * 4), 0x00020000);
  const unsigned voffA = (co0 + aq * 4 < p.Cout) ? (unsigned)(apix * p.Cout + co0 + aq * 4) * 4u : kOOB;
  const unsigned passA = (unsigned)(AROWS * p.Cout) * 4u;
  const unsigned dstA = (unsigned)((apix & (XH - 1)) * SA + aq * 8);
  const int ahalf = __builtin_amdgcn_readfirstlane(apix >> 4);          // AHALF: which half-step this wave's dy rows belong to (wave-uniform)

  // ---- x operand: thread = (pixel bpix of the half-step, float4 slot bq); 16 threads x BPQ float4 cover the BN columns
  const int bq = tid & 15, bpix = tid >> 4;
  const long biasB = ((long)p.pad * p.Wi + p.pad) * p.Cin;
  const __amdgpu_buffer_rsrc_t rsrcB = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.x) - biasB, 0, (int)((biasB + (long)p.N * p.Hi * p.Wi * p.Cin) * 4), 0x00020000);
  int tr[BPQ], ts[BPQ];
  unsigned offq[BPQ];                                  // ((tr*Wi + ts)*Cin + c)*4 + bias bytes
  bool colok[BPQ];
#pragma unroll
  for (int q = 0; q < BPQ; ++q) {
    const int nn = nn0 + (bq + 16 * q) * 4;
    colok[q] = nn < p.KK;
    const int nc = colok[q] ? nn : 0;
    const int tap = p.div_cin.div(nc), c = nc - tap * p.Cin;
    tr[q] = p.div_s.div(tap); ts[q] = tap - tr[q] * p.S;
    offq[q] = (unsigned)(((tr[q] * p.Wi + ts[q]) * p.Cin + c) * 4 + biasB * 4);
  }
  int dh_r[2], dw_r[2];                                // this thread's pixel of half 0 / half 1 relative to the step's first pixel
#pragma unroll
  for (int h = 0; h < 2; ++h) { dh_r[h] = (bpix + XH * h) / p.Wo; dw_r[h] = (bpix + XH * h) - dh_r[h] * p.Wo; }
  const unsigned dstB = (unsigned)(ASZ + bpix * SB + bq * 8);

  f32x16 acc[C::MI][C::NI];
#pragma unroll
  for (int mi = 0; mi < C::MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < C::NI; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  float4 ra_0[ASET], rb_0[2 * BPQ], ra_1[ASET], rb_1[2 * BPQ];
  auto load_k = [&](int mk, float4 (&ra)[ASET], float4 (&rb)[2 * BPQ]) {        // mk: first pixel of the K-step (uniform)
    const unsigned soffA = (unsigned)mk * (unsigned)p.Cout * 4u;
#pragma unroll
    for (int j = 0; j < ASET; ++j) ra[j] = buf_load16(rsrcA, voffA, soffA + j * passA);     // passes 0..APH-1 = half 0
    const int n0 = p.div_hw.div(mk), rem = mk - n0 * HWo;
    const int h0 = p.div_w.div(rem), w0 = rem - h0 * p.Wo;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int w = w0 + dw_r[h];
      const int c1 = w >= p.Wo ? 1 : 0;
      w -= c1 ? p.Wo : 0;
      int hr = h0 + dh_r[h] + c1;
      const int c2 = hr >= p.Ho ? 1 : 0;
      hr -= c2 ? p.Ho : 0;
      const int n = n0 + c2;
      const int hb = hr * p.stride - p.pad, wb = w * p.stride - p.pad;
      const unsigned pix = (unsigned)(((n * p.Hi + hb) * p.Wi + wb) * p.Cin) * 4u;
#pragma unroll
      for (int q = 0; q < BPQ; ++q) {
        const bool ok = colok[q] && (unsigned)(hb + tr[q]) < (unsigned)p.Hi && (unsigned)(wb + ts[q]) < (unsigned)p.Wi;
        rb[h * BPQ + q] = buf_load16(rsrcB, ok ? pix + offq[q] : kOOB, 0u);
      }
    }
  };
  auto store_half = [&](int buf, int h, const float4 (&ra)[ASET], const float4 (&rb)[2 * BPQ]) {
    unsigned char* sb = S + buf * HBUF;
#pragma unroll
    for (int j = 0; j < APH; ++j) {
      if (AHALF && ahalf != h) continue;               // (wave-uniform) this wave's dy rows belong to the other half
      const Pieces<P> pcs = split_pieces<P>(ra[AHALF ? 0 : h * APH + j], f16_sd);
#pragma unroll
      for (int pc = 0; pc < P; ++pc) *reinterpret_cast<uint2*>(sb + pc * XH * SA + AROWS * j * SA + dstA) = pcs.q[pc];
    }
#pragma unroll
    for (int q4 = 0; q4 < BPQ; ++q4) {
      const Pieces<P> pcs = split_pieces<P>(rb[h * BPQ + q4], f16_sx);
#pragma unroll
      for (int pc = 0; pc < P; ++pc) *reinterpret_cast<uint2*>(sb + pc * XH * SB + q4 * 128 + dstB) = pcs.q[pc];
    }
  };
  // transposing fragment read: lane (l16 = lane & 15 -> row qd = l16 >> 2, column quad pp = l16 & 3; g1 = channel half;
  // hh = k half) supplies the address of pixel row 8 hh + 4 t + qd, channels 16 g1 + 4 pp .. + 3 of its 32-channel block
  const int l16 = lane & 15, qd = l16 >> 2, pp = l16 & 3, g1 = (lane >> 4) & 1, hh = lane >> 5;
  const unsigned fragA = (unsigned)((8 * hh + qd) * SA + (wm * C::WM + 16 * g1 + 4 * pp) * 2);
  const unsigned fragB = (unsigned)(ASZ + (8 * hh + qd) * SB + (wn * C::WN + 16 * g1 + 4 * pp) * 2);
  typedef s16x4_t __attribute__((address_space(3))) * lds_s16x4_p;
  auto tr_read = [&](const unsigned char* base, unsigned off) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(base + off));
  };
  auto compute = [&](int buf) {
    const unsigned char* sb = S + buf * HBUF;
    uint4 fa[P][C::MI], fb[P][C::NI];
#pragma unroll
    for (int pc = P - 1; pc >= 0; --pc) {
#pragma unroll
      for (int mi = 0; mi < C::MI; ++mi) {
        const s16x4_t lo = tr_read(sb, fragA + pc * XH * SA + mi * 64);
        const s16x4_t hi = tr_read(sb, fragA + pc * XH * SA + mi * 64 + 4 * SA);
        fa[pc][mi] = __builtin_bit_cast(uint4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
      }
#pragma unroll
      for (int ni = 0; ni < C::NI; ++ni) {
        const s16x4_t lo = tr_read(sb, fragB + pc * XH * SB + ni * 64);
        const s16x4_t hi = tr_read(sb, fragB + pc * XH * SB + ni * 64 + 4 * SB);
        fb[pc][ni] = __builtin_bit_cast(uint4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
      }
    }
    frag_regs(fa);
    frag_regs(fb);
#pragma unroll
    for (int t = 0; t < Products<P>::N; ++t)
#pragma unroll
      for (int mi = 0; mi < C::MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < C::NI; ++ni)
          acc[mi][ni] = mfma_piece<P>(fa[Products<P>::A[t]][mi], fb[Products<P>::B[t]][ni], acc[mi][ni]);
  };
  const int nsteps = (mend > mbeg) ? (mend - mbeg + WBK - 1) / WBK : 0;
  const int last = nsteps - 1;
  auto mk_of = [&](int st) { return mbeg + (st < last ? st : last) * WBK; };
  if (nsteps > 0) {
    load_k(mk_of(0), ra_0, rb_0);
    load_k(mk_of(1), ra_1, rb_1);
    store_half(0, 0, ra_0, rb_0);
    int st = 0;
    for (; st + 1 < nsteps; st += 2) {
      __syncthreads();
      compute(0);
      store_half(1, 1, ra_0, rb_0);
      __syncthreads();
      load_k(mk_of(st + 2), ra_0, rb_0);
      compute(1);
      store_half(0, 0, ra_1, rb_1);
      __syncthreads();
      compute(0);
      store_half(1, 1, ra_1, rb_1);
      __syncthreads();
      load_k(mk_of(st + 3), ra_1, rb_1);
      compute(1);
      store_half(0, 0, ra_0, rb_0);
    }
    if (st < nsteps) {
      __syncthreads();
      compute(0);
      store_half(1, 1, ra_0, rb_0);
      __syncthreads();
      compute(1);
    }
  }
  // accumulators -> slab (row = channel co from the register index, column = nn from the lane)
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
        if (nn < p.KK) slab[(size_t)co * p.KK + nn] = P == 2 ? acc[mi][ni][reg] * f16_desc : acc[mi][ni][reg];
      }
    }
#ifdef XAS_DRAIN_AT_END
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
}

// ------------------------------------------------------------------------------------
// weight gradient of STRIDE-1 3x3 convolutions with tap re-use ("wgrad_x6t"), the counterpart of igemm_x6t_kernel.
// In wgrad_x6_kernel a block owns (pixel split, 128 output channels, 128 columns of (tap, input channel)): the x pixels of
// a split are loaded and split once per tap AND per Cout tile, the dy pixels once per column tile (18 for 3x3 x 256).
// Here a block owns (pixel split, BM output channels, 32 input channels) and ALL NINE TAPS: per 8 x 16 pixel patch the x
// halo (10 x 18 pixels x 32 channels) is staged as bf16 planes once and the nine taps take their operands (8 consecutive
// pixels of one channel per lane, by the transposing LDS read) from it at a constant pixel offset per tap; dy is staged
// per K-slice of 16 pixels (one patch row), two buffers.  4 waves = (BM / 32 output-channel blocks) x (tap groups): a wave
// holds one 32 x 32 accumulator per tap of its group (9 / 5 / 3 for BM = 128 / 64 / 32).
// ------------------------------------------------------------------------------------
template <int BM, int P>
__global__ __launch_bounds__(256, 2) void wgrad_x6t_kernel(WgradParams p) {
  constexpr int NCB = BM / 32, NTG = 4 / NCB;          // output-channel blocks, tap groups
  constexpr int NTW = (9 + NTG - 1) / NTG;             // taps per wave (at most)
  constexpr int SA = XStride<BM>::value;               // bytes per dy pixel row of a plane
  constexpr int DYB = P * XH * SA;                     // bytes per dy buffer
  constexpr int XPB = 64;                              // bytes per x halo pixel of a plane (32 bf16)
  constexpr int DQ = BM / 4;                           // dy float4 per pixel
  constexpr int DJ = (XH * DQ + 255) / 256;            // dy float4 per thread per K-slice (2 / 1 / 1)
  extern __shared__ __align__(16) float lds[];
  unsigned char* S = reinterpret_cast<unsigned char*>(lds);       // [2 dy buffers][x planes]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cb = wave % NCB, tg = wave / NCB;
  const int H = p.Ho, W = p.Wo;
  const int tw = p.t2d_tw, tws = tw == 16 ? 4 : 3;
  const int tn_cnt = 128 >> (3 + tws);
  const int hw = tw + 2, npix_img = 10 * hw, npix = tn_cnt * npix_img;
  const int xplane = npix * XPB;
  unsigned char* SX = S + 2 * DYB;
  // block -> (split, Cout tile, 32-channel tile): the tiles of a split are adjacent (they share its x and dy in L2)
  const int nci = p.Cin / 32;
  const int tiles = p.nct * nci;
  const int split = blockIdx.x / tiles, tile = blockIdx.x - split * tiles;
  const int co0 = (tile / nci) * BM, c0 = (tile - (tile / nci) * nci) * 32;
  const int tiles_x = W / tw, per_img = tiles_x * (H >> 3);
  const int np_total = p.M >> 7;
  const int pbeg = split * p.pps, pend = min(np_total, pbeg + p.pps);
  if (pbeg >= pend) return;
  float f16_sd = kF16AScale, f16_sx = kF16AScale, f16_desc = 1.f;      // P == 2: scales of dy and of x, scale of the result
  if (P == 2) {                                        // (each operand from its maximum; the launchers insist on both)
    float id = 1.f / kF16AScale, ix = 1.f / kF16AScale;
    if (p.a_amax) f16_sd = f16_grad_scale(p.a_amax, &id);
    if (p.b_amax) f16_sx = f16_grad_scale(p.b_amax, &ix);
    f16_desc = id * ix;
  }

  const __amdgpu_buffer_rsrc_t rsrcX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)((long)p.N * H * W * p.Cin * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcD = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, (int)((long)p.M * p.Cout * 4), 0x00020000);
  // x halo item = tid + 256 j -> (halo pixel, float4 of the 32 channels): the pixel offset INSIDE a patch is fixed per thread
  int hpy[XT_NJ], hpx[XT_NJ], hpt[XT_NJ];
#pragma unroll
  for (int j = 0; j < XT_NJ; ++j) {
    const int item = tid + 256 * j, pix = item >> 3;
    hpt[j] = -1; hpy[j] = 0; hpx[j] = 0;
    if (pix < npix) {
      const int tn = pix / npix_img, pr = pix - tn * npix_img;
      hpt[j] = tn; hpy[j] = pr / hw - 1; hpx[j] = pr - (pr / hw) * hw - 1;
    }
  }
  float4 rx[XT_NJ];
  auto load_x = [&](int patch) {                       // the halo of patch `patch` -> registers
    const int img0 = (patch / per_img) * tn_cnt, tt = patch % per_img;
    const int y0 = (tt / tiles_x) * 8, x0 = (tt % tiles_x) * tw;
#pragma unroll
    for (int j = 0; j < XT_NJ; ++j) {
      const int gy = y0 + hpy[j], gx = x0 + hpx[j], img = img0 + hpt[j];
      const bool ok = hpt[j] >= 0 && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W && img < p.N;
      const unsigned off = ok ? (unsigned)(((((long)img * H + gy) * W + gx) * p.Cin + c0 + ((tid + 256 * j) & 7) * 4) * 4) : kOOB;
      rx[j] = buf_load16(rsrcX, off, 0u);
    }
  };
  auto stage_x = [&]() {
#pragma unroll
    for (int j = 0; j < XT_NJ; ++j) {
      const int item = tid + 256 * j, pix = item >> 3, q = item & 7;
      if (pix < npix) {
        const Pieces<P> pcs = split_pieces<P>(rx[j], f16_sx);
        unsigned char* d = SX + pix * XPB + q * 8;
#pragma unroll
        for (int pc = 0; pc < P; ++pc) *reinterpret_cast<uint2*>(d + pc * xplane) = pcs.q[pc];
      }
    }
  };
  // dy K-slice: 16 consecutive pixels (one patch row; two rows of an 8-wide map) x BM channels
  float4 rd[DJ];
  auto slice_row0 = [&](int patch, int ks) {           // global pixel index of the slice's first pixel
    const int img0 = (patch / per_img) * tn_cnt, tt = patch % per_img;
    const int y0 = (tt / tiles_x) * 8, x0 = (tt % tiles_x) * tw;
    if (tw == 16) return (img0 * H + y0 + ks) * W + x0;
    return ((img0 + (ks >> 2)) * H + y0 + 2 * (ks & 3)) * W + x0;
  };
  auto load_dy = [&](int patch, int ks) {
    const unsigned soff = (unsigned)slice_row0(patch, ks) * (unsigned)p.Cout * 4u;
#pragma unroll
    for (int j = 0; j < DJ; ++j) {
      const int item = tid + 256 * j, apix = item / DQ, aq = item - apix * DQ;
      const bool ok = item < XH * DQ && co0 + aq * 4 < p.Cout;
      rd[j] = buf_load16(rsrcD, ok ? (unsigned)(apix * p.Cout + co0 + aq * 4) * 4u : kOOB, soff);
    }
  };
  auto store_dy = [&](int buf) {
    unsigned char* sb = S + buf * DYB;
#pragma unroll
    for (int j = 0; j < DJ; ++j) {
      const int item = tid + 256 * j, apix = item / DQ, aq = item - apix * DQ;
      if (item < XH * DQ) {
        const Pieces<P> pcs = split_pieces<P>(rd[j], f16_sd);
#pragma unroll
        for (int pc = 0; pc < P; ++pc) *reinterpret_cast<uint2*>(sb + pc * XH * SA + apix * SA + aq * 8) = pcs.q[pc];
      }
    }
  };
  // transposing fragment reads (see wgrad_x6_kernel): lane -> pixel row 8 hh + qd (+ 4), channels 16 g1 + 4 pp .. + 3
  const int l16 = lane & 15, qd = l16 >> 2, pp = l16 & 3, g1 = (lane >> 4) & 1, hh = lane >> 5;
  const unsigned fragA = (unsigned)((8 * hh + qd) * SA + (cb * 32 + 16 * g1 + 4 * pp) * 2);
  const int rs = tw == 16 ? 8 : hw;                    // pixels between the two k-halves of a slice inside the halo
  const unsigned fragX = (unsigned)((hh * rs + qd) * XPB + (16 * g1 + 4 * pp) * 2);
  typedef s16x4_t __attribute__((address_space(3))) * lds_s16x4_p;
  auto tr_read = [&](const unsigned char* base, unsigned off) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(base + off));
  };
  f32x16 acc[NTW];
#pragma unroll
  for (int t = 0; t < NTW; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
  auto compute = [&](int buf, int ks) {
    const unsigned char* sb = S + buf * DYB;
    uint4 fa[P];
#pragma unroll
    for (int pc = P - 1; pc >= 0; --pc) {
      const s16x4_t lo = tr_read(sb, fragA + pc * XH * SA);
      const s16x4_t hi = tr_read(sb, fragA + pc * XH * SA + 4 * SA);
      fa[pc] = __builtin_bit_cast(uint4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    }
    frag_regs(fa);
    // halo pixel of the slice's first pixel for tap (0, 0) offset: row ks (two rows 2 (ks & 3) of image ks >> 2 when 8-wide)
    const int wbase = tw == 16 ? (ks + 1) * hw + 1 : (ks >> 2) * npix_img + (2 * (ks & 3) + 1) * hw + 1;
#pragma unroll
    for (int ti = 0; ti < NTW; ++ti) {
      const int t = tg + ti * NTG;
      if (t < 9) {
        const int dy = t / 3 - 1, dx = t - (t / 3) * 3 - 1;
        const unsigned xo = (unsigned)((wbase + dy * hw + dx) * XPB) + fragX;
        uint4 fb[P];
#pragma unroll
        for (int pc = P - 1; pc >= 0; --pc) {
          const s16x4_t lo = tr_read(SX, xo + pc * xplane);
          const s16x4_t hi = tr_read(SX, xo + pc * xplane + 4 * XPB);
          fb[pc] = __builtin_bit_cast(uint4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        }
        frag_regs(fb);
#pragma unroll
        for (int k = 0; k < Products<P>::N; ++k)
          acc[ti] = mfma_piece<P>(fa[Products<P>::A[k]], fb[Products<P>::B[k]], acc[ti]);
      }
    }
  };
  // ---- slice stream: 8 K-slices per patch; dy one slice ahead in registers, x halo one patch ahead in registers
  load_x(pbeg);
  load_dy(pbeg, 0);
  int buf = 0;
  for (int patch = pbeg; patch < pend; ++patch) {
    __syncthreads();                                   // every wave is done with the previous patch's x planes
    stage_x();
    if (patch + 1 < pend) load_x(patch + 1);
    for (int ks = 0; ks < 8; ++ks) {
      store_dy(buf);
      if (ks < 7) load_dy(patch, ks + 1);
      else if (patch + 1 < pend) load_dy(patch + 1, 0);
      __syncthreads();
      compute(buf, ks);
      buf ^= 1;
    }
  }
  // ---- accumulators -> slab [split][co][tap * Cin + c]
  float* slab = p.out + (size_t)split * p.Cout * p.KK;
  const int col_l = lane & 31, rsub = 4 * (lane >> 5);
#pragma unroll
  for (int ti = 0; ti < NTW; ++ti) {
    const int t = tg + ti * NTG;
    if (t < 9) {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int co = co0 + cb * 32 + (reg & 3) + 8 * (reg >> 2) + rsub;
        if (co < p.Cout) slab[(size_t)co * p.KK + t * p.Cin + c0 + col_l] = P == 2 ? acc[ti][reg] * f16_desc : acc[ti][reg];
      }
    }
  }
#ifdef XAS_DRAIN_AT_END
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
}

// plan of the tap-reuse weight gradient; false: the shape is not taken (wgrad_x6_kernel does it)
bool wgrad_x6t_plan(int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad, int Ho, int Wo, int* bm, int* splits, int* pps) {
  if (R != 3 || S != 3 || stride != 1 || pad != 1 || Ho != H || Wo != W) return false;
  if (Cin % 32 != 0 || H % 8 != 0) return false;
  if (!(W % 16 == 0 || (W == 8 && N % 2 == 0))) return false;
  // measured (profiles/r03_wgrad_x6t_ab.txt): 64 and 32 output channels +15..63 %; 128 and more -5..6 % with 128- and with
  // 64-channel blocks (nine accumulators per wave: 245 VGPRs) - those stay on wgrad_x6_kernel
  if (Cout == 64) *bm = 64;
  else if (Cout == 32) *bm = 32;
  else return false;
  const long np = (long)N * H * W / 128;
  const long tiles = (long)(Cout / *bm) * (Cin / 32);
#ifndef XAS_WX6T_BLOCKS
#define XAS_WX6T_BLOCKS 1536
#endif
  long sp = cdiv(XAS_WX6T_BLOCKS, tiles);              // ~1 500 blocks; a block should see at least 2 patches
  if (sp > np / 2) sp = np / 2;
  if (sp < 1) sp = 1;
  *pps = (int)cdiv(np, sp);
  *splits = (int)cdiv(np, *pps);
  return true;
}

template <int BM, int P>
static int launch_wgrad_x6t_t(const WgradParams& p, int splits, int pps, hipStream_t st) {
  constexpr size_t lds = (size_t)2 * P * XH * XStride<BM>::value + (size_t)P * 200 * 64;
  static bool attr_set_dev[kMaxDevices] = {};
  bool& attr_set = attr_set_dev[current_device()];
  if (!attr_set && lds > 64 * 1024) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_x6t_kernel<BM, P>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  WgradParams q = p;
  q.nct = p.Cout / BM;
  q.nsplits = splits; q.pps = pps;
  q.t2d_tw = p.Wo % 16 == 0 ? 16 : 8;
  dim3 grid((unsigned)(splits * q.nct * (p.Cin / 32)));
  hipLaunchKernelGGL((wgrad_x6t_kernel<BM, P>), grid, dim3(256), lds, st, q);
  XAS_LAUNCH_CHECK();
  return 0;
}

int launch_wgrad_x6t(const WgradParams& p, int bm, int splits, int pps, int pieces, hipStream_t st) {
  if (pieces == 2) {                                   // f16x3: dy at the scale of p.a_amax, x at the scale of p.b_amax
    XAS_REQUIRE(p.a_amax && p.b_amax, "conv_wgrad: the f16x3 weight gradient needs the maxima of both tensor operands (xas_conv_shape.grad_amax, x_amax)");
    if (bm == 128) return launch_wgrad_x6t_t<128, 2>(p, splits, pps, st);
    if (bm == 64) return launch_wgrad_x6t_t<64, 2>(p, splits, pps, st);
    return launch_wgrad_x6t_t<32, 2>(p, splits, pps, st);
  }
  if (pieces == 3) {
    if (bm == 128) return launch_wgrad_x6t_t<128, 3>(p, splits, pps, st);
    if (bm == 64) return launch_wgrad_x6t_t<64, 3>(p, splits, pps, st);
    return launch_wgrad_x6t_t<32, 3>(p, splits, pps, st);
  }
  if (bm == 128) return launch_wgrad_x6t_t<128, 1>(p, splits, pps, st);
  if (bm == 64) return launch_wgrad_x6t_t<64, 1>(p, splits, pps, st);
  return launch_wgrad_x6t_t<32, 1>(p, splits, pps, st);
}

template <int BM, int BN, int P>
static int launch_wgrad_x6_t(const WgradParams& p, int splits, hipStream_t st) {
  constexpr size_t lds = (size_t)2 * P * XH * (XStride<BM>::value + XStride<BN>::value);
  WgradParams q = p;
  q.nct = (int)cdiv(p.Cout, BM);
  q.ntiles = q.nct * (int)cdiv(p.KK, BN);
  q.nsplits = splits;
  q.tune = (splits % 8 == 0) ? 1 : 0;                // wgrad_plan rounds the split count to a multiple of 8 when it can
  dim3 grid(q.tune ? (unsigned)(splits * q.ntiles) : (unsigned)(8 * cdiv((long)splits * q.nct, 8) * (q.ntiles / q.nct)));
  hipLaunchKernelGGL((wgrad_x6_kernel<BM, BN, P>), grid, dim3(256), lds, st, q);
  XAS_LAUNCH_CHECK();
  return 0;
}

// tile of the bf16-split weight gradient: BM over Cout, BN over KK = R*S*Cin
void wgrad_x6_tile(int Cout, long KK, int* bm, int* bn) {
  *bm = Cout >= 96 ? 128 : (Cout > 32 ? 64 : 32);
  *bn = KK <= 64 ? 64 : 128;
  if (*bm == 32) *bn = 128;
}

int launch_wgrad_x6(const WgradParams& p, int bm, int bn, int splits, int pieces, hipStream_t st) {
  if (pieces == 2) {
    XAS_REQUIRE(p.a_amax && p.b_amax, "conv_wgrad: the f16x3 weight gradient needs the maxima of both tensor operands (xas_conv_shape.grad_amax, x_amax)");
    if (bm == 32) return launch_wgrad_x6_t<32, 128, 2>(p, splits, st);
    if (bm == 128) return bn == 128 ? launch_wgrad_x6_t<128, 128, 2>(p, splits, st) : launch_wgrad_x6_t<128, 64, 2>(p, splits, st);
    return bn == 128 ? launch_wgrad_x6_t<64, 128, 2>(p, splits, st) : launch_wgrad_x6_t<64, 64, 2>(p, splits, st);
  }
  if (bm == 32) return pieces == 3 ? launch_wgrad_x6_t<32, 128, 3>(p, splits, st) : launch_wgrad_x6_t<32, 128, 1>(p, splits, st);
  if (pieces == 3) {
    if (bm == 128) return bn == 128 ? launch_wgrad_x6_t<128, 128, 3>(p, splits, st) : launch_wgrad_x6_t<128, 64, 3>(p, splits, st);
    return bn == 128 ? launch_wgrad_x6_t<64, 128, 3>(p, splits, st) : launch_wgrad_x6_t<64, 64, 3>(p, splits, st);
  }
  if (bm == 128) return bn == 128 ? launch_wgrad_x6_t<128, 128, 1>(p, splits, st) : launch_wgrad_x6_t<128, 64, 1>(p, splits, st);
  return bn == 128 ? launch_wgrad_x6_t<64, 128, 1>(p, splits, st) : launch_wgrad_x6_t<64, 64, 1>(p, splits, st);
}

}  // namespace xas

using namespace xas;

// ------------------------------------------------------------------------------------
// all layers of a network in ONE launch: OIHW fp32 -> pre-split fragment-ordered planes (pack + split of every conv; the
// per-layer launches were 189 small kernels = 1.1 ms at the head of every step).  Descriptor = 12 int64 per entry:
// src, dst, Cout, Cin, R, S, transposed, planes, rows, K, first block, -.  Same arithmetic as xas_pack_weight followed by
// xas_split_weight (the packing is a pure permutation), so the results are bit-identical to the per-layer path.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void prepare_weights_kernel(const long* __restrict__ d, int n) {
  int lo = 0, hi = n - 1;                              // entry whose block range holds blockIdx.x
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (d[mid * 12 + 10] <= (long)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const long* e = d + lo * 12;
  const float* src = reinterpret_cast<const float*>(e[0]);
  unsigned short* dst = reinterpret_cast<unsigned short*>(e[1]);
  const int Cout = (int)e[2], Cin = (int)e[3], R = (int)e[4], S = (int)e[5], transposed = (int)e[6], P = (int)e[7];
  const int rows = (int)e[8], K = (int)e[9];
  const int nhc = K / 16;
  const long total = (long)((rows + 31) / 32) * nhc * 64;
  const long id = ((long)blockIdx.x - e[10]) * 256 + threadIdx.x;
  if (id >= total) return;
  const int lane = (int)(id & 63);
  const long blk = id >> 6;
  const int hc = (int)(blk % nhc), rb = (int)(blk / nhc);
  const int row = rb * 32 + (lane & 31), k0 = hc * 16 + (lane >> 5) * 8;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    v[j] = 0.f;
    if (row < rows) {
      const int k = k0 + j;
      int co, ci, tap;
      if (!transposed) { ci = k % Cin; tap = k / Cin; co = row; }
      else { co = k % Cout; tap = k / Cout; ci = row; }
      const int q = tap % S, r = tap / S;
      v[j] = src[(((long)co * Cin + ci) * R + r) * S + q];
    }
  }
  float4 r0 = make_float4(v[0], v[1], v[2], v[3]), r1 = make_float4(v[4], v[5], v[6], v[7]);
  unsigned short* o = dst + (blk * P) * 512 + lane * 8;
  if (P == 2) {                                        // two fp16 pieces of 2^10 w (split_weight_kernel<2>)
    r0.x *= kF16WScale; r0.y *= kF16WScale; r0.z *= kF16WScale; r0.w *= kF16WScale;
    r1.x *= kF16WScale; r1.y *= kF16WScale; r1.z *= kF16WScale; r1.w *= kF16WScale;
    f16_weight_check(r0, r1);
    const uint2 q0 = pack_f16x4(r0), q1 = pack_f16x4(r1);
    *reinterpret_cast<uint4*>(o) = make_uint4(q0.x, q0.y, q1.x, q1.y);
    r0 = sub_f16x4(r0, q0); r1 = sub_f16x4(r1, q1);
    const uint2 t0 = pack_f16x4(r0), t1 = pack_f16x4(r1);
    *reinterpret_cast<uint4*>(o + 512) = make_uint4(t0.x, t0.y, t1.x, t1.y);
    return;
  }
  for (int pc = 0; pc < P; ++pc) {
    const uint2 q0 = pack_bf16x4(r0), q1 = pack_bf16x4(r1);
    *reinterpret_cast<uint4*>(o + pc * 512) = make_uint4(q0.x, q0.y, q1.x, q1.y);
    if (pc + 1 < P) { r0 = sub_bf16x4(r0, q0); r1 = sub_bf16x4(r1, q1); }
  }
}

extern "C" int xas_prepare_weights(const void* descs, int n, long blocks, void* stream) {
  XAS_REQUIRE(descs && n > 0 && blocks > 0 && blocks < 0x7fffffffl, "prepare_weights: bad arguments");
  hipLaunchKernelGGL(prepare_weights_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), reinterpret_cast<const long*>(descs), n);
  XAS_LAUNCH_CHECK();
  return 0;
}

__global__ void weight_overflow_peek_kernel(unsigned* out) { *out = g_f16_weight_overflow ? 1u : 0u; }

// The flag without a device synchronisation: written (0 / 1) to a device word on `stream` (the caller copies that word to pinned host
// memory asynchronously and looks at it a step later - engine.TrainStep).  Does not clear the flag.
extern "C" int xas_f16_weight_overflow_peek(unsigned* device_out, void* stream) {
  XAS_REQUIRE(device_out, "f16_weight_overflow_peek: null output");
  hipLaunchKernelGGL(weight_overflow_peek_kernel, dim3(1), dim3(1), 0, as_stream(stream), device_out);
  XAS_LAUNCH_CHECK();
  return stem_weight_overflow_peek(device_out, stream);
}

extern "C" int xas_f16_weight_overflow(int reset) {
  unsigned v = 0u;
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_f16_weight_overflow), sizeof(v)) != hipSuccess) return -1;     // (synchronises)
  if (v && reset) {
    const unsigned z = 0u;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_f16_weight_overflow), &z, sizeof(z));
  }
  const int stem = stem_weight_overflow(reset);        // the stem kernel splits its weights itself (conv.hip)
  if (stem < 0) return -1;
  return (v || stem) ? 1 : 0;
}

extern "C" size_t xas_split_weight_bytes(long rows, long K, int pieces) {
  return (size_t)((rows + 31) / 32) * 32 * (size_t)K * 2 * (pieces == 3 ? 3 : (pieces == 2 ? 2 : 1));
}

extern "C" int xas_split_weight(const float* w_packed, void* w_split, long rows, long K, int pieces, void* stream) {
  XAS_REQUIRE(w_packed && w_split && rows > 0 && K > 0 && K % 16 == 0, "split_weight: need a packed weight [rows][K] with K a multiple of 16");
  XAS_REQUIRE(pieces >= 1 && pieces <= 3, "split_weight: pieces must be 1 (bf16), 2 (two fp16 pieces of 2^10 w: f16x3) or 3 (bf16x6)");
  XAS_REQUIRE((((uintptr_t)w_packed | (uintptr_t)w_split) & 15) == 0, "split_weight: buffers must be 16-byte aligned");
  XAS_REQUIRE(((rows + 31) / 32) * 32 * K * 6 < 0x7fffff00l, "split_weight: weight too large");
  const long total = ((rows + 31) / 32) * (K / 16) * 64;
  hipStream_t st = as_stream(stream);
  if (pieces == 3)
    hipLaunchKernelGGL(split_weight_kernel<3>, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, st, w_packed,
                       reinterpret_cast<unsigned short*>(w_split), (int)rows, (int)K, total);
  else if (pieces == 2)
    hipLaunchKernelGGL(split_weight_kernel<2>, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, st, w_packed,
                       reinterpret_cast<unsigned short*>(w_split), (int)rows, (int)K, total);
  else
    hipLaunchKernelGGL(split_weight_kernel<1>, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, st, w_packed,
                       reinterpret_cast<unsigned short*>(w_split), (int)rows, (int)K, total);
  XAS_LAUNCH_CHECK();
  return 0;
}
