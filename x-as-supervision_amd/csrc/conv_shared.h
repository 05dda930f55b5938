// Shared pieces of the convolution family (conv.hip: exact-fp32 MFMA kernels; conv_x6.hip: bf16-split kernels):
// problem descriptors, tile configuration, buffer-load helper, bf16 split helpers and the epilogue that both kernel
// families end in.  gfx950 only.
#pragma once
#include "common.h"

namespace xas {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int BK = 32;        // K-step (channels of one tap)
constexpr int LDK = BK + 4;   // padded LDS row, dwords

struct FastDiv {              // n / d and n % d for 0 <= n < 2^31, d >= 1
  unsigned d, mul, shift;
  __host__ void init(unsigned dd) {
    d = dd;
    if (dd == 1) { mul = 0; shift = 0; return; }
    unsigned s = 0;
    while ((1ull << s) < dd) ++s;
    unsigned long long m = ((1ull << (31 + s)) + dd - 1) / dd;   // ceil(2^(31+s)/d) fits 32 bits
    mul = (unsigned)m; shift = s;
  }
  __device__ __forceinline__ unsigned div(unsigned n) const {
    return d == 1 ? n : (unsigned)(((unsigned long long)n * mul) >> (31 + shift));
  }
};

struct IgemmParams {
  const float* src;    // activations that are gathered (x for fwd, dy for dgrad)
  const float* wgt;    // packed weights [rows][R][S][Cs]
  const float* bias;   // per output column or null
  float* out;
  int N;
  int Hs, Ws, Cs;      // gathered tensor dims
  int Hd, Wd, Cd;      // destination dims; Cd = number of GEMM columns
  int R, S, stride, pad;
  FastDiv div_hw, div_w;   // row -> (n, a, b) decode over the row grid
  int Hrow, Wrow;          // row grid (fwd: Ho x Wo; dgrad: per-phase grid, set in kernel)
  int tune;                // kernel-variant selectors kept for coverage tests (xas_set_tuning): bit5 plain K-loop, bit6 global-load kernel
  int nMt, nNt, mt_per_xcd;   // tile counts and M-tiles per XCD for the XCD-aware block order
  int t2d_tw;                 // != 0: the rows of a 128-row tile are an 8 x t2d_tw pixel patch of one image (t2d_tw = 16) or of two
                              // images (t2d_tw = 8) instead of 128 consecutive pixels (igemm_x6t_kernel); the epilogue maps rows with tile_row()
  int tpb;                    // igemm_x6p_kernel: consecutive M-tiles per block (mt_per_xcd then counts groups of tpb tiles)
  int xn, nt_per_x;           // bf16-split kernels: the 8 XCDs form an (8 / xn) x xn grid over (M-tiles, N-tiles); xn = 1: every XCD owns all N-tiles of its M-tiles
  long src_elems, wgt_elems;  // sizes of src / wgt (buffer-load kernel: range of the buffer descriptors)
  int accumulate;             // epilogue: 1: out += result (residual gradient already in the buffer);
                              //           2: out = result + relu'(mask) * acc_src (the skip gradient is formed here from
                              //              the block-output gradient and the sign bytes of xas_bn_apply: no dres tensor)
  const float* acc_src;       // accumulate == 2: [rows][Cd] like out
  const unsigned char* acc_mask;   // accumulate == 2: one byte per float4 of out, bit e = element active
  // fwd only (xas_conv_fwd_bnstats): != null -> every tile also emits, per output channel, sum(v - pivot) and
  // sum((v - pivot)^2) over its BM rows: stat_partial[tile row][channel][2].  The batch-norm statistics of the result are
  // then a reduction over (rows / BM) partial rows instead of a second pass over the activation.
  float* stat_partial;
  const float* stat_pivot;         // per channel or null (= 0)
  // dgrad only, stride 1 (xas_conv_dgrad_bn_bwd): the result is the gradient wrt the OUTPUT h = relu(bn(xb)) of a batch
  // norm; the epilogue applies the ReLU mask (re-derived from xb exactly as bn_bwd_reduce does), writes the masked
  // gradient dz and emits per tile and channel sum(dz), sum(dz * xhat): bnb_partial[tile][2][Cd].
  const float* bnb_x;              // != null enables the path; [rows][Cd] like out
  const float* bnb_mean; const float* bnb_var; const float* bnb_gamma; const float* bnb_beta;   // mean / var: [groups][Cd]
  float bnb_eps;
  int bnb_rows_per_group;
  float* bnb_partial;
  // f16x3 launches: device pointer to max |v| over the gathered tensor (activation or gradient): scale = the power of two
  // that puts the maximum just below 2^15 (required: a launch without it runs as bf16x6)
  const float* a_amax;
  // fwd only, 64 x 256 tiles (xas_conv_fwd_head): the result is the logit tensor of the soft-argmax head
  // (keypoint_detector_integral_multi.py:70-74: channel = joint * 64 + depth bin, 64 x 64 maps): every 64-pixel x 64-channel
  // staging pass of the epilogue - ONE image row of ONE joint - also emits that joint's online-softmax partial record
  // {max, sum e * w, sum e * h, sum_pixels e per depth bin} while the logits are still on chip: head_partial[image][HW / 64]
  // [joint][3 + 64], the layout head_finalize_kernel merges (head.hip).  The head's own pass over the logits disappears.
  float* head_partial;
};

constexpr int kMaxDevices = 16;
static inline int current_device() {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= kMaxDevices) d = 0;
  return d;
}

// shared MFMA core: As[BM][LDK], Bs[BN][LDK] -> acc
// ------------------------------------------------------------------------------------
template <int BM, int BN>
struct TileCfg {
  // 64 x 256 (bf16-split kernels, wide layers): the four waves sit side by side, each 64 rows x 64 columns - the same wave
  // tile as 128 x 128, but the activation rows a block loads, splits and stores serve twice the MFMAs
  static constexpr int WAVES_M = (BM == 64 && BN == 256) ? 1 : (BM >= 64 && BN >= 64) ? 2 : (BM < 64 ? 1 : 4);
  static constexpr int WAVES_N = 4 / WAVES_M;
  static constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  static constexpr int MI = WM / 32, NI = WN / 32;
  static_assert(MI >= 1 && NI >= 1, "tile too small");
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kOOB = 0x80000000u;

__device__ __forceinline__ float4 buf_load16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
constexpr int LDKH = BK + 8;   // bf16 elements per LDS row

__device__ __forceinline__ uint2 pack_bf16x4(float4 v) {
  const bf16x2_t lo = __builtin_convertvector(f32x2_t{v.x, v.y}, bf16x2_t);
  const bf16x2_t hi = __builtin_convertvector(f32x2_t{v.z, v.w}, bf16x2_t);
  return make_uint2(__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi));
}

// v - float(q) for the four bf16 values packed in q (exact: q is the leading part of v)
__device__ __forceinline__ float4 sub_bf16x4(float4 v, uint2 q) {
  return make_float4(v.x - __uint_as_float(q.x << 16), v.y - __uint_as_float(q.x & 0xffff0000u),
                     v.z - __uint_as_float(q.y << 16), v.w - __uint_as_float(q.y & 0xffff0000u));
}

// ---- two-piece fp16 split ("f16x3", forward passes of XAS_PREC_F16X3): x = h1 + h2 + e, h1 = fp16(x), h2 = fp16(x - h1),
// |e| <= 2^-22 |x| (11 + 11 significant bits, round to nearest); three of the four partial products are kept.
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
constexpr float kF16WScale = 1024.f;           // weights are split as 2^10 w: their second pieces stay clear of fp16's subnormals
                                               // (|w| < 64 assumed - larger weights become inf, loudly); results are scaled back
constexpr float kF16AScale = 16.f;             // (initial value of the per-launch tensor scale only: every f16x3 launch derives
                                               // its scales from the recorded maxima of its operands, f16_grad_scale below)
int stem_weight_overflow(int reset);           // conv.hip: flag of the stem kernel's in-kernel weight split (-1: read failed)
int stem_weight_overflow_peek(unsigned* device_out, void* stream);   // conv.hip: ORs that flag into *device_out on `stream`
constexpr float kF16Descale = 1.f / (kF16WScale * kF16AScale);
// scale of a gradient operand from its maximum magnitude (wave-uniform): amax in [2^(e-127), 2^(e-126)) -> 2^(141 - e), i.e.
// amax * scale in [2^14, 2^15).  Zero / denormal-range maxima: 1 (nothing to resolve).  *inv = 1 / scale (exact).
__device__ __forceinline__ float f16_grad_scale(const float* amax, float* inv) {
  const unsigned bits = __builtin_amdgcn_readfirstlane(__float_as_uint(read_amax(amax)));
  const int e = (int)((bits >> 23) & 0xffu);
  if (e < 40 || e > 250) { *inv = 1.f; return 1.f; }
  *inv = __uint_as_float((unsigned)(e - 14) << 23);           // 2^(e - 141)
  return __uint_as_float((unsigned)(268 - e) << 23);           // 2^(141 - e)
}

__device__ __forceinline__ uint2 pack_f16x4(float4 v) {
  const f16x2_t lo = {(_Float16)v.x, (_Float16)v.y}, hi = {(_Float16)v.z, (_Float16)v.w};      // v_cvt_pk_f16_f32 (RNE)
  return make_uint2(__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi));
}
__device__ __forceinline__ float4 sub_f16x4(float4 v, uint2 q) {
  const f16x2_t lo = __builtin_bit_cast(f16x2_t, q.x), hi = __builtin_bit_cast(f16x2_t, q.y);
  return make_float4(v.x - (float)lo.x, v.y - (float)lo.y, v.z - (float)hi.x, v.w - (float)hi.y);
}
// Both fp16 pieces of four fp32 values at scale s in EIGHT vector-ALU instructions: h1 = fp16(s v) by v_fma_mixlo/hi_f16 (the
// scale folded into the conversion, the pair packed by the instruction), h2 = fp16(s v - h1) by the same instruction reading
// h1 as an fp16 source (op_sel picks the half).  s v is exact (s is a power of two) and s v - h1 is exactly representable in
// fp32, so every result is rounded once: bit-identical to pack_f16x4 / sub_f16x4 up to the sign of a zero piece (fma(-0, s, +0)
// = +0; a zero piece contributes nothing either way) - tools/micro/split_f16_check.hip, 50 M values incl. inf / NaN / denormals.  What the
// compiler makes of the generic formulation is 14 instructions per float4 (it computes h1 twice: once unpacked as the fp16
// source of h2, once by v_mul + v_cvt_pk for the packed store) - and the conversion work, not the matrix pipe, is what the
// weight-gradient kernels wait for (both operands are split in the kernel: ~1 400 vector-ALU cycles against 768 MFMA cycles
// per K-step and wave before this change).
__device__ __forceinline__ void split_f16x4(float4 v, float s, uint2& h1, uint2& h2) {
  unsigned a, b, c, d;
  asm("v_fma_mixlo_f16 %0, %2, %3, 0\n\t"
      "v_fma_mixlo_f16 %1, %2, %5, 0\n\t"
      "v_fma_mixhi_f16 %0, %2, %4, 0\n\t"
      "v_fma_mixhi_f16 %1, %2, %6, 0"
      : "=&v"(a), "=&v"(b) : "v"(s), "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
  asm("v_fma_mixlo_f16 %0, %2, %3, -%7 op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixlo_f16 %1, %2, %5, -%8 op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %0, %2, %4, -%7 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %1, %2, %6, -%8 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
      : "=&v"(c), "=&v"(d) : "v"(s), "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w), "v"(a), "v"(b));
  h1 = make_uint2(a, b);
  h2 = make_uint2(c, d);
}
// the pieces of four values, smallest plane index first (P = 2: fp16 at scale s; P = 3 / 1: bf16, s unused)
template <int P> struct Pieces { uint2 q[P]; };
template <int P> __device__ __forceinline__ Pieces<P> split_pieces(float4 r, float s);
// piece formats by plane count: 3 / 1 planes = bf16 pieces, 2 planes = fp16 pieces
template <int P> __device__ __forceinline__ uint2 pack_piece4(float4 v) { return P == 2 ? pack_f16x4(v) : pack_bf16x4(v); }
template <int P> __device__ __forceinline__ float4 sub_piece4(float4 v, uint2 q) { return P == 2 ? sub_f16x4(v, q) : sub_bf16x4(v, q); }
template <int P> __device__ __forceinline__ Pieces<P> split_pieces(float4 r, float s) {
  Pieces<P> o;
  if constexpr (P == 2) {
#ifdef XAS_F16_SPLIT_GENERIC                       // (A/B builds: the compiler's 14-instruction form)
    r.x *= s; r.y *= s; r.z *= s; r.w *= s;
    o.q[0] = pack_f16x4(r);
    o.q[1] = pack_f16x4(sub_f16x4(r, o.q[0]));
#else
    split_f16x4(r, s, o.q[0], o.q[1]);
#endif
  } else {
#pragma unroll
    for (int pc = 0; pc < P; ++pc) {
      o.q[pc] = pack_bf16x4(r);
      if (pc + 1 < P) r = sub_bf16x4(r, o.q[pc]);
    }
  }
  return o;
}
// kept partial products (a piece, b piece), smallest first
template <int P> struct Products;
template <> struct Products<3> { static constexpr int N = 6; static constexpr int A[6] = {2, 0, 1, 1, 0, 0}; static constexpr int B[6] = {0, 2, 1, 0, 1, 0}; };
template <> struct Products<2> { static constexpr int N = 3; static constexpr int A[3] = {1, 0, 0}; static constexpr int B[3] = {0, 1, 0}; };
template <> struct Products<1> { static constexpr int N = 1; static constexpr int A[1] = {0}; static constexpr int B[1] = {0}; };
// XAS_FRAG_AGPR (experiment, r05): a matrix-instruction operand fragment is moved to the ACCUMULATION half of the register file
// once, where it is formed; the K-doubled matrix instructions then read their 128-bit A / B operands from there.  With the
// operands read from AGPRs the multi-stream hazard of DESIGN.md section 5 did not show (0 / 500 steps on the three-stream schedule,
// shipped build on the same lease 11 / 200: profiles/r05_hazard_census_agpr.txt).
__device__ __forceinline__ uint4 frag_reg(uint4 v) {
#ifdef XAS_FRAG_AGPR
  typedef unsigned u32x4v_ __attribute__((ext_vector_type(4)));
  u32x4v_ t = {v.x, v.y, v.z, v.w}, o;
  asm("" : "=a"(o) : "0"(t));
  return make_uint4(o[0], o[1], o[2], o[3]);
#else
  return v;
#endif
}
template <int A_, int B_>
__device__ __forceinline__ void frag_regs(uint4 (&f)[A_][B_]) {
#ifdef XAS_FRAG_AGPR
#pragma unroll
  for (int i = 0; i < A_; ++i)
#pragma unroll
    for (int j = 0; j < B_; ++j) f[i][j] = frag_reg(f[i][j]);
#endif
}
template <int A_>
__device__ __forceinline__ void frag_regs(uint4 (&f)[A_]) {
#ifdef XAS_FRAG_AGPR
#pragma unroll
  for (int i = 0; i < A_; ++i) f[i] = frag_reg(f[i]);
#endif
}

template <int P>
__device__ __forceinline__ f32x16 mfma_piece(uint4 a, uint4 b, f32x16 c) {
#ifdef XAS_MFMA_X8
  // diagnosis build: the K = 16 matrix instructions of gfx950 replaced by two K = 8 ones of the previous generation (the low /
  // high 64 bits of each lane's operand pair up the same k indices on both sides, so every dot product has the same terms)
  typedef short s16x4v __attribute__((ext_vector_type(4)));
  typedef _Float16 f16x4v __attribute__((ext_vector_type(4)));
  const uint2 alo = make_uint2(a.x, a.y), ahi = make_uint2(a.z, a.w), blo = make_uint2(b.x, b.y), bhi = make_uint2(b.z, b.w);
  if constexpr (P == 2) {
    c = __builtin_amdgcn_mfma_f32_32x32x8f16(__builtin_bit_cast(f16x4v, alo), __builtin_bit_cast(f16x4v, blo), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x8f16(__builtin_bit_cast(f16x4v, ahi), __builtin_bit_cast(f16x4v, bhi), c, 0, 0, 0);
  } else {
    c = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(__builtin_bit_cast(s16x4v, alo), __builtin_bit_cast(s16x4v, blo), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(__builtin_bit_cast(s16x4v, ahi), __builtin_bit_cast(s16x4v, bhi), c, 0, 0, 0);
  }
#endif
#ifdef XAS_MFMA_16X16
  // diagnosis build (WRONG RESULTS ON PURPOSE - only run-to-run reproducibility is looked at): gfx950's other new shape,
  // 16x16x32, in place of 32x32x16 - four of them, one per quarter of the accumulator, on the same operand registers.  Every
  // accumulator element still receives a 32-term dot product of operand values of the right scale, so the step stays finite.
  {
    typedef float f32x4v __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4v t = {c[4 * q], c[4 * q + 1], c[4 * q + 2], c[4 * q + 3]};
      if constexpr (P == 2) t = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), t, 0, 0, 0);
      else t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), t, 0, 0, 0);
      c[4 * q] = t[0]; c[4 * q + 1] = t[1]; c[4 * q + 2] = t[2]; c[4 * q + 3] = t[3];
    }
    return c;
  }
#endif
#ifdef XAS_MFMA_AGPR_OPERANDS
  // diagnosis build: the A / B operands of the matrix instruction come from ACCUMULATION registers (the unified file's upper
  // half; copied there by v_accvgpr_write): does the multi-stream hazard depend on where the K-doubled instruction reads its
  // 128-bit operands from?  (A production form would load the fragments straight into AGPRs - ds_read / buffer_load can.)
  {
    typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
    u32x4v av = {a.x, a.y, a.z, a.w}, bv = {b.x, b.y, b.z, b.w}, aa, ab;
    asm volatile("" : "=a"(aa) : "0"(av));
    asm volatile("" : "=a"(ab) : "0"(bv));
    a = make_uint4(aa[0], aa[1], aa[2], aa[3]);
    b = make_uint4(ab[0], ab[1], ab[2], ab[3]);
  }
#endif
#ifdef XAS_MFMA_AGPR_DUMMY
  // diagnosis control for XAS_MFMA_AGPR_OPERANDS: the same eight copies into accumulation registers before every matrix
  // instruction, but the instruction still reads the ORIGINAL (vector-register) operands: is it the copies or the operand source?
  {
    typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
    u32x4v av = {a.x, a.y, a.z, a.w}, bv = {b.x, b.y, b.z, b.w}, aa, ab;
    asm volatile("" : "=a"(aa) : "0"(av));
    asm volatile("" : "=a"(ab) : "0"(bv));
    asm volatile("" :: "a"(aa), "a"(ab));
  }
#endif
#ifdef XAS_MFMA_X16_TWICE
  // diagnosis control for XAS_MFMA_X8: the K = 16 instruction issued TWICE (second result thrown away): the timing of the K = 8
  // build with the instruction of the shipped one
  {
    f32x16 waste = c;
    if constexpr (P == 2) waste = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), waste, 0, 0, 0);
    else waste = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), waste, 0, 0, 0);
    asm volatile("" :: "v"(waste));
  }
#endif
#ifdef XAS_MFMA_NOP
  // diagnosis build: XAS_MFMA_NOP idle issue slots of the wave after every matrix instruction (is it the issue density?)
  if constexpr (P == 2) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  else c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_nop %0" :: "n"(XAS_MFMA_NOP));
  __builtin_amdgcn_sched_barrier(0);
  return c;
#else
  if constexpr (P == 2) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
#endif
}

// dgrad epilogue with the batch-norm backward reduction folded in (see IgemmParams::bnb_x).  Stride 1: output row = m.
// Per (channel quad) the norm's parameters are loaded and 1/std formed ONCE, then applied to the MI row blocks; the two
// sums are pre-reduced over the row blocks in registers, so both [WAVES_M * 32][BN] arrays fit the operand LDS together:
// one staging pass, one barrier, one column pass.
template <int BM, int BN>
__device__ __forceinline__ void bnb_epilogue(const IgemmParams& p, f32x16 (&acc)[TileCfg<BM, BN>::MI][TileCfg<BM, BN>::NI],
                                             int m0, int n0, int wm, int wn, int lane, float* lds) {
  using C = TileCfg<BM, BN>;
  constexpr int LDT = BN + 4, TR = C::WAVES_M * 32;               // staged rows per array
  constexpr int PARTS = 256 / BN;
  float* T1 = lds;
  float* T2 = lds + TR * LDT;
  float* red = lds + 2 * TR * LDT;                                 // [PARTS][BN][2]
  static_assert((2 * TR * LDT + 2 * 256) <= 2 * (BM + BN) * LDK, "bn-backward staging does not fit the operand LDS");
  const int pix_l = lane & 31, csub = 4 * (lane >> 5);
  const int grp = m0 / p.bnb_rows_per_group;                       // tiles never straddle a group (launcher)
  const float* mean = p.bnb_mean + (size_t)grp * p.Cd;
  const float* var = p.bnb_var + (size_t)grp * p.Cd;
  const size_t row0 = (size_t)(m0 + wm * C::WM + pix_l);           // < Mrows: tiles are full (launcher)
  __syncthreads();                                                 // operand buffers are free
#pragma unroll
  for (int ni = 0; ni < C::NI; ++ni)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int nl = wn * C::WN + ni * 32 + 8 * g + csub, n = n0 + nl;
      float4 xv[C::MI];
#pragma unroll
      for (int mi = 0; mi < C::MI; ++mi) xv[mi] = *reinterpret_cast<const float4*>(p.bnb_x + (row0 + mi * 32) * p.Cd + n);
      const float4 mu = *reinterpret_cast<const float4*>(mean + n);
      const float4 vr = *reinterpret_cast<const float4*>(var + n);
      const float4 gm = *reinterpret_cast<const float4*>(p.bnb_gamma + n);
      const float4 bt = *reinterpret_cast<const float4*>(p.bnb_beta + n);
      const float mu_[4] = {mu.x, mu.y, mu.z, mu.w}, bt_[4] = {bt.x, bt.y, bt.z, bt.w};
      float rstd[4], rsg[4];
      rstd[0] = rsqrtf(vr.x + p.bnb_eps); rstd[1] = rsqrtf(vr.y + p.bnb_eps);
      rstd[2] = rsqrtf(vr.z + p.bnb_eps); rstd[3] = rsqrtf(vr.w + p.bnb_eps);
      rsg[0] = __fmul_rn(rstd[0], gm.x); rsg[1] = __fmul_rn(rstd[1], gm.y);
      rsg[2] = __fmul_rn(rstd[2], gm.z); rsg[3] = __fmul_rn(rstd[3], gm.w);
      float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int mi = 0; mi < C::MI; ++mi) {
        const float x_[4] = {xv[mi].x, xv[mi].y, xv[mi].z, xv[mi].w};
        float dz[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          dz[e] = bn_affine(x_[e], mu_[e], rsg[e], bt_[e]) > 0.f ? acc[mi][ni][4 * g + e] : 0.f;
          s1[e] += dz[e];
          s2[e] = fmaf(dz[e], (x_[e] - mu_[e]) * rstd[e], s2[e]);
        }
        *reinterpret_cast<float4*>(p.out + (row0 + mi * 32) * p.Cd + n) = make_float4(dz[0], dz[1], dz[2], dz[3]);
      }
      *reinterpret_cast<float4*>(T1 + (wm * 32 + pix_l) * LDT + nl) = make_float4(s1[0], s1[1], s1[2], s1[3]);
      *reinterpret_cast<float4*>(T2 + (wm * 32 + pix_l) * LDT + nl) = make_float4(s2[0], s2[1], s2[2], s2[3]);
    }
  __syncthreads();
  const int c = threadIdx.x % BN, part = threadIdx.x / BN;
  float a1 = 0.f, a2 = 0.f;
#pragma unroll 8
  for (int r = part; r < TR; r += PARTS) { a1 += T1[r * LDT + c]; a2 += T2[r * LDT + c]; }
  red[(part * BN + c) * 2] = a1; red[(part * BN + c) * 2 + 1] = a2;
  __syncthreads();
  if (part == 0) {
#pragma unroll
    for (int k = 1; k < PARTS; ++k) { a1 += red[(k * BN + c) * 2]; a2 += red[(k * BN + c) * 2 + 1]; }
    float* prow = p.bnb_partial + (size_t)(m0 / BM) * 2 * p.Cd;
    prow[n0 + c] = a1;
    prow[p.Cd + n0 + c] = a2;
  }
}

// Output row (pixel index n*H*W + y*W + x) of local row ml of tile m0 / 128 when the tile is a 2-D patch (IgemmParams::t2d_tw).
__device__ __forceinline__ size_t tile_row(const IgemmParams& p, int m0, int ml) {
  const int tws = p.t2d_tw == 16 ? 4 : 3, tw = p.t2d_tw;
  const int tn = ml >> (3 + tws), ty = (ml >> tws) & 7, tx = ml & (tw - 1);
  const int tiles_x = p.Wd / tw, per_img = tiles_x * (p.Hd >> 3);
  const int tile = m0 >> 7, tn_cnt = 128 >> (3 + tws);
  const int img = (tile / per_img) * tn_cnt + tn, t = tile % per_img;
  const int y = (t / tiles_x) * 8 + ty, x = (t % tiles_x) * tw + tx;
  return ((size_t)img * p.Hd + y) * p.Wd + x;
}

// Epilogue shared by the igemm kernels: accumulators -> global memory (+ bias), float4 per four output channels.
// HALVES = 2: the LDS staging of the row / statistics epilogue is done in two passes over BN / 2 columns each, so that the
// staging area is half as large (the bf16-split kernels need 36.9 KB of LDS for their operands; a full 128 x 128 staging
// tile would double their footprint and halve their residency).
template <int BM, int BN, int MODE, bool BNB = false, int HALVES = 1, bool HEAD = true>
__device__ __forceinline__ void igemm_epilogue(const IgemmParams& p, f32x16 (&acc)[TileCfg<BM, BN>::MI][TileCfg<BM, BN>::NI],
                                               f32x16& acc2, int m0, int n0, int wm, int wn, int lane, int Mrows, int HW,
                                               int Wrow, int ph, int pw, float* lds = nullptr) {
  using C = TileCfg<BM, BN>;
  if (C::MI == 1 && C::NI == 1) {
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[0][0][e] += acc2[e];
  }
  if constexpr (MODE == 1 && BNB) {                  // own instantiation: the extra live registers of this path would
    bnb_epilogue<BM, BN>(p, acc, m0, n0, wm, wn, lane, lds);   // otherwise cost the plain kernel its second block per CU
    return;
  }
  // ---- per-channel sums of the tile for the batch norm that follows (forward only).  The accumulators go through the
  // (now idle) operand LDS as T[pixel][channel]; after the global stores below, thread t sums column t % BNH over the rows
  // t / BNH, t / BNH + PARTS, ...  Launcher guarantees: every tile is full in M (rows-per-group % BM == 0), lds != null.
  constexpr int BNH = BN / HALVES;
  constexpr int LDT = BNH + 4;
  static_assert(HALVES == 1 || (C::WN <= BNH && BNH % 32 == 0), "a wave's columns must fall into one staging half");
  const bool want_stats = MODE == 0 && p.stat_partial != nullptr;
  // Row epilogue (default for full tiles): the tile goes through the idle operand LDS as T[pixel][channel] and is written
  // to memory whole rows at a time - a wave stores 1 KiB of contiguous channels per instruction, old values / sign bytes
  // of the accumulating forms are read the same way - instead of 64 scattered 16-byte pieces per instruction straight
  // from the MFMA register layout (64 -> 256 channels at 256 x 64 x 64: 0.671 -> 0.547 ms, r02).
  // Plain (non-accumulating) data gradients keep the register epilogue: their long K loops gain nothing and pay the extra
  // barrier (215.3 vs 215.9 ms/step, r02).
  const bool rows_from_lds = lds != nullptr && (p.Cd & 3) == 0 && m0 + BM <= Mrows && n0 + BN <= p.Cd &&
                             !(MODE == 1 && !p.accumulate);
  // ---- register epilogue.  Accumulator layout (operands swapped): column = lane & 31 = pixel row m of the tile,
  // row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) = output channel -> registers 4g..4g+3 are four
  // consecutive channels of one pixel: one 16-byte store each, row address computed once per lane.
  const int pix_l = lane & 31, csub = 4 * (lane >> 5);
  const bool vec_ok = (p.Cd & 3) == 0;
#pragma unroll
  for (int mi = 0; mi < C::MI; ++mi) {
    const int m = m0 + wm * C::WM + mi * 32 + pix_l;
    if (m >= Mrows || rows_from_lds) continue;
    size_t orow;
    if (p.t2d_tw) orow = tile_row(p, m0, m - m0);
    else if (MODE == 0) orow = (size_t)m;
    else {
      const int n = m / HW; const int rem = m - n * HW; const int a = rem / Wrow, b = rem - a * Wrow;
      orow = ((size_t)n * p.Hd + (ph + p.stride * a)) * p.Wd + (pw + p.stride * b);
    }
    float* orow_p = p.out + orow * p.Cd;
    // accumulating form (out += result): ALL the old values of this row are requested before the first one is used -
    // one load at a time (load, wait, add, store) left the epilogue waiting out a full memory round trip per float4
    // (rocprof: xas_conv_dgrad_acc at 46 TFLOP/s against 104 for the same shapes without the accumulation)
    float4 prev[C::NI][4];
    unsigned pmask[C::NI][4];
    if (p.accumulate && vec_ok) {
      const float* prow = p.accumulate == 2 ? p.acc_src + orow * p.Cd : orow_p;
#pragma unroll
      for (int ni = 0; ni < C::NI; ++ni)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = n0 + wn * C::WN + ni * 32 + 8 * g + csub;
          const int nc = n + 3 < p.Cd ? n : p.Cd - 4;                                   // clamped: unconditional load
          prev[ni][g] = *reinterpret_cast<const float4*>(prow + nc);
          pmask[ni][g] = p.accumulate == 2 ? p.acc_mask[(orow * p.Cd + nc) >> 2] : 15u;
        }
    }
#pragma unroll
    for (int ni = 0; ni < C::NI; ++ni) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = n0 + wn * C::WN + ni * 32 + 8 * g + csub;
        float4 v = make_float4(acc[mi][ni][4 * g], acc[mi][ni][4 * g + 1], acc[mi][ni][4 * g + 2], acc[mi][ni][4 * g + 3]);
        if (vec_ok && n + 3 < p.Cd) {
          if (MODE == 0 && p.bias) {
            const float4 bb = *reinterpret_cast<const float4*>(p.bias + n);
            v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
          }
          if (p.accumulate) {
            const float4 o = prev[ni][g];
            const unsigned mb = pmask[ni][g];
            v.x += (mb & 1u) ? o.x : 0.f; v.y += (mb & 2u) ? o.y : 0.f;
            v.z += (mb & 4u) ? o.z : 0.f; v.w += (mb & 8u) ? o.w : 0.f;
          }
          *reinterpret_cast<float4*>(orow_p + n) = v;      // (scattered 16-byte pieces: a non-temporal hint costs 2.3 ms here)
        } else {
          const float vv[4] = {v.x, v.y, v.z, v.w};
          for (int e = 0; e < 4; ++e)
            if (n + e < p.Cd)
              orow_p[n + e] = vv[e] + ((MODE == 0 && p.bias) ? p.bias[n + e] : 0.f) + (p.accumulate ? orow_p[n + e] : 0.f);
        }
      }
    }
  }
  const bool want_head = HEAD && MODE == 0 && BM == 64 && BN / HALVES == 64 && p.head_partial != nullptr;
  if (!(want_stats || rows_from_lds || want_head)) return;
  // ---- staged passes: T[pixel][channel of this half] -> whole rows to memory and / or per-channel sums
  constexpr int C4 = BNH / 4, RPP = 256 / C4, NR = BM / RPP;   // float4 per row, rows per pass of the block, rows per thread
  static_assert(BM % RPP == 0, "row pass does not tile the block rows");
  const int c4 = threadIdx.x % C4, r0 = threadIdx.x / C4;
  auto out_row = [&](int r) -> size_t {
    const int m = m0 + r;
    if (p.t2d_tw) return tile_row(p, m0, r);
    if (MODE == 0 || p.stride == 1) return (size_t)m;
    const int n = m / HW; const int rem = m - n * HW; const int a = rem / Wrow, b = rem - a * Wrow;
    return ((size_t)n * p.Hd + (ph + p.stride * a)) * p.Wd + (pw + p.stride * b);
  };
#pragma unroll
  for (int hf = 0; hf < HALVES; ++hf) {
    const int nh0 = n0 + hf * BNH;                     // first global channel of this half
    const int nn = nh0 + c4 * 4;
    // accumulating forms: the old values / skip-gradient rows and sign bytes of this pass are requested BEFORE the staging
    // and its two barriers, so that their HBM latency is spent there (the short-K layers that use these forms are all
    // epilogue: 3.6 TB/s with the loads issued after the barriers)
    float4 prevv[NR];
    unsigned pmask[NR];
    const bool pre = rows_from_lds && p.accumulate;
    if (pre) {
#pragma unroll
      for (int j = 0; j < NR; ++j) {
        const size_t orow = out_row(r0 + RPP * j);
        prevv[j] = stream_load(reinterpret_cast<const float4*>((p.accumulate == 2 ? p.acc_src : p.out) + orow * p.Cd + nn));
        pmask[j] = p.accumulate == 2 ? p.acc_mask[(orow * p.Cd + nn) >> 2] : 15u;
      }
    }
    __syncthreads();                                   // the operand buffers (first pass) / the previous half's tile are free
    {
      const int pl = lane & 31, cs = 4 * (lane >> 5);
#pragma unroll
      for (int mi = 0; mi < C::MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < C::NI; ++ni) {
          const int cl = wn * C::WN + ni * 32 - hf * BNH;           // first channel of this accumulator inside the half (wave-uniform)
          if (HALVES == 1 || (cl >= 0 && cl < BNH)) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
              *reinterpret_cast<float4*>(lds + (wm * C::WM + mi * 32 + pl) * LDT + cl + 8 * g + cs) =
                  make_float4(acc[mi][ni][4 * g], acc[mi][ni][4 * g + 1], acc[mi][ni][4 * g + 2], acc[mi][ni][4 * g + 3]);
          }
        }
    }
    __syncthreads();                                   // T complete
    if (rows_from_lds) {
      float4 bb = make_float4(0, 0, 0, 0);
      if (MODE == 0 && p.bias) bb = *reinterpret_cast<const float4*>(p.bias + nn);
#pragma unroll
      for (int j = 0; j < NR; ++j) {
        const int r = r0 + RPP * j;
        const size_t orow = out_row(r);
        float4 v = *reinterpret_cast<const float4*>(lds + r * LDT + c4 * 4);
        v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
        if (p.accumulate) {
          const float4 o = prevv[j];
          const unsigned mb = pmask[j];
          v.x += (mb & 1u) ? o.x : 0.f; v.y += (mb & 2u) ? o.y : 0.f;
          v.z += (mb & 4u) ? o.z : 0.f; v.w += (mb & 8u) ? o.w : 0.f;
        }
#ifdef XAS_CONV_ROWS_PLAIN_STORE
        *reinterpret_cast<float4*>(p.out + orow * p.Cd + nn) = v;
#else
        stream_store(reinterpret_cast<float4*>(p.out + orow * p.Cd + nn), v);
#endif
      }
    }
    if constexpr (HEAD && MODE == 0 && BM == 64 && BNH == 64) {
      if (p.head_partial != nullptr && nh0 < p.Cd) {
        // one joint (64 depth bins = the 64 channels of this half) over the 64 pixels of one image row: T[pixel][bin] + bias
        const int tid = threadIdx.x;
        const int c = tid & 63, part = tid >> 6;           // thread: depth bin c, pixels part, part + 4, ...
        const float bc = p.bias ? p.bias[nh0 + c] : 0.f;
        float v[16];
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < 16; ++j) { v[j] = lds[(part + 4 * j) * LDT + c] + bc; mx = fmaxf(mx, v[j]); }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        float* red = lds + BM * LDT;                       // 512 floats behind the tile
        if (c == 0) red[part] = mx;
        __syncthreads();
        const float Mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        float z = 0.f, sx = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const float e = __expf(v[j] - Mx);
          z += e;
          sx = fmaf(e, (float)(part + 4 * j), sx);         // pixel r of the tile row = image column w (m0 % 64 == 0)
        }
        red[8 + part * 64 + c] = z;                        // per (pixel part, bin)
        float zs = z;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { zs += __shfl_xor(zs, o, 64); sx += __shfl_xor(sx, o, 64); }
        if (c == 0) { red[264 + part] = zs; red[268 + part] = sx; }
        __syncthreads();
        const int img = m0 / HW, pix0 = m0 - img * HW;     // (forward: HW = Ho * Wo = 4096; the tile is one image row)
        float* rec = p.head_partial + (((size_t)img * (HW >> 6) + (pix0 >> 6)) * (p.Cd >> 6) + (nh0 >> 6)) * 67;
        if (part == 0) rec[3 + c] = (red[8 + c] + red[8 + 64 + c]) + (red[8 + 128 + c] + red[8 + 192 + c]);
        if (tid == 64) {
          const float tot = (red[264] + red[265]) + (red[266] + red[267]);
          rec[0] = Mx;
          rec[1] = (red[268] + red[269]) + (red[270] + red[271]);
          rec[2] = tot * (float)(pix0 >> 6);               // every pixel of the tile has the same h
        }
      }
    }
    if (MODE == 0 && want_stats) {
      constexpr int PARTS = 256 / BNH;
      const int tid = threadIdx.x;
      const int c = tid % BNH, part = tid / BNH;
      const int n = nh0 + c;
      const float pv = (p.stat_pivot && n < p.Cd) ? p.stat_pivot[n] : 0.f;
      float s = 0.f, q = 0.f;
#pragma unroll 8
      for (int r = part; r < BM; r += PARTS) {
        const float v = lds[r * LDT + c] - pv;
        s += v; q = fmaf(v, v, q);
      }
      float* red = lds + BM * LDT;                       // [PARTS][BNH][2]
      red[(part * BNH + c) * 2] = s; red[(part * BNH + c) * 2 + 1] = q;
      __syncthreads();
      if (part == 0 && n < p.Cd) {
#pragma unroll
        for (int k = 1; k < PARTS; ++k) { s += red[(k * BNH + c) * 2]; q += red[(k * BNH + c) * 2 + 1]; }
        *reinterpret_cast<float2*>(p.stat_partial + ((size_t)(m0 / BM) * p.Cd + n) * 2) = make_float2(s, q);
      }
    }
  }
}


struct WgradParams {
  const float* x; const float* dy; float* out;      // out: [splits][Cout][KK] slabs
  int N, Hi, Wi, Cin, Cout, R, S, stride, pad, Ho, Wo;
  int KK;                                            // R*S*Cin
  int M;                                             // N*Ho*Wo
  int m_per_split;
  int nct, ntiles, nsplits;                           // Cout tiles, tiles per split, pixel splits
  int tune;
  int t2d_tw, pps;                                    // wgrad_x6t_kernel: patch width (16 / 8), pixel patches per split
  const float* a_amax; const float* b_amax;           // f16x3 kernels: device pointers to max |dy| / max |x| when that operand is a
                                                      // gradient tensor (scale of its fp16 pieces); null: an activation (fixed scale)
  FastDiv div_hw, div_w, div_cin, div_s;
};

constexpr int WBK = 32;       // pixels per K-step

// Tile of the forward / data-gradient kernels for a problem (one rule for both kernel families, the launchers and
// xas_conv_fwd_bnstats / xas_conv_dgrad_bn_bwd, whose partial-sum grids follow the tile grid).
// stride-1 3x3 problems the tap re-use kernels take (igemm_x6t_kernel): 128-row tiles that are whole 8 x 16 / 8 x 8 patches
static inline bool tap_tile_ok(int R, int S, int stride, int pad, int Hs, int Ws, int Hd, int Wd, int Cs, int N) {
  if (R != 3 || S != 3 || stride != 1 || pad != 1 || Hd != Hs || Wd != Ws || Cs % 32 != 0 || Hd % 8 != 0) return false;
  return Wd % 16 == 0 || (Wd == 8 && N % 2 == 0);
}

#ifndef XAS_TILE128_MIN_BLOCKS
#define XAS_TILE128_MIN_BLOCKS 256     // fewer 128 x 128 tiles than this: 64 x 64 tiles (r03 sweep, in-box: 512 -> 256 -1.4 ms/step, 128 / 192 the same, 1024 +2.6)
#endif
static inline void pick_tile(int Cd, long Mrows_max, int phases, int* bm, int* bn, bool wide = false) {
  if (Cd >= 96) {
    // small problems (layer3/4: M = 8192 / 2048 rows per 32 images): 128x128 tiles leave most of the 256 CUs idle
    const long blocks128 = cdiv(Mrows_max, 128) * cdiv(Cd, 128) * phases;
    if (blocks128 <= XAS_TILE128_MIN_BLOCKS) { *bm = 64; *bn = 64; }
    else if (wide && Cd % 256 == 0) { *bm = 64; *bn = 256; }      // bf16-split kernels only (TileCfg<64, 256>)
    else { *bm = 128; *bn = 128; }
  } else if (Cd >= 48) { *bm = 128; *bn = 64; }
  else { *bm = 128; *bn = 32; }
}

// conv_x6.hip: bf16-split kernels (pieces = 3: bf16x6, fp32 accurate; 1: plain bf16).  mode: 0 forward, 1 data gradient.
// p.wgt points to PRE-SPLIT weights (xas_split_weight).
int launch_igemm_x6(const IgemmParams& p, int mode, int Mrows_max, int phases, int pieces, hipStream_t st);
void wgrad_x6_tile(int Cout, long KK, int* bm, int* bn);
int launch_wgrad_x6(const WgradParams& p, int bm, int bn, int splits, int pieces, hipStream_t st);
bool wgrad_x6t_plan(int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad, int Ho, int Wo, int* bm, int* splits, int* pps);
int launch_wgrad_x6t(const WgradParams& p, int bm, int splits, int pps, int pieces, hipStream_t st);

}  // namespace xas
