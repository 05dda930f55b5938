// HBM-bound NHWC layer kernels: training-mode batch norm (stats / apply / backward),
// 3x3-s2 max pool, bilinear x2 upsample, sigmoid, NCHW<->NHWC.  gfx950.
//
// Replaces ATen batch_norm_stats / batch_norm_elemt / batch_norm_backward_{reduce,elemt}
// (nn.BatchNorm2d + nn.SyncBatchNorm: resnet.py:18,40, deconv_head.py:30,
// physique_network.py:18,25,33), max_pool2d (resnet.py:20), upsample_bilinear2d
// (physique_network.py:31) and sigmoid (physique_network.py:57).
// All kernels move float4 (4 channels) per lane: a wave touches 1 KiB of contiguous rows.
#include "common.h"

namespace xas {


// ---------------------------------------------------------------- column reductions
// x is [M][C] = `G` independent groups of Mg = M / G consecutive rows (the camera-batched step sends the images of
// all cameras through the detector as ONE tensor; every camera keeps its own batch statistics, exactly as the
// reference's per-camera calls, model.py:64,147).  A block owns CB = min(C,256) channels (TX = CB/4 lanes along C)
// and a slab of rows of ONE group; it emits per channel sum(f1), sum(f2) of two per-element functions.
// The finalize pass is folded into the same launch: the block that arrives LAST at a ticket counter (one counter
// per channel block) sums the slab partials of all groups in a fixed order (deterministic) and writes the results
// (rocprof, round 1: 1 262 separate finalize launches per step were pure latency, 13 ms).
struct ColGeom { int TX, TY, CB, ncb, nslab, G; long rows_per_slab, Mg; };

static int col_geom(long M, int C, int G, ColGeom* g) {
  XAS_REQUIRE(M > 0 && C >= 4 && C % 4 == 0, "column reduce: channel count %d must be a positive multiple of 4", C);
  XAS_REQUIRE(G >= 1 && G <= 64 && M % G == 0, "column reduce: %ld rows do not split into %d equal groups", M, G);
  // largest power of two <= 64 dividing C: a 64-channel block still moves 256-byte row segments, and the block that
  // finalizes a channel block reads G * nslab * 512 B of partials (one CU pulls ~100 GB/s: keep that in the tens of KB)
  int cb = 4;
  while (cb < 64 && C % (cb * 2) == 0) cb *= 2;
  g->CB = cb; g->G = G; g->Mg = M / G;
  g->TX = g->CB / 4; g->TY = 256 / g->TX; g->ncb = C / g->CB;
  static const long kWant[4] = {256, 512, 128, 64};           // tune bits 15-16 (experiment)
  long want = kWant[(tune_flags() >> 15) & 3] * (G > 1 ? 2 : 1) / ((long)g->ncb * G);   // ~1-2 blocks per CU in total
  long maxslab = cdiv(g->Mg, (long)g->TY * 8);   // at least 8 rows per thread
  if (want > maxslab) want = maxslab;
  if (want < 1) want = 1;
  g->rows_per_slab = cdiv(g->Mg, want);
  g->nslab = (int)cdiv(g->Mg, g->rows_per_slab);
  return 0;
}

// Ticket counters of the folded finalize: a zero-initialised pool per device; a launch takes `ncb` consecutive words,
// the last arriver of each word resets it, so a word is zero again whenever a later launch re-uses it (launches that
// share a word are thousands of launches apart).
constexpr int kTicketWords = 65536;
static unsigned* g_tickets[16] = {};
static unsigned g_ticket_next[16] = {};

static unsigned* take_tickets(int n) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  if (!g_tickets[dev]) {
    unsigned* p = nullptr;
    if (hipMalloc(&p, kTicketWords * sizeof(unsigned)) != hipSuccess) return nullptr;
    if (hipMemset(p, 0, kTicketWords * sizeof(unsigned)) != hipSuccess) return nullptr;
    g_tickets[dev] = p;
  }
  if (g_ticket_next[dev] + (unsigned)n > (unsigned)kTicketWords) g_ticket_next[dev] = 0;
  unsigned* r = g_tickets[dev] + g_ticket_next[dev];
  g_ticket_next[dev] += (unsigned)n;
  return r;
}

// Parameter-gradient accumulation into the .grad arena: hardware float atomics (global_atomic_add_f32), because two
// backward chains of one step (xas_amd/streams.py: chains) may finish the same layer's reduction concurrently on two
// streams.  A parameter receives at most one contribution per chain starting from the zeroed arena, and a + b == b + a
// bit for bit: the result does not depend on which chain arrives first.
__device__ __forceinline__ void grad_add4(float* p, float4 v) {
  unsafeAtomicAdd(p + 0, v.x); unsafeAtomicAdd(p + 1, v.y); unsafeAtomicAdd(p + 2, v.z); unsafeAtomicAdd(p + 3, v.w);
}

// The norm kernels sit on the critical chain of the step and share the CUs with the weight-gradient kernels of the side
// stream: like the chain's convolutions (conv_x6.hip XAS_X6_PRIO) they win the issue arbitration.  r02 / early r03: no
// effect (the chain was MFMA-bound); with the f16x3 convolutions the norms are half of the chain: -1.0 ms per step
// (in-box A/B against no priority, 117.5 vs 118.5).
#ifndef XAS_BN_PRIO
#define XAS_BN_PRIO 3
#endif

struct ColArgs {
  const float* x; const float* y; const float* dy; const float* mean; const float* var; const float* aux;
  float eps; int act; long M; int C; ColGeom g;
  float* partial;              // [G][nslab][2][C]
  unsigned* ticket;            // [ncb]
  // finalize
  float* out1; float* out2;    // group g writes out1[g * out_stride + c], out2[g * out_stride + c]
  long out_stride;
  float* count_out;            // != null: count_out[g * out_stride] = rows per group (SyncBatchNorm message)
  float* running_mean; float* running_var; float momentum, unbias;
  float* acc1; float* acc2;    // != null: acc1[c] += sum over groups of out1 ... (parameter gradients, in place)
  const uint8_t* mask;         // MODE 1: != null -> activation sign bits (one byte per float4, bit e = z_e > 0) instead of y
  long rows_real;              // MODE 5: activation rows per group behind the partial rows
};

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

// write-through (sc1) store / L1-bypassing load of a float4 at a per-lane byte offset of ONE wave-uniform buffer: the
// slab partials cross workgroups inside one launch
__device__ __forceinline__ void store_wt(__amdgpu_buffer_rsrc_t r, unsigned byte_off, float4 v) {
  const u32x4_t u = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
  __builtin_amdgcn_raw_buffer_store_b128(u, r, (int)byte_off, 0, 16);          // aux 16 = sc1
}
__device__ __forceinline__ float4 load_wt(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  const u32x4_t u = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 16);
  return make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
}

// loads of the reduction passes: PLAIN since r05.  r03/r04 shipped non-temporal loads here like for every streamed operand
// (measured with the weight gradients sharing the chip); on ONE stream the apply kernel that follows reads
// the same x and dy again and finds part of them in the caches when the reduction did not mark them for early eviction:
// -0.7 ms/step (in-box, interleaved, 3 rounds: 124.8 -> 124.1).  XAS_BN_REDUCE_NT=1 brings the hint back.
#ifndef XAS_BN_REDUCE_NT
#define XAS_BN_REDUCE_NT 0
#endif
__device__ __forceinline__ float4 red_load(const float4* p) { return XAS_BN_REDUCE_NT ? stream_load(p) : *p; }

template <int MODE, int UNR = 4, int FL = 4>   // 0: stats of x around pivot ; 1: bn backward sums ; 2: plain column sums of x ;
                      // 3: bn backward sums WITHOUT x: xhat = (z - beta)/gamma with z recovered from y (act != 0)
                      // 4: bn backward sums WITHOUT y: the activation mask is re-derived from x (`y` carries gamma,
                      //    `aux` carries beta)
                      // 5: x = per-tile partial sums of a convolution epilogue, [rows][C/2 channels][sum(v-p), sum((v-p)^2)]
                      //    (C = 2 x channels, `aux` = pivot per channel or null): plain column sums, finalize -> statistics
                      // 6: x = per-tile partial sums of a data-gradient epilogue, [rows][2][C/2 channels] (sum dz | sum dz xhat):
                      //    plain column sums -> out1 = sums [G][2][channels]; acc1 / acc2 (dbeta / dgamma) get the two halves
__device__ __forceinline__ void col_reduce_body(const ColArgs& a) {
  __shared__ __align__(16) float4 red[2][256];         // 8 KB, also the double scratch of the finalize tail
  const ColGeom& g = a.g;
  const float* __restrict__ x = a.x; const float* __restrict__ y = a.y; const float* __restrict__ dy = a.dy;
  const int C = a.C, act = a.act;
  const float eps = a.eps;
  const int tx = threadIdx.x % g.TX, ty = threadIdx.x / g.TX;
  const int c = blockIdx.y * g.CB + tx * 4;
  const int grp = blockIdx.z;
  const float* mean = a.mean ? a.mean + (size_t)grp * C : nullptr;      // per-group statistics / parameters
  const float* var = a.var ? a.var + (size_t)grp * C : nullptr;
  const long r0 = grp * g.Mg + (long)blockIdx.x * g.rows_per_slab;
  const long r1 = min((grp + 1) * g.Mg, r0 + g.rows_per_slab);
  float4 s1 = make_float4(0, 0, 0, 0), s2 = make_float4(0, 0, 0, 0);
  float4 p0, p1;   // MODE 0: pivot ; MODE 1: mean, invstd
  float4 rsg = make_float4(0, 0, 0, 0), bt = make_float4(0, 0, 0, 0);      // MODE 4: rstd * gamma, beta
  if (MODE == 4) {
    p0 = *reinterpret_cast<const float4*>(mean + c);
    const float4 v = *reinterpret_cast<const float4*>(var + c);
    p1 = make_float4(rsqrtf(v.x + eps), rsqrtf(v.y + eps), rsqrtf(v.z + eps), rsqrtf(v.w + eps));
    const float4 gm = *reinterpret_cast<const float4*>(y + c);
    rsg = make_float4(__fmul_rn(p1.x, gm.x), __fmul_rn(p1.y, gm.y), __fmul_rn(p1.z, gm.z), __fmul_rn(p1.w, gm.w));
    bt = *reinterpret_cast<const float4*>(a.aux + c);
  } else if (MODE == 0) {
    p0 = *reinterpret_cast<const float4*>(x + (size_t)grp * g.Mg * C + c);          // pivot = first row of the group
    p1 = p0;
  } else if (MODE == 2 || MODE == 5 || MODE == 6) {
    p0 = make_float4(0, 0, 0, 0); p1 = p0;
  } else if (MODE == 3) {                                  // `aux` carries beta, `y`-side parameter pointer: see launcher
    p0 = *reinterpret_cast<const float4*>(a.aux + c);                                 // beta
    const float4 gm = *reinterpret_cast<const float4*>(a.x + c);                      // gamma (x unused in this mode)
    p1 = make_float4(gm.x != 0.f ? 1.f / gm.x : 0.f, gm.y != 0.f ? 1.f / gm.y : 0.f, gm.z != 0.f ? 1.f / gm.z : 0.f,
                     gm.w != 0.f ? 1.f / gm.w : 0.f);
  } else {
    p0 = *reinterpret_cast<const float4*>(mean + c);
    const float4 v = *reinterpret_cast<const float4*>(var + c);
    p1 = make_float4(rsqrtf(v.x + eps), rsqrtf(v.y + eps), rsqrtf(v.z + eps), rsqrtf(v.w + eps));
  }
#pragma unroll UNR
  for (long r = r0 + ty; r < r1; r += g.TY) {
    if (MODE == 3) {
      float4 g4 = red_load(reinterpret_cast<const float4*>(dy + r * C + c));
      const float4 yv = red_load(reinterpret_cast<const float4*>(y + r * C + c));
      const float neg = act == 1 ? 0.f : 0.01f, up = act == 1 ? 0.f : 100.f;
      float4 z;                                            // pre-activation value
      z.x = yv.x > 0.f ? yv.x : yv.x * up; z.y = yv.y > 0.f ? yv.y : yv.y * up;
      z.z = yv.z > 0.f ? yv.z : yv.z * up; z.w = yv.w > 0.f ? yv.w : yv.w * up;
      g4.x *= yv.x > 0.f ? 1.f : neg; g4.y *= yv.y > 0.f ? 1.f : neg;
      g4.z *= yv.z > 0.f ? 1.f : neg; g4.w *= yv.w > 0.f ? 1.f : neg;
      s1.x += g4.x; s1.y += g4.y; s1.z += g4.z; s1.w += g4.w;
      s2.x = fmaf(g4.x, (z.x - p0.x) * p1.x, s2.x); s2.y = fmaf(g4.y, (z.y - p0.y) * p1.y, s2.y);
      s2.z = fmaf(g4.z, (z.z - p0.z) * p1.z, s2.z); s2.w = fmaf(g4.w, (z.w - p0.w) * p1.w, s2.w);
      continue;
    }
    const float4 xv = red_load(reinterpret_cast<const float4*>(x + r * C + c));
    if (MODE == 4) {
      float4 g4 = red_load(reinterpret_cast<const float4*>(dy + r * C + c));
      const float neg = act == 1 ? 0.f : 0.01f;
      g4.x *= bn_affine(xv.x, p0.x, rsg.x, bt.x) > 0.f ? 1.f : neg; g4.y *= bn_affine(xv.y, p0.y, rsg.y, bt.y) > 0.f ? 1.f : neg;
      g4.z *= bn_affine(xv.z, p0.z, rsg.z, bt.z) > 0.f ? 1.f : neg; g4.w *= bn_affine(xv.w, p0.w, rsg.w, bt.w) > 0.f ? 1.f : neg;
      s1.x += g4.x; s1.y += g4.y; s1.z += g4.z; s1.w += g4.w;
      s2.x = fmaf(g4.x, (xv.x - p0.x) * p1.x, s2.x); s2.y = fmaf(g4.y, (xv.y - p0.y) * p1.y, s2.y);
      s2.z = fmaf(g4.z, (xv.z - p0.z) * p1.z, s2.z); s2.w = fmaf(g4.w, (xv.w - p0.w) * p1.w, s2.w);
      continue;
    }
    if (MODE == 2 || MODE == 5 || MODE == 6) {
      s1.x += xv.x; s1.y += xv.y; s1.z += xv.z; s1.w += xv.w;
    } else if (MODE == 0) {
      const float a0 = xv.x - p0.x, b = xv.y - p0.y, cc = xv.z - p0.z, d = xv.w - p0.w;
      s1.x += a0; s1.y += b; s1.z += cc; s1.w += d;
      s2.x = fmaf(a0, a0, s2.x); s2.y = fmaf(b, b, s2.y); s2.z = fmaf(cc, cc, s2.z); s2.w = fmaf(d, d, s2.w);
    } else {
      float4 g4 = red_load(reinterpret_cast<const float4*>(dy + r * C + c));
      if (act && a.mask) {
        const unsigned mb = a.mask[(r * C + c) >> 2];
        const float neg = act == 1 ? 0.f : 0.01f;
        g4.x *= (mb & 1u) ? 1.f : neg; g4.y *= (mb & 2u) ? 1.f : neg;
        g4.z *= (mb & 4u) ? 1.f : neg; g4.w *= (mb & 8u) ? 1.f : neg;
      } else if (act) {
        const float4 yv = red_load(reinterpret_cast<const float4*>(y + r * C + c));
        const float neg = act == 1 ? 0.f : 0.01f;
        g4.x *= yv.x > 0.f ? 1.f : neg; g4.y *= yv.y > 0.f ? 1.f : neg;
        g4.z *= yv.z > 0.f ? 1.f : neg; g4.w *= yv.w > 0.f ? 1.f : neg;
      }
      s1.x += g4.x; s1.y += g4.y; s1.z += g4.z; s1.w += g4.w;
      s2.x = fmaf(g4.x, (xv.x - p0.x) * p1.x, s2.x); s2.y = fmaf(g4.y, (xv.y - p0.y) * p1.y, s2.y);
      s2.z = fmaf(g4.z, (xv.z - p0.z) * p1.z, s2.z); s2.w = fmaf(g4.w, (xv.w - p0.w) * p1.w, s2.w);
    }
  }
  red[0][threadIdx.x] = s1; red[1][threadIdx.x] = s2;
  __syncthreads();
  for (int s = g.TY >> 1; s > 0; s >>= 1) {
    if (ty < s) {
      const float4 u0 = red[0][threadIdx.x + s * g.TX], v0 = red[1][threadIdx.x + s * g.TX];
      float4& u = red[0][threadIdx.x]; float4& v = red[1][threadIdx.x];
      u.x += u0.x; u.y += u0.y; u.z += u0.z; u.w += u0.w;
      v.x += v0.x; v.y += v0.y; v.z += v0.z; v.w += v0.w;
    }
    __syncthreads();
  }
  // ---- publish this block's partial sums (write-through), then take a ticket -------------------------------------
  const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(
      a.partial, 0, (int)((size_t)g.G * g.nslab * 2 * C * sizeof(float)), 0x00020000);
  if (ty == 0) {
    const unsigned o = (unsigned)(((((size_t)grp * g.nslab + blockIdx.x) * 2) * C + c) * sizeof(float));
    store_wt(prs, o, red[0][tx]);
    store_wt(prs, o + (unsigned)C * 4u, red[1][tx]);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave drains its stores ...
  __syncthreads();                                          // ... before one lane signals for the workgroup
  __shared__ int s_last;
  if (threadIdx.x == 0) {
    const unsigned total = (unsigned)(g.nslab * g.G);
    const unsigned old = __hip_atomic_fetch_add(a.ticket + blockIdx.y, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = old == total - 1u;
    if (last) {
      __hip_atomic_store(a.ticket + blockIdx.y, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for re-use
#ifndef XAS_BN_NO_ACQUIRE
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");    // drop this CU's stale L1 lines (other blocks' partials)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    }
    s_last = last;
  }
  __syncthreads();
  if (!s_last) return;

  // ---- finalize (last arriver of this channel block): all groups, slabs summed in slab order in double ------------
  double* dred = reinterpret_cast<double*>(&red[0][0]);     // [256][4] doubles = 8 KB
  const int nl = g.TY;                                      // slab lanes (same thread layout as the reduction)
  float4 accg1 = make_float4(0, 0, 0, 0), accg2 = make_float4(0, 0, 0, 0);      // sums over groups (parameter gradients)
  float4 rm = make_float4(0, 0, 0, 0), rv = make_float4(0, 0, 0, 0);
  if (MODE == 0 && a.running_mean && ty == 0) {
    rm = *reinterpret_cast<const float4*>(a.running_mean + c);
    rv = *reinterpret_cast<const float4*>(a.running_var + c);
  }
  const int ch = c >> 1;                                    // MODE 5: this thread's two channels are ch, ch + 1
  if (MODE == 5 && a.running_mean && ty == 0) {
    const float2 m2 = *reinterpret_cast<const float2*>(a.running_mean + ch);
    const float2 v2 = *reinterpret_cast<const float2*>(a.running_var + ch);
    rm.x = m2.x; rm.y = m2.y; rv.x = v2.x; rv.y = v2.y;
  }
  for (int gi = 0; gi < g.G; ++gi) {
    double d1[4] = {0, 0, 0, 0}, d2[4] = {0, 0, 0, 0};
    const unsigned base = (unsigned)((((size_t)gi * g.nslab * 2) * C + c) * sizeof(float));
    const unsigned slab_b = 2u * (unsigned)C * 4u, half_b = (unsigned)C * 4u;
    int s = ty;
    for (; s + (FL - 1) * nl < g.nslab; s += FL * nl) {      // 2 * FL independent 16-byte loads in flight per lane
      float4 u[FL], v[FL];
#pragma unroll
      for (int k = 0; k < FL; ++k) {
        u[k] = load_wt(prs, base + (unsigned)(s + k * nl) * slab_b);
        v[k] = load_wt(prs, base + (unsigned)(s + k * nl) * slab_b + half_b);
      }
#pragma unroll
      for (int k = 0; k < FL; ++k) {
        d1[0] += (double)u[k].x; d1[1] += (double)u[k].y; d1[2] += (double)u[k].z; d1[3] += (double)u[k].w;
        d2[0] += (double)v[k].x; d2[1] += (double)v[k].y; d2[2] += (double)v[k].z; d2[3] += (double)v[k].w;
      }
    }
    for (; s < g.nslab; s += nl) {
      const float4 u = load_wt(prs, base + (unsigned)s * slab_b), v = load_wt(prs, base + (unsigned)s * slab_b + half_b);
      d1[0] += (double)u.x; d1[1] += (double)u.y; d1[2] += (double)u.z; d1[3] += (double)u.w;
      d2[0] += (double)v.x; d2[1] += (double)v.y; d2[2] += (double)v.z; d2[3] += (double)v.w;
    }
    // cross-lane sums over the slab lanes, fixed order, first sums then second sums through the same 8 KB
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      __syncthreads();
#pragma unroll
      for (int e = 0; e < 4; ++e) dred[(size_t)threadIdx.x * 4 + e] = half ? d2[e] : d1[e];
      __syncthreads();
      if (ty == 0) {
        for (int k = 1; k < nl; ++k)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const double t = dred[(size_t)(k * g.TX + tx) * 4 + e];
            if (half) d2[e] += t; else d1[e] += t;
          }
      }
    }
    if (ty != 0) continue;
    if (MODE == 5) {
      const float2 pv = a.aux ? *reinterpret_cast<const float2*>(a.aux + ch) : make_float2(0.f, 0.f);
      const double inv = 1.0 / (double)a.rows_real;
      const double m0 = d1[0] * inv, m1 = d1[2] * inv;
      double v0 = d1[1] * inv - m0 * m0, v1 = d1[3] * inv - m1 * m1;
      if (v0 < 0.0) v0 = 0.0;
      if (v1 < 0.0) v1 = 0.0;
      const float mf0 = (float)((double)pv.x + m0), mf1 = (float)((double)pv.y + m1), vf0 = (float)v0, vf1 = (float)v1;
      *reinterpret_cast<float2*>(a.out1 + (size_t)gi * a.out_stride + ch) = make_float2(mf0, mf1);
      *reinterpret_cast<float2*>(a.out2 + (size_t)gi * a.out_stride + ch) = make_float2(vf0, vf1);
      if (a.running_mean) {
        const float mo = a.momentum, ub = a.unbias;
        rm.x = (1.f - mo) * rm.x + mo * mf0; rm.y = (1.f - mo) * rm.y + mo * mf1;
        rv.x = (1.f - mo) * rv.x + mo * (vf0 * ub); rv.y = (1.f - mo) * rv.y + mo * (vf1 * ub);
      }
      if (a.count_out && blockIdx.y == 0 && tx == 0) a.count_out[(size_t)gi * a.out_stride] = (float)a.rows_real;
      continue;
    }
    float* o1 = a.out1 + (size_t)gi * a.out_stride + c;
    float* o2 = a.out2 + (size_t)gi * a.out_stride + c;
    if (MODE == 0) {
      const float4 pv = *reinterpret_cast<const float4*>(x + (size_t)gi * g.Mg * C + c);
      const float piv[4] = {pv.x, pv.y, pv.z, pv.w};
      float mf[4], vf[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const double m = d1[e] / (double)g.Mg;
        double v = d2[e] / (double)g.Mg - m * m;
        if (v < 0.0) v = 0.0;
        mf[e] = (float)((double)piv[e] + m); vf[e] = (float)v;
      }
      *reinterpret_cast<float4*>(o1) = make_float4(mf[0], mf[1], mf[2], mf[3]);
      *reinterpret_cast<float4*>(o2) = make_float4(vf[0], vf[1], vf[2], vf[3]);
      if (a.running_mean) {                                  // one update per group, in group order = the order of the
        const float mo = a.momentum, ub = a.unbias;          // reference's per-camera calls
        rm.x = (1.f - mo) * rm.x + mo * mf[0]; rm.y = (1.f - mo) * rm.y + mo * mf[1];
        rm.z = (1.f - mo) * rm.z + mo * mf[2]; rm.w = (1.f - mo) * rm.w + mo * mf[3];
        rv.x = (1.f - mo) * rv.x + mo * (vf[0] * ub); rv.y = (1.f - mo) * rv.y + mo * (vf[1] * ub);
        rv.z = (1.f - mo) * rv.z + mo * (vf[2] * ub); rv.w = (1.f - mo) * rv.w + mo * (vf[3] * ub);
      }
    } else {
      const float4 f1 = make_float4((float)d1[0], (float)d1[1], (float)d1[2], (float)d1[3]);
      const float4 f2 = make_float4((float)d2[0], (float)d2[1], (float)d2[2], (float)d2[3]);
      *reinterpret_cast<float4*>(o1) = f1;
      if (MODE != 6) *reinterpret_cast<float4*>(o2) = f2;
      accg1.x += f1.x; accg1.y += f1.y; accg1.z += f1.z; accg1.w += f1.w;
      accg2.x += f2.x; accg2.y += f2.y; accg2.z += f2.z; accg2.w += f2.w;
    }
    if (a.count_out && blockIdx.y == 0 && tx == 0) a.count_out[(size_t)gi * a.out_stride] = (float)g.Mg;
  }
  if (ty != 0) return;
  if (MODE == 0 && a.running_mean) {
    *reinterpret_cast<float4*>(a.running_mean + c) = rm;
    *reinterpret_cast<float4*>(a.running_var + c) = rv;
  }
  if (MODE == 5) {
    if (a.running_mean) {
      *reinterpret_cast<float2*>(a.running_mean + ch) = make_float2(rm.x, rm.y);
      *reinterpret_cast<float2*>(a.running_var + ch) = make_float2(rv.x, rv.y);
    }
    return;
  }
  if (MODE == 6) {                                           // columns [0, C/2): dbeta, [C/2, C): dgamma
    if (a.acc1) {
      const int half = C >> 1;
      float* ap = c < half ? a.acc1 + c : a.acc2 + (c - half);
      grad_add4(ap, accg1);
    }
    return;
  }
  if (MODE != 0 && a.acc1) {                                 // parameter gradients accumulated in place (.grad arena)
    grad_add4(a.acc1 + c, accg1);
    grad_add4(a.acc2 + c, accg2);
  }
}

template <int MODE>
__global__ __launch_bounds__(256) void col_reduce_kernel(ColArgs a) {
#ifdef XAS_BN_PRIO
  __builtin_amdgcn_s_setprio(XAS_BN_PRIO);
#endif
  col_reduce_body<MODE, 4, 4>(a);
}

// Same kernel compiled for at most 64 VGPRs: shipped for the backward sums in r04, when they ran beside two weight-gradient
// blocks per CU (which leave 112 VGPRs per SIMD lane - room for two lean waves instead of one).  On ONE stream (r05) the
// 86-register build above is 0.7 ms/step faster (in-box, interleaved, 5 rounds: 126.95 -> 126.24) and is the shipped one;
// tune bit 18 selects this build.
template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8)))
void col_reduce_lean_kernel(ColArgs a) {
#ifdef XAS_BN_PRIO
  __builtin_amdgcn_s_setprio(XAS_BN_PRIO);
#endif
  col_reduce_body<MODE, 2, 2>(a);
}

// SyncBatchNorm: merge the per-rank statistics of every group.  gathered: [world][G][msg_stride] = mean | biased var |
// count (| padding) per (rank, group).  Count-weighted merge in double (Chan et al.: total M2 = sum n_i (var_i + (mean_i - mean)^2)),
// then the running-statistic update of each group in group order with the GLOBAL count.
__global__ void bn_sync_merge_kernel(const float* __restrict__ gathered, int world, int G, int C, long msg_stride,
                                     float* __restrict__ mean, float* __restrict__ var,
                                     float* __restrict__ running_mean, float* __restrict__ running_var, float momentum) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const size_t stride = (size_t)msg_stride;
  float rm = running_mean ? running_mean[c] : 0.f, rv = running_var ? running_var[c] : 0.f;
  for (int g = 0; g < G; ++g) {
    double n = 0.0, mu = 0.0;
    for (int r = 0; r < world; ++r) {
      const float* p = gathered + ((size_t)r * G + g) * stride;
      const double cnt = (double)p[2 * C];
      n += cnt; mu += cnt * (double)p[c];
    }
    mu /= n;
    double m2 = 0.0;
    for (int r = 0; r < world; ++r) {
      const float* p = gathered + ((size_t)r * G + g) * stride;
      const double d = (double)p[c] - mu;
      m2 += (double)p[2 * C] * ((double)p[C + c] + d * d);
    }
    const float mf = (float)mu, vf = (float)(m2 / n);
    mean[(size_t)g * C + c] = mf;
    var[(size_t)g * C + c] = vf;
    if (running_mean) {
      const float unbias = n > 1.0 ? (float)(n / (n - 1.0)) : 1.f;
      rm = (1.f - momentum) * rm + momentum * mf;
      rv = (1.f - momentum) * rv + momentum * (vf * unbias);
    }
  }
  if (running_mean) { running_mean[c] = rm; running_var[c] = rv; }
}

__device__ __forceinline__ float act_fwd(float v, int act) {
  return act == 0 ? v : (act == 1 ? fmaxf(v, 0.f) : (v > 0.f ? v : 0.01f * v));
}

// grid.y = group: the rows of group g use mean / var row g ([G][C]); n4g = float4 elements per group
__global__ void bn_apply_kernel(const float4* __restrict__ x, const float* __restrict__ mean,
                                const float* __restrict__ var, const float* __restrict__ gamma,
                                const float* __restrict__ beta, const float4* __restrict__ res, float eps, int act,
                                long n4g, int C4, float4* __restrict__ y, uint8_t* __restrict__ mask_out,
                                float* __restrict__ amax_out) {
  float amx = 0.f;
  const long goff = (long)blockIdx.y * n4g;
  mean += (size_t)blockIdx.y * C4 * 4; var += (size_t)blockIdx.y * C4 * 4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4g; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    const float4 m = *reinterpret_cast<const float4*>(mean + c), v = *reinterpret_cast<const float4*>(var + c);
    const float4 g = *reinterpret_cast<const float4*>(gamma + c), b = *reinterpret_cast<const float4*>(beta + c);
    const float4 xv = x[goff + i];
    float4 o;
    o.x = bn_affine(xv.x, m.x, __fmul_rn(rsqrtf(v.x + eps), g.x), b.x);
    o.y = bn_affine(xv.y, m.y, __fmul_rn(rsqrtf(v.y + eps), g.y), b.y);
    o.z = bn_affine(xv.z, m.z, __fmul_rn(rsqrtf(v.z + eps), g.z), b.z);
    o.w = bn_affine(xv.w, m.w, __fmul_rn(rsqrtf(v.w + eps), g.w), b.w);
    if (res) { const float4 r = res[goff + i]; o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w; }
    if (mask_out)          // sign bits of the pre-activation value: what the backward needs of y (1/16 of its bytes)
      mask_out[goff + i] = (uint8_t)((o.x > 0.f ? 1u : 0u) | (o.y > 0.f ? 2u : 0u) | (o.z > 0.f ? 4u : 0u) | (o.w > 0.f ? 8u : 0u));
    o.x = act_fwd(o.x, act); o.y = act_fwd(o.y, act); o.z = act_fwd(o.z, act); o.w = act_fwd(o.w, act);
    amx = amax4(amx, o);
    y[goff + i] = o;
  }
  if (amax_out) publish_amax(amax_out, amx);
}

// grid.y = group; sums: [G][2][C] = sum_dz | sum_dz_xhat of each group
__global__ void bn_bwd_apply_kernel(const float4* __restrict__ x, const float4* __restrict__ y,
                                    const float4* __restrict__ dy, const float* __restrict__ mean,
                                    const float* __restrict__ var, const float* __restrict__ gamma,
                                    const float* __restrict__ beta, const float* __restrict__ sums,
                                    float eps, int act, long n4g, int C4,
                                    float inv_count, float4* __restrict__ dx, float4* __restrict__ dres,
                                    const uint8_t* __restrict__ mask, float* __restrict__ amax_out) {
  float amx = 0.f;
  const long goff = (long)blockIdx.y * n4g;
  mean += (size_t)blockIdx.y * C4 * 4; var += (size_t)blockIdx.y * C4 * 4;
  const float* sdz = sums + (size_t)blockIdx.y * C4 * 8;
  const float* sdzx = sdz + (size_t)C4 * 4;
  for (long ii = (long)blockIdx.x * blockDim.x + threadIdx.x; ii < n4g; ii += (long)gridDim.x * blockDim.x) {
    const int c = (int)(ii % C4) * 4;
    const long i = goff + ii;
    const float4 m = *reinterpret_cast<const float4*>(mean + c), v = *reinterpret_cast<const float4*>(var + c);
    const float4 g = *reinterpret_cast<const float4*>(gamma + c);
    const float4 a = *reinterpret_cast<const float4*>(sdz + c), b = *reinterpret_cast<const float4*>(sdzx + c);
    float4 dz = dy[i];
    float4 yv = make_float4(0, 0, 0, 0);
    if (act && mask) {                                     // sign bits saved by the forward (layers with a residual)
      const unsigned mb = mask[i];
      const float neg = act == 1 ? 0.f : 0.01f;
      dz.x *= (mb & 1u) ? 1.f : neg; dz.y *= (mb & 2u) ? 1.f : neg;
      dz.z *= (mb & 4u) ? 1.f : neg; dz.w *= (mb & 8u) ? 1.f : neg;
    } else if (act && y) {
      yv = y[i];
      const float neg = act == 1 ? 0.f : 0.01f;
      dz.x *= yv.x > 0.f ? 1.f : neg; dz.y *= yv.y > 0.f ? 1.f : neg;
      dz.z *= yv.z > 0.f ? 1.f : neg; dz.w *= yv.w > 0.f ? 1.f : neg;
    } else if (act) {                                      // y not read: the mask is re-derived from x (no residual)
      const float4 xv = x[i];
      const float4 bt = *reinterpret_cast<const float4*>(beta + c);
      const float neg = act == 1 ? 0.f : 0.01f;
      dz.x *= bn_affine(xv.x, m.x, __fmul_rn(rsqrtf(v.x + eps), g.x), bt.x) > 0.f ? 1.f : neg;
      dz.y *= bn_affine(xv.y, m.y, __fmul_rn(rsqrtf(v.y + eps), g.y), bt.y) > 0.f ? 1.f : neg;
      dz.z *= bn_affine(xv.z, m.z, __fmul_rn(rsqrtf(v.z + eps), g.z), bt.z) > 0.f ? 1.f : neg;
      dz.w *= bn_affine(xv.w, m.w, __fmul_rn(rsqrtf(v.w + eps), g.w), bt.w) > 0.f ? 1.f : neg;
    }
    if (dres) dres[i] = dz;
    float4 xh;                                             // normalised input
    float4 is = make_float4(rsqrtf(v.x + eps), rsqrtf(v.y + eps), rsqrtf(v.z + eps), rsqrtf(v.w + eps));
    if (x) {
      const float4 xv = x[i];
      xh = make_float4((xv.x - m.x) * is.x, (xv.y - m.y) * is.y, (xv.z - m.z) * is.z, (xv.w - m.w) * is.w);
    } else {                                               // recovered from y: z = act^-1(y), xhat = (z - beta) / gamma
      const float4 bt = *reinterpret_cast<const float4*>(beta + c);
      const float up = act == 1 ? 0.f : 100.f;
      xh.x = g.x != 0.f ? ((yv.x > 0.f ? yv.x : yv.x * up) - bt.x) / g.x : 0.f;
      xh.y = g.y != 0.f ? ((yv.y > 0.f ? yv.y : yv.y * up) - bt.y) / g.y : 0.f;
      xh.z = g.z != 0.f ? ((yv.z > 0.f ? yv.z : yv.z * up) - bt.z) / g.z : 0.f;
      xh.w = g.w != 0.f ? ((yv.w > 0.f ? yv.w : yv.w * up) - bt.w) / g.w : 0.f;
    }
    float4 o;
    o.x = g.x * is.x * (dz.x - a.x * inv_count - xh.x * b.x * inv_count);
    o.y = g.y * is.y * (dz.y - a.y * inv_count - xh.y * b.y * inv_count);
    o.z = g.z * is.z * (dz.z - a.z * inv_count - xh.z * b.z * inv_count);
    o.w = g.w * is.w * (dz.w - a.w * inv_count - xh.w * b.w * inv_count);
    amx = fmaxf(fmaxf(amx, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
    dx[i] = o;
  }
  if (amax_out) publish_amax(amax_out, amx);
}

// ---- streaming forms of the two apply kernels.  When the number of threads in a grid row is a multiple of C/4 a thread
// sees ONE channel quadruple for its whole life: the per-channel parameters are read once, the loop body is only the
// activation stream, compiled per mode (no branches) and unrolled so that several 16-byte loads are in flight per lane -
// in the backward pass these kernels share the chip with two weight-gradient blocks per CU and get few wave slots.
template <int ACT, bool RES, bool MASKOUT>
__global__ __launch_bounds__(256) void bn_apply_stream_kernel(const float4* __restrict__ x, const float* __restrict__ mean,
                                                             const float* __restrict__ var, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const float4* __restrict__ res,
                                                             float eps, long n4g, int C4, float4* __restrict__ y,
                                                             uint8_t* __restrict__ mask_out, float* __restrict__ amax_out) {
  float amx = 0.f;                                     // max |y| of this thread (amax_out != null: xas_bn_apply_amax)
#ifdef XAS_BN_PRIO
  __builtin_amdgcn_s_setprio(XAS_BN_PRIO);
#endif
  const long goff = (long)blockIdx.y * n4g;
  const long stride = (long)gridDim.x * blockDim.x;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int c = (int)(i % C4) * 4;
  const float4 m = *reinterpret_cast<const float4*>(mean + (size_t)blockIdx.y * C4 * 4 + c);
  const float4 v = *reinterpret_cast<const float4*>(var + (size_t)blockIdx.y * C4 * 4 + c);
  const float4 g = *reinterpret_cast<const float4*>(gamma + c), b = *reinterpret_cast<const float4*>(beta + c);
  const float4 rs = make_float4(__fmul_rn(rsqrtf(v.x + eps), g.x), __fmul_rn(rsqrtf(v.y + eps), g.y),
                                __fmul_rn(rsqrtf(v.z + eps), g.z), __fmul_rn(rsqrtf(v.w + eps), g.w));
  auto one = [&](long k, float4 xv, float4 rv) {
    float4 o;
    o.x = bn_affine(xv.x, m.x, rs.x, b.x); o.y = bn_affine(xv.y, m.y, rs.y, b.y);
    o.z = bn_affine(xv.z, m.z, rs.z, b.z); o.w = bn_affine(xv.w, m.w, rs.w, b.w);
    if (RES) { o.x += rv.x; o.y += rv.y; o.z += rv.z; o.w += rv.w; }
    if (MASKOUT)
      mask_out[goff + k] = (uint8_t)((o.x > 0.f ? 1u : 0u) | (o.y > 0.f ? 2u : 0u) | (o.z > 0.f ? 4u : 0u) | (o.w > 0.f ? 8u : 0u));
    o.x = act_fwd(o.x, ACT); o.y = act_fwd(o.y, ACT); o.z = act_fwd(o.z, ACT); o.w = act_fwd(o.w, ACT);
    amx = amax4(amx, o);
#ifdef XAS_BN_APPLY_PLAIN_STORE
    y[goff + k] = o;
#else
    stream_store(y + goff + k, o);
#endif
  };
  const float4 z4 = make_float4(0, 0, 0, 0);
  for (; i + 3 * stride < n4g; i += 4 * stride) {
    float4 xv[4], rv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { xv[u] = stream_load(x + goff + i + u * stride); rv[u] = RES ? stream_load(res + goff + i + u * stride) : z4; }
#pragma unroll
    for (int u = 0; u < 4; ++u) one(i + u * stride, xv[u], rv[u]);
  }
  for (; i < n4g; i += stride) one(i, x[goff + i], RES ? res[goff + i] : z4);
  if (amax_out) publish_amax(amax_out, amx);
}

// SIGN: where the activation sign comes from: 0 no activation, 1 y, 2 x (re-derived, no residual), 3 mask bytes
// XH  : normalised input from x (true) or recovered from y (false: leaky ReLU, SIGN == 1)
template <int ACT, int SIGN, bool XH, bool DRES>
__global__ __launch_bounds__(256) void bn_bwd_apply_stream_kernel(
    const float4* __restrict__ x, const float4* __restrict__ y, const float4* __restrict__ dy, const float* __restrict__ mean,
    const float* __restrict__ var, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ sums, float eps, long n4g, int C4, float inv_count, float4* __restrict__ dx,
    float4* __restrict__ dres, const uint8_t* __restrict__ mask, float* __restrict__ amax_out) {
#ifdef XAS_BN_PRIO
  __builtin_amdgcn_s_setprio(XAS_BN_PRIO);
#endif
  float amx = 0.f;                                     // max |dx| of this thread (amax_out != null: xas_bn_bwd_apply_amax)
  const long goff = (long)blockIdx.y * n4g;
  const long stride = (long)gridDim.x * blockDim.x;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int c = (int)(i % C4) * 4;
  const size_t gc = (size_t)blockIdx.y * C4 * 4 + c;
  const float4 m = *reinterpret_cast<const float4*>(mean + gc), v = *reinterpret_cast<const float4*>(var + gc);
  const float4 g = *reinterpret_cast<const float4*>(gamma + c);
  const float4 bt = beta ? *reinterpret_cast<const float4*>(beta + c) : make_float4(0, 0, 0, 0);
  const float4 a = *reinterpret_cast<const float4*>(sums + (size_t)blockIdx.y * C4 * 8 + c);
  const float4 b = *reinterpret_cast<const float4*>(sums + (size_t)blockIdx.y * C4 * 8 + (size_t)C4 * 4 + c);
  const float4 is = make_float4(rsqrtf(v.x + eps), rsqrtf(v.y + eps), rsqrtf(v.z + eps), rsqrtf(v.w + eps));
  const float4 rs = make_float4(__fmul_rn(is.x, g.x), __fmul_rn(is.y, g.y), __fmul_rn(is.z, g.z), __fmul_rn(is.w, g.w));
  constexpr float neg = ACT == 1 ? 0.f : 0.01f, up = ACT == 1 ? 0.f : 100.f;
  auto one = [&](long k, float4 xv, float4 yv, float4 dz, unsigned mb) {
    if (SIGN == 1) {
      dz.x *= yv.x > 0.f ? 1.f : neg; dz.y *= yv.y > 0.f ? 1.f : neg; dz.z *= yv.z > 0.f ? 1.f : neg; dz.w *= yv.w > 0.f ? 1.f : neg;
    } else if (SIGN == 2) {
      dz.x *= bn_affine(xv.x, m.x, rs.x, bt.x) > 0.f ? 1.f : neg; dz.y *= bn_affine(xv.y, m.y, rs.y, bt.y) > 0.f ? 1.f : neg;
      dz.z *= bn_affine(xv.z, m.z, rs.z, bt.z) > 0.f ? 1.f : neg; dz.w *= bn_affine(xv.w, m.w, rs.w, bt.w) > 0.f ? 1.f : neg;
    } else if (SIGN == 3) {
      dz.x *= (mb & 1u) ? 1.f : neg; dz.y *= (mb & 2u) ? 1.f : neg; dz.z *= (mb & 4u) ? 1.f : neg; dz.w *= (mb & 8u) ? 1.f : neg;
    }
    if (DRES) stream_store(dres + goff + k, dz);
    float4 xh;
    if (XH) xh = make_float4((xv.x - m.x) * is.x, (xv.y - m.y) * is.y, (xv.z - m.z) * is.z, (xv.w - m.w) * is.w);
    else {
      xh.x = g.x != 0.f ? ((yv.x > 0.f ? yv.x : yv.x * up) - bt.x) / g.x : 0.f;
      xh.y = g.y != 0.f ? ((yv.y > 0.f ? yv.y : yv.y * up) - bt.y) / g.y : 0.f;
      xh.z = g.z != 0.f ? ((yv.z > 0.f ? yv.z : yv.z * up) - bt.z) / g.z : 0.f;
      xh.w = g.w != 0.f ? ((yv.w > 0.f ? yv.w : yv.w * up) - bt.w) / g.w : 0.f;
    }
    float4 o;
    o.x = g.x * is.x * (dz.x - a.x * inv_count - xh.x * b.x * inv_count);
    o.y = g.y * is.y * (dz.y - a.y * inv_count - xh.y * b.y * inv_count);
    o.z = g.z * is.z * (dz.z - a.z * inv_count - xh.z * b.z * inv_count);
    o.w = g.w * is.w * (dz.w - a.w * inv_count - xh.w * b.w * inv_count);
    amx = fmaxf(fmaxf(amx, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
#ifdef XAS_BN_BWD_PLAIN_STORE
    dx[goff + k] = o;
#else
    stream_store(dx + goff + k, o);
#endif
  };
  const float4 z4 = make_float4(0, 0, 0, 0);
  constexpr bool NEEDX = XH || SIGN == 2, NEEDY = SIGN == 1 || !XH;
  for (; i + 3 * stride < n4g; i += 4 * stride) {
    float4 xv[4], yv[4], dv[4];
    unsigned mb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long k = goff + i + u * stride;
#ifdef XAS_BN_BWD_PLAIN_LOAD
      dv[u] = dy[k];
      xv[u] = NEEDX ? x[k] : z4;
#else
      dv[u] = stream_load(dy + k);
      xv[u] = NEEDX ? stream_load(x + k) : z4;
#endif
      yv[u] = NEEDY ? stream_load(y + k) : z4;
      mb[u] = SIGN == 3 ? mask[k] : 0u;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) one(i + u * stride, xv[u], yv[u], dv[u], mb[u]);
  }
  for (; i < n4g; i += stride) {
    const long k = goff + i;
    one(i, NEEDX ? x[k] : z4, NEEDY ? y[k] : z4, dy[k], SIGN == 3 ? mask[k] : 0u);
  }
  if (amax_out) publish_amax(amax_out, amx);
}

// one update per group, in group order (mean / var: [G][C])
__global__ void bn_running_kernel(const float* mean, const float* var, float* rm, float* rv, float momentum,
                                  float unbias, int C, int G) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float m = rm[c], v = rv[c];
  for (int g = 0; g < G; ++g) {
    m = (1.f - momentum) * m + momentum * mean[(size_t)g * C + c];
    v = (1.f - momentum) * v + momentum * (var[(size_t)g * C + c] * unbias);
  }
  rm[c] = m; rv[c] = v;
}

// ---------------------------------------------------------------- max pool 3x3 s2 p1
// One block walks whole image ROWS (grid-stride over n * rows): the row decode is scalar and 32-bit, a lane splits only its
// element index into (column, channel quad).  (r04: the flat 64-bit index per thread cost three emulated 64-bit divisions -
// backward 697 us for 1.4 GB = 2.0 TB/s at 256 images.)
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x, int N, int H, int W, int C, int Ho, int Wo,
                                                          float* __restrict__ y, int8_t* __restrict__ idx) {
  const int C4 = C / 4, per_row = Wo * C4;
  for (int row = blockIdx.x; row < N * Ho; row += gridDim.x) {
    const int n = row / Ho, ho = row - n * Ho;
    const float* xn = x + (size_t)n * H * W * C;
    const size_t obase = (size_t)row * per_row;
    for (int e = threadIdx.x; e < per_row; e += blockDim.x) {
      const int wo = e / C4, c = (e - wo * C4) * 4;
      float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
      int b0 = 0, b1 = 0, b2 = 0, b3 = 0;
      bool first = true;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const int hi = ho * 2 - 1 + r;
        if ((unsigned)hi >= (unsigned)H) continue;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          const int wi = wo * 2 - 1 + s;
          if ((unsigned)wi >= (unsigned)W) continue;
          const float4 v = *reinterpret_cast<const float4*>(xn + ((size_t)hi * W + wi) * C + c);
          const int tap = r * 3 + s;
          if (first || v.x > best.x) { best.x = v.x; b0 = tap; }
          if (first || v.y > best.y) { best.y = v.y; b1 = tap; }
          if (first || v.z > best.z) { best.z = v.z; b2 = tap; }
          if (first || v.w > best.w) { best.w = v.w; b3 = tap; }
          first = false;
        }
      }
      *reinterpret_cast<float4*>(y + (obase + e) * 4) = best;
      *reinterpret_cast<char4*>(idx + (obase + e) * 4) = make_char4((char)b0, (char)b1, (char)b2, (char)b3);
    }
  }
}

__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dy, const int8_t* __restrict__ idx, int N, int H,
                                                          int W, int C, int Ho, int Wo, float* __restrict__ dx) {
  const int C4 = C / 4, per_row = W * C4;
  for (int row = blockIdx.x; row < N * H; row += gridDim.x) {
    const int n = row / H, hi = row - n * H;
    const size_t obase = (size_t)row * per_row;
    for (int e = threadIdx.x; e < per_row; e += blockDim.x) {
      const int wi = e / C4, c = (e - wi * C4) * 4;
      float4 acc = make_float4(0, 0, 0, 0);
      // output windows ho with ho*2-1 <= hi <= ho*2+1
      for (int ho = hi / 2; ho <= (hi + 1) / 2; ++ho) {
        if (ho >= Ho) continue;
        const int r = hi - (ho * 2 - 1);
        if (r < 0 || r > 2) continue;
        for (int wo = wi / 2; wo <= (wi + 1) / 2; ++wo) {
          if (wo >= Wo) continue;
          const int s = wi - (wo * 2 - 1);
          if (s < 0 || s > 2) continue;
          const size_t o = (((size_t)n * Ho + ho) * Wo + wo) * C + c;
          const char4 k = *reinterpret_cast<const char4*>(idx + o);
          const float4 g = *reinterpret_cast<const float4*>(dy + o);
          const int tap = r * 3 + s;
          if (k.x == tap) acc.x += g.x;
          if (k.y == tap) acc.y += g.y;
          if (k.z == tap) acc.z += g.z;
          if (k.w == tap) acc.w += g.w;
        }
      }
      stream_store(reinterpret_cast<float4*>(dx) + obase + e, acc);
    }
  }
}

// ---------------------------------------------------------------- bilinear x2 (align_corners=False)
// Workgroups are dealt round-robin over the 8 XCDs (private L2 each).  Neighbouring image rows share their source rows:
// give every XCD a CONTIGUOUS range of the logical block ids, so the shared rows are found in that XCD's L2.
// Launch with 8 * ceil(nblk / 8) blocks; ids >= nblk are padding.
__device__ __forceinline__ unsigned xcd_contiguous(unsigned b, unsigned nblk) {
  const unsigned per = (nblk + 7u) >> 3;
  return (b & 7u) * per + (b >> 3);
}

// One block = 256 consecutive float4 of ONE output row: the row decode is scalar (block-uniform), a lane only splits its
// element index into (column, channel quad) with 32-bit arithmetic.  (The first version decoded a flat 64-bit index per
// thread - three emulated 64-bit divisions - and ran at 1.2 TB/s.)
__global__ __launch_bounds__(256) void upsample2x_fwd_kernel(const float* __restrict__ x, int N, int H, int W, int C,
                                                             float* __restrict__ y, unsigned segs, unsigned nblk) {
  const unsigned C4 = (unsigned)C / 4, Ho = 2u * H, Wo = 2u * W;
  const unsigned bid = xcd_contiguous(blockIdx.x, nblk);
  if (bid >= nblk) return;
  const unsigned row = bid / segs, seg = bid - row * segs;                      // row = n * Ho + ho
  const unsigned n = row / Ho, ho = row - n * Ho;
  const unsigned e = seg * 256u + threadIdx.x;
  if (e >= Wo * C4) return;
  const unsigned wo = e / C4, c = (e - wo * C4) * 4u;
  const float sh = fmaxf(0.f, (ho + 0.5f) * 0.5f - 0.5f), sw = fmaxf(0.f, (wo + 0.5f) * 0.5f - 0.5f);
  const int h0 = (int)sh, w0 = (int)sw;
  const int h1 = min(h0 + 1, H - 1), w1 = min(w0 + 1, W - 1);
  const float lh = sh - h0, lw = sw - w0;
  const float* base = x + (size_t)n * H * W * C + c;
  const float4 a = *reinterpret_cast<const float4*>(base + ((size_t)h0 * W + w0) * C);
  const float4 b = *reinterpret_cast<const float4*>(base + ((size_t)h0 * W + w1) * C);
  const float4 cc = *reinterpret_cast<const float4*>(base + ((size_t)h1 * W + w0) * C);
  const float4 d = *reinterpret_cast<const float4*>(base + ((size_t)h1 * W + w1) * C);
  const float h0l = 1.f - lh, w0l = 1.f - lw;
  float4 o;
  o.x = h0l * (w0l * a.x + lw * b.x) + lh * (w0l * cc.x + lw * d.x);
  o.y = h0l * (w0l * a.y + lw * b.y) + lh * (w0l * cc.y + lw * d.y);
  o.z = h0l * (w0l * a.z + lw * b.z) + lh * (w0l * cc.z + lw * d.z);
  o.w = h0l * (w0l * a.w + lw * b.w) + lh * (w0l * cc.w + lw * d.w);
  stream_store(reinterpret_cast<float4*>(y + ((size_t)row * Wo + wo) * C + c), o);
}

// Adjoint of the x2 bilinear map.  Input i receives from outputs 2i-1 (0.25), 2i (0.75; 1 at i == 0), 2i+1 (0.75; 1 at
// i == last), 2i+2 (0.25): a lane owns one input pixel (4 channels) and gathers its 4 x 4 output taps - sixteen
// independent 16-byte loads from clamped addresses, taps outside the image carry weight 0.  Same block layout as the
// forward kernel (one block = 256 float4 of one input row).
__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(const float* __restrict__ dy, int N, int H, int W, int C,
                                                             float* __restrict__ dx, unsigned segs, unsigned nblk) {
  const unsigned C4 = (unsigned)C / 4;
  const int Ho = 2 * H, Wo = 2 * W;
  const unsigned bid = xcd_contiguous(blockIdx.x, nblk);
  if (bid >= nblk) return;
  const unsigned row = bid / segs, seg = bid - row * segs;                      // row = n * H + hi
  const unsigned n = row / (unsigned)H;
  const int hi = (int)(row - n * (unsigned)H);
  const unsigned e = seg * 256u + threadIdx.x;
  if (e >= (unsigned)W * C4) return;
  const int wi = (int)(e / C4);
  const unsigned c = (e - (unsigned)wi * C4) * 4u;
  const float wh[4] = {hi >= 1 ? 0.25f : 0.f, hi == 0 ? 1.f : 0.75f, hi == H - 1 ? 1.f : 0.75f, hi <= H - 2 ? 0.25f : 0.f};
  const float ww[4] = {wi >= 1 ? 0.25f : 0.f, wi == 0 ? 1.f : 0.75f, wi == W - 1 ? 1.f : 0.75f, wi <= W - 2 ? 0.25f : 0.f};
  const float* base = dy + (size_t)n * Ho * Wo * C + c;
  float4 g[4][4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = min(max(2 * hi - 1 + k, 0), Ho - 1);
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      const int q = min(max(2 * wi - 1 + l, 0), Wo - 1);
      g[k][l] = *reinterpret_cast<const float4*>(base + ((size_t)r * Wo + q) * C);
    }
  }
  float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float4 tr = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      tr.x = fmaf(ww[l], g[k][l].x, tr.x); tr.y = fmaf(ww[l], g[k][l].y, tr.y);
      tr.z = fmaf(ww[l], g[k][l].z, tr.z); tr.w = fmaf(ww[l], g[k][l].w, tr.w);
    }
    acc.x = fmaf(wh[k], tr.x, acc.x); acc.y = fmaf(wh[k], tr.y, acc.y);
    acc.z = fmaf(wh[k], tr.z, acc.z); acc.w = fmaf(wh[k], tr.w, acc.w);
  }
  stream_store(reinterpret_cast<float4*>(dx + ((size_t)row * W + wi) * C + c), acc);
}

__global__ void sigmoid_fwd_kernel(const float* __restrict__ x, long n, float* __restrict__ y) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    y[i] = 1.f / (1.f + __expf(-x[i]));
}
__global__ void sigmoid_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy, long n, float* __restrict__ dx) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    dx[i] = dy[i] * y[i] * (1.f - y[i]);
}

__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, int N, int C, int HW, float* __restrict__ y, int inverse) {
  const long total = (long)N * C * HW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    // i enumerates NHWC
    const int c = i % C; const long t = i / C; const long p = t % HW; const long n = t / HW;
    const long j = (n * C + c) * HW + p;
    if (inverse) y[j] = x[i]; else y[i] = x[j];
  }
}

static inline unsigned ew_grid(long n, int per_block = 256) {
  long b = cdiv(n, per_block);
  return (unsigned)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

}  // namespace xas

using namespace xas;

extern "C" size_t xas_bn_workspace_floats(long M, int C, int groups) {
  ColGeom g;
  if (col_geom(M, C, groups, &g)) return 0;
  return (size_t)g.G * g.nslab * 2 * C + C;
}

#ifndef XAS_EW_CAP
#define XAS_EW_CAP 8192                  // blocks of a streaming launch over all groups (r05 re-sweep on one stream, in-box: 2048 +0.9 ms/step, 4096 +0.3, 8192 best, 16384 +0.4)
#endif
static inline unsigned ew_grid_g(long n_per_group, int groups) {
  long b = cdiv(n_per_group, 256);
  const long cap = groups > 1 ? cdiv(XAS_EW_CAP, groups) : XAS_EW_CAP;
  return (unsigned)(b > cap ? cap : (b < 1 ? 1 : b));
}

static int col_args(ColArgs* a, const ColGeom& g, long M, int C, float* workspace) {
  a->M = M; a->C = C; a->g = g; a->partial = workspace;
  a->ticket = take_tickets(g.ncb);
  XAS_REQUIRE(a->ticket != nullptr, "column reduce: could not allocate the ticket counters");
  XAS_REQUIRE((size_t)g.G * g.nslab * 2 * C * sizeof(float) < 0x7fffff00ul, "column reduce: partial buffer too large");
  return 0;
}

extern "C" int xas_bn_stats(const float* x, long M, int C, int groups, float* mean, float* var_biased, long out_stride,
                            float* count_out, float* workspace, float* running_mean, float* running_var, float momentum,
                            long count, void* stream) {
  ColGeom g;
  if (col_geom(M, C, groups, &g)) return 1;
  XAS_REQUIRE(x && mean && var_biased && workspace && out_stride >= C, "bn_stats: null buffer / bad stride");
  XAS_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_stats: running buffers come in pairs");
  XAS_REQUIRE((((uintptr_t)mean | (uintptr_t)var_biased) & 15) == 0 && out_stride % 4 == 0,
              "bn_stats: outputs must be 16-byte aligned (out_stride a multiple of 4 floats)");
  ColArgs a{};
  if (col_args(&a, g, M, C, workspace)) return 1;
  a.x = x; a.out1 = mean; a.out2 = var_biased; a.out_stride = out_stride; a.count_out = count_out;
  a.running_mean = running_mean; a.running_var = running_var; a.momentum = momentum;
  a.unbias = count > 1 ? (float)((double)count / (double)(count - 1)) : 1.f;
  hipLaunchKernelGGL(col_reduce_kernel<0>, dim3(g.nslab, g.ncb, g.G), dim3(256), 0, as_stream(stream), a);
  XAS_LAUNCH_CHECK();
  return 0;
}

// Statistics from the per-tile partial sums a convolution epilogue left (xas_conv_fwd_bnstats): partial is
// [groups * tiles_per_group... rows][C][2] = (sum(v - pivot), sum((v - pivot)^2)) over rows_per_group / (rows / groups)
// activation rows each.  One launch: column sums of the [rows][2C] matrix + folded finalize.
extern "C" int xas_bn_stats_from_partials(const float* partial, long rows, int C, int groups, long rows_per_group,
                                          const float* pivot, float* mean, float* var_biased, long out_stride,
                                          float* count_out, float* workspace, float* running_mean, float* running_var,
                                          float momentum, void* stream) {
  ColGeom g;
  if (col_geom(rows, 2 * C, groups, &g)) return 1;
  XAS_REQUIRE(partial && mean && var_biased && workspace && out_stride >= C && rows_per_group > 0,
              "bn_stats_from_partials: null buffer / bad stride");
  XAS_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_stats_from_partials: running buffers come in pairs");
  XAS_REQUIRE((((uintptr_t)mean | (uintptr_t)var_biased | (uintptr_t)partial) & 15) == 0 && out_stride % 4 == 0 && C % 4 == 0,
              "bn_stats_from_partials: buffers must be 16-byte aligned, C and out_stride multiples of 4");
  ColArgs a{};
  if (col_args(&a, g, rows, 2 * C, workspace)) return 1;
  a.x = partial; a.aux = pivot; a.out1 = mean; a.out2 = var_biased; a.out_stride = out_stride; a.count_out = count_out;
  a.running_mean = running_mean; a.running_var = running_var; a.momentum = momentum;
  a.rows_real = rows_per_group;
  a.unbias = rows_per_group > 1 ? (float)((double)rows_per_group / (double)(rows_per_group - 1)) : 1.f;
  hipLaunchKernelGGL(col_reduce_kernel<5>, dim3(g.nslab, g.ncb, g.G), dim3(256), 0, as_stream(stream), a);
  XAS_LAUNCH_CHECK();
  return 0;
}

// Batch-norm backward sums from the per-tile partial sums a data-gradient epilogue left (xas_conv_dgrad_bn_bwd):
// partial [rows][2][C] (sum dz | sum dz * xhat over the activation rows of a tile) -> sums [groups][2][C], and the LOCAL
// parameter gradients added into dbeta_acc / dgamma_acc (may be NULL).
extern "C" int xas_bn_bwd_sums_from_partials(const float* partial, long rows, int C, int groups, float* sums,
                                             float* workspace, float* dbeta_acc, float* dgamma_acc, void* stream) {
  ColGeom g;
  if (col_geom(rows, 2 * C, groups, &g)) return 1;
  XAS_REQUIRE(partial && sums && workspace && C % 4 == 0, "bn_bwd_sums_from_partials: null buffer / C not a multiple of 4");
  XAS_REQUIRE((dbeta_acc == nullptr) == (dgamma_acc == nullptr), "bn_bwd_sums_from_partials: gradient accumulators come in pairs");
  ColArgs a{};
  if (col_args(&a, g, rows, 2 * C, workspace)) return 1;
  a.x = partial; a.out1 = sums; a.out_stride = 2 * (long)C;
  a.out2 = nullptr;                                                    // (no second sums in this mode)
  a.acc1 = dbeta_acc; a.acc2 = dgamma_acc;
  hipLaunchKernelGGL(col_reduce_kernel<6>, dim3(g.nslab, g.ncb, g.G), dim3(256), 0, as_stream(stream), a);
  XAS_LAUNCH_CHECK();
  return 0;
}

extern "C" int xas_bn_sync_merge(const float* gathered, int world, int groups, int C, long msg_stride, float* mean,
                                 float* var_biased, float* running_mean, float* running_var, float momentum,
                                 void* stream) {
  XAS_REQUIRE(gathered && mean && var_biased && world >= 1 && groups >= 1 && C >= 1 && msg_stride >= 2 * (long)C + 1,
              "bn_sync_merge: bad arguments");
  XAS_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_sync_merge: running buffers come in pairs");
  hipLaunchKernelGGL(bn_sync_merge_kernel, dim3((unsigned)cdiv(C, 64)), dim3(64), 0, as_stream(stream), gathered, world,
                     groups, C, msg_stride, mean, var_biased, running_mean, running_var, momentum);
  XAS_LAUNCH_CHECK();
  return 0;
}

extern "C" int xas_col_sum(const float* x, long M, int C, float* out, float* workspace, void* stream) {
  ColGeom g;
  if (col_geom(M, C, 1, &g)) return 1;
  XAS_REQUIRE(x && out && workspace, "col_sum: null buffer");
  ColArgs a{};
  if (col_args(&a, g, M, C, workspace)) return 1;
  a.x = x; a.out1 = out; a.out2 = workspace + (size_t)g.nslab * 2 * C;     // second sums are unused: workspace tail
  a.out_stride = C;
  hipLaunchKernelGGL(col_reduce_kernel<2>, dim3(g.nslab, g.ncb, 1), dim3(256), 0, as_stream(stream), a);
  XAS_LAUNCH_CHECK();
  return 0;
}

extern "C" int xas_bn_apply(const float* x, const float* mean, const float* var_biased, const float* gamma,
                            const float* beta, const float* residual, float eps, int act, long M, int C, int groups,
                            float* y, uint8_t* mask_out, void* stream) {
  return xas_bn_apply_amax(x, mean, var_biased, gamma, beta, residual, eps, act, M, C, groups, y, mask_out, nullptr, stream);
}

extern "C" int xas_bn_apply_amax(const float* x, const float* mean, const float* var_biased, const float* gamma,
                                 const float* beta, const float* residual, float eps, int act, long M, int C, int groups,
                                 float* y, uint8_t* mask_out, float* amax_out, void* stream) {
  XAS_REQUIRE(x && mean && var_biased && gamma && beta && y, "bn_apply: null buffer");
  XAS_REQUIRE(M > 0 && C >= 4 && C % 4 == 0 && act >= 0 && act <= 2 && groups >= 1 && M % groups == 0,
              "bn_apply: bad shape M=%ld C=%d act=%d groups=%d", M, C, act, groups);
  const long n4g = (M / groups) * (C / 4);
  {
    // streaming form: every thread keeps one channel quadruple (threads per grid row a multiple of C/4)
    const int C4 = C / 4;
    unsigned gx = ew_grid_g(n4g, groups);
    const bool pow2 = (C4 & (C4 - 1)) == 0;
    if (pow2 && C4 > 256) gx = (gx + (C4 / 256) - 1) / (C4 / 256) * (C4 / 256);
    if (pow2 && ((long)gx * 256) % C4 == 0 && !(tune_flags() & (1 << 22))) {
      const dim3 grid(gx, groups);
      const float4* x4 = reinterpret_cast<const float4*>(x);
      const float4* r4 = reinterpret_cast<const float4*>(residual);
      float4* y4 = reinterpret_cast<float4*>(y);
#define XAS_BN_FWD(ACT, RES, MK)                                                                                         \
  hipLaunchKernelGGL((bn_apply_stream_kernel<ACT, RES, MK>), grid, dim3(256), 0, as_stream(stream), x4, mean, var_biased, \
                     gamma, beta, r4, eps, n4g, C4, y4, mask_out, amax_out)
      if (residual && mask_out) { if (act == 1) XAS_BN_FWD(1, true, true); else if (act == 2) XAS_BN_FWD(2, true, true); else XAS_BN_FWD(0, true, true); }
      else if (residual) { if (act == 1) XAS_BN_FWD(1, true, false); else if (act == 2) XAS_BN_FWD(2, true, false); else XAS_BN_FWD(0, true, false); }
      else { if (act == 1) XAS_BN_FWD(1, false, false); else if (act == 2) XAS_BN_FWD(2, false, false); else XAS_BN_FWD(0, false, false); }
#undef XAS_BN_FWD
      XAS_LAUNCH_CHECK();
      return 0;
    }
  }
  hipLaunchKernelGGL(bn_apply_kernel, dim3(ew_grid_g(n4g, groups), groups), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const float4*>(x), mean, var_biased, gamma, beta,
                     reinterpret_cast<const float4*>(residual), eps, act, n4g, C / 4, reinterpret_cast<float4*>(y), mask_out,
                     amax_out);
  XAS_LAUNCH_CHECK();
  return 0;
}

extern "C" int xas_bn_update_running(const float* mean, const float* var_biased, float* running_mean,
                                     float* running_var, float momentum, long count, int C, int groups, void* stream) {
  XAS_REQUIRE(mean && var_biased && running_mean && running_var && C > 0 && count > 0 && groups >= 1,
              "bn_update_running: bad arguments");
  const float unbias = count > 1 ? (float)((double)count / (double)(count - 1)) : 1.f;
  hipLaunchKernelGGL(bn_running_kernel, dim3((unsigned)cdiv(C, 64)), dim3(64), 0, as_stream(stream), mean, var_biased,
                     running_mean, running_var, momentum, unbias, C, groups);
  XAS_LAUNCH_CHECK();
  return 0;
}

// acc[c] += sum over rows of x[:, c] (bias gradients added straight into the gradient arena, on the weight-gradient stream)
extern "C" int xas_col_sum_acc(const float* x, long M, int C, float* acc, float* workspace, void* stream) {
  ColGeom g;
  if (col_geom(M, C, 1, &g)) return 1;
  XAS_REQUIRE(x && acc && workspace && (((uintptr_t)acc) & 15) == 0, "col_sum_acc: null / misaligned buffer");
  ColArgs a{};
  if (col_args(&a, g, M, C, workspace)) return 1;
  float* scratch = workspace + (size_t)g.nslab * 2 * C;                 // C floats: plain sums, second sums and their accumulator
  a.x = x; a.out1 = scratch; a.out2 = scratch; a.out_stride = C;
  a.acc1 = acc; a.acc2 = scratch;
  hipLaunchKernelGGL(col_reduce_kernel<2>, dim3(g.nslab, g.ncb, 1), dim3(256), 0, as_stream(stream), a);
  XAS_LAUNCH_CHECK();
  return 0;
}

extern "C" int xas_bn_bwd_reduce(const float* x, const float* y, const float* dy, const float* mean,
                                 const float* var_biased, const float* gamma, const float* beta, float eps, int act,
                                 long M, int C, int groups, float* sums, float* workspace,
                                 float* dbeta_acc, float* dgamma_acc, const uint8_t* mask, void* stream) {
  ColGeom g;
  if (col_geom(M, C, groups, &g)) return 1;
  XAS_REQUIRE(!mask || (x && act != 0), "bn_bwd_reduce: the sign-mask form needs x and an activation");
  XAS_REQUIRE(dy && mean && var_biased && sums && workspace && (act == 0 || y || mask || (x && gamma && beta)),
              "bn_bwd_reduce: null buffer (an activation needs y, or x with gamma and beta)");
  XAS_REQUIRE(x || (act != 0 && y && gamma && beta), "bn_bwd_reduce: without x the layer needs an activation, y, gamma, beta");
  XAS_REQUIRE((dbeta_acc == nullptr) == (dgamma_acc == nullptr), "bn_bwd_reduce: gradient accumulators come in pairs");
  const bool lean = (tune_flags() & 262144) != 0;      // shipped: the 86-VGPR build (tune bit 18 selects the <= 64-VGPR one: see col_reduce_lean_kernel)
  ColArgs a{};
  if (col_args(&a, g, M, C, workspace)) return 1;
  a.dy = dy; a.mean = mean; a.var = var_biased; a.eps = eps; a.act = act;
  a.out1 = sums; a.out2 = sums + C; a.out_stride = 2 * (long)C;            // [G][2][C]
  a.acc1 = dbeta_acc; a.acc2 = dgamma_acc;
  const dim3 grid(g.nslab, g.ncb, g.G);
  if (mask) {                             // sign bits instead of y (layers with a residual: 1/16 of y's bytes)
    a.x = x; a.y = nullptr; a.mask = mask;
    hipLaunchKernelGGL(lean ? col_reduce_lean_kernel<1> : col_reduce_kernel<1>, grid, dim3(256), 0, as_stream(stream), a);
  } else
  if (x && act != 0 && y == nullptr) {   // y-free form: activation mask re-derived from x (layers without a residual)
    a.x = x; a.y = gamma; a.aux = beta;
    hipLaunchKernelGGL(lean ? col_reduce_lean_kernel<4> : col_reduce_kernel<4>, grid, dim3(256), 0, as_stream(stream), a);
  } else if (x) {
    a.x = x; a.y = y;
    hipLaunchKernelGGL(lean ? col_reduce_lean_kernel<1> : col_reduce_kernel<1>, grid, dim3(256), 0, as_stream(stream), a);
  } else {            // x-free form: xhat recovered from the saved output (one activation tensor less to read)
    a.x = gamma; a.y = y; a.aux = beta;
    hipLaunchKernelGGL(lean ? col_reduce_lean_kernel<3> : col_reduce_kernel<3>, grid, dim3(256), 0, as_stream(stream), a);
  }
  XAS_LAUNCH_CHECK();
  return 0;
}

extern "C" int xas_bn_bwd_apply(const float* x, const float* y, const float* dy, const float* mean,
                                const float* var_biased, const float* gamma, const float* beta, const float* sums,
                                float eps, int act, long M, int C, int groups, double count, float* dx,
                                float* dresidual, const uint8_t* mask, void* stream) {
  return xas_bn_bwd_apply_amax(x, y, dy, mean, var_biased, gamma, beta, sums, eps, act, M, C, groups, count, dx, dresidual, mask,
                               nullptr, stream);
}

extern "C" int xas_bn_bwd_apply_amax(const float* x, const float* y, const float* dy, const float* mean,
                                     const float* var_biased, const float* gamma, const float* beta, const float* sums,
                                     float eps, int act, long M, int C, int groups, double count, float* dx,
                                     float* dresidual, const uint8_t* mask, float* amax_out, void* stream) {
  XAS_REQUIRE(!mask || x, "bn_bwd_apply: the sign-mask form needs x");
  XAS_REQUIRE(dy && mean && var_biased && gamma && sums && dx && (act == 0 || y || mask || (x && beta)),
              "bn_bwd_apply: null buffer (an activation needs y, or x with beta)");
  XAS_REQUIRE(y || mask || !dresidual || act == 0, "bn_bwd_apply: the y-free form is for layers without a residual");
  XAS_REQUIRE(x || (act == 2 && y && beta),
              "bn_bwd_apply: without x the layer needs an INVERTIBLE activation (leaky ReLU), y and beta: dx needs xhat "
              "of every element, also where ReLU clipped the output");
  XAS_REQUIRE(M > 0 && C >= 4 && C % 4 == 0 && count > 0 && groups >= 1 && M % groups == 0, "bn_bwd_apply: bad shape");
  const long n4g = (M / groups) * (C / 4);
  {
    const int C4 = C / 4;
    unsigned gx = ew_grid_g(n4g, groups);
    const bool pow2 = (C4 & (C4 - 1)) == 0;
    if (pow2 && C4 > 256) gx = (gx + (C4 / 256) - 1) / (C4 / 256) * (C4 / 256);
    // modes of the model: ReLU without residual (sign from x), leaky ReLU without x (physique net), residual layers
    // (sign from y or from the mask bytes), no activation (projection norms); anything else takes the generic kernel
    int mode = -1;
    if (act == 0 && x && !dresidual) mode = 0;
    else if (act == 1 && x && !y && !mask && !dresidual) mode = 1;
    else if (act == 2 && !x && y && !dresidual) mode = 2;
    else if (act == 1 && x && y && !mask) mode = dresidual ? 3 : 4;
    else if (act == 1 && x && mask) mode = dresidual ? 5 : 6;
    if (mode >= 0 && pow2 && ((long)gx * 256) % C4 == 0 && !(tune_flags() & (1 << 22))) {
      const dim3 grid(gx, groups);
      const float4* x4 = reinterpret_cast<const float4*>(x);
      const float4* y4 = reinterpret_cast<const float4*>(y);
      const float4* d4 = reinterpret_cast<const float4*>(dy);
      float4* o4 = reinterpret_cast<float4*>(dx);
      float4* r4 = reinterpret_cast<float4*>(dresidual);
      const float ic = (float)(1.0 / count);
#define XAS_BN_BWD(ACT, SIGN, XH, DR)                                                                                      \
  hipLaunchKernelGGL((bn_bwd_apply_stream_kernel<ACT, SIGN, XH, DR>), grid, dim3(256), 0, as_stream(stream), x4, y4, d4, mean, \
                     var_biased, gamma, beta, sums, eps, n4g, C4, ic, o4, r4, mask, amax_out)
      switch (mode) {
        case 0: XAS_BN_BWD(0, 0, true, false); break;
        case 1: XAS_BN_BWD(1, 2, true, false); break;
        case 2: XAS_BN_BWD(2, 1, false, false); break;
        case 3: XAS_BN_BWD(1, 1, true, true); break;
        case 4: XAS_BN_BWD(1, 1, true, false); break;
        case 5: XAS_BN_BWD(1, 3, true, true); break;
        default: XAS_BN_BWD(1, 3, true, false); break;
      }
#undef XAS_BN_BWD
      XAS_LAUNCH_CHECK();
      return 0;
    }
  }
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_grid_g(n4g, groups), groups), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const float4*>(x), reinterpret_cast<const float4*>(y),
                     reinterpret_cast<const float4*>(dy), mean, var_biased, gamma, beta, sums, eps, act, n4g,
                     C / 4, (float)(1.0 / count), reinterpret_cast<float4*>(dx), reinterpret_cast<float4*>(dresidual), mask, amax_out);
  XAS_LAUNCH_CHECK();
  return 0;
}

// max |x| over a tensor -> *amax_out (merged with atomicMax: the caller zeroes it once).  For conv inputs that no kernel of
// this library wrote (images, rendered masks, tensors handed in from outside): one streaming read, HBM bound.
__global__ __launch_bounds__(256) void abs_max_kernel(const float* __restrict__ x, long n, float* __restrict__ amax_out) {
  float amx = 0.f;
  const long n4 = n / 4, stride = (long)gridDim.x * blockDim.x;
  const float4* x4 = reinterpret_cast<const float4*>(x);
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = stream_load(x4 + i + u * stride);
#pragma unroll
    for (int u = 0; u < 4; ++u) amx = amax4(amx, v[u]);
  }
  for (; i < n4; i += stride) amx = amax4(amx, x4[i]);
  if (blockIdx.x == 0 && threadIdx.x < (int)(n - n4 * 4)) amx = fmaxf(amx, fabsf(x[n4 * 4 + threadIdx.x]));
  publish_amax(amax_out, amx);
}

extern "C" int xas_abs_max(const float* x, long n, float* amax_out, void* stream) {
  XAS_REQUIRE(x && amax_out && n > 0 && (((uintptr_t)x) & 15) == 0, "abs_max: need a 16-byte aligned tensor and an output float");
  hipLaunchKernelGGL(abs_max_kernel, dim3(ew_grid(cdiv(n, 4))), dim3(256), 0, as_stream(stream), x, n, amax_out);
  XAS_LAUNCH_CHECK();
  return 0;
}

extern "C" int xas_maxpool3x3s2_fwd(const float* x, int N, int H, int W, int C, float* y, int8_t* idx, void* stream) {
  XAS_REQUIRE(x && y && idx && N > 0 && H > 0 && W > 0 && C % 4 == 0, "maxpool: bad arguments");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  XAS_REQUIRE((long)N * H < 0x7fffffffl, "maxpool: too many rows");
  const long rows = (long)N * Ho;                               // one block per output row (grid-stride beyond 8 192 blocks)
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3((unsigned)(rows < 8192 ? rows : 8192)), dim3(256), 0, as_stream(stream), x, N, H, W, C, Ho,
                     Wo, y, idx);
  XAS_LAUNCH_CHECK();
  return 0;
}

extern "C" int xas_maxpool3x3s2_bwd(const float* dy, const int8_t* idx, int N, int H, int W, int C, float* dx,
                                    void* stream) {
  XAS_REQUIRE(dy && dx && idx && N > 0 && H > 0 && W > 0 && C % 4 == 0, "maxpool bwd: bad arguments");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  XAS_REQUIRE((long)N * H < 0x7fffffffl, "maxpool bwd: too many rows");
  const long rows = (long)N * H;
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3((unsigned)(rows < 16384 ? rows : 16384)), dim3(256), 0, as_stream(stream), dy, idx, N, H,
                     W, C, Ho, Wo, dx);
  XAS_LAUNCH_CHECK();
  return 0;
}

extern "C" int xas_upsample2x_fwd(const float* x, int N, int H, int W, int C, float* y, void* stream) {
  XAS_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C % 4 == 0, "upsample2x: bad arguments");
  const long segs = cdiv((long)2 * W * (C / 4), 256), blocks = (long)N * 2 * H * segs;
  XAS_REQUIRE(blocks < 0x7ffffff0l && (long)N * 4 * H * W * C < (1l << 40), "upsample2x: tensor too large");
  hipLaunchKernelGGL(upsample2x_fwd_kernel, dim3((unsigned)(8 * cdiv(blocks, 8))), dim3(256), 0, as_stream(stream), x, N, H,
                     W, C, y, (unsigned)segs, (unsigned)blocks);
  XAS_LAUNCH_CHECK();
  return 0;
}

extern "C" int xas_upsample2x_bwd(const float* dy, int N, int H, int W, int C, float* dx, void* stream) {
  XAS_REQUIRE(dy && dx && N > 0 && H > 0 && W > 0 && C % 4 == 0, "upsample2x bwd: bad arguments");
  const long segs = cdiv((long)W * (C / 4), 256), blocks = (long)N * H * segs;
  XAS_REQUIRE(blocks < 0x7ffffff0l && (long)N * 4 * H * W * C < (1l << 40), "upsample2x bwd: tensor too large");
  hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3((unsigned)(8 * cdiv(blocks, 8))), dim3(256), 0, as_stream(stream), dy, N,
                     H, W, C, dx, (unsigned)segs, (unsigned)blocks);
  XAS_LAUNCH_CHECK();
  return 0;
}

extern "C" int xas_sigmoid_fwd(const float* x, long n, float* y, void* stream) {
  XAS_REQUIRE(x && y && n > 0, "sigmoid: bad arguments");
  hipLaunchKernelGGL(sigmoid_fwd_kernel, dim3(ew_grid(n)), dim3(256), 0, as_stream(stream), x, n, y);
  XAS_LAUNCH_CHECK();
  return 0;
}

extern "C" int xas_sigmoid_bwd(const float* y, const float* dy, long n, float* dx, void* stream) {
  XAS_REQUIRE(y && dy && dx && n > 0, "sigmoid bwd: bad arguments");
  hipLaunchKernelGGL(sigmoid_bwd_kernel, dim3(ew_grid(n)), dim3(256), 0, as_stream(stream), y, dy, n, dx);
  XAS_LAUNCH_CHECK();
  return 0;
}

extern "C" int xas_nchw_to_nhwc(const float* x, int N, int C, int H, int W, float* y, void* stream) {
  XAS_REQUIRE(x && y && N > 0 && C > 0 && H > 0 && W > 0, "nchw_to_nhwc: bad arguments");
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(ew_grid((long)N * C * H * W)), dim3(256), 0, as_stream(stream), x, N, C,
                     H * W, y, 0);
  XAS_LAUNCH_CHECK();
  return 0;
}

extern "C" int xas_nhwc_to_nchw(const float* x, int N, int C, int H, int W, float* y, void* stream) {
  XAS_REQUIRE(x && y && N > 0 && C > 0 && H > 0 && W > 0, "nhwc_to_nchw: bad arguments");
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(ew_grid((long)N * C * H * W)), dim3(256), 0, as_stream(stream), x, N, C,
                     H * W, y, 1);
  XAS_LAUNCH_CHECK();
  return 0;
}
