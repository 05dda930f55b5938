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

// z = (x - mean) * (rstd * gamma) + beta with a FIXED operation order (no compiler-chosen contraction), so the
// backward kernels that re-derive the ReLU mask from x reproduce the forward decision bit for bit.
__device__ __forceinline__ float bn_affine(float x, float m, float rs_g, float b) {
  return __fmaf_rn(__fsub_rn(x, m), rs_g, b);
}

// ---------------------------------------------------------------- column reductions
// x is [M][C]; a block owns CB = min(C,256) channels (TX = CB/4 lanes along C) and a slab
// of rows; it emits per channel sum(f1), sum(f2) of two per-element functions.
struct ColGeom { int TX, TY, CB, ncb, nslab; long rows_per_slab; };

static int col_geom(long M, int C, ColGeom* g) {
  XAS_REQUIRE(M > 0 && C >= 4 && C % 4 == 0, "column reduce: channel count %d must be a positive multiple of 4", C);
  int cb = 4;                                  // largest power of two <= 256 dividing C
  while (cb < 256 && C % (cb * 2) == 0) cb *= 2;
  g->CB = cb;
  g->TX = g->CB / 4; g->TY = 256 / g->TX; g->ncb = C / g->CB;
  static const long kWant[4] = {256, 512, 128, 64};           // tune bits 15-16 (experiment)
  long want = kWant[(tune_flags() >> 15) & 3] / g->ncb;         // ~1 block per CU in total; keeps the finalize pass short
  long maxslab = cdiv(M, (long)g->TY * 8);   // at least 8 rows per thread
  if (want > maxslab) want = maxslab;
  if (want < 1) want = 1;
  g->rows_per_slab = cdiv(M, want);
  g->nslab = (int)cdiv(M, g->rows_per_slab);
  return 0;
}

template <int MODE, int UNR = 4>   // 0: stats of x around pivot ; 1: bn backward sums ; 2: plain column sums of x ;
                      // 3: bn backward sums WITHOUT x: xhat = (z - beta)/gamma with z recovered from y (act != 0)
                      // 4: bn backward sums WITHOUT y: the activation mask is re-derived from x (`y` carries gamma,
                      //    `aux` carries beta)
__device__ __forceinline__ void col_reduce_body(const float* __restrict__ x, const float* __restrict__ y,
                                                const float* __restrict__ dy, const float* __restrict__ mean,
                                                const float* __restrict__ var, float eps, int act, long M,
                                                int C, ColGeom g, float* __restrict__ partial,
                                                const float* __restrict__ aux = nullptr) {
  __shared__ float4 red[2][256];
  const int tx = threadIdx.x % g.TX, ty = threadIdx.x / g.TX;
  const int c = blockIdx.y * g.CB + tx * 4;
  const long r0 = (long)blockIdx.x * g.rows_per_slab;
  const long r1 = min(M, r0 + g.rows_per_slab);
  float4 s1 = make_float4(0, 0, 0, 0), s2 = make_float4(0, 0, 0, 0);
  float4 p0, p1;   // MODE 0: pivot ; MODE 1: mean, invstd
  float4 rsg = make_float4(0, 0, 0, 0), bt = make_float4(0, 0, 0, 0);      // MODE 4: rstd * gamma, beta
  if (MODE == 4) {
    p0 = *reinterpret_cast<const float4*>(mean + c);
    const float4 v = *reinterpret_cast<const float4*>(var + c);
    p1 = make_float4(rsqrtf(v.x + eps), rsqrtf(v.y + eps), rsqrtf(v.z + eps), rsqrtf(v.w + eps));
    const float4 gm = *reinterpret_cast<const float4*>(y + c);
    rsg = make_float4(__fmul_rn(p1.x, gm.x), __fmul_rn(p1.y, gm.y), __fmul_rn(p1.z, gm.z), __fmul_rn(p1.w, gm.w));
    bt = *reinterpret_cast<const float4*>(aux + c);
  } else if (MODE == 0) {
    p0 = *reinterpret_cast<const float4*>(x + c);          // pivot = first row
    p1 = p0;
  } else if (MODE == 2) {
    p0 = make_float4(0, 0, 0, 0); p1 = p0;
  } else if (MODE == 3) {                                  // `mean` carries beta, `var` carries gamma
    p0 = *reinterpret_cast<const float4*>(mean + c);
    const float4 gm = *reinterpret_cast<const float4*>(var + c);
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
      float4 g4 = *reinterpret_cast<const float4*>(dy + r * C + c);
      const float4 yv = *reinterpret_cast<const float4*>(y + r * C + c);
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
    const float4 xv = *reinterpret_cast<const float4*>(x + r * C + c);
    if (MODE == 4) {
      float4 g4 = *reinterpret_cast<const float4*>(dy + r * C + c);
      const float neg = act == 1 ? 0.f : 0.01f;
      g4.x *= bn_affine(xv.x, p0.x, rsg.x, bt.x) > 0.f ? 1.f : neg; g4.y *= bn_affine(xv.y, p0.y, rsg.y, bt.y) > 0.f ? 1.f : neg;
      g4.z *= bn_affine(xv.z, p0.z, rsg.z, bt.z) > 0.f ? 1.f : neg; g4.w *= bn_affine(xv.w, p0.w, rsg.w, bt.w) > 0.f ? 1.f : neg;
      s1.x += g4.x; s1.y += g4.y; s1.z += g4.z; s1.w += g4.w;
      s2.x = fmaf(g4.x, (xv.x - p0.x) * p1.x, s2.x); s2.y = fmaf(g4.y, (xv.y - p0.y) * p1.y, s2.y);
      s2.z = fmaf(g4.z, (xv.z - p0.z) * p1.z, s2.z); s2.w = fmaf(g4.w, (xv.w - p0.w) * p1.w, s2.w);
      continue;
    }
    if (MODE == 2) {
      s1.x += xv.x; s1.y += xv.y; s1.z += xv.z; s1.w += xv.w;
    } else if (MODE == 0) {
      const float a = xv.x - p0.x, b = xv.y - p0.y, cc = xv.z - p0.z, d = xv.w - p0.w;
      s1.x += a; s1.y += b; s1.z += cc; s1.w += d;
      s2.x = fmaf(a, a, s2.x); s2.y = fmaf(b, b, s2.y); s2.z = fmaf(cc, cc, s2.z); s2.w = fmaf(d, d, s2.w);
    } else {
      float4 g4 = *reinterpret_cast<const float4*>(dy + r * C + c);
      if (act) {
        const float4 yv = *reinterpret_cast<const float4*>(y + r * C + c);
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
      const float4 a = red[0][threadIdx.x + s * g.TX], b = red[1][threadIdx.x + s * g.TX];
      float4& u = red[0][threadIdx.x]; float4& v = red[1][threadIdx.x];
      u.x += a.x; u.y += a.y; u.z += a.z; u.w += a.w;
      v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
    }
    __syncthreads();
  }
  if (ty == 0) {
    float* o = partial + ((size_t)blockIdx.x * 2) * C + c;
    *reinterpret_cast<float4*>(o) = red[0][tx];
    *reinterpret_cast<float4*>(o + C) = red[1][tx];
  }
}

template <int MODE>
__global__ __launch_bounds__(256) void col_reduce_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                         const float* __restrict__ dy, const float* __restrict__ mean,
                                                         const float* __restrict__ var, float eps, int act, long M,
                                                         int C, ColGeom g, float* __restrict__ partial,
                                                         const float* __restrict__ aux = nullptr) {
  col_reduce_body<MODE, 4>(x, y, dy, mean, var, eps, act, M, C, g, partial, aux);
}

// Same kernel compiled for at most 64 VGPRs (the shipped one for the backward sums): in the backward pass it runs beside two
// weight-gradient blocks per CU, which leave 112 VGPRs per SIMD lane - room for two lean waves instead of one.
template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8)))
void col_reduce_lean_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ dy,
                            const float* __restrict__ mean, const float* __restrict__ var, float eps, int act, long M,
                            int C, ColGeom g, float* __restrict__ partial, const float* __restrict__ aux = nullptr) {
  col_reduce_body<MODE, 2>(x, y, dy, mean, var, eps, act, M, C, g, partial, aux);
}

// partial: [nslab][2][C] -> out1[c], out2[c]; MODE 0 converts pivot sums to mean / biased var and, when
// running buffers are given, applies the running-statistic update in the same launch.
// Block = 64 channels x (blockDim.x / 64) slab lanes; slabs are summed in double, in a fixed order (deterministic).
// Launched with 256 threads: a 1024-thread block needs 16 free wave slots on ONE compute unit, and while the
// weight-gradient kernels of the side stream fill the chip this tiny kernel waited ~20 us for a CU to drain.
constexpr int kFinalizeThreads = 256;

template <int MODE>
__global__ __launch_bounds__(kFinalizeThreads) void col_finalize_kernel(const float* __restrict__ partial,
                                                           const float* __restrict__ x, int nslab, int C, long M,
                                                           float* __restrict__ out1, float* __restrict__ out2,
                                                           float* __restrict__ running_mean,
                                                           float* __restrict__ running_var, float momentum,
                                                           float unbias, float* __restrict__ acc1 = nullptr,
                                                           float* __restrict__ acc2 = nullptr) {
  // 4 KB of LDS, not 16: while two weight-gradient blocks (2 x 74 KB) sit on every CU, a block asking for 16 KB has to
  // wait for one of them to retire (seen as 25 us launches of this 5 us kernel in the backward pass)
  __shared__ double r1[kFinalizeThreads / 64][64], r2[kFinalizeThreads / 64][64];
  const int cx = threadIdx.x & 63, sy = threadIdx.x >> 6, nl = blockDim.x >> 6;
  const int c = blockIdx.x * 64 + cx;
  double s1 = 0.0, s2 = 0.0;
  if (c < C) {
    int s = sy;
    for (; s + 15 * nl < nslab; s += 16 * nl) {        // 32 independent loads in flight per lane, summed in slab order
      float a[16], b[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        a[u] = partial[((size_t)(s + u * nl) * 2) * C + c];
        b[u] = partial[((size_t)(s + u * nl) * 2 + 1) * C + c];
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) { s1 += (double)a[u]; s2 += (double)b[u]; }
    }
    for (; s + 3 * nl < nslab; s += 4 * nl) {
      float a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a[u] = partial[((size_t)(s + u * nl) * 2) * C + c];
        b[u] = partial[((size_t)(s + u * nl) * 2 + 1) * C + c];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) { s1 += (double)a[u]; s2 += (double)b[u]; }
    }
    for (; s < nslab; s += nl) {
      s1 += (double)partial[((size_t)s * 2) * C + c];
      s2 += (double)partial[((size_t)s * 2 + 1) * C + c];
    }
  }
  r1[sy][cx] = s1; r2[sy][cx] = s2;
  __syncthreads();
  if (sy != 0 || c >= C) return;
  for (int k = 1; k < nl; ++k) { s1 += r1[k][cx]; s2 += r2[k][cx]; }
  if (MODE == 0) {
    const double m = s1 / (double)M;
    double v = s2 / (double)M - m * m;
    if (v < 0.0) v = 0.0;
    const float mean_f = (float)((double)x[c] + m), var_f = (float)v;
    out1[c] = mean_f;
    out2[c] = var_f;
    if (running_mean) {
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean_f;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (var_f * unbias);
    }
  } else {
    out1[c] = (float)s1;
    out2[c] = (float)s2;
    if (acc1) { acc1[c] += (float)s1; acc2[c] += (float)s2; }   // parameter gradients accumulated in place (.grad arena)
  }
}

__device__ __forceinline__ float act_fwd(float v, int act) {
  return act == 0 ? v : (act == 1 ? fmaxf(v, 0.f) : (v > 0.f ? v : 0.01f * v));
}

__global__ void bn_apply_kernel(const float4* __restrict__ x, const float* __restrict__ mean,
                                const float* __restrict__ var, const float* __restrict__ gamma,
                                const float* __restrict__ beta, const float4* __restrict__ res, float eps, int act,
                                long n4, int C4, float4* __restrict__ y) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    const float4 m = *reinterpret_cast<const float4*>(mean + c), v = *reinterpret_cast<const float4*>(var + c);
    const float4 g = *reinterpret_cast<const float4*>(gamma + c), b = *reinterpret_cast<const float4*>(beta + c);
    const float4 xv = x[i];
    float4 o;
    o.x = bn_affine(xv.x, m.x, __fmul_rn(rsqrtf(v.x + eps), g.x), b.x);
    o.y = bn_affine(xv.y, m.y, __fmul_rn(rsqrtf(v.y + eps), g.y), b.y);
    o.z = bn_affine(xv.z, m.z, __fmul_rn(rsqrtf(v.z + eps), g.z), b.z);
    o.w = bn_affine(xv.w, m.w, __fmul_rn(rsqrtf(v.w + eps), g.w), b.w);
    if (res) { const float4 r = res[i]; o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w; }
    o.x = act_fwd(o.x, act); o.y = act_fwd(o.y, act); o.z = act_fwd(o.z, act); o.w = act_fwd(o.w, act);
    y[i] = o;
  }
}

__global__ void bn_bwd_apply_kernel(const float4* __restrict__ x, const float4* __restrict__ y,
                                    const float4* __restrict__ dy, const float* __restrict__ mean,
                                    const float* __restrict__ var, const float* __restrict__ gamma,
                                    const float* __restrict__ beta, const float* __restrict__ sdz,
                                    const float* __restrict__ sdzx, float eps, int act, long n4, int C4,
                                    float inv_count, float4* __restrict__ dx, float4* __restrict__ dres) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    const float4 m = *reinterpret_cast<const float4*>(mean + c), v = *reinterpret_cast<const float4*>(var + c);
    const float4 g = *reinterpret_cast<const float4*>(gamma + c);
    const float4 a = *reinterpret_cast<const float4*>(sdz + c), b = *reinterpret_cast<const float4*>(sdzx + c);
    float4 dz = dy[i];
    float4 yv = make_float4(0, 0, 0, 0);
    if (act && y) {
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
    dx[i] = o;
  }
}

__global__ void bn_running_kernel(const float* mean, const float* var, float* rm, float* rv, float momentum,
                                  float unbias, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  rm[c] = (1.f - momentum) * rm[c] + momentum * mean[c];
  rv[c] = (1.f - momentum) * rv[c] + momentum * (var[c] * unbias);
}

// ---------------------------------------------------------------- max pool 3x3 s2 p1
__global__ void maxpool_fwd_kernel(const float* __restrict__ x, int N, int H, int W, int C, int Ho, int Wo,
                                   float* __restrict__ y, int8_t* __restrict__ idx) {
  const int C4 = C / 4;
  const long total = (long)N * Ho * Wo * C4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    long t = i / C4;
    const int wo = t % Wo; t /= Wo;
    const int ho = t % Ho; const int n = t / Ho;
    float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    int b0 = 0, b1 = 0, b2 = 0, b3 = 0;
    bool first = true;
    for (int r = 0; r < 3; ++r) {
      const int hi = ho * 2 - 1 + r;
      if ((unsigned)hi >= (unsigned)H) continue;
      for (int s = 0; s < 3; ++s) {
        const int wi = wo * 2 - 1 + s;
        if ((unsigned)wi >= (unsigned)W) continue;
        const float4 v = *reinterpret_cast<const float4*>(x + (((size_t)n * H + hi) * W + wi) * C + c);
        const int tap = r * 3 + s;
        if (first || v.x > best.x) { best.x = v.x; b0 = tap; }
        if (first || v.y > best.y) { best.y = v.y; b1 = tap; }
        if (first || v.z > best.z) { best.z = v.z; b2 = tap; }
        if (first || v.w > best.w) { best.w = v.w; b3 = tap; }
        first = false;
      }
    }
    *reinterpret_cast<float4*>(y + i * 4) = best;
    *reinterpret_cast<char4*>(idx + i * 4) = make_char4((char)b0, (char)b1, (char)b2, (char)b3);
  }
}

__global__ void maxpool_bwd_kernel(const float* __restrict__ dy, const int8_t* __restrict__ idx, int N, int H, int W,
                                   int C, int Ho, int Wo, float* __restrict__ dx) {
  const int C4 = C / 4;
  const long total = (long)N * H * W * C4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    long t = i / C4;
    const int wi = t % W; t /= W;
    const int hi = t % H; const int n = t / H;
    float4 acc = make_float4(0, 0, 0, 0);
    // output windows ho with ho*2-1 <= hi <= ho*2+1
    for (int ho = (hi) / 2; ho <= (hi + 1) / 2; ++ho) {
      if (ho >= Ho) continue;
      const int r = hi - (ho * 2 - 1);
      if (r < 0 || r > 2) continue;
      for (int wo = (wi) / 2; wo <= (wi + 1) / 2; ++wo) {
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
    *reinterpret_cast<float4*>(dx + i * 4) = acc;
  }
}

// ---------------------------------------------------------------- bilinear x2 (align_corners=False)
__global__ void upsample2x_fwd_kernel(const float* __restrict__ x, int N, int H, int W, int C, float* __restrict__ y) {
  const int C4 = C / 4, Ho = 2 * H, Wo = 2 * W;
  const long total = (long)N * Ho * Wo * C4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    long t = i / C4;
    const int wo = t % Wo; t /= Wo;
    const int ho = t % Ho; const int n = t / Ho;
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
    *reinterpret_cast<float4*>(y + i * 4) = o;
  }
}

// 1-D adjoint taps of the x2 bilinear map: input i receives from outputs {2i-1,2i,2i+1,2i+2}
__device__ __forceinline__ int up_taps(int i, int H, int* o, float* w) {
  int n = 0;
  o[n] = 2 * i; w[n++] = (i == 0) ? 1.f : 0.75f;                 // out 2i: 0.75 (+0.25 clamp at i==0)
  o[n] = 2 * i + 1; w[n++] = (i == H - 1) ? 1.f : 0.75f;         // out 2i+1: 0.75 (+0.25 clamp at the end)
  if (i >= 1) { o[n] = 2 * i - 1; w[n++] = 0.25f; }
  if (i <= H - 2) { o[n] = 2 * i + 2; w[n++] = 0.25f; }
  return n;
}

// Adjoint of the x2 bilinear map.  A thread owns one input column (and 4 channels) over a strip of kUpStrip input rows:
// every output row of the strip is combined horizontally ONCE (4 loads) and added to the one or two input rows it
// belongs to, 9 loads per result instead of the 16 of a thread-per-pixel gather.
constexpr int kUpStrip = 8;

__global__ void upsample2x_bwd_kernel(const float* __restrict__ dy, int N, int H, int W, int C, float* __restrict__ dx) {
  const int C4 = C / 4, Ho = 2 * H, Wo = 2 * W;
  const int strips = (H + kUpStrip - 1) / kUpStrip;
  const long total = (long)N * strips * W * C4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    long t = i / C4;
    const int wi = t % W; t /= W;
    const int sidx = t % strips; const int n = t / strips;
    const int h0 = sidx * kUpStrip;
    int ow[4]; float ww[4];
    const int nw = up_taps(wi, W, ow, ww);
    float4 acc[kUpStrip];
#pragma unroll
    for (int k = 0; k < kUpStrip; ++k) acc[k] = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int rl = 0; rl < 2 * kUpStrip + 2; ++rl) {               // output rows 2*h0 - 1 ... 2*h0 + 2*kUpStrip
      const int r = 2 * h0 - 1 + rl;
      if (r < 0 || r >= Ho) continue;
      // output row r feeds input row r/2 (weight 0.75, or 1 at the clamped image border) and its neighbour (0.25)
      const int ia = r >> 1, ib = (r & 1) ? ia + 1 : ia - 1;
      const int ka = ia - h0, kb = ib - h0;                         // compile-time after unrolling: (rl-1)/2 etc.
      const bool use_a = ka >= 0 && ka < kUpStrip && ia < H;
      const bool use_b = kb >= 0 && kb < kUpStrip && ib >= 0 && ib < H;
      if (!use_a && !use_b) continue;
      float4 tr = make_float4(0, 0, 0, 0);
      const float* row = dy + (((size_t)n * Ho + r) * Wo) * C + c;
      for (int b = 0; b < nw; ++b) {
        const float4 g = *reinterpret_cast<const float4*>(row + (size_t)ow[b] * C);
        tr.x = fmaf(ww[b], g.x, tr.x); tr.y = fmaf(ww[b], g.y, tr.y); tr.z = fmaf(ww[b], g.z, tr.z); tr.w = fmaf(ww[b], g.w, tr.w);
      }
      if (use_a) {
        const float wa = ((r & 1) ? (ia == H - 1) : (ia == 0)) ? 1.f : 0.75f;
        const int k = (rl - 1) >> 1;                                // == ka for rl >= 1; rl == 0 has ka = -1 (unused)
        if (rl >= 1) { acc[k].x = fmaf(wa, tr.x, acc[k].x); acc[k].y = fmaf(wa, tr.y, acc[k].y); acc[k].z = fmaf(wa, tr.z, acc[k].z); acc[k].w = fmaf(wa, tr.w, acc[k].w); }
      }
      if (use_b) {
        const int k = (rl & 1) ? ((rl - 1) >> 1) - 1 : (rl >> 1);   // == kb: odd rl -> r even -> ia-1 ; even rl -> r odd -> ia+1
        if (k >= 0 && k < kUpStrip) { acc[k].x = fmaf(0.25f, tr.x, acc[k].x); acc[k].y = fmaf(0.25f, tr.y, acc[k].y); acc[k].z = fmaf(0.25f, tr.z, acc[k].z); acc[k].w = fmaf(0.25f, tr.w, acc[k].w); }
      }
    }
#pragma unroll
    for (int k = 0; k < kUpStrip; ++k)
      if (h0 + k < H) *reinterpret_cast<float4*>(dx + ((((size_t)n * H + h0 + k) * W + wi) * C4) * 4 + c) = acc[k];
  }
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

extern "C" size_t xas_bn_workspace_floats(long M, int C) {
  ColGeom g;
  if (col_geom(M, C, &g)) return 0;
  return (size_t)g.nslab * 2 * C + C;
}

extern "C" int xas_bn_stats(const float* x, long M, int C, float* mean, float* var_biased, float* workspace,
                            float* running_mean, float* running_var, float momentum, long count, void* stream) {
  ColGeom g;
  if (col_geom(M, C, &g)) return 1;
  XAS_REQUIRE(x && mean && var_biased && workspace, "bn_stats: null buffer");
  XAS_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_stats: running buffers come in pairs");
  const float unbias = count > 1 ? (float)((double)count / (double)(count - 1)) : 1.f;
  hipLaunchKernelGGL(col_reduce_kernel<0>, dim3(g.nslab, g.ncb), dim3(256), 0, as_stream(stream), x, nullptr, nullptr,
                     nullptr, nullptr, 0.f, 0, M, C, g, workspace);
  XAS_LAUNCH_CHECK();
  hipLaunchKernelGGL(col_finalize_kernel<0>, dim3((unsigned)cdiv(C, 64)), dim3(kFinalizeThreads), 0, as_stream(stream), workspace, x,
                     g.nslab, C, M, mean, var_biased, running_mean, running_var, momentum, unbias);
  XAS_LAUNCH_CHECK();
  return 0;
}

extern "C" int xas_col_sum(const float* x, long M, int C, float* out, float* workspace, void* stream) {
  ColGeom g;
  if (col_geom(M, C, &g)) return 1;
  XAS_REQUIRE(x && out && workspace, "col_sum: null buffer");
  hipLaunchKernelGGL(col_reduce_kernel<2>, dim3(g.nslab, g.ncb), dim3(256), 0, as_stream(stream), x, nullptr, nullptr,
                     nullptr, nullptr, 0.f, 0, M, C, g, workspace);
  XAS_LAUNCH_CHECK();
  // second output (unused sums of the second accumulator) lands in workspace tail
  hipLaunchKernelGGL(col_finalize_kernel<1>, dim3((unsigned)cdiv(C, 64)), dim3(kFinalizeThreads), 0, as_stream(stream), workspace, x,
                     g.nslab, C, M, out, workspace + (size_t)g.nslab * 2 * C, nullptr, nullptr, 0.f, 1.f);
  XAS_LAUNCH_CHECK();
  return 0;
}

extern "C" int xas_bn_apply(const float* x, const float* mean, const float* var_biased, const float* gamma,
                            const float* beta, const float* residual, float eps, int act, long M, int C, float* y,
                            void* stream) {
  XAS_REQUIRE(x && mean && var_biased && gamma && beta && y, "bn_apply: null buffer");
  XAS_REQUIRE(M > 0 && C >= 4 && C % 4 == 0 && act >= 0 && act <= 2, "bn_apply: bad shape M=%ld C=%d act=%d", M, C, act);
  const long n4 = M * (C / 4);
  hipLaunchKernelGGL(bn_apply_kernel, dim3(ew_grid(n4)), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const float4*>(x), mean, var_biased, gamma, beta,
                     reinterpret_cast<const float4*>(residual), eps, act, n4, C / 4, reinterpret_cast<float4*>(y));
  XAS_LAUNCH_CHECK();
  return 0;
}

extern "C" int xas_bn_update_running(const float* mean, const float* var_biased, float* running_mean,
                                     float* running_var, float momentum, long count, int C, void* stream) {
  XAS_REQUIRE(mean && var_biased && running_mean && running_var && C > 0 && count > 0, "bn_update_running: bad arguments");
  const float unbias = count > 1 ? (float)((double)count / (double)(count - 1)) : 1.f;
  hipLaunchKernelGGL(bn_running_kernel, dim3((unsigned)cdiv(C, 64)), dim3(64), 0, as_stream(stream), mean, var_biased,
                     running_mean, running_var, momentum, unbias, C);
  XAS_LAUNCH_CHECK();
  return 0;
}

extern "C" int xas_bn_bwd_reduce(const float* x, const float* y, const float* dy, const float* mean,
                                 const float* var_biased, const float* gamma, const float* beta, float eps, int act,
                                 long M, int C, float* sum_dz, float* sum_dz_xhat, float* workspace,
                                 float* dbeta_acc, float* dgamma_acc, void* stream) {
  ColGeom g;
  if (col_geom(M, C, &g)) return 1;
  XAS_REQUIRE(dy && mean && var_biased && sum_dz && sum_dz_xhat && workspace && (act == 0 || y || (x && gamma && beta)),
              "bn_bwd_reduce: null buffer (an activation needs y, or x with gamma and beta)");
  XAS_REQUIRE(x || (act != 0 && y && gamma && beta), "bn_bwd_reduce: without x the layer needs an activation, y, gamma, beta");
  XAS_REQUIRE((dbeta_acc == nullptr) == (dgamma_acc == nullptr), "bn_bwd_reduce: gradient accumulators come in pairs");
  const bool lean = (tune_flags() & 262144) == 0;      // shipped: the <= 64-VGPR build (tune bit18 selects the 86-VGPR one)
  if (x && act != 0 && y == nullptr) {   // y-free form: activation mask re-derived from x (layers without a residual)
    hipLaunchKernelGGL(lean ? col_reduce_lean_kernel<4> : col_reduce_kernel<4>, dim3(g.nslab, g.ncb), dim3(256), 0,
                       as_stream(stream), x, gamma, dy, mean, var_biased, eps, act, M, C, g, workspace, beta);
  } else
  if (x) {
    hipLaunchKernelGGL(lean ? col_reduce_lean_kernel<1> : col_reduce_kernel<1>, dim3(g.nslab, g.ncb), dim3(256), 0,
                       as_stream(stream), x, y, dy, mean, var_biased, eps, act, M, C, g, workspace, (const float*)nullptr);
  } else {            // x-free form: xhat recovered from the saved output (one activation tensor less to read)
    hipLaunchKernelGGL(lean ? col_reduce_lean_kernel<3> : col_reduce_kernel<3>, dim3(g.nslab, g.ncb), dim3(256), 0,
                       as_stream(stream), y, y, dy, beta, gamma, eps, act, M, C, g, workspace, (const float*)nullptr);
  }
  XAS_LAUNCH_CHECK();
  hipLaunchKernelGGL(col_finalize_kernel<1>, dim3((unsigned)cdiv(C, 64)), dim3(kFinalizeThreads), 0, as_stream(stream), workspace, x,
                     g.nslab, C, M, sum_dz, sum_dz_xhat, nullptr, nullptr, 0.f, 1.f, dbeta_acc, dgamma_acc);
  XAS_LAUNCH_CHECK();
  return 0;
}

extern "C" int xas_bn_bwd_apply(const float* x, const float* y, const float* dy, const float* mean,
                                const float* var_biased, const float* gamma, const float* beta, const float* sum_dz,
                                const float* sum_dz_xhat, float eps, int act, long M, int C, double count, float* dx,
                                float* dresidual, void* stream) {
  XAS_REQUIRE(dy && mean && var_biased && gamma && sum_dz && sum_dz_xhat && dx && (act == 0 || y || (x && beta)),
              "bn_bwd_apply: null buffer (an activation needs y, or x with beta)");
  XAS_REQUIRE(y || !dresidual || act == 0, "bn_bwd_apply: the y-free form is for layers without a residual");
  XAS_REQUIRE(x || (act == 2 && y && beta),
              "bn_bwd_apply: without x the layer needs an INVERTIBLE activation (leaky ReLU), y and beta: dx needs xhat "
              "of every element, also where ReLU clipped the output");
  XAS_REQUIRE(M > 0 && C >= 4 && C % 4 == 0 && count > 0, "bn_bwd_apply: bad shape");
  const long n4 = M * (C / 4);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_grid(n4)), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const float4*>(x), reinterpret_cast<const float4*>(y),
                     reinterpret_cast<const float4*>(dy), mean, var_biased, gamma, beta, sum_dz, sum_dz_xhat, eps, act, n4,
                     C / 4, (float)(1.0 / count), reinterpret_cast<float4*>(dx), reinterpret_cast<float4*>(dresidual));
  XAS_LAUNCH_CHECK();
  return 0;
}

extern "C" int xas_maxpool3x3s2_fwd(const float* x, int N, int H, int W, int C, float* y, int8_t* idx, void* stream) {
  XAS_REQUIRE(x && y && idx && N > 0 && H > 0 && W > 0 && C % 4 == 0, "maxpool: bad arguments");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const long total = (long)N * Ho * Wo * (C / 4);
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(ew_grid(total)), dim3(256), 0, as_stream(stream), x, N, H, W, C, Ho, Wo, y, idx);
  XAS_LAUNCH_CHECK();
  return 0;
}

extern "C" int xas_maxpool3x3s2_bwd(const float* dy, const int8_t* idx, int N, int H, int W, int C, float* dx,
                                    void* stream) {
  XAS_REQUIRE(dy && dx && idx && N > 0 && H > 0 && W > 0 && C % 4 == 0, "maxpool bwd: bad arguments");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const long total = (long)N * H * W * (C / 4);
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(ew_grid(total)), dim3(256), 0, as_stream(stream), dy, idx, N, H, W, C, Ho, Wo, dx);
  XAS_LAUNCH_CHECK();
  return 0;
}

extern "C" int xas_upsample2x_fwd(const float* x, int N, int H, int W, int C, float* y, void* stream) {
  XAS_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C % 4 == 0, "upsample2x: bad arguments");
  hipLaunchKernelGGL(upsample2x_fwd_kernel, dim3(ew_grid((long)N * 4 * H * W * (C / 4))), dim3(256), 0,
                     as_stream(stream), x, N, H, W, C, y);
  XAS_LAUNCH_CHECK();
  return 0;
}

extern "C" int xas_upsample2x_bwd(const float* dy, int N, int H, int W, int C, float* dx, void* stream) {
  XAS_REQUIRE(dy && dx && N > 0 && H > 0 && W > 0 && C % 4 == 0, "upsample2x bwd: bad arguments");
  hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3(ew_grid((long)N * ((H + kUpStrip - 1) / kUpStrip) * W * (C / 4))), dim3(256),
                     0, as_stream(stream), dy, N, H, W, C, dx);
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
