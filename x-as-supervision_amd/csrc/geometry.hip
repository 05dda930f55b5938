// Patch->world geometry and the fused line-mask renderer (max over lines).  gfx950.
//
// patch_to_world : modules/util.py:128-152 -> :61-95, all hypotheses in one launch, with
//                  closed-form 2x2 / 3x3 inverses instead of two batched LU solves.
// draw_lines_max : modules/util.py:21-59 + torch.max(dim=1) at modules/model.py:94,96.
//                  max_l exp(e_l) == exp(max_l e_l): one exp per pixel, nothing but the
//                  [B,1,S,S] mask ever reaches HBM (the reference keeps ten 419 MB
//                  intermediates for autograd).  Backward recomputes the winner line.
#include "common.h"

namespace xas {

struct CamParams {
  float i00, i01, i10, i11, t0, t1;   // inverse crop affine, translation
  float fx, fy, cx, cy, pz;           // intrinsics, pelvis depth
  float r[9];                         // R^-1 row-major
  float tw[3];
};

__device__ __forceinline__ CamParams load_cam(const float* ti, const float* km, const float* pv, const float* rw,
                                              const float* tw, int b, bool full = true) {
  CamParams c;
  const float* A = ti + b * 6;
  const float a00 = A[0], a01 = A[1], a10 = A[3], a11 = A[4];
  const float det = a00 * a11 - a01 * a10;
  c.i00 = a11 / det; c.i01 = -a01 / det; c.i10 = -a10 / det; c.i11 = a00 / det;
  c.t0 = A[2]; c.t1 = A[5];
  c.pz = pv[b * 3 + 2];
  if (!full) return c;                               // XAS_GEO_IMAGE: intrinsics / extrinsics are not needed (may be NULL)
  const float* Kc = km + b * 9;
  c.fx = Kc[0]; c.fy = Kc[4]; c.cx = Kc[2]; c.cy = Kc[5];
  const float* R = rw + b * 9;
  const float m00 = R[4] * R[8] - R[5] * R[7], m01 = R[2] * R[7] - R[1] * R[8], m02 = R[1] * R[5] - R[2] * R[4];
  const float m10 = R[5] * R[6] - R[3] * R[8], m11 = R[0] * R[8] - R[2] * R[6], m12 = R[2] * R[3] - R[0] * R[5];
  const float m20 = R[3] * R[7] - R[4] * R[6], m21 = R[1] * R[6] - R[0] * R[7], m22 = R[0] * R[4] - R[1] * R[3];
  const float d3 = R[0] * m00 + R[1] * m10 + R[2] * m20;
  c.r[0] = m00 / d3; c.r[1] = m01 / d3; c.r[2] = m02 / d3;
  c.r[3] = m10 / d3; c.r[4] = m11 / d3; c.r[5] = m12 / d3;
  c.r[6] = m20 / d3; c.r[7] = m21 / d3; c.r[8] = m22 / d3;
  c.tw[0] = tw[b * 3]; c.tw[1] = tw[b * 3 + 1]; c.tw[2] = tw[b * 3 + 2];
  return c;
}

#ifdef XAS_P2W_DEBUG
// diagnosis build (tools/diag_repro.py --bisect, profiles/r05_step_reproducibility.md): every launch records, per point, what the
// lane LOADED and what it COMPUTED, with the XCD it ran on: [32 launch slots][4096 points][32 floats]
__device__ float* g_p2w_dbg = nullptr;
static int g_p2w_launch = 0;
#define P2W_DBG_ARG , int dbg_slot
#else
#define P2W_DBG_ARG
#endif

__global__ void patch_to_world_fwd_kernel(const float* __restrict__ kps, const float* ti, const float* km,
                                          const float* pv, const float* rw, const float* tw, int B, int HK,
                                          float S, float rect, int flags, float* __restrict__ world P2W_DBG_ARG) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * HK) return;
  const int b = i / HK;
  const CamParams c = load_cam(ti, km, pv, rw, tw, b, !(flags & XAS_GEO_IMAGE));
  float p0 = kps[i * 3], p1 = kps[i * 3 + 1], p2 = kps[i * 3 + 2];
#ifdef XAS_P2W_DEBUG
  const float in0 = p0, in1 = p1, in2 = p2;
#endif
  if (flags & XAS_GEO_PATCH) {
    if (flags & XAS_GEO_NORM) {
      p0 = (p0 + 1.f) / 2.f * (S - 1.f);
      p1 = (p1 + 1.f) / 2.f * (S - 1.f);
      p2 = p2 * (S - 1.f);
    }
    const float du = p0 - c.t0, dv = p1 - c.t1;
    p0 = c.i00 * du + c.i01 * dv;
    p1 = c.i10 * du + c.i11 * dv;
    p2 = p2 * (1.0f / S * rect) + c.pz;
  }
  if (flags & XAS_GEO_IMAGE) {                       // patch -> image only (triangulation input, util.py:180-182)
    world[i * 3] = p0; world[i * 3 + 1] = p1; world[i * 3 + 2] = p2;
    return;
  }
  float o0, o1, o2;
#ifdef XAS_P2W_DEBUG
  float dq0 = 0.f, dq1 = 0.f, dq2 = 0.f;
#endif
  if (flags & XAS_GEO_MONO) {
    o0 = -p0; o1 = -(p2 + 128.f); o2 = -p1;
  } else {
    const float q0 = (p0 - c.cx) / c.fx * p2 - c.tw[0];
    const float q1 = (p1 - c.cy) / c.fy * p2 - c.tw[1];
    const float q2 = p2 - c.tw[2];
#ifdef XAS_P2W_DEBUG
    dq0 = q0; dq1 = q1; dq2 = q2;
#endif
    o0 = c.r[0] * q0 + c.r[1] * q1 + c.r[2] * q2;
    o1 = c.r[3] * q0 + c.r[4] * q1 + c.r[5] * q2;
    o2 = c.r[6] * q0 + c.r[7] * q1 + c.r[8] * q2;
  }
  world[i * 3] = o0; world[i * 3 + 1] = o1; world[i * 3 + 2] = o2;
#ifdef XAS_P2W_DEBUG
  if (g_p2w_dbg && i < 4096) {
    float* d = g_p2w_dbg + ((size_t)(dbg_slot & 31) * 4096 + i) * 32;
    unsigned xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    d[0] = in0; d[1] = in1; d[2] = in2; d[3] = o0; d[4] = o1; d[5] = o2; d[6] = (float)(xcc & 15); d[7] = c.pz;
    d[8] = p0; d[9] = p1; d[10] = p2; d[11] = dq0; d[12] = dq1; d[13] = dq2;      // after the patch stage; after the back projection
    d[14] = c.fx; d[15] = c.fy; d[16] = c.cx; d[17] = c.cy;
    for (int e = 0; e < 9; ++e) d[18 + e] = c.r[e];
    d[27] = c.tw[0]; d[28] = c.tw[1]; d[29] = c.tw[2];
    d[30] = __uint_as_float(hwid); d[31] = (float)b;
  }
#endif
}

__global__ void patch_to_world_bwd_kernel(const float* __restrict__ kps, const float* __restrict__ gw, const float* ti,
                                          const float* km, const float* pv, const float* rw, const float* tw, int B,
                                          int HK, float S, float rect, int flags, float* __restrict__ gk) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * HK) return;
  const int b = i / HK;
  const CamParams c = load_cam(ti, km, pv, rw, tw, b);
  float p0 = kps[i * 3], p1 = kps[i * 3 + 1], p2 = kps[i * 3 + 2];
  if (flags & XAS_GEO_PATCH) {
    if (flags & XAS_GEO_NORM) {
      p0 = (p0 + 1.f) / 2.f * (S - 1.f);
      p1 = (p1 + 1.f) / 2.f * (S - 1.f);
      p2 = p2 * (S - 1.f);
    }
    const float du = p0 - c.t0, dv = p1 - c.t1;
    p0 = c.i00 * du + c.i01 * dv;
    p1 = c.i10 * du + c.i11 * dv;
    p2 = p2 * (1.0f / S * rect) + c.pz;
  }
  const float g0 = gw[i * 3], g1 = gw[i * 3 + 1], g2 = gw[i * 3 + 2];
  float d0, d1, d2;      // gradient w.r.t. (p0,p1,p2) after the patch stage
  if (flags & XAS_GEO_MONO) {
    d0 = -g0; d2 = -g1; d1 = -g2;
  } else {
    const float q0 = c.r[0] * g0 + c.r[3] * g1 + c.r[6] * g2;
    const float q1 = c.r[1] * g0 + c.r[4] * g1 + c.r[7] * g2;
    const float q2 = c.r[2] * g0 + c.r[5] * g1 + c.r[8] * g2;
    d0 = q0 * p2 / c.fx;
    d1 = q1 * p2 / c.fy;
    d2 = q0 * (p0 - c.cx) / c.fx + q1 * (p1 - c.cy) / c.fy + q2;
  }
  if (flags & XAS_GEO_PATCH) {
    const float e0 = c.i00 * d0 + c.i10 * d1;
    const float e1 = c.i01 * d0 + c.i11 * d1;
    d0 = e0; d1 = e1;
    d2 = d2 * (1.0f / S * rect);
    if (flags & XAS_GEO_NORM) {
      d0 *= (S - 1.f) / 2.f; d1 *= (S - 1.f) / 2.f; d2 *= (S - 1.f);
    }
  }
  gk[i * 3] = d0; gk[i * 3 + 1] = d1; gk[i * 3 + 2] = d2;
}

// ------------------------------------------------------------------ line renderer
constexpr int kMaxLines = 32;
constexpr int kLineThreads = 256;
constexpr int kPixPerThread = 4;

struct LineSeg { float ax, ay, bx, by, vx, vy, den, scale; };

__device__ __forceinline__ void load_lines(LineSeg* segs, const float* kps, long sb, long sj, int b,
                                           const int* parents, const int* children, int L, unsigned fine,
                                           float body_width) {
  if (threadIdx.x < L) {
    const int l = threadIdx.x;
    const float* a = kps + b * sb + children[l] * sj;   // start = child, end = parent (util.py:35-36)
    const float* e = kps + b * sb + parents[l] * sj;
    LineSeg s;
    s.ax = a[0]; s.ay = a[1]; s.bx = e[0]; s.by = e[1];
    s.vx = s.bx - s.ax; s.vy = s.by - s.ay;
    s.den = 1e-8f + (s.vx * s.vx + s.vy * s.vy);
    s.scale = ((fine >> l) & 1u) ? 2.f : 1.f;
    segs[l] = s;
  }
  __syncthreads();
}

__device__ __forceinline__ float seg_exponent(const LineSeg& s, float gx, float gy, float body_width, float* t_out) {
  const float dax = gx - s.ax, day = gy - s.ay;
  const float t = (dax * s.vx + day * s.vy) / s.den;
  float d2;
  if (t <= 0.f) {
    d2 = dax * dax + day * day;
  } else if (t >= 1.f) {
    const float dbx = gx - s.bx, dby = gy - s.by;
    d2 = dbx * dbx + dby * dby;
  } else {
    const float rx = gx - (s.ax + t * s.vx), ry = gy - (s.ay + t * s.vy);
    d2 = rx * rx + ry * ry;
  }
  *t_out = t;
  return (-d2 / body_width) * s.scale;
}

__global__ void draw_lines_max_fwd_kernel(const float* __restrict__ kps, long sb, long sj, const int* parents,
                                          const int* children, int L, unsigned fine, float body_width, int S,
                                          float* __restrict__ mask) {
  __shared__ LineSeg segs[kMaxLines];
  const int b = blockIdx.y;
  load_lines(segs, kps, sb, sj, b, parents, children, L, fine, body_width);
  const int pix0 = (blockIdx.x * kLineThreads + threadIdx.x) * kPixPerThread;
  if (pix0 >= S * S) return;
  const float fs = (float)(S - 1);
  float out[kPixPerThread];
#pragma unroll
  for (int q = 0; q < kPixPerThread; ++q) {
    const int pix = pix0 + q;
    const float gx = 2.f * ((float)(pix % S) / fs) - 1.f, gy = 2.f * ((float)(pix / S) / fs) - 1.f;
    float best = -INFINITY, t;
    for (int l = 0; l < L; ++l) best = fmaxf(best, seg_exponent(segs[l], gx, gy, body_width, &t));
    out[q] = __expf(best);
  }
  float* o = mask + (size_t)b * S * S + pix0;
  if (pix0 + kPixPerThread <= S * S && ((S * S) % kPixPerThread) == 0) {
    *reinterpret_cast<float4*>(o) = make_float4(out[0], out[1], out[2], out[3]);
  } else {
    for (int q = 0; q < kPixPerThread && pix0 + q < S * S; ++q) o[q] = out[q];
  }
}

// Backward of the max-over-lines mask.  Every pixel sends a gradient to the two end points of its winning line.
// The per-block sums are formed in a FIXED order (per-line scan of the block's pixel records by one wave, butterfly
// wave sum, then lines -> joints in line order), never with floating-point atomics, so the detector gradient is
// reproducible run to run.
__global__ void draw_lines_max_bwd_kernel(const float* __restrict__ kps, long sb, long sj, const int* parents,
                                          const int* children, int L, unsigned fine, float body_width, int S, int K,
                                          const float* __restrict__ gmask, float* __restrict__ partial) {
  constexpr int kRec = kLineThreads * kPixPerThread;
  __shared__ LineSeg segs[kMaxLines];
  __shared__ int rec_line[kRec];
  __shared__ float rec_g[4][kRec];
  __shared__ float line_sum[kMaxLines][4];
  const int b = blockIdx.y;
  load_lines(segs, kps, sb, sj, b, parents, children, L, fine, body_width);
  const int pix0 = (blockIdx.x * kLineThreads + threadIdx.x) * kPixPerThread;
  const float fs = (float)(S - 1);
#pragma unroll
  for (int q = 0; q < kPixPerThread; ++q) {
    const int pix = pix0 + q;
    const int e = q * kLineThreads + threadIdx.x;
    int bl = -1;
    float gax = 0.f, gay = 0.f, gbx = 0.f, gby = 0.f;
    if (pix < S * S) {
      const float go = gmask[(size_t)b * S * S + pix];
      const float gx = 2.f * ((float)(pix % S) / fs) - 1.f, gy = 2.f * ((float)(pix / S) / fs) - 1.f;
      float best = -INFINITY, bt = 0.f, t;
      int win = 0;
      for (int l = 0; l < L; ++l) {
        const float ex = seg_exponent(segs[l], gx, gy, body_width, &t);
        if (ex > best) { best = ex; win = l; bt = t; }     // first maximum wins, as torch.max
      }
      const float ge = go * __expf(best);
      if (ge != 0.f) {
        bl = win;
        const LineSeg s = segs[win];
        const float gd2 = -ge * s.scale / body_width;
        if (bt <= 0.f) {
          gax = -2.f * (gx - s.ax) * gd2; gay = -2.f * (gy - s.ay) * gd2;
        } else if (bt >= 1.f) {
          gbx = -2.f * (gx - s.bx) * gd2; gby = -2.f * (gy - s.by) * gd2;
        } else {
          const float rx = gx - (s.ax + bt * s.vx), ry = gy - (s.ay + bt * s.vy);
          const float grx = 2.f * rx * gd2, gry = 2.f * ry * gd2;
          const float gt = -(grx * s.vx + gry * s.vy);
          const float gnum = gt / s.den, gden = -gt * bt / s.den;
          const float gvx = -bt * grx + (gx - s.ax) * gnum + 2.f * s.vx * gden;
          const float gvy = -bt * gry + (gy - s.ay) * gnum + 2.f * s.vy * gden;
          gax = -grx - s.vx * gnum - gvx; gay = -gry - s.vy * gnum - gvy;
          gbx = gvx; gby = gvy;
        }
      }
    }
    rec_line[e] = bl;
    rec_g[0][e] = gax; rec_g[1][e] = gay; rec_g[2][e] = gbx; rec_g[3][e] = gby;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int l = wave; l < L; l += kLineThreads / 64) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (int e = lane; e < kRec; e += 64) {
      if (rec_line[e] == l) { s0 += rec_g[0][e]; s1 += rec_g[1][e]; s2 += rec_g[2][e]; s3 += rec_g[3][e]; }
    }
    s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2); s3 = wave_sum(s3);
    if (lane == 0) { line_sum[l][0] = s0; line_sum[l][1] = s1; line_sum[l][2] = s2; line_sum[l][3] = s3; }
  }
  __syncthreads();
  float* o = partial + ((size_t)b * gridDim.x + blockIdx.x) * K * 2;
  for (int i = threadIdx.x; i < K * 2; i += blockDim.x) {
    const int j = i >> 1, c = i & 1;
    float s = 0.f;
    for (int l = 0; l < L; ++l) {
      if (children[l] == j) s += line_sum[l][c];
      if (parents[l] == j) s += line_sum[l][2 + c];
    }
    o[i] = s;
  }
}

__global__ void lines_bwd_reduce_kernel(const float* __restrict__ partial, int nblk, int K2, float* __restrict__ out) {
  const int b = blockIdx.x;
  for (int i = threadIdx.x; i < K2; i += blockDim.x) {
    float s = 0.f;
    for (int j = 0; j < nblk; ++j) s += partial[((size_t)b * nblk + j) * K2 + i];
    out[(size_t)b * K2 + i] = s;
  }
}

}  // namespace xas

using namespace xas;

extern "C" int xas_patch_to_world_fwd(const float* kps, const float* trans_image, const float* k_mat,
                                      const float* pelvis, const float* rot_world, const float* trans_world, int B,
                                      int Hy, int K, float image_size, float rect_width, int flags, float* world,
                                      void* stream) {
  XAS_REQUIRE(kps && world && trans_image && pelvis, "patch_to_world: null buffer");
  XAS_REQUIRE((flags & XAS_GEO_IMAGE) || (k_mat && rot_world && trans_world), "patch_to_world: null camera buffer");
  XAS_REQUIRE(B > 0 && Hy > 0 && K > 0, "patch_to_world: bad shape B=%d Hy=%d K=%d", B, Hy, K);
  const int n = B * Hy * K;
#ifdef XAS_P2W_DEBUG
  hipLaunchKernelGGL(patch_to_world_fwd_kernel, dim3(cdiv(n, 128)), dim3(128), 0, as_stream(stream), kps, trans_image,
                     k_mat, pelvis, rot_world, trans_world, B, Hy * K, image_size, rect_width, flags, world, g_p2w_launch++);
#else
  hipLaunchKernelGGL(patch_to_world_fwd_kernel, dim3(cdiv(n, 128)), dim3(128), 0, as_stream(stream), kps, trans_image,
                     k_mat, pelvis, rot_world, trans_world, B, Hy * K, image_size, rect_width, flags, world);
#endif
  XAS_LAUNCH_CHECK();
  return 0;
}

#ifdef XAS_P2W_DEBUG
// (diagnosis build only) buf: 32 * 4096 * 32 device floats, or NULL to stop recording; resets the launch counter
extern "C" int xas_debug_p2w(float* buf) {
  g_p2w_launch = 0;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_p2w_dbg), &buf, sizeof(buf)) == hipSuccess ? 0 : 1;
}
#endif

extern "C" int xas_patch_to_world_bwd(const float* kps, const float* grad_world, const float* trans_image,
                                      const float* k_mat, const float* pelvis, const float* rot_world,
                                      const float* trans_world, int B, int Hy, int K, float image_size,
                                      float rect_width, int flags, float* grad_kps, void* stream) {
  XAS_REQUIRE(kps && grad_world && grad_kps && trans_image && k_mat && pelvis && rot_world && trans_world,
              "patch_to_world bwd: null buffer");
  XAS_REQUIRE(B > 0 && Hy > 0 && K > 0, "patch_to_world bwd: bad shape");
  XAS_REQUIRE(!(flags & XAS_GEO_IMAGE), "patch_to_world bwd: XAS_GEO_IMAGE is a forward-only (evaluation) mode");
  const int n = B * Hy * K;
  hipLaunchKernelGGL(patch_to_world_bwd_kernel, dim3(cdiv(n, 128)), dim3(128), 0, as_stream(stream), kps, grad_world,
                     trans_image, k_mat, pelvis, rot_world, trans_world, B, Hy * K, image_size, rect_width, flags,
                     grad_kps);
  XAS_LAUNCH_CHECK();
  return 0;
}

extern "C" int xas_lines_nblk(int S) { return (int)cdiv((long)S * S, kLineThreads * kPixPerThread); }

extern "C" int xas_draw_lines_max_fwd(const float* kps, long kp_stride_b, long kp_stride_j, int B, int K,
                                      const int* parents, const int* children, int L, unsigned fine_mask,
                                      float body_width, int S, float* mask, void* stream) {
  XAS_REQUIRE(kps && parents && children && mask, "draw_lines: null buffer");
  XAS_REQUIRE(L >= 1 && L <= kMaxLines && K >= 1 && K <= 64 && S >= 2 && B >= 1, "draw_lines: bad shape L=%d K=%d S=%d", L, K, S);
  XAS_REQUIRE(body_width > 0.f, "draw_lines: body_width must be > 0");
  hipLaunchKernelGGL(draw_lines_max_fwd_kernel, dim3(xas_lines_nblk(S), B), dim3(kLineThreads), 0, as_stream(stream),
                     kps, kp_stride_b, kp_stride_j, parents, children, L, fine_mask, body_width, S, mask);
  XAS_LAUNCH_CHECK();
  return 0;
}

extern "C" int xas_draw_lines_max_bwd(const float* kps, long kp_stride_b, long kp_stride_j, int B, int K,
                                      const int* parents, const int* children, int L, unsigned fine_mask,
                                      float body_width, int S, const float* grad_mask, float* partial,
                                      float* grad_kps_xy, void* stream) {
  XAS_REQUIRE(kps && parents && children && grad_mask && partial && grad_kps_xy, "draw_lines bwd: null buffer");
  XAS_REQUIRE(L >= 1 && L <= kMaxLines && K >= 1 && K <= 64 && S >= 2 && B >= 1, "draw_lines bwd: bad shape");
  const int nblk = xas_lines_nblk(S);
  hipLaunchKernelGGL(draw_lines_max_bwd_kernel, dim3(nblk, B), dim3(kLineThreads), 0, as_stream(stream), kps,
                     kp_stride_b, kp_stride_j, parents, children, L, fine_mask, body_width, S, K, grad_mask, partial);
  XAS_LAUNCH_CHECK();
  hipLaunchKernelGGL(lines_bwd_reduce_kernel, dim3(B), dim3(64), 0, as_stream(stream), partial, nblk, K * 2,
                     grad_kps_xy);
  XAS_LAUNCH_CHECK();
  return 0;
}
