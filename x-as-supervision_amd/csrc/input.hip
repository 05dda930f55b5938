// GPU input pipeline: what the reference's CPU loader does per sample and camera with OpenCV and
// scikit-fmm (human_utils/dataloader/dataloader.py:17-91,150-191, common/imglib/affine.py:56-114,
// common/utility/geodesic.py:14-54), as HIP kernels over a whole batch of decoded images resident in HBM:
//
//   warp_affine_u8      cv2.warpAffine(img, trans, (P, P), INTER_LINEAR), BORDER_CONSTANT 0, for 8-bit images: OpenCV's
//                       fixed-point scheme restated exactly (10-bit coordinates, 5-bit sub-pixel position, 15-bit weights,
//                       round to nearest) - integer arithmetic, bit-exact against the oracle (oracle/input_pipeline.py)
//   mask_blur_threshold cv2.GaussianBlur(mask, (5,5), 0) + threshold(127) of the MPI-INF-3DHP masks (dataloader.py:62-65)
//   patch_finish        BGR->RGB, HWC->CHW, colour scale + clip, (x - mean) / std, mask / 255, image * mask (rm_bg)
//   geodesic_weight     centroid, two grid Eikonal solves (inside the mask from the centroid; outside from the mask),
//                       normalisation and combination of geodesic.py:42-52
//
// HBM-bound byte work except the Eikonal solve, which is latency-bound (one workgroup per image iterating the upwind
// update to its fixed point).  gfx950.
#include "common.h"

namespace xas {

constexpr int kInterBits = 5, kInterTab = 1 << kInterBits;           // INTER_BITS, INTER_TAB_SIZE
constexpr int kAbBits = 10, kAbScale = 1 << kAbBits;                 // AB_BITS, AB_SCALE
constexpr int kCoefBits = 15;                                        // INTER_REMAP_COEF_BITS

struct WarpGeom { int B, C, P; };

// minv: [B][6] doubles = the INVERTED 2x3 map (dst -> src), computed on the host exactly as cv::warpAffine does.
__global__ void warp_affine_u8_kernel(const uint8_t* __restrict__ src, const long* __restrict__ src_off,
                                      const int* __restrict__ src_hw, const double* __restrict__ minv, WarpGeom g,
                                      uint8_t* __restrict__ dst) {
  const int b = blockIdx.y;
  const int H = src_hw[2 * b], W = src_hw[2 * b + 1];
  const uint8_t* S = src + src_off[b];
  const double* M = minv + 6 * b;
  const int round_delta = kAbScale / kInterTab / 2;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < g.P * g.P; i += gridDim.x * blockDim.x) {
    const int y = i / g.P, x = i - y * g.P;
    // cv::warpAffine: integer coordinates with AB_BITS fractional bits, rounded (half to even) per term
    const int adelta = __double2int_rn(M[0] * x * kAbScale), bdelta = __double2int_rn(M[3] * x * kAbScale);
    const int X0 = __double2int_rn((M[1] * y + M[2]) * kAbScale) + round_delta;
    const int Y0 = __double2int_rn((M[4] * y + M[5]) * kAbScale) + round_delta;
    const int X = (X0 + adelta) >> (kAbBits - kInterBits), Y = (Y0 + bdelta) >> (kAbBits - kInterBits);
    int sx = X >> kInterBits, sy = Y >> kInterBits;
    sx = max(-32768, min(32767, sx)); sy = max(-32768, min(32767, sy));       // saturate_cast<short>
    const int fx = X & (kInterTab - 1), fy = Y & (kInterTab - 1);
    // bilinear weights of the 32 x 32 table, scaled by 2^15: (32-fx)(32-fy)*32 ... exact integers summing to 32768
    const int w00 = (kInterTab - fx) * (kInterTab - fy) * 32, w01 = fx * (kInterTab - fy) * 32;
    const int w10 = (kInterTab - fx) * fy * 32, w11 = fx * fy * 32;
    uint8_t* o = dst + ((size_t)b * g.P * g.P + i) * g.C;
    const bool in00 = (unsigned)sx < (unsigned)W && (unsigned)sy < (unsigned)H;
    const bool in01 = (unsigned)(sx + 1) < (unsigned)W && (unsigned)sy < (unsigned)H;
    const bool in10 = (unsigned)sx < (unsigned)W && (unsigned)(sy + 1) < (unsigned)H;
    const bool in11 = (unsigned)(sx + 1) < (unsigned)W && (unsigned)(sy + 1) < (unsigned)H;
    const size_t p00 = ((size_t)sy * W + sx) * g.C;
    for (int c = 0; c < g.C; ++c) {
      const int v00 = in00 ? S[p00 + c] : 0, v01 = in01 ? S[p00 + g.C + c] : 0;
      const int v10 = in10 ? S[p00 + (size_t)W * g.C + c] : 0, v11 = in11 ? S[p00 + (size_t)W * g.C + g.C + c] : 0;
      o[c] = (uint8_t)((v00 * w00 + v01 * w01 + v10 * w10 + v11 * w11 + (1 << (kCoefBits - 1))) >> kCoefBits);
    }
  }
}

// 5x5 Gaussian ([1 4 6 4 1] x [1 4 6 4 1] / 256, BORDER_REFLECT_101, rounded half up as OpenCV's 8-bit fixed-point path)
// followed by threshold(127) -> {0, 255}
__global__ void mask_blur_threshold_kernel(const uint8_t* __restrict__ m, int B, int P, uint8_t* __restrict__ out) {
  const int b = blockIdx.y;
  const uint8_t* S = m + (size_t)b * P * P;
  const int wk[5] = {1, 4, 6, 4, 1};
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P * P; i += gridDim.x * blockDim.x) {
    const int y = i / P, x = i - y * P;
    int s = 0;
#pragma unroll
    for (int dy = -2; dy <= 2; ++dy) {
      int yy = y + dy; yy = yy < 0 ? -yy : (yy >= P ? 2 * P - 2 - yy : yy);
#pragma unroll
      for (int dx = -2; dx <= 2; ++dx) {
        int xx = x + dx; xx = xx < 0 ? -xx : (xx >= P ? 2 * P - 2 - xx : xx);
        s += wk[dy + 2] * wk[dx + 2] * S[yy * P + xx];
      }
    }
    const int v = (s + 128) >> 8;
    out[(size_t)b * P * P + i] = v > 127 ? 255 : 0;
  }
}

// img_patch [B][P][P][3] BGR u8, mask_patch [B][P][P] u8 -> img [B][3][P][P] RGB float, mask [B][1][P][P] float
__global__ void patch_finish_kernel(const uint8_t* __restrict__ img, const uint8_t* __restrict__ mask,
                                    const float* __restrict__ color_scale /*[B][3] RGB order or null*/, float3 mean,
                                    float3 stdv, int rm_bg, int B, int P, float* __restrict__ out_img,
                                    float* __restrict__ out_mask) {
  const int b = blockIdx.y;
  const size_t plane = (size_t)P * P;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P * P; i += gridDim.x * blockDim.x) {
    const uint8_t* p = img + ((size_t)b * plane + i) * 3;
    const float mk = (float)mask[(size_t)b * plane + i] / 255.0f;
    const float mean3[3] = {mean.x, mean.y, mean.z}, std3[3] = {stdv.x, stdv.y, stdv.z};
#pragma unroll
    for (int c = 0; c < 3; ++c) {                     // output channel c = RGB index; source is BGR
      float v = (float)p[2 - c];
      if (color_scale) v = fminf(fmaxf(v * color_scale[b * 3 + c], 0.f), 255.f);
      v = (v - mean3[c]) / std3[c];
      if (rm_bg) v *= mk;
      out_img[((size_t)b * 3 + c) * plane + i] = v;
    }
    out_mask[(size_t)b * plane + i] = mk;
  }
}

// ---------------------------------------------------------------- geodesic weight map
// One workgroup per image.  Phase 1: centroid (geodesic.py:4-12, truncated to integers) or the given centre, early-out
// flag (a centre outside the mask -> map of ones, geodesic.py:25-27).  Phase 2: two upwind Eikonal solves on the pixel grid -
// scikit-fmm's default second-order scheme (order 2, r05) or first order - iterated to their fixed point (the solution the
// fast-marching method computes for the same stencil):
//   d_in : domain = mask pixels, source = centre pixel          (skfmm.distance of the masked array, geodesic.py:29-35)
//   d_bg : domain = all pixels, sources = mask pixels (value 0) (skfmm.distance(m_bg), geodesic.py:37-39)
// Phase 3: out = exp(p0 * d_in / max d_in) + p1 + p2 * d_bg / max d_bg + p3 (geodesic.py:45-52).
constexpr int kGeoThreads = 1024;
constexpr float kInf = 3.0e38f;

__device__ __forceinline__ float eikonal_update(float a, float b) {     // a, b: smaller neighbour per axis (kInf if none)
  const float lo = fminf(a, b), hi = fmaxf(a, b);
  if (lo >= kInf) return kInf;
  if (hi - lo >= 1.0f) return lo + 1.0f;
  return 0.5f * (a + b + sqrtf(2.0f - (a - b) * (a - b)));
}

// u: distances (sources 0, others kInf on entry); dom: 1 = pixel takes part
__device__ void eikonal_solve(float* __restrict__ u, const uint8_t* __restrict__ dom, int P, int max_sweeps) {
  const int n = P * P;
  for (int sweep = 0; sweep < max_sweeps; ++sweep) {
    int changed = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      if (!dom[i]) continue;
      const float cur = u[i];
      if (cur == 0.f) continue;
      const int y = i / P, x = i - y * P;
      const float l = (x > 0 && dom[i - 1]) ? u[i - 1] : kInf, r = (x < P - 1 && dom[i + 1]) ? u[i + 1] : kInf;
      const float t = (y > 0 && dom[i - P]) ? u[i - P] : kInf, d = (y < P - 1 && dom[i + P]) ? u[i + P] : kInf;
      const float nu = eikonal_update(fminf(l, r), fminf(t, d));
      if (nu < cur) { u[i] = nu; changed = 1; }
    }
    if (!__syncthreads_or(changed)) break;            // barrier: this workgroup's global writes are visible to its waves
  }
}

// ---- second order (what scikit-fmm's `distance` runs by default, order = 2).  Per axis: the smaller neighbour v1; when the
// next pixel in the same direction is not larger (in fast marching: frozen before v1), the one-sided second-order difference
// (3u - 4 v1 + v2) / 2 - the term 9/4 (u - t)^2 with t = v1 + (v1 - v2) / 3 - else the first-order term (u - v1)^2; the terms sum
// to 1.  Solved relative to the smaller t (the textbook form b^2 - 4ac cancels catastrophically in fp32 at distances of
// hundreds of pixels).  A neighbour larger than the result is not upwind and is dropped (fast marching: not frozen yet).
// The scheme is not monotone in its neighbours (a smaller v2 RAISES t), so the sweeps ASSIGN the value instead of keeping the
// minimum, until a sweep changes nothing: the values settle in causal order (every pixel depends on strictly smaller ones).
// A neighbour counts as upwind only if it is smaller than the result by this margin: pixels of (nearly) EQUAL distance - the whole
// first ring around a flat source region - must not take each other as upwind neighbours (fast marching breaks such ties by its
// heap order; parallel sweeps would chase each other an ulp at a time).  The result is continuous across the switch (at
// v = single-axis value both forms agree), so the margin moves values by O(1e-4) pixels.
constexpr float kGeoCausal = 1e-4f;

__device__ __forceinline__ float eikonal2_value(const float* __restrict__ u, const uint8_t* __restrict__ dom, int i, int P) {
  const int y = i / P, x = i - y * P;
  float t[2], w[2], v[2];
#pragma unroll
  for (int ax = 0; ax < 2; ++ax) {
    const int step = ax ? P : 1, pos = ax ? y : x;
    float v1 = kInf, v2 = kInf;
#pragma unroll
    for (int j = -1; j <= 1; j += 2) {
      if ((unsigned)(pos + j) >= (unsigned)P) continue;
      const int n1 = i + j * step;
      if (dom && !dom[n1]) continue;
      const float a = u[n1];
      if (a < v1) {
        v1 = a; v2 = kInf;
        if ((unsigned)(pos + 2 * j) < (unsigned)P) {
          const int n2 = i + 2 * j * step;
          if (!dom || dom[n2]) { const float b = u[n2]; if (b <= a) v2 = b; }
        }
      }
    }
    v[ax] = v1;
    if (v2 < kInf) { t[ax] = v1 + (v1 - v2) * (1.f / 3.f); w[ax] = 2.25f; }
    else { t[ax] = v1; w[ax] = 1.f; }
  }
  const int lo = t[1] < t[0] ? 1 : 0, hi = 1 - lo;              // (kInf on an axis without a reached neighbour)
  if (v[lo] >= kInf) return kInf;
  const float single = t[lo] + (w[lo] > 1.f ? (2.f / 3.f) : 1.f);
  if (v[hi] >= kInf) return single;
  const float d = t[hi] - t[lo], sw = w[lo] + w[hi];
  const float det = sw - w[lo] * w[hi] * d * d;
  if (det < 0.f) return single;
  const float r = t[lo] + (w[hi] * d + sqrtf(det)) / sw;
  return r - kGeoCausal >= v[hi] ? r : single;
}

// u: sources 0, others kInf on entry; dom: 1 = pixel takes part (nullptr: all); tmp: a second field.  JACOBI sweeps (every
// pixel from the PREVIOUS field): the result does not depend on the order in which the waves run - with in-place sweeps a
// comparison of two nearly equal neighbours could go either way from run to run, and with it a stencil.
__device__ void eikonal2_solve(float* __restrict__ u, float* __restrict__ tmp, const uint8_t* __restrict__ dom, int P, int max_sweeps) {
  const int n = P * P;
  float* src = u;
  float* dst = tmp;
  for (int sweep = 0; sweep < max_sweeps; ++sweep) {
    int changed = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      const float cur = src[i];
      float nu = cur;
      if ((!dom || dom[i]) && cur != 0.f) nu = eikonal2_value(src, dom, i, P);
      dst[i] = nu;
      changed |= nu != cur;
    }
    float* t = src; src = dst; dst = t;                 // (uniform)
    if (!__syncthreads_or(changed)) break;
  }
  if (src != u) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) u[i] = src[i];
    __syncthreads();
  }
}

__device__ float block_max(float v, float* red) {      // red: >= 33 floats; result valid in every thread
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x < 64) {
    float r = threadIdx.x < (blockDim.x >> 6) ? red[threadIdx.x] : -kInf;
    r = wave_max(r);
    if (threadIdx.x == 0) red[32] = r;
  }
  __syncthreads();
  return red[32];
}

__global__ __launch_bounds__(kGeoThreads) void geodesic_weight_kernel(const float* __restrict__ mask, const int* __restrict__ centers, int nc, int order,
                                                                      float p0, float p1, float p2, float p3, int P,
                                                                      float* __restrict__ work /*[B][3][P*P]*/,
                                                                      uint8_t* __restrict__ dom /*[B][P*P]*/,
                                                                      float* __restrict__ out, int* __restrict__ center_out) {
  __shared__ float red[33];
  __shared__ double sred[3][16];
  __shared__ int s_c[3];
  const int b = blockIdx.x, n = P * P;
  const float* m = mask + (size_t)b * n;
  float* din = work + (size_t)b * 3 * n;
  float* dbg = din + n;
  float* tmp = dbg + n;                                // second field of the Jacobi sweeps (order 2)
  uint8_t* dm = dom + (size_t)b * n;
  float* o = out + (size_t)b * n;
  // centroid of the BOOLEAN mask (np.bool_(img): any non-zero value is foreground), truncated to int16 (geodesic.py:4-12,15)
  double sx = 0.0, sy = 0.0, sm = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const double v = m[i] != 0.f ? 1.0 : 0.0;
    const int y = i / P, x = i - y * P;
    sx += (double)x * v; sy += (double)y * v; sm += v;
    dm[i] = v != 0.0;
  }
  for (int off = 32; off > 0; off >>= 1) { sx += __shfl_xor(sx, off, 64); sy += __shfl_xor(sy, off, 64); sm += __shfl_xor(sm, off, 64); }
  if ((threadIdx.x & 63) == 0) { sred[0][threadIdx.x >> 6] = sx; sred[1][threadIdx.x >> 6] = sy; sred[2][threadIdx.x >> 6] = sm; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0, c = 0, d = 0;
    for (int k = 0; k < (int)(blockDim.x >> 6); ++k) { a += sred[0][k]; c += sred[1][k]; d += sred[2][k]; }
    int cx, cy;
    if (centers) { cx = centers[2 * (size_t)b * nc]; cy = centers[2 * (size_t)b * nc + 1]; }
    else { cx = (int)(a / d); cy = (int)(c / d); }                 // astype(np.int16): truncation (NaN for an empty mask -> 0)
    if (!(d > 0.0) && !centers) { cx = 0; cy = 0; }
    s_c[0] = cx; s_c[1] = cy;
    if (center_out) { center_out[2 * b] = cx; center_out[2 * b + 1] = cy; }
    // geodesic.py:22-27: EVERY source must lie on the mask, else the map is all ones (several sources: geodesic_pt_list joints)
    bool inside = true;
    const int ncs = centers ? nc : 1;
    for (int k = 0; k < ncs; ++k) {
      const int x = centers ? centers[2 * ((size_t)b * nc + k)] : cx, y = centers ? centers[2 * ((size_t)b * nc + k) + 1] : cy;
      inside = inside && (unsigned)x < (unsigned)P && (unsigned)y < (unsigned)P && m[y * P + x] != 0.f;
    }
    s_c[2] = inside ? 1 : 0;
  }
  __syncthreads();
  const int cx = s_c[0], cy = s_c[1];
  if (!s_c[2]) {                                       // geodesic.py:25-27: a source on the background -> weights of one
    for (int i = threadIdx.x; i < n; i += blockDim.x) o[i] = 1.0f;
    return;
  }
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    din[i] = (i == cy * P + cx) ? 0.f : kInf;
    dbg[i] = dm[i] ? 0.f : kInf;
  }
  __syncthreads();
  if (centers && nc > 1) {                             // the other sources (geodesic.py:29-31: every centre is a zero of the level set)
    for (int k = 1 + threadIdx.x; k < nc; k += blockDim.x) din[centers[2 * ((size_t)b * nc + k) + 1] * P + centers[2 * ((size_t)b * nc + k)]] = 0.f;
    __syncthreads();
  }
  if (order == 2) {
    eikonal2_solve(din, tmp, dm, P, 8 * P);
    eikonal2_solve(dbg, tmp, nullptr, P, 8 * P);
  } else {
    eikonal_solve(din, dm, P, 4 * P);
    // background solve: every pixel is in the domain
    for (int sweep = 0; sweep < 4 * P; ++sweep) {
      int changed = 0;
      for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float cur = dbg[i];
        if (cur == 0.f) continue;
        const int y = i / P, x = i - y * P;
        const float l = x > 0 ? dbg[i - 1] : kInf, r = x < P - 1 ? dbg[i + 1] : kInf;
        const float t = y > 0 ? dbg[i - P] : kInf, d = y < P - 1 ? dbg[i + P] : kInf;
        const float nu = eikonal_update(fminf(l, r), fminf(t, d));
        if (nu < cur) { dbg[i] = nu; changed = 1; }
      }
      if (!__syncthreads_or(changed)) break;
    }
  }
  // maxima (pixels outside the mask count as 0 in d_in, as in the array scikit-fmm returns for masked cells)
  float mi = 0.f, mb = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const float a = (dm[i] && din[i] < kInf) ? din[i] : 0.f;
    mi = fmaxf(mi, a);
    mb = fmaxf(mb, dbg[i] < kInf ? dbg[i] : 0.f);
  }
  mi = block_max(mi, red);
  mb = block_max(mb, red);
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const float a = (dm[i] && din[i] < kInf) ? din[i] : 0.f;
    const float g = dbg[i] < kInf ? dbg[i] : 0.f;
    o[i] = __expf(p0 * (a / mi)) + p1 + p2 * (g / mb) + p3;
  }
}

}  // namespace xas

using namespace xas;

extern "C" int xas_warp_affine_u8(const uint8_t* src, const long* src_off, const int* src_hw, const double* minv, int B,
                                  int C, int P, uint8_t* dst, void* stream) {
  XAS_REQUIRE(src && src_off && src_hw && minv && dst && B > 0 && C >= 1 && C <= 4 && P > 0 && P <= 4096,
              "warp_affine_u8: bad arguments (B=%d C=%d P=%d)", B, C, P);
  WarpGeom g{B, C, P};
  const unsigned bx = (unsigned)cdiv((long)P * P, 256);
  hipLaunchKernelGGL(warp_affine_u8_kernel, dim3(bx > 256 ? 256 : bx, B), dim3(256), 0, as_stream(stream), src, src_off,
                     src_hw, minv, g, dst);
  XAS_LAUNCH_CHECK();
  return 0;
}

extern "C" int xas_mask_blur_threshold(const uint8_t* mask, int B, int P, uint8_t* out, void* stream) {
  XAS_REQUIRE(mask && out && mask != out && B > 0 && P >= 3, "mask_blur_threshold: bad arguments");
  const unsigned bx = (unsigned)cdiv((long)P * P, 256);
  hipLaunchKernelGGL(mask_blur_threshold_kernel, dim3(bx > 256 ? 256 : bx, B), dim3(256), 0, as_stream(stream), mask, B, P, out);
  XAS_LAUNCH_CHECK();
  return 0;
}

extern "C" int xas_patch_finish(const uint8_t* img_bgr, const uint8_t* mask, const float* color_scale, const float* mean3,
                                const float* std3, int rm_bg, int B, int P, float* out_img, float* out_mask, void* stream) {
  XAS_REQUIRE(img_bgr && mask && mean3 && std3 && out_img && out_mask && B > 0 && P > 0, "patch_finish: bad arguments");
  XAS_REQUIRE(std3[0] != 0.f && std3[1] != 0.f && std3[2] != 0.f, "patch_finish: zero std");
  const unsigned bx = (unsigned)cdiv((long)P * P, 256);
  hipLaunchKernelGGL(patch_finish_kernel, dim3(bx > 256 ? 256 : bx, B), dim3(256), 0, as_stream(stream), img_bgr, mask,
                     color_scale, make_float3(mean3[0], mean3[1], mean3[2]), make_float3(std3[0], std3[1], std3[2]), rm_bg,
                     B, P, out_img, out_mask);
  XAS_LAUNCH_CHECK();
  return 0;
}

extern "C" size_t xas_geodesic_workspace_bytes(int B, int P) {
  if (B <= 0 || P <= 0) return 0;
  return (size_t)B * P * P * (3 * sizeof(float) + 1);
}

extern "C" int xas_geodesic_weight(const float* mask, const int* centers, const float* params5, int B, int P, float* out,
                                   int* center_out, void* workspace, void* stream) {
  XAS_REQUIRE(mask && params5 && out && workspace && B > 0 && P >= 2 && P <= 1024, "geodesic_weight: bad arguments");
  XAS_REQUIRE(params5[4] == 0.f, "geodesic_weight: geodesic_param_list[4] = %g: only the shipped 0.0 (mask = zero level) is built",
              (double)params5[4]);
  float* work = reinterpret_cast<float*>(workspace);
  uint8_t* dom = reinterpret_cast<uint8_t*>(work + (size_t)B * 3 * P * P);
  hipLaunchKernelGGL(geodesic_weight_kernel, dim3(B), dim3(kGeoThreads), 0, as_stream(stream), mask, centers, 1, 1, params5[0],
                     params5[1], params5[2], params5[3], P, work, dom, out, center_out);
  XAS_LAUNCH_CHECK();
  return 0;
}

extern "C" int xas_geodesic_weight_multi(const float* mask, const int* centers, int num_centers, int order, const float* params5,
                                         int B, int P, float* out, int* center_out, void* workspace, void* stream) {
  XAS_REQUIRE(mask && params5 && out && workspace && B > 0 && P >= 2 && P <= 1024 && num_centers >= 1 && num_centers <= 64 &&
              (centers || num_centers == 1) && (order == 1 || order == 2),
              "geodesic_weight_multi: bad arguments (B=%d P=%d centres=%d order=%d)", B, P, num_centers, order);
  XAS_REQUIRE(params5[4] == 0.f, "geodesic_weight_multi: geodesic_param_list[4] = %g: only the shipped 0.0 (mask = zero level) is built",
              (double)params5[4]);
  float* work = reinterpret_cast<float*>(workspace);
  uint8_t* dom = reinterpret_cast<uint8_t*>(work + (size_t)B * 3 * P * P);
  hipLaunchKernelGGL(geodesic_weight_kernel, dim3(B), dim3(kGeoThreads), 0, as_stream(stream), mask, centers, num_centers, order,
                     params5[0], params5[1], params5[2], params5[3], P, work, dom, out, center_out);
  XAS_LAUNCH_CHECK();
  return 0;
}
