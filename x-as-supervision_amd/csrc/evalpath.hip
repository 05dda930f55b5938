// Evaluation path on the device (gfx950): hypothesis selection, multi-view DLT triangulation and the pose
// metrics.  All of it is tiny per-joint work (B*K ~ 600 points); the point of these kernels is that a whole
// evaluation batch stays on the GPU with ONE host synchronisation, instead of the reference's per-hypothesis
// tensor chains, a batched fp32 SVD and three numpy round trips per prediction set.
//
//   eval_select     : eval.py:117-148 + eval_utils.py:7-43 (switch_points, best/confident, per_act_mse)
//   projection      : modules/util.py:188 (P = K [R | t])
//   triangulate_dlt : modules/util.py:198-230 (null vector of the weighted DLT system)
//   pose_metrics    : metrics.py:5-244 (MPJPE under none / scale / procrustes alignment, 3DPCK, AUC hit counts)
//
// Small symmetric eigenproblems (4x4) are solved with cyclic Jacobi rotations in double precision, fully unrolled
// so that the matrices live in registers: the DLT null vector is the eigenvector of A^T A with the smallest
// eigenvalue, and the Procrustes rotation is Horn's unit quaternion, the eigenvector of a 4x4 matrix built from
// the 3x3 correlation with the LARGEST eigenvalue (always a proper rotation: the same matrix as the reference's
// SVD solution with its det(R) = +1 fix, metrics.py:41-49).
#include "common.h"

namespace xas {

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Cyclic Jacobi on a symmetric 4x4 (upper part used).  On return a[i][i] are the eigenvalues and the COLUMNS of v
// the eigenvectors.  12 sweeps: convergence is quadratic, 6-8 are enough for double precision.
__device__ __forceinline__ void jacobi4(double (&a)[4][4], double (&v)[4][4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) v[i][j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 12; ++sweep) {
    double off = 0.0, diag = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      diag += a[i][i] * a[i][i];
#pragma unroll
      for (int j = i + 1; j < 4; ++j) off += a[i][j] * a[i][j];
    }
    if (off <= 1e-60 * diag || off == 0.0) break;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
      for (int q = p + 1; q < 4; ++q) {
        const double apq = a[p][q];
        if (apq != 0.0) {
          const double theta = (a[q][q] - a[p][p]) / (2.0 * apq);
          const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
          const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
          a[p][p] -= t * apq;
          a[q][q] += t * apq;
          a[p][q] = 0.0; a[q][p] = 0.0;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (r != p && r != q) {
              const double arp = a[r][p], arq = a[r][q];
              a[r][p] = c * arp - s * arq; a[p][r] = a[r][p];
              a[r][q] = s * arp + c * arq; a[q][r] = a[r][q];
            }
            const double vrp = v[r][p], vrq = v[r][q];
            v[r][p] = c * vrp - s * vrq;
            v[r][q] = s * vrp + c * vrq;
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// eval_select: one block (one wave) per sample, lane = joint.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void eval_select_kernel(const float* __restrict__ kps, const float* __restrict__ joints,
                                                         const int* __restrict__ perm, int Hy, int K, int C, float S,
                                                         int flags, float* __restrict__ sel3d,
                                                         float* __restrict__ sel2d, float* __restrict__ err2d,
                                                         unsigned char* __restrict__ swapped) {
  const int b = blockIdx.x, k = threadIdx.x;
  const bool on = k < K;
  float g[3] = {0.f, 0.f, 0.f};
  if (on) {
    for (int c = 0; c < C; ++c) g[c] = joints[((size_t)b * K + k) * C + c];
    if (!(flags & XAS_EVAL_GT_NORMALISED)) {                       // eval.py:124-125
      g[0] = g[0] / (S - 1.f) * 2.f - 1.f;
      g[1] = g[1] / (S - 1.f) * 2.f - 1.f;
      g[2] = g[2] / (S - 1.f);
    }
  }
  const int kp = on ? perm[k] : 0;
  float best3[3] = {0.f, 0.f, 0.f}, best2[2] = {0.f, 0.f};
  float e3min = INFINITY, e2min = INFINITY;
  bool took = false;
  for (int h = 0; h < Hy; ++h) {
    float p[3] = {0.f, 0.f, 0.f}, m[3] = {0.f, 0.f, 0.f};
    if (on) {
      const float* src = kps + (((size_t)b * Hy + h) * K) * C;
      for (int c = 0; c < C; ++c) { p[c] = src[k * C + c]; m[c] = src[kp * C + c]; }
    }
    // eval_utils.py:16-27: L1 error over the first two coordinates, mirrored joint kept when strictly smaller
    float e_own = fabsf(p[0] - g[0]) + fabsf(p[1] - g[1]);
    float e_mir = fabsf(m[0] - g[0]) + fabsf(m[1] - g[1]);
    if (flags & XAS_EVAL_SWITCH_ALL) {                              // one decision per sample (sum over joints too)
      e_own = wave_sum(on ? e_own : 0.f);
      e_mir = wave_sum(on ? e_mir : 0.f);
    }
    took = e_mir < e_own;
    if (took) { p[0] = m[0]; p[1] = m[1]; p[2] = m[2]; }
    const float d0 = p[0] - g[0], d1 = p[1] - g[1], d2 = p[2] - g[2];
    const float e2 = __fadd_rn(__fmul_rn(d0, d0), __fmul_rn(d1, d1));
    const float e3 = __fadd_rn(e2, __fmul_rn(d2, d2));
    const bool first = (flags & XAS_EVAL_CONFIDENT) ? (h == 0) : false;
    const bool better3 = (flags & XAS_EVAL_CONFIDENT) ? first : (e3 < e3min);       // argmin: first minimum wins
    const bool better2 = (flags & XAS_EVAL_CONFIDENT) ? first : (e2 < e2min);
    if (better3) { e3min = e3; best3[0] = p[0]; best3[1] = p[1]; best3[2] = p[2]; }
    if (better2) { e2min = e2; best2[0] = p[0]; best2[1] = p[1]; }
  }
  if (on) {
    if (sel3d) for (int c = 0; c < C; ++c) sel3d[((size_t)b * K + k) * C + c] = best3[c];
    if (sel2d) { sel2d[((size_t)b * K + k) * 2] = best2[0]; sel2d[((size_t)b * K + k) * 2 + 1] = best2[1]; }
    if (swapped) swapped[(size_t)b * K + k] = took ? 1 : 0;          // the reference keeps the LAST hypothesis' flags
  }
  if (err2d) {                                                       // eval_utils.py:32-43
    const float a0 = (best2[0] + 1.f) / 2.f - (g[0] + 1.f) / 2.f, a1 = (best2[1] + 1.f) / 2.f - (g[1] + 1.f) / 2.f;
    const float d = on ? sqrtf(__fadd_rn(__fmul_rn(a0, a0), __fmul_rn(a1, a1))) : 0.f;
    const float s = wave_sum(d);
    if (k == 0) err2d[b] = s / (float)K;
  }
}

// ------------------------------------------------------------------------------------------------------
// projection matrices: P[b] = K[b] [R[b] | t[b]]   (3x4)
// ------------------------------------------------------------------------------------------------------
__global__ void projection_kernel(const float* __restrict__ km, const float* __restrict__ rw, const float* __restrict__ tw,
                                  int B, float* __restrict__ P) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * 12) return;
  const int b = i / 12, r = (i % 12) / 4, c = i % 4;
  const float* Kc = km + (size_t)b * 9;
  float acc = 0.f;
  for (int j = 0; j < 3; ++j) {
    const float rt = c < 3 ? rw[(size_t)b * 9 + j * 3 + c] : tw[(size_t)b * 3 + j];
    acc += Kc[r * 3 + j] * rt;
  }
  P[i] = acc;
}

// ------------------------------------------------------------------------------------------------------
// triangulate_dlt: thread per (b, joint).  points [B][V][K][3] = (u, v, weight), P [B][V][3][4].
// Rows are formed in fp32 exactly as the reference forms them (util.py:215-221), then A^T A and its smallest
// eigenvector are computed in double.
// ------------------------------------------------------------------------------------------------------
__global__ void triangulate_kernel(const float* __restrict__ pts, const float* __restrict__ P, int B, int V, int K,
                                   float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * K) return;
  const int b = i / K, k = i % K;
  double g[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) g[r][c] = 0.0;
  float wsum = 0.f;
  int wpos = 0;
  for (int v = 0; v < V; ++v) {
    const float* q = pts + (((size_t)b * V + v) * K + k) * 3;
    const float* Pm = P + ((size_t)b * V + v) * 12;
    const float u = q[0], w = q[1], c = q[2];
    wsum += c;
    wpos += c > 0.f ? 1 : 0;
    float au[4], av[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      au[j] = __fmul_rn(c, __fsub_rn(__fmul_rn(u, Pm[8 + j]), Pm[j]));
      av[j] = __fmul_rn(c, __fsub_rn(__fmul_rn(w, Pm[8 + j]), Pm[4 + j]));
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int cc = r; cc < 4; ++cc) g[r][cc] += (double)au[r] * (double)au[cc] + (double)av[r] * (double)av[cc];
  }
#pragma unroll
  for (int r = 1; r < 4; ++r)
#pragma unroll
    for (int cc = 0; cc < r; ++cc) g[r][cc] = g[cc][r];
  double vec[4][4];
  jacobi4(g, vec);
  int best = 0;
  double lo = g[0][0];
#pragma unroll
  for (int j = 1; j < 4; ++j)
    if (g[j][j] < lo) { lo = g[j][j]; best = j; }
  double X[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) X[r] = best == 0 ? vec[r][0] : (best == 1 ? vec[r][1] : (best == 2 ? vec[r][2] : vec[r][3]));
  float* o = out + (size_t)i * 4;
  o[0] = (float)(X[0] / X[3]); o[1] = (float)(X[1] / X[3]); o[2] = (float)(X[2] / X[3]);
  o[3] = wsum / (float)wpos;                                          // util.py:207-209 (mean weight)
}

// ------------------------------------------------------------------------------------------------------
// pose_metrics: one wave per sample, lane = joint.
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float norm3(float a, float b, float c) {
  return sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(a, a), __fmul_rn(b, b)), __fmul_rn(c, c)));
}

__global__ __launch_bounds__(64) void pose_metrics_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                          const unsigned char* __restrict__ mask, int K, float in_div,
                                                          int pck_align, float pck_threshold, float* __restrict__ err,
                                                          float* __restrict__ aligned, float* __restrict__ pck,
                                                          int* __restrict__ auc_hits, long NK) {
  const int n = blockIdx.x, k = threadIdx.x;
  const bool on = k < K;
  float p[3] = {0.f, 0.f, 0.f}, g[3] = {0.f, 0.f, 0.f};
  if (on) {
    for (int c = 0; c < 3; ++c) {
      p[c] = pred[((size_t)n * K + k) * 3 + c] / in_div;
      g[c] = gt[((size_t)n * K + k) * 3 + c] / in_div;
    }
  }
  const float vis = on ? (mask ? (mask[(size_t)n * K + k] ? 1.f : 0.f) : 1.f) : 0.f;
  float e[3];
  e[0] = norm3(p[0] - g[0], p[1] - g[1], p[2] - g[2]);
  // --- scale alignment (metrics.py:106-110): all joints, visible or not
  const double pp = wave_sum_d(on ? (double)p[0] * p[0] + (double)p[1] * p[1] + (double)p[2] * p[2] : 0.0);
  const double pg = wave_sum_d(on ? (double)p[0] * g[0] + (double)p[1] * g[1] + (double)p[2] * g[2] : 0.0);
  const float f = (float)(pg / pp);
  float a1[3] = {p[0] * f, p[1] * f, p[2] * f};
  e[1] = norm3(a1[0] - g[0], a1[1] - g[1], a1[2] - g[2]);
  // --- procrustes (metrics.py:5-62)
  const double invK = 1.0 / (double)K;
  double mu1[3], mu2[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    mu1[c] = wave_sum_d(on ? (double)p[c] : 0.0) * invK;
    mu2[c] = wave_sum_d(on ? (double)g[c] : 0.0) * invK;
  }
  double x1[3], x2[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) { x1[c] = on ? p[c] - mu1[c] : 0.0; x2[c] = on ? g[c] - mu2[c] : 0.0; }
  const double var1 = wave_sum_d(x1[0] * x1[0] + x1[1] * x1[1] + x1[2] * x1[2]);
  double S[3][3];                                                     // S = sum src dst^T  (K of metrics.py:38)
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) S[r][c] = wave_sum_d(x1[r] * x2[c]);
  double N4[4][4], Q[4][4];
  N4[0][0] = S[0][0] + S[1][1] + S[2][2];
  N4[0][1] = S[1][2] - S[2][1]; N4[0][2] = S[2][0] - S[0][2]; N4[0][3] = S[0][1] - S[1][0];
  N4[1][1] = S[0][0] - S[1][1] - S[2][2]; N4[1][2] = S[0][1] + S[1][0]; N4[1][3] = S[2][0] + S[0][2];
  N4[2][2] = -S[0][0] + S[1][1] - S[2][2]; N4[2][3] = S[1][2] + S[2][1];
  N4[3][3] = -S[0][0] - S[1][1] + S[2][2];
#pragma unroll
  for (int r = 1; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < r; ++c) N4[r][c] = N4[c][r];
  jacobi4(N4, Q);                                                     // every lane solves the same 4x4 (no broadcast needed)
  int best = 0;
  double hi = N4[0][0];
#pragma unroll
  for (int j = 1; j < 4; ++j)
    if (N4[j][j] > hi) { hi = N4[j][j]; best = j; }
  double q[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) q[r] = best == 0 ? Q[r][0] : (best == 1 ? Q[r][1] : (best == 2 ? Q[r][2] : Q[r][3]));
  const double qn = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const double w = q[0] / qn, x = q[1] / qn, y = q[2] / qn, z = q[3] / qn;
  double R[3][3] = {{1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)},
                    {2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)},
                    {2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)}};
  double trRK = 0.0;                                                  // trace(R K) = sum_ij R[i][j] S[j][i]
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) trRK += R[r][c] * S[c][r];
  const double scale = trRK / var1;
  float a2[3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
    a2[r] = (float)(scale * (R[r][0] * x1[0] + R[r][1] * x1[1] + R[r][2] * x1[2]) + mu2[r]);
  e[2] = norm3(a2[0] - g[0], a2[1] - g[1], a2[2] - g[2]);

  if (!on) return;
  const size_t o = (size_t)n * K + k;
  if (err) { err[o] = e[0] * vis; err[NK + o] = e[1] * vis; err[2 * NK + o] = e[2] * vis; }
  if (aligned) {
    for (int c = 0; c < 3; ++c) { aligned[o * 3 + c] = a1[c]; aligned[(NK + o) * 3 + c] = a2[c]; }
  }
  const float ep = pck_align == 0 ? e[0] : (pck_align == 1 ? e[1] : e[2]);
  if (pck) pck[o] = (ep < pck_threshold ? 100.f : 0.f) * vis;           // metrics.py:172-173
  if (auc_hits) {                                                       // metrics.py:236-240, np.linspace(0, 0.15, 31)
    const double step = 0.15 / 30.0;
    for (int t = 0; t < 31; ++t) {
      const double thr = t == 30 ? 0.15 : t * step;
      const int hit = ((double)ep < thr && vis != 0.f) ? 1 : 0;
      const unsigned long long bal = __ballot(hit);
      if (k == 0) auc_hits[(size_t)n * 31 + t] = __popcll(bal);
    }
  }
}

}  // namespace xas

using namespace xas;

extern "C" int xas_eval_select(const float* kps, const float* joints, const int* perm, int B, int Hy, int K, int C,
                               float image_size, int flags, float* sel3d, float* sel2d, float* err2d,
                               unsigned char* swapped, void* stream) {
  XAS_REQUIRE(kps && joints && perm, "eval_select: null buffer");
  XAS_REQUIRE(B > 0 && Hy > 0 && K > 0 && K <= 64 && (C == 2 || C == 3), "eval_select: bad shape B=%d Hy=%d K=%d C=%d (K <= 64, C in {2,3})", B, Hy, K, C);
  XAS_REQUIRE(image_size > 1.f, "eval_select: image_size %f", (double)image_size);
  hipLaunchKernelGGL(eval_select_kernel, dim3(B), dim3(64), 0, as_stream(stream), kps, joints, perm, Hy, K, C, image_size,
                     flags, sel3d, sel2d, err2d, swapped);
  XAS_LAUNCH_CHECK();
  return 0;
}

extern "C" int xas_projection_matrix(const float* k_mat, const float* rot_world, const float* trans_world, int B, float* P,
                                     void* stream) {
  XAS_REQUIRE(k_mat && rot_world && trans_world && P && B > 0, "projection_matrix: bad arguments");
  hipLaunchKernelGGL(projection_kernel, dim3((unsigned)cdiv((long)B * 12, 128)), dim3(128), 0, as_stream(stream), k_mat,
                     rot_world, trans_world, B, P);
  XAS_LAUNCH_CHECK();
  return 0;
}

extern "C" int xas_triangulate_dlt(const float* points, const float* P, int B, int V, int K, float* out, void* stream) {
  XAS_REQUIRE(points && P && out, "triangulate_dlt: null buffer");
  XAS_REQUIRE(B > 0 && V >= 2 && K > 0, "triangulate_dlt: bad shape B=%d V=%d K=%d (needs >= 2 views)", B, V, K);
  hipLaunchKernelGGL(triangulate_kernel, dim3((unsigned)cdiv((long)B * K, 64)), dim3(64), 0, as_stream(stream), points, P, B,
                     V, K, out);
  XAS_LAUNCH_CHECK();
  return 0;
}

extern "C" int xas_pose_metrics(const float* pred, const float* gt, const unsigned char* mask, int N, int K, float in_div,
                                int pck_align, float pck_threshold, float* err, float* aligned, float* pck,
                                int* auc_hits, void* stream) {
  XAS_REQUIRE(pred && gt, "pose_metrics: null buffer");
  XAS_REQUIRE(N > 0 && K >= 3 && K <= 64, "pose_metrics: bad shape N=%d K=%d (3 <= K <= 64)", N, K);
  XAS_REQUIRE(in_div > 0.f && pck_align >= 0 && pck_align <= 2, "pose_metrics: bad in_div / pck_align");
  hipLaunchKernelGGL(pose_metrics_kernel, dim3(N), dim3(64), 0, as_stream(stream), pred, gt, mask, K, in_div, pck_align,
                     pck_threshold, err, aligned, pck, auc_hits, (long)N * K);
  XAS_LAUNCH_CHECK();
  return 0;
}
