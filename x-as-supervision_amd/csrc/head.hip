// Fused D x H x W soft-argmax ("integral") head, forward and backward.  gfx950.
//
// Replaces keypoint_detector_integral_multi.py:69-88 / keypoint_detector_integral.py:48-63.
// HBM-bound: the forward reads the logits exactly once (online softmax, all three
// marginals in one pass), the backward reads them once and writes the gradient once.
//
// Data layout: logits are NHWC, i.e. [B][H*W][K*D]: one pixel holds K*D contiguous
// floats.  A thread owns one float4 of channels (joint k, depth bins 4q..4q+3) and walks
// pixels, so every wave-instruction loads 1 KiB of contiguous memory.
//
// Pass 1 (head_partial): per (b, pixel chunk) -> per joint {max, sum e*w, sum e*h, pz[D]}.
// Pass 2 (head_finalize): one wave per (b,k): merge chunks, normalise, pick the depth
// peaks (value desc, index asc), windowed expectation, write kps / indices / stats.
#include "common.h"

namespace xas {

struct HeadGeom {
  int B, K, D, HW, W, C4, G, R, P, nchunk, rec;  // rec = 3 + D floats per (b,chunk,k)
  int wshift;                                     // log2(W): W is a power of two, so pix -> (h, w) is shift/mask
};

static int make_geom(int B, int K, int D, HeadGeom* g) {
  XAS_REQUIRE(B > 0 && K > 0 && D >= 4 && D <= 64 && (D & (D - 1)) == 0,
              "head: need power-of-two depth_dim in [4,64], got B=%d K=%d D=%d", B, K, D);
  g->B = B; g->K = K; g->D = D; g->W = D; g->HW = D * D;
  g->wshift = 0;
  while ((1 << g->wshift) < D) ++g->wshift;
  g->C4 = K * D / 4;
  g->G = D / 4;
  int gcd = 1;
  while (gcd < 64 && (g->C4 % (gcd * 2)) == 0) gcd *= 2;
  int rstep = 64 / gcd;                       // R must be a multiple of this
  int R = rstep;
  while ((long)g->C4 * R * 2 <= 640 && R * 2 <= g->HW) R *= 2;
  XAS_REQUIRE((long)g->C4 * R <= 1024, "head: K*D=%d too wide for one workgroup", K * D);
  g->R = R;
  g->P = R > 128 ? R : 128;       // pixels per workgroup: 128 amortises the block-level merge (64: 3.8 TB/s)
  if (g->P > g->HW) g->P = g->HW;
  g->P = (g->P / R) * R;
  XAS_REQUIRE(g->P >= R, "head: heat-map too small");
  g->nchunk = (g->HW + g->P - 1) / g->P;
  g->rec = 3 + D;
  return 0;
}

__global__ void head_partial_kernel(const float4* __restrict__ logits, float* __restrict__ partial, HeadGeom g) {
  extern __shared__ float smem[];            // [R][K][3 + D]
  const int tid = threadIdx.x;
  const int c4 = tid % g.C4, slot = tid / g.C4;
  const int k = c4 / g.G, dq = c4 % g.G;
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int pix0 = chunk * g.P;
  const int pend = min(g.P, g.HW - pix0);
  const float4* base = logits + ((size_t)b * g.HW + pix0) * g.C4 + c4;

  float m = -INFINITY, sx = 0.f, sy = 0.f, z0 = 0.f, z1 = 0.f, z2 = 0.f, z3 = 0.f;
  // four pixels per trip: one running-max update (one rescale exp) per 16 values keeps the loop-carried
  // dependency short; the 4 loads are issued back to back
  int p = slot;
  for (; p + 3 * g.R < pend; p += 4 * g.R) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = stream_load(base + (size_t)(p + u * g.R) * g.C4);
    float mn = m;
#pragma unroll
    for (int u = 0; u < 4; ++u) mn = fmaxf(mn, fmaxf(fmaxf(v[u].x, v[u].y), fmaxf(v[u].z, v[u].w)));
    const float sc = __expf(m - mn);          // exp(-inf) = 0 on the first trip
    m = mn;
    z0 *= sc; z1 *= sc; z2 *= sc; z3 *= sc; sx *= sc; sy *= sc;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int pix = pix0 + p + u * g.R;
      const float fw = (float)(pix & (g.W - 1)), fh = (float)(pix >> g.wshift);
      const float e0 = __expf(v[u].x - mn), e1 = __expf(v[u].y - mn), e2 = __expf(v[u].z - mn), e3 = __expf(v[u].w - mn);
      const float es = (e0 + e1) + (e2 + e3);
      z0 += e0; z1 += e1; z2 += e2; z3 += e3;
      sx = fmaf(es, fw, sx);
      sy = fmaf(es, fh, sy);
    }
  }
  for (; p < pend; p += g.R) {
    const float4 v = base[(size_t)p * g.C4];
    const int pix = pix0 + p;
    const float fw = (float)(pix & (g.W - 1)), fh = (float)(pix >> g.wshift);
    const float mn = fmaxf(fmaxf(m, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
    const float sc = __expf(m - mn);
    m = mn;
    const float e0 = __expf(v.x - mn), e1 = __expf(v.y - mn), e2 = __expf(v.z - mn), e3 = __expf(v.w - mn);
    const float es = (e0 + e1) + (e2 + e3);
    z0 = z0 * sc + e0; z1 = z1 * sc + e1; z2 = z2 * sc + e2; z3 = z3 * sc + e3;
    sx = sx * sc + es * fw;
    sy = sy * sc + es * fh;
  }
  // unify the running max over the G lanes of this (slot, joint), then sum sx / sy
  float gm = m;
  for (int o = g.G >> 1; o > 0; o >>= 1) gm = fmaxf(gm, __shfl_xor(gm, o, 64));
  const float sc = (m == -INFINITY) ? 0.f : __expf(m - gm);
  z0 *= sc; z1 *= sc; z2 *= sc; z3 *= sc; sx *= sc; sy *= sc;
  for (int o = g.G >> 1; o > 0; o >>= 1) { sx += __shfl_xor(sx, o, 64); sy += __shfl_xor(sy, o, 64); }
  float* rec = smem + ((size_t)slot * g.K + k) * g.rec;
  if (dq == 0) { rec[0] = gm; rec[1] = sx; rec[2] = sy; }
  rec[3 + 4 * dq + 0] = z0; rec[3 + 4 * dq + 1] = z1; rec[3 + 4 * dq + 2] = z2; rec[3 + 4 * dq + 3] = z3;
  __syncthreads();
  // merge the R slots; thread t handles entry t of the [K][rec] record table
  float* out = partial + ((size_t)b * g.nchunk + chunk) * g.K * g.rec;
  for (int e = tid; e < g.K * g.rec; e += blockDim.x) {
    const int kk = e / g.rec, f = e % g.rec;
    float M = -INFINITY;
    for (int s = 0; s < g.R; ++s) M = fmaxf(M, smem[((size_t)s * g.K + kk) * g.rec]);
    if (f == 0) { out[e] = M; continue; }
    float acc = 0.f;
    for (int s = 0; s < g.R; ++s) {
      const float* r = smem + ((size_t)s * g.K + kk) * g.rec;
      const float ms = r[0];
      acc += (ms == -INFINITY) ? 0.f : r[f] * __expf(ms - M);
    }
    out[e] = acc;
  }
}

__global__ void head_finalize_kernel(const float* __restrict__ partial, HeadGeom g, int num_hypo, int neighbor,
                                     float* __restrict__ kps, int64_t* __restrict__ z_idx,
                                     float* __restrict__ depth_prob_map, int dmap_every, float* __restrict__ stats) {
  const int b = blockIdx.x / g.K, k = blockIdx.x % g.K;
  const int d = threadIdx.x;                  // one wave; lane = depth bin
  const bool live = d < g.D;
  const float* p0 = partial + ((size_t)b * g.nchunk * g.K + k) * g.rec;
  const size_t cstride = (size_t)g.K * g.rec;
  float M = -INFINITY;
  for (int c = 0; c < g.nchunk; ++c) M = fmaxf(M, p0[c * cstride]);
  float sd = 0.f, SX = 0.f, SY = 0.f;
  for (int c = 0; c < g.nchunk; ++c) {
    const float* r = p0 + c * cstride;
    const float sc = __expf(r[0] - M);
    if (live) sd += r[3 + d] * sc;
    SX += r[1] * sc;
    SY += r[2] * sc;
  }
  const float S = wave_sum(live ? sd : 0.f);
  const float pz = live ? sd / S : 0.f;
  const float X = SX / S, Y = SY / S;
  float* st = stats + ((size_t)b * g.K + k) * XAS_HEAD_STATS;
  if (b % dmap_every == 0 && live) depth_prob_map[((size_t)(b / dmap_every) * g.K + k) * g.D + d] = pz;
  const float fD = (float)g.D;
  const float xn = X / fD * 2.f - 1.f, yn = Y / fD * 2.f - 1.f;
  if (d == 0) { st[0] = M + __logf(S); st[1] = X; st[2] = Y; }

  if (neighbor == 0) {                        // single hypothesis: plain expectation
    const float Z = wave_sum(pz * (float)d);
    if (d == 0) {
      float* o = kps + ((size_t)b * g.K + k) * 3;
      o[0] = xn; o[1] = yn; o[2] = Z / fD * 2.f - 1.f;
      st[3] = Z;
      if (z_idx) z_idx[(size_t)b * g.K + k] = 0;
    }
    return;
  }
  const float left = __shfl_up(pz, 1, 64), right = __shfl_down(pz, 1, 64);
  const bool inner = d >= 1 && d <= g.D - 2;
  float score = inner ? ((pz >= left && pz >= right) ? pz : 0.f) : -1.f;
  const int r = neighbor / 2;
  for (int h = 0; h < num_hypo; ++h) {
    const float best = wave_max(score);
    const unsigned long long cand = __ballot(score == best);
    const int idx = __ffsll((long long)cand) - 1;   // lowest bin among equal scores
    if (d == idx) score = -2.f;
    const bool inwin = live && (d >= idx - r) && (d <= idx + r);
    const float sw = wave_sum(inwin ? pz : 0.f);
    const float swd = wave_sum(inwin ? pz * (float)d : 0.f);
    const float Z = swd / sw;
    if (d == 0) {
      float* o = kps + (((size_t)b * num_hypo + h) * g.K + k) * 3;
      o[0] = xn; o[1] = yn; o[2] = Z / fD * 2.f - 1.f;
      z_idx[((size_t)b * g.K + k) * num_hypo + h] = idx;
      st[3 + h] = Z;
      st[9 + h] = sw;
    }
  }
}

// per (b,k): {cx, cy, c0, lse, gz[D]}
__global__ void head_bwd_coef_kernel(const float* __restrict__ stats, const int64_t* __restrict__ z_idx,
                                     const float* __restrict__ grad_kps, HeadGeom g, int num_hypo, int neighbor,
                                     float* __restrict__ coef) {
  const int b = blockIdx.x / g.K, k = blockIdx.x % g.K;
  const int d = threadIdx.x;
  const float* st = stats + ((size_t)b * g.K + k) * XAS_HEAD_STATS;
  const float fD = (float)g.D;
  float gx = 0.f, gy = 0.f, gz = 0.f, c0 = 0.f;
  const int r = neighbor / 2;
  for (int h = 0; h < num_hypo; ++h) {
    const float* gk = grad_kps + (((size_t)b * num_hypo + h) * g.K + k) * 3;
    gx += gk[0];
    gy += gk[1];
    if (neighbor == 0) {
      const float cz = gk[2] * (2.f / fD);
      gz += cz * (float)d;
      c0 -= cz * st[3];
    } else {
      const int idx = (int)z_idx[((size_t)b * g.K + k) * num_hypo + h];
      if (d >= idx - r && d <= idx + r) gz += gk[2] * (2.f / fD) * ((float)d - st[3 + h]) / st[9 + h];
    }
  }
  const float cx = gx * (2.f / fD), cy = gy * (2.f / fD);
  c0 += -cx * st[1] - cy * st[2];
  float* o = coef + ((size_t)b * g.K + k) * (4 + g.D);
  if (d == 0) { o[0] = cx; o[1] = cy; o[2] = c0; o[3] = st[0]; }
  if (d < g.D) o[4 + d] = gz;
}

__global__ void head_bwd_kernel(const float4* __restrict__ logits, const float* __restrict__ coef, HeadGeom g,
                                float4* __restrict__ grad, float* __restrict__ amax_out) {
  __shared__ unsigned s_amax;                          // amax_out != null: max |grad| of the block (bit pattern), then of the launch
  if (threadIdx.x == 0) s_amax = 0u;
  if (amax_out) __syncthreads();
  float amx = 0.f;
  const int tid = threadIdx.x;
  const int c4 = tid % g.C4, slot = tid / g.C4;
  const int k = c4 / g.G, dq = c4 % g.G;
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int pix0 = chunk * g.P;
  const int pend = min(g.P, g.HW - pix0);
  const float* cf = coef + ((size_t)b * g.K + k) * (4 + g.D);
  const float cx = cf[0], cy = cf[1], c0 = cf[2], lse = cf[3];
  const float4 gz = *reinterpret_cast<const float4*>(cf + 4 + 4 * dq);
  const size_t off = ((size_t)b * g.HW + pix0) * g.C4 + c4;
#pragma unroll 4
  for (int p = slot; p < pend; p += g.R) {
    const float4 v = stream_load(logits + off + (size_t)p * g.C4);
    const int pix = pix0 + p;
    const float lin = cx * (float)(pix & (g.W - 1)) + cy * (float)(pix >> g.wshift) + c0;
    float4 o;
    o.x = __expf(v.x - lse) * (lin + gz.x);
    o.y = __expf(v.y - lse) * (lin + gz.y);
    o.z = __expf(v.z - lse) * (lin + gz.z);
    o.w = __expf(v.w - lse) * (lin + gz.w);
    amx = fmaxf(fmaxf(amx, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
    stream_store(grad + off + (size_t)p * g.C4, o);
  }
  if (amax_out) {                                      // (the block is not a whole number of waves: reduce through LDS)
    if (amx > 0.f) atomicMax(&s_amax, __float_as_uint(amx));
    __syncthreads();
    // one atomic per block, to the sub-maximum the block index selects (common.h: recorded maxima)
    const unsigned sub = (blockIdx.x + blockIdx.y * gridDim.x) % (unsigned)kAmaxSub;
    if (threadIdx.x == 0 && s_amax) atomicMax(reinterpret_cast<unsigned*>(amax_out + sub * kAmaxStride), s_amax);
  }
}

}  // namespace xas

using namespace xas;

extern "C" size_t xas_head_workspace_floats(int B, int K, int D) {
  HeadGeom g;
  if (make_geom(B, K, D, &g)) return 0;
  return (size_t)B * g.nchunk * K * g.rec;
}

extern "C" int xas_head_softargmax_fwd(const float* logits, int B, int K, int D, int num_hypo, int neighbor,
                                       float* kps, int64_t* z_idx, float* depth_prob_map, int groups, float* stats,
                                       float* partial, void* stream) {
  HeadGeom g;
  if (make_geom(B, K, D, &g)) return 1;
  XAS_REQUIRE(logits && kps && depth_prob_map && stats && partial, "head fwd: null buffer");
  XAS_REQUIRE(num_hypo >= 1 && num_hypo <= 6, "head fwd: num_hypo %d not in [1,6]", num_hypo);
  XAS_REQUIRE(neighbor >= 0 && (neighbor > 0 || num_hypo == 1), "head fwd: single-hypothesis mode needs num_hypo == 1");
  XAS_REQUIRE(neighbor == 0 || (z_idx != nullptr && num_hypo <= D - 2), "head fwd: z_idx required / too many hypotheses");
  XAS_REQUIRE(((uintptr_t)logits & 15) == 0, "head fwd: logits must be 16-byte aligned");
  XAS_REQUIRE(groups >= 1 && B % groups == 0, "head fwd: B=%d does not split into %d groups", B, groups);
  const size_t lds = (size_t)g.R * K * g.rec * sizeof(float);
  XAS_REQUIRE(lds <= 64 * 1024, "head fwd: LDS %zu too large", lds);
  hipLaunchKernelGGL(head_partial_kernel, dim3(g.nchunk, B), dim3(g.C4 * g.R), lds, as_stream(stream),
                     reinterpret_cast<const float4*>(logits), partial, g);
  XAS_LAUNCH_CHECK();
  hipLaunchKernelGGL(head_finalize_kernel, dim3(B * K), dim3(64), 0, as_stream(stream), partial, g, num_hypo,
                     neighbor, kps, z_idx, depth_prob_map, B / groups, stats);
  XAS_LAUNCH_CHECK();
  return 0;
}

// The second pass alone, over partial records that something else produced: xas_conv_fwd_head (conv.hip) emits them from the
// final convolution's epilogue, `nchunk` records of 64 pixels per image in the layout of head_partial_kernel
// ([image][chunk][joint][3 + D]).  Everything downstream (kps, int64 peak indices, depth maps, statistics for the backward)
// is what xas_head_softargmax_fwd writes.
extern "C" int xas_head_softargmax_from_partials(const float* partial, int B, int K, int D, int nchunk, int num_hypo, int neighbor,
                                                 float* kps, int64_t* z_idx, float* depth_prob_map, int groups, float* stats,
                                                 void* stream) {
  HeadGeom g;
  if (make_geom(B, K, D, &g)) return 1;
  XAS_REQUIRE(partial && kps && depth_prob_map && stats && nchunk >= 1, "head from partials: null buffer");
  XAS_REQUIRE(num_hypo >= 1 && num_hypo <= 6, "head from partials: num_hypo %d not in [1,6]", num_hypo);
  XAS_REQUIRE(neighbor >= 0 && (neighbor > 0 || num_hypo == 1), "head from partials: single-hypothesis mode needs num_hypo == 1");
  XAS_REQUIRE(neighbor == 0 || (z_idx != nullptr && num_hypo <= D - 2), "head from partials: z_idx required / too many hypotheses");
  XAS_REQUIRE(groups >= 1 && B % groups == 0, "head from partials: B=%d does not split into %d groups", B, groups);
  g.nchunk = nchunk;
  hipLaunchKernelGGL(head_finalize_kernel, dim3(B * K), dim3(64), 0, as_stream(stream), partial, g, num_hypo,
                     neighbor, kps, z_idx, depth_prob_map, B / groups, stats);
  XAS_LAUNCH_CHECK();
  return 0;
}

extern "C" int xas_head_softargmax_bwd(const float* logits, const float* stats, const int64_t* z_idx,
                                       const float* grad_kps, int B, int K, int D, int num_hypo, int neighbor,
                                       float* grad_logits, float* coef, void* stream) {
  return xas_head_softargmax_bwd_amax(logits, stats, z_idx, grad_kps, B, K, D, num_hypo, neighbor, grad_logits, coef, nullptr, stream);
}

extern "C" int xas_head_softargmax_bwd_amax(const float* logits, const float* stats, const int64_t* z_idx,
                                            const float* grad_kps, int B, int K, int D, int num_hypo, int neighbor,
                                            float* grad_logits, float* coef, float* amax_out, void* stream) {
  HeadGeom g;
  if (make_geom(B, K, D, &g)) return 1;
  XAS_REQUIRE(logits && stats && grad_kps && grad_logits && coef, "head bwd: null buffer");
  XAS_REQUIRE(num_hypo >= 1 && num_hypo <= 6, "head bwd: num_hypo %d not in [1,6]", num_hypo);
  XAS_REQUIRE(neighbor == 0 || z_idx != nullptr, "head bwd: z_idx required");
  XAS_REQUIRE((((uintptr_t)logits | (uintptr_t)grad_logits | (uintptr_t)coef) & 15) == 0, "head bwd: 16-byte alignment");
  hipLaunchKernelGGL(head_bwd_coef_kernel, dim3(B * K), dim3(64), 0, as_stream(stream), stats, z_idx, grad_kps, g,
                     num_hypo, neighbor, coef);
  XAS_LAUNCH_CHECK();
  hipLaunchKernelGGL(head_bwd_kernel, dim3(g.nchunk, B), dim3(g.C4 * g.R), 0, as_stream(stream),
                     reinterpret_cast<const float4*>(logits), coef, g, reinterpret_cast<float4*>(grad_logits), amax_out);
  XAS_LAUNCH_CHECK();
  return 0;
}
