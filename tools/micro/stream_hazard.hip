// Reproducer OUTSIDE PyTorch for the multi-stream hazard of profiles/r05_step_reproducibility.md.
//
//   stream M : a trivial "victim" kernel (128-thread blocks, three strided dword loads + three strided dword stores per lane -
//              the shape of patch_to_world_fwd_kernel) on constant inputs, followed by a checker that compares its output with
//              the result computed on an idle GPU and histograms the lane quarter of every mismatch;
//   stream B : the discriminator's linear layer as the library runs it (xas_conv_fwd, 1 image of 72 x 96 "pixels", 128 -> 128,
//              no operand maxima = bf16x6 kernels), in a loop;
//   stream C : a detector weight gradient (xas_conv_wgrad_acc), in a loop.
// usage: stream_hazard [iterations] [mask]   mask bit0: run B, bit1: run C, bit2: B on the exact-fp32 kernels, bit3: C on exact fp32,
//        bit4: victim on the NULL stream, bit5: cross-stream event waits every iteration
// build: hipcc -O3 --offload-arch=gfx950 -Iinclude tools/micro/stream_hazard.hip -o tools/micro/stream_hazard \
//        -Lx-as-supervision_amd/xas_amd -lxas_hip -Wl,-rpath,'$ORIGIN/../../x-as-supervision_amd/xas_amd'
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "xas_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
#define XK(x) do { int r_ = (x); if (r_ != 0) { printf("%s:%d xas rc %d: %s\n", __FILE__, __LINE__, r_, xas_last_error()); exit(1); } } while (0)

__global__ void victim(const float* __restrict__ in, const float* cam, int n, int per, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int b = i / per;
  const float a0 = cam[b * 4], a1 = cam[b * 4 + 1], a2 = cam[b * 4 + 2], a3 = cam[b * 4 + 3];
  const float x = in[i * 3], y = in[i * 3 + 1], z = in[i * 3 + 2];
  const float zc = z * 1992.f + a3;
  out[i * 3] = (x * 127.5f - a0) / a2 * zc;
  out[i * 3 + 1] = (y * 127.5f - a1) / a2 * zc;
  out[i * 3 + 2] = zc;
}

__global__ void copyk(const float* __restrict__ src, float* __restrict__ dst, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}

__global__ void check(const float* __restrict__ out, const float* __restrict__ ref, int n, unsigned* bad /* [5]: total, q0..q3 */) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  bool ok = true;
  for (int c = 0; c < 3; ++c) ok &= __float_as_uint(out[i * 3 + c]) == __float_as_uint(ref[i * 3 + c]);
  if (!ok) { atomicAdd(bad, 1u); atomicAdd(bad + 1 + ((i & 63) >> 4), 1u); }
}

static float* dev_rand(size_t n, float scale, unsigned seed) {
  std::vector<float> h(n);
  unsigned s = seed * 2654435761u + 12345u;
  for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((s >> 8) / 8388608.f - 1.f) * scale; }
  float* d; CK(hipMalloc(&d, n * 4)); CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
  return d;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  const int mask = argc > 2 ? atoi(argv[2]) : 3;
  hipStream_t M, B, C;
  if (mask & 16) M = nullptr;                        // bit4: the victim on the NULL stream (what torch's default stream is)
  else CK(hipStreamCreateWithFlags(&M, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&C, hipStreamNonBlocking));
  // victim
  const int NB = 32, PER = 54, n = NB * PER;
  float* in = dev_rand((size_t)n * 3, 1.f, 1); float* cam = dev_rand(NB * 4, 1.f, 2);
  { std::vector<float> hc(NB * 4); for (int b = 0; b < NB; ++b) { hc[b*4] = 100.f + b; hc[b*4+1] = 90.f - b; hc[b*4+2] = 1145.f; hc[b*4+3] = 5000.f + 10 * b; }
    CK(hipMemcpy(cam, hc.data(), NB * 16, hipMemcpyHostToDevice)); }
  float *out, *ref; unsigned* bad;
  CK(hipMalloc(&out, n * 12)); CK(hipMalloc(&ref, n * 12)); CK(hipMalloc(&bad, 5 * 4)); CK(hipMemset(bad, 0, 20));
  victim<<<(n + 127) / 128, 128, 0, M>>>(in, cam, n, PER, ref);
  CK(hipDeviceSynchronize());
  // bit6: the victim's INPUT is rewritten every iteration by a producer kernel on the same stream (eight rotating sources): a
  // consumer that reads the buffer's previous content shows up as a mismatch
  float* src[8]; float* refs[8];
  for (int k = 0; k < 8; ++k) {
    src[k] = dev_rand((size_t)n * 3, 1.f, 100 + k);
    CK(hipMalloc(&refs[k], n * 12));
    victim<<<(n + 127) / 128, 128, 0, M>>>(src[k], cam, n, PER, refs[k]);
  }
  CK(hipDeviceSynchronize());
  // stream B: linear 128 -> 128 over 6912 rows as one 72 x 96 image; bit7: a chip-filling forward conv instead (256 images of
  // 64 x 64, 64 -> 256: every SIMD hosts its waves, so the victim's waves must share SIMDs with them)
  const bool bigb = (mask & 128) != 0;
  const int bn_ = bigb ? 256 : 1, bh = bigb ? 64 : 72, bw = bigb ? 64 : 96, bci = bigb ? 64 : 128, bco = bigb ? 256 : 128;
  xas_conv_shape sb = {bn_, bh, bw, bci, bco, 1, 1, 1, 0, bh, bw, (mask & 4) ? 1 + XAS_PREC_F32 : 0, nullptr, nullptr};
  float* xb = dev_rand((size_t)bn_ * bh * bw * bci, 1.f, 3); float* wb = dev_rand((size_t)bco * bci, 0.1f, 4); float* bb = dev_rand(bco, 0.1f, 5);
  float* yb; CK(hipMalloc(&yb, (size_t)bn_ * bh * bw * bco * 4));
  const int planes_b = xas_conv_weight_planes(&sb, 0);
  void* wpb = wb;
  if (planes_b) { CK(hipMalloc(&wpb, xas_split_weight_bytes(bco, bci, planes_b))); XK(xas_split_weight(wb, wpb, bco, bci, planes_b, B)); }
  // stream C: weight gradient of a bottleneck 1x1 (64 -> 256 on 64 x 64 maps, 16 images)
  const int cn = bigb ? 256 : 16;
  xas_conv_shape sc = {cn, 64, 64, 64, 256, 1, 1, 1, 0, 64, 64, (mask & 8) ? 1 + XAS_PREC_F32 : 0, nullptr, nullptr};
  float* xc = dev_rand((size_t)cn * 4096 * 64, 1.f, 6); float* dyc = dev_rand((size_t)cn * 4096 * 256, 1e-3f, 7);
  float* gc; CK(hipMalloc(&gc, 256 * 64 * 4)); CK(hipMemset(gc, 0, 256 * 64 * 4));
  float* wsc; CK(hipMalloc(&wsc, (xas_conv_wgrad_workspace_floats(&sc) + 1) * 4));
  printf("B: kernel class %d (planes %d), C: kernel class %d; iterations %d, mask %d\n", xas_conv_kernel_class(&sb, 0), planes_b,
         xas_conv_kernel_class(&sc, 2), iters, mask);
  CK(hipDeviceSynchronize());
  for (int it = 0; it < iters; ++it) {
    CK(hipMemsetAsync(out, 0xFF, n * 12, M));
    if (mask & 64) {
      copyk<<<(n * 3 + 255) / 256, 256, 0, M>>>(src[it & 7], in, n * 3);
      victim<<<(n + 127) / 128, 128, 0, M>>>(in, cam, n, PER, out);
      check<<<(n + 127) / 128, 128, 0, M>>>(out, refs[it & 7], n, bad);
    } else {
      victim<<<(n + 127) / 128, 128, 0, M>>>(in, cam, n, PER, out);
      check<<<(n + 127) / 128, 128, 0, M>>>(out, ref, n, bad);
    }
    if ((mask & 1) && (!bigb || (it & 31) == 0)) { XK(xas_conv_fwd(xb, (const float*)wpb, bb, yb, &sb, B)); if (!bigb) XK(xas_conv_fwd(xb, (const float*)wpb, bb, yb, &sb, B)); }
    if ((mask & 2) && (!bigb || (it & 31) == 16)) XK(xas_conv_wgrad_acc(xc, dyc, gc, wsc, &sc, C));
    if (mask & 32) {                                   // bit5: cross-stream waits as torch's wait_stream issues them (fresh event each)
      hipEvent_t e1, e2;
      CK(hipEventCreateWithFlags(&e1, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
      CK(hipEventRecord(e1, M)); CK(hipStreamWaitEvent(C, e1, 0));            // side.wait_stream(main)
      if ((it & 7) == 7) { CK(hipEventRecord(e2, B)); CK(hipStreamWaitEvent(M, e2, 0)); }   // main.wait_stream(aux)
      CK(hipEventDestroy(e1)); CK(hipEventDestroy(e2));
    }
    if (it % 256 == 255) CK(hipDeviceSynchronize());
  }
  CK(hipDeviceSynchronize());
  unsigned h[5]; CK(hipMemcpy(h, bad, 20, hipMemcpyDeviceToHost));
  printf("mismatching points: %u of %ld checked; by lane quarter (0-15, 16-31, 32-47, 48-63): %u %u %u %u\n", h[0], (long)n * iters, h[1], h[2], h[3], h[4]);
  return 0;
}
