// Bit-exactness of the two-instruction-per-element fp16 split (conv_shared.h split_f16x4: v_fma_mixlo/hi_f16 with the scale
// folded in) against the reference formulation h1 = fp16(s x), h2 = fp16(s x - h1) over random, tiny, huge and special values.
//   hipcc --offload-arch=gfx950 -O3 -I include -I x-as-supervision_amd/csrc tools/micro/split_f16_check.hip -o tools/micro/split_f16_check
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "conv_shared.h"
using namespace xas;

__global__ void k(const float4* x, float s, uint2* a1, uint2* a2, uint2* b1, uint2* b2, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float4 v = x[i];
  uint2 h1, h2;
  split_f16x4(v, s, h1, h2);
  a1[i] = h1; a2[i] = h2;
  float4 r = make_float4(v.x * s, v.y * s, v.z * s, v.w * s);
  const uint2 q1 = pack_f16x4(r);
  r = sub_f16x4(r, q1);
  b1[i] = q1; b2[i] = pack_f16x4(r);
}

int main() {
  const int n = 1 << 20;
  std::vector<float4> h(n);
  srand(1);
  for (int i = 0; i < n; ++i) {
    float t[4];
    for (int e = 0; e < 4; ++e) {
      const int kind = rand() % 16;
      float v = (float)rand() / RAND_MAX * 2.f - 1.f;
      if (kind == 0) v *= 1e-30f; else if (kind == 1) v *= 1e-8f; else if (kind == 2) v *= 1e4f; else if (kind == 3) v = 0.f;
      else if (kind == 4) v = INFINITY; else if (kind == 5) v = NAN; else if (kind == 6) v *= 1e-41f; else if (kind == 7) v = -0.f;
      else v *= expf((float)(rand() % 40 - 20));
      t[e] = v;
    }
    h[i] = make_float4(t[0], t[1], t[2], t[3]);
  }
  float4* dx; uint2 *a1, *a2, *b1, *b2;
  hipMalloc(&dx, n * sizeof(float4));
  hipMalloc(&a1, n * 8); hipMalloc(&a2, n * 8); hipMalloc(&b1, n * 8); hipMalloc(&b2, n * 8);
  hipMemcpy(dx, h.data(), n * sizeof(float4), hipMemcpyHostToDevice);
  long bad = 0;
  for (float s : {1.f, 16.f, 1024.f, 0.0009765625f, 3.0517578125e-05f, 32768.f}) {
    k<<<n / 256, 256>>>(dx, s, a1, a2, b1, b2, n);
    std::vector<uint2> A1(n), A2(n), B1(n), B2(n);
    hipMemcpy(A1.data(), a1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(A2.data(), a2, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(B1.data(), b1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(B2.data(), b2, n * 8, hipMemcpyDeviceToHost);
    long nb = 0;
    auto isnan16 = [](unsigned hbits) { return (hbits & 0x7c00u) == 0x7c00u && (hbits & 0x3ffu); };
    for (int i = 0; i < n; ++i) {
      const unsigned a[4] = {A1[i].x, A1[i].y, A2[i].x, A2[i].y}, b[4] = {B1[i].x, B1[i].y, B2[i].x, B2[i].y};
      for (int e = 0; e < 4; ++e)
        for (int hf = 0; hf < 2; ++hf) {
          const unsigned u = (a[e] >> (16 * hf)) & 0xffffu, w = (b[e] >> (16 * hf)) & 0xffffu;
          // (+0 and -0 count as equal: fma(-0, s, +0) = +0)
          if (u != w && !(isnan16(u) && isnan16(w)) && ((u | w) & 0x7fffu)) { if (nb < 5) printf("s=%g i=%d e=%d.%d: %04x vs %04x (x=%g)\n", s, i, e, hf, u, w, ((float*)&h[i])[(e & 1) * 2 + hf]); ++nb; }
        }
    }
    printf("scale %-12g mismatches %ld of %d\n", s, nb, 8 * n);
    bad += nb;
  }
  printf(bad ? "FAILED\n" : "OK: bit-identical up to the sign of zero\n");
  return bad ? 1 : 0;
}
