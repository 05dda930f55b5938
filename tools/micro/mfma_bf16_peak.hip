// Sustained v_mfma_f32_32x32x16_bf16 rate on this device with NON-ZERO random operands: registers only, 4 accumulators per
// wave, 1-3 blocks of 4 waves per CU, long enough (tens of ms) for the clock to settle under the power limit.  The
// nominal dense bf16 peak (2 516.8 TFLOP/s) assumes 2.4 GHz; this prints what the part sustains, i.e. the ceiling any
// bf16 MFMA kernel - and the bf16x6 convolutions at 1/6 of it - can reach.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

template <bool ZERO>
__global__ __launch_bounds__(256) void k(float* out, const uint4* in, int iters) {
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a) for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
  uint4 xa = in[threadIdx.x], xb = in[threadIdx.x + 256], ya = in[threadIdx.x + 512], yb = in[threadIdx.x + 768];
  if (ZERO) { xa = xb = ya = yb = make_uint4(0, 0, 0, 0); }
  const bf16x8_t a0 = __builtin_bit_cast(bf16x8_t, xa), a1 = __builtin_bit_cast(bf16x8_t, xb);
  const bf16x8_t b0 = __builtin_bit_cast(bf16x8_t, ya), b1 = __builtin_bit_cast(bf16x8_t, yb);
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[3], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int a = 0; a < 4; ++a) for (int e = 0; e < 16; ++e) s += acc[a][e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  float* out; uint4* in;
  hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&in, 1024 * 16);
  unsigned short h[1024 * 8];
  for (int i = 0; i < 1024 * 8; ++i) { float f = (float)rand() / RAND_MAX - 0.5f; unsigned u; memcpy(&u, &f, 4); h[i] = (unsigned short)(u >> 16); }
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int zero = 0; zero < 2; ++zero)
    for (int bpc = 1; bpc <= 3; ++bpc) {
      const int grid = 256 * bpc, iters = 60000 / bpc;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        if (zero) hipLaunchKernelGGL(k<true>, dim3(grid), dim3(256), 0, 0, out, in, iters);
        else hipLaunchKernelGGL(k<false>, dim3(grid), dim3(256), 0, 0, out, in, iters);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        const double fl = (double)grid * 4 /*waves*/ * iters * 16.0 /*mfma*/ * 32768.0;
        printf("%s operands, blocks/CU=%d: %.2f ms  %.1f TFLOP/s bf16  = %.1f TFLOP/s bf16x6-equivalent\n", zero ? "zero" : "random", bpc, ms,
               fl / ms / 1e9, fl / ms / 1e9 / 6);
      }
    }
  return 0;
}
