// Raw v_mfma_f32_32x32x2_f32 issue rate on this device: registers only, N accumulators, W waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, const float* in, int iters) {
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a) for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
  float x = in[threadIdx.x], y = in[threadIdx.x + 256];
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
  }
  float s = 0.f;
  for (int a = 0; a < NACC; ++a) for (int e = 0; e < 16; ++e) s += acc[a][e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  float *out, *in;
  hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&in, 512 * 4);
  float h[512]; for (int i = 0; i < 512; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int blocks_per_cu = 1; blocks_per_cu <= 2; ++blocks_per_cu) {
    int grid = 256 * blocks_per_cu, iters = 4000;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(a); hipLaunchKernelGGL(k<4>, dim3(grid), dim3(256), 0, 0, out, in, iters); hipEventRecord(b);
      hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b);
      double fl = (double)grid * 4 * iters * 16 * 4096.0;
      printf("NACC=4 blocks/CU=%d: %.3f ms  %.1f TFLOP/s\n", blocks_per_cu, ms, fl / ms / 1e9);
    }
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(a); hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, out, in, iters * 4); hipEventRecord(b);
      hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b);
      double fl = (double)grid * 4 * iters * 4 * 4 * 4096.0;
      printf("NACC=1 blocks/CU=%d: %.3f ms  %.1f TFLOP/s\n", blocks_per_cu, ms, fl / ms / 1e9);
    }
  }
  return 0;
}
