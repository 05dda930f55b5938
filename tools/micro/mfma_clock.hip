// Shader clock under sustained fp32-MFMA load: the kernel brackets its MFMA loop with s_memtime (core-clock counter)
// and s_memrealtime (constant 100 MHz), for run lengths from ~10 ms to ~1 s, with and without LDS traffic.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <bool LDS>
__global__ __launch_bounds__(256) void k(float* out, const float* in, int iters, unsigned long long* stamps) {
  __shared__ float4 buf[2048];
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a) for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
  float4 x = make_float4(in[threadIdx.x], in[threadIdx.x + 1], in[threadIdx.x + 2], in[threadIdx.x + 3]);
  float4 y = x;
  if (LDS) { for (int i = threadIdx.x; i < 2048; i += 256) buf[i] = x; __syncthreads(); }
  unsigned long long t0, r0, t1, r1;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
  for (int i = 0; i < iters; ++i) {
    if (LDS) { x = buf[(threadIdx.x + i * 64) & 2047]; y = buf[(threadIdx.x * 3 + i * 32) & 2047]; }
#pragma unroll
    for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x.x, y.x, acc[a], 0, 0, 0);
#pragma unroll
    for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x.y, y.y, acc[a], 0, 0, 0);
#pragma unroll
    for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x.z, y.z, acc[a], 0, 0, 0);
#pragma unroll
    for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x.w, y.w, acc[a], 0, 0, 0);
  }
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
  float s = 0.f;
  for (int a = 0; a < 4; ++a) for (int e = 0; e < 16; ++e) s += acc[a][e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && (blockIdx.x % 64) == 0) { stamps[(blockIdx.x / 64) * 2] = t1 - t0; stamps[(blockIdx.x / 64) * 2 + 1] = r1 - r0; }
}

template <bool LDS>
void run(float* out, float* in, unsigned long long* st, int iters, const char* tag) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int grid = 512;
  hipEventRecord(a); hipLaunchKernelGGL(k<LDS>, dim3(grid), dim3(256), 0, 0, out, in, iters, st); hipEventRecord(b);
  hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b);
  unsigned long long h[16]; hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost);
  double fl = (double)grid * 4 * iters * 16 * 4096.0;
  double clk = 0; for (int i = 0; i < 8; ++i) clk += (double)h[2 * i] / (double)h[2 * i + 1] * 100.0;   // MHz
  printf("%-10s iters=%8d: %9.3f ms  %6.1f TFLOP/s  shader clock %.0f MHz  -> peak at that clock %.1f TFLOP/s\n", tag, iters, ms,
         fl / ms / 1e9, clk / 8, clk / 8 * 1e6 * 256 * 256 / 1e12);
}

int main() {
  float *out, *in; unsigned long long* st;
  hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&in, 1024 * 4); hipMalloc(&st, 16 * 8);
  float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  for (int iters : {2000, 20000, 200000, 1000000}) run<false>(out, in, st, iters, "regs");
  for (int iters : {2000, 20000, 200000, 1000000}) run<true>(out, in, st, iters, "regs+lds");
  return 0;
}
