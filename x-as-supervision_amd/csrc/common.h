// Shared host/device helpers for libxas_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "xas_hip.h"

namespace xas {

void set_error(const char* fmt, ...);
int tune_flags();   // xas_set_tuning value (experiments only; 0 = shipped configuration)

#define XAS_REQUIRE(cond, ...)        \
  do {                                \
    if (!(cond)) {                    \
      ::xas::set_error(__VA_ARGS__);  \
      return 1;                       \
    }                                 \
  } while (0)

#define XAS_LAUNCH_CHECK()                                               \
  do {                                                                   \
    hipError_t e_ = hipGetLastError();                                   \
    if (e_ != hipSuccess) {                                              \
      ::xas::set_error("%s:%d launch failed: %s", __FILE__, __LINE__,    \
                       hipGetErrorString(e_));                           \
      return 2;                                                          \
    }                                                                    \
  } while (0)

constexpr int kWave = 64;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// block-wide sum over blockDim.x threads (multiple of 64, <= 1024); result valid in all threads
__device__ __forceinline__ float block_sum(float v, float* smem /* >= 17 floats */) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) smem[wid] = v;
  __syncthreads();
  float r = (lane < nw) ? smem[lane] : 0.f;
  r = wave_sum(r);
  return r;
}

// z = (x - mean) * (rstd * gamma) + beta with a FIXED operation order (no compiler-chosen contraction), so the
// backward kernels that re-derive the ReLU mask from x reproduce the forward decision bit for bit.
__device__ __forceinline__ float bn_affine(float x, float m, float rs_g, float b) {
  return __fmaf_rn(__fsub_rn(x, m), rs_g, b);
}

// Store of a large result that is streamed out once: non-temporal (global_store ... nt).  Measured (build A/B on one box,
// XAS_HIPCC_DEFS=-DXAS_NO_NT_STORES): x2 upsample forward 3.28 -> 4.70 TB/s, step 213.5 -> 213.0 ms.
__device__ __forceinline__ void stream_store(float4* p, float4 v) {
#ifndef XAS_NO_NT_STORES
  typedef float f32x4_nt_t __attribute__((ext_vector_type(4)));
  __builtin_nontemporal_store(f32x4_nt_t{v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4_nt_t*>(p));
#else
  *p = v;
#endif
}

// Load of a large operand that is read once: non-temporal (global_load ... nt), so that it does not displace the conv
// kernels' re-used operands from the caches.  Batch-norm kernels: step 213.4 -> 211.6 ms (build A/B, -DXAS_NO_NT_LOADS);
// head logits and the accumulate operands of the conv epilogue: another -0.3 ms.
__device__ __forceinline__ float4 stream_load(const float4* p) {
#ifndef XAS_NO_NT_LOADS
  typedef float f32x4_ntl_t __attribute__((ext_vector_type(4)));
  const f32x4_ntl_t v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_ntl_t*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
#else
  return *p;
#endif
}

// ---- recorded maxima (the scales of the f16x3 tensor operands) ---------------------------------------------------------
// A maximum is kept as kAmaxSub sub-maxima kAmaxStride floats (128 bytes) apart - XAS_AMAX_SLOT_FLOATS floats in all; its value
// is the max over the sub-maxima.  One address per tensor serialised EVERY finishing wave of a producer at the L2 atomic unit:
// 16 384 same-address atomics = 190 us flat per launch (r04 trace: a 10 us norm kernel took 190 us with the maximum switched
// on; the backward apply kernels had carried that cost since r03).  Now a block reduces through LDS and issues ONE atomic to
// the sub-maximum its index selects: <= 128 atomics per address and launch.
constexpr int kAmaxSub = XAS_AMAX_SUB, kAmaxStride = XAS_AMAX_STRIDE;
static_assert(kAmaxSub == 32 && kAmaxSub * kAmaxStride == XAS_AMAX_SLOT_FLOATS, "amax slot layout");

// max |v| of a launch -> out (zeroed by the caller before the first launch that merges into it): non-negative floats order
// like their bit patterns, and +inf sorts above every finite value (the consumer then keeps scale 1: the inf reaches its fp16
// pieces and the result, as it would in fp32).  NaN elements do NOT enter the maximum (fmaxf drops them): they travel in the
// data itself.  EVERY thread of the block must call (block-wide barrier inside).
__device__ __forceinline__ void publish_amax(float* out, float v) {
  __shared__ float s_amx[16];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  const int nw = ((int)blockDim.x + 63) >> 6;
  if ((threadIdx.x & 63) == 0) s_amx[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < nw; ++w) v = fmaxf(v, s_amx[w]);
    const unsigned sub = (blockIdx.x + blockIdx.y * gridDim.x) % (unsigned)kAmaxSub;
    if (v > 0.f) atomicMax(reinterpret_cast<unsigned*>(out + sub * kAmaxStride), __float_as_uint(v));
  }
}
__device__ __forceinline__ float amax4(float a, float4 o) {
  return fmaxf(fmaxf(a, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
}
// the value of a recorded maximum (wave-uniform): lane l reads sub-maximum l mod 32, butterfly over the wave
__device__ __forceinline__ float read_amax(const float* amax) {
  float v = amax[(threadIdx.x & (kAmaxSub - 1)) * kAmaxStride];
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline long cdiv(long a, long b) { return (a + b - 1) / b; }

}  // namespace xas
