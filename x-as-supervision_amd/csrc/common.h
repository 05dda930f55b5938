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

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline long cdiv(long a, long b) { return (a + b - 1) / b; }

}  // namespace xas
