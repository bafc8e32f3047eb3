// Shared device/host helpers for libdgtta_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/dgtta.h"

#define WAVE 64

#include <atomic>

void dgtta_set_error(const char *fmt, ...);

// Test switches (Switches): a snapshot of the DGTTA_* environment taken ONCE (std::call_once) at first use and
// again only on dgtta_reload_env(); dispatch code reads the snapshot, never the environment.  Every one of them selects
// between kernels that compute the SAME result (a fast kernel and the predecessor it replaced; tests run both).
// -1 = variable not set; otherwise the first character ('0', '1', ...) of its value.
struct DgttaSwitches {
  int conv_rows, conv_s2, dgrad_s2_allcls, wgrad_tr, wgrad_tr8, wgrad_s2_onepass, convt_wgrad_onepass;
  int convt_gemm, rows_order, in_nt, wgrad_upw, wgrad_xcd, in_gstats, softdice16, conv_ring, wgrad_ring, ha_mfma, wgrad_f32_split, feature_head_mfma, headwarp_mfma, wgrad_flat, wgrad_reduce_taps;
  // Laboratory switches: timing models whose results are WRONG BY CONSTRUCTION (*_abl), cycle stamps written past the
  // caller's buffers, measured-no-gain variants.  They exist only in the diagnostic build (-DDGTTA_DIAG ->
  // libdgtta_hip_diag.so, `python -m dg_tta_amd.build --diag`, loaded by profiles/tools/ through DGTTA_LIB); the product
  // library neither reads these variables nor contains the kernel instantiations behind them.
  int rows_abl, rows_var, ring_nt, ring_abl, wgrad_ring_lab, ha_abl, warp_abl, convt_gemm_abl;
  int ncu;      // DGTTA_NCU=<n> (diagnostic build): persistent kernels size their grid for n CUs (CU-masked stream experiments); 0 = unset
};
const DgttaSwitches &dgtta_switches();
#ifdef DGTTA_DIAG
#define DG_LAB(field) (dgtta_switches().field)
#define DG_LAB_NCU(n) (dgtta_switches().ncu > 0 ? dgtta_switches().ncu : (n))
#else
#define DG_LAB(field) (-1)
#define DG_LAB_NCU(n) (n)
#endif

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device): lock-free and per device (a plain
// `static bool` was neither).  One static DynLdsOnce per launch site.
struct DynLdsOnce {
  std::atomic<unsigned long long> mask{0};
};
static inline hipError_t ensure_dyn_lds(DynLdsOnce &o, const void *fn, int bytes) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (o.mask.load(std::memory_order_acquire) & bit) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) o.mask.fetch_or(bit, std::memory_order_release);
  return e;
}

// context of a data-gradient launch that also produces the InstanceNorm backward statistics of the previous block: built by
// dgtta_conv3d_k3_dgrad_gstats and handed DOWN the dispatch chain as an argument (conv3_fwd_mfma -> dispatch_conv ->
// conv3_ring_launch / conv3_rows_launch); the launcher that takes it sets `produced`, the generic kernels ignore it
struct RowsGstCtx {
  const void *y;
  long long ldy;
  const float *mr, *gamma, *beta;
  float slope;
  double *out;
  int produced;
};

#define DG_REQUIRE(cond, code, ...)      \
  do {                                   \
    if (!(cond)) {                       \
      dgtta_set_error(__VA_ARGS__);      \
      return (code);                     \
    }                                    \
  } while (0)

#define DG_CHECK_LAUNCH(name)                                                     \
  do {                                                                            \
    hipError_t e__ = hipGetLastError();                                           \
    if (e__ != hipSuccess) {                                                      \
      dgtta_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));     \
      return DGTTA_ERR_LAUNCH;                                                    \
    }                                                                             \
  } while (0)

__host__ __device__ static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
__host__ __device__ static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---------------------------------------------------------------- bf16 storage helpers
typedef unsigned short bf16_t;  // raw bits

__device__ __forceinline__ float bf16_to_f32(bf16_t h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  // plain cast path: hipcc emits v_cvt_pk_bf16_f32 (round-to-nearest-even, NaN preserved)
  __hip_bfloat16 b = __float2bfloat16(f);
  return *reinterpret_cast<bf16_t *>(&b);
}

// fp16 storage (IEEE binary16) is a distinct C++ type so that templates can tell it from bf16; same size and alignment.
struct f16_t {
  unsigned short v;
};
__device__ __forceinline__ float f16_to_f32(unsigned short h) {
  return __half2float(__ushort_as_half(h));
}
__device__ __forceinline__ unsigned short f32_to_f16(float f) { return __half_as_ushort(__float2half_rn(f)); }

// 16-bit conversions selected by the storage type (bf16_t or f16_t)
template <typename T>
__device__ __forceinline__ unsigned short f32_to_16(float f);
template <>
__device__ __forceinline__ unsigned short f32_to_16<bf16_t>(float f) { return f32_to_bf16(f); }
template <>
__device__ __forceinline__ unsigned short f32_to_16<f16_t>(float f) { return f32_to_f16(f); }
template <typename T>
__device__ __forceinline__ unsigned pack2_16(float lo, float hi) {
  return (unsigned)f32_to_16<T>(lo) | ((unsigned)f32_to_16<T>(hi) << 16);
}
template <typename T>
__device__ __forceinline__ void unpack2_16(unsigned w, float &lo, float &hi);
template <>
__device__ __forceinline__ void unpack2_16<bf16_t>(unsigned w, float &lo, float &hi) {
  lo = __uint_as_float(w << 16);
  hi = __uint_as_float(w & 0xffff0000u);
}
template <>
__device__ __forceinline__ void unpack2_16<f16_t>(unsigned w, float &lo, float &hi) {
  lo = f16_to_f32((unsigned short)(w & 0xffffu));
  hi = f16_to_f32((unsigned short)(w >> 16));
}

template <typename T>
__device__ __forceinline__ float ld_f(const T *p);
template <>
__device__ __forceinline__ float ld_f<float>(const float *p) { return *p; }
template <>
__device__ __forceinline__ float ld_f<bf16_t>(const bf16_t *p) { return bf16_to_f32(*p); }
template <>
__device__ __forceinline__ float ld_f<f16_t>(const f16_t *p) { return f16_to_f32(p->v); }

template <typename T>
__device__ __forceinline__ void st_f(T *p, float v);
template <>
__device__ __forceinline__ void st_f<float>(float *p, float v) { *p = v; }
template <>
__device__ __forceinline__ void st_f<bf16_t>(bf16_t *p, float v) { *p = f32_to_bf16(v); }
template <>
__device__ __forceinline__ void st_f<f16_t>(f16_t *p, float v) { p->v = f32_to_f16(v); }

// ---------------------------------------------------------------- reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}

// Block-wide sum for blockDim.x <= 1024 (multiple of 64). `red` = LDS scratch of >= 16 floats.
// Result valid in every thread.  Deterministic (fixed tree).
__device__ __forceinline__ float block_sum(float v, float *red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < nw; ++i) r += red[i];
  return r;
}

__device__ __forceinline__ float lrelu(float a, float slope) { return a > 0.f ? a : a * slope; }
