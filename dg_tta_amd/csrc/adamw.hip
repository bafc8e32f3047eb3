// Multi-tensor AdamW step (decoupled weight decay, bias corrected), HBM-bound: 4 reads + 3 writes of fp32 per element.
// Replaces torch.optim.AdamW(model.parameters(), lr).step() of dg_tta/tta/tta.py:185,278 (PyTorch defaults:
// betas (0.9, 0.999), eps 1e-8, weight_decay 0.01, amsgrad off).  Math order follows torch's single-tensor path:
//   p *= 1 - lr*wd;  m = lerp(m, g, 1-b1);  v = b2*v + (1-b2)*g*g;
//   p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
// Tensors are passed in chunks of up to 32 per launch through the kernel-argument buffer (no device-side table,
// no host->device copies, graph-capture safe).
#include "common.h"
#include <math.h>

namespace {

constexpr int TPL = 32;  // tensors per launch

struct AdamArgs {
  float *p[TPL];
  const float *g[TPL];
  float *m[TPL];
  float *v[TPL];
  int64_t n[TPL];
  int blk_start[TPL + 1];  // first workgroup of tensor i
};

constexpr int CHUNK = 256 * 16;  // elements per workgroup

__global__ __launch_bounds__(256) void adamw_kernel(AdamArgs a, int nt, float lr_wd, float b1, float b2, float eps,
                                                    float step_size, float sqrt_bc2, float inv_scale,
                                                    const int *__restrict__ skip) {
  if (skip != nullptr && *skip != 0) return;      // a gradient overflowed (fp16 loss scale too high): the step is skipped
  int t = 0;
  while (t + 1 < nt && (int)blockIdx.x >= a.blk_start[t + 1]) ++t;
  const int64_t base = (int64_t)((int)blockIdx.x - a.blk_start[t]) * CHUNK;
  float *p = a.p[t], *m = a.m[t], *v = a.v[t];
  const float *g = a.g[t];
  const int64_t n = a.n[t];
#pragma unroll 4
  for (int k = 0; k < CHUNK / 256; ++k) {
    const int64_t i = base + k * 256 + threadIdx.x;
    if (i < n) {
      float gi = g[i] * inv_scale, pi = p[i], mi = m[i], vi = v[i];       // inv_scale = 1 / loss scale (exactly 1 if unused)
      pi = pi * (1.0f - lr_wd);
      mi = mi + (gi - mi) * (1.0f - b1);
      vi = vi * b2 + ((1.0f - b2) * gi) * gi;
      float denom = sqrtf(vi) / sqrt_bc2 + eps;
      pi = pi - step_size * (mi / denom);
      p[i] = pi;
      m[i] = mi;
      v[i] = vi;
    }
  }
}

// Sets *flag when any gradient element is inf / NaN (the fp16 storage path's activation gradients can overflow when the
// loss scale is too high); read-only pass over the gradients, same tensor chunking as the step itself.
__global__ __launch_bounds__(256) void grads_nonfinite_kernel(AdamArgs a, int nt, int *__restrict__ flag) {
  int t = 0;
  while (t + 1 < nt && (int)blockIdx.x >= a.blk_start[t + 1]) ++t;
  const int64_t base = (int64_t)((int)blockIdx.x - a.blk_start[t]) * CHUNK;
  const float *g = a.g[t];
  const int64_t n = a.n[t];
  bool bad = false;
#pragma unroll 4
  for (int k = 0; k < CHUNK / 256; ++k) {
    const int64_t i = base + k * 256 + threadIdx.x;
    if (i < n) bad |= !(fabsf(g[i]) <= 3.4028234e38f);      // false for inf and NaN
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

}  // namespace

extern "C" int dgtta_grads_nonfinite(const float *const *h_g, const int64_t *h_n, int ntensors, int *flag, void *stream) {
  DG_REQUIRE(h_g && h_n && flag, DGTTA_ERR_BADARG, "grads_nonfinite: null pointer");
  int i = 0;
  while (i < ntensors) {
    AdamArgs a;
    int nt = 0, blocks = 0;
    while (i < ntensors && nt < TPL) {
      if (h_g[i] != nullptr && h_n[i] > 0) {
        a.g[nt] = h_g[i];
        a.n[nt] = h_n[i];
        a.blk_start[nt] = blocks;
        blocks += (int)cdiv64(h_n[i], CHUNK);
        ++nt;
      }
      ++i;
    }
    if (nt == 0) break;
    a.blk_start[nt] = blocks;
    hipLaunchKernelGGL(grads_nonfinite_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, nt, flag);
    DG_CHECK_LAUNCH("grads_nonfinite_kernel");
  }
  return DGTTA_OK;
}

extern "C" int dgtta_adamw_step(float *const *h_p, const float *const *h_g, float *const *h_m, float *const *h_v,
                                const int64_t *h_n, int ntensors, float lr, float beta1, float beta2, float eps,
                                float weight_decay, int step, float grad_scale, const int *skip_if_nonzero,
                                void *stream) {
  DG_REQUIRE(h_p && h_g && h_m && h_v && h_n, DGTTA_ERR_BADARG, "adamw_step: null table");
  DG_REQUIRE(ntensors >= 0 && step >= 1, DGTTA_ERR_BADARG, "adamw_step: bad ntensors/step");
  DG_REQUIRE(grad_scale > 0.f, DGTTA_ERR_BADARG, "adamw_step: grad_scale must be positive (1 = gradients are unscaled)");
  hipStream_t st = (hipStream_t)stream;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr / bc1);
  const float sqrt_bc2 = (float)sqrt(bc2);
  int i = 0;
  while (i < ntensors) {
    AdamArgs a;
    int nt = 0, blocks = 0;
    while (i < ntensors && nt < TPL) {
      if (h_g[i] != nullptr && h_n[i] > 0) {
        DG_REQUIRE(h_p[i] && h_m[i] && h_v[i], DGTTA_ERR_BADARG, "adamw_step: null tensor %d", i);
        a.p[nt] = h_p[i];
        a.g[nt] = h_g[i];
        a.m[nt] = h_m[i];
        a.v[nt] = h_v[i];
        a.n[nt] = h_n[i];
        a.blk_start[nt] = blocks;
        blocks += (int)cdiv64(h_n[i], CHUNK);
        ++nt;
      }
      ++i;
    }
    if (nt == 0) break;
    a.blk_start[nt] = blocks;
    hipLaunchKernelGGL(adamw_kernel, dim3(blocks), dim3(256), 0, st, a, nt, lr * weight_decay, beta1, beta2, eps,
                       step_size, sqrt_bc2, 1.0f / grad_scale, skip_if_nonzero);
    DG_CHECK_LAUNCH("adamw_kernel");
  }
  return DGTTA_OK;
}
