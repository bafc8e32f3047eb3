// Affine resampling: F.affine_grid + F.grid_sample(align_corners=False) fused, grid never materialised.
// Replaces dg_tta/tta/tta.py:523-551 (image warp, border padding), :572-575 (logit warp back, zeros padding,
// with backward) and dg_tta/tta/torch_utils.py:55-73 (get_batch patch sampling, linear + nearest).
// HBM-bound gather: per output voxel 8 neighbour rows of C channels (L2 absorbs the 8x reuse) + one write;
// algorithmic bytes = 2 * C * 4 B per voxel.
#include "common.h"

namespace {

// normalised base coordinate j of n, as at::affine_grid builds it: linspace(-1,1,n) * (n-1) / n
__device__ __forceinline__ float base_coord(int j, int n) {
  if (n <= 1) return 0.f;
  const float step = 2.0f / (float)(n - 1);
  float v = (j < n / 2) ? (-1.0f + step * (float)j) : (1.0f - step * (float)(n - 1 - j));
  return (v * (float)(n - 1)) / (float)n;
}

struct Sample {
  float ix, iy, iz;
};

__device__ __forceinline__ Sample sample_pos(const float *th, int d, int h, int w, int Dd, int Hd, int Wd, int Ds,
                                             int Hs, int Ws, int algebra, int pad_mode) {
  const float x = base_coord(w, Wd), y = base_coord(h, Hd), z = base_coord(d, Dd);
  float gx = __builtin_fmaf(th[2], z, __builtin_fmaf(th[1], y, th[0] * x)) + th[3];
  float gy = __builtin_fmaf(th[6], z, __builtin_fmaf(th[5], y, th[4] * x)) + th[7];
  float gz = __builtin_fmaf(th[10], z, __builtin_fmaf(th[9], y, th[8] * x)) + th[11];
  if (algebra) {  // tta.py:523-548: grid = (affine_grid(R) - identity_grid) + identity_grid
    gx = (gx - x) + x;
    gy = (gy - y) + y;
    gz = (gz - z) + z;
  }
  Sample s;
  s.ix = ((gx + 1.0f) * (float)Ws - 1.0f) / 2.0f;
  s.iy = ((gy + 1.0f) * (float)Hs - 1.0f) / 2.0f;
  s.iz = ((gz + 1.0f) * (float)Ds - 1.0f) / 2.0f;
  if (pad_mode == DGTTA_PAD_BORDER) {
    s.ix = fminf((float)(Ws - 1), fmaxf(s.ix, 0.f));
    s.iy = fminf((float)(Hs - 1), fmaxf(s.iy, 0.f));
    s.iz = fminf((float)(Ds - 1), fmaxf(s.iz, 0.f));
  }
  return s;
}

struct Corners {
  int x0, y0, z0;
  float w[8];  // order tnw,tne,tsw,tse,bnw,bne,bsw,bse (t: z0, n: y0, w: x0) as ATen's grid_sampler_3d
};

__device__ __forceinline__ Corners corners(const Sample &s) {
  Corners c;
  const float fx = floorf(s.ix), fy = floorf(s.iy), fz = floorf(s.iz);
  c.x0 = (int)fx;
  c.y0 = (int)fy;
  c.z0 = (int)fz;
  const float ex = (fx + 1.0f) - s.ix, ey = (fy + 1.0f) - s.iy, ez = (fz + 1.0f) - s.iz;  // weights of the low side
  const float ux = s.ix - fx, uy = s.iy - fy, uz = s.iz - fz;                                // weights of the high side
  c.w[0] = ex * ey * ez;
  c.w[1] = ux * ey * ez;
  c.w[2] = ex * uy * ez;
  c.w[3] = ux * uy * ez;
  c.w[4] = ex * ey * uz;
  c.w[5] = ux * ey * uz;
  c.w[6] = ex * uy * uz;
  c.w[7] = ux * uy * uz;
  return c;
}

// VEC channels per thread (NDHWC: contiguous; NCDHW: VEC must be 1 and the thread loops over channels)
template <int VEC, bool NDHWC>
__global__ void warp_fwd_kernel(const float *__restrict__ src, const float *__restrict__ theta, float *__restrict__ dst,
                                int C, int Ds, int Hs, int Ws, int Dd, int Hd, int Wd, int src_ldc, int dst_ldc,
                                int pad_mode, int interp, int algebra, const float *__restrict__ sub_const,
                                int64_t total) {
  const int cg = NDHWC ? (C / VEC) : 1;  // channel groups per voxel
  const float sub = sub_const ? sub_const[0] : 0.f;
  const int64_t Vd = (int64_t)Dd * Hd * Wd, Vs = (int64_t)Ds * Hs * Ws;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int g = (int)(i % cg);
    const int64_t vox = i / cg;
    const int b = (int)(vox / Vd);
    const int64_t v = vox % Vd;
    const int w = (int)(v % Wd), h = (int)((v / Wd) % Hd), d = (int)(v / ((int64_t)Wd * Hd));
    const Sample s = sample_pos(theta + b * 12, d, h, w, Dd, Hd, Wd, Ds, Hs, Ws, algebra, pad_mode);
    if (interp == DGTTA_INTERP_NEAREST) {
      const int nx = (int)nearbyintf(s.ix), ny = (int)nearbyintf(s.iy), nz = (int)nearbyintf(s.iz);
      const bool ok = (unsigned)nx < (unsigned)Ws && (unsigned)ny < (unsigned)Hs && (unsigned)nz < (unsigned)Ds;
      const int64_t sv = ((int64_t)nz * Hs + ny) * Ws + nx;
      if (NDHWC) {
#pragma unroll
        for (int k = 0; k < VEC; ++k)
          dst[((int64_t)b * Vd + v) * dst_ldc + g * VEC + k] =
              (ok ? (src[((int64_t)b * Vs + sv) * src_ldc + g * VEC + k] - sub) : 0.f) + sub;
      } else {
        for (int c = 0; c < C; ++c)
          dst[((int64_t)b * C + c) * Vd + v] = (ok ? (src[((int64_t)b * C + c) * Vs + sv] - sub) : 0.f) + sub;
      }
      continue;
    }
    const Corners cr = corners(s);
    int64_t off[8];
    bool ok[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int xx = cr.x0 + (k & 1), yy = cr.y0 + ((k >> 1) & 1), zz = cr.z0 + (k >> 2);
      ok[k] = (unsigned)xx < (unsigned)Ws && (unsigned)yy < (unsigned)Hs && (unsigned)zz < (unsigned)Ds;
      off[k] = ((int64_t)zz * Hs + yy) * Ws + xx;
    }
    if (NDHWC) {
      float acc[VEC];
#pragma unroll
      for (int q = 0; q < VEC; ++q) acc[q] = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (ok[k]) {
          const float *p = src + ((int64_t)b * Vs + off[k]) * src_ldc + g * VEC;
          if (VEC == 4) {
            const float4 t = *reinterpret_cast<const float4 *>(p);
            acc[0] += (t.x - sub) * cr.w[k];
            acc[1] += (t.y - sub) * cr.w[k];
            acc[2] += (t.z - sub) * cr.w[k];
            acc[3] += (t.w - sub) * cr.w[k];
          } else {
#pragma unroll
            for (int q = 0; q < VEC; ++q) acc[q] += (p[q] - sub) * cr.w[k];
          }
        }
      }
      float *o = dst + ((int64_t)b * Vd + v) * dst_ldc + g * VEC;
      if (VEC == 4) {
        *reinterpret_cast<float4 *>(o) = make_float4(acc[0] + sub, acc[1] + sub, acc[2] + sub, acc[3] + sub);
      } else {
#pragma unroll
        for (int q = 0; q < VEC; ++q) o[q] = acc[q] + sub;
      }
    } else {
      for (int c = 0; c < C; ++c) {
        const float *p = src + ((int64_t)b * C + c) * Vs;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (ok[k]) acc += (p[off[k]] - sub) * cr.w[k];
        dst[((int64_t)b * C + c) * Vd + v] = acc + sub;
      }
    }
  }
}

// adjoint of the linear sampler w.r.t. src: scatter-add (fp32 atomics; sums are order dependent in the last bits)
template <int VEC, bool NDHWC>
__global__ void warp_bwd_kernel(const float *__restrict__ gdst, const float *__restrict__ theta, float *__restrict__ gsrc,
                                int C, int Ds, int Hs, int Ws, int Dd, int Hd, int Wd, int src_ldc, int dst_ldc,
                                int pad_mode, int algebra, int64_t total) {
  const int cg = NDHWC ? (C / VEC) : 1;
  const int64_t Vd = (int64_t)Dd * Hd * Wd, Vs = (int64_t)Ds * Hs * Ws;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int g = (int)(i % cg);
    const int64_t vox = i / cg;
    const int b = (int)(vox / Vd);
    const int64_t v = vox % Vd;
    const int w = (int)(v % Wd), h = (int)((v / Wd) % Hd), d = (int)(v / ((int64_t)Wd * Hd));
    const Sample s = sample_pos(theta + b * 12, d, h, w, Dd, Hd, Wd, Ds, Hs, Ws, algebra, pad_mode);
    const Corners cr = corners(s);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int xx = cr.x0 + (k & 1), yy = cr.y0 + ((k >> 1) & 1), zz = cr.z0 + (k >> 2);
      if (!((unsigned)xx < (unsigned)Ws && (unsigned)yy < (unsigned)Hs && (unsigned)zz < (unsigned)Ds)) continue;
      const int64_t off = ((int64_t)zz * Hs + yy) * Ws + xx;
      if (NDHWC) {
        const float *gp = gdst + ((int64_t)b * Vd + v) * dst_ldc + g * VEC;
        float *sp = gsrc + ((int64_t)b * Vs + off) * src_ldc + g * VEC;
#pragma unroll
        for (int q = 0; q < VEC; ++q) atomicAdd(sp + q, gp[q] * cr.w[k]);
      } else {
        for (int c = 0; c < C; ++c)
          atomicAdd(gsrc + ((int64_t)b * C + c) * Vs + off, gdst[((int64_t)b * C + c) * Vd + v] * cr.w[k]);
      }
    }
  }
}

int check_common(const char *name, const void *a, const void *t, const void *o, int B, int C, int Ds, int Hs, int Ws,
                 int Dd, int Hd, int Wd, int ndhwc, int src_ldc, int dst_ldc) {
  DG_REQUIRE(a && t && o, DGTTA_ERR_BADARG, "%s: null pointer", name);
  DG_REQUIRE(B > 0 && C > 0 && Ds > 0 && Hs > 0 && Ws > 0 && Dd > 0 && Hd > 0 && Wd > 0, DGTTA_ERR_BADARG,
             "%s: bad dims", name);
  DG_REQUIRE(!ndhwc || (src_ldc >= C && dst_ldc >= C), DGTTA_ERR_BADARG, "%s: ldc < C", name);
  return DGTTA_OK;
}

int grid_for(int64_t total) {
  int64_t b = (total + 255) / 256;
  return (int)(b < 8192 ? b : 8192);
}

}  // namespace

extern "C" int dgtta_affine_warp3d_fwd(const float *src, const float *theta, float *dst, int B, int C, int Ds, int Hs,
                                       int Ws, int Dd, int Hd, int Wd, int ndhwc, int src_ldc, int dst_ldc, int pad_mode,
                                       int interp_mode, int tta_grid_algebra, const float *sub_const_dev,
                                       void *stream) {
  int rc = check_common("affine_warp3d_fwd", src, theta, dst, B, C, Ds, Hs, Ws, Dd, Hd, Wd, ndhwc, src_ldc, dst_ldc);
  if (rc) return rc;
  DG_REQUIRE(pad_mode == DGTTA_PAD_ZEROS || pad_mode == DGTTA_PAD_BORDER, DGTTA_ERR_BADARG, "warp: bad pad_mode");
  DG_REQUIRE(interp_mode == DGTTA_INTERP_LINEAR || interp_mode == DGTTA_INTERP_NEAREST, DGTTA_ERR_BADARG,
             "warp: bad interp_mode");
  hipStream_t st = (hipStream_t)stream;
  const int64_t Vd = (int64_t)Dd * Hd * Wd;
  if (ndhwc) {
    const bool v4 = (C % 4 == 0) && (src_ldc % 4 == 0) && (dst_ldc % 4 == 0) && ((uintptr_t)src % 16 == 0) &&
                    ((uintptr_t)dst % 16 == 0);
    if (v4) {
      int64_t total = (int64_t)B * Vd * (C / 4);
      hipLaunchKernelGGL((warp_fwd_kernel<4, true>), dim3(grid_for(total)), dim3(256), 0, st, src, theta, dst, C, Ds, Hs,
                         Ws, Dd, Hd, Wd, src_ldc, dst_ldc, pad_mode, interp_mode, tta_grid_algebra, sub_const_dev,
                         total);
    } else {
      int64_t total = (int64_t)B * Vd * C;
      hipLaunchKernelGGL((warp_fwd_kernel<1, true>), dim3(grid_for(total)), dim3(256), 0, st, src, theta, dst, C, Ds, Hs,
                         Ws, Dd, Hd, Wd, src_ldc, dst_ldc, pad_mode, interp_mode, tta_grid_algebra, sub_const_dev,
                         total);
    }
  } else {
    int64_t total = (int64_t)B * Vd;
    hipLaunchKernelGGL((warp_fwd_kernel<1, false>), dim3(grid_for(total)), dim3(256), 0, st, src, theta, dst, C, Ds, Hs,
                       Ws, Dd, Hd, Wd, src_ldc, dst_ldc, pad_mode, interp_mode, tta_grid_algebra, sub_const_dev, total);
  }
  DG_CHECK_LAUNCH("warp_fwd_kernel");
  return DGTTA_OK;
}

extern "C" int dgtta_affine_warp3d_bwd(const float *grad_dst, const float *theta, float *grad_src, int B, int C, int Ds,
                                       int Hs, int Ws, int Dd, int Hd, int Wd, int ndhwc, int src_ldc, int dst_ldc,
                                       int pad_mode, int tta_grid_algebra, void *stream) {
  int rc = check_common("affine_warp3d_bwd", grad_dst, theta, grad_src, B, C, Ds, Hs, Ws, Dd, Hd, Wd, ndhwc, src_ldc,
                        dst_ldc);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const int64_t Vd = (int64_t)Dd * Hd * Wd;
  if (ndhwc) {
    // one lane per channel: a wave-instruction's atomics then cover whole 64-byte rows (C=16) instead of 16-byte pieces
    if (false && C % 4 == 0) {
      int64_t total = (int64_t)B * Vd * (C / 4);
      hipLaunchKernelGGL((warp_bwd_kernel<4, true>), dim3(grid_for(total)), dim3(256), 0, st, grad_dst, theta, grad_src,
                         C, Ds, Hs, Ws, Dd, Hd, Wd, src_ldc, dst_ldc, pad_mode, tta_grid_algebra, total);
    } else {
      int64_t total = (int64_t)B * Vd * C;
      hipLaunchKernelGGL((warp_bwd_kernel<1, true>), dim3(grid_for(total)), dim3(256), 0, st, grad_dst, theta, grad_src,
                         C, Ds, Hs, Ws, Dd, Hd, Wd, src_ldc, dst_ldc, pad_mode, tta_grid_algebra, total);
    }
  } else {
    int64_t total = (int64_t)B * Vd;
    hipLaunchKernelGGL((warp_bwd_kernel<1, false>), dim3(grid_for(total)), dim3(256), 0, st, grad_dst, theta, grad_src, C,
                       Ds, Hs, Ws, Dd, Hd, Wd, src_ldc, dst_ldc, pad_mode, tta_grid_algebra, total);
  }
  DG_CHECK_LAUNCH("warp_bwd_kernel");
  return DGTTA_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Gaussian-weighted sliding-window accumulation (post-TTA ensemble inference; nnU-Net's
// predict_sliding_window_return_logits [3P nnunetv2==2.2.1], reached from dg_tta/tta/nnunet_utils.py:116-125,208-230):
//   acc[v0 + p][c] += patch[p][c] * gauss[p];   nsum[v0 + p] += gauss[p]        (voxel-major fp32 accumulators)
// One lane per (patch voxel, channel): rows of C floats are contiguous, windows overlap only between launches.
namespace {
__global__ void window_accumulate_kernel(const float *__restrict__ patch, const float *__restrict__ gauss,
                                         float *__restrict__ acc, float *__restrict__ nsum, int C, int PD, int PH, int PW,
                                         int X, int Y, int Z, int x0, int y0, int z0, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int64_t p = i / C;
    const int pw = (int)(p % PW), ph = (int)((p / PW) % PH), pd = (int)(p / ((int64_t)PW * PH));
    const int64_t v = ((int64_t)(x0 + pd) * Y + (y0 + ph)) * Z + (z0 + pw);
    const float g = gauss[p];
    acc[v * C + c] += patch[i] * g;
    if (c == 0 && nsum) nsum[v] += g;
  }
}
}  // namespace

extern "C" int dgtta_window_accumulate(const float *patch, const float *gauss, float *acc, float *nsum, int C, int PD, int PH,
                                       int PW, int X, int Y, int Z, int x0, int y0, int z0, void *stream) {
  DG_REQUIRE(patch && gauss && acc, DGTTA_ERR_BADARG, "window_accumulate: null pointer");
  DG_REQUIRE(C > 0 && PD > 0 && PH > 0 && PW > 0 && x0 >= 0 && y0 >= 0 && z0 >= 0 && x0 + PD <= X && y0 + PH <= Y &&
                 z0 + PW <= Z,
             DGTTA_ERR_BADARG, "window_accumulate: window outside the volume");
  const int64_t total = (int64_t)PD * PH * PW * C;
  hipLaunchKernelGGL(window_accumulate_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, patch, gauss, acc,
                     nsum, C, PD, PH, PW, X, Y, Z, x0, y0, z0, total);
  DG_CHECK_LAUNCH("window_accumulate_kernel");
  return DGTTA_OK;
}
