// Affine resampling: F.affine_grid + F.grid_sample(align_corners=False) fused, grid never materialised.
// Replaces dg_tta/tta/tta.py:523-551 (image warp, border padding), :572-575 (logit warp back, zeros padding,
// with backward) and dg_tta/tta/torch_utils.py:55-73 (get_batch patch sampling, linear + nearest).
// HBM-bound gather: per output voxel 8 neighbour rows of C channels (L2 absorbs the 8x reuse) + one write;
// algorithmic bytes = 2 * C * 4 B per voxel.
#include "common.h"
#include <limits.h>

namespace {

// normalised base coordinate j of n, as at::affine_grid builds it: linspace(-1,1,n) * (n-1) / n
__device__ __forceinline__ float base_coord(int j, int n) {
  if (n <= 1) return 0.f;
  const float step = 2.0f / (float)(n - 1);
  float v = (j < n / 2) ? (-1.0f + step * (float)j) : (1.0f - step * (float)(n - 1 - j));
  return (v * (float)(n - 1)) / (float)n;
}

// (Round 3 measured an XCD-contiguous tile order for these kernels - no gain forward, 20 % slower backward: both wait on L1
// misses that hit in L2, not on HBM - and a cooperative owner / loader backward, bit-identical and no faster; both prototypes
// were removed in round 5, the numbers are in DESIGN.md.)

struct Sample {
  float ix, iy, iz;
};

__device__ __forceinline__ Sample sample_from_base(const float *th, float x, float y, float z, int Ds, int Hs, int Ws,
                                                   int algebra, int pad_mode);

__device__ __forceinline__ Sample sample_pos(const float *th, int d, int h, int w, int Dd, int Hd, int Wd, int Ds,
                                             int Hs, int Ws, int algebra, int pad_mode) {
  return sample_from_base(th, base_coord(w, Wd), base_coord(h, Hd), base_coord(d, Dd), Ds, Hs, Ws, algebra, pad_mode);
}

__device__ __forceinline__ Sample sample_from_base(const float *th, float x, float y, float z, int Ds, int Hs, int Ws,
                                                   int algebra, int pad_mode) {
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
  const bool idx32 = total <= 0x7fffffffll;      // (uniform) five 64-bit divisions per item cost more than the gather itself
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int g, b, w, h, d;
    int64_t v;
    if (idx32) {
      const unsigned ii = (unsigned)i, vox = ii / (unsigned)cg, vd = (unsigned)Vd;
      g = (int)(ii - vox * (unsigned)cg);
      b = (int)(vox / vd);
      const unsigned vv = vox - (unsigned)b * vd, t = vv / (unsigned)Wd;
      w = (int)(vv - t * (unsigned)Wd);
      d = (int)(t / (unsigned)Hd);
      h = (int)(t - (unsigned)d * (unsigned)Hd);
      v = vv;
    } else {
      g = (int)(i % cg);
      const int64_t vox = i / cg;
      b = (int)(vox / Vd);
      v = vox % Vd;
      w = (int)(v % Wd), h = (int)((v / Wd) % Hd), d = (int)(v / ((int64_t)Wd * Hd));
    }
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
      // (round 6) the offset of the CLAMPED corner: every corner is read - all eight loads in flight instead of eight
      // load - wait - blend blocks behind `if (ok)` - and a corner outside the volume is replaced by `sub` (it then adds +0)
      off[k] = ((int64_t)min(max(zz, 0), Ds - 1) * Hs + min(max(yy, 0), Hs - 1)) * Ws + min(max(xx, 0), Ws - 1);
    }
    if (NDHWC) {
      float acc[VEC];
#pragma unroll
      for (int q = 0; q < VEC; ++q) acc[q] = 0.f;
      float tv[8][VEC];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float *p = src + ((int64_t)b * Vs + off[k]) * src_ldc + g * VEC;
        if (VEC == 4) {
          const float4 t = *reinterpret_cast<const float4 *>(p);
          tv[k][0] = t.x;
          tv[k][1 % VEC] = t.y;
          tv[k][2 % VEC] = t.z;
          tv[k][3 % VEC] = t.w;
        } else {
#pragma unroll
          for (int q = 0; q < VEC; ++q) tv[k][q] = p[q];
        }
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
#pragma unroll
        for (int q = 0; q < VEC; ++q) acc[q] += ((ok[k] ? tv[k][q] : sub) - sub) * (ok[k] ? cr.w[k] : 0.f);
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
        float acc = 0.f, tv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) tv[k] = p[off[k]];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += ((ok[k] ? tv[k] : sub) - sub) * (ok[k] ? cr.w[k] : 0.f);
        dst[((int64_t)b * C + c) * Vd + v] = acc + sub;
      }
    }
  }
}

// Voxel-space linearisation of the sampling map (it is affine): S(v) = s0 + M v, with the inverse of M.  Only used to
// bound candidate ranges; the contributions themselves are always recomputed with the exact forward arithmetic.
struct InvMap {
  float m[3][3];     // rows (ix,iy,iz) x cols (w,h,d)
  float inv[3][3];   // rows (w,h,d) x cols (ix,iy,iz)
  float e[3];        // |inv| row sums = half-extent of the pre-image of a unit half-width box
  float s0[3];
  bool ok;           // invertible, finite, and at most 512 candidate voxels per source voxel
};

__device__ __forceinline__ InvMap inverse_map(const float *th, int Ds, int Hs, int Ws, int Dd, int Hd, int Wd, int algebra) {
  InvMap r;
  const Sample s0 = sample_pos(th, 0, 0, 0, Dd, Hd, Wd, Ds, Hs, Ws, algebra, DGTTA_PAD_ZEROS);
  r.s0[0] = s0.ix;
  r.s0[1] = s0.iy;
  r.s0[2] = s0.iz;
  const float m00 = th[0] * Ws / Wd, m01 = th[1] * Ws / Hd, m02 = th[2] * Ws / Dd;
  const float m10 = th[4] * Hs / Wd, m11 = th[5] * Hs / Hd, m12 = th[6] * Hs / Dd;
  const float m20 = th[8] * Ds / Wd, m21 = th[9] * Ds / Hd, m22 = th[10] * Ds / Dd;
  const float c00 = m11 * m22 - m12 * m21, c01 = m12 * m20 - m10 * m22, c02 = m10 * m21 - m11 * m20;
  const float det = m00 * c00 + m01 * c01 + m02 * c02;
  const float idet = 1.0f / det;
  r.m[0][0] = m00; r.m[0][1] = m01; r.m[0][2] = m02;
  r.m[1][0] = m10; r.m[1][1] = m11; r.m[1][2] = m12;
  r.m[2][0] = m20; r.m[2][1] = m21; r.m[2][2] = m22;
  r.inv[0][0] = c00 * idet;
  r.inv[0][1] = (m02 * m21 - m01 * m22) * idet;
  r.inv[0][2] = (m01 * m12 - m02 * m11) * idet;
  r.inv[1][0] = c01 * idet;
  r.inv[1][1] = (m00 * m22 - m02 * m20) * idet;
  r.inv[1][2] = (m02 * m10 - m00 * m12) * idet;
  r.inv[2][0] = c02 * idet;
  r.inv[2][1] = (m01 * m20 - m00 * m21) * idet;
  r.inv[2][2] = (m00 * m11 - m01 * m10) * idet;
  float vol = 1.f;
  bool fin = det != 0.f && isfinite(idet) && isfinite(s0.ix) && isfinite(s0.iy) && isfinite(s0.iz);
  for (int i = 0; i < 3; ++i) {
    r.e[i] = fabsf(r.inv[i][0]) + fabsf(r.inv[i][1]) + fabsf(r.inv[i][2]);
    fin = fin && isfinite(r.e[i]);
    vol *= 2.0f * r.e[i] + 1.0f;
  }
  r.ok = fin && vol <= 512.0f;
  return r;
}

// adjoint of the linear sampler w.r.t. src: scatter-add (fp32 atomics; sums are order dependent in the last bits)
template <int VEC, bool NDHWC>
__global__ void warp_bwd_kernel(const float *__restrict__ gdst, const float *__restrict__ theta, float *__restrict__ gsrc,
                                int C, int Ds, int Hs, int Ws, int Dd, int Hd, int Wd, int src_ldc, int dst_ldc,
                                int pad_mode, int algebra, int only_declined, int B, int64_t total) {
  __shared__ int declined[16];
  if (only_declined) {   // fallback role: only batches the gather kernel declined; normally none -> exit at once
    if (threadIdx.x < 16)
      declined[threadIdx.x] = (int)threadIdx.x < B &&
                              !inverse_map(theta + threadIdx.x * 12, Ds, Hs, Ws, Dd, Hd, Wd, algebra).ok;
    __syncthreads();
    int any = 0;
    for (int q = 0; q < 16; ++q) any |= declined[q];
    if (!any) return;
  }
  const int cg = NDHWC ? (C / VEC) : 1;
  const int64_t Vd = (int64_t)Dd * Hd * Wd, Vs = (int64_t)Ds * Hs * Ws;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int g = (int)(i % cg);
    const int64_t vox = i / cg;
    const int b = (int)(vox / Vd);
    if (only_declined && !declined[b]) continue;
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

// Atomic-free, deterministic adjoint for NDHWC + zeros padding: one thread OWNS one grad_src voxel (CH channels in
// registers).  Because the map is affine, the dst voxels whose trilinear footprint touches source voxel u lie in the
// pre-image of u +- 1, a small box around M^-1 (u - s0) (2-3 lattice points per axis for the near-identity maps of
// tta.py:523-548).  Each candidate's sample position and corner weight are recomputed with the forward's arithmetic,
// so every term equals the scatter formulation's; only the (now fixed) summation order differs.  Neighbouring lanes
// read neighbouring grad_dst rows (L1/L2 hits), stores are full coalesced rows, grad_src needs no zero-init.
template <int CH, bool VEC>
__global__ __launch_bounds__(256) void warp_bwd_gather_kernel(const float *__restrict__ gdst,
                                                              const float *__restrict__ theta, float *__restrict__ gsrc,
                                                              int C, int Ds, int Hs, int Ws, int Dd, int Hd, int Wd,
                                                              int src_ldc, int dst_ldc, int algebra, int gx, int gy) {
  const int64_t Vd = (int64_t)Dd * Hd * Wd, Vs = (int64_t)Ds * Hs * Ws;
  // workgroup = compact 16 x 4 x 4 tile of source voxels (small grad_dst footprint -> L1/L2 reuse of the 8x overlap).
  // (4 lanes per voxel with 4 channels each would make the row loads 4x denser per instruction, but repeats the
  // candidate search 4x and measured slower: 374 vs 206 us at 128^3 x 16.)
  const int tilesX = (Ws + 15) >> 4, tilesZ = (Ds + 3) >> 2;
  const int tile = (int)blockIdx.x;
  const int bx = tile % gx, by = (tile / gx) % gy, bz = tile / (gx * gy);
  const int c0 = (bx / tilesX) * CH;
  const int b = bz / tilesZ;
  const int x = (bx % tilesX) * 16 + (threadIdx.x & 15), y = by * 4 + ((threadIdx.x >> 4) & 3),
            z = (bz % tilesZ) * 4 + (threadIdx.x >> 6);
  // the inverse map is the same for the whole workgroup (one batch item): one thread evaluates it (~150 instructions,
  // 4 calls of the sample-position arithmetic), the others read the 22 numbers from LDS
  __shared__ InvMap s_im;
  const float *th = theta + b * 12;
  if (threadIdx.x == 0) s_im = inverse_map(th, Ds, Hs, Ws, Dd, Hd, Wd, algebra);
  __syncthreads();
  if (x < Ws && y < Hs && z < Ds) {
    const int64_t u = ((int64_t)z * Hs + y) * Ws + x;
    const InvMap im = s_im;
    if (!im.ok) return;   // the scatter kernels (launched next) take over (uniform per batch item)
    const float rel[3] = {(float)x - im.s0[0], (float)y - im.s0[1], (float)z - im.s0[2]};
    int lo[3], hi[3];
    const int dims[3] = {Wd, Hd, Dd};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float c = im.inv[a][0] * rel[0] + im.inv[a][1] * rel[1] + im.inv[a][2] * rel[2];
      const float slack = im.e[a] + 0.02f;
      lo[a] = max(0, (int)fmaxf(ceilf(c - slack), -1.0f));
      hi[a] = min(dims[a] - 1, (int)fminf(floorf(c + slack), (float)dims[a]));
    }
    float acc[CH];
#pragma unroll
    for (int q = 0; q < CH; ++q) acc[q] = 0.f;
    // per (d,h) line the three constraints |S_j(w,h,d) - u_j| < 1 bound w to ~1 lattice point (linear model + slack)
    float rcp0[3];
    bool bounds_w[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      bounds_w[j] = fabsf(im.m[j][0]) > 1e-6f;
      rcp0[j] = bounds_w[j] ? 1.0f / im.m[j][0] : 0.f;
    }
    for (int d = lo[2]; d <= hi[2]; ++d) {
      const float zc = base_coord(d, Dd);
      for (int h = lo[1]; h <= hi[1]; ++h) {
        const float yc = base_coord(h, Hd);
        float wl = (float)lo[0], wh = (float)hi[0];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const float r = -rel[j] + im.m[j][1] * (float)h + im.m[j][2] * (float)d;
          if (bounds_w[j]) {
            const float a = (-1.02f - r) * rcp0[j], bq = (1.02f - r) * rcp0[j];
            wl = fmaxf(wl, fminf(a, bq));
            wh = fminf(wh, fmaxf(a, bq));
          } else if (fabsf(r) > 1.02f) {
            wh = wl - 1.0f;
          }
        }
        const int w1 = (int)floorf(wh);
        for (int w = (int)ceilf(wl); w <= w1; ++w) {
          const Sample s = sample_from_base(th, base_coord(w, Wd), yc, zc, Ds, Hs, Ws, algebra, DGTTA_PAD_ZEROS);
          const float fx = floorf(s.ix), fy = floorf(s.iy), fz = floorf(s.iz);
          // weight of source voxel (x,y,z) in this sample: low corner -> (f+1)-i, high corner -> i-f (as corners())
          const float dx = (float)x - fx, dy = (float)y - fy, dz = (float)z - fz;
          if (!((dx == 0.f || dx == 1.f) && (dy == 0.f || dy == 1.f) && (dz == 0.f || dz == 1.f))) continue;
          const float wx = dx == 0.f ? (fx + 1.0f) - s.ix : s.ix - fx;
          const float wy = dy == 0.f ? (fy + 1.0f) - s.iy : s.iy - fy;
          const float wz = dz == 0.f ? (fz + 1.0f) - s.iz : s.iz - fz;
          const float wt = wx * wy * wz;
          const float *gp = gdst + ((int64_t)b * Vd + ((int64_t)d * Hd + h) * Wd + w) * dst_ldc + c0;
          if (VEC) {
#pragma unroll
            for (int q = 0; q < CH; q += 4) {
              if (c0 + q < C) {
                const float4 g = *reinterpret_cast<const float4 *>(gp + q);
                acc[q] += g.x * wt;
                acc[q + 1] += g.y * wt;
                acc[q + 2] += g.z * wt;
                acc[q + 3] += g.w * wt;
              }
            }
          } else {
#pragma unroll
            for (int q = 0; q < CH; ++q)
              if (c0 + q < C) acc[q] += gp[q] * wt;
          }
        }
      }
    }
    float *o = gsrc + ((int64_t)b * Vs + u) * src_ldc + c0;
    if (VEC) {
#pragma unroll
      for (int q = 0; q < CH; q += 4)
        if (c0 + q < C) *reinterpret_cast<float4 *>(o + q) = make_float4(acc[q], acc[q + 1], acc[q + 2], acc[q + 3]);
    } else {
#pragma unroll
      for (int q = 0; q < CH; ++q)
        if (c0 + q < C) o[q] = acc[q];
    }
  }
}

// Fallback for maps the gather kernel declines (singular / extreme minification): zero, then scatter with atomics.
__global__ void warp_bwd_zero_if_declined_kernel(const float *__restrict__ theta, float *__restrict__ gsrc, int B, int Ds,
                                                 int Hs, int Ws, int Dd, int Hd, int Wd, int algebra, int64_t per_batch) {
  for (int b = 0; b < B; ++b) {
    if (inverse_map(theta + b * 12, Ds, Hs, Ws, Dd, Hd, Wd, algebra).ok) continue;
    float *g = gsrc + (int64_t)b * per_batch;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < per_batch; i += (int64_t)gridDim.x * blockDim.x)
      g[i] = 0.f;
  }
}

// NDHWC, trilinear, 4 channels per thread: the general kernel above spends most of its time on the five 64-bit
// divisions that turn a flat index into (b, d, h, w, group).  Here the grid carries (h, d * B + b) and a workgroup walks
// one output row: 32-bit index arithmetic only, same sample_pos / corners arithmetic, same summation order (1069 -> 826
// us for 8 x 128^3 x 16 channels).  What is left is the L2 -> L1 traffic of the 8 corner rows (one thread per voxel with 16
// channels halves the VALU work but makes every access a 16-byte piece of a different line: 1150 us).
constexpr int WARP_ROWS = 8;
__global__ __launch_bounds__(256) void warp_fwd_rows4_kernel(const float *__restrict__ src, const float *__restrict__ theta,
                                                             float *__restrict__ dst, int C, int Ds, int Hs, int Ws, int Dd,
                                                             int Hd, int Wd, int src_ldc, int dst_ldc, int pad_mode,
                                                             int algebra, const float *__restrict__ sub_const, int gx, int gy) {
  const int cg = C >> 2;
  const float sub = sub_const ? sub_const[0] : 0.f;
  const int tile = (int)blockIdx.x;
  const int bx = tile % gx, by = (tile / gx) % gy, bz = tile / (gx * gy);
  const int d = bz % Dd, b = bz / Dd;
  const int64_t Vd = (int64_t)Dd * Hd * Wd, Vs = (int64_t)Ds * Hs * Ws;
  const float *sb = src + (int64_t)b * Vs * src_ldc;
  const int items = Wd * cg;
  const int i = bx * 256 + threadIdx.x;
  if (i >= items) return;
  const int w = i / cg, g = i - w * cg;
  // WARP_ROWS consecutive output rows per thread (the loads of a row are independent of the previous row's stores)
  const int h0 = by * WARP_ROWS;
#pragma unroll 2
  for (int h = h0; h < min(h0 + WARP_ROWS, Hd); ++h) {
    const Sample s = sample_pos(theta + b * 12, d, h, w, Dd, Hd, Wd, Ds, Hs, Ws, algebra, pad_mode);
    const Corners cr = corners(s);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    // (round 6) all eight corner rows in flight: clamped addresses, a corner outside the volume zeroed (it then adds +0)
    float4 t[8];
    bool ok[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int xx = cr.x0 + (k & 1), yy = cr.y0 + ((k >> 1) & 1), zz = cr.z0 + (k >> 2);
      ok[k] = (unsigned)xx < (unsigned)Ws && (unsigned)yy < (unsigned)Hs && (unsigned)zz < (unsigned)Ds;
      const int xc = min(max(xx, 0), Ws - 1), yc = min(max(yy, 0), Hs - 1), zc = min(max(zz, 0), Ds - 1);
      t[k] = *reinterpret_cast<const float4 *>(sb + (((int64_t)zc * Hs + yc) * Ws + xc) * src_ldc + g * 4);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float wk = ok[k] ? cr.w[k] : 0.f;
      acc[0] += ((ok[k] ? t[k].x : sub) - sub) * wk;
      acc[1] += ((ok[k] ? t[k].y : sub) - sub) * wk;
      acc[2] += ((ok[k] ? t[k].z : sub) - sub) * wk;
      acc[3] += ((ok[k] ? t[k].w : sub) - sub) * wk;
    }
    float *drow = dst + ((int64_t)b * Vd + ((int64_t)d * Hd + h) * Wd) * dst_ldc;
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    const f32x4_t ov = {acc[0] + sub, acc[1] + sub, acc[2] + sub, acc[3] + sub};
    // (non-temporal stores measured no different here, round 3)
    *reinterpret_cast<f32x4_t *>(drow + (int64_t)w * dst_ldc + g * 4) = ov;
  }
}

// =====================================================================================================================
// Segmentation head fused with the inverse warp of the logits (round 3).
//
// The reference maps a branch's logits back to the common frame with grid_sample (tta.py:572-575) after the 1x1x1 head of
// the network ([3P] PlainConvUNet seg layer) and map_label (torch_utils.py:214-221).  Both are linear and the head acts
// per voxel, so  warp(head(z)) = head(warp(z)) + bias * (sum of the in-bounds corner weights):  the forward kernel gathers
// the 8 corners of the 32-channel 16-bit feature map z (64 B per voxel, the size of the 16 fp32 logits it replaces),
// blends them and applies the selected head rows; the backward kernel gathers the logit gradient like
// warp_bwd_gather_kernel, and applies W^T to its 16 accumulators before storing.  The un-warped logits and their
// gradient (1 GB each per 8 x 128^3 pass) are never written or read: one launch each way instead of two.  Both gathers wait
// on L2 round trips with the vector ALU ~14 % busy (PMC), so the head's FMAs ride along.
// Numerics: same terms, associated differently (blend first, then the dot product) - fp32 rounding level.
constexpr int HW_CIN = 32, HW_NS = 16;

__device__ __forceinline__ float quad_xor_add(float v, int which) {     // v + (value of lane ^ 1) or (lane ^ 2): DPP quad_perm
  const int iv = __float_as_int(v);
  const int o = which == 1 ? __builtin_amdgcn_update_dpp(0, iv, 0xB1, 0xf, 0xf, false)
                           : __builtin_amdgcn_update_dpp(0, iv, 0x4E, 0xf, 0xf, false);
  return v + __int_as_float(o);
}

template <typename T>
__device__ __forceinline__ void unpack8_16(const uint4 &v, float *f) {
  const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) unpack2_16<T>(w[i], f[2 * i], f[2 * i + 1]);
}

template <typename T, bool ABL = false>
__global__ __launch_bounds__(256) void head_warp_fwd_kernel(const T *__restrict__ z, const float *__restrict__ theta,
                                                            const float *__restrict__ w, const float *__restrict__ bias,
                                                            const int *__restrict__ sel, int nsel, float *__restrict__ out,
                                                            int D, int H, int W, int algebra, int gx, int gy) {
  __shared__ float sw[HW_NS * HW_CIN];
  __shared__ float sb[HW_NS];
  for (int i = threadIdx.x; i < HW_NS * HW_CIN; i += 256) {
    const int k = i / HW_CIN;
    sw[i] = k < nsel ? w[(int64_t)(sel ? sel[k] : k) * HW_CIN + i % HW_CIN] : 0.f;
  }
  if (threadIdx.x < HW_NS) sb[threadIdx.x] = (int)threadIdx.x < nsel ? bias[sel ? sel[threadIdx.x] : threadIdx.x] : 0.f;
  __syncthreads();
  const int tile = (int)blockIdx.x;
  const int bx = tile % gx, by = (tile / gx) % gy, bz = tile / (gx * gy);
  const int d = bz % D, b = bz / D;
  const int64_t V = (int64_t)D * H * W;
  const T *zb = z + (int64_t)b * V * HW_CIN;
  const int i = bx * 256 + threadIdx.x;
  const int wv = i >> 2, g = i & 3;              // 4 lanes per voxel: channels 8 g .. 8 g + 7 in, classes 4 g .. 4 g + 3 out
  const bool live = wv < W;
  const int h0 = by * WARP_ROWS;
  for (int h = h0; h < min(h0 + WARP_ROWS, H); ++h) {
    float zbl[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float wsum = 0.f;
    if (live) {
      const Sample s = sample_pos(theta + b * 12, d, h, wv, D, H, W, D, H, W, algebra, DGTTA_PAD_ZEROS);
      const Corners cr = corners(s);
      // (round 6: all eight corner rows requested before the first is used - see head_warp_fwd_mfma_kernel)
      uint4 t[8];
      bool inb[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int xx = cr.x0 + (k & 1), yy = cr.y0 + ((k >> 1) & 1), zz = cr.z0 + (k >> 2);
        inb[k] = (unsigned)xx < (unsigned)W && (unsigned)yy < (unsigned)H && (unsigned)zz < (unsigned)D;
        const int xc = min(max(xx, 0), W - 1), yc = min(max(yy, 0), H - 1), zc = min(max(zz, 0), D - 1);
        // (abl: timing diagnostic DGTTA_WARP_ABL=1 - every corner read lands in a 2 x 4 x 16-voxel block that stays in L1)
        const int64_t sv = ABL ? (((int64_t)(zc & 1) * H + (yc & 3)) * W + (xc & 15)) : (((int64_t)zc * H + yc) * W + xc);
        t[k] = *reinterpret_cast<const uint4 *>(zb + sv * HW_CIN + g * 8);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const uint4 tk = inb[k] ? t[k] : make_uint4(0u, 0u, 0u, 0u);
        const float wk = inb[k] ? cr.w[k] : 0.f;
        float f[8];
        unpack8_16<T>(tk, f);
#pragma unroll
        for (int c = 0; c < 8; ++c) zbl[c] += f[c] * wk;
        wsum += wk;
      }
    }
    // partial logits over this lane's 8 channels, summed over the voxel's 4 lanes.  The weight slice is re-read from LDS
    // for every row on purpose: its offset is laundered, otherwise the compiler hoists the 32 reads out of the row loop
    // and keeps 128 registers live (190 VGPRs, 2 waves per SIMD - this kernel lives on occupancy)
    float pl[HW_NS];
    int goff = g * 8;
    asm volatile("" : "+v"(goff));
#pragma unroll
    for (int k = 0; k < HW_NS; ++k) {
      const float4 w0 = *reinterpret_cast<const float4 *>(sw + k * HW_CIN + goff);
      const float4 w1 = *reinterpret_cast<const float4 *>(sw + k * HW_CIN + goff + 4);
      float a = zbl[0] * w0.x;
      a = __builtin_fmaf(zbl[1], w0.y, a);
      a = __builtin_fmaf(zbl[2], w0.z, a);
      a = __builtin_fmaf(zbl[3], w0.w, a);
      a = __builtin_fmaf(zbl[4], w1.x, a);
      a = __builtin_fmaf(zbl[5], w1.y, a);
      a = __builtin_fmaf(zbl[6], w1.z, a);
      a = __builtin_fmaf(zbl[7], w1.w, a);
      a = quad_xor_add(a, 1);
      pl[k] = quad_xor_add(a, 2);
    }
    if (live && 4 * g < nsel) {
      float o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float v = g == 0 ? pl[j] : (g == 1 ? pl[4 + j] : (g == 2 ? pl[8 + j] : pl[12 + j]));
        o[j] = v + sb[4 * g + j] * wsum;
      }
      float *op = out + ((int64_t)b * V + ((int64_t)d * H + h) * W + wv) * nsel + 4 * g;
      *reinterpret_cast<float4 *>(op) = make_float4(o[0], o[1], o[2], o[3]);
    }
  }
}

// Round 6: the head's 32 -> 16 product on the fp32 MATRIX cores (v_mfma_f32_16x16x4_f32: exact f32, bit for bit a k-ordered
// fmaf chain).  The forward was instruction-bound (round 4: 1.03 -> 0.87 ms with every gather L1-resident): per voxel 128 FMAs,
// 32 LDS weight reads and 32 DPP adds for the head on top of the gather's blend.  A wave now takes 16 voxels x 4 channel groups
// (lane l: voxel l & 15, channels 8 (l >> 4) .. + 7 - the same 16-byte reads of the same rows as before); the blended channels
// ARE the B operand (k = lane >> 4: step s multiplies channel 8 g + s), the head's rows live in 8 registers per lane as the A
// operand (class l & 15), and D leaves classes 4 g .. 4 g + 3 of its voxel on the lane - the float4 the lane stores.  No LDS,
// no barrier; the vector ALU keeps the coordinate algebra and the blend, the matrix pipe (idle before) the head.
// Numerics: the same 32 products per logit, summed in the order (s = 0..7 even, then odd) x (g = 0..3) instead of per-lane chains
// and a quad butterfly: fp32 association level (tests: 2e-6 of the logit range against head-then-warp).
using hw_f32x4 = __attribute__((ext_vector_type(4))) float;

template <typename T>
__global__ __launch_bounds__(256) void head_warp_fwd_mfma_kernel(const T *__restrict__ z, const float *__restrict__ theta,
                                                                 const float *__restrict__ w, const float *__restrict__ bias,
                                                                 const int *__restrict__ sel, int nsel, float *__restrict__ out,
                                                                 int D, int H, int W, int algebra, int gx, int gy) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int vl = lane & 15, g = lane >> 4;
  float wa[8], b4[4];
  {
    const bool have = vl < nsel;
    const float *wr = w + (int64_t)(have ? (sel ? sel[vl] : vl) : 0) * HW_CIN + g * 8;
#pragma unroll
    for (int s = 0; s < 8; ++s) wa[s] = have ? wr[s] : 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = 4 * g + j;
      b4[j] = k < nsel ? bias[sel ? sel[k] : k] : 0.f;
    }
  }
  const int tile = (int)blockIdx.x;
  const int bx = tile % gx, by = (tile / gx) % gy, bz = tile / (gx * gy);
  const int d = bz % D, b = bz / D;
  const int64_t V = (int64_t)D * H * W;
  const T *zb = z + (int64_t)b * V * HW_CIN;
  const int wv = bx * 64 + wave * 16 + vl;
  const bool live = wv < W;
  const int h0 = by * WARP_ROWS;
  for (int h = h0; h < min(h0 + WARP_ROWS, H); ++h) {      // (wave-uniform: every lane reaches the MFMAs)
    float zbl[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float wsum = 0.f;
    if (live) {
      const Sample sp = sample_pos(theta + b * 12, d, h, wv, D, H, W, D, H, W, algebra, DGTTA_PAD_ZEROS);
      const Corners cr = corners(sp);
      // ALL EIGHT corner rows are requested before the first is used: a corner behind `if (in bounds)` is its own basic block
      // (load - s_waitcnt vmcnt(0) - blend), i.e. eight dependent L2 round trips per voxel row - what the kernel was waiting on
      // (round 3 PMC: 81 % of the wave cycles in s_waitcnt).  An out-of-bounds corner reads the clamped row and is zeroed.
      uint4 t[8];
      bool inb[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int xx = cr.x0 + (k & 1), yy = cr.y0 + ((k >> 1) & 1), zz = cr.z0 + (k >> 2);
        inb[k] = (unsigned)xx < (unsigned)W && (unsigned)yy < (unsigned)H && (unsigned)zz < (unsigned)D;
        const int xc = min(max(xx, 0), W - 1), yc = min(max(yy, 0), H - 1), zc = min(max(zz, 0), D - 1);
        t[k] = *reinterpret_cast<const uint4 *>(zb + ((int64_t)(zc * H + yc) * W + xc) * HW_CIN + g * 8);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const uint4 tk = inb[k] ? t[k] : make_uint4(0u, 0u, 0u, 0u);
        const float wk = inb[k] ? cr.w[k] : 0.f;
        float f[8];
        unpack8_16<T>(tk, f);
#pragma unroll
        for (int c = 0; c < 8; ++c) zbl[c] += f[c] * wk;       // (an excluded corner adds +0: the same bits as skipping it)
        wsum += wk;
      }
    }
    hw_f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};      // two chains: the dependent latency is 40 cycles
#pragma unroll
    for (int s = 0; s < 8; s += 2) {
      a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[s], zbl[s], a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[s + 1], zbl[s + 1], a1, 0, 0, 0);
    }
    if (live && 4 * g < nsel) {
      float *op = out + ((int64_t)b * V + ((int64_t)d * H + h) * W + wv) * nsel + 4 * g;
      *reinterpret_cast<float4 *>(op) = make_float4((a0[0] + a1[0]) + b4[0] * wsum, (a0[1] + a1[1]) + b4[1] * wsum,
                                                    (a0[2] + a1[2]) + b4[2] * wsum, (a0[3] + a1[3]) + b4[3] * wsum);
    }
  }
}

template <typename T>
__device__ __forceinline__ uint4 pack8_16(const float *f) {
  return make_uint4(pack2_16<T>(f[0], f[1]), pack2_16<T>(f[2], f[3]), pack2_16<T>(f[4], f[5]), pack2_16<T>(f[6], f[7]));
}

// backward: one thread owns one voxel of the feature-map lattice; candidate search and accumulation exactly as
// warp_bwd_gather_kernel<16, true> (same order), then d16 = 16-bit copy of the gathered logit gradient (operand of the
// head's MFMA weight gradient), gz = W^T acc in the network's storage type, and the block's partial sums for the bias gradient
// Round 6, G16: the logit gradient arrives in the network's 16-bit storage type (written so by dgtta_softdice_bwd_t): half the
// bytes per gathered candidate (1.89 instead of 2.13 ms per 8 x 128^3 launch); acc += float(g16) * weight in the same order.
// Measured beside it and NOT kept (profiles/r06_ab.txt): W^T on the fp32 matrix cores through a class-major LDS slab (same
// bits, but 82 instead of 80 registers = 5 instead of 6 waves per SIMD: 1.96 ms) and a chunked gather with 2 or 3 lines'
// candidate rows in flight (2.14 / 2.17 ms: the candidate search is the cost, not the round trips - more slots, more search).
template <typename T, bool ABL = false, bool G16 = false>
__global__ __launch_bounds__(256, 6) void head_warp_bwd_kernel(const void *__restrict__ gdst_, const float *__restrict__ theta,
                                                            const float *__restrict__ w, const int *__restrict__ sel,
                                                            int nsel, T *__restrict__ gz, unsigned short *__restrict__ d16,
                                                            double *__restrict__ bias_partial, int D, int H, int W,
                                                            int algebra, int gx, int gy) {
  __shared__ float sw[HW_NS * HW_CIN];
  __shared__ InvMap s_im;
  __shared__ float sred[4][HW_NS];
  const float *gdst = (const float *)gdst_;
  const unsigned short *gdst16 = (const unsigned short *)gdst_;
  for (int i = threadIdx.x; i < HW_NS * HW_CIN; i += 256) {
    const int k = i / HW_CIN;
    sw[i] = k < nsel ? w[(int64_t)(sel ? sel[k] : k) * HW_CIN + i % HW_CIN] : 0.f;
  }
  const int64_t V = (int64_t)D * H * W;
  const int tilesZ = (D + 3) >> 2;
  const int tile = (int)blockIdx.x;
  const int bx = tile % gx, by = (tile / gx) % gy, bz = tile / (gx * gy);
  const int b = bz / tilesZ;
  const int x = bx * 16 + (threadIdx.x & 15), y = by * 4 + ((threadIdx.x >> 4) & 3), zc_ = (bz % tilesZ) * 4 + (threadIdx.x >> 6);
  const float *th = theta + b * 12;
  if (threadIdx.x == 0) s_im = inverse_map(th, D, H, W, D, H, W, algebra);
  __syncthreads();
  const InvMap im = s_im;           // the launcher has checked on the host that every map of the batch is accepted
  float acc[HW_NS];
#pragma unroll
  for (int q = 0; q < HW_NS; ++q) acc[q] = 0.f;
  const bool inside = x < W && y < H && zc_ < D;
  if (inside && im.ok) {
    const float rel[3] = {(float)x - im.s0[0], (float)y - im.s0[1], (float)zc_ - im.s0[2]};
    int lo[3], hi[3];
    const int dims[3] = {W, H, D};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float c = im.inv[a][0] * rel[0] + im.inv[a][1] * rel[1] + im.inv[a][2] * rel[2];
      const float slack = im.e[a] + 0.02f;
      lo[a] = max(0, (int)fmaxf(ceilf(c - slack), -1.0f));
      hi[a] = min(dims[a] - 1, (int)fminf(floorf(c + slack), (float)dims[a]));
    }
    float rcp0[3];
    bool bounds_w[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      bounds_w[j] = fabsf(im.m[j][0]) > 1e-6f;
      rcp0[j] = bounds_w[j] ? 1.0f / im.m[j][0] : 0.f;
    }
    // w range of the candidates on line (d, h): the three constraints |S_j(w,h,d) - u_j| < 1 bound w (linear model + slack)
    auto w_range = [&](int d, int h, int &w0, int &w1) {
      float wl = (float)lo[0], wh = (float)hi[0];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const float r = -rel[j] + im.m[j][1] * (float)h + im.m[j][2] * (float)d;
        if (bounds_w[j]) {
          const float a = (-1.02f - r) * rcp0[j], bq = (1.02f - r) * rcp0[j];
          wl = fmaxf(wl, fminf(a, bq));
          wh = fminf(wh, fmaxf(a, bq));
        } else if (fabsf(r) > 1.02f) {
          wh = wl - 1.0f;
        }
      }
      w0 = (int)ceilf(wl);
      w1 = (int)floorf(wh);
    };
    // weight of this source voxel in the sample of dst voxel (wq, yc, zc); false when it is none of the sample's 8 corners
    auto corner_weight = [&](int wq, float yc, float zc, float &wt) -> bool {
      const Sample s = sample_from_base(th, base_coord(wq, W), yc, zc, D, H, W, algebra, DGTTA_PAD_ZEROS);
      const float fx = floorf(s.ix), fy = floorf(s.iy), fz = floorf(s.iz);
      const float dx = (float)x - fx, dy = (float)y - fy, dz = (float)zc_ - fz;
      const float wx = dx == 0.f ? (fx + 1.0f) - s.ix : s.ix - fx;
      const float wy = dy == 0.f ? (fy + 1.0f) - s.iy : s.iy - fy;
      const float wz = dz == 0.f ? (fz + 1.0f) - s.iz : s.iz - fz;
      wt = wx * wy * wz;
      return (dx == 0.f || dx == 1.f) && (dy == 0.f || dy == 1.f) && (dz == 0.f || dz == 1.f);
    };
    for (int d = lo[2]; d <= hi[2]; ++d) {
      const float zc = base_coord(d, D);
      for (int h = lo[1]; h <= hi[1]; ++h) {
        const float yc = base_coord(h, H);
        int w0, w1;
        w_range(d, h, w0, w1);
        for (int wq = w0; wq <= w1; ++wq) {
          float wt;
          if (!corner_weight(wq, yc, zc, wt)) continue;
          const int64_t grow = (ABL ? (((int64_t)(d & 1) * H + (h & 3)) * W + (wq & 15)) * nsel
                                    : ((int64_t)b * V + ((int64_t)d * H + h) * W + wq) * nsel);
          if (G16) {
            const unsigned short *gp = gdst16 + grow;
#pragma unroll
            for (int q = 0; q < HW_NS; q += 4) {
              if (q < nsel) {
                const uint2 t = *reinterpret_cast<const uint2 *>(gp + q);
                float f0, f1, f2, f3;
                unpack2_16<T>(t.x, f0, f1);
                unpack2_16<T>(t.y, f2, f3);
                acc[q] += f0 * wt;
                acc[q + 1] += f1 * wt;
                acc[q + 2] += f2 * wt;
                acc[q + 3] += f3 * wt;
              }
            }
          } else {
            const float *gp = gdst + grow;
#pragma unroll
            for (int q = 0; q < HW_NS; q += 4) {
              if (q < nsel) {
                const float4 g = *reinterpret_cast<const float4 *>(gp + q);
                acc[q] += g.x * wt;
                acc[q + 1] += g.y * wt;
                acc[q + 2] += g.z * wt;
                acc[q + 3] += g.w * wt;
              }
            }
          }
        }
      }
    }
  }
  if (inside) {
    const int64_t u = (int64_t)b * V + ((int64_t)zc_ * H + y) * W + x;
    // 16-bit copy of the gathered gradient, rows of nsel
    unsigned short *dp = d16 + u * nsel;
#pragma unroll
    for (int q = 0; q < HW_NS; q += 4)
      if (q < nsel) *reinterpret_cast<uint2 *>(dp + q) = make_uint2(pack2_16<T>(acc[q], acc[q + 1]), pack2_16<T>(acc[q + 2], acc[q + 3]));
    // gz = W^T acc, a quarter (8 channels) at a time to keep the register footprint of the gather phase
    T *gp = gz + u * HW_CIN;
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
      float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < HW_NS; ++k) {
        const float4 w0 = *reinterpret_cast<const float4 *>(sw + k * HW_CIN + qt * 8);
        const float4 w1 = *reinterpret_cast<const float4 *>(sw + k * HW_CIN + qt * 8 + 4);
        o[0] = __builtin_fmaf(acc[k], w0.x, o[0]);
        o[1] = __builtin_fmaf(acc[k], w0.y, o[1]);
        o[2] = __builtin_fmaf(acc[k], w0.z, o[2]);
        o[3] = __builtin_fmaf(acc[k], w0.w, o[3]);
        o[4] = __builtin_fmaf(acc[k], w1.x, o[4]);
        o[5] = __builtin_fmaf(acc[k], w1.y, o[5]);
        o[6] = __builtin_fmaf(acc[k], w1.z, o[6]);
        o[7] = __builtin_fmaf(acc[k], w1.w, o[7]);
      }
      *reinterpret_cast<uint4 *>(reinterpret_cast<unsigned short *>(gp) + qt * 8) = pack8_16<T>(o);
    }
  }
  // bias gradient: sum of the gathered gradient over the block (threads outside the volume hold zeros), fixed order
  if (bias_partial) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < HW_NS; ++k) {
      const float v = wave_sum(acc[k]);
      if (lane == 0) sred[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < HW_NS) {
      const double t = ((double)sred[0][threadIdx.x] + (double)sred[1][threadIdx.x]) +
                       ((double)sred[2][threadIdx.x] + (double)sred[3][threadIdx.x]);
      bias_partial[(int64_t)threadIdx.x * gridDim.x + blockIdx.x] = t;      // [class][block]: the finalize reads rows
    }
  }
}

__global__ __launch_bounds__(1024) void head_warp_bias_finalize_kernel(const double *__restrict__ partial, int nblk, int nsel,
                                                                       float *__restrict__ db, int accumulate) {
  // one workgroup per class over its row of block sums; fixed order: per-thread strided sums, butterfly per wave, waves in order
  __shared__ double red[16];
  const int k = blockIdx.x;
  const double *row = partial + (int64_t)k * nblk;
  double s = 0.0;
  for (int i = threadIdx.x; i < nblk; i += 1024) s += row[i];
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0 && k < nsel) {
    double t = 0.0;
    for (int wv = 0; wv < 16; ++wv) t += red[wv];
    db[k] = accumulate ? db[k] + (float)t : (float)t;
  }
}

// host copy of inverse_map()'s acceptance test (same arithmetic in float): the fused backward has no scatter fallback
bool host_map_ok(const float *th, int D, int H, int W) {
  const float m00 = th[0], m01 = th[1] * W / H, m02 = th[2] * W / D;
  const float m10 = th[4] * H / W, m11 = th[5], m12 = th[6] * H / D;
  const float m20 = th[8] * D / W, m21 = th[9] * D / H, m22 = th[10];
  const float c00 = m11 * m22 - m12 * m21, c01 = m12 * m20 - m10 * m22, c02 = m10 * m21 - m11 * m20;
  const float det = m00 * c00 + m01 * c01 + m02 * c02;
  if (det == 0.f) return false;
  const float idet = 1.0f / det;
  const float inv[3][3] = {{c00 * idet, (m02 * m21 - m01 * m22) * idet, (m01 * m12 - m02 * m11) * idet},
                           {c01 * idet, (m00 * m22 - m02 * m20) * idet, (m02 * m10 - m00 * m12) * idet},
                           {c02 * idet, (m01 * m20 - m00 * m21) * idet, (m00 * m11 - m01 * m10) * idet}};
  float vol = 1.f;
  for (int i = 0; i < 3; ++i) {
    const float e = fabsf(inv[i][0]) + fabsf(inv[i][1]) + fabsf(inv[i][2]);
    if (!(e < 1e30f)) return false;
    vol *= 2.0f * e + 1.0f;
  }
  for (int i = 0; i < 12; ++i)
    if (!(fabsf(th[i]) < 1e30f)) return false;
  return vol <= 256.0f;        // (the device test accepts up to 512: a margin for the differences in rounding)
}

// ---------------------------------------------------------------------------------------------------------------------
// Segmentation head fused with the Gaussian window accumulation (round 3, BASELINE config 3): a window's logits over ALL
// classes ([3P] nnU-Net predict_sliding_window_return_logits via dg_tta/tta/nnunet_utils.py:116-125) are 880 MB at
// 128^3 x 105 fp32 - written by the head and read back by window_accumulate_kernel.  Here a workgroup takes 64 consecutive
// voxels of a window row, evaluates all classes from the 64-byte feature rows (same FMA chain over the 32 channels and bias
// add as head_fwd_lds_kernel: identical logits), scales them by the voxel's Gaussian weight into an LDS tile and adds the
// tile to the accumulator as one contiguous run (the window row is contiguous in the volume along its last axis).
// Round 4: the run's accumulator values (27 per thread) are requested BEFORE the logits are evaluated and held in
// registers - the round-3 kernel read them one dependent load at a time after the barrier (a workgroup then spends ~27
// memory round trips per tile: 0.79 ms per window where the HBM traffic of 1.9 GB asks for 0.35 ms).  ACC: storage type of
// the accumulator - float, or f16_t (what nnU-Net's predictor keeps, 2.2.1 `predicted_logits` dtype torch.half); the sum is
// formed in fp32 either way and rounded once per window on the way back.
constexpr int HA_MAXC = 112;
constexpr int HA_RUN = (64 * HA_MAXC + 255) / 256;      // accumulator values per thread and tile (fp32 storage)
constexpr int HA_RUN2 = (64 * HA_MAXC / 2 + 1 + 255) / 256;   // 32-bit words per thread and tile (fp16 storage)

// One tile's run of the accumulator (n values from ap) held in registers between the early loads and the late stores.
// fp32 storage: thread t owns values t, t + 256, ...  fp16 storage: the run is addressed as 32-bit words from the aligned
// address at or below ap (a run starts on an odd half when its voxel offset x 105 classes is odd); a word that straddles
// the run's first or last half is accessed as that one half only - the other half belongs to the neighbouring run, which
// another workgroup may be updating.
template <typename ACC>
struct AccRun;
template <>
struct AccRun<float> {
  float r[HA_RUN];
  __device__ __forceinline__ void load(const float *ap, int n) {
#pragma unroll
    for (int j = 0; j < HA_RUN; ++j) {
      const int i = (int)threadIdx.x + 256 * j;
      r[j] = i < n ? ap[i] : 0.f;
    }
  }
  __device__ __forceinline__ void add_store(float *ap, int n, const float *tile) {
#pragma unroll
    for (int j = 0; j < HA_RUN; ++j) {
      const int i = (int)threadIdx.x + 256 * j;
      if (i < n) ap[i] = r[j] + tile[i];
    }
  }
};
template <>
struct AccRun<f16_t> {
  unsigned r[HA_RUN2];
  __device__ __forceinline__ void load(const f16_t *ap, int n) {
    const int h0 = (int)(((uintptr_t)ap >> 1) & 1);
    const unsigned short *hp = reinterpret_cast<const unsigned short *>(ap);
#pragma unroll
    for (int j = 0; j < HA_RUN2; ++j) {
      const int e0 = 2 * ((int)threadIdx.x + 256 * j) - h0;        // first half of the word, relative to ap
      const bool v0 = e0 >= 0 && e0 < n, v1 = e0 + 1 < n;
      unsigned v = 0;
      if (v0 && v1) v = *reinterpret_cast<const unsigned *>(hp + e0);
      else if (v0) v = hp[e0];
      else if (v1) v = (unsigned)hp[e0 + 1] << 16;
      r[j] = v;
    }
  }
  __device__ __forceinline__ void add_store(f16_t *ap, int n, const float *tile) {
    const int h0 = (int)(((uintptr_t)ap >> 1) & 1);
    unsigned short *hp = reinterpret_cast<unsigned short *>(ap);
#pragma unroll
    for (int j = 0; j < HA_RUN2; ++j) {
      const int e0 = 2 * ((int)threadIdx.x + 256 * j) - h0;
      const bool v0 = e0 >= 0 && e0 < n, v1 = e0 + 1 < n;
      const unsigned short lo = v0 ? f32_to_f16(f16_to_f32((unsigned short)(r[j] & 0xffffu)) + tile[e0]) : (unsigned short)0;
      const unsigned short hi = v1 ? f32_to_f16(f16_to_f32((unsigned short)(r[j] >> 16)) + tile[e0 + 1]) : (unsigned short)0;
      if (v0 && v1) *reinterpret_cast<unsigned *>(hp + e0) = (unsigned)lo | ((unsigned)hi << 16);
      else if (v0) hp[e0] = lo;
      else if (v1) hp[e0 + 1] = hi;
    }
  }
};

template <typename T, typename ACC>
__global__ __launch_bounds__(256) void head_accumulate_fma_kernel(const T *__restrict__ z, const float *__restrict__ w,
                                                              const float *__restrict__ bias, const float *__restrict__ gauss,
                                                              ACC *__restrict__ acc, float *__restrict__ nsum, int C, int PD,
                                                              int PH, int PW, int X, int Y, int Z, int x0, int y0, int z0,
                                                              int abl) {
  extern __shared__ float hsm[];
  float *tile = hsm;                        // [64][C]
  // the head's weights are the same for every lane of a wave (a wave = 64 voxels x one class at a time): they are read
  // through the scalar cache (13 KB at 105 classes; constant address space + a wave-uniform class index) and enter the
  // FMAs as scalar operands.  Round 3 staged them in LDS: a 16-byte broadcast read still moves 1 KB per wave, and 216 of
  // them per tile and wave kept the LDS pipe busier than HBM (0.42 of the 0.79 ms per window)
  typedef const __attribute__((address_space(4))) float *cptr_t;
  const cptr_t wc = (cptr_t)w, bc = (cptr_t)bias;
  const int runs = (PW + 63) >> 6, nblk = PD * PH * runs;
  const int vox = threadIdx.x & 63;
  const int grp = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));       // 4 class groups per voxel
  for (int blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const int row = blk / runs, pw0 = (blk % runs) * 64;
    const int pd = row / PH, ph = row % PH;
    const int nv = PW - pw0 < 64 ? PW - pw0 : 64;
    const int64_t p = ((int64_t)pd * PH + ph) * PW + pw0 + vox;   // voxel inside the window
    const int64_t vg = ((int64_t)(x0 + pd) * Y + (y0 + ph)) * Z + (z0 + pw0);
    ACC *ap = acc + vg * C;
    const int n = nv * C;
    AccRun<ACC> run;
    if (abl != 2) run.load(ap, n);           // (abl: timing diagnostics, DGTTA_HA_ABL - 1: no logits, 2: no accumulator traffic)
    float ns = 0.f, gs = 0.f;
    if (nsum && (int)threadIdx.x < nv) {
      ns = nsum[vg + threadIdx.x];
      gs = gauss[((int64_t)pd * PH + ph) * PW + pw0 + threadIdx.x];
    }
    __syncthreads();                                              // previous tile consumed
    if (vox < nv && abl != 1) {
      float xr[HW_CIN];
      const uint4 *zr = reinterpret_cast<const uint4 *>(z + p * HW_CIN);
#pragma unroll
      for (int g = 0; g < 4; ++g) unpack8_16<T>(zr[g], xr + 8 * g);
      const float gq = gauss[p];
      for (int k = grp; k < C; k += 4) {
        float a = 0.f;
#pragma unroll
        for (int ci = 0; ci < HW_CIN; ++ci) a = __builtin_fmaf(xr[ci], wc[k * HW_CIN + ci], a);
        tile[vox * C + k] = (a + bc[k]) * gq;
      }
    }
    __syncthreads();
    if (abl != 2) run.add_store(ap, n, tile);
    if (nsum && (int)threadIdx.x < nv) nsum[vg + threadIdx.x] = ns + gs;
  }
}

// The same on the matrix cores (round 4, the default; DGTTA_HA_MFMA=0 selects the FMA chain above).  Timed alone on one
// 128^3 x 105 window the FMA chain needs 0.31 ms for the logits and the accumulator traffic 0.35 ms (fp32) / 0.20 ms (fp16):
// the vector ALU, not HBM, sets the pace.  Here a wave takes 16 voxels: their feature rows are the B operand of
// v_mfma_f32_16x16x32 as they lie in memory (lane = voxel x 8-channel chunk, one 16-byte load), the head's weights are the A
// operand, held in registers for the life of the workgroup, and D[class][voxel] leaves 4 consecutive classes of one voxel in
// each lane.  The weights stay fp32-exact: every fp32 weight is split into three 16-bit terms (hi + mid + lo; 3 x 8 bits of
// bf16 significand = the 24 of fp32, exact; fp16 terms carry 3 x 11 bits but stop at 2^-24 in magnitude), 16-bit activations
// times 16-bit terms are exact in the fp32 accumulator, so the logits differ from the FMA chain by the order of the 32-term
// sum only (tests: <= a few 1e-7 of the row's magnitude, same labels outside float ties).
typedef __attribute__((ext_vector_type(8))) __bf16 hw_bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 hw_f16x8_t;
typedef __attribute__((ext_vector_type(4))) float hw_f32x4_t;
template <typename T>
__device__ __forceinline__ hw_f32x4_t hw_mfma(const uint4 &a, const uint4 &b, hw_f32x4_t acc);
template <>
__device__ __forceinline__ hw_f32x4_t hw_mfma<bf16_t>(const uint4 &a, const uint4 &b, hw_f32x4_t acc) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(hw_bf16x8_t, a), __builtin_bit_cast(hw_bf16x8_t, b), acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ hw_f32x4_t hw_mfma<f16_t>(const uint4 &a, const uint4 &b, hw_f32x4_t acc) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(hw_f16x8_t, a), __builtin_bit_cast(hw_f16x8_t, b), acc, 0, 0, 0);
}
template <typename T>
__device__ __forceinline__ float from16(unsigned short h);
template <>
__device__ __forceinline__ float from16<bf16_t>(unsigned short h) { return bf16_to_f32(h); }
template <>
__device__ __forceinline__ float from16<f16_t>(unsigned short h) { return f16_to_f32(h); }

static_assert(HA_MAXC <= 8 * 16, "a wave takes class groups wv and wv + 4: at most 8 groups of 16 classes");
template <typename T, typename ACC>
__global__ __launch_bounds__(256) void head_accumulate_kernel(const T *__restrict__ z, const float *__restrict__ w,
                                                              const float *__restrict__ bias, const float *__restrict__ gauss,
                                                              ACC *__restrict__ acc, float *__restrict__ nsum, int C, int PD,
                                                              int PH, int PW, int X, int Y, int Z, int x0, int y0, int z0,
                                                              int abl) {
  extern __shared__ float hsm[];
  float *tile = hsm;                        // [64][C]
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int ln = lane & 15, lq = lane >> 4;
  // wave wv evaluates class groups wv and wv + 4 for all 64 voxels of a tile (a wave per voxel group would hold all 7
  // groups x 3 terms = 84 registers of weights and run at 2 waves per SIMD; here it is 24, at 4 waves per SIMD - the
  // accumulator loads in flight are what keeps HBM busy).  A operand: lane = (class ln of the group, channels 8 lq .. + 7)
  uint4 wa[2][3];
  float bq[2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int cg = wv + 4 * j;
    const int m = cg * 16 + ln;
    unsigned short t[3][8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float x = m < C ? w[m * HW_CIN + lq * 8 + i] : 0.f;
      const unsigned short h = f32_to_16<T>(x);
      const float r1 = x - from16<T>(h);
      const unsigned short mid = f32_to_16<T>(r1);
      const float r2 = r1 - from16<T>(mid);
      t[0][i] = h;
      t[1][i] = mid;
      t[2][i] = f32_to_16<T>(r2);
    }
#pragma unroll
    for (int q = 0; q < 3; ++q)
      wa[j][q] = make_uint4((unsigned)t[q][0] | ((unsigned)t[q][1] << 16), (unsigned)t[q][2] | ((unsigned)t[q][3] << 16),
                            (unsigned)t[q][4] | ((unsigned)t[q][5] << 16), (unsigned)t[q][6] | ((unsigned)t[q][7] << 16));
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = cg * 16 + lq * 4 + r;
      bq[j][r] = k < C ? bias[k] : 0.f;
    }
  }
  const int runs = (PW + 63) >> 6, nblk = PD * PH * runs;
  // Software pipeline over the workgroup's tiles: the accumulator run and the feature rows of tile t + 1 are requested
  // before tile t is evaluated, so that every workgroup keeps loads in flight while its waves are in the MFMA / LDS phase
  // (loads return in order: a wave that asked for its 27 accumulator values first and its feature rows second cannot start
  // on the logits before the whole run has arrived - without the pipeline the phases of a workgroup run strictly one after
  // the other, and the fp32 accumulator, whose 27 registers per thread cost a wave of occupancy, ran at 2.6-3.2 TB/s)
  struct TilePos {
    int nv, n;
    int64_t prow, vg;
  };
  auto tile_pos = [&](int blk) {
    const int row = blk / runs, pw0 = (blk % runs) * 64;
    const int pd = row / PH, ph = row % PH;
    TilePos t;
    t.nv = PW - pw0 < 64 ? PW - pw0 : 64;
    t.n = t.nv * C;
    t.prow = ((int64_t)pd * PH + ph) * PW + pw0;                  // first voxel of the run inside the window
    t.vg = ((int64_t)(x0 + pd) * Y + (y0 + ph)) * Z + (z0 + pw0); // ... and inside the volume
    return t;
  };
  AccRun<ACC> run, run_n;
  uint4 zb[4], zb_n[4];                                           // B operand: lane = (voxel 16 vgp + ln, channels 8 lq .. + 7)
  float gq[4], gq_n[4];
  float ns = 0.f, gs = 0.f, ns_n = 0.f, gs_n = 0.f;
  auto fetch = [&](const TilePos &t, AccRun<ACC> &rn, uint4 *zz, float *gg, float &nss, float &gss) {
    if (abl != 2) rn.load(acc + t.vg * C, t.n);
    nss = 0.f;
    gss = 0.f;
    if (nsum && (int)threadIdx.x < t.nv) {
      nss = nsum[t.vg + threadIdx.x];
      gss = gauss[t.prow + threadIdx.x];
    }
#pragma unroll
    for (int vgp = 0; vgp < 4; ++vgp) {
      const int vox = vgp * 16 + ln;
      zz[vgp] = make_uint4(0u, 0u, 0u, 0u);
      gg[vgp] = 0.f;
      if (vox < t.nv) {
        zz[vgp] = *reinterpret_cast<const uint4 *>(z + (t.prow + vox) * HW_CIN + lq * 8);
        gg[vgp] = gauss[t.prow + vox];
      }
    }
  };
  int blk = blockIdx.x;
  if (blk < nblk) fetch(tile_pos(blk), run, zb, gq, ns, gs);
  for (; blk < nblk; blk += gridDim.x) {
    const TilePos t = tile_pos(blk);
    const bool more = blk + (int)gridDim.x < nblk;
    if (more) fetch(tile_pos(blk + gridDim.x), run_n, zb_n, gq_n, ns_n, gs_n);
    __syncthreads();                                              // previous tile consumed
    if (abl != 1) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int cg = wv + 4 * j;
        if (cg * 16 < C) {                                        // wave-uniform
#pragma unroll
          for (int vgp = 0; vgp < 4; ++vgp) {
            hw_f32x4_t d = {0.f, 0.f, 0.f, 0.f};
            d = hw_mfma<T>(wa[j][2], zb[vgp], d);
            d = hw_mfma<T>(wa[j][1], zb[vgp], d);
            d = hw_mfma<T>(wa[j][0], zb[vgp], d);
            const int vox = vgp * 16 + ln;
            if (vox < t.nv) {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int k = cg * 16 + lq * 4 + r;
                if (k < C) tile[vox * C + k] = (d[r] + bq[j][r]) * gq[vgp];
              }
            }
          }
        }
      }
    }
    __syncthreads();
    if (abl != 2) run.add_store(acc + t.vg * C, t.n, tile);
    if (nsum && (int)threadIdx.x < t.nv) nsum[t.vg + threadIdx.x] = ns + gs;
    if (more) {
      run = run_n;
#pragma unroll
      for (int vgp = 0; vgp < 4; ++vgp) {
        zb[vgp] = zb_n[vgp];
        gq[vgp] = gq_n[vgp];
      }
      ns = ns_n;
      gs = gs_n;
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
    const int items = Wd * (C / 4);
    const int64_t nblk = (int64_t)cdiv(items, 256) * cdiv(Hd, WARP_ROWS) * Dd * B;
    if (v4 && interp_mode == DGTTA_INTERP_LINEAR && nblk < (1ll << 31)) {
      const int gx = cdiv(items, 256), gy = cdiv(Hd, WARP_ROWS);
      hipLaunchKernelGGL(warp_fwd_rows4_kernel, dim3((unsigned)nblk), dim3(256), 0, st, src, theta, dst, C, Ds, Hs, Ws, Dd, Hd, Wd,
                         src_ldc, dst_ldc, pad_mode, tta_grid_algebra, sub_const_dev, gx, gy);
    } else if (v4) {
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
  const int64_t Vd = (int64_t)Dd * Hd * Wd, Vs = (int64_t)Ds * Hs * Ws;
  if (ndhwc && pad_mode == DGTTA_PAD_ZEROS) {
    // owner-computes gather (no atomics, no memset); batches whose map it declines are handled by the two launches after
    const bool vec = (C % 4 == 0) && (dst_ldc % 4 == 0) && (src_ldc % 4 == 0) && ((uintptr_t)grad_dst % 16 == 0) &&
                     ((uintptr_t)grad_src % 16 == 0);
    const int gx = cdiv(Ws, 16) * cdiv(C, 16), gy = cdiv(Hs, 4);
    const int64_t nblk = (int64_t)gx * gy * cdiv(Ds, 4) * B;
    DG_REQUIRE(B <= 16 && nblk < (1ll << 31), DGTTA_ERR_UNSUPPORTED, "warp_bwd: need B <= 16 and fewer than 2^31 tiles");
    dim3 grid((unsigned)nblk);
    if (vec)
      hipLaunchKernelGGL((warp_bwd_gather_kernel<16, true>), grid, dim3(256), 0, st, grad_dst, theta, grad_src, C, Ds, Hs,
                         Ws, Dd, Hd, Wd, src_ldc, dst_ldc, tta_grid_algebra, gx, gy);
    else
      hipLaunchKernelGGL((warp_bwd_gather_kernel<16, false>), grid, dim3(256), 0, st, grad_dst, theta, grad_src, C, Ds,
                         Hs, Ws, Dd, Hd, Wd, src_ldc, dst_ldc, tta_grid_algebra, gx, gy);
    DG_CHECK_LAUNCH("warp_bwd_gather_kernel");
    hipLaunchKernelGGL(warp_bwd_zero_if_declined_kernel, dim3(256), dim3(256), 0, st, theta, grad_src, B, Ds, Hs, Ws, Dd,
                       Hd, Wd, tta_grid_algebra, Vs * src_ldc);
    DG_CHECK_LAUNCH("warp_bwd_zero_if_declined_kernel");
    const int64_t tot2 = (int64_t)B * Vd * C;
    hipLaunchKernelGGL((warp_bwd_kernel<1, true>), dim3(grid_for(tot2)), dim3(256), 0, st, grad_dst, theta, grad_src, C,
                       Ds, Hs, Ws, Dd, Hd, Wd, src_ldc, dst_ldc, pad_mode, tta_grid_algebra, 1, B, tot2);
    DG_CHECK_LAUNCH("warp_bwd_kernel");
    return DGTTA_OK;
  }
  // general path: zero grad_src, then scatter with fp32 atomics
  {
    const size_t nb = ndhwc ? (size_t)B * Vs * src_ldc * sizeof(float) : (size_t)B * C * Vs * sizeof(float);
    hipError_t e = hipMemsetAsync(grad_src, 0, nb, st);
    DG_REQUIRE(e == hipSuccess, DGTTA_ERR_LAUNCH, "warp_bwd: memset failed: %s", hipGetErrorString(e));
  }
  if (ndhwc) {
    // one lane per channel: a wave-instruction's atomics then cover whole 64-byte rows (C=16) instead of 16-byte pieces
    int64_t total = (int64_t)B * Vd * C;
    hipLaunchKernelGGL((warp_bwd_kernel<1, true>), dim3(grid_for(total)), dim3(256), 0, st, grad_dst, theta, grad_src,
                       C, Ds, Hs, Ws, Dd, Hd, Wd, src_ldc, dst_ldc, pad_mode, tta_grid_algebra, 0, B, total);
  } else {
    int64_t total = (int64_t)B * Vd;
    hipLaunchKernelGGL((warp_bwd_kernel<1, false>), dim3(grid_for(total)), dim3(256), 0, st, grad_dst, theta, grad_src, C,
                       Ds, Hs, Ws, Dd, Hd, Wd, src_ldc, dst_ldc, pad_mode, tta_grid_algebra, 0, B, total);
  }
  DG_CHECK_LAUNCH("warp_bwd_kernel");
  return DGTTA_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// fused head + inverse warp (see head_warp_fwd_kernel)
size_t head_wgrad_mfma_ws_bytes(int Cin, int nsel, int64_t rows);
int head_wgrad_mfma(const void *x, int ldx, const float *dout, int lddo, float *dw_sel, void *ws, size_t ws_bytes, int Cin,
                    int nsel, int64_t rows, int accumulate, int dtype, hipStream_t st, bool have_d16);

static bool head_warp_shape_ok(int Cin, int nsel, int dtype) {
  return Cin == HW_CIN && nsel >= 4 && nsel <= HW_NS && nsel % 4 == 0 && (dtype == DGTTA_BF16 || dtype == DGTTA_F16);
}

static size_t head_warp_bias_region(int B, int D, int H, int W) {
  const int64_t nblk = (int64_t)cdiv(W, 16) * cdiv(H, 4) * cdiv(D, 4) * B;
  return align_up((size_t)nblk * HW_NS * sizeof(double), 256);
}

extern "C" int dgtta_seghead_warp_supported(const float *h_theta, int B, int Cin, int nsel, int D, int H, int W, int dtype) {
  if (!h_theta || B <= 0 || B > 16 || D <= 0 || H <= 0 || W <= 0 || !head_warp_shape_ok(Cin, nsel, dtype)) return 0;
  if (head_wgrad_mfma_ws_bytes(Cin, nsel, (int64_t)B * D * H * W) == 0) return 0;
  for (int b = 0; b < B; ++b)
    if (!host_map_ok(h_theta + 12 * b, D, H, W)) return 0;
  return 1;
}

extern "C" size_t dgtta_seghead_warp_bwd_ws_bytes(int B, int Cin, int nsel, int D, int H, int W) {
  if (B <= 0 || Cin <= 0 || nsel <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
  const int64_t rows = (int64_t)B * D * H * W;
  const size_t wg = head_wgrad_mfma_ws_bytes(Cin, nsel, rows);
  if (wg == 0) return 0;                       // the MFMA weight-gradient plan does not apply: the fused path is not offered
  return head_warp_bias_region(B, D, H, W) + align_up(wg, 256);
}

extern "C" int dgtta_seghead_warp_fwd(const void *z, const float *w, const float *bias, const int *sel, int nsel,
                                      const float *theta, float *out, int B, int Cin, int D, int H, int W,
                                      int tta_grid_algebra, int dtype, void *stream) {
  DG_REQUIRE(z && w && bias && theta && out, DGTTA_ERR_BADARG, "seghead_warp_fwd: null pointer");
  DG_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0, DGTTA_ERR_BADARG, "seghead_warp_fwd: bad dims");
  DG_REQUIRE(head_warp_shape_ok(Cin, nsel, dtype), DGTTA_ERR_UNSUPPORTED,
             "seghead_warp_fwd: built for 32 input channels, 4/8/12/16 selected classes and 16-bit storage (Cin %d, nsel %d, "
             "dtype %d)", Cin, nsel, dtype);
  DG_REQUIRE(((uintptr_t)z & 15) == 0 && ((uintptr_t)out & 15) == 0, DGTTA_ERR_BADARG, "seghead_warp_fwd: unaligned operand");
  const int gx = cdiv(W * 4, 256), gy = cdiv(H, WARP_ROWS);
  const int64_t nblk = (int64_t)gx * gy * D * B;
  DG_REQUIRE(nblk < (1ll << 31), DGTTA_ERR_UNSUPPORTED, "seghead_warp_fwd: too many tiles");
#define HWF_LAUNCH(T, A)                                                                                                   \
  hipLaunchKernelGGL((head_warp_fwd_kernel<T, A>), dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, (const T *)z, \
                     theta, w, bias, sel, nsel, out, D, H, W, tta_grid_algebra, gx, gy)
  bool lab = false;
#ifdef DGTTA_DIAG
  if (DG_LAB(warp_abl) == '1') {      // timing model (every gather read L1-resident): results wrong by construction
    lab = true;
    if (dtype == DGTTA_BF16) HWF_LAUNCH(bf16_t, true);
    else HWF_LAUNCH(f16_t, true);
  }
#endif
  if (lab) {
  } else if (dgtta_switches().headwarp_mfma != '0') {      // round 6: the head on the fp32 matrix cores (DGTTA_HEADWARP_MFMA=0: the FMA chain)
    if (dtype == DGTTA_BF16)
      hipLaunchKernelGGL((head_warp_fwd_mfma_kernel<bf16_t>), dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream,
                         (const bf16_t *)z, theta, w, bias, sel, nsel, out, D, H, W, tta_grid_algebra, gx, gy);
    else
      hipLaunchKernelGGL((head_warp_fwd_mfma_kernel<f16_t>), dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream,
                         (const f16_t *)z, theta, w, bias, sel, nsel, out, D, H, W, tta_grid_algebra, gx, gy);
  } else if (dtype == DGTTA_BF16) {
    HWF_LAUNCH(bf16_t, false);
  } else {
    HWF_LAUNCH(f16_t, false);
  }
#undef HWF_LAUNCH
  DG_CHECK_LAUNCH("head_warp_fwd_kernel");
  return DGTTA_OK;
}

static int seghead_warp_bwd_impl(const void *z, const void *gout, int gout16, const float *theta, const float *h_theta,
                                 const float *w, const int *sel, int nsel, void *gz, float *dw_sel, float *db_sel, void *ws,
                                 size_t ws_bytes, int B, int Cin, int D, int H, int W, int tta_grid_algebra, int accumulate,
                                 int dtype, void *stream) {
  DG_REQUIRE(z && gout && theta && h_theta && w && gz && ws, DGTTA_ERR_BADARG, "seghead_warp_bwd: null pointer");
  DG_REQUIRE(B > 0 && B <= 16 && D > 0 && H > 0 && W > 0, DGTTA_ERR_BADARG, "seghead_warp_bwd: bad dims");
  DG_REQUIRE(head_warp_shape_ok(Cin, nsel, dtype), DGTTA_ERR_UNSUPPORTED, "seghead_warp_bwd: unsupported shape / dtype");
  DG_REQUIRE(((uintptr_t)gout & 15) == 0 && ((uintptr_t)gz & 15) == 0, DGTTA_ERR_BADARG, "seghead_warp_bwd: unaligned operand");
  const size_t need = dgtta_seghead_warp_bwd_ws_bytes(B, Cin, nsel, D, H, W);
  DG_REQUIRE(need > 0, DGTTA_ERR_UNSUPPORTED, "seghead_warp_bwd: voxel count must be a multiple of 128");
  DG_REQUIRE(ws_bytes >= need, DGTTA_ERR_WORKSPACE, "seghead_warp_bwd: workspace too small");
  for (int b = 0; b < B; ++b)
    DG_REQUIRE(host_map_ok(h_theta + 12 * b, D, H, W), DGTTA_ERR_UNSUPPORTED,
               "seghead_warp_bwd: sample %d's map is singular or strongly minifying (use the unfused path)", b);
  hipStream_t st = (hipStream_t)stream;
  const int64_t rows = (int64_t)B * D * H * W;
  double *bias_partial = (double *)ws;
  void *ws_main = (char *)ws + head_warp_bias_region(B, D, H, W);
  const size_t main_bytes = ws_bytes - head_warp_bias_region(B, D, H, W);
  unsigned short *d16 = (unsigned short *)ws_main;           // first region of head_wgrad_mfma's workspace
  const int gx = cdiv(W, 16), gy = cdiv(H, 4);
  const int64_t nblk = (int64_t)gx * gy * cdiv(D, 4) * B;
  DG_REQUIRE(nblk < (1ll << 31), DGTTA_ERR_UNSUPPORTED, "seghead_warp_bwd: too many tiles");
#define HWB_LAUNCH(T, A, G)                                                                                               \
  hipLaunchKernelGGL((head_warp_bwd_kernel<T, A, G>), dim3((unsigned)nblk), dim3(256), 0, st, gout, theta, w, sel, nsel, \
                     (T *)gz, d16, db_sel ? bias_partial : nullptr, D, H, W, tta_grid_algebra, gx, gy)
  bool lab = false;
#ifdef DGTTA_DIAG
  if (DG_LAB(warp_abl) == '1' && !gout16) {      // timing model: results wrong by construction
    lab = true;
    if (dtype == DGTTA_BF16) HWB_LAUNCH(bf16_t, true, false);
    else HWB_LAUNCH(f16_t, true, false);
  }
#endif
  if (lab) {
  } else if (dtype == DGTTA_BF16) {
    if (gout16) HWB_LAUNCH(bf16_t, false, true);
    else HWB_LAUNCH(bf16_t, false, false);
  } else {
    if (gout16) HWB_LAUNCH(f16_t, false, true);
    else HWB_LAUNCH(f16_t, false, false);
  }
#undef HWB_LAUNCH
  DG_CHECK_LAUNCH("head_warp_bwd_kernel");
  if (dw_sel) {
    const int rc = head_wgrad_mfma(z, Cin, nullptr, nsel, dw_sel, ws_main, main_bytes, Cin, nsel, rows, accumulate, dtype, st, true);
    DG_REQUIRE(rc == DGTTA_OK, rc, "seghead_warp_bwd: head weight gradient failed (%d)", rc);
  }
  if (db_sel) {
    hipLaunchKernelGGL(head_warp_bias_finalize_kernel, dim3(HW_NS), dim3(1024), 0, st, bias_partial, (int)nblk, nsel, db_sel,
                       accumulate);
    DG_CHECK_LAUNCH("head_warp_bias_finalize_kernel");
  }
  return DGTTA_OK;
}

extern "C" int dgtta_seghead_warp_bwd(const void *z, const float *gout, const float *theta, const float *h_theta,
                                      const float *w, const int *sel, int nsel, void *gz, float *dw_sel, float *db_sel,
                                      void *ws, size_t ws_bytes, int B, int Cin, int D, int H, int W, int tta_grid_algebra,
                                      int accumulate, int dtype, void *stream) {
  return seghead_warp_bwd_impl(z, gout, 0, theta, h_theta, w, sel, nsel, gz, dw_sel, db_sel, ws, ws_bytes, B, Cin, D, H, W,
                               tta_grid_algebra, accumulate, dtype, stream);
}

// the same with the gradient of the warped logits in the network's 16-bit storage type `dtype` (rows of nsel values, as
// dgtta_softdice_bwd_t writes them): half the bytes per gathered candidate
extern "C" int dgtta_seghead_warp_bwd_g16(const void *z, const void *gout16, const float *theta, const float *h_theta,
                                          const float *w, const int *sel, int nsel, void *gz, float *dw_sel, float *db_sel,
                                          void *ws, size_t ws_bytes, int B, int Cin, int D, int H, int W, int tta_grid_algebra,
                                          int accumulate, int dtype, void *stream) {
  return seghead_warp_bwd_impl(z, gout16, 1, theta, h_theta, w, sel, nsel, gz, dw_sel, db_sel, ws, ws_bytes, B, Cin, D, H, W,
                               tta_grid_algebra, accumulate, dtype, stream);
}

extern "C" int dgtta_seghead_window_accumulate_t(const void *z, const float *w, const float *bias, const float *gauss, void *acc,
                                                 float *nsum, int Cin, int C, int PD, int PH, int PW, int X, int Y, int Z, int x0,
                                                 int y0, int z0, int dtype, int acc_dtype, void *stream) {
  DG_REQUIRE(z && w && bias && gauss && acc, DGTTA_ERR_BADARG, "seghead_window_accumulate: null pointer");
  DG_REQUIRE(Cin == HW_CIN && C > 0 && C <= HA_MAXC && (dtype == DGTTA_BF16 || dtype == DGTTA_F16), DGTTA_ERR_UNSUPPORTED,
             "seghead_window_accumulate: built for 32 input channels, up to %d classes, 16-bit storage", HA_MAXC);
  DG_REQUIRE(acc_dtype == DGTTA_F32 || acc_dtype == DGTTA_F16, DGTTA_ERR_UNSUPPORTED,
             "seghead_window_accumulate: the accumulator is fp32 or fp16 (acc_dtype %d)", acc_dtype);
  DG_REQUIRE(PD > 0 && PH > 0 && PW > 0 && x0 >= 0 && y0 >= 0 && z0 >= 0 && x0 + PD <= X && y0 + PH <= Y && z0 + PW <= Z,
             DGTTA_ERR_BADARG, "seghead_window_accumulate: window outside the volume");
  DG_REQUIRE(((uintptr_t)z & 15) == 0, DGTTA_ERR_BADARG, "seghead_window_accumulate: unaligned feature map");
  const int64_t nblk = (int64_t)PD * PH * cdiv(PW, 64);
  DG_REQUIRE(nblk < (1ll << 31), DGTTA_ERR_UNSUPPORTED, "seghead_window_accumulate: too many rows");
  const size_t lds = (size_t)64 * C * sizeof(float);
  static DynLdsOnce once[8];
  const dim3 grid((unsigned)(nblk < 2048 ? nblk : 2048));
  hipStream_t st = (hipStream_t)stream;
  const int abl = DG_LAB(ha_abl) == '1' ? 1 : (DG_LAB(ha_abl) == '2' ? 2 : 0);      // timing models: diagnostic build only
  const bool fma = dgtta_switches().ha_mfma == '0';
#define HA_LAUNCH(IDX, T, ACC)                                                                                               \
  do {                                                                                                                       \
    const void *fn = fma ? (const void *)head_accumulate_fma_kernel<T, ACC> : (const void *)head_accumulate_kernel<T, ACC>;  \
    DG_REQUIRE(ensure_dyn_lds(once[IDX + (fma ? 4 : 0)], fn, (int)lds) == hipSuccess, DGTTA_ERR_LAUNCH,                      \
               "seghead_window_accumulate: cannot raise the dynamic LDS limit");                                             \
    if (fma)                                                                                                                 \
      hipLaunchKernelGGL((head_accumulate_fma_kernel<T, ACC>), grid, dim3(256), lds, st, (const T *)z, w, bias, gauss,       \
                         (ACC *)acc, nsum, C, PD, PH, PW, X, Y, Z, x0, y0, z0, abl);                                         \
    else                                                                                                                     \
      hipLaunchKernelGGL((head_accumulate_kernel<T, ACC>), grid, dim3(256), lds, st, (const T *)z, w, bias, gauss,           \
                         (ACC *)acc, nsum, C, PD, PH, PW, X, Y, Z, x0, y0, z0, abl);                                         \
  } while (0)
  if (dtype == DGTTA_BF16 && acc_dtype == DGTTA_F32) HA_LAUNCH(0, bf16_t, float);
  else if (dtype == DGTTA_BF16) HA_LAUNCH(1, bf16_t, f16_t);
  else if (acc_dtype == DGTTA_F32) HA_LAUNCH(2, f16_t, float);
  else HA_LAUNCH(3, f16_t, f16_t);
#undef HA_LAUNCH
  DG_CHECK_LAUNCH("head_accumulate_kernel");
  return DGTTA_OK;
}

extern "C" int dgtta_seghead_window_accumulate(const void *z, const float *w, const float *bias, const float *gauss, float *acc,
                                               float *nsum, int Cin, int C, int PD, int PH, int PW, int X, int Y, int Z, int x0,
                                               int y0, int z0, int dtype, void *stream) {
  return dgtta_seghead_window_accumulate_t(z, w, bias, gauss, acc, nsum, Cin, C, PD, PH, PW, X, Y, Z, x0, y0, z0, dtype, DGTTA_F32,
                                           stream);
}

// ---------------------------------------------------------------------------------------------------------------------
// Gaussian-weighted sliding-window accumulation (post-TTA ensemble inference; nnU-Net's
// predict_sliding_window_return_logits [3P nnunetv2==2.2.1], reached from dg_tta/tta/nnunet_utils.py:116-125,208-230):
//   acc[v0 + p][c] += patch[p][c] * gauss[p];   nsum[v0 + p] += gauss[p]        (voxel-major fp32 accumulators)
// One lane per (patch voxel, channel): rows of C floats are contiguous, windows overlap only between launches.
namespace {
template <typename ACC>
__global__ void window_accumulate_kernel(const float *__restrict__ patch, const float *__restrict__ gauss,
                                         ACC *__restrict__ acc, float *__restrict__ nsum, int C, int PD, int PH, int PW,
                                         int X, int Y, int Z, int x0, int y0, int z0, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int64_t p = i / C;
    const int pw = (int)(p % PW), ph = (int)((p / PW) % PH), pd = (int)(p / ((int64_t)PW * PH));
    const int64_t v = ((int64_t)(x0 + pd) * Y + (y0 + ph)) * Z + (z0 + pw);
    const float g = gauss[p];
    st_f<ACC>(acc + v * C + c, ld_f<ACC>(acc + v * C + c) + patch[i] * g);
    if (c == 0 && nsum) nsum[v] += g;
  }
}
}  // namespace

extern "C" int dgtta_window_accumulate_t(const float *patch, const float *gauss, void *acc, float *nsum, int C, int PD, int PH,
                                         int PW, int X, int Y, int Z, int x0, int y0, int z0, int acc_dtype, void *stream) {
  DG_REQUIRE(patch && gauss && acc, DGTTA_ERR_BADARG, "window_accumulate: null pointer");
  DG_REQUIRE(C > 0 && PD > 0 && PH > 0 && PW > 0 && x0 >= 0 && y0 >= 0 && z0 >= 0 && x0 + PD <= X && y0 + PH <= Y &&
                 z0 + PW <= Z,
             DGTTA_ERR_BADARG, "window_accumulate: window outside the volume");
  DG_REQUIRE(acc_dtype == DGTTA_F32 || acc_dtype == DGTTA_F16, DGTTA_ERR_UNSUPPORTED,
             "window_accumulate: the accumulator is fp32 or fp16 (acc_dtype %d)", acc_dtype);
  const int64_t total = (int64_t)PD * PH * PW * C;
  if (acc_dtype == DGTTA_F32)
    hipLaunchKernelGGL(window_accumulate_kernel<float>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, patch, gauss,
                       (float *)acc, nsum, C, PD, PH, PW, X, Y, Z, x0, y0, z0, total);
  else
    hipLaunchKernelGGL(window_accumulate_kernel<f16_t>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, patch, gauss,
                       (f16_t *)acc, nsum, C, PD, PH, PW, X, Y, Z, x0, y0, z0, total);
  DG_CHECK_LAUNCH("window_accumulate_kernel");
  return DGTTA_OK;
}

extern "C" int dgtta_window_accumulate(const float *patch, const float *gauss, float *acc, float *nsum, int C, int PD, int PH,
                                       int PW, int X, int Y, int Z, int x0, int y0, int z0, void *stream) {
  return dgtta_window_accumulate_t(patch, gauss, acc, nsum, C, PD, PH, PW, X, Y, Z, x0, y0, z0, DGTTA_F32, stream);
}
