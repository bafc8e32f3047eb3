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

// Workgroups are handed to the 8 XCDs (each with its own L2) round robin in launch order, and neighbouring tiles of these
// kernels share the rows / slices their trilinear footprints overlap in.  With the launch order as the tile order every
// XCD fetched its own copy of those rows from HBM; this maps launch index L to a tile index such that each XCD walks a
// CONTIGUOUS eighth of the tiles.  Measured (round 3, 8 x 128^3 x 16): no gain forward, 20 % SLOWER backward - the
// counters show both kernels waiting on L1 misses that hit in L2 (L2 hit rate 93 %), not on HBM - so it is opt-in
// (DGTTA_WARP_XCD=1) and the launch order stays the tile order.
__device__ __forceinline__ int xcd_contiguous_tile(int L, int n, int enable) {
  const int n8 = n & ~7;
  return (enable && L < n8) ? (L & 7) * (n8 >> 3) + (L >> 3) : L;
}

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
                                                              int src_ldc, int dst_ldc, int algebra, int gx, int gy,
                                                              int xcd) {
  const int64_t Vd = (int64_t)Dd * Hd * Wd, Vs = (int64_t)Ds * Hs * Ws;
  // workgroup = compact 16 x 4 x 4 tile of source voxels (small grad_dst footprint -> L1/L2 reuse of the 8x overlap).
  // (4 lanes per voxel with 4 channels each would make the row loads 4x denser per instruction, but repeats the
  // candidate search 4x and measured slower: 374 vs 206 us at 128^3 x 16.)
  const int tilesX = (Ws + 15) >> 4, tilesZ = (Ds + 3) >> 2;
  const int tile = xcd_contiguous_tile((int)blockIdx.x, (int)gridDim.x, xcd);
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

// Cooperative form of warp_bwd_gather_kernel<16, true> (round 3).  The owner-computes kernel above gives a lane ONE
// grad_src voxel with 16 channels: each of its grad_dst loads is four 16-byte pieces per lane at a 64-byte lane stride, so a
// wave instruction touches 64 cache lines for 1 KiB of data - the texture path, not HBM, set its 1.4 TB/s.  Here the wave
// splits the two jobs: (1) as OWNER, lane o still searches the candidate dst voxels of its source voxel (same arithmetic, same
// order) but only writes (dst voxel index, weight) pairs into a per-wave LDS list; (2) as LOADER, lane l takes channel
// slot l & 3 of the four owners 16 q + (l >> 2), q = 0..3 (one row of the 16 x 4 tile each), reads their list entries
// (broadcast reads) and accumulates 4 channels per owner: a wave load covers 16 voxels x 64 contiguous bytes, and the final
// stores are whole coalesced rows.  Per (voxel, channel) the terms and their order are those of the kernel above: the
// results are bit-identical (tests/test_gpu_ops.py).  Lists hold WB_KMAX entries; an owner with more candidates (strongly
// minifying maps) resumes its search in further rounds.
constexpr int WB_KMAX = 12;
__global__ __launch_bounds__(256) void warp_bwd_gather_coop_kernel(const float *__restrict__ gdst,
                                                                   const float *__restrict__ theta,
                                                                   float *__restrict__ gsrc, int C, int Ds, int Hs, int Ws,
                                                                   int Dd, int Hd, int Wd, int src_ldc, int dst_ldc,
                                                                   int algebra, int nt, int gx, int gy, int xcd) {
  const int64_t Vd = (int64_t)Dd * Hd * Wd, Vs = (int64_t)Ds * Hs * Ws;
  const int tilesX = (Ws + 15) >> 4, tilesZ = (Ds + 3) >> 2;
  const int tile = xcd_contiguous_tile((int)blockIdx.x, (int)gridDim.x, xcd);
  const int bx = tile % gx, by = (tile / gx) % gy, bz = tile / (gx * gy);
  const int c0 = (bx / tilesX) * 16;
  const int b = bz / tilesZ;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int x0 = (bx % tilesX) * 16, y0 = by * 4, z = (bz % tilesZ) * 4 + wave;
  const int x = x0 + (lane & 15), y = y0 + (lane >> 4);
  __shared__ InvMap s_im;
  __shared__ int2 s_list[4][WB_KMAX][64];
  __shared__ int s_cnt[4][64];
  // normalised base coordinates of the dst lattice (two IEEE divisions each): evaluated once per workgroup instead of once
  // per candidate; same function, same bits
  constexpr int BC_MAX = 512;
  __shared__ float s_bc[3][BC_MAX];
  const bool bc_tab = Wd <= BC_MAX && Hd <= BC_MAX && Dd <= BC_MAX;
  if (bc_tab) {
    for (int i = threadIdx.x; i < Wd; i += 256) s_bc[0][i] = base_coord(i, Wd);
    for (int i = threadIdx.x; i < Hd; i += 256) s_bc[1][i] = base_coord(i, Hd);
    for (int i = threadIdx.x; i < Dd; i += 256) s_bc[2][i] = base_coord(i, Dd);
  }
  const float *th = theta + b * 12;
  if (threadIdx.x == 0) s_im = inverse_map(th, Ds, Hs, Ws, Dd, Hd, Wd, algebra);
  __syncthreads();
  const InvMap im = s_im;
  if (!im.ok) return;   // the scatter kernels (launched next) take over (uniform per batch item)
  // ---- owner role: candidate search state
  bool done = !(x < Ws && y < Hs && z < Ds);
  const float rel[3] = {(float)x - im.s0[0], (float)y - im.s0[1], (float)z - im.s0[2]};
  int lo[3], hi[3];
  {
    const int dims[3] = {Wd, Hd, Dd};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float c = im.inv[a][0] * rel[0] + im.inv[a][1] * rel[1] + im.inv[a][2] * rel[2];
      const float slack = im.e[a] + 0.02f;
      lo[a] = max(0, (int)fmaxf(ceilf(c - slack), -1.0f));
      hi[a] = min(dims[a] - 1, (int)fminf(floorf(c + slack), (float)dims[a]));
    }
  }
  float rcp0[3];
  bool bounds_w[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    bounds_w[j] = fabsf(im.m[j][0]) > 1e-6f;
    rcp0[j] = bounds_w[j] ? 1.0f / im.m[j][0] : 0.f;
  }
  int d = lo[2], h = lo[1], wres = INT_MIN;      // next (d, h) line and, inside a line cut short by a full list, the next w
  if (lo[1] > hi[1] || lo[2] > hi[2]) done = true;
  // ---- loader role
  const int slot = lane & 3, oq = lane >> 2;
  const bool ch_ok = c0 + slot * 4 < C;
  float acc[4][4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[q][e] = 0.f;
  const float *gb = gdst + (int64_t)b * Vd * dst_ldc + c0 + slot * 4;
  for (;;) {
    int n = 0;
    while (!done && n < WB_KMAX) {
      const float zc = bc_tab ? s_bc[2][d] : base_coord(d, Dd), yc = bc_tab ? s_bc[1][h] : base_coord(h, Hd);
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
      int w = max((int)ceilf(wl), wres);
      for (; w <= w1; ++w) {
        const Sample sp = sample_from_base(th, bc_tab ? s_bc[0][w] : base_coord(w, Wd), yc, zc, Ds, Hs, Ws, algebra,
                                           DGTTA_PAD_ZEROS);
        const float fx = floorf(sp.ix), fy = floorf(sp.iy), fz = floorf(sp.iz);
        const float dx = (float)x - fx, dy = (float)y - fy, dz = (float)z - fz;
        if (!((dx == 0.f || dx == 1.f) && (dy == 0.f || dy == 1.f) && (dz == 0.f || dz == 1.f))) continue;
        if (n == WB_KMAX) break;      // list full: this candidate opens the next round
        const float wx = dx == 0.f ? (fx + 1.0f) - sp.ix : sp.ix - fx;
        const float wy = dy == 0.f ? (fy + 1.0f) - sp.iy : sp.iy - fy;
        const float wz = dz == 0.f ? (fz + 1.0f) - sp.iz : sp.iz - fz;
        s_list[wave][n][lane] = make_int2((d * Hd + h) * Wd + w, __float_as_int(wx * wy * wz));
        ++n;
      }
      if (w <= w1) {
        wres = w;
        break;
      }
      wres = INT_MIN;
      if (++h > hi[1]) {
        h = lo[1];
        if (++d > hi[2]) done = true;
      }
    }
    s_cnt[wave][lane] = n;
    int maxn = n;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) maxn = max(maxn, __shfl_xor(maxn, o, 64));
    // the list is written and read by the same wave: its LDS operations complete in order; the compiler must not move them
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (ch_ok) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int owner = 16 * q + oq;
        const int nq = s_cnt[wave][owner];
        // four list entries per step: their loads are in flight together (a load per step left the kernel waiting on
        // one L2 round trip per candidate); entries past the owner's count read a valid address with weight 0 * skipped
        for (int k0 = 0; k0 < maxn; k0 += 4) {
          int2 e[4];
          float4 g[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            e[u] = (k0 + u < nq) ? s_list[wave][k0 + u][owner] : make_int2(-1, 0);
            if (e[u].x >= 0) g[u] = *reinterpret_cast<const float4 *>(gb + (int64_t)e[u].x * dst_ldc);
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            if (e[u].x >= 0) {
              const float wt = __int_as_float(e[u].y);
              acc[q][0] += g[u].x * wt;
              acc[q][1] += g[u].y * wt;
              acc[q][2] += g[u].z * wt;
              acc[q][3] += g[u].w * wt;
            }
          }
        }
      }
    }
    if (__all(done)) break;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  if (ch_ok && z < Ds) {
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ox = x0 + oq, oy = y0 + q;
      if (ox < Ws && oy < Hs) {
        float *o = gsrc + ((int64_t)b * Vs + ((int64_t)z * Hs + oy) * Ws + ox) * src_ldc + c0 + slot * 4;
        const f32x4_t v = {acc[q][0], acc[q][1], acc[q][2], acc[q][3]};
        if (nt) __builtin_nontemporal_store(v, reinterpret_cast<f32x4_t *>(o));
        else *reinterpret_cast<f32x4_t *>(o) = v;
      }
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
                                                             int algebra, const float *__restrict__ sub_const, int nt,
                                                             int gx, int gy, int xcd) {
  const int cg = C >> 2;
  const float sub = sub_const ? sub_const[0] : 0.f;
  const int tile = xcd_contiguous_tile((int)blockIdx.x, (int)gridDim.x, xcd);
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
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int xx = cr.x0 + (k & 1), yy = cr.y0 + ((k >> 1) & 1), zz = cr.z0 + (k >> 2);
      if ((unsigned)xx < (unsigned)Ws && (unsigned)yy < (unsigned)Hs && (unsigned)zz < (unsigned)Ds) {
        const float4 t = *reinterpret_cast<const float4 *>(sb + (((int64_t)zz * Hs + yy) * Ws + xx) * src_ldc + g * 4);
        acc[0] += (t.x - sub) * cr.w[k];
        acc[1] += (t.y - sub) * cr.w[k];
        acc[2] += (t.z - sub) * cr.w[k];
        acc[3] += (t.w - sub) * cr.w[k];
      }
    }
    float *drow = dst + ((int64_t)b * Vd + ((int64_t)d * Hd + h) * Wd) * dst_ldc;
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    const f32x4_t ov = {acc[0] + sub, acc[1] + sub, acc[2] + sub, acc[3] + sub};
    // the output is consumed by another kernel after ~1 GB has gone by: a non-temporal store keeps it from displacing
    // the source rows the neighbouring output rows are about to read (DGTTA_WARP_NT=0: plain stores)
    if (nt) __builtin_nontemporal_store(ov, reinterpret_cast<f32x4_t *>(drow + (int64_t)w * dst_ldc + g * 4));
    else *reinterpret_cast<f32x4_t *>(drow + (int64_t)w * dst_ldc + g * 4) = ov;
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
                         src_ldc, dst_ldc, pad_mode, tta_grid_algebra, sub_const_dev,
                         (dgtta_switches().warp_nt == '1' && C >= 8) ? 1 : 0, gx, gy, dgtta_switches().warp_xcd == '1');
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
    const int xcd = dgtta_switches().warp_xcd == '1';
    // DGTTA_WARP_COOP=1: the cooperative owner / loader kernel (bit-identical results; measured slower)
    if (vec && dgtta_switches().warp_coop == '1')
      hipLaunchKernelGGL(warp_bwd_gather_coop_kernel, grid, dim3(256), 0, st, grad_dst, theta, grad_src, C, Ds, Hs, Ws, Dd,
                         Hd, Wd, src_ldc, dst_ldc, tta_grid_algebra, dgtta_switches().warp_nt == '1', gx, gy, xcd);
    else if (vec)
      hipLaunchKernelGGL((warp_bwd_gather_kernel<16, true>), grid, dim3(256), 0, st, grad_dst, theta, grad_src, C, Ds, Hs,
                         Ws, Dd, Hd, Wd, src_ldc, dst_ldc, tta_grid_algebra, gx, gy, xcd);
    else
      hipLaunchKernelGGL((warp_bwd_gather_kernel<16, false>), grid, dim3(256), 0, st, grad_dst, theta, grad_src, C, Ds,
                         Hs, Ws, Dd, Hd, Wd, src_ldc, dst_ldc, tta_grid_algebra, gx, gy, xcd);
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
