// Masked-softmax soft-Dice consistency loss, forward and backward, single pass each (HBM-bound).
// Replaces dg_tta/tta/tta.py:263-271 and soft_dice_loss (dg_tta/tta/torch_utils.py:90-104):
//   mask = (sum_c la > 0) * (sum_c lb > 0);  a = softmax_c(la)*mask;  b = softmax_c(lb)*mask
//   nom_c = mean_v(2ab); den_c = 0.5*mean_v((a+b)^2); dice = nom/den (no eps; all-zero guard -> 1)
//   loss = 1 - mean_{b, c>=start} dice
// Logits are voxel-major [B][V][ldc] fp32 (one row of C classes per voxel = one coalesced read per lane).
// Rows are moved between HBM and per-wave LDS tiles as contiguous runs (coalesced); lane = voxel inside the tile.
// fwd: per-lane class/voxel-group partial sums -> per-workgroup partials (double) -> fixed-order finalize.
// bwd: recomputes the softmax per voxel from the logits (cheaper than saving it) and applies the closed-form
//      gradient with per-class coefficients P,Q prepared by the finalize kernel.
// Algorithmic HBM bytes per voxel: fwd 2*C*4 read; bwd 2*C*4 read + 2*C*4 write.
#include "common.h"

namespace {

constexpr int MAXC = 128;
constexpr int MAXW = 4;                 // waves per workgroup (fewer when the tile does not fit)
constexpr size_t LDS_BUDGET = 96 * 1024;

// Every wave owns two LDS tiles [64 voxels][LDP] (branch a, branch b).  Global rows are read/written as one contiguous
// run of 64*ldc floats (fully coalesced); inside the tile lane = voxel and the odd row pitch LDP keeps the strided
// per-lane walks over the classes bank-conflict free.
__host__ __device__ inline int tile_pitch(int ldc) { return ldc | 1; }

__device__ __forceinline__ void tile_load(float *tile, const float *src, int count, int ldc, int LDP, int lane, bool vec) {
  if (vec) {
    const float4 *s4 = reinterpret_cast<const float4 *>(src);
    for (int i = lane; i < (count >> 2); i += 64) {
      float4 x = s4[i];
      int e = i << 2, v = e / ldc, ch = e - v * ldc;
      float *d = tile + v * LDP + ch;
      d[0] = x.x; d[1] = x.y; d[2] = x.z; d[3] = x.w;
    }
  } else {
    for (int e = lane; e < count; e += 64) {
      int v = e / ldc, ch = e - v * ldc;
      tile[v * LDP + ch] = src[e];
    }
  }
}

__device__ __forceinline__ void tile_store(const float *tile, float *dst, int count, int ldc, int LDP, int lane, bool vec) {
  if (vec) {
    float4 *d4 = reinterpret_cast<float4 *>(dst);
    for (int i = lane; i < (count >> 2); i += 64) {
      int e = i << 2, v = e / ldc, ch = e - v * ldc;
      const float *t = tile + v * LDP + ch;
      d4[i] = make_float4(t[0], t[1], t[2], t[3]);
    }
  } else {
    for (int e = lane; e < count; e += 64) {
      int v = e / ldc, ch = e - v * ldc;
      dst[e] = tile[v * LDP + ch];
    }
  }
}

// lane = voxel: replaces the row of logits by exp(x - max) in place; returns the exp-sum and the (sum_c x > 0) mask part
__device__ __forceinline__ float row_softmax_inplace(float *row, int C, float *mask_part) {
  float mx = row[0], s = row[0];
  for (int c = 1; c < C; ++c) {
    float v = row[c];
    mx = fmaxf(mx, v);
    s += v;
  }
  float se = 0.f;
  for (int c = 0; c < C; ++c) {
    float e = expf(row[c] - mx);
    row[c] = e;
    se += e;
  }
  *mask_part = s > 0.0f ? 1.0f : 0.0f;
  return se;
}

__global__ __launch_bounds__(MAXW * 64) void softdice_fwd_kernel(const float *__restrict__ la,
                                                                 const float *__restrict__ lb,
                                                                 double *__restrict__ partial, int C, int64_t V, int ldc,
                                                                 int vec) {
  extern __shared__ float smem[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, NW = blockDim.x >> 6;
  const int LDP = tile_pitch(ldc);
  float *ta = smem + (size_t)wv * 2 * 64 * LDP, *tb = ta + 64 * LDP;
  const int b = blockIdx.y;
  // reduction role of this lane: class c, voxel group g of `groups`
  const int groups = C <= 64 ? 64 / C : 1;
  const int NR = (C + 63) >> 6;
  const int rc = C <= 64 ? lane % C : lane, rg = C <= 64 ? lane / C : 0;
  const bool ron = C <= 64 ? lane < groups * C : true;
  float acc1[2] = {0.f, 0.f}, acc2[2] = {0.f, 0.f};
  const int64_t ntiles = (V + 63) >> 6;
  for (int64_t t0 = (int64_t)blockIdx.x * NW; t0 < ntiles; t0 += (int64_t)gridDim.x * NW) {
    const int64_t t = t0 + wv;
    const int64_t rem = V - t * 64;
    const int nvalid = rem <= 0 ? 0 : (rem < 64 ? (int)rem : 64);
    const int64_t off = ((int64_t)b * V + t * 64) * ldc;
    if (nvalid > 0) {
      tile_load(ta, la + off, nvalid * ldc, ldc, LDP, lane, vec);
      tile_load(tb, lb + off, nvalid * ldc, ldc, LDP, lane, vec);
    }
    __syncthreads();
    {
      float *ra = ta + lane * LDP, *rb = tb + lane * LDP;
      if (lane < nvalid) {
        float ma, mb;
        float sa = row_softmax_inplace(ra, C, &ma), sb = row_softmax_inplace(rb, C, &mb);
        const float m = ma * mb;
        for (int c = 0; c < C; ++c) {
          ra[c] = (ra[c] / sa) * m;
          rb[c] = (rb[c] / sb) * m;
        }
      } else {
        for (int c = 0; c < C; ++c) ra[c] = rb[c] = 0.f;
      }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int c = rc + r * 64;
      if (r < NR && ron && c < C) {
        float s1 = 0.f, s2 = 0.f;
        for (int v = rg; v < 64; v += groups) {
          float a = ta[v * LDP + c], bq = tb[v * LDP + c];
          s1 += (2.0f * a) * bq;
          float tt = a + bq;
          s2 += tt * tt;
        }
        acc1[r] += s1;
        acc2[r] += s2;
      }
    }
    __syncthreads();
  }
  // combine lanes (groups) and waves in a fixed order, in double
  float *red = smem;  // [NW][64][4]
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    red[((wv * 64 + lane) * 2 + r) * 2 + 0] = acc1[r];
    red[((wv * 64 + lane) * 2 + r) * 2 + 1] = acc2[r];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < C * 2; i += blockDim.x) {
    const int c = i >> 1, k = i & 1;
    double s = 0.0;
    for (int w = 0; w < NW; ++w) {
      if (C <= 64) {
        for (int g = 0; g < groups; ++g) s += (double)red[((w * 64 + g * C + c) * 2 + 0) * 2 + k];
      } else {
        s += (double)red[((w * 64 + (c & 63)) * 2 + (c >> 6)) * 2 + k];
      }
    }
    partial[(((int64_t)b * C + c) * 2 + k) * gridDim.x + blockIdx.x] = s;
  }
}

// one workgroup (16 waves): sums partials in fixed order, forms dice/loss and the backward coefficients
__global__ __launch_bounds__(1024) void softdice_finalize_kernel(const double *__restrict__ partial, int nblk, int B,
                                                                 int C, int64_t V, int start_class,
                                                                 float *__restrict__ dice, float *__restrict__ loss,
                                                                 float *__restrict__ coef, int guard_items) {
  __shared__ float s_nom[8 * MAXC], s_den[8 * MAXC];
  __shared__ float s_flag[8], s_loss;
  const int n = B * C;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int j = wv; j < 2 * n; j += nw) {          // j = (b*C + c)*2 + k, partial row of nblk doubles
    double s = 0.0;
    for (int q = lane; q < nblk; q += 64) s += partial[(int64_t)j * nblk + q];
    s = wave_sum_d(s);
    if (lane == 0) {
      if (j & 1) s_den[j >> 1] = 0.5f * (float)(s / (double)V);
      else s_nom[j >> 1] = (float)(s / (double)V);
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < B / guard_items) {       // one guard per group of guard_items batch items
    const int g0 = threadIdx.x * guard_items * C;
    float tot = 0.f;
    for (int i = 0; i < guard_items * C; ++i) tot += s_den[g0 + i];
    s_flag[threadIdx.x] = (tot == 0.0f) ? 1.f : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += blockDim.x)
    dice[i] = s_flag[(i / C) / guard_items] != 0.f ? 1.0f : s_nom[i] / s_den[i];
  __syncthreads();
  if (threadIdx.x == 0) {
    float acc = 0.f;
    int cnt = 0;
    for (int b = 0; b < B; ++b)
      for (int c = start_class; c < C; ++c) {
        acc += dice[b * C + c];
        ++cnt;
      }
    s_loss = 1.0f - acc / (float)cnt;
    loss[0] = s_loss;
  }
  const float invN = 1.0f / (float)(B * (C - start_class));
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    int c = i % C;
    float P = 0.f, Q = 0.f;
    if (s_flag[(i / C) / guard_items] == 0.f && c >= start_class) {
      float den = s_den[i], nom = s_nom[i];
      P = -invN * 2.0f / ((float)V * den);
      Q = invN * nom / (den * den * (float)V);
    }
    coef[2 * i + 0] = P;
    coef[2 * i + 1] = Q;
  }
}

__global__ __launch_bounds__(MAXW * 64) void softdice_bwd_kernel(const float *__restrict__ la,
                                                                 const float *__restrict__ lb, float *__restrict__ ga,
                                                                 float *__restrict__ gb, const float *__restrict__ coef,
                                                                 float scale_h, const float *__restrict__ scale_dev,
                                                                 int C, int64_t V, int ldc, int vec) {
  extern __shared__ float smem[];
  __shared__ float sc[MAXC * 2];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, NW = blockDim.x >> 6;
  const int LDP = tile_pitch(ldc);
  float *ta = smem + (size_t)wv * 2 * 64 * LDP, *tb = ta + 64 * LDP;
  const float scale = scale_dev ? scale_h * scale_dev[0] : scale_h;
  const int b = blockIdx.y;
  for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) sc[i] = coef[(int64_t)b * C * 2 + i];
  const int64_t ntiles = (V + 63) >> 6;
  for (int64_t t0 = (int64_t)blockIdx.x * NW; t0 < ntiles; t0 += (int64_t)gridDim.x * NW) {
    const int64_t t = t0 + wv;
    const int64_t rem = V - t * 64;
    const int nvalid = rem <= 0 ? 0 : (rem < 64 ? (int)rem : 64);
    const int64_t off = ((int64_t)b * V + t * 64) * ldc;
    if (nvalid > 0) {
      tile_load(ta, la + off, nvalid * ldc, ldc, LDP, lane, vec);
      tile_load(tb, lb + off, nvalid * ldc, ldc, LDP, lane, vec);
    }
    __syncthreads();
    if (lane < nvalid) {
      float *ra = ta + lane * LDP, *rb = tb + lane * LDP;
      float ma, mb;
      const float sa = row_softmax_inplace(ra, C, &ma), sb = row_softmax_inplace(rb, C, &mb);
      const float m = ma * mb;
      float dot_a = 0.f, dot_b = 0.f;
      for (int c = 0; c < C; ++c) {
        float pa = ra[c] / sa, pb = rb[c] / sb;
        float a = pa * m, bq = pb * m;
        float P = sc[2 * c], Q = sc[2 * c + 1];
        float tt = Q * (a + bq);
        dot_a += (P * bq + tt) * pa;
        dot_b += (P * a + tt) * pb;
      }
      for (int c = 0; c < C; ++c) {
        float pa = ra[c] / sa, pb = rb[c] / sb;
        float a = pa * m, bq = pb * m;
        float P = sc[2 * c], Q = sc[2 * c + 1];
        float tt = Q * (a + bq);
        ra[c] = scale * m * pa * ((P * bq + tt) - dot_a);
        rb[c] = scale * m * pb * ((P * a + tt) - dot_b);
      }
      for (int c = C; c < ldc; ++c) ra[c] = rb[c] = 0.f;   // padding columns of the gradient rows are zeroed
    }
    __syncthreads();
    if (nvalid > 0) {
      tile_store(ta, ga + off, nvalid * ldc, ldc, LDP, lane, vec);
      tile_store(tb, gb + off, nvalid * ldc, ldc, LDP, lane, vec);
    }
    __syncthreads();
  }
}


// ---- stand-alone soft_dice_loss(a, b) on probability maps (torch_utils.py:90-104): a, b [B][C][V] with element strides
// (sb, sc, sv), i.e. NCDHW (sv = 1) or channels-last (sc = 1).  One workgroup reduces a voxel chunk of one (b, c) plane;
// partials and the finalize kernel are shared with the fused loss (guard over the whole call, start_class 0).
__global__ __launch_bounds__(256) void softdice_probs_fwd_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                                 double *__restrict__ partial, int C, int64_t V,
                                                                 int64_t sb, int64_t sc, int64_t sv) {
  __shared__ float red[16];
  const int bc = blockIdx.y, bi = bc / C, c = bc - bi * C;
  const float *pa = a + (int64_t)bi * sb + (int64_t)c * sc, *pb = b + (int64_t)bi * sb + (int64_t)c * sc;
  const int64_t per = cdiv64(V, gridDim.x), v0 = (int64_t)blockIdx.x * per, v1 = v0 + per < V ? v0 + per : V;
  double s1 = 0.0, s2 = 0.0;
  for (int64_t c0 = v0; c0 < v1; c0 += 256 * 64) {        // float partials over <= 64 voxels per thread, then double
    float f1 = 0.f, f2 = 0.f;
    const int64_t c1 = c0 + 256 * 64 < v1 ? c0 + 256 * 64 : v1;
    for (int64_t v = c0 + threadIdx.x; v < c1; v += 256) {
      const float x = pa[v * sv], y = pb[v * sv];
      f1 += (2.0f * x) * y;
      const float t = x + y;
      f2 += t * t;
    }
    s1 += (double)f1;
    s2 += (double)f2;
  }
  const float r1 = block_sum((float)s1, red), r2 = block_sum((float)s2, red);
  if (threadIdx.x == 0) {
    partial[((int64_t)bc * 2 + 0) * gridDim.x + blockIdx.x] = (double)r1;
    partial[((int64_t)bc * 2 + 1) * gridDim.x + blockIdx.x] = (double)r2;
  }
}

// d dice[b,c] / d a_v = 2 b_v / (V den) - nom (a_v + b_v) / (V den^2)  (and a <-> b); coef = (P, Q) of the finalize kernel
// with invN = 1 / (B C): the factor is undone here and replaced by the upstream gradient gdice[b,c].
__global__ __launch_bounds__(256) void softdice_probs_bwd_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                                 const float *__restrict__ gdice,
                                                                 const float *__restrict__ coef, float *__restrict__ ga,
                                                                 float *__restrict__ gb, int B, int C, int64_t V,
                                                                 int64_t sb, int64_t sc, int64_t sv) {
  const int bc = blockIdx.y, bi = bc / C, c = bc - bi * C;
  const int64_t base = (int64_t)bi * sb + (int64_t)c * sc;
  const float g = gdice[bc] * (float)(B * C);
  const float P = -coef[2 * bc] * g, Q = -coef[2 * bc + 1] * g;
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < V; v += (int64_t)gridDim.x * 256) {
    const float x = a[base + v * sv], y = b[base + v * sv];
    const float t = Q * (x + y);
    ga[base + v * sv] = P * y + t;
    gb[base + v * sv] = P * x + t;
  }
}

// ---- 16 classes in rows of 16 (the TTA plan's C_opt at the bench shape), round 3: four lanes per voxel, one float4 of
// either branch per lane - every load and store instruction covers 16 voxels x 64 contiguous bytes, the softmax lives in
// registers (quad reductions by DPP), no LDS tile, no second pass.  Same formulas as the generic kernels above; the class
// sums of a voxel are taken as a quad tree instead of sequentially (the results differ in the last bits only), the
// per-class partial sums keep the layout the finalize kernel reads.  DGTTA_SOFTDICE16=0: the generic kernels.
__device__ __forceinline__ float quad_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false));
  return v;
}
__device__ __forceinline__ float quad_max(float v) {
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false)));
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false)));
  return v;
}

// masked softmax of this lane's 4 classes of one voxel: p[j] = exp(x_j - max) / sum_exp; returns the mask part (sum x > 0)
__device__ __forceinline__ float quad_softmax4(const float4 &x, float *p) {
  const float s = quad_sum((x.x + x.y) + (x.z + x.w));
  const float mx = quad_max(fmaxf(fmaxf(x.x, x.y), fmaxf(x.z, x.w)));
  const float e0 = expf(x.x - mx), e1 = expf(x.y - mx), e2 = expf(x.z - mx), e3 = expf(x.w - mx);
  const float se = quad_sum((e0 + e1) + (e2 + e3));
  p[0] = e0 / se;
  p[1] = e1 / se;
  p[2] = e2 / se;
  p[3] = e3 / se;
  return s > 0.0f ? 1.0f : 0.0f;
}


__global__ __launch_bounds__(256) void softdice_fwd16_kernel(const float *__restrict__ la, const float *__restrict__ lb,
                                                             double *__restrict__ partial, int64_t V) {
  __shared__ float red[4][4][8];      // [wave][class quarter][4 x (nom, den)]
  const int b = blockIdx.y, q = threadIdx.x & 3;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const float4 *pa = reinterpret_cast<const float4 *>(la + (int64_t)b * V * 16);
  const float4 *pb = reinterpret_cast<const float4 *>(lb + (int64_t)b * V * 16);
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  // a workgroup pass covers 64 voxels; passes are dealt to the workgroups round robin
  for (int64_t v0 = (int64_t)blockIdx.x * 64; v0 < V; v0 += (int64_t)gridDim.x * 64) {
    const int64_t v = v0 + (threadIdx.x >> 2);
    if (v < V) {         // (whole quads are in or out)
      const float4 xa = pa[v * 4 + q], xb = pb[v * 4 + q];
      float a[4], bq[4];
      const float m = quad_softmax4(xa, a) * quad_softmax4(xb, bq);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float aj = a[j] * m, bj = bq[j] * m;
        s1[j] += (2.0f * aj) * bj;
        const float tt = aj + bj;
        s2[j] += tt * tt;
      }
    }
  }
  // lanes with the same class quarter (lane & 3) hold different voxels: butterfly over the other lane bits, fixed order
#pragma unroll
  for (int j = 0; j < 4; ++j) {
#pragma unroll
    for (int o = 4; o < 64; o <<= 1) {
      s1[j] += __shfl_xor(s1[j], o, 64);
      s2[j] += __shfl_xor(s2[j], o, 64);
    }
  }
  if (lane < 4) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      red[wv][lane][2 * j] = s1[j];
      red[wv][lane][2 * j + 1] = s2[j];
    }
  }
  __syncthreads();
  if (threadIdx.x < 32) {       // (class c = 4 quarter + j, k): waves in index order, in double
    const int c = threadIdx.x >> 1, k = threadIdx.x & 1;
    double t = 0.0;
    for (int w = 0; w < 4; ++w) t += (double)red[w][c >> 2][2 * (c & 3) + k];
    partial[(((int64_t)b * 16 + c) * 2 + k) * gridDim.x + blockIdx.x] = t;
  }
}

// TO: type of the gradient rows - float, or (round 6) the network's 16-bit storage type: the fused head + warp backward then
// gathers half the bytes (dgtta_seghead_warp_bwd_g16); the values are the fp32 ones rounded once
template <typename TO>
__device__ __forceinline__ void store_grad4(TO *base, int64_t idx4, float x, float y, float z, float w);
template <>
__device__ __forceinline__ void store_grad4<float>(float *base, int64_t idx4, float x, float y, float z, float w) {
  reinterpret_cast<float4 *>(base)[idx4] = make_float4(x, y, z, w);
}
template <>
__device__ __forceinline__ void store_grad4<bf16_t>(bf16_t *base, int64_t idx4, float x, float y, float z, float w) {
  reinterpret_cast<uint2 *>(base)[idx4] = make_uint2(pack2_16<bf16_t>(x, y), pack2_16<bf16_t>(z, w));
}
template <>
__device__ __forceinline__ void store_grad4<f16_t>(f16_t *base, int64_t idx4, float x, float y, float z, float w) {
  reinterpret_cast<uint2 *>(base)[idx4] = make_uint2(pack2_16<f16_t>(x, y), pack2_16<f16_t>(z, w));
}

template <typename TO>
__global__ __launch_bounds__(256) void softdice_bwd16_kernel(const float *__restrict__ la, const float *__restrict__ lb,
                                                             TO *__restrict__ ga, TO *__restrict__ gb,
                                                             const float *__restrict__ coef, float scale_h,
                                                             const float *__restrict__ scale_dev, int64_t V) {
  const int b = blockIdx.y, q = threadIdx.x & 3;
  const float scale = scale_dev ? scale_h * scale_dev[0] : scale_h;
  float P[4], Q[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    P[j] = coef[((int64_t)b * 16 + 4 * q + j) * 2];
    Q[j] = coef[((int64_t)b * 16 + 4 * q + j) * 2 + 1];
  }
  const float4 *pa = reinterpret_cast<const float4 *>(la + (int64_t)b * V * 16);
  const float4 *pb = reinterpret_cast<const float4 *>(lb + (int64_t)b * V * 16);
  TO *oa = ga + (int64_t)b * V * 16, *ob = gb + (int64_t)b * V * 16;
  for (int64_t v0 = (int64_t)blockIdx.x * 64; v0 < V; v0 += (int64_t)gridDim.x * 64) {
    const int64_t v = v0 + (threadIdx.x >> 2);
    if (v < V) {
      const float4 xa = pa[v * 4 + q], xb = pb[v * 4 + q];
      float a[4], bq[4], ta[4], tb[4];
      const float m = quad_softmax4(xa, a) * quad_softmax4(xb, bq);
      float da = 0.f, db = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float aj = a[j] * m, bj = bq[j] * m;
        const float tt = Q[j] * (aj + bj);
        ta[j] = P[j] * bj + tt;
        tb[j] = P[j] * aj + tt;
        da += ta[j] * a[j];
        db += tb[j] * bq[j];
      }
      da = quad_sum(da);
      db = quad_sum(db);
      const float sm = scale * m;
      store_grad4<TO>(oa, v * 4 + q, sm * a[0] * (ta[0] - da), sm * a[1] * (ta[1] - da), sm * a[2] * (ta[2] - da), sm * a[3] * (ta[3] - da));
      store_grad4<TO>(ob, v * 4 + q, sm * bq[0] * (tb[0] - db), sm * bq[1] * (tb[1] - db), sm * bq[2] * (tb[2] - db),
                      sm * bq[3] * (tb[3] - db));
    }
  }
}

bool use16(const void *a, const void *b, const void *c, const void *d, int C, int ldc) {
  return C == 16 && ldc == 16 && (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)d) & 15) == 0 &&
         dgtta_switches().softdice16 != '0';
}

int waves_for(int ldc) {
  size_t per_wave = (size_t)2 * 64 * tile_pitch(ldc) * sizeof(float);
  int nw = (int)(LDS_BUDGET / per_wave);
  return nw > MAXW ? MAXW : (nw < 1 ? 1 : nw);
}

size_t lds_for(int ldc, int nw) {
  size_t tiles = (size_t)nw * 2 * 64 * tile_pitch(ldc) * sizeof(float);
  size_t red = (size_t)nw * 64 * 4 * sizeof(float);
  return tiles > red ? tiles : red;
}

int nblocks_for(int64_t V) {
  int64_t b = (V + 255) / 256;
  return (int)(b < 1024 ? b : 1024);
}

bool vec_ok(const void *a, const void *b, const void *c, const void *d, int ldc) {
  return ldc % 4 == 0 && (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)d) & 15) == 0;
}

int allow_big_lds() {      // per device, once (common.h: DynLdsOnce)
  static DynLdsOnce fwd_once, bwd_once;
  const hipError_t e1 = ensure_dyn_lds(fwd_once, (const void *)softdice_fwd_kernel, (int)LDS_BUDGET);
  const hipError_t e2 = ensure_dyn_lds(bwd_once, (const void *)softdice_bwd_kernel, (int)LDS_BUDGET);
  return (e1 == hipSuccess && e2 == hipSuccess) ? 0 : -1;
}

}  // namespace

// ws layout: [partials: B*nblk*C*2 double][coef: B*C*2 float]
extern "C" size_t dgtta_softdice_ws_bytes(int B, int C, int64_t V) {
  if (B <= 0 || C <= 0 || V <= 0) return 0;      // a size query of an empty problem (the launchers reject it with DGTTA_ERR_BADARG)
  return align_up((size_t)B * nblocks_for(V) * C * 2 * sizeof(double), 256) +
         align_up((size_t)B * C * 2 * sizeof(float), 256) + 256 /* scratch scalar of the stand-alone op */;
}

extern "C" int dgtta_softdice_fwd(const float *la, const float *lb, float *dice, float *loss, void *ws, size_t ws_bytes,
                                  int B, int C, int64_t V, int ldc, int start_class, int guard_items, void *stream) {
  DG_REQUIRE(la && lb && dice && loss && ws, DGTTA_ERR_BADARG, "softdice_fwd: null pointer");
  DG_REQUIRE(B > 0 && B <= 8 && C > 0 && C <= MAXC && V > 0 && ldc >= C, DGTTA_ERR_BADARG,
             "softdice_fwd: need 1<=B<=8, 1<=C<=%d, ldc>=C (B=%d C=%d ldc=%d)", MAXC, B, C, ldc);
  DG_REQUIRE(start_class >= 0 && start_class < C, DGTTA_ERR_BADARG, "softdice_fwd: bad start_class");
  DG_REQUIRE(guard_items >= 1 && B % guard_items == 0, DGTTA_ERR_BADARG, "softdice_fwd: guard_items must divide B");
  DG_REQUIRE(ws_bytes >= dgtta_softdice_ws_bytes(B, C, V), DGTTA_ERR_WORKSPACE, "softdice_fwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int nblk = nblocks_for(V);
  double *partial = (double *)ws;
  float *coef = (float *)((char *)ws + align_up((size_t)B * nblk * C * 2 * sizeof(double), 256));
  if (use16(la, lb, nullptr, nullptr, C, ldc)) {
    hipLaunchKernelGGL(softdice_fwd16_kernel, dim3(nblk, B), dim3(256), 0, st, la, lb, partial, V);
    DG_CHECK_LAUNCH("softdice_fwd16_kernel");
  } else {
    DG_REQUIRE(allow_big_lds() == 0, DGTTA_ERR_LAUNCH, "softdice: cannot raise the dynamic LDS limit");
    const int nw = waves_for(ldc);
    hipLaunchKernelGGL(softdice_fwd_kernel, dim3(nblk, B), dim3(nw * 64), lds_for(ldc, nw), st, la, lb, partial, C, V, ldc,
                       (int)vec_ok(la, lb, nullptr, nullptr, ldc));
    DG_CHECK_LAUNCH("softdice_fwd_kernel");
  }
  hipLaunchKernelGGL(softdice_finalize_kernel, dim3(1), dim3(1024), 0, st, partial, nblk, B, C, V, start_class, dice,
                     loss, coef, guard_items);
  DG_CHECK_LAUNCH("softdice_finalize_kernel");
  return DGTTA_OK;
}

extern "C" int dgtta_softdice_bwd(const float *la, const float *lb, float *grad_la, float *grad_lb, const void *ws,
                                  float grad_scale, const float *grad_scale_dev, int B, int C, int64_t V, int ldc,
                                  int start_class, void *stream) {
  DG_REQUIRE(la && lb && grad_la && grad_lb && ws, DGTTA_ERR_BADARG, "softdice_bwd: null pointer");
  DG_REQUIRE(B > 0 && B <= 8 && C > 0 && C <= MAXC && V > 0 && ldc >= C, DGTTA_ERR_BADARG, "softdice_bwd: bad dims");
  (void)start_class;
  hipStream_t st = (hipStream_t)stream;
  const int nblk = nblocks_for(V);
  const float *coef = (const float *)((const char *)ws + align_up((size_t)B * nblk * C * 2 * sizeof(double), 256));
  if (use16(la, lb, grad_la, grad_lb, C, ldc)) {
    int64_t g16 = (V + 63) / 64;
    if (g16 > 4096) g16 = 4096;
    hipLaunchKernelGGL(softdice_bwd16_kernel<float>, dim3((int)g16, B), dim3(256), 0, st, la, lb, grad_la, grad_lb, coef, grad_scale,
                       grad_scale_dev, V);
    DG_CHECK_LAUNCH("softdice_bwd16_kernel");
    return DGTTA_OK;
  }
  DG_REQUIRE(allow_big_lds() == 0, DGTTA_ERR_LAUNCH, "softdice: cannot raise the dynamic LDS limit");
  const int nw = waves_for(ldc);
  int64_t gb = (V + 255) / 256;
  if (gb > 2048) gb = 2048;
  hipLaunchKernelGGL(softdice_bwd_kernel, dim3((int)gb, B), dim3(nw * 64), lds_for(ldc, nw), st, la, lb, grad_la,
                     grad_lb, coef, grad_scale, grad_scale_dev, C, V, ldc, (int)vec_ok(la, lb, grad_la, grad_lb, ldc));
  DG_CHECK_LAUNCH("softdice_bwd_kernel");
  return DGTTA_OK;
}

// Gradient rows in a 16-bit storage type (DGTTA_BF16 / DGTTA_F16): the 16-class form only (C = ldc = 16, 8-byte aligned rows) -
// what the fused head + warp backward consumes (dgtta_seghead_warp_bwd_g16).  Same arithmetic as dgtta_softdice_bwd, each value
// rounded once on the way out.
extern "C" int dgtta_softdice_bwd_t(const float *la, const float *lb, void *grad_la, void *grad_lb, const void *ws,
                                    float grad_scale, const float *grad_scale_dev, int B, int C, int64_t V, int ldc,
                                    int start_class, int grad_dtype, void *stream) {
  if (grad_dtype == DGTTA_F32)
    return dgtta_softdice_bwd(la, lb, (float *)grad_la, (float *)grad_lb, ws, grad_scale, grad_scale_dev, B, C, V, ldc, start_class,
                              stream);
  DG_REQUIRE(la && lb && grad_la && grad_lb && ws, DGTTA_ERR_BADARG, "softdice_bwd_t: null pointer");
  DG_REQUIRE(B > 0 && B <= 8 && V > 0, DGTTA_ERR_BADARG, "softdice_bwd_t: bad dims");
  DG_REQUIRE(grad_dtype == DGTTA_BF16 || grad_dtype == DGTTA_F16, DGTTA_ERR_BADARG, "softdice_bwd_t: unknown gradient dtype %d", grad_dtype);
  DG_REQUIRE(C == 16 && ldc == 16 && (((uintptr_t)la | (uintptr_t)lb) & 15) == 0 && (((uintptr_t)grad_la | (uintptr_t)grad_lb) & 7) == 0,
             DGTTA_ERR_UNSUPPORTED, "softdice_bwd_t: 16-bit gradient rows are built for C = ldc = 16 and aligned operands (C %d, ldc %d)", C, ldc);
  (void)start_class;
  hipStream_t st = (hipStream_t)stream;
  const int nblk = nblocks_for(V);
  const float *coef = (const float *)((const char *)ws + align_up((size_t)B * nblk * C * 2 * sizeof(double), 256));
  int64_t g16 = (V + 63) / 64;
  if (g16 > 4096) g16 = 4096;
  if (grad_dtype == DGTTA_BF16)
    hipLaunchKernelGGL(softdice_bwd16_kernel<bf16_t>, dim3((int)g16, B), dim3(256), 0, st, la, lb, (bf16_t *)grad_la, (bf16_t *)grad_lb,
                       coef, grad_scale, grad_scale_dev, V);
  else
    hipLaunchKernelGGL(softdice_bwd16_kernel<f16_t>, dim3((int)g16, B), dim3(256), 0, st, la, lb, (f16_t *)grad_la, (f16_t *)grad_lb,
                       coef, grad_scale, grad_scale_dev, V);
  DG_CHECK_LAUNCH("softdice_bwd16_kernel");
  return DGTTA_OK;
}

extern "C" int dgtta_softdice_probs_fwd(const float *a, const float *b, float *dice, void *ws, size_t ws_bytes, int B, int C,
                                        int64_t V, int64_t stride_b, int64_t stride_c, int64_t stride_v, void *stream) {
  DG_REQUIRE(a && b && dice && ws, DGTTA_ERR_BADARG, "softdice_probs_fwd: null pointer");
  DG_REQUIRE(B > 0 && B <= 8 && C > 0 && C <= MAXC && V > 0, DGTTA_ERR_BADARG,
             "softdice_probs_fwd: need 1<=B<=8, 1<=C<=%d (B=%d C=%d)", MAXC, B, C);
  DG_REQUIRE(ws_bytes >= dgtta_softdice_ws_bytes(B, C, V), DGTTA_ERR_WORKSPACE, "softdice_probs_fwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int nblk = nblocks_for(V);
  double *partial = (double *)ws;
  float *coef = (float *)((char *)ws + align_up((size_t)B * nblk * C * 2 * sizeof(double), 256));
  float *scratch = coef + align_up((size_t)B * C * 2 * sizeof(float), 256) / sizeof(float);
  hipLaunchKernelGGL(softdice_probs_fwd_kernel, dim3(nblk, B * C), dim3(256), 0, st, a, b, partial, C, V, stride_b,
                     stride_c, stride_v);
  DG_CHECK_LAUNCH("softdice_probs_fwd_kernel");
  hipLaunchKernelGGL(softdice_finalize_kernel, dim3(1), dim3(1024), 0, st, partial, nblk, B, C, V, 0, dice, scratch, coef,
                     B);
  DG_CHECK_LAUNCH("softdice_finalize_kernel");
  return DGTTA_OK;
}

extern "C" int dgtta_softdice_probs_bwd(const float *a, const float *b, const float *grad_dice, float *grad_a, float *grad_b,
                                        const void *ws, int B, int C, int64_t V, int64_t stride_b, int64_t stride_c,
                                        int64_t stride_v, void *stream) {
  DG_REQUIRE(a && b && grad_dice && grad_a && grad_b && ws, DGTTA_ERR_BADARG, "softdice_probs_bwd: null pointer");
  DG_REQUIRE(B > 0 && B <= 8 && C > 0 && C <= MAXC && V > 0, DGTTA_ERR_BADARG, "softdice_probs_bwd: bad dims");
  hipStream_t st = (hipStream_t)stream;
  const int nblk = nblocks_for(V);
  const float *coef = (const float *)((const char *)ws + align_up((size_t)B * nblk * C * 2 * sizeof(double), 256));
  int64_t gx = (V + 255) / 256;
  if (gx > 512) gx = 512;
  hipLaunchKernelGGL(softdice_probs_bwd_kernel, dim3((int)gx, B * C), dim3(256), 0, st, a, b, grad_dice, coef, grad_a,
                     grad_b, B, C, V, stride_b, stride_c, stride_v);
  DG_CHECK_LAUNCH("softdice_probs_bwd_kernel");
  return DGTTA_OK;
}
