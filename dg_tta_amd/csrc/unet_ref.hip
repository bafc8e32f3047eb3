// Network building blocks, general-shape VALU kernels ("impl 1"): correct for every Cin/Cout/stride/ld, used
//  (a) as the on-GPU cross-check for the MFMA implicit-GEMM kernels (conv_mfma.hip) at sizes where the CPU oracle
//      is too slow, and (b) for layer shapes the MFMA kernels do not cover.
// Also hosts the HBM-bound pieces that stay on the VALU: InstanceNorm+LeakyReLU fwd/bwd, the per-channel
// reductions, layout converters, weight packing, argmax/Dice counting.
// Semantics follow torch.nn.{Conv3d, InstanceNorm3d, LeakyReLU, ConvTranspose3d} as used by nnUNet's PlainConvUNet
// (dynamic-network-architectures==0.2; built at dg_tta/pretraining/nnUNetTrainer_GIN_MIND.py:46-53).
#include "common.h"

namespace {

// ============================================================================ weight packing
template <typename T>
__global__ void pack_weights_kernel(const float *__restrict__ w, T *__restrict__ wf, T *__restrict__ wb, int Cin,
                                    int Cout, int CinP, int CoutP) {
  const int64_t nf = (int64_t)27 * CinP * CoutP;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nf; i += (int64_t)gridDim.x * blockDim.x) {
    {  // wf[tap][ci][co]
      int co = (int)(i % CoutP), ci = (int)((i / CoutP) % CinP), tap = (int)(i / ((int64_t)CoutP * CinP));
      float v = (co < Cout && ci < Cin) ? w[((int64_t)co * Cin + ci) * 27 + tap] : 0.f;
      if (wf) st_f<T>(wf + i, v);
    }
    {  // wb[tap'][co][ci] = w[co][ci][26-tap']
      int ci = (int)(i % CinP), co = (int)((i / CinP) % CoutP), tap = (int)(i / ((int64_t)CoutP * CinP));
      float v = (co < Cout && ci < Cin) ? w[((int64_t)co * Cin + ci) * 27 + (26 - tap)] : 0.f;
      if (wb) st_f<T>(wb + i, v);
    }
  }
}

// ============================================================================ conv 3x3x3 forward (reference grade)
// one thread per (voxel, co); lanes run over co so the x value is a broadcast and wf[tap][ci][co] is coalesced.
template <typename T>
__global__ void conv3_fwd_ref_kernel(const T *__restrict__ x, int ldx, const T *__restrict__ wf,
                                     const float *__restrict__ bias, T *__restrict__ y, int ldy, int Cin, int Cout,
                                     int CinP, int CoutP, int Di, int Hi, int Wi, int Do, int Ho, int Wo, int s,
                                     int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int co = (int)(i % Cout);
    const int64_t vox = i / Cout;
    const int wo = (int)(vox % Wo), ho = (int)((vox / Wo) % Ho);
    const int d_o = (int)((vox / ((int64_t)Wo * Ho)) % Do), b = (int)(vox / ((int64_t)Wo * Ho * Do));
    float acc = 0.f;
    for (int kd = 0; kd < 3; ++kd) {
      const int di = d_o * s + kd - 1;
      if ((unsigned)di >= (unsigned)Di) continue;
      for (int kh = 0; kh < 3; ++kh) {
        const int hi = ho * s + kh - 1;
        if ((unsigned)hi >= (unsigned)Hi) continue;
        for (int kw = 0; kw < 3; ++kw) {
          const int wi = wo * s + kw - 1;
          if ((unsigned)wi >= (unsigned)Wi) continue;
          const T *xp = x + ((((int64_t)b * Di + di) * Hi + hi) * Wi + wi) * ldx;
          const T *wp = wf + ((int64_t)(kd * 9 + kh * 3 + kw) * CinP) * CoutP + co;
          for (int ci = 0; ci < Cin; ++ci) acc = __builtin_fmaf(ld_f<T>(xp + ci), ld_f<T>(wp + (int64_t)ci * CoutP), acc);
        }
      }
    }
    st_f<T>(y + vox * ldy + co, acc + (bias ? bias[co] : 0.f));
  }
}

// data gradient, any stride: thread per (input voxel, ci); wb[26-tap][co][ci] is coalesced over ci.
template <typename T>
__global__ void conv3_dgrad_ref_kernel(const T *__restrict__ dy, int lddy, const T *__restrict__ wb, T *__restrict__ dx,
                                       int lddx, int Cin, int Cout, int CinP, int CoutP, int Di, int Hi, int Wi, int Do,
                                       int Ho, int Wo, int s, int accumulate, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ci = (int)(i % Cin);
    const int64_t vox = i / Cin;
    const int wi = (int)(vox % Wi), hi = (int)((vox / Wi) % Hi);
    const int di = (int)((vox / ((int64_t)Wi * Hi)) % Di), b = (int)(vox / ((int64_t)Wi * Hi * Di));
    float acc = 0.f;
    for (int kd = 0; kd < 3; ++kd) {
      int td = di + 1 - kd;
      if (td < 0 || td % s) continue;
      td /= s;
      if (td >= Do) continue;
      for (int kh = 0; kh < 3; ++kh) {
        int th = hi + 1 - kh;
        if (th < 0 || th % s) continue;
        th /= s;
        if (th >= Ho) continue;
        for (int kw = 0; kw < 3; ++kw) {
          int tw = wi + 1 - kw;
          if (tw < 0 || tw % s) continue;
          tw /= s;
          if (tw >= Wo) continue;
          const T *gp = dy + ((((int64_t)b * Do + td) * Ho + th) * Wo + tw) * lddy;
          const T *wp = wb + ((int64_t)(26 - (kd * 9 + kh * 3 + kw)) * CoutP) * CinP + ci;
          for (int co = 0; co < Cout; ++co) acc = __builtin_fmaf(ld_f<T>(gp + co), ld_f<T>(wp + (int64_t)co * CinP), acc);
        }
      }
    }
    T *o = dx + vox * lddx + ci;
    st_f<T>(o, accumulate ? ld_f<T>(o) + acc : acc);
  }
}

// weight gradient partials: grid (pairs/256, 27, nsplit); thread = one (ci,co) pair, loops over a voxel slice.
// partial layout [split][co][ci][tap] (torch order) so the final reduction is a plain sum over splits.
template <typename T>
__global__ void conv3_wgrad_ref_kernel(const T *__restrict__ x, int ldx, const T *__restrict__ dy, int lddy,
                                       float *__restrict__ part, int Cin, int Cout, int B, int Di, int Hi, int Wi,
                                       int Do, int Ho, int Wo, int s) {
  const int pair = blockIdx.x * blockDim.x + threadIdx.x;
  if (pair >= Cin * Cout) return;
  const int co = pair % Cout, ci = pair / Cout;
  const int tap = blockIdx.y, kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
  const int64_t nvox = (int64_t)B * Do * Ho * Wo;
  const int64_t per = cdiv64(nvox, gridDim.z);
  const int64_t v0 = (int64_t)blockIdx.z * per, v1 = (v0 + per < nvox) ? v0 + per : nvox;
  float acc = 0.f;
  for (int64_t vox = v0; vox < v1; ++vox) {
    const int wo = (int)(vox % Wo), ho = (int)((vox / Wo) % Ho);
    const int d_o = (int)((vox / ((int64_t)Wo * Ho)) % Do), b = (int)(vox / ((int64_t)Wo * Ho * Do));
    const int di = d_o * s + kd - 1, hi = ho * s + kh - 1, wi = wo * s + kw - 1;
    if ((unsigned)di >= (unsigned)Di || (unsigned)hi >= (unsigned)Hi || (unsigned)wi >= (unsigned)Wi) continue;
    const float xv = ld_f<T>(x + ((((int64_t)b * Di + di) * Hi + hi) * Wi + wi) * ldx + ci);
    acc = __builtin_fmaf(xv, ld_f<T>(dy + vox * lddy + co), acc);
  }
  part[(((int64_t)blockIdx.z * Cout + co) * Cin + ci) * 27 + tap] = acc;
}

__global__ void reduce_splits_kernel(const float *__restrict__ part, float *__restrict__ out, int64_t n, int nsplit,
                                     int accumulate) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int k = 0; k < nsplit; ++k) s += part[(int64_t)k * n + i];
    out[i] = accumulate ? out[i] + s : s;
  }
}

// ============================================================================ per-channel reductions over rows
// rows = voxels of one batch sample; MODE 0: (sum y, sum y^2)          [InstanceNorm statistics]
//                                     MODE 1: (sum da, sum da*xhat)     [InstanceNorm backward]; da = gz*lrelu'(a)
//                                     MODE 2: (sum y, 0)                [bias gradient]
// grid (nblk, B); partial[b][blk][c][2] double; the finalize kernels sum blocks in fixed order.
template <typename T, int MODE>
__global__ __launch_bounds__(256) void chan_reduce_kernel(const T *__restrict__ y, int ldy, const T *__restrict__ gz,
                                                          int ldgz, const float *__restrict__ mean_rstd,
                                                          const float *__restrict__ gamma,
                                                          const float *__restrict__ beta, float slope,
                                                          double *__restrict__ partial, int C, int64_t V) {
  __shared__ float red[2][256];
  const int b = blockIdx.y;
  int CL = 1;
  while (CL < C && CL < 64) CL <<= 1;  // channel lanes (power of two <= 64)
  const int RG = 256 / CL;             // row groups
  const int cl = threadIdx.x % CL, rg = threadIdx.x / CL;
  const int64_t rows_per_blk = cdiv64(V, gridDim.x);
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_blk, r1 = (r0 + rows_per_blk < V) ? r0 + rows_per_blk : V;
  for (int c0 = 0; c0 < C; c0 += CL) {
    const int c = c0 + cl;
    float s0 = 0.f, s1 = 0.f;
    if (c < C) {
      float mu = 0.f, rs = 0.f, ga = 0.f, be = 0.f;
      if (MODE == 1) {
        mu = mean_rstd[((int64_t)b * C + c) * 2];
        rs = mean_rstd[((int64_t)b * C + c) * 2 + 1];
        ga = gamma[c];
        be = beta[c];
      }
      for (int64_t r = r0 + rg; r < r1; r += RG) {
        const float v = ld_f<T>(y + ((int64_t)b * V + r) * ldy + c);
        if (MODE == 0) {
          s0 += v;
          s1 += v * v;
        } else if (MODE == 2) {
          s0 += v;
        } else {
          const float xh = (v - mu) * rs;
          const float a = xh * ga + be;
          float g = ld_f<T>(gz + ((int64_t)b * V + r) * ldgz + c);
          g = a > 0.f ? g : g * slope;
          s0 += g;
          s1 += g * xh;
        }
      }
    }
    __syncthreads();
    red[0][threadIdx.x] = s0;
    red[1][threadIdx.x] = s1;
    __syncthreads();
    if (rg == 0 && c < C) {
      double t0 = 0.0, t1 = 0.0;
      for (int k = 0; k < RG; ++k) {
        t0 += (double)red[0][k * CL + cl];
        t1 += (double)red[1][k * CL + cl];
      }
      double *p = partial + ((((int64_t)b * gridDim.x + blockIdx.x) * C) + c) * 2;
      p[0] = t0;
      p[1] = t1;
    }
  }
}

// 16-byte vectorised variant of chan_reduce_kernel (needs C, ld multiples of EPV = 16/sizeof(T) and aligned rows):
// a thread owns EPV consecutive channels, 256/G rows are in flight per iteration (G = C/EPV), two rows per thread and
// iteration for memory-level parallelism.  Same partial layout [b][blk][c][2] (double).
template <typename T>
struct VecOf {
  static constexpr int EPV = 16 / sizeof(T);
};
template <typename T>
__device__ __forceinline__ void unpack16(const uint4 &v, float *f);
template <>
__device__ __forceinline__ void unpack16<float>(const uint4 &v, float *f) {
  f[0] = __uint_as_float(v.x);
  f[1] = __uint_as_float(v.y);
  f[2] = __uint_as_float(v.z);
  f[3] = __uint_as_float(v.w);
}
template <>
__device__ __forceinline__ void unpack16<bf16_t>(const uint4 &v, float *f) {
  const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = __uint_as_float(w[i] << 16);
    f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
  }
}
template <>
__device__ __forceinline__ void unpack16<f16_t>(const uint4 &v, float *f) {
  const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) unpack2_16<f16_t>(w[i], f[2 * i], f[2 * i + 1]);
}

template <typename T, int MODE>
__global__ __launch_bounds__(256) void chan_reduce_vec_kernel(const T *__restrict__ y, int ldy, const T *__restrict__ gz,
                                                              int ldgz, const float *__restrict__ mean_rstd,
                                                              const float *__restrict__ gamma,
                                                              const float *__restrict__ beta, float slope,
                                                              double *__restrict__ partial, int C, int64_t V) {
  constexpr int EPV = VecOf<T>::EPV;
  extern __shared__ float sred[];          // [rpi][C][2]
  const int b = blockIdx.y;
  const int G = C / EPV, rpi = 256 / G;
  const int cg = threadIdx.x % G, rg = threadIdx.x / G;
  const bool active = rg < rpi;
  // rows are dealt to the workgroups in turn, 2 rpi at a time (round 4): with one contiguous chunk of V / gridDim.x rows
  // per workgroup the workgroups in flight read addresses a fixed 256 KiB (128^3 x 32 channels) apart - a few HBM channels
  // at a time, 3.1-3.8 TB/s where the apply passes, which walk the tensor in this interleaved order, stream at 5
  const int64_t r1 = V;
  float s0[EPV], s1[EPV], mu[EPV], rs[EPV], ga[EPV], be[EPV];
#pragma unroll
  for (int e = 0; e < EPV; ++e) {
    s0[e] = s1[e] = 0.f;
    mu[e] = rs[e] = ga[e] = be[e] = 0.f;
    if (MODE == 1 && active) {
      const int c = cg * EPV + e;
      mu[e] = mean_rstd[((int64_t)b * C + c) * 2];
      rs[e] = mean_rstd[((int64_t)b * C + c) * 2 + 1];
      ga[e] = gamma[c];
      be[e] = beta[c];
    }
  }
  if (active) {
    const T *yb = y + (int64_t)b * V * ldy + cg * EPV;
    const T *gb = (MODE == 1) ? gz + (int64_t)b * V * ldgz + cg * EPV : nullptr;
    for (int64_t r = (int64_t)blockIdx.x * (2 * rpi) + rg; r < r1; r += (int64_t)gridDim.x * (2 * rpi)) {
      const bool two = r + rpi < r1;
      // streaming loads, as in the apply passes (the tensors are far larger than the caches and are read once per pass)
      typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
      auto ldnt = [](const T *p) {
        const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t *>(p));
        return make_uint4(v[0], v[1], v[2], v[3]);
      };
      uint4 v0 = ldnt(yb + r * ldy), v1 = make_uint4(0, 0, 0, 0);
      uint4 g0 = make_uint4(0, 0, 0, 0), g1 = g0;
      if (two) v1 = ldnt(yb + (r + rpi) * ldy);
      if (MODE == 1) {
        g0 = ldnt(gb + r * ldgz);
        if (two) g1 = ldnt(gb + (r + rpi) * ldgz);
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        if (k == 1 && !two) break;
        float f[EPV], g[EPV];
        unpack16<T>(k ? v1 : v0, f);
        if (MODE == 1) unpack16<T>(k ? g1 : g0, g);
#pragma unroll
        for (int e = 0; e < EPV; ++e) {
          if (MODE == 0) {
            s0[e] += f[e];
            s1[e] += f[e] * f[e];
          } else if (MODE == 2) {
            s0[e] += f[e];
          } else {
            const float xh = (f[e] - mu[e]) * rs[e];
            const float a = xh * ga[e] + be[e];
            const float gg = a > 0.f ? g[e] : g[e] * slope;
            s0[e] += gg;
            s1[e] += gg * xh;
          }
        }
      }
    }
#pragma unroll
    for (int e = 0; e < EPV; ++e) {
      sred[((rg * C) + cg * EPV + e) * 2 + 0] = s0[e];
      sred[((rg * C) + cg * EPV + e) * 2 + 1] = s1[e];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    double t0 = 0.0, t1 = 0.0;
    for (int k = 0; k < rpi; ++k) {
      t0 += (double)sred[(k * C + c) * 2];
      t1 += (double)sred[(k * C + c) * 2 + 1];
    }
    double *p = partial + ((((int64_t)b * gridDim.x + blockIdx.x) * C) + c) * 2;
    p[0] = t0;
    p[1] = t1;
  }
}

template <typename T>
static bool vec_ok(const void *p, int ld, int C) {
  constexpr int EPV = 16 / sizeof(T);
  return p && ((uintptr_t)p & 15) == 0 && ld % EPV == 0 && C % EPV == 0 && C / EPV <= 256;
}

// launches the vectorised reduction when the operands allow it, else the scalar kernel
template <typename T, int MODE>
static void launch_chan_reduce(const void *y, int ldy, const void *gz, int ldgz, const float *mean_rstd,
                               const float *gamma, const float *beta, float slope, double *partial, int nblk, int B, int C,
                               int64_t V, hipStream_t st) {
  if (vec_ok<T>(y, ldy, C) && (MODE != 1 || vec_ok<T>(gz, ldgz, C))) {
    constexpr int EPV = 16 / sizeof(T);
    const int rpi = 256 / (C / EPV);
    hipLaunchKernelGGL((chan_reduce_vec_kernel<T, MODE>), dim3(nblk, B), dim3(256), (size_t)rpi * C * 2 * sizeof(float), st,
                       (const T *)y, ldy, (const T *)gz, ldgz, mean_rstd, gamma, beta, slope, partial, C, V);
  } else {
    hipLaunchKernelGGL((chan_reduce_kernel<T, MODE>), dim3(nblk, B), dim3(256), 0, st, (const T *)y, ldy, (const T *)gz,
                       ldgz, mean_rstd, gamma, beta, slope, partial, C, V);
  }
}

template <typename T>
__device__ __forceinline__ uint4 pack16(const float *f);
template <>
__device__ __forceinline__ uint4 pack16<float>(const float *f) {
  return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
}
template <>
__device__ __forceinline__ uint4 pack16<bf16_t>(const float *f) {
  unsigned w[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) w[i] = (unsigned)f32_to_bf16(f[2 * i]) | ((unsigned)f32_to_bf16(f[2 * i + 1]) << 16);
  return make_uint4(w[0], w[1], w[2], w[3]);
}
template <>
__device__ __forceinline__ uint4 pack16<f16_t>(const float *f) {
  return make_uint4(pack2_16<f16_t>(f[0], f[1]), pack2_16<f16_t>(f[2], f[3]), pack2_16<f16_t>(f[4], f[5]),
                    pack2_16<f16_t>(f[6], f[7]));
}

// 16-byte vectorised InstanceNorm+LeakyReLU apply kernels (forward MODE 0, backward MODE 1): grid (blocks, B); the
// per-channel constants of sample b are staged in LDS once per workgroup; each thread streams 16-byte channel groups.
// Same arithmetic as in_lrelu_apply_kernel / in_lrelu_bwd_apply_kernel.
template <typename T, int MODE>
__global__ __launch_bounds__(256) void in_apply_vec_kernel(const T *__restrict__ y, int ldy, const T *__restrict__ gz,
                                                           int ldgz, const float *__restrict__ mean_rstd,
                                                           const float *__restrict__ gamma,
                                                           const float *__restrict__ beta, const float *__restrict__ c12,
                                                           T *__restrict__ out, int ldo, int C, int64_t V, float slope,
                                                           int nt) {
  constexpr int EPV = 16 / sizeof(T);
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  extern __shared__ float sc[];     // fwd: [C][2] (alpha, beta'); bwd: [C][6] (mu, rs, ga, be, c1, c2)
  const int b = blockIdx.y;
  constexpr int NK = MODE == 0 ? 2 : 6;
  for (int c = threadIdx.x; c < C; c += 256) {
    const float mu = mean_rstd[((int64_t)b * C + c) * 2], rs = mean_rstd[((int64_t)b * C + c) * 2 + 1];
    if (MODE == 0) {
      const float al = rs * gamma[c];
      sc[c * NK] = al;
      sc[c * NK + 1] = beta[c] - mu * al;
    } else {
      sc[c * NK] = mu;
      sc[c * NK + 1] = rs;
      sc[c * NK + 2] = gamma[c];
      sc[c * NK + 3] = beta[c];
      sc[c * NK + 4] = c12[((int64_t)b * C + c) * 2];
      sc[c * NK + 5] = c12[((int64_t)b * C + c) * 2 + 1];
    }
  }
  __syncthreads();
  const int G = C / EPV;
  const int64_t items = V * G;
  const T *yb = y + (int64_t)b * V * ldy;
  const T *gb = MODE == 1 ? gz + (int64_t)b * V * ldgz : nullptr;
  T *ob = out + (int64_t)b * V * ldo;
  if (256 % G == 0) {
    // fast path (G = 4, 8, 16, 32: every layer except the 320-channel bottleneck): a thread keeps ONE channel group for the
    // whole launch, so its per-channel constants sit in registers (no LDS read per element, no 64-bit division per item),
    // and two rows are in flight per iteration
    const int c0 = (threadIdx.x % G) * EPV;
    float kc[EPV][NK];
#pragma unroll
    for (int e = 0; e < EPV; ++e)
#pragma unroll
      for (int q = 0; q < NK; ++q) kc[e][q] = sc[(c0 + e) * NK + q];
    const int rpb = 256 / G;                                             // rows per workgroup per step
    const int64_t rstep = (int64_t)gridDim.x * rpb;
    auto one = [&](const uint4 &yv, const uint4 &gv, int64_t row) {
      float f[EPV], g[EPV], o[EPV];
      unpack16<T>(yv, f);
      if (MODE == 1) unpack16<T>(gv, g);
#pragma unroll
      for (int e = 0; e < EPV; ++e) {
        if constexpr (MODE == 0) {
          o[e] = lrelu(f[e] * kc[e][0] + kc[e][1], slope);
        } else {
          const float xh = (f[e] - kc[e][0]) * kc[e][1];
          const float a = xh * kc[e][2] + kc[e][3];
          const float gg = a > 0.f ? g[e] : g[e] * slope;
          o[e] = (kc[e][2] * kc[e][1]) * ((gg - kc[e][4]) - xh * kc[e][5]);
        }
      }
      const uint4 pk = pack16<T>(o);
      if (nt) {
        const u32x4_t nv = {pk.x, pk.y, pk.z, pk.w};
        __builtin_nontemporal_store(nv, reinterpret_cast<u32x4_t *>(ob + row * ldo + c0));
      } else {
        *reinterpret_cast<uint4 *>(ob + row * ldo + c0) = pk;
      }
    };
    auto ld = [&](const T *p) {
      if (nt) {
        const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t *>(p));
        return make_uint4(v[0], v[1], v[2], v[3]);
      }
      return *reinterpret_cast<const uint4 *>(p);
    };
    for (int64_t row = (int64_t)blockIdx.x * rpb + threadIdx.x / G; row < V; row += 2 * rstep) {
      const int64_t row2 = row + rstep;
      const bool two = row2 < V;
      const uint4 y0 = ld(yb + row * ldy + c0);
      uint4 y1 = y0, g0 = y0, g1 = y0;
      if (two) y1 = ld(yb + row2 * ldy + c0);
      if (MODE == 1) {
        g0 = ld(gb + row * ldgz + c0);
        if (two) g1 = ld(gb + row2 * ldgz + c0);
      }
      one(y0, g0, row);
      if (two) one(y1, g1, row2);
    }
    return;
  }
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < items; i += (int64_t)gridDim.x * 256) {
    const int64_t row = i / G;
    const int c0 = (int)(i % G) * EPV;
    float f[EPV], g[EPV], o[EPV];
    unpack16<T>(*reinterpret_cast<const uint4 *>(yb + row * ldy + c0), f);
    if (MODE == 1) unpack16<T>(*reinterpret_cast<const uint4 *>(gb + row * ldgz + c0), g);
#pragma unroll
    for (int e = 0; e < EPV; ++e) {
      const float *k = sc + (c0 + e) * NK;
      if (MODE == 0) {
        o[e] = lrelu(f[e] * k[0] + k[1], slope);
      } else {
        const float xh = (f[e] - k[0]) * k[1];
        const float a = xh * k[2] + k[3];
        const float gg = a > 0.f ? g[e] : g[e] * slope;
        o[e] = (k[2] * k[1]) * ((gg - k[4]) - xh * k[5]);
      }
    }
    *reinterpret_cast<uint4 *>(ob + row * ldo + c0) = pack16<T>(o);
  }
}

// sum of two doubles over a 256-thread workgroup (fixed order: lanes by butterfly, then waves 0..3); result in all threads
__device__ __forceinline__ void block_sum2_d(double &a, double &b, double *red /* >= 8 doubles of LDS */) {
  a = wave_sum_d(a);
  b = wave_sum_d(b);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
    red[2 * w] = a;
    red[2 * w + 1] = b;
  }
  __syncthreads();
  a = (red[0] + red[2]) + (red[4] + red[6]);
  b = (red[1] + red[3]) + (red[5] + red[7]);
}


// InstanceNorm statistics finalize: mean, rstd = 1/sqrt(biased var + eps).  One 256-thread workgroup per (b,c); the number of
// partial blocks is read from the device-side header when hdr != NULL (statistics produced by the conv epilogue).
__global__ void in_stats_finalize_kernel(const double *__restrict__ partial, const long long *__restrict__ hdr, int nblk_h,
                                         int B, int C, int64_t V, float eps, float *__restrict__ mean_rstd) {
  const int i = blockIdx.x;
  const int b = i / C, c = i % C;
  const int nblk = hdr ? (int)hdr[0] : nblk_h;
  __shared__ double red[8];
  double s = 0.0, ss = 0.0;
  for (int k = threadIdx.x; k < nblk; k += 256) {
    const double2 v = *reinterpret_cast<const double2 *>(partial + ((((int64_t)b * nblk + k) * C) + c) * 2);
    s += v.x;
    ss += v.y;
  }
  block_sum2_d(s, ss, red);
  if (threadIdx.x == 0) {
    const double mean = s / (double)V;
    double var = ss / (double)V - mean * mean;
    if (var < 0.0) var = 0.0;
    mean_rstd[2 * i] = (float)mean;
    mean_rstd[2 * i + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

// z = lrelu(y*alpha + beta'), alpha = rstd*gamma, beta' = beta - mean*alpha  (ATen's batch_norm transform form)
template <typename T>
__global__ void in_lrelu_apply_kernel(const T *__restrict__ y, int ldy, const float *__restrict__ mean_rstd,
                                      const float *__restrict__ gamma, const float *__restrict__ beta,
                                      T *__restrict__ z, int ldz, int C, int64_t V, float slope, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int64_t row = i / C;  // b*V + v
    const int b = (int)(row / V);
    const float mu = mean_rstd[((int64_t)b * C + c) * 2], rs = mean_rstd[((int64_t)b * C + c) * 2 + 1];
    const float al = rs * gamma[c];
    const float bt = beta[c] - mu * al;
    const float a = ld_f<T>(y + row * ldy + c) * al + bt;
    st_f<T>(z + row * ldz + c, lrelu(a, slope));
  }
}

// backward finalize: c1 = mean(da), c2 = mean(da*xhat); dgamma (+)= sum_b sum(da*xhat); dbeta (+)= sum_b sum(da).
// one 256-thread workgroup per channel
__global__ void in_bwd_finalize_kernel(const double *__restrict__ partial, int nblk, int B, int C, int64_t V,
                                       float *__restrict__ c12, float *__restrict__ dgamma, float *__restrict__ dbeta,
                                       int accumulate) {
  const int c = blockIdx.x;
  __shared__ double red[8];
  double g_acc = 0.0, b_acc = 0.0;
  for (int b = 0; b < B; ++b) {
    double s0 = 0.0, s1 = 0.0;
    for (int k = threadIdx.x; k < nblk; k += 256) {
      const double2 v = *reinterpret_cast<const double2 *>(partial + ((((int64_t)b * nblk + k) * C) + c) * 2);
      s0 += v.x;
      s1 += v.y;
    }
    block_sum2_d(s0, s1, red);
    if (threadIdx.x == 0) {
      c12[((int64_t)b * C + c) * 2] = (float)(s0 / (double)V);
      c12[((int64_t)b * C + c) * 2 + 1] = (float)(s1 / (double)V);
    }
    b_acc += s0;
    g_acc += s1;
  }
  if (threadIdx.x == 0) {
    dgamma[c] = accumulate ? dgamma[c] + (float)g_acc : (float)g_acc;
    dbeta[c] = accumulate ? dbeta[c] + (float)b_acc : (float)b_acc;
  }
}

// The same from the partial sums the data-gradient kernel left (conv_rows.hip, GST): per (sample, tile) sums of g' and
// g' * y in the layout of the forward statistics (header = tiles per sample); sum g' xhat = rstd (sum g' y - mean sum g').
__global__ void in_bwd_finalize_gstats_kernel(const double *__restrict__ stats, int B, int C, int64_t V,
                                              const float *__restrict__ mean_rstd, float *__restrict__ c12,
                                              float *__restrict__ dgamma, float *__restrict__ dbeta, int accumulate) {
  const int c = blockIdx.x;
  __shared__ double red[8];
  const int nblk = (int)reinterpret_cast<const long long *>(stats)[0];
  const double *partial = stats + 32;
  double g_acc = 0.0, b_acc = 0.0;
  for (int b = 0; b < B; ++b) {
    double s0 = 0.0, s1 = 0.0;
    for (int k = threadIdx.x; k < nblk; k += 256) {
      const double2 v = *reinterpret_cast<const double2 *>(partial + ((((int64_t)b * nblk + k) * C) + c) * 2);
      s0 += v.x;
      s1 += v.y;
    }
    block_sum2_d(s0, s1, red);
    const double mu = (double)mean_rstd[((int64_t)b * C + c) * 2], rs = (double)mean_rstd[((int64_t)b * C + c) * 2 + 1];
    const double s1x = rs * (s1 - mu * s0);
    if (threadIdx.x == 0) {
      c12[((int64_t)b * C + c) * 2] = (float)(s0 / (double)V);
      c12[((int64_t)b * C + c) * 2 + 1] = (float)(s1x / (double)V);
    }
    b_acc += s0;
    g_acc += s1x;
  }
  if (threadIdx.x == 0) {
    dgamma[c] = accumulate ? dgamma[c] + (float)g_acc : (float)g_acc;
    dbeta[c] = accumulate ? dbeta[c] + (float)b_acc : (float)b_acc;
  }
}

// dy = gamma*rstd*(da - c1 - xhat*c2)
template <typename T>
__global__ void in_lrelu_bwd_apply_kernel(const T *__restrict__ gz, int ldgz, const T *__restrict__ y, int ldy,
                                          const float *__restrict__ mean_rstd, const float *__restrict__ gamma,
                                          const float *__restrict__ beta, const float *__restrict__ c12,
                                          T *__restrict__ dy, int lddy, int C, int64_t V, float slope, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int64_t row = i / C;
    const int b = (int)(row / V);
    const int64_t bc = (int64_t)b * C + c;
    const float mu = mean_rstd[bc * 2], rs = mean_rstd[bc * 2 + 1], ga = gamma[c];
    const float xh = (ld_f<T>(y + row * ldy + c) - mu) * rs;
    const float a = xh * ga + beta[c];
    float g = ld_f<T>(gz + row * ldgz + c);
    g = a > 0.f ? g : g * slope;
    st_f<T>(dy + row * lddy + c, (ga * rs) * ((g - c12[bc * 2]) - xh * c12[bc * 2 + 1]));
  }
}

__global__ void bias_finalize_kernel(const double *__restrict__ partial, int nblk, int B, int C, float *__restrict__ db,
                                     int accumulate) {
  const int c = blockIdx.x;   // one wave per channel
  double s = 0.0;
  for (int k = threadIdx.x; k < B * nblk; k += 64) s += partial[(((int64_t)k * C) + c) * 2];
  s = wave_sum_d(s);
  if (threadIdx.x == 0) db[c] = accumulate ? db[c] + (float)s : (float)s;
}

// ============================================================================ ConvTranspose3d k2 s2
template <typename T>
__global__ void convT_fwd_ref_kernel(const T *__restrict__ x, int ldx, const float *__restrict__ w,
                                     const float *__restrict__ bias, T *__restrict__ out, int ldo, int Cin, int Cout,
                                     int Di, int Hi, int Wi, int64_t total) {
  const int Do = 2 * Di, Ho = 2 * Hi, Wo = 2 * Wi;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int co = (int)(i % Cout);
    const int64_t vox = i / Cout;
    const int wo = (int)(vox % Wo), ho = (int)((vox / Wo) % Ho);
    const int d_o = (int)((vox / ((int64_t)Wo * Ho)) % Do), b = (int)(vox / ((int64_t)Wo * Ho * Do));
    const int o = ((d_o & 1) * 2 + (ho & 1)) * 2 + (wo & 1);
    const T *xp = x + ((((int64_t)b * Di + (d_o >> 1)) * Hi + (ho >> 1)) * Wi + (wo >> 1)) * ldx;
    float acc = 0.f;
    for (int ci = 0; ci < Cin; ++ci) acc = __builtin_fmaf(ld_f<T>(xp + ci), w[((int64_t)ci * Cout + co) * 8 + o], acc);
    st_f<T>(out + vox * ldo + co, acc + (bias ? bias[co] : 0.f));
  }
}

template <typename T>
__global__ void convT_dgrad_ref_kernel(const T *__restrict__ dout, int lddo, const float *__restrict__ w,
                                       T *__restrict__ dx, int lddx, int Cin, int Cout, int Di, int Hi, int Wi,
                                       int64_t total) {
  const int Ho = 2 * Hi, Wo = 2 * Wi, Do = 2 * Di;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ci = (int)(i % Cin);
    const int64_t vox = i / Cin;
    const int wi = (int)(vox % Wi), hi = (int)((vox / Wi) % Hi);
    const int di = (int)((vox / ((int64_t)Wi * Hi)) % Di), b = (int)(vox / ((int64_t)Wi * Hi * Di));
    float acc = 0.f;
    for (int o = 0; o < 8; ++o) {
      const T *gp =
          dout + ((((int64_t)b * Do + 2 * di + (o >> 2)) * Ho + 2 * hi + ((o >> 1) & 1)) * Wo + 2 * wi + (o & 1)) * lddo;
      for (int co = 0; co < Cout; ++co) acc = __builtin_fmaf(ld_f<T>(gp + co), w[((int64_t)ci * Cout + co) * 8 + o], acc);
    }
    st_f<T>(dx + vox * lddx + ci, acc);
  }
}

// partial[split][ci][co][o]; grid (pairs/256, 8, nsplit)
template <typename T>
__global__ void convT_wgrad_ref_kernel(const T *__restrict__ x, int ldx, const T *__restrict__ dout, int lddo,
                                       float *__restrict__ part, int Cin, int Cout, int B, int Di, int Hi, int Wi) {
  const int pair = blockIdx.x * blockDim.x + threadIdx.x;
  if (pair >= Cin * Cout) return;
  const int co = pair % Cout, ci = pair / Cout, o = blockIdx.y;
  const int Ho = 2 * Hi, Wo = 2 * Wi, Do = 2 * Di;
  const int64_t nvox = (int64_t)B * Di * Hi * Wi;
  const int64_t per = cdiv64(nvox, gridDim.z);
  const int64_t v0 = (int64_t)blockIdx.z * per, v1 = (v0 + per < nvox) ? v0 + per : nvox;
  float acc = 0.f;
  for (int64_t vox = v0; vox < v1; ++vox) {
    const int wi = (int)(vox % Wi), hi = (int)((vox / Wi) % Hi);
    const int di = (int)((vox / ((int64_t)Wi * Hi)) % Di), b = (int)(vox / ((int64_t)Wi * Hi * Di));
    const T *gp =
        dout + ((((int64_t)b * Do + 2 * di + (o >> 2)) * Ho + 2 * hi + ((o >> 1) & 1)) * Wo + 2 * wi + (o & 1)) * lddo;
    acc = __builtin_fmaf(ld_f<T>(x + vox * ldx + ci), ld_f<T>(gp + co), acc);
  }
  part[(((int64_t)blockIdx.z * Cin + ci) * Cout + co) * 8 + o] = acc;
}

// ============================================================================ 1x1x1 head on selected rows
template <typename T, bool NDHWC>
__global__ void head_fwd_kernel(const T *__restrict__ x, int ldx, const float *__restrict__ w,
                                const float *__restrict__ bias, const int *__restrict__ sel, int nsel,
                                float *__restrict__ out, int ldo, int Cin, int64_t V, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int k;
    int64_t row;
    if (NDHWC) {
      k = (int)(i % nsel);
      row = i / nsel;
    } else {  // i = (b*nsel + k)*V + v
      const int64_t v = i % V;
      k = (int)((i / V) % nsel);
      row = (i / (V * nsel)) * V + v;
    }
    const int r = sel ? sel[k] : k;
    const T *xp = x + row * ldx;
    const float *wp = w + (int64_t)r * Cin;
    float acc = 0.f;
    for (int ci = 0; ci < Cin; ++ci) acc = __builtin_fmaf(ld_f<T>(xp + ci), wp[ci], acc);
    acc += bias[r];
    if (NDHWC) out[row * ldo + k] = acc;
    else out[i] = acc;
  }
}

template <typename T>
__global__ void head_dgrad_kernel(const float *__restrict__ dout, int lddo, const float *__restrict__ w,
                                  const int *__restrict__ sel, int nsel, T *__restrict__ dx, int lddx, int Cin,
                                  int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ci = (int)(i % Cin);
    const int64_t row = i / Cin;
    float acc = 0.f;
    for (int k = 0; k < nsel; ++k) acc = __builtin_fmaf(dout[row * lddo + k], w[(int64_t)(sel ? sel[k] : k) * Cin + ci], acc);
    st_f<T>(dx + row * lddx + ci, acc);
  }
}

// Fast head kernels for the production shape (CIN input channels, nsel <= 32 selected rows): one thread per voxel, the
// activation row lives in registers (16-byte loads), the selected weight rows in LDS (broadcast reads).
template <typename T, int CIN, bool NDHWC>
__global__ __launch_bounds__(256) void head_fwd_fast_kernel(const T *__restrict__ x, int ldx, const float *__restrict__ w,
                                                            const float *__restrict__ bias, const int *__restrict__ sel,
                                                            int nsel, float *__restrict__ out, int ldo, int64_t V,
                                                            int64_t rows) {
  constexpr int EPV = 16 / sizeof(T);
  __shared__ float sw[128 * CIN + 128];
  for (int i = threadIdx.x; i < nsel * CIN; i += 256) sw[i] = w[(int64_t)(sel ? sel[i / CIN] : i / CIN) * CIN + i % CIN];
  for (int i = threadIdx.x; i < nsel; i += 256) sw[128 * CIN + i] = bias[sel ? sel[i] : i];
  __syncthreads();
  for (int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x; row < rows; row += (int64_t)gridDim.x * 256) {
    float xr[CIN];
#pragma unroll
    for (int g = 0; g < CIN / EPV; ++g)
      unpack16<T>(*reinterpret_cast<const uint4 *>(x + row * ldx + g * EPV), xr + g * EPV);
    for (int k = 0; k < nsel; ++k) {
      float acc = 0.f;
#pragma unroll
      for (int ci = 0; ci < CIN; ++ci) acc = __builtin_fmaf(xr[ci], sw[k * CIN + ci], acc);
      acc += sw[128 * CIN + k];
      if (NDHWC) out[row * ldo + k] = acc;
      else out[((row / V) * nsel + k) * V + row % V] = acc;
    }
  }
}

template <typename T, int CIN>
__global__ __launch_bounds__(256) void head_dgrad_fast_kernel(const float *__restrict__ dout, int lddo,
                                                              const float *__restrict__ w, const int *__restrict__ sel,
                                                              int nsel, T *__restrict__ dx, int lddx, int64_t rows) {
  __shared__ float sw[32 * CIN];
  for (int i = threadIdx.x; i < nsel * CIN; i += 256) sw[i] = w[(int64_t)(sel ? sel[i / CIN] : i / CIN) * CIN + i % CIN];
  __syncthreads();
  for (int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x; row < rows; row += (int64_t)gridDim.x * 256) {
    float acc[CIN];
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci) acc[ci] = 0.f;
    for (int k = 0; k < nsel; ++k) {
      const float g = dout[row * lddo + k];
#pragma unroll
      for (int ci = 0; ci < CIN; ++ci) acc[ci] = __builtin_fmaf(g, sw[k * CIN + ci], acc[ci]);
    }
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci) st_f<T>(dx + row * lddx + ci, acc[ci]);
  }
}

// Coalesced variants for the production shape (32 input channels in contiguous 64-byte bf16 rows / 128-byte fp32 rows,
// nsel <= 16 selected classes in contiguous fp32 rows): a workgroup moves 256 rows through LDS in both directions, so
// that every global access is a contiguous 16-byte-per-lane stream (thread-per-row loads touch 64 cache lines per
// instruction, and the per-class 4-byte stores 64 lines for 256 bytes).
template <typename T>
__global__ __launch_bounds__(256) void head_fwd_lds_kernel(const T *__restrict__ x, const float *__restrict__ w,
                                                           const float *__restrict__ bias, const int *__restrict__ sel,
                                                           int nsel, float *__restrict__ out, int64_t rows) {
  constexpr int CIN = 32, EPV = 16 / sizeof(T), XU = CIN / EPV;      // uint4 per input row
  constexpr int XP = XU + 1;                                          // padded row pitch in uint4
  __shared__ float sw[16 * CIN + 16];
  __shared__ uint4 sx[256 * XP];
  __shared__ float so[256 * 17];
  for (int i = threadIdx.x; i < nsel * CIN; i += 256) sw[i] = w[(int64_t)(sel ? sel[i / CIN] : i / CIN) * CIN + i % CIN];
  for (int i = threadIdx.x; i < nsel; i += 256) sw[16 * CIN + i] = bias[sel ? sel[i] : i];
  for (int64_t r0 = (int64_t)blockIdx.x * 256; r0 < rows; r0 += (int64_t)gridDim.x * 256) {
    const int nr = rows - r0 < 256 ? (int)(rows - r0) : 256;
    __syncthreads();
    const uint4 *gx = reinterpret_cast<const uint4 *>(x + r0 * CIN);
    for (int i = threadIdx.x; i < nr * XU; i += 256) sx[(i / XU) * XP + i % XU] = gx[i];
    __syncthreads();
    if ((int)threadIdx.x < nr) {
      float xr[CIN];
#pragma unroll
      for (int g = 0; g < XU; ++g) unpack16<T>(sx[threadIdx.x * XP + g], xr + g * EPV);
      for (int k = 0; k < nsel; ++k) {
        float acc = 0.f;
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) acc = __builtin_fmaf(xr[ci], sw[k * CIN + ci], acc);
        so[threadIdx.x * 17 + k] = acc + sw[16 * CIN + k];
      }
    }
    __syncthreads();
    float *go = out + r0 * nsel;
    for (int i = threadIdx.x; i < nr * nsel; i += 256) go[i] = so[(i / nsel) * 17 + i % nsel];
  }
}

template <typename T>
__global__ __launch_bounds__(256) void head_dgrad_lds_kernel(const float *__restrict__ dout, const float *__restrict__ w,
                                                             const int *__restrict__ sel, int nsel, T *__restrict__ dx,
                                                             int64_t rows, unsigned short *__restrict__ d16) {
  // d16 != NULL: also leave the 16-bit copy of dout [rows][nsel] that the MFMA weight gradient of the head reads (saves
  // its own conversion pass over the fp32 gradient)
  constexpr int CIN = 32, EPV = 16 / sizeof(T), XU = CIN / EPV, XP = XU + 1;
  __shared__ float sw[16 * CIN];
  __shared__ float sg[256 * 17];
  __shared__ uint4 sx[256 * XP];
  for (int i = threadIdx.x; i < nsel * CIN; i += 256) sw[i] = w[(int64_t)(sel ? sel[i / CIN] : i / CIN) * CIN + i % CIN];
  for (int64_t r0 = (int64_t)blockIdx.x * 256; r0 < rows; r0 += (int64_t)gridDim.x * 256) {
    const int nr = rows - r0 < 256 ? (int)(rows - r0) : 256;
    __syncthreads();
    const float *gg = dout + r0 * nsel;
    for (int i = threadIdx.x; i < nr * nsel; i += 256) {
      const float v = gg[i];
      sg[(i / nsel) * 17 + i % nsel] = v;
      if constexpr (sizeof(T) == 2) {
        if (d16 && nsel != 16) d16[r0 * nsel + i] = f32_to_16<T>(v);
      }
    }
    __syncthreads();
    if constexpr (sizeof(T) == 2) {
      if (d16 && nsel == 16 && (int)threadIdx.x < nr) {      // one 32-byte row per thread: two 16-byte stores
        uint4 *o = reinterpret_cast<uint4 *>(d16 + (r0 + threadIdx.x) * 16);
        const float *g = sg + threadIdx.x * 17;
        o[0] = make_uint4(pack2_16<T>(g[0], g[1]), pack2_16<T>(g[2], g[3]), pack2_16<T>(g[4], g[5]), pack2_16<T>(g[6], g[7]));
        o[1] = make_uint4(pack2_16<T>(g[8], g[9]), pack2_16<T>(g[10], g[11]), pack2_16<T>(g[12], g[13]),
                          pack2_16<T>(g[14], g[15]));
      }
    }
    if ((int)threadIdx.x < nr) {
      float acc[CIN];
#pragma unroll
      for (int ci = 0; ci < CIN; ++ci) acc[ci] = 0.f;
      for (int k = 0; k < nsel; ++k) {
        const float g = sg[threadIdx.x * 17 + k];
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) acc[ci] = __builtin_fmaf(g, sw[k * CIN + ci], acc[ci]);
      }
#pragma unroll
      for (int g = 0; g < XU; ++g) sx[threadIdx.x * XP + g] = pack16<T>(acc + g * EPV);
    }
    __syncthreads();
    uint4 *gx = reinterpret_cast<uint4 *>(dx + r0 * CIN);
    for (int i = threadIdx.x; i < nr * XU; i += 256) gx[i] = sx[(i / XU) * XP + i % XU];
  }
}

// partial[split][k][ci] ; grid (pairs/256, nsplit)
template <typename T>
__global__ void head_wgrad_kernel(const T *__restrict__ x, int ldx, const float *__restrict__ dout, int lddo,
                                  float *__restrict__ part, int Cin, int nsel, int64_t rows) {
  const int pair = blockIdx.x * blockDim.x + threadIdx.x;
  if (pair >= Cin * nsel) return;
  const int ci = pair % Cin, k = pair / Cin;
  const int64_t per = cdiv64(rows, gridDim.y);
  const int64_t r0 = (int64_t)blockIdx.y * per, r1 = (r0 + per < rows) ? r0 + per : rows;
  float acc = 0.f;
  for (int64_t r = r0; r < r1; ++r) acc = __builtin_fmaf(dout[r * lddo + k], ld_f<T>(x + r * ldx + ci), acc);
  part[((int64_t)blockIdx.y * nsel + k) * Cin + ci] = acc;
}

// Wide head (32 input channels, 33..128 evaluated classes - the full 105-class head of a pre-training step or of a TTA run whose
// model-output modifier is user code, so that map_label cannot be folded into the head): round 5.  The one-thread-per-output
// kernels above walk a 420-byte row per lane (head_dgrad_kernel 9.1 ms, head_wgrad_kernel 11.9 ms per 2 x 128^3 step); here a
// workgroup stages 64 rows of dout (one contiguous run) and of x through LDS.
constexpr int HWD_ROWS = 64, HWD_MAXK = 128;
// n contiguous floats -> LDS rows of `cols` values at pitch LDP: EIGHT loads in flight per thread before the first LDS store (a
// load - store loop pays the memory latency once per element: the first version of these kernels spent 40 us per tile in it)
__device__ __forceinline__ void hwd_stage_rows(float *tile, const float *src, int n, int cols, int LDP) {
  for (int e0 = threadIdx.x; e0 < n; e0 += 256 * 8) {
    float tmp[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + 256 * u;
      tmp[u] = e < n ? src[e] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + 256 * u;
      if (e < n) {
        const int v = e / cols;
        tile[v * LDP + (e - v * cols)] = tmp[u];
      }
    }
  }
}
//   dx[r][ci] = sum_k dout[r][k] w[sel k][ci]: thread = (row, 8-channel group); dout row element broadcast to the row's 4 threads,
//   weight rows broadcast to all rows
template <typename T>
__global__ __launch_bounds__(256) void head_dgrad_wide_kernel(const float *__restrict__ dout, int lddo, const float *__restrict__ w,
                                                            const int *__restrict__ sel, int nsel, T *__restrict__ dx, int lddx,
                                                            int64_t rows) {
  extern __shared__ float hw_smem[];
  float *sw = hw_smem;                    // [nsel][32]
  float *sg = hw_smem + HWD_MAXK * 32;    // [64][nsel | 1]
  const int LDP = nsel | 1;
  for (int i = threadIdx.x; i < nsel * 32; i += 256) sw[i] = w[(int64_t)(sel ? sel[i >> 5] : i >> 5) * 32 + (i & 31)];
  const int r = threadIdx.x >> 2, g = threadIdx.x & 3;
  const int64_t ntile = (rows + HWD_ROWS - 1) / HWD_ROWS;
  for (int64_t t = blockIdx.x; t < ntile; t += gridDim.x) {
    const int64_t r0 = t * HWD_ROWS;
    const int nv = rows - r0 < HWD_ROWS ? (int)(rows - r0) : HWD_ROWS;
    __syncthreads();
    if (lddo == nsel) {
      hwd_stage_rows(sg, dout + r0 * lddo, nv * nsel, nsel, LDP);
    } else {
      for (int e = threadIdx.x; e < nv * nsel; e += 256) {
        const int v = e / nsel, k = e - v * nsel;
        sg[v * LDP + k] = dout[(r0 + v) * lddo + k];
      }
    }
    __syncthreads();
    if (r < nv) {
      float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      const float *gr = sg + r * LDP;
      for (int k = 0; k < nsel; ++k) {
        const float gk = gr[k];
        const float4 w0 = *reinterpret_cast<const float4 *>(sw + k * 32 + 8 * g), w1 = *reinterpret_cast<const float4 *>(sw + k * 32 + 8 * g + 4);
        acc[0] = __builtin_fmaf(gk, w0.x, acc[0]);
        acc[1] = __builtin_fmaf(gk, w0.y, acc[1]);
        acc[2] = __builtin_fmaf(gk, w0.z, acc[2]);
        acc[3] = __builtin_fmaf(gk, w0.w, acc[3]);
        acc[4] = __builtin_fmaf(gk, w1.x, acc[4]);
        acc[5] = __builtin_fmaf(gk, w1.y, acc[5]);
        acc[6] = __builtin_fmaf(gk, w1.z, acc[6]);
        acc[7] = __builtin_fmaf(gk, w1.w, acc[7]);
      }
      T *o = dx + (r0 + r) * lddx + 8 * g;
#pragma unroll
      for (int q = 0; q < 8; ++q) st_f<T>(o + q, acc[q]);
    }
  }
}

//   part[split][k][ci] = sum over the split's rows of dout[r][k] x[r][ci]: thread = (ci, class residue t >> 5 of 8), accumulators for
//   classes (t >> 5) + 8 j; grid.x = splits, each a contiguous range of 64-row tiles (deterministic: fixed order inside a split,
//   reduce_splits_kernel adds the splits in order)
template <typename T>
__global__ __launch_bounds__(256) void head_wgrad_wide_kernel(const T *__restrict__ x, int ldx, const float *__restrict__ dout,
                                                            int lddo, float *__restrict__ part, int nsel, int64_t rows) {
  extern __shared__ float hw_smem[];
  float *sx = hw_smem;                    // [64][33]
  float *sg = hw_smem + HWD_ROWS * 33;    // [64][nsel | 1]
  const int LDP = nsel | 1;
  const int ci = threadIdx.x & 31, cg = threadIdx.x >> 5;
  constexpr int NJ = HWD_MAXK / 8;
  float acc[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) acc[j] = 0.f;
  const int64_t ntile = (rows + HWD_ROWS - 1) / HWD_ROWS;
  const int64_t per = (ntile + gridDim.x - 1) / gridDim.x;
  const int64_t t0 = (int64_t)blockIdx.x * per, t1 = t0 + per < ntile ? t0 + per : ntile;
  for (int64_t t = t0; t < t1; ++t) {
    const int64_t r0 = t * HWD_ROWS;
    const int nv = rows - r0 < HWD_ROWS ? (int)(rows - r0) : HWD_ROWS;
    __syncthreads();
    {     // x rows: 8 loads in flight per thread (64 x 32 values = 8 per thread)
      float tx[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = threadIdx.x + 256 * u, v = e >> 5, c = e & 31;
        tx[u] = v < nv ? ld_f<T>(x + (r0 + v) * ldx + c) : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = threadIdx.x + 256 * u;
        sx[(e >> 5) * 33 + (e & 31)] = tx[u];
      }
    }
    if (lddo == nsel) {
      hwd_stage_rows(sg, dout + r0 * lddo, nv * nsel, nsel, LDP);
      for (int e = nv * nsel + threadIdx.x; e < HWD_ROWS * nsel; e += 256) {      // ragged last tile: zero rows
        const int v = e / nsel;
        sg[v * LDP + (e - v * nsel)] = 0.f;
      }
    } else {
      for (int e = threadIdx.x; e < HWD_ROWS * nsel; e += 256) {
        const int v = e / nsel, k = e - v * nsel;
        sg[v * LDP + k] = v < nv ? dout[(r0 + v) * lddo + k] : 0.f;
      }
    }
    __syncthreads();
    // branch free: all NJ products per row, also for classes >= nsel (they read the next row's values - the tile is padded by
    // HWD_MAXK floats - into accumulators that are never stored).  A per-class guard turned every product into its own
    // read - wait - fma - branch block: 9.5 ms instead of 1
#pragma unroll 4
    for (int v = 0; v < HWD_ROWS; ++v) {
      const float xv = sx[v * 33 + ci];
      const float *gr = sg + v * LDP + cg;
      float gv[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) gv[j] = gr[8 * j];
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[j] = __builtin_fmaf(gv[j], xv, acc[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j)
    if (cg + 8 * j < nsel) part[((int64_t)blockIdx.x * nsel + cg + 8 * j) * 32 + ci] = acc[j];
}

// ============================================================================ layout converters, argmax/dice
template <typename T>
__global__ void ncdhw_to_ndhwc_kernel(const float *__restrict__ src, T *__restrict__ dst, int C, int64_t V, int ldc,
                                      int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % ldc);
    const int64_t row = i / ldc;
    const int64_t b = row / V, v = row % V;
    st_f<T>(dst + i, c < C ? src[(b * C + c) * V + v] : 0.f);
  }
}
template <typename T>
__global__ void ndhwc_to_ncdhw_kernel(const T *__restrict__ src, float *__restrict__ dst, int C, int64_t V, int ldc,
                                      int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t v = i % V;
    const int c = (int)((i / V) % C);
    const int64_t b = i / (V * C);
    dst[i] = ld_f<T>(src + (b * V + v) * ldc + c);
  }
}

__global__ void argmax_dice_kernel(const float *__restrict__ logits, int ldc, int C, const int64_t *__restrict__ labels,
                                   int64_t *__restrict__ amax, unsigned long long *__restrict__ counts, int64_t total) {
  extern __shared__ unsigned int scnt[];  // [3*C]
  for (int i = threadIdx.x; i < 3 * C; i += blockDim.x) scnt[i] = 0;
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int best = 0;
    if (logits) {
      const float *p = logits + i * ldc;
      float bv = p[0];
      for (int c = 1; c < C; ++c)
        if (p[c] > bv) {  // first maximum wins, as torch.argmax
          bv = p[c];
          best = c;
        }
      if (amax) amax[i] = best;
    } else {
      best = (int)amax[i];  // predictions given
      if ((unsigned)best >= (unsigned)C) best = -1;
    }
    if (labels) {
      const int gt = (int)labels[i];
      if (best >= 0) atomicAdd(&scnt[best], 1u);
      if ((unsigned)gt < (unsigned)C) {
        atomicAdd(&scnt[C + gt], 1u);
        if (gt == best) atomicAdd(&scnt[2 * C + gt], 1u);
      }
    }
  }
  __syncthreads();
  if (labels)
    for (int i = threadIdx.x; i < 3 * C; i += blockDim.x)
      if (scnt[i]) atomicAdd(&counts[i], (unsigned long long)scnt[i]);
}

// argmax over the classes of voxel-major rows that are stored back to back (ldc == C: the sliding-window accumulator,
// 105 classes x 512^3 = 56 GB).  argmax_dice_kernel reads one row per thread - 64 lanes 420 bytes apart, 0.3 TB/s; here a
// workgroup streams 64 rows (one contiguous run, all loads in flight at once) into LDS and scans them from there: four
// threads per row take a quarter of the classes each (odd C: rows C words apart are conflict free), combined in class
// order with the same strict comparison, so the first maximum wins as before.
constexpr int AR_MAXC = 112;
constexpr int AR_RUN = (64 * AR_MAXC + 255) / 256;
template <typename ACC>
__global__ __launch_bounds__(256) void argmax_rows_kernel(const ACC *__restrict__ logits, int C, int64_t *__restrict__ amax,
                                                          int64_t total) {
  extern __shared__ float ar_tile[];          // [64][C]
  __shared__ float pv[4][64];
  __shared__ int pi[4][64];
  const int vox = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int cq = (C + 3) >> 2;
  const int64_t ntile = (total + 63) >> 6;
  for (int64_t t = blockIdx.x; t < ntile; t += gridDim.x) {
    const int64_t v0 = t << 6;
    const int nv = total - v0 < 64 ? (int)(total - v0) : 64;
    const int n = nv * C;
    const ACC *lp = logits + v0 * C;
    if constexpr (sizeof(ACC) == 4) {
      float r[AR_RUN];
#pragma unroll
      for (int j = 0; j < AR_RUN; ++j) {
        const int i = (int)threadIdx.x + 256 * j;
        r[j] = i < n ? ld_f<ACC>(lp + i) : 0.f;
      }
      __syncthreads();                        // previous tile scanned
#pragma unroll
      for (int j = 0; j < AR_RUN; ++j) {
        const int i = (int)threadIdx.x + 256 * j;
        if (i < n) ar_tile[i] = r[j];
      }
    } else {
      // 16-bit rows: two classes per 32-bit load (a tile starts at voxel 64 t: 4-byte aligned for any C); the odd last
      // half of the last tile is read on its own - the word would reach past the end of the buffer
      constexpr int RUN2 = (AR_RUN + 1) / 2;
      const unsigned short *hp = reinterpret_cast<const unsigned short *>(lp);
      unsigned r[RUN2];
#pragma unroll
      for (int j = 0; j < RUN2; ++j) {
        const int i = 2 * ((int)threadIdx.x + 256 * j);
        r[j] = i + 1 < n ? *reinterpret_cast<const unsigned *>(hp + i) : (i < n ? (unsigned)hp[i] : 0u);
      }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < RUN2; ++j) {
        const int i = 2 * ((int)threadIdx.x + 256 * j);
        if (i < n) ar_tile[i] = f16_to_f32((unsigned short)(r[j] & 0xffffu));
        if (i + 1 < n) ar_tile[i + 1] = f16_to_f32((unsigned short)(r[j] >> 16));
      }
    }
    __syncthreads();
    if (vox < nv) {
      const float *row = ar_tile + vox * C;
      const int c0 = q * cq, c1 = min(C, c0 + cq);
      // quarter 0 starts from class 0 as the sequential scan does; the others from "nothing yet" (-inf, replaced by
      // anything greater), so that a NaN inside a quarter is passed over exactly as in the sequential scan
      float bv = q == 0 ? row[0] : -__builtin_inff();
      int best = c0 < C ? c0 : C - 1;
      for (int c = q == 0 ? 1 : c0; c < c1; ++c)
        if (row[c] > bv) {
          bv = row[c];
          best = c;
        }
      pv[q][vox] = bv;
      pi[q][vox] = best;
    }
    __syncthreads();
    if (threadIdx.x < (unsigned)nv) {
      float bv = pv[0][threadIdx.x];
      int best = pi[0][threadIdx.x];
#pragma unroll
      for (int k = 1; k < 4; ++k)
        if (pv[k][threadIdx.x] > bv) {
          bv = pv[k][threadIdx.x];
          best = pi[k][threadIdx.x];
        }
      amax[v0 + threadIdx.x] = best;
    }
  }
}

// More classes than the LDS tile of argmax_rows_kernel holds (e.g. the 118 classes of TotalSegmentator v2 weights): one WAVE
// per row, lane l scans classes l, l + 64, ...; the partial maxima are combined so that the FIRST class that reaches the
// maximum wins and NaNs are passed over, exactly as the sequential scan does (row[0] = NaN keeps class 0).
template <typename ACC>
__global__ __launch_bounds__(256) void argmax_rows_wide_kernel(const ACC *__restrict__ logits, int C, int64_t *__restrict__ amax,
                                                               int64_t total) {
  const int lane = threadIdx.x & 63;
  const int64_t w0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * 4;
  for (int64_t v = w0; v < total; v += nw) {
    const ACC *row = logits + v * C;
    float bv = -__builtin_inff();
    int best = lane < C ? lane : C - 1;
    if (lane == 0) bv = ld_f<ACC>(row);
    for (int c = lane == 0 ? 64 : lane; c < C; c += 64) {
      const float x = ld_f<ACC>(row + c);
      if (x > bv) bv = x, best = c;
    }
    const bool first_nan = __shfl(bv != bv ? 1 : 0, 0, 64) != 0;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      const float ov = __shfl_xor(bv, m, 64);
      const int ob = __shfl_xor(best, m, 64);
      if (ov > bv || (ov == bv && ob < best)) bv = ov, best = ob;
    }
    if (lane == 0) amax[v] = first_nan ? 0 : best;
  }
}

int gs_blocks(int64_t total, int cap = 16384) {
  int64_t b = (total + 255) / 256;
  return (int)(b < cap ? (b > 0 ? b : 1) : cap);
}

// per-channel reductions: >= 32 rows per block (small volumes used to run on 1-4 workgroups) and about 4096 workgroups
// per launch over the whole batch: 16 per CU keep an HBM stream saturated, while the finalize kernels that read the
// B x nblk partial rows stay short (with 2048 blocks per SAMPLE at batch 8 they took 30-84 us each, ~5 ms per epoch)
int reduce_blocks(int64_t V, int B) {
  int64_t b = cdiv64(V, 32);
  int64_t cap = 4096 / (B > 0 ? B : 1);
  if (cap < 128) cap = 128;
  if (cap > 2048) cap = 2048;
  return (int)(b < cap ? (b > 0 ? b : 1) : cap);
}

int wgrad_splits(int64_t nvox) {
  int64_t s = cdiv64(nvox, 4096);
  return (int)(s < 64 ? (s > 0 ? s : 1) : 64);
}

}  // namespace

#define DISPATCH_T(dtype, CALL)                                                        \
  do {                                                                                 \
    if ((dtype) == DGTTA_F32) {                                                        \
      typedef float T;                                                                 \
      CALL;                                                                            \
    } else if ((dtype) == DGTTA_BF16) {                                                \
      typedef bf16_t T;                                                                \
      CALL;                                                                            \
    } else if ((dtype) == DGTTA_F16) {                                                 \
      typedef f16_t T;                                                                 \
      CALL;                                                                            \
    } else {                                                                           \
      dgtta_set_error("bad dtype %d", (int)(dtype));                                   \
      return DGTTA_ERR_BADARG;                                                         \
    }                                                                                  \
  } while (0)

// MFMA implementations (conv_mfma.hip); return DGTTA_ERR_UNSUPPORTED when the shape is not covered.
int conv3_fwd_mfma(const void *x, int ldx, const void *w_kmajor, int mirror, const float *bias, void *y, int ldy, int B,
                   int Cin, int Cout, int CinP, int CoutP, int Di, int Hi, int Wi, int stride, int dtype,
                   hipStream_t st, double *stats, RowsGstCtx *gst = nullptr, long long xkh = 0, bool dry = false);
bool conv3_wgrad_blocked_ok(int B, int Cin, int Cout, int D, int H, int W, int dtype);
int64_t conv3_mfma_max_tiles(int Do, int Ho, int Wo);
int conv3_dgrad_s2_mfma(const void *dy, int lddy, const void *w_kmajor, void *dx, int lddx, int B, int Cin, int Cout,
                        int CinP, int CoutP, int Di, int Hi, int Wi, int accumulate, int dtype, hipStream_t st);
size_t convT_packed_bytes(int CinP, int CoutP, int dtype);
int convT_fwd_mfma(const void *x, int ldx, const float *w_t, const float *bias, void *out, int ldo, void *ws, int B, int Cin,
                   int Cout, int Di, int Hi, int Wi, int dtype, hipStream_t st);
int convT_dgrad_mfma(const void *dout, int lddo, const float *w_t, void *dx, int lddx, void *ws, int B, int Cin, int Cout,
                     int Di, int Hi, int Wi, int dtype, hipStream_t st);
int convT_wgrad_mfma(const void *x, int ldx, const void *dout, int lddo, float *dw_t, void *ws, size_t ws_bytes, int B,
                     int Cin, int Cout, int Di, int Hi, int Wi, int accumulate, int dtype, hipStream_t st, float *bias_part,
                     size_t bias_part_bytes, int *bias_units);
int convT_bias_finalize(const float *part, int units, int Cout, float *db, int accumulate, hipStream_t st);
int conv3_wgrad_mfma(const void *x, int ldx, const void *dy, int lddy, float *dw_t, float *db, void *ws, size_t ws_bytes,
                     int B, int Cin, int Cout, int Di, int Hi, int Wi, int stride, int accumulate, int dtype,
                     hipStream_t st, long long xkh = 0);

static size_t esize(int dtype) { return dtype == DGTTA_F32 ? 4 : 2; }
static const void *wb_of(const void *wpack, int CinP, int CoutP, int dtype) {
  return (const char *)wpack + (size_t)27 * CinP * CoutP * esize(dtype);
}

size_t conv_image_bytes(int CinP, int CoutP, int dtype);
size_t conv_imgB_offset_bytes(int CinP, int CoutP, int dtype);
int conv_pack_images(const float *w_t, void *img, int Cin, int Cout, int CinP, int CoutP, int dtype, hipStream_t st);

// blob = [wf | wb] (general kernels) followed by [imgF | imgB] (LDS-image order for the MFMA kernels)
static const void *img_of(const void *wpack, int CinP, int CoutP, int dtype) {
  return (const char *)wpack + (size_t)2 * 27 * CinP * CoutP * esize(dtype);
}
static const void *imgB_of(const void *wpack, int CinP, int CoutP, int dtype) {
  return (const char *)img_of(wpack, CinP, CoutP, dtype) + conv_imgB_offset_bytes(CinP, CoutP, dtype);
}

extern "C" size_t dgtta_conv3d_packed_bytes(int CinP, int CoutP, int dtype) {
  if (CinP <= 0 || CoutP <= 0) return 0;      // a size query of an empty problem (the launchers reject it with DGTTA_ERR_BADARG)
  return (size_t)2 * 27 * CinP * CoutP * esize(dtype) + conv_image_bytes(CinP, CoutP, dtype);
}

extern "C" int dgtta_conv3d_pack_weights(const float *w_t, void *wpack, int Cin, int Cout, int CinP, int CoutP,
                                         int dtype, void *stream) {
  DG_REQUIRE(w_t && wpack, DGTTA_ERR_BADARG, "pack_weights: null pointer");
  void *wf = wpack;
  void *wb = const_cast<void *>(wb_of(wpack, CinP, CoutP, dtype));
  DG_REQUIRE(Cin > 0 && Cout > 0 && CinP >= Cin && CoutP >= Cout, DGTTA_ERR_BADARG, "pack_weights: bad channel counts");
  const int64_t n = (int64_t)27 * CinP * CoutP;
  DISPATCH_T(dtype, hipLaunchKernelGGL((pack_weights_kernel<T>), dim3(gs_blocks(n)), dim3(256), 0, (hipStream_t)stream,
                                       w_t, (T *)wf, (T *)wb, Cin, Cout, CinP, CoutP));
  DG_CHECK_LAUNCH("pack_weights_kernel");
  if (CinP % (dtype == DGTTA_F32 ? 8 : 16) == 0 && CoutP % (dtype == DGTTA_F32 ? 8 : 16) == 0)
    return conv_pack_images(w_t, const_cast<void *>(img_of(wpack, CinP, CoutP, dtype)), Cin, Cout, CinP, CoutP, dtype,
                            (hipStream_t)stream);
  return DGTTA_OK;
}

static int out_dim(int i, int s) { return (i + 2 - 3) / s + 1; }

// statistics buffer: [256-byte header: int64 nblk][partial sums: B x nblk x Cout x 2 doubles]
extern "C" size_t dgtta_conv3d_stats_bytes(int B, int Cout, int Do, int Ho, int Wo) {
  if (B <= 0 || Cout <= 0 || Do <= 0 || Ho <= 0 || Wo <= 0) return 0;      // a size query of an empty problem (the launchers reject it with DGTTA_ERR_BADARG)
  int64_t nb = conv3_mfma_max_tiles(Do, Ho, Wo);
  const int64_t rb = reduce_blocks((int64_t)Do * Ho * Wo, B);
  if (rb > nb) nb = rb;
  return 256 + (size_t)B * nb * Cout * 2 * sizeof(double);
}

__global__ void set_header_kernel(long long *hdr, long long v) { hdr[0] = v; }

static int k3_fwd(const void *x, int ldx, long long x_block_stride, const void *wpack, const float *bias, void *y, int ldy,
                  void *stats, int B, int Cin, int Cout, int CinP, int CoutP, int Di, int Hi, int Wi, int stride, int dtype, int impl,
                  void *stream);

extern "C" int dgtta_conv3d_k3_fwd(const void *x, int ldx, const void *wpack, const float *bias, void *y, int ldy,
                                   void *stats, int B, int Cin, int Cout, int CinP, int CoutP, int Di, int Hi, int Wi,
                                   int stride, int dtype, int impl, void *stream) {
  return k3_fwd(x, ldx, 0, wpack, bias, y, ldy, stats, B, Cin, Cout, CinP, CoutP, Di, Hi, Wi, stride, dtype, impl, stream);
}

// Round 6: x as 32-channel BLOCKS - block c of the input channels is a dense tensor [B][D][H][W][32] (ldx = 32) at element
// offset c * x_block_stride from x.  The level-0 concat buffer of the U-Net is kept that way ([up | skip] as two planes): the
// kernels that read ONE half of it (the stride-2 conv of the skip, the transposed conv's backward) then use whole 128-byte lines.
// Only the D-ring kernels read this layout (64 input channels, 16-bit storage, launches they take): ask
// dgtta_conv3d_k3_blocked_supported first; anything else returns DGTTA_ERR_UNSUPPORTED and launches nothing.
extern "C" int dgtta_conv3d_k3_blocked_supported(int B, int Cin, int Cout, int D, int H, int W, int dtype) {
  if (B <= 0 || Cin != 64 || Cout <= 0 || Cout % 32 || D <= 0 || H <= 0 || W <= 0) return 0;
  if (dtype != DGTTA_BF16 && dtype != DGTTA_F16) return 0;
  const int rc = conv3_fwd_mfma((const void *)16, 32, (const void *)16, 0, nullptr, (void *)16, Cout, B, Cin, Cout, Cin, Cout, D, H, W, 1,
                                dtype, nullptr, nullptr, nullptr, (long long)B * D * H * W * 32, true);
  return rc == DGTTA_OK && conv3_wgrad_blocked_ok(B, Cin, Cout, D, H, W, dtype);
}

extern "C" int dgtta_conv3d_k3_fwd_blocked(const void *x, long long x_block_stride, const void *wpack, const float *bias, void *y,
                                           int ldy, void *stats, int B, int Cin, int Cout, int CinP, int CoutP, int Di, int Hi,
                                           int Wi, int dtype, void *stream) {
  DG_REQUIRE(x_block_stride > 0 && x_block_stride % 8 == 0, DGTTA_ERR_BADARG, "conv3d_k3_fwd_blocked: block stride must be a positive multiple of 8 elements");
  return k3_fwd(x, 32, x_block_stride, wpack, bias, y, ldy, stats, B, Cin, Cout, CinP, CoutP, Di, Hi, Wi, 1, dtype, 0, stream);
}

static int k3_fwd(const void *x, int ldx, long long x_block_stride, const void *wpack, const float *bias, void *y, int ldy,
                  void *stats, int B, int Cin, int Cout, int CinP, int CoutP, int Di, int Hi, int Wi, int stride, int dtype, int impl,
                  void *stream) {
  DG_REQUIRE(x && wpack && y, DGTTA_ERR_BADARG, "conv3d_k3_fwd: null pointer");
  const void *wf = wpack;
  DG_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && CinP >= Cin && CoutP >= Cout && Di > 0 && Hi > 0 && Wi > 0,
             DGTTA_ERR_BADARG, "conv3d_k3_fwd: bad dims");
  DG_REQUIRE(stride == 1 || stride == 2, DGTTA_ERR_UNSUPPORTED, "conv3d_k3_fwd: stride %d", stride);
  DG_REQUIRE((x_block_stride ? 2 * ldx : ldx) >= Cin && ldy >= Cout, DGTTA_ERR_BADARG, "conv3d_k3_fwd: ld < C");
  hipStream_t st = (hipStream_t)stream;
  if (impl != 1) {
    int rc = conv3_fwd_mfma(x, ldx, img_of(wpack, CinP, CoutP, dtype), 0, bias, y, ldy, B, Cin, Cout, CinP, CoutP, Di, Hi,
                            Wi, stride, dtype, st, (double *)stats, nullptr, x_block_stride);
    if (rc != DGTTA_ERR_UNSUPPORTED) return rc;
    DG_REQUIRE(x_block_stride == 0, DGTTA_ERR_UNSUPPORTED,
               "conv3d_k3_fwd_blocked: only the D-ring kernel reads x as 32-channel planes (ask dgtta_conv3d_k3_blocked_supported)");
    DG_REQUIRE(impl == 0, DGTTA_ERR_UNSUPPORTED, "conv3d_k3_fwd: shape not covered by the MFMA kernel");
  }
  const int Do = out_dim(Di, stride), Ho = out_dim(Hi, stride), Wo = out_dim(Wi, stride);
  const int64_t total = (int64_t)B * Do * Ho * Wo * Cout;
  DISPATCH_T(dtype, hipLaunchKernelGGL((conv3_fwd_ref_kernel<T>), dim3(gs_blocks(total, 1 << 20)), dim3(256), 0, st,
                                       (const T *)x, ldx, (const T *)wf, bias, (T *)y, ldy, Cin, Cout, CinP, CoutP, Di,
                                       Hi, Wi, Do, Ho, Wo, stride, total));
  DG_CHECK_LAUNCH("conv3_fwd_ref_kernel");
  if (stats) {   // general kernel: statistics by a separate reduction pass, same buffer layout
    const int64_t V = (int64_t)Do * Ho * Wo;
    const int nblk = reduce_blocks(V, B);
    hipLaunchKernelGGL(set_header_kernel, dim3(1), dim3(1), 0, st, (long long *)stats, (long long)nblk);
    DISPATCH_T(dtype, hipLaunchKernelGGL((chan_reduce_kernel<T, 0>), dim3(nblk, B), dim3(256), 0, st, (const T *)y, ldy,
                                         (const T *)nullptr, 0, nullptr, nullptr, nullptr, 0.f, (double *)stats + 32,
                                         Cout, V));
    DG_CHECK_LAUNCH("chan_reduce_kernel<0>");
  }
  return DGTTA_OK;
}

// gst: the InstanceNorm-backward context of dgtta_conv3d_k3_dgrad_gstats (null for the plain data gradient); it travels down
// the dispatch chain as an argument and only the ring / row-reuse launchers act on it
static int k3_dgrad(const void *dy, int lddy, const void *wpack, void *dx, int lddx, int B, int Cin, int Cout, int CinP, int CoutP,
                    int Di, int Hi, int Wi, int stride, int accumulate, int dtype, int impl, void *stream, RowsGstCtx *gst) {
  DG_REQUIRE(dy && wpack && dx, DGTTA_ERR_BADARG, "conv3d_k3_dgrad: null pointer");
  const void *wb = wb_of(wpack, CinP, CoutP, dtype);
  DG_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && CinP >= Cin && CoutP >= Cout && Di > 0 && Hi > 0 && Wi > 0,
             DGTTA_ERR_BADARG, "conv3d_k3_dgrad: bad dims");
  DG_REQUIRE(stride == 1 || stride == 2, DGTTA_ERR_UNSUPPORTED, "conv3d_k3_dgrad: stride %d", stride);
  DG_REQUIRE(lddx >= Cin && lddy >= Cout, DGTTA_ERR_BADARG, "conv3d_k3_dgrad: ld < C");
  hipStream_t st = (hipStream_t)stream;
  if (impl != 1 && stride == 1 && !accumulate) {
    // stride-1 data gradient == forward conv of dy with the mirrored, transposed weights (wb)
    // (imgB: N = ci, K = co; taps mirrored)
    int rc = conv3_fwd_mfma(dy, lddy, imgB_of(wpack, CinP, CoutP, dtype), 1, nullptr, dx, lddx, B, Cout, Cin, CoutP, CinP, Di,
                            Hi, Wi, 1, dtype, st, nullptr, gst);
    if (rc != DGTTA_ERR_UNSUPPORTED) return rc;
    DG_REQUIRE(impl == 0, DGTTA_ERR_UNSUPPORTED, "conv3d_k3_dgrad: shape not covered by the MFMA kernel");
  }
  if (impl != 1 && stride == 2 && !((Di | Hi | Wi) & 1)) {
    int rc = conv3_dgrad_s2_mfma(dy, lddy, imgB_of(wpack, CinP, CoutP, dtype), dx, lddx, B, Cin, Cout, CinP, CoutP, Di, Hi, Wi,
                                 accumulate, dtype, st);
    if (rc != DGTTA_ERR_UNSUPPORTED) return rc;
    DG_REQUIRE(impl == 0, DGTTA_ERR_UNSUPPORTED, "conv3d_k3_dgrad: shape not covered by the MFMA kernel");
  }
  const int Do = out_dim(Di, stride), Ho = out_dim(Hi, stride), Wo = out_dim(Wi, stride);
  const int64_t total = (int64_t)B * Di * Hi * Wi * Cin;
  DISPATCH_T(dtype, hipLaunchKernelGGL((conv3_dgrad_ref_kernel<T>), dim3(gs_blocks(total, 1 << 20)), dim3(256), 0, st,
                                       (const T *)dy, lddy, (const T *)wb, (T *)dx, lddx, Cin, Cout, CinP, CoutP, Di, Hi,
                                       Wi, Do, Ho, Wo, stride, accumulate, total));
  DG_CHECK_LAUNCH("conv3_dgrad_ref_kernel");
  return DGTTA_OK;
}

extern "C" int dgtta_conv3d_k3_dgrad(const void *dy, int lddy, const void *wpack, void *dx, int lddx, int B, int Cin,
                                     int Cout, int CinP, int CoutP, int Di, int Hi, int Wi, int stride, int accumulate,
                                     int dtype, int impl, void *stream) {
  return k3_dgrad(dy, lddy, wpack, dx, lddx, B, Cin, Cout, CinP, CoutP, Di, Hi, Wi, stride, accumulate, dtype, impl, stream, nullptr);
}


extern "C" int dgtta_conv3d_k3_dgrad_gstats(const void *dy, int lddy, const void *wpack, void *dx, int lddx, int B, int Cin,
                                            int Cout, int CinP, int CoutP, int Di, int Hi, int Wi, const void *y_prev,
                                            int ldy_prev, const float *mean_rstd_prev, const float *gamma_prev,
                                            const float *beta_prev, float slope, void *gstats, size_t gstats_bytes,
                                            int *h_produced, int dtype, int impl, void *stream) {
  DG_REQUIRE(y_prev && mean_rstd_prev && gamma_prev && beta_prev && gstats && h_produced, DGTTA_ERR_BADARG,
             "conv3d_k3_dgrad_gstats: null pointer");
  DG_REQUIRE(ldy_prev >= Cin, DGTTA_ERR_BADARG, "conv3d_k3_dgrad_gstats: ldy_prev < Cin");
  DG_REQUIRE(B > 0 && Cin > 0 && Di > 0 && Hi > 0 && Wi > 0 && gstats_bytes >= dgtta_conv3d_stats_bytes(B, Cin, Di, Hi, Wi),
             DGTTA_ERR_WORKSPACE, "conv3d_k3_dgrad_gstats: statistics buffer too small");
  RowsGstCtx ctx{y_prev, ldy_prev, mean_rstd_prev, gamma_prev, beta_prev, slope, (double *)gstats, 0};
  // only the row-reuse kernel (16-bit storage, large whole-tile volumes) knows the fused form; any other dispatch ignores
  // the context and *h_produced stays 0: the caller then runs the plain dgtta_instnorm_lrelu_bwd
  const bool fuse = dtype != DGTTA_F32 && impl != 1 && dgtta_switches().in_gstats != '0';
  const int rc = k3_dgrad(dy, lddy, wpack, dx, lddx, B, Cin, Cout, CinP, CoutP, Di, Hi, Wi, 1, 0, dtype, impl, stream,
                          fuse ? &ctx : nullptr);
  *h_produced = rc == DGTTA_OK ? ctx.produced : 0;
  return rc;
}

size_t conv3_wgrad_mfma_ws_bytes(int B, int Cin, int Cout, int D, int H, int W);
size_t conv3_wgrad_split_extra_bytes(int B, int Cin, int Cout, int D, int H, int W, int stride);

// workspace layout: [bias partials][main: split partials of the VALU kernel | slabs of the MFMA kernel]
static size_t wgrad_bias_bytes(int B, int Cout, int Do, int Ho, int Wo) {
  return align_up((size_t)B * reduce_blocks((int64_t)Do * Ho * Wo, B) * Cout * 2 * sizeof(double), 256);
}

extern "C" size_t dgtta_conv3d_wgrad_ws_bytes(int B, int Cin, int Cout, int Do, int Ho, int Wo) {
  if (B <= 0 || Cin <= 0 || Cout <= 0 || Do <= 0 || Ho <= 0 || Wo <= 0) return 0;      // a size query of an empty problem (the launchers reject it with DGTTA_ERR_BADARG)
  const int64_t nvox = (int64_t)B * Do * Ho * Wo;
  size_t a = align_up((size_t)wgrad_splits(nvox) * Cout * Cin * 27 * sizeof(float), 256);
  size_t c = align_up(conv3_wgrad_mfma_ws_bytes(B, Cin, Cout, Do, Ho, Wo), 256);   // stride 1: input dims == output dims
  return wgrad_bias_bytes(B, Cout, Do, Ho, Wo) + (a > c ? a : c);
}

// workspace that lets an fp32 stride-1 weight gradient run as six 16-bit launches on bf16 split planes (conv_wgrad.hip): the
// plain workspace followed by three planes of x and three of dy
extern "C" size_t dgtta_conv3d_wgrad_split_ws_bytes(int B, int Cin, int Cout, int Do, int Ho, int Wo, int stride) {
  if (B <= 0 || Cin <= 0 || Cout <= 0 || Do <= 0 || Ho <= 0 || Wo <= 0 || (stride != 1 && stride != 2)) return 0;
  const size_t base = dgtta_conv3d_wgrad_ws_bytes(B, Cin, Cout, Do, Ho, Wo);
  // the split planes start behind the 256-aligned slab region of the MAIN part (workspace = [bias partials][main])
  return base + 256 + conv3_wgrad_split_extra_bytes(B, Cin, Cout, Do, Ho, Wo, stride);
}

static int bias_grad(const void *dy, int lddy, float *db, void *ws, int B, int C, int64_t V, int accumulate, int dtype,
                     hipStream_t st) {
  const int nblk = reduce_blocks(V, B);
  double *partial = (double *)ws;
  DISPATCH_T(dtype, (launch_chan_reduce<T, 2>(dy, lddy, nullptr, 0, nullptr, nullptr, nullptr, 0.f, partial, nblk, B, C, V, st)));
  DG_CHECK_LAUNCH("chan_reduce_kernel<2>");
  hipLaunchKernelGGL(bias_finalize_kernel, dim3(C), dim3(64), 0, st, partial, nblk, B, C, db, accumulate);
  DG_CHECK_LAUNCH("bias_finalize_kernel");
  return DGTTA_OK;
}

static int k3_wgrad(const void *x, int ldx, long long x_block_stride, const void *dy, int lddy, float *dw_t, float *db, void *ws,
                    size_t ws_bytes, int B, int Cin, int Cout, int Di, int Hi, int Wi, int stride, int accumulate, int dtype, int impl,
                    void *stream);

extern "C" int dgtta_conv3d_k3_wgrad(const void *x, int ldx, const void *dy, int lddy, float *dw_t, float *db, void *ws,
                                     size_t ws_bytes, int B, int Cin, int Cout, int Di, int Hi, int Wi, int stride,
                                     int accumulate, int dtype, int impl, void *stream) {
  return k3_wgrad(x, ldx, 0, dy, lddy, dw_t, db, ws, ws_bytes, B, Cin, Cout, Di, Hi, Wi, stride, accumulate, dtype, impl, stream);
}

// weight gradient of a stride-1 conv whose x lies as 32-channel blocks (see dgtta_conv3d_k3_fwd_blocked); same workspace
extern "C" int dgtta_conv3d_k3_wgrad_blocked(const void *x, long long x_block_stride, const void *dy, int lddy, float *dw_t, float *db,
                                             void *ws, size_t ws_bytes, int B, int Cin, int Cout, int Di, int Hi, int Wi,
                                             int accumulate, int dtype, void *stream) {
  DG_REQUIRE(x_block_stride > 0 && x_block_stride % 8 == 0, DGTTA_ERR_BADARG, "conv3d_k3_wgrad_blocked: block stride must be a positive multiple of 8 elements");
  return k3_wgrad(x, 32, x_block_stride, dy, lddy, dw_t, db, ws, ws_bytes, B, Cin, Cout, Di, Hi, Wi, 1, accumulate, dtype, 0, stream);
}

static int k3_wgrad(const void *x, int ldx, long long x_block_stride, const void *dy, int lddy, float *dw_t, float *db, void *ws,
                    size_t ws_bytes, int B, int Cin, int Cout, int Di, int Hi, int Wi, int stride, int accumulate, int dtype, int impl,
                    void *stream) {
  DG_REQUIRE(x && dy && dw_t && ws, DGTTA_ERR_BADARG, "conv3d_k3_wgrad: null pointer");
  DG_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && Di > 0 && Hi > 0 && Wi > 0, DGTTA_ERR_BADARG, "conv3d_k3_wgrad: bad dims");
  DG_REQUIRE(stride == 1 || stride == 2, DGTTA_ERR_UNSUPPORTED, "conv3d_k3_wgrad: stride %d", stride);
  const int Do = out_dim(Di, stride), Ho = out_dim(Hi, stride), Wo = out_dim(Wi, stride);
  DG_REQUIRE(ws_bytes >= dgtta_conv3d_wgrad_ws_bytes(B, Cin, Cout, Do, Ho, Wo), DGTTA_ERR_WORKSPACE,
             "conv3d_k3_wgrad: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int64_t nvox = (int64_t)B * Do * Ho * Wo;
  const int nsplit = wgrad_splits(nvox);
  const size_t bias_bytes = wgrad_bias_bytes(B, Cout, Do, Ho, Wo);
  void *ws2 = ws;                                   // bias partials
  float *part = (float *)((char *)ws + bias_bytes); // main region
  bool done = false;
  if (impl != 1) {
    int rc = conv3_wgrad_mfma(x, ldx, dy, lddy, dw_t, nullptr, part, ws_bytes - bias_bytes, B, Cin, Cout, Di, Hi, Wi,
                              stride, accumulate, dtype, st, x_block_stride);
    if (rc == DGTTA_OK) done = true;
    else if (rc != DGTTA_ERR_UNSUPPORTED) return rc;
    else DG_REQUIRE(impl == 0 && x_block_stride == 0, DGTTA_ERR_UNSUPPORTED,
                    "conv3d_k3_wgrad: shape not covered by the MFMA kernel (x as 32-channel planes: only the ring sweep, ask dgtta_conv3d_k3_blocked_supported)");
  }
  if (!done) {
    DISPATCH_T(dtype, hipLaunchKernelGGL((conv3_wgrad_ref_kernel<T>), dim3(cdiv(Cin * Cout, 256), 27, nsplit), dim3(256),
                                         0, st, (const T *)x, ldx, (const T *)dy, lddy, part, Cin, Cout, B, Di, Hi, Wi,
                                         Do, Ho, Wo, stride));
    DG_CHECK_LAUNCH("conv3_wgrad_ref_kernel");
    const int64_t n = (int64_t)Cout * Cin * 27;
    hipLaunchKernelGGL(reduce_splits_kernel, dim3(gs_blocks(n)), dim3(256), 0, st, part, dw_t, n, nsplit, accumulate);
    DG_CHECK_LAUNCH("reduce_splits_kernel");
  }
  if (db) return bias_grad(dy, lddy, db, ws2, B, Cout, (int64_t)Do * Ho * Wo, accumulate, dtype, st);
  return DGTTA_OK;
}

extern "C" size_t dgtta_instnorm_ws_bytes(int B, int C, int64_t V) {
  if (B <= 0 || C <= 0 || V <= 0) return 0;      // a size query of an empty problem (the launchers reject it with DGTTA_ERR_BADARG)
  return align_up((size_t)B * reduce_blocks(V, B) * C * 2 * sizeof(double), 256) +
         align_up((size_t)B * C * 2 * sizeof(float), 256);
}

extern "C" int dgtta_instnorm_lrelu_fwd(const void *y, int ldy, const void *stats, const float *gamma, const float *beta,
                                        float *mean_rstd, void *z, int ldz, void *ws, size_t ws_bytes, int B, int C,
                                        int64_t V, float eps, float slope, int dtype, void *stream) {
  // z == NULL (round 5): the statistics only - the caller applies them itself (dgtta_feature_window_accumulate_norm)
  DG_REQUIRE(y && gamma && beta && mean_rstd && ws, DGTTA_ERR_BADARG, "instnorm_lrelu_fwd: null pointer");
  DG_REQUIRE(B > 0 && C > 0 && V > 0 && ldy >= C && (!z || ldz >= C), DGTTA_ERR_BADARG, "instnorm_lrelu_fwd: bad dims");
  DG_REQUIRE(ws_bytes >= dgtta_instnorm_ws_bytes(B, C, V), DGTTA_ERR_WORKSPACE, "instnorm_lrelu_fwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int nblk = reduce_blocks(V, B);
  if (stats) {   // partial sums came with the conv epilogue (header + partials)
    hipLaunchKernelGGL(in_stats_finalize_kernel, dim3(B * C), dim3(256), 0, st, (const double *)stats + 32,
                       (const long long *)stats, 0, B, C, V, eps, mean_rstd);
  } else {
    double *partial = (double *)ws;
    DISPATCH_T(dtype, (launch_chan_reduce<T, 0>(y, ldy, nullptr, 0, nullptr, nullptr, nullptr, 0.f, partial, nblk, B, C, V, st)));
    DG_CHECK_LAUNCH("chan_reduce_kernel<0>");
    hipLaunchKernelGGL(in_stats_finalize_kernel, dim3(B * C), dim3(256), 0, st, (const double *)partial,
                       (const long long *)nullptr, nblk, B, C, V, eps, mean_rstd);
  }
  DG_CHECK_LAUNCH("in_stats_finalize_kernel");
  if (!z) return DGTTA_OK;
  const int64_t total = (int64_t)B * V * C;
  const int esz = dtype == DGTTA_F32 ? 4 : 2, epv = 16 / esz;
  if (C % epv == 0 && ldy % epv == 0 && ldz % epv == 0 && !((uintptr_t)y & 15) && !((uintptr_t)z & 15) && C <= 2048) {
    const int64_t items = V * (C / epv);
    const int blocks = (int)(cdiv64(items, 256 * 4) < 4096 ? (cdiv64(items, 256 * 4) > 0 ? cdiv64(items, 256 * 4) : 1) : 4096);
    DISPATCH_T(dtype, hipLaunchKernelGGL((in_apply_vec_kernel<T, 0>), dim3(blocks, B), dim3(256), (size_t)C * 2 * 4, st,
                                         (const T *)y, ldy, (const T *)nullptr, 0, mean_rstd, gamma, beta, nullptr, (T *)z,
                                         ldz, C, V, slope, dgtta_switches().in_nt - '0'));
    DG_CHECK_LAUNCH("in_apply_vec_kernel<0>");
    return DGTTA_OK;
  }
  DISPATCH_T(dtype, hipLaunchKernelGGL((in_lrelu_apply_kernel<T>), dim3(gs_blocks(total)), dim3(256), 0, st,
                                       (const T *)y, ldy, mean_rstd, gamma, beta, (T *)z, ldz, C, V, slope, total));
  DG_CHECK_LAUNCH("in_lrelu_apply_kernel");
  return DGTTA_OK;
}

static int instnorm_bwd_impl(const void *gz, int ldgz, const void *y, int ldy, const float *gamma, const float *beta,
                             const float *mean_rstd, void *dy, int lddy, float *dgamma, float *dbeta, const void *gstats,
                             void *ws, size_t ws_bytes, int B, int C, int64_t V, float slope, int accumulate, int dtype,
                             void *stream, const char *name) {
  DG_REQUIRE(gz && y && gamma && beta && mean_rstd && dy && dgamma && dbeta && ws, DGTTA_ERR_BADARG, "%s: null pointer", name);
  DG_REQUIRE(B > 0 && C > 0 && V > 0 && ldy >= C && ldgz >= C && lddy >= C, DGTTA_ERR_BADARG, "%s: bad dims", name);
  DG_REQUIRE(ws_bytes >= dgtta_instnorm_ws_bytes(B, C, V), DGTTA_ERR_WORKSPACE, "%s: workspace too small", name);
  hipStream_t st = (hipStream_t)stream;
  const int nblk = reduce_blocks(V, B);
  double *partial = (double *)ws;
  float *c12 = (float *)((char *)ws + align_up((size_t)B * nblk * C * 2 * sizeof(double), 256));
  if (gstats) {     // the sums came out of the data-gradient kernel that produced gz: no pass over y and gz
    hipLaunchKernelGGL(in_bwd_finalize_gstats_kernel, dim3(C), dim3(256), 0, st, (const double *)gstats, B, C, V, mean_rstd, c12,
                       dgamma, dbeta, accumulate);
    DG_CHECK_LAUNCH("in_bwd_finalize_gstats_kernel");
  } else {
    DISPATCH_T(dtype, (launch_chan_reduce<T, 1>(y, ldy, gz, ldgz, mean_rstd, gamma, beta, slope, partial, nblk, B, C, V, st)));
    DG_CHECK_LAUNCH("chan_reduce_kernel<1>");
    hipLaunchKernelGGL(in_bwd_finalize_kernel, dim3(C), dim3(256), 0, st, partial, nblk, B, C, V, c12, dgamma, dbeta,
                       accumulate);
    DG_CHECK_LAUNCH("in_bwd_finalize_kernel");
  }
  const int64_t total = (int64_t)B * V * C;
  const int esz = dtype == DGTTA_F32 ? 4 : 2, epv = 16 / esz;
  if (C % epv == 0 && ldy % epv == 0 && ldgz % epv == 0 && lddy % epv == 0 && !((uintptr_t)y & 15) && !((uintptr_t)gz & 15) &&
      !((uintptr_t)dy & 15) && C <= 2048) {
    const int64_t items = V * (C / epv);
    const int blocks = (int)(cdiv64(items, 256 * 4) < 4096 ? (cdiv64(items, 256 * 4) > 0 ? cdiv64(items, 256 * 4) : 1) : 4096);
    DISPATCH_T(dtype, hipLaunchKernelGGL((in_apply_vec_kernel<T, 1>), dim3(blocks, B), dim3(256), (size_t)C * 6 * 4, st,
                                         (const T *)y, ldy, (const T *)gz, ldgz, mean_rstd, gamma, beta, c12, (T *)dy, lddy,
                                         C, V, slope, dgtta_switches().in_nt - '0'));
    DG_CHECK_LAUNCH("in_apply_vec_kernel<1>");
    return DGTTA_OK;
  }
  DISPATCH_T(dtype, hipLaunchKernelGGL((in_lrelu_bwd_apply_kernel<T>), dim3(gs_blocks(total)), dim3(256), 0, st,
                                       (const T *)gz, ldgz, (const T *)y, ldy, mean_rstd, gamma, beta, c12, (T *)dy, lddy,
                                       C, V, slope, total));
  DG_CHECK_LAUNCH("in_lrelu_bwd_apply_kernel");
  return DGTTA_OK;
}

extern "C" int dgtta_instnorm_lrelu_bwd(const void *gz, int ldgz, const void *y, int ldy, const float *gamma,
                                        const float *beta, const float *mean_rstd, void *dy, int lddy, float *dgamma,
                                        float *dbeta, void *ws, size_t ws_bytes, int B, int C, int64_t V, float slope,
                                        int accumulate, int dtype, void *stream) {
  return instnorm_bwd_impl(gz, ldgz, y, ldy, gamma, beta, mean_rstd, dy, lddy, dgamma, dbeta, nullptr, ws, ws_bytes, B, C, V,
                           slope, accumulate, dtype, stream, "instnorm_lrelu_bwd");
}

extern "C" int dgtta_instnorm_lrelu_bwd_gstats(const void *gz, int ldgz, const void *y, int ldy, const float *gamma,
                                               const float *beta, const float *mean_rstd, void *dy, int lddy,
                                               float *dgamma, float *dbeta, const void *gstats, void *ws, size_t ws_bytes,
                                               int B, int C, int64_t V, float slope, int accumulate, int dtype,
                                               void *stream) {
  DG_REQUIRE(gstats, DGTTA_ERR_BADARG, "instnorm_lrelu_bwd_gstats: null statistics");
  return instnorm_bwd_impl(gz, ldgz, y, ldy, gamma, beta, mean_rstd, dy, lddy, dgamma, dbeta, gstats, ws, ws_bytes, B, C, V,
                           slope, accumulate, dtype, stream, "instnorm_lrelu_bwd_gstats");
}

static size_t convT_pack_region(int Cin, int Cout, int dtype) {
  const int g = (dtype == DGTTA_F32) ? 8 : 16;
  return align_up(convT_packed_bytes((Cin + g - 1) / g * g, (Cout + g - 1) / g * g, dtype), 256);
}

extern "C" size_t dgtta_convT3d_fwd_ws_bytes(int Cin, int Cout, int dtype) {
  if (Cin <= 0 || Cout <= 0) return 0;
  return convT_pack_region(Cin, Cout, dtype);
}

extern "C" int dgtta_convT3d_k2s2_fwd(const void *x, int ldx, const float *w_t, const float *bias, void *out, int ldo,
                                      void *ws, size_t ws_bytes, int B, int Cin, int Cout, int Di, int Hi, int Wi,
                                      int dtype, int impl, void *stream) {
  DG_REQUIRE(x && w_t && out, DGTTA_ERR_BADARG, "convT3d_k2s2_fwd: null pointer");
  DG_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && Di > 0 && Hi > 0 && Wi > 0 && ldx >= Cin && ldo >= Cout, DGTTA_ERR_BADARG,
             "convT3d_k2s2_fwd: bad dims");
  hipStream_t st = (hipStream_t)stream;
  if (impl != 1 && ws && ws_bytes >= convT_pack_region(Cin, Cout, dtype)) {
    int rc = convT_fwd_mfma(x, ldx, w_t, bias, out, ldo, ws, B, Cin, Cout, Di, Hi, Wi, dtype, st);
    if (rc != DGTTA_ERR_UNSUPPORTED) return rc;
  }
  DG_REQUIRE(impl != 2, DGTTA_ERR_UNSUPPORTED, "convT3d_k2s2_fwd: shape not covered by the MFMA kernel");
  const int64_t total = (int64_t)B * Di * Hi * Wi * 8 * Cout;
  DISPATCH_T(dtype, hipLaunchKernelGGL((convT_fwd_ref_kernel<T>), dim3(gs_blocks(total, 1 << 20)), dim3(256), 0, st,
                                       (const T *)x, ldx, w_t, bias, (T *)out, ldo, Cin, Cout, Di, Hi, Wi, total));
  DG_CHECK_LAUNCH("convT_fwd_ref_kernel");
  return DGTTA_OK;
}

// workspace layout: [bias partials][packed weights][main: split partials (VALU) | slabs (MFMA)]
static size_t convT_bias_region(int B, int Cout, int Di, int Hi, int Wi) {
  return align_up((size_t)B * reduce_blocks((int64_t)Di * Hi * Wi * 8, B) * Cout * 2 * sizeof(double), 256);
}

extern "C" size_t dgtta_convT3d_bwd_ws_bytes(int B, int Cin, int Cout, int Di, int Hi, int Wi) {
  if (B <= 0 || Cin <= 0 || Cout <= 0 || Di <= 0 || Hi <= 0 || Wi <= 0) return 0;      // a size query of an empty problem (the launchers reject it with DGTTA_ERR_BADARG)
  const int64_t nvox = (int64_t)B * Di * Hi * Wi;
  size_t a = align_up((size_t)wgrad_splits(nvox) * Cin * Cout * 8 * sizeof(float), 256);
  size_t c = align_up(conv3_wgrad_mfma_ws_bytes(B, Cin, Cout, Di, Hi, Wi), 256);
  return convT_bias_region(B, Cout, Di, Hi, Wi) + convT_pack_region(Cin, Cout, DGTTA_F32) + (a > c ? a : c);
}

// ... with room for the fp32 weight gradient as six 16-bit launches on three-term bf16 splits (conv_wgrad.hip)
size_t convT_wgrad_split_extra_bytes(int B, int Cin, int Cout, int Di, int Hi, int Wi);
extern "C" size_t dgtta_convT3d_bwd_split_ws_bytes(int B, int Cin, int Cout, int Di, int Hi, int Wi) {
  if (B <= 0 || Cin <= 0 || Cout <= 0 || Di <= 0 || Hi <= 0 || Wi <= 0) return 0;
  const int64_t nvox = (int64_t)B * Di * Hi * Wi;
  size_t a = align_up((size_t)wgrad_splits(nvox) * Cin * Cout * 8 * sizeof(float), 256);
  size_t c = align_up(conv3_wgrad_mfma_ws_bytes(B, Cin, Cout, Di, Hi, Wi), 256) + convT_wgrad_split_extra_bytes(B, Cin, Cout, Di, Hi, Wi);
  return convT_bias_region(B, Cout, Di, Hi, Wi) + convT_pack_region(Cin, Cout, DGTTA_F32) + (a > c ? a : c);
}

extern "C" int dgtta_convT3d_k2s2_bwd(const void *x, int ldx, const void *dout, int lddo, const float *w_t, void *dx,
                                      int lddx, float *dw_t, float *db, void *ws, size_t ws_bytes, int B, int Cin,
                                      int Cout, int Di, int Hi, int Wi, int accumulate, int dtype, int impl,
                                      void *stream) {
  DG_REQUIRE(x && dout && w_t && ws, DGTTA_ERR_BADARG, "convT3d_k2s2_bwd: null pointer");
  DG_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && Di > 0 && Hi > 0 && Wi > 0 && ldx >= Cin && lddo >= Cout,
             DGTTA_ERR_BADARG, "convT3d_k2s2_bwd: bad dims");
  DG_REQUIRE(ws_bytes >= dgtta_convT3d_bwd_ws_bytes(B, Cin, Cout, Di, Hi, Wi), DGTTA_ERR_WORKSPACE,
             "convT3d_k2s2_bwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int64_t nvox = (int64_t)B * Di * Hi * Wi;
  void *ws_bias = ws;
  void *ws_pack = (char *)ws + convT_bias_region(B, Cout, Di, Hi, Wi);
  void *ws_main = (char *)ws_pack + convT_pack_region(Cin, Cout, DGTTA_F32);
  const size_t main_bytes = ws_bytes - ((char *)ws_main - (char *)ws);
  int bias_units = 0;
  if (dx) {
    DG_REQUIRE(lddx >= Cin, DGTTA_ERR_BADARG, "convT3d_k2s2_bwd: lddx < Cin");
    int rc = DGTTA_ERR_UNSUPPORTED;
    if (impl != 1) rc = convT_dgrad_mfma(dout, lddo, w_t, dx, lddx, ws_pack, B, Cin, Cout, Di, Hi, Wi, dtype, st);
    if (rc == DGTTA_ERR_UNSUPPORTED) {
      DG_REQUIRE(impl != 2, DGTTA_ERR_UNSUPPORTED, "convT3d_k2s2_bwd: dgrad shape not covered by the MFMA kernel");
      const int64_t total = nvox * Cin;
      DISPATCH_T(dtype, hipLaunchKernelGGL((convT_dgrad_ref_kernel<T>), dim3(gs_blocks(total, 1 << 20)), dim3(256), 0, st,
                                           (const T *)dout, lddo, w_t, (T *)dx, lddx, Cin, Cout, Di, Hi, Wi, total));
      DG_CHECK_LAUNCH("convT_dgrad_ref_kernel");
    } else if (rc != DGTTA_OK) {
      return rc;
    }
  }
  if (dw_t) {
    int rc = DGTTA_ERR_UNSUPPORTED;
    if (impl != 1)      // (the one-pass kernel also leaves the bias gradient's partial sums in the bias region when asked)
      rc = convT_wgrad_mfma(x, ldx, dout, lddo, dw_t, ws_main, main_bytes, B, Cin, Cout, Di, Hi, Wi, accumulate, dtype, st,
                            db ? (float *)ws_bias : nullptr, convT_bias_region(B, Cout, Di, Hi, Wi), &bias_units);
    if (rc == DGTTA_ERR_UNSUPPORTED) {
      DG_REQUIRE(impl != 2, DGTTA_ERR_UNSUPPORTED, "convT3d_k2s2_bwd: wgrad shape not covered by the MFMA kernel");
      const int nsplit = wgrad_splits(nvox);
      float *part = (float *)ws_main;
      DISPATCH_T(dtype, hipLaunchKernelGGL((convT_wgrad_ref_kernel<T>), dim3(cdiv(Cin * Cout, 256), 8, nsplit), dim3(256),
                                           0, st, (const T *)x, ldx, (const T *)dout, lddo, part, Cin, Cout, B, Di, Hi, Wi));
      DG_CHECK_LAUNCH("convT_wgrad_ref_kernel");
      const int64_t n = (int64_t)Cin * Cout * 8;
      hipLaunchKernelGGL(reduce_splits_kernel, dim3(gs_blocks(n)), dim3(256), 0, st, part, dw_t, n, nsplit, accumulate);
      DG_CHECK_LAUNCH("reduce_splits_kernel");
    } else if (rc != DGTTA_OK) {
      return rc;
    }
  }
  if (db && bias_units > 0) {
    DG_REQUIRE(convT_bias_finalize((const float *)ws_bias, bias_units, Cout, db, accumulate, st) == DGTTA_OK, DGTTA_ERR_LAUNCH,
               "convT3d_k2s2_bwd: bias finalize launch failed");
    return DGTTA_OK;
  }
  if (db) return bias_grad(dout, lddo, db, ws_bias, B, Cout, (int64_t)Di * Hi * Wi * 8, accumulate, dtype, st);
  return DGTTA_OK;
}

extern "C" int dgtta_seghead_fwd(const void *x, int ldx, const float *w, const float *bias, const int *sel, int nsel,
                                 float *out, int out_ndhwc, int ldo, int B, int Cin, int64_t V, int dtype,
                                 void *stream) {
  DG_REQUIRE(x && w && bias && out, DGTTA_ERR_BADARG, "seghead_fwd: null pointer");
  DG_REQUIRE(B > 0 && Cin > 0 && nsel > 0 && V > 0 && ldx >= Cin && (!out_ndhwc || ldo >= nsel), DGTTA_ERR_BADARG,
             "seghead_fwd: bad dims");
  if (Cin == 32 && nsel <= 128 && ldx % 8 == 0 && ((uintptr_t)x & 15) == 0) {
    const int64_t rows = (int64_t)B * V;
    const int blocks = (int)(cdiv64(rows, 256) < 8192 ? cdiv64(rows, 256) : 8192);
    if (out_ndhwc && ldx == 32 && ldo == nsel && nsel <= 16 && ((uintptr_t)out & 15) == 0) {
      DISPATCH_T(dtype, hipLaunchKernelGGL((head_fwd_lds_kernel<T>), dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                                           (const T *)x, w, bias, sel, nsel, out, rows));
      DG_CHECK_LAUNCH("head_fwd_lds_kernel");
      return DGTTA_OK;
    }
    if (out_ndhwc)
      DISPATCH_T(dtype, hipLaunchKernelGGL((head_fwd_fast_kernel<T, 32, true>), dim3(blocks), dim3(256), 0,
                                           (hipStream_t)stream, (const T *)x, ldx, w, bias, sel, nsel, out, ldo, V, rows));
    else
      DISPATCH_T(dtype, hipLaunchKernelGGL((head_fwd_fast_kernel<T, 32, false>), dim3(blocks), dim3(256), 0,
                                           (hipStream_t)stream, (const T *)x, ldx, w, bias, sel, nsel, out, ldo, V, rows));
    DG_CHECK_LAUNCH("head_fwd_fast_kernel");
    return DGTTA_OK;
  }
  const int64_t total = (int64_t)B * V * nsel;
  if (out_ndhwc)
    DISPATCH_T(dtype, hipLaunchKernelGGL((head_fwd_kernel<T, true>), dim3(gs_blocks(total, 1 << 20)), dim3(256), 0,
                                         (hipStream_t)stream, (const T *)x, ldx, w, bias, sel, nsel, out, ldo, Cin, V,
                                         total));
  else
    DISPATCH_T(dtype, hipLaunchKernelGGL((head_fwd_kernel<T, false>), dim3(gs_blocks(total, 1 << 20)), dim3(256), 0,
                                         (hipStream_t)stream, (const T *)x, ldx, w, bias, sel, nsel, out, ldo, Cin, V,
                                         total));
  DG_CHECK_LAUNCH("head_fwd_kernel");
  return DGTTA_OK;
}

static int head_splits(int64_t rows) {
  int64_t s = cdiv64(rows, 2048);
  return (int)(s < 256 ? (s > 0 ? s : 1) : 256);
}

size_t head_wgrad_mfma_ws_bytes(int Cin, int nsel, int64_t rows);
int head_wgrad_mfma(const void *x, int ldx, const float *dout, int lddo, float *dw_sel, void *ws, size_t ws_bytes, int Cin,
                    int nsel, int64_t rows, int accumulate, int dtype, hipStream_t st, bool have_d16);

// workspace layout: [bias partials][main: split partials (VALU) | bf16 copy + slabs (MFMA)]
static size_t head_bias_region(int B, int nsel, int64_t V) {
  return align_up((size_t)B * reduce_blocks(V, B) * nsel * 2 * sizeof(double), 256);
}

extern "C" size_t dgtta_seghead_bwd_ws_bytes(int B, int Cin, int nsel, int64_t V) {
  if (B <= 0 || Cin <= 0 || nsel <= 0 || V <= 0) return 0;      // a size query of an empty problem (the launchers reject it with DGTTA_ERR_BADARG)
  size_t a = align_up((size_t)head_splits((int64_t)B * V) * nsel * Cin * sizeof(float), 256);
  size_t c = align_up(head_wgrad_mfma_ws_bytes(Cin, nsel, (int64_t)B * V), 256);
  return head_bias_region(B, nsel, V) + (a > c ? a : c);
}

extern "C" int dgtta_seghead_bwd(const void *x, int ldx, const float *dout, int lddo, const float *w, const int *sel,
                                 int nsel, void *dx, int lddx, float *dw_sel, float *db_sel, void *ws, size_t ws_bytes,
                                 int B, int Cin, int64_t V, int accumulate, int dtype, void *stream) {
  DG_REQUIRE(x && dout && w && ws, DGTTA_ERR_BADARG, "seghead_bwd: null pointer");
  DG_REQUIRE(B > 0 && Cin > 0 && nsel > 0 && V > 0 && ldx >= Cin && lddo >= nsel, DGTTA_ERR_BADARG, "seghead_bwd: bad dims");
  DG_REQUIRE(ws_bytes >= dgtta_seghead_bwd_ws_bytes(B, Cin, nsel, V), DGTTA_ERR_WORKSPACE, "seghead_bwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int64_t rows = (int64_t)B * V;
  void *ws_bias = ws;
  void *ws_main = (char *)ws + head_bias_region(B, nsel, V);
  const size_t main_bytes = ws_bytes - head_bias_region(B, nsel, V);
  unsigned short *d16 = nullptr;      // 16-bit copy of dout written by the data-gradient kernel for the weight gradient
  if (dx) {
    DG_REQUIRE(lddx >= Cin, DGTTA_ERR_BADARG, "seghead_bwd: lddx < Cin");
    if (Cin == 32 && nsel <= 32) {
      const int blocks = (int)(cdiv64(rows, 256) < 8192 ? cdiv64(rows, 256) : 8192);
      if (lddx == 32 && lddo == nsel && nsel <= 16 && ((uintptr_t)dx & 15) == 0) {
        if (dw_sel && dtype != DGTTA_F32 && head_wgrad_mfma_ws_bytes(Cin, nsel, rows) > 0 &&
            main_bytes >= head_wgrad_mfma_ws_bytes(Cin, nsel, rows)) {
          d16 = (unsigned short *)ws_main;       // the first region of head_wgrad_mfma's workspace
        }
        DISPATCH_T(dtype, hipLaunchKernelGGL((head_dgrad_lds_kernel<T>), dim3(blocks), dim3(256), 0, st, dout, w, sel, nsel,
                                             (T *)dx, rows, d16));
        DG_CHECK_LAUNCH("head_dgrad_lds_kernel");
      } else {
        DISPATCH_T(dtype, hipLaunchKernelGGL((head_dgrad_fast_kernel<T, 32>), dim3(blocks), dim3(256), 0, st, dout, lddo, w,
                                             sel, nsel, (T *)dx, lddx, rows));
        DG_CHECK_LAUNCH("head_dgrad_fast_kernel");
      }
    } else if (Cin == 32 && nsel <= HWD_MAXK) {      // the wide head: 64-row tiles through LDS
      const size_t lds = ((size_t)HWD_MAXK * 32 + (size_t)HWD_ROWS * (nsel | 1)) * sizeof(float);
      const int64_t nt = cdiv64(rows, HWD_ROWS);
      DISPATCH_T(dtype, {
        static DynLdsOnce once;
        DG_REQUIRE(ensure_dyn_lds(once, reinterpret_cast<const void *>(head_dgrad_wide_kernel<T>),
                                  (HWD_MAXK * 32 + HWD_ROWS * (HWD_MAXK | 1)) * (int)sizeof(float)) == hipSuccess,
                   DGTTA_ERR_LAUNCH, "seghead_bwd: cannot raise the dynamic LDS limit");
        hipLaunchKernelGGL((head_dgrad_wide_kernel<T>), dim3((unsigned)(nt < 4096 ? nt : 4096)), dim3(256), lds, st, dout, lddo, w, sel,
                           nsel, (T *)dx, lddx, rows);
      });
      DG_CHECK_LAUNCH("head_dgrad_wide_kernel");
    } else {
      const int64_t total = rows * Cin;
      DISPATCH_T(dtype, hipLaunchKernelGGL((head_dgrad_kernel<T>), dim3(gs_blocks(total, 1 << 20)), dim3(256), 0, st, dout,
                                           lddo, w, sel, nsel, (T *)dx, lddx, Cin, total));
      DG_CHECK_LAUNCH("head_dgrad_kernel");
    }
  }
  if (dw_sel) {
    int rc = head_wgrad_mfma(x, ldx, dout, lddo, dw_sel, ws_main, main_bytes, Cin, nsel, rows, accumulate, dtype, st,
                             d16 != nullptr);
    if (rc == DGTTA_ERR_UNSUPPORTED) {
      const int ns = head_splits(rows);
      float *part = (float *)ws_main;
      if (Cin == 32 && nsel > 32 && nsel <= HWD_MAXK) {
        const size_t lds = ((size_t)HWD_ROWS * 33 + (size_t)HWD_ROWS * (nsel | 1) + HWD_MAXK) * sizeof(float);
        DISPATCH_T(dtype, {
          static DynLdsOnce once;
          DG_REQUIRE(ensure_dyn_lds(once, reinterpret_cast<const void *>(head_wgrad_wide_kernel<T>),
                                    (HWD_ROWS * 33 + HWD_ROWS * (HWD_MAXK | 1) + HWD_MAXK) * (int)sizeof(float)) == hipSuccess,
                     DGTTA_ERR_LAUNCH, "seghead_bwd: cannot raise the dynamic LDS limit");
          hipLaunchKernelGGL((head_wgrad_wide_kernel<T>), dim3(ns), dim3(256), lds, st, (const T *)x, ldx, dout, lddo, part, nsel,
                             rows);
        });
        DG_CHECK_LAUNCH("head_wgrad_wide_kernel");
      } else {
        DISPATCH_T(dtype, hipLaunchKernelGGL((head_wgrad_kernel<T>), dim3(cdiv(Cin * nsel, 256), ns), dim3(256), 0, st,
                                             (const T *)x, ldx, dout, lddo, part, Cin, nsel, rows));
        DG_CHECK_LAUNCH("head_wgrad_kernel");
      }
      const int64_t n = (int64_t)nsel * Cin;
      hipLaunchKernelGGL(reduce_splits_kernel, dim3(gs_blocks(n)), dim3(256), 0, st, part, dw_sel, n, ns, accumulate);
      DG_CHECK_LAUNCH("reduce_splits_kernel");
    } else if (rc != DGTTA_OK) {
      return rc;
    }
  }
  if (db_sel) return bias_grad(dout, lddo, db_sel, ws_bias, B, nsel, V, accumulate, DGTTA_F32, st);
  return DGTTA_OK;
}

extern "C" int dgtta_ncdhw_to_ndhwc(const float *src, void *dst, int B, int C, int64_t V, int ldc, int dtype,
                                    void *stream) {
  DG_REQUIRE(src && dst && B > 0 && C > 0 && V > 0 && ldc >= C, DGTTA_ERR_BADARG, "ncdhw_to_ndhwc: bad args");
  const int64_t total = (int64_t)B * V * ldc;
  DISPATCH_T(dtype, hipLaunchKernelGGL((ncdhw_to_ndhwc_kernel<T>), dim3(gs_blocks(total)), dim3(256), 0,
                                       (hipStream_t)stream, src, (T *)dst, C, V, ldc, total));
  DG_CHECK_LAUNCH("ncdhw_to_ndhwc_kernel");
  return DGTTA_OK;
}

extern "C" int dgtta_ndhwc_to_ncdhw(const void *src, float *dst, int B, int C, int64_t V, int ldc, int dtype,
                                    void *stream) {
  DG_REQUIRE(src && dst && B > 0 && C > 0 && V > 0 && ldc >= C, DGTTA_ERR_BADARG, "ndhwc_to_ncdhw: bad args");
  const int64_t total = (int64_t)B * V * C;
  DISPATCH_T(dtype, hipLaunchKernelGGL((ndhwc_to_ncdhw_kernel<T>), dim3(gs_blocks(total)), dim3(256), 0,
                                       (hipStream_t)stream, (const T *)src, dst, C, V, ldc, total));
  DG_CHECK_LAUNCH("ndhwc_to_ncdhw_kernel");
  return DGTTA_OK;
}

extern "C" int dgtta_argmax_rows(const void *logits, int acc_dtype, int C, int64_t rows, int64_t *argmax_out, void *stream) {
  DG_REQUIRE(logits && argmax_out && rows > 0, DGTTA_ERR_BADARG, "argmax_rows: bad args");
  DG_REQUIRE(C > 0 && C <= 65536, DGTTA_ERR_UNSUPPORTED, "argmax_rows: C %d", C);
  DG_REQUIRE(acc_dtype == DGTTA_F32 || acc_dtype == DGTTA_F16, DGTTA_ERR_UNSUPPORTED, "argmax_rows: rows are fp32 or fp16");
  DG_REQUIRE(((uintptr_t)logits & 3) == 0, DGTTA_ERR_BADARG, "argmax_rows: rows must start on a 4-byte boundary");
  if (C > AR_MAXC) {      // wider than the LDS tile: one wave per row
    const dim3 gridw((unsigned)(cdiv64(rows, 4) < 16384 ? cdiv64(rows, 4) : 16384));
    if (acc_dtype == DGTTA_F32)
      hipLaunchKernelGGL(argmax_rows_wide_kernel<float>, gridw, dim3(256), 0, (hipStream_t)stream, (const float *)logits, C, argmax_out, rows);
    else
      hipLaunchKernelGGL(argmax_rows_wide_kernel<f16_t>, gridw, dim3(256), 0, (hipStream_t)stream, (const f16_t *)logits, C, argmax_out, rows);
    DG_CHECK_LAUNCH("argmax_rows_wide_kernel");
    return DGTTA_OK;
  }
  const int64_t ntile = cdiv64(rows, 64);
  const dim3 grid((unsigned)(ntile < 4096 ? ntile : 4096));
  const size_t lds = (size_t)64 * C * sizeof(float);
  if (acc_dtype == DGTTA_F32)
    hipLaunchKernelGGL(argmax_rows_kernel<float>, grid, dim3(256), lds, (hipStream_t)stream, (const float *)logits, C, argmax_out,
                       rows);
  else
    hipLaunchKernelGGL(argmax_rows_kernel<f16_t>, grid, dim3(256), lds, (hipStream_t)stream, (const f16_t *)logits, C, argmax_out,
                       rows);
  DG_CHECK_LAUNCH("argmax_rows_kernel");
  return DGTTA_OK;
}

extern "C" int dgtta_argmax_dice(const float *logits, int ldc, int C, const int64_t *labels, int64_t *argmax_out,
                                 int64_t *counts, int B, int64_t V, void *stream) {
  DG_REQUIRE((logits || argmax_out) && C > 0 && C <= 1024 && (!logits || ldc >= C) && B > 0 && V > 0, DGTTA_ERR_BADARG,
             "argmax_dice: bad args");
  DG_REQUIRE(!labels || counts, DGTTA_ERR_BADARG, "argmax_dice: labels without counts");
  const int64_t total = (int64_t)B * V;
  // back-to-back rows, no Dice counts asked for (the sliding-window label map): the streaming kernel
  if (logits && !labels && argmax_out && ldc == C && C <= AR_MAXC && total >= 4096)
    return dgtta_argmax_rows(logits, DGTTA_F32, C, total, argmax_out, stream);
  hipLaunchKernelGGL(argmax_dice_kernel, dim3(gs_blocks(total, 2048)), dim3(256), 3 * C * sizeof(unsigned int),
                     (hipStream_t)stream, logits, ldc, C, labels, argmax_out, (unsigned long long *)counts, total);
  DG_CHECK_LAUNCH("argmax_dice_kernel");
  return DGTTA_OK;
}
