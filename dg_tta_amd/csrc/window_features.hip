// Sliding-window inference in FEATURE space (round 5; BASELINE config 3).
//
// What the reference does (nnunetv2==2.2.1 predict_sliding_window_return_logits, reached from
// dg_tta/tta/nnunet_utils.py:116-125, 208-230; ensemble loop dg_tta/tta/tta.py:396-413): every window's logits over ALL
// pretrain classes are multiplied by the Gaussian importance map and added into a [classes, X, Y, Z] volume; the volume is
// divided by the summed weights, the members' volumes are averaged and the label map is the argmax.
//
// The segmentation head is a 1x1x1 convolution - LINEAR - and it is the last thing in the network:
//     sum_w g_w(v) (W z_w(v) + b) = W (sum_w g_w(v) z_w(v)) + b sum_w g_w(v).
// So the volume-sized accumulator can hold the 32 Gaussian-weighted FEATURE channels of the last decoder block instead of
// the 105 logits, and the head runs ONCE per voxel at the end, fused with the argmax:
//   * read-modify-write per window voxel: 64 B of features + 2 x 128 B instead of 64 B + 2 x 420 B (2.8x less HBM traffic in
//     the pass that was 20 % of a member's time: 343 windows x 1.9 GB at 512^3);
//   * accumulator of a 512^3 volume: 16 GiB per member instead of 52.5 GiB - and a member needs its own (the members' heads
//     differ), so an ensemble of three is 48 GiB: still below ONE logits volume;
//   * the 105-class volume never exists: the label map is argmax_c sum_m (W_m F_m(v) + b_m n(v)) straight from the features
//     (no division: n(v) > 0 is shared by all classes and members).
// Floating point: the sums are re-associated (fp32 throughout; products g * z are exact up to one rounding, the head runs on
// fp32 features with the fp32 weights instead of on 16-bit features window by window), so labels can differ from the
// logits-space accumulator where the top-2 margin is at rounding level - tests/test_inference.py compares both forms with the
// CPU restatement outside such ties.  nnU-Net itself is absent from /root/reference: this stage is "parity unpinned" either way.
#include "common.h"

namespace {

constexpr int WF_CIN = 32;

// 4 channels of a voxel as floats
template <typename T>
__device__ __forceinline__ void wf_load4(const T *p, float (&f)[4]);
template <>
__device__ __forceinline__ void wf_load4<float>(const float *p, float (&f)[4]) {
  const float4 v = *reinterpret_cast<const float4 *>(p);
  f[0] = v.x, f[1] = v.y, f[2] = v.z, f[3] = v.w;
}
template <>
__device__ __forceinline__ void wf_load4<bf16_t>(const bf16_t *p, float (&f)[4]) {
  const uint2 v = *reinterpret_cast<const uint2 *>(p);
  f[0] = __uint_as_float(v.x << 16), f[1] = __uint_as_float(v.x & 0xffff0000u);
  f[2] = __uint_as_float(v.y << 16), f[3] = __uint_as_float(v.y & 0xffff0000u);
}
template <>
__device__ __forceinline__ void wf_load4<f16_t>(const f16_t *p, float (&f)[4]) {
  const uint2 v = *reinterpret_cast<const uint2 *>(p);
  f[0] = f16_to_f32((unsigned short)(v.x & 0xffffu)), f[1] = f16_to_f32((unsigned short)(v.x >> 16));
  f[2] = f16_to_f32((unsigned short)(v.y & 0xffffu)), f[3] = f16_to_f32((unsigned short)(v.y >> 16));
}

// facc[(x0 + pd, y0 + ph, z0 + pw)][c] += gauss[p] * z[p][c], nsum[..] += gauss[p]; 8 threads per voxel, 4 channels each: every
// load and store instruction of a wave covers one contiguous run (z: 512 B, facc: 1 KiB; a window row is contiguous in the
// volume along its last axis).  A launch touches every accumulator element once - overlapping windows are separate launches
// in stream order, as in the logits-space form.
template <typename T>
__global__ __launch_bounds__(256) void feature_accumulate_kernel(const T *__restrict__ z, const float *__restrict__ gauss,
                                                                 float *__restrict__ facc, float *__restrict__ nsum, int PH, int PW,
                                                                 int Y, int Z, int x0, int y0, int z0, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int q = (int)(i & 7);
  const int64_t p = i >> 3;
  const int pw = (int)(p % PW), ph = (int)((p / PW) % PH);
  const int64_t pd = p / ((int64_t)PW * PH);
  const int64_t v = ((pd + x0) * Y + (ph + y0)) * Z + (pw + z0);
  const float g = gauss[p];
  float f[4];
  wf_load4<T>(z + p * WF_CIN + q * 4, f);
  float4 *dst = reinterpret_cast<float4 *>(facc + v * WF_CIN + q * 4);
  float4 a = *dst;
  a.x += g * f[0], a.y += g * f[1], a.z += g * f[2], a.w += g * f[3];
  *dst = a;
  if (q == 0 && nsum) nsum[v] += g;
}

// value -> storage type T and back (what the network's apply pass writes and the head would read)
template <typename T>
__device__ __forceinline__ float wf_round(float v);
template <>
__device__ __forceinline__ float wf_round<float>(float v) { return v; }
template <>
__device__ __forceinline__ float wf_round<bf16_t>(float v) { return bf16_to_f32(f32_to_bf16(v)); }
template <>
__device__ __forceinline__ float wf_round<f16_t>(float v) { return f16_to_f32(f32_to_f16(v)); }

// The same with the last block's InstanceNorm + LeakyReLU apply folded in: z = round_T(lrelu(y * alpha + beta')) is formed from the
// raw conv output y in registers (alpha = rstd * gamma, beta' = beta - mean * alpha: the arithmetic of in_apply_vec_kernel, so the
// SAME z values) and never written - the apply pass (read y, write z) and the read of z disappear for the layer in front of the
// head: 64 + 256 B per voxel instead of 128 + 320.  No backward exists in inference, so nothing else needs that z.
template <typename T>
__global__ __launch_bounds__(256) void feature_accumulate_norm_kernel(const T *__restrict__ y, const float *__restrict__ mean_rstd,
                                                                      const float *__restrict__ gamma, const float *__restrict__ beta,
                                                                      float slope, const float *__restrict__ gauss,
                                                                      float *__restrict__ facc, float *__restrict__ nsum, int PH, int PW,
                                                                      int Y, int Z, int x0, int y0, int z0, int64_t total) {
  __shared__ float sal[WF_CIN], sbe[WF_CIN];      // the window's per-channel constants, once per workgroup
  if (threadIdx.x < WF_CIN) {
    const int c = threadIdx.x;
    const float al = mean_rstd[c * 2 + 1] * gamma[c];
    sal[c] = al;
    sbe[c] = beta[c] - mean_rstd[c * 2] * al;
  }
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int q = (int)(i & 7);
  const int64_t p = i >> 3;
  const int pw = (int)(p % PW), ph = (int)((p / PW) % PH);
  const int64_t pd = p / ((int64_t)PW * PH);
  const int64_t v = ((pd + x0) * Y + (ph + y0)) * Z + (pw + z0);
  const float g = gauss[p];
  float f[4];
  wf_load4<T>(y + p * WF_CIN + q * 4, f);
  float4 *dst = reinterpret_cast<float4 *>(facc + v * WF_CIN + q * 4);
  float4 a = *dst;
  float zz[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) zz[e] = wf_round<T>(lrelu(f[e] * sal[q * 4 + e] + sbe[q * 4 + e], slope));
  a.x += g * zz[0], a.y += g * zz[1], a.z += g * zz[2], a.w += g * zz[3];
  *dst = a;
  if (q == 0 && nsum) nsum[v] += g;
}

// Several windows of one network pass that overlap along the last axis (consecutive origins of a sliding-window row), one SEGMENT of
// that axis per launch: every voxel of the segment receives the contributions of all NS windows that cover it, added in window order
// in registers - (acc + t_0) + t_1: the bits the one-window launches produce - and the accumulator is read and written ONCE instead
// of NS times.  With 50 % overlap a row of 7 windows is 8 half-window segments: 57 % of the read-modify-write traffic.
// NORM: the sources are raw conv outputs and the InstanceNorm + LeakyReLU apply runs here (feature_accumulate_norm_kernel).
struct WfSources {
  const void *src[4];            // window k: [PD][PH][PW][32]
  const float *mean_rstd[4];     // NORM: the window's statistics [32][2]
  int zoff[4];                   // the segment's first voxel along the last axis, in window k's coordinates
};

template <typename T, int NS, bool NORM>
__global__ __launch_bounds__(256) void feature_accumulate_multi_kernel(WfSources ws, const float *__restrict__ gamma,
                                                                       const float *__restrict__ beta, float slope,
                                                                       const float *__restrict__ gauss, float *__restrict__ facc,
                                                                       float *__restrict__ nsum, int PH, int PW, int SL, int Y, int Z, int x0,
                                                                       int y0, int z0, int64_t total) {
  __shared__ float sal[NS][WF_CIN], sbe[NS][WF_CIN];
  if (NORM) {
    if (threadIdx.x < NS * WF_CIN) {
      const int k = threadIdx.x / WF_CIN, c = threadIdx.x % WF_CIN;
      const float al = ws.mean_rstd[k][c * 2 + 1] * gamma[c];
      sal[k][c] = al;
      sbe[k][c] = beta[c] - ws.mean_rstd[k][c * 2] * al;
    }
    __syncthreads();
  }
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int q = (int)(i & 7);
  const int64_t p = i >> 3;                  // voxel of the segment [PD][PH][SL]
  const int zz = (int)(p % SL), ph = (int)((p / SL) % PH);
  const int64_t pd = p / ((int64_t)SL * PH);
  const int64_t v = ((pd + x0) * Y + (ph + y0)) * Z + (zz + z0);
  float4 *dst = reinterpret_cast<float4 *>(facc + v * WF_CIN + q * 4);
  float4 a = *dst;
  float n = (q == 0 && nsum) ? nsum[v] : 0.f;
  float f[NS][4], g[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) {             // all loads first
    const int64_t pk = (pd * PH + ph) * PW + ws.zoff[k] + zz;
    g[k] = gauss[pk];
    wf_load4<T>(reinterpret_cast<const T *>(ws.src[k]) + pk * WF_CIN + q * 4, f[k]);
  }
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    float zv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) zv[e] = NORM ? wf_round<T>(lrelu(f[k][e] * sal[k][q * 4 + e] + sbe[k][q * 4 + e], slope)) : f[k][e];
    a.x += g[k] * zv[0], a.y += g[k] * zv[1], a.z += g[k] * zv[2], a.w += g[k] * zv[3];
    n += g[k];
  }
  *dst = a;
  if (q == 0 && nsum) nsum[v] = n;
}

// Label map from the members' feature accumulators: argmax_c sum_m (W_m[c] . F_m(v)) + n(v) bsum[c], bsum = sum_m b_m (formed
// by the caller).  One thread per voxel; a member's 32 features sit in registers, the weights are wave-uniform (scalar
// loads).  With several members the classes' partial sums wait in LDS ([C][256] floats, own column per thread: no bank
// conflicts, no barrier); with one member they are compared as they are produced.  First maximum wins, as argmax does.
__global__ __launch_bounds__(256) void feature_head_argmax_kernel(const float *__restrict__ facc, int64_t member_stride,
                                                                  const float *__restrict__ nsum, const float *__restrict__ w,
                                                                  const float *__restrict__ bsum, int M, int C, int64_t V,
                                                                  int64_t *__restrict__ amax) {
  extern __shared__ float part[];      // [C][256] when M > 1
  const int tid = threadIdx.x;
  for (int64_t v0 = (int64_t)blockIdx.x * 256; v0 < V; v0 += (int64_t)gridDim.x * 256) {
    const int64_t v = v0 + tid < V ? v0 + tid : V - 1;      // (the tail repeats the last voxel: no divergent exit above LDS use)
    const float n = nsum[v];
    float best = -INFINITY;
    int bi = 0;
    for (int m = 0; m < M; ++m) {
      float f[WF_CIN];
      const float4 *src = reinterpret_cast<const float4 *>(facc + m * member_stride + v * WF_CIN);
#pragma unroll
      for (int k = 0; k < WF_CIN / 4; ++k) {
        const float4 t = src[k];
        f[4 * k] = t.x, f[4 * k + 1] = t.y, f[4 * k + 2] = t.z, f[4 * k + 3] = t.w;
      }
      const float *wm = w + (int64_t)m * C * WF_CIN;
      for (int c = 0; c < C; ++c) {
        // four independent FMA chains over the channels (a single chain of 32 waits for itself)
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
        for (int k = 0; k < WF_CIN; k += 4) {
          s0 = fmaf(wm[c * WF_CIN + k], f[k], s0);
          s1 = fmaf(wm[c * WF_CIN + k + 1], f[k + 1], s1);
          s2 = fmaf(wm[c * WF_CIN + k + 2], f[k + 2], s2);
          s3 = fmaf(wm[c * WF_CIN + k + 3], f[k + 3], s3);
        }
        float s = (s0 + s1) + (s2 + s3);
        if (m == 0) s += n * bsum[c];
        else s += part[c * 256 + tid];
        if (m + 1 < M) {
          part[c * 256 + tid] = s;
        } else if (s > best) {
          best = s;
          bi = c;
        }
      }
    }
    if (v0 + tid < V) amax[v] = bi;
  }
}

// The same on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: M = 32 classes, N = 32 voxels, K = the 32 channels of every member in
// turn): a wave owns 32 voxels, the weights of all members sit in LDS ([member][class, padded to a multiple of 32][36 floats]: the
// 144-byte pitch keeps the 16-byte reads of 8 consecutive class rows on distinct bank groups), a lane's 16 bytes of features /
// weights feed four instructions (lane half h holds channels 8 s + 4 h .. + 3 of k-step s, as conv_common.h's fp32 mfma_step).
// Accumulators start at n(v) bsum[c]; the members add up in the accumulators (no LDS partials).  C/D map: a lane holds voxel
// lane & 31 and classes 32 cb + (q & 3) + 8 (q >> 2) + 4 (lane >> 5) - ascending in (cb, q), so a strict comparison in that order
// keeps the first maximum; the two lane halves of a voxel are merged with the lower class winning a tie.
// Measured at 512^3 x 105 classes (profiles/tools/headargmax_bench.py): one member 14.3 ms (vector-ALU kernel 15.5: the per-tile
// initialisation and argmax scan cost as much as one member's products), the plan's ensemble of three 29.0 ms against 179 ms - the
// vector-ALU kernel keeps several members' partial sums in LDS and is bound by that traffic; here a further member is 7.3 ms of MFMAs.
typedef float wf_f32x16_t __attribute__((ext_vector_type(16)));
constexpr int WF_WPITCH = 36;

template <int NCB>      // class blocks of 32 (C <= 32 NCB)
__global__ __launch_bounds__(256) void feature_head_argmax_mfma_kernel(const float *__restrict__ facc, int64_t member_stride,
                                                                       const float *__restrict__ nsum, const float *__restrict__ w,
                                                                       const float *__restrict__ bsum, int M, int C, int64_t V,
                                                                       int64_t *__restrict__ amax) {
  extern __shared__ __attribute__((aligned(16))) float sw[];      // [M][32 NCB][WF_WPITCH], then bsum [32 NCB]
  float *sb = sw + (size_t)M * 32 * NCB * WF_WPITCH;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < M * 32 * NCB * WF_CIN; i += 256) {
    const int k = i % WF_CIN, c = (i / WF_CIN) % (32 * NCB), m = i / (WF_CIN * 32 * NCB);
    sw[(m * 32 * NCB + c) * WF_WPITCH + k] = c < C ? w[((int64_t)m * C + c) * WF_CIN + k] : 0.f;
  }
  for (int c = tid; c < 32 * NCB; c += 256) sb[c] = c < C ? bsum[c] : 0.f;
  __syncthreads();
  const int j = lane & 31, h = lane >> 5;
  const int64_t ntile = (V + 31) / 32;
  const int64_t tstep = (int64_t)gridDim.x * 4;
  // the features of (tile, member) pair p + 1 are requested before the 16 NCB MFMAs of pair p issue: at 2 waves per SIMD (161
  // registers with four class blocks) a load - wait - multiply sequence per pair left the matrix pipe idle for the memory latency
  auto voxel_of = [&](int64_t t) { return t * 32 + j < V ? t * 32 + j : V - 1; };
  auto load_fb = [&](int64_t t, int m, float4(&fb)[4]) {
    const float4 *fv = reinterpret_cast<const float4 *>(facc + m * member_stride + voxel_of(t) * WF_CIN) + h;
#pragma unroll
    for (int s = 0; s < 4; ++s) fb[s] = fv[2 * s];      // channels 8 s + 4 h .. + 3
  };
  float4 fb[4], fn[4];
  int64_t t = (int64_t)blockIdx.x * 4 + wave;
  if (t < ntile) load_fb(t, 0, fb);
  for (; t < ntile; t += tstep) {
    const int64_t v = voxel_of(t);
    const float n = nsum[v];
    wf_f32x16_t acc[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {      // a lane's classes come in runs of four
        const float4 b4 = *reinterpret_cast<const float4 *>(sb + cb * 32 + 8 * q4 + 4 * h);
        acc[cb][4 * q4 + 0] = n * b4.x, acc[cb][4 * q4 + 1] = n * b4.y, acc[cb][4 * q4 + 2] = n * b4.z, acc[cb][4 * q4 + 3] = n * b4.w;
      }
    for (int m = 0; m < M; ++m) {
      if (m + 1 < M) load_fb(t, m + 1, fn);
      else if (t + tstep < ntile) load_fb(t + tstep, 0, fn);
      const float *wm = sw + ((size_t)m * 32 * NCB + j) * WF_WPITCH + 4 * h;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        float4 wa[NCB];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) wa[cb] = *reinterpret_cast<const float4 *>(wm + (size_t)cb * 32 * WF_WPITCH + 8 * s);
        // (consecutive instructions on different accumulators: a chain on one accumulator waits for itself)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[cb].x, fb[s].x, acc[cb], 0, 0, 0);
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[cb].y, fb[s].y, acc[cb], 0, 0, 0);
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[cb].z, fb[s].z, acc[cb], 0, 0, 0);
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[cb].w, fb[s].w, acc[cb], 0, 0, 0);
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) fb[s] = fn[s];
    }
    // the sequential scan's rule (best = -inf, class 0; strictly greater replaces; NaNs never do), per lane half and then merged
    float best = -INFINITY;
    int bi = 4 * h < C ? 4 * h : 0x7fffffff;      // this half's first class
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int c = cb * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
        const float x = acc[cb][q];
        if ((cb + 1 < NCB || c < C) && x > best) {      // (only the last class block can hold padding: C > 32 (NCB - 1))
          best = x;
          bi = c;
        }
      }
    const float ob = __shfl_xor(best, 32, 64);
    const int oi = __shfl_xor(bi, 32, 64);
    if (ob > best || (ob == best && oi < bi)) {
      best = ob;
      bi = oi;
    }
    if (h == 0 && t * 32 + j < V) amax[v] = bi;
  }
}

// Export path (original geometry): classes c0 .. c0 + cg - 1 of the normalised ensemble logits as doubles,
// dst[x][y][z][j] = sum_m (W_m[c0 + j] . F_m(sv)) / n(sv) + bsum[c0 + j] - the input of the dgtta_resample_axis passes, as
// dgtta_logits_chunk_f64 is for the logits-space accumulator (products and sums in double).
__global__ void feature_logits_chunk_kernel(const float *__restrict__ facc, int64_t member_stride, const float *__restrict__ nsum,
                                            const float *__restrict__ w, const float *__restrict__ bsum, double *__restrict__ dst,
                                            int M, int C, int Y, int Z, int x0, int y0, int z0, int ys, int zs, int c0, int cg,
                                            int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int j = (int)(i % cg);
  const int64_t v = i / cg;
  const int z = (int)(v % zs), y = (int)((v / zs) % ys);
  const int64_t x = v / ((int64_t)zs * ys);
  const int64_t sv = ((x + x0) * Y + (y + y0)) * Z + (z + z0);
  double s = 0.0;
  for (int m = 0; m < M; ++m) {
    const float *f = facc + m * member_stride + sv * WF_CIN;
    const float *wr = w + ((int64_t)m * C + c0 + j) * WF_CIN;
    for (int k = 0; k < WF_CIN; ++k) s += (double)wr[k] * (double)f[k];
  }
  dst[i] = s / (double)nsum[sv] + (double)bsum[c0 + j];
}

}  // namespace

extern "C" int dgtta_feature_window_accumulate(const void *z, const float *gauss, float *facc, float *nsum, int Cin, int PD, int PH,
                                               int PW, int X, int Y, int Z, int x0, int y0, int z0, int dtype, void *stream) {
  DG_REQUIRE(z && gauss && facc, DGTTA_ERR_BADARG, "feature_window_accumulate: null pointer");
  DG_REQUIRE(Cin == WF_CIN, DGTTA_ERR_UNSUPPORTED, "feature_window_accumulate: built for %d feature channels (got %d)", WF_CIN, Cin);
  DG_REQUIRE(dtype == DGTTA_F32 || dtype == DGTTA_BF16 || dtype == DGTTA_F16, DGTTA_ERR_BADARG, "feature_window_accumulate: dtype %d", dtype);
  DG_REQUIRE(PD > 0 && PH > 0 && PW > 0 && x0 >= 0 && y0 >= 0 && z0 >= 0 && x0 + PD <= X && y0 + PH <= Y && z0 + PW <= Z,
             DGTTA_ERR_BADARG, "feature_window_accumulate: window outside the volume");
  DG_REQUIRE(((uintptr_t)z & 15) == 0 && ((uintptr_t)facc & 15) == 0, DGTTA_ERR_BADARG, "feature_window_accumulate: unaligned buffer");
  const int64_t total = (int64_t)PD * PH * PW * 8;
  const int64_t nblk = cdiv64(total, 256);
  DG_REQUIRE(nblk < (1ll << 31), DGTTA_ERR_UNSUPPORTED, "feature_window_accumulate: window too large");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DGTTA_F32)
    hipLaunchKernelGGL(feature_accumulate_kernel<float>, dim3((unsigned)nblk), dim3(256), 0, st, (const float *)z, gauss, facc, nsum, PH,
                       PW, Y, Z, x0, y0, z0, total);
  else if (dtype == DGTTA_BF16)
    hipLaunchKernelGGL(feature_accumulate_kernel<bf16_t>, dim3((unsigned)nblk), dim3(256), 0, st, (const bf16_t *)z, gauss, facc, nsum,
                       PH, PW, Y, Z, x0, y0, z0, total);
  else
    hipLaunchKernelGGL(feature_accumulate_kernel<f16_t>, dim3((unsigned)nblk), dim3(256), 0, st, (const f16_t *)z, gauss, facc, nsum, PH,
                       PW, Y, Z, x0, y0, z0, total);
  DG_CHECK_LAUNCH("feature_accumulate_kernel");
  return DGTTA_OK;
}

extern "C" int dgtta_feature_window_accumulate_norm(const void *y, const float *mean_rstd, const float *gamma, const float *beta, float slope,
                                                    const float *gauss, float *facc, float *nsum, int Cin, int PD, int PH, int PW, int X,
                                                    int Y, int Z, int x0, int y0, int z0, int dtype, void *stream) {
  DG_REQUIRE(y && mean_rstd && gamma && beta && gauss && facc, DGTTA_ERR_BADARG, "feature_window_accumulate_norm: null pointer");
  DG_REQUIRE(Cin == WF_CIN, DGTTA_ERR_UNSUPPORTED, "feature_window_accumulate_norm: built for %d feature channels (got %d)", WF_CIN, Cin);
  DG_REQUIRE(dtype == DGTTA_F32 || dtype == DGTTA_BF16 || dtype == DGTTA_F16, DGTTA_ERR_BADARG, "feature_window_accumulate_norm: dtype %d", dtype);
  DG_REQUIRE(PD > 0 && PH > 0 && PW > 0 && x0 >= 0 && y0 >= 0 && z0 >= 0 && x0 + PD <= X && y0 + PH <= Y && z0 + PW <= Z,
             DGTTA_ERR_BADARG, "feature_window_accumulate_norm: window outside the volume");
  DG_REQUIRE(((uintptr_t)y & 15) == 0 && ((uintptr_t)facc & 15) == 0, DGTTA_ERR_BADARG, "feature_window_accumulate_norm: unaligned buffer");
  const int64_t total = (int64_t)PD * PH * PW * 8;
  const int64_t nblk = cdiv64(total, 256);
  DG_REQUIRE(nblk < (1ll << 31), DGTTA_ERR_UNSUPPORTED, "feature_window_accumulate_norm: window too large");
  hipStream_t st = (hipStream_t)stream;
#define WF_LAUNCH(T)                                                                                                              \
  hipLaunchKernelGGL(feature_accumulate_norm_kernel<T>, dim3((unsigned)nblk), dim3(256), 0, st, (const T *)y, mean_rstd, gamma, beta, \
                     slope, gauss, facc, nsum, PH, PW, Y, Z, x0, y0, z0, total)
  if (dtype == DGTTA_F32) WF_LAUNCH(float);
  else if (dtype == DGTTA_BF16) WF_LAUNCH(bf16_t);
  else WF_LAUNCH(f16_t);
#undef WF_LAUNCH
  DG_CHECK_LAUNCH("feature_accumulate_norm_kernel");
  return DGTTA_OK;
}

extern "C" int dgtta_feature_window_accumulate_multi(const void *const *h_srcs, const float *const *h_mean_rstds, const int *h_zoffs, int nsrc,
                                                     const float *gamma, const float *beta, float slope, const float *gauss, float *facc,
                                                     float *nsum, int Cin, int PD, int PH, int PW, int seg_len, int X, int Y, int Z, int x0,
                                                     int y0, int z0, int dtype, void *stream) {
  DG_REQUIRE(h_srcs && h_zoffs && gauss && facc, DGTTA_ERR_BADARG, "feature_window_accumulate_multi: null pointer");
  DG_REQUIRE(nsrc >= 1 && nsrc <= 4, DGTTA_ERR_UNSUPPORTED, "feature_window_accumulate_multi: 1..4 windows per segment (got %d)", nsrc);
  DG_REQUIRE(Cin == WF_CIN, DGTTA_ERR_UNSUPPORTED, "feature_window_accumulate_multi: built for %d feature channels (got %d)", WF_CIN, Cin);
  DG_REQUIRE(dtype == DGTTA_F32 || dtype == DGTTA_BF16 || dtype == DGTTA_F16, DGTTA_ERR_BADARG, "feature_window_accumulate_multi: dtype %d", dtype);
  const bool norm = h_mean_rstds != nullptr;
  DG_REQUIRE(!norm || (gamma && beta), DGTTA_ERR_BADARG, "feature_window_accumulate_multi: statistics without gamma / beta");
  DG_REQUIRE(PD > 0 && PH > 0 && PW > 0 && seg_len > 0 && seg_len <= PW && x0 >= 0 && y0 >= 0 && z0 >= 0 && x0 + PD <= X && y0 + PH <= Y &&
                 z0 + seg_len <= Z, DGTTA_ERR_BADARG, "feature_window_accumulate_multi: segment outside the volume");
  WfSources ws{};
  for (int k = 0; k < nsrc; ++k) {
    DG_REQUIRE(h_srcs[k] && ((uintptr_t)h_srcs[k] & 15) == 0 && h_zoffs[k] >= 0 && h_zoffs[k] + seg_len <= PW && (!norm || h_mean_rstds[k]),
               DGTTA_ERR_BADARG, "feature_window_accumulate_multi: window %d (null / unaligned source or segment outside the window)", k);
    ws.src[k] = h_srcs[k];
    ws.mean_rstd[k] = norm ? h_mean_rstds[k] : nullptr;
    ws.zoff[k] = h_zoffs[k];
  }
  DG_REQUIRE(((uintptr_t)facc & 15) == 0, DGTTA_ERR_BADARG, "feature_window_accumulate_multi: unaligned accumulator");
  const int64_t total = (int64_t)PD * PH * seg_len * 8;
  const int64_t nblk = cdiv64(total, 256);
  DG_REQUIRE(nblk < (1ll << 31), DGTTA_ERR_UNSUPPORTED, "feature_window_accumulate_multi: segment too large");
  hipStream_t st = (hipStream_t)stream;
#define WFM(T, NS, NRM)                                                                                                                \
  hipLaunchKernelGGL((feature_accumulate_multi_kernel<T, NS, NRM>), dim3((unsigned)nblk), dim3(256), 0, st, ws, gamma, beta, slope, gauss, \
                     facc, nsum, PH, PW, seg_len, Y, Z, x0, y0, z0, total)
#define WFM_NS(T, NRM)                  \
  do {                                  \
    if (nsrc == 1) WFM(T, 1, NRM);      \
    else if (nsrc == 2) WFM(T, 2, NRM); \
    else if (nsrc == 3) WFM(T, 3, NRM); \
    else WFM(T, 4, NRM);                \
  } while (0)
#define WFM_T(NRM)                                   \
  do {                                               \
    if (dtype == DGTTA_F32) WFM_NS(float, NRM);      \
    else if (dtype == DGTTA_BF16) WFM_NS(bf16_t, NRM); \
    else WFM_NS(f16_t, NRM);                         \
  } while (0)
  if (norm) WFM_T(true);
  else WFM_T(false);
#undef WFM_T
#undef WFM_NS
#undef WFM
  DG_CHECK_LAUNCH("feature_accumulate_multi_kernel");
  return DGTTA_OK;
}

extern "C" int dgtta_feature_head_argmax(const float *facc, int64_t member_stride, const float *nsum, const float *w, const float *bsum,
                                         int M, int Cin, int C, int64_t V, int64_t *argmax_out, void *stream) {
  DG_REQUIRE(facc && nsum && w && bsum && argmax_out, DGTTA_ERR_BADARG, "feature_head_argmax: null pointer");
  DG_REQUIRE(Cin == WF_CIN, DGTTA_ERR_UNSUPPORTED, "feature_head_argmax: built for %d feature channels (got %d)", WF_CIN, Cin);
  DG_REQUIRE(M >= 1 && C >= 1 && V >= 1 && (M == 1 || member_stride >= V * WF_CIN), DGTTA_ERR_BADARG, "feature_head_argmax: bad sizes");
  DG_REQUIRE(((uintptr_t)facc & 15) == 0 && (member_stride & 3) == 0, DGTTA_ERR_BADARG, "feature_head_argmax: unaligned accumulator");
  // the matrix-core kernel when its weight image fits the LDS (DGTTA_FEATURE_HEAD_MFMA=0: the vector-ALU kernel)
  const int ncb = (C + 31) / 32;
  const size_t lds_m = ((size_t)M * 32 * ncb * WF_WPITCH + 32 * ncb) * sizeof(float);
  const bool use_mfma = dgtta_switches().feature_head_mfma != '0';
  if (use_mfma && ncb <= 4 && lds_m <= 160 * 1024) {
    static DynLdsOnce once_m[4];
    const int64_t ntile = cdiv64(V, 32);
    const unsigned grid = (unsigned)(cdiv64(ntile, 4) < 2048 ? cdiv64(ntile, 4) : 2048);
#define WF_MFMA(N)                                                                                                                   \
  do {                                                                                                                               \
    DG_REQUIRE(ensure_dyn_lds(once_m[N - 1], (const void *)feature_head_argmax_mfma_kernel<N>, 160 * 1024) == hipSuccess,            \
               DGTTA_ERR_LAUNCH, "feature_head_argmax: cannot raise the dynamic LDS limit");                                         \
    hipLaunchKernelGGL(feature_head_argmax_mfma_kernel<N>, dim3(grid), dim3(256), lds_m, (hipStream_t)stream, facc, member_stride,    \
                       nsum, w, bsum, M, C, V, argmax_out);                                                                          \
  } while (0)
    if (ncb == 1) WF_MFMA(1);
    else if (ncb == 2) WF_MFMA(2);
    else if (ncb == 3) WF_MFMA(3);
    else WF_MFMA(4);
#undef WF_MFMA
    DG_CHECK_LAUNCH("feature_head_argmax_mfma_kernel");
    return DGTTA_OK;
  }
  const size_t lds = M > 1 ? (size_t)C * 256 * sizeof(float) : 0;
  DG_REQUIRE(lds <= 160 * 1024, DGTTA_ERR_UNSUPPORTED, "feature_head_argmax: %d classes x several members exceed the LDS (<= 160)", C);
  static DynLdsOnce once;
  if (lds > 64 * 1024)
    DG_REQUIRE(ensure_dyn_lds(once, (const void *)feature_head_argmax_kernel, 160 * 1024) == hipSuccess, DGTTA_ERR_LAUNCH,
               "feature_head_argmax: cannot raise the dynamic LDS limit");
  const int64_t nblk = cdiv64(V, 256);
  const unsigned grid = (unsigned)(nblk < 8192 ? nblk : 8192);
  hipLaunchKernelGGL(feature_head_argmax_kernel, dim3(grid), dim3(256), lds, (hipStream_t)stream, facc, member_stride, nsum, w, bsum, M, C,
                     V, argmax_out);
  DG_CHECK_LAUNCH("feature_head_argmax_kernel");
  return DGTTA_OK;
}

extern "C" int dgtta_feature_logits_chunk_f64(const float *facc, int64_t member_stride, const float *nsum, const float *w,
                                              const float *bsum, double *dst, int M, int Cin, int C, int X, int Y, int Z, int x0, int y0,
                                              int z0, int xs, int ys, int zs, int c0, int cg, void *stream) {
  DG_REQUIRE(facc && nsum && w && bsum && dst, DGTTA_ERR_BADARG, "feature_logits_chunk: null pointer");
  DG_REQUIRE(Cin == WF_CIN, DGTTA_ERR_UNSUPPORTED, "feature_logits_chunk: built for %d feature channels (got %d)", WF_CIN, Cin);
  DG_REQUIRE(M >= 1 && C > 0 && cg > 0 && c0 >= 0 && c0 + cg <= C, DGTTA_ERR_BADARG, "feature_logits_chunk: class range outside [0,%d)", C);
  DG_REQUIRE(x0 >= 0 && y0 >= 0 && z0 >= 0 && xs > 0 && ys > 0 && zs > 0 && x0 + xs <= X && y0 + ys <= Y && z0 + zs <= Z, DGTTA_ERR_BADARG,
             "feature_logits_chunk: crop outside the volume");
  DG_REQUIRE(M == 1 || member_stride >= (int64_t)X * Y * Z * WF_CIN, DGTTA_ERR_BADARG, "feature_logits_chunk: member stride");
  const int64_t total = (int64_t)xs * ys * zs * cg;
  DG_REQUIRE(cdiv64(total, 256) < (1ll << 31), DGTTA_ERR_UNSUPPORTED, "feature_logits_chunk: too many values");
  hipLaunchKernelGGL(feature_logits_chunk_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream, facc, member_stride,
                     nsum, w, bsum, dst, M, C, Y, Z, x0, y0, z0, ys, zs, c0, cg, total);
  DG_CHECK_LAUNCH("feature_logits_chunk_kernel");
  return DGTTA_OK;
}
