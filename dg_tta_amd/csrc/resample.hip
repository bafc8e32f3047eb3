// Spline resampling of a volume along ONE axis, as nnU-Net's preprocessing does it through skimage.transform.resize
// (order 0 / 1 / 3, mode='edge', anti_aliasing=False) = scipy.ndimage.zoom(order, mode='nearest', grid_mode=True)
// [3P: nnunetv2==2.2.1 preprocessing/resampling/default_resampling.py, reached from DefaultPreprocessor.run_case at
// dg_tta/tta/nnunet_utils.py:183-189].  The nD operation is separable, so the caller applies this pass per axis.
//
//   x(o) = (o + 0.5) * n / m - 0.5                 output sample o of m on an input line of n (grid_mode)
//   order 0: nearest, floor(x + 0.5) clamped      order 1: linear between floor(x), floor(x)+1 (clamped = edge)
//   order 3: cubic B-spline: the line is padded by 12 edge samples per side (scipy: _prepad_for_spline_filter for
//            mode 'nearest'), prefiltered (pole z = sqrt(3) - 2, gain 6, mirror initialisation on the padded line) and
//            evaluated with the 4 B-spline weights at x + 12.  All arithmetic in double, like scipy.
// One thread per line; a line is (outer o, inner i): element k at ((o * n + k) * inner + i).  Consecutive threads take
// consecutive inner indices, i.e. consecutive addresses for every axis but the last.  The prefiltered coefficients of a
// line live in a workspace laid out [k][line] (coalesced across lines).  HBM-bound; runs once per case.
#include "common.h"

namespace {

constexpr int NPAD = 12;

__global__ void resample_axis_kernel(const double *__restrict__ src, double *__restrict__ dst, double *__restrict__ coef,
                                     int64_t nlines, int n, int m, int64_t inner, int order) {
  const int64_t line = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (line >= nlines) return;
  const int64_t o = line / inner, i = line % inner;
  const double *s = src + o * n * inner + i;
  double *d = dst + o * m * inner + i;
  const double ratio = (double)n / (double)m;
  if (order == 0) {
    for (int k = 0; k < m; ++k) {
      const double x = ((double)k + 0.5) * ratio - 0.5;
      int j = (int)floor(x + 0.5);
      j = j < 0 ? 0 : (j > n - 1 ? n - 1 : j);
      d[(int64_t)k * inner] = s[(int64_t)j * inner];
    }
    return;
  }
  if (order == 1) {
    for (int k = 0; k < m; ++k) {
      const double x = ((double)k + 0.5) * ratio - 0.5;
      const double fl = floor(x), t = x - fl;
      int j0 = (int)fl, j1 = j0 + 1;
      j0 = j0 < 0 ? 0 : (j0 > n - 1 ? n - 1 : j0);
      j1 = j1 < 0 ? 0 : (j1 > n - 1 ? n - 1 : j1);
      d[(int64_t)k * inner] = (1.0 - t) * s[(int64_t)j0 * inner] + t * s[(int64_t)j1 * inner];
    }
    return;
  }
  // ---- order 3
  const int L = n + 2 * NPAD;
  double *c = coef + line;                 // element k at c[k * nlines]
  const double z = -0.26794919243112270647;   // sqrt(3) - 2
  const double gain = 6.0;                  // (1 - z)(1 - 1/z)
  auto xin = [&](int k) {                   // padded input, edge replicated
    int j = k - NPAD;
    j = j < 0 ? 0 : (j > n - 1 ? n - 1 : j);
    return s[(int64_t)j * inner] * gain;
  };
  // causal initialisation, mirror boundary on the padded line (scipy ni_splines.c: _init_causal_mirror)
  {
    const double z_n_1 = pow(z, (double)(L - 1));
    double c0 = xin(0) + z_n_1 * xin(L - 1);
    double z_i = z;
    for (int k = 1; k < L - 1; ++k) {
      c0 += z_i * (xin(k) + z_n_1 * xin(L - 1 - k));
      z_i *= z;
      if (fabs(z_i) < 1e-300) break;       // the remaining terms are exact zeros in double
    }
    c0 /= 1.0 - z_n_1 * z_n_1;
    c[0] = c0;
    double prev = c0;
    for (int k = 1; k < L; ++k) {
      prev = xin(k) + z * prev;
      c[(int64_t)k * nlines] = prev;
    }
  }
  // anticausal pass (scipy: _init_anticausal_mirror)
  {
    double last = c[(int64_t)(L - 1) * nlines], before = c[(int64_t)(L - 2) * nlines];
    double cur = (z * before + last) * z / (z * z - 1.0);
    c[(int64_t)(L - 1) * nlines] = cur;
    for (int k = L - 2; k >= 0; --k) {
      cur = z * (cur - c[(int64_t)k * nlines]);
      c[(int64_t)k * nlines] = cur;
    }
  }
  for (int k = 0; k < m; ++k) {
    const double x = ((double)k + 0.5) * ratio - 0.5 + (double)NPAD;
    const double fl = floor(x), t = x - fl;
    const int j = (int)fl - 1;
    // cubic B-spline weights at offsets -1, 0, 1, 2 (scipy get_spline_interpolation_weights, order 3)
    const double w1 = (t * t * (t - 2.0) * 3.0 + 4.0) / 6.0;
    const double u = 1.0 - t;
    const double w2 = (u * u * (u - 2.0) * 3.0 + 4.0) / 6.0;
    const double w0 = u * u * u / 6.0;
    const double w3 = 1.0 - w0 - w1 - w2;
    d[(int64_t)k * inner] = w0 * c[(int64_t)j * nlines] + w1 * c[(int64_t)(j + 1) * nlines] + w2 * c[(int64_t)(j + 2) * nlines] +
                            w3 * c[(int64_t)(j + 3) * nlines];
  }
}

// ---- export of a prediction in the case's original geometry (nnU-Net's
// convert_predicted_logits_to_segmentation_with_correct_shape, reached from dg_tta/tta/nnunet_utils.py:208-230): the
// accumulated window logits are normalised, resampled class group by class group (the passes above) and reduced to a label
// map by a running argmax, so that the 105-class volume never exists twice.
template <typename ACC>
__global__ void logits_chunk_kernel(const ACC *__restrict__ acc, const float *__restrict__ nsum, double *__restrict__ dst,
                                    int C, int Y, int Z, int x0, int y0, int z0, int ys, int zs, int c0, int cg,
                                    int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int j = (int)(i % cg);
  const int64_t v = i / cg;
  const int z = (int)(v % zs), y = (int)((v / zs) % ys);
  const int64_t x = v / ((int64_t)zs * ys);
  const int64_t sv = ((x + x0) * Y + (y + y0)) * Z + (z + z0);
  dst[i] = (double)(ld_f<ACC>(acc + sv * C + c0 + j) / nsum[sv]);
}

__global__ void argmax_merge_kernel(const double *__restrict__ vals, int64_t V, int cg, int c0, double *__restrict__ best_val,
                                    int *__restrict__ best_idx, int first) {
  const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  double bv = first ? vals[v * cg] : best_val[v];
  int bi = first ? c0 : best_idx[v];
  for (int j = first ? 1 : 0; j < cg; ++j) {
    const double x = vals[v * cg + j];
    if (x > bv) {            // strict: the first maximum wins, as argmax does
      bv = x;
      bi = c0 + j;
    }
  }
  best_val[v] = bv;
  best_idx[v] = bi;
}

}  // namespace

extern "C" int dgtta_logits_chunk_f64_t(const void *acc, const float *nsum, double *dst, int C, int X, int Y, int Z, int x0,
                                        int y0, int z0, int xs, int ys, int zs, int c0, int cg, int acc_dtype, void *stream) {
  DG_REQUIRE(acc && nsum && dst, DGTTA_ERR_BADARG, "logits_chunk: null pointer");
  DG_REQUIRE(C > 0 && cg > 0 && c0 >= 0 && c0 + cg <= C, DGTTA_ERR_BADARG, "logits_chunk: class range outside [0,%d)", C);
  DG_REQUIRE(x0 >= 0 && y0 >= 0 && z0 >= 0 && xs > 0 && ys > 0 && zs > 0 && x0 + xs <= X && y0 + ys <= Y && z0 + zs <= Z,
             DGTTA_ERR_BADARG, "logits_chunk: crop outside the volume");
  DG_REQUIRE(acc_dtype == DGTTA_F32 || acc_dtype == DGTTA_F16, DGTTA_ERR_UNSUPPORTED, "logits_chunk: the accumulator is fp32 or fp16");
  const int64_t total = (int64_t)xs * ys * zs * cg;
  if (acc_dtype == DGTTA_F32)
    hipLaunchKernelGGL(logits_chunk_kernel<float>, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float *)acc, nsum, dst, C, Y, Z, x0, y0, z0, ys, zs, c0, cg, total);
  else
    hipLaunchKernelGGL(logits_chunk_kernel<f16_t>, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const f16_t *)acc, nsum, dst, C, Y, Z, x0, y0, z0, ys, zs, c0, cg, total);
  DG_CHECK_LAUNCH("logits_chunk_kernel");
  return DGTTA_OK;
}

extern "C" int dgtta_logits_chunk_f64(const float *acc, const float *nsum, double *dst, int C, int X, int Y, int Z, int x0,
                                      int y0, int z0, int xs, int ys, int zs, int c0, int cg, void *stream) {
  return dgtta_logits_chunk_f64_t(acc, nsum, dst, C, X, Y, Z, x0, y0, z0, xs, ys, zs, c0, cg, DGTTA_F32, stream);
}

extern "C" int dgtta_argmax_merge_f64(const double *vals, int64_t V, int cg, int c0, double *best_val, int *best_idx,
                                      int first, void *stream) {
  DG_REQUIRE(vals && best_val && best_idx, DGTTA_ERR_BADARG, "argmax_merge: null pointer");
  DG_REQUIRE(V > 0 && cg > 0 && c0 >= 0, DGTTA_ERR_BADARG, "argmax_merge: bad dims");
  hipLaunchKernelGGL(argmax_merge_kernel, dim3((unsigned)cdiv64(V, 256)), dim3(256), 0, (hipStream_t)stream, vals, V, cg, c0,
                     best_val, best_idx, first);
  DG_CHECK_LAUNCH("argmax_merge_kernel");
  return DGTTA_OK;
}

extern "C" size_t dgtta_resample_axis_ws_bytes(int64_t outer, int n, int64_t inner, int order) {
  if (outer <= 0 || n <= 0 || inner <= 0) return 0;      // a size query of an empty problem (the launchers reject it with DGTTA_ERR_BADARG)
  return order == 3 ? (size_t)outer * (size_t)inner * (size_t)(n + 2 * NPAD) * sizeof(double) : 256;
}

extern "C" int dgtta_resample_axis(const double *src, double *dst, void *ws, size_t ws_bytes, int64_t outer, int n, int m,
                                   int64_t inner, int order, void *stream) {
  DG_REQUIRE(src && dst, DGTTA_ERR_BADARG, "resample_axis: null pointer");
  DG_REQUIRE(outer > 0 && inner > 0 && n > 0 && m > 0, DGTTA_ERR_BADARG, "resample_axis: bad dims");
  DG_REQUIRE(order == 0 || order == 1 || order == 3, DGTTA_ERR_UNSUPPORTED, "resample_axis: order %d not in {0,1,3}", order);
  DG_REQUIRE(order != 3 || (ws && ws_bytes >= dgtta_resample_axis_ws_bytes(outer, n, inner, order)), DGTTA_ERR_WORKSPACE,
             "resample_axis: workspace too small");
  DG_REQUIRE(order != 3 || n >= 2, DGTTA_ERR_UNSUPPORTED, "resample_axis: cubic needs at least 2 samples per line");
  const int64_t nlines = outer * inner;
  DG_REQUIRE(nlines < (1ll << 31) * 256, DGTTA_ERR_UNSUPPORTED, "resample_axis: too many lines");
  hipLaunchKernelGGL(resample_axis_kernel, dim3((unsigned)cdiv64(nlines, 256)), dim3(256), 0, (hipStream_t)stream, src, dst,
                     (double *)ws, nlines, n, m, inner, order);
  DG_CHECK_LAUNCH("resample_axis_kernel");
  return DGTTA_OK;
}
