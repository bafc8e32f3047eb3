// Masked-softmax soft-Dice consistency loss, forward and backward, single pass each (HBM-bound).
// Replaces dg_tta/tta/tta.py:263-271 and soft_dice_loss (dg_tta/tta/torch_utils.py:90-104):
//   mask = (sum_c la > 0) * (sum_c lb > 0);  a = softmax_c(la)*mask;  b = softmax_c(lb)*mask
//   nom_c = mean_v(2ab); den_c = 0.5*mean_v((a+b)^2); dice = nom/den (no eps; all-zero guard -> 1)
//   loss = 1 - mean_{b, c>=start} dice
// Logits are voxel-major [B][V][ldc] fp32 (one row of C classes per voxel = one coalesced read per lane).
// fwd: wavefront reductions per class -> per-workgroup partials (double) -> fixed-order finalize.
// bwd: recomputes the softmax per voxel from the logits (cheaper than saving it) and applies the closed-form
//      gradient with per-class coefficients P,Q prepared by the finalize kernel.
// Algorithmic HBM bytes per voxel: fwd 2*C*4 read; bwd 2*C*4 read + 2*C*4 write.
#include "common.h"

namespace {

constexpr int NT = 256;
constexpr int MAXC = 128;

struct VoxelSoftmax {
  float mx, inv_sum, mask_part;  // mask_part = (sum_c logits > 0)
};

__device__ __forceinline__ VoxelSoftmax voxel_stats(const float *row, int C) {
  float mx = row[0], s = row[0];
  for (int c = 1; c < C; ++c) {
    float v = row[c];
    mx = fmaxf(mx, v);
    s += v;
  }
  float se = 0.f;
  for (int c = 0; c < C; ++c) se += expf(row[c] - mx);
  VoxelSoftmax r;
  r.mx = mx;
  r.inv_sum = se;  // holds the SUM; callers divide
  r.mask_part = s > 0.0f ? 1.0f : 0.0f;
  return r;
}

__global__ __launch_bounds__(NT) void softdice_fwd_kernel(const float *__restrict__ la, const float *__restrict__ lb,
                                                          double *__restrict__ partial, int C, int64_t V, int ldc) {
  __shared__ float acc[NT / 64][MAXC][2];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int b = blockIdx.y;
  for (int i = threadIdx.x; i < (NT / 64) * MAXC * 2; i += NT) (&acc[0][0][0])[i] = 0.f;
  __syncthreads();
  const float *pa = la + (int64_t)b * V * ldc, *pb = lb + (int64_t)b * V * ldc;
  // every lane of a wave runs the same number of iterations (wave-uniform loop; inactive voxels contribute 0)
  const int64_t stride = (int64_t)gridDim.x * NT;
  for (int64_t base = (int64_t)blockIdx.x * NT + wv * 64; base < V; base += stride) {
    const int64_t v = base + lane;
    const bool on = v < V;
    const float *ra = pa + (on ? v : 0) * ldc, *rb = pb + (on ? v : 0) * ldc;
    VoxelSoftmax sa = voxel_stats(ra, C), sb = voxel_stats(rb, C);
    const float m = on ? sa.mask_part * sb.mask_part : 0.f;
    for (int c = 0; c < C; ++c) {
      float a = (expf(ra[c] - sa.mx) / sa.inv_sum) * m;
      float bq = (expf(rb[c] - sb.mx) / sb.inv_sum) * m;
      float s1 = wave_sum((2.0f * a) * bq);
      float t = a + bq;
      float s2 = wave_sum(t * t);
      if (lane == 0) {
        acc[wv][c][0] += s1;
        acc[wv][c][1] += s2;
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < C * 2; i += NT) {
    int c = i >> 1, k = i & 1;
    double s = 0.0;
    for (int w = 0; w < NT / 64; ++w) s += (double)acc[w][c][k];
    partial[(((int64_t)b * gridDim.x + blockIdx.x) * C + c) * 2 + k] = s;
  }
}

// one workgroup: sums partials in fixed order, forms dice/loss and the backward coefficients
__global__ void softdice_finalize_kernel(const double *__restrict__ partial, int nblk, int B, int C, int64_t V,
                                         int start_class, float *__restrict__ dice, float *__restrict__ loss,
                                         float *__restrict__ coef) {
  __shared__ float s_nom[8 * MAXC], s_den[8 * MAXC];
  __shared__ float s_flag, s_loss;
  const int n = B * C;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    int b = i / C, c = i % C;
    double s1 = 0.0, s2 = 0.0;
    for (int k = 0; k < nblk; ++k) {
      s1 += partial[(((int64_t)b * nblk + k) * C + c) * 2 + 0];
      s2 += partial[(((int64_t)b * nblk + k) * C + c) * 2 + 1];
    }
    s_nom[i] = (float)(s1 / (double)V);
    s_den[i] = 0.5f * (float)(s2 / (double)V);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.f;
    for (int i = 0; i < n; ++i) tot += s_den[i];
    s_flag = (tot == 0.0f) ? 1.f : 0.f;
  }
  __syncthreads();
  const bool all_zero = s_flag != 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) dice[i] = all_zero ? 1.0f : s_nom[i] / s_den[i];
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
    if (!all_zero && c >= start_class) {
      float den = s_den[i], nom = s_nom[i];
      P = -invN * 2.0f / ((float)V * den);
      Q = invN * nom / (den * den * (float)V);
    }
    coef[2 * i + 0] = P;
    coef[2 * i + 1] = Q;
  }
}

__global__ __launch_bounds__(NT) void softdice_bwd_kernel(const float *__restrict__ la, const float *__restrict__ lb,
                                                          float *__restrict__ ga, float *__restrict__ gb,
                                                          const float *__restrict__ coef, float scale_h,
                                                          const float *__restrict__ scale_dev, int C, int64_t V,
                                                          int ldc) {
  __shared__ float sc[MAXC * 2];
  const float scale = scale_dev ? scale_h * scale_dev[0] : scale_h;
  const int b = blockIdx.y;
  for (int i = threadIdx.x; i < 2 * C; i += NT) sc[i] = coef[(int64_t)b * C * 2 + i];
  __syncthreads();
  for (int64_t v = (int64_t)blockIdx.x * NT + threadIdx.x; v < V; v += (int64_t)gridDim.x * NT) {
    const int64_t ro = ((int64_t)b * V + v) * ldc;
    const float *ra = la + ro, *rb = lb + ro;
    VoxelSoftmax sa = voxel_stats(ra, C), sb = voxel_stats(rb, C);
    const float m = sa.mask_part * sb.mask_part;
    float dot_a = 0.f, dot_b = 0.f;
    for (int c = 0; c < C; ++c) {
      float pa = expf(ra[c] - sa.mx) / sa.inv_sum, pb = expf(rb[c] - sb.mx) / sb.inv_sum;
      float a = pa * m, bq = pb * m;
      float P = sc[2 * c], Q = sc[2 * c + 1];
      float t = Q * (a + bq);
      dot_a += (P * bq + t) * pa;
      dot_b += (P * a + t) * pb;
    }
    for (int c = 0; c < C; ++c) {
      float pa = expf(ra[c] - sa.mx) / sa.inv_sum, pb = expf(rb[c] - sb.mx) / sb.inv_sum;
      float a = pa * m, bq = pb * m;
      float P = sc[2 * c], Q = sc[2 * c + 1];
      float t = Q * (a + bq);
      ga[ro + c] = scale * m * pa * ((P * bq + t) - dot_a);
      gb[ro + c] = scale * m * pb * ((P * a + t) - dot_b);
    }
  }
}

int nblocks_for(int64_t V) {
  int64_t b = (V + NT - 1) / NT;
  return (int)(b < 1024 ? b : 1024);
}

}  // namespace

// ws layout: [partials: B*nblk*C*2 double][coef: B*C*2 float]
extern "C" size_t dgtta_softdice_ws_bytes(int B, int C, int64_t V) {
  return align_up((size_t)B * nblocks_for(V) * C * 2 * sizeof(double), 256) +
         align_up((size_t)B * C * 2 * sizeof(float), 256);
}

extern "C" int dgtta_softdice_fwd(const float *la, const float *lb, float *dice, float *loss, void *ws, size_t ws_bytes,
                                  int B, int C, int64_t V, int ldc, int start_class, void *stream) {
  DG_REQUIRE(la && lb && dice && loss && ws, DGTTA_ERR_BADARG, "softdice_fwd: null pointer");
  DG_REQUIRE(B > 0 && B <= 8 && C > 0 && C <= MAXC && V > 0 && ldc >= C, DGTTA_ERR_BADARG,
             "softdice_fwd: need 1<=B<=8, 1<=C<=%d, ldc>=C (B=%d C=%d ldc=%d)", MAXC, B, C, ldc);
  DG_REQUIRE(start_class >= 0 && start_class < C, DGTTA_ERR_BADARG, "softdice_fwd: bad start_class");
  DG_REQUIRE(ws_bytes >= dgtta_softdice_ws_bytes(B, C, V), DGTTA_ERR_WORKSPACE, "softdice_fwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int nblk = nblocks_for(V);
  double *partial = (double *)ws;
  float *coef = (float *)((char *)ws + align_up((size_t)B * nblk * C * 2 * sizeof(double), 256));
  hipLaunchKernelGGL(softdice_fwd_kernel, dim3(nblk, B), dim3(NT), 0, st, la, lb, partial, C, V, ldc);
  DG_CHECK_LAUNCH("softdice_fwd_kernel");
  hipLaunchKernelGGL(softdice_finalize_kernel, dim3(1), dim3(256), 0, st, partial, nblk, B, C, V, start_class, dice, loss,
                     coef);
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
  int64_t gb = (V + NT - 1) / NT;
  if (gb > 4096) gb = 4096;
  hipLaunchKernelGGL(softdice_bwd_kernel, dim3((int)gb, B), dim3(NT), 0, st, la, lb, grad_la, grad_lb, coef, grad_scale,
                     grad_scale_dev, C, V, ldc);
  DG_CHECK_LAUNCH("softdice_bwd_kernel");
  return DGTTA_OK;
}
