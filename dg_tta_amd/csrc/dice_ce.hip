// Supervised segmentation loss of the PRE-TRAINING side: soft Dice + cross-entropy against an integer
// label map, forward and backward.  The trainers of dg_tta/pretraining/nnUNetTrainer_{GIN,MIND,GIN_MIND}.py:38-59 inherit
// nnU-Net's nnUNetTrainer, whose loss is DC_and_CE_loss [3P nnunetv2==2.2.1: SoftDiceLoss(batch_dice=False, do_bg=False,
// smooth=1e-5) + CrossEntropyLoss, weights 1 / 1].  Restated here so that a model can be pre-trained THROUGH this engine
// (bench.py pre-trains the synthetic source domain before it adapts to the shifted target; dg_tta_amd/pretraining):
//   p = softmax_c(z);  y = label (voxels whose label is outside [0, C) are ignored)
//   ce   = -(1/N) sum_v log p[v][y_v]                                   N = valid voxels of the whole batch
//   dc_bc = (2 sum_v p y + s) / (sum_v p + sum_v y + s)                  per sample b and class c, y one-hot
//   loss = ce - mean_{b, c >= first} dc_bc                               first = 1 (do_bg = False) or 0
// Logits are voxel-major [B][V][ldc] fp32 (what the head writes); a workgroup stages 128 rows through LDS as one contiguous run,
// one lane = one voxel inside the tile; HBM-bound and NOT on the timed path.
// Deterministic: per-workgroup partial sums (double) in fixed slots, one finalize workgroup adds them in slot order.
#include "common.h"

namespace {

constexpr int DC_MAXC = 128;
constexpr int DC_TILE = 128;               // voxels per tile: rows are staged through LDS as ONE contiguous run (a lane that walks
                                           // its own 420-byte row straight from memory runs at 0.3 TB/s, DESIGN.md "20x trap")
constexpr int DC_THREADS = 256;

__host__ __device__ inline int dc_pitch(int C) { return C | 1; }
inline int dc_blocks(int64_t V) {
  const int64_t b = (V + DC_TILE - 1) / DC_TILE;
  return (int)(b < 1024 ? b : 1024);
}
inline size_t dc_partial_doubles(int B, int C, int64_t V) { return (size_t)B * dc_blocks(V) * (3 * C + 2); }

// rows [nv][ld] in memory <-> tile [nv][C | 1] in LDS (odd pitch: a lane per voxel walks its row conflict free)
__device__ __forceinline__ void dc_tile_load(float *tile, const float *src, int nv, int C, int ld, int LDP) {
  if (ld == C) {
    // eight loads in flight per thread before the first LDS store (a load - store loop pays the memory latency per element)
    const int n = nv * C;
    for (int e0 = threadIdx.x; e0 < n; e0 += DC_THREADS * 8) {
      float tmp[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + DC_THREADS * u;
        tmp[u] = e < n ? src[e] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + DC_THREADS * u;
        if (e < n) {
          const int v = e / C;
          tile[v * LDP + (e - v * C)] = tmp[u];
        }
      }
    }
  } else {
    for (int e = threadIdx.x; e < nv * C; e += DC_THREADS) {
      const int v = e / C, c = e - v * C;
      tile[v * LDP + c] = src[(int64_t)v * ld + c];
    }
  }
}
__device__ __forceinline__ void dc_tile_store(const float *tile, float *dst, int nv, int C, int ld, int LDP) {
  for (int e = threadIdx.x; e < nv * C; e += DC_THREADS) {
    const int v = e / C, c = e - v * C;
    dst[(int64_t)v * ld + c] = tile[v * LDP + c];
  }
}

// partial[(b * nblk + blk) * (3C + 2) + {c, C + c, 2C + c, 3C, 3C + 1}] = {sum p y, sum p, sum y, sum -log p_y, valid voxels}
__global__ __launch_bounds__(DC_THREADS) void dice_ce_fwd_kernel(const float *__restrict__ logits, int ldc,
                                                               const int64_t *__restrict__ labels, double *__restrict__ partial,
                                                               int C, int64_t V) {
  extern __shared__ float dc_tile[];               // [DC_TILE][C | 1]
  __shared__ int s_lab[DC_TILE];
  __shared__ float s_red[16];
  const int LDP = dc_pitch(C);
  const int b = blockIdx.y, t = threadIdx.x;
  const float *lb = logits + (int64_t)b * V * ldc;
  const int64_t *yb = labels + (int64_t)b * V;
  double aI = 0.0, aP = 0.0, aY = 0.0, ce = 0.0, nv_tot = 0.0;
  const int64_t ntile = (V + DC_TILE - 1) / DC_TILE;
  for (int64_t tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
    const int64_t v0 = tile * DC_TILE;
    const int nv = V - v0 < DC_TILE ? (int)(V - v0) : DC_TILE;
    dc_tile_load(dc_tile, lb + v0 * ldc, nv, C, ldc, LDP);
    __syncthreads();
    float nlp = 0.f, ok = 0.f;
    if (t < DC_TILE) {
      float *row = dc_tile + t * LDP;
      int y = -1;
      if (t < nv) {
        const int64_t yl = yb[v0 + t];
        y = (yl >= 0 && yl < C) ? (int)yl : -1;
      }
      if (y >= 0) {
        float mx = row[0];
        for (int c = 1; c < C; ++c) mx = fmaxf(mx, row[c]);
        const float zy = row[y];
        float se = 0.f;
        for (int c = 0; c < C; ++c) {
          const float e = expf(row[c] - mx);
          row[c] = e;
          se += e;
        }
        const float inv = 1.0f / se;
        for (int c = 0; c < C; ++c) row[c] *= inv;
        nlp = logf(se) - (zy - mx);
        ok = 1.f;
      } else {
        for (int c = 0; c < C; ++c) row[c] = 0.f;
      }
      s_lab[t] = y;
    }
    const float tce = block_sum(nlp, s_red);          // (also the barrier that publishes the probabilities)
    const float tnv = block_sum(ok, s_red);
    if (t == 0) ce += (double)tce, nv_tot += (double)tnv;
    if (t < C) {
      float sI = 0.f, sP = 0.f, sY = 0.f;
      for (int q = 0; q < DC_TILE; ++q) {
        const float p = dc_tile[q * LDP + t];
        const float hit = s_lab[q] == t ? 1.f : 0.f;
        sI += p * hit;
        sP += p;
        sY += hit;
      }
      aI += (double)sI;
      aP += (double)sP;
      aY += (double)sY;
    }
    __syncthreads();
  }
  double *out = partial + ((int64_t)b * gridDim.x + blockIdx.x) * (3 * C + 2);
  if (t < C) {
    out[t] = aI;
    out[C + t] = aP;
    out[2 * C + t] = aY;
  }
  if (t == 0) {
    out[3 * C] = ce;
    out[3 * C + 1] = nv_tot;
  }
}

// one workgroup: loss[0] = total, loss[1] = ce, loss[2] = -mean dice; dice[B][C]; coef[(b * C + c) * 2 + {0, 1}] = {alpha, beta}
// with dL/dp[v][c] = alpha * [y_v == c] + beta, coef[2 B C] = 1 / N
__global__ __launch_bounds__(1024) void dice_ce_finalize_kernel(const double *__restrict__ partial, int nblk, int B, int C,
                                                             float smooth, int first, float *__restrict__ loss,
                                                             float *__restrict__ dice, float *__restrict__ coef) {
  __shared__ double s_dc[8 * DC_MAXC];
  __shared__ double s_ce, s_nv;
  const int n = B * C, W = 3 * C + 2;
  const double w = 1.0 / (double)(B * (C - first));
  // four lanes per (b, c): lane p adds the partial slots p, p + 4, ... (three independent loads per slot), the quad combines in
  // lane order - a fixed summation order, so the loss is reproducible run to run
  const int p4 = threadIdx.x & 3;
  for (int i = threadIdx.x >> 2; i < ((n + 255) / 256) * 256; i += blockDim.x >> 2) {
    const bool on = i < n;
    const int b = on ? i / C : 0, c = on ? i - b * C : 0;
    double I = 0.0, P = 0.0, Y = 0.0;
    if (on)
      for (int q = p4; q < nblk; q += 4) {
        const double *r = partial + ((int64_t)b * nblk + q) * W;
        I += r[c];
        P += r[C + c];
        Y += r[2 * C + c];
      }
#pragma unroll
    for (int m = 1; m <= 2; m <<= 1) {
      I += __shfl_xor(I, m, 64);
      P += __shfl_xor(P, m, 64);
      Y += __shfl_xor(Y, m, 64);
    }
    if (on && p4 == 0) {
      const double N = 2.0 * I + (double)smooth;
      double D = P + Y + (double)smooth;
      if (D < 1e-8) D = 1e-8;
      const double dc = N / D;
      s_dc[i] = dc;
      dice[i] = (float)dc;
      coef[2 * i] = c >= first ? (float)(-2.0 * w / D) : 0.f;
      coef[2 * i + 1] = c >= first ? (float)(w * N / (D * D)) : 0.f;
    }
  }
  if (threadIdx.x < 64) {      // cross-entropy sum and valid-voxel count: one wave, slots dealt to its lanes, fixed tree
    double ce = 0.0, nv = 0.0;
    for (int q = threadIdx.x; q < B * nblk; q += 64) {
      const double *r = partial + (int64_t)q * W;
      ce += r[3 * C];
      nv += r[3 * C + 1];
    }
    ce = wave_sum_d(ce);
    nv = wave_sum_d(nv);
    if (threadIdx.x == 0) s_ce = ce, s_nv = nv;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double acc = 0.0;
    for (int b = 0; b < B; ++b)
      for (int c = first; c < C; ++c) acc += s_dc[b * C + c];
    const double nv = s_nv > 0.0 ? s_nv : 1.0;
    const double cem = s_ce / nv, dl = -acc * w;
    loss[0] = (float)(cem + dl);
    loss[1] = (float)cem;
    loss[2] = (float)dl;
    coef[2 * n] = (float)(1.0 / nv);
  }
}

__global__ __launch_bounds__(DC_THREADS) void dice_ce_bwd_kernel(const float *__restrict__ logits, int ldc,
                                                               const int64_t *__restrict__ labels, const float *__restrict__ coef,
                                                               float scale, const float *__restrict__ scale_dev,
                                                               float *__restrict__ grad, int ldg, int B, int C, int64_t V) {
  extern __shared__ float dc_tile[];               // [DC_TILE][C | 1]: logits in, gradient out
  __shared__ float s_coef[2 * DC_MAXC];
  const int LDP = dc_pitch(C);
  const int b = blockIdx.y, t = threadIdx.x;
  for (int i = t; i < 2 * C; i += DC_THREADS) s_coef[i] = coef[(int64_t)b * 2 * C + i];
  const float inv_n = coef[(int64_t)B * 2 * C];
  const float sc = scale * (scale_dev ? scale_dev[0] : 1.0f);
  const int64_t ntile = (V + DC_TILE - 1) / DC_TILE;
  for (int64_t tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
    const int64_t v0 = tile * DC_TILE;
    const int nv = V - v0 < DC_TILE ? (int)(V - v0) : DC_TILE;
    __syncthreads();                                 // previous tile stored (and s_coef published)
    dc_tile_load(dc_tile, logits + ((int64_t)b * V + v0) * ldc, nv, C, ldc, LDP);
    __syncthreads();
    if (t < nv) {
      float *row = dc_tile + t * LDP;
      const int64_t yl = labels[(int64_t)b * V + v0 + t];
      if (yl < 0 || yl >= C) {
        for (int c = 0; c < C; ++c) row[c] = 0.f;
      } else {
        const int y = (int)yl;
        float mx = row[0];
        for (int c = 1; c < C; ++c) mx = fmaxf(mx, row[c]);
        float se = 0.f, sb = 0.f;
        for (int c = 0; c < C; ++c) {
          const float e = expf(row[c] - mx);
          row[c] = e;
          se += e;
          sb += e * s_coef[2 * c + 1];
        }
        const float inv = 1.0f / se;
        const float S = row[y] * inv * s_coef[2 * y] + sb * inv;          // sum_c p_c dL/dp_c
        for (int c = 0; c < C; ++c) {
          const float p = row[c] * inv;
          const float hit = c == y ? 1.f : 0.f;
          const float gp = s_coef[2 * c] * hit + s_coef[2 * c + 1];
          row[c] = sc * ((p - hit) * inv_n + p * (gp - S));
        }
      }
    }
    __syncthreads();
    dc_tile_store(dc_tile, grad + ((int64_t)b * V + v0) * ldg, nv, C, ldg, LDP);
  }
}

}  // namespace

// ws: [partials: B * nblk * (3C + 2) double][coef: 2 B C + 1 float], kept between fwd and bwd
extern "C" size_t dgtta_dice_ce_ws_bytes(int B, int C, int64_t V) {
  if (B <= 0 || C <= 0 || V <= 0) return 0;
  return align_up(dc_partial_doubles(B, C, V) * sizeof(double), 256) + align_up(((size_t)2 * B * C + 1) * sizeof(float), 256);
}

extern "C" int dgtta_dice_ce_fwd(const float *logits, int ldc, const int64_t *labels, float *loss3, float *dice, void *ws,
                                 size_t ws_bytes, int B, int C, int64_t V, float smooth, int do_bg, void *stream) {
  DG_REQUIRE(logits && labels && loss3 && dice && ws, DGTTA_ERR_BADARG, "dice_ce_fwd: null pointer");
  DG_REQUIRE(B > 0 && B <= 8 && C >= 2 && C <= DC_MAXC && V > 0 && ldc >= C, DGTTA_ERR_BADARG,
             "dice_ce_fwd: need 1<=B<=8, 2<=C<=%d, ldc>=C (B=%d C=%d ldc=%d)", DC_MAXC, B, C, ldc);
  DG_REQUIRE(ws_bytes >= dgtta_dice_ce_ws_bytes(B, C, V), DGTTA_ERR_WORKSPACE, "dice_ce_fwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int nblk = dc_blocks(V);
  double *partial = (double *)ws;
  float *coef = (float *)((char *)ws + align_up(dc_partial_doubles(B, C, V) * sizeof(double), 256));
  const size_t lds = (size_t)DC_TILE * dc_pitch(C) * sizeof(float);
  static DynLdsOnce once;
  DG_REQUIRE(ensure_dyn_lds(once, (const void *)dice_ce_fwd_kernel, DC_TILE * dc_pitch(DC_MAXC) * (int)sizeof(float)) == hipSuccess,
             DGTTA_ERR_LAUNCH, "dice_ce_fwd: cannot raise the dynamic LDS limit");
  hipLaunchKernelGGL(dice_ce_fwd_kernel, dim3(nblk, B), dim3(DC_THREADS), lds, st, logits, ldc, labels, partial, C, V);
  DG_CHECK_LAUNCH("dice_ce_fwd_kernel");
  hipLaunchKernelGGL(dice_ce_finalize_kernel, dim3(1), dim3(1024), 0, st, partial, nblk, B, C, smooth, do_bg ? 0 : 1, loss3, dice,
                     coef);
  DG_CHECK_LAUNCH("dice_ce_finalize_kernel");
  return DGTTA_OK;
}

extern "C" int dgtta_dice_ce_bwd(const float *logits, int ldc, const int64_t *labels, const void *ws, float grad_scale,
                                 const float *grad_scale_dev, float *grad_logits, int ldg, int B, int C, int64_t V,
                                 void *stream) {
  DG_REQUIRE(logits && labels && ws && grad_logits, DGTTA_ERR_BADARG, "dice_ce_bwd: null pointer");
  DG_REQUIRE(B > 0 && B <= 8 && C >= 2 && C <= DC_MAXC && V > 0 && ldc >= C && ldg >= C, DGTTA_ERR_BADARG, "dice_ce_bwd: bad dims");
  const float *coef = (const float *)((const char *)ws + align_up(dc_partial_doubles(B, C, V) * sizeof(double), 256));
  int64_t gx = (V + DC_TILE - 1) / DC_TILE;
  if (gx > 2048) gx = 2048;
  const size_t lds = (size_t)DC_TILE * dc_pitch(C) * sizeof(float);
  static DynLdsOnce once;
  DG_REQUIRE(ensure_dyn_lds(once, (const void *)dice_ce_bwd_kernel, DC_TILE * dc_pitch(DC_MAXC) * (int)sizeof(float)) == hipSuccess,
             DGTTA_ERR_LAUNCH, "dice_ce_bwd: cannot raise the dynamic LDS limit");
  hipLaunchKernelGGL(dice_ce_bwd_kernel, dim3((unsigned)gx, B), dim3(DC_THREADS), lds, (hipStream_t)stream, logits, ldc, labels, coef,
                     grad_scale, grad_scale_dev, grad_logits, ldg, B, C, V);
  DG_CHECK_LAUNCH("dice_ce_bwd_kernel");
  return DGTTA_OK;
}
