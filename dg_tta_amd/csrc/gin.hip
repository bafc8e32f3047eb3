// GIN random-convolution augmentation chain, fused (1->2->2->2->1 channels, k in {1,3}, LeakyReLU 0.01 on the
// first three layers), alpha-blend with the input and Frobenius re-normalisation.
// Replaces GINGroupConv.forward + GradlessGCReplayNonlinBlock.forward (dg_tta/gin.py:168-230, :59-122).
//
// Pass A (gin_chain_kernel): one workgroup per 8x8x16 output tile; the input tile (+4 halo, zero padded as the
//   reference's padding=k//2 does layer by layer) and all intermediate 2-channel activations live in LDS
//   (ping-pong), so HBM sees one read of x and one write of `mixed`.  k=1 layers are run as 3x3x3 with a
//   zero-filled kernel (adds exact zeros).  Emits per-workgroup partial sums of x^2 and mixed^2.
// Pass B (gin_norm_kernel): fixed-order double sum -> ||x||_F, ||mixed||_F per sample.
// Pass C (gin_scale_kernel): out = mixed * (1/(||mixed||+1e-5)) * ||x||   (gin.py:228, same operation order).
// Algorithmic HBM bytes per voxel: 4 r + 4 w + 4 r + 4 w = 16 B.
#include "common.h"

namespace {

constexpr int TD = 8, TH = 8, TW = 16, HALO = 4, NT = 256;
constexpr int R0D = TD + 8, R0H = TH + 8, R0W = TW + 8;
constexpr int SZ_A = R0D * R0H * R0W;                         // 6144 (also holds layer-2 output: 2*12*12*20=5760)
constexpr int SZ_B = 2 * (TD + 6) * (TH + 6) * (TW + 6);      // 8624 (layer-1 output; layer-3 output 3600)

struct GinArgs {
  const float *ker[4];
  const float *shift[4];
  int ksz[4];
};

__device__ __forceinline__ bool inside(int d, int h, int w, int D, int H, int W) {
  return (unsigned)d < (unsigned)D && (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W;
}

// One layer on an LDS region. in: [CIN][id][ih][iw] (dims = out dims + 2), out: [COUT][od][oh][ow].
// (gd0,gh0,gw0) = global coordinate of out-region origin. Outside-volume outputs are forced to 0.
template <int CIN, int COUT, bool ACT>
__device__ __forceinline__ void gin_layer(const float *sin, float *sout, const float *w /*[COUT][CIN][27]*/,
                                          const float *sh, int od, int oh, int ow, int gd0, int gh0, int gw0, int D,
                                          int H, int W) {
  const int ih = oh + 2, iw = ow + 2, id = od + 2;
  const int n = od * oh * ow;
  for (int i = threadIdx.x; i < n; i += NT) {
    int x = i % ow, y = (i / ow) % oh, z = i / (ow * oh);
    float acc[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[co] = 0.f;
    const bool in_vol = inside(gd0 + z, gh0 + y, gw0 + x, D, H, W);
    if (in_vol) {
#pragma unroll
      for (int ci = 0; ci < CIN; ++ci) {
        const float *p = sin + ((ci * id + z) * ih + y) * iw + x;
#pragma unroll
        for (int kd = 0; kd < 3; ++kd)
#pragma unroll
          for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
              float v = p[(kd * ih + kh) * iw + kw];
#pragma unroll
              for (int co = 0; co < COUT; ++co)
                acc[co] = __builtin_fmaf(w[(co * CIN + ci) * 27 + kd * 9 + kh * 3 + kw], v, acc[co]);
            }
      }
#pragma unroll
      for (int co = 0; co < COUT; ++co) {
        float v = acc[co] + sh[co];
        acc[co] = ACT ? lrelu(v, 0.01f) : v;
      }
    }
#pragma unroll
    for (int co = 0; co < COUT; ++co) sout[co * n + i] = acc[co];
  }
}

__global__ __launch_bounds__(NT) void gin_chain_kernel(const float *__restrict__ x, const float *__restrict__ alpha,
                                                       GinArgs a, float *__restrict__ mixed,
                                                       double *__restrict__ partial, int D, int H, int W, int tilesD) {
  __shared__ float bufA[SZ_A];
  __shared__ float bufB[SZ_B];
  __shared__ float sw[324];   // L0 [2][1][27] @0, L1 [2][2][27] @54, L2 @162, L3 [1][2][27] @270
  __shared__ float ssh[8];    // shifts: L0 @0, L1 @2, L2 @4, L3 @6
  __shared__ float sred[16];

  const int tid = threadIdx.x;
  const int b = blockIdx.z / tilesD;
  const int d0 = (blockIdx.z % tilesD) * TD, h0 = blockIdx.y * TH, w0 = blockIdx.x * TW;
  const int64_t V = (int64_t)D * H * W;
  const float *xb = x + (int64_t)b * V;

  // weights of this sample: rows [b*cout, (b+1)*cout) of each layer's [cout*nb, cin, k,k,k] tensor (groups=nb)
  {
    const int cin[4] = {1, 2, 2, 2}, cout[4] = {2, 2, 2, 1}, off[4] = {0, 54, 162, 270};
    for (int l = 0; l < 4; ++l) {
      const int k = a.ksz[l], k3 = k * k * k, rows = cout[l] * cin[l];
      for (int i = tid; i < rows * 27; i += NT) {
        int r = i / 27, t = i % 27;
        float v;
        if (k == 3) v = a.ker[l][((int64_t)b * rows + r) * 27 + t];
        else v = (t == 13) ? a.ker[l][(int64_t)b * rows + r] : 0.f;
        (void)k3;
        sw[off[l] + i] = v;
      }
      if (tid < cout[l]) ssh[2 * l + tid] = a.shift[l][b * cout[l] + tid];
    }
  }
  for (int i = tid; i < SZ_A; i += NT) {
    int iw = i % R0W, ih = (i / R0W) % R0H, id = i / (R0W * R0H);
    int gd = d0 - HALO + id, gh = h0 - HALO + ih, gw = w0 - HALO + iw;
    bufA[i] = inside(gd, gh, gw, D, H, W) ? xb[((int64_t)gd * H + gh) * W + gw] : 0.f;
  }
  __syncthreads();
  gin_layer<1, 2, true>(bufA, bufB, sw + 0, ssh + 0, TD + 6, TH + 6, TW + 6, d0 - 3, h0 - 3, w0 - 3, D, H, W);
  __syncthreads();
  gin_layer<2, 2, true>(bufB, bufA, sw + 54, ssh + 2, TD + 4, TH + 4, TW + 4, d0 - 2, h0 - 2, w0 - 2, D, H, W);
  __syncthreads();
  gin_layer<2, 2, true>(bufA, bufB, sw + 162, ssh + 4, TD + 2, TH + 2, TW + 2, d0 - 1, h0 - 1, w0 - 1, D, H, W);
  __syncthreads();
  gin_layer<2, 1, false>(bufB, bufA, sw + 270, ssh + 6, TD, TH, TW, d0, h0, w0, D, H, W);
  __syncthreads();

  const float al = alpha[b];
  const float om = 1.0f - al;
  float s_in = 0.f, s_mx = 0.f;
  for (int i = tid; i < TD * TH * TW; i += NT) {
    int xw = i % TW, y = (i / TW) % TH, z = i / (TW * TH);
    int gd = d0 + z, gh = h0 + y, gw = w0 + xw;
    if (inside(gd, gh, gw, D, H, W)) {
      int64_t g = ((int64_t)gd * H + gh) * W + gw;
      float xi = xb[g];
      float t1 = al * bufA[i];
      float t2 = om * xi;
      float mx = t1 + t2;
      mixed[(int64_t)b * V + g] = mx;
      s_in += xi * xi;
      s_mx += mx * mx;
    }
  }
  float tin = block_sum(s_in, sred);
  float tmx = block_sum(s_mx, sred);
  if (tid == 0) {
    int bid = ((blockIdx.z % tilesD) * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    int per = tilesD * gridDim.y * gridDim.x;
    partial[((int64_t)b * per + bid) * 2 + 0] = (double)tin;
    partial[((int64_t)b * per + bid) * 2 + 1] = (double)tmx;
  }
}

__global__ void gin_norm_kernel(const double *__restrict__ partial, int per, float *__restrict__ norms) {
  __shared__ double sh[2][256];
  const int b = blockIdx.x;
  double s0 = 0.0, s1 = 0.0;
  for (int i = threadIdx.x; i < per; i += 256) {
    s0 += partial[((int64_t)b * per + i) * 2 + 0];
    s1 += partial[((int64_t)b * per + i) * 2 + 1];
  }
  sh[0][threadIdx.x] = s0;
  sh[1][threadIdx.x] = s1;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      sh[0][threadIdx.x] += sh[0][threadIdx.x + o];
      sh[1][threadIdx.x] += sh[1][threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    norms[2 * b + 0] = sqrtf((float)sh[0][0]);
    norms[2 * b + 1] = sqrtf((float)sh[1][0]);
  }
}

__global__ void gin_scale_kernel(float *__restrict__ out, const float *__restrict__ norms, int64_t V, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t b = i / V;
    float r = 1.0f / (norms[2 * b + 1] + 1e-5f);
    out[i] = (out[i] * r) * norms[2 * b + 0];
  }
}

}  // namespace

extern "C" size_t dgtta_gin_ws_bytes(int B, int D, int H, int W) {
  size_t per = (size_t)cdiv(D, TD) * cdiv(H, TH) * cdiv(W, TW);
  return align_up((size_t)B * per * 2 * sizeof(double), 256) + align_up((size_t)B * 2 * sizeof(float), 256);
}

extern "C" int dgtta_gin_chain_fwd(const float *x, const float *alpha, const int *h_ksz, const float *const *h_ker,
                                   const float *const *h_shift, float *out, void *ws, size_t ws_bytes, int B, int D,
                                   int H, int W, void *stream) {
  DG_REQUIRE(x && alpha && h_ksz && h_ker && h_shift && out && ws, DGTTA_ERR_BADARG, "gin_chain_fwd: null pointer");
  DG_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0, DGTTA_ERR_BADARG, "gin_chain_fwd: bad dims");
  DG_REQUIRE(ws_bytes >= dgtta_gin_ws_bytes(B, D, H, W), DGTTA_ERR_WORKSPACE, "gin_chain_fwd: workspace too small");
  GinArgs a;
  for (int l = 0; l < 4; ++l) {
    DG_REQUIRE(h_ksz[l] == 1 || h_ksz[l] == 3, DGTTA_ERR_UNSUPPORTED, "gin_chain_fwd: kernel size %d not in {1,3}",
               h_ksz[l]);
    DG_REQUIRE(h_ker[l] && h_shift[l], DGTTA_ERR_BADARG, "gin_chain_fwd: null layer pointer");
    a.ker[l] = h_ker[l];
    a.shift[l] = h_shift[l];
    a.ksz[l] = h_ksz[l];
  }
  hipStream_t st = (hipStream_t)stream;
  const int td = cdiv(D, TD), th = cdiv(H, TH), tw = cdiv(W, TW);
  DG_REQUIRE((int64_t)td * B <= 65535 && th <= 65535, DGTTA_ERR_UNSUPPORTED, "gin_chain_fwd: volume too large");
  const int per = td * th * tw;
  double *partial = (double *)ws;
  float *norms = (float *)((char *)ws + align_up((size_t)B * per * 2 * sizeof(double), 256));
  const int64_t V = (int64_t)D * H * W;
  hipLaunchKernelGGL(gin_chain_kernel, dim3(tw, th, td * B), dim3(NT), 0, st, x, alpha, a, out, partial, D, H, W, td);
  DG_CHECK_LAUNCH("gin_chain_kernel");
  hipLaunchKernelGGL(gin_norm_kernel, dim3(B), dim3(256), 0, st, partial, per, norms);
  DG_CHECK_LAUNCH("gin_norm_kernel");
  const int64_t total = (int64_t)B * V;
  int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(gin_scale_kernel, dim3(blocks), dim3(256), 0, st, out, norms, V, total);
  DG_CHECK_LAUNCH("gin_scale_kernel");
  return DGTTA_OK;
}
