// GIN random-convolution augmentation chain, fused (1->2->2->2->1 channels, k in {1,3}, LeakyReLU 0.01 on the
// first three layers), alpha-blend with the input and Frobenius re-normalisation.
// Replaces GINGroupConv.forward + GradlessGCReplayNonlinBlock.forward (dg_tta/gin.py:168-230, :59-122).
//
// Pass A (gin_chain_kernel): one workgroup per 8x8x32 output tile; the input tile (+4 halo, zero padded as the
//   reference's padding=k//2 does layer by layer) and all intermediate 2-channel activations live in LDS
//   (ping-pong), so HBM sees one read of x and one write of `mixed`.  k=1 layers take a pointwise path (same regions
//   and zero masking; the 3x3x3 form would add 26 exact zeros per tap).  Emits per-workgroup partial sums of x^2 and mixed^2.
// Pass B (gin_norm_kernel): fixed-order double sum -> ||x||_F, ||mixed||_F per sample.
// Pass C (gin_scale_kernel): out = mixed * (1/(||mixed||+1e-5)) * ||x||   (gin.py:228, same operation order).
// Algorithmic HBM bytes per voxel: 4 r + 4 w + 4 r + 4 w = 16 B.
#include "common.h"

namespace {

// Tile 8 x 8 x 32 outputs, 8 waves.  Every layer's output region shrinks by one voxel per side; rows are padded to a
// multiple of 4 floats so that a thread can own a strip of 4 consecutive outputs along W and fetch its 6 inputs per
// (channel, kd, kh) row with one 16-byte and one 8-byte LDS read (1.5 reads per output instead of 3 + 1 per weight).
// Weights: the per-sample kernels, expanded to 27 taps, sit in a small global table and are read with wave-uniform
// addresses (scalar loads, an SGPR operand per FMA) - the first version kept them in LDS and paid a read per FMA.
constexpr int TD = 8, TH = 8, TW = 32, HALO = 4, NT = 512;
constexpr int R0D = TD + 8, R0H = TH + 8, R0W = TW + 8;                      // input tile 16 x 16 x 40
constexpr int PW0 = 40, PW1 = 36, PW2 = 36, PW3 = 32;                        // row pitch of the outputs of layers 0..3
constexpr int SZ_A = 2 * (TD + 4) * (TH + 4) * PW1 + 8;                      // input (10240) / layer-1 output (10368) / result
constexpr int SZ_B = 2 * (TD + 6) * (TH + 6) * PW0 + 8;                      // layer-0 output (15680) / layer-2 output (7200)
constexpr int WTAB = 336;                                                     // 324 weights + 8 shifts (+ pad) per sample

struct GinArgs {
  const float *ker[4];
  const float *shift[4];
  int ksz[4];
};

__device__ __forceinline__ bool inside(int d, int h, int w, int D, int H, int W) {
  return (unsigned)d < (unsigned)D && (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W;
}

// weights of sample b: rows [b*cout, (b+1)*cout) of each layer's [cout*nb, cin, k,k,k] tensor (groups=nb); k=1 layers
// become 3x3x3 kernels with a single centre tap (the other taps add exact zeros).  Table layout per layer:
// [cin][27 taps][cout] (the two output channels of a tap adjacent = one packed-FMA operand); L0 @0, L1 @54, L2 @162,
// L3 @270, shifts @324.
__global__ void gin_prep_kernel(GinArgs a, float *__restrict__ wtab) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const int cin[4] = {1, 2, 2, 2}, cout[4] = {2, 2, 2, 1}, off[4] = {0, 54, 162, 270};
  float *wt = wtab + (int64_t)b * WTAB;
  for (int l = 0; l < 4; ++l) {
    const int k = a.ksz[l], rows = cout[l] * cin[l];
    for (int i = tid; i < rows * 27; i += blockDim.x) {
      const int r = i / 27, t = i % 27, co = r / cin[l], ci = r % cin[l];
      const float v = (k == 3) ? a.ker[l][((int64_t)b * rows + r) * 27 + t]
                               : ((t == 13) ? a.ker[l][(int64_t)b * rows + r] : 0.f);
      wt[off[l] + (ci * 27 + t) * cout[l] + co] = v;
    }
    if (tid < cout[l]) wt[324 + 2 * l + tid] = a.shift[l][b * cout[l] + tid];
  }
}

typedef __attribute__((ext_vector_type(2))) float f32x2_t;

// acc.{lo,hi} += w.{lo,hi} * v.lo   /   * v.hi   (v_pk_fma_f32 with the second operand's half broadcast by op_sel)
__device__ __forceinline__ void pk_fma_blo(f32x2_t &acc, const f32x2_t &w, const f32x2_t &v) {
  asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(w), "v"(v));
}
__device__ __forceinline__ void pk_fma_bhi(f32x2_t &acc, const f32x2_t &w, const f32x2_t &v) {
  asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc) : "v"(w), "v"(v));
}

// One layer on LDS regions.  in: [CIN][OD+2][OH+2][IPW], out: [COUT][OD][OH][OPW]; (gd0,gh0,gw0) = global coordinate of
// the output region's origin.  Outputs outside the volume are forced to 0 (= the reference's zero padding, layer by
// layer).  The last strip of a row may compute up to 3 columns beyond OW from whatever follows the valid inputs: those
// columns land in the row padding and are never read by a valid output of the next layer.
// COUT == 2: the two output channels of a voxel are one packed accumulator (v_pk_fma_f32: weight pair x broadcast input).
template <int CIN, int COUT, bool ACT, int OD, int OH, int OW, int IPW, int OPW>
__device__ __forceinline__ void gin_layer(const float *sin, float *sout, const float *__restrict__ w /*[CIN][27][COUT]*/,
                                          const float *__restrict__ sh, int gd0, int gh0, int gw0, int D, int H, int W) {
  constexpr int ID = OD + 2, IH = OH + 2, NSX = (OW + 3) / 4, N = OD * OH * NSX;
  // COUT == 2: all weight pairs of the layer in registers (wave-uniform values; 54 pairs for a 2 -> 2 layer)
  f32x2_t wreg[COUT == 2 ? CIN * 27 : 1];
  if (COUT == 2) {
#pragma unroll
    for (int i = 0; i < CIN * 27; ++i) wreg[i] = f32x2_t{w[2 * i], w[2 * i + 1]};
  }
  for (int s = threadIdx.x; s < N; s += NT) {
    const int xs = (s % NSX) * 4, y = (s / NSX) % OH, z = s / (NSX * OH);
    float o[COUT][4];
    if (COUT == 2) {
      f32x2_t acc[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = f32x2_t{0.f, 0.f};
#pragma unroll
      for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
        for (int kd = 0; kd < 3; ++kd)
#pragma unroll
          for (int kh = 0; kh < 3; ++kh) {
            const float *p = sin + ((ci * ID + z + kd) * IH + y + kh) * IPW + xs;
            const float4 q0 = *reinterpret_cast<const float4 *>(p);
            const float2 q1 = *reinterpret_cast<const float2 *>(p + 4);
            const f32x2_t vp[3] = {f32x2_t{q0.x, q0.y}, f32x2_t{q0.z, q0.w}, f32x2_t{q1.x, q1.y}};
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
              const f32x2_t wv = wreg[ci * 27 + kd * 9 + kh * 3 + kw];
              // issue order pinned: the 4 strip accumulators are independent, so consecutive packed FMAs never wait on
              // each other (left to itself the scheduler builds one 54-long dependent chain per accumulator)
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                if ((j + kw) & 1) pk_fma_bhi(acc[j], wv, vp[(j + kw) >> 1]);
                else pk_fma_blo(acc[j], wv, vp[(j + kw) >> 1]);
              }
            }
          }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        o[0][j] = acc[j][0];
        o[COUT - 1][j] = acc[j][1];
      }
    } else {
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
        for (int kd = 0; kd < 3; ++kd)
#pragma unroll
          for (int kh = 0; kh < 3; ++kh) {
            const float *p = sin + ((ci * ID + z + kd) * IH + y + kh) * IPW + xs;
            const float4 q0 = *reinterpret_cast<const float4 *>(p);
            const float2 q1 = *reinterpret_cast<const float2 *>(p + 4);
            const float v[6] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y};
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
              const float wv = w[ci * 27 + kd * 9 + kh * 3 + kw];
#pragma unroll
              for (int j = 0; j < 4; ++j) acc[j] = __builtin_fmaf(wv, v[j + kw], acc[j]);
            }
          }
#pragma unroll
      for (int j = 0; j < 4; ++j) o[0][j] = acc[j];
    }
    const bool zy_in = (unsigned)(gd0 + z) < (unsigned)D && (unsigned)(gh0 + y) < (unsigned)H;
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float v = o[co][j] + sh[co];
        const bool in = zy_in && (unsigned)(gw0 + xs + j) < (unsigned)W;
        o[co][j] = in ? (ACT ? lrelu(v, 0.01f) : v) : 0.f;
      }
      *reinterpret_cast<float4 *>(sout + ((co * OD + z) * OH + y) * OPW + xs) =
          make_float4(o[co][0], o[co][1], o[co][2], o[co][3]);
    }
  }
}

// k = 1 layer: out[co] = act(sum_ci w[co][ci] * in[ci][centre] + shift), same regions and masking as gin_layer (which
// would spend 26 of its 27 taps on exact zeros).  w: the centre taps of the expanded table ([CIN][27][COUT] layout).
template <int CIN, int COUT, bool ACT, int OD, int OH, int OW, int IPW, int OPW>
__device__ __forceinline__ void gin_layer_k1(const float *sin, float *sout, const float *__restrict__ w,
                                             const float *__restrict__ sh, int gd0, int gh0, int gw0, int D, int H, int W) {
  constexpr int ID = OD + 2, IH = OH + 2, NSX = (OW + 3) / 4, N = OD * OH * NSX;
  float wc[COUT][CIN];
#pragma unroll
  for (int co = 0; co < COUT; ++co)
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci) wc[co][ci] = w[(ci * 27 + 13) * COUT + co];
  for (int s = threadIdx.x; s < N; s += NT) {
    const int xs = (s % NSX) * 4, y = (s / NSX) % OH, z = s / (NSX * OH);
    float v[CIN][4];
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci) {
      const float *p = sin + ((ci * ID + z + 1) * IH + y + 1) * IPW + xs + 1;      // centre tap: unaligned by one float
#pragma unroll
      for (int j = 0; j < 4; ++j) v[ci][j] = p[j];
    }
    const bool zy_in = (unsigned)(gd0 + z) < (unsigned)D && (unsigned)(gh0 + y) < (unsigned)H;
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
      float o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float a = 0.f;
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) a = __builtin_fmaf(wc[co][ci], v[ci][j], a);
        a += sh[co];
        const bool in = zy_in && (unsigned)(gw0 + xs + j) < (unsigned)W;
        o[j] = in ? (ACT ? lrelu(a, 0.01f) : a) : 0.f;
      }
      *reinterpret_cast<float4 *>(sout + ((co * OD + z) * OH + y) * OPW + xs) = make_float4(o[0], o[1], o[2], o[3]);
    }
  }
}

// Persistent over tiles: the input tile of the NEXT tile is fetched into registers while the current one is computed
// (one workgroup per CU holds 102 KB of LDS, so nothing else would hide the load latency).
__global__ __launch_bounds__(NT) void gin_chain_kernel(const float *__restrict__ x, const float *__restrict__ alpha,
                                                       const float *__restrict__ wtab, float *__restrict__ mixed,
                                                       double *__restrict__ partial, int D, int H, int W, int tilesD,
                                                       int tilesH, int tilesW, int ntiles, int k0, int k1, int k2,
                                                       int k3) {
  __shared__ __attribute__((aligned(16))) float bufA[SZ_A];
  __shared__ __attribute__((aligned(16))) float bufB[SZ_B];
  __shared__ float sred[16];
  constexpr int NPRE = R0D * R0H * R0W / NT;       // 20 input elements per thread
  static_assert(R0D * R0H * R0W % NT == 0, "input tile must divide over the threads");

  const int tid = threadIdx.x;
  const int64_t V = (int64_t)D * H * W;
  const int per = tilesD * tilesH * tilesW;
  float pre[NPRE];
  auto fetch = [&](int tile) {
    const int b = tile / per, bid = tile % per;
    const int d0 = (bid / (tilesH * tilesW)) * TD, h0 = ((bid / tilesW) % tilesH) * TH, w0 = (bid % tilesW) * TW;
    const float *xb = x + (int64_t)b * V;
#pragma unroll
    for (int k = 0; k < NPRE; ++k) {
      const int i = tid + k * NT;
      const int iw = i % R0W, ih = (i / R0W) % R0H, id = i / (R0W * R0H);
      const int gd = d0 - HALO + id, gh = h0 - HALO + ih, gw = w0 - HALO + iw;
      const bool in = inside(gd, gh, gw, D, H, W);
      const float v = xb[in ? ((int64_t)gd * H + gh) * W + gw : 0];
      pre[k] = in ? v : 0.f;
    }
  };
  int tile = blockIdx.x;
  if (tile < ntiles) fetch(tile);
  for (; tile < ntiles; tile += gridDim.x) {
    const int b = tile / per, bid = tile % per;
    const int d0 = (bid / (tilesH * tilesW)) * TD, h0 = ((bid / tilesW) % tilesH) * TH, w0 = (bid % tilesW) * TW;
    const float *xb = x + (int64_t)b * V;
    const float *wt = wtab + (int64_t)b * WTAB;
#pragma unroll
    for (int k = 0; k < NPRE; ++k) bufA[tid + k * NT] = pre[k];
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) fetch(tile + gridDim.x);
    if (k0 == 1) gin_layer_k1<1, 2, true, TD + 6, TH + 6, TW + 6, R0W, PW0>(bufA, bufB, wt + 0, wt + 324, d0 - 3, h0 - 3, w0 - 3, D, H, W);
    else gin_layer<1, 2, true, TD + 6, TH + 6, TW + 6, R0W, PW0>(bufA, bufB, wt + 0, wt + 324, d0 - 3, h0 - 3, w0 - 3, D, H, W);
    __syncthreads();
    if (k1 == 1) gin_layer_k1<2, 2, true, TD + 4, TH + 4, TW + 4, PW0, PW1>(bufB, bufA, wt + 54, wt + 326, d0 - 2, h0 - 2, w0 - 2, D, H, W);
    else gin_layer<2, 2, true, TD + 4, TH + 4, TW + 4, PW0, PW1>(bufB, bufA, wt + 54, wt + 326, d0 - 2, h0 - 2, w0 - 2, D, H, W);
    __syncthreads();
    if (k2 == 1) gin_layer_k1<2, 2, true, TD + 2, TH + 2, TW + 2, PW1, PW2>(bufA, bufB, wt + 162, wt + 328, d0 - 1, h0 - 1, w0 - 1, D, H, W);
    else gin_layer<2, 2, true, TD + 2, TH + 2, TW + 2, PW1, PW2>(bufA, bufB, wt + 162, wt + 328, d0 - 1, h0 - 1, w0 - 1, D, H, W);
    __syncthreads();
    if (k3 == 1) gin_layer_k1<2, 1, false, TD, TH, TW, PW2, PW3>(bufB, bufA, wt + 270, wt + 330, d0, h0, w0, D, H, W);
    else gin_layer<2, 1, false, TD, TH, TW, PW2, PW3>(bufB, bufA, wt + 270, wt + 330, d0, h0, w0, D, H, W);
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
    float tmx = block_sum(s_mx, sred);      // (block_sum's barriers also separate the reads of bufA from the next tile's writes)
    if (tid == 0) {
      partial[((int64_t)b * per + bid) * 2 + 0] = (double)tin;
      partial[((int64_t)b * per + bid) * 2 + 1] = (double)tmx;
    }
    __syncthreads();
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
  if (B <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;      // a size query of an empty problem (the launchers reject it with DGTTA_ERR_BADARG)
  size_t per = (size_t)cdiv(D, TD) * cdiv(H, TH) * cdiv(W, TW);
  return align_up((size_t)B * per * 2 * sizeof(double), 256) + align_up((size_t)B * 2 * sizeof(float), 256) +
         align_up((size_t)B * WTAB * sizeof(float), 256);
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
  DG_REQUIRE((int64_t)td * th * tw * B < (1ll << 31), DGTTA_ERR_UNSUPPORTED, "gin_chain_fwd: volume too large");
  const int per = td * th * tw;
  double *partial = (double *)ws;
  float *norms = (float *)((char *)ws + align_up((size_t)B * per * 2 * sizeof(double), 256));
  float *wtab = (float *)((char *)norms + align_up((size_t)B * 2 * sizeof(float), 256));
  const int64_t V = (int64_t)D * H * W;
  hipLaunchKernelGGL(gin_prep_kernel, dim3(B), dim3(128), 0, st, a, wtab);
  DG_CHECK_LAUNCH("gin_prep_kernel");
  const int ntiles = B * per;
  static const int ncu = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n > 0 ? n : 256;
  }();
  hipLaunchKernelGGL(gin_chain_kernel, dim3(ntiles < ncu ? ntiles : ncu), dim3(NT), 0, st, x, alpha, wtab, out, partial, D,
                     H, W, td, th, tw, ntiles, a.ksz[0], a.ksz[1], a.ksz[2], a.ksz[3]);
  DG_CHECK_LAUNCH("gin_chain_kernel");
  hipLaunchKernelGGL(gin_norm_kernel, dim3(B), dim3(256), 0, st, partial, per, norms);
  DG_CHECK_LAUNCH("gin_norm_kernel");
  const int64_t total = (int64_t)B * V;
  int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(gin_scale_kernel, dim3(blocks), dim3(256), 0, st, out, norms, V, total);
  DG_CHECK_LAUNCH("gin_scale_kernel");
  return DGTTA_OK;
}
