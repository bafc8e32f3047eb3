// MIND3D 12-channel self-similarity descriptor, fused (HBM-bound).
// Replaces MIND3D.forward / smooth / filter1D of the reference (dg_tta/mind.py:142-164, :27-43, :5-24).
//
// Pass A (mind_ssd_kernel): one workgroup per 8x8x32 output tile.  The image tile (+3 halo, replicate
//   = clamped coordinates) is staged once in LDS; the squared edge responses
//   q_c = (p[v+s1_c]-p[v+s2_c] + rw*noise_c)^2 are formed on the tile +2 halo from the six face neighbours and
//   smoothed by the separable 5-tap Gaussian D (register window) -> H -> W (LDS), 4 channels per pass; the 12
//   smoothed values of a voxel stay in registers.
//   Writes m_c = ssd_c - min_c (fp32, voxel-major [B][V][12]) to the workspace and a per-workgroup
//   partial sum of var = mean_c m_c.
// Pass B (mind_reduce_kernel): fixed-order sum of the partials in double -> global mean of var.
// Pass C (mind_finish_kernel): var clamp to [1e-3, 1e3] x global mean, out = exp(-m/var), written
//   NCDHW fp32 or NDHWC fp32/bf16.
// Algorithmic HBM bytes per voxel: img 4 + noise 48 + ws 48 w + 48 r + out 48 (fp32) = 196 B.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int TD = 8, TH = 8, TW = 32;
constexpr int NT = 512;                                 // 8 waves
constexpr int VPT = TD * TH * TW / NT;                 // 4 voxels per thread (consecutive along W)
static_assert(VPT == 4 && TW == 32 && TD * TH * (TW / 4) == NT, "stage-3 thread map: (d, h, w/4)");

__device__ __forceinline__ int clampi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }

struct Taps { float g[7]; };      // 2 R + 1 Gaussian taps (R <= 3, i.e. sigma <= 2)

// The 12 shift pairs of mshift1/mshift2 (mind.py:112-135) only ever touch the SIX face neighbours of a voxel (at distance
// delta): e_c = img[v + delta s1_c] - img[v + delta s2_c] with (d,h,w) offsets
//   s1: c0 (0,0,-1) c1,c2 (0,-1,0) c3,c4 (0,0,1) c5..c7 (1,0,0) c8..c11 (0,1,0)
//   s2: c0,c1,c3,c8 (-1,0,0)  c2,c5,c9 (0,0,-1)  c4,c6 (0,-1,0)  c7,c10 (0,0,1)  c11 (1,0,0)
// so one fetch of the neighbours (dp = d+delta, dm = d-delta, hp, hm, wp, wm) serves every channel.
template <int C>
__device__ __forceinline__ float edge(float dp, float dm, float hp, float hm, float wp, float wm) {
  constexpr int S1[12] = {5, 3, 3, 4, 4, 0, 0, 0, 2, 2, 2, 2};      // index into {dp, dm, hp, hm, wp, wm}
  constexpr int S2[12] = {1, 1, 5, 1, 3, 5, 3, 4, 1, 5, 4, 0};
  const float n[6] = {dp, dm, hp, hm, wp, wm};
  return n[S1[C]] - n[S2[C]];
}

__device__ __forceinline__ float edge_sel(int c, float dp, float dm, float hp, float hm, float wp, float wm) {
  switch (c) {
    case 0: return edge<0>(dp, dm, hp, hm, wp, wm);
    case 1: return edge<1>(dp, dm, hp, hm, wp, wm);
    case 2: return edge<2>(dp, dm, hp, hm, wp, wm);
    case 3: return edge<3>(dp, dm, hp, hm, wp, wm);
    case 4: return edge<4>(dp, dm, hp, hm, wp, wm);
    case 5: return edge<5>(dp, dm, hp, hm, wp, wm);
    case 6: return edge<6>(dp, dm, hp, hm, wp, wm);
    case 7: return edge<7>(dp, dm, hp, hm, wp, wm);
    case 8: return edge<8>(dp, dm, hp, hm, wp, wm);
    case 9: return edge<9>(dp, dm, hp, hm, wp, wm);
    case 10: return edge<10>(dp, dm, hp, hm, wp, wm);
    default: return edge<11>(dp, dm, hp, hm, wp, wm);
  }
}

// Geometry for neighbour distance DELTA (mind.py:137: ReplicationPad3d(delta), dilation delta) and filter radius R
// (mind.py:30-31: N = 2 ceil(1.5 sigma) + 1 taps), CG channels per pass through the LDS filter pipeline.
template <int DELTA, int R, int CG>
struct MG {
  static constexpr int HL = R + DELTA;                                          // image halo
  static constexpr int ID = TD + 2 * HL, IH = TH + 2 * HL, IW = TW + 2 * HL;    // image tile
  static constexpr int QD = TD + 2 * R, QH = TH + 2 * R, QW = TW + 2 * R;       // q tile (halo R)
  static constexpr int QWP = (QW + 3) / 4 * 4;                                  // row pitch of the filter buffers
  static constexpr int NTAP = 2 * R + 1, NG = 12 / CG;
  static constexpr int NV3 = (4 + 2 * R + 3) / 4;                               // 16-byte reads per row in stage 3
  static constexpr size_t LDS_BYTES = (size_t)(ID * IH * IW + CG * TD * QH * QWP + CG * TD * TH * QWP + 8 + 16) * 4;
};

// Pass A.  One workgroup per 8 x 8 x 32 output tile, channels in NG groups of CG:
//   stage 1  thread = one (h', w') column of the q tile (halo R): walks the depths, forms q_c = (e_c + rw n_c)^2 for the
//            group's channels from the six neighbours and runs the D filter on a (2R+1)-deep register window per channel
//            -> r1[c][d][h'][w'] in LDS (no q tile in LDS at all);
//   stage 2  H filter, 4 adjacent w' columns per thread (16-byte LDS accesses): r1 -> r2[c][d][h][w'];
//   stage 3  W filter: thread = 4 consecutive output voxels of a row, 16-byte reads per channel -> registers.
// 2 barriers per group instead of 5 per channel, ~1/3 of the LDS instructions of the first version.
template <int DELTA, int R, int CG>
__global__ __launch_bounds__(NT) void mind_ssd_kernel(const float *__restrict__ img, const float *__restrict__ noise,
                                                      float rw, float *__restrict__ mws, double *__restrict__ partial,
                                                      int D, int H, int W, int tilesD, Taps taps) {
  typedef MG<DELTA, R, CG> G;
  constexpr int ID = G::ID, IH = G::IH, IW = G::IW, QD = G::QD, QH = G::QH, QW = G::QW, QWP = G::QWP, NTAP = G::NTAP,
                NG = G::NG, HL = G::HL;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *simg = smem;                                   // [ID][IH][IW]
  float *sr1 = simg + (ID * IH * IW + 3) / 4 * 4;       // [CG][TD][QH][QWP]
  float *sr2 = sr1 + CG * TD * QH * QWP;                // [CG][TD][TH][QWP] (+ pad: stage 3 reads past the last row)
  float *sred = sr2 + CG * TD * TH * QWP + 8;

  const int tid = threadIdx.x;
  // Workgroups are dispatched round robin over the 8 XCDs; tiles that are neighbours in H or W (their q tiles overlap by
  // the +2 halo of the 48-byte noise rows, 2.5x the tile) would sit on different XCDs and each L2 would fetch the overlap
  // from HBM.  Give every XCD a contiguous run of tiles (W fastest, then H, then D) instead.
  int tlin = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
  const int ntile = gridDim.x * gridDim.y * gridDim.z;
  if ((ntile & 7) == 0) tlin = (tlin & 7) * (ntile >> 3) + (tlin >> 3);
  const int bx = tlin % gridDim.x, by = (tlin / gridDim.x) % gridDim.y, bz = tlin / (gridDim.x * gridDim.y);
  const int b = bz / tilesD;
  const int d0 = (bz % tilesD) * TD, h0 = by * TH, w0 = bx * TW;
  const int64_t V = (int64_t)D * H * W;
  const float *imgb = img + (int64_t)b * V;

  {   // image tile: all loads of a thread issued back to back (a load-store loop pays the HBM latency once per element)
    constexpr int NI = (ID * IH * IW + NT - 1) / NT;
    float pre[NI];
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      const int i = tid + k * NT < ID * IH * IW ? tid + k * NT : 0;
      const int iw = i % IW, ih = (i / IW) % IH, id = i / (IW * IH);
      const int gd = clampi(d0 - HL + id, 0, D - 1), gh = clampi(h0 - HL + ih, 0, H - 1), gw = clampi(w0 - HL + iw, 0, W - 1);
      pre[k] = imgb[((int64_t)gd * H + gh) * W + gw];
    }
#pragma unroll
    for (int k = 0; k < NI; ++k)
      if (tid + k * NT < ID * IH * IW) simg[tid + k * NT] = pre[k];
  }

  // stage-1 role: column (qh, qw) of the q tile; replicate padding = clamped global coordinates (mind.py:143, :13-14)
  const bool col_on = tid < QH * QW;
  const int qw = tid % QW, qh = (tid / QW) % QH;
  const int gh = clampi(h0 - R + qh, 0, H - 1), gw = clampi(w0 - R + qw, 0, W - 1);
  const int ih_c = gh - (h0 - HL), iw_c = gw - (w0 - HL);
  const int ih_p = clampi(gh + DELTA, 0, H - 1) - (h0 - HL), ih_m = clampi(gh - DELTA, 0, H - 1) - (h0 - HL);
  const int iw_p = clampi(gw + DELTA, 0, W - 1) - (w0 - HL), iw_m = clampi(gw - DELTA, 0, W - 1) - (w0 - HL);
  // stage-3 role: 4 consecutive output voxels
  const int od = tid / (TH * (TW / 4)), oh = (tid / (TW / 4)) % TH, ow4 = (tid % (TW / 4)) * 4;

  float ssd[VPT][12];
  // noise of the whole column for one channel group, fetched a full group ahead: the loads of group g+1 are issued while
  // group g runs
  float nbuf[QD][CG];
  const float *nzb = noise + (int64_t)b * 12 * V;
  int gidx[QD];
#pragma unroll
  for (int qd = 0; qd < QD; ++qd) gidx[qd] = (clampi(d0 - R + qd, 0, D - 1) * H + gh) * W + gw;
  if (col_on) {
#pragma unroll
    for (int qd = 0; qd < QD; ++qd)
#pragma unroll
      for (int c = 0; c < CG; ++c) nbuf[qd][c] = nzb[(int64_t)c * V + gidx[qd]];
  }
  __syncthreads();      // simg ready

#pragma unroll
  for (int g = 0; g < NG; ++g) {
    // ---- stage 1
    if (col_on) {
      float win[CG][NTAP];
      float nv[CG];
#pragma unroll
      for (int qd = 0; qd < QD; ++qd) {
#pragma unroll
        for (int c = 0; c < CG; ++c) nv[c] = nbuf[qd][c];
        if (g + 1 < NG) {
#pragma unroll
          for (int c = 0; c < CG; ++c) nbuf[qd][c] = nzb[(int64_t)((g + 1) * CG + c) * V + gidx[qd]];
        }
        const int gd = clampi(d0 - R + qd, 0, D - 1);
        const int id_c = gd - (d0 - HL), id_p = clampi(gd + DELTA, 0, D - 1) - (d0 - HL),
                  id_m = clampi(gd - DELTA, 0, D - 1) - (d0 - HL);
        const float dp = simg[(id_p * IH + ih_c) * IW + iw_c], dm = simg[(id_m * IH + ih_c) * IW + iw_c];
        const float hp = simg[(id_c * IH + ih_p) * IW + iw_c], hm = simg[(id_c * IH + ih_m) * IW + iw_c];
        const float wp = simg[(id_c * IH + ih_c) * IW + iw_p], wm = simg[(id_c * IH + ih_c) * IW + iw_m];
        float e[CG];
#pragma unroll
        for (int c = 0; c < CG; ++c) e[c] = edge_sel(g * CG + c, dp, dm, hp, hm, wp, wm);     // constant after unrolling
#pragma unroll
        for (int c = 0; c < CG; ++c) {
          const float ev = e[c] + rw * nv[c];
          win[c][qd % NTAP] = ev * ev;
        }
        if (qd >= NTAP - 1) {     // D filter output d = qd - 2R: taps over q[d .. d+2R]
          const int d = qd - (NTAP - 1);
#pragma unroll
          for (int c = 0; c < CG; ++c) {
            float acc = taps.g[0] * win[c][d % NTAP];
#pragma unroll
            for (int t = 1; t < NTAP; ++t) acc += taps.g[t] * win[c][(d + t) % NTAP];
            sr1[((c * TD + d) * QH + qh) * QWP + qw] = acc;
          }
        }
      }
    }
    if constexpr (QH * QW > NT) {
      // R = 3: the q tile has 14 x 38 = 532 columns for 512 threads; the last ones are walked by the first threads in
      // a second, unpipelined pass (noise loaded where it is used)
      const int col = tid + NT;
      if (col < QH * QW) {
        const int qw2 = col % QW, qh2 = col / QW;
        const int gh2 = clampi(h0 - R + qh2, 0, H - 1), gw2 = clampi(w0 - R + qw2, 0, W - 1);
        const int jh_c = gh2 - (h0 - HL), jw_c = gw2 - (w0 - HL);
        const int jh_p = clampi(gh2 + DELTA, 0, H - 1) - (h0 - HL), jh_m = clampi(gh2 - DELTA, 0, H - 1) - (h0 - HL);
        const int jw_p = clampi(gw2 + DELTA, 0, W - 1) - (w0 - HL), jw_m = clampi(gw2 - DELTA, 0, W - 1) - (w0 - HL);
        float win[CG][NTAP];
#pragma unroll
        for (int qd = 0; qd < QD; ++qd) {
          const int gd = clampi(d0 - R + qd, 0, D - 1);
          const int64_t gi = ((int64_t)gd * H + gh2) * W + gw2;
          const int id_c = gd - (d0 - HL), id_p = clampi(gd + DELTA, 0, D - 1) - (d0 - HL),
                    id_m = clampi(gd - DELTA, 0, D - 1) - (d0 - HL);
          const float dp = simg[(id_p * IH + jh_c) * IW + jw_c], dm = simg[(id_m * IH + jh_c) * IW + jw_c];
          const float hp = simg[(id_c * IH + jh_p) * IW + jw_c], hm = simg[(id_c * IH + jh_m) * IW + jw_c];
          const float wp = simg[(id_c * IH + jh_c) * IW + jw_p], wm = simg[(id_c * IH + jh_c) * IW + jw_m];
#pragma unroll
          for (int c = 0; c < CG; ++c) {
            const float ev = edge_sel(g * CG + c, dp, dm, hp, hm, wp, wm) + rw * nzb[(int64_t)(g * CG + c) * V + gi];
            win[c][qd % NTAP] = ev * ev;
          }
          if (qd >= NTAP - 1) {
            const int d = qd - (NTAP - 1);
#pragma unroll
            for (int c = 0; c < CG; ++c) {
              float acc = taps.g[0] * win[c][d % NTAP];
#pragma unroll
              for (int t = 1; t < NTAP; ++t) acc += taps.g[t] * win[c][(d + t) % NTAP];
              sr1[((c * TD + d) * QH + qh2) * QWP + qw2] = acc;
            }
          }
        }
      }
    }
    __syncthreads();
    // ---- stage 2: H filter, r2[c][d][h][w'] = sum_t g[t] r1[c][d][h+t][w'], 4 columns per thread
    for (int i = tid; i < CG * TD * (QWP / 4); i += NT) {
      const int w4 = (i % (QWP / 4)) * 4, d = (i / (QWP / 4)) % TD, c = i / ((QWP / 4) * TD);
      const float *r = sr1 + ((c * TD + d) * QH) * QWP + w4;
      float4 rowv[QH];
#pragma unroll
      for (int k = 0; k < QH; ++k) rowv[k] = *reinterpret_cast<const float4 *>(r + k * QWP);
#pragma unroll
      for (int h = 0; h < TH; ++h) {
        float4 acc;
        acc.x = taps.g[0] * rowv[h].x; acc.y = taps.g[0] * rowv[h].y; acc.z = taps.g[0] * rowv[h].z; acc.w = taps.g[0] * rowv[h].w;
#pragma unroll
        for (int t = 1; t < NTAP; ++t) {
          acc.x += taps.g[t] * rowv[h + t].x; acc.y += taps.g[t] * rowv[h + t].y;
          acc.z += taps.g[t] * rowv[h + t].z; acc.w += taps.g[t] * rowv[h + t].w;
        }
        *reinterpret_cast<float4 *>(sr2 + ((c * TD + d) * TH + h) * QWP + w4) = acc;
      }
    }
    __syncthreads();
    // ---- stage 3: W filter into registers
#pragma unroll
    for (int c = 0; c < CG; ++c) {
      const float *r = sr2 + ((c * TD + od) * TH + oh) * QWP + ow4;
      float v[4 * G::NV3];
#pragma unroll
      for (int q = 0; q < G::NV3; ++q) {
        const float4 a = *reinterpret_cast<const float4 *>(r + 4 * q);
        v[4 * q] = a.x; v[4 * q + 1] = a.y; v[4 * q + 2] = a.z; v[4 * q + 3] = a.w;
      }
#pragma unroll
      for (int k = 0; k < VPT; ++k) {
        float acc = taps.g[0] * v[k];
#pragma unroll
        for (int t = 1; t < NTAP; ++t) acc += taps.g[t] * v[k + t];
        ssd[k][g * CG + c] = acc;
      }
    }
    // (the next group's stage 1 writes sr1 only; sr2 is rewritten after the barrier that follows it)
  }

  float vsum = 0.f;
#pragma unroll
  for (int k = 0; k < VPT; ++k) {
    const int w = w0 + ow4 + k, h = h0 + oh, d = d0 + od;
    if (d < D && h < H && w < W) {
      float mn = ssd[k][0];
#pragma unroll
      for (int c = 1; c < 12; ++c) mn = fminf(mn, ssd[k][c]);
      float s = 0.f;
      float m[12];
#pragma unroll
      for (int c = 0; c < 12; ++c) {
        m[c] = ssd[k][c] - mn;
        s += m[c];
      }
      vsum += s / 12.0f;
      float4 *dst = reinterpret_cast<float4 *>(mws + ((int64_t)b * V + ((int64_t)d * H + h) * W + w) * 12);
      dst[0] = make_float4(m[0], m[1], m[2], m[3]);
      dst[1] = make_float4(m[4], m[5], m[6], m[7]);
      dst[2] = make_float4(m[8], m[9], m[10], m[11]);
    }
  }
  float tot = block_sum(vsum, sred);
  if (tid == 0) partial[tlin] = (double)tot;
}

template <int DELTA, int R, int CG>
int launch_mind_ssd(const float *img, const float *noise, float rw, float *mws, double *partial, int D, int H, int W, int td,
                    dim3 grid, const Taps &taps, hipStream_t st) {
  typedef MG<DELTA, R, CG> G;
  static_assert(G::LDS_BYTES <= 163840, "MIND tile does not fit the LDS");
  static DynLdsOnce once;
  DG_REQUIRE(ensure_dyn_lds(once, (const void *)mind_ssd_kernel<DELTA, R, CG>, (int)G::LDS_BYTES) == hipSuccess,
             DGTTA_ERR_LAUNCH, "mind3d: cannot raise the dynamic LDS limit to %zu", (size_t)G::LDS_BYTES);
  hipLaunchKernelGGL((mind_ssd_kernel<DELTA, R, CG>), grid, dim3(NT), G::LDS_BYTES, st, img, noise, rw, mws, partial, D, H, W,
                     td, taps);
  DG_CHECK_LAUNCH("mind_ssd_kernel");
  return DGTTA_OK;
}

__global__ void mind_reduce_kernel(const double *__restrict__ partial, int n, double inv_count, float *gmean) {
  __shared__ double sh[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) gmean[0] = (float)(sh[0] * inv_count);
}

template <typename TO, bool NDHWC>
__global__ void mind_finish_kernel(const float *__restrict__ mws, const float *__restrict__ gmean, TO *__restrict__ out,
                                   int ldc, int64_t V, int64_t total) {
  const float gm = gmean[0];
  const float lo = gm * 0.001f, hi = gm * 1000.0f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 *src = reinterpret_cast<const float4 *>(mws + i * 12);
    float4 a = src[0], bq = src[1], cq = src[2];
    float m[12] = {a.x, a.y, a.z, a.w, bq.x, bq.y, bq.z, bq.w, cq.x, cq.y, cq.z, cq.w};
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 12; ++c) s += m[c];
    float var = s / 12.0f;
    var = fminf(fmaxf(var, lo), hi);
    if (NDHWC) {
      TO *o = out + i * ldc;
#pragma unroll
      for (int c = 0; c < 12; ++c) st_f<TO>(o + c, expf(-(m[c] / var)));
      for (int c = 12; c < ldc; ++c) st_f<TO>(o + c, 0.f);
    } else {
      int64_t b = i / V, v = i % V;
      TO *o = out + b * 12 * V + v;
#pragma unroll
      for (int c = 0; c < 12; ++c) st_f<TO>(o + (int64_t)c * V, expf(-(m[c] / var)));
    }
  }
}

void tile_counts(int D, int H, int W, int &td, int &th, int &tw) {
  td = cdiv(D, TD);
  th = cdiv(H, TH);
  tw = cdiv(W, TW);
}

}  // namespace

extern "C" size_t dgtta_mind3d_ws_bytes(int B, int D, int H, int W) {
  if (B <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;      // a size query of an empty problem (the launchers reject it with DGTTA_ERR_BADARG)
  int td, th, tw;
  tile_counts(D, H, W, td, th, tw);
  size_t m = align_up((size_t)B * D * H * W * 12 * sizeof(float), 256);
  size_t p = align_up((size_t)B * td * th * tw * sizeof(double), 256);
  return m + p + 256;
}

extern "C" int dgtta_mind3d_fwd(const float *img, const float *noise, float rw, int delta, const float *h_taps, int ntaps,
                                void *out, int out_ndhwc, int out_ldc, int out_dtype, void *ws, size_t ws_bytes, int B,
                                int D, int H, int W, void *stream) {
  DG_REQUIRE(img && noise && out && ws, DGTTA_ERR_BADARG, "mind3d_fwd: null pointer");
  DG_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0, DGTTA_ERR_BADARG, "mind3d_fwd: bad dims %d %d %d %d", B, D, H, W);
  DG_REQUIRE(h_taps && (ntaps == 3 || ntaps == 5 || ntaps == 7), DGTTA_ERR_UNSUPPORTED,
             "mind3d_fwd: %d filter taps (sigma up to 2, i.e. 3 / 5 / 7 taps, are built)", ntaps);
  DG_REQUIRE(delta == 1 || delta == 2, DGTTA_ERR_UNSUPPORTED, "mind3d_fwd: delta %d (1 and 2 are built)", delta);
  DG_REQUIRE(D <= 1024 && H <= 1024 && W <= 1024 && (int64_t)D * H * W < (1ll << 31), DGTTA_ERR_UNSUPPORTED,
             "mind3d_fwd: each dim must be <= 1024 (got %d %d %d)", D, H, W);
  DG_REQUIRE(ws_bytes >= dgtta_mind3d_ws_bytes(B, D, H, W), DGTTA_ERR_WORKSPACE, "mind3d_fwd: workspace too small");
  DG_REQUIRE(out_ndhwc ? (out_ldc >= 12) : (out_dtype == DGTTA_F32), DGTTA_ERR_BADARG,
             "mind3d_fwd: NCDHW output must be fp32; NDHWC needs ldc >= 12");
  DG_REQUIRE(out_dtype == DGTTA_F32 || out_dtype == DGTTA_BF16 || out_dtype == DGTTA_F16, DGTTA_ERR_BADARG,
             "mind3d_fwd: bad dtype");
  hipStream_t st = (hipStream_t)stream;
  int td, th, tw;
  tile_counts(D, H, W, td, th, tw);
  DG_REQUIRE((int64_t)td * B <= 65535 && th <= 65535, DGTTA_ERR_UNSUPPORTED, "mind3d_fwd: volume too large for grid");
  const int64_t V = (int64_t)D * H * W;
  float *mws = (float *)ws;
  double *partial = (double *)((char *)ws + align_up((size_t)B * V * 12 * sizeof(float), 256));
  const int nblk = B * td * th * tw;
  float *gmean = (float *)((char *)partial + align_up((size_t)nblk * sizeof(double), 256));

  // taps as mind.py:30-37 evaluates them (fp32, computed by the caller with the reference's expression)
  Taps taps;
  for (int t = 0; t < 7; ++t) taps.g[t] = t < ntaps ? h_taps[t] : 0.f;
  dim3 grid(tw, th, td * B);
  const int R = ntaps / 2;
  int rc = DGTTA_ERR_UNSUPPORTED;
#define MIND_CASE(DL, RR, CGG) \
  if (delta == DL && R == RR) rc = launch_mind_ssd<DL, RR, CGG>(img, noise, rw, mws, partial, D, H, W, td, grid, taps, st);
  MIND_CASE(1, 1, 4) MIND_CASE(1, 2, 4) MIND_CASE(1, 3, 4) MIND_CASE(2, 1, 4) MIND_CASE(2, 2, 4) MIND_CASE(2, 3, 2)
#undef MIND_CASE
  if (rc != DGTTA_OK) return rc;
  hipLaunchKernelGGL(mind_reduce_kernel, dim3(1), dim3(256), 0, st, partial, nblk, 1.0 / ((double)B * V), gmean);
  DG_CHECK_LAUNCH("mind_reduce_kernel");
  const int64_t total = (int64_t)B * V;
  int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  if (!out_ndhwc) {
    hipLaunchKernelGGL((mind_finish_kernel<float, false>), dim3(blocks), dim3(256), 0, st, mws, gmean, (float *)out, 12,
                       V, total);
  } else if (out_dtype == DGTTA_F32) {
    hipLaunchKernelGGL((mind_finish_kernel<float, true>), dim3(blocks), dim3(256), 0, st, mws, gmean, (float *)out,
                       out_ldc, V, total);
  } else if (out_dtype == DGTTA_BF16) {
    hipLaunchKernelGGL((mind_finish_kernel<bf16_t, true>), dim3(blocks), dim3(256), 0, st, mws, gmean, (bf16_t *)out,
                       out_ldc, V, total);
  } else {
    hipLaunchKernelGGL((mind_finish_kernel<f16_t, true>), dim3(blocks), dim3(256), 0, st, mws, gmean, (f16_t *)out,
                       out_ldc, V, total);
  }
  DG_CHECK_LAUNCH("mind_finish_kernel");
  return DGTTA_OK;
}
