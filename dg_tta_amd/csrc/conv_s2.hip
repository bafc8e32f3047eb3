// Stride-2 3x3x3 convolution (forward) of the two large encoder transitions (32 -> 64 at 128^3 -> 64^3, 64 -> 128 at
// 64^3 -> 32^3 in the 3d_fullres plan) with register operands [3P: the stride-2 first conv of an encoder stage of
// PlainConvUNet, built at dg_tta/pretraining/nnUNetTrainer_GIN_MIND.py:46-53].
//
// The layer reads 4x the bytes it writes and has 3 % of the network's FLOPs: it is bound by how the input gets to the
// MFMA, not by the MFMA.  The generic kernel stages a halo tile through LDS per 16-channel chunk (one workgroup per CU,
// load and multiply serialised): 105 us per 128^3 sample against ~35 us of HBM time.  Here
//   * the weights of the workgroup's output-channel group (27 taps x Cin x 32..64 channels = 108 KB) stay in LDS for the
//     lifetime of a persistent workgroup, in MFMA fragment order (the packed weight image already is);
//   * a wave owns 32 consecutive output voxels of one output row and reads the input operands straight from global
//     memory into the MFMA's A layout (lane = voxel, 16 bytes = 8 channels): for a stride-2 row the two voxels a lane
//     needs for kw = 1, 2 share a 128-byte line and kw = 0 is the neighbour lane's line, so the 6 loads of a (kd, kh) row
//     hit the same 33 lines back to back (L1) and every line leaves L2 once per row that uses it;
//   * no barrier after the weights are in: the 8 waves of a workgroup run independent job lists, a wave waiting for its
//     loads leaves the SIMD to the other one.
// Output through a per-wave slab (64-byte voxel rows, non-temporal stores); the InstanceNorm statistics of the block ride
// along as in the other conv kernels (per-chunk fp32 partial sums -> double slots, deterministic).
// Predecessor: conv3_mfma_kernel<.., S = 2, ..> (DGTTA_CONV_S2=1 or 4), which also serves every other shape.
#include "conv_common.h"

namespace {

template <typename T, int KC, int NBW>
__global__ __launch_bounds__(512) void conv_s2_regs_kernel(const T *__restrict__ x, View xv, const T *__restrict__ wimg,
                                                           const float *__restrict__ bias, T *__restrict__ y, View yv, int Cout,
                                                           int NG, int nWB, int mbs, int CH, int nslots, int B,
                                                           double *__restrict__ stats) {
  constexpr int NF = 27 * KC * NBW;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint4 *sW = reinterpret_cast<uint4 *>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  // workgroup -> (channel group, position in the group's job list).  Dispatch is round robin over the 8 XCDs: the NG
  // workgroups that read the same rows and the neighbours in the job list (adjacent output rows share input rows) are
  // put on the SAME XCD, whose L2 then serves the 3.4x re-reads of an input line
  const int ngw = (int)(gridDim.x / NG);            // workgroups per channel group
  int grp = (int)(blockIdx.x % NG), gpos = (int)(blockIdx.x / NG);
  if ((gridDim.x & 7) == 0 && (ngw & 7) == 0) {
    const int xcd = (int)(blockIdx.x & 7), j = (int)(blockIdx.x >> 3);
    grp = j % NG;
    gpos = xcd * (ngw >> 3) + j / NG;
  }
  const int n0 = grp * NBW * 32;
  // weight fragments of this channel group: image order [N/32][KC][27][64 lanes][8] -> LDS [tap][c][nb][64 lanes]
  for (int i = tid; i < NF * 64; i += 512) {
    const int l = i & 63;
    int f = i >> 6;
    const int nb = f % NBW;
    f /= NBW;
    const int c = f % KC, tap = f / KC;
    sW[i] = reinterpret_cast<const uint4 *>(wimg)[((((long long)(n0 / 32 + nb) * KC + c) * 27 + tap) << 6) + l];
  }
  unsigned short *slab = reinterpret_cast<unsigned short *>(smem + NF * 1024 + wave * 2048);
  __syncthreads();

  const int Di = xv.D, Hi = xv.H, Wi = xv.W, Ho = yv.H, Wo = yv.W;
  float bv[NBW];
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb) bv[nb] = bias ? bias[n0 + nb * 32 + r] : 0.f;
  if (stats && blockIdx.x == 0 && tid == 0) reinterpret_cast<long long *>(stats)[0] = nslots;

  const long long njobs = (long long)B * nslots;
  const long long wstride = (long long)ngw * 8;
  for (long long job = (long long)gpos * 8 + wave; job < njobs; job += wstride) {
    const int b = (int)(job / nslots), chunk = (int)(job % nslots);
    float st1[NBW], st2[NBW];
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) st1[nb] = st2[nb] = 0.f;
    const int mb1 = min(mbs, (chunk + 1) * CH);
    for (int mbi = chunk * CH; mbi < mb1; ++mbi) {
      const int wb = mbi % nWB, oh = (mbi / nWB) % Ho, od = mbi / (nWB * Ho);
      const int ow = wb * 32 + r;
      const bool vox_ok = ow < Wo;
      f32x16_t acc[NBW];
#pragma unroll
      for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[nb][q] = bv[nb];
      const T *xb = x + (long long)b * xv.sb + h * 8;
#pragma unroll 1
      for (int kd = 0; kd < 3; ++kd) {
        const int id = 2 * od + kd - 1;
        const bool d_ok = (unsigned)id < (unsigned)Di;
        uint4 xa[3][3][KC];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const int ih = 2 * oh + kh - 1;
          const bool r_ok = d_ok && (unsigned)ih < (unsigned)Hi && vox_ok;
          const T *rowp = xb + (long long)(d_ok ? id : 0) * xv.sd + (long long)((unsigned)ih < (unsigned)Hi ? ih : 0) * xv.sh;
          // kw = 1, 2: this lane's voxels 2 ow, 2 ow + 1 (one 128-byte line for 32 channels).  kw = 0 is voxel 2 ow - 1 = the
          // kw = 2 operand of the lane to the left: taken from there (row shift inside each 32-lane half) instead of a
          // third set of loads that would request the neighbour's lines again; only lane r = 0 loads its own.
#pragma unroll
          for (int kw = 1; kw < 3; ++kw) {
            const int iw = 2 * ow + kw - 1;
            const bool ok = r_ok && (unsigned)iw < (unsigned)Wi;
            const T *p = rowp + (long long)(ok ? iw : 0) * xv.sw;
#pragma unroll
            for (int c = 0; c < KC; ++c) {
              const uint4 v = *reinterpret_cast<const uint4 *>(p + c * 16);
              xa[kh][kw][c] = ok ? v : make_uint4(0, 0, 0, 0);
            }
          }
          {     // the edge voxel 2 ow0 - 1: every lane issues the (clamped) load, lane r = 0 uses it - no divergent branch
            const int iw = 2 * (wb * 32) - 1;
            const bool ok = d_ok && (unsigned)ih < (unsigned)Hi && (unsigned)iw < (unsigned)Wi;
            const T *p = rowp + (long long)(ok ? iw : 0) * xv.sw;
#pragma unroll
            for (int c = 0; c < KC; ++c) {
              const uint4 v = *reinterpret_cast<const uint4 *>(p + c * 16);
              xa[kh][0][c] = ok ? v : make_uint4(0, 0, 0, 0);
            }
          }
        }
        // (after ALL loads of the plane are issued: a shuffle waits for its source)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int c = 0; c < KC; ++c) {
            const uint4 s2 = xa[kh][2][c];
            uint4 sh;
            sh.x = __shfl_up(s2.x, 1, 32);
            sh.y = __shfl_up(s2.y, 1, 32);
            sh.z = __shfl_up(s2.z, 1, 32);
            sh.w = __shfl_up(s2.w, 1, 32);
            if (r != 0) xa[kh][0][c] = sh;
          }
        const uint4 *wf = sW + ((kd * 9 * KC * NBW) << 6) + lane;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw)
#pragma unroll
            for (int c = 0; c < KC; ++c)
#pragma unroll
              for (int nb = 0; nb < NBW; ++nb)
                mfma_step<T>(xa[kh][kw][c], wf[((((kh * 3 + kw) * KC + c) * NBW) + nb) << 6], acc[nb]);
      }
      // epilogue: lane (channel r, half h) holds voxels m = (q & 3) + 8 (q >> 2) + 4 h
      T *yrow = y + (long long)b * yv.sb + (long long)od * yv.sd + (long long)oh * yv.sh + n0;
#pragma unroll
      for (int nb = 0; nb < NBW; ++nb) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int m = (q & 3) + 8 * (q >> 2) + 4 * h;
          const float v = acc[nb][q];
          slab[m * 32 + r] = f32_to_16<T>(v);
          if (wb * 32 + m < Wo) {
            st1[nb] += v;
            st2[nb] += v * v;
          }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int m = t * 16 + (lane >> 2), cq = (lane & 3) * 8, owm = wb * 32 + m;
          const uint4 val = *reinterpret_cast<const uint4 *>(slab + m * 32 + cq);
          if (owm < Wo) {
            typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
            const u32x4_t nv = {val.x, val.y, val.z, val.w};
            __builtin_nontemporal_store(nv, reinterpret_cast<u32x4_t *>(yrow + (long long)owm * yv.sw + nb * 32 + cq));
          }
        }
      }
    }
    if (stats) {
#pragma unroll
      for (int nb = 0; nb < NBW; ++nb) {
        const float a = st1[nb] + __shfl_xor(st1[nb], 32, 64), c2 = st2[nb] + __shfl_xor(st2[nb], 32, 64);
        if (h == 0) {
          double *pp = stats + 32 + (((long long)b * nslots + chunk) * Cout + n0 + nb * 32 + r) * 2;
          pp[0] = (double)a;
          pp[1] = (double)c2;
        }
      }
    }
  }
}

template <typename T, int KC, int NBW>
int launch_s2(const void *x, const View &xv, const void *wimg, const float *bias, void *y, const View &yv, int B, int Cout,
              double *stats, int64_t cap_slots, hipStream_t st) {
  constexpr int LDS = 27 * KC * NBW * 1024 + 8 * 2048;
  auto kern = conv_s2_regs_kernel<T, KC, NBW>;
  static DynLdsOnce once;
  DG_REQUIRE(ensure_dyn_lds(once, reinterpret_cast<const void *>(kern), LDS) == hipSuccess, DGTTA_ERR_LAUNCH,
             "conv_s2_regs: cannot raise the dynamic LDS limit to %d", LDS);
  const int NG = Cout / (32 * NBW), nWB = cdiv(yv.W, 32);
  const long long mbs = (long long)yv.D * yv.H * nWB;
  DG_REQUIRE(mbs < (1ll << 30), DGTTA_ERR_UNSUPPORTED, "conv_s2_regs: volume too large");
  long long CH = cdiv64(mbs, cap_slots);
  if (CH < 2) CH = 2;
  const int nslots = (int)cdiv64(mbs, CH);
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  long long grid = (long long)cus / NG * NG;                       // one workgroup per CU, a multiple of the channel groups
  const long long want = cdiv64((long long)B * nslots, 8) * NG;     // no more workgroups than there are jobs
  if (grid > want) grid = want;
  if (grid < NG) grid = NG;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), LDS, st, (const T *)x, xv, (const T *)wimg, bias, (T *)y, yv, Cout,
                     NG, nWB, (int)mbs, (int)CH, nslots, B, stats);
  DG_CHECK_LAUNCH("conv_s2_regs_kernel");
  return DGTTA_OK;
}

}  // namespace

// Shapes served: 16-bit storage, Cin in {32, 64} unpadded, Cout a multiple of the channel group, >= 32 output columns.
// cap_slots: statistics slots per sample the caller's buffer holds (dgtta_conv3d_stats_bytes).
int conv3_s2_regs(const void *x, const View &xv, const void *wimg, const float *bias, void *y, const View &yv, int B, int Cin,
                  int Cout, int CinP, int CoutP, double *stats, int64_t cap_slots, int dtype, hipStream_t st) {
  if (dtype != DGTTA_BF16 && dtype != DGTTA_F16) return DGTTA_ERR_UNSUPPORTED;
  if (!((Cin == 32 && Cout % 64 == 0) || (Cin == 64 && Cout % 32 == 0)) || CinP != Cin || CoutP != Cout) return DGTTA_ERR_UNSUPPORTED;
  if (yv.W < 32 || xv.sw % 8 || xv.sh % 8 || xv.sd % 8 || xv.sb % 8 || ((uintptr_t)x & 15)) return DGTTA_ERR_UNSUPPORTED;
  if (yv.sw % 8 || yv.sh % 8 || yv.sd % 8 || yv.sb % 8 || ((uintptr_t)y & 15) || cap_slots < 1) return DGTTA_ERR_UNSUPPORTED;
#define GO(TT, K, N) return launch_s2<TT, K, N>(x, xv, wimg, bias, y, yv, B, Cout, stats, cap_slots, st)
  if (dtype == DGTTA_BF16) {
    if (Cin == 32) GO(bf16_t, 2, 2);
    GO(bf16_t, 4, 1);
  }
  if (Cin == 32) GO(f16_t, 2, 2);
  GO(f16_t, 4, 1);
#undef GO
}
