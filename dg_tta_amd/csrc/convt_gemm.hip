// ConvTranspose3d(k=2, s=2) forward and data gradient of the two large decoder stages (64 -> 32 at 64^3 -> 128^3 and
// 128 -> 64 at 32^3 -> 64^3 in the 3d_fullres plan) as register-operand GEMMs [3P: nn.ConvTranspose3d inside
// PlainConvUNet's decoder, built at dg_tta/pretraining/nnUNetTrainer_GIN_MIND.py:46-53].
//
// Both are HBM-bound (K is one or four short channel runs per voxel; the forward writes 4x what it reads), so the kernel
// is organised around the memory stream, not the MFMA:
//   * the whole weight tensor (8 * Cin * Cout 16-bit values = 32 / 128 KB) sits in LDS for the lifetime of a persistent
//     workgroup, already in MFMA fragment order (one 1-KB fragment = one conflict-free ds_read_b128 per lane);
//   * a wave owns 32 voxels of one row of the COARSE lattice at a time and holds ALL their K values in registers
//     (16-byte global loads straight into the MFMA operand layout, issued back to back: no LDS staging, no barrier in the
//     loop) - every byte of the activations is requested from memory once;
//   * the weights are the MFMA's M operand and the voxels its N operand, so a lane ends up with 4 consecutive channels
//     of ONE voxel per accumulator quad: packed 8-byte writes into a small per-wave slab, read back as the 64
//     contiguous bytes of a voxel by 4 lanes (16-byte global stores).
// forward (MODE 0):  out[2v + (od,oh,ow)][co] = bias[co] + sum_ci x[v][ci] w[ci][co][od][oh][ow]
//                    4 GEMMs (one per (od,oh)) with K = Cin and N = (ow, co): the two fine voxels 2w, 2w+1 of a row
// data grad (MODE 1): dx[v][ci] = sum_{od,oh,ow,co} dout[2v + (od,oh,ow)][co] w[ci][co][od][oh][ow]
//                    1 GEMM with K = 4 row segments x (ow, co) and N = ci
// The predecessor (8 pointwise launches' worth of classes in conv3_mfma_kernel, each re-staging x through LDS in
// 16-channel chunks) stays behind DGTTA_CONVT_GEMM=0 and serves every other shape.
#include "conv_common.h"

namespace {

// fragment f = ((r * NBR + nb) * S + s) * KSP + ks;  lane l: channel m = l % 32 of block nb, k = ks*16 + (l/32)*8 + e
template <typename T>
__global__ void convT_frag_pack_kernel(const float *__restrict__ w, T *__restrict__ blob, int Cin, int Cout, int mode) {
  const int KSP = (mode == 0 ? Cin : 2 * Cout) / 16, NBR = (mode == 0 ? 2 * Cout : Cin) / 32;
  const int R = mode == 0 ? 4 : 1, S = mode == 0 ? 1 : 4;
  const int64_t total = (int64_t)R * NBR * S * KSP * 512;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int e = (int)(i & 7), l = (int)((i >> 3) & 63);
    int64_t f = i >> 9;
    const int ks = (int)(f % KSP);
    f /= KSP;
    const int s = (int)(f % S);
    f /= S;
    const int nb = (int)(f % NBR), r = (int)(f / NBR);
    const int n = nb * 32 + (l & 31), k = ks * 16 + (l >> 5) * 8 + e;
    int ci, co, od, oh, ow;
    if (mode == 0) {
      ci = k, ow = n / Cout, co = n % Cout, od = r >> 1, oh = r & 1;
    } else {
      ci = n, ow = k / Cout, co = k % Cout, od = s >> 1, oh = s & 1;
    }
    st_f<T>(blob + i, w[((((int64_t)ci * Cout + co) * 2 + od) * 2 + oh) * 2 + ow]);
  }
}

template <typename T, int MODE, int KSP, int NBR>
__global__ __launch_bounds__(512) void convT_gemm_kernel(const T *__restrict__ in, int ldin, const T *__restrict__ wfrag,
                                                        const float *__restrict__ bias, T *__restrict__ out, int ldout,
                                                        int Di, int Hi, int Wi, int Cout, int nWB, long long nMB, int abl) {
  constexpr int R = MODE == 0 ? 4 : 1, S = MODE == 0 ? 1 : 4, NF = R * NBR * S * KSP;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint4 *sW = reinterpret_cast<uint4 *>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  for (int i = tid; i < NF * 64; i += 512) sW[i] = reinterpret_cast<const uint4 *>(wfrag)[i];

  float *sBias = reinterpret_cast<float *>(smem + NF * 1024);      // [NBR * 32]: bias of N index n = (ow, co)
  if (MODE == 0) {
    for (int i = tid; i < NBR * 32; i += 512) sBias[i] = bias ? bias[i % Cout] : 0.f;
  }
  unsigned char *slab = smem + NF * 1024 + NBR * 32 * 4 + wave * (32 * 72);
  __syncthreads();

  for (long long mb = (long long)blockIdx.x * 8 + wave; mb < nMB; mb += (long long)gridDim.x * 8) {
    const int wb = (int)(mb % nWB);
    long long row = mb / nWB;
    const int hh = (int)(row % Hi);
    row /= Hi;
    const int d = (int)(row % Di), b = (int)(row / Di);
    const int w = wb * 32 + r;
    const bool valid = w < Wi;
    const int wc = valid ? w : Wi - 1;
    const long long crow = (((long long)b * Di + d) * Hi + hh) * Wi;                        // coarse row start (voxels)
    auto frow = [&](int od, int oh) {                                                       // fine row start (voxels)
      return (((long long)b * 2 * Di + 2 * d + od) * 2 * Hi + 2 * hh + oh) * 2 * Wi;
    };
    uint4 xa[S][KSP];
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
      for (int ks = 0; ks < KSP; ++ks) {
        const T *p;
        if (MODE == 0) {
          p = in + (crow + wc) * ldin + ks * 16 + h * 8;
        } else {
          constexpr int KH = KSP / 2;     // k-steps per fine voxel
          p = in + (frow(s >> 1, s & 1) + 2 * wc + ks / KH) * ldin + (ks % KH) * 16 + h * 8;
        }
        xa[s][ks] = *reinterpret_cast<const uint4 *>(p);
      }
    // (rr, nb) deliberately NOT unrolled: unrolled, the scheduler hoists every weight fragment read of the 8-16 blocks
    // in front of the first MFMA and runs out of registers
#pragma unroll 1
    for (int rr = 0; rr < R; ++rr) {
      const long long orow = MODE == 0 ? frow(rr >> 1, rr & 1) : crow;
#pragma unroll 1
      for (int nb = 0; nb < NBR; ++nb) {
        f32x16_t acc;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = 0.f;
        const uint4 *wf = sW + (((rr * NBR + nb) * S * KSP) << 6) + lane;
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
          for (int ks = 0; ks < KSP; ++ks) {
            if (abl == 3) acc[0] += __uint_as_float(xa[s][ks].x);
            else mfma_step<T>(wf[(s * KSP + ks) << 6], xa[s][ks], acc);
          }
        // lane (voxel r, half h) holds channels 8j + 4h .. +3 of the block: through a per-wave slab [32 voxels][72 bytes]
        // (pitch 72: the 8-byte writes of 32 voxels hit 64 distinct banks) so that 4 consecutive lanes store the 64
        // contiguous bytes of one voxel - 8-byte stores straight from the accumulator layout ran at 2.7 TB/s
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float4 bq = make_float4(0.f, 0.f, 0.f, 0.f);
          if (MODE == 0) bq = *reinterpret_cast<const float4 *>(sBias + nb * 32 + 8 * j + 4 * h);
          uint2 pk;
          pk.x = pack2_16<T>(acc[4 * j] + bq.x, acc[4 * j + 1] + bq.y);
          pk.y = pack2_16<T>(acc[4 * j + 2] + bq.z, acc[4 * j + 3] + bq.w);
          *reinterpret_cast<uint2 *>(slab + r * 72 + 16 * j + 8 * h) = pk;
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int v = t * 16 + (lane >> 2), c0 = nb * 32 + (lane & 3) * 8, wv = wb * 32 + v;
          const uint4 val = *reinterpret_cast<const uint4 *>(slab + v * 72 + (lane & 3) * 16);
          T *o;
          if (MODE == 0) o = out + (orow + 2 * wv + c0 / Cout) * ldout + c0 % Cout;
          else o = out + (orow + wv) * ldout + c0;
          if (wv < Wi && (abl != 2 || val.x == 0x12345u)) *reinterpret_cast<uint4 *>(o) = val;
        }
      }
    }
  }
}

template <typename T, int MODE, int KSP, int NBR>
int launch_gemm(const void *in, int ldin, const T *blob, const float *bias, void *out, int ldout, int B, int Di, int Hi, int Wi,
                int Cout, hipStream_t st) {
  constexpr int R = MODE == 0 ? 4 : 1, S = MODE == 0 ? 1 : 4, LDS = R * NBR * S * KSP * 1024 + NBR * 32 * 4 + 8 * 32 * 72;
  auto kern = convT_gemm_kernel<T, MODE, KSP, NBR>;
  static DynLdsOnce once;
  DG_REQUIRE(ensure_dyn_lds(once, reinterpret_cast<const void *>(kern), LDS) == hipSuccess, DGTTA_ERR_LAUNCH,
             "convT_gemm: cannot raise the dynamic LDS limit to %d", LDS);
  const int nWB = cdiv(Wi, 32);
  const long long nMB = (long long)B * Di * Hi * nWB;
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const long long want = cdiv64(nMB, 8), cap = (long long)cus * (LDS <= 64 * 1024 ? 2 : 1);
  hipLaunchKernelGGL(kern, dim3((unsigned)(want < cap ? want : cap)), dim3(512), LDS, st, (const T *)in, ldin, blob, bias,
                     (T *)out, ldout, Di, Hi, Wi, Cout, nWB, nMB,
                     DG_LAB(convt_gemm_abl) >= '2' ? DG_LAB(convt_gemm_abl) - '0' : 0);      // timing models: diagnostic build only
  DG_CHECK_LAUNCH("convT_gemm_kernel");
  return DGTTA_OK;
}

template <typename T>
int run_gemm(int mode, const void *in, int ldin, const float *w_t, const float *bias, void *out, int ldout, void *ws, int B,
             int Cin, int Cout, int Di, int Hi, int Wi, hipStream_t st) {
  T *blob = (T *)ws;
  const int64_t total = (int64_t)8 * Cin * Cout;
  hipLaunchKernelGGL((convT_frag_pack_kernel<T>), dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, st, w_t, blob, Cin, Cout,
                     mode);
  DG_CHECK_LAUNCH("convT_frag_pack_kernel");
  const int KSP = (mode == 0 ? Cin : 2 * Cout) / 16;
#define GO(M, K, N) return launch_gemm<T, M, K, N>(in, ldin, blob, bias, out, ldout, B, Di, Hi, Wi, Cout, st)
  if (mode == 0) {
    if (KSP == 4) GO(0, 4, 2);
    GO(0, 8, 4);
  }
  if (KSP == 4) GO(1, 4, 2);
  GO(1, 8, 4);
#undef GO
}

}  // namespace

// shapes the register-operand kernel serves: 16-bit storage, Cin = 2 * Cout in {64, 128} (the weights fit in LDS and the
// per-wave K fits in registers), whole 32-voxel blocks worth having (Wi >= 32), vector-aligned operands
bool convT_gemm_eligible(int mode, const void *in, int ldin, const void *out, int ldout, int Cin, int Cout, int Wi, int dtype) {
  if (dgtta_switches().convt_gemm == '0') return false;
  if (dtype != DGTTA_BF16 && dtype != DGTTA_F16) return false;
  if (!((Cin == 64 && Cout == 32) || (Cin == 128 && Cout == 64))) return false;
  if (Wi < 32) return false;
  return ldin % 8 == 0 && ((uintptr_t)in & 15) == 0 && ldout % 8 == 0 && ((uintptr_t)out & 15) == 0;
}

int convT_gemm_run(int mode, const void *in, int ldin, const float *w_t, const float *bias, void *out, int ldout, void *ws, int B,
                   int Cin, int Cout, int Di, int Hi, int Wi, int dtype, hipStream_t st) {
  if (dtype == DGTTA_BF16) return run_gemm<bf16_t>(mode, in, ldin, w_t, bias, out, ldout, ws, B, Cin, Cout, Di, Hi, Wi, st);
  return run_gemm<f16_t>(mode, in, ldin, w_t, bias, out, ldout, ws, B, Cin, Cout, Di, Hi, Wi, st);
}
