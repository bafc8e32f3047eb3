// 3x3x3 convolution as an implicit GEMM on the gfx950 matrix cores (stride 1 and 2, zero padding 1).
//
//   GEMM view:  M = output voxels (32 per MFMA block, consecutive along W), N = output channels, K = 27 taps x Cin.
//   MFMA:       bf16 storage -> v_mfma_f32_32x32x16_bf16 (8 k-values per lane and operand),
//               fp32 storage -> v_mfma_f32_32x32x2_f32 x4 (a lane's 16 B = 4 k-values feed 4 instructions);
//               fp32 accumulation in both cases.
//   Workgroup:  256 threads = 4 waves, output tile TD x TH x TW voxels (16 or 8 M-blocks) x 32*NB output channels.
//   LDS:        A = input halo tile for one K-chunk, laid out [16-byte channel group][halo voxel] so that the 32 lanes
//               of an M-block (consecutive voxels along W) read consecutive 16-byte slots -> ds_read_b128 without bank
//               conflicts for every tap shift; B = the chunk's weights [tap][group][cout] (same property over cout).
//               (32,4,4) tile, bf16: A 38.3 KiB + B 27 KiB (NB=1) -> two workgroups per CU overlap staging and MFMA.
//   K loop:     for each chunk of Cin: stage A (zero filled outside the volume = the conv's zero padding) and B,
//               barrier, 27 taps x k-steps of MFMA straight from LDS, barrier.
//   Data gradient of a stride-1 conv = the same kernel on dy with mirrored taps and swapped channel roles.
//   Stride 2:   the halo tile is staged de-interleaved by W parity so that lane reads stay contiguous.
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

template <typename T>
struct Elem;
template <>
struct Elem<float> {
  static constexpr int EPV = 4;  // elements per 16 bytes
};
template <>
struct Elem<bf16_t> {
  static constexpr int EPV = 8;
};

template <typename T>
__device__ __forceinline__ void mfma_step(const uint4 &a, const uint4 &b, f32x16_t &acc);
template <>
__device__ __forceinline__ void mfma_step<bf16_t>(const uint4 &a, const uint4 &b, f32x16_t &acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0,
                                                0, 0);
}
template <>
__device__ __forceinline__ void mfma_step<float>(const uint4 &a, const uint4 &b, f32x16_t &acc) {
  // lane half h holds channels 4h..4h+3 of the 8-channel k-step; instruction j contracts the pair {j, 4+j}
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
}

// Tile geometry.  MBW: voxels of an M-block along W (32/16/8); an M-block spans RPM = 32/MBW rows of H.
// MBH x MBD M-blocks per workgroup (MPW = MBH*MBD/4 per wave).  S = stride.
template <int MBW, int MBH, int MBD, int S>
struct Geo {
  static constexpr int RPM = 32 / MBW;
  static constexpr int TW = MBW, TH = RPM * MBH, TD = MBD;
  static constexpr int MB = MBH * MBD, MPW = MB / 4;
  // input halo extents
  static constexpr int ID = (TD - 1) * S + 3, IH = (TH - 1) * S + 3, IW = (TW - 1) * S + 3;
  // LDS row of W: for S=2 the row is split into even / odd columns, each IWH long
  static constexpr int IWH = (S == 1) ? IW : (IW + 1) / 2;
  static constexpr int ROW = (S == 1) ? IW : 2 * IWH;
  static constexpr int NV = ID * IH * ROW;
  __host__ __device__ static constexpr int lds_col(int wx) { return (S == 1) ? wx : (wx & 1) * IWH + (wx >> 1); }
};

template <typename T, int MBW, int MBH, int MBD, int S, int NB, int KSPC>
struct ConvCfg {
  typedef Geo<MBW, MBH, MBD, S> G;
  static constexpr int EPV = Elem<T>::EPV;
  static constexpr int NG = 2 * KSPC;        // 16-byte channel groups per K-chunk
  static constexpr int CK = NG * EPV;        // channels per K-chunk
  static constexpr int NC = 32 * NB;
  static constexpr size_t A_BYTES = (size_t)NG * G::NV * 16;
  static constexpr size_t B_BYTES = (size_t)27 * NG * NC * 16;
  static constexpr size_t LDS_BYTES = A_BYTES + B_BYTES;
};

// x: [B][Di][Hi][Wi][ldx];  w: [27][CoutP][CinP] (k contiguous), tap index mirrored when `mirror`;  y: [B][Do][Ho][Wo][ldy]
template <typename T, int MBW, int MBH, int MBD, int S, int NB, int KSPC>
__global__ __launch_bounds__(256) void conv3_mfma_kernel(const T *__restrict__ x, int ldx, const T *__restrict__ w,
                                                         int mirror, const float *__restrict__ bias,
                                                         T *__restrict__ y, int ldy, int Cin, int Cout, int CinP,
                                                         int CoutP, int Di, int Hi, int Wi, int Do, int Ho, int Wo,
                                                         int tilesW, int tilesH, int tilesD) {
  typedef ConvCfg<T, MBW, MBH, MBD, S, NB, KSPC> Cfg;
  typedef typename Cfg::G G;
  constexpr int EPV = Cfg::EPV, NG = Cfg::NG, CK = Cfg::CK, NC = Cfg::NC, NV = G::NV, MPW = G::MPW;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint4 *sA = reinterpret_cast<uint4 *>(smem);                    // [NG][NV]
  uint4 *sB = reinterpret_cast<uint4 *>(smem + Cfg::A_BYTES);     // [27][NG][NC]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;

  int t = blockIdx.x;
  const int tw = t % tilesW;
  t /= tilesW;
  const int th = t % tilesH;
  t /= tilesH;
  const int td = t % tilesD;
  const int b = t / tilesD;
  const int n0 = blockIdx.y * NC;
  const int od0 = td * G::TD, oh0 = th * G::TH, ow0 = tw * G::TW;      // output tile origin
  const int id0 = od0 * S - 1, ih0 = oh0 * S - 1, iw0 = ow0 * S - 1;    // input halo origin

  // per-lane LDS voxel offsets of this wave's M-blocks (tap offset is added as a compile-time constant)
  int a_off[MPW];
#pragma unroll
  for (int i = 0; i < MPW; ++i) {
    const int mb = wave * MPW + i;
    const int mbd = mb / MBH, mbh = mb % MBH;
    const int row = mbh * G::RPM + r / MBW, col = r % MBW;
    a_off[i] = ((mbd * S) * G::IH + row * S) * G::ROW + ((S == 1) ? col : col);   // S=2: column index in the half row
  }

  f32x16_t acc[MPW][NB];
#pragma unroll
  for (int i = 0; i < MPW; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

  const int64_t xb = (int64_t)b * Di * Hi * Wi;
  const int cin_lim = (Cin + EPV - 1) / EPV * EPV;       // channels that may be read (caller guarantees ldx >= this)

  for (int kc = 0; kc < CinP; kc += CK) {
    __syncthreads();   // previous chunk's LDS reads are done
    // ---- stage A: halo voxels x channel groups of this chunk
    for (int idx = tid; idx < NV * NG; idx += 256) {
      const int g = idx % NG, v = idx / NG;
      const int wx_l = v % G::ROW, hy = (v / G::ROW) % G::IH, dz = v / (G::ROW * G::IH);
      int wx = wx_l;
      if (S == 2) wx = (wx_l >= G::IWH) ? 2 * (wx_l - G::IWH) + 1 : 2 * wx_l;    // inverse of lds_col
      const int gd = id0 + dz, gh = ih0 + hy, gw = iw0 + wx;
      const int c = kc + g * EPV;
      uint4 val = make_uint4(0, 0, 0, 0);
      if ((S == 1 || wx < G::IW) && (unsigned)gd < (unsigned)Di && (unsigned)gh < (unsigned)Hi &&
          (unsigned)gw < (unsigned)Wi && c < cin_lim)
        val = *reinterpret_cast<const uint4 *>(x + (xb + ((int64_t)gd * Hi + gh) * Wi + gw) * ldx + c);
      sA[g * NV + v] = val;
    }
    // ---- stage B: weights of this chunk for output channels n0..n0+NC
    for (int idx = tid; idx < 27 * NC * NG; idx += 256) {
      const int g = idx % NG, n = (idx / NG) % NC, tap = idx / (NG * NC);
      const int wt = mirror ? 26 - tap : tap;
      uint4 val = make_uint4(0, 0, 0, 0);
      if (n0 + n < CoutP) val = *reinterpret_cast<const uint4 *>(w + ((int64_t)wt * CoutP + n0 + n) * CinP + kc + g * EPV);
      sB[(tap * NG + g) * NC + n] = val;
    }
    __syncthreads();
    // ---- 27 taps x KSPC k-steps of MFMA from LDS
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) {
      const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
      const int tap_off = (kd * G::IH + kh) * G::ROW + G::lds_col(kw);
#pragma unroll
      for (int ks = 0; ks < KSPC; ++ks) {
        const int g = 2 * ks + h;
        uint4 bf[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) bf[j] = sB[(tap * NG + g) * NC + j * 32 + r];
#pragma unroll
        for (int i = 0; i < MPW; ++i) {
          const uint4 af = sA[g * NV + a_off[i] + tap_off];
#pragma unroll
          for (int j = 0; j < NB; ++j) mfma_step<T>(af, bf[j], acc[i][j]);
        }
      }
    }
  }

  // ---- epilogue: bias, convert, store (acc row m = (q&3) + 8*(q>>2) + 4*h, column = r)
#pragma unroll
  for (int i = 0; i < MPW; ++i) {
    const int mb = wave * MPW + i;
    const int mbd = mb / MBH, mbh = mb % MBH;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int co = n0 + j * 32 + r;
      const float bv = (bias && co < Cout) ? bias[co] : 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int m = (q & 3) + 8 * (q >> 2) + 4 * h;
        const int od = od0 + mbd, oh = oh0 + mbh * G::RPM + m / MBW, ow = ow0 + m % MBW;
        if (co < Cout && od < Do && oh < Ho && ow < Wo)
          st_f<T>(y + (((int64_t)b * Do + od) * Ho + oh) * Wo * (int64_t)ldy + (int64_t)ow * ldy + co, acc[i][j][q] + bv);
      }
    }
  }
}

template <typename T, int MBW, int MBH, int MBD, int S, int NB, int KSPC>
int launch_conv(const void *x, int ldx, const void *w, int mirror, const float *bias, void *y, int ldy, int B, int Cin,
                int Cout, int CinP, int CoutP, int Di, int Hi, int Wi, int Do, int Ho, int Wo, hipStream_t st) {
  typedef ConvCfg<T, MBW, MBH, MBD, S, NB, KSPC> Cfg;
  typedef typename Cfg::G G;
  static bool attr_set = false;
  auto kern = conv3_mfma_kernel<T, MBW, MBH, MBD, S, NB, KSPC>;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                        (int)Cfg::LDS_BYTES);
    attr_set = true;
  }
  const int tW = cdiv(Wo, G::TW), tH = cdiv(Ho, G::TH), tD = cdiv(Do, G::TD);
  const int64_t tiles = (int64_t)tW * tH * tD * B;
  DG_REQUIRE(tiles < (1ll << 31), DGTTA_ERR_UNSUPPORTED, "conv3_mfma: too many tiles");
  dim3 grid((unsigned)tiles, (unsigned)cdiv(CoutP, Cfg::NC));
  hipLaunchKernelGGL(kern, grid, dim3(256), Cfg::LDS_BYTES, st, (const T *)x, ldx, (const T *)w, mirror, bias, (T *)y,
                     ldy, Cin, Cout, CinP, CoutP, Di, Hi, Wi, Do, Ho, Wo, tW, tH, tD);
  DG_CHECK_LAUNCH("conv3_mfma_kernel");
  return DGTTA_OK;
}

template <typename T>
int dispatch_conv(const void *x, int ldx, const void *w, int mirror, const float *bias, void *y, int ldy, int B, int Cin,
                  int Cout, int CinP, int CoutP, int Di, int Hi, int Wi, int stride, hipStream_t st) {
  constexpr int EPV = Elem<T>::EPV;
  // shape requirements of the vectorised staging
  if (ldx % EPV != 0 || ((uintptr_t)x & 15) != 0 || CinP % (2 * EPV) != 0 || ldx < (Cin + EPV - 1) / EPV * EPV)
    return DGTTA_ERR_UNSUPPORTED;
  const int Do = (Di - 1) / stride + 1, Ho = (Hi - 1) / stride + 1, Wo = (Wi - 1) / stride + 1;
#define ARGS x, ldx, w, mirror, bias, y, ldy, B, Cin, Cout, CinP, CoutP, Di, Hi, Wi, Do, Ho, Wo, st
  if (stride == 1) {
    if (Wo >= 32) return launch_conv<T, 32, 4, 4, 1, 1, 1>(ARGS);
    if (Wo >= 16) return launch_conv<T, 16, 4, 4, 1, 1, 1>(ARGS);
    return launch_conv<T, 8, 2, 8, 1, 1, 1>(ARGS);
  }
  if (stride == 2) {
    if (Wo >= 16) return launch_conv<T, 16, 2, 4, 2, 1, 1>(ARGS);
    return launch_conv<T, 8, 2, 4, 2, 1, 1>(ARGS);
  }
#undef ARGS
  return DGTTA_ERR_UNSUPPORTED;
}

}  // namespace

// w_kmajor: [27][CoutP][CinP] with the K (input-channel) index contiguous; mirror: use tap 26-t.
int conv3_fwd_mfma(const void *x, int ldx, const void *w_kmajor, int mirror, const float *bias, void *y, int ldy, int B,
                   int Cin, int Cout, int CinP, int CoutP, int Di, int Hi, int Wi, int stride, int dtype,
                   hipStream_t st) {
  if (dtype == DGTTA_F32)
    return dispatch_conv<float>(x, ldx, w_kmajor, mirror, bias, y, ldy, B, Cin, Cout, CinP, CoutP, Di, Hi, Wi, stride, st);
  if (dtype == DGTTA_BF16)
    return dispatch_conv<bf16_t>(x, ldx, w_kmajor, mirror, bias, y, ldy, B, Cin, Cout, CinP, CoutP, Di, Hi, Wi, stride,
                                 st);
  return DGTTA_ERR_UNSUPPORTED;
}

int conv3_wgrad_mfma(const void *, int, const void *, int, float *, float *, void *, size_t, int, int, int, int, int, int,
                     int, int, int, hipStream_t) {
  return DGTTA_ERR_UNSUPPORTED;
}
