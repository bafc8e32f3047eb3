// 3x3x3 convolution as an implicit GEMM on the gfx950 matrix cores (stride 1 and 2, zero padding 1).
//
//   GEMM view:  M = output voxels (32 per MFMA block, consecutive along W), N = output channels, K = 27 taps x Cin.
//   MFMA:       bf16 storage -> v_mfma_f32_32x32x16_bf16 (8 k-values per lane and operand),
//               fp32 storage -> v_mfma_f32_32x32x2_f32 x4 (a lane's 16 B = 4 k-values feed 4 instructions);
//               fp32 accumulation in both cases.
//   Workgroup:  NW = 8 (or 4) waves, output tile TD x TH x TW voxels (16 or 8 M-blocks) x 32*NB output channels.
//   LDS:        A = input halo tile for one K-chunk, laid out [16-byte channel group][halo voxel] so that the 32 lanes
//               of an M-block (consecutive voxels along W) read consecutive 16-byte slots -> ds_read_b128 without bank
//               conflicts for every tap shift; B = the chunk's weights [tap][group][cout] (same property over cout).
//               (32,4,4) tile, bf16: A 38.3 KiB + B 27 KiB (NB=1) -> two workgroups per CU overlap staging and MFMA.
//   K loop:     for each chunk of Cin: stage A (zero filled outside the volume = the conv's zero padding) and B,
//               barrier, 27 taps x k-steps of MFMA straight from LDS, barrier.
//   Data gradient of a stride-1 conv = the same kernel on dy with mirrored taps and swapped channel roles.
//   Stride 2:   the halo tile is staged de-interleaved by W parity so that lane reads stay contiguous.
#include "common.h"
#include <stdlib.h>

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
// MBH x MBD M-blocks per workgroup (MPW = MBH*MBD/4 per wave).  S = stride; S == 0 selects the POINTWISE variant
// (stride 1, centre tap only, no halo) used by the 2x2x2 transposed-conv compositions.
template <int MBW, int MBH, int MBD, int S>
struct Geo {
  static constexpr int SE = (S == 0) ? 1 : S;        // effective stride
  static constexpr int HALO = (S == 0) ? 0 : 1;
  static constexpr int NTAP = (S == 0) ? 1 : 27;     // taps staged in LDS
  static constexpr int RPM = 32 / MBW;
  static constexpr int TW = MBW, TH = RPM * MBH, TD = MBD;
  static constexpr int MB = MBH * MBD, MPW = MB / 4;
  // input halo extents
  static constexpr int ID = (TD - 1) * SE + 1 + 2 * HALO, IH = (TH - 1) * SE + 1 + 2 * HALO,
                       IW = (TW - 1) * SE + 1 + 2 * HALO;
  // LDS row of W: for S=2 the row is split into even / odd columns, each IWH long
  static constexpr int IWH = (S != 2) ? IW : (IW + 1) / 2;
  static constexpr int ROW = (S != 2) ? IW : 2 * IWH;
  static constexpr int NV = ID * IH * ROW;
  __host__ __device__ static constexpr int lds_col(int wx) { return (S != 2) ? wx : (wx & 1) * IWH + (wx >> 1); }
};

template <typename T, int MBW, int MBH, int MBD, int S, int NB, int KSPC>
struct ConvCfg {
  typedef Geo<MBW, MBH, MBD, S> G;
  static constexpr int EPV = Elem<T>::EPV;
  static constexpr int NG = 2 * KSPC;        // 16-byte channel groups per K-chunk
  static constexpr int CK = NG * EPV;        // channels per K-chunk
  static constexpr int NC = 32 * NB;
  static constexpr size_t A_BYTES = (size_t)NG * G::NV * 16;
  static constexpr size_t B_BYTES = (size_t)G::NTAP * NG * NC * 16;
  static constexpr size_t LDS_BYTES = A_BYTES + B_BYTES;
};

// Strided view of a channels-last volume: element strides (channel stride 1) + logical extent.  Lets the same kernel
// run on parity sub-lattices (stride-2 data gradient, 2x2x2 transposed conv) without copies.
struct View {
  long long sb, sd, sh, sw;
  int D, H, W;
};
// weight tap used by each of the 27 virtual taps (-1: tap not present)
struct Taps {
  signed char wt[27];
};

// A launch can run up to 8 independent "classes" (blockIdx.z) that share shapes but differ in operand offsets and tap
// tables: the 8 parity sub-lattices of a stride-2 data gradient, or the 8 output offsets of a 2x2x2 transposed conv.
struct ConvClasses {
  int n;
  int acc[8];
  long long xoff[8], yoff[8];   // element offsets of the operands of class c
  Taps taps[8];
  // K concatenation (pointwise variant): input channel c lives in segment c / kseg at element offset segoff[c / kseg]
  // (the 8 parity sub-lattices of a transposed conv's output gradient); kseg == 0: plain channels
  int kseg;
  long long segoff[8];
};

// x: view xv;  y: view yv;  virtual tap t uses weight tap taps.wt[t].
// w is stored in "LDS image order" [N/32][K/(2*EPV)][ntaps_src][2][32][EPV]: the B tile of a K-chunk is one contiguous
// run, so its staging is a linear, fully coalesced copy (see conv_weight_image_index).
// AC ("all classes"): data gradient of a stride-2 conv in ONE pass over dy.  The 8 parity classes of the input lattice
// need dy at 8 shifts (0/+1 per axis) only; a wave keeps 8 accumulators (one per class), reads each shifted A fragment
// once and feeds the (class, tap) pairs that use it (27 in total = every real tap once).  Class c is written to
// y + cs.yoff[c] with the strides of yv (the parity view).  Replaces 8 class launches that each re-staged the dy tile.
template <typename T, int MBW, int MBH, int MBD, int S, int NB, int KSPC, int ABL = 0, int NW = 4, bool AC = false>   // ABL: diagnostic ablation; NW waves
__global__ __launch_bounds__(NW * 64) void conv3_mfma_kernel(const T *__restrict__ x, View xv, const T *__restrict__ w,
                                                         ConvClasses cs, const float *__restrict__ bias,
                                                         T *__restrict__ y, View yv, int Cin, int Cout, int CinP,
                                                         int CoutP, int tilesW, int tilesH, int tilesD,
                                                         double *__restrict__ stats, int ntaps_src) {
  const int cls = blockIdx.z;
  x += cs.xoff[cls];
  if (!AC) y += cs.yoff[cls];
  const Taps &taps = cs.taps[cls];
  const int accumulate = cs.acc[cls];
  const int Di = xv.D, Hi = xv.H, Wi = xv.W, Do = yv.D, Ho = yv.H, Wo = yv.W;
  typedef ConvCfg<T, MBW, MBH, MBD, S, NB, KSPC> Cfg;
  typedef typename Cfg::G G;
  constexpr int EPV = Cfg::EPV, NG = Cfg::NG, CK = Cfg::CK, NC = Cfg::NC, NV = G::NV, MPW = G::MB / NW, NT = NW * 64;
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
  constexpr int SE = G::SE, NTAP = G::NTAP;
  const int od0 = td * G::TD, oh0 = th * G::TH, ow0 = tw * G::TW;      // output tile origin
  const int id0 = od0 * SE - G::HALO, ih0 = oh0 * SE - G::HALO, iw0 = ow0 * SE - G::HALO;    // input halo origin

  // per-lane LDS voxel offsets of this wave's M-blocks (tap offset is added as a compile-time constant)
  int a_off[MPW];
#pragma unroll
  for (int i = 0; i < MPW; ++i) {
    const int mb = wave * MPW + i;
    const int mbd = mb / MBH, mbh = mb % MBH;
    const int row = mbh * G::RPM + r / MBW, col = r % MBW;
    a_off[i] = ((mbd * SE) * G::IH + row * SE) * G::ROW + col;   // S=2: column index in the half row
  }

  constexpr int NACC = AC ? 8 : NB;
  f32x16_t acc[MPW][NACC];
#pragma unroll
  for (int i = 0; i < MPW; ++i)
#pragma unroll
    for (int j = 0; j < NACC; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

  const T *xbp = x + (int64_t)b * xv.sb;
  const int cin_lim = (Cin + EPV - 1) / EPV * EPV;       // channels that may be read (caller guarantees ldx >= this)

  // Register staging, software pipelined: all global loads of a chunk are issued back to back (unconditional loads from
  // a clamped address + select, so the compiler emits no per-load branch / wait), written to LDS one chunk later, and
  // the loads of chunk k+1 are in flight while chunk k is being multiplied.
  constexpr int NA = (NV * NG + NT - 1) / NT, NBL = (NTAP * NC * NG + NT - 1) / NT;
  uint4 ra[NA], rb[NBL];
  auto load_chunk = [&](int kc) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int idx = tid + i * NT;
      const int g = idx % NG, v = idx / NG;
      const int wx_l = v % G::ROW, hy = (v / G::ROW) % G::IH, dz = v / (G::ROW * G::IH);
      int wx = wx_l;
      if (S == 2) wx = (wx_l >= G::IWH) ? 2 * (wx_l - G::IWH) + 1 : 2 * wx_l;    // inverse of lds_col
      const int gd = id0 + dz, gh = ih0 + hy, gw = iw0 + wx;
      int c = kc + g * EPV;
      long long soff = 0;
      if (S == 0 && cs.kseg > 0) {        // K concatenation over parity segments
        const int seg = c / cs.kseg;
        c -= seg * cs.kseg;
        soff = cs.segoff[seg];
      }
      const bool ok = idx < NV * NG && (S != 2 || wx < G::IW) && (unsigned)gd < (unsigned)Di &&
                      (unsigned)gh < (unsigned)Hi && (unsigned)gw < (unsigned)Wi && c < cin_lim;
      const T *p = ok ? xbp + soff + gd * xv.sd + gh * xv.sh + gw * xv.sw + c : x;
      if (ABL == 1) {
        ra[i] = make_uint4(idx, 0, 0, 0);
        continue;
      }
      const uint4 val = *reinterpret_cast<const uint4 *>(p);
      ra[i] = ok ? val : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NBL; ++i) {
      const int idx = tid + i * NT;       // == LDS index (tap*NG + g)*NC + n
      const int n = idx % NC, g = (idx / NC) % NG, tap = (S == 0) ? 13 : (idx / (NG * NC)) % 27;
      const int wt = AC ? tap : taps.wt[tap];
      const bool ok = idx < NTAP * NC * NG && wt >= 0;
      const int64_t chunk2 = (int64_t)(kc / (2 * EPV)) + g / 2;     // K-chunk of 2*EPV channels
      const int64_t off = (((((int64_t)(n0 + n) / 32) * (CinP / (2 * EPV)) + chunk2) * ntaps_src + wt) * 2 + (g & 1)) * 32 +
                          (n0 + n) % 32;
      const T *p = ok ? w + off * EPV : w;
      if (ABL == 1 || ABL == 4) {
        rb[i] = make_uint4(idx, 0, 0, 0);
        continue;
      }
      const uint4 val = *reinterpret_cast<const uint4 *>(p);
      rb[i] = ok ? val : make_uint4(0, 0, 0, 0);
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int idx = tid + i * NT;
      if (ABL == 2) {
        asm volatile("" ::"v"(ra[i].x));
        continue;
      }
      if (idx < NV * NG) sA[(idx % NG) * NV + idx / NG] = ra[i];
    }
#pragma unroll
    for (int i = 0; i < NBL; ++i) {
      const int idx = tid + i * NT;
      if (idx < NTAP * NC * NG) sB[idx] = rb[i];
    }
  };

  load_chunk(0);
  for (int kc = 0; kc < CinP; kc += CK) {
    __syncthreads();   // previous chunk's LDS reads are done
    store_chunk();
    __syncthreads();
    if (kc + CK < CinP) load_chunk(kc + CK);
    if (AC) {
      // ---- 8 shifts x KSPC k-steps: A fragment of shift (sd,sh,sw) feeds class/tap pairs: per axis shift 1 <- (parity 1,
      //      real tap 0); shift 0 <- (parity 0, tap 1) and (parity 1, tap 2)
#pragma unroll
      for (int ks = 0; ks < KSPC; ++ks) {
        const int g = 2 * ks + h;
#pragma unroll
        for (int sh8 = 0; sh8 < 8; ++sh8) {
          const int sd = sh8 >> 2, shh = (sh8 >> 1) & 1, sw = sh8 & 1;
          const int tap_off = ((1 + sd) * G::IH + (1 + shh)) * G::ROW + G::lds_col(1 + sw);
          uint4 af[MPW];
#pragma unroll
          for (int i = 0; i < MPW; ++i) af[i] = sA[g * NV + a_off[i] + tap_off];
#pragma unroll
          for (int od = 0; od < 2 - sd; ++od)
#pragma unroll
            for (int oh = 0; oh < 2 - shh; ++oh)
#pragma unroll
              for (int ow = 0; ow < 2 - sw; ++ow) {
                // option 0 on a shift-0 axis: parity 0 / tap 1; option 1: parity 1 / tap 2; shift-1 axis: parity 1 / tap 0
                const int pd = sd ? 1 : od, ph = shh ? 1 : oh, pw = sw ? 1 : ow;
                const int td = sd ? 0 : 1 + od, th = shh ? 0 : 1 + oh, tw = sw ? 0 : 1 + ow;
                const int tap = td * 9 + th * 3 + tw, c8 = pd * 4 + ph * 2 + pw;
                const uint4 bfr = sB[(tap * NG + g) * NC + r];
#pragma unroll
                for (int i = 0; i < MPW; ++i) mfma_step<T>(af[i], bfr, acc[i][c8]);
              }
        }
      }
    } else
    // ---- 27 taps x KSPC k-steps of MFMA from LDS
#pragma unroll
    for (int tap = (S == 0 ? 13 : 0); tap < (S == 0 ? 14 : 27); ++tap) {
      if (taps.wt[tap] < 0) continue;     // wave-uniform
      const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
      const int tap_off = (S == 0) ? 0 : (kd * G::IH + kh) * G::ROW + G::lds_col(kw);
      const int tb = (S == 0) ? 0 : tap;   // tap slot in the LDS weight tile
#pragma unroll
      for (int ks = 0; ks < KSPC; ++ks) {
        const int g = 2 * ks + h;
        uint4 bf[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) bf[j] = (ABL == 7) ? make_uint4(tap, j, tid, 1) : sB[(tb * NG + g) * NC + j * 32 + r];
#pragma unroll
        for (int i = 0; i < MPW; ++i) {
          const uint4 af = (ABL == 6 || ABL == 7) ? make_uint4(tap, i, tid, 0) : sA[g * NV + a_off[i] + tap_off];
#pragma unroll
          for (int j = 0; j < NB; ++j) {
            if (ABL == 3) acc[i][j][0] += __uint_as_float(af.x ^ bf[j].x);
            else mfma_step<T>(af, bf[j], acc[i][j]);
          }
        }
      }
    }
  }

  // ---- epilogue: bias, convert, store (acc row m = (q&3) + 8*(q>>2) + 4*h, column = r); optional per-channel
  //      sum / sum of squares of this tile for the following InstanceNorm (fp32 within the tile, double partials)
  float st1[NB], st2[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) st1[j] = st2[j] = 0.f;
  if (AC) {
    // epilogue: one class at a time through a per-wave fp32 slab [32 voxels][32 channels] (row pitch 36 floats) in the
    // finished A/B tiles, so that a lane handles 8 consecutive channels of a voxel: 16-byte (bf16) loads / stores of
    // whole 64-byte rows instead of 2-byte accesses (the skip-connection sum makes this a read-modify-write)
    const bool vec = Cout % 8 == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0 && yv.sw % 8 == 0 && yv.sh % 8 == 0 &&
                     yv.sd % 8 == 0 && yv.sb % 8 == 0;
    __syncthreads();
    float *slab = reinterpret_cast<float *>(smem) + wave * (32 * 36);
#pragma unroll
    for (int i = 0; i < MPW; ++i) {
      const int mb = wave * MPW + i;
      const int mbd = mb / MBH, mbh = mb % MBH;
      const int co = n0 + r;
#pragma unroll
      for (int c8 = 0; c8 < 8; ++c8) {
        if (!vec) {
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const int m = (q & 3) + 8 * (q >> 2) + 4 * h;
            const int od = od0 + mbd, oh = oh0 + mbh * G::RPM + m / MBW, ow = ow0 + m % MBW;
            if (co < Cout && od < Do && oh < Ho && ow < Wo) {
              T *o = y + cs.yoff[c8] + b * yv.sb + od * yv.sd + oh * yv.sh + ow * yv.sw + co;
              float v = acc[i][c8][q];
              if (accumulate) v += ld_f<T>(o);
              st_f<T>(o, v);
            }
          }
          continue;
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) slab[((q & 3) + 8 * (q >> 2) + 4 * h) * 36 + r] = acc[i][c8][q];
        // (a wave reads back only its own slab: LDS operations of one wave complete in order)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int m = t * 16 + (lane >> 2), cq = (lane & 3) * 8;
          const float4 v0 = *reinterpret_cast<const float4 *>(slab + m * 36 + cq);
          const float4 v1 = *reinterpret_cast<const float4 *>(slab + m * 36 + cq + 4);
          float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
          const int od = od0 + mbd, oh = oh0 + mbh * G::RPM + m / MBW, ow = ow0 + m % MBW;
          if (n0 + cq < Cout && od < Do && oh < Ho && ow < Wo) {
            T *o = y + cs.yoff[c8] + b * yv.sb + od * yv.sd + oh * yv.sh + ow * yv.sw + n0 + cq;
            if (sizeof(T) == 2) {
              uint4 *o4 = reinterpret_cast<uint4 *>(o);
              if (accumulate) {
                const uint4 old = *o4;
                const unsigned wv[4] = {old.x, old.y, old.z, old.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  v[2 * e] += __uint_as_float(wv[e] << 16);
                  v[2 * e + 1] += __uint_as_float(wv[e] & 0xffff0000u);
                }
              }
              uint4 pk;
              pk.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
              pk.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
              pk.z = (unsigned)f32_to_bf16(v[4]) | ((unsigned)f32_to_bf16(v[5]) << 16);
              pk.w = (unsigned)f32_to_bf16(v[6]) | ((unsigned)f32_to_bf16(v[7]) << 16);
              *o4 = pk;
            } else {
              float4 *o4 = reinterpret_cast<float4 *>(o);
              if (accumulate) {
                const float4 a0 = o4[0], a1 = o4[1];
                v[0] += a0.x; v[1] += a0.y; v[2] += a0.z; v[3] += a0.w;
                v[4] += a1.x; v[5] += a1.y; v[6] += a1.z; v[7] += a1.w;
              }
              o4[0] = make_float4(v[0], v[1], v[2], v[3]);
              o4[1] = make_float4(v[4], v[5], v[6], v[7]);
            }
          }
        }
      }
    }
    return;
  }
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
        if (co < Cout && od < Do && oh < Ho && ow < Wo) {
          T *o = y + b * yv.sb + od * yv.sd + oh * yv.sh + ow * yv.sw + co;
          float v = acc[i][j][q] + bv;
          if (accumulate) v += ld_f<T>(o);
          if (ABL == 5) {
            if (v == 1234.5f) st_f<T>(o, v);
          } else {
            st_f<T>(o, v);
          }
          st1[j] += v;
          st2[j] += v * v;
        }
      }
    }
  }
  if (stats) {
    __syncthreads();                       // all waves are done with the A/B tiles: reuse LDS for the reduction
    float *red = reinterpret_cast<float *>(smem);      // [4 waves][NC][2]
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const float a = st1[j] + __shfl_xor(st1[j], 32, 64), c = st2[j] + __shfl_xor(st2[j], 32, 64);
      if (h == 0) {
        red[(wave * NC + j * 32 + r) * 2 + 0] = a;
        red[(wave * NC + j * 32 + r) * 2 + 1] = c;
      }
    }
    __syncthreads();
    const int tiles_per_b = tilesW * tilesH * tilesD;
    if (tid < NC && n0 + tid < Cout) {
      double s = 0.0, ss = 0.0;
#pragma unroll
      for (int wv = 0; wv < NW; ++wv) {
        s += (double)red[(wv * NC + tid) * 2 + 0];
        ss += (double)red[(wv * NC + tid) * 2 + 1];
      }
      double *p = stats + 32 + (((int64_t)b * tiles_per_b + (blockIdx.x % tiles_per_b)) * Cout + n0 + tid) * 2;
      p[0] = s;
      p[1] = ss;
    }
    if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) reinterpret_cast<long long *>(stats)[0] = tiles_per_b;
  }
}

// all-classes stride-2 data gradient (AC = true): tiles over the dy lattice, grid.z = 1
template <typename T, int MBW, int MBH, int MBD>
int launch_conv_allcls(const void *x, const View &xv, const void *w, const ConvClasses &cs, void *y, const View &yv, int B,
                       int Cin, int Cout, int CinP, int CoutP, hipStream_t st) {
  typedef ConvCfg<T, MBW, MBH, MBD, 1, 1, 1> Cfg;
  typedef typename Cfg::G G;
  auto kern = conv3_mfma_kernel<T, MBW, MBH, MBD, 1, 1, 1, 0, 8, true>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)Cfg::LDS_BYTES);
    attr_set = true;
  }
  const int tW = cdiv(yv.W, G::TW), tH = cdiv(yv.H, G::TH), tD = cdiv(yv.D, G::TD);
  const int64_t tiles = (int64_t)tW * tH * tD * B;
  DG_REQUIRE(tiles < (1ll << 31), DGTTA_ERR_UNSUPPORTED, "conv3_mfma: too many tiles");
  dim3 grid((unsigned)tiles, (unsigned)cdiv(CoutP, Cfg::NC), 1u);
  hipLaunchKernelGGL(kern, grid, dim3(8 * 64), Cfg::LDS_BYTES, st, (const T *)x, xv, (const T *)w, cs, (const float *)nullptr,
                     (T *)y, yv, Cin, Cout, CinP, CoutP, tW, tH, tD, (double *)nullptr, 27);
  DG_CHECK_LAUNCH("conv3_mfma_kernel<all classes>");
  return DGTTA_OK;
}

template <typename T, int MBW, int MBH, int MBD, int S, int NB, int KSPC, int NW = 4>
int launch_conv(const void *x, const View &xv, const void *w, const ConvClasses &cs, const float *bias, void *y,
                const View &yv, int B, int Cin, int Cout, int CinP, int CoutP, double *stats, int ntaps_src,
                hipStream_t st) {
  typedef ConvCfg<T, MBW, MBH, MBD, S, NB, KSPC> Cfg;
  typedef typename Cfg::G G;
  static bool attr_set = false;
  auto kern = conv3_mfma_kernel<T, MBW, MBH, MBD, S, NB, KSPC, 0, NW>;
  if (MBW == 32 && S == 1 && NW == 8) {
    static const char *abl = getenv("DGTTA_CONV_ABL");      // diagnostic only
    if (abl && abl[0] == '1') kern = conv3_mfma_kernel<T, MBW, MBH, MBD, S, NB, KSPC, 1, NW>;
    if (abl && abl[0] == '2') kern = conv3_mfma_kernel<T, MBW, MBH, MBD, S, NB, KSPC, 2, NW>;
    if (abl && abl[0] == '3') kern = conv3_mfma_kernel<T, MBW, MBH, MBD, S, NB, KSPC, 3, NW>;
    if (abl && abl[0] == '4') kern = conv3_mfma_kernel<T, MBW, MBH, MBD, S, NB, KSPC, 4, NW>;
    if (abl && abl[0] == '5') kern = conv3_mfma_kernel<T, MBW, MBH, MBD, S, NB, KSPC, 5, NW>;
    if (abl && abl[0] == '6') kern = conv3_mfma_kernel<T, MBW, MBH, MBD, S, NB, KSPC, 6, NW>;
    if (abl && abl[0] == '7') kern = conv3_mfma_kernel<T, MBW, MBH, MBD, S, NB, KSPC, 7, NW>;
    if (abl) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)Cfg::LDS_BYTES);
  }
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)Cfg::LDS_BYTES);
    attr_set = true;
  }
  const int tW = cdiv(yv.W, G::TW), tH = cdiv(yv.H, G::TH), tD = cdiv(yv.D, G::TD);
  const int64_t tiles = (int64_t)tW * tH * tD * B;
  DG_REQUIRE(tiles < (1ll << 31), DGTTA_ERR_UNSUPPORTED, "conv3_mfma: too many tiles");
  dim3 grid((unsigned)tiles, (unsigned)cdiv(CoutP, Cfg::NC), (unsigned)cs.n);
  hipLaunchKernelGGL(kern, grid, dim3(NW * 64), Cfg::LDS_BYTES, st, (const T *)x, xv, (const T *)w, cs, bias, (T *)y, yv,
                     Cin, Cout, CinP, CoutP, tW, tH, tD, stats, ntaps_src);
  DG_CHECK_LAUNCH("conv3_mfma_kernel");
  return DGTTA_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Row-reuse kernel (bf16, stride 1, all 27 taps, W tiles of 32): the LDS-lean variant for the large layers.
//
// In the kernel above every MFMA fetches a fresh A fragment from LDS (1.5 ds_read_b128 per MFMA incl. B): with 32
// output channels the LDS pipe saturates before the matrix cores do.  Here a wave owns a PD x PH patch of output rows
// (one row = 32 voxels along W = one 32x32 accumulator) and walks the (PD+2) x (PH+2) INPUT rows of its patch: an
// input-row fragment (one per kw shift) is read once and feeds every (kd,kh) tap whose output row lies in the patch
// (up to 9 MFMAs per read), and the 27 weight fragments of the K-step live in registers.  PD = PH = 2: 48 A reads + 27
// B reads per 108 MFMAs (0.7 per MFMA instead of 1.5).
//   Workgroup: WD x WH waves, tile (PD*WD) x (PH*WH) x 32 voxels x 32 output channels; persistent over a contiguous
//   range of (spatial tile, channel block) jobs, so the first K-chunk of the next tile is in flight during the last
//   MFMA phase of the current one.
//   Staging:   global_load_lds_dwordx4 (no staging registers, no ds_write): A chunk (16 channels) double buffered,
//   B chunk (27 x 1 KiB, contiguous in the packed image) single buffered -- it is copied to registers at phase start.
//   Padding voxels read a 16-byte zero constant.
template <int PD, int PH, int WD, int WH>
struct RowsCfg {
  static constexpr int NW = WD * WH, NT = NW * 64;
  static constexpr int TD = PD * WD, TH = PH * WH, TW = 32;
  static constexpr int ID = TD + 2, IH = TH + 2, ROW = TW + 2;
  // A chunk in LDS: one 68-entry block (16 B entries) per input row (dz,hy):
  //   [0,32) channel group 0, columns 0..31 | [32,64) group 1, columns 0..31 | 64,65 group 0, columns 32,33 | 66,67 group 1
  // = one full 1-KiB DMA piece + one 4-lane piece per row; all per-piece address arithmetic is scalar.
  static constexpr int NROW = ID * IH, RB = 68;
  static constexpr size_t A_BYTES = ((size_t)NROW * RB * 16 + 1023) / 1024 * 1024;
  static constexpr size_t B_BYTES = 27 * 1024;
  static constexpr size_t RED_BYTES = (size_t)NW * 32 * 2 * sizeof(float);
  static constexpr size_t LDS_BYTES = 2 * A_BYTES + B_BYTES + RED_BYTES;
};

__device__ const uint4 g_zero16 = {0u, 0u, 0u, 0u};

typedef __attribute__((address_space(3))) void lds_void_t;

// LDS-DMA of 16 bytes per lane: lane i's bytes land at lds_addr + 16*i (lds_addr wave-uniform).  Issued from inline asm
// on purpose: the compiler then keeps no s_waitcnt bookkeeping for it (with the builtin it drains vmcnt(0) before the
// next ds_read, i.e. before the MFMA phase the copy is meant to overlap); the kernel waits with dma_wait_all() before
// the barrier that publishes the buffer.  M0 is compiler-reserved, so it is saved and restored in the same statement.
__device__ __forceinline__ void dma16_to_lds(const void *gsrc, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_addr)
               : "memory");
}
__device__ __forceinline__ void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// workgroup barrier that orders LDS traffic only: __syncthreads() would also drain vmcnt, i.e. wait for DMA in flight
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ unsigned lds_addr_of(const void *p) {
  return __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void_t *)p);
}

template <int PD, int PH, int WD, int WH, int ABL = 0>   // ABL: diagnostic ablation
__global__ __launch_bounds__(WD *WH * 64) void conv3_rows_kernel(const bf16_t *__restrict__ x, View xv,
                                                                 const bf16_t *__restrict__ w, Taps taps,
                                                                 const float *__restrict__ bias, bf16_t *__restrict__ y,
                                                                 View yv, int Cin, int Cout, int CinP, int tilesW,
                                                                 int tilesH, int tilesD, int nblkN, int njobs,
                                                                 double *__restrict__ stats, int ntaps_src) {
  typedef RowsCfg<PD, PH, WD, WH> Cfg;
  constexpr int NW = Cfg::NW, IH = Cfg::IH;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char *sAb = smem;                                   // two A buffers
  unsigned char *sBb = smem + 2 * Cfg::A_BYTES;                // [27][2][32] x 16 B
  float *red = reinterpret_cast<float *>(smem + 2 * Cfg::A_BYTES + Cfg::B_BYTES);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wd = wave / WH, wh = wave % WH;
  const int Di = xv.D, Hi = xv.H, Wi = xv.W, Do = yv.D, Ho = yv.H, Wo = yv.W;
  const int nk = CinP / 16;
  const int cin_lim = (Cin + 7) / 8 * 8;

  // contiguous job range of this workgroup; workgroups of one XCD (blockIdx % 8) get neighbouring ranges
  const int G = gridDim.x;
  const int lw = (G % 8 == 0) ? (int)(blockIdx.x % 8) * (G / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
  const int j0 = (int)(((long long)njobs * lw) / G), j1 = (int)(((long long)njobs * (lw + 1)) / G);
  const int nph = (j1 - j0) * nk;
  if (nph == 0) return;

  struct Job {
    int b, n0, od0, oh0, ow0, tile;
  };
  auto decode = [&](int j) {
    Job q;
    const int td = j % tilesD;
    j /= tilesD;
    const int nb = j % nblkN;
    j /= nblkN;
    const int th = j % tilesH;
    j /= tilesH;
    const int tw = j % tilesW;
    q.b = j / tilesW;
    q.n0 = nb * 32;
    q.od0 = td * Cfg::TD;
    q.oh0 = th * Cfg::TH;
    q.ow0 = tw * 32;
    q.tile = (tw * tilesH + th) * tilesD + td;
    return q;
  };

  // source taps of the B pieces this wave copies (read from the kernel arguments before any DMA is in flight)
  int my_wt[(27 + NW - 1) / NW];
#pragma unroll
  for (int i = 0; i < (27 + NW - 1) / NW; ++i) {
    const int tap = wave + i * NW;
    my_wt[i] = __builtin_amdgcn_readfirstlane(tap < 27 ? (int)taps.wt[tap] : -1);
  }

  // DMA of K-chunk kc of job q into A buffer `buf` and the B buffer, one piece per call: this wave's share is NPR input
  // rows (main + tail piece each) then NPB B pieces.  Issued one at a time between MFMA groups: a burst of all pieces
  // blocks the issuing wave until the memory pipeline has absorbed them (measured: half of the kernel's cycles).
  constexpr int NPR = (Cfg::NROW + NW - 1) / NW, NPB = (27 + NW - 1) / NW, NPIECE = 2 * NPR + NPB;
  // lane roles inside a piece: main = (group lane>>5, column lane&31); tail (lanes 0..3) = (group lane>>1, column 32 + lane&1)
  const int m_g = lane >> 5, m_wx = lane & 31, t_g = (lane >> 1) & 1, t_wx = 32 + (lane & 1);
  auto issue_piece = [&](const Job &q, int kc, int buf, int i) {
    if (i < 2 * NPR) {
      const int row = wave + (i >> 1) * NW;          // wave-uniform
      const bool tail = i & 1;
      if (row < Cfg::NROW && (!tail || lane < 4) && !(ABL == 7 && tail)) {
        const int dz = row / IH, hy = row % IH;
        const int gd = q.od0 - 1 + dz, gh = q.oh0 - 1 + hy;
        const int g = tail ? t_g : m_g, gw = q.ow0 - 1 + (tail ? t_wx : m_wx);
        const int c = kc * 16 + g * 8;
        const bool ok = (unsigned)gd < (unsigned)Di && (unsigned)gh < (unsigned)Hi && (unsigned)gw < (unsigned)Wi &&
                        c < cin_lim;
        const bf16_t *rowp = x + (long long)q.b * xv.sb + gd * xv.sd + gh * xv.sh;      // scalar part
        const void *src = ok ? (const void *)(rowp + gw * xv.sw + c) : (const void *)&g_zero16;
        if (ABL == 1 || ABL == 4) return;
        dma16_to_lds(src, lds_addr_of(sAb + (size_t)buf * Cfg::A_BYTES + (row * Cfg::RB + (tail ? 64 : 0)) * 16));
      }
    } else {
      const int tap = wave + (i - 2 * NPR) * NW;
      if (tap < 27) {
        const int wt = my_wt[i - 2 * NPR];
        const void *src = wt >= 0 ? (const void *)(w + ((((long long)(q.n0 / 32) * nk + kc) * ntaps_src + wt) * 64 + lane) * 8)
                                  : (const void *)&g_zero16;
        if (ABL == 1 || ABL == 5) return;
        dma16_to_lds(src, lds_addr_of(sBb + tap * 1024));
      }
    }
  };

  f32x16_t acc[PD][PH];
  // this lane's A read bases (16-byte entries) for the three kw shifts: patch origin row block + entry of column r + kw
  int abase[3];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) {
    const int col = r + kw;
    abase[kw] = ((wd * PD) * IH + wh * PH) * Cfg::RB + (col < 32 ? h * 32 + col : 64 + h * 2 + (col - 32));
  }

  Job cur = decode(j0);
  float bv = 0.f;
#pragma unroll
  for (int i = 0; i < NPIECE; ++i) issue_piece(cur, 0, 0, i);
  int kc = 0, jn = j0;
  unsigned long long tseg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;     // ABL 6: cycle stamps per segment (diagnostic)
  auto stamp = [&](int k) {
    if (ABL == 6) {
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      __builtin_amdgcn_sched_barrier(0);
      tseg[k] += t - tprev;
      tprev = t;
    }
  };
  if (ABL == 6) tprev = __builtin_amdgcn_s_memtime();
  for (int p = 0; p < nph; ++p) {
    dma_wait_all();
    stamp(0);                 // waiting for the DMA
    lds_barrier();            // chunk p has landed; every wave is done with phase p-1
    uint4 breg[27];
    {
      const uint4 *sB = reinterpret_cast<const uint4 *>(sBb);
#pragma unroll
      for (int t = 0; t < 27; ++t) breg[t] = sB[t * 64 + lane];
    }
    if (kc == 0) {            // bias of this job, fetched while no DMA is in flight (its wait would drain them)
      const int co = cur.n0 + r;
      bv = (bias && co < Cout) ? bias[co] : 0.f;
      asm volatile("" ::"v"(bv));
    }
    lds_barrier();            // B buffer is free again
    stamp(1);                 // barrier + B fragments + barrier
    // prefetch the next phase (next K-chunk of this job, or chunk 0 of the next job)
    Job nxt = cur;
    int kn = kc + 1;
    if (kn == nk) {
      kn = 0;
      if (p + 1 < nph) nxt = decode(jn + 1);
    }
    const bool more = p + 1 < nph;
    stamp(2);

    if (kc == 0) {
#pragma unroll
      for (int i = 0; i < PD; ++i)
#pragma unroll
        for (int j = 0; j < PH; ++j)
#pragma unroll
          for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    }
    {
      const uint4 *sA = reinterpret_cast<const uint4 *>(sAb + (size_t)(p & 1) * Cfg::A_BYTES);
#pragma unroll
      for (int dz = 0; dz < PD + 2; ++dz)
#pragma unroll
        for (int hy = 0; hy < PH + 2; ++hy)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const uint4 af = sA[abase[kw] + (dz * IH + hy) * Cfg::RB];
            {   // spread this wave's DMA pieces of the next chunk evenly over the A-read steps
              constexpr int NSTEP = (PD + 2) * (PH + 2) * 3;
              const int step = (dz * (PH + 2) + hy) * 3 + kw;
#pragma unroll
              for (int i = 0; i < NPIECE; ++i)
                if (step == (i * NSTEP) / NPIECE + 1 && more) issue_piece(nxt, kn, (p + 1) & 1, i);
            }
#pragma unroll
            for (int kd = 0; kd < 3; ++kd)
#pragma unroll
              for (int kh = 0; kh < 3; ++kh) {
                const int od = dz - kd, oh = hy - kh;
                if (od >= 0 && od < PD && oh >= 0 && oh < PH) {
                  if (ABL == 3) acc[od][oh][0] += __uint_as_float(af.x ^ breg[(kd * 3 + kh) * 3 + kw].x);
                  else mfma_step<bf16_t>(af, breg[(kd * 3 + kh) * 3 + kw], acc[od][oh]);
                }
              }
          }
    }

    stamp(3);                 // MFMA loop
    if (kc == nk - 1) {
      // ---- epilogue of job `cur`: bias, convert, transpose through LDS (this phase's A buffer, one 8-KiB slab per
      //      wave) so that a lane stores 16 bytes = 8 channels of a voxel; optional per-channel sum / sum of squares of
      //      the unrounded values for the following InstanceNorm
      float st1 = 0.f, st2 = 0.f;
      lds_barrier();          // every wave has finished reading this phase's A buffer
      stamp(5);               // (diagnostic) barrier skew
      // lane-derived epilogue indices are rebuilt from a laundered lane id: otherwise the compiler hoists them out of
      // the phase loop, keeps them live across the MFMA phase and spills (a scratch reload's wait drains the DMA)
      int le = lane;
      asm volatile("" : "+v"(le));
      const int re = le & 31, he = le >> 5;
      const int co = cur.n0 + re;
      bf16_t *slab = reinterpret_cast<bf16_t *>(sAb + (size_t)(p & 1) * Cfg::A_BYTES) + wave * (PD * PH * 1024);
#pragma unroll
      for (int i = 0; i < PD; ++i)
#pragma unroll
        for (int j = 0; j < PH; ++j) {
          const int od = cur.od0 + wd * PD + i, oh = cur.oh0 + wh * PH + j;
          const bool row_ok = od < Do && oh < Ho;
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const int m = (q & 3) + 8 * (q >> 2) + 4 * he;
            const float v = acc[i][j][q] + bv;
            slab[(i * PH + j) * 1024 + m * 32 + re] = f32_to_bf16(v);
            if (row_ok && co < Cout && cur.ow0 + m < Wo) {
              st1 += v;
              st2 += v * v;
            }
          }
        }
      stamp(6);               // (diagnostic) convert + slab writes
      // (a wave reads back only its own slab: LDS operations of one wave complete in order)
#pragma unroll
      for (int i = 0; i < PD; ++i)
#pragma unroll
        for (int j = 0; j < PH; ++j) {
          const int od = cur.od0 + wd * PD + i, oh = cur.oh0 + wh * PH + j;
          bf16_t *orow = y + (long long)cur.b * yv.sb + od * yv.sd + oh * yv.sh + cur.n0;
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const int m = t * 16 + (le >> 2), cq = (le & 3) * 8;
            const uint4 val = *reinterpret_cast<const uint4 *>(slab + (i * PH + j) * 1024 + m * 32 + cq);
            const int ow = cur.ow0 + m;
            if (ABL != 2 && od < Do && oh < Ho && ow < Wo && cur.n0 + cq < Cout)
              *reinterpret_cast<uint4 *>(orow + ow * yv.sw + cq) = val;
          }
        }
      stamp(7);               // (diagnostic) slab reads + global stores
      if (stats) {
        const float a = st1 + __shfl_xor(st1, 32, 64), c2 = st2 + __shfl_xor(st2, 32, 64);
        if (he == 0) {
          red[(wave * 32 + re) * 2 + 0] = a;
          red[(wave * 32 + re) * 2 + 1] = c2;
        }
        lds_barrier();
        const int tiles_per_b = tilesW * tilesH * tilesD;
        if (tid < 32 && cur.n0 + tid < Cout) {
          double s = 0.0, ss = 0.0;
#pragma unroll
          for (int wv = 0; wv < NW; ++wv) {
            s += (double)red[(wv * 32 + tid) * 2 + 0];
            ss += (double)red[(wv * 32 + tid) * 2 + 1];
          }
          double *pp = stats + 32 + (((int64_t)cur.b * tiles_per_b + cur.tile) * Cout + cur.n0 + tid) * 2;
          pp[0] = s;
          pp[1] = ss;
        }
        if (lw == 0 && tid == 0 && p == nk - 1) reinterpret_cast<long long *>(stats)[0] = tiles_per_b;
      }
    }
    stamp(4);                 // epilogue
    kc = kn;
    if (kn == 0) {
      cur = nxt;
      ++jn;
    }
  }
  if (ABL == 6 && stats && lane == 0) {
    for (int k = 0; k < 8; ++k) stats[4096 + ((size_t)blockIdx.x * NW + wave) * 8 + k] = (double)tseg[k];
  }
}

template <int PD, int PH, int WD, int WH>
int launch_conv_rows(const void *x, const View &xv, const void *w, const Taps &taps, const float *bias, void *y,
                     const View &yv, int B, int Cin, int Cout, int CinP, int CoutP, double *stats, int ntaps_src,
                     hipStream_t st) {
  typedef RowsCfg<PD, PH, WD, WH> Cfg;
  auto kern = conv3_rows_kernel<PD, PH, WD, WH>;
  static const char *abl = getenv("DGTTA_ROWS_ABL");      // diagnostic only
  if (abl && abl[0] == '1') kern = conv3_rows_kernel<PD, PH, WD, WH, 1>;
  if (abl && abl[0] == '2') kern = conv3_rows_kernel<PD, PH, WD, WH, 2>;
  if (abl && abl[0] == '3') kern = conv3_rows_kernel<PD, PH, WD, WH, 3>;
  if (abl && abl[0] == '4') kern = conv3_rows_kernel<PD, PH, WD, WH, 4>;
  if (abl && abl[0] == '5') kern = conv3_rows_kernel<PD, PH, WD, WH, 5>;
  if (abl && abl[0] == '6') kern = conv3_rows_kernel<PD, PH, WD, WH, 6>;
  if (abl && abl[0] == '7') kern = conv3_rows_kernel<PD, PH, WD, WH, 7>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)Cfg::LDS_BYTES);
    DG_REQUIRE(e == hipSuccess, DGTTA_ERR_LAUNCH, "conv3_rows: cannot raise the dynamic LDS limit to %zu", Cfg::LDS_BYTES);
    attr_set = true;
  }
  const int tW = cdiv(yv.W, 32), tH = cdiv(yv.H, Cfg::TH), tD = cdiv(yv.D, Cfg::TD), nblkN = cdiv(CoutP, 32);
  const long long njobs = (long long)tW * tH * tD * nblkN * B;
  DG_REQUIRE(njobs < (1ll << 31), DGTTA_ERR_UNSUPPORTED, "conv3_rows: too many tiles");
  static int ncu = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n > 0 ? n : 256;
  }();
  const int wg_per_cu = (int)((160 * 1024) / Cfg::LDS_BYTES) > 0 ? (int)((160 * 1024) / Cfg::LDS_BYTES) : 1;
  const int grid = (int)(njobs < (long long)ncu * wg_per_cu ? njobs : (long long)ncu * wg_per_cu);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(Cfg::NT), Cfg::LDS_BYTES, st, (const bf16_t *)x, xv, (const bf16_t *)w, taps,
                     bias, (bf16_t *)y, yv, Cin, Cout, CinP, tW, tH, tD, nblkN, (int)njobs, stats, ntaps_src);
  DG_CHECK_LAUNCH("conv3_rows_kernel");
  return DGTTA_OK;
}

View dense_view(int B, int D, int H, int W, int ld) {
  (void)B;
  View v;
  v.sw = ld;
  v.sh = (long long)W * ld;
  v.sd = (long long)H * W * ld;
  v.sb = (long long)D * H * W * ld;
  v.D = D;
  v.H = H;
  v.W = W;
  return v;
}
// sub-lattice of parity (pd,ph,pw) of a dense volume: elements 2v+p
View parity_view(int D, int H, int W, int ld, int pd, int ph, int pw, long long *offset) {
  View v = dense_view(1, D, H, W, ld);
  *offset = ((long long)pd * H * W + (long long)ph * W + pw) * ld;
  v.sd *= 2;
  v.sh *= 2;
  v.sw *= 2;
  v.D = (D - pd + 1) / 2;
  v.H = (H - ph + 1) / 2;
  v.W = (W - pw + 1) / 2;
  return v;
}

template <typename T>
bool operand_ok(const void *p, long long ld_elems, int Cin, int CinP) {
  constexpr int EPV = Elem<T>::EPV;
  return ld_elems % EPV == 0 && ((uintptr_t)p & 15) == 0 && CinP % (2 * EPV) == 0 &&
         ld_elems >= (Cin + EPV - 1) / EPV * EPV;
}

// picks the tile shape from the (virtual) output extent
template <typename T>
int dispatch_conv_classes(const void *x, const View &xv, const void *w, const ConvClasses &cs, const float *bias, void *y,
                          const View &yv, int B, int Cin, int Cout, int CinP, int CoutP, int stride, hipStream_t st,
                          double *stats = nullptr, int ntaps_src = 27) {
#define ARGS x, xv, w, cs, bias, y, yv, B, Cin, Cout, CinP, CoutP, stats, ntaps_src, st
  const long long vox = (long long)yv.D * yv.H * yv.W * B;
  if (stride == 0) {      // pointwise (centre tap only, no halo)
    if (yv.W >= 32) return launch_conv<T, 32, 4, 4, 0, 1, 1, 8>(ARGS);
    if (yv.W >= 16) return launch_conv<T, 16, 4, 4, 0, 1, 1, 8>(ARGS);
    if (vox <= 4096) return launch_conv<T, 8, 2, 2, 0, 1, 1>(ARGS);
    return launch_conv<T, 8, 2, 8, 0, 1, 1, 8>(ARGS);
  }
  if (stride == 1 && sizeof(T) == 2 && yv.W >= 32 && cs.n == 1 && cs.acc[0] == 0 && cs.kseg == 0 && cs.xoff[0] == 0 &&
      cs.yoff[0] == 0) {
    const char *rows = getenv("DGTTA_CONV_ROWS");   // diagnostic / tests: "0" forces the generic kernel, "1" this one
    bool all_taps = true;
    for (int t = 0; t < 27; ++t) all_taps = all_taps && cs.taps[0].wt[t] >= 0;
    const bool vec_out = Cout % 8 == 0 && ((uintptr_t)y & 15) == 0 && yv.sw % 8 == 0 && yv.sh % 8 == 0 &&
                         yv.sd % 8 == 0 && yv.sb % 8 == 0;
    // enough (tile, channel block) jobs to fill the chip with one persistent workgroup per CU; below that the generic kernel wins
    const long long njobs = (long long)cdiv(yv.W, 32) * cdiv(yv.H, 8) * cdiv(yv.D, 4) * cdiv(CoutP, 32) * B;
    if (all_taps && vec_out && (njobs >= 256 || (rows && rows[0] == '1')) && !(rows && rows[0] == '0'))
      return launch_conv_rows<2, 2, 2, 4>(x, xv, w, cs.taps[0], bias, y, yv, B, Cin, Cout, CinP, CoutP, stats, ntaps_src,
                                          st);
  }
  if (stride == 1) {
    static const char *var = getenv("DGTTA_CONV_VARIANT");      // diagnostic: tile-shape experiments
    if (yv.W >= 32 && var && var[0] == 'a') return launch_conv<T, 32, 4, 4, 1, 1, 2>(ARGS);   // CK = 2 k-steps
    if (yv.W >= 32 && var && var[0] == 'b') return launch_conv<T, 32, 4, 4, 1, 2, 1>(ARGS);   // 64 output channels
    if (yv.W >= 32 && var && var[0] == 'd') return launch_conv<T, 32, 4, 2, 1, 1, 1>(ARGS);   // 256-voxel tile, 3 WG/CU
    if (yv.W >= 32 && var && var[0] == 'w') return launch_conv<T, 32, 4, 4, 1, 1, 1, 8>(ARGS);   // 8 waves per workgroup
    if (yv.W >= 32 && var && var[0] == 'e') return launch_conv<T, 32, 2, 4, 1, 1, 1>(ARGS);
    // 8 waves per workgroup (2 M-blocks each): 16 waves per CU hide the LDS / barrier latency (+27 % over 4 waves)
    if (yv.W >= 32 && var && var[0] == 'x') return launch_conv<T, 32, 4, 4, 1, 1, 1, 4>(ARGS);
    if (yv.W >= 32) return launch_conv<T, 32, 4, 4, 1, 1, 1, 8>(ARGS);
    // tiny volumes are latency bound (few workgroups, a long serial K loop): when the channel padding allows it, 2
    // k-steps per chunk halve the barrier / load round trips.  (128-voxel tiles for the 16^3 layers measured faster in
    // isolation but slower inside the network: 23.0 vs 20.9 ms per epoch.)
    const bool k2 = CinP % (4 * Elem<T>::EPV) == 0 && !(var && var[0] == 'k');
    if (yv.W >= 16) return launch_conv<T, 16, 4, 4, 1, 1, 1, 8>(ARGS);
    if (vox <= 4096 && k2) return launch_conv<T, 8, 2, 2, 1, 1, 2>(ARGS);
    if (vox <= 4096) return launch_conv<T, 8, 2, 2, 1, 1, 1>(ARGS);     // tiny volumes: more, smaller workgroups
    return launch_conv<T, 8, 2, 8, 1, 1, 1, 8>(ARGS);
  }
  if (stride == 2) {
    // 8 waves per workgroup (one M-block each): 231 -> 153 us at 128^3 -> 64^3, 32 -> 64 channels
    static const char *v2 = getenv("DGTTA_CONV_S2");      // diagnostic: "4" = the former 4-wave tiles
    if (v2 && v2[0] == '4') {
      if (yv.W >= 16) return launch_conv<T, 16, 2, 4, 2, 1, 1>(ARGS);
      return launch_conv<T, 8, 2, 4, 2, 1, 1>(ARGS);
    }
    if (yv.W >= 32) return launch_conv<T, 32, 4, 2, 2, 1, 1, 8>(ARGS);
    if (yv.W >= 16) return launch_conv<T, 16, 2, 4, 2, 1, 1, 8>(ARGS);
    return launch_conv<T, 8, 2, 4, 2, 1, 1, 8>(ARGS);
  }
#undef ARGS
  return DGTTA_ERR_UNSUPPORTED;
}

template <typename T>
int dispatch_conv(const void *x, const View &xv, const void *w, const Taps &taps, const float *bias, void *y,
                  const View &yv, int B, int Cin, int Cout, int CinP, int CoutP, int stride, int accumulate,
                  hipStream_t st, double *stats = nullptr, int ntaps_src = 27) {
  ConvClasses cs;
  cs.n = 1;
  cs.kseg = 0;
  cs.acc[0] = accumulate;
  cs.xoff[0] = cs.yoff[0] = 0;
  cs.taps[0] = taps;
  return dispatch_conv_classes<T>(x, xv, w, cs, bias, y, yv, B, Cin, Cout, CinP, CoutP, stride, st, stats, ntaps_src);
}

Taps identity_taps(int mirror) {
  Taps t;
  for (int i = 0; i < 27; ++i) t.wt[i] = (signed char)(mirror ? 26 - i : i);
  return t;
}

// index of element (n, k, tap) in the LDS-image-ordered weight array [N/32][K/(2*EPV)][ntaps][2][32][EPV]
__host__ __device__ inline int64_t conv_weight_image_index(int n, int k, int tap, int KP, int ntaps, int EPV) {
  const int64_t chunk2 = k / (2 * EPV);
  const int g = (k / EPV) % 2, e = k % EPV;
  return ((((((int64_t)(n / 32)) * (KP / (2 * EPV)) + chunk2) * ntaps + tap) * 2 + g) * 32 + n % 32) * EPV + e;
}

// ConvTranspose3d k2 s2 weight packing: w_t[ci][co][o] fp32 -> image-ordered wf (N=co, K=ci) and wb (N=ci, K=co),
// 8 "taps" = the 8 output offsets, zero padded
template <typename T>
__global__ void convT_pack_kernel(const float *__restrict__ w, T *__restrict__ wf, T *__restrict__ wb, int Cin, int Cout,
                                  int CinP, int CoutP) {
  constexpr int EPV = Elem<T>::EPV;
  const int CoutN = (CoutP + 31) / 32 * 32, CinN = (CinP + 31) / 32 * 32;
  const int64_t nf = (int64_t)8 * CinP * CoutN, nb = (int64_t)8 * CoutP * CinN;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (nf > nb ? nf : nb);
       i += (int64_t)gridDim.x * blockDim.x) {
    if (i < nf) {
      int ci = (int)(i % CinP), co = (int)((i / CinP) % CoutN), o = (int)(i / ((int64_t)CinP * CoutN));
      st_f<T>(wf + conv_weight_image_index(co, ci, o, CinP, 8, EPV),
              (ci < Cin && co < Cout) ? w[((int64_t)ci * Cout + co) * 8 + o] : 0.f);
    }
    if (i < nb) {   // data-gradient role: N = ci, K = o*CoutP + co (the 8 parity segments concatenated along K), one tap
      int co = (int)(i % CoutP), ci = (int)((i / CoutP) % CinN), o = (int)(i / ((int64_t)CoutP * CinN));
      st_f<T>(wb + conv_weight_image_index(ci, o * CoutP + co, 0, 8 * CoutP, 1, EPV),
              (ci < Cin && co < Cout) ? w[((int64_t)ci * Cout + co) * 8 + o] : 0.f);
    }
  }
}

// conv 3x3x3 weight images: imgF (N=co, K=ci) and imgB (N=ci, K=co) with REAL tap indices
template <typename T>
__global__ void conv_pack_image_kernel(const float *__restrict__ w, T *__restrict__ imgF, T *__restrict__ imgB, int Cin,
                                       int Cout, int CinP, int CoutP) {
  constexpr int EPV = Elem<T>::EPV;
  const int CoutN = (CoutP + 31) / 32 * 32, CinN = (CinP + 31) / 32 * 32;
  const int64_t nf = (int64_t)27 * CinP * CoutN, nb = (int64_t)27 * CoutP * CinN;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (nf > nb ? nf : nb);
       i += (int64_t)gridDim.x * blockDim.x) {
    if (i < nf) {
      int ci = (int)(i % CinP), co = (int)((i / CinP) % CoutN), tap = (int)(i / ((int64_t)CinP * CoutN));
      st_f<T>(imgF + conv_weight_image_index(co, ci, tap, CinP, 27, EPV),
              (ci < Cin && co < Cout) ? w[((int64_t)co * Cin + ci) * 27 + tap] : 0.f);
    }
    if (i < nb) {
      int co = (int)(i % CoutP), ci = (int)((i / CoutP) % CinN), tap = (int)(i / ((int64_t)CoutP * CinN));
      st_f<T>(imgB + conv_weight_image_index(ci, co, tap, CoutP, 27, EPV),
              (ci < Cin && co < Cout) ? w[((int64_t)co * Cin + ci) * 27 + tap] : 0.f);
    }
  }
}

}  // namespace

// w_kmajor: [27][CoutP][CinP] with the K (input-channel) index contiguous; mirror: use tap 26-t.
// upper bound of output tiles per batch sample over all tile shapes the dispatcher may pick
int64_t conv3_mfma_max_tiles(int Do, int Ho, int Wo) {
  const int shapes[6][3] = {{4, 4, 32}, {4, 8, 16}, {2, 8, 8}, {8, 8, 8}, {4, 4, 16}, {4, 8, 8}};
  int64_t best = 0;
  for (auto &t : shapes) {
    int64_t n = (int64_t)cdiv(Do, t[0]) * cdiv(Ho, t[1]) * cdiv(Wo, t[2]);
    best = n > best ? n : best;
  }
  return best;
}

int conv3_fwd_mfma(const void *x, int ldx, const void *w_kmajor, int mirror, const float *bias, void *y, int ldy, int B,
                   int Cin, int Cout, int CinP, int CoutP, int Di, int Hi, int Wi, int stride, int dtype,
                   hipStream_t st, double *stats) {
  const int Do = (Di - 1) / stride + 1, Ho = (Hi - 1) / stride + 1, Wo = (Wi - 1) / stride + 1;
  const View xv = dense_view(B, Di, Hi, Wi, ldx), yv = dense_view(B, Do, Ho, Wo, ldy);
  const Taps taps = identity_taps(mirror);
  if (dtype == DGTTA_F32) {
    if (!operand_ok<float>(x, ldx, Cin, CinP)) return DGTTA_ERR_UNSUPPORTED;
    return dispatch_conv<float>(x, xv, w_kmajor, taps, bias, y, yv, B, Cin, Cout, CinP, CoutP, stride, 0, st, stats);
  }
  if (dtype == DGTTA_BF16) {
    if (!operand_ok<bf16_t>(x, ldx, Cin, CinP)) return DGTTA_ERR_UNSUPPORTED;
    return dispatch_conv<bf16_t>(x, xv, w_kmajor, taps, bias, y, yv, B, Cin, Cout, CinP, CoutP, stride, 0, st, stats);
  }
  return DGTTA_ERR_UNSUPPORTED;
}

// Data gradient of a stride-2 conv: 8 parity classes of the input lattice, each a stride-1 gather of dy with the
// 1/2/4/8 taps that reach that class.  w_kmajor = blob first half [27][CinP][CoutP] (K = co contiguous), real taps.
template <typename T>
static int dgrad_s2(const void *dy, int lddy, const void *w_kmajor, void *dx, int lddx, int B, int Cin, int Cout, int CinP,
                    int CoutP, int Di, int Hi, int Wi, int accumulate, hipStream_t st) {
  if (!operand_ok<T>(dy, lddy, Cout, CoutP)) return DGTTA_ERR_UNSUPPORTED;
  const int Do = (Di - 1) / 2 + 1, Ho = (Hi - 1) / 2 + 1, Wo = (Wi - 1) / 2 + 1;
  const View xv = dense_view(B, Do, Ho, Wo, lddy);
  // even extents: all 8 parity classes have the same shape -> ONE launch, class = blockIdx.z
  ConvClasses cs;
  cs.n = 8;
  cs.kseg = 0;
  View yv;
  for (int p = 0; p < 8; ++p) {
    const int pd = p >> 2, ph = (p >> 1) & 1, pw = p & 1;
    long long off;
    yv = parity_view(Di, Hi, Wi, lddx, pd, ph, pw, &off);
    yv.sb = (long long)Di * Hi * Wi * lddx;
    cs.xoff[p] = 0;
    cs.yoff[p] = off;
    cs.acc[p] = accumulate;
    for (int t = 0; t < 27; ++t) {
      const int k[3] = {t / 9, (t / 3) % 3, t % 3}, par[3] = {pd, ph, pw};
      int real[3];
      bool ok = true;
      for (int a = 0; a < 3; ++a) {
        if (par[a] == 0) {
          ok = ok && (k[a] == 1);
          real[a] = 1;
        } else {
          ok = ok && (k[a] >= 1);
          real[a] = (k[a] == 1) ? 2 : 0;
        }
      }
      cs.taps[p].wt[t] = ok ? (signed char)(real[0] * 9 + real[1] * 3 + real[2]) : (signed char)-1;
    }
  }
  {
    // one pass over dy with all 8 classes accumulated per wave (yv: extent of the dy lattice, strides of the parity view)
    const char *ac = getenv("DGTTA_DGRAD_S2_ALLCLS");      // diagnostic / tests: "0" = the 8-class launch
    if (!(ac && ac[0] == '0') && CinP % 8 == 0) {
      if (yv.W >= 32) return launch_conv_allcls<T, 32, 4, 2>(dy, xv, w_kmajor, cs, dx, yv, B, Cout, Cin, CoutP, CinP, st);
      if (yv.W >= 16) return launch_conv_allcls<T, 16, 2, 4>(dy, xv, w_kmajor, cs, dx, yv, B, Cout, Cin, CoutP, CinP, st);
      return launch_conv_allcls<T, 8, 2, 4>(dy, xv, w_kmajor, cs, dx, yv, B, Cout, Cin, CoutP, CinP, st);
    }
  }
  return dispatch_conv_classes<T>(dy, xv, w_kmajor, cs, nullptr, dx, yv, B, Cout, Cin, CoutP, CinP, 1, st);
}

int conv3_dgrad_s2_mfma(const void *dy, int lddy, const void *w_kmajor, void *dx, int lddx, int B, int Cin, int Cout,
                        int CinP, int CoutP, int Di, int Hi, int Wi, int accumulate, int dtype, hipStream_t st) {
  if (dtype == DGTTA_F32) return dgrad_s2<float>(dy, lddy, w_kmajor, dx, lddx, B, Cin, Cout, CinP, CoutP, Di, Hi, Wi, accumulate, st);
  if (dtype == DGTTA_BF16) return dgrad_s2<bf16_t>(dy, lddy, w_kmajor, dx, lddx, B, Cin, Cout, CinP, CoutP, Di, Hi, Wi, accumulate, st);
  return DGTTA_ERR_UNSUPPORTED;
}

// ConvTranspose3d(k2,s2) forward / data gradient as 8 single-tap launches (one per output offset o).
static size_t n32(int c) { return (size_t)(c + 31) / 32 * 32; }
size_t convT_packed_bytes(int CinP, int CoutP, int dtype) {
  return (size_t)8 * (CinP * n32(CoutP) + CoutP * n32(CinP)) * (dtype == DGTTA_BF16 ? 2 : 4);
}
// image-ordered conv weights (imgF | imgB) appended to the [wf | wb] blob
size_t conv_image_bytes(int CinP, int CoutP, int dtype) {
  return (size_t)27 * (CinP * n32(CoutP) + CoutP * n32(CinP)) * (dtype == DGTTA_BF16 ? 2 : 4);
}
size_t conv_imgB_offset_bytes(int CinP, int CoutP, int dtype) {
  return (size_t)27 * CinP * n32(CoutP) * (dtype == DGTTA_BF16 ? 2 : 4);
}
int conv_pack_images(const float *w_t, void *img, int Cin, int Cout, int CinP, int CoutP, int dtype, hipStream_t st) {
  const int64_t n = (int64_t)27 * (CinP > CoutP ? CinP : CoutP) * n32(CinP > CoutP ? CinP : CoutP);
  const unsigned blocks = (unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  void *imgB = (char *)img + conv_imgB_offset_bytes(CinP, CoutP, dtype);
  if (dtype == DGTTA_F32)
    hipLaunchKernelGGL((conv_pack_image_kernel<float>), dim3(blocks), dim3(256), 0, st, w_t, (float *)img, (float *)imgB, Cin,
                       Cout, CinP, CoutP);
  else
    hipLaunchKernelGGL((conv_pack_image_kernel<bf16_t>), dim3(blocks), dim3(256), 0, st, w_t, (bf16_t *)img, (bf16_t *)imgB,
                       Cin, Cout, CinP, CoutP);
  DG_CHECK_LAUNCH("conv_pack_image_kernel");
  return DGTTA_OK;
}

template <typename T>
static int convT_run(int mode /*0 fwd, 1 dgrad*/, const void *in, int ldin, const float *w_t, const float *bias, void *out,
                     int ldout, void *ws, int B, int Cin, int Cout, int Di, int Hi, int Wi, hipStream_t st) {
  constexpr int EPV = Elem<T>::EPV;
  const int CinP = (Cin + 2 * EPV - 1) / (2 * EPV) * (2 * EPV), CoutP = (Cout + 2 * EPV - 1) / (2 * EPV) * (2 * EPV);
  T *wf = (T *)ws, *wb = wf + (size_t)8 * CinP * n32(CoutP);
  const int64_t n = (int64_t)8 * (CinP > CoutP ? CinP : CoutP) * n32(CinP > CoutP ? CinP : CoutP);
  hipLaunchKernelGGL((convT_pack_kernel<T>), dim3((unsigned)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024)), dim3(256),
                     0, st, w_t, wf, wb, Cin, Cout, CinP, CoutP);
  DG_CHECK_LAUNCH("convT_pack_kernel");
  const int Do = 2 * Di, Ho = 2 * Hi, Wo = 2 * Wi;
  if (mode == 0 ? !operand_ok<T>(in, ldin, Cin, CinP) : !operand_ok<T>(in, ldin, Cout, CoutP)) return DGTTA_ERR_UNSUPPORTED;
  if (mode == 0) {
    // forward: the 8 output offsets write disjoint sub-lattices -> one pointwise launch with 8 classes
    ConvClasses cs;
    cs.n = 8;
    cs.kseg = 0;
    View yv;
    for (int o = 0; o < 8; ++o) {
      long long off;
      yv = parity_view(Do, Ho, Wo, ldout, o >> 2, (o >> 1) & 1, o & 1, &off);
      yv.sb = (long long)Do * Ho * Wo * ldout;
      cs.xoff[o] = 0;
      cs.yoff[o] = off;
      cs.acc[o] = 0;
      for (int t = 0; t < 27; ++t) cs.taps[o].wt[t] = -1;
      cs.taps[o].wt[13] = (signed char)o;
    }
    const View xv = dense_view(B, Di, Hi, Wi, ldin);
    return dispatch_conv_classes<T>(in, xv, wf, cs, bias, out, yv, B, Cin, Cout, CinP, CoutP, 0, st, nullptr, 8);
  }
  // data gradient: dx[v][ci] = sum_o sum_co dout[2v+o][co] w[ci][co][o] = ONE pointwise GEMM with K = 8 x Cout, the 8
  // parity sub-lattices of dout concatenated along K (segment offsets), no read-modify-write passes
  ConvClasses cs;
  cs.n = 1;
  cs.acc[0] = 0;
  cs.xoff[0] = cs.yoff[0] = 0;
  for (int t = 0; t < 27; ++t) cs.taps[0].wt[t] = -1;
  cs.taps[0].wt[13] = 0;
  cs.kseg = CoutP;
  View xv;
  for (int o = 0; o < 8; ++o) {
    long long off;
    xv = parity_view(Do, Ho, Wo, ldin, o >> 2, (o >> 1) & 1, o & 1, &off);
    cs.segoff[o] = off;
  }
  xv.sb = (long long)Do * Ho * Wo * ldin;
  const View yv = dense_view(B, Di, Hi, Wi, ldout);
  return dispatch_conv_classes<T>(in, xv, wb, cs, nullptr, out, yv, B, Cout, Cin, 8 * CoutP, CinP, 0, st, nullptr, 1);
}

int convT_fwd_mfma(const void *x, int ldx, const float *w_t, const float *bias, void *out, int ldo, void *ws, int B, int Cin,
                   int Cout, int Di, int Hi, int Wi, int dtype, hipStream_t st) {
  if (dtype == DGTTA_F32) return convT_run<float>(0, x, ldx, w_t, bias, out, ldo, ws, B, Cin, Cout, Di, Hi, Wi, st);
  if (dtype == DGTTA_BF16) return convT_run<bf16_t>(0, x, ldx, w_t, bias, out, ldo, ws, B, Cin, Cout, Di, Hi, Wi, st);
  return DGTTA_ERR_UNSUPPORTED;
}
int convT_dgrad_mfma(const void *dout, int lddo, const float *w_t, void *dx, int lddx, void *ws, int B, int Cin, int Cout,
                     int Di, int Hi, int Wi, int dtype, hipStream_t st) {
  if (dtype == DGTTA_F32) return convT_run<float>(1, dout, lddo, w_t, nullptr, dx, lddx, ws, B, Cin, Cout, Di, Hi, Wi, st);
  if (dtype == DGTTA_BF16) return convT_run<bf16_t>(1, dout, lddo, w_t, nullptr, dx, lddx, ws, B, Cin, Cout, Di, Hi, Wi, st);
  return DGTTA_ERR_UNSUPPORTED;
}

namespace {

// =====================================================================================================================
// Weight gradient on the matrix cores (stride 1):  dW[tap][ci][co] = sum_v x[v + tap - 1][ci] * dy[v][co]
//   GEMM view: M = ci, N = co, K = voxels (runs of 32 along W).  A workgroup owns one 32(ci) x 32(co) channel tile and a
//   column of the volume: TH=4 output rows x 32 voxels, D range [d0,d1); its 4 waves own the four 16x16 sub-blocks and
//   keep all 27 tap accumulators (27 x f32x4) in registers while the column is swept slice by slice.
//   MFMA: bf16 v_mfma_f32_16x16x32_bf16 (K=32 = one voxel row per instruction), fp32 v_mfma_f32_16x16x4_f32 x8.
//   LDS: x and dy are staged TRANSPOSED (channel-major, 16-byte runs of consecutive voxels) with an in-register
//   EPV x EPV transpose, as a ring of 4 x-slices (halo of 1 in D and H) and 2 dy-slices; global loads for slice d+2 are
//   issued before the MFMAs of slice d and written to LDS after them.  The W shift of a tap (kw-1) is a funnel shift
//   of the aligned 16-byte run plus the next run's first dword(s).  Layout [row][run][channel][16 B] makes the 16
//   lanes of a k-group read consecutive 16-byte slots (no bank conflicts).
//   Each workgroup writes one fp32 partial slab; wgrad_reduce_kernel sums slabs in fixed order (deterministic).
// =====================================================================================================================
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

template <typename T>
struct WG {
  static constexpr int EPV = Elem<T>::EPV;
  static constexpr int GC = 32 / EPV;            // channel groups (of EPV channels) per 32-channel tile
  static constexpr int NCH_Y = 32 / EPV;         // voxel runs per dy row
  static constexpr int NCH_X = 32 / EPV + 1;     // voxel runs per x row; run c covers wx = EPV*c - 1 .. EPV*c + EPV - 2
  static constexpr int TH = 4, XR = TH + 2;
  static constexpr int XSLOT = XR * NCH_X * 32;  // uint4 per x slice
  static constexpr int YSLOT = TH * NCH_Y * 32;
  static constexpr int NUX = XR * NCH_X * GC, NUY = TH * NCH_Y * GC, NU = NUX + NUY;
  static constexpr int ROUNDS = (NU + 255) / 256;
  static constexpr size_t LDS_BYTES = (size_t)(4 * XSLOT + 2 * YSLOT) * 16;
  static constexpr int NSTEP = 32 / (4 * EPV);   // MFMA k-steps per voxel row (bf16 1, fp32 2)
  // LDS slot (in uint4) of channel c (0..31) of voxel run `run` in row `row`.  Within a channel group the EPV slots are
  // XOR-swizzled so that the 8 lanes of a ds_write_b128 group (which differ in channel group / run parity and all write
  // the same in-group channel j) hit 8 different 16-byte bank slots; readers apply the same map (still one distinct
  // slot per lane of a 16-lane read group).
  __device__ static __forceinline__ int slot(int row, int run, int nruns, int c) {
    const int cg = c / EPV, j = c % EPV;
    const int sw = (EPV == 8) ? ((cg | ((run & 1) << 2)) & 7) : ((cg >> 1) & 3);
    return (row * nruns + run) * 32 + cg * EPV + (j ^ sw);
  }
};

template <typename T>
__device__ __forceinline__ void transpose_unit(const uint4 *in, uint4 *out);
template <>
__device__ __forceinline__ void transpose_unit<float>(const uint4 *in, uint4 *out) {   // 4 voxels x 4 channels
  out[0] = make_uint4(in[0].x, in[1].x, in[2].x, in[3].x);
  out[1] = make_uint4(in[0].y, in[1].y, in[2].y, in[3].y);
  out[2] = make_uint4(in[0].z, in[1].z, in[2].z, in[3].z);
  out[3] = make_uint4(in[0].w, in[1].w, in[2].w, in[3].w);
}
__device__ __forceinline__ unsigned pack_lo(unsigned a, unsigned b) { return (a & 0xffffu) | (b << 16); }
__device__ __forceinline__ unsigned pack_hi(unsigned a, unsigned b) { return (a >> 16) | (b & 0xffff0000u); }
template <>
__device__ __forceinline__ void transpose_unit<bf16_t>(const uint4 *in, uint4 *out) {  // 8 voxels x 8 channels
#define TR_PAIR(c, fld)                                                                                      \
  out[c] = make_uint4(pack_lo(in[0].fld, in[1].fld), pack_lo(in[2].fld, in[3].fld), pack_lo(in[4].fld, in[5].fld), \
                      pack_lo(in[6].fld, in[7].fld));                                                        \
  out[c + 1] = make_uint4(pack_hi(in[0].fld, in[1].fld), pack_hi(in[2].fld, in[3].fld),                       \
                          pack_hi(in[4].fld, in[5].fld), pack_hi(in[6].fld, in[7].fld));
  TR_PAIR(0, x) TR_PAIR(2, y) TR_PAIR(4, z) TR_PAIR(6, w)
#undef TR_PAIR
}

// A operand for tap column kw from the aligned run `c` and the next run's first dwords (e0, e1)
template <typename T>
__device__ __forceinline__ uint4 shift_run(const uint4 &c, unsigned e0, unsigned e1, int kw);
template <>
__device__ __forceinline__ uint4 shift_run<bf16_t>(const uint4 &c, unsigned e0, unsigned, int kw) {
  if (kw == 0) return c;
  if (kw == 2) return make_uint4(c.y, c.z, c.w, e0);
  return make_uint4((c.x >> 16) | (c.y << 16), (c.y >> 16) | (c.z << 16), (c.z >> 16) | (c.w << 16),
                    (c.w >> 16) | (e0 << 16));
}
template <>
__device__ __forceinline__ uint4 shift_run<float>(const uint4 &c, unsigned e0, unsigned e1, int kw) {
  if (kw == 0) return c;
  if (kw == 1) return make_uint4(c.y, c.z, c.w, e0);
  return make_uint4(c.z, c.w, e0, e1);
}

template <typename T>
__device__ __forceinline__ void mfma16(const uint4 &a, const uint4 &b, f32x4_t &acc);
template <>
__device__ __forceinline__ void mfma16<bf16_t>(const uint4 &a, const uint4 &b, f32x4_t &acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0,
                                                0, 0);
}
template <>
__device__ __forceinline__ void mfma16<float>(const uint4 &a, const uint4 &b, f32x4_t &acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
}

// Up to 8 independent classes per launch (blockIdx.z): operand offsets + tap mask per class, one slab set per class.
struct WgradClasses {
  int n;
  unsigned mask[8];
  long long xoff[8], yoff[8];
};
struct RealTaps {
  Taps t[8];
};

// x: view xv (input lattice of the virtual stride-1 problem), dy: view yv (output lattice; tiles run over it).
// mask bit t set = virtual tap t is accumulated.
template <typename T, int ABL = 0>   // ABL: diagnostic ablation (1 no global loads, 2 no LDS stores, 3 no MFMA); 0 = product
__global__ __launch_bounds__(256, (sizeof(T) == 2 ? 2 : 1)) void conv3_wgrad_mfma_kernel(const T *__restrict__ x, View xv, const T *__restrict__ dy,
                                                               View yv, float *__restrict__ slabs, int Cin, int Cout,
                                                               int tilesW, int tilesH, int nsd, int DR, int cobs,
                                                               WgradClasses wc) {
  const int cls = blockIdx.z;
  x += wc.xoff[cls];
  dy += wc.yoff[cls];
  const unsigned tapmask = wc.mask[cls];
  const int D = yv.D, H = yv.H, W = yv.W;
  typedef WG<T> C;
  constexpr int EPV = C::EPV;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint4 *sX = reinterpret_cast<uint4 *>(smem);                 // [4][XR][NCH_X][32]
  uint4 *sY = sX + 4 * C::XSLOT;                               // [2][TH][NCH_Y][32]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, kg = lane >> 4;
  const int cih = wave >> 1, coh = wave & 1;

  int t = blockIdx.x;
  const int tw = t % tilesW;
  t /= tilesW;
  const int th = t % tilesH;
  t /= tilesH;
  const int ds = t % nsd;
  const int b = t / nsd;
  const int cib = blockIdx.y / cobs, cob = blockIdx.y % cobs;
  const int h0 = th * C::TH, w0 = tw * 32;
  const int d_begin = ds * DR, d_end = (d_begin + DR < D) ? d_begin + DR : D;
  const T *xb = x + b * xv.sb;
  const T *yb0 = dy + b * yv.sb;
  const int cin_lim = (Cin + EPV - 1) / EPV * EPV;

  uint4 stg[C::ROUNDS][EPV];

  // all loads of a slice are issued back to back: unconditional loads from a clamped address + select (a conditional
  // load makes hipcc branch and wait per element, which serialises the round trips)
  auto load_units = [&](int dx_slice, bool do_x, int dy_slice, bool do_y) {
#pragma unroll
    for (int rd = 0; rd < C::ROUNDS; ++rd) {
      const int u = tid + rd * 256;
      const bool is_x = u < C::NUX;
      const int v = is_x ? u : u - C::NUX;
      const int nch = is_x ? C::NCH_X : C::NCH_Y;
      const int cg = v % C::GC, ch = (v / C::GC) % nch, row = v / (C::GC * nch);
      const View &vw = is_x ? xv : yv;
      const T *bp = is_x ? xb : yb0;
      const int gd = is_x ? dx_slice : dy_slice, gh = is_x ? h0 - 1 + row : h0 + row;
      const int c = (is_x ? cib : cob) * 32 + cg * EPV;
      const bool rowok = (ABL != 1) && u < C::NU && (is_x ? do_x : do_y) && (unsigned)gd < (unsigned)vw.D &&
                         (unsigned)gh < (unsigned)vw.H && c < (is_x ? cin_lim : Cout);
      const T *base = bp + (rowok ? gd * vw.sd + gh * vw.sh + c : 0);
      const int gw0 = w0 + EPV * ch - (is_x ? 1 : 0);
#pragma unroll
      for (int j = 0; j < EPV; ++j) {
        const int gw = gw0 + j;
        const bool ok = rowok && (unsigned)gw < (unsigned)vw.W;
        const uint4 val = *reinterpret_cast<const uint4 *>(base + (ok ? gw * vw.sw : 0));
        stg[rd][j] = ok ? val : make_uint4(0, 0, 0, 0);
      }
    }
  };
  auto store_units = [&](int xslot, bool do_x, int yslot, bool do_y) {
#pragma unroll
    for (int rd = 0; rd < C::ROUNDS; ++rd) {
      const int u = tid + rd * 256;
      uint4 o[EPV];
      if (ABL == 2) {
        asm volatile("" ::"v"(stg[rd][0].x));
        continue;
      }
      if (u < C::NUX) {
        if (!do_x) continue;
        transpose_unit<T>(stg[rd], o);
        const int cg = u % C::GC, ch = (u / C::GC) % C::NCH_X, row = u / (C::GC * C::NCH_X);
        uint4 *dst = sX + xslot * C::XSLOT;
#pragma unroll
        for (int j = 0; j < EPV; ++j) dst[C::slot(row, ch, C::NCH_X, cg * EPV + j)] = o[j];
      } else if (u < C::NU) {
        if (!do_y) continue;
        transpose_unit<T>(stg[rd], o);
        const int v = u - C::NUX;
        const int cg = v % C::GC, ch = (v / C::GC) % C::NCH_Y, row = v / (C::GC * C::NCH_Y);
        uint4 *dst = sY + yslot * C::YSLOT;
#pragma unroll
        for (int j = 0; j < EPV; ++j) dst[C::slot(row, ch, C::NCH_Y, cg * EPV + j)] = o[j];
      }
    }
  };

  f32x4_t acc[27];
#pragma unroll
  for (int i = 0; i < 27; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  unsigned long long tseg[6] = {0, 0, 0, 0, 0, 0}, tprev = 0;     // ABL 6: cycle stamps per segment (diagnostic)
  auto stamp = [&](int k) {
    if (ABL == 6) {
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      __builtin_amdgcn_sched_barrier(0);
      tseg[k] += t - tprev;
      tprev = t;
    }
  };
  if (ABL == 6) tprev = __builtin_amdgcn_s_memtime();

  // prologue: x slices d_begin-1, d_begin, d_begin+1 and dy slice d_begin
  load_units(d_begin - 1, true, d_begin, true);
  store_units((d_begin - 1) & 3, true, d_begin & 1, true);
  load_units(d_begin, true, 0, false);
  store_units(d_begin & 3, true, 0, false);
  load_units(d_begin + 1, true, 0, false);
  store_units((d_begin + 1) & 3, true, 0, false);
  __syncthreads();
  stamp(0);                                   // prologue

  for (int d = d_begin; d < d_end; ++d) {
    const bool more = d + 1 < d_end;
    load_units(d + 2, more, d + 1, more);     // in flight during the MFMAs below
    stamp(1);                                 // load issue
    const uint4 *yb = sY + (d & 1) * C::YSLOT;
#pragma unroll
    for (int oh = 0; oh < C::TH; ++oh) {
#pragma unroll
      for (int stp = 0; stp < C::NSTEP; ++stp) {
        const int run = stp * 4 + kg;
        const uint4 bf = yb[C::slot(oh, run, C::NCH_Y, coh * 16 + m)];
#pragma unroll
        for (int kd = 0; kd < 3; ++kd) {
          const uint4 *xs = sX + ((d + kd - 1) & 3) * C::XSLOT;
#pragma unroll
          for (int kh = 0; kh < 3; ++kh) {
            if (((tapmask >> (kd * 9 + kh * 3)) & 7u) == 0) continue;      // wave-uniform: no tap of this (kd,kh) wanted
            const uint4 c0 = xs[C::slot(oh + kh, run, C::NCH_X, cih * 16 + m)];
            const uint2 ex = *reinterpret_cast<const uint2 *>(xs + C::slot(oh + kh, run + 1, C::NCH_X, cih * 16 + m));
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
              if ((tapmask >> (kd * 9 + kh * 3 + kw)) & 1u) {
                if (ABL == 3) {
                  const uint4 a_ = shift_run<T>(c0, ex.x, ex.y, kw);
                  acc[kd * 9 + kh * 3 + kw][0] += __uint_as_float(a_.x ^ bf.x);
                } else {
                  mfma16<T>(shift_run<T>(c0, ex.x, ex.y, kw), bf, acc[kd * 9 + kh * 3 + kw]);
                }
              }
          }
        }
      }
    }
    stamp(2);                                 // MFMA loop
    store_units((d + 2) & 3, more, (d + 1) & 1, more);
    stamp(3);                                 // wait for loads + transpose + LDS writes
    __syncthreads();
    stamp(4);                                 // barrier
  }

  // partial slab [27][32 ci][32 co]; C/D map of the 16x16 MFMA: col = lane&15, row = (lane>>4)*4 + reg
  float *slab = slabs + (((int64_t)cls * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * (27 * 1024);
#pragma unroll
  for (int tap = 0; tap < 27; ++tap)
#pragma unroll
    for (int q = 0; q < 4; ++q) slab[(tap * 32 + cih * 16 + kg * 4 + q) * 32 + coh * 16 + m] = acc[tap][q];
  if (ABL == 6) {
    stamp(5);                                 // slab write issue
    __syncthreads();
    if (lane == 0)
      for (int k = 0; k < 6; ++k) slab[27 * 1024 - 64 + wave * 8 + k] = (float)tseg[k];      // overwrites a slab corner
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// bf16 weight gradient with hardware-transposed operand reads (stride 1, all 27 taps).  Same decomposition and slab
// format as conv3_wgrad_mfma_kernel, but:
//   * x / dy slices stay VOXEL-major in LDS ([row][voxel][32 channels = 64 B]) and are filled by LDS-DMA
//     (global_load_lds_dwordx4: 16 voxels x 64 B per instruction, no staging registers, no register transposes, no
//     ds_write); the K-contiguous MFMA operands (8 consecutive voxels of one channel per lane) come out of
//     ds_read_b64_tr_b16, so a tap's W shift is an address offset instead of a funnel shift per operand;
//   * MFMA 32x32x16: a wave owns the whole 32(ci) x 32(co) tile for 7 (or 6) of the 27 taps (tap = wave + 4 i), the dy
//     fragment of a (row, 16-voxel step) is shared by its taps; per MFMA: 2 transposed reads, ~1 VALU, no shifts.
// (The predecessor spent its issue slots on funnel shifts and 8x8 register transposes: measured 2.5x the MFMA time.)
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((address_space(3))) s16x4_t lds_s16x4_t;

struct WT {
  static constexpr int TH = 4, XR = TH + 2;
  static constexpr int XW = 36;                         // voxels per x row in LDS (34 used)
  static constexpr int X_ROW_B = XW * 64, X_SLICE_B = XR * X_ROW_B;
  static constexpr int Y_ROW_B = 32 * 64, Y_SLICE_B = TH * Y_ROW_B;
  static constexpr int LDS_BYTES = 4 * X_SLICE_B + 2 * Y_SLICE_B;
  static constexpr int NPX = XR * 3, NPY = TH * 2, NP = NPX + NPY;      // DMA pieces per slice
};

__device__ __forceinline__ bf16x8_t tr_operand(const unsigned char *p) {
  // two 4-voxel transposed reads = 8 consecutive voxels (k) of this lane's channel
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t *)p);
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t *)(p + 4 * 64));
  typedef __attribute__((ext_vector_type(8))) short s16x8_t;
  const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}

// CLS: class launch (blockIdx.z selects operand offsets and a tap subset, as in conv3_wgrad_mfma_kernel): the set taps
// are dealt round-robin to the 4 waves, slots beyond a wave's share are skipped with wave-uniform branches.
template <int ABL = 0, bool CLS = false>
__global__ __launch_bounds__(256, 2) void conv3_wgrad_tr_kernel(const bf16_t *__restrict__ x, View xv,
                                                                const bf16_t *__restrict__ dy, View yv,
                                                                float *__restrict__ slabs, int Cin, int Cout, int tilesW,
                                                                int tilesH, int nsd, int DR, int cobs, WgradClasses wc) {
  const int cls = CLS ? blockIdx.z : 0;
  if (CLS) {
    x += wc.xoff[cls];
    dy += wc.yoff[cls];
  }
  const int D = yv.D, H = yv.H, W = yv.W;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char *sX = smem;                                   // ring of 4 x slices
  unsigned char *sY = smem + 4 * WT::X_SLICE_B;               // ring of 2 dy slices
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  int t = blockIdx.x;
  const int tw = t % tilesW;
  t /= tilesW;
  const int th = t % tilesH;
  t /= tilesH;
  const int ds = t % nsd;
  const int b = t / nsd;
  const int cib = blockIdx.y / cobs, cob = blockIdx.y % cobs;
  const int h0 = th * WT::TH, w0 = tw * 32;
  const int d_begin = ds * DR, d_end = (d_begin + DR < D) ? d_begin + DR : D;
  const bf16_t *xb = x + b * xv.sb + cib * 32;
  const bf16_t *yb = dy + b * yv.sb + cob * 32;
  const int cin_lim = (Cin + 7) / 8 * 8;

  // DMA of one slice: piece idx (wave-uniform) -> x row r (3 pieces: voxels 0-15, 16-31, 32-33) or dy row (2 pieces);
  // lane l of a piece = voxel 16*pi + l/4, 16-byte channel chunk l%4
  const int l_vox = lane >> 2, l_chunk = lane & 3;
  constexpr int NPW = (WT::NP + 3) / 4;        // pieces per wave and slice
  auto issue_piece = [&](int i, int xd, int xslot, bool do_x, int yd, int yslot, bool do_y) __attribute__((always_inline)) {
    const int idx = wave + 4 * i;
    if (idx < WT::NPX) {
      if (!do_x) return;
      const int r = idx / 3, pi = idx % 3;
      if (pi == 2 && lane >= 8) return;
      const int gh = h0 - 1 + r, wx = 16 * pi + l_vox, gw = w0 - 1 + wx;
      const bool ok = (unsigned)xd < (unsigned)xv.D && (unsigned)gh < (unsigned)xv.H && (unsigned)gw < (unsigned)xv.W &&
                      cib * 32 + l_chunk * 8 < cin_lim;
      const void *src = ok ? (const void *)(xb + xd * xv.sd + gh * xv.sh + gw * xv.sw + l_chunk * 8) : (const void *)&g_zero16;
      if (ABL == 1) return;
      dma16_to_lds(src, lds_addr_of(sX + xslot * WT::X_SLICE_B + r * WT::X_ROW_B + pi * 1024));
    } else if (idx < WT::NP) {
      if (!do_y) return;
      const int j = idx - WT::NPX, r = j / 2, pi = j % 2;
      const int gh = h0 + r, gw = w0 + 16 * pi + l_vox;
      const bool ok = (unsigned)yd < (unsigned)D && gh < H && gw < W && cob * 32 + l_chunk * 8 < Cout;
      const void *src = ok ? (const void *)(yb + yd * yv.sd + gh * yv.sh + gw * yv.sw + l_chunk * 8) : (const void *)&g_zero16;
      if (ABL == 1) return;
      dma16_to_lds(src, lds_addr_of(sY + yslot * WT::Y_SLICE_B + r * WT::Y_ROW_B + pi * 1024));
    }
  };
  auto issue_slice = [&](int xd, int xslot, bool do_x, int yd, int yslot, bool do_y) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NPW; ++i) issue_piece(i, xd, xslot, do_x, yd, yslot, do_y);
  };

  // transposed-read lane address inside a 16-voxel x 32-channel block (64-byte voxel rows): group lane 4q+p supplies
  // voxel row q, channels 4p..4p+3 of the group's 16 channels; groups 0/1 = channels 0-15 / 16-31, lanes >= 32 = k 8..15
  const int lane_off = ((lane >> 5) * 8 + ((lane & 15) >> 2)) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8;

  // this wave's taps: tap = wave + 4 i (i < 7) -- with classes, the (wave + 4 i)-th set bit of the class mask;
  // wave-uniform offsets of the x operand
  int tap_id[7], tap_kd[7], tap_off[7];
  int ntap_w = 7;
  if (CLS) {
    const unsigned mask = wc.mask[cls];
    ntap_w = 0;
#pragma unroll
    for (int i = 0; i < 7; ++i) tap_id[i] = 26;
    int seen = 0;
    for (int tp = 0; tp < 27; ++tp)
      if ((mask >> tp) & 1u) {
        if ((seen & 3) == wave) {
#pragma unroll
          for (int i = 0; i < 7; ++i)
            if (i == (seen >> 2)) tap_id[i] = tp;
          ntap_w = (seen >> 2) + 1;
        }
        ++seen;
      }
  } else {
#pragma unroll
    for (int i = 0; i < 7; ++i) tap_id[i] = wave + 4 * i < 27 ? wave + 4 * i : 26;
  }
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int tc = tap_id[i];
    tap_kd[i] = tc / 9;
    tap_off[i] = ((tc / 3) % 3) * WT::X_ROW_B + (tc % 3) * 64;
  }

  f32x16_t acc[7];
#pragma unroll
  for (int i = 0; i < 7; ++i)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;

  // prologue: x slices d_begin-1, d_begin, d_begin+1 and dy slice d_begin
  issue_slice(d_begin - 1, (d_begin - 1) & 3, true, d_begin, d_begin & 1, true);
  issue_slice(d_begin, d_begin & 3, true, 0, 0, false);
  issue_slice(d_begin + 1, (d_begin + 1) & 3, true, 0, 0, false);
  dma_wait_all();
  lds_barrier();

  for (int d = d_begin; d < d_end; ++d) {
    const bool more = d + 1 < d_end;
    const unsigned char *ys = sY + (d & 1) * WT::Y_SLICE_B + lane_off;
    int slice_off[3];
#pragma unroll
    for (int kd = 0; kd < 3; ++kd) slice_off[kd] = ((d + kd - 1) & 3) * WT::X_SLICE_B;
#pragma unroll
    for (int oh = 0; oh < WT::TH; ++oh) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8_t bfr = tr_operand(ys + oh * WT::Y_ROW_B + ks * 1024);
        // one DMA piece of the next slices per (row, k-step): a burst at the top of the slice would block this wave
        // until the memory pipeline has taken all of them
        if (oh * 2 + ks < NPW) issue_piece(oh * 2 + ks, d + 2, (d + 2) & 3, more, d + 1, (d + 1) & 1, more);
        // all 7 operand reads first, then 7 MFMAs (wave 3's seventh slot repeats tap 26 into a discarded accumulator,
        // so the code is branch-free and the reads pipeline ahead of the matrix instructions)
        bf16x8_t afr[7];
#pragma unroll
        for (int i = 0; i < 7; ++i) {
          if (CLS && i >= ntap_w) continue;      // wave-uniform
          const int so = (tap_kd[i] == 0 ? slice_off[0] : (tap_kd[i] == 1 ? slice_off[1] : slice_off[2])) + tap_off[i];
          afr[i] = tr_operand(sX + lane_off + so + oh * WT::X_ROW_B + ks * 1024);
        }
#pragma unroll
        for (int i = 0; i < 7; ++i) {
          if (CLS && i >= ntap_w) continue;
          if (ABL == 3) acc[i][0] += (float)afr[i][0] * (float)bfr[1];
          else acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[i], bfr, acc[i], 0, 0, 0);
        }
      }
    }
    dma_wait_all();
    lds_barrier();
  }

  // partial slab [27][32 ci][32 co]; C/D map of the 32x32 MFMA: col = lane&31 (co), row = (q&3) + 8(q>>2) + 4(lane>>5) (ci)
  float *slab = slabs + (((int64_t)cls * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * (27 * 1024);
  const int co = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int tap = tap_id[i];
    if (CLS ? i < ntap_w : wave + 4 * i < 27) {
#pragma unroll
      for (int q = 0; q < 16; ++q) slab[(tap * 32 + (q & 3) + 8 * (q >> 2) + 4 * hh) * 32 + co] = acc[i][q];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// 8-wave variant of conv3_wgrad_tr_kernel for Cout >= 64: a workgroup owns a 32(ci) x 64(co) channel tile, so the x tile
// (the larger one, with its halo) is staged once for two output-channel blocks: 58 instead of 94 DMA bytes per MFMA
// (the 4-wave kernel sits on the ~11 B/clk/CU fill rate).  Wave w owns taps w, w+8, w+16, w+24 (27 of the 32 slots are
// real) for both blocks: an x fragment feeds 2 MFMAs, 1.5 transposed reads per MFMA instead of 2.3.
struct WT8 {
  static constexpr int Y_ROW_B = 32 * 128, Y_SLICE_B = WT::TH * Y_ROW_B;       // dy rows of 64 channels
  static constexpr int LDS_BYTES = 4 * WT::X_SLICE_B + 2 * Y_SLICE_B;
  static constexpr int NPY = WT::TH * 4;                                        // 8 voxels x 128 B per piece
  static constexpr int NP = WT::NPX + NPY;
};

__global__ __launch_bounds__(512, 1) void conv3_wgrad_tr8_kernel(const bf16_t *__restrict__ x, View xv,
                                                                 const bf16_t *__restrict__ dy, View yv,
                                                                 float *__restrict__ slabs, int Cin, int Cout, int tilesW,
                                                                 int tilesH, int nsd, int DR, int cobs) {
  const int D = yv.D, H = yv.H, W = yv.W;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char *sX = smem;
  unsigned char *sY = smem + 4 * WT::X_SLICE_B;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int t = blockIdx.x;
  const int tw = t % tilesW;
  t /= tilesW;
  const int th = t % tilesH;
  t /= tilesH;
  const int ds = t % nsd;
  const int b = t / nsd;
  const int cobs2 = (cobs + 1) / 2;
  const int cib = blockIdx.y / cobs2, cob2 = blockIdx.y % cobs2;          // channel-block pair (2 cob2, 2 cob2 + 1)
  const int h0 = th * WT::TH, w0 = tw * 32;
  const int d_begin = ds * DR, d_end = (d_begin + DR < D) ? d_begin + DR : D;
  const bf16_t *xb = x + b * xv.sb + cib * 32;
  const bf16_t *yb = dy + b * yv.sb + cob2 * 64;
  const int cin_lim = (Cin + 7) / 8 * 8;

  constexpr int NPW = (WT8::NP + 7) / 8;
  auto issue_piece = [&](int i, int xd, int xslot, int yd, int yslot, bool more, bool with_y = true) __attribute__((always_inline)) {
    const int idx = wave + 8 * i;
    if (!more) return;
    if (idx < WT::NPX) {
      const int r = idx / 3, pi = idx % 3;
      if (pi == 2 && lane >= 8) return;
      const int l_vox = lane >> 2, l_chunk = lane & 3;
      const int gh = h0 - 1 + r, gw = w0 - 1 + 16 * pi + l_vox;
      const bool ok = (unsigned)xd < (unsigned)xv.D && (unsigned)gh < (unsigned)xv.H && (unsigned)gw < (unsigned)xv.W &&
                      cib * 32 + l_chunk * 8 < cin_lim;
      const void *src = ok ? (const void *)(xb + xd * xv.sd + gh * xv.sh + gw * xv.sw + l_chunk * 8) : (const void *)&g_zero16;
      dma16_to_lds(src, lds_addr_of(sX + xslot * WT::X_SLICE_B + r * WT::X_ROW_B + pi * 1024));
    } else if (idx < WT8::NP && with_y) {
      const int j = idx - WT::NPX, r = j / 4, pi = j % 4;
      const int l_vox = lane >> 3, l_chunk = lane & 7;       // 8 voxels x 8 chunks of 16 B
      const int gh = h0 + r, gw = w0 + 8 * pi + l_vox;
      const bool ok = (unsigned)yd < (unsigned)D && gh < H && gw < W && cob2 * 64 + l_chunk * 8 < Cout;
      const void *src = ok ? (const void *)(yb + yd * yv.sd + gh * yv.sh + gw * yv.sw + l_chunk * 8) : (const void *)&g_zero16;
      dma16_to_lds(src, lds_addr_of(sY + yslot * WT8::Y_SLICE_B + r * WT8::Y_ROW_B + pi * 1024));
    }
  };

  const int kq = (lane >> 5) * 8 + ((lane & 15) >> 2), cpart = ((lane >> 4) & 1) * 32 + (lane & 3) * 8;
  const int lane_off_x = kq * 64 + cpart, lane_off_y = kq * 128 + cpart;

  int tap_kd[4], tap_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int tc = wave + 8 * i < 27 ? wave + 8 * i : 26;
    tap_kd[i] = tc / 9;
    tap_off[i] = ((tc / 3) % 3) * WT::X_ROW_B + (tc % 3) * 64;
  }
  f32x16_t acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][c][q] = 0.f;

  // prologue: x slices d_begin-1, d_begin, d_begin+1 and dy slice d_begin
#pragma unroll
  for (int i = 0; i < NPW; ++i) issue_piece(i, d_begin - 1, (d_begin - 1) & 3, d_begin, d_begin & 1, true);
#pragma unroll
  for (int sl = 0; sl <= 1; ++sl)
#pragma unroll
    for (int i = 0; i < (WT::NPX + 7) / 8; ++i) issue_piece(i, d_begin + sl, (d_begin + sl) & 3, 0, 0, true, false);
  dma_wait_all();
  lds_barrier();
  typedef __attribute__((ext_vector_type(8))) short s16x8_t;
  for (int d = d_begin; d < d_end; ++d) {
    const bool more = d + 1 < d_end;
    const unsigned char *ys = sY + (d & 1) * WT8::Y_SLICE_B + lane_off_y;
    int slice_off[3];
#pragma unroll
    for (int kd = 0; kd < 3; ++kd) slice_off[kd] = ((d + kd - 1) & 3) * WT::X_SLICE_B;
#pragma unroll
    for (int oh = 0; oh < WT::TH; ++oh) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        if (oh * 2 + ks < NPW) issue_piece(oh * 2 + ks, d + 2, (d + 2) & 3, d + 1, (d + 1) & 1, more);
        bf16x8_t bfr[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const unsigned char *pb = ys + oh * WT8::Y_ROW_B + ks * 16 * 128 + c * 64;
          const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t *)pb);
          const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t *)(pb + 4 * 128));
          const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          bfr[c] = __builtin_bit_cast(bf16x8_t, v);
        }
        bf16x8_t afr[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int so = (tap_kd[i] == 0 ? slice_off[0] : (tap_kd[i] == 1 ? slice_off[1] : slice_off[2])) + tap_off[i];
          afr[i] = tr_operand(sX + lane_off_x + so + oh * WT::X_ROW_B + ks * 1024);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int c = 0; c < 2; ++c) acc[i][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[i], bfr[c], acc[i][c], 0, 0, 0);
      }
    }
    dma_wait_all();
    lds_barrier();
  }
  const int co = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int cob = 2 * cob2 + c;
    if (cob >= cobs) continue;
    float *slab = slabs + (((int64_t)cib * cobs + cob) * gridDim.x + blockIdx.x) * (27 * 1024);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int tap = wave + 8 * i;
      if (tap < 27) {
#pragma unroll
        for (int q = 0; q < 16; ++q) slab[(tap * 32 + (q & 3) + 8 * (q >> 2) + 4 * hh) * 32 + co] = acc[i][c][q];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// bf16 weight gradient of a STRIDE-2 conv in one pass:  dW[tap][ci][co] = sum_vo x[2 vo + tap - 1][ci] * dy[vo][co].
// Same scheme as conv3_wgrad_tr_kernel (LDS-DMA staging, ds_read_b64_tr_b16 operands, 7 taps per wave, slab output) with
// the x tile kept at FULL resolution: output tile 2 rows x 16 voxels needs x rows 2h0-1 .. 2h0+3 and voxels 2w0-1 ..
// 2w0+31; output slice d needs x slices 2d-1, 2d, 2d+1 (ring of 5: 3 live + 2 arriving).  The transposed read takes one
// row address per lane, so "every second voxel" is just a 128-byte row stride of the operand block.  dy is read once and x
// once (+ halo), instead of 8 parity-class passes that each re-read dy and gathered x with half-used cache lines.
struct WT2 {
  static constexpr int TH = 2, TWO = 16;                // output rows / voxels per tile
  static constexpr int XR = 2 * TH + 1, XW = 36;        // x rows per slice, voxels per x row in LDS (33 used)
  static constexpr int X_ROW_B = XW * 64, X_SLICE_B = XR * X_ROW_B;
  static constexpr int Y_ROW_B = TWO * 64, Y_SLICE_B = TH * Y_ROW_B;
  static constexpr int NXS = 5;                         // x ring slots
  static constexpr int LDS_BYTES = NXS * X_SLICE_B + 2 * Y_SLICE_B;
  static constexpr int NPX1 = XR * 3;                   // DMA pieces per x slice (16 + 16 + 1 voxels per row)
  static constexpr int NP = 2 * NPX1 + TH;              // pieces per output slice: two x slices + one dy slice
};

__global__ __launch_bounds__(256, 2) void conv3_wgrad_tr_s2_kernel(const bf16_t *__restrict__ x, View xv,
                                                                   const bf16_t *__restrict__ dy, View yv,
                                                                   float *__restrict__ slabs, int Cin, int Cout, int tilesW,
                                                                   int tilesH, int nsd, int DR, int cobs) {
  const int D = yv.D, H = yv.H, W = yv.W;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char *sX = smem;
  unsigned char *sY = smem + WT2::NXS * WT2::X_SLICE_B;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  int t = blockIdx.x;
  const int tw = t % tilesW;
  t /= tilesW;
  const int th = t % tilesH;
  t /= tilesH;
  const int ds = t % nsd;
  const int b = t / nsd;
  const int cib = blockIdx.y / cobs, cob = blockIdx.y % cobs;
  const int h0 = th * WT2::TH, w0 = tw * WT2::TWO;
  const int d_begin = ds * DR, d_end = (d_begin + DR < D) ? d_begin + DR : D;
  const bf16_t *xb = x + b * xv.sb + cib * 32;
  const bf16_t *yb = dy + b * yv.sb + cob * 32;
  const int cin_lim = (Cin + 7) / 8 * 8;
  const int l_vox = lane >> 2, l_chunk = lane & 3;

  auto xslot = [&](int xd) { return (xd + WT2::NXS) % WT2::NXS; };
  // piece i of this wave for output slice `od`: x slices 2od-1+{s} (s given by the piece index) and the dy slice
  auto issue_x_slice = [&](int xd, int i) __attribute__((always_inline)) {      // piece index idx = wave + 4 i < NPX1
    const int idx = wave + 4 * i;
    if (idx >= WT2::NPX1) return;
    const int r = idx / 3, pi = idx % 3;
    if (pi == 2 && lane >= 4) return;
    const int gh = 2 * h0 - 1 + r, wx = 16 * pi + l_vox, gw = 2 * w0 - 1 + wx;
    const bool ok = (unsigned)xd < (unsigned)xv.D && (unsigned)gh < (unsigned)xv.H && (unsigned)gw < (unsigned)xv.W &&
                    cib * 32 + l_chunk * 8 < cin_lim;
    const void *src = ok ? (const void *)(xb + xd * xv.sd + gh * xv.sh + gw * xv.sw + l_chunk * 8) : (const void *)&g_zero16;
    dma16_to_lds(src, lds_addr_of(sX + xslot(xd) * WT2::X_SLICE_B + r * WT2::X_ROW_B + pi * 1024));
  };
  auto issue_y_slice = [&](int yd) __attribute__((always_inline)) {             // rows 0/1 by waves 0/1
    if (wave >= WT2::TH) return;
    const int gh = h0 + wave, gw = w0 + l_vox;
    const bool ok = (unsigned)yd < (unsigned)D && gh < H && gw < W && cob * 32 + l_chunk * 8 < Cout;
    const void *src = ok ? (const void *)(yb + yd * yv.sd + gh * yv.sh + gw * yv.sw + l_chunk * 8) : (const void *)&g_zero16;
    dma16_to_lds(src, lds_addr_of(sY + (yd & 1) * WT2::Y_SLICE_B + wave * WT2::Y_ROW_B));
  };
  constexpr int NPXW = (WT2::NPX1 + 3) / 4;      // x pieces per wave and x slice

  // transposed-read lane addresses: dy block rows are consecutive voxels (64 B), x block rows every second voxel (128 B)
  const int kq = (lane >> 5) * 8 + ((lane & 15) >> 2), cpart = ((lane >> 4) & 1) * 32 + (lane & 3) * 8;
  const int lane_off_y = kq * 64 + cpart, lane_off_x = kq * 128 + cpart;

  int tap_kd[7], tap_off[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int tc = wave + 4 * i < 27 ? wave + 4 * i : 26;
    tap_kd[i] = tc / 9;
    tap_off[i] = ((tc / 3) % 3) * WT2::X_ROW_B + (tc % 3) * 64;
  }
  f32x16_t acc[7];
#pragma unroll
  for (int i = 0; i < 7; ++i)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;

  // prologue: x slices 2 d_begin - 1 .. 2 d_begin + 1, dy slice d_begin
#pragma unroll
  for (int sl = -1; sl <= 1; ++sl)
#pragma unroll
    for (int i = 0; i < NPXW; ++i) issue_x_slice(2 * d_begin + sl, i);
  issue_y_slice(d_begin);
  dma_wait_all();
  lds_barrier();

  for (int d = d_begin; d < d_end; ++d) {
    if (d + 1 < d_end) {       // next output slice: x slices 2d+2, 2d+3 and dy slice d+1 land during the MFMAs below
#pragma unroll
      for (int i = 0; i < NPXW; ++i) issue_x_slice(2 * d + 2, i);
#pragma unroll
      for (int i = 0; i < NPXW; ++i) issue_x_slice(2 * d + 3, i);
      issue_y_slice(d + 1);
    }
    const unsigned char *ys = sY + (d & 1) * WT2::Y_SLICE_B + lane_off_y;
    int slice_off[3];
#pragma unroll
    for (int kd = 0; kd < 3; ++kd) slice_off[kd] = xslot(2 * d + kd - 1) * WT2::X_SLICE_B;
#pragma unroll
    for (int oh = 0; oh < WT2::TH; ++oh) {
      // K-step = the 16 output voxels of the row
      const s16x4_t blo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t *)(ys + oh * WT2::Y_ROW_B));
      const s16x4_t bhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t *)(ys + oh * WT2::Y_ROW_B + 4 * 64));
      typedef __attribute__((ext_vector_type(8))) short s16x8_t;
      const s16x8_t bv = {blo[0], blo[1], blo[2], blo[3], bhi[0], bhi[1], bhi[2], bhi[3]};
      const bf16x8_t bfr = __builtin_bit_cast(bf16x8_t, bv);
      bf16x8_t afr[7];
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        const int so = (tap_kd[i] == 0 ? slice_off[0] : (tap_kd[i] == 1 ? slice_off[1] : slice_off[2])) + tap_off[i];
        const unsigned char *pa = sX + lane_off_x + so + 2 * oh * WT2::X_ROW_B;
        const s16x4_t alo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t *)pa);
        const s16x4_t ahi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t *)(pa + 4 * 128));
        const s16x8_t av = {alo[0], alo[1], alo[2], alo[3], ahi[0], ahi[1], ahi[2], ahi[3]};
        afr[i] = __builtin_bit_cast(bf16x8_t, av);
      }
#pragma unroll
      for (int i = 0; i < 7; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[i], bfr, acc[i], 0, 0, 0);
    }
    dma_wait_all();
    lds_barrier();
  }

  float *slab = slabs + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (27 * 1024);
  const int co = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int tap = wave + 4 * i;
    if (tap < 27) {
#pragma unroll
      for (int q = 0; q < 16; ++q) slab[(tap * 32 + (q & 3) + 8 * (q >> 2) + 4 * hh) * 32 + co] = acc[i][q];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// bf16 weight gradient of ConvTranspose3d(k2,s2) in one pass:  dW[ci][co][o] = sum_v x[v][ci] * dout[2v + o][co].
// Tile = 2 rows x 16 voxels of the INPUT lattice; the dout tile is kept at full resolution (4 rows x 32 voxels, slices 2d and
// 2d+1) and read with a 2-voxel row stride, the x fragment of a row is shared by the 8 offsets (2 per wave).  x and dout are
// read once, instead of 8 single-tap class launches that each re-read x and gathered a dout parity sub-lattice.
struct WT3 {
  static constexpr int TH = 2, TWI = 16;
  static constexpr int X_ROW_B = TWI * 64, X_SLICE_B = TH * X_ROW_B;             // 2 KiB
  static constexpr int Y_ROW_B = 2 * TWI * 64, Y_SLICE_B = 2 * TH * Y_ROW_B;     // one dout slice: 4 rows x 2 KiB
  static constexpr int Y_PAIR_B = 2 * Y_SLICE_B;                                 // dout slices 2d, 2d+1
  static constexpr int LDS_BYTES = 2 * X_SLICE_B + 2 * Y_PAIR_B;
  static constexpr int NPY = 2 * 2 * TH * 2;                                     // dout pieces per x slice (16 KiB)
};

__global__ __launch_bounds__(256, 2) void convT_wgrad_tr_kernel(const bf16_t *__restrict__ x, View xv,
                                                                const bf16_t *__restrict__ dout, View yv,
                                                                float *__restrict__ slabs, int Cin, int Cout, int tilesW,
                                                                int tilesH, int nsd, int DR, int cobs) {
  const int D = xv.D, H = xv.H, W = xv.W;                  // input lattice; yv = dense view of dout (2D x 2H x 2W)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char *sX = smem;                                // 2 slots
  unsigned char *sY = smem + 2 * WT3::X_SLICE_B;           // 2 slots of a slice pair
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int t = blockIdx.x;
  const int tw = t % tilesW;
  t /= tilesW;
  const int th = t % tilesH;
  t /= tilesH;
  const int ds = t % nsd;
  const int b = t / nsd;
  const int cib = blockIdx.y / cobs, cob = blockIdx.y % cobs;
  const int h0 = th * WT3::TH, w0 = tw * WT3::TWI;
  const int d_begin = ds * DR, d_end = (d_begin + DR < D) ? d_begin + DR : D;
  const bf16_t *xb = x + b * xv.sb + cib * 32;
  const bf16_t *yb = dout + b * yv.sb + cob * 32;
  const int cin_lim = (Cin + 7) / 8 * 8;
  const int l_vox = lane >> 2, l_chunk = lane & 3;

  // pieces of x slice d: 2 rows (waves 0,1); pieces of the dout pair: 2 slices x 4 rows x 2 halves = 16 (4 per wave)
  auto issue = [&](int d) __attribute__((always_inline)) {
    if (wave < WT3::TH) {
      const int gh = h0 + wave, gw = w0 + l_vox;
      const bool ok = (unsigned)d < (unsigned)D && gh < H && gw < W && cib * 32 + l_chunk * 8 < cin_lim;
      const void *src = ok ? (const void *)(xb + d * xv.sd + gh * xv.sh + gw * xv.sw + l_chunk * 8) : (const void *)&g_zero16;
      dma16_to_lds(src, lds_addr_of(sX + (d & 1) * WT3::X_SLICE_B + wave * WT3::X_ROW_B));
    }
#pragma unroll
    for (int i = 0; i < WT3::NPY / 4; ++i) {
      const int idx = wave + 4 * i;                     // (slice s, row r, half pi)
      const int sl = idx >> 3, r = (idx >> 1) & 3, pi = idx & 1;
      const int gd = 2 * d + sl, gh = 2 * h0 + r, gw = 2 * w0 + 16 * pi + l_vox;
      const bool ok = (unsigned)d < (unsigned)D && gd < yv.D && gh < yv.H && gw < yv.W && cob * 32 + l_chunk * 8 < Cout;
      const void *src = ok ? (const void *)(yb + gd * yv.sd + gh * yv.sh + gw * yv.sw + l_chunk * 8) : (const void *)&g_zero16;
      dma16_to_lds(src, lds_addr_of(sY + (d & 1) * WT3::Y_PAIR_B + sl * WT3::Y_SLICE_B + r * WT3::Y_ROW_B + pi * 1024));
    }
  };

  const int kq = (lane >> 5) * 8 + ((lane & 15) >> 2), cpart = ((lane >> 4) & 1) * 32 + (lane & 3) * 8;
  const int lane_off_x = kq * 64 + cpart, lane_off_y = kq * 128 + cpart;
  // this wave's two output offsets o = 2 wave, 2 wave + 1  (o = od*4 + oh*2 + ow)
  f32x16_t acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
  int ooff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int o = 2 * wave + i;
    ooff[i] = (o >> 2) * WT3::Y_SLICE_B + ((o >> 1) & 1) * WT3::Y_ROW_B + (o & 1) * 64;
  }

  issue(d_begin);
  dma_wait_all();
  lds_barrier();
  typedef __attribute__((ext_vector_type(8))) short s16x8_t;
  for (int d = d_begin; d < d_end; ++d) {
    if (d + 1 < d_end) issue(d + 1);
    const unsigned char *xs = sX + (d & 1) * WT3::X_SLICE_B + lane_off_x;
    const unsigned char *ys = sY + (d & 1) * WT3::Y_PAIR_B + lane_off_y;
#pragma unroll
    for (int r = 0; r < WT3::TH; ++r) {
      const s16x4_t alo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t *)(xs + r * WT3::X_ROW_B));
      const s16x4_t ahi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t *)(xs + r * WT3::X_ROW_B + 4 * 64));
      const s16x8_t av = {alo[0], alo[1], alo[2], alo[3], ahi[0], ahi[1], ahi[2], ahi[3]};
      const bf16x8_t afr = __builtin_bit_cast(bf16x8_t, av);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const unsigned char *pb = ys + ooff[i] + 2 * r * WT3::Y_ROW_B;
        const s16x4_t blo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t *)pb);
        const s16x4_t bhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t *)(pb + 4 * 128));
        const s16x8_t bv = {blo[0], blo[1], blo[2], blo[3], bhi[0], bhi[1], bhi[2], bhi[3]};
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr, __builtin_bit_cast(bf16x8_t, bv), acc[i], 0, 0, 0);
      }
    }
    dma_wait_all();
    lds_barrier();
  }
  // slab "tap" slot = output offset o
  float *slab = slabs + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (27 * 1024);
  const int co = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int o = 2 * wave + i;
#pragma unroll
    for (int q = 0; q < 16; ++q) slab[(o * 32 + (q & 3) + 8 * (q >> 2) + 4 * hh) * 32 + co] = acc[i][q];
  }
}

// dw[co*s_co + ci*s_ci + real_tap*s_tap] (+)= sum over slabs of virtual tap t (real_tap = real.wt[t], -1: skip).
// Workgroup = 32 consecutive output channels (one coalesced 128-byte row of every slab) x 8 slab groups; the 8 partial
// sums are combined through LDS in fixed order (deterministic).
template <int G>   // G slab groups per output row (8: many slabs, 1: few slabs -> 8 output rows per workgroup)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float *__restrict__ slabs, float *__restrict__ dw, int Cin,
                                                           int Cout, int cobs, int npairs, int nslab, int accumulate,
                                                           RealTaps reals, long long s_co, long long s_ci,
                                                           long long s_tap) {
  const Taps &real = reals.t[blockIdx.y];
  slabs += (int64_t)blockIdx.y * npairs * nslab * (27 * 1024);
  constexpr int R = 8 / G;                  // output rows (tap, ci, co-block) per workgroup
  __shared__ float part[8][32];
  const int lane = threadIdx.x & 31, sub = threadIdx.x >> 5;
  const int grp = sub % G, rsel = sub / G;
  const int cobs32 = (Cout + 31) / 32;
  const int64_t nrows = (int64_t)27 * Cin * cobs32;
  int64_t t = (int64_t)blockIdx.x * R + rsel;
  const bool live = t < nrows;
  if (!live) t = 0;
  const int cb = (int)(t % cobs32);
  t /= cobs32;
  const int ci = (int)(t % Cin);
  const int tap = (int)(t / Cin);
  const int rt = real.wt[tap];
  const int co = cb * 32 + lane;
  const bool ok = live && rt >= 0 && co < Cout;
  float s = 0.f;
  if (ok) {
    const int pair = (ci >> 5) * cobs + cb;
    const float *p = slabs + (int64_t)pair * nslab * (27 * 1024) + (tap * 32 + (ci & 31)) * 32 + lane;
    for (int k = grp; k < nslab; k += G) s += p[(int64_t)k * (27 * 1024)];
  }
  if (G > 1) {
    part[sub][lane] = s;
    __syncthreads();
    if (grp == 0) {
      s = 0.f;
#pragma unroll
      for (int g = 0; g < G; ++g) s += part[rsel * G + g][lane];
    }
  }
  if (grp == 0 && ok) {
    float *o = dw + co * s_co + ci * s_ci + rt * s_tap;
    *o = accumulate ? *o + s : s;
  }
}

struct WgradPlan {
  int tW, tH, nsd, DR, cibs, cobs;
  int64_t units;
};

WgradPlan wgrad_plan(int B, int Cin, int Cout, int D, int H, int W, int ncls = 1) {
  WgradPlan p;
  p.tW = cdiv(W, 32);
  p.tH = cdiv(H, 4);
  p.cibs = cdiv(Cin, 32);
  p.cobs = cdiv(Cout, 32);
  const int64_t base = (int64_t)B * p.tW * p.tH * p.cibs * p.cobs * ncls;
  int want = (int)cdiv64(512, base);                  // aim for >= ~512 workgroups (2 per CU) over all classes
  int maxsplit = D / 4 > 0 ? D / 4 : 1;
  p.nsd = want < 1 ? 1 : (want > maxsplit ? maxsplit : want);
  p.DR = cdiv(D, p.nsd);
  p.nsd = cdiv(D, p.DR);
  p.units = (int64_t)B * p.tW * p.tH * p.nsd;
  return p;
}

}  // namespace

// one-pass stride-2 kernel: tiles of 2 rows x 16 voxels of the output lattice
static WgradPlan wgrad_plan_s2(int B, int Cin, int Cout, int D, int H, int W) {
  WgradPlan p;
  p.tW = cdiv(W, WT2::TWO);
  p.tH = cdiv(H, WT2::TH);
  p.cibs = cdiv(Cin, 32);
  p.cobs = cdiv(Cout, 32);
  const int64_t base = (int64_t)B * p.tW * p.tH * p.cibs * p.cobs;
  int want = (int)cdiv64(512, base);
  int maxsplit = D / 4 > 0 ? D / 4 : 1;
  p.nsd = want < 1 ? 1 : (want > maxsplit ? maxsplit : want);
  p.DR = cdiv(D, p.nsd);
  p.nsd = cdiv(D, p.DR);
  p.units = (int64_t)B * p.tW * p.tH * p.nsd;
  return p;
}

// sized for the 8-class launches (stride-2 conv, transposed conv) and the one-pass stride-2 plan; single-class launches
// use the first part
size_t conv3_wgrad_mfma_ws_bytes(int B, int Cin, int Cout, int D, int H, int W) {
  WgradPlan p1 = wgrad_plan(B, Cin, Cout, D, H, W, 1), p8 = wgrad_plan(B, Cin, Cout, D, H, W, 8),
            p2 = wgrad_plan_s2(B, Cin, Cout, D, H, W);
  size_t a = (size_t)p1.units * p1.cibs * p1.cobs, b = (size_t)8 * p8.units * p8.cibs * p8.cobs,
         c = (size_t)p2.units * p2.cibs * p2.cobs;
  a = a > b ? a : b;
  return (a > c ? a : c) * 27 * 1024 * sizeof(float);
}

template <typename T>
static int wgrad_launch_classes(const void *x, const View &xv, const void *dy, const View &yv, float *dw, void *ws,
                                size_t ws_bytes, int B, int Cin, int Cout, const WgradClasses &wc, const RealTaps &reals,
                                long long s_co, long long s_ci, long long s_tap, int accumulate, hipStream_t st) {
  constexpr int EPV = Elem<T>::EPV;
  // Cin may be ragged (first layer: 12 channels in rows of 16): the pad channels only feed gradient rows ci >= Cin,
  // which the reduction never writes.  The rows must be long enough to be read in whole 16-byte groups.
  if (Cout % EPV || xv.sw % EPV || yv.sw % EPV || ((uintptr_t)x & 15) || ((uintptr_t)dy & 15) ||
      xv.sw < (Cin + EPV - 1) / EPV * EPV)
    return DGTTA_ERR_UNSUPPORTED;
  for (int c = 0; c < wc.n; ++c)
    if ((wc.xoff[c] * (long long)sizeof(T)) % 16 || (wc.yoff[c] * (long long)sizeof(T)) % 16) return DGTTA_ERR_UNSUPPORTED;
  WgradPlan p = wgrad_plan(B, Cin, Cout, yv.D, yv.H, yv.W, wc.n);
  const size_t need = (size_t)wc.n * p.units * p.cibs * p.cobs * 27 * 1024 * sizeof(float);
  if (ws_bytes < need || p.units >= (1ll << 31) || p.cibs * p.cobs > 65535) return DGTTA_ERR_UNSUPPORTED;
  if (sizeof(T) == 2) {
    const char *tr = getenv("DGTTA_WGRAD_TR");        // diagnostic / tests: "0" forces the register-transpose kernel
    if (!(tr && tr[0] == '0')) {
      const bool plain = wc.n == 1 && wc.mask[0] == 0x7ffffffu && wc.xoff[0] == 0 && wc.yoff[0] == 0;
      auto ktr = plain ? conv3_wgrad_tr_kernel<0, false> : conv3_wgrad_tr_kernel<0, true>;
      static const char *abl = getenv("DGTTA_WGRAD_ABL");      // diagnostic only
      if (abl && abl[0] == '1' && plain) ktr = conv3_wgrad_tr_kernel<1, false>;
      if (abl && abl[0] == '3' && plain) ktr = conv3_wgrad_tr_kernel<3, false>;
      static bool tr_attr[2] = {false, false};
      if (!tr_attr[plain] || abl) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(ktr), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)WT::LDS_BYTES);
        DG_REQUIRE(e == hipSuccess, DGTTA_ERR_LAUNCH, "wgrad_tr: cannot raise the dynamic LDS limit");
        tr_attr[plain] = true;
      }
      const char *w8 = getenv("DGTTA_WGRAD_TR8");      // diagnostic / tests: "0" = always the 4-wave kernel
      if (plain && Cout >= 64 && !abl && !(w8 && w8[0] == '0')) {
        static bool a8 = false;
        if (!a8) {
          hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv3_wgrad_tr8_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)WT8::LDS_BYTES);
          DG_REQUIRE(e == hipSuccess, DGTTA_ERR_LAUNCH, "wgrad_tr8: cannot raise the dynamic LDS limit");
          a8 = true;
        }
        hipLaunchKernelGGL(conv3_wgrad_tr8_kernel, dim3((unsigned)p.units, (unsigned)(p.cibs * ((p.cobs + 1) / 2))), dim3(512),
                           WT8::LDS_BYTES, st, (const bf16_t *)x, xv, (const bf16_t *)dy, yv, (float *)ws, Cin, Cout, p.tW, p.tH,
                           p.nsd, p.DR, p.cobs);
        DG_CHECK_LAUNCH("conv3_wgrad_tr8_kernel");
        goto reduce;
      }
      hipLaunchKernelGGL(ktr, dim3((unsigned)p.units, (unsigned)(p.cibs * p.cobs), (unsigned)wc.n), dim3(256), WT::LDS_BYTES,
                         st, (const bf16_t *)x, xv, (const bf16_t *)dy, yv, (float *)ws, Cin, Cout, p.tW, p.tH, p.nsd,
                         p.DR, p.cobs, wc);
      DG_CHECK_LAUNCH("conv3_wgrad_tr_kernel");
      goto reduce;
    }
  }
  {
  static bool attr_set = false;
  auto kern = conv3_wgrad_mfma_kernel<T, 0>;
  {
    static const char *abl = getenv("DGTTA_WGRAD_ABL");      // diagnostic only
    if (abl && abl[0] == '1') kern = conv3_wgrad_mfma_kernel<T, 1>;
    if (abl && abl[0] == '2') kern = conv3_wgrad_mfma_kernel<T, 2>;
    if (abl && abl[0] == '3') kern = conv3_wgrad_mfma_kernel<T, 3>;
    if (abl && abl[0] == '6') kern = conv3_wgrad_mfma_kernel<T, 6>;
    if (abl) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)WG<T>::LDS_BYTES);
  }
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)WG<T>::LDS_BYTES);
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)p.units, (unsigned)(p.cibs * p.cobs), (unsigned)wc.n), dim3(256), WG<T>::LDS_BYTES,
                     st, (const T *)x, xv, (const T *)dy, yv, (float *)ws, Cin, Cout, p.tW, p.tH, p.nsd, p.DR, p.cobs, wc);
  DG_CHECK_LAUNCH("conv3_wgrad_mfma_kernel");
  }
reduce:
  const int64_t rrows = (int64_t)27 * Cin * ((Cout + 31) / 32);
  const int npairs = p.cibs * p.cobs;
  if (p.units >= 64)
    hipLaunchKernelGGL(wgrad_reduce_kernel<8>, dim3((unsigned)rrows, (unsigned)wc.n), dim3(256), 0, st, (const float *)ws, dw,
                       Cin, Cout, p.cobs, npairs, (int)p.units, accumulate, reals, s_co, s_ci, s_tap);
  else
    hipLaunchKernelGGL(wgrad_reduce_kernel<1>, dim3((unsigned)cdiv64(rrows, 8), (unsigned)wc.n), dim3(256), 0, st,
                       (const float *)ws, dw, Cin, Cout, p.cobs, npairs, (int)p.units, accumulate, reals, s_co, s_ci, s_tap);
  DG_CHECK_LAUNCH("wgrad_reduce_kernel");
  return DGTTA_OK;
}

template <typename T>
static int wgrad_launch(const void *x, const View &xv, const void *dy, const View &yv, float *dw, void *ws, size_t ws_bytes,
                        int B, int Cin, int Cout, unsigned tapmask, const Taps &real, long long s_co, long long s_ci,
                        long long s_tap, int accumulate, hipStream_t st) {
  WgradClasses wc;
  wc.n = 1;
  wc.mask[0] = tapmask;
  wc.xoff[0] = wc.yoff[0] = 0;
  RealTaps reals;
  reals.t[0] = real;
  return wgrad_launch_classes<T>(x, xv, dy, yv, dw, ws, ws_bytes, B, Cin, Cout, wc, reals, s_co, s_ci, s_tap, accumulate, st);
}

template <typename T>
static int wgrad_conv(const void *x, int ldx, const void *dy, int lddy, float *dw_t, void *ws, size_t ws_bytes, int B,
                      int Cin, int Cout, int Di, int Hi, int Wi, int stride, int accumulate, hipStream_t st) {
  const long long s_co = (long long)Cin * 27, s_ci = 27, s_tap = 1;
  if (stride == 1) {
    const View xv = dense_view(B, Di, Hi, Wi, ldx), yv = dense_view(B, Di, Hi, Wi, lddy);
    return wgrad_launch<T>(x, xv, dy, yv, dw_t, ws, ws_bytes, B, Cin, Cout, 0x7ffffffu, identity_taps(0), s_co, s_ci, s_tap,
                           accumulate, st);
  }
  // stride 2: x[2*vo + tap - 1] lives on parity sub-lattices of x; per axis parity 0 <- tap 1 (offset 0),
  // parity 1 <- tap 0 (offset -1) and tap 2 (offset 0).  Each real tap belongs to exactly one of the 8 classes.
  const int Do = (Di - 1) / 2 + 1, Ho = (Hi - 1) / 2 + 1, Wo = (Wi - 1) / 2 + 1;
  const View yv = dense_view(B, Do, Ho, Wo, lddy);
  if (sizeof(T) == 2) {
    // one pass over x (full resolution tile) and dy with all 27 taps: conv3_wgrad_tr_s2_kernel
    const char *one = getenv("DGTTA_WGRAD_S2_ONEPASS");      // diagnostic / tests: "0" = the 8-class launch
    const View xfull = dense_view(B, Di, Hi, Wi, ldx);
    WgradPlan p = wgrad_plan_s2(B, Cin, Cout, Do, Ho, Wo);
    const size_t need = (size_t)p.units * p.cibs * p.cobs * 27 * 1024 * sizeof(float);
    const bool ok = Cout % 8 == 0 && ldx % 8 == 0 && lddy % 8 == 0 && !((uintptr_t)x & 15) && !((uintptr_t)dy & 15) &&
                    ldx >= (Cin + 7) / 8 * 8 && ws_bytes >= need && p.units < (1ll << 31) && p.cibs * p.cobs <= 65535;
    if (ok && !(one && one[0] == '0')) {
      static bool attr = false;
      if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv3_wgrad_tr_s2_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)WT2::LDS_BYTES);
        DG_REQUIRE(e == hipSuccess, DGTTA_ERR_LAUNCH, "wgrad_tr_s2: cannot raise the dynamic LDS limit");
        attr = true;
      }
      hipLaunchKernelGGL(conv3_wgrad_tr_s2_kernel, dim3((unsigned)p.units, (unsigned)(p.cibs * p.cobs)), dim3(256),
                         WT2::LDS_BYTES, st, (const bf16_t *)x, xfull, (const bf16_t *)dy, yv, (float *)ws, Cin, Cout, p.tW,
                         p.tH, p.nsd, p.DR, p.cobs);
      DG_CHECK_LAUNCH("conv3_wgrad_tr_s2_kernel");
      RealTaps ident;
      ident.t[0] = identity_taps(0);
      const int64_t rrows = (int64_t)27 * Cin * ((Cout + 31) / 32);
      const int npairs = p.cibs * p.cobs;
      if (p.units >= 64)
        hipLaunchKernelGGL(wgrad_reduce_kernel<8>, dim3((unsigned)rrows, 1u), dim3(256), 0, st, (const float *)ws, dw_t, Cin,
                           Cout, p.cobs, npairs, (int)p.units, accumulate, ident, s_co, s_ci, s_tap);
      else
        hipLaunchKernelGGL(wgrad_reduce_kernel<1>, dim3((unsigned)cdiv64(rrows, 8), 1u), dim3(256), 0, st, (const float *)ws,
                           dw_t, Cin, Cout, p.cobs, npairs, (int)p.units, accumulate, ident, s_co, s_ci, s_tap);
      DG_CHECK_LAUNCH("wgrad_reduce_kernel");
      return DGTTA_OK;
    }
  }
  WgradClasses wc;
  RealTaps reals;
  wc.n = 8;
  View xv;
  for (int p = 0; p < 8; ++p) {
    const int par[3] = {p >> 2, (p >> 1) & 1, p & 1};
    long long off;
    xv = parity_view(Di, Hi, Wi, ldx, par[0], par[1], par[2], &off);   // even extents: same shape for all classes
    xv.sb = (long long)Di * Hi * Wi * ldx;
    wc.xoff[p] = off;
    wc.yoff[p] = 0;
    unsigned mask = 0;
    for (int t = 0; t < 27; ++t) {
      const int k[3] = {t / 9, (t / 3) % 3, t % 3};
      int rl[3];
      bool ok = true;
      for (int a = 0; a < 3; ++a) {
        if (par[a] == 0) {
          ok = ok && (k[a] == 1);
          rl[a] = 1;
        } else {
          ok = ok && (k[a] <= 1);
          rl[a] = (k[a] == 0) ? 0 : 2;
        }
      }
      reals.t[p].wt[t] = ok ? (signed char)(rl[0] * 9 + rl[1] * 3 + rl[2]) : (signed char)-1;
      if (ok) mask |= 1u << t;
    }
    wc.mask[p] = mask;
  }
  return wgrad_launch_classes<T>(x, xv, dy, yv, dw_t, ws, ws_bytes, B, Cin, Cout, wc, reals, s_co, s_ci, s_tap, accumulate, st);
}

int conv3_wgrad_mfma(const void *x, int ldx, const void *dy, int lddy, float *dw_t, float *db, void *ws, size_t ws_bytes,
                     int B, int Cin, int Cout, int Di, int Hi, int Wi, int stride, int accumulate, int dtype,
                     hipStream_t st) {
  (void)db;
  if (stride != 1 && stride != 2) return DGTTA_ERR_UNSUPPORTED;
  if (stride == 2 && ((Di | Hi | Wi) & 1)) return DGTTA_ERR_UNSUPPORTED;   // odd extents: leave to the general kernel
  if (dtype == DGTTA_F32) return wgrad_conv<float>(x, ldx, dy, lddy, dw_t, ws, ws_bytes, B, Cin, Cout, Di, Hi, Wi, stride, accumulate, st);
  if (dtype == DGTTA_BF16) return wgrad_conv<bf16_t>(x, ldx, dy, lddy, dw_t, ws, ws_bytes, B, Cin, Cout, Di, Hi, Wi, stride, accumulate, st);
  return DGTTA_ERR_UNSUPPORTED;
}

// ConvTranspose3d k2 s2 weight gradient: dw_t[ci][co][o] (+)= sum_v x[v][ci] * dout[2v+o][co]  (8 single-tap launches)
template <typename T>
static int convT_wgrad(const void *x, int ldx, const void *dout, int lddo, float *dw_t, void *ws, size_t ws_bytes, int B,
                       int Cin, int Cout, int Di, int Hi, int Wi, int accumulate, hipStream_t st) {
  const View xv = dense_view(B, Di, Hi, Wi, ldx);
  if (sizeof(T) == 2) {
    const char *one = getenv("DGTTA_CONVT_WGRAD_ONEPASS");      // diagnostic / tests: "0" = the 8-class launch
    const View yfull = dense_view(B, 2 * Di, 2 * Hi, 2 * Wi, lddo);
    WgradPlan p = wgrad_plan_s2(B, Cin, Cout, Di, Hi, Wi);        // same tile shape (2 rows x 16 voxels) on the input lattice
    const size_t need = (size_t)p.units * p.cibs * p.cobs * 27 * 1024 * sizeof(float);
    const bool ok = Cout % 8 == 0 && ldx % 8 == 0 && lddo % 8 == 0 && !((uintptr_t)x & 15) && !((uintptr_t)dout & 15) &&
                    ldx >= (Cin + 7) / 8 * 8 && ws_bytes >= need && p.units < (1ll << 31) && p.cibs * p.cobs <= 65535;
    if (ok && !(one && one[0] == '0')) {
      static bool attr = false;
      if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(convT_wgrad_tr_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)WT3::LDS_BYTES);
        DG_REQUIRE(e == hipSuccess, DGTTA_ERR_LAUNCH, "convT_wgrad_tr: cannot raise the dynamic LDS limit");
        attr = true;
      }
      hipLaunchKernelGGL(convT_wgrad_tr_kernel, dim3((unsigned)p.units, (unsigned)(p.cibs * p.cobs)), dim3(256),
                         WT3::LDS_BYTES, st, (const bf16_t *)x, xv, (const bf16_t *)dout, yfull, (float *)ws, Cin, Cout, p.tW,
                         p.tH, p.nsd, p.DR, p.cobs);
      DG_CHECK_LAUNCH("convT_wgrad_tr_kernel");
      RealTaps rt;
      for (int t = 0; t < 27; ++t) rt.t[0].wt[t] = (signed char)(t < 8 ? t : -1);      // slab tap slot o -> dw_t[..][o]
      const int64_t rrows = (int64_t)27 * Cin * ((Cout + 31) / 32);
      const int npairs = p.cibs * p.cobs;
      if (p.units >= 64)
        hipLaunchKernelGGL(wgrad_reduce_kernel<8>, dim3((unsigned)rrows, 1u), dim3(256), 0, st, (const float *)ws, dw_t, Cin,
                           Cout, p.cobs, npairs, (int)p.units, accumulate, rt, 8, (long long)Cout * 8, 1);
      else
        hipLaunchKernelGGL(wgrad_reduce_kernel<1>, dim3((unsigned)cdiv64(rrows, 8), 1u), dim3(256), 0, st, (const float *)ws,
                           dw_t, Cin, Cout, p.cobs, npairs, (int)p.units, accumulate, rt, 8, (long long)Cout * 8, 1);
      DG_CHECK_LAUNCH("wgrad_reduce_kernel");
      return DGTTA_OK;
    }
  }
  WgradClasses wc;
  RealTaps reals;
  wc.n = 8;
  View yv;
  for (int o = 0; o < 8; ++o) {
    long long off;
    yv = parity_view(2 * Di, 2 * Hi, 2 * Wi, lddo, o >> 2, (o >> 1) & 1, o & 1, &off);
    yv.sb = (long long)8 * Di * Hi * Wi * lddo;
    wc.xoff[o] = 0;
    wc.yoff[o] = off;
    wc.mask[o] = 1u << 13;
    for (int t = 0; t < 27; ++t) reals.t[o].wt[t] = -1;
    reals.t[o].wt[13] = (signed char)o;
  }
  return wgrad_launch_classes<T>(x, xv, dout, yv, dw_t, ws, ws_bytes, B, Cin, Cout, wc, reals, 8, (long long)Cout * 8, 1,
                                 accumulate, st);
}

int convT_wgrad_mfma(const void *x, int ldx, const void *dout, int lddo, float *dw_t, void *ws, size_t ws_bytes, int B,
                     int Cin, int Cout, int Di, int Hi, int Wi, int accumulate, int dtype, hipStream_t st) {
  if (dtype == DGTTA_F32) return convT_wgrad<float>(x, ldx, dout, lddo, dw_t, ws, ws_bytes, B, Cin, Cout, Di, Hi, Wi, accumulate, st);
  if (dtype == DGTTA_BF16) return convT_wgrad<bf16_t>(x, ldx, dout, lddo, dw_t, ws, ws_bytes, B, Cin, Cout, Di, Hi, Wi, accumulate, st);
  return DGTTA_ERR_UNSUPPORTED;
}

// 1x1x1 head weight gradient dw[k][ci] = sum_rows dout[row][k] * x[row][ci] as a single-tap run of the wgrad kernel:
// the [rows] axis is folded into a D x 4 x 32 lattice (no neighbour access with one tap, so any folding is valid).
namespace {
__global__ void f32_to_bf16_rows_kernel(const float *__restrict__ src, int lds_, bf16_t *__restrict__ dst, int C,
                                        int64_t rows) {
  const int64_t n = rows * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dst[i] = f32_to_bf16(src[(i / C) * lds_ + i % C]);
}
}  // namespace

size_t head_wgrad_mfma_ws_bytes(int Cin, int nsel, int64_t rows) {
  if (rows % 128) return 0;
  const int D = (int)(rows / 128);
  return conv3_wgrad_mfma_ws_bytes(1, Cin, nsel, D, 4, 32) + align_up((size_t)rows * nsel * 2, 256);
}

int head_wgrad_mfma(const void *x, int ldx, const float *dout, int lddo, float *dw_sel, void *ws, size_t ws_bytes, int Cin,
                    int nsel, int64_t rows, int accumulate, int dtype, hipStream_t st) {
  if (rows % 128 || rows / 128 >= (1ll << 30)) return DGTTA_ERR_UNSUPPORTED;
  const int D = (int)(rows / 128);
  if (ws_bytes < head_wgrad_mfma_ws_bytes(Cin, nsel, rows)) return DGTTA_ERR_UNSUPPORTED;
  Taps real;
  for (int t = 0; t < 27; ++t) real.wt[t] = -1;
  real.wt[13] = 0;
  const View xv = dense_view(1, D, 4, 32, ldx);
  if (dtype == DGTTA_F32) {
    const View yv = dense_view(1, D, 4, 32, lddo);
    return wgrad_launch<float>(x, xv, dout, yv, dw_sel, ws, ws_bytes, 1, Cin, nsel, 1u << 13, real, Cin, 1, 0, accumulate,
                               st);
  }
  if (dtype == DGTTA_BF16) {
    const size_t cbytes = align_up((size_t)rows * nsel * 2, 256);
    bf16_t *d16 = (bf16_t *)ws;
    hipLaunchKernelGGL(f32_to_bf16_rows_kernel, dim3(2048), dim3(256), 0, st, dout, lddo, d16, nsel, rows);
    DG_CHECK_LAUNCH("f32_to_bf16_rows_kernel");
    const View yv = dense_view(1, D, 4, 32, nsel);
    return wgrad_launch<bf16_t>(x, xv, d16, yv, dw_sel, (char *)ws + cbytes, ws_bytes - cbytes, 1, Cin, nsel, 1u << 13, real,
                                Cin, 1, 0, accumulate, st);
  }
  return DGTTA_ERR_UNSUPPORTED;
}
