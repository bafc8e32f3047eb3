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
#include "conv_common.h"
#include <stdlib.h>
#include <type_traits>

int conv3_rows_launch(const void *x, const View &xv, const void *w, const Taps &taps, const float *bias, void *y, const View &yv,
                      int B, int Cin, int Cout, int CinP, int CoutP, double *stats, int ntaps_src, int is_f16, hipStream_t st,
                      RowsGstCtx *gst);

int conv3_ring_launch(const void *x, const View &xv, const void *w, const Taps &taps, const float *bias, void *y, const View &yv,
                      int B, int Cin, int Cout, int CinP, int CoutP, double *stats, int64_t stats_cap_slots, int ntaps_src,
                      int is_f16, hipStream_t st, RowsGstCtx *gst, long long xkh = 0, bool dry = false);      // conv_ring.hip
int64_t conv3_mfma_max_tiles(int Do, int Ho, int Wo);

namespace {

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

  // tile index: a contiguous run of tiles per XCD (workgroups are dispatched round robin over the 8 XCDs; neighbouring
  // tiles share their input halo, which one L2 should fetch once)
  int tlin = blockIdx.x;
  if ((gridDim.x & 7) == 0) tlin = (tlin & 7) * (int)(gridDim.x >> 3) + (tlin >> 3);
  int t = tlin;
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
  // Weight slots of this thread, resolved ONCE (round 6).  The tap table lives in the kernel arguments and a lane's slot picks
  // its tap at run time, so `taps.wt[tap]` inside load_chunk was a per-lane BYTE LOAD FROM MEMORY in front of every weight load:
  // global_load_sbyte - s_waitcnt vmcnt(0) - branch - global_load_dwordx4, i.e. every one of a chunk's 4-14 weight loads per
  // thread waited for the one before it AND for the activation loads in flight (two dependent round trips each; the tiny layers,
  // one wave per SIMD, spent ~6 us per chunk there against 0.8 us of MFMAs).  The table goes through LDS once, and a slot's
  // element offset for chunk 0 (or -1: tap absent / slot beyond the tile) is kept in a register; a chunk only adds its K offset.
  __shared__ signed char s_wt[32];
  if (tid < 27) s_wt[tid] = AC ? (signed char)tid : taps.wt[tid];
  __syncthreads();
  int wslot[NBL];
#pragma unroll
  for (int i = 0; i < NBL; ++i) {
    const int idx = tid + i * NT;       // == LDS index (tap*NG + g)*NC + n
    const int n = idx % NC, g = (idx / NC) % NG, tap = (S == 0) ? 13 : (idx / (NG * NC)) % 27;
    const int wt = s_wt[tap];
    const bool ok = idx < NTAP * NC * NG && wt >= 0;
    const int off = ((((n0 + n) / 32) * (CinP / (2 * EPV)) + g / 2) * ntaps_src + wt) * 64 + (g & 1) * 32 + (n0 + n) % 32;
    wslot[i] = ok ? off : -1;
  }
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
    const int kterm = (kc / (2 * EPV)) * ntaps_src * 64;      // K-chunk of 2*EPV channels: ntaps_src x 2 x 32 slots each
#pragma unroll
    for (int i = 0; i < NBL; ++i) {
      const bool ok = wslot[i] >= 0;
      const T *p = w + (int64_t)(ok ? wslot[i] + kterm : 0) * EPV;
      if (ABL == 1 || ABL == 4) {
        rb[i] = make_uint4(tid + i * NT, 0, 0, 0);
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
      // the skip-connection sum reads the old rows: all 16 of a tile's 16-byte pieces are requested before the first class is
      // written back (round 4) - one load, one wait, one store per class made the epilogue a chain of 16 memory round trips
      // per tile, about as long as the tile's MFMA loop
      uint4 olds[8][2];
      if constexpr (sizeof(T) == 2) {
        if (vec && accumulate) {
#pragma unroll
          for (int c8 = 0; c8 < 8; ++c8)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              const int m = t * 16 + (lane >> 2), cq = (lane & 3) * 8;
              const int od = od0 + mbd, oh = oh0 + mbh * G::RPM + m / MBW, ow = ow0 + m % MBW;
              olds[c8][t] = make_uint4(0u, 0u, 0u, 0u);
              if (n0 + cq < Cout && od < Do && oh < Ho && ow < Wo)
                olds[c8][t] = *reinterpret_cast<const uint4 *>(y + cs.yoff[c8] + b * yv.sb + od * yv.sd + oh * yv.sh + ow * yv.sw + n0 + cq);
            }
        }
      }
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
            if constexpr (sizeof(T) == 2) {
              uint4 *o4 = reinterpret_cast<uint4 *>(o);
              if (accumulate) {
                const uint4 old = olds[c8][t];
                const unsigned wv[4] = {old.x, old.y, old.z, old.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  float lo, hi;
                  unpack2_16<T>(wv[e], lo, hi);
                  v[2 * e] += lo;
                  v[2 * e + 1] += hi;
                }
              }
              uint4 pk;
              pk.x = pack2_16<T>(v[0], v[1]);
              pk.y = pack2_16<T>(v[2], v[3]);
              pk.z = pack2_16<T>(v[4], v[5]);
              pk.w = pack2_16<T>(v[6], v[7]);
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
      double *p = stats + 32 + (((int64_t)b * tiles_per_b + (tlin % tiles_per_b)) * Cout + n0 + tid) * 2;
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
  static DynLdsOnce lds_once;
  DG_REQUIRE(ensure_dyn_lds(lds_once, reinterpret_cast<const void *>(kern), (int)Cfg::LDS_BYTES) == hipSuccess,
             DGTTA_ERR_LAUNCH, "conv3_mfma: cannot raise the dynamic LDS limit to %zu", (size_t)Cfg::LDS_BYTES);
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
  auto kern = conv3_mfma_kernel<T, MBW, MBH, MBD, S, NB, KSPC, 0, NW>;
  static DynLdsOnce lds_once;
  DG_REQUIRE(ensure_dyn_lds(lds_once, reinterpret_cast<const void *>(kern), (int)Cfg::LDS_BYTES) == hipSuccess,
             DGTTA_ERR_LAUNCH, "conv3_mfma: cannot raise the dynamic LDS limit to %zu", (size_t)Cfg::LDS_BYTES);
  const int tW = cdiv(yv.W, G::TW), tH = cdiv(yv.H, G::TH), tD = cdiv(yv.D, G::TD);
  const int64_t tiles = (int64_t)tW * tH * tD * B;
  DG_REQUIRE(tiles < (1ll << 31), DGTTA_ERR_UNSUPPORTED, "conv3_mfma: too many tiles");
  dim3 grid((unsigned)tiles, (unsigned)cdiv(CoutP, Cfg::NC), (unsigned)cs.n);
  hipLaunchKernelGGL(kern, grid, dim3(NW * 64), Cfg::LDS_BYTES, st, (const T *)x, xv, (const T *)w, cs, bias, (T *)y, yv,
                     Cin, Cout, CinP, CoutP, tW, tH, tD, stats, ntaps_src);
  DG_CHECK_LAUNCH("conv3_mfma_kernel");
  return DGTTA_OK;
}

// picks the tile shape from the (virtual) output extent
template <typename T>
int dispatch_conv_classes(const void *x, const View &xv, const void *w, const ConvClasses &cs, const float *bias, void *y,
                          const View &yv, int B, int Cin, int Cout, int CinP, int CoutP, int stride, hipStream_t st,
                          double *stats = nullptr, int ntaps_src = 27, RowsGstCtx *gst = nullptr, long long xkh = 0, bool dry = false) {
  // xkh != 0: x holds its two 32-channel K halves as dense planes xkh elements apart - only the ring kernel reads that layout;
  // dry: no launch, DGTTA_OK iff the ring kernel would take this call (dgtta_conv3d_k3_blocked_supported)
#define ARGS x, xv, w, cs, bias, y, yv, B, Cin, Cout, CinP, CoutP, stats, ntaps_src, st
  const long long vox = (long long)yv.D * yv.H * yv.W * B;
  if ((xkh || dry) && stride != 1) return DGTTA_ERR_UNSUPPORTED;
  if (stride == 0) {      // pointwise (centre tap only, no halo)
    if (yv.W >= 32) return launch_conv<T, 32, 4, 4, 0, 1, 1, 8>(ARGS);
    if (yv.W >= 16) return launch_conv<T, 16, 4, 4, 0, 1, 1, 8>(ARGS);
    if (vox <= 4096) return launch_conv<T, 8, 2, 2, 0, 1, 1>(ARGS);
    return launch_conv<T, 8, 2, 8, 0, 1, 1, 8>(ARGS);
  }
  if (stride == 1 && sizeof(T) == 2 && yv.W >= 32 && cs.n == 1 && cs.acc[0] == 0 && cs.kseg == 0 && cs.xoff[0] == 0 &&
      cs.yoff[0] == 0) {
    const int rows = dgtta_switches().conv_rows;   // DGTTA_CONV_ROWS (tests): '0' forces the generic kernel, '1' this one
    bool all_taps = true;
    for (int t = 0; t < 27; ++t) all_taps = all_taps && cs.taps[0].wt[t] >= 0;
    const bool vec_out = Cout % 8 == 0 && ((uintptr_t)y & 15) == 0 && yv.sw % 8 == 0 && yv.sh % 8 == 0 &&
                         yv.sd % 8 == 0 && yv.sb % 8 == 0;
    // enough (tile, channel block) jobs to fill the chip with one persistent workgroup per CU; below that the generic kernel wins
    const long long njobs = (long long)cdiv(yv.W, 32) * cdiv(yv.H, 8) * cdiv(yv.D, 4) * cdiv(CoutP, 32) * B;
    // 32 or 64 input channels: the D-ring kernel (conv_ring.hip; DGTTA_CONV_RING=0: its predecessor below, =3: only 32 channels)
    if (all_taps && vec_out && ((Cin == 32 && CinP == 32) || (Cin == 64 && CinP == 64)) && (njobs >= 512 || dgtta_switches().conv_ring == '1') &&
        dgtta_switches().conv_ring != '0' && rows != '1') {
      const int rc = conv3_ring_launch(x, xv, w, cs.taps[0], bias, y, yv, B, Cin, Cout, CinP, CoutP, stats,
                                       conv3_mfma_max_tiles(yv.D, yv.H, yv.W), ntaps_src, (int)std::is_same<T, f16_t>::value, st, gst,
                                       xkh, dry);
      if (rc != DGTTA_ERR_UNSUPPORTED || xkh || dry) return rc;
    }
    if (xkh || dry) return DGTTA_ERR_UNSUPPORTED;
    if (all_taps && vec_out && (njobs >= 256 || rows == '1') && rows != '0')
      return conv3_rows_launch(x, xv, w, cs.taps[0], bias, y, yv, B, Cin, Cout, CinP, CoutP, stats, ntaps_src,
                               (int)std::is_same<T, f16_t>::value, st, gst);
  }
  if (xkh || dry) return DGTTA_ERR_UNSUPPORTED;
  if (stride == 1) {
    // 8 waves per workgroup (2 M-blocks each): 16 waves per CU hide the LDS / barrier latency (+27 % over 4 waves; the
    // other tile shapes tried in round 1 - 2 k-steps per chunk, 64 output channels, 256-voxel tiles - were slower)
    if (yv.W >= 32) return launch_conv<T, 32, 4, 4, 1, 1, 1, 8>(ARGS);
    // tiny volumes are latency bound (few workgroups, a long serial K loop): when the channel padding allows it, 2
    // k-steps per chunk halve the barrier / load round trips.  (128-voxel tiles for the 16^3 layers measured faster in
    // isolation but slower inside the network: 23.0 vs 20.9 ms per epoch.)
    const bool k2 = CinP % (4 * Elem<T>::EPV) == 0;
    if (yv.W >= 16) return launch_conv<T, 16, 4, 4, 1, 1, 1, 8>(ARGS);
    if (vox <= 4096 && k2) return launch_conv<T, 8, 2, 2, 1, 1, 2>(ARGS);
    if (vox <= 4096) return launch_conv<T, 8, 2, 2, 1, 1, 1>(ARGS);     // tiny volumes: more, smaller workgroups
    return launch_conv<T, 8, 2, 8, 1, 1, 1, 8>(ARGS);
  }
  if (stride == 2) {
    // 8 waves per workgroup (one M-block each): 231 -> 153 us at 128^3 -> 64^3, 32 -> 64 channels
    if (dgtta_switches().conv_s2 == '4') {      // DGTTA_CONV_S2=4: the former 4-wave tiles
      if (yv.W >= 16) return launch_conv<T, 16, 2, 4, 2, 1, 1>(ARGS);
      return launch_conv<T, 8, 2, 4, 2, 1, 1>(ARGS);
    }
    // two output-channel blocks per workgroup when there are that many: the input tile is staged once for 64 channels
    // (157 -> 112 us at 128^3 -> 64^3, 32 -> 64; DGTTA_CONV_S2=1: one block per workgroup, the round-1 shape)
    if (dgtta_switches().conv_s2 != '1' && yv.W >= 32 && CoutP % 64 == 0) return launch_conv<T, 32, 4, 2, 2, 2, 1, 8>(ARGS);
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
                  hipStream_t st, double *stats = nullptr, int ntaps_src = 27, RowsGstCtx *gst = nullptr, long long xkh = 0,
                  bool dry = false) {
  ConvClasses cs;
  cs.n = 1;
  cs.kseg = 0;
  cs.acc[0] = accumulate;
  cs.xoff[0] = cs.yoff[0] = 0;
  cs.taps[0] = taps;
  return dispatch_conv_classes<T>(x, xv, w, cs, bias, y, yv, B, Cin, Cout, CinP, CoutP, stride, st, stats, ntaps_src, gst, xkh, dry);
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

int conv3_s2_regs(const void *x, const View &xv, const void *wimg, const float *bias, void *y, const View &yv, int B, int Cin,
                  int Cout, int CinP, int CoutP, double *stats, int64_t cap_slots, int dtype, hipStream_t st);      // conv_s2.hip

int conv3_fwd_mfma(const void *x, int ldx, const void *w_kmajor, int mirror, const float *bias, void *y, int ldy, int B,
                   int Cin, int Cout, int CinP, int CoutP, int Di, int Hi, int Wi, int stride, int dtype,
                   hipStream_t st, double *stats, RowsGstCtx *gst, long long xkh, bool dry) {
  if ((xkh || dry) && (stride != 1 || dtype == DGTTA_F32)) return DGTTA_ERR_UNSUPPORTED;
  const int Do = (Di - 1) / stride + 1, Ho = (Hi - 1) / stride + 1, Wo = (Wi - 1) / stride + 1;
  const View xv = dense_view(B, Di, Hi, Wi, ldx), yv = dense_view(B, Do, Ho, Wo, ldy);
  const Taps taps = identity_taps(mirror);
  if (stride == 2 && !mirror && dtype != DGTTA_F32 && dgtta_switches().conv_s2 != '1' && dgtta_switches().conv_s2 != '4') {
    // the two large encoder transitions: register-operand kernel (conv_s2.hip)
    const int rc = conv3_s2_regs(x, xv, w_kmajor, bias, y, yv, B, Cin, Cout, CinP, CoutP, stats,
                                 conv3_mfma_max_tiles(Do, Ho, Wo), dtype, st);
    if (rc != DGTTA_ERR_UNSUPPORTED) return rc;
  }
  if (dtype == DGTTA_F32) {
    if (!operand_ok<float>(x, ldx, Cin, CinP)) return DGTTA_ERR_UNSUPPORTED;
    return dispatch_conv<float>(x, xv, w_kmajor, taps, bias, y, yv, B, Cin, Cout, CinP, CoutP, stride, 0, st, stats, 27, gst);
  }
  if (dtype == DGTTA_BF16) {
    if (!operand_ok<bf16_t>(x, xkh ? 2 * ldx : ldx, Cin, CinP)) return DGTTA_ERR_UNSUPPORTED;
    return dispatch_conv<bf16_t>(x, xv, w_kmajor, taps, bias, y, yv, B, Cin, Cout, CinP, CoutP, stride, 0, st, stats, 27, gst, xkh, dry);
  }
  if (dtype == DGTTA_F16) {
    if (!operand_ok<f16_t>(x, xkh ? 2 * ldx : ldx, Cin, CinP)) return DGTTA_ERR_UNSUPPORTED;
    return dispatch_conv<f16_t>(x, xv, w_kmajor, taps, bias, y, yv, B, Cin, Cout, CinP, CoutP, stride, 0, st, stats, 27, gst, xkh, dry);
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
    const int ac = dgtta_switches().dgrad_s2_allcls;      // DGTTA_DGRAD_S2_ALLCLS (tests): '0' = the 8-class launch
    if (ac != '0' && CinP % 8 == 0) {
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
  if (dtype == DGTTA_F16) return dgrad_s2<f16_t>(dy, lddy, w_kmajor, dx, lddx, B, Cin, Cout, CinP, CoutP, Di, Hi, Wi, accumulate, st);
  return DGTTA_ERR_UNSUPPORTED;
}

// ConvTranspose3d(k2,s2) forward / data gradient as 8 single-tap launches (one per output offset o).
static size_t n32(int c) { return (size_t)(c + 31) / 32 * 32; }
size_t convT_packed_bytes(int CinP, int CoutP, int dtype) {
  return (size_t)8 * (CinP * n32(CoutP) + CoutP * n32(CinP)) * (dtype == DGTTA_F32 ? 4 : 2);
}
// image-ordered conv weights (imgF | imgB) appended to the [wf | wb] blob
size_t conv_image_bytes(int CinP, int CoutP, int dtype) {
  return (size_t)27 * (CinP * n32(CoutP) + CoutP * n32(CinP)) * (dtype == DGTTA_F32 ? 4 : 2);
}
size_t conv_imgB_offset_bytes(int CinP, int CoutP, int dtype) {
  return (size_t)27 * CinP * n32(CoutP) * (dtype == DGTTA_F32 ? 4 : 2);
}
int conv_pack_images(const float *w_t, void *img, int Cin, int Cout, int CinP, int CoutP, int dtype, hipStream_t st) {
  const int64_t n = (int64_t)27 * (CinP > CoutP ? CinP : CoutP) * n32(CinP > CoutP ? CinP : CoutP);
  const unsigned blocks = (unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  void *imgB = (char *)img + conv_imgB_offset_bytes(CinP, CoutP, dtype);
  if (dtype == DGTTA_F32)
    hipLaunchKernelGGL((conv_pack_image_kernel<float>), dim3(blocks), dim3(256), 0, st, w_t, (float *)img, (float *)imgB, Cin,
                       Cout, CinP, CoutP);
  else if (dtype == DGTTA_BF16)
    hipLaunchKernelGGL((conv_pack_image_kernel<bf16_t>), dim3(blocks), dim3(256), 0, st, w_t, (bf16_t *)img, (bf16_t *)imgB,
                       Cin, Cout, CinP, CoutP);
  else
    hipLaunchKernelGGL((conv_pack_image_kernel<f16_t>), dim3(blocks), dim3(256), 0, st, w_t, (f16_t *)img, (f16_t *)imgB, Cin,
                       Cout, CinP, CoutP);
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

// convt_gemm.hip: register-operand kernels of the two large decoder stages
bool convT_gemm_eligible(int mode, const void *in, int ldin, const void *out, int ldout, int Cin, int Cout, int Wi, int dtype);
int convT_gemm_run(int mode, const void *in, int ldin, const float *w_t, const float *bias, void *out, int ldout, void *ws, int B,
                   int Cin, int Cout, int Di, int Hi, int Wi, int dtype, hipStream_t st);

int convT_fwd_mfma(const void *x, int ldx, const float *w_t, const float *bias, void *out, int ldo, void *ws, int B, int Cin,
                   int Cout, int Di, int Hi, int Wi, int dtype, hipStream_t st) {
  if (ldx >= Cin && ldo >= Cout && convT_gemm_eligible(0, x, ldx, out, ldo, Cin, Cout, Wi, dtype))
    return convT_gemm_run(0, x, ldx, w_t, bias, out, ldo, ws, B, Cin, Cout, Di, Hi, Wi, dtype, st);
  if (dtype == DGTTA_F32) return convT_run<float>(0, x, ldx, w_t, bias, out, ldo, ws, B, Cin, Cout, Di, Hi, Wi, st);
  if (dtype == DGTTA_BF16) return convT_run<bf16_t>(0, x, ldx, w_t, bias, out, ldo, ws, B, Cin, Cout, Di, Hi, Wi, st);
  if (dtype == DGTTA_F16) return convT_run<f16_t>(0, x, ldx, w_t, bias, out, ldo, ws, B, Cin, Cout, Di, Hi, Wi, st);
  return DGTTA_ERR_UNSUPPORTED;
}
int convT_dgrad_mfma(const void *dout, int lddo, const float *w_t, void *dx, int lddx, void *ws, int B, int Cin, int Cout,
                     int Di, int Hi, int Wi, int dtype, hipStream_t st) {
  if (lddo >= Cout && lddx >= Cin && convT_gemm_eligible(1, dout, lddo, dx, lddx, Cin, Cout, Wi, dtype))
    return convT_gemm_run(1, dout, lddo, w_t, nullptr, dx, lddx, ws, B, Cin, Cout, Di, Hi, Wi, dtype, st);
  if (dtype == DGTTA_F32) return convT_run<float>(1, dout, lddo, w_t, nullptr, dx, lddx, ws, B, Cin, Cout, Di, Hi, Wi, st);
  if (dtype == DGTTA_BF16) return convT_run<bf16_t>(1, dout, lddo, w_t, nullptr, dx, lddx, ws, B, Cin, Cout, Di, Hi, Wi, st);
  if (dtype == DGTTA_F16) return convT_run<f16_t>(1, dout, lddo, w_t, nullptr, dx, lddx, ws, B, Cin, Cout, Di, Hi, Wi, st);
  return DGTTA_ERR_UNSUPPORTED;
}

