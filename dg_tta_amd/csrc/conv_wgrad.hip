// Weight gradients of the 3x3x3 / 2x2x2-transposed convolutions and of the head on the matrix cores.
#include "conv_common.h"
#include <stdlib.h>

int conv3_wgrad_ring_launch(const void *x, const View &xv, const void *dy, const View &yv, float *slabs, size_t ws_bytes, int B,
                            int Cin, int Cout, int is_f16, hipStream_t st, int *rc, long long xkh = 0, bool dry = false);      // conv_wgrad_ring.hip

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

// Workgroups are dispatched round robin over the 8 XCDs (each with its own L2), units are columns of the volume whose x
// tiles overlap their H neighbours' by the halo rows: give every XCD a CONTIGUOUS range of units, so that the halo is
// fetched from HBM by one L2 instead of two (DGTTA_WGRAD_XCD=0: units in dispatch order)
__device__ __forceinline__ int xcd_unit(int xcd_map) {
  const int i = blockIdx.x, n = gridDim.x;
  if (!xcd_map || (n & 7)) return i;
  return (i & 7) * (n >> 3) + (i >> 3);
}

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

template <>
__device__ __forceinline__ void transpose_unit<f16_t>(const uint4 *in, uint4 *out) { transpose_unit<bf16_t>(in, out); }

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
__device__ __forceinline__ uint4 shift_run<f16_t>(const uint4 &c, unsigned e0, unsigned e1, int kw) {
  return shift_run<bf16_t>(c, e0, e1, kw);          // pure 16-bit lane moves
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
__device__ __forceinline__ void mfma16<f16_t>(const uint4 &a, const uint4 &b, f32x4_t &acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), acc, 0, 0, 0);
}
// 32x32x16 step on transposed-read operands (kept as bf16x8 bit patterns; the storage type picks the instruction)
template <typename T16>
__device__ __forceinline__ f32x16_t mfma32_tr(const bf16x8_t &a, const bf16x8_t &b, const f32x16_t &acc);
template <>
__device__ __forceinline__ f32x16_t mfma32_tr<bf16_t>(const bf16x8_t &a, const bf16x8_t &b, const f32x16_t &acc) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x16_t mfma32_tr<f16_t>(const bf16x8_t &a, const bf16x8_t &b, const f32x16_t &acc) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), acc, 0, 0, 0);
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
  const int D = yv.D;
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
template <int ABL = 0, bool CLS = false, typename T16 = bf16_t>
__global__ __launch_bounds__(256, 2) void conv3_wgrad_tr_kernel(const bf16_t *__restrict__ x, View xv,
                                                                const bf16_t *__restrict__ dy, View yv,
                                                                float *__restrict__ slabs, int Cin, int Cout, int tilesW,
                                                                int tilesH, int nsd, int DR, int cobs, WgradClasses wc, int upw,
                                                                int units, int xcd_map) {
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

  int tap_id[7], tap_kd[7], tap_off[7];
  int ntap_w = 7;
  // a workgroup sweeps `upw` consecutive units (columns of the volume) into the same accumulators: one slab per
  // workgroup, i.e. upw times fewer partial slabs to write and to reduce
  f32x16_t acc[7];
#pragma unroll
  for (int i = 0; i < 7; ++i)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
  const int cib = blockIdx.y / cobs, cob = blockIdx.y % cobs;
  auto sweep = [&](int t) __attribute__((always_inline)) {
  const int tw = t % tilesW;
  t /= tilesW;
  const int th = t % tilesH;
  t /= tilesH;
  const int ds = t % nsd;
  const int b = t / nsd;
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
    // the tap's ring slot is selected once per slice (round 4: inside the unrolled row loop every operand read carried its own
    // compare / select chain - two transposed reads per MFMA made the sweep issue bound)
    int so_t[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) so_t[i] = (tap_kd[i] == 0 ? slice_off[0] : (tap_kd[i] == 1 ? slice_off[1] : slice_off[2])) + tap_off[i];
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
          afr[i] = tr_operand(sX + lane_off + so_t[i] + oh * WT::X_ROW_B + ks * 1024);
        }
#pragma unroll
        for (int i = 0; i < 7; ++i) {
          if (CLS && i >= ntap_w) continue;
          if (ABL == 3) acc[i][0] += (float)afr[i][0] * (float)bfr[1];
          else acc[i] = mfma32_tr<T16>(afr[i], bfr, acc[i]);
        }
      }
    }
    dma_wait_all();
    lds_barrier();
  }
  };
  if (CLS) {                 // class launches: one unit per workgroup, the body specialised as before
    sweep((int)blockIdx.x);
  } else {
    for (int uu = 0; uu < upw; ++uu) {
      const int t = xcd_unit(xcd_map) * upw + uu;
      if (t >= units) break;
      sweep(t);
    }
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
// Weight gradient of the SMALL planes (W <= 16: the 16^3, 8^3 and 4^3 levels), round 6.  The row kernels above and the ring
// sweep give every output row a 32-voxel MFMA k-run of its own, so a row of 16 / 8 / 4 voxels leaves 50 / 75 / 87 % of the
// matrix instructions multiplying zeros.  Here an (H, W) plane is ONE flat run of slots with row pitch P = W + 2 (the
// two zero-padding voxels of a row are slots of their own): slot g = h P + (w + 1) of dy, and the x operand of tap
// (kh, kw) for slot g is slot g + kh P + kw of the padded x plane - a plain address offset for the transposed LDS reads,
// exactly as a W shift is in conv3_wgrad_tr_kernel.  dy is zero in its pad slots, x in its pad rows / columns (both
// come from the DMA's zero source), so the products of the pad slots vanish and W / (W + 2) = 89 / 80 / 67 % of the
// k dimension is real.  Otherwise the scheme of conv3_wgrad_tr_kernel: LDS-DMA of whole planes into rings (G planes per
// step), a wave owns the 32(ci) x 32(co) tile for 7 of the 27 taps, a workgroup sweeps `upw` units (sample x D segment)
// into one slab.
constexpr int WF_MAXPW = 16;      // DMA pieces (16 slots) per wave and plane
// NCO output-channel blocks per wave: with one, a chunk step reads 8 operands (7 x, 1 dy) from LDS for 7 MFMAs - four waves
// ask for 146 B / clk of a 128 B / clk LDS (measured: 0.39 of the MFMA peak at 16^2); with two the x operand feeds two
// MFMAs: 9 operands for 14 MFMAs, 82 B / clk.  224 accumulator registers: one wave per SIMD, so NCO = 2 is always PIPE.
template <typename T16, bool PIPE, int NCO>
__global__ __launch_bounds__(256, PIPE ? 1 : 2) void conv3_wgrad_flat_kernel(const bf16_t *__restrict__ x, View xv,
                                                                  const bf16_t *__restrict__ dy, View yv,
                                                                  float *__restrict__ slabs, int Cin, int Cout, int cobs,
                                                                  int nseg, int DR, int upw, int units, int P, int NCH, int XS, int G) {
  const int D = yv.D, H = yv.H, W = yv.W;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int YS = NCH * 16;
  unsigned char *sX = smem;                            // ring of 2 G + 2 padded x planes, XS slots of 64 B each
  unsigned char *sY = smem + (2 * G + 2) * XS * 64;    // ring of 2 G dy planes, NCO blocks of YS slots each
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cobs2 = (cobs + NCO - 1) / NCO;
  const int cib = blockIdx.y / cobs2, cob0 = (blockIdx.y % cobs2) * NCO;
  const int cin_lim = (Cin + 7) / 8 * 8;

  int tap_id[7], tap_kd[7], tap_off[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int tc = wave + 4 * i < 27 ? wave + 4 * i : 26;      // (wave 3's seventh slot: a discarded accumulator)
    tap_id[i] = tc;
    tap_kd[i] = tc / 9;
    tap_off[i] = (((tc / 3) % 3) * P + tc % 3) * 64;
  }
  f32x16_t acc[7][NCO];
#pragma unroll
  for (int i = 0; i < 7; ++i)
#pragma unroll
    for (int n = 0; n < NCO; ++n)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][n][q] = 0.f;

  // DMA pieces of a plane: XP pieces of the x plane (buffer slot s = padded-plane slot + 1: the tap (0, 0) of slot 0 reads
  // one slot in front of the plane), then NCH pieces of each dy block; piece idx = wave + 4 i.  Per lane: the element
  // offset of its 16 bytes inside the (sample, depth) plane, -1 = a pad slot (zero source).
  const int XP = XS / 16, NP = XP + NCO * NCH;
  const int l_vox = lane >> 2, l_chunk = lane & 3;
  int poff[WF_MAXPW];
#pragma unroll
  for (int i = 0; i < WF_MAXPW; ++i) {
    const int idx = wave + 4 * i;
    poff[i] = -1;
    if (idx < XP) {
      const int s = idx * 16 + l_vox, q = s > 0 ? s - 1 : 0;
      const int hp = q / P, wp = q - hp * P;
      if (s > 0 && hp >= 1 && hp <= H && wp >= 1 && wp <= W && cib * 32 + l_chunk * 8 < cin_lim)
        poff[i] = (int)((hp - 1) * xv.sh + (wp - 1) * xv.sw) + l_chunk * 8;
    } else if (idx < NP) {
      const int n = (idx - XP) / NCH;
      const int g = (idx - XP - n * NCH) * 16 + l_vox;
      const int h = g / P, wp = g - h * P;
      if (h < H && wp >= 1 && wp <= W && (cob0 + n) * 32 + l_chunk * 8 < Cout)
        poff[i] = (int)(h * yv.sh + (wp - 1) * yv.sw) + n * 32 + l_chunk * 8;
    }
  }
  const int lane_off = ((lane >> 5) * 8 + ((lane & 15) >> 2)) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8;

  // A step of the sweep takes G planes (G = 1, 2, 4).  x ring of 2 G + 2 planes (G + 2 in use: d0 - 1 .. d0 + G, G landing),
  // dy ring of 2 G; ring slots count from the unit's first plane.
  const int NRX = 2 * G + 2, NRY = 2 * G;
  for (int uu = 0; uu < upw; ++uu) {
    const int t = blockIdx.x * upw + uu;
    if (t >= units) break;
    const int b = t / nseg, seg = t - b * nseg;
    const int d_begin = seg * DR, d_end = (d_begin + DR < D) ? d_begin + DR : D;
    const bf16_t *xb = x + b * xv.sb + cib * 32;
    const bf16_t *yb = dy + b * yv.sb + cob0 * 32;
    auto issue_x = [&](int xd, int slot) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < WF_MAXPW; ++i) {
        const int idx = wave + 4 * i;
        if (idx < XP) {
          const bool ok = poff[i] >= 0 && (unsigned)xd < (unsigned)D;
          const void *src = ok ? (const void *)(xb + xd * xv.sd + poff[i]) : (const void *)&g_zero16;
          dma16_to_lds(src, lds_addr_of(sX + (slot * XS + idx * 16) * 64));
        }
      }
    };
    auto issue_y = [&](int yd, int slot) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < WF_MAXPW; ++i) {
        const int idx = wave + 4 * i;
        if (idx >= XP && idx < NP) {
          const void *src = poff[i] >= 0 ? (const void *)(yb + yd * yv.sd + poff[i]) : (const void *)&g_zero16;
          dma16_to_lds(src, lds_addr_of(sY + (slot * NCO * YS + (idx - XP) * 16) * 64));      // (block n at + n YS slots)
        }
      }
    };
    // prologue (the sweep of the previous unit ended with a barrier behind its last reads)
    for (int j = 0; j < G + 2; ++j) issue_x(d_begin - 1 + j, j);
    for (int j = 0; j < G; ++j)
      if (d_begin + j < d_end) issue_y(d_begin + j, j);
    dma_wait_all();
    lds_barrier();
    int xs0 = 0, ys0 = 0;      // ring slots of x plane d0 - 1 and dy plane d0
    for (int d0 = d_begin; d0 < d_end; d0 += G) {
      if (d0 + G < d_end) {
        for (int j = 0; j < G; ++j) {
          int sx = xs0 + G + 2 + j, sy = ys0 + G + j;
          sx -= sx >= NRX ? NRX : 0;
          sy -= sy >= NRY ? NRY : 0;
          issue_x(d0 + G + 1 + j, sx);
          if (d0 + G + j < d_end) issue_y(d0 + G + j, sy);
        }
      }
      for (int j = 0; j < G && d0 + j < d_end; ++j) {
        int sy = ys0 + j;
        sy -= sy >= NRY ? NRY : 0;
        const unsigned char *ys = sY + sy * NCO * YS * 64 + lane_off;
        int so_t[7];
#pragma unroll
        for (int i = 0; i < 7; ++i) {
          int sx = xs0 + j + tap_kd[i];
          sx -= sx >= NRX ? NRX : 0;
          so_t[i] = sx * XS * 64 + tap_off[i] + lane_off;
        }
        // PIPE (one wave per SIMD, nothing else covers the LDS latency): operands of chunk c + 1 are read while the MFMAs
        // of chunk c run, two named register sets (the 512-register budget)
        bf16x8_t a0[7], a1[7], b0[NCO], b1[NCO];
        auto load = [&](int c, bf16x8_t (&a)[7], bf16x8_t (&bb)[NCO]) __attribute__((always_inline)) {
#pragma unroll
          for (int n = 0; n < NCO; ++n) bb[n] = tr_operand(ys + (n * YS + c * 16) * 64);
#pragma unroll
          for (int i = 0; i < 7; ++i) a[i] = tr_operand(sX + so_t[i] + c * 1024);
        };
        auto mm = [&](const bf16x8_t (&a)[7], const bf16x8_t (&bb)[NCO]) __attribute__((always_inline)) {
#pragma unroll
          for (int i = 0; i < 7; ++i)
#pragma unroll
            for (int n = 0; n < NCO; ++n) acc[i][n] = mfma32_tr<T16>(a[i], bb[n], acc[i][n]);
        };
        if (PIPE) {
          load(0, a0, b0);
          int c = 0;
          for (; c + 2 <= NCH; c += 2) {
            load(c + 1, a1, b1);
            mm(a0, b0);
            if (c + 2 < NCH) load(c + 2, a0, b0);
            mm(a1, b1);
          }
          if (c < NCH) mm(a0, b0);
        } else {
          for (int c = 0; c < NCH; ++c) {
            load(c, a0, b0);
            mm(a0, b0);
          }
        }
      }
      xs0 += G;
      xs0 -= xs0 >= NRX ? NRX : 0;
      ys0 += G;
      ys0 -= ys0 >= NRY ? NRY : 0;
      dma_wait_all();
      lds_barrier();
    }
  }

  // partial slabs [27][32 ci][32 co], as conv3_wgrad_tr_kernel writes them: slab blockIdx.x of pair (cib, cob0 + n)
  const int co = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int n = 0; n < NCO; ++n) {
    if (cob0 + n >= cobs) break;
    float *slab = slabs + ((int64_t)(cib * cobs + cob0 + n) * gridDim.x + blockIdx.x) * (27 * 1024);
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      if (wave + 4 * i < 27) {
        const int tap = tap_id[i];
#pragma unroll
        for (int q = 0; q < 16; ++q) slab[(tap * 32 + (q & 3) + 8 * (q >> 2) + 4 * hh) * 32 + co] = acc[i][n][q];
      }
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

template <typename T16 = bf16_t>
__global__ __launch_bounds__(512, 1) void conv3_wgrad_tr8_kernel(const bf16_t *__restrict__ x, View xv,
                                                                 const bf16_t *__restrict__ dy, View yv,
                                                                 float *__restrict__ slabs, int Cin, int Cout, int tilesW,
                                                                 int tilesH, int nsd, int DR, int cobs, int upw, int units, int xcd_map) {
  const int D = yv.D, H = yv.H, W = yv.W;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char *sX = smem;
  unsigned char *sY = smem + 4 * WT::X_SLICE_B;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // a workgroup sweeps `upw` consecutive units (columns of the volume) into the same accumulators: one slab per
  // workgroup, i.e. upw times fewer partial slabs to write and to reduce
  f32x16_t acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][c][q] = 0.f;
  const int cobs2 = (cobs + 1) / 2;
  const int cib = blockIdx.y / cobs2, cob2 = blockIdx.y % cobs2;          // channel-block pair (2 cob2, 2 cob2 + 1)
  for (int uu = 0; uu < upw; ++uu) {
  int t = xcd_unit(xcd_map) * upw + uu;
  if (t >= units) break;
  const int tw = t % tilesW;
  t /= tilesW;
  const int th = t % tilesH;
  t /= tilesH;
  const int ds = t % nsd;
  const int b = t / nsd;
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
    int so_t[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) so_t[i] = (tap_kd[i] == 0 ? slice_off[0] : (tap_kd[i] == 1 ? slice_off[1] : slice_off[2])) + tap_off[i];
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
          afr[i] = tr_operand(sX + lane_off_x + so_t[i] + oh * WT::X_ROW_B + ks * 1024);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int c = 0; c < 2; ++c) acc[i][c] = mfma32_tr<T16>(afr[i], bfr[c], acc[i][c]);
      }
    }
    dma_wait_all();
    lds_barrier();
  }
  }   // units
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
};

template <typename T16 = bf16_t>
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

  int t = xcd_unit(1);
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
    int so_t[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) so_t[i] = (tap_kd[i] == 0 ? slice_off[0] : (tap_kd[i] == 1 ? slice_off[1] : slice_off[2])) + tap_off[i];
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
        const unsigned char *pa = sX + lane_off_x + so_t[i] + 2 * oh * WT2::X_ROW_B;
        const s16x4_t alo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t *)pa);
        const s16x4_t ahi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t *)(pa + 4 * 128));
        const s16x8_t av = {alo[0], alo[1], alo[2], alo[3], ahi[0], ahi[1], ahi[2], ahi[3]};
        afr[i] = __builtin_bit_cast(bf16x8_t, av);
      }
#pragma unroll
      for (int i = 0; i < 7; ++i) acc[i] = mfma32_tr<T16>(afr[i], bfr, acc[i]);
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
// The same for TWO output-channel blocks per workgroup (round 4): 8 waves, waves 0-3 / 4-7 take the 27 taps of block 0 / 1 and
// share the x tile.  With one block per workgroup a 32 -> 64 layer read x - four times the bytes of dy, the whole traffic of this
// HBM-bound kernel - once per block (2.8 GB at the 128^3 -> 64^3 transition where 1.3 GB are the operands; 853 us).  One
// 512-thread workgroup per CU halves the loads in flight, so the ring is two output slices ahead instead of one (7 x slots,
// 3 dy slots) and a step waits with a counted vmcnt for the slices of the NEXT step only; every wave issues the same five
// pieces per step (a wave without a piece of its own repeats a neighbour's: same bytes to the same address).
struct WT2X {
  static constexpr int NCO = 2, LA = 2;
  static constexpr int NXS = 3 + 2 * LA, NYS = LA + 1;
  static constexpr int Y_STEP_B = NCO * WT2::Y_SLICE_B;
  static constexpr int LDS_BYTES = NXS * WT2::X_SLICE_B + NYS * Y_STEP_B;
};

template <typename T16 = bf16_t>
__global__ __launch_bounds__(512, 2) void conv3_wgrad_tr_s2x_kernel(const bf16_t *__restrict__ x, View xv,
                                                                    const bf16_t *__restrict__ dy, View yv,
                                                                    float *__restrict__ slabs, int Cin, int Cout, int tilesW,
                                                                    int tilesH, int nsd, int DR, int cobs) {
  const int D = yv.D, H = yv.H, W = yv.W;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char *sX = smem;
  unsigned char *sY = smem + WT2X::NXS * WT2::X_SLICE_B;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tg = wave & 3, cb = wave >> 2;                 // tap group, output-channel block of the pair

  int t = xcd_unit(1);
  const int tw = t % tilesW;
  t /= tilesW;
  const int th = t % tilesH;
  t /= tilesH;
  const int ds = t % nsd;
  const int b = t / nsd;
  const int cogs = cobs / 2;
  const int cib = blockIdx.y / cogs, cog = blockIdx.y % cogs;
  const int h0 = th * WT2::TH, w0 = tw * WT2::TWO;
  const int d_begin = ds * DR, d_end = (d_begin + DR < D) ? d_begin + DR : D;
  const bf16_t *xb = x + b * xv.sb + cib * 32;
  const bf16_t *yb = dy + b * yv.sb + cog * 64;
  const int cin_lim = (Cin + 7) / 8 * 8;
  const int l_vox = lane >> 2, l_chunk = lane & 3;

  auto xslot = [&](int xd) { return (xd + WT2X::NXS) % WT2X::NXS; };
  // 15 pieces per x slice over 8 waves: pieces wave and wave + 8 (the 16th repeats piece 14)
  auto issue_x_slice = [&](int xd) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int idx = wave + 8 * i;
      idx = idx < WT2::NPX1 ? idx : WT2::NPX1 - 1;
      const int r = idx / 3, pi = idx % 3;
      const int gh = 2 * h0 - 1 + r, wx = 16 * pi + l_vox, gw = 2 * w0 - 1 + wx;
      // the third piece of a row is one voxel (4 lanes): the other lanes write zeros into the unused tail of the LDS row
      const bool ok = (unsigned)xd < (unsigned)xv.D && (unsigned)gh < (unsigned)xv.H && (unsigned)gw < (unsigned)xv.W &&
                      cib * 32 + l_chunk * 8 < cin_lim && (pi < 2 || lane < 4);
      const void *src = ok ? (const void *)(xb + xd * xv.sd + gh * xv.sh + gw * xv.sw + l_chunk * 8) : (const void *)&g_zero16;
      if (pi < 2 || lane < 12)      // voxels 32 .. 34 of the 36-voxel LDS row
        dma16_to_lds(src, lds_addr_of(sX + xslot(xd) * WT2::X_SLICE_B + r * WT2::X_ROW_B + pi * 1024));
    }
  };
  // 4 pieces per dy slice (2 rows x 2 blocks): piece wave & 3
  auto issue_y_slice = [&](int yd) __attribute__((always_inline)) {
    const int row = wave & 1, blk = (wave >> 1) & 1;
    const int gh = h0 + row, gw = w0 + l_vox;
    const bool ok = (unsigned)yd < (unsigned)D && gh < H && gw < W && cog * 64 + blk * 32 + l_chunk * 8 < Cout;
    const void *src = ok ? (const void *)(yb + yd * yv.sd + gh * yv.sh + gw * yv.sw + blk * 32 + l_chunk * 8)
                         : (const void *)&g_zero16;
    dma16_to_lds(src, lds_addr_of(sY + (yd % WT2X::NYS) * WT2X::Y_STEP_B + blk * WT2::Y_SLICE_B + row * WT2::Y_ROW_B));
  };
  auto issue_step = [&](int od) __attribute__((always_inline)) {      // x slices 2 od, 2 od + 1 and dy slice od: 5 pieces per wave
    issue_x_slice(2 * od);
    issue_x_slice(2 * od + 1);
    issue_y_slice(od);
  };

  const int kq = (lane >> 5) * 8 + ((lane & 15) >> 2), cpart = ((lane >> 4) & 1) * 32 + (lane & 3) * 8;
  const int lane_off_y = kq * 64 + cpart, lane_off_x = kq * 128 + cpart;

  int tap_kd[7], tap_off[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int tc = tg + 4 * i < 27 ? tg + 4 * i : 26;
    tap_kd[i] = tc / 9;
    tap_off[i] = ((tc / 3) % 3) * WT2::X_ROW_B + (tc % 3) * 64;
  }
  f32x16_t acc[7];
#pragma unroll
  for (int i = 0; i < 7; ++i)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;

  // prologue: x slice 2 d_begin - 1, then the steps d_begin and d_begin + 1
  issue_x_slice(2 * d_begin - 1);
  issue_step(d_begin);
  if (d_begin + 1 < d_end) issue_step(d_begin + 1);
  dma_wait_all();
  lds_barrier();

  for (int d = d_begin; d < d_end; ++d) {
    const bool ahead = d + WT2X::LA < d_end;
    if (ahead) issue_step(d + WT2X::LA);      // lands during this step and the next
    const unsigned char *ys = sY + (d % WT2X::NYS) * WT2X::Y_STEP_B + cb * WT2::Y_SLICE_B + lane_off_y;
    int slice_off[3];
#pragma unroll
    for (int kd = 0; kd < 3; ++kd) slice_off[kd] = xslot(2 * d + kd - 1) * WT2::X_SLICE_B;
    int so_t[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) so_t[i] = (tap_kd[i] == 0 ? slice_off[0] : (tap_kd[i] == 1 ? slice_off[1] : slice_off[2])) + tap_off[i];
#pragma unroll
    for (int oh = 0; oh < WT2::TH; ++oh) {
      const s16x4_t blo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t *)(ys + oh * WT2::Y_ROW_B));
      const s16x4_t bhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t *)(ys + oh * WT2::Y_ROW_B + 4 * 64));
      typedef __attribute__((ext_vector_type(8))) short s16x8_t;
      const s16x8_t bv = {blo[0], blo[1], blo[2], blo[3], bhi[0], bhi[1], bhi[2], bhi[3]};
      const bf16x8_t bfr = __builtin_bit_cast(bf16x8_t, bv);
      bf16x8_t afr[7];
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        const unsigned char *pa = sX + lane_off_x + so_t[i] + 2 * oh * WT2::X_ROW_B;
        const s16x4_t alo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t *)pa);
        const s16x4_t ahi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t *)(pa + 4 * 128));
        const s16x8_t av = {alo[0], alo[1], alo[2], alo[3], ahi[0], ahi[1], ahi[2], ahi[3]};
        afr[i] = __builtin_bit_cast(bf16x8_t, av);
      }
#pragma unroll
      for (int i = 0; i < 7; ++i) acc[i] = mfma32_tr<T16>(afr[i], bfr, acc[i]);
    }
    // the slices of step d + 1 were issued one step ago: everything but the five pieces issued above must have landed
    if (ahead) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else dma_wait_all();
    lds_barrier();
  }

  const int pair = cib * cobs + cog * 2 + cb;
  float *slab = slabs + ((int64_t)pair * gridDim.x + blockIdx.x) * (27 * 1024);
  const int co = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int tap = tg + 4 * i;
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
template <int NCI>
struct WT3 {
  static constexpr int TH = 2, TWI = 16;
  static constexpr int X_ROW_B = TWI * 64, X_BLK_B = TH * X_ROW_B;               // one 32-channel block of an x slice: 2 KiB
  static constexpr int X_SLICE_B = NCI * X_BLK_B;
  static constexpr int Y_ROW_B = 2 * TWI * 64, Y_SLICE_B = 2 * TH * Y_ROW_B;     // one dout slice: 4 rows x 2 KiB
  static constexpr int Y_PAIR_B = 2 * Y_SLICE_B;                                 // dout slices 2d, 2d+1
  static constexpr int LDS_BYTES = 2 * X_SLICE_B + 2 * Y_PAIR_B;
  static constexpr int NPY = 2 * 2 * TH * 2;                                     // dout pieces per x slice (16 KiB)
};

// NCI (round 4): input-channel blocks of 32 per workgroup.  With one block per workgroup a 64-channel layer read dout - four
// times the bytes of x, the whole traffic of this HBM-bound kernel - once per block: 2.4 GB instead of 1.3 GB at the
// 64^3 -> 128^3 stage (836 us at 2.9 TB/s).  NCI = 2 shares the dout tile between two blocks of x.
template <typename T16 = bf16_t, int NCI = 1>
__global__ __launch_bounds__(256, 2) void convT_wgrad_tr_kernel(const bf16_t *__restrict__ x, View xv,
                                                                const bf16_t *__restrict__ dout, View yv,
                                                                float *__restrict__ slabs, int Cin, int Cout, int tilesW,
                                                                int tilesH, int nsd, int DR, int cobs, int cibs,
                                                                float *__restrict__ bias_part) {
  typedef WT3<NCI> WT;
  const int D = xv.D, H = xv.H, W = xv.W;                  // input lattice; yv = dense view of dout (2D x 2H x 2W)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char *sX = smem;                                // 2 slots
  unsigned char *sY = smem + 2 * WT::X_SLICE_B;            // 2 slots of a slice pair
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int t = xcd_unit(1);
  const int tw = t % tilesW;
  t /= tilesW;
  const int th = t % tilesH;
  t /= tilesH;
  const int ds = t % nsd;
  const int b = t / nsd;
  const int cig = blockIdx.y / cobs, cob = blockIdx.y % cobs;      // group of NCI input-channel blocks
  const int h0 = th * WT::TH, w0 = tw * WT::TWI;
  const int d_begin = ds * DR, d_end = (d_begin + DR < D) ? d_begin + DR : D;
  const bf16_t *xb = x + b * xv.sb + cig * NCI * 32;
  const bf16_t *yb = dout + b * yv.sb + cob * 32;
  const int cin_lim = (Cin + 7) / 8 * 8;
  const int l_vox = lane >> 2, l_chunk = lane & 3;

  // pieces of x slice d: NCI blocks x 2 rows (one per wave while they last); pieces of the dout pair: 2 slices x 4 rows x 2
  // halves = 16 (4 per wave)
  auto issue = [&](int d) __attribute__((always_inline)) {
    if (wave < WT::TH * NCI) {
      const int row = wave % WT::TH, blk = wave / WT::TH;
      const int gh = h0 + row, gw = w0 + l_vox;
      const bool ok = (unsigned)d < (unsigned)D && gh < H && gw < W && (cig * NCI + blk) * 32 + l_chunk * 8 < cin_lim;
      const void *src = ok ? (const void *)(xb + d * xv.sd + gh * xv.sh + gw * xv.sw + blk * 32 + l_chunk * 8)
                           : (const void *)&g_zero16;
      dma16_to_lds(src, lds_addr_of(sX + (d & 1) * WT::X_SLICE_B + blk * WT::X_BLK_B + row * WT::X_ROW_B));
    }
#pragma unroll
    for (int i = 0; i < WT::NPY / 4; ++i) {
      const int idx = wave + 4 * i;                     // (slice s, row r, half pi)
      const int sl = idx >> 3, r = (idx >> 1) & 3, pi = idx & 1;
      const int gd = 2 * d + sl, gh = 2 * h0 + r, gw = 2 * w0 + 16 * pi + l_vox;
      const bool ok = (unsigned)d < (unsigned)D && gd < yv.D && gh < yv.H && gw < yv.W && cob * 32 + l_chunk * 8 < Cout;
      const void *src = ok ? (const void *)(yb + gd * yv.sd + gh * yv.sh + gw * yv.sw + l_chunk * 8) : (const void *)&g_zero16;
      dma16_to_lds(src, lds_addr_of(sY + (d & 1) * WT::Y_PAIR_B + sl * WT::Y_SLICE_B + r * WT::Y_ROW_B + pi * 1024));
    }
  };

  const int kq = (lane >> 5) * 8 + ((lane & 15) >> 2), cpart = ((lane >> 4) & 1) * 32 + (lane & 3) * 8;
  const int lane_off_x = kq * 64 + cpart, lane_off_y = kq * 128 + cpart;
  // this wave's two output offsets o = 2 wave, 2 wave + 1  (o = od*4 + oh*2 + ow), for every input-channel block
  f32x16_t acc[NCI][2];
#pragma unroll
  for (int c = 0; c < NCI; ++c)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[c][i][q] = 0.f;
  int ooff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int o = 2 * wave + i;
    ooff[i] = (o >> 2) * WT::Y_SLICE_B + ((o >> 1) & 1) * WT::Y_ROW_B + (o & 1) * 64;
  }
  // bias gradient sum_v dout[v][co] on the side (round 4; the first input-channel group's workgroups only): an x operand that is 1
  // in row 0 and 0 elsewhere leaves the column sums of the dout fragments in row 0 of a third accumulator pair - the pass
  // over dout that chan_reduce_vec_kernel<., 2> made for them (1.5 ms per epoch) is not needed
  typedef __attribute__((ext_vector_type(8))) short s16x8_t;
  const bool do_bias = bias_part != nullptr && cig == 0;
  const short one16 = sizeof(T16) == 2 && std::is_same<T16, f16_t>::value ? (short)0x3C00 : (short)0x3F80;
  const short o1 = (lane & 31) == 0 ? one16 : (short)0;
  const s16x8_t onesv = {o1, o1, o1, o1, o1, o1, o1, o1};
  const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, onesv);
  f32x16_t bacc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int q = 0; q < 16; ++q) bacc[i][q] = 0.f;

  issue(d_begin);
  dma_wait_all();
  lds_barrier();
  for (int d = d_begin; d < d_end; ++d) {
    if (d + 1 < d_end) issue(d + 1);
    const unsigned char *xs = sX + (d & 1) * WT::X_SLICE_B + lane_off_x;
    const unsigned char *ys = sY + (d & 1) * WT::Y_PAIR_B + lane_off_y;
#pragma unroll
    for (int r = 0; r < WT::TH; ++r) {
      bf16x8_t bfr[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const unsigned char *pb = ys + ooff[i] + 2 * r * WT::Y_ROW_B;
        const s16x4_t blo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t *)pb);
        const s16x4_t bhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t *)(pb + 4 * 128));
        const s16x8_t bv = {blo[0], blo[1], blo[2], blo[3], bhi[0], bhi[1], bhi[2], bhi[3]};
        bfr[i] = __builtin_bit_cast(bf16x8_t, bv);
      }
      if (do_bias) {
#pragma unroll
        for (int i = 0; i < 2; ++i) bacc[i] = mfma32_tr<T16>(ones, bfr[i], bacc[i]);
      }
#pragma unroll
      for (int c = 0; c < NCI; ++c) {
        const unsigned char *pa = xs + c * WT::X_BLK_B + r * WT::X_ROW_B;
        const s16x4_t alo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t *)pa);
        const s16x4_t ahi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t *)(pa + 4 * 64));
        const s16x8_t av = {alo[0], alo[1], alo[2], alo[3], ahi[0], ahi[1], ahi[2], ahi[3]};
        const bf16x8_t afr = __builtin_bit_cast(bf16x8_t, av);
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[c][i] = mfma32_tr<T16>(afr, bfr[i], acc[c][i]);
      }
    }
    dma_wait_all();
    lds_barrier();
  }
  // slab "tap" slot = output offset o; one slab per (input-channel block, output-channel block) pair and unit
  const int co = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int c = 0; c < NCI; ++c) {
    const int cib = cig * NCI + c;
    if (cib < cibs) {
      float *slab = slabs + ((int64_t)(cib * cobs + cob) * gridDim.x + blockIdx.x) * (27 * 1024);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int o = 2 * wave + i;
#pragma unroll
        for (int q = 0; q < 16; ++q) slab[(o * 32 + (q & 3) + 8 * (q >> 2) + 4 * hh) * 32 + co] = acc[c][i][q];
      }
    }
  }
  if (do_bias) {      // row 0 of the accumulator = lanes 0..31, element 0; offsets, then waves, in order
    float *red = reinterpret_cast<float *>(smem);
    if (lane < 32) red[wave * 32 + lane] = bacc[0][0] + bacc[1][0];
    __syncthreads();
    if (tid < 32) bias_part[(int64_t)blockIdx.x * (cobs * 32) + cob * 32 + tid] = ((red[tid] + red[32 + tid]) + red[64 + tid]) + red[96 + tid];
  }
}

// sums the per-unit bias partials of convT_wgrad_tr_kernel: one workgroup per 32 output channels, 8 unit groups (unit mod 8)
// with eight loads in flight each, the groups added in order (double): a fixed summation order
__global__ __launch_bounds__(256) void convT_bias_finalize_kernel(const float *__restrict__ part, int units, int ldp, int Cout,
                                                                  float *__restrict__ db, int accumulate) {
  __shared__ double red[8][32];
  const int co = blockIdx.x * 32 + (threadIdx.x & 31), grp = threadIdx.x >> 5;
  double s = 0.0;
  for (int u = grp; u < units; u += 64) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = u + 8 * j < units ? part[(int64_t)(u + 8 * j) * ldp + co] : 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (u + 8 * j < units) s += (double)v[j];
  }
  red[grp][threadIdx.x & 31] = s;
  __syncthreads();
  if (threadIdx.x < 32 && co < Cout) {
    double t = 0.0;
#pragma unroll
    for (int g = 0; g < 8; ++g) t += red[g][threadIdx.x];
    db[co] = accumulate ? db[co] + (float)t : (float)t;
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
    // eight slabs are requested before the first is added (round 4): one dependent load per slab made the launch a chain
    // of nslab / G memory round trips (32 us for 256 slabs).  The additions keep their order: same sums, bit for bit.
    for (int k = grp; k < nslab; k += 8 * G) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (k + j * G < nslab) ? p[(int64_t)(k + j * G) * (27 * 1024)] : 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (k + j * G < nslab) s += v[j];
    }
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

// Few slabs (< 64) into the dense layout dw_t[co][ci][27], round 6.  The kernel above gives each output row (tap, ci, co-block)
// to 32 lanes: coalesced slab reads, but the 27 floats of a (ci, co) are written by workgroups far apart in the grid, as
// 4-byte stores 100+ KB apart, so every 128-byte line of dw goes to memory in pieces (the 22 MB of the 640 -> 320 layer took
// 77 us, 4x the 11 MB of a 320 -> 320 layer).  Here a workgroup owns 8 input channels x 32 output channels: thread (ci, co)
// sums its 27 taps over the slabs (9 taps x 4 slabs in flight; slabs in ascending order: the same bits as above), the sums
// meet in LDS, and each output channel's 8 x 27 = 216 consecutive floats leave as 54 float4 of one wave instruction.
__global__ __launch_bounds__(256) void wgrad_reduce_taps_kernel(const float *__restrict__ slabs, float *__restrict__ dw, int Cin,
                                                                int Cout, int cobs, int nslab, int accumulate, long long s_co) {
  __shared__ float stg[32][217];
  const int lane = threadIdx.x & 31, sub = threadIdx.x >> 5;
  const int cobs32 = (Cout + 31) / 32;
  const int cb = blockIdx.x % cobs32, ci0 = (blockIdx.x / cobs32) * 8;      // (Cin % 8 == 0: the launcher checks)
  const int ci = ci0 + sub;
  const float *p = slabs + (int64_t)((ci >> 5) * cobs + cb) * nslab * (27 * 1024) + (ci & 31) * 32 + lane;
#pragma unroll 1
  for (int t0 = 0; t0 < 27; t0 += 9) {
    float s[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) s[t] = 0.f;
    for (int k = 0; k < nslab; k += 4) {
      float v[9][4];
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) v[t][j] = p[(int64_t)(k + j < nslab ? k + j : k) * (27 * 1024) + (t0 + t) * 1024];
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (k + j < nslab) s[t] += v[t][j];
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) stg[lane][sub * 27 + t0 + t] = s[t];
  }
  __syncthreads();
  const int wv = threadIdx.x >> 6, l = threadIdx.x & 63;
  if (l < 54) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int cl = wv + 4 * r, co = cb * 32 + cl;
      if (co >= Cout) break;
      float4 o = make_float4(stg[cl][4 * l], stg[cl][4 * l + 1], stg[cl][4 * l + 2], stg[cl][4 * l + 3]);
      float4 *dst = (float4 *)(dw + co * s_co + (int64_t)ci0 * 27) + l;
      if (accumulate) {
        const float4 old = *dst;
        o.x += old.x, o.y += old.y, o.z += old.z, o.w += old.w;
      }
      *dst = o;
    }
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
  int want = (int)cdiv64(512, base > 0 ? base : 1);   // aim for >= ~512 workgroups (2 per CU) over all classes
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
  int want = (int)cdiv64(512, base > 0 ? base : 1);
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

// units a workgroup sweeps into one slab: as many as keep >= `slots` workgroups in the launch (at most 4; DGTTA_WGRAD_UPW=1:
// one unit per workgroup, the round-2a partition; =2..4: forced, for the tests)
static int units_per_workgroup(int64_t units, int64_t gy, int slots) {
  const int sw = dgtta_switches().wgrad_upw;
  if (sw == '1') return 1;
  if (sw >= '2' && sw <= '4') return (int)(units < sw - '0' ? units : sw - '0');      // forced (tests)
  int64_t u = units * gy / slots;
  if (u > 4) u = 4;
  if (u > units) u = units;
  return u < 1 ? 1 : (int)u;
}

// the dense-layout reduction (wgrad_reduce_taps_kernel) takes: all 27 taps in place, dw_t[co][ci][27], whole groups of 8 input
// channels, 16-byte aligned rows (DGTTA_WGRAD_REDUCE_TAPS=0, tests: always the row kernel)
static bool reduce_taps_ok(const Taps *real, const float *dw, int Cin, long long s_ci, long long s_tap) {
  if (!real || dgtta_switches().wgrad_reduce_taps == '0' || s_tap != 1 || s_ci != 27 || Cin % 8 || ((uintptr_t)dw & 15)) return false;
  for (int t = 0; t < 27; ++t)
    if (real->wt[t] != t) return false;
  return true;
}

// launch plan of conv3_wgrad_flat_kernel; returns the number of slabs per channel-block pair (0: shape not taken)
template <typename T16>
static int64_t wgrad_flat_launch(const void *x, const View &xv, const void *dy, const View &yv, float *slabs, size_t ws_bytes,
                                 int B, int Cin, int Cout, hipStream_t st, int *rc) {
  *rc = DGTTA_OK;
  const int D = yv.D, H = yv.H, W = yv.W;
  if (W > 16 || xv.D != D || xv.H != H || xv.W != W) return 0;
  const int P = W + 2, NCH = cdiv(H * P, 16);
  int XS = NCH * 16 + 2 * P + 2;                  // last slot a tap reads: (NCH 16 - 1) + 2 P + 2
  if (XS < (H + 2) * P + 1) XS = (H + 2) * P + 1;
  XS = (XS + 15) / 16 * 16;
  const int cibs = cdiv(Cin, 32), cobs = cdiv(Cout, 32), pairs = cibs * cobs;
  if (pairs > 65535) return 0;
  // planes per step: as many (4, 2, 1) as fit.  Small planes (8^2, 4^2: a plane set fits twice into a CU's LDS): one
  // output-channel block per wave, two workgroups per CU (measured 43 / 76 / 24 us against 58 / 97 / 35 us with two blocks per
  // wave on the 320 -> 320 and 640 -> 320 layers at 8^3 and 320 -> 320 at 4^3).  Larger planes: one workgroup per CU either
  // way, two blocks per wave where there are two (16^3, 256 -> 256: 123 against 133 us)
  auto lds_of = [&](int nco, int g) { return ((2 * g + 2) * XS + 2 * g * nco * NCH * 16) * 64; };
  int NCO = 1, G = 4;
  while (G > 1 && (lds_of(1, G) > 80 * 1024 || G > D)) G /= 2;
  if (lds_of(1, G) > 80 * 1024 && cobs >= 2) {
    int g2 = 4;
    while (g2 > 1 && (lds_of(2, g2) > 160 * 1024 || g2 > D)) g2 /= 2;
    if (lds_of(2, g2) <= 160 * 1024 && cdiv(XS / 16 + 2 * NCH, 4) <= WF_MAXPW) NCO = 2, G = g2;
  }
  const int lds = lds_of(NCO, G);
  if (lds > 160 * 1024 || cdiv(XS / 16 + NCO * NCH, 4) > WF_MAXPW) return 0;
  if ((long long)H * xv.sh >= (1ll << 30) || (long long)H * yv.sh >= (1ll << 30)) return 0;      // plane offsets as int
  const bool pipe = NCO == 2 || lds > 80 * 1024;
  const int wgs_y = cibs * cdiv(cobs, NCO);
  static int ncu = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n > 0 ? n : 256;
  }();
  // slabs per pair: fill the chip once (two workgroups per CU in the non-PIPE form); every slab is 110 KB written and read
  // again, so no more than that
  int64_t want = (int64_t)ncu * (pipe ? 1 : 2) / wgs_y;
  const int64_t fit = (int64_t)(ws_bytes / ((size_t)pairs * 27 * 1024 * sizeof(float)));
  if (fit < 1) return 0;
  if (want > fit) want = fit;
  if (want < 1) want = 1;
  int nseg = 1;                                   // D segments only when the batch alone gives too few units
  if (B < want) nseg = (int)(cdiv64(want, B) < D ? cdiv64(want, B) : D);
  const int DR = cdiv(D, nseg);
  nseg = cdiv(D, DR);
  const int units = B * nseg;
  const int upw = cdiv(units, (int)(want < units ? want : units));
  const int nslab = cdiv(units, upw);
  auto kern = NCO == 2 ? conv3_wgrad_flat_kernel<T16, true, 2>
                       : (pipe ? conv3_wgrad_flat_kernel<T16, true, 1> : conv3_wgrad_flat_kernel<T16, false, 1>);
  static DynLdsOnce once[3];
  if (ensure_dyn_lds(once[NCO == 2 ? 2 : (int)pipe], reinterpret_cast<const void *>(kern), pipe ? 160 * 1024 : 80 * 1024) != hipSuccess) {
    dgtta_set_error("wgrad_flat: cannot raise the dynamic LDS limit");
    *rc = DGTTA_ERR_LAUNCH;
    return 0;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)nslab, (unsigned)wgs_y), dim3(256), lds, st, (const bf16_t *)x, xv,
                     (const bf16_t *)dy, yv, slabs, Cin, Cout, cobs, nseg, DR, upw, units, P, NCH, XS, G);
  if (hipGetLastError() != hipSuccess) {
    dgtta_set_error("conv3_wgrad_flat_kernel: launch failed");
    *rc = DGTTA_ERR_LAUNCH;
    return 0;
  }
  return nslab;
}

template <typename T>
static int wgrad_launch_classes(const void *x, const View &xv, const void *dy, const View &yv, float *dw, void *ws,
                                size_t ws_bytes, int B, int Cin, int Cout, const WgradClasses &wc, const RealTaps &reals,
                                long long s_co, long long s_ci, long long s_tap, int accumulate, hipStream_t st, long long xkh = 0,
                                bool split_leg = false) {
  constexpr int EPV = Elem<T>::EPV;
  // Cin may be ragged (first layer: 12 channels in rows of 16): the pad channels only feed gradient rows ci >= Cin,
  // which the reduction never writes.  The rows must be long enough to be read in whole 16-byte groups.
  if (Cout % EPV || xv.sw % EPV || yv.sw % EPV || ((uintptr_t)x & 15) || ((uintptr_t)dy & 15) ||
      xv.sw < (xkh ? 32 : (Cin + EPV - 1) / EPV * EPV))      // (x as 32-channel planes: rows of one block)
    return DGTTA_ERR_UNSUPPORTED;
  for (int c = 0; c < wc.n; ++c)
    if ((wc.xoff[c] * (long long)sizeof(T)) % 16 || (wc.yoff[c] * (long long)sizeof(T)) % 16) return DGTTA_ERR_UNSUPPORTED;
  if (xkh && (sizeof(T) != 2 || dgtta_switches().wgrad_tr == '0' || dgtta_switches().wgrad_ring == '0')) return DGTTA_ERR_UNSUPPORTED;
  WgradPlan p = wgrad_plan(B, Cin, Cout, yv.D, yv.H, yv.W, wc.n);
  const size_t need = (size_t)wc.n * p.units * p.cibs * p.cobs * 27 * 1024 * sizeof(float);
  if (ws_bytes < need || p.units >= (1ll << 31) || p.cibs * p.cobs > 65535) return DGTTA_ERR_UNSUPPORTED;
  int upw = 1;
  int64_t nslab = p.units;            // partial slabs per (channel-block pair, class)
  if constexpr (sizeof(T) == 2) {
    typedef T T16;
    const DgttaSwitches &sw = dgtta_switches();
    if (sw.wgrad_tr != '0') {        // DGTTA_WGRAD_TR=0 (tests): the register-transpose predecessor
      const bool plain = wc.n == 1 && wc.mask[0] == 0x7ffffffu && wc.xoff[0] == 0 && wc.yoff[0] == 0;
      // planes of W <= 16 as flat runs (DGTTA_WGRAD_FLAT=0: the kernels below).  Not for the six launches of an fp32 weight
      // gradient (split_leg): that path's fixtures compare label maps and Adam update signs of near-tied values bit for bit
      // with the reference run, i.e. they are pinned on the summation order of the kernels below
      if (plain && !xkh && !split_leg && sw.wgrad_flat != '0') {
        int rc = DGTTA_OK;
        const int64_t g = wgrad_flat_launch<T16>(x, xv, dy, yv, (float *)ws, ws_bytes, B, Cin, Cout, st, &rc);
        if (rc != DGTTA_OK) return rc;
        if (g > 0) {
          nslab = g;
          goto reduce;
        }
      }
      if (plain && sw.wgrad_ring != '0') {      // the persistent ring sweep (conv_wgrad_ring.hip; DGTTA_WGRAD_RING=0: its predecessors)
        int rc = DGTTA_OK;
        const int g = conv3_wgrad_ring_launch(x, xv, dy, yv, (float *)ws, ws_bytes, B, Cin, Cout, (int)std::is_same<T16, f16_t>::value,
                                              st, &rc, xkh);
        if (rc != DGTTA_OK) return rc;
        if (g > 0) {
          nslab = g;
          goto reduce;
        }
      }
      if (xkh) return DGTTA_ERR_UNSUPPORTED;      // only the ring sweep reads x as 32-channel planes
      auto ktr = plain ? conv3_wgrad_tr_kernel<0, false, T16> : conv3_wgrad_tr_kernel<0, true, T16>;
      static DynLdsOnce tr_once[2];
      DG_REQUIRE(ensure_dyn_lds(tr_once[plain], reinterpret_cast<const void *>(ktr), (int)WT::LDS_BYTES) == hipSuccess,
                 DGTTA_ERR_LAUNCH, "wgrad_tr: cannot raise the dynamic LDS limit");
      if (plain && Cout >= 64 && sw.wgrad_tr8 != '0') {      // DGTTA_WGRAD_TR8=0 (tests): always the 4-wave kernel
        static DynLdsOnce a8;
        DG_REQUIRE(ensure_dyn_lds(a8, reinterpret_cast<const void *>(conv3_wgrad_tr8_kernel<T16>), (int)WT8::LDS_BYTES) ==
                       hipSuccess, DGTTA_ERR_LAUNCH, "wgrad_tr8: cannot raise the dynamic LDS limit");
        const int64_t gy = (int64_t)p.cibs * ((p.cobs + 1) / 2);
        upw = units_per_workgroup(p.units, gy, 256);          // one 8-wave workgroup per CU
        nslab = cdiv64(p.units, upw);
        hipLaunchKernelGGL(conv3_wgrad_tr8_kernel<T16>, dim3((unsigned)nslab, (unsigned)gy), dim3(512),
                           WT8::LDS_BYTES, st, (const bf16_t *)x, xv, (const bf16_t *)dy, yv, (float *)ws, Cin, Cout, p.tW, p.tH,
                           p.nsd, p.DR, p.cobs, upw, (int)p.units, dgtta_switches().wgrad_xcd != '0');
        DG_CHECK_LAUNCH("conv3_wgrad_tr8_kernel");
        goto reduce;
      }
      // two 4-wave workgroups per CU; class launches keep one unit per workgroup (their classes carry 1..8 taps: many
      // small workgroups balance that, 4 units each measured 1.5x slower)
      upw = plain ? units_per_workgroup(p.units, (int64_t)p.cibs * p.cobs, 512) : 1;
      nslab = cdiv64(p.units, upw);
      hipLaunchKernelGGL(ktr, dim3((unsigned)nslab, (unsigned)(p.cibs * p.cobs), (unsigned)wc.n), dim3(256), WT::LDS_BYTES,
                         st, (const bf16_t *)x, xv, (const bf16_t *)dy, yv, (float *)ws, Cin, Cout, p.tW, p.tH, p.nsd,
                         p.DR, p.cobs, wc, upw, (int)p.units, dgtta_switches().wgrad_xcd != '0');
      DG_CHECK_LAUNCH("conv3_wgrad_tr_kernel");
      goto reduce;
    }
  }
  {
  auto kern = conv3_wgrad_mfma_kernel<T, 0>;
  static DynLdsOnce mf_once;
  DG_REQUIRE(ensure_dyn_lds(mf_once, reinterpret_cast<const void *>(kern), (int)WG<T>::LDS_BYTES) == hipSuccess,
             DGTTA_ERR_LAUNCH, "wgrad_mfma: cannot raise the dynamic LDS limit");
  hipLaunchKernelGGL(kern, dim3((unsigned)p.units, (unsigned)(p.cibs * p.cobs), (unsigned)wc.n), dim3(256), WG<T>::LDS_BYTES,
                     st, (const T *)x, xv, (const T *)dy, yv, (float *)ws, Cin, Cout, p.tW, p.tH, p.nsd, p.DR, p.cobs, wc);
  DG_CHECK_LAUNCH("conv3_wgrad_mfma_kernel");
  }
reduce:
  const int64_t rrows = (int64_t)27 * Cin * ((Cout + 31) / 32);
  const int npairs = p.cibs * p.cobs;
  if (nslab >= 64)
    hipLaunchKernelGGL(wgrad_reduce_kernel<8>, dim3((unsigned)rrows, (unsigned)wc.n), dim3(256), 0, st, (const float *)ws, dw,
                       Cin, Cout, p.cobs, npairs, (int)nslab, accumulate, reals, s_co, s_ci, s_tap);
  else if (reduce_taps_ok(wc.n == 1 ? &reals.t[0] : nullptr, dw, Cin, s_ci, s_tap))
    hipLaunchKernelGGL(wgrad_reduce_taps_kernel, dim3((unsigned)((Cin / 8) * ((Cout + 31) / 32))), dim3(256), 0, st,
                       (const float *)ws, dw, Cin, Cout, p.cobs, (int)nslab, accumulate, s_co);
  else
    hipLaunchKernelGGL(wgrad_reduce_kernel<1>, dim3((unsigned)cdiv64(rrows, 8), (unsigned)wc.n), dim3(256), 0, st,
                       (const float *)ws, dw, Cin, Cout, p.cobs, npairs, (int)nslab, accumulate, reals, s_co, s_ci, s_tap);
  DG_CHECK_LAUNCH("wgrad_reduce_kernel");
  return DGTTA_OK;
}

template <typename T>
static int wgrad_launch(const void *x, const View &xv, const void *dy, const View &yv, float *dw, void *ws, size_t ws_bytes,
                        int B, int Cin, int Cout, unsigned tapmask, const Taps &real, long long s_co, long long s_ci,
                        long long s_tap, int accumulate, hipStream_t st, long long xkh = 0, bool split_leg = false) {
  WgradClasses wc;
  wc.n = 1;
  wc.mask[0] = tapmask;
  wc.xoff[0] = wc.yoff[0] = 0;
  RealTaps reals;
  reals.t[0] = real;
  return wgrad_launch_classes<T>(x, xv, dy, yv, dw, ws, ws_bytes, B, Cin, Cout, wc, reals, s_co, s_ci, s_tap, accumulate, st, xkh, split_leg);
}

template <typename T>
static int wgrad_conv(const void *x, int ldx, const void *dy, int lddy, float *dw_t, void *ws, size_t ws_bytes, int B,
                      int Cin, int Cout, int Di, int Hi, int Wi, int stride, int accumulate, hipStream_t st, long long xkh = 0,
                      bool split_leg = false) {
  const long long s_co = (long long)Cin * 27, s_ci = 27, s_tap = 1;
  if (stride == 1) {
    const View xv = dense_view(B, Di, Hi, Wi, ldx), yv = dense_view(B, Di, Hi, Wi, lddy);
    return wgrad_launch<T>(x, xv, dy, yv, dw_t, ws, ws_bytes, B, Cin, Cout, 0x7ffffffu, identity_taps(0), s_co, s_ci, s_tap,
                           accumulate, st, xkh, split_leg);
  }
  if (xkh) return DGTTA_ERR_UNSUPPORTED;
  // stride 2: x[2*vo + tap - 1] lives on parity sub-lattices of x; per axis parity 0 <- tap 1 (offset 0),
  // parity 1 <- tap 0 (offset -1) and tap 2 (offset 0).  Each real tap belongs to exactly one of the 8 classes.
  const int Do = (Di - 1) / 2 + 1, Ho = (Hi - 1) / 2 + 1, Wo = (Wi - 1) / 2 + 1;
  const View yv = dense_view(B, Do, Ho, Wo, lddy);
  if constexpr (sizeof(T) == 2) {
    typedef T T16;
    // one pass over x (full resolution tile) and dy with all 27 taps: conv3_wgrad_tr_s2_kernel
    const int one = dgtta_switches().wgrad_s2_onepass;      // DGTTA_WGRAD_S2_ONEPASS=0 (tests): the 8-class launch
    const View xfull = dense_view(B, Di, Hi, Wi, ldx);
    WgradPlan p = wgrad_plan_s2(B, Cin, Cout, Do, Ho, Wo);
    const size_t need = (size_t)p.units * p.cibs * p.cobs * 27 * 1024 * sizeof(float);
    const bool ok = Cout % 8 == 0 && ldx % 8 == 0 && lddy % 8 == 0 && !((uintptr_t)x & 15) && !((uintptr_t)dy & 15) &&
                    ldx >= (Cin + 7) / 8 * 8 && ws_bytes >= need && p.units < (1ll << 31) && p.cibs * p.cobs <= 65535;
    if (ok && one != '0') {
      static DynLdsOnce once, once_x;
      if (p.cobs % 2 == 0 && Cout % 64 == 0 && one != '1') {      // two output-channel blocks share the x tile (DGTTA_WGRAD_S2_ONEPASS=1: one)
        DG_REQUIRE(ensure_dyn_lds(once_x, reinterpret_cast<const void *>(conv3_wgrad_tr_s2x_kernel<T16>), (int)WT2X::LDS_BYTES) ==
                       hipSuccess, DGTTA_ERR_LAUNCH, "wgrad_tr_s2x: cannot raise the dynamic LDS limit");
        hipLaunchKernelGGL(conv3_wgrad_tr_s2x_kernel<T16>, dim3((unsigned)p.units, (unsigned)(p.cibs * (p.cobs / 2))), dim3(512),
                           WT2X::LDS_BYTES, st, (const bf16_t *)x, xfull, (const bf16_t *)dy, yv, (float *)ws, Cin, Cout, p.tW,
                           p.tH, p.nsd, p.DR, p.cobs);
      } else {
        DG_REQUIRE(ensure_dyn_lds(once, reinterpret_cast<const void *>(conv3_wgrad_tr_s2_kernel<T16>), (int)WT2::LDS_BYTES) ==
                       hipSuccess, DGTTA_ERR_LAUNCH, "wgrad_tr_s2: cannot raise the dynamic LDS limit");
        hipLaunchKernelGGL(conv3_wgrad_tr_s2_kernel<T16>, dim3((unsigned)p.units, (unsigned)(p.cibs * p.cobs)), dim3(256),
                           WT2::LDS_BYTES, st, (const bf16_t *)x, xfull, (const bf16_t *)dy, yv, (float *)ws, Cin, Cout, p.tW,
                           p.tH, p.nsd, p.DR, p.cobs);
      }
      DG_CHECK_LAUNCH("conv3_wgrad_tr_s2_kernel");
      RealTaps ident;
      ident.t[0] = identity_taps(0);
      const int64_t rrows = (int64_t)27 * Cin * ((Cout + 31) / 32);
      const int npairs = p.cibs * p.cobs;
      if (p.units >= 64)
        hipLaunchKernelGGL(wgrad_reduce_kernel<8>, dim3((unsigned)rrows, 1u), dim3(256), 0, st, (const float *)ws, dw_t, Cin,
                           Cout, p.cobs, npairs, (int)p.units, accumulate, ident, s_co, s_ci, s_tap);
      else if (reduce_taps_ok(&ident.t[0], dw_t, Cin, s_ci, s_tap))
        hipLaunchKernelGGL(wgrad_reduce_taps_kernel, dim3((unsigned)((Cin / 8) * ((Cout + 31) / 32))), dim3(256), 0, st,
                           (const float *)ws, dw_t, Cin, Cout, p.cobs, (int)p.units, accumulate, s_co);
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

// ---------------------------------------------------------------------------------------------------------------------
// fp32 weight gradient on the 16-bit matrix-core kernels (round 5).  The reference computes in fp32
// (dg_tta/tta/tta.py:560 never autocasts) and fp32 is what `dgtta run_tta` defaults to; the fp32 MFMA weight-gradient kernel
// (v_mfma_f32_32x32x2_f32, 157 TFLOP/s peak) took 40 % of an fp32 epoch at 0.48 of that peak.  An fp32 number is EXACTLY the
// sum of three bf16 numbers (3 x 8 significand bits, same exponent range):  x = x0 + x1 + x2,  dy = g0 + g1 + g2, and a
// bf16 x bf16 product is exact in the fp32 accumulator of v_mfma_f32_*_bf16.  So
//     dW = sum_v x dy = sum over (i, j) of [sum_v x_i g_j]:  the six products with i + j <= 2 carry everything above 2^-24 of the
// largest term (what fp32 rounding leaves anyway); each is ONE launch of the 16-bit weight-gradient kernels above (ring sweep,
// tr, tr8) on the split planes, accumulated into dW in a fixed order (0,0) (0,1) (1,0) (0,2) (1,1) (2,0).  The split is one
// streaming pass per operand (4 B read, 6 B written per element).  Six launches at 0.40 of the 2.5 PF peak = 167 TFLOP/s
// fp32-equivalent before the split passes, against 75 for the kernel it replaces.
// Workspace: [slab region of the 16-bit plan][x planes 3 x B V ldxs][dy planes 3 x B V ldys] (dgtta_conv3d_wgrad_split_ws_bytes).
namespace {
__device__ __forceinline__ void split3(float v, bf16_t &a0, bf16_t &a1, bf16_t &a2) {
  a0 = f32_to_bf16(v);
  const float r1 = v - bf16_to_f32(a0);          // exact
  a1 = f32_to_bf16(r1);
  const float r2 = r1 - bf16_to_f32(a1);         // exact
  a2 = f32_to_bf16(r2);
}

// rows [n][ld] fp32 (C used channels) -> three planes [n][lds] bf16 (channels >= C zero); one thread = 8 channels of a row
__global__ __launch_bounds__(256) void split3_bf16_kernel(const float *__restrict__ x, int ld, int C, int lds, int64_t n,
                                                        bf16_t *__restrict__ p0, bf16_t *__restrict__ p1,
                                                        bf16_t *__restrict__ p2) {
  const int groups = lds >> 3;
  const int64_t total = n * groups;
  const bool vec = (ld & 3) == 0 && (C & 7) == 0 && (((uintptr_t)x) & 15) == 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / groups;
    const int g = (int)(i - r * groups);
    float v[8];
    if (vec) {
      const float4 lo = *reinterpret_cast<const float4 *>(x + r * ld + 8 * g), hi = *reinterpret_cast<const float4 *>(x + r * ld + 8 * g + 4);
      v[0] = lo.x, v[1] = lo.y, v[2] = lo.z, v[3] = lo.w, v[4] = hi.x, v[5] = hi.y, v[6] = hi.z, v[7] = hi.w;
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = (8 * g + k) < C ? x[r * ld + 8 * g + k] : 0.f;
    }
    bf16_t a[3][8];
#pragma unroll
    for (int k = 0; k < 8; ++k) split3(v[k], a[0][k], a[1][k], a[2][k]);
    bf16_t *dst[3] = {p0, p1, p2};
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      uint4 w;
      w.x = (unsigned)a[q][0] | ((unsigned)a[q][1] << 16);
      w.y = (unsigned)a[q][2] | ((unsigned)a[q][3] << 16);
      w.z = (unsigned)a[q][4] | ((unsigned)a[q][5] << 16);
      w.w = (unsigned)a[q][6] | ((unsigned)a[q][7] << 16);
      *reinterpret_cast<uint4 *>(dst[q] + r * lds + 8 * g) = w;
    }
  }
}
}  // namespace

static size_t wgrad_split_plane_bytes(int B, int C, int D, int H, int W) {
  return align_up((size_t)B * D * H * W * ((C + 7) / 8 * 8) * sizeof(bf16_t), 256);
}
// extra bytes behind the slab region: 3 planes of x (input extent = stride x output extent) and 3 of dy (output extent D, H, W)
size_t conv3_wgrad_split_extra_bytes(int B, int Cin, int Cout, int D, int H, int W, int stride) {
  return 3 * wgrad_split_plane_bytes(B, Cin, D * stride, H * stride, W * stride) + 3 * wgrad_split_plane_bytes(B, Cout, D, H, W);
}

// Di, Hi, Wi: input extent; stride 1 or 2 (even extents)
static int wgrad_conv_f32_split(const float *x, int ldx, const float *dy, int lddy, float *dw_t, void *ws, size_t slab_bytes,
                                void *planes, int B, int Cin, int Cout, int Di, int Hi, int Wi, int stride, int accumulate,
                                hipStream_t st) {
  const int ldxs = (Cin + 7) / 8 * 8, ldys = (Cout + 7) / 8 * 8;
  const int D = Di / stride, H = Hi / stride, W = Wi / stride;
  const int64_t rows_x = (int64_t)B * Di * Hi * Wi, rows = (int64_t)B * D * H * W;
  const size_t xb = wgrad_split_plane_bytes(B, Cin, Di, Hi, Wi), yb = wgrad_split_plane_bytes(B, Cout, D, H, W);
  bf16_t *xs[3], *gs[3];
  for (int i = 0; i < 3; ++i) {
    xs[i] = (bf16_t *)((char *)planes + i * xb);
    gs[i] = (bf16_t *)((char *)planes + 3 * xb + i * yb);
  }
  const int64_t tx = rows_x * (ldxs / 8), ty = rows * (ldys / 8);
  hipLaunchKernelGGL(split3_bf16_kernel, dim3((unsigned)(cdiv64(tx, 256) < 8192 ? cdiv64(tx, 256) : 8192)), dim3(256), 0, st, x, ldx,
                     Cin, ldxs, rows_x, xs[0], xs[1], xs[2]);
  hipLaunchKernelGGL(split3_bf16_kernel, dim3((unsigned)(cdiv64(ty, 256) < 8192 ? cdiv64(ty, 256) : 8192)), dim3(256), 0, st, dy,
                     lddy, Cout, ldys, rows, gs[0], gs[1], gs[2]);
  DG_CHECK_LAUNCH("split3_bf16_kernel");
  static const int PAIRS[6][2] = {{0, 0}, {0, 1}, {1, 0}, {0, 2}, {1, 1}, {2, 0}};
  for (int q = 0; q < 6; ++q) {
    const int rc = wgrad_conv<bf16_t>(xs[PAIRS[q][0]], ldxs, gs[PAIRS[q][1]], ldys, dw_t, ws, slab_bytes, B, Cin, Cout, Di, Hi, Wi,
                                      stride, (accumulate || q > 0) ? 1 : 0, st, 0, true);
    if (rc != DGTTA_OK) return rc;        // (q == 0: nothing written yet, the caller falls back to the fp32 kernel)
  }
  return DGTTA_OK;
}

int conv3_wgrad_mfma(const void *x, int ldx, const void *dy, int lddy, float *dw_t, float *db, void *ws, size_t ws_bytes,
                     int B, int Cin, int Cout, int Di, int Hi, int Wi, int stride, int accumulate, int dtype,
                     hipStream_t st, long long xkh) {
  (void)db;
  if (stride != 1 && stride != 2) return DGTTA_ERR_UNSUPPORTED;
  if (stride == 2 && ((Di | Hi | Wi) & 1)) return DGTTA_ERR_UNSUPPORTED;   // odd extents: leave to the general kernel
  if (dtype == DGTTA_F32) {
    if (xkh) return DGTTA_ERR_UNSUPPORTED;      // (x as 32-channel planes: 16-bit storage only)
    // the caller offered the split workspace (dgtta_conv3d_wgrad_split_ws_bytes) behind the plain one: six 16-bit launches
    // (DGTTA_WGRAD_F32_SPLIT=0: the fp32 MFMA kernel, its predecessor)
    const int Do = Di / stride, Ho = Hi / stride, Wo = Wi / stride;
    const size_t base = conv3_wgrad_mfma_ws_bytes(B, Cin, Cout, Do, Ho, Wo);
    const size_t slab = align_up(base, 256);
    if (Cout % 8 == 0 && dgtta_switches().wgrad_f32_split != '0' &&
        ws_bytes >= slab + conv3_wgrad_split_extra_bytes(B, Cin, Cout, Do, Ho, Wo, stride)) {
      const int rc = wgrad_conv_f32_split((const float *)x, ldx, (const float *)dy, lddy, dw_t, ws, slab, (char *)ws + slab, B, Cin,
                                          Cout, Di, Hi, Wi, stride, accumulate, st);
      if (rc != DGTTA_ERR_UNSUPPORTED) return rc;
    }
    return wgrad_conv<float>(x, ldx, dy, lddy, dw_t, ws, ws_bytes, B, Cin, Cout, Di, Hi, Wi, stride, accumulate, st);
  }
  if (dtype == DGTTA_BF16) return wgrad_conv<bf16_t>(x, ldx, dy, lddy, dw_t, ws, ws_bytes, B, Cin, Cout, Di, Hi, Wi, stride, accumulate, st, xkh);
  if (dtype == DGTTA_F16) return wgrad_conv<f16_t>(x, ldx, dy, lddy, dw_t, ws, ws_bytes, B, Cin, Cout, Di, Hi, Wi, stride, accumulate, st, xkh);
  return DGTTA_ERR_UNSUPPORTED;
}

// ConvTranspose3d k2 s2 weight gradient: dw_t[ci][co][o] (+)= sum_v x[v][ci] * dout[2v+o][co]  (8 single-tap launches)
template <typename T>
static int convT_wgrad(const void *x, int ldx, const void *dout, int lddo, float *dw_t, void *ws, size_t ws_bytes, int B,
                       int Cin, int Cout, int Di, int Hi, int Wi, int accumulate, hipStream_t st, float *bias_part,
                       size_t bias_part_bytes, int *bias_units) {
  if (bias_units) *bias_units = 0;
  const View xv = dense_view(B, Di, Hi, Wi, ldx);
  if constexpr (sizeof(T) == 2) {
    typedef T T16;
    const int one = dgtta_switches().convt_wgrad_onepass;      // DGTTA_CONVT_WGRAD_ONEPASS=0 (tests): the 8-class launch
    const View yfull = dense_view(B, 2 * Di, 2 * Hi, 2 * Wi, lddo);
    WgradPlan p = wgrad_plan_s2(B, Cin, Cout, Di, Hi, Wi);        // same tile shape (2 rows x 16 voxels) on the input lattice
    const size_t need = (size_t)p.units * p.cibs * p.cobs * 27 * 1024 * sizeof(float);
    const bool ok = Cout % 8 == 0 && ldx % 8 == 0 && lddo % 8 == 0 && !((uintptr_t)x & 15) && !((uintptr_t)dout & 15) &&
                    ldx >= (Cin + 7) / 8 * 8 && ws_bytes >= need && p.units < (1ll << 31) && p.cibs * p.cobs <= 65535;
    if (ok && one != '0') {
      static DynLdsOnce once1, once2;
      // bias partials [unit][32 cobs] ride along when the caller offers room for them
      float *bp = (bias_part && bias_units && bias_part_bytes >= (size_t)p.units * p.cobs * 32 * sizeof(float)) ? bias_part : nullptr;
      if (bp) *bias_units = (int)p.units;
      if (p.cibs >= 2) {      // two input-channel blocks share a dout tile
        DG_REQUIRE(ensure_dyn_lds(once2, reinterpret_cast<const void *>(convT_wgrad_tr_kernel<T16, 2>), (int)WT3<2>::LDS_BYTES) ==
                       hipSuccess, DGTTA_ERR_LAUNCH, "convT_wgrad_tr: cannot raise the dynamic LDS limit");
        hipLaunchKernelGGL((convT_wgrad_tr_kernel<T16, 2>), dim3((unsigned)p.units, (unsigned)(cdiv(p.cibs, 2) * p.cobs)), dim3(256),
                           WT3<2>::LDS_BYTES, st, (const bf16_t *)x, xv, (const bf16_t *)dout, yfull, (float *)ws, Cin, Cout, p.tW,
                           p.tH, p.nsd, p.DR, p.cobs, p.cibs, bp);
      } else {
        DG_REQUIRE(ensure_dyn_lds(once1, reinterpret_cast<const void *>(convT_wgrad_tr_kernel<T16, 1>), (int)WT3<1>::LDS_BYTES) ==
                       hipSuccess, DGTTA_ERR_LAUNCH, "convT_wgrad_tr: cannot raise the dynamic LDS limit");
        hipLaunchKernelGGL((convT_wgrad_tr_kernel<T16, 1>), dim3((unsigned)p.units, (unsigned)(p.cibs * p.cobs)), dim3(256),
                           WT3<1>::LDS_BYTES, st, (const bf16_t *)x, xv, (const bf16_t *)dout, yfull, (float *)ws, Cin, Cout, p.tW,
                           p.tH, p.nsd, p.DR, p.cobs, p.cibs, bp);
      }
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

// fp32 transposed-conv weight gradient as six launches of the 16-bit kernel on exact three-term bf16 splits (see
// wgrad_conv_f32_split): extra bytes behind the slab region = 3 planes of x (input lattice) and 3 of dout (output lattice)
size_t convT_wgrad_split_extra_bytes(int B, int Cin, int Cout, int Di, int Hi, int Wi) {
  return 3 * wgrad_split_plane_bytes(B, Cin, Di, Hi, Wi) + 3 * wgrad_split_plane_bytes(B, Cout, 2 * Di, 2 * Hi, 2 * Wi);
}

static int convT_wgrad_f32_split(const float *x, int ldx, const float *dout, int lddo, float *dw_t, void *ws, size_t slab_bytes,
                                 void *planes, int B, int Cin, int Cout, int Di, int Hi, int Wi, int accumulate, hipStream_t st) {
  const int ldxs = (Cin + 7) / 8 * 8, ldys = (Cout + 7) / 8 * 8;
  const int64_t rows_x = (int64_t)B * Di * Hi * Wi, rows = rows_x * 8;
  const size_t xb = wgrad_split_plane_bytes(B, Cin, Di, Hi, Wi), yb = wgrad_split_plane_bytes(B, Cout, 2 * Di, 2 * Hi, 2 * Wi);
  bf16_t *xs[3], *gs[3];
  for (int i = 0; i < 3; ++i) {
    xs[i] = (bf16_t *)((char *)planes + i * xb);
    gs[i] = (bf16_t *)((char *)planes + 3 * xb + i * yb);
  }
  const int64_t tx = rows_x * (ldxs / 8), ty = rows * (ldys / 8);
  hipLaunchKernelGGL(split3_bf16_kernel, dim3((unsigned)(cdiv64(tx, 256) < 8192 ? cdiv64(tx, 256) : 8192)), dim3(256), 0, st, x, ldx,
                     Cin, ldxs, rows_x, xs[0], xs[1], xs[2]);
  hipLaunchKernelGGL(split3_bf16_kernel, dim3((unsigned)(cdiv64(ty, 256) < 8192 ? cdiv64(ty, 256) : 8192)), dim3(256), 0, st, dout,
                     lddo, Cout, ldys, rows, gs[0], gs[1], gs[2]);
  DG_CHECK_LAUNCH("split3_bf16_kernel");
  static const int PAIRS[6][2] = {{0, 0}, {0, 1}, {1, 0}, {0, 2}, {1, 1}, {2, 0}};
  for (int q = 0; q < 6; ++q) {
    const int rc = convT_wgrad<bf16_t>(xs[PAIRS[q][0]], ldxs, gs[PAIRS[q][1]], ldys, dw_t, ws, slab_bytes, B, Cin, Cout, Di, Hi, Wi,
                                       (accumulate || q > 0) ? 1 : 0, st, nullptr, 0, nullptr);
    if (rc != DGTTA_OK) return rc;        // (q == 0: nothing written yet, the caller falls back to the fp32 kernel)
  }
  return DGTTA_OK;
}

// bias_part / bias_units (optional): room for [units][ceil(Cout / 32) * 32] floats; *bias_units > 0 on return means the launch left
// the per-unit sums of dout there (convT_bias_finalize adds them up), 0 means the caller runs its own pass over dout
int convT_wgrad_mfma(const void *x, int ldx, const void *dout, int lddo, float *dw_t, void *ws, size_t ws_bytes, int B,
                     int Cin, int Cout, int Di, int Hi, int Wi, int accumulate, int dtype, hipStream_t st, float *bias_part,
                     size_t bias_part_bytes, int *bias_units) {
  if (bias_units) *bias_units = 0;
  if (dtype == DGTTA_F32) {
    // the caller offered the split workspace (dgtta_convT3d_bwd_split_ws_bytes): six 16-bit launches (DGTTA_WGRAD_F32_SPLIT=0: never)
    const size_t slab = align_up(conv3_wgrad_mfma_ws_bytes(B, Cin, Cout, Di, Hi, Wi), 256);
    if (Cout % 8 == 0 && dgtta_switches().wgrad_f32_split != '0' &&
        ws_bytes >= slab + convT_wgrad_split_extra_bytes(B, Cin, Cout, Di, Hi, Wi)) {
      const int rc = convT_wgrad_f32_split((const float *)x, ldx, (const float *)dout, lddo, dw_t, ws, slab, (char *)ws + slab, B, Cin,
                                           Cout, Di, Hi, Wi, accumulate, st);
      if (rc != DGTTA_ERR_UNSUPPORTED) return rc;
    }
    return convT_wgrad<float>(x, ldx, dout, lddo, dw_t, ws, ws_bytes, B, Cin, Cout, Di, Hi, Wi, accumulate, st, nullptr, 0, nullptr);
  }
  if (dtype == DGTTA_BF16)
    return convT_wgrad<bf16_t>(x, ldx, dout, lddo, dw_t, ws, ws_bytes, B, Cin, Cout, Di, Hi, Wi, accumulate, st, bias_part,
                               bias_part_bytes, bias_units);
  if (dtype == DGTTA_F16)
    return convT_wgrad<f16_t>(x, ldx, dout, lddo, dw_t, ws, ws_bytes, B, Cin, Cout, Di, Hi, Wi, accumulate, st, bias_part,
                              bias_part_bytes, bias_units);
  return DGTTA_ERR_UNSUPPORTED;
}

int convT_bias_finalize(const float *part, int units, int Cout, float *db, int accumulate, hipStream_t st) {
  hipLaunchKernelGGL(convT_bias_finalize_kernel, dim3((unsigned)cdiv(Cout, 32)), dim3(256), 0, st, part, units, cdiv(Cout, 32) * 32,
                     Cout, db, accumulate);
  return hipGetLastError() == hipSuccess ? DGTTA_OK : DGTTA_ERR_LAUNCH;
}

// 1x1x1 head weight gradient dw[k][ci] = sum_rows dout[row][k] * x[row][ci] as a single-tap run of the wgrad kernel:
// the [rows] axis is folded into a D x 4 x 32 lattice (no neighbour access with one tap, so any folding is valid).
namespace {
template <typename T16>
__global__ void f32_to_16_rows_kernel(const float *__restrict__ src, int lds_, unsigned short *__restrict__ dst, int C,
                                      int64_t rows) {
  const int64_t n = rows * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dst[i] = f32_to_16<T16>(src[(i / C) * lds_ + i % C]);
}
}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient of a POINTWISE layer with 32 input channels (the 1x1x1 segmentation head: dW[k][c] = sum_v d[v][k] z[v][c])
// as a stream (round 4).  head_wgrad_mfma used to view the rows as a D x 4 x 32 volume and run the 27-tap class kernel with the
// centre tap only: one of its four waves multiplies, every 128-voxel step ends in a DMA drain and a barrier with 12 KB in
// flight - 642 us for 1.6 GB at 8 x 128^3 (2.5 TB/s).  Here a persistent workgroup takes chunks of 128 rows, keeps two chunks in
// flight behind the one it multiplies (LDS-DMA into a 4-deep ring, counted vmcnt), every wave multiplies its own 32 rows of the
// chunk (two v_mfma_f32_32x32x16 with transposed LDS reads, as in conv3_wgrad_tr_kernel), the four waves' accumulators are
// added in wave order at the end and the workgroups' 32 x 32 partials in workgroup order by the finalize kernel: deterministic.
// d has nsel <= 32 columns (a multiple of 8); its rows are zero-extended to 32 columns on the way into LDS.
struct PWG {
  static constexpr int CHUNK = 128, NBUF = 4, LA = 2;
  static constexpr int X_B = CHUNK * 64, BUF_B = 2 * X_B;
  static constexpr int LDS_BYTES = NBUF * BUF_B;
};

template <typename T16>
__global__ __launch_bounds__(256, 2) void pointwise_wgrad_kernel(const bf16_t *__restrict__ x, int ldx,
                                                                 const unsigned short *__restrict__ d16, int nsel,
                                                                 float *__restrict__ partial, int64_t nchunks) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l_vox = lane >> 2, l_chunk = lane & 3;
  const int64_t G = gridDim.x;
  const int64_t n = (nchunks - (int64_t)blockIdx.x + G - 1) / G;      // this workgroup's chunks: blockIdx.x, + G, ...
  // every wave issues 4 pieces of 1 KiB per chunk: x pieces wave, wave + 4 (16 rows x 64 B each) and the same two of d
  auto issue = [&](int64_t i) __attribute__((always_inline)) {
    const int64_t row0 = (blockIdx.x + i * G) * PWG::CHUNK;
    unsigned char *buf = smem + (int)(i % PWG::NBUF) * PWG::BUF_B;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int pc = wave + 4 * j;
      const int64_t row = row0 + pc * 16 + l_vox;
      dma16_to_lds(x + row * ldx + l_chunk * 8, lds_addr_of(buf + pc * 1024));
      const void *src = l_chunk * 8 < nsel ? (const void *)(d16 + row * nsel + l_chunk * 8) : (const void *)&g_zero16;
      dma16_to_lds(src, lds_addr_of(buf + PWG::X_B + pc * 1024));
    }
  };
  const int lane_off = ((lane >> 5) * 8 + ((lane & 15) >> 2)) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8;
  f32x16_t acc;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.f;
  if (n > 0) issue(0);
  if (n > 1) issue(1);
  for (int64_t i = 0; i < n; ++i) {
    if (i + 2 < n) {
      issue(i + 2);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");        // all but the two newest chunks of this wave have landed
    } else if (i + 1 < n) {
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      dma_wait_all();
    }
    lds_barrier();                                            // ... and everybody else's pieces of chunk i
    const unsigned char *buf = smem + (int)(i % PWG::NBUF) * PWG::BUF_B + wave * 32 * 64 + lane_off;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const bf16x8_t a = tr_operand(buf + ks * 1024);
      const bf16x8_t b = tr_operand(buf + PWG::X_B + ks * 1024);
      acc = mfma32_tr<T16>(a, b, acc);
    }
  }
  // waves in order through LDS (the ring is free after a barrier), then one 32 x 32 partial per workgroup: [c][k]
  lds_barrier();
  float *red = reinterpret_cast<float *>(smem);
  const int co = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int q = 0; q < 16; ++q) red[wave * 1024 + ((q & 3) + 8 * (q >> 2) + 4 * hh) * 32 + co] = acc[q];
  __syncthreads();
  for (int e = tid; e < 1024; e += 256)
    partial[(int64_t)blockIdx.x * 1024 + e] = ((red[e] + red[1024 + e]) + red[2048 + e]) + red[3072 + e];
}

__global__ __launch_bounds__(1024) void pointwise_wgrad_finalize_kernel(const float *__restrict__ partial, int G, float *__restrict__ dw,
                                                                        int nsel, int Cin, int accumulate) {
  const int c = threadIdx.x >> 5, k = threadIdx.x & 31;      // partial layout [c][k]
  float s = 0.f;
  for (int g = 0; g < G; g += 8) {      // eight loads in flight, added in workgroup order
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = g + j < G ? partial[(int64_t)(g + j) * 1024 + threadIdx.x] : 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (g + j < G) s += v[j];
  }
  if (k < nsel && c < Cin) {
    float *o = dw + (int64_t)k * Cin + c;
    *o = accumulate ? *o + s : s;
  }
}

size_t head_wgrad_mfma_ws_bytes(int Cin, int nsel, int64_t rows) {
  if (rows <= 0 || rows % 128 || rows / 128 >= (1ll << 30)) return 0;      // the MFMA plan does not apply (as head_wgrad_mfma)
  const int D = (int)(rows / 128);
  return conv3_wgrad_mfma_ws_bytes(1, Cin, nsel, D, 4, 32) + align_up((size_t)rows * nsel * 2, 256);
}

// have_d16: the 16-bit copy of dout already sits at the start of ws (written by the head's data-gradient kernel)
int head_wgrad_mfma(const void *x, int ldx, const float *dout, int lddo, float *dw_sel, void *ws, size_t ws_bytes, int Cin,
                    int nsel, int64_t rows, int accumulate, int dtype, hipStream_t st, bool have_d16) {
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
  if (dtype == DGTTA_BF16 || dtype == DGTTA_F16) {
    const size_t cbytes = align_up((size_t)rows * nsel * 2, 256);
    unsigned short *d16 = (unsigned short *)ws;
    if (!have_d16) {
      if (dtype == DGTTA_BF16)
        hipLaunchKernelGGL(f32_to_16_rows_kernel<bf16_t>, dim3(2048), dim3(256), 0, st, dout, lddo, d16, nsel, rows);
      else
        hipLaunchKernelGGL(f32_to_16_rows_kernel<f16_t>, dim3(2048), dim3(256), 0, st, dout, lddo, d16, nsel, rows);
      DG_CHECK_LAUNCH("f32_to_16_rows_kernel");
    }
    // the streaming kernel (DGTTA_WGRAD_TR=0, tests: the class kernel below)
    const int G = 512;
    if (Cin == 32 && ldx >= 32 && ldx % 8 == 0 && nsel % 8 == 0 && nsel <= 32 && !((uintptr_t)x & 15) &&
        ws_bytes - cbytes >= (size_t)G * 1024 * sizeof(float) && dgtta_switches().wgrad_tr != '0') {
      float *partial = reinterpret_cast<float *>((char *)ws + cbytes);
      const int64_t nchunks = rows / PWG::CHUNK;
      const int g = (int)(nchunks < G ? nchunks : G);
      static DynLdsOnce once_b, once_h;
      if (dtype == DGTTA_BF16) {
        DG_REQUIRE(ensure_dyn_lds(once_b, reinterpret_cast<const void *>(pointwise_wgrad_kernel<bf16_t>), PWG::LDS_BYTES) == hipSuccess,
                   DGTTA_ERR_LAUNCH, "pointwise_wgrad: cannot raise the dynamic LDS limit");
        hipLaunchKernelGGL(pointwise_wgrad_kernel<bf16_t>, dim3((unsigned)g), dim3(256), PWG::LDS_BYTES, st, (const bf16_t *)x, ldx, d16,
                           nsel, partial, nchunks);
      } else {
        DG_REQUIRE(ensure_dyn_lds(once_h, reinterpret_cast<const void *>(pointwise_wgrad_kernel<f16_t>), PWG::LDS_BYTES) == hipSuccess,
                   DGTTA_ERR_LAUNCH, "pointwise_wgrad: cannot raise the dynamic LDS limit");
        hipLaunchKernelGGL(pointwise_wgrad_kernel<f16_t>, dim3((unsigned)g), dim3(256), PWG::LDS_BYTES, st, (const bf16_t *)x, ldx, d16,
                           nsel, partial, nchunks);
      }
      DG_CHECK_LAUNCH("pointwise_wgrad_kernel");
      hipLaunchKernelGGL(pointwise_wgrad_finalize_kernel, dim3(1), dim3(1024), 0, st, (const float *)partial, g, dw_sel, nsel, Cin,
                         accumulate);
      DG_CHECK_LAUNCH("pointwise_wgrad_finalize_kernel");
      return DGTTA_OK;
    }
    const View yv = dense_view(1, D, 4, 32, nsel);
    if (dtype == DGTTA_BF16)
      return wgrad_launch<bf16_t>(x, xv, d16, yv, dw_sel, (char *)ws + cbytes, ws_bytes - cbytes, 1, Cin, nsel, 1u << 13, real,
                                  Cin, 1, 0, accumulate, st);
    return wgrad_launch<f16_t>(x, xv, d16, yv, dw_sel, (char *)ws + cbytes, ws_bytes - cbytes, 1, Cin, nsel, 1u << 13, real,
                               Cin, 1, 0, accumulate, st);
  }
  return DGTTA_ERR_UNSUPPORTED;
}



// would the weight gradient of a stride-1 conv on these dims take x as 32-channel planes (only the ring sweep does)?
bool conv3_wgrad_blocked_ok(int B, int Cin, int Cout, int D, int H, int W, int dtype) {
  if ((dtype != DGTTA_BF16 && dtype != DGTTA_F16) || Cin % 32 || Cout % 8) return false;
  if (dgtta_switches().wgrad_tr == '0' || dgtta_switches().wgrad_ring == '0') return false;
  const View xv = dense_view(B, D, H, W, 32), yv = dense_view(B, D, H, W, Cout);
  int rc = DGTTA_OK;
  const size_t ws_bytes = conv3_wgrad_mfma_ws_bytes(B, Cin, Cout, D, H, W);
  const int g = conv3_wgrad_ring_launch((const void *)16, xv, (const void *)16, yv, nullptr, ws_bytes, B, Cin, Cout, dtype == DGTTA_F16, nullptr,
                                        &rc, (long long)B * D * H * W * 32, true);
  return rc == DGTTA_OK && g > 0;
}
