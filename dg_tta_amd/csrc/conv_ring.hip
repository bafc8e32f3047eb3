// D-ring 3x3x3 convolution kernel (16-bit storage, stride 1, 32 input channels) -- the large 32-channel layers of the
// 128^3 level (PlainConvUNet stages enc.0 / dec.3, reference topology dg_tta/__resources__/dummy_results/*/plans.json:279-401;
// forward of the 32 -> 32 blocks and the data gradients whose dy has 32 channels).
//
// Why another kernel (round 4).  The row-reuse kernel (conv_rows.hip) re-stages, for every 4 x 8 x 32 tile and 16-channel
// chunk, a 6 x 10 x 34 halo (1.99x the tile) AND the chunk's 27 weight fragments, behind two workgroup barriers per chunk:
// its MFMA loop is 59 % of the wave time and the matrix pipe is busy 46 % of a phase (profiles/r04_mfma_util.json).
// Here nothing is staged twice along D and the weights are never staged at all:
//   * a persistent workgroup owns a COLUMN of the volume (8 rows x 32 voxels, all of D or a segment of it) and marches
//     along D two planes at a time.  The LDS holds a ring of 6 input planes (10 x 34 halo voxels x 32 channels = 21.8 KB
//     each): 4 in use, 2 arriving by LDS-DMA for the next step.  Halo factor 1.33 instead of 1.99, 5-6 DMA pieces per wave
//     and step instead of 40 for the same FLOPs, ONE barrier per step (6.9 k MFMA cycles) instead of four.
//   * v_mfma_f32_16x16x32 with the WEIGHTS as the M operand: a wave owns 16 output channels, and the 27 taps x 32 input
//     channels x 16 output channels it needs are 27 fragments = 108 registers, loaded once per workgroup.  K = 32 is the
//     whole channel extent, so there are no K chunks, no weight buffer in LDS, no reload.  (The guide's clock effect of
//     the 16x16x32 shape comes on top.)
//   * 8 waves = 4 row pairs x 2 output-channel halves; a wave computes 2 planes x 2 rows x 32 voxels x 16 channels per step
//     (8 accumulator tiles = 32 registers) from 4 x 4 input rows: 96 fragment reads feed 216 MFMAs.
//   * the accumulator of D = W^T X has the output CHANNELS along the registers and the voxel on the lane: four packed
//     converts, two v_permlane16_swap and one 16-byte store per output row and wave - no LDS transpose, no epilogue barrier;
//     an output row is converted and stored as soon as its last input row is done, between the MFMAs of the rows that remain.
//   * one barrier per step, in its middle; the DMA of a plane pair has more than a step to land (see "Synchronisation" below).
// Measured (MI355X, fp16, 8 x 128^3 x 32 -> 32, 400 back-to-back launches): 0.75-0.79 ms = 1170-1240 TFLOP/s against 0.975 ms
// of the row-reuse kernel; in-kernel clock 1.65-1.8 GHz: the launch is POWER bound - fewer cycles come back as a lower
// clock (stamps: the streaming epilogue took 1600 -> 480 cycles off a step and the clock fell from 1.69 to 1.80 ... the wall
// time stayed) - so what helps now is less energy per FLOP, not a tighter schedule (profiles/r04_ab.txt).
// LDS image of a plane: voxel-major rows of 34 voxels x 64 B; inside a voxel the four 16-byte channel groups sit at
// position g ^ 2*((u >> 2) & 1) (u = voxel index in the row): with that, the 16 lanes that ds_read_b128 serves per cycle
// (4 voxels apart in two channel groups) fall on 16 different 16-byte bank slots for every tap shift.  The permutation
// is applied on the SOURCE address of the DMA (the LDS side of an LDS-DMA is lane-linear).
// Zero padding and ragged edges: the DMA and the stores are raw BUFFER operations; a lane outside the volume gets an
// offset beyond the descriptor's range, which reads as zero and drops the store.
#include "conv_common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

// KH = number of 32-channel K halves: 1 = 32 input channels (8 waves: 4 row pairs x 2 output-channel halves, 8-row tiles),
// 2 = 64 input channels (round 4b).  With 64 input channels a wave's weights are 27 taps x 2 K halves = 216 registers, which
// only fits ONE wave per SIMD: 4 waves of up to 512 registers (2 row pairs x 2 output-channel halves, 4-row tiles); a plane of
// the ring is then two 32-channel sub-planes [K half][row][voxel][64 B], each with the bank layout described above.
template <int KH>
struct RingCfgT {
  static constexpr int TH = 8 / KH, TW = 32;                         // (two output planes per step)
  static constexpr int IH = TH + 2, IW = TW + 2;
  static constexpr int VB = 64;                                      // bytes per voxel of a sub-plane: 32 channels x 2
  static constexpr int ROWB = IW * VB;                               // 2176
  static constexpr int SUB_PIECES = (IH * ROWB + 1023) / 1024;       // 1-KiB DMA pieces per sub-plane (the last one partly pad): 22 / 13
  static constexpr int SUB = SUB_PIECES * 1024;
  static constexpr int PIECES = KH * SUB_PIECES, PLANE = PIECES * 1024;
  static constexpr int NVOX = IH * IW;                               // halo voxels per plane
  static constexpr int NSLOT = 6;
  static constexpr int NW = 8 / KH, NT = NW * 64;
  static constexpr int NPW = (2 * PIECES + NW - 1) / NW;             // pieces per wave and step (6 / 13)
  static constexpr int RED_BYTES = NW * 16 * 2 * (int)sizeof(float) + 32 * 2 * (int)sizeof(float);      // + GST constants
  static constexpr int LDS_BYTES = NSLOT * PLANE + RED_BYTES;
};
typedef RingCfgT<1> RingCfg;
static_assert(RingCfgT<2>::LDS_BYTES <= 160 * 1024 && RingCfgT<1>::LDS_BYTES <= 160 * 1024, "ring does not fit the LDS");

// scheduling pattern of one input row: NM MFMAs with NR LDS reads dealt out behind the first ones, an MFMA first
template <int NM, int NR>
__device__ __forceinline__ void ring_sched_interleave() {
  if constexpr (NM > 0) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    constexpr int r = NR > 0 ? 1 : 0;
    if constexpr (r > 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    ring_sched_interleave<NM - 1, NR - r>();
  }
}

template <typename T16>
__device__ __forceinline__ void mfma16(const uint4 &a, const uint4 &b, f32x4_t &acc);
template <>
__device__ __forceinline__ void mfma16<bf16_t>(const uint4 &a, const uint4 &b, f32x4_t &acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mfma16<f16_t>(const uint4 &a, const uint4 &b, f32x4_t &acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), acc, 0, 0, 0);
}

// two fp32 -> one packed 16-bit pair by ONE instruction (v_cvt_pk_f16_f32 / v_cvt_pk_bf16_f32: what a vector fptrunc selects on
// gfx950, round to nearest even like the scalar conversions of common.h, which cost three instructions per pair)
template <typename T16>
__device__ __forceinline__ unsigned pack2_pk(float lo, float hi);
template <>
__device__ __forceinline__ unsigned pack2_pk<f16_t>(float lo, float hi) {
  typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{lo, hi}, h2_t));
}
template <>
__device__ __forceinline__ unsigned pack2_pk<bf16_t>(float lo, float hi) {
  typedef __bf16 b2_t __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{lo, hi}, b2_t));
}

// LDS-DMA through a buffer descriptor: lane i's 16 bytes land at lds_addr + 16 i; a lane whose offset is outside the
// descriptor's range delivers zeros.  Inline asm for the reason given at dma16_to_lds (no compiler-side vmcnt bookkeeping).
__device__ __forceinline__ void dma16_buf_to_lds(u32x4_t rsrc, unsigned voff, unsigned soff, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_addr)
               : "memory");
}
template <bool NT_ST>
__device__ __forceinline__ void store16_buf(u32x4_t rsrc, unsigned voff, unsigned soff, u32x4_t val) {
  if (NT_ST) asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen nt" ::"v"(val), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
  else asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen" ::"v"(val), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
__device__ __forceinline__ void vm_wait_all_but(int n) {      // the n youngest vector-memory operations may stay in flight
  if (n > 40) n = 40;       // (fewer in flight than allowed: always safe)
  switch (n) {
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
    case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
    case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
    case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
    case 17: asm volatile("s_waitcnt vmcnt(17)" ::: "memory"); break;
    case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
    case 19: asm volatile("s_waitcnt vmcnt(19)" ::: "memory"); break;
    case 20: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
    case 21: asm volatile("s_waitcnt vmcnt(21)" ::: "memory"); break;
    case 22: asm volatile("s_waitcnt vmcnt(22)" ::: "memory"); break;
    case 23: asm volatile("s_waitcnt vmcnt(23)" ::: "memory"); break;
    case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
    case 25: asm volatile("s_waitcnt vmcnt(25)" ::: "memory"); break;
    case 26: asm volatile("s_waitcnt vmcnt(26)" ::: "memory"); break;
    case 27: asm volatile("s_waitcnt vmcnt(27)" ::: "memory"); break;
    case 28: asm volatile("s_waitcnt vmcnt(28)" ::: "memory"); break;
    case 29: asm volatile("s_waitcnt vmcnt(29)" ::: "memory"); break;
    case 30: asm volatile("s_waitcnt vmcnt(30)" ::: "memory"); break;
    case 31: asm volatile("s_waitcnt vmcnt(31)" ::: "memory"); break;
    case 32: asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); break;
    case 33: asm volatile("s_waitcnt vmcnt(33)" ::: "memory"); break;
    case 34: asm volatile("s_waitcnt vmcnt(34)" ::: "memory"); break;
    case 35: asm volatile("s_waitcnt vmcnt(35)" ::: "memory"); break;
    case 36: asm volatile("s_waitcnt vmcnt(36)" ::: "memory"); break;
    case 37: asm volatile("s_waitcnt vmcnt(37)" ::: "memory"); break;
    case 38: asm volatile("s_waitcnt vmcnt(38)" ::: "memory"); break;
    case 39: asm volatile("s_waitcnt vmcnt(39)" ::: "memory"); break;
    case 40: asm volatile("s_waitcnt vmcnt(40)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}
__device__ __forceinline__ u32x4_t make_rsrc(const void *base, unsigned bytes) {
  const unsigned long long a = (unsigned long long)base;
  u32x4_t r;
  r[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
  r[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);      // stride 0
  r[2] = __builtin_amdgcn_readfirstlane(bytes);
  r[3] = 0x00020000u;                                                         // raw 32-bit data format (gfx9 family)
  return r;
}
constexpr unsigned OOB = 0x80000000u;      // beyond every descriptor this kernel builds (the launcher checks < 2^31 bytes)

// GST: the launch is the DATA GRADIENT of a conv whose input was z = LeakyReLU(InstanceNorm(y_prev)); the epilogue also leaves the
// sums the InstanceNorm backward of that previous block needs (conv_rows.hip, GST: same definition, same partial-sum layout,
// same finalize kernel): sum g' and sum g' y_prev with g' = gz * lrelu'(A y_prev + B), from the ROUNDED gz it stores.
struct RingGst {
  const bf16_t *y;          // y_prev, same lattice as the output
  View v;
  const float *mr, *gamma, *beta;
  float slope;
  unsigned bytes;
};

// ABL (diagnostic builds, DGTTA_RING_ABL; results are wrong for 1 and 2): 1 no DMA after a job's first four planes, 2 no stores,
// 3 a step's DMA pieces issued in one burst behind the barrier instead of one per input row, 6 per-segment cycle stamps and
// the in-kernel clock, written behind the statistics (profiles/tools/ring_stamps.py)
template <typename T16, bool NT_ST, int ABL = 0, bool GST = false, int KH = 1>
__global__ __launch_bounds__(RingCfgT<KH>::NT) void conv3_ring_kernel(const bf16_t *__restrict__ x, View xv, const bf16_t *__restrict__ w,
                                                                 Taps taps, const float *__restrict__ bias, bf16_t *__restrict__ y,
                                                                 View yv, int Cout, int tilesW, int tilesH, int nblkN, int nseg,
                                                                 int steps_per_seg, int njobs, double *__restrict__ stats,
                                                                 int ntaps_src, unsigned x_bytes, unsigned y_bytes, RingGst gst,
                                                                 long long xkh) {
  // xkh (round 6, KH = 2): element distance between the two 32-channel K halves of x.  0: they interleave in rows of 64 channels
  // (half h at channel 32 h of a voxel's row); > 0: two DENSE 32-channel tensors xkh elements apart - the level-0 concat buffer as
  // planes [up | skip], so that the kernels that read ONE half (the stride-2 conv of the skip, the transposed conv's backward)
  // use whole 128-byte lines instead of 64 bytes of each (profiles/r05_ab.txt: 4.3x input fetched by the stride-2 forward).
  typedef RingCfgT<KH> C;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float *red = reinterpret_cast<float *>(smem + C::NSLOT * C::PLANE);
  float *gcst = red + C::NW * 16 * 2;      // GST: (A, B) of the job's 32 channels
  const unsigned lds0 = lds_addr_of(smem);

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int v = lane & 15, q = lane >> 4;
  const int chalf = wave & 1, ws = wave >> 1;      // output-channel half, row pair
  const int D = xv.D, H = xv.H, W = xv.W;
  const int steps_total = (D + 1) / 2;

  // this lane's fragment read offsets for the three tap shifts along W (voxel v + kw of rows 2 ws .. 2 ws + 3)
  int aoff[3];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) {
    const int u = v + kw;
    aoff[kw] = (2 * ws) * C::ROWB + u * C::VB + ((q ^ (((u >> 2) & 1) << 1)) << 4);
  }

  uint4 wreg[KH][27];
  int nb_loaded = -1;
  // ABL 6 (diagnostic): cycle stamps per segment - 0 job prologue, 1 rows 0..6, 2 wait + barrier, 3 rows 7..15, 4 epilogue
  unsigned long long tseg[5] = {0, 0, 0, 0, 0}, tprev = 0, t_begin = 0, rt_begin = 0;
  auto stamp = [&](int i) {
    if (ABL == 6) {
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      __builtin_amdgcn_sched_barrier(0);
      tseg[i] += t - tprev;
      tprev = t;
    }
  };
  if (ABL == 6) {
    t_begin = __builtin_amdgcn_s_memtime();
    rt_begin = __builtin_amdgcn_s_memrealtime();
  }
  const int G = gridDim.x;
  const bool xcd_order = (G % 8) == 0;
  const int rounds = (njobs + G - 1) / G;
  for (int rd = 0; rd < rounds; ++rd) {
    // job order: the 32 workgroups of an XCD (blockIdx % 8) take 32 consecutive jobs = columns that are neighbours along H
    // (then W) and march in step, so the halo rows they share are fetched from HBM once
    int j = xcd_order ? (rd * 8 + (int)(blockIdx.x % 8)) * (G / 8) + (int)(blockIdx.x / 8) : rd * G + (int)blockIdx.x;
    if (j >= njobs) continue;       // (uniform per workgroup; rounds are independent: no barrier is skipped by others)
    const int nb = j % nblkN;
    j /= nblkN;
    const int th = j % tilesH;
    j /= tilesH;
    const int tw = j % tilesW;
    j /= tilesW;
    const int seg = j % nseg;
    const int b = j / nseg;
    const int oh0 = th * C::TH, ow0 = tw * C::TW, n0 = nb * 32;
    const int s_beg = seg * steps_per_seg;
    const int s_end = (s_beg + steps_per_seg < steps_total) ? s_beg + steps_per_seg : steps_total;
    const int nsteps = s_end - s_beg;
    const int d0 = 2 * s_beg;
    const u32x4_t rx = make_rsrc(x + (long long)b * xv.sb, x_bytes), ry = make_rsrc(y + (long long)b * yv.sb, y_bytes);
    const u32x4_t rx1 = (KH == 2 && xkh) ? make_rsrc(x + xkh + (long long)b * xv.sb, x_bytes) : rx;      // K half 1
    const int khoff = (KH == 2 && xkh) ? 0 : 32;

    // weights: fragment (tap) of this wave's 16 output channels x 32 input channels, from the image of the generic kernels
    // [N/32][K/16][ntaps][2][32][8]: this lane holds W[n0 + 16 chalf + v][8 q .. 8 q + 7]
    if (nb != nb_loaded) {
#pragma unroll
      for (int t = 0; t < 27; ++t) {
        const int wt = taps.wt[t];
#pragma unroll
        for (int kh = 0; kh < KH; ++kh) {
          const long long idx = ((((long long)nb * (2 * KH) + 2 * kh + (q >> 1)) * ntaps_src + wt) * 2 + (q & 1)) * 32 + chalf * 16 + v;
          wreg[kh][t] = *reinterpret_cast<const uint4 *>(w + idx * 8);
        }
      }
      nb_loaded = nb;
    }
    float bv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = (!GST && bias) ? bias[n0 + chalf * 16 + 4 * q + r] : 0.f;
    const __amdgpu_buffer_rsrc_t rg =
        __builtin_amdgcn_make_buffer_rsrc((void *)(gst.y + (long long)b * gst.v.sb), (short)0, (int)gst.bytes, 0x00020000);
    if (GST) {      // the job's InstanceNorm constants (published by the barriers in front of the first epilogue)
      if (tid < 32) {
        const int c = n0 + tid;
        const float mu = gst.mr[((long long)b * Cout + c) * 2], rs = gst.mr[((long long)b * Cout + c) * 2 + 1];
        const float a = gst.gamma[c] * rs;
        gcst[tid * 2] = a;
        gcst[tid * 2 + 1] = gst.beta[c] - mu * a;
      }
    }
    // the compiler's wait for these loads belongs HERE: left to the first use inside the step loop it becomes a
    // vmcnt(0) per step, which also waits for the epilogue's stores
#pragma unroll
    for (int t = 0; t < 27; ++t)
#pragma unroll
      for (int kh = 0; kh < KH; ++kh) asm volatile("" ::"v"(wreg[kh][t].x), "v"(wreg[kh][t].w));
#pragma unroll
    for (int r = 0; r < 4; ++r) asm volatile("" ::"v"(bv[r]));

    // per-lane source offsets (bytes inside a plane) of this wave's DMA pieces: piece idx = wave + 8 i of the 44 that make
    // up two planes; lane -> voxel e = 16 P + lane / 4 of the plane, position lane & 3 inside the voxel
    // (KH = 2: computed at the issue instead - 13 more live registers would spill inside the step loop, and a scratch reload's
    // wait also waits for the DMA in flight)
    auto piece_off = [&](int i) -> unsigned {
      const int idx = wave + C::NW * i;
      const int P = idx % C::SUB_PIECES, kh = (idx / C::SUB_PIECES) % KH;      // piece of sub-plane kh of plane idx / PIECES
      const int e = P * 16 + (lane >> 2), row = e / C::IW, u = e - row * C::IW;
      const int g = (lane & 3) ^ (((u >> 2) & 1) << 1);
      const int gh = oh0 - 1 + row, gw = ow0 - 1 + u;
      const bool ok = idx < 2 * C::PIECES && e < C::NVOX && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W;
      return ok ? (unsigned)((gh * xv.sh + gw * xv.sw + kh * khoff + g * 8) * 2) : OOB;
    };
    unsigned poff[KH == 1 ? C::NPW : 1];
    if (KH == 1) {
#pragma unroll
      for (int i = 0; i < C::NPW; ++i) poff[i] = piece_off(i);
    }
    // output: lane (row rho = q, voxel v) stores 8 channels of voxel v + 16 (rho & 1) after the row swaps
    const int ovox = v + 16 * (q & 1);
    const unsigned ooff = (ow0 + ovox < W) ? (unsigned)((ovox * yv.sw + chalf * 16 + (q >> 1) * 8) * 2) : OOB;
    const unsigned goff = (GST && ow0 + ovox < W) ? (unsigned)((ovox * gst.v.sw + chalf * 16 + (q >> 1) * 8) * 2) : OOB;
    const bool wv0 = ow0 + v < W, wv1 = ow0 + 16 + v < W;      // this lane's voxel of tile half 0 / 1 inside the volume
    const bool full_w = ow0 + 32 <= W;

    // DMA piece i of the plane pair whose first plane is input plane ip (global plane d0 - 1 + ip) into ring pair pr
    auto issue_piece = [&](int i, int pr, int ip) -> int {
      const int idx = wave + C::NW * i;
      if (C::NW * i + C::NW - 1 < 2 * C::PIECES || idx < 2 * C::PIECES) {
        const int gd = d0 - 1 + ip + idx / C::PIECES;
        const bool dok = (unsigned)gd < (unsigned)D;
        const unsigned soff = dok ? (unsigned)(gd * xv.sd * 2) : 0u;
        const unsigned voff = (KH == 1 ? poff[KH == 1 ? i : 0] : piece_off(i)) | (dok ? 0u : OOB);
        const bool half1 = KH == 2 && ((idx / C::SUB_PIECES) % KH) == 1;          // (wave-uniform)
        dma16_buf_to_lds(half1 ? rx1 : rx, voff, soff, lds0 + pr * (2 * C::PLANE) + idx * 1024);
        return 1;
      }
      return 0;
    };

    f32x4_t acc[2][2][2];
    f32x2_t s1[2] = {{0.f, 0.f}, {0.f, 0.f}}, s2[2] = {{0.f, 0.f}, {0.f, 0.f}};      // forward sums of this lane's 4 channels
    f32x2_t gs1[4], gs2[4];       // GST: sums of this lane's 8 channels (channel pairs: packed fp32 math)
#pragma unroll
    for (int t = 0; t < 4; ++t) gs1[t] = gs2[t] = f32x2_t{0.f, 0.f};
    u32x4_t gy[2];                // GST: y_prev of this lane's voxel and 8 channels, the two rows of the output plane in work
    int issued = 0;     // ... and by this wave in this job

    // GST: the y_prev values of output plane od of step k, fetched two input rows ahead of their use (compiler-visible loads:
    // it waits for them itself - its count ignores the inline-asm operations, which only makes the wait stricter)
    auto load_gy = [&](int k, int od) {
#pragma unroll
      for (int oh = 0; oh < 2; ++oh) {
        const int odg = d0 + 2 * k + od, ohg = oh0 + 2 * ws + oh;
        if (odg < D && ohg < H) {
          const unsigned soff = (unsigned)(((long long)odg * gst.v.sd + (long long)ohg * gst.v.sh + (long long)ow0 * gst.v.sw + n0) * 2);
          gy[oh] = __builtin_amdgcn_raw_buffer_load_b128(rg, goff, soff, 0);
          ++issued;
        }
      }
    };

    // Output row (od, oh) of step k is complete once input row (dz, hy) = (od + 2, oh + 2) has been processed: it is converted
    // and stored right then, between the MFMAs of the remaining input rows (3 of the 4 output rows of a step), not in a block at
    // the end of the step - where all eight waves queued on the store path at once (stamps: 1600-1900 cycles per step).
    auto epilogue_row = [&](int k, int od, int oh) {
      const int odg = d0 + 2 * k + od, ohg = oh0 + 2 * ws + oh;
      if (odg < D && ohg < H) {       // wave-uniform
        const f32x4_t a0 = acc[od][oh][0], a1 = acc[od][oh][1];
        if (!GST) {      // forward statistics of the unrounded values: channel pairs as packed fp32 (v_pk_add_f32 / v_pk_fma_f32)
          auto sums = [&](const f32x2_t l0, const f32x2_t h0, const f32x2_t l1, const f32x2_t h1) {
            s1[0] += l0 + l1;
            s1[1] += h0 + h1;
            s2[0] = __builtin_elementwise_fma(l0, l0, __builtin_elementwise_fma(l1, l1, s2[0]));
            s2[1] = __builtin_elementwise_fma(h0, h0, __builtin_elementwise_fma(h1, h1, s2[1]));
          };
          const f32x2_t z2 = {0.f, 0.f};
          if (full_w)      // (two code paths on purpose: merged, the compiler copies all eight values of the common case)
            sums(__builtin_shufflevector(a0, a0, 0, 1), __builtin_shufflevector(a0, a0, 2, 3),
                 __builtin_shufflevector(a1, a1, 0, 1), __builtin_shufflevector(a1, a1, 2, 3));
          else             // a voxel beyond W contributes nothing
            sums(wv0 ? __builtin_shufflevector(a0, a0, 0, 1) : z2, wv0 ? __builtin_shufflevector(a0, a0, 2, 3) : z2,
                 wv1 ? __builtin_shufflevector(a1, a1, 0, 1) : z2, wv1 ? __builtin_shufflevector(a1, a1, 2, 3) : z2);
        }
        const unsigned x0 = pack2_pk<T16>(a0[0], a0[1]), x1 = pack2_pk<T16>(a0[2], a0[3]);
        const unsigned y0 = pack2_pk<T16>(a1[0], a1[1]), y1 = pack2_pk<T16>(a1[2], a1[3]);
        // rows (16 lanes) 1 and 3 of the half-0 tile change places with rows 0 and 2 of the half-1 tile: afterwards a lane
        // holds 8 consecutive channels of ONE voxel (rows 0, 2: voxel v; rows 1, 3: voxel 16 + v)
        const auto p0 = __builtin_amdgcn_permlane16_swap(x0, y0, false, false);
        const auto p1 = __builtin_amdgcn_permlane16_swap(x1, y1, false, false);
        const u32x4_t val = {p0[0], p1[0], p0[1], p1[1]};
        const unsigned soff = (unsigned)(((long long)odg * yv.sd + (long long)ohg * yv.sh + (long long)ow0 * yv.sw + n0) * 2);
        if (ABL != 2) {
          store16_buf<NT_ST>(ry, ooff, soff, val);
          ++issued;
        } else {
          asm volatile("" ::"v"(val), "s"(soff));
        }
        if (GST && goff != OOB) {       // (a voxel beyond W contributes nothing)
          const u32x4_t yq = gy[oh];
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const float4 c4 = *reinterpret_cast<const float4 *>(gcst + (chalf * 16 + (q >> 1) * 8 + 2 * t) * 2);      // (A, B) of two channels
            const f32x2_t gA = {c4.x, c4.z}, gB = {c4.y, c4.w};
            float g0, g1, yy0, yy1;
            unpack2_16<T16>(val[t], g0, g1);
            unpack2_16<T16>(yq[t], yy0, yy1);
            const f32x2_t g2 = {g0, g1}, y2 = {yy0, yy1};
            const f32x2_t a2 = __builtin_elementwise_fma(gA, y2, gB);
            const f32x2_t gsl = g2 * gst.slope;
            const f32x2_t q2 = {a2[0] > 0.f ? g2[0] : gsl[0], a2[1] > 0.f ? g2[1] : gsl[1]};
            gs1[t] += q2;
            gs2[t] = __builtin_elementwise_fma(q2, y2, gs2[t]);
          }
        }
      }
    };

    // Synchronisation: ONE barrier per step, in its middle.  Step k reads planes 2k, 2k+1 (input rows rs 0..7, ring pair k % 3)
    // and 2k+2, 2k+3 (rs 8..15, pair (k+1) % 3).  Barrier M_k sits before the first read of rs 8: in front of it every wave has
    // waited for its own DMA pieces of planes 2k+2, 2k+3, so behind it they have all landed; and every wave has finished rs 0..7
    // of step k - 1... of every earlier step, so pair (k+2) % 3 (planes 2k-2, 2k-1, last read in front of M_{k-1}) is free: it takes
    // planes 2k+4, 2k+5 during rs 0..5 of step k (one DMA piece per row), first read behind M_{k+1}: more than a step to land.
    // A wave's vector-memory operations complete in issue order; `issued` counts them, a mark remembers the count behind each
    // DMA group, and the wait in front of M_k leaves exactly the younger operations (the next group, the epilogue's stores, GST's
    // y loads) in flight.  Tried and dropped (profiles/r04_ab.txt): the two waves of a SIMD on opposite sides of the barrier
    // (waves 4-7 running whole steps between barriers, so that each wave's conversion / store phase falls into its partner's MFMA
    // stream): 0.786 vs 0.774 ms (static wave priorities either way: no change) - the older wave of a pair wins the arbitration, finishes its interval early and waits ~2900
    // cycles per step at the barrier while the younger one runs alone, at the ~60 % MFMA rate one wave's LDS latencies allow.
    uint4 fr[2][6 * KH];
    auto load_row = [&](int rs, int kkv, uint4(&f)[6 * KH]) {
      const int dz = rs >> 2, hy = rs & 3;
      int slot = 2 * kkv + dz;
      slot = slot >= 6 ? slot - 6 : slot;
      const int sb = slot * C::PLANE;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
          for (int kh = 0; kh < KH; ++kh)
            f[(kw * 2 + hf) * KH + kh] = *reinterpret_cast<const uint4 *>(smem + (aoff[kw] + sb) + kh * C::SUB + hy * C::ROWB + hf * 1024);
    };
    auto init_acc = [&]() {
#pragma unroll
      for (int od = 0; od < 2; ++od)
#pragma unroll
        for (int oh = 0; oh < 2; ++oh)
#pragma unroll
          for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[od][oh][hf][r] = bv[r];
    };
    int mark_cur = 0, mark_nxt = 0;
    auto issue_group = [&](int pr, int ip) {
#pragma unroll
      for (int i = 0; i < C::NPW; ++i) issued += issue_piece(i, pr, ip);
      return issued;
    };
    // input rows [RS0, RS1) of step k (ring phase kkv): the six fragments of row rs + 1 (3 tap shifts x 2 voxel halves) are read
    // while the MFMAs of row rs issue; fr[RS0 & 1] holds row RS0 on entry; on exit fr[RS1 & 1] holds row RS1 (RS1 < 16).
    // BAR7: the barrier M_k sits between rs 7 and rs 8 (in front of the first read of rs 8: the MFMAs of row 7 cover the latency
    // of those reads).  DMA0: the row at which this wave issues its first DMA piece of planes 2k+4, 2k+5 (-1: not in this block).
    auto rows = [&](auto rs0c, auto rs1c, auto bar7c, auto dma0c, int k, int kkv, bool more) {
      constexpr int RS0 = decltype(rs0c)::value, RS1 = decltype(rs1c)::value, DMA0 = decltype(dma0c)::value;
      constexpr bool BAR7 = decltype(bar7c)::value;
      const int prn = kkv == 0 ? 2 : kkv - 1;      // (k + 2) % 3
#pragma unroll
      for (int rs = RS0; rs < RS1; ++rs) {
        const int dz = rs >> 2, hy = rs & 3;
        if (BAR7 && rs == 7) {
          stamp(1);
          vm_wait_all_but(issued - mark_cur);
          lds_barrier();
          stamp(2);
        }
        // the row's DMA piece (wave-uniform branches) first; then ONE basic block: the six (twelve) fragment reads of row rs + 1
        // dealt out BETWEEN the MFMAs of row rs.  Round 5: with the reads in front of the MFMAs and the DMA between them, a wave
        // issued no MFMA for ~150 cycles per row - hidden by the SIMD's second wave with 32 input channels, plain idle time of
        // the matrix pipe with 64 (one wave per SIMD): profiles/r05_ab.txt.
        if (DMA0 >= 0 && more) {
          if (ABL == 3) {
            if (rs == DMA0) mark_nxt = issue_group(prn, 2 * k + 4);
          } else if (rs >= DMA0 && rs < DMA0 + C::NPW) {
            issued += issue_piece(rs - DMA0, prn, 2 * k + 4);
            if (rs == DMA0 + C::NPW - 1) mark_nxt = issued;
          }
        }
        if (GST && rs == 8) load_gy(k, 0);
        if (GST && rs == 12) load_gy(k, 1);
        __builtin_amdgcn_sched_barrier(0);
        if (rs + 1 < 16) load_row(rs + 1, kkv, fr[(rs + 1) & 1]);
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
          for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int kd = 0; kd < 3; ++kd)
#pragma unroll
              for (int kh = 0; kh < 3; ++kh) {
                const int od = dz - kd, oh = hy - kh;
                if (od >= 0 && od < 2 && oh >= 0 && oh < 2) {
#pragma unroll
                  for (int kq = 0; kq < KH; ++kq)
                    mfma16<T16>(wreg[kq][(kd * 3 + kh) * 3 + kw], fr[rs & 1][(kw * 2 + hf) * KH + kq], acc[od][oh][hf]);
                }
              }
        if (rs + 1 < 16) {
          const int nm = ((dz == 1 || dz == 2) ? 2 : 1) * ((hy == 1 || hy == 2) ? 2 : 1);      // output rows this input row feeds
          if (nm == 1) ring_sched_interleave<6 * KH, 6 * KH>();
          else if (nm == 2) ring_sched_interleave<12 * KH, 6 * KH>();
          else ring_sched_interleave<24 * KH, 6 * KH>();
        }
        __builtin_amdgcn_sched_barrier(0);
        if (rs == 15) stamp(3);
        if (rs == 10 || rs == 11 || rs == 14 || rs == 15) epilogue_row(k, (rs >> 2) - 2, (rs & 3) - 2);
      }
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 16> I16;

    lds_barrier();      // every wave is done with the previous job's planes
    if (ABL == 6) tprev = __builtin_amdgcn_s_memtime();
    const int mark_p = issue_group(0, 0);
    mark_cur = issue_group(1, 2);      // planes 2, 3: needed behind M_0
    vm_wait_all_but(issued - mark_p);
    lds_barrier();      // planes 0, 1 are there
    stamp(0);
    int kk = 0;         // k mod 3: ring pair that holds the step's first two planes
    {
      for (int k = 0; k < nsteps; ++k) {
        const bool more = k + 1 < nsteps && ABL != 1;
        mark_nxt = issued;
        init_acc();
        load_row(0, kk, fr[0]);
        rows(I0{}, I16{}, std::true_type{}, I0{}, k, kk, more);
        stamp(4);
        mark_cur = mark_nxt;
        kk = kk == 2 ? 0 : kk + 1;
      }
    }

    if (stats && GST) {
      // this lane's 8 channels are shared by the 32 lanes of rows {0, 1} or {2, 3}
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float a = gs1[e >> 1][e & 1], c2 = gs2[e >> 1][e & 1];
#pragma unroll
        for (int m = 1; m < 32; m <<= 1) {
          a += __shfl_xor(a, m, 64);
          c2 += __shfl_xor(c2, m, 64);
        }
        if ((lane & 31) == 0) {
          red[(wave * 16 + (q >> 1) * 8 + e) * 2 + 0] = a;
          red[(wave * 16 + (q >> 1) * 8 + e) * 2 + 1] = c2;
        }
      }
    }
    if (stats) {
      // per-channel sums of the job: over the 16 lanes (voxels) of a row, then over the four row-pair waves of a channel half
#pragma unroll
      for (int r = 0; r < 4 && !GST; ++r) {
        float a = s1[r >> 1][r & 1], c2 = s2[r >> 1][r & 1];
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) {
          a += __shfl_xor(a, m, 64);
          c2 += __shfl_xor(c2, m, 64);
        }
        if (v == 0) {
          red[(wave * 16 + 4 * q + r) * 2 + 0] = a;
          red[(wave * 16 + 4 * q + r) * 2 + 1] = c2;
        }
      }
      lds_barrier();
      const int nblk = tilesH * tilesW * nseg;
      if (tid < 32 && n0 + tid < Cout) {
        const int ch = tid >> 4, cc = tid & 15;
        double s = 0.0, ss = 0.0;
#pragma unroll
        for (int wq = 0; wq < C::NW / 2; ++wq) {
          s += (double)red[((wq * 2 + ch) * 16 + cc) * 2 + 0];
          ss += (double)red[((wq * 2 + ch) * 16 + cc) * 2 + 1];
        }
        const int slot = (seg * tilesW + tw) * tilesH + th;
        double *pp = stats + 32 + (((int64_t)b * nblk + slot) * Cout + n0 + tid) * 2;
        pp[0] = s;
        pp[1] = ss;
      }
      if (blockIdx.x == 0 && tid == 0 && rd == 0) reinterpret_cast<long long *>(stats)[0] = nblk;
    }
  }
  if (ABL == 6 && stats && lane == 0) {      // behind everything the finalize kernels read (profiles/tools/ring_stamps.py knows the place)
    double *o = stats + (1 << 20) + ((size_t)blockIdx.x * C::NW + wave) * 8;
    for (int i = 0; i < 5; ++i) o[i] = (double)tseg[i];
    o[5] = (double)(__builtin_amdgcn_s_memtime() - t_begin);
    o[6] = (double)(__builtin_amdgcn_s_memrealtime() - rt_begin);
  }
}

}  // namespace

namespace {

template <int KH>
int ring_launch_kh(const void *x, const View &xv, const void *w, const Taps &taps, const float *bias, void *y, const View &yv,
                   int B, int Cout, double *stats, int64_t stats_cap_slots, int ntaps_src, int is_f16, hipStream_t st,
                   RowsGstCtx *gctx, long long xkh, bool dry) {
  typedef RingCfgT<KH> C;
  if (xkh && (KH != 2 || xkh % 8 || xv.sw != 32)) return DGTTA_ERR_UNSUPPORTED;      // K halves as planes: 64 input channels, dense 32-channel rows
  // the buffer descriptors address one sample with 32-bit byte offsets
  const long long xb = ((long long)(xv.D - 1) * xv.sd + (long long)(xv.H - 1) * xv.sh + (long long)(xv.W - 1) * xv.sw + (xkh ? 32 : 32 * KH)) * 2;
  const long long yb = ((long long)(yv.D - 1) * yv.sd + (long long)(yv.H - 1) * yv.sh + (long long)(yv.W - 1) * yv.sw + Cout) * 2;
  if (xb >= (1ll << 31) || yb >= (1ll << 31)) return DGTTA_ERR_UNSUPPORTED;
  if (xv.sw % 8 || xv.sh % 8 || xv.sd % 8 || xv.sb % 8 || yv.sw % 8 || yv.sh % 8 || yv.sd % 8 || yv.sb % 8 || ((uintptr_t)x & 15) ||
      ((uintptr_t)y & 15))
    return DGTTA_ERR_UNSUPPORTED;
  static int ncu_dev = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n > 0 ? n : 256;
  }();
  const int ncu = DG_LAB_NCU(ncu_dev);
  const int tW = cdiv(yv.W, C::TW), tH = cdiv(yv.H, C::TH), nblkN = Cout / 32;
  const int steps_total = (yv.D + 1) / 2;
  const long long ncol = (long long)B * nblkN * tW * tH;
  // segments along D: as few as fill the chip evenly (a segment start costs about one and a half steps of exposed latency)
  int nseg = 1;
  {
    double best = 1e300;
    for (int s = 1; s <= 32 && s <= steps_total; s *= 2) {
      const int sps = cdiv(steps_total, s);
      const int ns = cdiv(steps_total, sps);
      const double t = (double)cdiv64(ncol * ns, ncu) * (sps + 1.5);
      if (t < best * 0.97) best = t, nseg = ns;
    }
  }
  const int sps = cdiv(steps_total, nseg);
  nseg = cdiv(steps_total, sps);
  if (stats && (int64_t)tW * tH * nseg > stats_cap_slots) return DGTTA_ERR_UNSUPPORTED;
  const long long njobs = ncol * nseg;
  if (njobs >= (1ll << 31)) return DGTTA_ERR_UNSUPPORTED;
  // a data gradient that is asked to leave the InstanceNorm backward sums of the previous block (dgtta_conv3d_k3_dgrad_gstats)
  RingGst ga{};
  const int abl = DG_LAB(ring_abl);      // laboratory builds (libdgtta_hip_diag.so only): timing models and cycle stamps
  const long long gb = gctx ? ((long long)(yv.D - 1) * yv.H * yv.W * gctx->ldy + (long long)(yv.H - 1) * yv.W * gctx->ldy +
                               (long long)(yv.W - 1) * gctx->ldy + Cout) * 2 : 0;
  const bool gst_on = gctx && !stats && !bias && gctx->ldy % 8 == 0 && ((uintptr_t)gctx->y & 15) == 0 && gb < (1ll << 31) &&
                      (int64_t)tW * tH * nseg <= stats_cap_slots && abl < 0;
  if (gst_on) {
    ga.y = (const bf16_t *)gctx->y;
    ga.v = dense_view(B, yv.D, yv.H, yv.W, (int)gctx->ldy);
    ga.mr = gctx->mr;
    ga.gamma = gctx->gamma;
    ga.beta = gctx->beta;
    ga.slope = gctx->slope;
    ga.bytes = (unsigned)gb;
    stats = gctx->out;
    if (!dry) gctx->produced = 1;
  }
  if (dry) return DGTTA_OK;      // (dgtta_conv3d_k3_blocked_supported: would this launch be taken?)
  const int grid = (int)(njobs < ncu ? njobs : ncu);
  // DGTTA_RING_NT=1 (diagnostic build): non-temporal output stores (measured 5 % slower: the two 32-byte halves of a voxel come
  // from two waves and merge in L2)
  const bool nt = DG_LAB(ring_nt) == '1';
#define RING_LAUNCH(T16, NTS, ABLV, GSTV)                                                                                     \
  do {                                                                                                                        \
    auto kern = conv3_ring_kernel<T16, NTS, ABLV, GSTV, KH>;                                                                  \
    static DynLdsOnce once;                                                                                                   \
    DG_REQUIRE(ensure_dyn_lds(once, reinterpret_cast<const void *>(kern), C::LDS_BYTES) == hipSuccess, DGTTA_ERR_LAUNCH,      \
               "conv3_ring: cannot raise the dynamic LDS limit to %d", C::LDS_BYTES);                                         \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NT), C::LDS_BYTES, st, (const bf16_t *)x, xv, (const bf16_t *)w, taps, bias, \
                       (bf16_t *)y, yv, Cout, tW, tH, nblkN, nseg, sps, (int)njobs, stats, ntaps_src, (unsigned)xb,           \
                       (unsigned)yb, ga, xkh);                                                                                \
  } while (0)
  bool diag = false;
#ifdef DGTTA_DIAG
  if constexpr (KH == 1) {      // diagnostic builds exist for the fp16 / 32-channel instantiation only; exact values only
    if (is_f16 && (abl == '1' || abl == '2' || abl == '3')) {
      diag = true;
      if (abl == '1') RING_LAUNCH(f16_t, false, 1, false);
      else if (abl == '2') RING_LAUNCH(f16_t, false, 2, false);
      else RING_LAUNCH(f16_t, false, 3, false);
    } else if (is_f16 && abl == '6') {
      // cycle stamps go to stats + 2^20 doubles, past what dgtta_conv3d_stats_bytes covers: only profiles/tools/ring_stamps.py,
      // which allocates that area itself, may ask for this build (it does not exist in the product library)
      diag = true;
      RING_LAUNCH(f16_t, false, 6, false);
    }
    if (!diag && nt) {
      diag = true;
      if (is_f16) RING_LAUNCH(f16_t, true, 0, false);
      else RING_LAUNCH(bf16_t, true, 0, false);
    }
  }
#endif
  if (diag) {
  } else if (gst_on) {
    if (is_f16) RING_LAUNCH(f16_t, false, 0, true);
    else RING_LAUNCH(bf16_t, false, 0, true);
  } else if (is_f16) {
    RING_LAUNCH(f16_t, false, 0, false);
  } else {
    RING_LAUNCH(bf16_t, false, 0, false);
  }
#undef RING_LAUNCH
  (void)nt;
  DG_CHECK_LAUNCH("conv3_ring_kernel");
  return DGTTA_OK;
}

}  // namespace

// Entry point used by the dispatcher in conv_mfma.hip: DGTTA_ERR_UNSUPPORTED when the shape is not this kernel's.
int conv3_ring_launch(const void *x, const View &xv, const void *w, const Taps &taps, const float *bias, void *y, const View &yv,
                      int B, int Cin, int Cout, int CinP, int CoutP, double *stats, int64_t stats_cap_slots, int ntaps_src,
                      int is_f16, hipStream_t st, RowsGstCtx *gst, long long xkh, bool dry) {
  if (Cin != CinP || (Cin != 32 && Cin != 64) || Cout % 32 != 0 || CoutP != Cout) return DGTTA_ERR_UNSUPPORTED;
  if (xv.D != yv.D || xv.H != yv.H || xv.W != yv.W) return DGTTA_ERR_UNSUPPORTED;
  if (Cin == 64) {
    if (dgtta_switches().conv_ring == '3') return DGTTA_ERR_UNSUPPORTED;      // DGTTA_CONV_RING=3: the ring for 32 input channels only
    return ring_launch_kh<2>(x, xv, w, taps, bias, y, yv, B, Cout, stats, stats_cap_slots, ntaps_src, is_f16, st, gst, xkh, dry);
  }
  return ring_launch_kh<1>(x, xv, w, taps, bias, y, yv, B, Cout, stats, stats_cap_slots, ntaps_src, is_f16, st, gst, xkh, dry);
}
