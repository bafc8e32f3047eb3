// Weight gradient of the stride-1 3x3x3 conv, 16-bit storage, as a persistent D-sweep with a deep LDS ring (round 4):
//   dW[tap][ci][co] = sum_v x[v + tap - 1][ci] * dy[v][co]      (reference: autograd of the Conv3d blocks of the PlainConvUNet,
//   dg_tta/tta/tta.py:275 `loss_accum.backward()`; topology dg_tta/__resources__/dummy_results/*/plans.json:279-401)
// Same arithmetic, operand reads (ds_read_b64_tr_b16 out of voxel-major LDS slices filled by LDS-DMA), MFMA shape
// (32x32x16: M = ci, N = co, K = 16 voxels along W), tap split (7 taps per wave) and slab output as conv3_wgrad_tr_kernel
// (conv_wgrad.hip).  What differs is the schedule that kernel was bound by - its DMA of slice d + 2 was issued during slice
// d and waited for at the END of slice d (under one slice = ~0.9 us to land, with a workgroup-wide drain and barrier):
//   * ONE 8-wave workgroup per CU, persistent over (column, D-segment) jobs, column = 8 rows x 32 voxels (halo 1.33x instead
//     of 1.59x); waves 2r and 2r + 1 ... wave w owns row half (w & 1) and taps (w >> 1) + 4 i: 56 MFMAs per wave and slice;
//   * x ring of 5 slices, dy ring of 3: the DMA of x(d + 3), dy(d + 2) is issued during slice d and first read in slice d + 2;
//     the wait in front of the (single) barrier of a slice is a counted vmcnt that leaves the youngest group in flight;
//   * buffer-addressed DMA with per-lane offsets precomputed per job (zero fill = out-of-range offset), as conv_ring.hip;
//   * also built: v_mfma_f32_16x16x32 (K = the row's 32 voxels, four 16x16 accumulators per tap, DGTTA_WGRAD_RING=4), with the
//     half-swapped LDS image that makes its operand reads conflict free - measured within +-2 % of the 32x32x16 form;
//   * the accumulators live across all jobs of the workgroup: one slab per workgroup (<= 256 per channel-block pair), the two
//     row halves of a tap are added through LDS once, at the very end.
// Measured (MI355X, fp16, 8 x 128^3, 200 back-to-back launches, same box): 32 -> 32 0.948 -> 0.933 ms, 64 -> 32 1.96 -> 1.81 ms
// (one dy stream per channel-block pair less halo).  Stamps (profiles/tools/wring_clock.py): in-kernel clock 1.87-1.98 GHz, MFMA
// busy 60 % of the wave time, 19 % at the barrier (the older wave of a SIMD pair 30 %, the younger 8 %).  Timing models (wrong
// results): without the DMA inside the sweep the cycles fall 8 % and the CLOCK rises from 1.98 to 2.39 GHz (0.82 -> 0.64 ms);
// without the LDS operand reads -16 % cycles at 2.21 GHz: the sweep is power bound, and what it spends on moving its two
// streamed operands (1.7x the DMA bytes per FLOP of the forward ring kernel) sets the clock.
// Round 5 (PAIR, REUSE below): two taps per MFMA for the 12-channel first layer; whole (kd, kw) tap columns per wave group so that
// four of a wave's seven x operands per row are the previous row's registers; the next iteration's operand reads placed between
// the MFMAs; ring slots as counters; DMA pieces without branches.  32 -> 32: 0.886 -> 0.80 ms per 8 x 128^3 launch (1155 TFLOP/s),
// 74 % of the wave time in MFMAs at 1.62 GHz (profiles/r05_ab.txt); DGTTA_WGRAD_RING=6 / =5 select the forms before.
#include "conv_common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

struct WR {
  static constexpr int TH = 8, XR = TH + 2, XW = 34;
  static constexpr int X_ROW_B = XW * 64, XP = (XR * X_ROW_B + 1023) / 1024, X_SLICE_B = XP * 1024;      // 22 pieces
  static constexpr int Y_ROW_B = 32 * 64, YP = TH * 2, Y_SLICE_B = TH * Y_ROW_B;                          // 16 pieces
  static constexpr int NXS = 5, NYS = 3;
  static constexpr int NP = XP + YP, NW = 8, NPW = (NP + NW - 1) / NW;                                    // 38 pieces, 5 per wave
  static constexpr int LDS_BYTES = NXS * X_SLICE_B + NYS * Y_SLICE_B;                                     // 161,792
  static constexpr int NVOX = XR * XW;
};
// behind the rings: 1 KiB that swallows the DMA pieces a wave issues for nothing (REUSE form: every wave issues one piece in each of a
// slice's first NPW iterations, branch-free; a piece that does not exist - 38 pieces on 8 waves, the tail of a job - reads
// out of range and lands here)
constexpr int WR_LDS_TOTAL = WR::LDS_BYTES + 1024;
static_assert(WR_LDS_TOTAL <= 160 * 1024, "wgrad ring does not fit the LDS");
static_assert(4 * 7 * 16 * 64 * 4 <= WR::LDS_BYTES, "final combine buffer");

__device__ __forceinline__ void wr_dma16(u32x4_t rsrc, unsigned voff, unsigned soff, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_addr)
               : "memory");
}
__device__ __forceinline__ void wr_wait_all_but(int n) {      // the n youngest vector-memory operations may stay in flight
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
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;      // (n > 10 never happens; 0 is always safe)
  }
}
__device__ __forceinline__ u32x4_t wr_rsrc(const void *base, unsigned bytes) {
  const unsigned long long a = (unsigned long long)base;
  u32x4_t r;
  r[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
  r[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);
  r[2] = __builtin_amdgcn_readfirstlane(bytes);
  r[3] = 0x00020000u;
  return r;
}
constexpr unsigned WR_OOB = 0x80000000u;

typedef __attribute__((ext_vector_type(8))) short wr_s16x8_t;
__device__ __forceinline__ bf16x8_t wr_operand(const unsigned char *p) {      // 8 consecutive voxels (k) of this lane's channel
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t *)p);
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t *)(p + 4 * 64));
  const wr_s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}
__device__ __forceinline__ bf16x8_t wr_operand2(const unsigned char *p0, const unsigned char *p1) {      // the two reads at their own addresses
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t *)p0);
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t *)p1);
  const wr_s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}
template <typename T16>
__device__ __forceinline__ f32x16_t wr_mfma(const bf16x8_t &a, const bf16x8_t &b, const f32x16_t &acc);
template <>
__device__ __forceinline__ f32x16_t wr_mfma<bf16_t>(const bf16x8_t &a, const bf16x8_t &b, const f32x16_t &acc) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x16_t wr_mfma<f16_t>(const bf16x8_t &a, const bf16x8_t &b, const f32x16_t &acc) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), acc, 0, 0, 0);
}

typedef __attribute__((ext_vector_type(4))) float f32x4_t;
template <typename T16>
__device__ __forceinline__ f32x4_t wr_mfma16(const bf16x8_t &a, const bf16x8_t &b, const f32x4_t &acc);
template <>
__device__ __forceinline__ f32x4_t wr_mfma16<bf16_t>(const bf16x8_t &a, const bf16x8_t &b, const f32x4_t &acc) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x4_t wr_mfma16<f16_t>(const bf16x8_t &a, const bf16x8_t &b, const f32x4_t &acc) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), acc, 0, 0, 0);
}

// scheduling pattern of one (row, k-step) iteration: NM MFMAs with NR LDS reads dealt out behind them, the first MFMA first
template <int NM, int NR>
__device__ __forceinline__ void wr_sched_interleave() {
  if constexpr (NM > 0) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    constexpr int r = (NR + NM - 1) / NM;
    if constexpr (r > 0) __builtin_amdgcn_sched_group_barrier(0x100, r, 0);
    wr_sched_interleave<NM - 1, NR - r>();
  }
}

// REUSE tap of slot i in wave group wq (see the kernel's comment); < 0 never: slots that are not stored repeat a real tap
__device__ __forceinline__ int wr_reuse_tap(int wq, int i, bool pair, int hsel) {
  const int last = pair ? 3 : 6;
  const int col = i == last ? 8 : (pair ? 2 * wq + hsel : 2 * wq + i / 3);
  const int kh = i == last ? (wq < 2 ? wq : 2) : i % 3;
  return (col / 3) * 9 + kh * 3 + col % 3;
}
__device__ __forceinline__ bool wr_reuse_stored(int wq, int i, bool pair, int half) {      // half: PAIR only (which 16 rows)
  const int last = pair ? 3 : 6;
  return i != last || (wq < 3 && half == 0);
}

// grid: (G workgroups, channel-block pairs); jobs (b, d-segment, tw, th) dealt so that the 32 workgroups of an XCD hold
// columns that are neighbours along H (their halo rows meet in one L2); slabs[(pair * G + blockIdx.x)][27][32 ci][32 co]
// PAIR (round 5, Cin <= 16: the network's first layer, 12 MIND channels in rows of 16): the 32 M rows of an MFMA carry TWO taps x 16
// input channels instead of one tap x 32 of which 16-20 are padding - lanes 16..31 of each half wave read channels 0..15 at the
// SECOND tap's shift instead of channels 16..31 at the first's.  14 tap pairs over the 4 wave groups = 4 accumulators per wave
// instead of 7: 4/7 of the MFMAs and operand reads for the same result (the slab keeps one [32 ci][32 co] block per tap; rows
// ci >= 16 of a block are not written and the reduction never reads rows ci >= Cin).
// REUSE (round 5): the x operand of (output row oh, tap kh + 1) IS the operand of (row oh + 1, tap kh).  A wave group therefore takes
// whole (kd, kw) COLUMNS of taps - columns 2 wq and 2 wq + 1 in slots 0..2 / 3..5 (kh = slot % 3), and tap kh = wq of the ninth
// column in slot 6 (group 3: a discarded repeat) - walks the four rows of its half with the k-step OUTSIDE, and reads only the
// three operands that are new in a row (kh = 2 of each column and the single tap); the other four are the previous row's
// registers.  16 + 4 instead of 28 + 4 operand reads per row half and k-step: the sweep was bound by the LDS read rate
// (512 KiB per CU and slice = 4096 clk against 3584 clk of MFMA), not by the matrix pipe.  With PAIR the
// pair slot s < 3 holds tap kh = s of columns 2 wq (rows 0..15) and 2 wq + 1 (rows 16..31), slot 3 the ninth column's tap.
template <typename T16, bool MF16, bool CLK = false, bool PAIR = false, bool REUSE = false>      // MF16: v_mfma_f32_16x16x32 (K = the row's 32 voxels) instead of 32x32x16;
                                                                           // CLK (diagnostic): cycle / real-time stamps behind the slabs
__global__ __launch_bounds__(WR::NW * 64) void conv3_wgrad_ring_kernel(const bf16_t *__restrict__ x, View xv, const bf16_t *__restrict__ dy,
                                                                       View yv, float *__restrict__ slabs, int Cin, int Cout, int tilesW,
                                                                       int tilesH, int nseg, int DR, int cobs, int njobs, unsigned x_bytes,
                                                                       unsigned y_bytes, long long xkh) {
  // xkh (round 6): element distance between the 32-channel blocks of x; 0 = they interleave in the voxel rows (block c at channel
  // 32 c), > 0 = dense 32-channel tensors xkh elements apart (the level-0 concat buffer as planes, see conv_ring.hip)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char *sX = smem, *sY = smem + WR::NXS * WR::X_SLICE_B;
  const unsigned lds0 = lds_addr_of(smem);
  unsigned long long t_begin = 0, rt_begin = 0, t_wait = 0;
  if (CLK) {
    t_begin = __builtin_amdgcn_s_memtime();
    rt_begin = __builtin_amdgcn_s_memrealtime();
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rh = wave & 1, wq = wave >> 1;      // row half, tap residue
  const int D = yv.D, H = yv.H, W = yv.W;
  const int cib = blockIdx.y / cobs, cob = blockIdx.y % cobs;
  const int cin_lim = (Cin + 7) / 8 * 8;

  static_assert(!(PAIR && MF16) && !(REUSE && MF16), "the pair and operand-reuse forms exist for the 32x32x16 MFMA only");
  constexpr int NA = PAIR ? 4 : 7;                         // accumulators (taps, or tap pairs) per wave
  const int hsel = (lane >> 4) & 1;                        // PAIR: which tap of the pair this lane's M rows belong to
  int tap_kd[7], tap_off[7], tap_row[7], tap_kw[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    // (the fourth residue class has 6 taps: its seventh slot repeats tap 26 into a discarded accumulator; PAIR: pair wq + 4 i =
    // taps 2 (wq + 4 i) and + 1, per lane; pairs 14, 15 and the second tap of pair 13 repeat tap 26 into rows that are not stored)
    const int tc = REUSE ? wr_reuse_tap(wq, i, PAIR, hsel)
                         : (PAIR ? ((2 * (wq + 4 * i) + hsel) < 27 ? 2 * (wq + 4 * i) + hsel : 26) : (wq + 4 * i < 27 ? wq + 4 * i : 26));
    tap_kd[i] = tc / 9;
    tap_off[i] = ((tc / 3) % 3) * WR::X_ROW_B + (tc % 3) * 64;
    tap_row[i] = ((tc / 3) % 3) * WR::X_ROW_B;
    tap_kw[i] = tc % 3;
  }
  // transposed-read lane address inside a 16-voxel x 32-channel block (64-byte voxel rows), as conv3_wgrad_tr_kernel
  // (MF16: 16-lane group g = lane >> 4 takes voxels 8 g .. 8 g + 7 of 16 channels; the channel half is an immediate offset)
  const int lane_off = MF16 ? ((lane >> 4) * 8 + ((lane & 15) >> 2)) * 64 + (lane & 3) * 8
                            : ((lane >> 5) * 8 + ((lane & 15) >> 2)) * 64 + (PAIR ? 0 : ((lane >> 4) & 1) * 32) + (lane & 3) * 8;
  // (the dy operand keeps its own half select: N = 32 output channels in both forms)
  const int lane_off_y = MF16 ? lane_off : ((lane >> 5) * 8 + ((lane & 15) >> 2)) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8;

  // MF16 operand addresses.  A 16-lane group reads 4 voxels x 32 bytes (16 channels), and the two groups of a 32-lane half sit 8
  // voxels = 512 bytes apart: on the plain image both hit the same banks (PMC, profiles/r04_mfma_util.json: SQ_LDS_BANK_CONFLICT =
  // 50 % of the LDS cycles).  So voxels with bit 3 of their row index set store their channel halves swapped, and a lane reads
  // half h of voxel u at byte (h ^ bit3(u)) * 32.  lane_y[s] = offset of half 0 for read s (voxels +0 / +4) without W shift; with
  // the W shift kw of a tap the offset is (lane_y + 64 kw) ^ flip_kw, flip = 32 in the lanes where the shift carries into bit 3;
  // half 1 is that ^ 32.  A tap's kw depends on the wave; it is selected with plain bit masks, because every form of
  // `kw == 1 ? a : b` on these registers ends up as a table in scratch memory behind a pointer select (and a scratch load
  // waits for the DMA in flight).
  int lane_y[2], flip1[2], flip2[2], kw_off[7];
  unsigned kw_m1[7], kw_m2[7];
#pragma unroll
  for (int sr = 0; sr < 2; ++sr) {
    const int uy = (lane >> 4) * 8 + ((lane & 15) >> 2) + 4 * sr;
    lane_y[sr] = uy * 64 + ((uy >> 3) & 1) * 32 + (lane & 3) * 8;
    flip1[sr] = ((((uy + 1) >> 3) ^ (uy >> 3)) & 1) * 32;
    flip2[sr] = ((((uy + 2) >> 3) ^ (uy >> 3)) & 1) * 32;
  }
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    kw_off[i] = tap_kw[i] * 64;
    kw_m1[i] = tap_kw[i] == 1 ? 0xffffffffu : 0u;
    kw_m2[i] = tap_kw[i] == 2 ? 0xffffffffu : 0u;
  }
  // 7 taps x (32 ci x 32 co): one 32x32 accumulator per tap, or four 16x16 ones [ci half][co half] - 112 registers either way
  float accf[7][16];
#pragma unroll
  for (int i = 0; i < 7; ++i)
#pragma unroll
    for (int q = 0; q < 16; ++q) accf[i][q] = 0.f;

  const int G = gridDim.x;
  const bool xcd_order = (G % 8) == 0;
  const int rounds = (njobs + G - 1) / G;
  for (int rd = 0; rd < rounds; ++rd) {
    int j = xcd_order ? (rd * 8 + (int)(blockIdx.x % 8)) * (G / 8) + (int)(blockIdx.x / 8) : rd * G + (int)blockIdx.x;
    if (j >= njobs) continue;
    const int th = j % tilesH;
    j /= tilesH;
    const int tw = j % tilesW;
    j /= tilesW;
    const int seg = j % nseg;
    const int b = j / nseg;
    const int h0 = th * WR::TH, w0 = tw * 32;
    const int d_begin = seg * DR, d_end = (d_begin + DR < D) ? d_begin + DR : D;
    const u32x4_t rx = wr_rsrc(x + (long long)b * xv.sb + (xkh ? cib * xkh : (long long)cib * 32), x_bytes),
                  ry = wr_rsrc(dy + (long long)b * yv.sb + cob * 32, y_bytes);

    // per-lane source offsets of this wave's DMA pieces (piece idx = wave + 8 i of the 22 x + 16 dy pieces of a slice pair):
    // lane -> voxel 16 P + lane / 4, 16-byte channel chunk lane & 3
    unsigned poff[WR::NPW];
#pragma unroll
    for (int i = 0; i < WR::NPW; ++i) {
      const int idx = wave + WR::NW * i;
      // MF16: a voxel whose index in its LDS row has bit 3 set keeps its two 32-byte channel halves swapped (see lane_y below);
      // the LDS side of an LDS-DMA is lane-linear, so the permutation is applied to the SOURCE chunk
      int chunk = lane & 3;
      if (idx < WR::XP) {
        const int e = idx * 16 + (lane >> 2), row = e / WR::XW, u = e - row * WR::XW;
        if (MF16) chunk ^= ((u >> 3) & 1) << 1;
        const int gh = h0 - 1 + row, gw = w0 - 1 + u;
        const bool ok = e < WR::NVOX && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W && cib * 32 + chunk * 8 < cin_lim;
        poff[i] = ok ? (unsigned)((gh * xv.sh + gw * xv.sw + chunk * 8) * 2) : WR_OOB;
      } else {
        const int p = idx - WR::XP, row = p >> 1, vox = 16 * (p & 1) + (lane >> 2);
        if (MF16) chunk ^= ((vox >> 3) & 1) << 1;
        const int gh = h0 + row, gw = w0 + vox;
        const bool ok = idx < WR::NP && gh < H && gw < W && cob * 32 + chunk * 8 < Cout;
        poff[i] = ok ? (unsigned)((gh * yv.sh + gw * yv.sw + chunk * 8) * 2) : WR_OOB;
      }
    }
    // piece i of x slice xd into ring slot xslot / dy slice yd into ring slot yslot (either may be switched off)
    auto issue_piece_at = [&](int i, int xd, int xslot, bool do_x, int yd, int yslot, bool do_y) -> int {
      const int idx = wave + WR::NW * i;
      if (idx < WR::XP) {
        if (!do_x) return 0;
        const bool dok = (unsigned)xd < (unsigned)D;
        wr_dma16(rx, poff[i] | (dok ? 0u : WR_OOB), dok ? (unsigned)(xd * xv.sd * 2) : 0u, lds0 + xslot * WR::X_SLICE_B + idx * 1024);
        return 1;
      }
      if (idx < WR::NP) {
        if (!do_y) return 0;
        const bool dok = (unsigned)yd < (unsigned)D;
        wr_dma16(ry, poff[i] | (dok ? 0u : WR_OOB), dok ? (unsigned)(yd * yv.sd * 2) : 0u,
                 lds0 + WR::NXS * WR::X_SLICE_B + yslot * WR::Y_SLICE_B + (idx - WR::XP) * 1024);
        return 1;
      }
      return 0;
    };
    // the same without a branch (REUSE form; `on` = the slices exist at all): selects on wave-uniform values, always one DMA
    auto issue_piece_always = [&](int i, int xd, int xslot, int yd, int yslot, bool on) {
      const int idx = wave + WR::NW * i;
      const bool isx = idx < WR::XP, valid = on && idx < WR::NP;
      const int sd = isx ? xd : yd;
      const bool dok = valid && (unsigned)sd < (unsigned)D;
      u32x4_t r;
#pragma unroll
      for (int q = 0; q < 4; ++q) r[q] = isx ? rx[q] : ry[q];
      const unsigned lds_piece = isx ? (unsigned)(xslot * WR::X_SLICE_B + idx * 1024)
                                     : (unsigned)(WR::NXS * WR::X_SLICE_B + yslot * WR::Y_SLICE_B + (idx - WR::XP) * 1024);
      wr_dma16(r, poff[i] | (dok ? 0u : WR_OOB), dok ? (unsigned)(sd * (isx ? xv.sd : yv.sd) * 2) : 0u,
               lds0 + (valid ? lds_piece : (unsigned)WR::LDS_BYTES));
    };
    // (x(xd) lives in slot (xd + 1) mod NXS, dy(yd) in slot yd mod NYS)
    auto issue_piece = [&](int i, int xd, bool do_x, int yd, bool do_y) -> int {
      return issue_piece_at(i, xd, (xd + 1 + WR::NXS) % WR::NXS, do_x, yd, (yd + WR::NYS) % WR::NYS, do_y);
    };
    int issued = 0;
    auto issue_group = [&](int xd, bool do_x, int yd, bool do_y) {
#pragma unroll
      for (int i = 0; i < WR::NPW; ++i) issued += issue_piece(i, xd, do_x, yd, do_y);
      return issued;
    };

    lds_barrier();      // every wave is done with the previous job's slices
    issue_group(d_begin - 1, true, 0, false);
    issue_group(d_begin, true, d_begin, true);
    int mark_cur = issue_group(d_begin + 1, true, 0, false);            // needed in slice d_begin
    int mark_nxt = issue_group(d_begin + 2, true, d_begin + 1, true);   // needed in slice d_begin + 1
    // ring slots of x(d - 1) and dy(d), carried along the sweep: no division per slice (round 5: the slot arithmetic of a slice -
    // four magic-number divisions and seven slot selects - sat between the barrier and the slice's first operand read)
    int xr0 = (d_begin + WR::NXS) % WR::NXS, yr0 = (d_begin + WR::NYS) % WR::NYS;
    for (int d = d_begin; d < d_end; ++d) {
      unsigned long long tw0 = 0;
      if (CLK) tw0 = __builtin_amdgcn_s_memtime();
      wr_wait_all_but(issued - mark_cur);      // this wave's pieces of x(d + 1), dy(d) have landed ...
      lds_barrier();                           // ... and everyone's; every wave is done with slice d - 1
      if (CLK) t_wait += __builtin_amdgcn_s_memtime() - tw0;
      const bool more = d + 2 < d_end;         // x(d + 3), dy(d + 2): first read in slice d + 2
      int mark_new = issued;
      const unsigned char *ys0 = sY + yr0 * WR::Y_SLICE_B + rh * 4 * WR::Y_ROW_B;
      const unsigned char *ys = ys0 + lane_off_y;
      // x(d + 3) takes the slot x(d - 2) left, dy(d + 2) the one of dy(d - 1)
      const int xs_new = xr0 == 0 ? WR::NXS - 1 : xr0 - 1, ys_new = yr0 == 0 ? WR::NYS - 1 : yr0 - 1;
      int slice_off[3];
#pragma unroll
      for (int kd = 0; kd < 3; ++kd)
        slice_off[kd] = (xr0 + kd >= WR::NXS ? xr0 + kd - WR::NXS : xr0 + kd) * WR::X_SLICE_B + rh * 4 * WR::X_ROW_B;
      if (!MF16) {
        // operands of (row, k-step) iteration it + 1 are read while the MFMAs of iteration it issue
        // One address register per tap and slice (lane offset + ring slot of the tap's kd + its (kh, kw) offset, the slot picked
        // without a branch): every operand read of the slice is then that register plus an immediate.  Before, each read carried
        // a scalar compare / branch chain for the slot and its own address arithmetic - with two transposed reads per MFMA the
        // sweep was bound by instruction issue, not by the matrix pipe (profiles/r04_ab.txt).
        int xa[NA];
#pragma unroll
        for (int i = 0; i < NA; ++i) {
          int sl = xr0 + tap_kd[i];
          sl = sl >= WR::NXS ? sl - WR::NXS : sl;
          xa[i] = lane_off + sl * WR::X_SLICE_B + rh * 4 * WR::X_ROW_B + tap_off[i];
        }
        bf16x8_t afr[2][NA], bfr[2];
        // (REUSE: k-step outside, rows inside; a slot that is not new in a row takes the previous row's next slot)
        auto load_it = [&](int it, bf16x8_t(&a)[NA], const bf16x8_t(&prev)[NA], bf16x8_t &bb) {
          const int oh = REUSE ? (it & 3) : (it >> 1), ks = REUSE ? (it >> 2) : (it & 1);
          bb = wr_operand(ys + oh * WR::Y_ROW_B + ks * 1024);
#pragma unroll
          for (int i = 0; i < NA; ++i) {
            const bool fresh = !REUSE || oh == 0 || i == NA - 1 || (PAIR ? i == 2 : (i == 2 || i == 5));
            if (fresh) a[i] = wr_operand(sX + xa[i] + oh * WR::X_ROW_B + ks * 1024);
            else a[i] = prev[i + 1];
          }
        };
        auto mfma_it = [&](int it) {
#pragma unroll
          for (int i = 0; i < NA; ++i) {
            f32x16_t c;
#pragma unroll
            for (int q = 0; q < 16; ++q) c[q] = accf[i][q];
            c = wr_mfma<T16>(afr[it & 1][i], bfr[it & 1], c);
#pragma unroll
            for (int q = 0; q < 16; ++q) accf[i][q] = c[q];
          }
        };
        if constexpr (REUSE) {
          // the slice's first operands in the order the MFMAs want them (left to itself the scheduler sorts the reads of
          // iterations 0 and 1 by address, and the first MFMA after the barrier waits for fourteen reads of all eight waves)
          bfr[0] = wr_operand(ys);
#pragma unroll
          for (int i = 0; i < NA; ++i) {
            afr[0][i] = wr_operand(sX + xa[i]);
            if (i % 2 == 0) __builtin_amdgcn_sched_barrier(0);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int it = 0; it < 8; ++it) {
            // one basic block per iteration: the slice's DMA piece (branch-free), then the reads of iteration it + 1 dealt out
            // BETWEEN the MFMAs of iteration it.  (With the piece behind wave-uniform branches and the reads in front of the
            // MFMAs, a wave whose SIMD partner had finished its slice ran at half the MFMA rate: profiles/r05_ab.txt.)
            if (it < WR::NPW) {
              issue_piece_always(it, d + 3, xs_new, d + 2, ys_new, more);
              ++issued;
              if (it == WR::NPW - 1) mark_new = issued;
            }
            if (it + 1 < 8) load_it(it + 1, afr[(it + 1) & 1], afr[it & 1], bfr[(it + 1) & 1]);
            mfma_it(it);
            if (it + 1 < 8) {
              if (((it + 1) & 3) == 0) wr_sched_interleave<NA, 2 * (NA + 1)>();
              else wr_sched_interleave<NA, 2 * ((PAIR ? 2 : 3) + 1)>();
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        } else {
          load_it(0, afr[0], afr[1], bfr[0]);
#pragma unroll
          for (int it = 0; it < 8; ++it) {
            if (it + 1 < 8) load_it(it + 1, afr[(it + 1) & 1], afr[it & 1], bfr[(it + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            if (more && it < WR::NPW) {
              issued += issue_piece_at(it, d + 3, xs_new, true, d + 2, ys_new, true);
              if (it == WR::NPW - 1) mark_new = issued;
            }
            mfma_it(it);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      } else {
        // the row's 32 voxels are ONE k-step: per (row, tap) unit 2 x operands (ci halves) and 4 MFMAs against the row's 2 dy
        // operands (co halves).  28 units per slice; the x operands of unit u + 1 and the dy operands of the next row are read
        // while the MFMAs of unit u issue (a whole row of operands in flight would not fit the registers)
        bf16x8_t fb[2][2], bb[2][2];
        auto load_x = [&](int u, bf16x8_t(&a)[2]) {
          const int oh = u / 7, i = u % 7;
          const int so = (tap_kd[i] == 0 ? slice_off[0] : (tap_kd[i] == 1 ? slice_off[1] : slice_off[2])) + tap_row[i];
          // (laundered: otherwise the 28 addresses of the 7 taps are hoisted out of the unit loop and spill)
          int ly0 = lane_y[0], ly1 = lane_y[1];
          asm volatile("" : "+v"(ly0), "+v"(ly1));
          const int l0 = (ly0 + kw_off[i]) ^ (int)((flip1[0] & kw_m1[i]) | (flip2[0] & kw_m2[i]));
          const int l1 = (ly1 + kw_off[i]) ^ (int)((flip1[1] & kw_m1[i]) | (flip2[1] & kw_m2[i]));
#pragma unroll
          for (int h = 0; h < 2; ++h)
            a[h] = wr_operand2(sX + so + oh * WR::X_ROW_B + (l0 ^ (h * 32)), sX + so + oh * WR::X_ROW_B + (l1 ^ (h * 32)));
        };
        auto load_y = [&](int oh, bf16x8_t(&y2)[2]) {
#pragma unroll
          for (int h = 0; h < 2; ++h)
            y2[h] = wr_operand2(ys0 + oh * WR::Y_ROW_B + (lane_y[0] ^ (h * 32)), ys0 + oh * WR::Y_ROW_B + (lane_y[1] ^ (h * 32)));
        };
        load_y(0, bb[0]);
        load_x(0, fb[0]);
#pragma unroll
        for (int u = 0; u < 28; ++u) {
          const int oh = u / 7, i = u % 7;
          if (u + 1 < 28) load_x(u + 1, fb[(u + 1) & 1]);
          if (i == 3 && oh + 1 < 4) load_y(oh + 1, bb[(oh + 1) & 1]);
          __builtin_amdgcn_sched_barrier(0);
          if (more && i == 0) {      // the slice's DMA pieces: two after the first row's first unit, then one per row
#pragma unroll
            for (int pi = 0; pi < WR::NPW; ++pi)
              if ((pi == 0 ? 0 : pi - 1) == oh) {
                issued += issue_piece_at(pi, d + 3, xs_new, true, d + 2, ys_new, true);
                if (pi == WR::NPW - 1) mark_new = issued;
              }
          }
#pragma unroll
          for (int ca = 0; ca < 2; ++ca)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
              f32x4_t c = {accf[i][(ca * 2 + cb) * 4 + 0], accf[i][(ca * 2 + cb) * 4 + 1], accf[i][(ca * 2 + cb) * 4 + 2],
                           accf[i][(ca * 2 + cb) * 4 + 3]};
              c = wr_mfma16<T16>(fb[u & 1][ca], bb[oh & 1][cb], c);
#pragma unroll
              for (int r = 0; r < 4; ++r) accf[i][(ca * 2 + cb) * 4 + r] = c[r];
            }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      mark_cur = mark_nxt;
      mark_nxt = mark_new;
      xr0 = xr0 + 1 == WR::NXS ? 0 : xr0 + 1;
      yr0 = yr0 + 1 == WR::NYS ? 0 : yr0 + 1;
    }
  }

  if (CLK && lane == 0) {
    float *o = slabs + (int64_t)gridDim.y * gridDim.x * (27 * 1024) + ((int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * WR::NW + wave) * 4;
    o[0] = (float)(__builtin_amdgcn_s_memtime() - t_begin);
    o[1] = (float)(__builtin_amdgcn_s_memrealtime() - rt_begin);
    o[2] = (float)t_wait;
  }
  // the two row halves of a tap: waves with rh = 1 hand their accumulators over through LDS, which overlays the rings.  A
  // workgroup whose LAST job has a single slice (DR == 1 or a one-slice tail segment) has only waited for the first three of
  // its prologue's four DMA groups: drain them all before the overlay is written (free for every other job: already drained)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  lds_barrier();
  float *xch = reinterpret_cast<float *>(smem);
  if (rh == 1) {
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
      for (int q = 0; q < 16; ++q) xch[((wq * 7 + i) * 16 + q) * 64 + lane] = accf[i][q];
  }
  lds_barrier();
  if (rh == 0) {
    // partial slab [27][32 ci][32 co].  C/D map of the 32x32 MFMA: col = lane & 31 (co), row = (q & 3) + 8 (q >> 2) + 4 (lane >> 5)
    // (ci); of the 16x16 one [ci half ca][co half cb]: col = lane & 15, row = 4 (lane >> 4) + r
    float *slab = slabs + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (27 * 1024);
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      if constexpr (PAIR) {
        // rows 0..15 of the accumulator: ci of tap 2 p, rows 16..31: ci of tap 2 p + 1 (row = (q & 3) + 8 (q >> 2) + 4 (lane >> 5):
        // q < 8 is the first tap, q >= 8 the second)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int row = (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
          const int tap = REUSE ? wr_reuse_tap(wq, i, true, row >> 4) : 2 * (wq + 4 * i) + (row >> 4), co = lane & 31;
          if (REUSE ? wr_reuse_stored(wq, i, true, row >> 4) : tap < 27) slab[(tap * 32 + (row & 15)) * 32 + co] = accf[i][q] + xch[((wq * 7 + i) * 16 + q) * 64 + lane];
        }
      } else {
        const int tap = REUSE ? wr_reuse_tap(wq, i, false, 0) : wq + 4 * i;
        if (REUSE ? wr_reuse_stored(wq, i, false, 0) : tap < 27) {
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const int ci = MF16 ? (q >> 3) * 16 + 4 * (lane >> 4) + (q & 3) : (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
            const int co = MF16 ? ((q >> 2) & 1) * 16 + (lane & 15) : (lane & 31);
            slab[(tap * 32 + ci) * 32 + co] = accf[i][q] + xch[((wq * 7 + i) * 16 + q) * 64 + lane];
          }
        }
      }
    }
  }
}

}  // namespace

// Entry point used by wgrad_launch_classes (conv_wgrad.hip): plain stride-1 launches with 16-bit storage.  Returns the number
// of slabs per channel-block pair it wrote (> 0), or 0 if the shape is not this kernel's (then nothing was launched).
int conv3_wgrad_ring_launch(const void *x, const View &xv, const void *dy, const View &yv, float *slabs, size_t ws_bytes, int B,
                            int Cin, int Cout, int is_f16, hipStream_t st, int *rc, long long xkh, bool dry) {
  *rc = DGTTA_OK;
  if (Cout % 32 != 0 || xv.D != yv.D || xv.H != yv.H || xv.W != yv.W) return 0;
  if (xkh && (xkh % 8 || xv.sw != 32 || Cin % 32)) return 0;      // channel blocks as planes: dense 32-channel rows
  const int cibs = cdiv(Cin, 32), cobs = Cout / 32;
  const long long xb = ((long long)(xv.D - 1) * xv.sd + (long long)(xv.H - 1) * xv.sh + (long long)(xv.W - 1) * xv.sw + 32) * 2;
  const long long yb = ((long long)(yv.D - 1) * yv.sd + (long long)(yv.H - 1) * yv.sh + (long long)(yv.W - 1) * yv.sw + 32) * 2;
  if (xb >= (1ll << 31) || yb >= (1ll << 31)) return 0;
  if (xv.sw % 8 || xv.sh % 8 || xv.sd % 8 || xv.sb % 8 || yv.sw % 8 || yv.sh % 8 || yv.sd % 8 || yv.sb % 8) return 0;
  static int ncu = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n > 0 ? n : 256;
  }();
  const int pairs = cibs * cobs;
  if (pairs > 65535) return 0;
  const int tW = cdiv(yv.W, 32), tH = cdiv(yv.H, WR::TH);
  const long long ncol = (long long)B * tW * tH;
  // workgroups per pair: a whole number of XCD groups (8) that fills the chip once over all pairs
  int G = (ncu / pairs) / 8 * 8;
  if (G < 8) G = 8;
  // D segments: enough jobs to give every workgroup the same number (a segment start costs about three slices of exposed latency)
  int nseg = 1;
  {
    double best = 1e300;
    for (int s = 1; s <= 64 && s <= yv.D; s *= 2) {
      const int dr = cdiv(yv.D, s), ns = cdiv(yv.D, dr);
      const double t = (double)cdiv64(ncol * ns, G) * (dr + 3.0);
      if (t < best * 0.97) best = t, nseg = ns;
    }
  }
  const int DR = cdiv(yv.D, nseg);
  nseg = cdiv(yv.D, DR);
  const long long njobs = ncol * nseg;
  if (njobs >= (1ll << 31)) return 0;
  if (njobs < G) G = (int)njobs;
  if (ws_bytes < (size_t)pairs * G * 27 * 1024 * sizeof(float)) return 0;
  // small problems stay with the many-small-workgroups kernels: a persistent sweep needs a few slices per job to amortise its prologue
  const char sw = dgtta_switches().wgrad_ring;
  if (ncol * yv.D < 4ll * ncu * 8 / pairs && sw != '1' && sw != '5' && sw != '6')
    return 0;      // (=1 / =5 / =6: forced, for the tests; 5 = forced, one tap per MFMA for Cin <= 16 and no operand reuse; 6 = forced, no operand reuse)
  if (dry) return G;      // (dgtta_conv3d_k3_blocked_supported: would this launch be taken?)
#define WR_LAUNCH(T16, MF, CK) WR_LAUNCH_P(T16, MF, CK, false, false)
#define WR_LAUNCH_P(T16, MF, CK, PR, RU)                                                                                          \
  do {                                                                                                                         \
    auto kern = conv3_wgrad_ring_kernel<T16, MF, CK, PR, RU>;                                                                      \
    static DynLdsOnce once;                                                                                                    \
    if (ensure_dyn_lds(once, reinterpret_cast<const void *>(kern), WR_LDS_TOTAL) != hipSuccess) {                              \
      dgtta_set_error("wgrad_ring: cannot raise the dynamic LDS limit to %d", WR_LDS_TOTAL);                                   \
      *rc = DGTTA_ERR_LAUNCH;                                                                                                  \
      return 0;                                                                                                                \
    }                                                                                                                          \
    hipLaunchKernelGGL(kern, dim3((unsigned)G, (unsigned)pairs), dim3(WR::NW * 64), WR_LDS_TOTAL, st, (const bf16_t *)x, xv,  \
                       (const bf16_t *)dy, yv, slabs, Cin, Cout, tW, tH, nseg, DR, cobs, (int)njobs, (unsigned)xb, (unsigned)yb, \
                       xkh);                                                                                                   \
  } while (0)
  // DGTTA_WGRAD_RING=4: the v_mfma_f32_16x16x32 form (conflict-free after the half swap above; measured within +-2 % of the
  // 32x32x16 form on three boxes, which is the default: 36 registers less and no swizzle)
  const bool mf16 = sw == '4';
  // operands shared between neighbouring rows (REUSE; DGTTA_WGRAD_RING=6: its predecessor)
  const bool reuse = !mf16 && sw != '5' && sw != '6';
  bool lab = false;
  // Cin <= 16 (the first layer): two taps per MFMA (DGTTA_WGRAD_RING=5: the one-tap form, its predecessor)
  if (Cin <= 16 && !mf16 && sw != '5') {
    lab = true;
    if (is_f16) {
      if (reuse) WR_LAUNCH_P(f16_t, false, false, true, true);
      else WR_LAUNCH_P(f16_t, false, false, true, false);
    } else {
      if (reuse) WR_LAUNCH_P(bf16_t, false, false, true, true);
      else WR_LAUNCH_P(bf16_t, false, false, true, false);
    }
  }
#ifdef DGTTA_DIAG
  if (!lab && is_f16 && DG_LAB(wgrad_ring_lab) == '6') {      // DGTTA_WGRAD_RING_CLK=6: cycle stamps BEHIND the slabs; only
    lab = true;                                        // profiles/tools/wring_clock.py, which allocates that area, asks for it
    if (reuse) WR_LAUNCH_P(f16_t, false, true, false, true);
    else WR_LAUNCH(f16_t, false, true);
  }
#endif
  if (lab) {
  } else if (is_f16) {
    if (mf16) WR_LAUNCH(f16_t, true, false);
    else if (reuse) WR_LAUNCH_P(f16_t, false, false, false, true);
    else WR_LAUNCH(f16_t, false, false);
  } else {
    if (mf16) WR_LAUNCH(bf16_t, true, false);
    else if (reuse) WR_LAUNCH_P(bf16_t, false, false, false, true);
    else WR_LAUNCH(bf16_t, false, false);
  }
#undef WR_LAUNCH
#undef WR_LAUNCH_P
  if (hipGetLastError() != hipSuccess) {
    dgtta_set_error("conv3_wgrad_ring_kernel: launch failed");
    *rc = DGTTA_ERR_LAUNCH;
    return 0;
  }
  return G;
}
