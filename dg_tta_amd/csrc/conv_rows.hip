// Row-reuse 3x3x3 convolution kernel (bf16, stride 1) -- see the block comment below.
#include "conv_common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

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
  static constexpr size_t RED_BYTES = (size_t)NW * 32 * 2 * sizeof(float) + 32 * 2 * sizeof(float);     // + GST constants
  static constexpr size_t LDS_BYTES = 2 * A_BYTES + B_BYTES + RED_BYTES;
};

// VAR (round 3, bit mask; default 5 = 1 | 4, DGTTA_ROWS_VAR=0: the round-2 kernel): 1 = the next job's coordinates come from a mixed-radix
// counter (a few scalar adds) instead of ten integer divisions on the CU's one scalar unit, which all 8 waves queued on
// (stamps: 5.7 % of the kernel); 2 = the DMA pieces of the next chunk are issued in the first 5/8 of the MFMA loop, so
// that they have landed when the loop ends (measured: no gain, not in the default); 4 = the 27 weight fragments are read from LDS just in time inside the MFMA
// loop (taps (kd,kh) in order of first use) instead of in a block before it, and the barrier that frees the weight
// buffer for the next chunk's DMA sits in the middle of the loop.  Stamps (128^3 32->32): each feature shortens its own
// segment (job decode 13.7 k -> 8.6 k cycles per wave, weight reload + barriers 31.9 k -> 17.1 k of 238 k) and the MFMA
// loop of the lock-stepped waves absorbs most of it; net 1.5-2.5 % per launch.  Also tried in round 3 and dropped: the A image
// filled as flat 1-KiB pieces crossing row boundaries (12 instead of 20 piece issues per wave and phase): 8 % SLOWER -
// the per-lane row select and address arithmetic cost more than the issue slots they save.
// GST (round 3): the launch is the DATA GRADIENT of a conv whose input was z = LeakyReLU(InstanceNorm(y_prev)): the
// epilogue also accumulates the statistics the InstanceNorm backward of that previous layer needs,
//   sum_v g'  and  sum_v g' * y_prev   with  g' = gz * lrelu'(A y_prev + B),  A = gamma * rstd,  B = beta - mean * A
// (sum g' xhat = rstd (sum g' y - mean sum g')), from the gz values it is about to store (rounded, as the apply pass reads
// them) and the y_prev tile read with the stores' own address pattern.  Replaces the reduction pass over y_prev and gz
// (chan_reduce_vec_kernel<., 1>) for that layer; same partial-sum layout and finalize order as the forward statistics.
struct GstArgs {
  const bf16_t *y;          // y_prev, same lattice as the output
  View v;
  const float *mr, *gamma, *beta;
  float slope;
};

template <int PD, int PH, int WD, int WH, int ABL = 0, typename T16 = bf16_t, int VAR = 0, bool GST = false>   // ABL: diagnostic ablation
__global__ __launch_bounds__(WD *WH * 64) void conv3_rows_kernel(const bf16_t *__restrict__ x, View xv,
                                                                 const bf16_t *__restrict__ w, Taps taps,
                                                                 const float *__restrict__ bias, bf16_t *__restrict__ y,
                                                                 View yv, int Cin, int Cout, int CinP, int tilesW,
                                                                 int tilesH, int tilesD, int nblkN, int njobs,
                                                                 double *__restrict__ stats, int ntaps_src, int order,
                                                                 GstArgs gst) {
  typedef RowsCfg<PD, PH, WD, WH> Cfg;
  constexpr int NW = Cfg::NW, IH = Cfg::IH;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char *sAb = smem;                                   // two A buffers
  unsigned char *sBb = smem + 2 * Cfg::A_BYTES;                // [27][2][32] x 16 B
  float *red = reinterpret_cast<float *>(smem + 2 * Cfg::A_BYTES + Cfg::B_BYTES);
  float *gcst = red + NW * 32 * 2;                             // GST: (A, B) of the run's 32 channels
  bool gst_newrun = true;

  // the wave id is made explicitly scalar: derived from threadIdx it is a 'divergent' VGPR value to the compiler, and
  // every per-row address of a DMA piece (64-bit multiplies and adds) was computed per lane - ~25 VALU per piece, 20
  // pieces per wave and phase, on a VALU shared with the partner wave: the MFMA loop was VALU-bound on address math
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int wd = wave / WH, wh = wave % WH;
  const int Di = xv.D, Hi = xv.H, Wi = xv.W, Do = yv.D, Ho = yv.H, Wo = yv.W;
  const int nk = CinP / 16;
  const int cin_lim = (Cin + 7) / 8 * 8;

  // Job order.  The workgroups of one XCD (blockIdx % 8; one per CU) share an L2, and a job's input tile overlaps its
  // neighbours' by the halo.  A workgroup that walks a contiguous range meets each neighbour one job later, by which time
  // the XCD has streamed ~8 MB through its 4 MB L2: the shared D-halo planes came from HBM again (PMC: 1.47x the input).
  // So the XCD's share of the jobs is dealt out ROUND ROBIN: at any time its workgroups hold ~32 consecutive jobs, and
  // the job index is decoded so that 32 consecutive jobs form a compact block of tiles (all channel blocks of a tile
  // first - they read the same input -, then 4-8 tiles along D, then 2-4 along H): the halos are shared while hot.
  const int G = gridDim.x;
  int jbeg, jstep, jend;
  if (order == 0) {           // DGTTA_ROWS_ORDER=0: the round-2a order (a contiguous range per workgroup, D fastest)
    const int lw = (G % 8 == 0) ? (int)(blockIdx.x % 8) * (G / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
    jbeg = (int)(((long long)njobs * lw) / G), jend = (int)(((long long)njobs * (lw + 1)) / G), jstep = 1;
  } else if (G % 8 == 0) {
    const int xcd = (int)(blockIdx.x % 8), nl = G / 8;
    jbeg = (int)(((long long)njobs * xcd) / 8) + (int)(blockIdx.x / 8);
    jend = (int)(((long long)njobs * (xcd + 1)) / 8);
    jstep = nl;
  } else {
    jbeg = (int)blockIdx.x, jend = njobs, jstep = G;
  }
  const int nmine = jbeg < jend ? (jend - jbeg + jstep - 1) / jstep : 0;
  const int nph = nmine * nk;
  if (nph == 0) return;
  // block shape (NBL x TDL x THL jobs, ~32): powers of two that divide the tile counts
  const int NBL = (order != 0 && (nblkN == 2 || nblkN == 4)) ? nblkN : 1;
  int TDL = NBL == 1 ? 8 : 4, THL = NBL == 4 ? 2 : 4;
  while (tilesD % TDL) TDL >>= 1;
  while (tilesH % THL) THL >>= 1;
  if (order == 0) TDL = tilesD, THL = 1;      // (td, nb, th, tw, b)
  const int nbH = nblkN / NBL, tdH = tilesD / TDL, thH = tilesH / THL;
  struct Job {
    int b, n0, od0, oh0, ow0, tile;
  };
  auto decode = [&](int j) {
    Job q;
    const int nb_lo = j % NBL;
    j /= NBL;
    const int td_lo = j % TDL;
    j /= TDL;
    const int th_lo = j % THL;
    j /= THL;
    const int td = (j % tdH) * TDL + td_lo;
    j /= tdH;
    int th, nb;
    if (order == 0) {
      nb = j % nblkN;
      j /= nblkN;
      th = j % tilesH;
      j /= tilesH;
    } else {
      th = (j % thH) * THL + th_lo;
      j /= thH;
      nb = (j % nbH) * NBL + nb_lo;
      j /= nbH;
    }
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
        if (ABL == 1 || ABL == 4 || (ABL == 9 && dz >= 4)) return;      // 9: timing model of a D-ring (4 new planes of 6)
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

  // mixed-radix job counter (VAR & 1): digits (nb_lo, td_lo, th_lo, td_hi, th_hi, nb_hi, tw | b) of the current job index
  // and of the stride between this workgroup's jobs; advancing = digit-wise add with carry, all on wave-uniform values
  int dg[8], sdg[8];
  const int rad[7] = {NBL, TDL, THL, tdH, thH, nbH, tilesW};
  {
    int a = jbeg, c = jstep;
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      dg[k] = a % rad[k];
      a /= rad[k];
      sdg[k] = c % rad[k];
      c /= rad[k];
    }
    dg[7] = a;
    sdg[7] = c;
  }
  auto advance = [&]() {
    int carry = 0;
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      const int v = dg[k] + sdg[k] + carry;
      carry = v >= rad[k] ? 1 : 0;
      dg[k] = carry ? v - rad[k] : v;
    }
    dg[7] += sdg[7] + carry;
    Job q;
    const int nb = dg[5] * NBL + dg[0], td = dg[3] * TDL + dg[1], th = dg[4] * THL + dg[2], tw = dg[6];
    q.b = dg[7];
    q.n0 = nb * 32;
    q.od0 = td * Cfg::TD;
    q.oh0 = th * Cfg::TH;
    q.ow0 = tw * 32;
    q.tile = (tw * tilesH + th) * tilesD + td;
    return q;
  };
  Job cur = decode(jbeg);
  float bv = 0.f;
  float st1 = 0.f, st2 = 0.f;       // InstanceNorm partial sums of this lane's channel over the current run of jobs
#pragma unroll
  for (int i = 0; i < NPIECE; ++i) issue_piece(cur, 0, 0, i);
  int kc = 0, jn = jbeg;
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
  unsigned long long t_begin = 0, rt_begin = 0;      // ABL 6: in-kernel clock = d(s_memtime) / d(s_memrealtime) x 100 MHz
  if (ABL == 6) {
    tprev = t_begin = __builtin_amdgcn_s_memtime();
    rt_begin = __builtin_amdgcn_s_memrealtime();
  }
  for (int p = 0; p < nph; ++p) {
    dma_wait_all();
    stamp(0);                 // waiting for the DMA
    lds_barrier();            // chunk p has landed; every wave is done with phase p-1
    uint4 breg[27];
    const uint4 *sB = reinterpret_cast<const uint4 *>(sBb);
    auto load_b = [&](int kd, int kh) {       // the three kw fragments of tap row (kd, kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) breg[(kd * 3 + kh) * 3 + kw] = sB[((kd * 3 + kh) * 3 + kw) * 64 + lane];
    };
    if (VAR & 4) {
      static_assert(!(VAR & 4) || (PD == 2 && PH == 2), "the just-in-time weight schedule is written for 2 x 2 patches");
      load_b(0, 0);
    } else {
#pragma unroll
      for (int t = 0; t < 27; ++t) breg[t] = sB[t * 64 + lane];
    }
    if (kc == 0) {            // bias of this job, fetched while no DMA is in flight (its wait would drain them)
      const int co = cur.n0 + r;
      bv = (bias && co < Cout) ? bias[co] : 0.f;
      asm volatile("" ::"v"(bv));
      if (GST && gst_newrun) {      // a new (sample, channel block) run: its constants, and zeroed per-wave accumulators
        if (tid < 32) {
          const int c = cur.n0 + tid;
          const float mu = gst.mr[((long long)cur.b * Cout + c) * 2], rs = gst.mr[((long long)cur.b * Cout + c) * 2 + 1];
          const float a = gst.gamma[c] * rs;
          gcst[tid * 2] = a;
          gcst[tid * 2 + 1] = gst.beta[c] - mu * a;
        }
        for (int i = tid; i < NW * 32 * 2; i += Cfg::NT) red[i] = 0.f;
        gst_newrun = false;         // (published by the barriers between here and the epilogue)
      }
    }
    if (!(VAR & 4)) lds_barrier();            // B buffer is free again (VAR & 4: in the middle of the MFMA loop)
    stamp(1);                 // barrier + B fragments + barrier
    // prefetch the next phase (next K-chunk of this job, or chunk 0 of the next job)
    Job nxt = cur;
    int kn = kc + 1;
    if (kn == nk) {
      kn = 0;
      if (p + 1 < nph) nxt = ((VAR & 1) && order != 0) ? advance() : decode(jn + jstep);
    }
    const bool more = p + 1 < nph;
    stamp(2);

    if (kc == 0) {          // accumulators start at the bias of this lane's output channel (column re of every row)
#pragma unroll
      for (int i = 0; i < PD; ++i)
#pragma unroll
        for (int j = 0; j < PH; ++j)
#pragma unroll
          for (int q = 0; q < 16; ++q) acc[i][j][q] = bv;
    }
    {
      const uint4 *sA = reinterpret_cast<const uint4 *>(sAb + (size_t)(p & 1) * Cfg::A_BYTES);
#pragma unroll
      for (int dz = 0; dz < PD + 2; ++dz)
#pragma unroll
        for (int hy = 0; hy < PH + 2; ++hy) {
          if (VAR & 4) {
            // weight fragments just in time: tap row (kd, kh) is first used at input row (dz, hy) = (kd, kh); row step rs
            // reads the fragments whose first use is 1-3 steps ahead.  After step 7 every wave holds all 27: the barrier
            // at step 8 frees the weight buffer, the DMA pieces of the next weight chunk are issued after it.
            const int rs = dz * (PH + 2) + hy;
            constexpr int sched[8][2] = {{0, 1}, {0, 2}, {1, 0}, {1, 1}, {1, 2}, {2, 0}, {2, 1}, {2, 2}};
            if (rs < 8) {
              __builtin_amdgcn_sched_barrier(0);
              load_b(sched[rs][0], sched[rs][1]);
            } else if (rs == 8) {
              __builtin_amdgcn_sched_barrier(0);
              lds_barrier();
            }
          }
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const uint4 af = sA[abase[kw] + (dz * IH + hy) * Cfg::RB];
            {   // spread this wave's DMA pieces of the next chunk over the A-read steps (VAR & 2: over the first 5/8 of
                // them; the weight pieces - the last ones - then still come after the barrier of step 24)
              constexpr int NSTEP = (PD + 2) * (PH + 2) * 3;
              constexpr int NISSUE = (VAR & 2) ? NSTEP * 5 / 8 : NSTEP;
              static_assert(!(VAR & 4) || (2 * NPR * NISSUE) / NPIECE + 1 > 24, "weight pieces must follow the mid-loop barrier");
              const int step = (dz * (PH + 2) + hy) * 3 + kw;
#pragma unroll
              for (int i = 0; i < NPIECE; ++i)
                if (step == (i * NISSUE) / NPIECE + 1 && more) issue_piece(nxt, kn, (p + 1) & 1, i);
            }
#pragma unroll
            for (int kd = 0; kd < 3; ++kd)
#pragma unroll
              for (int kh = 0; kh < 3; ++kh) {
                const int od = dz - kd, oh = hy - kh;
                if (od >= 0 && od < PD && oh >= 0 && oh < PH) {
                  if (ABL == 3) acc[od][oh][0] += __uint_as_float(af.x ^ breg[(kd * 3 + kh) * 3 + kw].x);
                  else mfma_step<T16>(af, breg[(kd * 3 + kh) * 3 + kw], acc[od][oh]);
                }
              }
          }
        }
    }

    stamp(3);                 // MFMA loop
    if (kc == nk - 1) {
      // ---- epilogue of job `cur`: convert, transpose through LDS (this phase's A buffer, one 8-KiB slab per wave) so
      //      that a lane stores 16 bytes = 8 channels of a voxel; per-channel sum / sum of squares of the unrounded
      //      values for the following InstanceNorm, kept in registers over the jobs of one (sample, channel block) run.
      lds_barrier();          // every wave has finished reading this phase's A buffer
      stamp(5);               // (diagnostic) barrier skew
      // lane-derived epilogue indices are rebuilt from a laundered lane id: otherwise the compiler hoists them out of
      // the phase loop, keeps them live across the MFMA phase and spills (a scratch reload's wait drains the DMA)
      int le = lane;
      asm volatile("" : "+v"(le));
      const int re = le & 31, he = le >> 5;
      const int co = cur.n0 + re;
      unsigned char *slabb = sAb + (size_t)(p & 1) * Cfg::A_BYTES + wave * (PD * PH * 2048);
      const bool full = cur.od0 + Cfg::TD <= Do && cur.oh0 + Cfg::TH <= Ho && cur.ow0 + 32 <= Wo && cur.n0 + 32 <= Cout;
      if (full && ABL != 7) {
        // Fast path (whole tile inside the volume).  The MFMA leaves a lane with ONE channel (re) and 16 voxels, 4
        // consecutive ones per q group: the slab is CHANNEL-major [32 ch][32 voxels] bf16 (64-byte rows), written 8 bytes
        // (4 voxels) at a time - 4 ds_write_b64 + 8 packed converts per accumulator instead of 16 2-byte writes and 16
        // converts - and read back through ds_read_b64_tr_b16 (hardware transpose: a lane receives 4 channels of one
        // voxel).  8-byte slot s of row ch sits at s ^ ((ch >> 1) & 7): conflict free for the writes (16-lane groups
        // over 128 B) and for the transposed reads (32-lane halves over 256 B).
        const int swz = (re >> 1) & 7;
        // transposed read: 16-lane group cq = le >> 4 takes channels 8 cq .. 8 cq + 7 (two 4-row blocks) of 16 voxels;
        // lane 4 q + p of a group supplies the address of block row q, voxels 4 p .. 4 p + 3
        const int cq = le >> 4, li = le & 15, rq = li >> 2, rp = li & 3;
        f32x2_t s1v = {0.f, 0.f}, s2v = {0.f, 0.f};       // packed fp32 math: two voxels per instruction
        // GST: this lane's y_prev values (8 channels of each of its 8 output voxels), fetched with the stores' pattern
        uint4 gy[GST ? PD * PH * 2 : 1];
        f32x2_t gs1[4], gs2[4], gA[4], gB[4];       // channel pairs: packed fp32 math (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32)
        if (GST) {
#pragma unroll
          for (int k = 0; k < PD * PH; ++k) {
            const int od = cur.od0 + wd * PD + k / PH, oh = cur.oh0 + wh * PH + k % PH;
            const bf16_t *yrow = gst.y + (long long)cur.b * gst.v.sb + od * gst.v.sd + oh * gst.v.sh + cur.n0 + cq * 8;
#pragma unroll
            for (int vb = 0; vb < 2; ++vb)
              gy[k * 2 + vb] = *reinterpret_cast<const uint4 *>(yrow + (long long)(cur.ow0 + 16 * vb + li) * gst.v.sw);
          }
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const float4 c4 = *reinterpret_cast<const float4 *>(gcst + (cq * 8 + 2 * t) * 2);      // (A, B) of two channels
            gs1[t] = gs2[t] = f32x2_t{0.f, 0.f};
            gA[t] = f32x2_t{c4.x, c4.z};
            gB[t] = f32x2_t{c4.y, c4.w};
          }
        }
        auto part1 = [&](int i, int j) {
          unsigned char *sl = slabb + (i * PH + j) * 2048 + re * 64;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x2_t a = {acc[i][j][4 * g], acc[i][j][4 * g + 1]}, c = {acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
            s1v += a;
            s1v += c;
            s2v = __builtin_elementwise_fma(a, a, s2v);
            s2v = __builtin_elementwise_fma(c, c, s2v);
            uint2 pk;
            pk.x = pack2_16<T16>(a[0], a[1]);
            pk.y = pack2_16<T16>(c[0], c[1]);
            *reinterpret_cast<uint2 *>(sl + (((2 * g + he) ^ swz) << 3)) = pk;
          }
        };
        auto part2 = [&](int i, int j) {
          const int od = cur.od0 + wd * PD + i, oh = cur.oh0 + wh * PH + j;
          bf16_t *orow = y + (long long)cur.b * yv.sb + od * yv.sd + oh * yv.sh + cur.n0 + cq * 8;
          const unsigned char *sl = slabb + (i * PH + j) * 2048;
#pragma unroll
          for (int vb = 0; vb < 2; ++vb) {
            const int c0 = 8 * cq + rq, c1 = c0 + 4, sidx = 4 * vb + rp;
            const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (lds_s16x4_t *)(sl + c0 * 64 + ((sidx ^ ((c0 >> 1) & 7)) << 3)));
            const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (lds_s16x4_t *)(sl + c1 * 64 + ((sidx ^ ((c1 >> 1) & 7)) << 3)));
            uint4 val;
            val.x = (unsigned)(unsigned short)lo[0] | ((unsigned)(unsigned short)lo[1] << 16);
            val.y = (unsigned)(unsigned short)lo[2] | ((unsigned)(unsigned short)lo[3] << 16);
            val.z = (unsigned)(unsigned short)hi[0] | ((unsigned)(unsigned short)hi[1] << 16);
            val.w = (unsigned)(unsigned short)hi[2] | ((unsigned)(unsigned short)hi[3] << 16);
            if (GST) {
              const uint4 yq = gy[(i * PH + j) * 2 + vb];
              const unsigned gw[4] = {val.x, val.y, val.z, val.w}, yw[4] = {yq.x, yq.y, yq.z, yq.w};
#pragma unroll
              for (int t = 0; t < 4; ++t) {
                float g0, g1, y0, y1;
                unpack2_16<T16>(gw[t], g0, g1);
                unpack2_16<T16>(yw[t], y0, y1);
                const f32x2_t g2 = {g0, g1}, y2 = {y0, y1};
                const f32x2_t a2 = __builtin_elementwise_fma(gA[t], y2, gB[t]);
                const f32x2_t gsl = g2 * gst.slope;
                const f32x2_t q2 = {a2[0] > 0.f ? g2[0] : gsl[0], a2[1] > 0.f ? g2[1] : gsl[1]};
                gs1[t] += q2;
                gs2[t] = __builtin_elementwise_fma(q2, y2, gs2[t]);
              }
            }
            const int ow = cur.ow0 + 16 * vb + li;
            // NON-TEMPORAL store: the output streams through L2 instead of displacing the input lines, whose second
            // 16-channel half is fetched one phase later.  PMC, 128^3 32->32, one sample: FETCH 353 -> 197 MB (134 MB
            // of input; the rest is the D-halo of consecutive jobs), WRITE unchanged (DGTTA_ROWS_ABL=8: plain stores)
            if (ABL == 8) *reinterpret_cast<uint4 *>(orow + ow * yv.sw) = val;
            else if (ABL != 2) {
              typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
              const u32x4_t nv = {val.x, val.y, val.z, val.w};
              __builtin_nontemporal_store(nv, reinterpret_cast<u32x4_t *>(orow + ow * yv.sw));
            }
          }
        };
        // (a wave reads back only its own slab: LDS operations of one wave complete in order); the read-back and stores
        // of accumulator k are issued after the conversion of accumulator k+1, which hides the LDS round trip
#pragma unroll
        for (int k = 0; k <= PD * PH; ++k) {
          if (k < PD * PH) part1(k / PH, k % PH);
          if (k > 0) part2((k - 1) / PH, (k - 1) % PH);
        }
        st1 += s1v[0] + s1v[1];
        st2 += s2v[0] + s2v[1];
        if (GST) {
          // sum over the 16 lanes (voxels) that share this lane's channel group: quad swaps, half-row and row mirrors (DPP);
          // lane li == 0 adds the job's sums to this wave's accumulators (one writer per slot: fixed order)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float a = gs1[e >> 1][e & 1], c2 = gs2[e >> 1][e & 1];
            a += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0xB1, 0xf, 0xf, false));
            c2 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(c2), 0xB1, 0xf, 0xf, false));
            a += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x4E, 0xf, 0xf, false));
            c2 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(c2), 0x4E, 0xf, 0xf, false));
            a += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x141, 0xf, 0xf, false));
            c2 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(c2), 0x141, 0xf, 0xf, false));
            a += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x140, 0xf, 0xf, false));
            c2 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(c2), 0x140, 0xf, 0xf, false));
            if (li == 0) {
              float2 *slot = reinterpret_cast<float2 *>(red + (wave * 32 + cq * 8 + e) * 2);
              float2 cur2 = *slot;
              cur2.x += a;
              cur2.y += c2;
              *slot = cur2;
            }
          }
        }
        stamp(7);               // (diagnostic) slab reads + global stores
      } else {
        // ragged tile: voxel-major slab [32 voxels][32 ch] with 2-byte writes and per-element bounds
        bf16_t *slab = reinterpret_cast<bf16_t *>(slabb);
#pragma unroll
        for (int i = 0; i < PD; ++i)
#pragma unroll
          for (int j = 0; j < PH; ++j) {
            const int od = cur.od0 + wd * PD + i, oh = cur.oh0 + wh * PH + j;
            const bool row_ok = od < Do && oh < Ho;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
              const int m = (q & 3) + 8 * (q >> 2) + 4 * he;
              const float v = acc[i][j][q];
              slab[(i * PH + j) * 1024 + m * 32 + re] = f32_to_16<T16>(v);
              if (row_ok && co < Cout && cur.ow0 + m < Wo) {
                st1 += v;
                st2 += v * v;
              }
            }
          }
        stamp(6);
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
        stamp(7);
      }
      if (stats) {
        // the partial sums of a run of jobs over one (sample, channel block) are flushed once, into the slot of the run's
        // last tile; the other tiles of the run hold zeros (the finalize kernel adds all tile slots in a fixed order)
        const bool flush = !(p + 1 < nph) || nxt.b != cur.b || nxt.n0 != cur.n0;
        const int tiles_per_b = tilesW * tilesH * tilesD;
        double *pp = stats + 32 + (((int64_t)cur.b * tiles_per_b + cur.tile) * Cout + cur.n0 + tid) * 2;
        if (flush) {
          if (!GST) {
            const float a = st1 + __shfl_xor(st1, 32, 64), c2 = st2 + __shfl_xor(st2, 32, 64);
            if (he == 0) {
              red[(wave * 32 + re) * 2 + 0] = a;
              red[(wave * 32 + re) * 2 + 1] = c2;
            }
          } else {
            gst_newrun = true;      // (the per-wave accumulators already hold the run's sums)
          }
          st1 = st2 = 0.f;
          lds_barrier();
          if (tid < 32 && cur.n0 + tid < Cout) {
            double s = 0.0, ss = 0.0;
#pragma unroll
            for (int wv = 0; wv < NW; ++wv) {
              s += (double)red[(wv * 32 + tid) * 2 + 0];
              ss += (double)red[(wv * 32 + tid) * 2 + 1];
            }
            pp[0] = s;
            pp[1] = ss;
          }
        } else if (tid < 32 && cur.n0 + tid < Cout) {
          pp[0] = 0.0;
          pp[1] = 0.0;
        }
        if (blockIdx.x == 0 && tid == 0 && p == nk - 1) reinterpret_cast<long long *>(stats)[0] = tiles_per_b;
      }
    }
    stamp(4);                 // epilogue
    kc = kn;
    if (kn == 0) {
      cur = nxt;
      jn += jstep;
    }
  }
  if (ABL == 6 && stats && lane == 0) {
    for (int k = 0; k < 8; ++k) stats[4096 + ((size_t)blockIdx.x * NW + wave) * 8 + k] = (double)tseg[k];
    stats[4096 + (size_t)gridDim.x * NW * 8 + ((size_t)blockIdx.x * NW + wave) * 2 + 0] =
        (double)(__builtin_amdgcn_s_memtime() - t_begin);
    stats[4096 + (size_t)gridDim.x * NW * 8 + ((size_t)blockIdx.x * NW + wave) * 2 + 1] =
        (double)(__builtin_amdgcn_s_memrealtime() - rt_begin);
  }
}

}  // namespace

namespace {

template <int PD, int PH, int WD, int WH, typename T16>
int launch_conv_rows(const void *x, const View &xv, const void *w, const Taps &taps, const float *bias, void *y,
                     const View &yv, int B, int Cin, int Cout, int CinP, int CoutP, double *stats, int ntaps_src,
                     hipStream_t st, RowsGstCtx *gctx) {
  typedef RowsCfg<PD, PH, WD, WH> Cfg;
  GstArgs ga{};
  // laboratory builds of the same kernel (libdgtta_hip_diag.so only).  DGTTA_ROWS_ABL: 1 no DMA, 3 no MFMA, 6 per-segment cycle
  // stamps, 7 the voxel-major (ragged-tile) epilogue for every tile, 8 plain (temporal) output stores; DGTTA_ROWS_VAR: '0' =
  // the round-2 kernel, default = feature mask 5
  const int abl = DG_LAB(rows_abl);
  const int var = DG_LAB(rows_var);
  // eligible: no forward statistics asked for, every tile whole, whole 32-channel blocks
  const bool gst_on = gctx && !stats && !bias && yv.D % Cfg::TD == 0 && yv.H % Cfg::TH == 0 && yv.W % 32 == 0 && Cout % 32 == 0 &&
                      CoutP == Cout && gctx->ldy % 8 == 0 && ((uintptr_t)gctx->y & 15) == 0 && abl < 0 &&
                      var != '7' && var != '0';      // (those builds have no GST form)
  if (gst_on) {
    ga.y = (const bf16_t *)gctx->y;
    ga.v = dense_view(B, yv.D, yv.H, yv.W, (int)gctx->ldy);
    ga.mr = gctx->mr;
    ga.gamma = gctx->gamma;
    ga.beta = gctx->beta;
    ga.slope = gctx->slope;
    stats = gctx->out;
    gctx->produced = 1;
  }
  auto kern = conv3_rows_kernel<PD, PH, WD, WH, 0, T16, 5>;
  static DynLdsOnce once[16];
  int slot = 0;
  if (gst_on) kern = conv3_rows_kernel<PD, PH, WD, WH, 0, T16, 5, true>, slot = 15;
#ifdef DGTTA_DIAG
  if (var == '0') kern = conv3_rows_kernel<PD, PH, WD, WH, 0, T16, 0>, slot = 6;
  if (std::is_same<T16, bf16_t>::value) {      // diagnostic builds exist for the bf16 instantiation only
    if (var == '7') kern = conv3_rows_kernel<PD, PH, WD, WH, 0, T16, 7>, slot = 7;
    if (abl == '1') kern = conv3_rows_kernel<PD, PH, WD, WH, 1, T16, 0>, slot = 1;
    if (abl == '3') kern = conv3_rows_kernel<PD, PH, WD, WH, 3, T16, 0>, slot = 2;
    if (abl == '6') {
      kern = conv3_rows_kernel<PD, PH, WD, WH, 6, T16, 5>, slot = 3;
      if (var == '0') kern = conv3_rows_kernel<PD, PH, WD, WH, 6, T16, 0>, slot = 9;
    }
    if (abl == '7') kern = conv3_rows_kernel<PD, PH, WD, WH, 7, T16, 0>, slot = 4;
    if (abl == '4') kern = conv3_rows_kernel<PD, PH, WD, WH, 4, T16, 0>, slot = 12;
    if (abl == '5') kern = conv3_rows_kernel<PD, PH, WD, WH, 5, T16, 0>, slot = 13;
    if (abl == '9') kern = conv3_rows_kernel<PD, PH, WD, WH, 9, T16, 0>, slot = 14;
    if (abl == '8') kern = conv3_rows_kernel<PD, PH, WD, WH, 8, T16, 0>, slot = 5;
  }
#endif
  DG_REQUIRE(ensure_dyn_lds(once[slot], reinterpret_cast<const void *>(kern), (int)Cfg::LDS_BYTES) == hipSuccess,
             DGTTA_ERR_LAUNCH, "conv3_rows: cannot raise the dynamic LDS limit to %zu", (size_t)Cfg::LDS_BYTES);
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
                     bias, (bf16_t *)y, yv, Cin, Cout, CinP, tW, tH, tD, nblkN, (int)njobs, stats, ntaps_src,
                     dgtta_switches().rows_order == '0' ? 0 : 1, ga);
  DG_CHECK_LAUNCH("conv3_rows_kernel");
  return DGTTA_OK;
}

}  // namespace

// entry point used by the dispatcher in conv_mfma.hip
int conv3_rows_launch(const void *x, const View &xv, const void *w, const Taps &taps, const float *bias, void *y, const View &yv,
                      int B, int Cin, int Cout, int CinP, int CoutP, double *stats, int ntaps_src, int is_f16, hipStream_t st,
                      RowsGstCtx *gst) {
  if (is_f16)
    return launch_conv_rows<2, 2, 2, 4, f16_t>(x, xv, w, taps, bias, y, yv, B, Cin, Cout, CinP, CoutP, stats, ntaps_src, st, gst);
  return launch_conv_rows<2, 2, 2, 4, bf16_t>(x, xv, w, taps, bias, y, yv, B, Cin, Cout, CinP, CoutP, stats, ntaps_src, st, gst);
}
