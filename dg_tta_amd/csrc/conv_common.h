// Shared declarations of the MFMA convolution translation units (conv_mfma.hip, conv_rows.hip, conv_wgrad.hip).
#pragma once
#include "common.h"
#include <stdlib.h>

// operand descriptors shared by the launchers of all three translation units (external linkage)
namespace dgconv {
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

}  // namespace dgconv
using namespace dgconv;


namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;                  // result of ds_read_b64_tr_b16
typedef __attribute__((address_space(3))) s16x4_t lds_s16x4_t;

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
template <>
struct Elem<f16_t> {
  static constexpr int EPV = 8;
};

template <typename T>
__device__ __forceinline__ void mfma_step(const uint4 &a, const uint4 &b, f32x16_t &acc);
template <>
__device__ __forceinline__ void mfma_step<bf16_t>(const uint4 &a, const uint4 &b, f32x16_t &acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0,
                                                0, 0);
}
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
template <>
__device__ __forceinline__ void mfma_step<f16_t>(const uint4 &a, const uint4 &b, f32x16_t &acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), acc, 0, 0, 0);
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

inline View dense_view(int B, int D, int H, int W, int ld) {
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

inline Taps identity_taps(int mirror) {
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
}  // namespace

