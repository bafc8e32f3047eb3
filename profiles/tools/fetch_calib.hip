// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access patterns of this engine's kernels (MI355X_MICROARCH.md, HBM: "other
// access widths are uncalibrated: calibrate on a known byte count in your own access pattern").  Each kernel reads a 2 GiB buffer
// (far beyond the 256 MiB Infinity Cache) exactly once in the way its name says; FETCH_SIZE (KiB) / launches against 2 GiB tells the
// factor.  usage: hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip; rocprofv3 --pmc FETCH_SIZE --kernel-trace -- ./fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>

// (a) 16 B per lane on consecutive addresses: a wave reads 1 KiB contiguous
__global__ void read_coalesced16(const uint4 *p, size_t n, unsigned *sink) {
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc ^= p[i].x;
  if (acc == 0x12345678u) *sink = acc;
}
// (b) conv_s2_regs_kernel's pattern: lane r of a 32-lane half reads 16 B of voxel 2 r (64-byte voxel rows: lines 128 B apart), the
//     other half the next 16 B; four instructions (kw = 1, 2 x channel chunks 0, 1) cover the 128-byte line of the lane's voxel pair
__global__ void read_strided16(const uint4 *p, size_t nlines, unsigned *sink) {
  unsigned acc = 0;
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwave = ((size_t)gridDim.x * blockDim.x) >> 6;
  for (size_t l0 = wave * 32; l0 < nlines; l0 += nwave * 32) {
    const uint4 *line = p + (l0 + r) * 8;      // 8 x 16 B = 128 B
#pragma unroll
    for (int kw = 0; kw < 2; ++kw)
#pragma unroll
      for (int c = 0; c < 2; ++c) acc ^= line[kw * 4 + c * 2 + h].x;
  }
  if (acc == 0x12345678u) *sink = acc;
}
// (c) the same lines, but only ONE 16-byte piece of each 128-byte line is ever read
__global__ void read_strided16_sparse(const uint4 *p, size_t nlines, unsigned *sink) {
  unsigned acc = 0;
  for (size_t l = (size_t)blockIdx.x * blockDim.x + threadIdx.x; l < nlines; l += (size_t)gridDim.x * blockDim.x) acc ^= p[l * 8].x;
  if (acc == 0x12345678u) *sink = acc;
}

int main() {
  const size_t bytes = (size_t)2 << 30;
  uint4 *buf;
  unsigned *sink;
  if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) return 1;
  hipMemset(buf, 1, bytes);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(read_coalesced16, dim3(8192), dim3(256), 0, 0, buf, bytes / 16, sink);
    hipLaunchKernelGGL(read_strided16, dim3(8192), dim3(256), 0, 0, buf, bytes / 128, sink);
    hipLaunchKernelGGL(read_strided16_sparse, dim3(8192), dim3(256), 0, 0, buf, bytes / 128, sink);
  }
  hipDeviceSynchronize();
  printf("done: every launch read %zu bytes of lines\n", bytes);
  return 0;
}
