// pmc_calib.hip -- known-byte-count kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on this chip
// (MI355X_MICROARCH.md, HBM section: "Other access widths and WRITE_SIZE are uncalibrated: calibrate on a known byte
// count in your own access pattern").  The encoder's coefficient planes are read and written one int16 per lane
// (64 lanes = one 128-byte line per wave access, 63 planes per block, plane stride apart); its pixel rows and sample
// planes 16 bytes per lane.  Each kernel below moves exactly BYTES bytes in one of those patterns over buffers far larger
// than the 256 MiB Infinity Cache:
//   rd16B / wr16B    : 16 bytes per lane, consecutive
//   rd2B  / wr2B     : 2 bytes per lane, consecutive lanes, one "plane" after the other (63 planes of n elements)
// build: hipcc --offload-arch=gfx950 -O2 -o pmc_calib pmc_calib.hip ; run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void __launch_bounds__(256) rd16B(const uint4 *__restrict__ a, unsigned *__restrict__ sink, size_t n)
{
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint4 v = a[i];
  if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345u) sink[0] = 1;
}
__global__ void __launch_bounds__(256) wr16B(uint4 *__restrict__ a, size_t n)
{
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) a[i] = make_uint4((unsigned)i, 1u, 2u, 3u);
}
// one lane per block of 63 coefficients in plane-major layout: plane k at a + k * stride
__global__ void __launch_bounds__(64) rd2B(const int16_t *__restrict__ a, unsigned *__restrict__ sink, size_t stride)
{
  const size_t b = (size_t)blockIdx.x * 64 + threadIdx.x;
  int acc = 0;
#pragma unroll
  for (int k = 1; k < 64; k++) acc ^= a[(size_t)k * stride + b];
  if (acc == (int)stride) sink[0] = 1;
}
__global__ void __launch_bounds__(64) wr2B(int16_t *__restrict__ a, size_t stride)
{
  const size_t b = (size_t)blockIdx.x * 64 + threadIdx.x;
#pragma unroll
  for (int k = 1; k < 64; k++) a[(size_t)k * stride + b] = (int16_t)(k + (int)b);
}

// the encoder's actual layout: [image][64 planes][kstride = 32448 int16] (4K luma: 32400 blocks rounded up to 64), one
// wave per 64 blocks, 63 plane accesses per lane 64 896 bytes apart
#define KSTRIDE 32448
__global__ void __launch_bounds__(64) rd2B_planes(const int16_t *__restrict__ a, unsigned *__restrict__ sink, int magic)
{
  const int16_t *p = a + ((size_t)blockIdx.y * 64) * KSTRIDE + (size_t)blockIdx.x * 64 + threadIdx.x;
  int acc = 0;
#pragma unroll
  for (int k = 1; k < 64; k++) acc ^= p[(size_t)k * KSTRIDE];
  if (acc == magic) sink[0] = 1;
}
__global__ void __launch_bounds__(64) wr2B_planes(int16_t *__restrict__ a)
{
  int16_t *p = a + ((size_t)blockIdx.y * 64) * KSTRIDE + (size_t)blockIdx.x * 64 + threadIdx.x;
#pragma unroll
  for (int k = 1; k < 64; k++) p[(size_t)k * KSTRIDE] = (int16_t)(k + threadIdx.x);
}
// the same stores with arithmetic between them (the FDCT kernel computes one coefficient, stores it, computes the next)
__global__ void __launch_bounds__(64) wr2B_planes_spaced(int16_t *__restrict__ a, int m)
{
  int16_t *p = a + ((size_t)blockIdx.y * 64) * KSTRIDE + (size_t)blockIdx.x * 64 + threadIdx.x;
  int v = threadIdx.x;
#pragma unroll
  for (int k = 1; k < 64; k++) {
    for (int j = 0; j < 40; j++) v = v * m + k;
    p[(size_t)k * KSTRIDE] = (int16_t)v;
  }
}

int main()
{
  const size_t BYTES = (size_t)1 << 30;         // 1 GiB per pattern
  void *buf; unsigned *sink;
  hipMalloc(&buf, BYTES + (1 << 20)); hipMalloc(&sink, 64);
  hipMemset(buf, 1, BYTES);
  hipDeviceSynchronize();
  const size_t n16 = BYTES / 16;
  const size_t stride = BYTES / 2 / 64;         // 64 planes of `stride` int16 (plane 0 is not touched, like the DC plane)
  for (int rep = 0; rep < 3; rep++) {
    hipLaunchKernelGGL(rd16B, dim3((unsigned)(n16 / 256)), dim3(256), 0, 0, (const uint4 *)buf, sink, n16);
    hipLaunchKernelGGL(wr16B, dim3((unsigned)(n16 / 256)), dim3(256), 0, 0, (uint4 *)buf, n16);
    hipLaunchKernelGGL(rd2B, dim3((unsigned)(stride / 64)), dim3(64), 0, 0, (const int16_t *)buf, sink, stride);
    hipLaunchKernelGGL(wr2B, dim3((unsigned)(stride / 64)), dim3(64), 0, 0, (int16_t *)buf, stride);
    hipLaunchKernelGGL(rd2B_planes, dim3(KSTRIDE / 64, 256), dim3(64), 0, 0, (const int16_t *)buf, sink, 12345 + rep);
    hipLaunchKernelGGL(wr2B_planes, dim3(KSTRIDE / 64, 256), dim3(64), 0, 0, (int16_t *)buf);
    hipLaunchKernelGGL(wr2B_planes_spaced, dim3(KSTRIDE / 64, 256), dim3(64), 0, 0, (int16_t *)buf, 3 + rep);
  }
  hipDeviceSynchronize();
  printf("{\"rd16B_bytes\": %zu, \"wr16B_bytes\": %zu, \"rd2B_bytes\": %zu, \"wr2B_bytes\": %zu, \"planes_bytes\": %zu}\n", BYTES, BYTES, stride * 63 * 2, stride * 63 * 2,
         (size_t)256 * 63 * KSTRIDE * 2);
  return 0;
}
