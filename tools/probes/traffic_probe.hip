// Probe: what FETCH_SIZE / WRITE_SIZE report for the access shapes the QP kernels use, on a buffer read exactly once.
// Each kernel reads BYTES of a buffer far larger than the Infinity Cache (4 GiB) exactly once with one of the shapes:
//   shape 0: 8 B per lane, contiguous across the wave (the ELL value streams: 512 B per wave instruction)
//   shape 1: 32 B per lane, contiguous (factor blocks, row layout: 2 KiB per wave instruction)
//   shape 2: 4 x 8 B per lane at a 128-B stride (factor blocks read transposed: load_blk<true>)
//   shape 3: 16 B per lane, contiguous (the guide's calibrated case: expect FETCH_SIZE = bytes / 2)
//   shape 4: 4 B per lane, contiguous (the ELL index streams)
// run under:  rocprofv3 --kernel-trace --pmc FETCH_SIZE -- ./traffic_probe   (then WRITE_SIZE in a second run)
// build: hipcc -O3 --offload-arch=gfx950 -o traffic_probe traffic_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2 __attribute__((ext_vector_type(2)));
typedef double d4 __attribute__((ext_vector_type(4)));
template <int SHAPE>
__global__ void __launch_bounds__(256) read_once(const char *buf, size_t bytes, double *sink) {
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nthr = (size_t)gridDim.x * blockDim.x;
  double acc = 0.0;
  if (SHAPE == 0) { const double *p = (const double *)buf; for (size_t i = tid; i < bytes / 8; i += nthr) acc += p[i]; }
  else if (SHAPE == 1) { const d4 *p = (const d4 *)buf; for (size_t i = tid; i < bytes / 32; i += nthr) { d4 v = p[i]; acc += v[0] + v[1] + v[2] + v[3]; } }
  else if (SHAPE == 2) {
    // a wave takes 2 KiB blocks; lane (r = lane >> 2, j = lane & 3) reads element (4 j + c, r), c = 0..3, of a row-major 16 x 16 block
    const double *p = (const double *)buf; const int lane = threadIdx.x & 63, r = lane >> 2, j = lane & 3;
    const size_t wave = tid >> 6, nwave = nthr >> 6;
    for (size_t b = wave; b < bytes / 2048; b += nwave) { const double *B = p + b * 256 + (4 * j) * 16 + r; acc += B[0] + B[16] + B[32] + B[48]; }
  }
  else if (SHAPE == 3) { const d2 *p = (const d2 *)buf; for (size_t i = tid; i < bytes / 16; i += nthr) { d2 v = p[i]; acc += v[0] + v[1]; } }
  else { const int *p = (const int *)buf; for (size_t i = tid; i < bytes / 4; i += nthr) acc += (double)p[i]; }
  if (acc == 1.2345e301) sink[0] = acc;
}
__global__ void __launch_bounds__(256) write_once(double *buf, size_t bytes) {
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nthr = (size_t)gridDim.x * blockDim.x;
  for (size_t i = tid; i < bytes / 8; i += nthr) buf[i] = (double)i;
}
int main() {
  const size_t bytes = 4ull << 30;
  char *buf; double *sink;
  if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(buf, 0, bytes);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char *names[] = {"8 B/lane contiguous", "32 B/lane contiguous", "4 x 8 B/lane stride 128 B (transposed block)", "16 B/lane contiguous", "4 B/lane contiguous"};
  for (int s = 0; s < 5; s++) {
    hipEventRecord(e0);
    switch (s) { case 0: read_once<0><<<4096, 256>>>(buf, bytes, sink); break; case 1: read_once<1><<<4096, 256>>>(buf, bytes, sink); break;
                 case 2: read_once<2><<<4096, 256>>>(buf, bytes, sink); break; case 3: read_once<3><<<4096, 256>>>(buf, bytes, sink); break;
                 default: read_once<4><<<4096, 256>>>(buf, bytes, sink); }
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("read_once<%d> %-46s %zu bytes in %.3f ms = %.2f TB/s\n", s, names[s], bytes, ms, bytes / ms / 1e9);
  }
  hipEventRecord(e0); write_once<<<4096, 256>>>((double *)buf, bytes); hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("write_once 8 B/lane contiguous %zu bytes in %.3f ms = %.2f TB/s\n", bytes, ms, bytes / ms / 1e9);
  return 0;
}
