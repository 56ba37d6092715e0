// Probe: issue / dependent latency of v_mfma_f64_16x16x4_f64 and v_fma_f64 on gfx950, one wave per CU.
// build: hipcc -O3 --offload-arch=gfx950 -o mfma_f64_probe mfma_f64_probe.hip ; prints cycles per instruction
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void probe(double *out, long long *cyc, int reps) {
  const int lane = threadIdx.x;
  double a = 1.0 + lane * 1e-3, b = 0.5 + lane * 1e-4;
  d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  double f0 = a, f1 = b, f2 = a + b, f3 = a - b;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; r++) {
    if (MODE == 0) {          // 16 dependent MFMAs
#pragma unroll
      for (int k = 0; k < 16; k++) c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    } else if (MODE == 1) {   // 16 MFMAs over 4 independent accumulators
#pragma unroll
      for (int k = 0; k < 4; k++) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
      }
    } else if (MODE == 2) {   // 16 dependent v_fma_f64
#pragma unroll
      for (int k = 0; k < 16; k++) f0 = __builtin_fma(f0, a, b);
    } else if (MODE == 3) {   // 16 v_fma_f64 over 4 chains
#pragma unroll
      for (int k = 0; k < 4; k++) { f0 = __builtin_fma(f0, a, b); f1 = __builtin_fma(f1, a, b); f2 = __builtin_fma(f2, a, b); f3 = __builtin_fma(f3, a, b); }
    } else if (MODE == 4) {   // chain where the MFMA result feeds the next MFMA's B operand (the solve's chain: acc -> v)
#pragma unroll
      for (int k = 0; k < 4; k++) {
        d4 c = {0, 0, 0, 0};
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, c0[0], c, 0, 0, 0); c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, c0[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, c0[2], c, 0, 0, 0); c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, c0[3], c, 0, 0, 0);
        c0 = c;
      }
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 64 + lane] = c0[0] + c1[1] + c2[2] + c3[3] + f0 + f1 + f2 + f3;
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  double *out; long long *cyc; hipMalloc(&out, 256 * 64 * 8); hipMalloc(&cyc, 256 * 8);
  const int reps = 1000;
  const char *names[] = {"mfma_f64_16x16x4 dependent (same acc)", "mfma_f64_16x16x4 4 independent accs", "v_fma_f64 dependent", "v_fma_f64 4 chains", "mfma chain through B operand (4 per stage)"};
  for (int mode = 0; mode < 5; mode++) {
    for (int grid : {1, 256}) {
      for (int it = 0; it < 2; it++) {
        switch (mode) {
          case 0: probe<0><<<grid, 64>>>(out, cyc, reps); break; case 1: probe<1><<<grid, 64>>>(out, cyc, reps); break;
          case 2: probe<2><<<grid, 64>>>(out, cyc, reps); break; case 3: probe<3><<<grid, 64>>>(out, cyc, reps); break;
          default: probe<4><<<grid, 64>>>(out, cyc, reps); break;
        }
        hipDeviceSynchronize();
      }
      long long h[256]; hipMemcpy(h, cyc, grid * 8, hipMemcpyDeviceToHost);
      printf("%-45s grid %3d: %.1f s_memtime ticks per instruction\n", names[mode], grid, (double)h[0] / (reps * 16.0));
    }
  }
  // s_memtime tick vs wall clock
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); probe<0><<<1, 64>>>(out, cyc, 20000); hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1); long long h0; hipMemcpy(&h0, cyc, 8, hipMemcpyDeviceToHost);
  printf("s_memtime: %lld ticks in %.3f ms -> %.1f MHz\n", h0, ms, h0 / ms / 1e3);
  return 0;
}
