// Operand layout and issue cost of v_mfma_f64_4x4x4_4b_f64 on gfx950 (4 blocks of 4x4x4, one double per lane and operand).
// Layout: wave (la, lb) feeds A = [lane == la], B = [lane == lb] and records which lanes of D are 1.  Timing: a dependent chain.
// build: hipcc -O3 --offload-arch=gfx950 -o bin/mfma_4x4_probe mfma_4x4_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void layout(unsigned long long *out) {
  const int w = blockIdx.x, la = w >> 6, lb = w & 63, lane = threadIdx.x;
  const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
  const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
  const unsigned long long m = __ballot(d != 0.0);
  if (lane == 0) out[w] = m;
}
__global__ void timing(double *out, long long *cyc, int n) {
  double a = 1.0 + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3, d = 0.0, e = 0.0;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; i++) d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d, 0, 0, 0);           // dependent through C
  long long t1 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; i++) { e = __builtin_amdgcn_mfma_f64_4x4x4f64(a, e + b, 0.0, 0, 0, 0); }  // dependent through B (+ one VALU add)
  long long t2 = __builtin_amdgcn_s_memtime();
  double f = 0, g = 0, h = 0, k = 0;
  for (int i = 0; i < n; i++) { f = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, f, 0, 0, 0); g = __builtin_amdgcn_mfma_f64_4x4x4f64(b, a, g, 0, 0, 0);
                                h = __builtin_amdgcn_mfma_f64_4x4x4f64(a, a, h, 0, 0, 0); k = __builtin_amdgcn_mfma_f64_4x4x4f64(b, b, k, 0, 0, 0); }   // four independent chains
  long long t3 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = d + e + f + g + h + k;
  if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t3 - t2; }
}
int main() {
  unsigned long long *o; hipMalloc(&o, 4096 * 8);
  layout<<<4096, 64>>>(o);
  std::vector<unsigned long long> h(4096); hipMemcpy(h.data(), o, 4096 * 8, hipMemcpyDeviceToHost);
  // A lane la <-> (block, i, k); B lane lb <-> (block, k, j); D lane <-> (block, i, j).  Print, for every la, the B lanes it meets and where the product lands.
  for (int la = 0; la < 64; la++) {
    printf("A lane %2d:", la);
    for (int lb = 0; lb < 64; lb++) if (h[la * 64 + lb]) { printf("  B%2d->D", lb); for (int l = 0; l < 64; l++) if (h[la * 64 + lb] >> l & 1) printf("%d,", l); }
    printf("\n");
  }
  double *d; long long *c; hipMalloc(&d, 64 * 8); hipMalloc(&c, 3 * 8);
  const int n = 4096;
  timing<<<1, 64>>>(d, c, n); timing<<<1, 64>>>(d, c, n);
  long long hc[3]; hipMemcpy(hc, c, 24, hipMemcpyDeviceToHost);
  printf("cycles per MFMA: dependent through C %.1f, through B (+ v_add_f64) %.1f, four independent chains %.1f per instruction\n", hc[0] / (double)n, hc[1] / (double)n, hc[2] / (4.0 * n));
  return 0;
}
