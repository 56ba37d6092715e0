// Probe: cycles per stage of the on-chip solve's chain loop (kernel_onchip.hpp oc_solve F1 / B2) in isolation: one workgroup of 4 waves,
// wave 0 and 1 run a chain of `len` positions over LDS-resident blocks.  build: hipcc -O3 -std=c++17 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../optimal_control_problem_amd/csrc/plan.hpp"
using namespace mpcqp;
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NW> __device__ __forceinline__ void bsync() { __syncthreads(); }
#include "../../optimal_control_problem_amd/csrc/kernel_onchip.hpp"

template <int MODE>
__global__ void __launch_bounds__(256, 2) probe(const int *gtab, int len, int reps, long long *cyc, double *out) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  double *BL = lds, *R = lds + 24 * BLK; int *tab = reinterpret_cast<int *>(R + 32 * BS);
  for (int k = tid; k < 24 * BLK; k += 256) BL[k] = 1e-3 * ((k * 7) % 13 - 6);
  for (int k = tid; k < 32 * BS; k += 256) R[k] = 0.01 * (k % 17);
  for (int k = tid; k < 2 + 4 * len; k += 256) tab[k] = gtab[k];
  __syncthreads();
  const OcLane ln = oc_lane(lane);
  const int cb = wid == 0 ? 2 : 2 + 2 * len;
  long long t0 = __builtin_amdgcn_s_memtime();
  d4 keep = {0, 0, 0, 0};
  for (int r = 0; r < reps; r++) {
    if (wid < ((MODE & 8) ? 1 : 2)) {
      const int nst = len - 1;
      int2 e0 = oc_pair(tab, cb), e1 = oc_pair(tab, cb + 2 * min(1, nst)), e2 = oc_pair(tab, cb + 2 * min(2, nst)), e3 = oc_pair(tab, cb + 2 * min(3, nst));
      d4 x = oc_ldB(R, e0.x, ln), y = x;
      d4 a = oc_ldF(BL + (long)e0.y * BLK, ln), c = oc_ldB(R, e1.x, ln);
      int k = 0;
#pragma unroll
      for (int trip = 0; trip < 8; trip++) {
        if (k + 2 > nst) break;
        const d4 a1 = (MODE & 4) ? a : oc_ldF(BL + (long)e1.y * BLK, ln), c1 = oc_ldB(R, e2.x, ln);
        const int2 e4 = oc_pair(tab, cb + 2 * min(k + 4, nst)), e5 = oc_pair(tab, cb + 2 * min(k + 5, nst));     // entries of the next trip
        if (MODE & 1) __builtin_amdgcn_sched_barrier(0);
        y = oc_mv(a, x, c);
        if (!(MODE & 2)) oc_stB(R, e1.x, ln, y);
        if (!(MODE & 4)) a = oc_ldF(BL + (long)e2.y * BLK, ln);
        c = oc_ldB(R, e3.x, ln);
          if (MODE & 1) __builtin_amdgcn_sched_barrier(0);
        x = oc_mv(a1, y, c1);
        if (!(MODE & 2)) oc_stB(R, e2.x, ln, x);
        e0 = e2; e1 = e3; e2 = e4; e3 = e5; k += 2;
      }
      keep += x;
    }
    __syncthreads();
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[blockIdx.x * 4 + wid] = t1 - t0;
  out[blockIdx.x * 256 + tid] = keep[0] + keep[1] + keep[2] + keep[3];
}
int main() {
  const int len = 11, reps = 2000;
  std::vector<int> tab(2 + 4 * len);
  tab[0] = len; tab[1] = len;
  for (int k = 0; k < len; k++) { tab[2 + 2 * k] = 2 * k; tab[3 + 2 * k] = k; tab[2 + 2 * len + 2 * k] = 2 * k + 1; tab[3 + 2 * len + 2 * k] = 11 + k; }
  int *dt; long long *cyc; double *out;
  hipMalloc(&dt, tab.size() * 4); hipMemcpy(dt, tab.data(), tab.size() * 4, hipMemcpyHostToDevice);
  hipMalloc(&cyc, 512 * 4 * 8); hipMalloc(&out, 512 * 256 * 8);
  const size_t lds = (24 * BLK + 32 * BS) * 8 + 256;
  const char *names[] = {"as written", "loads pinned before the multiplies", "no vector stores", "one chain wave only", "constant block (no block loads)", "constant block, no stores, one wave"};
  for (int mode = 0; mode < 6; mode++) for (int grid : {1, 512}) {
    for (int it = 0; it < 2; it++) {
      switch (mode) { case 0: probe<0><<<grid, 256, lds>>>(dt, len, reps, cyc, out); break; case 1: probe<1><<<grid, 256, lds>>>(dt, len, reps, cyc, out); break;
                      case 2: probe<2><<<grid, 256, lds>>>(dt, len, reps, cyc, out); break; case 3: probe<8><<<grid, 256, lds>>>(dt, len, reps, cyc, out); break;
                      case 4: probe<4><<<grid, 256, lds>>>(dt, len, reps, cyc, out); break; default: probe<14><<<grid, 256, lds>>>(dt, len, reps, cyc, out); }
      hipDeviceSynchronize();
    }
    long long h[4]; hipMemcpy(h, cyc, 32, hipMemcpyDeviceToHost);
    printf("%-40s grid %3d: %.0f cycles per stage (10 stages per chain pass incl. barrier)\n", names[mode], grid, (double)h[0] / reps / (len - 1));
  }
  return 0;
}
