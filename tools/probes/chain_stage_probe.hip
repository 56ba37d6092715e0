// What one stage of the on-chip solve's chain costs (t_next = c + W t, 16x16 block W in LDS, four steps of v_mfma_f64_4x4x4_4b_f64), by the way
// the result becomes the next stage's operand:
//   V0  ds_swizzle broadcast of quad K over the row of 16 lanes (kernel_onchip.hpp oc_bc4: through the LDS pipe, in order behind the wave's loads)
//   V1  DPP row rotations by 4 / 8 / 12 lanes (VALU): block b of step d takes the column piece K = (b + d) & 3, so piece 0 is the result itself
//   V2  V1 with the four steps on independent accumulators, summed afterwards
//   V3  V1 with the stage's LDS traffic issued behind its first MFMA
//   V4  V3 with four independent products summed as (p0 + p1) + (p2 + p3);  V5  V3 as two chains of two products
// with and without other waves of the workgroup reading LDS at the same time.  Also prints which way row_ror turns.
// build: hipcc -O3 --offload-arch=gfx950 -o bin/chain_stage_probe chain_stage_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int BS = 16, BLK = 256, NSTAGE = 24;
__host__ __device__ inline int swz(int r, int c) { return ((r ^ ((r >> 2) & 1)) << 4) | (c ^ (((r >> 1) & 3) << 2) ^ (((r >> 3) & 1) << 1)); }
__device__ __forceinline__ double mv4(double a, double v, double acc) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, v, acc, 0, 0, 0); }
template <int K>
__device__ __forceinline__ double bc4k(const double v) {
  union { double d; int i[2]; } a, r; a.d = v;
  r.i[0] = __builtin_amdgcn_ds_swizzle(a.i[0], (K << 7) | 0x13); r.i[1] = __builtin_amdgcn_ds_swizzle(a.i[1], (K << 7) | 0x13);
  return r.d;
}
template <int N>
__device__ __forceinline__ double ror(const double v) {
  union { double d; int i[2]; } a, r; a.d = v;
  r.i[0] = __builtin_amdgcn_update_dpp(0, a.i[0], 0x120 + N, 0xf, 0xf, false); r.i[1] = __builtin_amdgcn_update_dpp(0, a.i[1], 0x120 + N, 0xf, 0xf, false);
  return r.d;
}
__global__ void which_way(int *out) {
  const int lane = threadIdx.x;
  out[lane] = __builtin_amdgcn_update_dpp(0, lane, 0x120 + 4, 0xf, 0xf, false);      // row_ror:4
}
template <int V>
__global__ void __launch_bounds__(512) chain(const double *Wg, const double *cg, double *out, long long *ticks, const int reps, const int noisy) {
  extern __shared__ double lds[];
  double *BL = lds, *R = lds + NSTAGE * BLK, *T = R + (NSTAGE + 1) * BS, *junk = T + (NSTAGE + 1) * BS;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  for (int t = tid; t < NSTAGE * BLK; t += blockDim.x) { const int s = t / BLK, e = t % BLK; BL[s * BLK + swz(e / BS, e % BS)] = Wg[t]; }
  for (int t = tid; t < (NSTAGE + 1) * BS; t += blockDim.x) R[t] = cg[t];
  for (int t = tid; t < 4096; t += blockDim.x) junk[t] = t;
  __syncthreads();
  const int n = lane & 15, kk = lane >> 4, b = (lane >> 2) & 3, o4 = 4 * b + kk;
  int off[4], voff[4];
  for (int d = 0; d < 4; d++) { const int K = V == 0 ? d : ((b + d) & 3); /* V >= 1: rotated pieces */ off[d] = swz(n, kk + 4 * K); voff[d] = kk + 4 * K; }
  if (wid == 0) {
    double last = 0.0;
    const long long t0 = __builtin_amdgcn_s_memtime();
    if (V >= 3) {
      // V3: the LDS traffic of a stage (store of the previous result, loads of the next block and right-hand side) is issued BEHIND the
      // stage's first MFMA: nothing recent is outstanding where the next stage starts, whatever wait the compiler puts there
      for (int r = 0; r < reps; r++) {
        d4 x = {R[voff[0]], R[voff[1]], R[voff[2]], R[voff[3]]}, y = x;
        d4 a = {BL[off[0]], BL[off[1]], BL[off[2]], BL[off[3]]}, a1 = a; double c = R[BS + o4], c1 = c;
        auto stage = [&](const d4 &A, const d4 &X, const double C, d4 &Y, d4 &An, double &Cn, const int sprev, const int snext) {
          double acc = mv4(A[0], X[0], C), p1 = 0.0, p2 = 0.0, p3 = 0.0;
          if (V == 4) { p1 = mv4(A[1], X[1], 0.0); p2 = mv4(A[2], X[2], 0.0); p3 = mv4(A[3], X[3], 0.0); }      // V4: four independent products
          if (V == 5) p2 = mv4(A[2], X[2], 0.0);                                                                  // V5: two chains of two
          __builtin_amdgcn_sched_barrier(0);
          T[BS * sprev + o4] = X[0];
          const double *nb = BL + snext * BLK;
          An = d4{nb[off[0]], nb[off[1]], nb[off[2]], nb[off[3]]}; Cn = R[BS * (snext + 1) + o4];
          __builtin_amdgcn_sched_barrier(0);
          if (V == 3) { acc = mv4(A[1], X[1], acc); acc = mv4(A[2], X[2], acc); acc = mv4(A[3], X[3], acc); }
          if (V == 4) acc = (acc + p1) + (p2 + p3);
          if (V == 5) { acc = mv4(A[1], X[1], acc); p2 = mv4(A[3], X[3], p2); acc += p2; }
          Y = d4{acc, ror<12>(acc), ror<8>(acc), ror<4>(acc)};
          __builtin_amdgcn_sched_barrier(0);
        };
#pragma unroll 1
        for (int s = 0; s < NSTAGE; s += 2) {
          stage(a, x, c, y, a1, c1, s, s + 1);
          stage(a1, y, c1, x, a, c, s + 1, s + 2 < NSTAGE ? s + 2 : s + 1);
        }
        T[BS * NSTAGE + o4] = x[0];
        last = x[0];
      }
    } else
    for (int r = 0; r < reps; r++) {
      // two stages per trip, the block and right-hand side of a stage loaded one stage ahead (the shape of oc_solve's trip)
      d4 x = {R[voff[0]], R[voff[1]], R[voff[2]], R[voff[3]]}, y = x;
      d4 a = {BL[off[0]], BL[off[1]], BL[off[2]], BL[off[3]]}; double c = R[BS + o4];
      auto stage = [&](const d4 &A, const d4 &X, const double C, d4 &Y) -> double {
        double acc;
        if (V == 2) {
          const double p0 = mv4(A[0], X[0], C), p1 = mv4(A[1], X[1], 0.0), p2 = mv4(A[2], X[2], 0.0), p3 = mv4(A[3], X[3], 0.0);
          acc = (p0 + p1) + (p2 + p3);
        } else {
          acc = mv4(A[0], X[0], C); acc = mv4(A[1], X[1], acc); acc = mv4(A[2], X[2], acc); acc = mv4(A[3], X[3], acc);
        }
        if (V == 0) Y = d4{bc4k<0>(acc), bc4k<1>(acc), bc4k<2>(acc), bc4k<3>(acc)};
        else Y = d4{acc, ror<12>(acc), ror<8>(acc), ror<4>(acc)};
        return acc;
      };
#pragma unroll 1
      for (int s = 0; s < NSTAGE; s += 2) {
        const int s1 = s + 1, s2 = s + 2 < NSTAGE ? s + 2 : s + 1;
        const double *nb = BL + s1 * BLK;
        const d4 a1 = {nb[off[0]], nb[off[1]], nb[off[2]], nb[off[3]]}; const double c1 = R[BS * (s1 + 1) + o4];
        __builtin_amdgcn_sched_barrier(0);
        const double r0 = stage(a, x, c, y);
        __builtin_amdgcn_sched_barrier(0);
        T[BS * (s + 1) + o4] = r0;
        nb = BL + s2 * BLK;
        a = d4{nb[off[0]], nb[off[1]], nb[off[2]], nb[off[3]]}; c = R[BS * (s2 + 1) + o4];
        __builtin_amdgcn_sched_barrier(0);
        const double r1 = stage(a1, y, c1, x);
        __builtin_amdgcn_sched_barrier(0);
        T[BS * (s1 + 1) + o4] = r1;
        last = r1;
      }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) ticks[0] = t1 - t0;
    for (int t = lane; t < (NSTAGE + 1) * BS; t += 64) out[t] = T[t];
    out[(NSTAGE + 2) * BS + lane] = last;
  } else if (noisy) {
    // the other waves gather from LDS the way the sweeps of the iteration do
    double s = 0.0; int p = lane;
    for (int r = 0; r < reps * NSTAGE * 4; r++) { s += junk[p & 4095]; p = p * 5 + 17 + (int)s % 3; }
    out[(NSTAGE + 8) * BS + tid] = s;
  }
}
template <int V>
static void run(const double *W, const double *c, const std::vector<double> &ref, const int noisy) {
  double *o; long long *tk; hipMalloc(&o, 8192 * 8); hipMalloc(&tk, 8); hipMemset(o, 0, 8192 * 8);
  const int reps = 2000; const size_t lds = (NSTAGE * BLK + 2 * (NSTAGE + 1) * BS + 4096) * 8;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  chain<V><<<1, noisy ? 512 : 64, lds>>>(W, c, o, tk, 10, noisy);
  hipEventRecord(e0); chain<V><<<1, noisy ? 512 : 64, lds>>>(W, c, o, tk, reps, noisy); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long t; hipMemcpy(&t, tk, 8, hipMemcpyDeviceToHost);
  std::vector<double> h((NSTAGE + 1) * BS); hipMemcpy(h.data(), o, h.size() * 8, hipMemcpyDeviceToHost);
  double err = 0; for (int i = BS; i < (NSTAGE + 1) * BS; i++) err = fmax(err, fabs(h[i] - ref[i]));
  printf("V%d %s: %.1f s_memtime ticks per stage (the shader clock, 2.4 GHz), kernel %.1f ns per stage; max |t - reference| %.2e\n", V, noisy ? "7 other waves reading LDS" : "alone            ",
         t / (double(reps) * NSTAGE), ms * 1e6 / (double(reps) * NSTAGE), err);
  hipFree(o); hipFree(tk);
}

// The chains as the eight-wave kernel runs them: every CU busy, one workgroup of 512 threads per CU, two waves (wa, wb) run a chain each, the
// others wait at the barrier.  Records the SIMD of each wave (HW_ID bits 5:4) and the time of the slower chain.
__global__ void __launch_bounds__(512) chip(const double *Wg, const double *cg, double *out, long long *ticks, int *simd, const int reps, const int wa, const int wb) {
  extern __shared__ double lds[];
  double *BL = lds, *R = lds + NSTAGE * BLK, *T = R + (NSTAGE + 1) * BS;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  for (int t = tid; t < NSTAGE * BLK; t += blockDim.x) { const int s = t / BLK, e = t % BLK; BL[s * BLK + swz(e / BS, e % BS)] = Wg[t]; }
  for (int t = tid; t < (NSTAGE + 1) * BS; t += blockDim.x) R[t] = cg[t];
  __syncthreads();
  const int hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_REG_HW_ID
  if (lane == 0 && blockIdx.x < 64) simd[blockIdx.x * 8 + wid] = hw;
  const int n = lane & 15, kk = lane >> 4, b = (lane >> 2) & 3, o4 = 4 * b + kk;
  int off[4], voff[4];
  for (int d = 0; d < 4; d++) { off[d] = swz(n, kk + 4 * d); voff[d] = kk + 4 * d; }
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; r++) {
    if (wid == wa || wid == wb) {
      double *To = T + (wid == wa ? 0 : (NSTAGE + 1) * BS);
      d4 x = {R[voff[0]], R[voff[1]], R[voff[2]], R[voff[3]]}, y = x;
      d4 a = {BL[off[0]], BL[off[1]], BL[off[2]], BL[off[3]]}; double c = R[BS + o4];
#pragma unroll 1
      for (int s = 0; s < NSTAGE; s += 2) {
        const int s1 = s + 1, s2 = s + 2 < NSTAGE ? s + 2 : s + 1;
        const double *nb = BL + s1 * BLK;
        const d4 a1 = {nb[off[0]], nb[off[1]], nb[off[2]], nb[off[3]]}; const double c1 = R[BS * (s1 + 1) + o4];
        __builtin_amdgcn_sched_barrier(0);
        double r0 = mv4(a[0], x[0], c); r0 = mv4(a[1], x[1], r0); r0 = mv4(a[2], x[2], r0); r0 = mv4(a[3], x[3], r0);
        y = d4{bc4k<0>(r0), bc4k<1>(r0), bc4k<2>(r0), bc4k<3>(r0)};
        __builtin_amdgcn_sched_barrier(0);
        To[BS * (s + 1) + o4] = r0;
        nb = BL + s2 * BLK;
        a = d4{nb[off[0]], nb[off[1]], nb[off[2]], nb[off[3]]}; c = R[BS * (s2 + 1) + o4];
        __builtin_amdgcn_sched_barrier(0);
        double r1 = mv4(a1[0], y[0], c1); r1 = mv4(a1[1], y[1], r1); r1 = mv4(a1[2], y[2], r1); r1 = mv4(a1[3], y[3], r1);
        x = d4{bc4k<0>(r1), bc4k<1>(r1), bc4k<2>(r1), bc4k<3>(r1)};
        __builtin_amdgcn_sched_barrier(0);
        To[BS * (s1 + 1) + o4] = r1;
      }
    }
    __syncthreads();
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0) ticks[blockIdx.x] = t1 - t0;
  if (blockIdx.x == 0 && wid == wa) for (int t = lane; t < (NSTAGE + 1) * BS; t += 64) out[t] = T[t];
}
static void run_chip(const double *W, const double *c, const int nwg, const int wa, const int wb) {
  double *o; long long *tk; int *sd; hipMalloc(&o, 8192 * 8); hipMalloc(&tk, nwg * 8); hipMalloc(&sd, 512 * 4);
  const int reps = 500; const size_t lds = 150 * 1024;      // (one workgroup per CU)
  hipFuncSetAttribute((const void *)chip, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  chip<<<nwg, 512, lds>>>(W, c, o, tk, sd, 10, wa, wb);
  hipEventRecord(e0); chip<<<nwg, 512, lds>>>(W, c, o, tk, sd, reps, wa, wb); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> t(nwg); hipMemcpy(t.data(), tk, nwg * 8, hipMemcpyDeviceToHost);
  std::vector<int> h(512); hipMemcpy(h.data(), sd, 512 * 4, hipMemcpyDeviceToHost);
  double mean = 0; for (int i = 0; i < nwg; i++) mean += t[i]; mean /= nwg;
  printf("chip: %4d workgroups, chains on waves %d and %d: %.1f ticks per stage (mean over workgroups), kernel %.1f ns per stage; SIMD of waves 0..7 in workgroups 0, 1, 2:", nwg, wa, wb,
         mean / (double(reps) * NSTAGE), ms * 1e6 / (double(reps) * NSTAGE * ((nwg + 255) / 256)));
  for (int g = 0; g < 3; g++) { printf(" ["); for (int w = 0; w < 8; w++) printf("%d", (h[g * 8 + w] >> 4) & 3); printf("]"); }
  printf("\n");
  hipFree(o); hipFree(tk); hipFree(sd);
}
int main() {
  int *w; hipMalloc(&w, 256); which_way<<<1, 64>>>(w); int hw[64]; hipMemcpy(hw, w, 256, hipMemcpyDeviceToHost);
  printf("row_ror:4 -- lane 0 receives lane %d, lane 5 receives lane %d, lane 17 receives lane %d (receives from lane - 4 mod 16: rotation to the right)\n", hw[0], hw[5], hw[17]);
  std::vector<double> W(NSTAGE * BLK), c((NSTAGE + 1) * BS), ref((NSTAGE + 1) * BS);
  for (size_t i = 0; i < W.size(); i++) W[i] = 0.11 * sin(0.37 * i + 1.0);
  for (size_t i = 0; i < c.size(); i++) c[i] = cos(0.91 * i);
  for (int i = 0; i < BS; i++) ref[i] = c[i];
  for (int s = 0; s < NSTAGE; s++)
    for (int i = 0; i < BS; i++) { double a = c[BS * (s + 1) + i]; for (int k = 0; k < BS; k++) a += W[s * BLK + i * BS + k] * ref[BS * s + k]; ref[BS * (s + 1) + i] = a; }
  double *dW, *dc; hipMalloc(&dW, W.size() * 8); hipMalloc(&dc, c.size() * 8);
  hipMemcpy(dW, W.data(), W.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dc, c.data(), c.size() * 8, hipMemcpyHostToDevice);
  for (int noisy = 0; noisy < 2; noisy++) { run<0>(dW, dc, ref, noisy); run<1>(dW, dc, ref, noisy); run<2>(dW, dc, ref, noisy); run<3>(dW, dc, ref, noisy); run<4>(dW, dc, ref, noisy); run<5>(dW, dc, ref, noisy); }
  run_chip(dW, dc, 1, 0, 1); run_chip(dW, dc, 1, 0, 4); run_chip(dW, dc, 256, 0, 1); run_chip(dW, dc, 256, 0, 4); run_chip(dW, dc, 1024, 0, 1); run_chip(dW, dc, 1024, 0, 2);
  return 0;
}
