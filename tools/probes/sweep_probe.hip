// What bounds the two ELL sweeps of an ADMM iteration (A'w before the solve, A x after it) on the north-star size, measured in isolation: one workgroup of 4 waves per QP,
// two workgroups per CU (80 KB of LDS each), every QP with its own value arrays in a slab (85 + 81 slots x 64 lanes x 8 B: what the real launch streams), index arrays
// shared, the gather source in LDS, and between the sweeps a pause of the length of the solve (s_sleep), so that the L2 sees the access pattern of the real kernel.
// Variants:  V0 the kernel's form (chunks round-robin over the waves, up to 16 slots in flight per lane, one batch after the other)
//            V1 8 slots in flight            V2 every load of the wave's chunks issued before the first use (what free registers would allow)
//            V3 V0 with the chunks balanced over the waves by slots            V4 V0 without the pause (the sweeps back to back)
// Prints ticks (s_memtime: shader clock) per sweep as wave 0 sees them between barriers, and the bytes per second the whole chip streams.
// build: hipcc -O3 --offload-arch=gfx950 -o bin/sweep_probe sweep_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
constexpr int WAVE = 64, NW = 4, NT = NW * WAVE;
constexpr int NCA = 9, NCT = 6;
__constant__ int c_wA[NCA + 1], c_wT[NCT + 1];       // chunk offsets (slots)
struct Args { const double *val; const int *idxA, *idxT; long stride; int iters, pause, nA, nT; long long *ticks; double *out; };

template <int U>
__device__ __forceinline__ double batch(const double *__restrict__ &vp, const int *__restrict__ &ip, const double *in, double acc) {
  double v[U]; int ix[U];
#pragma unroll
  for (int u = 0; u < U; u++) { v[u] = vp[u * WAVE]; ix[u] = ip[u * WAVE]; }
#pragma unroll
  for (int u = 0; u < U; u++) acc += v[u] * in[ix[u]];
  vp += U * WAVE; ip += U * WAVE;
  return acc;
}
template <int UMAX>
__device__ __forceinline__ double chunk(const double *val, const int *idx, const double *in, int s0, int s1, int lane) {
  const double *__restrict__ vp = val + ((long)s0 * WAVE + lane); const int *__restrict__ ip = idx + ((long)s0 * WAVE + lane);
  double acc = 0.0; int rem = s1 - s0;
  if (UMAX >= 16) { for (; rem >= 16; rem -= 16) acc = batch<16>(vp, ip, in, acc); if (rem & 8) acc = batch<8>(vp, ip, in, acc); }
  else for (; rem >= 8; rem -= 8) acc = batch<8>(vp, ip, in, acc);
  if (rem & 4) acc = batch<4>(vp, ip, in, acc);
  if (rem & 2) acc = batch<2>(vp, ip, in, acc);
  if (rem & 1) acc = batch<1>(vp, ip, in, acc);
  return acc;
}
// every slot of the wave's (up to three) chunks in flight at once: W = the widest chunk; statically indexed registers, loads predicated by the chunk's width
template <int W>
__device__ __forceinline__ void all_at_once(const double *val, const int *idx, const double *in, double *outv, const int *co, int nch, int wid, int lane) {
  double v[3][W]; int ix[3][W];
#pragma unroll
  for (int j = 0; j < 3; j++) {
    const int c = wid + j * NW;
    const int s0 = c < nch ? co[c] : 0, w = c < nch ? co[c + 1] - s0 : 0;
#pragma unroll
    for (int s = 0; s < W; s++) { v[j][s] = 0.0; ix[j][s] = 0; if (s < w) { v[j][s] = val[(long)(s0 + s) * WAVE + lane]; ix[j][s] = idx[(long)(s0 + s) * WAVE + lane]; } }
  }
#pragma unroll
  for (int j = 0; j < 3; j++) {
    const int c = wid + j * NW;
    if (c < nch) {
      double acc = 0.0;
#pragma unroll
      for (int s = 0; s < W; s++) acc += v[j][s] * in[ix[j][s]];
      outv[c * WAVE + lane] = acc;
    }
  }
}

template <int V>
__global__ void __launch_bounds__(NT, 2) sweep(Args a) {
  extern __shared__ double lds[];
  double *X = lds, *R = lds + 1024, *Z = lds + 2048;       // gather sources / results
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, b = blockIdx.x;
  const double *valA = a.val + (long)b * a.stride, *valT = valA + (long)c_wA[NCA] * WAVE;
  for (int t = tid; t < 1024; t += NT) { X[t] = 1.0 + 1e-3 * t; R[t] = 0.5; Z[t] = 0.0; }
  __syncthreads();
  long long tT = 0, tA = 0, tP = 0;
  // V3: chunks by slots: A [1,1,1,1,1,20,20,20,20] -> wave w takes wide chunk 5 + w and narrow chunk w (wave 0 also chunk 4); T as is
  for (int it = 0; it < a.iters; it++) {
    long long t0 = __builtin_amdgcn_s_memtime();
    if (V == 2) all_at_once<16>(valT, a.idxT, X, R, c_wT, NCT, wid, lane);
    else for (int c = wid; c < NCT; c += NW) R[c * WAVE + lane] = chunk<(V == 1 ? 8 : 16)>(valT, a.idxT, X, c_wT[c], c_wT[c + 1], lane);
    __syncthreads();
    long long t1 = __builtin_amdgcn_s_memtime();
    if (V != 4) for (int p = 0; p < a.pause; p++) __builtin_amdgcn_s_sleep(100);      // ~6400 cycles per 100 x 64
    __syncthreads();
    long long t2 = __builtin_amdgcn_s_memtime();
    if (V == 2) all_at_once<20>(valA, a.idxA, R, Z, c_wA, NCA, wid, lane);
    else if (V == 3) {
      Z[(5 + wid) * WAVE + lane] = chunk<16>(valA, a.idxA, R, c_wA[5 + wid], c_wA[6 + wid], lane);
      Z[wid * WAVE + lane] = chunk<16>(valA, a.idxA, R, c_wA[wid], c_wA[wid + 1], lane);
      if (wid == 0) Z[4 * WAVE + lane] = chunk<16>(valA, a.idxA, R, c_wA[4], c_wA[5], lane);
    } else for (int c = wid; c < NCA; c += NW) Z[c * WAVE + lane] = chunk<(V == 1 ? 8 : 16)>(valA, a.idxA, R, c_wA[c], c_wA[c + 1], lane);
    __syncthreads();
    long long t3 = __builtin_amdgcn_s_memtime();
    for (int t = tid; t < 1024; t += NT) X[t] = 0.999 * X[t] + 1e-6 * Z[t & 511];
    __syncthreads();
    tT += t1 - t0; tP += t2 - t1; tA += t3 - t2;
  }
  if (tid == 0) { a.ticks[3L * b] = tT; a.ticks[3L * b + 1] = tP; a.ticks[3L * b + 2] = tA; }
  if (tid < 8) a.out[8L * b + tid] = X[tid] + Z[tid];
}

template <int V>
static void run(Args a, int batch, const char *what) {
  hipFuncSetAttribute((const void *)sweep<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  sweep<V><<<batch, NT, 80 * 1024>>>(a); hipDeviceSynchronize();
  hipEventRecord(e0); sweep<V><<<batch, NT, 80 * 1024>>>(a); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> t(3L * batch); hipMemcpy(t.data(), a.ticks, t.size() * 8, hipMemcpyDeviceToHost);
  double m[3] = {0, 0, 0}; for (int b = 0; b < batch; b++) for (int k = 0; k < 3; k++) m[k] += t[3L * b + k];
  for (int k = 0; k < 3; k++) m[k] /= (double)batch * a.iters;
  const double bytes = (double)batch * a.iters * (a.nA + a.nT) * WAVE * 12.0;      // 8 B value + 4 B index per slot and lane
  printf("V%d %-58s A' sweep %6.0f ticks, pause %6.0f, A sweep %6.0f  | launch %.2f ms, %.2f TB/s of values + indices\n", V, what, m[0], m[1], m[2], ms, bytes / (ms * 1e-3) / 1e12);
}
int main(int argc, char **argv) {
  const int batch = argc > 1 ? atoi(argv[1]) : 8192, iters = argc > 2 ? atoi(argv[2]) : 28, pause = argc > 3 ? atoi(argv[3]) : 2;
  const int wA[NCA] = {1, 1, 1, 1, 1, 20, 20, 20, 20}, wT[NCT] = {16, 16, 16, 16, 16, 1};
  int oA[NCA + 1] = {0}, oT[NCT + 1] = {0};
  for (int c = 0; c < NCA; c++) oA[c + 1] = oA[c] + wA[c];
  for (int c = 0; c < NCT; c++) oT[c + 1] = oT[c] + wT[c];
  hipMemcpyToSymbol(HIP_SYMBOL(c_wA), oA, sizeof oA); hipMemcpyToSymbol(HIP_SYMBOL(c_wT), oT, sizeof oT);
  const int nA = oA[NCA], nT = oT[NCT];
  const long stride = (long)(nA + nT) * WAVE;
  double *val; int *idxA, *idxT; long long *ticks; double *out;
  hipMalloc(&val, (size_t)batch * stride * 8); hipMalloc(&idxA, (size_t)nA * WAVE * 4); hipMalloc(&idxT, (size_t)nT * WAVE * 4);
  hipMalloc(&ticks, (size_t)batch * 24); hipMalloc(&out, (size_t)batch * 64);
  std::vector<double> hv((size_t)stride); for (size_t i = 0; i < hv.size(); i++) hv[i] = 1e-3 * (double)(i % 97);
  for (int b = 0; b < batch; b += 1) hipMemcpy(val + (long)b * stride, hv.data(), hv.size() * 8, hipMemcpyHostToDevice);
  std::vector<int> hi((size_t)nA * WAVE), ht((size_t)nT * WAVE);
  for (size_t i = 0; i < hi.size(); i++) hi[i] = (int)((i * 37 + 11) % 336);       // columns of A: positions
  for (size_t i = 0; i < ht.size(); i++) ht[i] = (int)((i * 53 + 7) % 576);        // columns of A': rows
  hipMemcpy(idxA, hi.data(), hi.size() * 4, hipMemcpyHostToDevice); hipMemcpy(idxT, ht.data(), ht.size() * 4, hipMemcpyHostToDevice);
  Args a{val, idxA, idxT, stride, iters, pause, nA, nT, ticks, out};
  printf("batch %d, %d iterations, %ld KB of values per QP and iteration, pause %d x s_sleep(100)\n", batch, iters, stride * 8 / 1024, pause);
  run<0>(a, batch, "as in the kernel (16 slots in flight, chunk after chunk)");
  run<1>(a, batch, "8 slots in flight");
  run<2>(a, batch, "every load of the wave's chunks before the first use");
  run<3>(a, batch, "chunks of A balanced over the waves (wide chunk first)");
  run<4>(a, batch, "as V0, no pause between the sweeps");
  return 0;
}
