// kernels_util.hpp -- small kernels around a solve: block-primitive self test, data validation (crossed bounds), dispatch-order hint
// Part of the single translation unit mpcqp.hip (included there in order; not a stand-alone header).
#pragma once

// block-primitive self test (mpcqp_debug_blockops)
extern "C" __global__ void __launch_bounds__(WAVE) mpcqp_blockops_kernel(const double *A, const double *B, const double *C, double *S,
                                                                          double *out_gemm, double *Sb, int *fail) {
  __shared__ double t0[BS * 17], t1[BS * 17];
  const int lane = threadIdx.x, row0 = lane >> 4, col = lane & 15;
  d4 prod = {0, 0, 0, 0};
  prod = mfma_abt(A, B, prod);
#pragma unroll
  for (int g = 0; g < 4; g++) out_gemm[(row0 + 4 * g) * BS + col] = C[(row0 + 4 * g) * BS + col] - prod[g];
  const bool ok = potrf_inv(S, Sb, t0, t1);
  if (lane == 0) *fail = ok ? 0 : 1;
}

// OSQP validates its data at setup and refuses a problem with l_i > u_i (OSQP_DATA_VALIDATION_ERROR: OsqpEigen's initSolver,
// reference src/sqp_solver/CuCaQP.cpp:183-197, returns false and nothing is solved).  Batched form of that refusal, run behind the
// solve kernel so that the hot kernels carry no extra state: an instance with crossed bounds reports MPCQP_UNSOLVED with 0
// iterations and NaN in x, y, z and the residuals (rho, info[3], is left for a kept workspace).  One wave per instance.
extern "C" __global__ void __launch_bounds__(256) mpcqp_validate_kernel(int batch, int n, int m, const double *__restrict__ l, long sl,
                                                                          const double *__restrict__ u, long su, double *x, double *y, double *z,
                                                                          int *status, int *iters, double *info) {
  const int lane = threadIdx.x & (WAVE - 1), b = blockIdx.x * (blockDim.x / WAVE) + threadIdx.x / WAVE;
  if (b >= batch) return;
  const double *lb = l + (long)b * sl, *ub = u + (long)b * su;
  int crossed = 0;
  for (int i = lane; i < m; i += WAVE) crossed |= lb[i] > ub[i];
  if (!__any(crossed)) return;
  for (int j = lane; j < n; j += WAVE) x[(long)b * n + j] = NAN;
  for (int i = lane; i < m; i += WAVE) { y[(long)b * m + i] = NAN; z[(long)b * m + i] = NAN; }
  if (lane == 0) { status[b] = MPCQP_UNSOLVED; iters[b] = 0; info[4L * b] = NAN; info[4L * b + 1] = NAN; info[4L * b + 2] = NAN; }
}

// Dispatch hint for the NEXT solve on a handle: instances ordered by descending iteration count of the solve that just
// finished (counting sort over iters / unit).  Instances are independent, so the order changes no result -- it only lets the
// long ones start first instead of wherever they sit in the batch: with one QP per workgroup and 25 / 50 / 75-iteration
// instances mixed, the in-order tail leaves CUs idle while the last long instance finishes (longest-processing-time-first
// scheduling; in an MPC loop consecutive solves of the same plants have correlated iteration counts).
__global__ void __launch_bounds__(1024) mpcqp_order_kernel(const int *__restrict__ iters, int *__restrict__ order, int batch, int unit) {
  constexpr int NB = 256;
  __shared__ int start[NB];
  const int tid = threadIdx.x;
  for (int k = tid; k < NB; k += blockDim.x) start[k] = 0;
  __syncthreads();
  for (int i = tid; i < batch; i += blockDim.x) atomicAdd(&start[min(max(iters[i], 0) / unit, NB - 1)], 1);
  __syncthreads();
  if (tid == 0) { int acc = 0; for (int k = NB - 1; k >= 0; k--) { const int c = start[k]; start[k] = acc; acc += c; } }
  __syncthreads();
  for (int i = tid; i < batch; i += blockDim.x) order[atomicAdd(&start[min(max(iters[i], 0) / unit, NB - 1)], 1)] = i;
}
