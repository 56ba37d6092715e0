// kernel_resident.hpp -- mpcqp_res_kernel<NW, MINW, GB, REUSE, ZYG>: block LDL' kernels with the factor in LDS or in the HBM slab, and their schedule executor
// Part of the single translation unit mpcqp.hip (included there in order; not a stand-alone header).
#pragma once

// =========================================================================================================
// Resident variant: NW waves per QP, block LDL' factor (G_J = D_J^-1, W_IJ) held in LDS for the whole solve.
// HBM is touched per iteration only for the ELL sweeps of A / A' (L2-resident per-QP slabs) and l, u.
// =========================================================================================================
#ifdef MPCQP_TIMING
#define TS_DECL unsigned long long ts_last = __builtin_amdgcn_s_memtime(), ts_acc[16] = {0}; const unsigned long long ts_first = ts_last, ts_rt0 = __builtin_amdgcn_s_memrealtime()
#define TS(k) do { unsigned long long t_ = __builtin_amdgcn_s_memtime(); ts_acc[k] += t_ - ts_last; ts_last = t_; } while (0)
#define TS_STORE(ptr) do { if (tid == 0 && (ptr)) { for (int k_ = 0; k_ < 16; k_++) (ptr)[16L * b + k_] = (long long)ts_acc[k_]; \
    if (b == 0) { (ptr)[16L * gridDim.x + 126] = (long long)(__builtin_amdgcn_s_memtime() - ts_first); (ptr)[16L * gridDim.x + 127] = (long long)(__builtin_amdgcn_s_memrealtime() - ts_rt0); } } } while (0)     /* (the last two: what s_memtime counts, against the constant 100 MHz counter) */
#else
// (phase boundaries stay scheduling boundaries in the product build: the timing build, whose time stamps make them so, was 6.5 % FASTER on the
// 12-state quadrotor -- without them the compiler moves loads of the next phase up into a phase whose registers are all spoken for)
#define TS_DECL
#ifdef MPCQP_ASM_MARKS
// (tools/isa_scratch_map.py: the phase boundaries as comments in the device assembly, to tell which phase a scratch access belongs to)
#define TS(k) do { __builtin_amdgcn_sched_barrier(0); asm volatile("; TS " #k ::: "memory"); } while (0)
#else
#define TS(k) __builtin_amdgcn_sched_barrier(0)
#endif
#define TS_STORE(ptr)
#endif
struct DevRes {
  int nphase, ntemp, nconst, rext;   // nconst constant blocks behind the factor blocks (slot nblk = -I), rext partial-sum doubles behind the solve vector
  const int *lv_ptr, *lv_diag, *lw_ptr, *lw_slot, *lw_g, *lu_ptr, *lu_dst, *lu_tmp, *lu_b, *g_ptr, *g_seg;
  int n_seg, nlev; long stage;
  int tmp_alias;   // global-block plans: the factorisation's temp tiles alias w (plan.hpp gb_tmp_alias)
};


struct RCtx {
  const DevPlan *pl; const DevRes *rs; const mpcqp_settings *st; double *ws;
  double *BL, *TMP, *X, *Q, *R, *Z, *Y, *W, *RB, *RED;
  double c, cinv, rho; int unscale; int wid, lane;
  const int *coA, *coAt, *coP;      // chunk offsets of the three ELL structures (on-chip mode: a copy in LDS -- every chunk of every sweep starts by reading two of them)
  unsigned long long fts[4];
};

// reduce K per-thread values over the workgroup: the first K - NSUM by max, the last NSUM by sum
template <int NW, int K, int NSUM>
__device__ __forceinline__ void block_combine(double (&v)[K], double *red, int wid, int lane) {
#pragma unroll
  for (int k = 0; k < K; k++) v[k] = k >= K - NSUM ? wave_sum(v[k]) : wave_max(v[k]);
  if (NW > 1) {
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < K; k++) red[wid * K + k] = v[k];
    }
    bsync<NW>();
#pragma unroll
    for (int k = 0; k < K; k++) {
      double r = red[k];
      for (int w = 1; w < NW; w++) r = k >= K - NSUM ? r + red[w * K + k] : fmax(r, red[w * K + k]);
      v[k] = r;
    }
    bsync<NW>();
  }
}

// One ELL chunk (64 rows, slots s0..s1) for this lane's row: sum_s val * in[idx]  (or max_s |val| * in[idx]).
// Pointer bumps + compile-time strides keep the address arithmetic in the loads' immediate offsets (the generic
// indexed form spent ~12 instructions per slot on 64-bit address math); batches of 8 / 4 / 2 / 1 slots issue all
// their loads before the first use.
template <bool MAXABS, int U>
__device__ __forceinline__ double ell_batch(const double *__restrict__ &vp, const int *__restrict__ &ip, const double *in, double acc) {
  double v[U]; int ix[U];
#pragma unroll
  for (int u = 0; u < U; u++) { v[u] = vp[u * WAVE]; ix[u] = ip[u * WAVE]; }
#pragma unroll
  for (int u = 0; u < U; u++) {
    const double x = in[ix[u]];
    acc = MAXABS ? fmax(acc, fabs(v[u]) * x) : acc + v[u] * x;
  }
  vp += U * WAVE; ip += U * WAVE;
  return acc;
}
template <bool MAXABS, int UMAX = 16>
__device__ __forceinline__ double ell_chunk(const double *__restrict__ val, const int *__restrict__ idx, const double *in, const int s0, const int s1, const int lane) {
  const double *__restrict__ vp = val + ((long)s0 * WAVE + lane);
  const int *__restrict__ ip = idx + ((long)s0 * WAVE + lane);
  double acc = 0.0;
  int rem = s1 - s0;
  if (UMAX >= 16) {
    for (; rem >= 16; rem -= 16) acc = ell_batch<MAXABS, 16>(vp, ip, in, acc);
    if (rem & 8) acc = ell_batch<MAXABS, 8>(vp, ip, in, acc);
  } else {
    for (; rem >= 8; rem -= 8) acc = ell_batch<MAXABS, 8>(vp, ip, in, acc);     // 16 loads in flight spill in the 128-VGPR instances
  }
  if (rem & 4) acc = ell_batch<MAXABS, 4>(vp, ip, in, acc);
  if (rem & 2) acc = ell_batch<MAXABS, 2>(vp, ip, in, acc);
  if (rem & 1) acc = ell_batch<MAXABS, 1>(vp, ip, in, acc);
  return acc;
}
// ell_chunk with the gathers of a batch issued TOGETHER.  In ell_batch the compiler serialises "x = in[idx]; acc += v * x" slot by slot: wait for the
// index, compute the LDS address, ds_read, wait for it (lgkmcnt(0)), fma -- one exposed LDS round trip of ~100 cycles per slot, 1,500 cycles for a
// batch of 16 (measured with the MPCQP_TIMING_SWEEP build: issue of the 32 loads 1,512 cycles, wait for them 858, gathers + fma 1,552).  Here all U
// LDS reads of a batch are in flight before the first fma (a compiler barrier between the two loops); the sum is still accumulated slot by slot in
// ascending order with one fma each: the same bits.
template <int U>
__device__ __forceinline__ double ell_batch_g(const double *__restrict__ &vp, const int *__restrict__ &ip, const double *in, double acc) {
  double v[U], x[U]; int ix[U];
#pragma unroll
  for (int u = 0; u < U; u++) { v[u] = vp[u * WAVE]; ix[u] = ip[u * WAVE]; }
#pragma unroll
  for (int u = 0; u < U; u++) x[u] = in[ix[u]];
  asm volatile("" ::: "memory");
#pragma unroll
  for (int u = 0; u < U; u++) acc = __builtin_fma(v[u], x[u], acc);
  vp += U * WAVE; ip += U * WAVE;
  return acc;
}
template <int UMAX>
__device__ __forceinline__ double ell_chunk_g(const double *__restrict__ val, const int *__restrict__ idx, const double *in, const int s0, const int s1, const int lane) {
  const double *__restrict__ vp = val + ((long)s0 * WAVE + lane);
  const int *__restrict__ ip = idx + ((long)s0 * WAVE + lane);
  double acc = 0.0;
  int rem = s1 - s0;
  if (UMAX >= 16) {
    for (; rem >= 16; rem -= 16) acc = ell_batch_g<16>(vp, ip, in, acc);
    if (rem & 8) acc = ell_batch_g<8>(vp, ip, in, acc);
  } else {
    for (; rem >= 8; rem -= 8) acc = ell_batch_g<8>(vp, ip, in, acc);
  }
  if (rem & 4) acc = ell_batch_g<4>(vp, ip, in, acc);
  if (rem & 2) acc = ell_batch_g<2>(vp, ip, in, acc);
  if (rem & 1) acc = ell_batch_g<1>(vp, ip, in, acc);
  return acc;
}

// Ruiz equilibration, one sweep over A by rows for both norms: max_j |a_ij| d_j stays in the row's lane, and |a_ij| e_i goes to column
// j's accumulator with an LDS atomic max on the bit pattern (non-negative doubles order like their bit patterns) -- A' is not read.
typedef __attribute__((address_space(3))) unsigned long long lds_u64;
// (Every lane walks its row's slots in its own rotation, slot (u + lane) mod width in step u: the rows of one stage share their columns slot by
// slot, and same-address LDS atomics of one instruction serialise -- 12 rows of a stage hitting one column cost the sweep half its time.  A maximum
// does not care about the order.)
template <int U, class IT>
__device__ __forceinline__ double ell_batch_rc(const double *__restrict__ val, const IT *__restrict__ idx, const double *din, const double ei, double *colacc,
                                               const long e0, const int u0, const int r0, const int w, double acc) {
  double v[U]; int ix[U];
#pragma unroll
  for (int u = 0; u < U; u++) {
    int su = u0 + u + r0; su -= su >= w ? w : 0;
    const long e = e0 + (long)su * WAVE;
    v[u] = fabs(val[e]); ix[u] = idx[e];
  }
#pragma unroll
  for (int u = 0; u < U; u++) {
    acc = fmax(acc, v[u] * din[ix[u]]);
    if (v[u] != 0.0) __hip_atomic_fetch_max((lds_u64 *)(colacc + ix[u]), (unsigned long long)__double_as_longlong(v[u] * ei), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  return acc;
}
template <class IT>
__device__ __forceinline__ double ell_chunk_rc(const double *__restrict__ val, const IT *__restrict__ idx, const double *din, const double ei, double *colacc, const int s0, const int s1, const int lane) {
  const int w = s1 - s0;
  if (w <= 0) return 0.0;
  const long e0 = (long)s0 * WAVE + lane;
  const int r0 = lane % w;
  double acc = 0.0;
  int u0 = 0;
  for (; u0 + 8 <= w; u0 += 8) acc = ell_batch_rc<8>(val, idx, din, ei, colacc, e0, u0, r0, w, acc);
  if ((w - u0) & 4) { acc = ell_batch_rc<4>(val, idx, din, ei, colacc, e0, u0, r0, w, acc); u0 += 4; }
  if ((w - u0) & 2) { acc = ell_batch_rc<2>(val, idx, din, ei, colacc, e0, u0, r0, w, acc); u0 += 2; }
  if ((w - u0) & 1) acc = ell_batch_rc<1>(val, idx, din, ei, colacc, e0, u0, r0, w, acc);
  return acc;
}
// max_s |val| * in[idx] of one chunk with the gathers of a batch in flight together (a maximum does not care about the order); the index array may be
// a 16-bit table in LDS
template <int U, class IT>
__device__ __forceinline__ double ell_batch_mx(const double *__restrict__ &vp, const IT *__restrict__ &ip, const double *in, double acc) {
  double v[U], x[U]; int ix[U];
#pragma unroll
  for (int u = 0; u < U; u++) { v[u] = vp[u * WAVE]; ix[u] = ip[u * WAVE]; }
#pragma unroll
  for (int u = 0; u < U; u++) x[u] = in[ix[u]];
  asm volatile("" ::: "memory");
#pragma unroll
  for (int u = 0; u < U; u++) acc = fmax(acc, fabs(v[u]) * x[u]);
  vp += U * WAVE; ip += U * WAVE;
  return acc;
}
template <class IT>
__device__ __forceinline__ double ell_chunk_mx(const double *__restrict__ val, const IT *__restrict__ idx, const double *in, const int s0, const int s1, const int lane) {
  const double *__restrict__ vp = val + ((long)s0 * WAVE + lane);
  const IT *__restrict__ ip = idx + ((long)s0 * WAVE + lane);
  double acc = 0.0;
  int rem = s1 - s0;
  for (; rem >= 8; rem -= 8) acc = ell_batch_mx<8>(vp, ip, in, acc);
  if (rem & 4) acc = ell_batch_mx<4>(vp, ip, in, acc);
  if (rem & 2) acc = ell_batch_mx<2>(vp, ip, in, acc);
  if (rem & 1) acc = ell_batch_mx<1>(vp, ip, in, acc);
  return acc;
}
// out[e] = f(value, index) over the slots of one chunk, 8 / 4 / 2 / 1 slots per batch with all loads of a batch issued before the first use.
// INDIRECT: the value is gathered from the caller's array through the slot's source index (-1 = padding)
template <bool INDIRECT, int U, class F>
__device__ __forceinline__ void ell_map_batch(const double *&vp, const int *__restrict__ &sp, const double *in, const int *__restrict__ &ip, double *&op, F &f) {
  double v[U]; int ix[U], sr[U];
#pragma unroll
  for (int u = 0; u < U; u++) { if (INDIRECT) sr[u] = sp[u * WAVE]; else v[u] = vp[u * WAVE]; ix[u] = ip[u * WAVE]; }
  if (INDIRECT) {
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = sr[u] >= 0 ? in[sr[u]] : 0.0;
  }
#pragma unroll
  for (int u = 0; u < U; u++) op[u * WAVE] = f(v[u], ix[u]);
  if (INDIRECT) sp += U * WAVE; else vp += U * WAVE;
  ip += U * WAVE; op += U * WAVE;
}
template <bool INDIRECT, class F>
__device__ __forceinline__ void ell_map_chunk(const double *val, const int *__restrict__ src, const double *in, const int *__restrict__ idx, double *out,
                                              const int s0, const int s1, const int lane, F &&f) {
  // (val and out may be the same array: scaled in place)
  const long o = (long)s0 * WAVE + lane;
  const double *vp = val + o; const int *__restrict__ sp = src + o; const int *__restrict__ ip = idx + o; double *op = out + o;
  int rem = s1 - s0;
  for (; rem >= 8; rem -= 8) ell_map_batch<INDIRECT, 8>(vp, sp, in, ip, op, f);
  if (rem & 4) ell_map_batch<INDIRECT, 4>(vp, sp, in, ip, op, f);
  if (rem & 2) ell_map_batch<INDIRECT, 2>(vp, sp, in, ip, op, f);
  if (rem & 1) ell_map_batch<INDIRECT, 1>(vp, sp, in, ip, op, f);
}
// ELL sweeps for the multi-wave kernels: wave `wid` takes chunks wid, wid + NW, ... (A 16-deep clamped full unroll
// and a 4-lanes-per-row split were both measured slower on MI355X: spills / more latency rounds; see DESIGN.md.)
template <int NW, int UMAX = 16, class F>
__device__ __forceinline__ void ell_rows_w(const DevEll &E, const int *co, const double *__restrict__ val, const double *in, int wid, int lane, F &&f) {
  for (int c = wid; c < E.nchunks; c += NW) f(c * WAVE + lane, ell_chunk<false, UMAX>(val, E.idx, in, co[c], co[c + 1], lane));
}
template <int NW, class F>
__device__ __forceinline__ void ell_rowmax_w(const DevEll &E, const int *co, const double *__restrict__ val, const double *in, int wid, int lane, F &&f) {
  for (int c = wid; c < E.nchunks; c += NW) f(c * WAVE + lane, ell_chunk<true>(val, E.idx, in, co[c], co[c + 1], lane));
}

// sum over the 4 lanes of a quad with DPP quad_perm (no LDS crossbar round trip)
__device__ __forceinline__ double quad_sum(double v) {
  union { double d; int i[2]; } a, t;
  a.d = v;
  t.i[0] = __builtin_amdgcn_mov_dpp(a.i[0], 0xB1, 0xF, 0xF, true);   // quad_perm:[1,0,3,2]
  t.i[1] = __builtin_amdgcn_mov_dpp(a.i[1], 0xB1, 0xF, 0xF, true);
  a.d += t.d;
  t.i[0] = __builtin_amdgcn_mov_dpp(a.i[0], 0x4E, 0xF, 0xF, true);   // quad_perm:[2,3,0,1]
  t.i[1] = __builtin_amdgcn_mov_dpp(a.i[1], 0x4E, 0xF, 0xF, true);
  return a.d + t.d;
}

// acc += A * B^T, operands row-major 16x16 tiles (LDS or global), lane = lane within the wave
__device__ __forceinline__ d4 mfma_abt_l(const double *A, const double *B, d4 acc, int lane) {
  const int rr = lane & 15, kk = lane >> 4;
#pragma unroll
  for (int k0 = 0; k0 < BS; k0 += 4) {
    const double a = A[rr * BS + k0 + kk];
    const double b = B[rr * BS + k0 + kk];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  }
  return acc;
}

// ---- solve-schedule executor -------------------------------------------------------------------------------
// A lone wave issues roughly one instruction per 4-8 cycles, so the executor is written for instruction count:
// the host compresses each wave's op list into segments whose block / src / dst byte offsets are arithmetic
// progressions; each segment is one tight, branch-free, software-pipelined loop (4 lanes per output row; the next
// op's 4 block entries and old destination value are fetched while the current op is reduced with a quad DPP sum).
template <bool T>
__device__ __forceinline__ d4 load_blk(const char *BLb, const int off, const int offN, const int offT) {
  if (T) {
    const char *B = BLb + off + offT;
    d4 r;
    r[0] = *reinterpret_cast<const double *>(B); r[1] = *reinterpret_cast<const double *>(B + BS * 8);
    r[2] = *reinterpret_cast<const double *>(B + 2 * BS * 8); r[3] = *reinterpret_cast<const double *>(B + 3 * BS * 8);
    return r;
  }
  return *reinterpret_cast<const d4 *>(BLb + off + offN);
}
__device__ __forceinline__ void wave_order() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// blocks streamed from the HBM slab (GB variants): PD blocks of the segment in flight, destination read at use time
template <bool T, bool SET, int PD>
__device__ __forceinline__ void seg_each_g(const char *BLb, char *vecb, int b, int s, int d, const int cnt, const int db, const int ds, const int dd,
                                           const int offN, const int offT, const int offV, const int offD, const int j, double carry) {
  d4 ring[PD];
#pragma unroll
  for (int u = 0; u < PD; u++) { ring[u] = d4{0, 0, 0, 0}; if (u < cnt) ring[u] = load_blk<T>(BLb, b + u * db, offN, offT); }
  for (int k0 = 0; k0 < cnt; k0 += PD) {
#pragma unroll
    for (int u = 0; u < PD; u++) {
      if (k0 + u < cnt) {
        const d4 bb = ring[u];
        if (k0 + u + PD < cnt) ring[u] = load_blk<T>(BLb, b + (k0 + u + PD) * db, offN, offT);
        const d4 v = *reinterpret_cast<const d4 *>(vecb + s + offV);
        const double sum = quad_sum(carry + (bb[0] * v[0] + bb[1] * v[1] + bb[2] * v[2] + bb[3] * v[3]));
        carry = 0.0;
        if (j == 0) { double *o = reinterpret_cast<double *>(vecb + d + offD); *o = SET ? sum : *o - sum; }
        wave_order();
        s += ds; d += dd;
      }
    }
  }
}
// every op writes its own destination: dst_k = (SET ? 0 : dst_k) -/+ B_k * src_k ; `carry` joins the first op
template <bool T, bool SET>
__device__ __forceinline__ void seg_each(const char *BLb, char *vecb, int b, int s, int d, const int cnt, const int db, const int ds, const int dd,
                                         const int offN, const int offT, const int offV, const int offD, const int j, double carry) {
  d4 bb = load_blk<T>(BLb, b, offN, offT);
  double old = SET ? 0.0 : *reinterpret_cast<const double *>(vecb + d + offD);
  for (int k = 0; k < cnt; k++) {
    const d4 v = *reinterpret_cast<const d4 *>(vecb + s + offV);
    d4 nb = bb; double nold = old;
    if (k + 1 < cnt) {
      nb = load_blk<T>(BLb, b + db, offN, offT);
      if (!SET) nold = *reinterpret_cast<const double *>(vecb + d + dd + offD);
    }
    const double sum = quad_sum(carry + (bb[0] * v[0] + bb[1] * v[1] + bb[2] * v[2] + bb[3] * v[3]));
    carry = 0.0;
    if (j == 0) *reinterpret_cast<double *>(vecb + d + offD) = SET ? sum : old - sum;
    wave_order();
    bb = nb; old = nold; b += db; s += ds; d += dd;
  }
}
// SG_IND segments: the ops are mutually independent, so two are in flight at once (loads, FMAs, DPP sums interleave)
template <bool T, bool SET>
__device__ __forceinline__ void seg_each2(const char *BLb, char *vecb, int b, int s, int d, const int cnt, const int db, const int ds, const int dd,
                                          const int offN, const int offT, const int offV, const int offD, const int j) {
  int k = 0;
  for (; k + 2 <= cnt; k += 2) {
    const d4 b0 = load_blk<T>(BLb, b, offN, offT), b1 = load_blk<T>(BLb, b + db, offN, offT);
    const d4 v0 = *reinterpret_cast<const d4 *>(vecb + s + offV), v1 = *reinterpret_cast<const d4 *>(vecb + s + ds + offV);
    double o0 = 0.0, o1 = 0.0;
    if (!SET) { o0 = *reinterpret_cast<const double *>(vecb + d + offD); o1 = *reinterpret_cast<const double *>(vecb + d + dd + offD); }
    const double s0 = quad_sum(b0[0] * v0[0] + b0[1] * v0[1] + b0[2] * v0[2] + b0[3] * v0[3]);
    const double s1 = quad_sum(b1[0] * v1[0] + b1[1] * v1[1] + b1[2] * v1[2] + b1[3] * v1[3]);
    if (j == 0) {
      *reinterpret_cast<double *>(vecb + d + offD) = SET ? s0 : o0 - s0;
      *reinterpret_cast<double *>(vecb + d + dd + offD) = SET ? s1 : o1 - s1;
    }
    b += 2 * db; s += 2 * ds; d += 2 * dd;
  }
  wave_order();
  if (k < cnt) seg_each<T, SET>(BLb, vecb, b, s, d, cnt - k, db, ds, dd, offN, offT, offV, offD, j, 0.0);
}
// the ops accumulate into one destination; returns the per-lane partial sum
template <bool T, bool GB>
__device__ __forceinline__ double seg_run(const char *BLb, const char *vecb, int b, int s, const int cnt, const int db, const int ds,
                                          const int offN, const int offT, const int offV, double acc) {
#pragma unroll 8
  for (int k = 0; k < cnt; k++) {
    const d4 bb = load_blk<T>(BLb, b, offN, offT);
    const d4 v = *reinterpret_cast<const d4 *>(vecb + s + offV);
    acc += bb[0] * v[0] + bb[1] * v[1] + bb[2] * v[2] + bb[3] * v[3];
    b += db; s += ds;
  }
  return acc;
}
template <int NW, bool GB, int PD = 6>
__device__ __forceinline__ void run_schedule(const int4 *segs, const int g0, const int g1, const char *BLb, char *vecb, const int lane, const int ni_off, long long *trace = nullptr) {
  const int r = lane >> 2, j = lane & 3;
  const int offN = (r * BS + 4 * j) * 8, offT = ((4 * j) * BS + r) * 8, offV = 32 * j, offD = 8 * r;
  double acc = 0.0;
  for (int g = g0; g < g1; g++) {
    int4 a = segs[2 * g], c = segs[2 * g + 1];
    const int b0 = __builtin_amdgcn_readfirstlane(a.x), s0 = __builtin_amdgcn_readfirstlane(a.y), d0 = __builtin_amdgcn_readfirstlane(a.z);
    const int fl = __builtin_amdgcn_readfirstlane(a.w), cnt = __builtin_amdgcn_readfirstlane(c.x);
    const int db = __builtin_amdgcn_readfirstlane(c.y), ds = __builtin_amdgcn_readfirstlane(c.z), dd = __builtin_amdgcn_readfirstlane(c.w);
    if (fl & SG_NOP) { bsync<NW>(); continue; }
    if (GB && (fl & SG_EACH)) {
      if (fl & SG_SET) seg_each_g<false, true, PD>(BLb, vecb, b0, s0, d0, cnt, db, ds, dd, offN, offT, offV, offD, j, 0.0);
      else if (fl & SG_T) seg_each_g<true, false, PD>(BLb, vecb, b0, s0, d0, cnt, db, ds, dd, offN, offT, offV, offD, j, acc);
      else seg_each_g<false, false, PD>(BLb, vecb, b0, s0, d0, cnt, db, ds, dd, offN, offT, offV, offD, j, acc);
      acc = 0.0;
    } else if ((fl & (SG_EACH | SG_IND)) == (SG_EACH | SG_IND)) {
      if (fl & SG_SET) seg_each2<false, true>(BLb, vecb, b0, s0, d0, cnt, db, ds, dd, offN, offT, offV, offD, j);
      else if (fl & SG_T) seg_each2<true, false>(BLb, vecb, b0, s0, d0, cnt, db, ds, dd, offN, offT, offV, offD, j);
      else seg_each2<false, false>(BLb, vecb, b0, s0, d0, cnt, db, ds, dd, offN, offT, offV, offD, j);
    } else if (fl & SG_EACH) {
      if (fl & SG_SET) seg_each<false, true>(BLb, vecb, b0, s0, d0, cnt, db, ds, dd, offN, offT, offV, offD, j, 0.0);
      else if (fl & SG_T) seg_each<true, false>(BLb, vecb, b0, s0, d0, cnt, db, ds, dd, offN, offT, offV, offD, j, acc);
      else seg_each<false, false>(BLb, vecb, b0, s0, d0, cnt, db, ds, dd, offN, offT, offV, offD, j, acc);
      acc = 0.0;
    } else if (GB && b0 == ni_off && db == 0 && (fl & SG_END)) {
      // the combine run of a split accumulation: every op is (-I) * partial, so the run is a plain vector sum -- the constant
      // block is not fetched from the slab
      if (j == 0) {
        double a = 0.0;
        for (int k = 0; k < cnt; k++) a += *reinterpret_cast<const double *>(vecb + s0 + k * ds + offD);
        double *o = reinterpret_cast<double *>(vecb + d0 + offD); *o = *o + a;
      }
      wave_order();
    } else {
      acc = (fl & SG_T) ? seg_run<true, GB>(BLb, vecb, b0, s0, cnt, db, ds, offN, offT, offV, acc)
                        : seg_run<false, GB>(BLb, vecb, b0, s0, cnt, db, ds, offN, offT, offV, acc);
      if (fl & SG_END) {
        const double sum = quad_sum(acc);
        acc = 0.0;
        if (j == 0) { double *o = reinterpret_cast<double *>(vecb + d0 + offD); *o = *o - sum; }
        wave_order();
      }
    }
    if (fl & SG_BAR) bsync<NW>();
#ifdef MPCQP_TIMING
    if (trace && g - g0 < 60) { trace[2 + 2 * (g - g0)] = (long long)__builtin_amdgcn_s_memtime(); trace[3 + 2 * (g - g0)] = ((long long)fl << 32) | cnt; }
#endif
  }
}

// factorisation set-up loops over ELL slots, 8 / 4 / 2 / 1 per batch with every load of a batch issued before the first use
template <int U>
__device__ __forceinline__ void dvec_batch(const double *&vp, const int *&fp, const int *&ip, const double *w, double &acc) {
  double v[U]; int fl[U], ix[U];
#pragma unroll
  for (int u = 0; u < U; u++) { v[u] = vp[u * WAVE]; fl[u] = fp[u * WAVE]; ix[u] = ip[u * WAVE]; }
#pragma unroll
  for (int u = 0; u < U; u++) if (fl[u]) acc += w[ix[u]] * v[u] * v[u];
  vp += U * WAVE; fp += U * WAVE; ip += U * WAVE;
}
template <int U>
__device__ __forceinline__ void tscatter_batch(const double *&vp, const int *&tp, double *T, const double sr) {
  double v[U]; int t[U];
#pragma unroll
  for (int u = 0; u < U; u++) { v[u] = vp[u * WAVE]; t[u] = tp[u * WAVE]; }
#pragma unroll
  for (int u = 0; u < U; u++) if (t[u] >= 0) T[t[u]] = v[u] * sr;
  vp += U * WAVE; tp += U * WAVE;
}
// OCM: 0 the level loop below; 1 / 2 the on-chip topology without / with an arrow head: kernel_onchip.hpp's oc_ldl on the assembled blocks
// TAIL: leave w = rho z - y behind (the set-up kernel of kernel_oc_split.hpp has no iterate yet: false)
template <int NW, int OCM = 0, bool TAIL = true>
__device__ __forceinline__ bool factorize_res(RCtx &cx, const DevOc *oc = nullptr, const int *octab = nullptr, double *scr = nullptr) {
  const DevPlan &pl = *cx.pl; const DevRes &rs = *cx.rs; double *ws = cx.ws;
  const int wid = cx.wid, lane = cx.lane, tid = wid * WAVE + lane; constexpr int NT = NW * WAVE;
  const double *lb = ws + pl.o_l, *ub = ws + pl.o_u;
  const double *valA = ws + pl.o_ellA, *valAt = ws + pl.o_ellAt, *valP = ws + pl.o_ellP;
  double *T = ws + pl.o_T;
#ifdef MPCQP_TIMING
  unsigned long long f0 = __builtin_amdgcn_s_memtime();
#endif
  for (int i = tid; i < pl.mpad; i += NT) cx.W[i] = i < pl.m ? rho_of(lb[i], ub[i], cx.rho) : 0.0;
  // (the T tiles' structural zeros were written once, at creation: only the non-zeros are refreshed below)
  bsync<NW>();
  {
    // d_t = sigma + sum_i rho_i a_it^2 over the singleton rows of column t (the rest of A' rho A goes through the T tiles)
    const double sigma = cx.st->sigma;
    const DevEll &E = pl.At;
    for (int c = wid; c < E.nchunks; c += NW) {
      double acc = 0.0;
      const int s0 = cx.coAt[c], s1 = cx.coAt[c + 1];
      const long o = (long)s0 * WAVE + lane;
      const double *vp = valAt + o; const int *fp = E.flag + o, *ip = E.idx + o;
      int rem = s1 - s0;
      for (; rem >= 8; rem -= 8) dvec_batch<8>(vp, fp, ip, cx.W, acc);
      if (rem & 4) dvec_batch<4>(vp, fp, ip, cx.W, acc);
      if (rem & 2) dvec_batch<2>(vp, fp, ip, cx.W, acc);
      if (rem & 1) dvec_batch<1>(vp, fp, ip, cx.W, acc);
      const int t = c * WAVE + lane;
      if (t < pl.npad) cx.R[t] = pl.perm[t] >= 0 ? sigma + acc : 1.0;
    }
  }
  {
    const DevEll &E = pl.A;
    for (int c = wid; c < E.nchunks; c += NW) {
      const int i = c * WAVE + lane;
      const double sr = sqrt(cx.W[i]);
      const int s0 = cx.coA[c], s1 = cx.coA[c + 1];
      const long o = (long)s0 * WAVE + lane;
      const double *vp = valA + o; const int *tp = pl.tpos + o;
      int rem = s1 - s0;
      for (; rem >= 8; rem -= 8) tscatter_batch<8>(vp, tp, T, sr);
      if (rem & 4) tscatter_batch<4>(vp, tp, T, sr);
      if (rem & 2) tscatter_batch<2>(vp, tp, T, sr);
      if (rem & 1) tscatter_batch<1>(vp, tp, T, sr);
    }
  }
  int *rec = nullptr;
  if constexpr (OCM > 0) {      // the assembly recipe of every block, one 32-byte record each, next to the factorisation's scratch blocks
    rec = reinterpret_cast<int *>(scr + OC_LDL_SCR * BLK);
    for (int k = tid; k < 8 * pl.nblk; k += NT) rec[k] = oc->asm_rec[k];
  }
  bsync<NW>();
#ifdef MPCQP_TIMING
  unsigned long long f1 = __builtin_amdgcn_s_memtime(); cx.fts[0] += f1 - f0;
#endif
  const int row0 = lane >> 4, col = lane & 15;
  if constexpr (OCM > 0) {
    // S_b = sum_g T_a(g) T_b(g)' + P entries + the diagonal, three blocks in flight per wave: while block b multiplies, the T tiles of the
    // wave's next block and its P values are on their way, and the P indices of the one after that (every stage a slab round trip that the
    // loop used to wait for, block after block)
    // Every fetch is unconditional -- unused terms read the zero tile behind the T tiles, a block past the wave's last one is fetched again --
    // so the loop body is straight-line code and the compiler's wait counts stay exact (a conditional load makes it wait for everything)
    const int nblk = pl.nblk;
    auto rfl = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    auto fetch_ops = [&](const int bb, d4 (&A)[3], d4 (&B)[3], int &n, int &J) {
      const int4 r0 = *reinterpret_cast<const int4 *>(rec + 8 * bb), r1 = *reinterpret_cast<const int4 *>(rec + 8 * bb + 4);
      n = rfl(r0.x); J = rfl(r0.y);
      // (both tiles of a product as row pieces T[row][4 kk .. 4 kk + 3]: one 32-byte read per lane, the k index of MFMA i being 4 kk + i for both)
      const int o = (lane & 15) * BS + 4 * (lane >> 4);
      A[0] = *reinterpret_cast<const d4 *>(T + (long)rfl(r0.z) * BLK + o); B[0] = *reinterpret_cast<const d4 *>(T + (long)rfl(r0.w) * BLK + o);
      if (n > 1) { A[1] = *reinterpret_cast<const d4 *>(T + (long)rfl(r1.x) * BLK + o); B[1] = *reinterpret_cast<const d4 *>(T + (long)rfl(r1.y) * BLK + o); }
      if (n > 2) { A[2] = *reinterpret_cast<const d4 *>(T + (long)rfl(r1.z) * BLK + o); B[2] = *reinterpret_cast<const d4 *>(T + (long)rfl(r1.w) * BLK + o); }
    };
    auto fetch_pidx = [&](const int bb, int (&pi)[4]) {
#pragma unroll
      for (int g = 0; g < 4; g++) pi[g] = pl.asm_pidx[(long)bb * BLK + g * WAVE + lane];
    };
    if (wid < nblk) {
      const int bl = wid + (nblk - 1 - wid) / NW * NW;       // the wave's last block
      // two register sets in turn (no copy of a register with a load in flight): while one block multiplies, the other set is being filled
      d4 A0[3], B0[3], A1[3], B1[3];
      int n0, J0, n1, J1, pin[4];
      double pv0[4], pv1[4]; bool m0[4], m1[4];
      auto gather = [&](double (&pv)[4], bool (&mk)[4]) {       // P values of the block whose indices sit in pin; then pin moves on
#pragma unroll
        for (int g = 0; g < 4; g++) { pv[g] = valP[max(pin[g], 0)]; mk[g] = pin[g] >= 0; }
      };
      auto finish = [&](const int bb, const d4 (&A)[3], const d4 (&B)[3], const int n, const int J, const double (&pv)[4], const bool (&mk)[4]) {
        d4 acc = {0, 0, 0, 0};
        acc = oc_mm(A[0], B[0], acc);
        if (n > 1) acc = oc_mm(A[1], B[1], acc);
        if (n > 2) acc = oc_mm(A[2], B[2], acc);
        if (n > 3) for (int g = pl.asm_ptr[bb] + 3; g < pl.asm_ptr[bb + 1]; g++) acc = mfma_abt_l(T + (long)pl.asm_a[g] * BLK, T + (long)pl.asm_b[g] * BLK, acc, lane);
#pragma unroll
        for (int g = 0; g < 4; g++) {
          if (mk[g]) acc[g] += pv[g];
          const int row = row0 + 4 * g;
          if (J >= 0 && row == col) acc[g] += cx.R[J * BS + row];
          cx.BL[(long)bb * BLK + row * BS + col] = acc[g];
        }
      };
      fetch_ops(wid, A0, B0, n0, J0); fetch_pidx(wid, pin);
      gather(pv0, m0); fetch_pidx(min(wid + NW, bl), pin);
      for (int b = wid; b < nblk; b += 2 * NW) {
        const int b1 = min(b + NW, bl), b2 = min(b + 2 * NW, bl), b3 = min(b + 3 * NW, bl);     // (past the end: the last block once more, same result)
        fetch_ops(b1, A1, B1, n1, J1); gather(pv1, m1); fetch_pidx(b2, pin);
        finish(b, A0, B0, n0, J0, pv0, m0);
        fetch_ops(b2, A0, B0, n0, J0); gather(pv0, m0); fetch_pidx(b3, pin);
        finish(b1, A1, B1, n1, J1, pv1, m1);
      }
    }
  } else {
  for (int b = wid; b < pl.nblk; b += NW) {
    d4 acc = {0, 0, 0, 0};
    for (int g = pl.asm_ptr[b]; g < pl.asm_ptr[b + 1]; g++) acc = mfma_abt_l(T + (long)pl.asm_a[g] * BLK, T + (long)pl.asm_b[g] * BLK, acc, lane);
    const int J = pl.blk_diag[b];
#pragma unroll
    for (int g = 0; g < 4; g++) {
      const int pi = pl.asm_pidx[(long)b * BLK + g * WAVE + lane];
      if (pi >= 0) acc[g] += valP[pi];
      const int row = row0 + 4 * g;
      if (J >= 0 && row == col) acc[g] += cx.R[J * BS + row];
      cx.BL[(long)b * BLK + row * BS + col] = acc[g];
    }
  }
  }
  bsync<NW>();
#ifdef MPCQP_TIMING
  unsigned long long f2 = __builtin_amdgcn_s_memtime(); cx.fts[1] += f2 - f1;
#endif
  if constexpr (OCM > 0) {
#ifdef MPCQP_TIMING
    const bool okf = oc_ldl<NW, (OCM > 1)>(*oc, octab, cx.BL, scr, cx.RED, wid, lane, &cx.fts[3]);
#else
    const bool okf = oc_ldl<NW, (OCM > 1)>(*oc, octab, cx.BL, scr, cx.RED, wid, lane);
#endif
    if (!okf) return false;
  } else {
  // right-looking block LDL' by elimination-tree levels: G_K = S_KK^-1 (one wave per column of the level);
  // W_IK = S_IK G_K into temp tiles; S_IJ -= W_IK S_JK' (same-destination updates on one wave); the W tiles replace
  // the S_IK slots one phase later, once every update that still needs S_JK has read it.
  int nprev = 0, prev0 = 0;
  for (int lev = 0; lev < rs.nlev; lev++) {
    bool ok = true;
#ifdef MPCQP_TIMING
    const unsigned long long s0_ = __builtin_amdgcn_s_memtime();
#endif
    for (int ci = rs.lv_ptr[lev] + wid; ci < rs.lv_ptr[lev + 1]; ci += NW) {
      // (the in-register inverse of the on-chip mode, kernel_onchip.hpp oc_sweep: 16 rank-1 MFMAs instead of 16 LDS round trips)
      double *blk = cx.BL + (long)rs.lv_diag[ci] * BLK;
      d4 dg = oc_ldD(blk, lane);
      ok = oc_sweep(dg, lane) && ok;
      oc_stD(blk, lane, dg);
    }
#ifdef MPCQP_TIMING
    if (wid == 0) cx.fts[3] += __builtin_amdgcn_s_memtime() - s0_;
#endif
    for (int a = wid; a < nprev; a += NW)
      reinterpret_cast<d4 *>(cx.BL + (long)rs.lw_slot[prev0 + a] * BLK)[lane] = reinterpret_cast<const d4 *>(cx.TMP + (long)a * BLK)[lane];
    if (lane == 0) cx.RED[wid] = ok ? 1.0 : 0.0;
    bsync<NW>();
    bool all_ok = true;
    for (int w = 0; w < NW; w++) all_ok = all_ok && cx.RED[w] != 0.0;
    if (!all_ok) return false;
    const int w0 = rs.lw_ptr[lev], nwk = rs.lw_ptr[lev + 1] - w0;
    for (int a = wid; a < nwk; a += NW) {
      d4 acc = {0, 0, 0, 0};
      acc = mfma_abt_l(cx.BL + (long)rs.lw_slot[w0 + a] * BLK, cx.BL + (long)rs.lw_g[w0 + a] * BLK, acc, lane);
      double *t = cx.TMP + (long)a * BLK;
#pragma unroll
      for (int g = 0; g < 4; g++) t[(row0 + 4 * g) * BS + col] = acc[g];
    }
    bsync<NW>();
    for (int u = rs.lu_ptr[lev * NW + wid]; u < rs.lu_ptr[lev * NW + wid + 1]; u++) {
      d4 acc = {0, 0, 0, 0};
      acc = mfma_abt_l(cx.TMP + (long)rs.lu_tmp[u] * BLK, cx.BL + (long)rs.lu_b[u] * BLK, acc, lane);
      double *dst = cx.BL + (long)rs.lu_dst[u] * BLK;
#pragma unroll
      for (int g = 0; g < 4; g++) dst[(row0 + 4 * g) * BS + col] -= acc[g];
    }
    bsync<NW>();
    nprev = nwk; prev0 = w0;
  }
  for (int a = wid; a < nprev; a += NW)   // (the last level has no off-diagonal block; kept for generality)
    reinterpret_cast<d4 *>(cx.BL + (long)rs.lw_slot[prev0 + a] * BLK)[lane] = reinterpret_cast<const d4 *>(cx.TMP + (long)a * BLK)[lane];
  }
  if (rs.nconst) {   // the constant block -I that folds the partial sums of a split run into their destination
    double *ni = cx.BL + (long)pl.nblk * BLK;
    for (int e = tid; e < BLK; e += NT) ni[e] = (e / BS == e % BS) ? -1.0 : 0.0;
  }
  if constexpr (TAIL) {
  if (rs.tmp_alias) {   // the temp tiles lived in w: its rho vector is gone, recompute it (bit-identical) from the bounds
    for (int i = tid; i < pl.mpad; i += NT) cx.W[i] = i < pl.m ? rho_of(lb[i], ub[i], cx.rho) * cx.Z[i] - cx.Y[i] : 0.0;
  } else {
    for (int i = tid; i < pl.mpad; i += NT) cx.W[i] = cx.W[i] * cx.Z[i] - cx.Y[i];
  }
  }
  bsync<NW>();
#ifdef MPCQP_TIMING
  cx.fts[2] += __builtin_amdgcn_s_memtime() - f2;
#endif
  return true;
}

// GATHERS: the sweeps through ell_chunk_g (the LDS gathers of a batch in flight together; the two-kernel on-chip mode)
template <int NW, bool GATHERS = false>
__device__ __forceinline__ void update_info_res(RCtx &cx, Info &in) {
  const DevPlan &pl = *cx.pl; double *ws = cx.ws; const int wid = cx.wid, lane = cx.lane;
  const double *Dg = ws + pl.o_D, *Eg = ws + pl.o_E;
  const double *valA = ws + pl.o_ellA, *valAt = ws + pl.o_ellAt, *valP = ws + pl.o_ellP;
  const int unscale = cx.unscale;
  double v[15];
#pragma unroll
  for (int k = 0; k < 15; k++) v[k] = 0.0;
  // 0 pr 1 nz 2 nax 3 prs 4 nzs 5 naxs 6 dr 7 nq 8 naty 9 npx 10 drs 11 nqs 12 natys 13 npxs | 14 obj (sum)
  auto rows_A = [&](auto &&f) {
    if constexpr (GATHERS) { for (int c = wid; c < pl.A.nchunks; c += NW) f(c * WAVE + lane, ell_chunk_g<8>(valA, pl.A.idx, cx.X, cx.coA[c], cx.coA[c + 1], lane)); }
    else ell_rows_w<NW>(pl.A, cx.coA, valA, cx.X, wid, lane, f);
  };
  rows_A([&](int i, double ax) {
    if (i < pl.m) {
      const double einv = unscale ? 1.0 / Eg[i] : 1.0, zi = cx.Z[i];
      v[0] = fmax(v[0], fabs(einv * (ax - zi))); v[2] = fmax(v[2], fabs(einv * ax)); v[1] = fmax(v[1], fabs(einv * zi));
      v[3] = fmax(v[3], fabs(ax - zi)); v[5] = fmax(v[5], fabs(ax)); v[4] = fmax(v[4], fabs(zi));
    }
  });
  // P x and A' y land on the same rows for a given wave (both chunked by wid), so no barrier is needed in between
  for (int c = wid; c < pl.P.nchunks; c += NW) {
    const int la = lane;
    const double px = GATHERS ? ell_chunk_g<8>(valP, pl.P.idx, cx.X, cx.coP[c], cx.coP[c + 1], la) : ell_chunk<false>(valP, pl.P.idx, cx.X, cx.coP[c], cx.coP[c + 1], la);
    const double aty = GATHERS ? ell_chunk_g<8>(valAt, pl.At.idx, cx.Y, cx.coAt[c], cx.coAt[c + 1], la) : ell_chunk<false>(valAt, pl.At.idx, cx.Y, cx.coAt[c], cx.coAt[c + 1], la);
    const int t = c * WAVE + lane;
    if (t < pl.npad) {
      const double dinv = unscale ? 1.0 / Dg[t] : 1.0, qv = cx.Q[t], du = qv + px + aty;
      v[6] = fmax(v[6], fabs(dinv * du)); v[7] = fmax(v[7], fabs(dinv * qv)); v[8] = fmax(v[8], fabs(dinv * aty)); v[9] = fmax(v[9], fabs(dinv * px));
      v[10] = fmax(v[10], fabs(du)); v[11] = fmax(v[11], fabs(qv)); v[12] = fmax(v[12], fabs(aty)); v[13] = fmax(v[13], fabs(px));
      v[14] += cx.X[t] * (0.5 * px + qv);
    }
  }
  block_combine<NW, 15, 1>(v, cx.RED, wid, lane);
  in.prim_res = uni(v[0]); in.nz = uni(v[1]); in.nax = uni(v[2]); in.prs = uni(v[3]); in.nzs = uni(v[4]); in.naxs = uni(v[5]);
  in.dual_res = uni(unscale ? cx.cinv * v[6] : v[6]); in.nq = uni(v[7]); in.naty = uni(v[8]); in.npx = uni(v[9]);
  in.drs = uni(v[10]); in.nqs = uni(v[11]); in.natys = uni(v[12]); in.npxs = uni(v[13]);
  in.obj = uni(cx.st->scaling ? cx.cinv * v[14] : v[14]);
}

template <int NW>
__device__ __forceinline__ bool primal_infeasible_res(RCtx &cx, double eps) {
  const DevPlan &pl = *cx.pl; double *ws = cx.ws; const int wid = cx.wid, lane = cx.lane, tid = wid * WAVE + lane; constexpr int NT = NW * WAVE;
  const double *lb = ws + pl.o_l, *ub = ws + pl.o_u, *Eg = ws + pl.o_E, *Dg = ws + pl.o_D, *dy = ws + pl.o_dy;
  double v[2] = {0.0, 0.0};   // 0 nrm (max) 1 lhs (sum)
  for (int i = tid; i < pl.mpad; i += NT) {
    double x = 0.0;
    if (i < pl.m) {
      x = dy[i];
      const double lo = lb[i], up = ub[i];
      if (up > Q_INFTY * Q_MIN_SCALING) { if (lo < -Q_INFTY * Q_MIN_SCALING) x = 0.0; else x = fmin(x, 0.0); }
      else if (lo < -Q_INFTY * Q_MIN_SCALING) x = fmax(x, 0.0);
      v[0] = fmax(v[0], fabs(cx.unscale ? Eg[i] * x : x));
      v[1] += up * fmax(x, 0.0) + lo * fmin(x, 0.0);
    }
    cx.W[i] = x;
  }
  block_combine<NW, 2, 1>(v, cx.RED, wid, lane);
  if (NW == 1) bsync<NW>();
  const double nrm = v[0], lhs = v[1];
  bool res = false;
  if (nrm > eps && lhs < -eps * nrm) {
    double a[1] = {0.0};
    ell_rows_w<NW>(pl.At, cx.coAt, ws + pl.o_ellAt, cx.W, wid, lane, [&](int t, double x) { if (t < pl.npad) a[0] = fmax(a[0], fabs(cx.unscale ? (1.0 / Dg[t]) * x : x)); });
    block_combine<NW, 1, 0>(a, cx.RED, wid, lane);
    res = a[0] < eps * nrm;
  }
  bsync<NW>();
  for (int i = tid; i < pl.mpad; i += NT) cx.W[i] = i < pl.m ? rho_of(lb[i], ub[i], cx.rho) * cx.Z[i] - cx.Y[i] : 0.0;
  bsync<NW>();
  return res;
}

template <int NW>
__device__ __forceinline__ bool dual_infeasible_res(RCtx &cx, double eps) {
  const DevPlan &pl = *cx.pl; double *ws = cx.ws; const int wid = cx.wid, lane = cx.lane, tid = wid * WAVE + lane; constexpr int NT = NW * WAVE;
  const double *lb = ws + pl.o_l, *ub = ws + pl.o_u, *Eg = ws + pl.o_E, *Dg = ws + pl.o_D, *dx = ws + pl.o_dx;
  double v[2] = {0.0, 0.0};   // 0 nrm (max) 1 q'dx (sum)
  for (int t = tid; t < pl.npad; t += NT) {
    const double x = dx[t];
    cx.R[t] = x;
    v[0] = fmax(v[0], fabs(cx.unscale ? Dg[t] * x : x));
    v[1] += cx.Q[t] * x;
  }
  block_combine<NW, 2, 1>(v, cx.RED, wid, lane);
  if (NW == 1) bsync<NW>();
  const double nrm = v[0], qdx = v[1], cs = cx.unscale ? cx.c : 1.0;
  bool res = false;
  if (nrm > eps && qdx < -cs * eps * nrm) {
    double a[1] = {0.0};
    ell_rows_w<NW>(pl.P, cx.coP, ws + pl.o_ellP, cx.R, wid, lane, [&](int t, double x) { if (t < pl.npad) a[0] = fmax(a[0], fabs(cx.unscale ? (1.0 / Dg[t]) * x : x)); });
    block_combine<NW, 1, 0>(a, cx.RED, wid, lane);
    if (a[0] < cs * eps * nrm) {
      double bad[1] = {0.0};
      ell_rows_w<NW>(pl.A, cx.coA, ws + pl.o_ellA, cx.R, wid, lane, [&](int i, double x) {
        if (i < pl.m) {
          if (cx.unscale) x = (1.0 / Eg[i]) * x;
          if ((ub[i] < Q_INFTY * Q_MIN_SCALING && x > eps * nrm) || (lb[i] > -Q_INFTY * Q_MIN_SCALING && x < -eps * nrm)) bad[0] = 1.0;
        }
      });
      block_combine<NW, 1, 0>(bad, cx.RED, wid, lane);
      res = bad[0] == 0.0;
    }
  }
  bsync<NW>();
  return res;
}

template <int NW>
__device__ __forceinline__ int check_termination_res(RCtx &cx, Info &in, int approximate) {
  const mpcqp_settings &st = *cx.st;
  double eps_abs = st.eps_abs, eps_rel = st.eps_rel, epi = st.eps_prim_inf, edi = st.eps_dual_inf;
  if (in.prim_res > Q_INFTY || in.dual_res > Q_INFTY || in.prim_res != in.prim_res || in.dual_res != in.dual_res) { in.obj = NAN; return MPCQP_NON_CVX; }
  if (approximate) { eps_abs *= 10; eps_rel *= 10; epi *= 10; edi *= 10; }
  bool pc = false, dc = false, pic = false, dic = false;
  if (cx.pl->m == 0) pc = true;
  else {
    const double eps_prim = eps_abs + eps_rel * fmax(in.nz, in.nax);
    if (in.prim_res < eps_prim) pc = true; else pic = primal_infeasible_res<NW>(cx, epi);
  }
  {
    double mx = fmax(in.nq, fmax(in.naty, in.npx));
    if (cx.unscale) mx *= cx.cinv;
    const double eps_dual = eps_abs + eps_rel * mx;
    if (in.dual_res < eps_dual) dc = true; else dic = dual_infeasible_res<NW>(cx, edi);
  }
  if (pc && dc) return approximate ? MPCQP_SOLVED_INACCURATE : MPCQP_SOLVED;
  if (pic) { in.obj = Q_INFTY; return approximate ? MPCQP_PRIMAL_INFEASIBLE_INACCURATE : MPCQP_PRIMAL_INFEASIBLE; }
  if (dic) { in.obj = -Q_INFTY; return approximate ? MPCQP_DUAL_INFEASIBLE_INACCURATE : MPCQP_DUAL_INFEASIBLE; }
  return MPCQP_UNSOLVED;
}

// MINW = waves per SIMD the register allocation must leave room for: 1 when the LDS footprint allows only one QP per
// CU anyway (the kernel may then use the whole register file), 2 otherwise
// GB = the factor blocks stay in the per-QP HBM slab (factors that do not fit LDS); LDS then holds only the temp
// tiles, the ADMM vectors and the schedule, and the segment loops keep several blocks in flight.
// OCG > 0 (with GB): the on-chip solve of kernel_onchip.hpp -- the factorisation still works in the slab (its tiles and the temp
// tiles are dead outside it), then the factor is brought on chip: LDS block slots + OCG inverse diagonal blocks and OCH hub blocks
// per wave in registers.
#ifdef MPCQP_TIMING_RUIZ
#define RZ_T0 unsigned long long rz_ = __builtin_amdgcn_s_memtime()
#define RZ_T(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); rzacc[(k) - 4] += t_ - rz_; rz_ = t_; } while (0)
#else
#define RZ_T0
#define RZ_T(k)
#endif
// TL (on-chip instances): the two sweeps of the iteration run on dense tiles of A + the remainder ELL layouts (plan.hpp build_tile_plan)
template <int NW, int MINW, bool GB, bool REUSE, bool ZYG = false, int OCG = 0, int OCH = 0, bool TL = false>
__global__ void __launch_bounds__(NW * WAVE, MINW) mpcqp_res_kernel(const DevPlan pl, const DevRes rs, const mpcqp_settings st, const DevIO io, const DevOc oc) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  constexpr int NT = NW * WAVE;
  constexpr bool OC = OCG > 0;
  static_assert(!OC || (GB && (NW == 4 || NW == 8)), "the on-chip solve is a mode of the 4- and 8-wave global-block kernels");
  static_assert(!TL || OC, "tiles are a mode of the on-chip kernels");
  constexpr int OCU = NW == 4 ? 16 : 8;      // on-chip mode: ELL slots in flight per lane (eight waves split the chunks further and hold more resident blocks)
  constexpr int SPD = MINW == 3 ? 8 : 6;       // factor blocks in flight per wave in the global-block segment loops
  constexpr int EU = OC ? 8 : 16;   // ELL slots in flight per lane in the two sweeps of every iteration (8 for the 128-VGPR instances: 0.5 % slower; the on-chip mode needs the registers)
  const int lane = threadIdx.x & 63;
  const int b = __builtin_amdgcn_readfirstlane(io.order ? io.order[blockIdx.x] : (int)blockIdx.x);
  int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if constexpr (OC && NW == 4) {
    // Which wave plays which part is free.  The two workgroups of a CU put their chain waves (0, 1: the only ones busy in the chain phases of
    // the solve, bound by dependent MFMAs) on different SIMDs: the wave on SIMD s of the workgroup in LDS slot k takes part (s + 2 k) mod 4.
    // HW_ID[5:4] = SIMD, LDS_ALLOC[7:0] = LDS base (0: the CU's first slot).  Only when the four waves do sit on four SIMDs.
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), la = __builtin_amdgcn_s_getreg((31 << 11) | 6);
    const int simd = (hw >> 4) & 3, slot = (la & 0xff) != 0;
    int *xs = reinterpret_cast<int *>(lds);
    if (lane == 0) xs[wid] = simd;
    bsync<NW>();
    const int seen = (1 << xs[0]) | (1 << xs[1]) | (1 << xs[2]) | (1 << xs[3]);
    bsync<NW>();
    if (seen == 15 && !io.no_remap) wid = __builtin_amdgcn_readfirstlane((simd + 2 * slot) & 3);
  }
  const int tid = wid * WAVE + lane;
  RCtx cx;
  cx.pl = &pl; cx.rs = &rs; cx.st = &st; cx.wid = wid; cx.lane = lane;
  cx.fts[0] = cx.fts[1] = cx.fts[2] = cx.fts[3] = 0;
  double *ws = io.ws + (long)b * pl.ws_stride; cx.ws = ws;
  if (GB) { cx.BL = ws + pl.o_Lf; cx.TMP = lds; }
  else { cx.BL = lds; cx.TMP = cx.BL + ((long)pl.nblk + rs.nconst) * BLK; }
  cx.X = lds + rs.stage; cx.Q = cx.X + pl.npad; cx.R = cx.Q + pl.npad;
  double *rend = cx.R + pl.npad + rs.rext;      // the solve vector is followed by the per-wave partial sums of split runs
  if (GB && ZYG) {   // z and y are only ever indexed by their own row: in the slab they cost two coalesced streams per iteration and
    cx.Z = ws + pl.o_Zg; cx.Y = ws + pl.o_Yg; cx.W = rend;   // free 2 * mpad doubles of LDS (one more workgroup per CU for long horizons)
  } else { cx.Z = rend; cx.Y = cx.Z + pl.mpad; cx.W = cx.Y + pl.mpad; }
  if (GB && rs.tmp_alias) cx.TMP = cx.W;
  cx.RB = cx.W + pl.mpad; cx.RED = cx.RB + 16 * NW + 16;
  int4 *segs = reinterpret_cast<int4 *>(cx.RED + 16 * NW);     // (block_combine needs 15 * NW) [2 * n_seg] schedule segments, then [NW + 1] list bounds
  int *lptr = reinterpret_cast<int *>(segs + 2 * rs.n_seg);
  volatile int *octicket = nullptr;                             // on-chip solve: wave 3's ticket for the late rows of the right-hand side
  const int *coAr = nullptr, *coAtr = nullptr;                  // tiles: chunk offsets of the two remainder layouts (LDS)
  double *tiles = nullptr, *valAr = nullptr, *valAtr = nullptr;
  if constexpr (TL) { tiles = ws + oc.tl.o_tile; valAr = ws + oc.tl.o_ellAr; valAtr = ws + oc.tl.o_ellAtr; }
  int *octab = reinterpret_cast<int *>(cx.RED + 16 * NW) + 8;   // on-chip solve: its table (8-byte aligned pairs) instead of schedule segments
  double *ocBL = lds;                                           // ... and the LDS block slots (the temp tiles of the factorisation alias them)
  d4 ocG[OC ? OCG : 1], ocHF[OCH > 0 ? OCH : 1], ocHT[OCH > 0 ? OCH : 1];
  OcLane ocl; OcWave<OC ? OCG : 1> ocw;
  double *valA = ws + pl.o_ellA, *valAt = ws + pl.o_ellAt, *valP = ws + pl.o_ellP;
  double *lb = ws + pl.o_l, *ub = ws + pl.o_u, *Dg = ws + pl.o_D, *Eg = ws + pl.o_E;
  const double *inP = io.P + (long)b * io.sP, *inA = io.A + (long)b * io.sA, *inq = io.q + (long)b * io.sq;
  const double *inl = io.l + (long)b * io.sl, *inu = io.u + (long)b * io.su;
  const int n = pl.n, m = pl.m, npad = pl.npad, mpad = pl.mpad;
  cx.unscale = st.scaling && !st.scaled_termination;

  TS_DECL;
#ifdef MPCQP_TIMING_RUIZ
  unsigned long long rzacc[5] = {0, 0, 0, 0, 0};
#endif
  cx.coA = pl.A.chunk_off; cx.coAt = pl.At.chunk_off; cx.coP = pl.P.chunk_off;
  if constexpr (OC) {
    for (int k = tid; k < oc.o_pos; k += NT) octab[k] = oc.tab[k];      // the chain tables; the rest is only read when the factor is loaded
    int *co = octab + ((oc.o_pos + 1) & ~1);                            // ... and the chunk offsets: a slab round trip less at the head of every chunk
    for (int k = tid; k <= pl.A.nchunks; k += NT) co[k] = pl.A.chunk_off[k];
    for (int k = tid; k <= pl.At.nchunks; k += NT) co[pl.A.nchunks + 1 + k] = pl.At.chunk_off[k];
    for (int k = tid; k <= pl.P.nchunks; k += NT) co[pl.A.nchunks + pl.At.nchunks + 2 + k] = pl.P.chunk_off[k];
    cx.coA = co; cx.coAt = co + pl.A.nchunks + 1; cx.coP = co + pl.A.nchunks + pl.At.nchunks + 2;
    octicket = co + pl.A.nchunks + pl.At.nchunks + pl.P.nchunks + 3;
    if (tid == 0) *octicket = 0;
    if constexpr (TL) {      // ... and of the two remainder layouts, behind the ticket
      int *cr = co + pl.A.nchunks + pl.At.nchunks + pl.P.nchunks + 4;
      for (int k = tid; k <= oc.tl.nAr; k += NT) cr[k] = oc.tl.Ar_off[k];
      for (int k = tid; k <= oc.tl.nAtr; k += NT) cr[oc.tl.nAr + 1 + k] = oc.tl.Atr_off[k];
      coAr = cr; coAtr = cr + oc.tl.nAr + 1;
    }
    ocl = oc_lane(lane);
  } else {
    for (int k = tid; k < 2 * rs.n_seg; k += NT) segs[k] = reinterpret_cast<const int4 *>(rs.g_seg)[k];
    if (tid <= NW) lptr[tid] = rs.g_ptr[tid];
  }
  double c = 1.0;
  int refactor = 1, prev_status = MPCQP_UNSOLVED;
  constexpr bool REUSE_T = REUSE;
  const bool reuse = REUSE_T && io.reuse;
  if (reuse) {
    // ---- kept workspace (mpcqp_update_vectors; OSQP's osqp_update_data_vec): P, A, their scaling D, E, c, the factor and rho
    // stay from the previous solve of this instance; q, l, u are replaced and scaled with the kept D, E, c.  The factor is
    // rebuilt only when a row changed between loose / inequality / equality, because rho_i depends on that class.
    prev_status = io.status[b];
    c = io.cscale[b];
    for (int t = tid; t < npad; t += NT) { cx.Q[t] = 0.0; cx.R[t] = Dg[t]; }
    bsync<NW>();
    for (int j = tid; j < n; j += NT) cx.Q[pl.pos[j]] = inq[j];
    bsync<NW>();
    for (int t = tid; t < npad; t += NT) cx.Q[t] *= c * cx.R[t];
    double chg[1] = {0.0};
    for (int i = tid; i < mpad; i += NT) {
      const double ei = Eg[i];
      const double nl = i < m ? ei * fmax(inl[i], -Q_INFTY) : 0.0, nu = i < m ? ei * fmin(inu[i], Q_INFTY) : 0.0;
      if (i < m) {
        const double ol = lb[i], ou = ub[i];
        const int oc = (ol < -Q_INFTY * Q_MIN_SCALING && ou > Q_INFTY * Q_MIN_SCALING) ? 0 : (ou - ol < Q_RHO_TOL ? 2 : 1);
        const int nc = (nl < -Q_INFTY * Q_MIN_SCALING && nu > Q_INFTY * Q_MIN_SCALING) ? 0 : (nu - nl < Q_RHO_TOL ? 2 : 1);
        if (oc != nc) chg[0] = 1.0;
      }
      lb[i] = nl; ub[i] = nu;
    }
    block_combine<NW, 1, 0>(chg, cx.RED, wid, lane);
    refactor = chg[0] != 0.0;
    c = uni(c); cx.c = c; cx.cinv = uni(1.0 / c);
    if (!GB) {   // the factor blocks come back from the slab
      const double *src = ws + pl.o_Lf;
      for (long e = tid; e < ((long)pl.nblk + rs.nconst) * BLK; e += NT) lds[e] = src[e];
    }
    bsync<NW>();
  } else {
    // ---- load: caller's CSC values -> ELL arrays. The block region of LDS is idle until the factorisation, so the
    // ELL values of A, A', P live there for the whole scaling phase (host guarantees they fit) and are written to
    // the HBM slab once, already scaled.
    // (on-chip mode: the block slots hold A when it fits -- oc.a_lds, decided by the host -- and A', P are read from the slab)
    const bool a_lds = OC && oc.a_lds;
    const bool p_lds = OC && oc.p_lds;
    double *sA = GB ? (a_lds ? lds : valA) : lds, *sAt = GB ? valAt : sA + pl.A.entries, *sP = GB ? (p_lds ? lds + pl.A.entries : valP) : sAt + pl.At.entries;
    for (long e = tid; e < pl.A.entries; e += NT) { const int s = pl.A.src[e]; sA[e] = s >= 0 ? inA[s] : 0.0; }
    if (!GB) for (long e = tid; e < pl.At.entries; e += NT) { const int s = pl.At.src[e]; sAt[e] = s >= 0 ? inA[s] : 0.0; }   // (global-block kernels gather A' from the caller's array when they scale it)
    for (long e = tid; e < pl.P.entries; e += NT) { const int s = pl.P.src[e]; sP[e] = s >= 0 ? inP[s] : 0.0; }
    for (int t = tid; t < npad; t += NT) { cx.Q[t] = 0.0; cx.R[t] = 1.0; cx.X[t] = 0.0; }
    for (int i = tid; i < mpad; i += NT) cx.W[i] = 1.0;
    bsync<NW>();
    for (int j = tid; j < n; j += NT) cx.Q[pl.pos[j]] = inq[j];
    bsync<NW>();

    TS(0);
    // ---- modified Ruiz equilibration: D in R, E in W; X = the column-norm accumulators of the sweep over A; nP = max_k |P_tk| d_k of the
    // current D is needed twice -- for this pass's column norm and, with the updated D, for the cost scaling, which is also the next
    // pass's value -- and swept once per pass (it lives in y, which is idle until the iteration starts; in the slab when m < n)
    c = 1.0;
    double *nPv = mpad >= npad ? cx.Y : ws + pl.o_dx;
    if (st.scaling > 0) ell_rowmax_w<NW>(pl.P, cx.coP, sP, cx.R, wid, lane, [&](int t, double x) { if (t < npad) nPv[t] = x; });
    for (int it = 0; it < st.scaling; it++) {
      RZ_T0;
      for (int ch = wid; ch < pl.A.nchunks; ch += NW) {
        const int i = ch * WAVE + lane;
        const double ei = i < mpad ? cx.W[i] : 0.0;
        const double v = ell_chunk_rc(sA, pl.A.idx, cx.R, ei, cx.X, cx.coA[ch], cx.coA[ch + 1], lane);
        if (i < mpad) cx.W[i] = ei * (1.0 / sqrt(limit_scaling(ei * v)));       // (e_i is read by its own lane only: updated in place)
      }
      RZ_T(4);
      bsync<NW>();
      RZ_T(5);
      for (int t = tid; t < npad; t += NT) {       // (the thread that wrote nPv[t])
        const double dj = cx.R[t];
        cx.R[t] = dj * (1.0 / sqrt(limit_scaling(fmax(c * dj * nPv[t], dj * cx.X[t]))));
        cx.X[t] = 0.0;
      }
      bsync<NW>();
      RZ_T(6);
      double v[2] = {0.0, 0.0};   // 0 qn (max) 1 sum
      ell_rowmax_w<NW>(pl.P, cx.coP, sP, cx.R, wid, lane, [&](int t, double x) { if (t < npad) { nPv[t] = x; v[1] += c * cx.R[t] * x; v[0] = fmax(v[0], fabs(c * cx.R[t] * cx.Q[t])); } });
      RZ_T(7);
      block_combine<NW, 2, 1>(v, cx.RED, wid, lane);
      const double ct = 1.0 / limit_scaling(fmax(v[1] / (double)n, limit_scaling(v[0])));
      c *= ct;
      bsync<NW>();
      RZ_T(8);
    }
    c = uni(c); cx.c = c; cx.cinv = uni(1.0 / c);
    TS(1);
    // scale and write out: A <- E A D, A' likewise, P <- c D P D (coalesced stores of whole 512 B slots; the loads of up to 8 slots in flight)
    for (int ch = wid; ch < pl.A.nchunks; ch += NW) {
      const int i = ch * WAVE + lane; const double ei = i < mpad ? cx.W[i] : 0.0;
      ell_map_chunk<false>(sA, pl.A.idx, nullptr, pl.A.idx, valA, cx.coA[ch], cx.coA[ch + 1], lane, [&](double v, int j) { return v * (ei * cx.R[j]); });
    }
    for (int ch = wid; ch < pl.At.nchunks; ch += NW) {
      const int t = ch * WAVE + lane; const double dj = t < npad ? cx.R[t] : 0.0;
      // (global-block kernels gather A' from the caller's array here)
      ell_map_chunk<GB>(sAt, pl.At.src, inA, pl.At.idx, valAt, cx.coAt[ch], cx.coAt[ch + 1], lane, [&](double v, int i) { return v * (dj * cx.W[i]); });
      ell_map_chunk<false>(sP, pl.P.idx, nullptr, pl.P.idx, valP, cx.coP[ch], cx.coP[ch + 1], lane, [&](double v, int k) { return v * (c * dj * cx.R[k]); });
    }
    if constexpr (TL) {
      // the same scaled numbers once more in the layouts of the iteration's sweeps: dense tiles ([lane][K] order; element (r, c) of tile t is
      // row rowid[16 t + r], position 16 tJ[t] + c) and the two remainder layouts, gathered from the caller's array
      const DevTile &tl = oc.tl;
      for (long e = tid; e < (long)tl.ntile * BLK; e += NT) {
        const int sidx = tl.tsrc[e];
        double v = 0.0;
        if (sidx >= 0) {
          const int t = (int)(e >> 8), lt = (int)(e >> 2) & 63, K = (int)e & 3;
          const int i = tl.rowid[t * BS + (lt & 15)], j = BS * tl.tJ[t] + (lt >> 4) + 4 * K;
          v = inA[sidx] * (cx.W[i] * cx.R[j]);
        }
        tiles[e] = v;
      }
      for (int ch = wid; ch < tl.nAr; ch += NW) {
        const int i = ch * WAVE + lane; const double ei = i < mpad ? cx.W[i] : 0.0;
        ell_map_chunk<true>(valAr, tl.Ar_src, inA, tl.Ar_idx, valAr, coAr[ch], coAr[ch + 1], lane, [&](double v, int j) { return v * (ei * cx.R[j]); });
      }
      for (int ch = wid; ch < tl.nAtr; ch += NW) {
        const int t = ch * WAVE + lane; const double dj = t < npad ? cx.R[t] : 0.0;
        ell_map_chunk<true>(valAtr, tl.Atr_src, inA, tl.Atr_idx, valAtr, coAtr[ch], coAtr[ch + 1], lane, [&](double v, int i) { return v * (dj * cx.W[i]); });
      }
    }
    bsync<NW>();
    for (int t = tid; t < npad; t += NT) { cx.Q[t] *= c * cx.R[t]; Dg[t] = cx.R[t]; }
    for (int i = tid; i < mpad; i += NT) {
      const double ei = cx.W[i];
      Eg[i] = ei;
      lb[i] = i < m ? ei * fmax(inl[i], -Q_INFTY) : 0.0;
      ub[i] = i < m ? ei * fmin(inu[i], Q_INFTY) : 0.0;
    }
  }
  for (int t = tid; t < npad; t += NT) cx.X[t] = 0.0;
  for (int i = tid; i < mpad; i += NT) { cx.Z[i] = 0.0; cx.Y[i] = 0.0; }
  bsync<NW>();
  if (st.warm_start && io.x0 && io.y0) {
    for (int j = tid; j < n; j += NT) { const int t = pl.pos[j]; cx.X[t] = io.x0[(long)b * n + j] * (1.0 / Dg[t]); }
    for (int i = tid; i < m; i += NT) cx.Y[i] = io.y0[(long)b * m + i] * (1.0 / Eg[i]) * c;
    bsync<NW>();
    ell_rows_w<NW>(pl.A, cx.coA, valA, cx.X, wid, lane, [&](int i, double ax) { if (i < m) cx.Z[i] = ax; });
    bsync<NW>();
  }
  // a kept factor belongs to the rho it was built with: that instance's final rho of the previous solve
  cx.rho = uni(reuse ? io.info[4L * b + 3] : fmin(fmax(io.rho0 && io.rho0[b] > 0.0 ? io.rho0[b] : st.rho, Q_RHO_MIN), Q_RHO_MAX));
  int status = MPCQP_UNSOLVED, iter_done = 0;
  Info in; memset(&in, 0, sizeof(in));
  TS(2);
  bool ok = !(reuse && prev_status == MPCQP_NON_CVX);
  if (ok && refactor) ok = factorize_res<NW, (OC ? (OCH > 0 ? 2 : 1) : 0)>(cx, &oc, octab, ocBL);
  else if (ok) {   // kept factor: only w = rho z - y, which the factorisation leaves behind otherwise
    for (int i = tid; i < mpad; i += NT) cx.W[i] = i < m ? rho_of(lb[i], ub[i], cx.rho) * cx.Z[i] - cx.Y[i] : 0.0;
    bsync<NW>();
  }
  if (!ok) status = MPCQP_NON_CVX;
  if constexpr (OC) {
    if (ok) {
      ocw = oc_wave<NW, OCG, OCH>(oc, oc.tab, wid, npad);
      oc_load_factor<NW, OCG, OCH>(oc, oc.tab, ws + pl.o_Lf, ocBL, ocl, ocG, ocHF, ocHT, wid, lane);
    }
  }
  TS(3);

  int interval = st.adaptive_rho_interval;
  if (st.adaptive_rho && interval == 0) interval = st.check_termination ? 4 * st.check_termination : 100;
  const double alpha = st.alpha, sigma = st.sigma;
  double *dxg = ws + pl.o_dx, *dyg = ws + pl.o_dy;
  int can_check = 0;
  const int sq0 = OC ? 0 : lptr[wid], sq1 = OC ? 0 : lptr[wid + 1];
  const int ni_off = rs.nconst ? pl.nblk * BLK * 8 : -1;      // byte offset of the constant -I block (split accumulation runs)
  // on-chip mode: the rows of the right-hand side that waves 2 / 3 compute while the chains run (kernel_onchip.hpp oc_solve)
  auto late_rows = [&](const int w) {
    const int c = w == 3 ? oc.at_poll : oc.at_free;
    if (c >= 0) {
      const int t = c * WAVE + lane;
      const double v = ell_chunk<false, OCU>(valAt, pl.At.idx, cx.W, cx.coAt[c], cx.coAt[c + 1], lane);
      if (t < npad) cx.R[t] = sigma * cx.X[t] - cx.Q[t] + v;
    }
  };
  // ... and what they do while the chains run backwards: pull the values of A, l, u -- streamed right after the solve -- and of A' -- at
  // the start of the next iteration -- into L2 (every iteration re-reads them, and 512 resident QPs x 90 KB do not stay in L2 by themselves)
  auto idle_touch = [&](const int w) {
    if (io.no_touch) return;
    if constexpr (TL) {
      if (w == 2) oc_touch_pinned(tiles, (long)oc.tl.ntile * BLK * 8, lane);
      else if (NW == 4 || w == 3) { oc_touch_pinned(lb, (long)mpad * 8, lane); oc_touch_pinned(ub, (long)mpad * 8, lane); oc_touch_pinned(valAr, oc.tl.Ar_entries * 8, lane); oc_touch_pinned(valAtr, oc.tl.Atr_entries * 8, lane); }
      return;
    }
    if constexpr (NW == 8) {      // (the eight-wave instances are new in round 3: pinned form; off unless MPCQP_TOUCH8)
      if (w == 2) oc_touch_pinned(valA, pl.A.entries * 8, lane);
      else if (w == 3) { oc_touch_pinned(lb, (long)mpad * 8, lane); oc_touch_pinned(ub, (long)mpad * 8, lane); oc_touch_pinned(valAt, pl.At.entries * 8, lane); }
      return;
    }
    if (w == 2) oc_touch_pinned(valA, pl.A.entries * 8, lane);
    else if (NW == 4 || w == 3) { oc_touch_pinned(lb, (long)mpad * 8, lane); oc_touch_pinned(ub, (long)mpad * 8, lane); oc_touch_pinned(valAt, pl.At.entries * 8, lane); }
  };
  OcTileRec trec;
  if constexpr (TL) trec = oc_tile_records<NW>(oc.tl, pl.A.nchunks, pl.At.nchunks, wid, lane);      // this wave's chunk records, for the whole solve
  if (ok) {
    int iter;
    for (iter = 1; iter <= st.max_iter; iter++) {
      if constexpr (TL) {     // tiles: per chunk the four column blocks' tile products into R, then the remainder layout on top
#pragma unroll
        for (int k = 0; k < OC_TILE_MAXT; k++) {
          const int c = wid + k * NW;
          if (c < pl.At.nchunks)
            oc_tiles_at<4>(oc.tl, trec.t[k], tiles, valAtr, coAtr, cx.W, cx.R, c, pl.nb, ocl, lane, [&](const int t, const double v) { if (t < npad) cx.R[t] = sigma * cx.X[t] - cx.Q[t] + (v + cx.R[t]); });
        }
      } else
      if constexpr (OC) {     // (the chunks named in oc.at_poll / oc.at_free are computed during the chain phase of the solve: oc_solve)
        for (int c = wid; c < pl.At.nchunks; c += NW) if (c != oc.at_poll && c != oc.at_free) {
          const int t = c * WAVE + lane;
          const double v = ell_chunk<false, OCU>(valAt, pl.At.idx, cx.W, cx.coAt[c], cx.coAt[c + 1], lane);
          if (t < npad) cx.R[t] = sigma * cx.X[t] - cx.Q[t] + v;
        }
      } else
      ell_rows_w<NW, EU>(pl.At, cx.coAt, valAt, cx.W, wid, lane, [&](int t, double v) { if (t < npad) cx.R[t] = sigma * cx.X[t] - cx.Q[t] + v; });
      for (int t = tid; t < rs.rext; t += NT) cx.R[npad + t] = 0.0;
      bsync<NW>();
      TS(4);
      if constexpr (OC) {
#ifdef MPCQP_TIMING
        if constexpr (NW == 4 && !TL) oc_solve<NW, OCG, OCH, (OCH > 0)>(oc, octab, ocBL, cx.R, npad, ocl, ocw, ocG, ocHF, ocHT, wid, octicket, iter, late_rows, idle_touch, ts_acc + 9);   // slots 9..11: F1, F2 + F3, B1 (B2 = the rest of the solve)
        else oc_solve_long<NW, OCG, OCH, (OCH > 0)>(oc, octab, ocBL, cx.R, npad, ocl, ocw, ocG, ocHF, ocHT, wid, octicket, iter, late_rows, idle_touch, oc_chain_info<NW>(oc, octab, wid, lane), ts_acc + 9);
#else
        if constexpr (NW == 4 && !TL) oc_solve<NW, OCG, OCH, (OCH > 0)>(oc, octab, ocBL, cx.R, npad, ocl, ocw, ocG, ocHF, ocHT, wid, octicket, iter, late_rows, idle_touch);
        else oc_solve_long<NW, OCG, OCH, (OCH > 0)>(oc, octab, ocBL, cx.R, npad, ocl, ocw, ocG, ocHF, ocHT, wid, octicket, iter, late_rows, idle_touch, oc_chain_info<NW>(oc, octab, wid, lane));
#endif
      } else {
#ifdef MPCQP_TIMING
        long long *trace = (b == 0 && wid == 0 && iter == 3 && io.dbg) ? io.dbg + 16L * gridDim.x : nullptr;
        if (trace) trace[0] = (long long)__builtin_amdgcn_s_memtime();
        run_schedule<NW, GB, SPD>(segs, sq0, sq1, reinterpret_cast<const char *>(cx.BL), reinterpret_cast<char *>(cx.R), lane, ni_off, trace);
        if (trace) trace[1] = (long long)__builtin_amdgcn_s_memtime();
#else
        run_schedule<NW, GB, SPD>(segs, sq0, sq1, reinterpret_cast<const char *>(cx.BL), reinterpret_cast<char *>(cx.R), lane, ni_off);
#endif
      }
      if (NW == 1) bsync<NW>();
      TS(5);
      can_check = st.check_termination && (iter % st.check_termination == 0);
      const int do_rho = st.adaptive_rho && interval && (iter % interval == 0);
      const int save = can_check || do_rho;
      {
        // ztilde = A xtilde fused with relaxation, projection onto [l, u], dual update and w = rho z - y.
        // l, u are fetched before the row sum is accumulated; rho_i and 1/rho_i are selected from the three values
        // the rho rule can produce (no per-row division).
        const double rho_eq = uni(Q_RHO_EQ * cx.rho), ri_min = 1.0 / Q_RHO_MIN, ri_eq = uni(1.0 / rho_eq), ri_in = uni(1.0 / cx.rho);
        auto row_update = [&](const int i, const double lo, const double up, const double zo, const double yp, const double zt) {
          if (i < m) {
            const bool loose = lo < -Q_INFTY * Q_MIN_SCALING && up > Q_INFTY * Q_MIN_SCALING, eq = up - lo < Q_RHO_TOL;
            const double rh = loose ? Q_RHO_MIN : (eq ? rho_eq : cx.rho), rinv = loose ? ri_min : (eq ? ri_eq : ri_in);
            const double zr = alpha * zt + (1.0 - alpha) * ((GB && ZYG) ? zo : cx.Z[i]), yo = (GB && ZYG) ? yp : cx.Y[i];
            const double zn = fmin(fmax(zr + rinv * yo, lo), up);
            const double dy = rh * (zr - zn), yn = yo + dy;
            cx.Z[i] = zn; cx.Y[i] = yn; cx.W[i] = rh * zn - yn;
            if (save) dyg[i] = dy;
          }
        };
        if constexpr (TL) {     // tiles: the rows of a chunk that lie in a tile get that part through w (dead until row_update rewrites it)
#pragma unroll
          for (int k = 0; k < OC_TILE_MAXA; k++) {
            const int c = wid + k * NW;
            if (c < pl.A.nchunks) {
              const int i = c * WAVE + lane;
              const double lo = lb[i], up = ub[i];
              const double zo = (ZYG && i < mpad) ? cx.Z[i] : 0.0, yp = (ZYG && i < mpad) ? cx.Y[i] : 0.0;
              bool tiled;
              const double ze = oc_tiles_a<2>(oc.tl, trec.a[k], tiles, valAr, coAr, cx.R, cx.W, c, ocl, lane, &tiled);
              row_update(i, lo, up, zo, yp, tiled ? ze + cx.W[i] : ze);
            }
          }
        } else
        for (int c = wid; c < pl.A.nchunks; c += NW) {
          const int i = c * WAVE + lane;
          const double lo = lb[i], up = ub[i];
          // z, y of this row as well when they live in the slab: their latency hides behind the row sum like that of l, u
          const double zo = (GB && ZYG && i < mpad) ? cx.Z[i] : 0.0, yp = (GB && ZYG && i < mpad) ? cx.Y[i] : 0.0;   // (the last chunk may run past mpad)
          const double zt = ell_chunk<false, (OC ? OCU : EU)>(valA, pl.A.idx, cx.R, cx.coA[c], cx.coA[c + 1], lane);
          row_update(i, lo, up, zo, yp, zt);
        }
      }
      bsync<NW>();     // every wave has finished reading xtilde (R) as the gather source before X/R move on
      if (__builtin_expect(save, 0)) {     // (its own loop: the address of dx stayed live across the iteration otherwise -- spilled, and reloaded here behind a full wait)
        for (int t = tid; t < npad; t += NT) { const double xo = cx.X[t]; dxg[t] = (alpha * cx.R[t] + (1.0 - alpha) * xo) - xo; }
      }
      for (int t = tid; t < npad; t += NT) cx.X[t] = alpha * cx.R[t] + (1.0 - alpha) * cx.X[t];
      bsync<NW>();
      TS(6);
      iter_done = iter;
      if (__builtin_expect(can_check, 0)) {     // (rare paths are marked cold: their register pressure must not cost the hot loop its registers)
        update_info_res<NW>(cx, in);
        status = check_termination_res<NW>(cx, in, 0);
        TS(7);
        if (status != MPCQP_UNSOLVED) break;
      }
      if (__builtin_expect(do_rho, 0)) {
        if (!can_check) update_info_res<NW>(cx, in);
        const double pr = in.prs / (fmax(in.nzs, in.naxs) + Q_DIV_TOL);
        const double dr = in.drs / (fmax(in.nqs, fmax(in.natys, in.npxs)) + Q_DIV_TOL);
        double rn = cx.rho * sqrt(pr / (dr + Q_DIV_TOL));
        rn = fmin(fmax(rn, Q_RHO_MIN), Q_RHO_MAX);
        if (rn > cx.rho * st.adaptive_rho_tolerance || rn < cx.rho / st.adaptive_rho_tolerance) {
          cx.rho = uni(rn);
          if (!factorize_res<NW, (OC ? (OCH > 0 ? 2 : 1) : 0)>(cx, &oc, octab, ocBL)) { status = MPCQP_NON_CVX; break; }
          if constexpr (OC) oc_load_factor<NW, OCG, OCH>(oc, oc.tab, ws + pl.o_Lf, ocBL, ocl, ocG, ocHF, ocHT, wid, lane);
        }
      }
    }
    if (iter > st.max_iter) iter_done = st.max_iter;
    if (status == MPCQP_UNSOLVED) {
      if (!can_check) { update_info_res<NW>(cx, in); status = check_termination_res<NW>(cx, in, 0); }
      if (status == MPCQP_UNSOLVED) { status = check_termination_res<NW>(cx, in, 1); if (status == MPCQP_UNSOLVED) status = MPCQP_MAX_ITER_REACHED; }
    }
  }
  const bool bad = status == MPCQP_PRIMAL_INFEASIBLE || status == MPCQP_PRIMAL_INFEASIBLE_INACCURATE ||
                   status == MPCQP_DUAL_INFEASIBLE || status == MPCQP_DUAL_INFEASIBLE_INACCURATE || status == MPCQP_NON_CVX;
  for (int j = tid; j < n; j += NT) { const int t = pl.pos[j]; io.x[(long)b * n + j] = bad ? NAN : Dg[t] * cx.X[t]; }
  for (int i = tid; i < m; i += NT) {
    io.y[(long)b * m + i] = bad ? NAN : cx.cinv * Eg[i] * cx.Y[i];
    io.z[(long)b * m + i] = bad ? NAN : (1.0 / Eg[i]) * cx.Z[i];
  }
  if (!GB && io.keep) {   // resident variants: park the factor in the slab for a following mpcqp_update_vectors solve
    double *dst = ws + pl.o_Lf;
    for (long e = tid; e < ((long)pl.nblk + rs.nconst) * BLK; e += NT) dst[e] = lds[e];
  }
  if (tid == 0) {
    io.status[b] = status; io.iters[b] = iter_done;
    io.info[4L * b] = in.obj; io.info[4L * b + 1] = in.prim_res; io.info[4L * b + 2] = in.dual_res; io.info[4L * b + 3] = cx.rho;
    io.cscale[b] = c;
  }
  TS(8);
#ifdef MPCQP_TIMING
  ts_acc[12] = cx.fts[0]; ts_acc[13] = cx.fts[1]; ts_acc[14] = cx.fts[2]; ts_acc[15] = cx.fts[3];
#ifdef MPCQP_TIMING_RUIZ
  ts_acc[9] = rzacc[0]; ts_acc[10] = rzacc[1]; ts_acc[11] = rzacc[2]; ts_acc[12] = rzacc[3]; ts_acc[13] = rzacc[4];
#endif
#endif
  TS_STORE(io.dbg);
}
