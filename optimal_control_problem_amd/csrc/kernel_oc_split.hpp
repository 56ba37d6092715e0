// kernel_oc_split.hpp -- the on-chip mode as TWO kernels, one per call of the reference's seam:
//   mpcqp_oc_setup_kernel  = CuCaQP::initSolver (reference src/sqp_solver/CuCaQP.cpp:183-197 -> osqp_setup): load the caller's CSC values, modified
//                            Ruiz equilibration, scaled ELL copies, rho vector, assembly of M, block LDL' on the matrix cores (oc_ldl); the
//                            factor lands in the slab exactly where oc_load_factor reads it
//   mpcqp_oc_admm_kernel   = CuCaQP::solve (:199-211 -> osqp_solve): factor brought on chip (LDS + registers), ADMM iterations, termination /
//                            infeasibility tests, adaptive rho (re-factorisation in place, a rare path), un-scaled outputs
// Same stream, same dispatch order, one launch each per mpcqp_solve.  Why two: in the single kernel (kernel_resident.hpp mpcqp_res_kernel<..., OCG,
// OCH>) the set-up and the iteration shared one register allocation -- 654 - 1,139 spilled VGPRs, 464 - 980 B of scratch per lane in the three
// BASELINE instances -- and nothing could be done for one phase without moving the other's schedule.  Here each phase has its own allocation, and the
// set-up, in which no factor block is resident yet, may use every wave and register of the workgroup.
// What crosses from one kernel to the other, per QP, through the slab: the scaled ELL values, l, u, D, E (as before), the scaled q (the slab's Lb
// region, which the resident kernels never used), c (DevIO.cscale), rho (DevIO.info[3]) and the set-up's verdict (DevIO.status: UNSOLVED = go on,
// NON_CVX = M is not positive definite).  The arithmetic, and therefore every result bit, is that of the single kernel.
//
// Adaptive rho.  A new rho means a new factor: rare (every 100th iteration at most, and only when the residual ratio moved by more than a factor
// of five), but inlined into the iteration kernel it costs that kernel its register allocation (600 - 900 of its 700 - 1,100 spilled VGPRs).  So the
// iteration kernel <RF = 0> LEAVES when rho changes: it parks x, z, y in the slab, the iteration number in DevIO.iters, the new rho in DevIO.info[3]
// and marks the instance OC_PENDING.  mpcqp_solve queues, unconditionally and on the same stream: the set-up kernel in `resume` mode (only the
// re-factorisation, only for marked instances -- every other workgroup returns at once), the iteration kernel in resume mode, and, so that any
// number of rho updates is served, a last pair whose iteration kernel <RF = 1> re-factorises in place.  A launch of workgroups that return at
// once costs ~10 us; no instance of the MPC workloads ever reaches the last pair.
// Part of the kernel translation units (included through kernels_all.hpp, in order; not a stand-alone header).
#pragma once

constexpr int OC_PENDING = 100, OC_PENDING_NONCVX = 101;      // DevIO.status between the launches of one mpcqp_solve (never seen by the caller)
// LDS carve-up shared by both kernels (the same as the single kernel's on-chip mode, so that one footprint -- plan.hpp lds_bytes_oc -- serves both)
template <int NW>
struct OcLds {
  double *X, *Q, *R, *Z, *Y, *W, *RB, *RED;
  int *octab, *co;
};
template <int NW, bool SETUP = false>
__device__ __forceinline__ OcLds<NW> oc_lds(double *lds, const DevPlan &pl, const DevRes &rs, const DevOc &oc) {
  OcLds<NW> L;
  L.X = lds + rs.stage; L.Q = L.X + pl.npad; L.R = SETUP ? L.Q : L.Q + pl.npad;      // (SETUP: q lives in the slab, where the iteration kernel picks it up)
  double *rend = L.R + pl.npad + rs.rext;
  // (SETUP: the set-up kernel's own vector layout -- it never touches z, whose region holds its 16-bit index tables and is oc.zpad doubles long, and keeps
  // one n-vector of the Ruiz passes in y)
  L.Z = rend; L.Y = L.Z + (SETUP ? oc.zpad : pl.mpad); L.W = L.Y + (SETUP ? pl.npad : pl.mpad);
  L.RB = L.W + pl.mpad; L.RED = L.RB + 16 * NW + 16;
  L.octab = reinterpret_cast<int *>(L.RED + 16 * NW) + 8;
  L.co = L.octab + ((oc.o_pos + 1) & ~1);
  return L;
}
// Which wave plays which part is free.  The two workgroups of a CU put their chain waves (0, 1: the only ones busy in the chain phases of the solve
// and of oc_ldl, bound by dependent MFMAs) on different SIMDs: the wave on SIMD s of the workgroup in LDS slot k takes part (s + 2 k) mod 4.
// HW_ID[5:4] = SIMD, LDS_ALLOC[7:0] = LDS base (0: the CU's first slot).  Only when the four waves do sit on four SIMDs.
__device__ __forceinline__ int oc_wave_role4(double *lds, int wid, const int lane, const int no_remap) {
  const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), la = __builtin_amdgcn_s_getreg((31 << 11) | 6);
  const int simd = (hw >> 4) & 3, slot = (la & 0xff) != 0;
  int *xs = reinterpret_cast<int *>(lds);
  if (lane == 0) xs[wid] = simd;
  bsync<4>();
  const int seen = (1 << xs[0]) | (1 << xs[1]) | (1 << xs[2]) | (1 << xs[3]);
  bsync<4>();
  if (seen == 15 && !no_remap) wid = __builtin_amdgcn_readfirstlane((simd + 2 * slot) & 3);
  return wid;
}
// chain tables and chunk offsets into LDS (every chunk of every sweep starts by reading two offsets: a slab round trip less each)
template <int NW>
__device__ __forceinline__ void oc_tables_to_lds(const DevPlan &pl, const DevOc &oc, RCtx &cx, const OcLds<NW> &L, const int tid) {
  constexpr int NT = NW * WAVE;
  for (int k = tid; k < oc.o_pos; k += NT) L.octab[k] = oc.tab[k];
  int *co = L.co;
  for (int k = tid; k <= pl.A.nchunks; k += NT) co[k] = pl.A.chunk_off[k];
  for (int k = tid; k <= pl.At.nchunks; k += NT) co[pl.A.nchunks + 1 + k] = pl.At.chunk_off[k];
  for (int k = tid; k <= pl.P.nchunks; k += NT) co[pl.A.nchunks + pl.At.nchunks + 2 + k] = pl.P.chunk_off[k];
  cx.coA = co; cx.coAt = co + pl.A.nchunks + 1; cx.coP = co + pl.A.nchunks + pl.At.nchunks + 2;
}

// This wave's chunk offsets of an ELL structure in the lanes of ONE register -- lane 2 k: first slot of its k-th chunk (chunk wid + k NW), lane 2 k + 1:
// its end -- read once before the ADMM loop and picked out with v_readlane: the offsets used to come from LDS at the head of every chunk, a round trip
// of ~120 cycles in front of the chunk's loads, six times per iteration on the north-star size.
template <int NW>
__device__ __forceinline__ int oc_my_chunks(const int *__restrict__ chunk_off, const int nchunks, const int wid, const int lane) {
  const int c = wid + (lane >> 1) * NW;
  return c < nchunks ? chunk_off[c + (lane & 1)] : 0;
}
// a workgroup-uniform double the compiler may not look through (it re-derived 1 / rho_i per lane and chunk from the rho it was selected from:
// a twelve-instruction division sequence in every row update)
__device__ __forceinline__ double opaque_uni(const double v) {
  const long long b = __double_as_longlong(v);
  unsigned lo = (unsigned)(b & 0xffffffffll), hi = (unsigned)((unsigned long long)b >> 32);
  asm volatile("" : "+v"(lo), "+v"(hi));
  lo = __builtin_amdgcn_readfirstlane(lo); hi = __builtin_amdgcn_readfirstlane(hi);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
#ifdef MPCQP_TIMING
// (the two kernels share a QP's row of 16 slots: each adds its own)
#define TS_STORE_ADD(ptr, k0, k1) do { if (tid == 0 && (ptr)) { for (int k_ = (k0); k_ < (k1); k_++) (ptr)[16L * b + k_] += (long long)ts_acc[k_]; \
    if (b == 0) { (ptr)[16L * gridDim.x + 126] += (long long)(__builtin_amdgcn_s_memtime() - ts_first); (ptr)[16L * gridDim.x + 127] += (long long)(__builtin_amdgcn_s_memrealtime() - ts_rt0); } } } while (0)
#else
#define TS_STORE_ADD(ptr, k0, k1)
#endif

// =========================================================================================================
// Set-up: what osqp_setup does per QP.  REUSE = the kept-workspace entry (mpcqp_update_vectors; OSQP's osqp_update_data_vec): P, A, D, E, c, rho and
// the factor stay, q, l, u are replaced and scaled with the kept D, E, c; the factor is rebuilt only when a row changed its class (rho_i depends on it).
// =========================================================================================================
template <int NW, bool REUSE, bool HUB>
__global__ void __launch_bounds__(NW * WAVE, 2) mpcqp_oc_setup_kernel(const DevPlan pl, const DevRes rs, const mpcqp_settings st, const DevIO io, const DevOc oc) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  constexpr int NT = NW * WAVE;
  const int lane = threadIdx.x & 63;
  const int b = __builtin_amdgcn_readfirstlane(io.order ? io.order[blockIdx.x] : (int)blockIdx.x);
  int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if constexpr (NW == 4) wid = oc_wave_role4(lds, wid, lane, io.no_remap);
  const int tid = wid * WAVE + lane;
  const OcLds<NW> L = oc_lds<NW, true>(lds, pl, rs, oc);
  RCtx cx;
  cx.pl = &pl; cx.rs = &rs; cx.st = &st; cx.wid = wid; cx.lane = lane;
  cx.fts[0] = cx.fts[1] = cx.fts[2] = cx.fts[3] = 0;
  double *ws = io.ws + (long)b * pl.ws_stride; cx.ws = ws;
  cx.BL = ws + pl.o_Lf; cx.TMP = lds;
  cx.X = L.X; cx.Q = L.Q; cx.R = L.R; cx.Z = L.Z; cx.Y = L.Y; cx.W = L.W; cx.RB = L.RB; cx.RED = L.RED;
  double *valA = ws + pl.o_ellA, *valAt = ws + pl.o_ellAt, *valP = ws + pl.o_ellP;
  double *lb = ws + pl.o_l, *ub = ws + pl.o_u, *Dg = ws + pl.o_D, *Eg = ws + pl.o_E, *Qs = ws + pl.o_Lb;
  const double *inP = io.P + (long)b * io.sP, *inA = io.A + (long)b * io.sA, *inq = io.q + (long)b * io.sq;
  const double *inl = io.l + (long)b * io.sl, *inu = io.u + (long)b * io.su;
  const int n = pl.n, m = pl.m, npad = pl.npad, mpad = pl.mpad;
  cx.unscale = st.scaling && !st.scaled_termination;
  TS_DECL;
#ifdef MPCQP_TIMING_RUIZ
  unsigned long long rzacc[5] = {0, 0, 0, 0, 0};
#endif
  if (oc.resume && io.status[b] != OC_PENDING) return;      // (resume mode: only the instances that left the iteration kernel for a new factor)
  oc_tables_to_lds<NW>(pl, oc, cx, L, tid);
  if (oc.resume) {
    bsync<NW>();
    cx.rho = uni(io.info[4L * b + 3]);
    cx.c = 1.0; cx.cinv = 1.0;
    const bool okr = factorize_res<NW, (HUB ? 2 : 1), false>(cx, &oc, L.octab, lds);
    if (tid == 0) io.status[b] = okr ? OC_PENDING : OC_PENDING_NONCVX;
    return;
  }
  double c = 1.0;
  int refactor = 1, prev_status = MPCQP_UNSOLVED;
  const bool reuse = REUSE && io.reuse;
  if (reuse) {
    prev_status = io.status[b];
    c = io.cscale[b];
    for (int t = tid; t < npad; t += NT) { if (pl.perm[t] < 0) Qs[t] = 0.0; cx.R[t] = Dg[t]; }      // (q in place in the slab: padding positions here, the variables' below)
    bsync<NW>();
    for (int j = tid; j < n; j += NT) { const int t = pl.pos[j]; Qs[t] = inq[j] * (c * cx.R[t]); }
    double chg[1] = {0.0};
    for (int i = tid; i < mpad; i += NT) {
      const double ei = Eg[i];
      const double nl = i < m ? ei * fmax(inl[i], -Q_INFTY) : 0.0, nu = i < m ? ei * fmin(inu[i], Q_INFTY) : 0.0;
      if (i < m) {
        const double ol = lb[i], ou = ub[i];
        const int oc_ = (ol < -Q_INFTY * Q_MIN_SCALING && ou > Q_INFTY * Q_MIN_SCALING) ? 0 : (ou - ol < Q_RHO_TOL ? 2 : 1);
        const int nc = (nl < -Q_INFTY * Q_MIN_SCALING && nu > Q_INFTY * Q_MIN_SCALING) ? 0 : (nu - nl < Q_RHO_TOL ? 2 : 1);
        if (oc_ != nc) chg[0] = 1.0;
      }
      lb[i] = nl; ub[i] = nu;
    }
    block_combine<NW, 1, 0>(chg, cx.RED, wid, lane);
    refactor = chg[0] != 0.0;
    c = uni(c); cx.c = c; cx.cinv = uni(1.0 / c);
    bsync<NW>();
  } else {
    // ---- load: caller's CSC values -> ELL arrays.  The block slots of LDS are idle until the factor comes on chip (in the other kernel), so the
    // ELL values of A (and P behind them) live there for the whole scaling phase when they fit (oc.a_lds / oc.p_lds, decided by the host) and
    // are written to the slab once, already scaled; A' is gathered from the caller's array when it is scaled.
    // (the staged values through pointers whose memory the compiler KNOWS: `a_lds ? lds : valA` is a pointer to either, every access through it a
    // flat load that takes both memory paths and waits for both counters -- the whole scaling phase ran on those; one copy of the phase per case)
    const bool a_lds = oc.a_lds, p_lds = oc.p_lds;
    auto scale_phase = [&](double *sA, double *sP, auto iA, auto iP) __attribute__((always_inline)) {
    // (eight source indices, then the eight values they point at, in flight at a time: one element per trip was two dependent round trips to memory each)
    auto gather8 = [&](const int *__restrict__ src, const double *__restrict__ in, double *dst, const long entries) __attribute__((always_inline)) {
      long e = tid;
      for (; e + 7L * NT < entries; e += 8L * NT) {
        int sr[8]; double v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) sr[u] = src[e + (long)u * NT];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = in[max(sr[u], 0)];
#pragma unroll
        for (int u = 0; u < 8; u++) dst[e + (long)u * NT] = sr[u] >= 0 ? v[u] : 0.0;
      }
      for (; e < entries; e += NT) { const int sr = src[e]; dst[e] = sr >= 0 ? in[sr] : 0.0; }
    };
    gather8(pl.A.src, inA, sA, pl.A.entries);
    gather8(pl.P.src, inP, sP, pl.P.entries);
    // (the ten passes gather through these: a read from LDS instead of a round trip to the L2 in front of every batch)
    if constexpr (sizeof(*iA) == 2) { for (long e = tid; e < pl.A.entries; e += NT) iA[e] = (unsigned short)pl.A.idx[e]; }
    if constexpr (sizeof(*iP) == 2) { for (long e = tid; e < pl.P.entries; e += NT) iP[e] = (unsigned short)pl.P.idx[e]; }
    for (int t = tid; t < npad; t += NT) { if (pl.perm[t] < 0) Qs[t] = 0.0; cx.R[t] = 1.0; cx.X[t] = 0.0; }      // (q in place in the slab: one workgroup writes it, padding here)
    for (int i = tid; i < mpad; i += NT) cx.W[i] = 1.0;
    bsync<NW>();
    for (int j = tid; j < n; j += NT) Qs[pl.pos[j]] = inq[j];
    bsync<NW>();
    TS(0);
    // ---- modified Ruiz equilibration: D in R, E in W; X = the column-norm accumulators of the sweep over A (one sweep by rows gives the row norm,
    // lane-local, and the column norms, LDS atomic max on the bit pattern); nP = max_k |P_tk| d_k is needed twice per pass -- before and after D
    // is updated -- and swept once (the second value is the next pass's first; it lives in the y region, n doubles long in this kernel)
    c = 1.0;
    double *nPv = cx.Y;
    if (st.scaling > 0) { for (int ch = wid; ch < pl.P.nchunks; ch += NW) { const int t = ch * WAVE + lane; const double x = ell_chunk_mx(sP, iP, cx.R, cx.coP[ch], cx.coP[ch + 1], lane); if (t < npad) nPv[t] = x; } }
    // Two barriers per pass: the cost scale's reduction (one max, one sum over the columns of P) is finished by every wave for itself behind the NEXT pass's
    // first barrier -- the partial results wait in RED meanwhile; c is first needed by the D update.  The chunks of A start at the wave where those of P
    // end, so that the pair of sweeps between two barriers is balanced.
    const int shA = pl.P.nchunks % NW;
    auto cost_scale = [&]() {
      double r0 = cx.RED[0], r1 = cx.RED[1];
      for (int w = 1; w < NW; w++) { r0 = fmax(r0, cx.RED[2 * w]); r1 = r1 + cx.RED[2 * w + 1]; }
      c *= 1.0 / limit_scaling(fmax(r1 / (double)n, limit_scaling(r0)));
    };
    for (int it = 0; it < st.scaling; it++) {
      RZ_T0;
      for (int ch = (wid + NW - shA) % NW; ch < pl.A.nchunks; ch += NW) {
        const int i = ch * WAVE + lane;
        const double ei = i < mpad ? cx.W[i] : 0.0;
        const double v = ell_chunk_rc(sA, iA, cx.R, ei, cx.X, cx.coA[ch], cx.coA[ch + 1], lane);
        if (i < mpad) cx.W[i] = ei * (1.0 / sqrt(limit_scaling(ei * v)));       // (e_i is read by its own lane only: updated in place)
      }
      RZ_T(4);
      bsync<NW>();
      RZ_T(5);
      if (it > 0) cost_scale();
      for (int t = tid; t < npad; t += NT) {       // (the thread that wrote nPv[t])
        const double dj = cx.R[t];
        cx.R[t] = dj * (1.0 / sqrt(limit_scaling(fmax(c * dj * nPv[t], dj * cx.X[t]))));
        cx.X[t] = 0.0;
      }
      bsync<NW>();
      RZ_T(6);
      double v[2] = {0.0, 0.0};   // 0 qn (max) 1 sum
      for (int ch = wid; ch < pl.P.nchunks; ch += NW) {
        const int t = ch * WAVE + lane;
        const double x = ell_chunk_mx(sP, iP, cx.R, cx.coP[ch], cx.coP[ch + 1], lane);
        if (t < npad) { nPv[t] = x; v[1] += c * cx.R[t] * x; v[0] = fmax(v[0], fabs(c * cx.R[t] * Qs[t])); }
      }
      v[0] = wave_max(v[0]); v[1] = wave_sum(v[1]);
      if (lane == 0) { cx.RED[2 * wid] = v[0]; cx.RED[2 * wid + 1] = v[1]; }
      RZ_T(7);
    }
    if (st.scaling > 0) { bsync<NW>(); cost_scale(); bsync<NW>(); }
    c = uni(c); cx.c = c; cx.cinv = uni(1.0 / c);
    TS(1);
    // scale and write out: A <- E A D, A' likewise (gathered from the caller's array), P <- c D P D (coalesced stores of whole 512 B slots; up to 8 slots in flight)
    for (int ch = wid; ch < pl.A.nchunks; ch += NW) {
      const int i = ch * WAVE + lane; const double ei = i < mpad ? cx.W[i] : 0.0;
      ell_map_chunk<false>(sA, pl.A.idx, nullptr, pl.A.idx, valA, cx.coA[ch], cx.coA[ch + 1], lane, [&](double v, int j) { return v * (ei * cx.R[j]); });
    }
    for (int ch = wid; ch < pl.At.nchunks; ch += NW) {
      const int t = ch * WAVE + lane; const double dj = t < npad ? cx.R[t] : 0.0;
      ell_map_chunk<true>(valAt, pl.At.src, inA, pl.At.idx, valAt, cx.coAt[ch], cx.coAt[ch + 1], lane, [&](double v, int i) { return v * (dj * cx.W[i]); });
      ell_map_chunk<false>(sP, pl.P.idx, nullptr, pl.P.idx, valP, cx.coP[ch], cx.coP[ch + 1], lane, [&](double v, int k) { return v * (c * dj * cx.R[k]); });
    }
    if (oc.tl.on) {
      // the same scaled numbers once more in the layouts of the tile sweeps (experiment, MPCQP_VTILES=1): dense 16 x 16 tiles, row-major (element (r, c)
      // of tile t is row rowid[16 t + r], position 16 tJ[t] + c), and the two remainder ELL layouts, gathered from the caller's array
      const DevTile &tl = oc.tl;
      double *tiles = ws + tl.o_tile, *valAr = ws + tl.o_ellAr, *valAtr = ws + tl.o_ellAtr;
      for (long e = tid; e < (long)tl.ntile * BLK; e += NT) {
        const int sidx = tl.tsrc[e];
        double v = 0.0;
        if (sidx >= 0) {
          const int t = (int)(e >> 8), r = (int)(e >> 4) & 15, cc = (int)e & 15;
          v = inA[sidx] * (cx.W[tl.rowid[t * BS + r]] * cx.R[BS * tl.tJ[t] + cc]);
        }
        tiles[e] = v;
      }
      for (int ch = wid; ch < tl.nAr; ch += NW) {
        const int i = ch * WAVE + lane; const double ei = i < mpad ? cx.W[i] : 0.0;
        ell_map_chunk<true>(valAr, tl.Ar_src, inA, tl.Ar_idx, valAr, tl.Ar_off[ch], tl.Ar_off[ch + 1], lane, [&](double v, int j) { return v * (ei * cx.R[j]); });
      }
      for (int ch = wid; ch < tl.nAtr; ch += NW) {
        const int t = ch * WAVE + lane; const double dj = t < npad ? cx.R[t] : 0.0;
        ell_map_chunk<true>(valAtr, tl.Atr_src, inA, tl.Atr_idx, valAtr, tl.Atr_off[ch], tl.Atr_off[ch + 1], lane, [&](double v, int i) { return v * (dj * cx.W[i]); });
      }
    }
    bsync<NW>();
    for (int t = tid; t < npad; t += NT) { Qs[t] *= c * cx.R[t]; Dg[t] = cx.R[t]; }
    for (int i = tid; i < mpad; i += NT) {
      const double ei = cx.W[i];
      Eg[i] = ei;
      lb[i] = i < m ? ei * fmax(inl[i], -Q_INFTY) : 0.0;
      ub[i] = i < m ? ei * fmin(inu[i], Q_INFTY) : 0.0;
    }
    };
    unsigned short *iA16 = reinterpret_cast<unsigned short *>(lds) + oc.ixo_a, *iP16 = reinterpret_cast<unsigned short *>(lds) + oc.ixo_p;
    if (a_lds && p_lds && oc.ix16 == 3) scale_phase(lds, lds + pl.A.entries, iA16, iP16);
    else if (a_lds && p_lds && oc.ix16 == 1) scale_phase(lds, lds + pl.A.entries, iA16, pl.P.idx);
    else if (a_lds && p_lds) scale_phase(lds, lds + pl.A.entries, pl.A.idx, pl.P.idx);
    else if (a_lds) scale_phase(lds, valP, pl.A.idx, pl.P.idx);
    // (values in the slab: the index tables alone -- a quarter less to read per pass, and the gathers' addresses come from LDS; quadrotor N=50 set-up 8.05 -> 6.93 ms with A's)
    else if (oc.ix16 == 3) scale_phase(valA, valP, iA16, iP16);
    else if (oc.ix16 == 1) scale_phase(valA, valP, iA16, pl.P.idx);
    else scale_phase(valA, valP, pl.A.idx, pl.P.idx);
  }
  bsync<NW>();
  // a kept factor belongs to the rho it was built with: that instance's final rho of the previous solve
  cx.rho = uni(reuse ? io.info[4L * b + 3] : fmin(fmax(io.rho0 && io.rho0[b] > 0.0 ? io.rho0[b] : st.rho, Q_RHO_MIN), Q_RHO_MAX));
  TS(2);
  bool ok = !(reuse && prev_status == MPCQP_NON_CVX);
  if (ok && refactor) ok = factorize_res<NW, (HUB ? 2 : 1), false>(cx, &oc, L.octab, lds);
  if (tid == 0) {
    io.status[b] = ok ? MPCQP_UNSOLVED : MPCQP_NON_CVX; io.iters[b] = 0;
    io.info[4L * b + 3] = cx.rho;
    io.cscale[b] = c;
  }
  TS(3);
#ifdef MPCQP_TIMING
  ts_acc[12] = cx.fts[0]; ts_acc[13] = cx.fts[1]; ts_acc[14] = cx.fts[2]; ts_acc[15] = cx.fts[3];
#ifdef MPCQP_TIMING_RUIZ
  ts_acc[9] = rzacc[0]; ts_acc[10] = rzacc[1]; ts_acc[11] = rzacc[2]; ts_acc[12] = rzacc[3]; ts_acc[13] = rzacc[4];
#endif
#endif
  TS_STORE(io.dbg);
}

// =========================================================================================================
// Iteration: what osqp_solve does per QP, on the factor the set-up kernel left in the slab.
// =========================================================================================================
// RF = 1: an adaptive-rho step re-factorises in place (the last launch of a solve); RF = 0: the instance leaves for the set-up kernel's resume mode
// TL: the two sweeps of the iteration on dense tiles of A + remainder ELL layouts (experiment; below)
// One instance; called once per workgroup (a grid of `count` workgroups) or, with DevIO.queue set, again and again by a resident workgroup that draws tickets.
template <int NW, int OCG, int OCH, int RF, bool TL, bool PAIRS>
__device__ __forceinline__ void oc_admm_one(const DevPlan &pl, const DevRes &rs, const mpcqp_settings &st, const DevIO &io, const DevOc &oc, double *lds, const int b, const int wid, const int lane) {
  constexpr int NT = NW * WAVE;
  [[maybe_unused]] constexpr int OCU = NW == 4 ? 16 : 8;      // ELL slots in flight per lane (eight waves split the chunks further and hold more resident blocks)
  constexpr bool HUB = OCH > 0;
  const int tid = wid * WAVE + lane;
  const OcLds<NW> L = oc_lds<NW>(lds, pl, rs, oc);
  RCtx cx;
  cx.pl = &pl; cx.rs = &rs; cx.st = &st; cx.wid = wid; cx.lane = lane;
  cx.fts[0] = cx.fts[1] = cx.fts[2] = cx.fts[3] = 0;
  double *ws = io.ws + (long)b * pl.ws_stride; cx.ws = ws;
  cx.BL = ws + pl.o_Lf; cx.TMP = lds;
  cx.X = L.X; cx.Q = L.Q; cx.R = L.R; cx.Z = L.Z; cx.Y = L.Y; cx.W = L.W; cx.RB = L.RB; cx.RED = L.RED;
  int *octab = L.octab;
  double *ocBL = lds;
  d4 ocG[OCG], ocHF[OCH > 0 ? OCH : 1], ocHT[OCH > 0 ? OCH : 1];
  OcLane ocl; OcWave<OCG> ocw;
  double *valA = ws + pl.o_ellA, *valAt = ws + pl.o_ellAt;
  double *lb = ws + pl.o_l, *ub = ws + pl.o_u, *Dg = ws + pl.o_D, *Eg = ws + pl.o_E;
  double *Qs = ws + pl.o_Lb, *Xs = Qs + pl.npad, *Zs = ws + pl.o_Zg, *Ys = ws + pl.o_Yg;     // the scaled q; x, z, y of an instance that left for a new factor
  const int n = pl.n, m = pl.m, npad = pl.npad, mpad = pl.mpad;
  cx.unscale = st.scaling && !st.scaled_termination;
  int status = io.status[b], iter0 = 0;
  const bool resume = oc.resume != 0;
  if (resume && status != OC_PENDING && status != OC_PENDING_NONCVX) return;      // (not waiting for this launch)
  TS_DECL;
  oc_tables_to_lds<NW>(pl, oc, cx, L, tid);
  ocl = oc_lane(lane);
  const double c = uni(io.cscale[b]); cx.c = c; cx.cinv = uni(1.0 / c);
  if (resume) {
    iter0 = io.iters[b];
    for (int t = tid; t < npad; t += NT) { cx.Q[t] = Qs[t]; cx.X[t] = Xs[t]; }
    for (int i = tid; i < mpad; i += NT) { cx.Z[i] = Zs[i]; cx.Y[i] = Ys[i]; }
  } else {
    for (int t = tid; t < npad; t += NT) { cx.Q[t] = Qs[t]; cx.X[t] = 0.0; }
    for (int i = tid; i < mpad; i += NT) { cx.Z[i] = 0.0; cx.Y[i] = 0.0; }
  }
  bsync<NW>();
  if (!resume && st.warm_start && io.x0 && io.y0) {
    for (int j = tid; j < n; j += NT) { const int t = pl.pos[j]; cx.X[t] = io.x0[(long)b * n + j] * (1.0 / Dg[t]); }
    for (int i = tid; i < m; i += NT) cx.Y[i] = io.y0[(long)b * m + i] * (1.0 / Eg[i]) * c;
    bsync<NW>();
    ell_rows_w<NW>(pl.A, cx.coA, valA, cx.X, wid, lane, [&](int i, double ax) { if (i < m) cx.Z[i] = ax; });
    bsync<NW>();
  }
  cx.rho = uni(io.info[4L * b + 3]);
  int iter_done = iter0;
  Info in; memset(&in, 0, sizeof(in));
  if (resume) { in.obj = io.info[4L * b]; in.prim_res = io.info[4L * b + 1]; in.dual_res = io.info[4L * b + 2]; }      // (what a failed re-factorisation reports: the last check's)
  const bool ok = status != MPCQP_NON_CVX && status != OC_PENDING_NONCVX;
  status = ok ? MPCQP_UNSOLVED : MPCQP_NON_CVX;
  if (ok) {
    // w = rho z - y (what a factorisation leaves behind in the single kernel: the same product, bit for bit)
    for (int i = tid; i < mpad; i += NT) cx.W[i] = i < m ? rho_of(lb[i], ub[i], cx.rho) * cx.Z[i] - cx.Y[i] : 0.0;
    bsync<NW>();
    ocw = oc_wave<NW, OCG, OCH>(oc, oc.tab, wid, npad);
    oc_load_factor<NW, OCG, OCH>(oc, oc.tab, ws + pl.o_Lf, ocBL, ocl, ocG, ocHF, ocHT, wid, lane);
  }
  TS(3);

  int interval = st.adaptive_rho_interval;
  if (st.adaptive_rho && interval == 0) interval = st.check_termination ? 4 * st.check_termination : 100;
  const double alpha = st.alpha, sigma = st.sigma;
  double *dxg = ws + pl.o_dx, *dyg = ws + pl.o_dy;
  int can_check = 0;
  auto late_rows = [&](const int) {};
  // what waves 2, 3 do while the chains run backwards: pull the values of A, l, u -- streamed right after the solve -- and of A' -- at the start of
  // the next iteration -- into L2 (every iteration re-reads them, and 512 resident QPs x 90 KB do not stay in L2 by themselves)
  auto idle_touch = [&](const int w) {
    if (io.no_touch) return;
    if (w == 2) oc_touch_pinned(valA, pl.A.entries * 8, lane);
    else if (w == 3) { oc_touch_pinned(lb, (long)mpad * 8, lane); oc_touch_pinned(ub, (long)mpad * 8, lane); oc_touch_pinned(valAt, pl.At.entries * 8, lane); }
  };
  [[maybe_unused]] const OcChain occ = oc_chain_info<NW>(oc, octab, wid, lane);      // (after the tables are in LDS)
  const int myAt = oc_my_chunks<NW>(TL ? oc.tl.Atr_off : pl.At.chunk_off, pl.At.nchunks, wid, lane);      // (a wave has at most 32 chunks of either: the host checks)
  int myA = oc_my_chunks<NW>(TL ? oc.tl.Ar_off : pl.A.chunk_off, pl.A.nchunks, wid, lane);
  [[maybe_unused]] int myAid = wid + lane * NW;      // lane k: the wave's k-th row chunk of A
  if constexpr (NW == 8 && !TL) {
    if (oc.a_assign) {
      const int ck = oc.a_assign[wid * 32 + min(lane >> 1, 31)];
      myA = (ck >= 0 && ck < pl.A.nchunks) ? pl.A.chunk_off[ck + (lane & 1)] : 0;
      myAid = oc.a_assign[wid * 32 + min(lane, 31)];
    }
    if (myAid >= pl.A.nchunks || lane >= 32) myAid = -1;
  }
  // TL: this wave's tile records, read once into the lanes of registers (picked out with v_readlane inside the sweeps: no table access in front of a chunk's loads).
  //   A' sweep, lane 4 k + u: the tile of column block u of the wave's k-th chunk (or the zero tile) and the first of its sixteen rows of w
  //   A sweep,  lane 8 k + u: the u-th tile with a row in the wave's k-th chunk {tile, column block, first row, rows}; lane k of taCnt: how many; bit k of
  //             tiled: this lane's row of that chunk gets a tile contribution
  int ttId = 0, ttFirst = 0, taId = 0, taJ = 0, taFirst = 0, taRows = 0, taCnt = 0, tiled = 0;
  const double *tiles = nullptr, *valAr = nullptr, *valAtr = nullptr;
  if constexpr (TL) {
    const DevTile &tl = oc.tl;
    tiles = ws + tl.o_tile; valAr = ws + tl.o_ellAr; valAtr = ws + tl.o_ellAtr;
    const int cT = wid + (lane >> 2) * NW, cA = wid + (lane >> 3) * NW;
    ttId = cT < pl.At.nchunks ? tl.tt_info[16 * cT + 4 * (lane & 3)] : tl.ntile; ttFirst = cT < pl.At.nchunks ? tl.tt_info[16 * cT + 4 * (lane & 3) + 1] : 0;
    const bool okA = cA < pl.A.nchunks;
    taId = okA ? tl.ta_info[32 * cA + 4 * (lane & 7)] : tl.ntile; taJ = okA ? tl.ta_info[32 * cA + 4 * (lane & 7) + 1] : 0;
    taFirst = okA ? tl.ta_info[32 * cA + 4 * (lane & 7) + 2] : 0; taRows = okA ? tl.ta_info[32 * cA + 4 * (lane & 7) + 3] : 0;
    taCnt = (lane < 8 && wid + lane * NW < pl.A.nchunks) ? tl.ta_cnt[wid + lane * NW] : 0;
#pragma unroll
    for (int k = 0; k < 8; k++) { const int c = wid + k * NW; if (c < pl.A.nchunks) tiled |= (int)((tl.ta_mask[c] >> lane) & 1ull) << k; }
  }
  // rho_i and 1 / rho_i of a row are selected from the three values the rho rule can produce (no per-row division); they change with rho only
  double rho_in = 0, rho_eq = 0, ri_eq = 0, ri_in = 0; const double ri_min = 1.0 / Q_RHO_MIN;
  auto rho_constants = [&]() { rho_in = opaque_uni(cx.rho); rho_eq = opaque_uni(Q_RHO_EQ * cx.rho); ri_eq = opaque_uni(1.0 / (Q_RHO_EQ * cx.rho)); ri_in = opaque_uni(1.0 / cx.rho); };
  rho_constants();
  if (ok) {
    int iter;
    for (iter = iter0 + 1; iter <= st.max_iter; iter++) {
#ifdef MPCQP_TIMING_SWEEP
      // diagnostic build: where the time of the A' sweep goes, wave by wave (first chunk of the wave, up to 16 slots): issue of the loads, wait for
      // them, gathers + arithmetic + store, the rest of the wave's chunks, the barrier; slots 9 .. 13 of the QP's row, wave 0 (and wave NW - 1 in 14, 15)
      {
        const unsigned long long q0 = __builtin_amdgcn_s_memtime();
        const int ch0 = wid;
        const int s0 = cx.coAt[ch0], s1 = cx.coAt[ch0 + 1];
        double v[16]; int ix[16];
        const long e0 = (long)s0 * WAVE + lane;
#pragma unroll
        for (int u = 0; u < 16; u++) { const int o = max(min(u, s1 - s0 - 1), 0) * WAVE; v[u] = valAt[e0 + o]; ix[u] = pl.At.idx[e0 + o]; }      // (unconditional: a load under a branch makes the compiler wait for it at the join)
        const unsigned long long q1 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long q2 = __builtin_amdgcn_s_memtime();
        double a = 0.0;
#pragma unroll
        for (int u = 0; u < 16; u++) a += (s0 + u < s1 ? v[u] : 0.0) * cx.W[ix[u]];
        if (s1 - s0 > 16) a += ell_chunk<false, OCU>(valAt, pl.At.idx, cx.W, s0 + 16, s1, lane);
        { const int t = ch0 * WAVE + lane; if (t < npad) cx.R[t] = sigma * cx.X[t] - cx.Q[t] + a; }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned long long q3 = __builtin_amdgcn_s_memtime();
        for (int ch = wid + NW; ch < pl.At.nchunks; ch += NW) {
          const int t = ch * WAVE + lane;
          const double vv = ell_chunk<false, OCU>(valAt, pl.At.idx, cx.W, cx.coAt[ch], cx.coAt[ch + 1], lane);
          if (t < npad) cx.R[t] = sigma * cx.X[t] - cx.Q[t] + vv;
        }
        const unsigned long long q4 = __builtin_amdgcn_s_memtime();
        if (wid == 0) { ts_acc[9] += q1 - q0; ts_acc[10] += q2 - q1; ts_acc[11] += q3 - q2; ts_acc[12] += q4 - q3; }
        if (wid == NW - 1) { cx.fts[2] += q2 - q0; cx.fts[3] += q4 - q2; }
      }
#else
      for (int ch = wid; ch < pl.At.nchunks; ch += NW) {
        const int t = ch * WAVE + lane;
        const int k = (ch - wid) / NW;
        const double xt = t < npad ? cx.X[t] : 0.0, qt = t < npad ? cx.Q[t] : 0.0;      // (read while the loads fly)
        if constexpr (TL) {
          // the four column blocks of the chunk: their tiles read TRANSPOSED (lane (c = l >> 2, j = l & 3) takes entries [4 j .. 4 j + 3][c]: each load instruction
          // four whole 128-byte lines), requested before the remainder layout's slots; the tile sums reach the variables' rows of R behind the owner's write
          const int tc = lane >> 2, tj = 4 * (lane & 3);
          d4 ta[4]; int first[4];
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const double *T = tiles + (long)__builtin_amdgcn_readlane(ttId, 4 * k + u) * BLK + tj * BS + tc;
            ta[u] = d4{T[0], T[BS], T[2 * BS], T[3 * BS]};
            first[u] = __builtin_amdgcn_readlane(ttFirst, 4 * k + u);
          }
          const double v = ell_chunk_g<8>(valAtr, oc.tl.Atr_idx, cx.W, __builtin_amdgcn_readlane(myAt, 2 * k), __builtin_amdgcn_readlane(myAt, 2 * k + 1), lane);
          if (t < npad) cx.R[t] = v + __builtin_fma(sigma, xt, -qt);
          double ts[4];
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const double *wp = cx.W + first[u] + tj;
            ts[u] = oc_quad_sum(oc_dot4(ta[u], d4{wp[0], wp[1], wp[2], wp[3]}, 0.0));
          }
#pragma unroll
          for (int u = 0; u < 4; u++) if ((lane & 3) == 0 && 4 * ch + u < pl.nb) cx.R[BS * (4 * ch + u) + tc] += ts[u];
        } else {
        const double v = ell_chunk_g<OCU>(valAt, pl.At.idx, cx.W, __builtin_amdgcn_readlane(myAt, 2 * k), __builtin_amdgcn_readlane(myAt, 2 * k + 1), lane);
        if (t < npad) cx.R[t] = v + __builtin_fma(sigma, xt, -qt);      // (written out: which products the compiler fuses in a * b - c + d depends on the order it meets them in)
        }
      }
#endif
      for (int t = tid; t < rs.rext; t += NT) cx.R[npad + t] = 0.0;
      bsync<NW>();
      TS(4);
#if defined(MPCQP_TIMING) && !defined(MPCQP_TIMING_SWEEP) && !defined(MPCQP_TIMING_RUIZ)
      unsigned long long *const stamps = ts_acc + 9;   // slots 9..11: F1, F2 + F3 + B1, the barrier behind them (B2 = the rest of the solve)
#else
      unsigned long long *const stamps = nullptr;
#endif
#ifndef MPCQP_VALU_CHAINS     // (experiment, -DMPCQP_VALU_CHAINS: the chains on the vector ALUs too -- oc_solve_v: parity-green, 4 - 20 % slower)
      if constexpr (NW == 4 && !PAIRS) oc_solve<NW, OCG, OCH, HUB>(oc, octab, ocBL, cx.R, npad, ocl, ocw, ocG, ocHF, ocHT, wid, nullptr, iter, late_rows, idle_touch, stamps);
      else oc_solve_long<NW, OCG, OCH, HUB>(oc, octab, ocBL, cx.R, npad, ocl, ocw, ocG, ocHF, ocHT, wid, nullptr, iter, late_rows, idle_touch, occ, stamps);
#else
      oc_solve_v<NW, OCG, OCH, HUB>(oc, octab, ocBL, cx.R, npad, ocl, ocw, ocG, ocHF, ocHT, wid, idle_touch, stamps);
#endif
      TS(5);
      can_check = st.check_termination && (iter % st.check_termination == 0);
      const int do_rho = st.adaptive_rho && interval && (iter % interval == 0);
      const int save = can_check || do_rho;
      {
        // ztilde = A xtilde fused with relaxation, projection onto [l, u], dual update and w = rho z - y.  l, u, z, y of the row are fetched before
        // the row sum is accumulated.
        // (eight-wave instances: the wave's row chunks come from a list the host balanced by load batches -- ten wide chunks of the quadrotor's N=50 on eight
        // waves left two waves with seven round trips where six are enough; rows are independent, any wave may take any chunk.  Four waves: chunk wid + k NW)
        constexpr bool LPT = NW == 8 && !TL;
        for (int kq = 0, ch = LPT ? __builtin_amdgcn_readlane(myAid, 0) : wid; LPT ? (kq < 32 && ch >= 0) : ch < pl.A.nchunks;
             ++kq, ch = LPT ? __builtin_amdgcn_readlane(myAid, min(kq, 31)) : ch + NW) {
          const int i = ch * WAVE + lane, k = LPT ? kq : (ch - wid) / NW;
          const double lo = lb[i], up = ub[i];
          const double zo = i < mpad ? cx.Z[i] : 0.0, yo = i < mpad ? cx.Y[i] : 0.0;
          double zt;
          if constexpr (TL) {
            // the tiles with a row in this chunk, as they lie (one 32-byte row piece per lane), four in flight; a tile sum reaches its row's owner through w,
            // which is dead until the row update rewrites it (a tile whose rows straddle two chunks is computed for both; each wave keeps its own rows)
            const int cnt = __builtin_amdgcn_readlane(taCnt, k), tr = lane >> 2, tj = 4 * (lane & 3);
#pragma unroll
            for (int g = 0; g < 2; g++) {
              if (cnt > 4 * g) {
                d4 ta[4]; d4 xv[4]; int rid[4]; bool mine[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                  const int l8 = 8 * k + 4 * g + u;
                  ta[u] = reinterpret_cast<const d4 *>(tiles + (long)__builtin_amdgcn_readlane(taId, l8) * BLK)[lane];
                  const double *xp = cx.R + BS * __builtin_amdgcn_readlane(taJ, l8) + tj;
                  xv[u] = d4{xp[0], xp[1], xp[2], xp[3]};
                  rid[u] = __builtin_amdgcn_readlane(taFirst, l8) + tr;
                  mine[u] = tr < __builtin_amdgcn_readlane(taRows, l8) && (rid[u] >> 6) == ch && (lane & 3) == 0;
                }
#pragma unroll
                for (int u = 0; u < 4; u++) { const double sv = oc_quad_sum(oc_dot4(ta[u], xv[u], 0.0)); if (mine[u]) cx.W[rid[u]] = sv; }
              }
            }
            zt = ell_chunk_g<8>(valAr, oc.tl.Ar_idx, cx.R, __builtin_amdgcn_readlane(myA, 2 * k), __builtin_amdgcn_readlane(myA, 2 * k + 1), lane);
            if ((tiled >> k) & 1) zt += cx.W[i];
          } else
          zt = ell_chunk_g<OCU>(valA, pl.A.idx, cx.R, __builtin_amdgcn_readlane(myA, 2 * k), __builtin_amdgcn_readlane(myA, 2 * k + 1), lane);
          if (i < m) {
            const bool loose = lo < -Q_INFTY * Q_MIN_SCALING && up > Q_INFTY * Q_MIN_SCALING, eq = up - lo < Q_RHO_TOL;
            const double rh = loose ? Q_RHO_MIN : (eq ? rho_eq : rho_in), rinv = loose ? ri_min : (eq ? ri_eq : ri_in);
            // (every fused multiply-add written out, in the form the single kernel's build has: a * b + c * d may be contracted either way round)
            const double zr = __builtin_fma(alpha, zt, (1.0 - alpha) * zo);
            const double zn = fmin(fmax(__builtin_fma(rinv, yo, zr), lo), up);
            const double dz = zr - zn, yn = __builtin_fma(rh, dz, yo);
            cx.Z[i] = zn; cx.Y[i] = yn; cx.W[i] = __builtin_fma(rh, zn, -yn);
            const double dy = rh * dz;
            if (save) dyg[i] = dy;
          }
        }
      }
      bsync<NW>();     // every wave has finished reading xtilde (R) as the gather source before X/R move on
      if (__builtin_expect(save, 0)) {     // (its own loop: the address of dx stays out of the iteration's live set)
        for (int t = tid; t < npad; t += NT) { const double xo = cx.X[t]; dxg[t] = __builtin_fma(alpha, cx.R[t], (1.0 - alpha) * xo) - xo; }
      }
      for (int t = tid; t < npad; t += NT) cx.X[t] = __builtin_fma(alpha, cx.R[t], (1.0 - alpha) * cx.X[t]);
      // (no barrier here in an ordinary iteration: x_t is next read by the thread that wrote it -- the A' sweep hands out the same indices --, R is not
      // written before that sweep's own rows, and the barrier above already separates them from the A sweep's gathers; only the residual sweeps of a
      // termination check gather x)
      if (__builtin_expect(save, 0)) bsync<NW>();
      TS(6);
      iter_done = iter;
      if (__builtin_expect(can_check, 0)) {     // (rare paths are marked cold: their register pressure must not cost the hot loop its registers)
        update_info_res<NW, true>(cx, in);
        status = check_termination_res<NW>(cx, in, 0);
        TS(7);
        if (status != MPCQP_UNSOLVED) break;
      }
      if (__builtin_expect(do_rho, 0)) {
        if (!can_check) update_info_res<NW, true>(cx, in);
        const double pr = in.prs / (fmax(in.nzs, in.naxs) + Q_DIV_TOL);
        const double dr = in.drs / (fmax(in.nqs, fmax(in.natys, in.npxs)) + Q_DIV_TOL);
        double rn = cx.rho * sqrt(pr / (dr + Q_DIV_TOL));
        rn = fmin(fmax(rn, Q_RHO_MIN), Q_RHO_MAX);
        if (rn > cx.rho * st.adaptive_rho_tolerance || rn < cx.rho / st.adaptive_rho_tolerance) {
          cx.rho = uni(rn);
          if constexpr (RF == 0) {
            // leave for a new factor: the set-up kernel's resume mode builds it, the next launch of this kernel goes on from here
            for (int t = tid; t < npad; t += NT) Xs[t] = cx.X[t];
            for (int i = tid; i < mpad; i += NT) { Zs[i] = cx.Z[i]; Ys[i] = cx.Y[i]; }
            if (tid == 0) {
              io.status[b] = OC_PENDING; io.iters[b] = iter;
              io.info[4L * b] = in.obj; io.info[4L * b + 1] = in.prim_res; io.info[4L * b + 2] = in.dual_res; io.info[4L * b + 3] = cx.rho;
            }
            return;
          } else {
            if (!factorize_res<NW, (HUB ? 2 : 1)>(cx, &oc, octab, ocBL)) { status = MPCQP_NON_CVX; break; }
            oc_load_factor<NW, OCG, OCH>(oc, oc.tab, ws + pl.o_Lf, ocBL, ocl, ocG, ocHF, ocHT, wid, lane);
            rho_constants();
          }
        }
      }
    }
    if (iter > st.max_iter) iter_done = st.max_iter;
    if (status == MPCQP_UNSOLVED) {
      if (!can_check) { update_info_res<NW, true>(cx, in); status = check_termination_res<NW>(cx, in, 0); }
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
  if (tid == 0) {
    io.status[b] = status; io.iters[b] = iter_done;
    io.info[4L * b] = in.obj; io.info[4L * b + 1] = in.prim_res; io.info[4L * b + 2] = in.dual_res; io.info[4L * b + 3] = cx.rho;
  }
  TS(8);
#ifdef MPCQP_TIMING_SWEEP
  // (wave NW - 1's stamps travel through LDS to the wave that stores the row)
  if (wid == NW - 1 && lane == 0) { cx.RED[0] = (double)cx.fts[2]; cx.RED[1] = (double)cx.fts[3]; }
  bsync<NW>();
  ts_acc[13] = 0; ts_acc[14] = (unsigned long long)cx.RED[0]; ts_acc[15] = (unsigned long long)cx.RED[1];
  if (tid == 0 && io.dbg) for (int k_ = 9; k_ < 16; k_++) io.dbg[16L * b + k_] = 0;
#elif defined(MPCQP_TIMING) && !defined(MPCQP_TIMING_RUIZ)
  ts_acc[12] += cx.fts[0]; ts_acc[13] += cx.fts[1]; ts_acc[14] += cx.fts[2]; ts_acc[15] += cx.fts[3];
#endif
  TS_STORE_ADD(io.dbg, 0, 16);
}
// The launch.  Workgroups are handed to the XCDs and their shader engines round-robin, statically: a grid of one workgroup per instance is 8 x 4 separate
// queues, and instances whose iteration counts differ (25, 50, 75 ...) leave some of them busy long after the others have drained -- on quadrotor N=50 x 8192
// the iteration kernel took 23.5 ms in batch order and 21.0 ms with the longest-first order of a repeated solve.  With DevIO.queue the grid is as many
// workgroups as the GPU holds at once, and each draws the next instance from ONE counter until none is left: the same batch order, balanced to within one
// instance.  (Every wave reaches the exit: the ticket is read behind a barrier, uniformly.)
// PAIRS: a four-wave instance whose plan has two twisted pairs (the dissected order): the solve of the eight-wave instances' text, chain waves 0 .. 3
template <int NW, int OCG, int OCH, int RF, bool TL = false, bool PAIRS = false>
__global__ void __launch_bounds__(NW * WAVE, 2) mpcqp_oc_admm_kernel(const DevPlan pl, const DevRes rs, const mpcqp_settings st, const DevIO io, const DevOc oc) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int lane = threadIdx.x & 63;
  int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if constexpr (NW == 4) wid = oc_wave_role4(lds, wid, lane, io.no_remap);
  if constexpr (NW != 8) {      // (the four-wave instances, two workgroups per CU, keep a grid of one workgroup per instance: measured below)
    oc_admm_one<NW, OCG, OCH, RF, TL, PAIRS>(pl, rs, st, io, oc, lds, __builtin_amdgcn_readfirstlane(io.order ? io.order[blockIdx.x] : (int)blockIdx.x), wid, lane);
    return;
  }
  const OcLds<NW> L = oc_lds<NW>(lds, pl, rs, oc);
  int *ticket = reinterpret_cast<int *>(L.RED + 16 * NW);      // (the spare words in front of the chain tables)
  if (threadIdx.x == 0) { ticket[1] = 0; ticket[2] = 0; ticket[3] = 0; }
  for (;;) {
    if (!oc.resume) { if (threadIdx.x == 0) *ticket = atomicAdd(io.queue, 1); }
    else if (threadIdx.x < WAVE) {
      // a resume launch serves the few instances that left the launch before: tickets are drawn 64 at a time, the wave looks at their status words together
      // and keeps the waiting ones as a bit mask (one ticket and one barrier pair per instance made an all-idle launch of 8192 cost 0.1 - 0.2 ms)
      int base = ticket[1]; unsigned long long mask = ((unsigned long long)(unsigned)ticket[3] << 32) | (unsigned)ticket[2];
      while (mask == 0ull && base < io.count) {
        int b0 = 0;
        if (threadIdx.x == 0) b0 = atomicAdd(io.queue, WAVE);
        base = __builtin_amdgcn_readfirstlane(b0);
        const int t = base + (int)threadIdx.x;
        int st_ = 0;
        if (t < io.count) st_ = io.status[io.order ? io.order[t] : t];
        mask = __ballot(st_ == OC_PENDING || st_ == OC_PENDING_NONCVX);
      }
      if (threadIdx.x == 0) {
        if (mask == 0ull) *ticket = io.count;
        else { *ticket = base + __builtin_ctzll(mask); mask &= mask - 1; }
        ticket[1] = base; ticket[2] = (int)(unsigned)(mask & 0xffffffffull); ticket[3] = (int)(unsigned)(mask >> 32);
      }
    }
    bsync<NW>();
    const int t = __builtin_amdgcn_readfirstlane(*ticket);
    if (t >= io.count) break;
    // (lane and wave index made opaque per instance: everything derived from them is computed again for each instance, as in a fresh workgroup -- hoisted out of
    // this loop it stays live across the whole body: 240 B of scratch in an instance that had none)
    int lane_q = lane, wid_q = wid;
    asm volatile("" : "+v"(lane_q), "+s"(wid_q));
    oc_admm_one<NW, OCG, OCH, RF, TL, PAIRS>(pl, rs, st, io, oc, lds, __builtin_amdgcn_readfirstlane(io.order ? io.order[t] : t), wid_q, lane_q);
    bsync<NW>();      // (LDS is this instance's until every wave is through with it -- and the ticket until every wave has read it)
  }
}
