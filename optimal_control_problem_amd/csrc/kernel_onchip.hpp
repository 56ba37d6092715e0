// kernel_onchip.hpp -- the "on-chip" solve of mpcqp_res_kernel<..., OCG, OCH>: the whole block LDL' factor stays on the CU at two
// workgroups per CU, and every 16x16 mat-vec of the triangular solves runs on the matrix cores, register to register along a chain
// Part of the single translation unit mpcqp.hip (included there in order; not a stand-alone header).
#pragma once

// =========================================================================================================
// Why: the global-block kernels re-stream each QP's factor (120 KB on the 12-state quadrotor, N = 20) from the HBM slab in every
// ADMM iteration -- 197x the algorithmic bytes, the launch pinned to the fabric ceiling -- and the LDS-resident kernel fits one
// such QP per CU and is bound by the latency of its dependent chain ops (LDS write -> wave fence -> LDS read -> FMAs -> DPP
// quad sum per stage, 330-550 cycles).  Here the factor is split over both on-chip memories so that TWO workgroups fit a CU:
//   * chain blocks W_succ(p),p, the hub's inverse diagonal block and as many hub blocks W_hub,p as fit: LDS (<= 80 KB per
//     workgroup), one swizzled copy that is read conflict-free in both orientations (forward sweep: rows, backward: columns);
//   * the inverse diagonal blocks G_p (symmetric: one orientation) and the remaining hub blocks (both orientations): registers
//     of the wave that owns position p (p mod 4), statically indexed.
// and a mat-vec is 4 x v_mfma_f64_16x16x4_f64 with the vector in column 0 of the B operand.  With the index map phi below the
// C/D layout of column 0 IS the B layout of the next op, so a chain t_succ(p) = rhs - W t_p runs register to register: 4
// dependent MFMAs per stage and no LDS round trip on the critical path.  (Only 1 of 16 MFMA columns carries data; the
// matrix pipe has the room: 404 MFMAs per solve and QP = 13k pipe-cycles over 4 SIMDs.)
//
// MFMA index m (0..15) <-> position a inside a 16-block: a = phi(m) = 4 (m & 3) + (m >> 2).  Lane l = 16 kk + n:
//   A operand  A[m = n][k = kk + 4 i], i = 0..3  ->  block[phi(n)][4 kk + i]   (a row piece: one 32-byte read)
//   B operand  B[k = kk + 4 i][n]               ->  vec[4 kk + i]              (the same 32 bytes for the 16 lanes of a group)
//   C/D        D[m = kk + 4 g][n], g = 0..3     ->  out[4 kk + g]              (every column n holds the same vector)
// The transposed op y = W' v reads A[m = n][k] = block[4 kk + i][phi(n)] (four 8-byte reads down a column).
// =========================================================================================================
struct DevOc {
  int nbc, has_hub, junc, npw, nhr, nlds, ntab;
  int o_chainE, o_chainF, o_pos, o_fill, ghub_slot;
  const int *tab;
};

// LDS image of a 16x16 block: rows alternate between the two 32-bank halves in a pattern that also separates rows 4 apart
// (transposed reads touch rows i, 4 + i, 8 + i, 12 + i at once), the four 32-byte pieces of a row are rotated by the row pair,
// and the two 16-byte halves of a piece are swapped in the lower eight rows: a row-piece read (ds_read_b128 x 2) of 16 lanes
// covers all 64 banks once, a column read (ds_read_b64) of 32 lanes as well.
__host__ __device__ __forceinline__ int oc_swz(int r, int c) {
  return ((r ^ ((r >> 2) & 1)) << 4) | (c ^ (((r >> 1) & 3) << 2) ^ (((r >> 3) & 1) << 1));
}
__host__ __device__ __forceinline__ int oc_phi(int m) { return 4 * (m & 3) + (m >> 2); }

typedef double d2 __attribute__((ext_vector_type(2)));

struct OcLane {
  int fLo, fHi;      // doubles inside a swizzled LDS block: elements (phi(n), 4 kk + 0..1) and (phi(n), 4 kk + 2..3)
  int t0, t1, t2, t3;   // elements (4 kk + i, phi(n))
  int gF;            // row-major offset of (phi(n), 4 kk): A-operand reads from the slab
  int gT;            // row-major offset of (4 kk, phi(n)); row i is + 16 i
  int vb;            // 4 kk: this lane group's piece of a vector block
  bool col0;         // n == 0: the lane whose vector stores count
};
__device__ __forceinline__ OcLane oc_lane(int lane) {
  OcLane ln;
  const int n = lane & 15, kk = lane >> 4, r = oc_phi(n);
  ln.fLo = oc_swz(r, 4 * kk); ln.fHi = oc_swz(r, 4 * kk + 2);
  ln.t0 = oc_swz(4 * kk, r); ln.t1 = oc_swz(4 * kk + 1, r); ln.t2 = oc_swz(4 * kk + 2, r); ln.t3 = oc_swz(4 * kk + 3, r);
  ln.gF = r * BS + 4 * kk; ln.gT = 4 * kk * BS + r;
  ln.vb = 4 * kk; ln.col0 = n == 0;
  return ln;
}
__device__ __forceinline__ d4 oc_ldF(const double *blk, const OcLane &ln) {
  const d2 lo = *reinterpret_cast<const d2 *>(blk + ln.fLo), hi = *reinterpret_cast<const d2 *>(blk + ln.fHi);
  return d4{lo[0], lo[1], hi[0], hi[1]};
}
__device__ __forceinline__ d4 oc_ldT(const double *blk, const OcLane &ln) { return d4{blk[ln.t0], blk[ln.t1], blk[ln.t2], blk[ln.t3]}; }
__device__ __forceinline__ d4 oc_ldB(const double *vec, int p, const OcLane &ln) { return *reinterpret_cast<const d4 *>(vec + BS * p + ln.vb); }
__device__ __forceinline__ void oc_stB(double *vec, int p, const OcLane &ln, const d4 v) {
  if (ln.col0) *reinterpret_cast<d4 *>(vec + BS * p + ln.vb) = v;
}
// acc += Block * v on the matrix cores (a: the block's A-operand registers of this lane, v: the vector in the B layout)
__device__ __forceinline__ d4 oc_mv(const d4 a, const d4 v, d4 acc) {
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], v[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1], v[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[2], v[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[3], v[3], acc, 0, 0, 0);
  return acc;
}
__device__ __forceinline__ int oc_tab(const int *tab, int k) { return __builtin_amdgcn_readfirstlane(tab[k]); }

// After a factorisation (or on a kept workspace): bring the factor from the slab on chip.  Off-diagonal blocks are stored
// negated, so that every op of the sweeps is an accumulation  acc += block * v.
template <int NW, int NG, int NH>
__device__ __forceinline__ void oc_load_factor(const DevOc &oc, const int *tab, const double *slab, double *BL, const OcLane &ln, d4 (&G)[NG], d4 (&HF)[NH > 0 ? NH : 1],
                                               d4 (&HT)[NH > 0 ? NH : 1], const int wid, const int lane) {
  for (int j = wid; j < oc.nlds; j += NW) {
    const int src = oc_tab(tab, oc.o_fill + 3 * j), slot = oc_tab(tab, oc.o_fill + 3 * j + 1), neg = oc_tab(tab, oc.o_fill + 3 * j + 2);
    d4 v = reinterpret_cast<const d4 *>(slab + (long)src * BLK)[lane];            // row lane >> 2, columns 4 (lane & 3) ...
    if (neg) v = -v;
    const int r = lane >> 2, c = 4 * (lane & 3);
    double *dst = BL + (long)slot * BLK;
    *reinterpret_cast<d2 *>(dst + oc_swz(r, c)) = d2{v[0], v[1]};
    *reinterpret_cast<d2 *>(dst + oc_swz(r, c + 2)) = d2{v[2], v[3]};
  }
#pragma unroll
  for (int s = 0; s < NG; s++) {
    const int p = wid + NW * s;
    G[s] = d4{0, 0, 0, 0};
    if (s < NH) { HF[s] = d4{0, 0, 0, 0}; HT[s] = d4{0, 0, 0, 0}; }
    if (p < oc.nbc) {
      const int gs = oc_tab(tab, oc.o_pos + 5 * p), hs = oc_tab(tab, oc.o_pos + 5 * p + 2);
      G[s] = *reinterpret_cast<const d4 *>(slab + (long)gs * BLK + ln.gF);
      if (s < NH) {
        if (s < oc.nhr && hs >= 0) {
          const double *hb = slab + (long)hs * BLK;
          HF[s] = -*reinterpret_cast<const d4 *>(hb + ln.gF);
          HT[s] = -d4{hb[ln.gT], hb[ln.gT + BS], hb[ln.gT + 2 * BS], hb[ln.gT + 3 * BS]};
        }
      }
    }
  }
  bsync<NW>();
}

// what a wave needs to know about its own positions p = wid + NW s, read once into scalar registers
template <int NG>
struct OcWave { int hub[NG]; };     // >= 0: LDS slot of W_hub,p; -1: in this wave's registers; -2: no hub block (or no position)
template <int NW, int NG, int NH>
__device__ __forceinline__ OcWave<NG> oc_wave(const DevOc &oc, const int *tab, const int wid) {
  OcWave<NG> ow;
#pragma unroll
  for (int s = 0; s < NG; s++) {
    const int p = wid + NW * s;
    int h = -2;
    if (p < oc.nbc && oc.has_hub && oc_tab(tab, oc.o_pos + 5 * p + 2) >= 0) h = (s < NH && s < oc.nhr) ? -1 : oc_tab(tab, oc.o_pos + 5 * p + 4);
    ow.hub[s] = h;
  }
  return ow;
}
__device__ __forceinline__ int2 oc_pair(const int *tab, int k) {
  const int2 e = *reinterpret_cast<const int2 *>(tab + k);
  return make_int2(__builtin_amdgcn_readfirstlane(e.x), __builtin_amdgcn_readfirstlane(e.y));
}

// x = M^-1 rhs in place on the solve vector R (positions = blocks of 16; behind it the junction term and the waves' hub partials).
//   F1  the chains, one wave each: t_succ(p) = rhs_succ(p) - W t_p, register to register; the wave of chain E ends with the
//       junction term -W_f,e t_e (f = the shared last element, kept by chain F)
//   F2  every wave, for its own positions p = wid + 4 s: hub partial sum  -sum W_hub,p t_p
//   F3  every wave: t_hub, x_hub = G_hub t_hub
//   B1  every wave, own positions: d_p = G_p t_p - W_hub,p' x_hub
//   B2  the chains backwards: x_p = d_p - W_succ(p),p' x_succ(p)
// Chain tables are {position, LDS slot of the block below it} pairs; the entries, blocks and right-hand sides of the next stage
// are fetched while the current stage multiplies.
template <int NW, int NG, int NH>
__device__ __forceinline__ void oc_solve(const DevOc &oc, const int *tab, const double *BL, double *R, const int npad, const OcLane &ln, const OcWave<NG> &ow,
                                         const d4 (&G)[NG], const d4 (&HF)[NH > 0 ? NH : 1], const d4 (&HT)[NH > 0 ? NH : 1], const int wid) {
  double *JUNC = R + npad, *HP = R + npad + BS;
  const int LE = oc_tab(tab, 0), LF = oc_tab(tab, 1);
  const int len = wid == 0 ? LE : LF, cb = wid == 0 ? oc.o_chainE : oc.o_chainF;
  const int f = (oc.junc && LF > 0) ? oc_tab(tab, oc.o_chainF + 2 * (LF - 1)) : -1;
  const int H = oc.nbc;
  // ---- F1
  if (wid < 2 && len > 0) {
    int2 e = oc_pair(tab, cb);                      // stage k multiplies block e.y into position en.x
    d4 v = oc_ldB(R, e.x, ln);
    if (len > 1) {
      int2 en = oc_pair(tab, cb + 2);
      d4 a = oc_ldF(BL + (long)e.y * BLK, ln), c = oc_ldB(R, en.x, ln);
      int2 enn = len > 2 ? oc_pair(tab, cb + 4) : en;
      for (int k = 1; k < len; k++) {
        d4 an = a, cn = c; int2 e3 = enn;
        if (k + 1 < len) {
          an = oc_ldF(BL + (long)en.y * BLK, ln); cn = oc_ldB(R, enn.x, ln);
          if (k + 2 < len) e3 = oc_pair(tab, cb + 2 * (k + 2));
        }
        c = oc_mv(a, v, c);
        oc_stB(R, en.x, ln, c);
        v = c; e = en; en = enn; enn = e3; a = an; c = cn;
      }
    }
    if (wid == 0 && oc.junc) oc_stB(JUNC, 0, ln, oc_mv(oc_ldF(BL + (long)e.y * BLK, ln), v, d4{0, 0, 0, 0}));
  }
  bsync<NW>();
  d4 xh = {0, 0, 0, 0};
  if (oc.has_hub) {
    // ---- F2
    d4 hacc = {0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < NG; s++) {
      const int p = wid + NW * s;
      if (ow.hub[s] > -2) {
        d4 t = oc_ldB(R, p, ln);
        if (p == f) t += oc_ldB(JUNC, 0, ln);
        if (s < NH && ow.hub[s] == -1) hacc = oc_mv(HF[s < NH ? s : 0], t, hacc);
        else hacc = oc_mv(oc_ldF(BL + (long)ow.hub[s] * BLK, ln), t, hacc);
      }
    }
    oc_stB(HP, wid, ln, hacc);
    bsync<NW>();
    // ---- F3
    d4 th = oc_ldB(R, H, ln);
#pragma unroll
    for (int w = 0; w < NW; w++) th += oc_ldB(HP, w, ln);
    xh = oc_mv(oc_ldF(BL + (long)oc.ghub_slot * BLK, ln), th, d4{0, 0, 0, 0});
  }
  // ---- B1
#pragma unroll
  for (int s = 0; s < NG; s++) {
    const int p = wid + NW * s;
    if (p < oc.nbc) {
      d4 t = oc_ldB(R, p, ln);
      if (p == f) t += oc_ldB(JUNC, 0, ln);
      d4 d = oc_mv(G[s], t, d4{0, 0, 0, 0});
      if (ow.hub[s] > -2) {
        if (s < NH && ow.hub[s] == -1) d = oc_mv(HT[s < NH ? s : 0], xh, d);
        else d = oc_mv(oc_ldT(BL + (long)ow.hub[s] * BLK, ln), xh, d);
      }
      oc_stB(R, p, ln, d);
    }
  }
  bsync<NW>();
  // ---- B2
  if (oc.has_hub && wid == NW - 1) oc_stB(R, H, ln, xh);     // only now: every wave has read the hub's right-hand side
  if (wid < 2 && len > 0) {
    int k = len - 2;
    d4 v;
    if (wid == 0 && oc.junc) { v = oc_ldB(R, f, ln); k = len - 1; }       // chain E starts below f, which chain F's owner finished in B1
    else v = oc_ldB(R, oc_tab(tab, cb + 2 * (len - 1)), ln);
    if (k >= 0) {
      int2 e = oc_pair(tab, cb + 2 * k);             // stage k: x_e.x = d_e.x + block(e.y)' v
      d4 a = oc_ldT(BL + (long)e.y * BLK, ln), c = oc_ldB(R, e.x, ln);
      int2 en = k > 0 ? oc_pair(tab, cb + 2 * (k - 1)) : e;
      for (; k >= 0; k--) {
        d4 an = a, cn = c; int2 e3 = en;
        if (k > 0) {
          an = oc_ldT(BL + (long)en.y * BLK, ln); cn = oc_ldB(R, en.x, ln);
          if (k > 1) e3 = oc_pair(tab, cb + 2 * (k - 2));
        }
        c = oc_mv(a, v, c);
        oc_stB(R, e.x, ln, c);
        v = c; e = en; en = e3; a = an; c = cn;
      }
    }
  }
  bsync<NW>();
}
