// kernel_onchip.hpp -- the "on-chip" mode of mpcqp_res_kernel<..., OCG, OCH>: the whole block LDL' factor stays on the CU at two workgroups
// per CU; the triangular solves, and the factorisation of this topology, run on the matrix cores
// Part of the single translation unit mpcqp.hip (included there in order; not a stand-alone header).
#pragma once

// =========================================================================================================
// Why: the global-block kernels re-stream each QP's factor (120 KB on the 12-state quadrotor, N = 20) from the HBM slab in every
// ADMM iteration -- 197x the algorithmic bytes, the launch pinned to the fabric ceiling -- and the LDS-resident kernel fits one
// such QP per CU and is bound by the latency of its dependent chain ops (LDS write -> wave fence -> LDS read -> FMAs -> DPP
// quad sum per stage, 330-550 cycles).  Here the factor is split over both on-chip memories so that TWO workgroups fit a CU:
//   * chain blocks W_succ(p),p, the hub's inverse diagonal block and as many hub blocks W_hub,p as fit: LDS (<= 80 KB per
//     workgroup), one swizzled copy (oc_swz) that is read in both orientations (forward sweep: rows, backward: columns);
//   * the inverse diagonal blocks G_p (symmetric: one orientation) and the remaining hub blocks (both orientations): registers
//     of the wave that owns position p (p mod 4), statically indexed.
// A 16x16 mat-vec of the solve is four steps of v_mfma_f64_4x4x4_4b_f64 (oc_mv4 below: layout, cost and how a result becomes
// the next operand); the products of the factorisation (oc_ldl: 16x16 times 16x16, all sixteen columns in use) are four
// v_mfma_f64_16x16x4_f64 each, with the operand layouts D(X) / A(X) described there.
// =========================================================================================================
// Dense tiles of A for the two sweeps of the iteration (plan.hpp build_tile_plan), device side
struct DevTile {
  int on, ntile;                  // tile ntile is the zero tile (no rows, all zeros): what a batch of four is padded with
  int nAr, nAtr;                  // chunks of the two remainder layouts (= those of A and A')
  const int *Ar_off, *Ar_idx, *Ar_src, *Atr_off, *Atr_idx, *Atr_src;
  long Ar_entries, Atr_entries;
  const int *tJ, *rowid, *tsrc;   // [ntile + 1], [(ntile + 1) * 16], [(ntile + 1) * 256] (set-up only)
  const int *ta_info, *ta_cnt;    // [8 * chunks of A] {tile, column block, first row, rows} of the tiles with a row in the chunk (padded with the zero tile), [chunks] how many
  const unsigned long long *ta_mask;   // [chunks of A] bit r: row 64 c + r gets a tile contribution
  const int *tt_info;             // [4 * chunks of A'] {tile of column block J or the zero tile, first row, rows, 0}
  long o_tile, o_ellAr, o_ellAtr; // slab offsets (doubles)
};
struct DevOc {
  int nbc, has_hub, junc, npw, nhr, nlds, ntab;
  int o_chainE, o_chainF, o_pos, o_fill, ghub_slot, ghub_src;
  const int *a_assign;  // eight-wave instances: [8][32] row chunks of A per wave for the iteration's A sweep, balanced by load batches (-1: none); NULL = chunk wid + 8 k
  int npair, o_pair;   // twisted pairs of chains (plan.hpp OcPlan::pairs) and their records in tab: {LE, LF, oE, oF, junc, sLE, sLF, soE, soF, -, -, -}
  int nfill, o_s, o_dbl, ndbl, o_pp;   // LDS slots filled from the slab; the chains the solve walks; its double stages and their product blocks (plan.hpp oc_add_doubles)
  int at_poll, at_free;   // chunks of A' whose rows the iteration computes during the chain phase instead of before it (-1: none); see oc_solve
  int a_lds, p_lds; // the ELL values of A (and of P behind them) fit the LDS block slots: they stay there while the problem is scaled
  const int *tab;
  const int *asm_rec;   // [8 nblk] assembly recipe per block {terms, diagonal block row or -1, a0, b0, a1, b1, a2, b2} (T tile ids of the first three terms)
  DevTile tl;           // (the instances without tiles keep their argument layout)
  int resume;           // two-kernel form: this launch continues instances that left the iteration kernel for a re-factorisation (kernel_oc_split.hpp)
  int ixo_a, ixo_p;     // ... where those tables start, in 16-bit units from the LDS base (the z region; for shapes without staged values also the factorisation's scratch, idle until then)
  int ix16, zpad;       // set-up kernel (its own vector layout, kernel_oc_split.hpp oc_lds): 16-bit index tables of A (& 1) and P (& 2) in its z region of zpad doubles
};

// LDS image of a 16x16 block: rows alternate between the two 32-bank halves in a pattern that also separates rows 4 apart, the four
// 32-byte pieces of a row are rotated by the row pair, and the two 16-byte halves of a piece are swapped in the lower eight rows.  The
// solve reads a block element by element -- sixteen rows x one 32-byte piece per step (forward), four rows x sixteen columns (backward):
// at most two lanes per bank either way.
__host__ __device__ __forceinline__ int oc_swz(int r, int c) {
  return ((r ^ ((r >> 2) & 1)) << 4) | (c ^ (((r >> 1) & 3) << 2) ^ (((r >> 3) & 1) << 1));
}

typedef double d2 __attribute__((ext_vector_type(2)));

// what a lane of the solve keeps about itself (oc_mv4 below): row r4 = lane & 15, k4 = lane >> 4.  The eight block offsets are bytes of two
// registers: every register the solve keeps live across the iteration counts (a spilled offset is a scratch reload with a full wait in front of a chain)
struct OcLane {
  unsigned f4;       // byte K: element (r4, k4 + 4 K) of a swizzled LDS block
  unsigned t4;       // byte K: element (k4 + 4 K, r4)
  int k4;            // this lane's piece of a vector: v[k4 + 4 K]
  int o4;            // the element of the result this lane holds: 4 ((lane >> 2) & 3) + (lane >> 4)
};
__device__ __forceinline__ OcLane oc_lane(int lane) {
  OcLane ln;
  const int n = lane & 15, kk = lane >> 4;
  ln.f4 = 0; ln.t4 = 0;
#pragma unroll
  for (int K = 0; K < 4; K++) { ln.f4 |= (unsigned)oc_swz(n, kk + 4 * K) << (8 * K); ln.t4 |= (unsigned)oc_swz(kk + 4 * K, n) << (8 * K); }
  ln.k4 = kk; ln.o4 = 4 * ((lane >> 2) & 3) + (lane >> 4);
  return ln;
}
// acc += A B on the matrix cores, one 16x16x4 step per register pair (the factorisation's products: oc_mm below)
__device__ __forceinline__ d4 oc_mv(const d4 a, const d4 v, d4 acc) {
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], v[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1], v[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[2], v[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[3], v[3], acc, 0, 0, 0);
  return acc;
}
// The hub and diagonal phases multiply independent blocks by independent vectors, and a 16x16x4 MFMA spends 64 cycles of the matrix pipe on each
// of its four steps whether one column carries data or sixteen.  v_mfma_f64_4x4x4_4b_f64 -- four blocks of 4x4x4 per instruction, one double
// per lane and operand -- issues every 20 cycles when independent (52 dependent; tools/probes/mfma_4x4_probe.hip, which also gave the layout:
// A lane l <-> (i = l & 3, block = (l >> 2) & 3, k = l >> 4), B lane (j, block, k), D lane (j, block, i = l >> 4)).  For out = W v, block b of
// step K takes the tile W[4 b .. + 3][4 K .. + 3] and the piece v[4 K .. + 3] in every column: lane l supplies W[l & 15][(l >> 4) + 4 K] and
// v[(l >> 4) + 4 K], and receives out[4 ((l >> 2) & 3) + (l >> 4)].  Several such products run interleaved (oc_mv4 is one step of one of them).
__device__ __forceinline__ double oc_mv4(const double a, const double v, const double acc) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, v, acc, 0, 0, 0); }
__device__ __forceinline__ d4 oc_ld4(const double *blk, const unsigned pk) { return d4{blk[pk & 255], blk[(pk >> 8) & 255], blk[(pk >> 16) & 255], blk[pk >> 24]}; }
__device__ __forceinline__ d4 oc_ldF4(const double *blk, const OcLane &ln) { return oc_ld4(blk, ln.f4); }
__device__ __forceinline__ d4 oc_ldT4(const double *blk, const OcLane &ln) { return oc_ld4(blk, ln.t4); }
__device__ __forceinline__ d4 oc_ldB4(const double *vec, int p, const OcLane &ln) { const double *q = vec + BS * p + ln.k4; return d4{q[0], q[4], q[8], q[12]}; }
__device__ __forceinline__ void oc_stB4(double *vec, int p, const OcLane &ln, const double v) { vec[BS * p + ln.o4] = v; }      // (the four lanes of a quad store the same value)
// A result of oc_mv4 as the operand pieces of the next product: piece K of lane (k, block, j) is the result element 4 K + k, which sits in the
// lanes (k, block = K, j) -- quad K of every row of 16 lanes broadcast over the row: ds_swizzle in bit mode, lane' = (lane & 0x13) | (K << 2)
// (no LDS memory access, no register hand-over through a store and a load)
template <int K>
__device__ __forceinline__ double oc_bc4k(const double v) {
  union { double d; int i[2]; } a, r;
  a.d = v;
  r.i[0] = __builtin_amdgcn_ds_swizzle(a.i[0], (K << 7) | 0x13); r.i[1] = __builtin_amdgcn_ds_swizzle(a.i[1], (K << 7) | 0x13);
  return r.d;
}
__device__ __forceinline__ d4 oc_bc4(const double v) { return d4{oc_bc4k<0>(v), oc_bc4k<1>(v), oc_bc4k<2>(v), oc_bc4k<3>(v)}; }
__device__ __forceinline__ double oc_mv4x4(const d4 a, const d4 v, double acc) {
  acc = oc_mv4(a[0], v[0], acc); acc = oc_mv4(a[1], v[1], acc); acc = oc_mv4(a[2], v[2], acc); acc = oc_mv4(a[3], v[3], acc);
  return acc;
}
__device__ __forceinline__ double oc_ldE4(const double *vec, int p, const OcLane &ln) { return vec[BS * p + ln.o4]; }      // this lane's element of a vector block (the accumulator a product starts from)
__device__ __forceinline__ int oc_tab(const int *tab, int k) { return __builtin_amdgcn_readfirstlane(tab[k]); }

// ---- the two sweeps of the iteration on dense tiles (plan.hpp build_tile_plan): a tile is one 16 x 16 block of A's general rows, stored
// once in the A-operand layout of the 4-block MFMA ([lane][K]: element (r, c) at lane r + 16 (c & 3), K = c >> 2).
//   z~ = A x~ :  out = tile * x_J, the operand as it lies (one 32-byte load per lane), results (rows of A) in the lanes o4
//   A' w      :  out = tile' * w_rows, the operand read transposed -- four 8-byte loads per lane, each instruction four whole 128-byte lines
// Four tiles are in flight at a time (loads first, then sixteen interleaved MFMAs); a batch is padded with the zero tile.
__device__ __forceinline__ int oc_tile_toff(const int lane) { return 4 * (lane >> 4) + 64 * (lane & 3) + ((lane >> 2) & 3); }     // doubles; + 16 K
// Everything a chunk needs is requested at once -- the tile operands of its (up to four / eight) tiles and the first slots of its remainder
// layout -- so that a chunk costs one round trip to memory, not one per step; the tables are per-chunk records of fixed size read with scalar
// loads that depend on nothing but the chunk index (tile id, column block, first row, number of rows; the rows of a tile are consecutive).
template <bool MAXABS, int UMAX>
__device__ __forceinline__ double ell_chunk(const double *__restrict__ val, const int *__restrict__ idx, const double *in, const int s0, const int s1, const int lane);     // (kernel_resident.hpp)
template <int U>
struct EllPre { double v[U]; int ix[U]; };
template <int U>
__device__ __forceinline__ EllPre<U> ell_pre_load(const double *__restrict__ val, const int *__restrict__ idx, const int s0, const int s1, const int lane) {
  EllPre<U> p;
#pragma unroll
  for (int u = 0; u < U; u++) {
    p.v[u] = 0.0; p.ix[u] = 0;
    if (s0 + u < s1) { p.v[u] = val[(long)(s0 + u) * WAVE + lane]; p.ix[u] = idx[(long)(s0 + u) * WAVE + lane]; }
  }
  return p;
}
template <int U>
__device__ __forceinline__ double ell_pre_sum(const EllPre<U> &p, const double *in) {
  double acc = 0.0;
#pragma unroll
  for (int u = 0; u < U; u++) acc += p.v[u] * in[p.ix[u]];
  return acc;
}
// The per-chunk records ({tile, first row, rows} x 4 for a chunk of A' rows; {tile, column block, first row, rows} x 8, the tile count and
// the 64-bit row mask for a chunk of A's rows) never change: each wave reads the records of ITS chunks once, before the ADMM loop, into the
// lanes of a register per chunk (oc_tile_records), and the sweeps pick fields out with v_readlane -- no table access, scalar or vector, sits
// in front of a chunk's loads any more (each was a full round trip to memory: the first tile version was twice as slow as the ELL sweeps).
constexpr int OC_TILE_MAXA = 3, OC_TILE_MAXT = 2;      // chunks of A / A' rows per wave the tile instances keep records for (the host checks)
struct OcTileRec { int a[OC_TILE_MAXA], t[OC_TILE_MAXT]; };
template <int NW>
__device__ __forceinline__ OcTileRec oc_tile_records(const DevTile &tl, const int nA, const int nAt, const int wid, const int lane) {
  OcTileRec r;
#pragma unroll
  for (int k = 0; k < OC_TILE_MAXA; k++) {
    const int c = wid + k * NW;
    int v = (lane < 32 && (lane & 3) == 0) ? tl.ntile : 0;           // (no such chunk: eight zero tiles, no rows)
    if (c < nA) {
      if (lane < 32) v = tl.ta_info[32 * c + lane];
      else if (lane == 32) v = tl.ta_cnt[c];
      else if (lane == 33) v = (int)(tl.ta_mask[c] & 0xffffffffull);
      else if (lane == 34) v = (int)(tl.ta_mask[c] >> 32);
    }
    r.a[k] = v;
  }
#pragma unroll
  for (int k = 0; k < OC_TILE_MAXT; k++) {
    const int c = wid + k * NW;
    int v = (lane < 16 && (lane & 3) == 0) ? tl.ntile : 0;
    if (c < nAt && lane < 16) v = tl.tt_info[16 * c + lane];
    r.t[k] = v;
  }
  return r;
}
// the A' sweep of chunk c: R[t] = sigma x - q + sum over the chunk's four column blocks of tile' w_rows + remainder
template <int U, class F>
__device__ __forceinline__ void oc_tiles_at(const DevTile &tl, const int rec, const double *tiles, const double *valAtr, const int *coAtr, const double *W, double *R, const int c, const int nb,
                                            const OcLane &ln, const int lane, F &&finish) {
  const int toff = oc_tile_toff(lane);
  int tid[4], first[4], rows[4]; d4 a[4], v[4];
#pragma unroll
  for (int u = 0; u < 4; u++) { tid[u] = __builtin_amdgcn_readlane(rec, 4 * u); first[u] = __builtin_amdgcn_readlane(rec, 4 * u + 1); rows[u] = __builtin_amdgcn_readlane(rec, 4 * u + 2); }
#pragma unroll
  for (int u = 0; u < 4; u++) {
    const double *tp = tiles + (long)tid[u] * BLK + toff;
    a[u] = d4{tp[0], tp[16], tp[32], tp[48]};
  }
  const int s0 = coAtr[c], s1 = coAtr[c + 1];
  const EllPre<U> pre = ell_pre_load<U>(valAtr, tl.Atr_idx, s0, s1, lane);
#pragma unroll
  for (int u = 0; u < 4; u++) {
    const double *wp = W + first[u] + ln.k4;
    const int n = rows[u] - ln.k4;          // rows k4, k4 + 4, k4 + 8, k4 + 12 of the tile exist while below n
    v[u] = d4{n > 0 ? wp[0] : 0.0, n > 4 ? wp[4] : 0.0, n > 8 ? wp[8] : 0.0, n > 12 ? wp[12] : 0.0};
  }
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int K = 0; K < 4; K++)
#pragma unroll
    for (int u = 0; u < 4; u++) acc[u] = oc_mv4(a[u][K], v[u][K], acc[u]);
#pragma unroll
  for (int u = 0; u < 4; u++) if (4 * c + u < nb) R[BS * (4 * c + u) + ln.o4] = acc[u];
  double e = ell_pre_sum<U>(pre, W);
  if (s1 - s0 > U) e += ell_chunk<false, 8>(valAtr, tl.Atr_idx, W, s0 + U, s1, lane);
  finish(c * WAVE + lane, e);
}
// the tile part of the A sweep of chunk c: W[row] = tile * x~_J for the rows of chunk c that lie in a tile (a tile whose rows straddle two
// chunks is computed for both; each wave keeps its own rows); returns the remainder layout's row sums, *tiled = this lane's row has a tile part
template <int U>
__device__ __forceinline__ double oc_tiles_a(const DevTile &tl, const int rec, const double *tiles, const double *valAr, const int *coAr, const double *R, double *W, const int c,
                                             const OcLane &ln, const int lane, bool *tiled) {
  int tid[8], J[8], first[8], rows[8]; d4 a[8];
#pragma unroll
  for (int u = 0; u < 8; u++) {
    tid[u] = __builtin_amdgcn_readlane(rec, 4 * u); J[u] = __builtin_amdgcn_readlane(rec, 4 * u + 1);
    first[u] = __builtin_amdgcn_readlane(rec, 4 * u + 2); rows[u] = __builtin_amdgcn_readlane(rec, 4 * u + 3);
  }
  const int nt = __builtin_amdgcn_readlane(rec, 32);
  const unsigned long long mask = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(rec, 34) << 32) | (unsigned)__builtin_amdgcn_readlane(rec, 33);
  *tiled = (mask >> lane) & 1ull;
  if (nt > 0) {
#pragma unroll
    for (int u = 0; u < 4; u++) a[u] = reinterpret_cast<const d4 *>(tiles + (long)tid[u] * BLK)[lane];
  }
  if (nt > 4) {
#pragma unroll
    for (int u = 4; u < 8; u++) a[u] = reinterpret_cast<const d4 *>(tiles + (long)tid[u] * BLK)[lane];
  }
  const int s0 = coAr[c], s1 = coAr[c + 1];
  const EllPre<U> pre = ell_pre_load<U>(valAr, tl.Ar_idx, s0, s1, lane);
#pragma unroll
  for (int h = 0; h < 2; h++) {
    if (nt > 4 * h) {
      d4 v[4]; double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int u = 0; u < 4; u++) v[u] = oc_ldB4(R, J[4 * h + u], ln);
#pragma unroll
      for (int K = 0; K < 4; K++)
#pragma unroll
        for (int u = 0; u < 4; u++) acc[u] = oc_mv4(a[4 * h + u][K], v[u][K], acc[u]);
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int rid = first[4 * h + u] + ln.o4;
        if (ln.o4 < rows[4 * h + u] && (rid >> 6) == c) W[rid] = acc[u];
      }
    }
  }
  double e = ell_pre_sum<U>(pre, R);
  if (s1 - s0 > U) e += ell_chunk<false, 8>(valAr, tl.Ar_idx, R, s0 + U, s1, lane);
  return e;
}

// ---- the hub and diagonal phases of the solve (F2, F3, B1) on the VECTOR ALUs.  They are wave-parallel products of independent blocks with
// independent vectors, not dependent chains: what counts is throughput, and a 16 x 16 f64 mat-vec is four v_mfma_f64_4x4x4 = ~190 cycles of a
// SIMD's matrix pipe (48 each, measured: tools/probes/chain_stage_probe.hip) against four v_fma_f64 and a quad sum = ~40 cycles of its vector
// pipe.  (The chains stay on the matrix cores: there the point is latency -- no cross-lane reduction, the result already in operand layout.)
// Row-piece layout: lane l = (r = l >> 2, j = l & 3) holds the four consecutive entries [r][4 j .. 4 j + 3] of a block and multiplies them with
// the piece v[4 j .. 4 j + 3] of the vector (one 32-byte LDS read); the four lanes of a quad then hold the four partial sums of row r.
__device__ __forceinline__ double oc_quad_sum(double v) {       // sum over the 4 lanes of a quad with DPP quad_perm (no LDS crossbar round trip)
  union { double d; int i[2]; } a, t;
  a.d = v;
  t.i[0] = __builtin_amdgcn_mov_dpp(a.i[0], 0xB1, 0xF, 0xF, true);   // quad_perm:[1,0,3,2]
  t.i[1] = __builtin_amdgcn_mov_dpp(a.i[1], 0xB1, 0xF, 0xF, true);
  a.d += t.d;
  t.i[0] = __builtin_amdgcn_mov_dpp(a.i[0], 0x4E, 0xF, 0xF, true);   // quad_perm:[2,3,0,1]
  t.i[1] = __builtin_amdgcn_mov_dpp(a.i[1], 0x4E, 0xF, 0xF, true);
  return a.d + t.d;
}
__device__ __forceinline__ d4 oc_ldV4(const double *vec, const int p, const int lane) { return *reinterpret_cast<const d4 *>(vec + BS * p + 4 * (lane & 3)); }
__device__ __forceinline__ void oc_stV(double *vec, const int p, const int lane, const double v) { vec[BS * p + (lane >> 2)] = v; }      // (the four lanes of a quad store the same value)
__device__ __forceinline__ double oc_dot4(const d4 a, const d4 v, double acc) {
  acc = __builtin_fma(a[0], v[0], acc); acc = __builtin_fma(a[1], v[1], acc); acc = __builtin_fma(a[2], v[2], acc); acc = __builtin_fma(a[3], v[3], acc);
  return acc;
}
// row piece [r][4 j ..] of a block in its swizzled LDS image (oc_swz: the piece is one aligned group of four, its halves swapped in the lower rows)
__device__ __forceinline__ d4 oc_ldsRow(const double *blk, const int lane) {
  const int r = lane >> 2, j = lane & 3;
  const d4 q = *reinterpret_cast<const d4 *>(blk + (((r ^ ((r >> 2) & 1)) << 4) | ((j ^ ((r >> 1) & 3)) << 2)));
  return ((r >> 3) & 1) ? d4{q[2], q[3], q[0], q[1]} : q;
}
// ... and of its transpose: [4 j + i][c], c = lane >> 2
__device__ __forceinline__ d4 oc_ldsRowT(const double *blk, const int lane) {
  const int c = lane >> 2, j = 4 * (lane & 3);
  return d4{blk[oc_swz(j, c)], blk[oc_swz(j + 1, c)], blk[oc_swz(j + 2, c)], blk[oc_swz(j + 3, c)]};
}
__device__ __forceinline__ d4 oc_ldD(const double *blk, const int lane);
__device__ __forceinline__ d4 oc_ldA(const double *blk, const int lane);
__device__ __forceinline__ d4 oc_mm(const d4 a, const d4 b, d4 acc);
// After a factorisation (or on a kept workspace): bring the factor from the slab on chip.  Off-diagonal blocks are stored
// negated, so that every op of the sweeps is an accumulation  acc += block * v.
template <int NW, int NG, int NH>
__device__ __forceinline__ void oc_load_factor(const DevOc &oc, const int *tab, const double *slab, double *BL, const OcLane &ln, d4 (&G)[NG], d4 (&HF)[NH > 0 ? NH : 1],
                                               d4 (&HT)[NH > 0 ? NH : 1], const int wid, const int lane) {
  // This wave's table entries first, all of them, in the lanes of two registers (a table entry read where it is needed was a round trip to memory in
  // front of every block's own round trip): lane 3 k + {0, 1, 2} = {source block, LDS slot, negate} of its k-th LDS block (the wave's blocks are
  // j = wid + NW k; at most 21 of them: the host checks), lane 2 s + {0, 1} = {G block, hub block or -1} of its s-th position.
  const int kf = lane / 3, jf = wid + NW * kf;
  const int ftab = (kf < 21 && jf < oc.nfill) ? tab[oc.o_fill + 3 * jf + (lane - 3 * kf)] : 0;
  const int ps = lane >> 1, pp = wid + NW * ps;
  const int ptab = (ps < NG && pp < oc.nbc) ? tab[oc.o_pos + 5 * pp + 2 * (lane & 1)] : -1;
  auto put = [&](const d4 v0, const int slot, const int neg) {
    const d4 v = neg ? -v0 : v0;
    const int r = lane >> 2, c = 4 * (lane & 3);
    double *dst = BL + (long)slot * BLK;
    *reinterpret_cast<d2 *>(dst + oc_swz(r, c)) = d2{v[0], v[1]};
    *reinterpret_cast<d2 *>(dst + oc_swz(r, c + 2)) = d2{v[2], v[3]};
  };
  // four blocks in flight at a time (row lane >> 2, columns 4 (lane & 3) ...: one 32-byte load per lane and block)
  int k = 0;
  for (int j = wid; j < oc.nfill; j += 4 * NW, k += 4) {
    d4 v[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int src = __builtin_amdgcn_readlane(ftab, 3 * min(k + u, 20));
      v[u] = reinterpret_cast<const d4 *>(slab + (long)src * BLK)[lane];         // (past the wave's last block: block 0 once more, dropped)
    }
#pragma unroll
    for (int u = 0; u < 4; u++)
      if (j + u * NW < oc.nfill) put(v[u], __builtin_amdgcn_readlane(ftab, 3 * min(k + u, 20) + 1), __builtin_amdgcn_readlane(ftab, 3 * min(k + u, 20) + 2));
  }
  // register blocks, in the row-piece layout of the hub / diagonal phases (lane l: row l >> 2, entries 4 (l & 3) ...: one 32-byte load per lane; the
  // transposed hub block as four strided loads), every load of the wave issued before the first use; a slot without a position reads block 0 and is zeroed
  const int tc = lane >> 2, tj = 4 * (lane & 3);
#pragma unroll
  for (int s = 0; s < NG; s++) {
    const int gs = __builtin_amdgcn_readlane(ptab, 2 * s);
    G[s] = reinterpret_cast<const d4 *>(slab + (long)max(gs, 0) * BLK)[lane];
    if (s < NH) {
      const int hs = __builtin_amdgcn_readlane(ptab, 2 * s + 1);
      const double *hb = slab + (long)max(hs, 0) * BLK;
      HF[s] = reinterpret_cast<const d4 *>(hb)[lane];
      HT[s] = d4{hb[tj * BS + tc], hb[(tj + 1) * BS + tc], hb[(tj + 2) * BS + tc], hb[(tj + 3) * BS + tc]};
    }
  }
#pragma unroll
  for (int s = 0; s < NG; s++) {
    if (__builtin_amdgcn_readlane(ptab, 2 * s) < 0) G[s] = d4{0, 0, 0, 0};
    if (s < NH) {
      const bool has = __builtin_amdgcn_readlane(ptab, 2 * s + 1) >= 0;
      HF[s] = has ? -HF[s] : d4{0, 0, 0, 0};
      HT[s] = has ? -HT[s] : d4{0, 0, 0, 0};
    }
  }
  // product blocks of the double stages: W_b,m W_m,a (= the product of the two negated blocks the sweeps use) on the matrix cores, into their LDS slots
  for (int j = wid; j < oc.ndbl; j += NW) {
    const int sm = oc_tab(tab, oc.o_pp + 3 * j), sa = oc_tab(tab, oc.o_pp + 3 * j + 1), slot = oc_tab(tab, oc.o_pp + 3 * j + 2);
    const d4 pr = oc_mm(oc_ldA(slab + (long)sm * BLK, lane), oc_ldD(slab + (long)sa * BLK, lane), d4{0, 0, 0, 0});      // D layout: lane 16 kk + n holds X[kk + 4 g][n]
    double *dst = BL + (long)slot * BLK;
    const int n = lane & 15, kk = lane >> 4;
#pragma unroll
    for (int g = 0; g < 4; g++) dst[oc_swz(kk + 4 * g, n)] = pr[g];
  }
  bsync<NW>();
}

// =========================================================================================================
// The block LDL' of this topology, on the matrix cores and in registers.  The generic level loop of factorize_res works in the slab:
// every level is a handful of dependent L2 round trips (plan indices, operand tiles, temp tiles) between three workgroup barriers, and
// the 16 pivots of an in-LDS sweep cost ~900 cycles each.  Here the two chain waves (0: chain E, 1: chain F) carry the diagonal
// block from step to step in the MFMA C/D layout, and waves 2 / 3 follow one step behind with the hub row of "their" chain:
//   chain wave, step J:  G_J = S_JJ^-1 (oc_sweep);  W' = G_J L';  S_next,next -= W L'        (L = S_next,J: the block below the diagonal)
//   helper,  one behind: W_h' = G_J H_J';  S_hh -= W_h H_J';  H_next' = S_h,next' - L W_h'    (H_J = S_hub,J as updated so far)
// With the layouts  D(X): lane 16 kk + n holds X[kk + 4 g][n]  and  A(X) = D(X'): X[n][kk + 4 g],  oc_mm(a, b) = D(Xa Xb) for a = A(Xa),
// b = D(Xb), so every product above takes its operands from registers as the previous product (or a strided load from the slab)
// left them.  Hand-overs go through LDS scratch blocks: G_J to the helper (double-buffered by step parity), chain E's term for the
// junction block to chain F, the helpers' sums for the hub's diagonal block to wave 0, which inverts it last.
// Results land in the slab exactly where factorize_res leaves them (G_J in the diagonal slots, W in the off-diagonal ones).
// =========================================================================================================
__device__ __forceinline__ d4 oc_ldD(const double *blk, const int lane) { const int o = (lane >> 4) * BS + (lane & 15); return d4{blk[o], blk[o + 4 * BS], blk[o + 8 * BS], blk[o + 12 * BS]}; }
__device__ __forceinline__ void oc_stD(double *blk, const int lane, const d4 v) { const int o = (lane >> 4) * BS + (lane & 15); blk[o] = v[0]; blk[o + 4 * BS] = v[1]; blk[o + 8 * BS] = v[2]; blk[o + 12 * BS] = v[3]; }
__device__ __forceinline__ d4 oc_ldA(const double *blk, const int lane) { const int o = (lane & 15) * BS + (lane >> 4); return d4{blk[o], blk[o + 4], blk[o + 8], blk[o + 12]}; }
__device__ __forceinline__ void oc_stA(double *blk, const int lane, const d4 v) { const int o = (lane & 15) * BS + (lane >> 4); blk[o] = v[0]; blk[o + 4] = v[1]; blk[o + 8] = v[2]; blk[o + 12] = v[3]; }
__device__ __forceinline__ d4 oc_ldS(const double *scr, const int lane) { return reinterpret_cast<const d4 *>(scr)[lane]; }       // LDS scratch block: a wave's four registers as they are
__device__ __forceinline__ void oc_stS(double *scr, const int lane, const d4 v) { reinterpret_cast<d4 *>(scr)[lane] = v; }
__device__ __forceinline__ d4 oc_mm(const d4 a, const d4 b, d4 acc) { return oc_mv(a, b, acc); }
__device__ __forceinline__ double oc_readlane(const double v, const int l) {
  union { double d; int i[2]; } a, r;
  a.d = v;
  r.i[0] = __builtin_amdgcn_readlane(a.i[0], l); r.i[1] = __builtin_amdgcn_readlane(a.i[1], l);
  return r.d;
}
// 1 / d for a positive, normal d: v_rcp_f64 and two Newton steps (a full division adds scaling and a fix-up that the pivots of
// an equilibrated KKT matrix do not need; the sweep is a chain of 16 of these)
__device__ __forceinline__ double oc_rcp(const double d) {
  double p = __builtin_amdgcn_rcp(d);
  double e = __builtin_fma(-d, p, 1.0); p = __builtin_fma(p, e, p);
  e = __builtin_fma(-d, p, 1.0); p = __builtin_fma(p, e, p);
  return p;
}
// In-register inverse of an SPD block held in the D layout, by symmetric sweeps: sweeping pivot k (row u, pivot d = u_k) is the
// rank-1 update  a <- a0 - (1/d) w w'  with w = u except w_k = -1, a0 = a with row and column k cleared -- one MFMA whose only
// non-zero K slot is k & 3, which is also where the D layout keeps row k (register k >> 2 of lane group k & 3), for both operands:
// a is symmetric.  Pivots > 0 <=> positive definite (they are the Cholesky pivots squared).  After 16 sweeps a = -inverse.
__device__ __forceinline__ bool oc_sweep(d4 &a, const int lane) {
  const int n = lane & 15, kk = lane >> 4;
  bool ok = true;
#pragma unroll
  for (int k = 0; k < BS; k++) {
    const int g = k >> 2, ks = k & 3;
    const double row = a[g];
    const double d = oc_readlane(row, 16 * ks + k);
    ok = ok && d > 0.0;
    const double p = oc_rcp(d);
    const bool grp = kk == ks, colk = n == k;
    const double w = grp ? (colk ? -1.0 : row) : 0.0;
    const double bw = -p * w;
    a[g] = grp ? 0.0 : a[g];
#pragma unroll
    for (int gg = 0; gg < 4; gg++) a[gg] = colk ? 0.0 : a[gg];
    a = __builtin_amdgcn_mfma_f64_16x16x4f64(w, bw, a, 0, 0, 0);
  }
  a = -a;
  return ok;
}

constexpr int OC_LDL_SCR = 8;     // LDS scratch blocks of oc_ldl (the stage region is at least this large: plan.hpp oc_stage_doubles)
template <int NW, bool HUB>
__device__ __forceinline__ bool oc_ldl(const DevOc &oc, const int *ctab, double *slab, double *scr, double *red, const int wid, const int lane,
                                       unsigned long long *t_sweep = nullptr) {
  static_assert(NW >= 4, "two chain waves and their two helpers (further waves only take part in the barriers)");
  // scratch blocks: 0..3 G hand-over [chain][step parity], 4 chain E's term for the junction's diagonal block, 5 for its hub block, 6..7 the helpers' sums
  const int ch = wid & 1;
  const bool helper = NW > 4 ? (wid == 2 || wid == 3) : wid >= 2, idle = NW > 4 && wid >= 4;
  const int *pt = oc.tab + oc.o_pos;
  bool ok = true;
  d4 Sh = {0, 0, 0, 0};
  // (several twisted pairs -- the dissected order -- are taken one after the other by the same four waves: their chains are short, and the pairs only meet in
  // the hub's diagonal block, whose sums the helpers carry across)
  for (int pi = 0; pi < oc.npair; pi++) {
  const int rec = oc.o_pair + 12 * pi;
  const int LE = oc_tab(ctab, rec), LF = oc_tab(ctab, rec + 1);
  const bool junc = oc_tab(ctab, rec + 4) && LF > 0;
  const int S = junc ? max(LE + 1, LF) : LE;          // chain steps run in phases 0 .. S - 1 (chain F keeps its last step, the junction, for phase S - 1); helpers one phase behind
  const int L = ch == 0 ? LE : LF, cb = ch == 0 ? oc_tab(ctab, rec + 2) : oc_tab(ctab, rec + 3);
  // block ids of the chain's steps, step i in lane i (read back with v_readlane: no memory latency inside a phase)
  const int pv = L > 0 ? ctab[cb + 2 * min(lane, L - 1)] : 0;
  const int gsv = pt[5 * pv], csv = pt[5 * pv + 1], hsv = pt[5 * pv + 2];
  // the chain step worked on in phase s: chain E step s; chain F step s, its last one held back until chain E has finished
  auto step_of = [&](const int s) -> int {
    if (s < 0) return -1;
    if (ch == 0) return s < LE ? s : -1;
    if (s < LF - 1) return s;
    return (s == S - 1 && LF > 0) ? LF - 1 : -1;
  };
  d4 Dc = {0, 0, 0, 0}, Wp = {0, 0, 0, 0}, Hr = {0, 0, 0, 0}, Lc = {0, 0, 0, 0}, Ln = {0, 0, 0, 0}, Sn = {0, 0, 0, 0}, Hc = {0, 0, 0, 0}, Hn = {0, 0, 0, 0};
  int pend = -1;
  // Slab operands are fetched a whole step ahead (a slab read is an L2 / fabric round trip of a few thousand cycles).  What a fetch may
  // read early: L_k and the diagonal block below it are first written by this chain wave at step k (W over L: one phase after, see
  // below), S_hub,next by this helper at its step for `next`.
  auto chain_fetch = [&](const int k) {            // operands of chain step k: L_k = S_next,k and S_next,next
    const bool last = k == L - 1, tojunc = ch == 0 && last && junc;
    if (!last || tojunc) Ln = oc_ldA(slab + (long)__builtin_amdgcn_readlane(csv, k) * BLK, lane);
    if (!last) Sn = oc_ldD(slab + (long)__builtin_amdgcn_readlane(gsv, k + 1) * BLK, lane);
  };
  if (L > 0 && !idle) {
    if (!helper) { Dc = oc_ldD(slab + (long)__builtin_amdgcn_readlane(gsv, 0) * BLK, lane); chain_fetch(0); }
    else if (HUB) Hr = oc_ldA(slab + (long)__builtin_amdgcn_readlane(hsv, 0) * BLK, lane);
  }
  const int nph = HUB ? S + 1 : S;
  for (int s = 0; s < nph; s++) {
    if (idle) {
    } else if (!helper) {
      if (pend >= 0) { oc_stA(slab + (long)pend * BLK, lane, Wp); pend = -1; }     // one phase late: the helper has read L by now
      const int k = step_of(s);
      if (k >= 0) {
        const int gs = __builtin_amdgcn_readlane(gsv, k), cs = __builtin_amdgcn_readlane(csv, k);
        const bool last = k == L - 1, tojunc = ch == 0 && last && junc, has_next = !last || tojunc;
        const d4 Ls = Ln, Sd = tojunc ? d4{0, 0, 0, 0} : Sn;
        if (!last) chain_fetch(k + 1);
        if (ch == 1 && last && junc) Dc += oc_ldS(scr + 4 * BLK, lane);
#ifdef MPCQP_TIMING
        const unsigned long long s0_ = __builtin_amdgcn_s_memtime();
#endif
        ok = oc_sweep(Dc, lane) && ok;
#ifdef MPCQP_TIMING
        if (t_sweep && wid == 0) *t_sweep += __builtin_amdgcn_s_memtime() - s0_;
#endif
        oc_stD(slab + (long)gs * BLK, lane, Dc);
        if (HUB) oc_stS(scr + (2 * ch + (s & 1)) * BLK, lane, Dc);
        if (has_next) {
          const d4 Wt = oc_mm(Dc, Ls, d4{0, 0, 0, 0});
          Wp = Wt; pend = cs;
          const d4 acc = oc_mm(-Wt, Ls, Sd);
          if (tojunc) oc_stS(scr + 4 * BLK, lane, acc); else Dc = acc;
        }
      }
    } else if (HUB) {
      const int kc = step_of(s), kh = step_of(s - 1);
      if (kc >= 0) {       // operands of the helper's next step; L_kc is what the chain wave overwrites one phase from now
        const bool last = kc == L - 1, tojunc = ch == 0 && last && junc;
        if (!last || tojunc) Ln = oc_ldA(slab + (long)__builtin_amdgcn_readlane(csv, kc) * BLK, lane);
        if (!last) Hn = oc_ldA(slab + (long)__builtin_amdgcn_readlane(hsv, kc + 1) * BLK, lane);
      }
      if (kh >= 0) {
        const int hs = __builtin_amdgcn_readlane(hsv, kh);
        const bool last = kh == L - 1, tojunc = ch == 0 && last && junc, has_next = !last || tojunc;
        if (ch == 1 && last && junc) Hr += oc_ldS(scr + 5 * BLK, lane);
        const d4 G = oc_ldS(scr + (2 * ch + ((s - 1) & 1)) * BLK, lane);
        const d4 WhT = oc_mm(G, Hr, d4{0, 0, 0, 0});
        oc_stA(slab + (long)hs * BLK, lane, WhT);
        const d4 nW = -WhT;
        Sh = oc_mm(nW, Hr, Sh);
        if (has_next) {
          const d4 acc = oc_mm(Lc, nW, tojunc ? d4{0, 0, 0, 0} : Hc);
          if (tojunc) oc_stS(scr + 5 * BLK, lane, acc); else Hr = acc;
        }
      }
      Lc = Ln; Hc = Hn;
      __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): the loads of L have landed before the chain wave may store W over it
    }
    bsync<NW>();
  }
  if (!helper && !idle && pend >= 0) oc_stA(slab + (long)pend * BLK, lane, Wp);
  }
  if (HUB) {
    if (helper) oc_stS(scr + (6 + ch) * BLK, lane, Sh);
    bsync<NW>();
    if (wid == 0) {
      d4 Dh = oc_ldD(slab + (long)oc.ghub_src * BLK, lane) + oc_ldS(scr + 6 * BLK, lane) + oc_ldS(scr + 7 * BLK, lane);
      ok = oc_sweep(Dh, lane) && ok;
      oc_stD(slab + (long)oc.ghub_src * BLK, lane, Dh);
    }
  }
  if (lane == 0) red[wid] = ok ? 1.0 : 0.0;
  bsync<NW>();
  bool all_ok = true;
#pragma unroll
  for (int w = 0; w < NW; w++) all_ok = all_ok && red[w] != 0.0;
  bsync<NW>();
  return all_ok;
}

// what a wave needs to know about its own positions p = wid + NW s, read once into scalar registers.  Every loop over s is
// straight-line code: a slot without a position ("phantom": nbc is not a multiple of NW) reads the zero block behind the solve
// vector, multiplies a zero G and has its store switched off.
template <int NG>
struct OcWave {
  int vpos[NG];      // vector block to read: p, or the zero block for a phantom slot
  int hslot[NG];     // LDS slot of W_hub,p for the slots s >= NH (any valid slot for a phantom)
  int ok[NG];        // the slot has a position
};
constexpr int OC_MAXT = 8;       // chain loops are unrolled for up to 2 * OC_MAXT stages + 1: chains of at most 17 positions (plan.hpp checks)
// vector blocks behind the solve vector: 0 junction term, 1..NW hub partial sums, NW + 1 zeros (plan.hpp oc_rext)
template <int NW, int NG, int NH>
__device__ __forceinline__ OcWave<NG> oc_wave(const DevOc &oc, const int *tab, const int wid, const int npad) {
  OcWave<NG> ow;
#pragma unroll
  for (int s = 0; s < NG; s++) {
    const int p = wid + NW * s, pe = p < oc.nbc ? p : oc.nbc - 1;
    ow.ok[s] = p < oc.nbc;
    ow.vpos[s] = p < oc.nbc ? p : npad / BS + NW + 1;
    const int hs = oc_tab(tab, oc.o_pos + 5 * pe + 4);
    ow.hslot[s] = hs >= 0 ? hs : 0;
  }
  return ow;
}
// a chain table entry {position, LDS slot}.  It stays in vector registers: it only feeds LDS addresses, which are per-lane anyway, and
// a readfirstlane would make the wave wait for the entry -- and every LDS read issued before it -- in the middle of a stage
__device__ __forceinline__ int2 oc_pair(const int *tab, int k) { return *reinterpret_cast<const int2 *>(tab + k); }

// x = M^-1 rhs in place on the solve vector R (positions = blocks of 16; behind it the junction term, the waves' hub partials and a
// zero block).
//   F1  the chains, one wave each: t_succ(p) = rhs_succ(p) - W t_p, register to register; the wave of chain E ends with the
//       junction term -W_f,e t_e (f = the shared last element, kept by chain F)
//   F2  the owner of f adds the junction term; every wave, for its own positions p = wid + 4 s: hub partial sum  -sum W_hub,p t_p
//   F3  every wave: t_hub, x_hub = G_hub t_hub
//   B1  every wave, own positions: d_p = G_p t_p - W_hub,p' x_hub
//   B2  the chains backwards: x_p = d_p - W_succ(p),p' x_succ(p)
// Chain tables are {position, LDS slot of the block below it} pairs; the entries, blocks and right-hand sides of the next stage
// are fetched while the current stage multiplies.  HUB: the pattern has an arrow head (has_hub); its first NH blocks per wave are
// in registers (the host lays the plan out for exactly this instance).
// Touch every 128-byte line of [base, base + bytes): one dword per lane and line, results dropped -- brings a slab region that the next
// phase streams into L2 while this wave has nothing else to do (the wave waits for its own loads: nothing outstanding when it moves on).
// Every load writes ONE register that stays an operand of the inline assembly up to the final wait: the compiler does not know that a load
// issued from inline assembly returns its result later, and with a plain output operand it may hand the destination register to something else
// before the load has landed (round 3 found that on an instance with tiles; the unpinned form is gone).
__device__ __forceinline__ void oc_touch_pinned(const void *base, const long bytes, const int lane) {
  const char *p = reinterpret_cast<const char *>(base) + (long)lane * 128;
  int sink = 0;
  for (long o = 0; o < bytes; o += 64 * 128) {
    if (o + (long)lane * 128 < bytes) asm volatile("global_load_dword %0, %1, off" : "+v"(sink) : "v"(p + o) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(sink) : : "memory");
}
// F2, F3 and B1 of the solve on the vector ALUs (layout and cost: above oc_quad_sum).  On entry the chains have run forward and the junction term is
// folded in; on exit every position holds d_p = G_p t_p - W_hub,p' x_hub and a workgroup barrier is due.  Returns this lane's element of x_hub (row
// lane >> 2), which the caller stores once every wave has read the hub's right-hand side.
template <int NW, int NG, int NH, bool HUB>
__device__ __forceinline__ double oc_hub_phases(const DevOc &oc, const double *BL, double *R, double *EXT, const int lane, const OcWave<NG> &ow, const d4 (&G)[NG],
                                                const d4 (&HF)[NH > 0 ? NH : 1], const d4 (&HT)[NH > 0 ? NH : 1], const int wid) {
  const int H = oc.nbc;
  d4 xh = {0, 0, 0, 0};       // x_hub as this lane's operand piece
  double xhd = 0.0;           // ... and as its row's element
  // one pass over the wave's positions serves F2 (-sum W_hub,p t_p; two accumulators: a v_fma_f64 issues every 4 cycles, its result takes 8) and the
  // part of B1 that does not wait for x_hub (G_p t_p): t_p is read once, and the products run in the shadow of the barrier
  // (... where the instance has the registers for NG sums across the barrier: with seven positions per wave -- 168 resident VGPRs -- the fourteen
  // more spilled into the iteration, 13.1 against 12.3 ms on quadrotor N=50 x 4096; there G_p t_p waits for B1)
  constexpr bool EARLY = NG <= 5;
  double p0 = 0.0, p1 = 0.0, acc[NG];
#pragma unroll
  for (int s = 0; s < NG; s++) {
    const d4 t = oc_ldV4(R, ow.vpos[s], lane);
    if (HUB) {
      const d4 a = s < NH ? HF[s < NH ? s : 0] : oc_ldsRow(BL + (long)ow.hslot[s] * BLK, lane);
      if (s & 1) p1 = oc_dot4(a, t, p1); else p0 = oc_dot4(a, t, p0);
    }
    if (EARLY) acc[s] = oc_dot4(G[s], t, 0.0);
  }
  if (HUB) {
    oc_stV(EXT, 1 + wid, lane, oc_quad_sum(p0 + p1));
    bsync<NW>();
    // ---- F3: t_hub, x_hub = G_hub t_hub on every wave (cheaper than a barrier); the result goes through LDS once to become an operand piece (every
    // wave writes the same values to the junction block, which is free by now)
    d4 th = oc_ldV4(R, H, lane);
#pragma unroll
    for (int w = 0; w < NW; w++) th += oc_ldV4(EXT, 1 + w, lane);
    xhd = oc_quad_sum(oc_dot4(oc_ldsRow(BL + (long)oc.ghub_slot * BLK, lane), th, 0.0));
    oc_stV(EXT, 0, lane, xhd);
    xh = oc_ldV4(EXT, 0, lane);
  }
  // ---- B1: d_p = G_p t_p - W_hub,p' x_hub for this wave's positions
#pragma unroll
  for (int s = 0; s < NG; s++) {
    if (!EARLY) acc[s] = oc_dot4(G[s], oc_ldV4(R, ow.vpos[s], lane), 0.0);
    if (HUB) acc[s] = oc_dot4(s < NH ? HT[s < NH ? s : 0] : oc_ldsRowT(BL + (long)ow.hslot[s] * BLK, lane), xh, acc[s]);
    const double d = oc_quad_sum(acc[s]);
    if (ow.ok[s]) oc_stV(R, ow.vpos[s], lane, d);
  }
  return xhd;
}
// Late rows of the right-hand side: the chains need the last vector blocks last, and waves 2, 3 are idle while they run.  So the sweep
// over A' before the solve leaves out the chunk of the last chain positions (oc.at_poll) and the chunk of the hub's rows (oc.at_free, read
// only after the barrier behind the chains); `late(wid)` computes them here, on wave 3 / wave 2.  The chain waves wait for wave 3's rows
// at a fixed place -- the top of trip OC_POLL_TRIP, before any fetch of such a row (the host checks that: plan.hpp oc_late_chunks) -- on a
// ticket in LDS that wave 3 sets to the iteration number once its rows are written.
constexpr int OC_POLL_TRIP = 3;
// The four wave-parallel phases around the chains of a solve with double stages (plan.hpp oc_add_doubles; double stage j = {a, m, b, slot of W_m,a,
// slot of W_b,m}; all blocks negated: every op an accumulation):
//   MODE 0, before the forward chains:   r_b += W_b,m r_m          (the bracket of  t_b = (r_b - W r_m) + (W W) t_a)
//   MODE 1, behind them:                 t_m  = r_m + W_m,a t_a     (the position the double stage jumped over)
//   MODE 2, before the backward chains:  d_a += W_m,a' d_m
//   MODE 3, behind them:                 x_m  = d_m + W_b,m' x_b
// Double stage j goes to wave j mod NW, up to three of a wave's interleaved (independent products hide the MFMA's 52-cycle dependency).  No two
// stages touch the same destination, and a destination of one is never the source of another within a phase.
template <int NW, int MODE>
__device__ __forceinline__ void oc_double_phase(const DevOc &oc, const int *tab, const double *BL, double *R, const OcLane &ln, const int wid) {
  for (int j0 = wid; j0 < oc.ndbl; j0 += 3 * NW) {
    d4 a[3], v[3]; double acc[3]; int dst[3];
#pragma unroll
    for (int u = 0; u < 3; u++) {
      const int j = min(j0 + u * NW, oc.ndbl - 1);
      const int4 e = *reinterpret_cast<const int4 *>(tab + oc.o_dbl + 8 * j);      // {a, m, b, slot of W_m,a}; stays in vector registers: it only feeds LDS addresses
      const int sm = tab[oc.o_dbl + 8 * j + 4];
      const int src = MODE == 0 ? e.y : MODE == 1 ? e.x : MODE == 2 ? e.y : e.z;
      dst[u] = MODE == 0 ? e.z : MODE == 1 ? e.y : MODE == 2 ? e.x : e.y;
      const double *blk = BL + (long)((MODE == 0 || MODE == 3) ? sm : e.w) * BLK;
      a[u] = MODE < 2 ? oc_ldF4(blk, ln) : oc_ldT4(blk, ln);
      v[u] = d4{R[BS * src + ln.k4], R[BS * src + ln.k4 + 4], R[BS * src + ln.k4 + 8], R[BS * src + ln.k4 + 12]};
      acc[u] = R[BS * dst[u] + ln.o4];
    }
#pragma unroll
    for (int K = 0; K < 4; K++)
#pragma unroll
      for (int u = 0; u < 3; u++) acc[u] = oc_mv4(a[u][K], v[u][K], acc[u]);
#pragma unroll
    for (int u = 0; u < 3; u++) if (j0 + u * NW < oc.ndbl) R[BS * dst[u] + ln.o4] = acc[u];
  }
}
// (Two texts of the same solve.  oc_solve is the four-wave form exactly as it was tuned in round 2 -- chain loops unrolled for OC_MAXT trips, two
// position groups -- and oc_solve_long, further down, the form for the eight-wave instances: chain loops of any length, any number of position
// groups, per-solve recomputation of the LDS addresses.  The product of merging them ran 5 % slower on the north-star size: this kernel's
// schedule does not survive a change of its text.)
template <int NW, int NG, int NH, bool HUB, class Late, class Idle>
__device__ __forceinline__ void oc_solve(const DevOc &oc, const int *tab, const double *BL, double *R, const int npad, const OcLane &ln, const OcWave<NG> &ow,
                                         const d4 (&G)[NG], const d4 (&HF)[NH > 0 ? NH : 1], const d4 (&HT)[NH > 0 ? NH : 1], const int wid,
                                         volatile int *ticket, const int iter, Late &&late, Idle &&idle, unsigned long long *stamp = nullptr) {
#ifdef MPCQP_TIMING
#define OC_TS(k) do { if (stamp) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); stamp[k] += t_ - t0_; t0_ = t_; } } while (0)
  unsigned long long t0_ = __builtin_amdgcn_s_memtime();
#else
#define OC_TS(k) __builtin_amdgcn_sched_barrier(0)
#endif
  double *EXT = R + npad;                          // vector blocks behind the solve vector
  // (the chains the SOLVE walks: with double stages they visit every other position; plan.hpp oc_add_doubles)
  const int LE = oc_tab(tab, oc.o_s), LF = oc_tab(tab, oc.o_s + 1);
  const int len = wid == 0 ? LE : LF, cb = wid == 0 ? oc.o_s + 2 : oc.o_s + 2 + 2 * LE;
  const int f = (oc.junc && LF > 0) ? oc_tab(tab, oc.o_s + 2 + 2 * LE + 2 * (LF - 1)) : -1;
  const int H = oc.nbc;
  if (oc.ndbl) { oc_double_phase<NW, 0>(oc, tab, BL, R, ln, wid); bsync<NW>(); }
  // ---- F1
  // Two stages per trip with the roles of the two vector register sets swapped (x -> y -> x): no copy sits between an MFMA result
  // and the MFMAs that read it as their B operand.  The next stage's block and right-hand side are loaded while a stage multiplies;
  // table entries are read two stages ahead.
  if (wid < 2 && len > 0) {
    const int nst = len - 1;                           // stages: stage k multiplies block e[k].y into position e[k + 1].x
    // (table reads past the end are clamped to the last entry: the trip body has no branch, which keeps the compiler's wait counts exact)
    int2 e0 = oc_pair(tab, cb), e1 = oc_pair(tab, cb + 2 * min(1, nst)), e2 = oc_pair(tab, cb + 2 * min(2, nst)), e3 = oc_pair(tab, cb + 2 * min(3, nst));
    // (4-block MFMA: a stage is 4 dependent steps of 52 cycles, its result element goes to LDS and -- as the next stage's operand -- through ds_swizzle)
    d4 x = oc_ldB4(R, e0.x, ln), y = x;
    d4 a = oc_ldF4(BL + (long)e0.y * BLK, ln); double c = oc_ldE4(R, e1.x, ln);
    int k = 0;
#pragma unroll
    for (int trip = 0; trip < OC_MAXT; trip++) {        // fully unrolled: no loop-carried register copies between an MFMA result and its readers
      if (k + 2 > nst) break;
      if (trip == OC_POLL_TRIP && oc.at_poll >= 0) {     // the late rows of the right-hand side are in place from here on
        while (*ticket != iter) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      }
      // stage k: t = c + block(e0.y) x -> position e1.x ; prefetch stage k + 1: block e1.y, rhs of e2.x
      const d4 a1 = oc_ldF4(BL + (long)e1.y * BLK, ln); const double c1 = oc_ldE4(R, e2.x, ln);
      const int2 e4 = oc_pair(tab, cb + 2 * min(k + 4, nst)), e5 = oc_pair(tab, cb + 2 * min(k + 5, nst));     // entries of the next trip
      __builtin_amdgcn_sched_barrier(0);               // the loads above are issued before the multiplies below: a whole stage to land
      const double r0 = oc_mv4x4(a, x, c);
      y = oc_bc4(r0);                                  // (the hand-over first: the LDS pipe is in order, and the next stage waits for exactly this)
      __builtin_amdgcn_sched_barrier(0);
      oc_stB4(R, e1.x, ln, r0);
      // stage k + 1: t = c1 + block(e1.y) y -> position e2.x ; prefetch stage k + 2: block e2.y, rhs of e3.x
      a = oc_ldF4(BL + (long)e2.y * BLK, ln); c = oc_ldE4(R, e3.x, ln);
      __builtin_amdgcn_sched_barrier(0);
      const double r1 = oc_mv4x4(a1, y, c1);
      x = oc_bc4(r1);
      __builtin_amdgcn_sched_barrier(0);
      oc_stB4(R, e2.x, ln, r1);
      e0 = e2; e1 = e3; e2 = e4; e3 = e5; k += 2;
    }
    if (k < nst) {                                     // odd stage count: one more
      const double r0 = oc_mv4x4(a, x, c);
      oc_stB4(R, e1.x, ln, r0);
      x = oc_bc4(r0); e0 = e1;
    }
    // x = t of the chain's last position e0.x
    if (wid == 0 && oc.junc) oc_stB4(EXT, 0, ln, oc_mv4x4(oc_ldF4(BL + (long)e0.y * BLK, ln), x, 0.0));
  } else if (wid >= 2) {
    late(wid);
    if (wid == 3 && oc.at_poll >= 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (ln.o4 == 0) *ticket = iter;
    }
  }
  bsync<NW>();
  if (oc.ndbl) { oc_double_phase<NW, 1>(oc, tab, BL, R, ln, wid); bsync<NW>(); }
  OC_TS(0);
  if (f >= 0 && wid == (f & (NW - 1))) oc_stB4(R, f, ln, oc_ldE4(R, f, ln) + oc_ldE4(EXT, 0, ln));      // t_f complete (its owner reads it back in order)
  const double xhd = oc_hub_phases<NW, NG, NH, HUB>(oc, BL, R, EXT, threadIdx.x & 63, ow, G, HF, HT, wid);
  OC_TS(1);
  bsync<NW>();
  if (oc.ndbl) { oc_double_phase<NW, 2>(oc, tab, BL, R, ln, wid); bsync<NW>(); }
  OC_TS(2);
  // ---- B2
  if (HUB && wid == NW - 1) oc_stV(R, H, threadIdx.x & 63, xhd);     // only now: every wave has read the hub's right-hand side
  if (wid < 2 && len > 0) {
    // stages run down the chain table: stage k computes x_e[k].x = d_e[k].x + block(e[k].y)' v, v = x of the position above it
    int k = len - 2;
    d4 x;
    if (wid == 0 && oc.junc) { x = oc_ldB4(R, f, ln); k = len - 1; }       // chain E starts below f, which chain F's owner finished in B1
    else x = oc_ldB4(R, tab[cb + 2 * (len - 1)], ln);
    if (k >= 0) {
      int2 e0 = oc_pair(tab, cb + 2 * k), e1 = oc_pair(tab, cb + 2 * max(k - 1, 0)), e2 = oc_pair(tab, cb + 2 * max(k - 2, 0)), e3 = oc_pair(tab, cb + 2 * max(k - 3, 0));
      d4 y = x;
      d4 a = oc_ldT4(BL + (long)e0.y * BLK, ln); double c = oc_ldE4(R, e0.x, ln);
#pragma unroll
      for (int trip = 0; trip < OC_MAXT; trip++) {
        if (k < 1) break;
        const d4 a1 = oc_ldT4(BL + (long)e1.y * BLK, ln); const double c1 = oc_ldE4(R, e1.x, ln);
        const int2 e4 = oc_pair(tab, cb + 2 * max(k - 4, 0)), e5 = oc_pair(tab, cb + 2 * max(k - 5, 0));
        __builtin_amdgcn_sched_barrier(0);
        const double r0 = oc_mv4x4(a, x, c);
        y = oc_bc4(r0);
        __builtin_amdgcn_sched_barrier(0);
        oc_stB4(R, e0.x, ln, r0);
        a = oc_ldT4(BL + (long)e2.y * BLK, ln); c = oc_ldE4(R, e2.x, ln);
        __builtin_amdgcn_sched_barrier(0);
        const double r1 = oc_mv4x4(a1, y, c1);
        x = oc_bc4(r1);
        __builtin_amdgcn_sched_barrier(0);
        oc_stB4(R, e1.x, ln, r1);
        e0 = e2; e1 = e3; e2 = e4; e3 = e5; k -= 2;
      }
      if (k == 0) oc_stB4(R, e0.x, ln, oc_mv4x4(a, x, c));
    }
  } else if (wid >= 2) idle(wid);       // (waves 2, 3 have nothing to do in this phase: the caller's prefetch of what the next phase streams)
  bsync<NW>();
  if (oc.ndbl) { oc_double_phase<NW, 3>(oc, tab, BL, R, ln, wid); bsync<NW>(); }
#undef OC_TS
}
// what a wave needs to know about the chain it walks in the solve (oc_solve_long): read once per kernel, not once per ADMM iteration -- the pair's record,
// the chain's table and the junction are three dependent table reads, and with several pairs the junctions of all of them are wanted by their owners
struct OcChain {
  int chainw, role, pjunc, len, cb, f, xs;   // chain wave?  E (0) or F (1); the pair has a junction; positions; table offset; the junction (or -1); the junction term's vector block
  int fv;                                    // lane i: the junction of pair i (or -1)
};
template <int NW>
__device__ __forceinline__ OcChain oc_chain_info(const DevOc &oc, const int *tab, const int wid, const int lane) {
  OcChain c;
  const int np = oc.npair, pair = wid >> 1;
  c.chainw = wid < 2 * np; c.role = wid & 1;
  const int rec = oc.o_pair + 12 * (c.chainw ? pair : 0);
  const int LE = oc_tab(tab, rec + 5), LF = oc_tab(tab, rec + 6);
  c.pjunc = oc_tab(tab, rec + 4);
  c.len = c.role == 0 ? LE : LF; c.cb = c.role == 0 ? oc_tab(tab, rec + 7) : oc_tab(tab, rec + 8);
  c.xs = pair == 0 ? 0 : NW + 1 + pair;
  int fv = -1;
  if (lane < np) { const int ri = oc.o_pair + 12 * lane, lf = tab[ri + 6]; if (tab[ri + 4] && lf > 0) fv = tab[tab[ri + 8] + 2 * (lf - 1)]; }
  c.fv = fv;
  c.f = c.chainw ? __builtin_amdgcn_readlane(fv, c.chainw ? pair : 0) : -1;
  return c;
}
template <int NW, int NG, int NH, bool HUB, class Late, class Idle>
__device__ __forceinline__ void oc_solve_long(const DevOc &oc, const int *tab, const double *BL, double *R, const int npad, const OcLane &ln_, const OcWave<NG> &ow_,
                                         const d4 (&G)[NG], const d4 (&HF)[NH > 0 ? NH : 1], const d4 (&HT)[NH > 0 ? NH : 1], const int wid,
                                         volatile int *ticket, const int iter, Late &&late, Idle &&idle, const OcChain &ci, unsigned long long *stamp = nullptr) {
  // The LDS addresses of a wave's positions and block slots do not change from one ADMM iteration to the next, so the compiler computes them
  // once, outside the iteration -- and in the eight-wave instances, whose registers mostly hold resident blocks, spills them: every use then
  // is a scratch reload with a full wait in front of an LDS read (38 per solve, ~19k cycles per iteration measured).  Their inputs are made
  // opaque here, once per solve: the addresses are recomputed where they are used (a few VALU instructions each).
  OcLane ln = ln_; OcWave<NG> ow = ow_;
  asm volatile("" : "+v"(ln.f4), "+v"(ln.t4), "+v"(ln.k4), "+v"(ln.o4));
#pragma unroll
  for (int s = 0; s < NG; s++) asm volatile("" : "+s"(ow.vpos[s]), "+s"(ow.hslot[s]));
#ifdef MPCQP_TIMING
#define OC_TS(k) do { if (stamp) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); stamp[k] += t_ - t0_; t0_ = t_; } } while (0)
  unsigned long long t0_ = __builtin_amdgcn_s_memtime();
#else
#define OC_TS(k) __builtin_amdgcn_sched_barrier(0)
#endif
  double *EXT = R + npad;                          // vector blocks behind the solve vector
  // (the chains the SOLVE walks: with double stages they visit every other position; plan.hpp oc_add_doubles)
  // chain waves 2 i (E) and 2 i + 1 (F) walk twisted pair i (one pair: the plain or twisted order; several: the dissected order, plan.hpp); the junction term
  // of pair i waits in vector block xs = 0 (pair 0) or NW + 1 + i behind the solve vector
  const int np = oc.npair, role = ci.role, pjunc = ci.pjunc, len = ci.len, cb = ci.cb, f = ci.f, xs = ci.xs;
  const bool chainw = ci.chainw;
  const int H = oc.nbc;
  if (oc.ndbl) { oc_double_phase<NW, 0>(oc, tab, BL, R, ln, wid); bsync<NW>(); }
  // ---- F1
  // Two stages per trip with the roles of the two vector register sets swapped (x -> y -> x): no copy sits between an MFMA result
  // and the MFMAs that read it as their B operand.  The next stage's block and right-hand side are loaded while a stage multiplies;
  // table entries are read two stages ahead.
  if (chainw && len > 0) {
    const int nst = len - 1;                           // stages: stage k multiplies block e[k].y into position e[k + 1].x
    // (table reads past the end are clamped to the last entry: the trip body has no branch, which keeps the compiler's wait counts exact)
    int2 e0 = oc_pair(tab, cb), e1 = oc_pair(tab, cb + 2 * min(1, nst)), e2 = oc_pair(tab, cb + 2 * min(2, nst)), e3 = oc_pair(tab, cb + 2 * min(3, nst));
    // (4-block MFMA: a stage is 4 dependent steps of 52 cycles, its result element goes to LDS and -- as the next stage's operand -- through ds_swizzle)
    d4 x = oc_ldB4(R, e0.x, ln), y = x;
    d4 a = oc_ldF4(BL + (long)e0.y * BLK, ln); double c = oc_ldE4(R, e1.x, ln);
    int k = 0;
    auto trip2 = [&]() {
      // stage k: t = c + block(e0.y) x -> position e1.x ; prefetch stage k + 1: block e1.y, rhs of e2.x
      const d4 a1 = oc_ldF4(BL + (long)e1.y * BLK, ln); const double c1 = oc_ldE4(R, e2.x, ln);
      const int2 e4 = oc_pair(tab, cb + 2 * min(k + 4, nst)), e5 = oc_pair(tab, cb + 2 * min(k + 5, nst));     // entries of the next trip
      __builtin_amdgcn_sched_barrier(0);               // the loads above are issued before the multiplies below: a whole stage to land
      const double r0 = oc_mv4x4(a, x, c);
      y = oc_bc4(r0);                                  // (the hand-over first: the LDS pipe is in order, and the next stage waits for exactly this)
      __builtin_amdgcn_sched_barrier(0);
      oc_stB4(R, e1.x, ln, r0);
      // stage k + 1: t = c1 + block(e1.y) y -> position e2.x ; prefetch stage k + 2: block e2.y, rhs of e3.x
      a = oc_ldF4(BL + (long)e2.y * BLK, ln); c = oc_ldE4(R, e3.x, ln);
      __builtin_amdgcn_sched_barrier(0);
      const double r1 = oc_mv4x4(a1, y, c1);
      x = oc_bc4(r1);
      __builtin_amdgcn_sched_barrier(0);
      oc_stB4(R, e2.x, ln, r1);
      e0 = e2; e1 = e3; e2 = e4; e3 = e5; k += 2;
    };
    // the trip as a loop (chains of any length): the carried values are the block, the right-hand side element, the operand and four table
    // entries, all rewritten inside the trip
#pragma unroll 1
    while (k + 2 <= nst) trip2();
    if (k < nst) {                                     // odd stage count: one more
      const double r0 = oc_mv4x4(a, x, c);
      oc_stB4(R, e1.x, ln, r0);
      x = oc_bc4(r0); e0 = e1;
    }
    // x = t of the chain's last position e0.x
    if (role == 0 && pjunc) oc_stB4(EXT, xs, ln, oc_mv4x4(oc_ldF4(BL + (long)e0.y * BLK, ln), x, 0.0));
  } else if (!chainw) late(wid);
  bsync<NW>();
  if (oc.ndbl) { oc_double_phase<NW, 1>(oc, tab, BL, R, ln, wid); bsync<NW>(); }
  OC_TS(0);
  for (int i = 0; i < np; i++) {      // t_f complete, pair by pair (its owner in the wave-parallel phases reads it back in order)
    const int fi = __builtin_amdgcn_readlane(ci.fv, i);
    if (fi >= 0 && wid == (fi & (NW - 1))) oc_stB4(R, fi, ln, oc_ldE4(R, fi, ln) + oc_ldE4(EXT, i == 0 ? 0 : NW + 1 + i, ln));
  }
  const double xhd = oc_hub_phases<NW, NG, NH, HUB>(oc, BL, R, EXT, threadIdx.x & 63, ow, G, HF, HT, wid);
  OC_TS(1);
  bsync<NW>();
  if (oc.ndbl) { oc_double_phase<NW, 2>(oc, tab, BL, R, ln, wid); bsync<NW>(); }
  OC_TS(2);
  // ---- B2
  if (HUB && wid == NW - 1) oc_stV(R, H, threadIdx.x & 63, xhd);     // only now: every wave has read the hub's right-hand side
  if (chainw && len > 0) {
    // stages run down the chain table: stage k computes x_e[k].x = d_e[k].x + block(e[k].y)' v, v = x of the position above it
    int k = len - 2;
    d4 x;
    if (role == 0 && pjunc) { x = oc_ldB4(R, f, ln); k = len - 1; }       // chain E starts below f, which chain F's owner finished in B1
    else x = oc_ldB4(R, tab[cb + 2 * (len - 1)], ln);
    if (k >= 0) {
      int2 e0 = oc_pair(tab, cb + 2 * k), e1 = oc_pair(tab, cb + 2 * max(k - 1, 0)), e2 = oc_pair(tab, cb + 2 * max(k - 2, 0)), e3 = oc_pair(tab, cb + 2 * max(k - 3, 0));
      d4 y = x;
      d4 a = oc_ldT4(BL + (long)e0.y * BLK, ln); double c = oc_ldE4(R, e0.x, ln);
      auto trip2 = [&]() {
        const d4 a1 = oc_ldT4(BL + (long)e1.y * BLK, ln); const double c1 = oc_ldE4(R, e1.x, ln);
        const int2 e4 = oc_pair(tab, cb + 2 * max(k - 4, 0)), e5 = oc_pair(tab, cb + 2 * max(k - 5, 0));
        __builtin_amdgcn_sched_barrier(0);
        const double r0 = oc_mv4x4(a, x, c);
        y = oc_bc4(r0);
        __builtin_amdgcn_sched_barrier(0);
        oc_stB4(R, e0.x, ln, r0);
        a = oc_ldT4(BL + (long)e2.y * BLK, ln); c = oc_ldE4(R, e2.x, ln);
        __builtin_amdgcn_sched_barrier(0);
        const double r1 = oc_mv4x4(a1, y, c1);
        x = oc_bc4(r1);
        __builtin_amdgcn_sched_barrier(0);
        oc_stB4(R, e1.x, ln, r1);
        e0 = e2; e1 = e3; e2 = e4; e3 = e5; k -= 2;
      };
#pragma unroll 1
      while (k >= 1) trip2();
      if (k == 0) oc_stB4(R, e0.x, ln, oc_mv4x4(a, x, c));
    }
  } else if (!chainw) idle(wid);       // (waves without a chain have nothing to do in this phase: the caller's prefetch of what the next phase streams)
  bsync<NW>();
  if (oc.ndbl) { oc_double_phase<NW, 3>(oc, tab, BL, R, ln, wid); bsync<NW>(); }
#undef OC_TS
}

// =========================================================================================================
// The solve with its CHAINS on the vector ALUs as well (round 4).  A chain stage  t_next = rhs_next - W t  is a dependent 16 x 16 mat-vec.  On the
// matrix cores it is four v_mfma_f64_4x4x4_4b_f64 behind each other, 52 cycles each (the f64 4-block MFMA occupies the matrix pipe for ~48 cycles
// whether the next one depends on it or not), then eight ds_swizzle to turn the result into the next operand: 330 - 400 cycles in the kernels.  On the
// vector ALUs, in the row-piece layout (lane l = (r = l >> 2, j = l & 3) holds W[r][4 j .. 4 j + 3]): four dependent v_fma_f64 (a wave64 f64 fma issues
// in 4 cycles), a quad sum by DPP, and eight ds_bpermute that fetch the next operand piece x[4 j .. 4 j + 3] from the quads 4 j .. 4 j + 3 -- the same
// trip through the LDS crossbar as the swizzles, behind ~90 cycles of arithmetic instead of ~210.  A block is read from its swizzled LDS image as one
// 32-byte row piece (forward) or four 8-byte entries (backward); the right-hand side element joins the sum in the lane with j = 0.
// Same phases, barriers and tables as oc_solve / oc_solve_long; chain loops of any length.
// MEASURED SLOWER, not the default (build with -DMPCQP_VALU_CHAINS): quadrotor N=20 x 8192 iteration kernel 5.67 against 5.36 ms, N=50 x 4096 14.8 against
// 12.3, cart-pole N=100 x 4096 26.8 against 25.8 -- the eight ds_bpermute (a full crossbar gather with an address register each) and the two dependent
// DPP rounds of the quad sum cost more than the matrix pipe's 4 x 52 cycles save; the hub and diagonal phases, which have no hand-over, did gain.
__device__ __forceinline__ double oc_bperm(const double v, const int byte_addr) {
  union { double d; int i[2]; } a, r;
  a.d = v;
  r.i[0] = __builtin_amdgcn_ds_bpermute(byte_addr, a.i[0]); r.i[1] = __builtin_amdgcn_ds_bpermute(byte_addr, a.i[1]);
  return r.d;
}
// the operand piece of the next stage from the row sums of this one: entry 4 j + i sits in (every lane of) quad 4 j + i = lanes 16 j + 4 i ...
__device__ __forceinline__ d4 oc_next_piece(const double y, const int lane) {
  const int b = 64 * (lane & 3);       // byte address of lane 16 j
  return d4{oc_bperm(y, b), oc_bperm(y, b + 16), oc_bperm(y, b + 32), oc_bperm(y, b + 48)};
}
template <int NW, int NG, int NH, bool HUB, class Idle>
__device__ __forceinline__ void oc_solve_v(const DevOc &oc, const int *tab, const double *BL, double *R, const int npad, const OcLane &ln, const OcWave<NG> &ow_,
                                           const d4 (&G)[NG], const d4 (&HF)[NH > 0 ? NH : 1], const d4 (&HT)[NH > 0 ? NH : 1], const int wid, Idle &&idle,
                                           unsigned long long *stamp = nullptr) {
  // (the LDS addresses of a wave's positions are recomputed per solve from opaque inputs: hoisted out of the ADMM loop they are spilled -- see oc_solve_long)
  OcWave<NG> ow = ow_;
#pragma unroll
  for (int s = 0; s < NG; s++) asm volatile("" : "+s"(ow.vpos[s]), "+s"(ow.hslot[s]));
#ifdef MPCQP_TIMING
#define OC_TS(k) do { if (stamp) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); stamp[k] += t_ - t0_; t0_ = t_; } } while (0)
  unsigned long long t0_ = __builtin_amdgcn_s_memtime();
#else
#define OC_TS(k) __builtin_amdgcn_sched_barrier(0)
#endif
  const int lane = threadIdx.x & 63;
  const double jz = (lane & 3) == 0 ? 1.0 : 0.0;     // the right-hand side element joins the row sum in one lane of the quad
  double *EXT = R + npad;                            // vector blocks behind the solve vector
  const int LE = oc_tab(tab, oc.o_s), LF = oc_tab(tab, oc.o_s + 1);
  const int len = wid == 0 ? LE : LF, cb = wid == 0 ? oc.o_s + 2 : oc.o_s + 2 + 2 * LE;
  const int f = (oc.junc && LF > 0) ? oc_tab(tab, oc.o_s + 2 + 2 * LE + 2 * (LF - 1)) : -1;
  const int H = oc.nbc;
  if (oc.ndbl) { oc_double_phase<NW, 0>(oc, tab, BL, R, ln, wid); bsync<NW>(); }
  // ---- F1: the chains, one wave each.  Two stages per trip with the roles of the two operand register sets swapped; the next stage's block, right-hand
  // side element and table entries are fetched while a stage multiplies (table reads past the end are clamped to the last entry: no branch in a trip)
  if (wid < 2 && len > 0) {
    const int nst = len - 1;                           // stage k multiplies block e[k].y into position e[k + 1].x
    int2 e0 = oc_pair(tab, cb), e1 = oc_pair(tab, cb + 2 * min(1, nst)), e2 = oc_pair(tab, cb + 2 * min(2, nst)), e3 = oc_pair(tab, cb + 2 * min(3, nst));
    d4 x = oc_ldV4(R, e0.x, lane), y = x;
    d4 a = oc_ldsRow(BL + (long)e0.y * BLK, lane); double c = R[BS * e1.x + (lane >> 2)];
    int k = 0;
    auto trip2 = [&]() {
      const d4 a1 = oc_ldsRow(BL + (long)e1.y * BLK, lane); const double c1 = R[BS * e2.x + (lane >> 2)];
      const int2 e4 = oc_pair(tab, cb + 2 * min(k + 4, nst)), e5 = oc_pair(tab, cb + 2 * min(k + 5, nst));
      __builtin_amdgcn_sched_barrier(0);
      const double r0 = oc_quad_sum(oc_dot4(a, x, jz * c));
      y = oc_next_piece(r0, lane);                     // (the hand-over first: the LDS pipe is in order, and the next stage waits for exactly this)
      __builtin_amdgcn_sched_barrier(0);
      oc_stV(R, e1.x, lane, r0);
      a = oc_ldsRow(BL + (long)e2.y * BLK, lane); c = R[BS * e3.x + (lane >> 2)];
      __builtin_amdgcn_sched_barrier(0);
      const double r1 = oc_quad_sum(oc_dot4(a1, y, jz * c1));
      x = oc_next_piece(r1, lane);
      __builtin_amdgcn_sched_barrier(0);
      oc_stV(R, e2.x, lane, r1);
      e0 = e2; e1 = e3; e2 = e4; e3 = e5; k += 2;
    };
#pragma unroll 1
    while (k + 2 <= nst) trip2();
    if (k < nst) {                                     // odd stage count: one more
      const double r0 = oc_quad_sum(oc_dot4(a, x, jz * c));
      oc_stV(R, e1.x, lane, r0);
      x = oc_next_piece(r0, lane); e0 = e1;
    }
    // x = t of the chain's last position e0.x; the wave of chain E ends with the junction term
    if (wid == 0 && oc.junc) oc_stV(EXT, 0, lane, oc_quad_sum(oc_dot4(oc_ldsRow(BL + (long)e0.y * BLK, lane), x, 0.0)));
  }
  bsync<NW>();
  if (oc.ndbl) { oc_double_phase<NW, 1>(oc, tab, BL, R, ln, wid); bsync<NW>(); }
  OC_TS(0);
  if (f >= 0 && wid == (f & (NW - 1))) oc_stV(R, f, lane, R[BS * f + (lane >> 2)] + EXT[lane >> 2]);      // t_f complete (its owner reads it back in order)
  const double xhd = oc_hub_phases<NW, NG, NH, HUB>(oc, BL, R, EXT, lane, ow, G, HF, HT, wid);
  OC_TS(1);
  bsync<NW>();
  if (oc.ndbl) { oc_double_phase<NW, 2>(oc, tab, BL, R, ln, wid); bsync<NW>(); }
  OC_TS(2);
  // ---- B2: the chains backwards: stage k computes x_e[k].x = d_e[k].x + block(e[k].y)' v, v = x of the position above it
  if (HUB && wid == NW - 1) oc_stV(R, H, lane, xhd);     // only now: every wave has read the hub's right-hand side
  if (wid < 2 && len > 0) {
    int k = len - 2;
    d4 x;
    if (wid == 0 && oc.junc) { x = oc_ldV4(R, f, lane); k = len - 1; }       // chain E starts below f, which chain F's owner finished in B1
    else x = oc_ldV4(R, tab[cb + 2 * (len - 1)], lane);
    if (k >= 0) {
      int2 e0 = oc_pair(tab, cb + 2 * k), e1 = oc_pair(tab, cb + 2 * max(k - 1, 0)), e2 = oc_pair(tab, cb + 2 * max(k - 2, 0)), e3 = oc_pair(tab, cb + 2 * max(k - 3, 0));
      d4 y = x;
      d4 a = oc_ldsRowT(BL + (long)e0.y * BLK, lane); double c = R[BS * e0.x + (lane >> 2)];
      auto trip2 = [&]() {
        const d4 a1 = oc_ldsRowT(BL + (long)e1.y * BLK, lane); const double c1 = R[BS * e1.x + (lane >> 2)];
        const int2 e4 = oc_pair(tab, cb + 2 * max(k - 4, 0)), e5 = oc_pair(tab, cb + 2 * max(k - 5, 0));
        __builtin_amdgcn_sched_barrier(0);
        const double r0 = oc_quad_sum(oc_dot4(a, x, jz * c));
        y = oc_next_piece(r0, lane);
        __builtin_amdgcn_sched_barrier(0);
        oc_stV(R, e0.x, lane, r0);
        a = oc_ldsRowT(BL + (long)e2.y * BLK, lane); c = R[BS * e2.x + (lane >> 2)];
        __builtin_amdgcn_sched_barrier(0);
        const double r1 = oc_quad_sum(oc_dot4(a1, y, jz * c1));
        x = oc_next_piece(r1, lane);
        __builtin_amdgcn_sched_barrier(0);
        oc_stV(R, e1.x, lane, r1);
        e0 = e2; e1 = e3; e2 = e4; e3 = e5; k -= 2;
      };
#pragma unroll 1
      while (k >= 1) trip2();
      if (k == 0) oc_stV(R, e0.x, lane, oc_quad_sum(oc_dot4(a, x, jz * c)));
    }
  } else if (wid >= 2) idle(wid);       // (the other waves have nothing to do in this phase: the caller's prefetch of what the next phase streams)
  bsync<NW>();
  if (oc.ndbl) { oc_double_phase<NW, 3>(oc, tab, BL, R, ln, wid); bsync<NW>(); }
#undef OC_TS
}
