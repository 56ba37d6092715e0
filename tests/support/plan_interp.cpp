// plan_interp.cpp -- TEST INFRASTRUCTURE ONLY (built by __graft_entry__.build(), loaded by tests/test_plan.py).
//
// Scalar CPU interpreter of the host-side plan (optimal_control_problem_amd/csrc/plan.hpp): executes the
// ELL layouts, the assembly recipe, the block-Cholesky op list and the forward/backward block streams
// exactly as the HIP kernel is meant to, so that indexing mistakes in the plan show up in the CPU test
// suite. It is never linked into libmpcqp.so and is not a fallback for it.
#include <cmath>
#include <cstring>
#include <vector>

#include "../../optimal_control_problem_amd/csrc/plan.hpp"

using namespace mpcqp;

static void gemm_abt(const double *A, const double *B, double *C, double sign) {  // C += sign * A B'
  for (int i = 0; i < BS; i++) for (int j = 0; j < BS; j++) {
    double s = 0;
    for (int k = 0; k < BS; k++) s += A[i * BS + k] * B[j * BS + k];
    C[i * BS + j] += sign * s;
  }
}

static bool potrf_inv(double *S, double *ST) {
  double L[BLK] = {0}, X[BLK] = {0};
  for (int j = 0; j < BS; j++) {
    double d = S[j * BS + j];
    for (int k = 0; k < j; k++) d -= L[j * BS + k] * L[j * BS + k];
    if (!(d > 0)) return false;
    d = std::sqrt(d); L[j * BS + j] = d;
    for (int i = j + 1; i < BS; i++) {
      double s = S[i * BS + j];
      for (int k = 0; k < j; k++) s -= L[i * BS + k] * L[j * BS + k];
      L[i * BS + j] = s / d;
    }
  }
  for (int c = 0; c < BS; c++) for (int i = c; i < BS; i++) {
    if (i == c) X[i * BS + c] = 1.0 / L[i * BS + i];
    else { double s = 0; for (int k = c; k < i; k++) s += L[i * BS + k] * X[k * BS + c]; X[i * BS + c] = -s / L[i * BS + i]; }
  }
  for (int i = 0; i < BS; i++) for (int j = 0; j < BS; j++) { S[i * BS + j] = X[i * BS + j]; ST[j * BS + i] = X[i * BS + j]; }
  return true;
}

static void run_stream(const std::vector<double> &blk, const std::vector<int> &ops, std::vector<double> &vec) {
  for (size_t t = 0; t < ops.size(); t++) {
    int op = ops[t], kind = op & 1, s = (op >> 1) & 0x7fff, d = op >> 16;
    double out[BS];
    for (int r = 0; r < BS; r++) { double a = 0; for (int c = 0; c < BS; c++) a += blk[t * BLK + r * BS + c] * vec[s * BS + c]; out[r] = a; }
    for (int r = 0; r < BS; r++) vec[d * BS + r] = kind ? vec[d * BS + r] - out[r] : out[r];
  }
}

static void ell_fill(const Ell &e, const double *in, std::vector<double> &val) {
  val.assign(e.entries(), 0.0);
  for (long p = 0; p < e.entries(); p++) if (e.src[p] >= 0) val[p] = in[e.src[p]];
}
static void ell_spmv(const Ell &e, const std::vector<double> &val, const std::vector<double> &in, std::vector<double> &out) {
  out.assign((size_t)e.nchunks * WAVE, 0.0);
  for (int c = 0; c < e.nchunks; c++) for (int lane = 0; lane < WAVE; lane++) {
    double a = 0;
    for (int s = e.chunk_off[c]; s < e.chunk_off[c + 1]; s++) { long p = (long)s * WAVE + lane; a += val[p] * in[e.idx[p]]; }
    out[c * WAVE + lane] = a;
  }
}

extern "C" {

// info[0..7] = npad, mpad, nb, nblk, nT, n_fac, ordering, lds_bytes
int plan_describe(int n, int m, const int *Pp, const int *Pi, const int *Ap, const int *Ai, int force_ordering, long *info, int *pos_out) {
  Plan pl = build_plan(n, m, Pp, Pi, Ap, Ai, force_ordering);
  if (!pl.error.empty()) return 1;
  info[0] = pl.npad; info[1] = pl.mpad; info[2] = pl.nb; info[3] = pl.nblk; info[4] = pl.nT; info[5] = (long)pl.fac.size();
  info[6] = pl.ordering; info[7] = lds_bytes(pl);
  if (pos_out) for (int j = 0; j < n; j++) pos_out[j] = pl.pos[j];
  return 0;
}

// Executes the plan for one QP's matrices: x = (P + sigma I + A' diag(rho) A)^-1 rhs, plus the three ELL
// products Ax = A*xin, Atw = A'*win, Px = P*xin (xin/Atw/Px in original variable order).
// returns 0 ok, 1 plan error, 2 not positive definite
int plan_execute(int n, int m, const int *Pp, const int *Pi, const int *Ap, const int *Ai, int force_ordering,
                 const double *Pval, const double *Aval, const double *rho, double sigma,
                 const double *rhs, double *sol, const double *xin, const double *win, double *Ax, double *Atw, double *Px) {
  Plan pl = build_plan(n, m, Pp, Pi, Ap, Ai, force_ordering);
  if (!pl.error.empty()) return 1;
  std::vector<double> vA, vAt, vP;
  ell_fill(pl.A, Aval, vA); ell_fill(pl.At, Aval, vAt); ell_fill(pl.P, Pval, vP);
  // ELL products
  std::vector<double> xp(pl.npad, 0.0), wp(pl.mpad, 0.0), o;
  for (int j = 0; j < n; j++) xp[pl.pos[j]] = xin[j];
  for (int i = 0; i < m; i++) wp[i] = win[i];
  ell_spmv(pl.A, vA, xp, o); for (int i = 0; i < m; i++) Ax[i] = o[i];
  ell_spmv(pl.At, vAt, wp, o); for (int j = 0; j < n; j++) Atw[j] = o[pl.pos[j]];
  ell_spmv(pl.P, vP, xp, o); for (int j = 0; j < n; j++) Px[j] = o[pl.pos[j]];
  // dvec
  std::vector<double> dvec(pl.npad, 1.0);
  for (int c = 0; c < pl.At.nchunks; c++) for (int lane = 0; lane < WAVE; lane++) {
    int t = c * WAVE + lane; if (t >= pl.npad) continue;
    double a = 0;
    for (int s = pl.At.chunk_off[c]; s < pl.At.chunk_off[c + 1]; s++) { long p = (long)s * WAVE + lane; if (pl.At.flag[p]) a += rho[pl.At.idx[p]] * vAt[p] * vAt[p]; }
    dvec[t] = pl.perm[t] >= 0 ? sigma + a : 1.0;
  }
  // T
  std::vector<double> T((size_t)std::max(pl.nT, 1) * BLK, 0.0);
  for (int c = 0; c < pl.A.nchunks; c++) for (int lane = 0; lane < WAVE; lane++) {
    int i = c * WAVE + lane; if (i >= m) continue;
    for (int s = pl.A.chunk_off[c]; s < pl.A.chunk_off[c + 1]; s++) { long p = (long)s * WAVE + lane; if (pl.tpos[p] >= 0) T[pl.tpos[p]] = vA[p] * std::sqrt(rho[i]); }
  }
  // assemble in the MFMA C layout order
  std::vector<double> Lf((size_t)pl.nblk * BLK, 0.0), Lb((size_t)pl.nblk * BLK, 0.0);
  for (int b = 0; b < pl.nblk; b++) {
    double *C = &Lf[(size_t)b * BLK];
    for (int g = pl.asm_ptr[b]; g < pl.asm_ptr[b + 1]; g++) gemm_abt(&T[(size_t)pl.asm_a[g] * BLK], &T[(size_t)pl.asm_b[g] * BLK], C, 1.0);
    for (int g = 0; g < 4; g++) for (int lane = 0; lane < WAVE; lane++) {
      int row = (lane >> 4) + 4 * g, col = lane & 15;
      int pi = pl.asm_pidx[(size_t)b * BLK + g * WAVE + lane];
      if (pi >= 0) C[row * BS + col] += vP[pi];
      if (pl.blk_diag[b] >= 0 && row == col) C[row * BS + col] += dvec[pl.blk_diag[b] * BS + row];
    }
  }
  for (auto &op : pl.fac) {
    double *dst = &Lf[(size_t)op.dst * BLK];
    if (op.type == FAC_SUB) gemm_abt(&Lf[(size_t)op.a * BLK], &Lf[(size_t)op.b * BLK], dst, -1.0);
    else if (op.type == FAC_POTRF) { if (!potrf_inv(dst, &Lb[(size_t)pl.bwd_of[op.dst] * BLK])) return 2; }
    else {
      double tmp[BLK] = {0};
      gemm_abt(dst, &Lf[(size_t)op.a * BLK], tmp, 1.0);
      double *bt = &Lb[(size_t)pl.bwd_of[op.dst] * BLK];
      for (int i = 0; i < BS; i++) for (int j = 0; j < BS; j++) { dst[i * BS + j] = tmp[i * BS + j]; bt[j * BS + i] = tmp[i * BS + j]; }
    }
  }
  std::vector<double> v(pl.npad, 0.0);
  for (int j = 0; j < n; j++) v[pl.pos[j]] = rhs[j];
  run_stream(Lf, pl.fwd_ops, v);
  run_stream(Lb, pl.bwd_ops, v);
  for (int j = 0; j < n; j++) sol[j] = v[pl.pos[j]];
  return 0;
}

}  // extern "C"

// symmetric sweep: in-place inverse of an SPD 16x16 block (what the resident kernel's wave does)
static bool sweep_inverse(double *a) {
  for (int k = 0; k < BS; k++) {
    double row[BS];
    for (int c = 0; c < BS; c++) row[c] = a[k * BS + c];
    const double d = row[k];
    if (!(d > 0)) return false;
    const double p = 1.0 / d;
    for (int r = 0; r < BS; r++) for (int c = 0; c < BS; c++) {
      const double colk = row[r];
      if (r != k && c != k) a[r * BS + c] -= colk * row[c] * p;
      else if (r == k && c != k) a[r * BS + c] = row[c] * p;
      else if (r != k && c == k) a[r * BS + c] = colk * p;
      else a[r * BS + c] = -p;
    }
  }
  for (int i = 0; i < BLK; i++) a[i] = -a[i];
  return true;
}

extern "C" {

// Resident-variant plan: block LDL' factor plan + phase schedule over nw waves, executed serially with race checks.
// returns 0 ok, 1 plan error, 2 not positive definite, 3 schedule hazard (two waves touch one vector block in a phase)
int plan_execute_res(int n, int m, const int *Pp, const int *Pi, const int *Ap, const int *Ai, int nw,
                     const double *Pval, const double *Aval, const double *rho, double sigma,
                     const double *rhs, double *sol, long *info) {
  const bool split = nw >= 10000;                 // + 10000: split long accumulation runs (what the global-block kernels use)
  nw %= 10000;
  const int force = nw >= 100 ? nw / 100 : -1;
  nw %= 100;
  Plan pl = build_plan(n, m, Pp, Pi, Ap, Ai, force);
  if (!pl.error.empty()) return 1;
  ResPlan rp = build_res_plan(pl, nw, split);
  if (info) { info[0] = rp.ntemp + 100 * pl.ordering; info[1] = rp.nphase; info[2] = lds_bytes_res(pl, rp); int nb = 0; for (int b : rp.s_bar) nb += b; info[3] = nb; }
  // the segment compression must expand back to exactly the record list (offsets, kinds, flush and barrier points)
  for (int w = 0; w < nw; w++) {
    std::vector<int> ex;   // expanded {b, s, d, T|SET, flush, bar} per op; barriers-only as {-1,...}
    bool open_run = false;
    for (int g = rp.g_ptr[w]; g < rp.g_ptr[w + 1]; g++) {
      const int *sg = &rp.g_seg[8 * g];
      if (sg[3] & SG_NOP) { ex.insert(ex.end(), {-1, 0, 0, 0, 0, 1}); continue; }
      for (int k = 0; k < sg[4]; k++) {
        const bool last = k + 1 == sg[4];
        const int flush = (sg[3] & SG_EACH) ? 1 : (last && (sg[3] & SG_END) ? 1 : 0);
        ex.insert(ex.end(), {sg[0] + k * sg[5], sg[1] + k * sg[6], sg[2] + ((sg[3] & SG_EACH) ? k * sg[7] : 0), sg[3] & (SG_T | SG_SET), flush, last && (sg[3] & SG_BAR) ? 1 : 0});
        open_run = !flush;
      }
    }
    if (open_run) return 4;
    size_t e = 0;
    for (int q = rp.r_ptr[w]; q < rp.r_ptr[w + 1]; q++, e += 6) {
      const int *rc = &rp.r_rec[4 * q];
      if (e + 6 > ex.size()) return 4;
      if (rc[3] & RF_NOP) { if (ex[e] != -1) return 4; continue; }
      const int tf = ((rc[3] & RF_T) ? SG_T : 0) | ((rc[3] & RF_SET) ? SG_SET : 0);
      if (ex[e] != rc[0] || ex[e + 1] != rc[1] || ex[e + 2] != rc[2] || ex[e + 3] != tf || ex[e + 4] != ((rc[3] & RF_FLUSH) ? 1 : 0) ||
          ex[e + 5] != ((rc[3] & RF_BAR) ? 1 : 0)) return 4;
    }
    if (e != ex.size()) return 4;
  }
  if (info) info[3] = (long)rp.g_seg.size() / 8 * 1000 + info[3];   // segments * 1000 + barriers
  std::vector<double> vA, vAt, vP;
  ell_fill(pl.A, Aval, vA); ell_fill(pl.At, Aval, vAt); ell_fill(pl.P, Pval, vP);
  std::vector<double> dvec(pl.npad, 1.0);
  for (int c = 0; c < pl.At.nchunks; c++) for (int lane = 0; lane < WAVE; lane++) {
    int t = c * WAVE + lane; if (t >= pl.npad) continue;
    double a = 0;
    for (int s = pl.At.chunk_off[c]; s < pl.At.chunk_off[c + 1]; s++) { long p = (long)s * WAVE + lane; if (pl.At.flag[p]) a += rho[pl.At.idx[p]] * vAt[p] * vAt[p]; }
    dvec[t] = pl.perm[t] >= 0 ? sigma + a : 1.0;
  }
  std::vector<double> T((size_t)std::max(pl.nT, 1) * BLK, 0.0);
  for (int c = 0; c < pl.A.nchunks; c++) for (int lane = 0; lane < WAVE; lane++) {
    int i = c * WAVE + lane; if (i >= m) continue;
    for (int s = pl.A.chunk_off[c]; s < pl.A.chunk_off[c + 1]; s++) { long p = (long)s * WAVE + lane; if (pl.tpos[p] >= 0) T[pl.tpos[p]] = vA[p] * std::sqrt(rho[i]); }
  }
  std::vector<double> S((size_t)pl.nblk * BLK, 0.0), tmp((size_t)std::max(rp.ntemp, 1) * BLK, 0.0);
  for (int b = 0; b < pl.nblk; b++) {
    double *C = &S[(size_t)b * BLK];
    for (int g = pl.asm_ptr[b]; g < pl.asm_ptr[b + 1]; g++) gemm_abt(&T[(size_t)pl.asm_a[g] * BLK], &T[(size_t)pl.asm_b[g] * BLK], C, 1.0);
    for (int g = 0; g < 4; g++) for (int lane = 0; lane < WAVE; lane++) {
      int row = (lane >> 4) + 4 * g, col = lane & 15;
      int pi = pl.asm_pidx[(size_t)b * BLK + g * WAVE + lane];
      if (pi >= 0) C[row * BS + col] += vP[pi];
      if (pl.blk_diag[b] >= 0 && row == col) C[row * BS + col] += dvec[pl.blk_diag[b] * BS + row];
    }
  }
  // level-parallel factor plan (what the resident kernel executes); same-destination Schur updates must share a wave
  tmp.assign((size_t)std::max(rp.ntemp, 1) * BLK, 0.0);
  std::vector<int> pend_slot; std::vector<int> pend_tmp;
  for (int lev = 0; lev < rp.nlev; lev++) {
    for (int ci = rp.lv_ptr[lev]; ci < rp.lv_ptr[lev + 1]; ci++) if (!sweep_inverse(&S[(size_t)rp.lv_diag[ci] * BLK])) return 2;
    for (size_t a = 0; a < pend_slot.size(); a++) std::memcpy(&S[(size_t)pend_slot[a] * BLK], &tmp[(size_t)pend_tmp[a] * BLK], BLK * sizeof(double));
    pend_slot.clear(); pend_tmp.clear();
    const int w0 = rp.lw_ptr[lev], nwk = rp.lw_ptr[lev + 1] - w0;
    if (nwk > rp.ntemp) return 4;
    std::vector<double> tnew((size_t)std::max(nwk, 1) * BLK, 0.0);
    for (int a = 0; a < nwk; a++) gemm_abt(&S[(size_t)rp.lw_slot[w0 + a] * BLK], &S[(size_t)rp.lw_g[w0 + a] * BLK], &tnew[(size_t)a * BLK], 1.0);
    for (int a = 0; a < nwk; a++) std::memcpy(&tmp[(size_t)a * BLK], &tnew[(size_t)a * BLK], BLK * sizeof(double));
    std::map<int, int> dst_wave;
    for (int w = 0; w < nw; w++) for (int u = rp.lu_ptr[lev * nw + w]; u < rp.lu_ptr[lev * nw + w + 1]; u++) {
      auto it = dst_wave.find(rp.lu_dst[u]);
      if (it != dst_wave.end() && it->second != w) return 3;
      dst_wave[rp.lu_dst[u]] = w;
      for (int a = 0; a < nwk; a++) if (rp.lu_dst[u] == rp.lw_slot[w0 + a]) return 3;   // an update may not hit a slot still holding S_IK of this level
      gemm_abt(&tmp[(size_t)rp.lu_tmp[u] * BLK], &S[(size_t)rp.lu_b[u] * BLK], &S[(size_t)rp.lu_dst[u] * BLK], -1.0);
    }
    for (int a = 0; a < nwk; a++) { pend_slot.push_back(rp.lw_slot[w0 + a]); pend_tmp.push_back(a); }
  }
  for (size_t a = 0; a < pend_slot.size(); a++) std::memcpy(&S[(size_t)pend_slot[a] * BLK], &tmp[(size_t)pend_tmp[a] * BLK], BLK * sizeof(double));
  std::vector<double> v(pl.npad + rp.rext, 0.0);      // + the zero-initialised partial sums of split accumulation runs
  for (int j = 0; j < n; j++) v[pl.pos[j]] = rhs[j];
  if (rp.nconst) {                                     // the constant block -I behind the factor blocks
    S.resize(((size_t)pl.nblk + 1) * BLK, 0.0);
    for (int e = 0; e < BLK; e++) S[(size_t)pl.nblk * BLK + e] = (e / BS == e % BS) ? -1.0 : 0.0;
  }
  // hazard check, independent of the scheduler: since the last workgroup barrier no wave may read or overwrite a
  // vector block another wave has written, nor overwrite one another wave has read
  {
    std::vector<int> wby(pl.nb + nw, -1), rby(pl.nb + nw, 0);
    for (int p = 0; p < rp.nphase; p++) {
      for (int w = 0; w < nw; w++) for (int q = rp.s_ptr[p * nw + w]; q < rp.s_ptr[p * nw + w + 1]; q++) {
        unsigned op = (unsigned)rp.s_ops[q]; int src = (op >> 14) & 0x1ff, dst = op >> 23;
        if ((wby[src] >= 0 && wby[src] != w) || (wby[dst] >= 0 && wby[dst] != w) || (rby[dst] & ~(1 << w))) return 3;
        wby[dst] = w; rby[src] |= 1 << w; rby[dst] |= 1 << w;
      }
      if (rp.s_bar[p] || nw == 1) { std::fill(wby.begin(), wby.end(), -1); std::fill(rby.begin(), rby.end(), 0); }
    }
    if (nw > 1 && !rp.s_bar[rp.nphase - 1]) return 3;
  }
  for (int p = 0; p < rp.nphase; p++) {
    for (int w = 0; w < nw; w++) for (int q = rp.s_ptr[p * nw + w]; q < rp.s_ptr[p * nw + w + 1]; q++) {
      unsigned op = (unsigned)rp.s_ops[q]; int kind = op & 3, slot = (op >> 2) & 0xfff, src = (op >> 14) & 0x1ff, dst = op >> 23;
      const double *B = &S[(size_t)slot * BLK];
      double out[BS];
      for (int r = 0; r < BS; r++) { double a = 0; for (int c = 0; c < BS; c++) a += (kind == SOP_SUBT ? B[c * BS + r] : B[r * BS + c]) * v[src * BS + c]; out[r] = a; }
      for (int r = 0; r < BS; r++) v[dst * BS + r] = kind == SOP_SET ? out[r] : v[dst * BS + r] - out[r];
    }
  }
  for (int j = 0; j < n; j++) sol[j] = v[pl.pos[j]];
  return 0;
}

}  // extern "C"

// Bounds of everything the kernel will address through the solve schedule: every record and every expanded segment must stay
// inside the block array ((nblk + nconst) blocks) and the solve vector (npad + rext doubles).  Returns 0 ok, else a code.
extern "C" int plan_check_segments(int n, int m, const int *Pp, const int *Pi, const int *Ap, const int *Ai, int nw_enc, long *info) {
  const bool split = nw_enc >= 10000;
  nw_enc %= 10000;
  const int nw = nw_enc % 100, force = nw_enc / 100 - 1;
  Plan pl = build_plan(n, m, Pp, Pi, Ap, Ai, force, true);
  if (!pl.error.empty()) return 1;
  ResPlan rp = build_res_plan(pl, nw, split);
  const long bmax = ((long)pl.nblk + rp.nconst) * BLK * 8, vmax = ((long)pl.npad + rp.rext) * 8;
  for (size_t i = 0; i + 3 < rp.r_rec.size(); i += 4) {
    const int b = rp.r_rec[i], s = rp.r_rec[i + 1], d = rp.r_rec[i + 2], f = rp.r_rec[i + 3];
    if (f & RF_NOP) continue;
    if (b < 0 || b + BLK * 8 > bmax || s < 0 || s + BS * 8 > vmax || d < 0 || d + BS * 8 > vmax) return 2;
  }
  long nseg = 0;
  for (size_t i = 0; i + 7 < rp.g_seg.size(); i += 8) {
    const int b0 = rp.g_seg[i], s0 = rp.g_seg[i + 1], d0 = rp.g_seg[i + 2], fl = rp.g_seg[i + 3], cnt = rp.g_seg[i + 4];
    const int db = rp.g_seg[i + 5], ds = rp.g_seg[i + 6], dd = rp.g_seg[i + 7];
    nseg++;
    if (fl & SG_NOP) continue;
    for (int k = 0; k < cnt; k++) {
      const long b = b0 + (long)k * db, s = s0 + (long)k * ds, d = (fl & SG_EACH) ? d0 + (long)k * dd : d0;
      if (b < 0 || b + BLK * 8 > bmax || s < 0 || s + BS * 8 > vmax || d < 0 || d + BS * 8 > vmax) return 3;
    }
  }
  if (info) { info[0] = nseg; info[1] = rp.nconst; info[2] = rp.rext; info[3] = (long)lds_bytes_res(pl, rp); info[4] = (long)lds_bytes_res_gb(pl, rp); }
  return 0;
}

// The per-QP HBM slab: every region offset must be set, 16-double aligned, in increasing order and inside the stride, with room for
// its content (a kernel that faults can reset the GPUs of a host, so the layout is checked here, on the CPU).
extern "C" int plan_check_layout(int n, int m, const int *Pp, const int *Pi, const int *Ap, const int *Ai, int force_ordering) {
  Plan pl = build_plan(n, m, Pp, Pi, Ap, Ai, force_ordering, true);
  if (!pl.error.empty()) return 1;
  const WsLayout w = ws_layout(pl);
  const long off[] = {w.ellA, w.ellAt, w.ellP, w.Lf, w.Lb, w.T, w.l, w.u, w.D, w.E, w.dx, w.dy, w.Zg, w.Yg, w.stride};
  const long need[] = {pl.A.entries(), pl.At.entries(), pl.P.entries(), ((long)pl.nblk + 1) * BLK, (long)pl.nblk * BLK, (long)std::max(pl.nT, 1) * BLK,
                       pl.mpad, pl.mpad, pl.npad, pl.mpad, pl.npad, pl.mpad, pl.mpad, pl.mpad};
  if (off[0] != 0) return 2;
  for (int k = 0; k < 14; k++) {
    if (off[k] % 16 != 0) return 3;
    if (off[k + 1] < off[k] + need[k]) return 4 + k;
  }
  return 0;
}

// ---- on-chip mode (kernel_onchip.hpp): lane-accurate emulation of the solve and of the factorisation -- the MFMA operand / result layouts
// (v_mfma_f64_16x16x4_f64: A[m = l & 15][k = l >> 4], B[k = l >> 4][n = l & 15], D[m = (l >> 4) + 4 g][n = l & 15]; the 4-block
// v_mfma_f64_4x4x4_4b_f64 further down), the swizzled LDS block image read element by element, the ds_swizzle hand-over of a chain stage, the
// chain tables, the per-wave position slots with phantoms, the junction term and the hub phases -- on the factor of the level-parallel plan
// or of the in-register LDL', with a hazard check between the waves of every phase.
// returns 0 ok, 1 plan error, 2 not positive definite, 3 hazard, 5 the pattern is not taken by the on-chip plan; info: nbc, has_hub, junc, nlds, nhr, lds bytes
namespace {
struct Wave { double v[64][4]; };   // one d4 per lane
inline int swz(int r, int c) { return ((r ^ ((r >> 2) & 1)) << 4) | (c ^ (((r >> 1) & 3) << 2) ^ (((r >> 3) & 1) << 1)); }
void mfma(const double a[64], const double b[64], Wave &acc) {
  double A[16][4], B[4][16];
  for (int l = 0; l < 64; l++) { A[l & 15][l >> 4] = a[l]; B[l >> 4][l & 15] = b[l]; }
  for (int l = 0; l < 64; l++) for (int g = 0; g < 4; g++) {
    const int m = (l >> 4) + 4 * g, n = l & 15;
    double s = 0; for (int k = 0; k < 4; k++) s += A[m][k] * B[k][n];
    acc.v[l][g] += s;
  }
}
void mv(const Wave &a, const Wave &v, Wave &acc) {     // oc_mv: 4 MFMAs, register i of the block against register i of the vector
  for (int i = 0; i < 4; i++) { double x[64], y[64]; for (int l = 0; l < 64; l++) { x[l] = a.v[l][i]; y[l] = v.v[l][i]; } mfma(x, y, acc); }
}
Wave ldB(const double *vec, int p) { Wave w; for (int l = 0; l < 64; l++) for (int i = 0; i < 4; i++) w.v[l][i] = vec[BS * p + 4 * (l >> 4) + i]; return w; }
bool stB(double *vec, int p, const Wave &w) {          // every lane of a group stores: they must agree
  for (int l = 0; l < 64; l++) for (int i = 0; i < 4; i++) {
    if (w.v[l][i] != w.v[l & ~15][i]) return false;
    vec[BS * p + 4 * (l >> 4) + i] = w.v[l][i];
  }
  return true;
}
Wave zero() { Wave w; std::memset(&w, 0, sizeof(w)); return w; }
Wave add(Wave a, const Wave &b) { for (int l = 0; l < 64; l++) for (int i = 0; i < 4; i++) a.v[l][i] += b.v[l][i]; return a; }
}  // namespace

// v_mfma_f64_4x4x4_4b_f64 (layout probed on hardware, tools/probes/mfma_4x4_probe.hip): A lane (i = l & 3, block = (l >> 2) & 3, k = l >> 4),
// B lane (j, block, k), D lane (j, block, i = l >> 4); one double per lane and operand
namespace {
struct Wave1 { double v[64]; };
Wave1 zero1() { Wave1 w; std::memset(&w, 0, sizeof(w)); return w; }
void mfma4(const double a[64], const double b[64], Wave1 &acc) {
  for (int l = 0; l < 64; l++) {
    const int j = l & 3, blk = (l >> 2) & 3, i = l >> 4;
    double s = 0;
    for (int k = 0; k < 4; k++) s += a[i + 4 * blk + 16 * k] * b[j + 4 * blk + 16 * k];
    acc.v[l] += s;
  }
}
void mv4(const Wave &a, const Wave &v, Wave1 &acc) {      // one product: the four steps K of oc_mv4
  for (int K = 0; K < 4; K++) { double x[64], y[64]; for (int l = 0; l < 64; l++) { x[l] = a.v[l][K]; y[l] = v.v[l][K]; } mfma4(x, y, acc); }
}
Wave ldA4g(const double *blk, double sign) { Wave w; for (int l = 0; l < 64; l++) for (int K = 0; K < 4; K++) w.v[l][K] = sign * blk[(l & 15) * BS + (l >> 4) + 4 * K]; return w; }
Wave ldD4g(const double *blk, double sign) { Wave w; for (int l = 0; l < 64; l++) for (int K = 0; K < 4; K++) w.v[l][K] = sign * blk[((l >> 4) + 4 * K) * BS + (l & 15)]; return w; }
Wave ldF4(const double *blk) { Wave w; for (int l = 0; l < 64; l++) for (int K = 0; K < 4; K++) w.v[l][K] = blk[swz(l & 15, (l >> 4) + 4 * K)]; return w; }
Wave ldT4(const double *blk) { Wave w; for (int l = 0; l < 64; l++) for (int K = 0; K < 4; K++) w.v[l][K] = blk[swz((l >> 4) + 4 * K, l & 15)]; return w; }
Wave ldB4(const double *vec, int p) { Wave w; for (int l = 0; l < 64; l++) for (int K = 0; K < 4; K++) w.v[l][K] = vec[BS * p + (l >> 4) + 4 * K]; return w; }
bool stB4(double *vec, int p, const Wave1 &d) {            // the four lanes of a quad store one value: they must agree
  for (int l = 0; l < 64; l++) {
    if (d.v[l] != d.v[l & ~3]) return false;
    vec[BS * p + 4 * ((l >> 2) & 3) + (l >> 4)] = d.v[l];
  }
  return true;
}
}  // namespace

// kernel_onchip.hpp oc_ldl: the in-register block LDL' of the chain + hub topology, wave by wave and phase by phase, on the slab S of assembled
// blocks; layouts D(X): lane 16 kk + n holds X[kk + 4 g][n], A(X) = D(X'); a hazard = two waves touching a slab block or a scratch block in one
// phase with at least one of them writing.  returns 0 ok, 2 not positive definite, 3 hazard
namespace {
Wave ldD(const double *b) { Wave w; for (int l = 0; l < 64; l++) for (int g = 0; g < 4; g++) w.v[l][g] = b[((l >> 4) + 4 * g) * BS + (l & 15)]; return w; }
void stD(double *b, const Wave &w) { for (int l = 0; l < 64; l++) for (int g = 0; g < 4; g++) b[((l >> 4) + 4 * g) * BS + (l & 15)] = w.v[l][g]; }
Wave ldA(const double *b) { Wave w; for (int l = 0; l < 64; l++) for (int g = 0; g < 4; g++) w.v[l][g] = b[(l & 15) * BS + (l >> 4) + 4 * g]; return w; }
void stA(double *b, const Wave &w) { for (int l = 0; l < 64; l++) for (int g = 0; g < 4; g++) b[(l & 15) * BS + (l >> 4) + 4 * g] = w.v[l][g]; }
Wave neg(Wave a) { for (int l = 0; l < 64; l++) for (int i = 0; i < 4; i++) a.v[l][i] = -a.v[l][i]; return a; }
bool sweep_d(Wave &a) {
  bool ok = true;
  for (int k = 0; k < BS; k++) {
    const int g = k >> 2, ks = k & 3;
    const double d = a.v[16 * ks + k][g];
    ok = ok && d > 0.0;
    const double p = 1.0 / d;
    double w[64], bw[64];
    for (int l = 0; l < 64; l++) {
      const bool grp = (l >> 4) == ks, colk = (l & 15) == k;
      w[l] = grp ? (colk ? -1.0 : a.v[l][g]) : 0.0; bw[l] = -p * w[l];
      if (grp) a.v[l][g] = 0.0;
      if (colk) for (int gg = 0; gg < 4; gg++) a.v[l][gg] = 0.0;
    }
    mfma(w, bw, a);
  }
  a = neg(a);
  return ok;
}
int emu_oc_ldl(const OcPlan &oc, std::vector<double> &S, bool HUB) {
  const int *tab = oc.tab.data();
  const int *pt = tab + oc.o_pos;
  std::vector<double> scr(8 * BLK, 0.0);
  struct St { Wave Dc, Wp, Hr, Lc, Ln, Sn, Hc, Hn, Sh; int pend; bool ok; } st[4];
  for (auto &x : st) { x.Dc = x.Wp = x.Hr = x.Lc = x.Ln = x.Sn = x.Hc = x.Hn = x.Sh = zero(); x.pend = -1; x.ok = true; }
  const int nblk = (int)(S.size() / BLK);
  std::vector<int> wr(nblk + 8, -1), rd(nblk + 8, 0);
  bool hazard = false;
  auto touch = [&](int w, int blk, bool write) {
    if (wr[blk] >= 0 && wr[blk] != w) hazard = true;
    if (write) { if (rd[blk] & ~(1 << w)) hazard = true; wr[blk] = w; } else rd[blk] |= 1 << w;
  };
  auto barrier = [&]() { std::fill(wr.begin(), wr.end(), -1); std::fill(rd.begin(), rd.end(), 0); };
  auto SB = [&](int b) { return &S[(size_t)b * BLK]; };
  auto SC = [&](int b) { return &scr[(size_t)b * BLK]; };
  for (int pi = 0; pi < (int)oc.pairs.size(); pi++) {      // the twisted pairs one after the other (kernel_onchip.hpp oc_ldl); the helpers' hub sums carry across
  const int rec = oc.o_pair + 12 * pi;
  const int LE = tab[rec], LF = tab[rec + 1];
  const bool junc = tab[rec + 4] && LF > 0;
  const int Sph = junc ? std::max(LE + 1, LF) : LE;
  for (auto &x : st) { x.Dc = x.Wp = x.Hr = x.Lc = x.Ln = x.Sn = x.Hc = x.Hn = zero(); x.pend = -1; }
  const int nph = HUB ? Sph + 1 : Sph;
  for (int s = -1; s < nph; s++) {                       // s = -1: the fetches ahead of the first phase
    for (int wid = 0; wid < 4; wid++) {
      const int ch = wid & 1; const bool helper = wid >= 2;
      const int L = ch == 0 ? LE : LF, cb = ch == 0 ? tab[rec + 2] : tab[rec + 3];
      auto step_of = [&](int q) -> int { if (q < 0) return -1; if (ch == 0) return q < LE ? q : -1; if (q < LF - 1) return q; return (q == Sph - 1 && LF > 0) ? LF - 1 : -1; };
      auto gsv = [&](int k) { return pt[5 * tab[cb + 2 * k]]; };
      auto csv = [&](int k) { return pt[5 * tab[cb + 2 * k] + 1]; };
      auto hsv = [&](int k) { return pt[5 * tab[cb + 2 * k] + 2]; };
      St &x = st[wid];
      auto chain_fetch = [&](int k) {
        const bool last = k == L - 1, tojunc = ch == 0 && last && junc;
        if (!last || tojunc) { touch(wid, csv(k), false); x.Ln = ldA(SB(csv(k))); }
        if (!last) { touch(wid, gsv(k + 1), false); x.Sn = ldD(SB(gsv(k + 1))); }
      };
      if (s < 0) {
        if (L > 0) {
          if (!helper) { touch(wid, gsv(0), false); x.Dc = ldD(SB(gsv(0))); chain_fetch(0); }
          else if (HUB) { touch(wid, hsv(0), false); x.Hr = ldA(SB(hsv(0))); }
        }
        continue;
      }
      if (!helper) {
        if (x.pend >= 0) { touch(wid, x.pend, true); stA(SB(x.pend), x.Wp); x.pend = -1; }
        const int k = step_of(s);
        if (k < 0) continue;
        const int gs = gsv(k), cs = csv(k);
        const bool last = k == L - 1, tojunc = ch == 0 && last && junc, has_next = !last || tojunc;
        const Wave Ls = x.Ln, Sd = tojunc ? zero() : x.Sn;
        if (!last) chain_fetch(k + 1);
        if (ch == 1 && last && junc) { touch(wid, nblk + 4, false); x.Dc = add(x.Dc, ldD(SC(4))); }
        x.ok = sweep_d(x.Dc) && x.ok;
        touch(wid, gs, true); stD(SB(gs), x.Dc);
        if (HUB) { touch(wid, nblk + 2 * ch + (s & 1), true); stD(SC(2 * ch + (s & 1)), x.Dc); }
        if (has_next) {
          Wave Wt = zero(); mv(x.Dc, Ls, Wt);
          x.Wp = Wt; x.pend = cs;
          Wave acc = Sd; mv(neg(Wt), Ls, acc);
          if (tojunc) { touch(wid, nblk + 4, true); stD(SC(4), acc); } else x.Dc = acc;
        }
      } else if (HUB) {
        const int kc = step_of(s), kh = step_of(s - 1);
        if (kc >= 0) {
          const bool last = kc == L - 1, tojunc = ch == 0 && last && junc;
          if (!last || tojunc) { touch(wid, csv(kc), false); x.Ln = ldA(SB(csv(kc))); }
          if (!last) { touch(wid, hsv(kc + 1), false); x.Hn = ldA(SB(hsv(kc + 1))); }
        }
        if (kh >= 0) {
          const int hs = hsv(kh);
          const bool last = kh == L - 1, tojunc = ch == 0 && last && junc, has_next = !last || tojunc;
          if (ch == 1 && last && junc) { touch(wid, nblk + 5, false); x.Hr = add(x.Hr, ldD(SC(5))); }
          touch(wid, nblk + 2 * ch + ((s - 1) & 1), false);
          const Wave G = ldD(SC(2 * ch + ((s - 1) & 1)));
          Wave WhT = zero(); mv(G, x.Hr, WhT);
          touch(wid, hs, true); stA(SB(hs), WhT);
          const Wave nW = neg(WhT);
          mv(nW, x.Hr, x.Sh);
          if (has_next) { Wave acc = tojunc ? zero() : x.Hc; mv(x.Lc, nW, acc); if (tojunc) { touch(wid, nblk + 5, true); stD(SC(5), acc); } else x.Hr = acc; }
        }
        x.Lc = x.Ln; x.Hc = x.Hn;
      }
    }
    if (s >= 0) barrier();                               // (no barrier between the fetches ahead and phase 0)
  }
  for (int wid = 0; wid < 2; wid++) if (st[wid].pend >= 0) { touch(wid, st[wid].pend, true); stA(SB(st[wid].pend), st[wid].Wp); }
  barrier();
  }
  bool ok = st[0].ok && st[1].ok;
  if (HUB) {
    stD(SC(6), st[2].Sh); stD(SC(7), st[3].Sh);
    barrier();
    Wave Dh = add(add(ldD(SB(oc.ghub_src)), ldD(SC(6))), ldD(SC(7)));
    ok = sweep_d(Dh) && ok;
    stD(SB(oc.ghub_src), Dh);
  }
  if (hazard) return 3;
  return ok ? 0 : 2;
}
}  // namespace

// nw = 4: the two-workgroups-per-CU instances (twisted order, chains of at most 17 positions); nw = 8: the long-chain instances (padded
// twist -- ordering 3 --, chains of any length, zyg: z and y in the slab for the LDS figure)
// (ordering 4: the dissected order -- separators of the chain in the hub block, several twisted pairs; info[2] then reports the number of pairs)
static int plan_execute_oc_impl(int n, int m, const int *Pp, const int *Pi, const int *Ap, const int *Ai, int nw, int NG, int NH, int ldl, int zyg, int ordering,
                                const double *Pval, const double *Aval, const double *rho, double sigma, const double *rhs, double *sol, long *info);
extern "C" int plan_execute_oc_nw(int n, int m, const int *Pp, const int *Pi, const int *Ap, const int *Ai, int nw, int NG, int NH, int ldl, int zyg,
                                  const double *Pval, const double *Aval, const double *rho, double sigma, const double *rhs, double *sol, long *info) {
  return plan_execute_oc_impl(n, m, Pp, Pi, Ap, Ai, nw, NG, NH, ldl, zyg, nw == 8 ? 3 : 2, Pval, Aval, rho, sigma, rhs, sol, info);
}
extern "C" int plan_execute_oc_dissected(int n, int m, const int *Pp, const int *Pi, const int *Ap, const int *Ai, int NG, int NH, int ldl,
                                         const double *Pval, const double *Aval, const double *rho, double sigma, const double *rhs, double *sol, long *info) {
  return plan_execute_oc_impl(n, m, Pp, Pi, Ap, Ai, 8, NG, NH, ldl, 0, 4, Pval, Aval, rho, sigma, rhs, sol, info);
}
extern "C" int plan_execute_oc_dissected4(int n, int m, const int *Pp, const int *Pi, const int *Ap, const int *Ai, int NG, int NH, int ldl,
                                          const double *Pval, const double *Aval, const double *rho, double sigma, const double *rhs, double *sol, long *info) {
  return plan_execute_oc_impl(n, m, Pp, Pi, Ap, Ai, 4, NG, NH, ldl, 0, 4, Pval, Aval, rho, sigma, rhs, sol, info);
}
static int plan_execute_oc_impl(int n, int m, const int *Pp, const int *Pi, const int *Ap, const int *Ai, int nw, int NG, int NH, int ldl, int zyg, int ordering,
                                const double *Pval, const double *Aval, const double *rho, double sigma, const double *rhs, double *sol, long *info) {
  if (nw != 4 && nw != 8) return 1;
  Plan pl = build_plan(n, m, Pp, Pi, Ap, Ai, ordering, nw == 8 ? 2 : 1, nw == 8 ? 3 : 1);      // (dissected order: three separators for eight waves, one for four)
  if (!pl.error.empty()) return ordering == 4 ? 5 : 1;
  ResPlan rp = build_res_plan(pl, nw, false);
  OcPlan oc = build_oc_plan(pl, nw, 1 << 20, NG, NH, nw == 8 ? 64 : OC_CHAIN_SHORT);
  if (info) { info[0] = oc.nbc; info[1] = oc.has_hub; info[2] = oc.junc; info[3] = oc.nlds; info[4] = oc.nhr; info[5] = oc.ok ? lds_bytes_oc(pl, rp, oc, zyg != 0) : 0;
              info[6] = (long)oc.chainE.size(); info[7] = (long)oc.chainF.size(); if (ordering == 4) info[2] = (long)oc.pairs.size(); }
  if (!oc.ok) return 5;
  // ---- factor (as plan_execute_res: assembly + level-parallel LDL')
  std::vector<double> vA, vAt, vP;
  ell_fill(pl.A, Aval, vA); ell_fill(pl.At, Aval, vAt); ell_fill(pl.P, Pval, vP);
  std::vector<double> dvec(pl.npad, 1.0);
  for (int c = 0; c < pl.At.nchunks; c++) for (int lane = 0; lane < WAVE; lane++) {
    int t = c * WAVE + lane; if (t >= pl.npad) continue;
    double a = 0;
    for (int s = pl.At.chunk_off[c]; s < pl.At.chunk_off[c + 1]; s++) { long p = (long)s * WAVE + lane; if (pl.At.flag[p]) a += rho[pl.At.idx[p]] * vAt[p] * vAt[p]; }
    dvec[t] = pl.perm[t] >= 0 ? sigma + a : 1.0;
  }
  std::vector<double> T((size_t)std::max(pl.nT, 1) * BLK, 0.0);
  for (int c = 0; c < pl.A.nchunks; c++) for (int lane = 0; lane < WAVE; lane++) {
    int i = c * WAVE + lane; if (i >= m) continue;
    for (int s = pl.A.chunk_off[c]; s < pl.A.chunk_off[c + 1]; s++) { long p = (long)s * WAVE + lane; if (pl.tpos[p] >= 0) T[pl.tpos[p]] = vA[p] * std::sqrt(rho[i]); }
  }
  std::vector<double> S((size_t)pl.nblk * BLK, 0.0), tmp((size_t)std::max(rp.ntemp, 1) * BLK, 0.0);
  for (int b = 0; b < pl.nblk; b++) {
    double *C = &S[(size_t)b * BLK];
    for (int g = pl.asm_ptr[b]; g < pl.asm_ptr[b + 1]; g++) gemm_abt(&T[(size_t)pl.asm_a[g] * BLK], &T[(size_t)pl.asm_b[g] * BLK], C, 1.0);
    for (int g = 0; g < 4; g++) for (int lane = 0; lane < WAVE; lane++) {
      int row = (lane >> 4) + 4 * g, col = lane & 15;
      int pi = pl.asm_pidx[(size_t)b * BLK + g * WAVE + lane];
      if (pi >= 0) C[row * BS + col] += vP[pi];
      if (pl.blk_diag[b] >= 0 && row == col) C[row * BS + col] += dvec[pl.blk_diag[b] * BS + row];
    }
  }
  if (ldl) {
    if ((NH > 0) != (oc.has_hub != 0)) return 5;
    const int rcl = emu_oc_ldl(oc, S, NH > 0);
    if (rcl) return rcl;
  } else {
  std::vector<int> pend_slot, pend_tmp;
  for (int lev = 0; lev < rp.nlev; lev++) {
    for (int ci = rp.lv_ptr[lev]; ci < rp.lv_ptr[lev + 1]; ci++) if (!sweep_inverse(&S[(size_t)rp.lv_diag[ci] * BLK])) return 2;
    for (size_t a = 0; a < pend_slot.size(); a++) std::memcpy(&S[(size_t)pend_slot[a] * BLK], &tmp[(size_t)pend_tmp[a] * BLK], BLK * sizeof(double));
    pend_slot.clear(); pend_tmp.clear();
    const int w0 = rp.lw_ptr[lev], nwk = rp.lw_ptr[lev + 1] - w0;
    std::vector<double> tnew((size_t)std::max(nwk, 1) * BLK, 0.0);
    for (int a = 0; a < nwk; a++) gemm_abt(&S[(size_t)rp.lw_slot[w0 + a] * BLK], &S[(size_t)rp.lw_g[w0 + a] * BLK], &tnew[(size_t)a * BLK], 1.0);
    for (int a = 0; a < nwk; a++) std::memcpy(&tmp[(size_t)a * BLK], &tnew[(size_t)a * BLK], BLK * sizeof(double));
    for (int w = 0; w < nw; w++) for (int u = rp.lu_ptr[lev * nw + w]; u < rp.lu_ptr[lev * nw + w + 1]; u++)
      gemm_abt(&tmp[(size_t)rp.lu_tmp[u] * BLK], &S[(size_t)rp.lu_b[u] * BLK], &S[(size_t)rp.lu_dst[u] * BLK], -1.0);
    for (int a = 0; a < nwk; a++) { pend_slot.push_back(rp.lw_slot[w0 + a]); pend_tmp.push_back(a); }
  }
  for (size_t a = 0; a < pend_slot.size(); a++) std::memcpy(&S[(size_t)pend_slot[a] * BLK], &tmp[(size_t)pend_tmp[a] * BLK], BLK * sizeof(double));
  }
  // ---- oc_load_factor: LDS images (negated off-diagonal blocks, swizzled) and per-wave register blocks
  const int *tab = oc.tab.data();
  std::vector<double> BL((size_t)oc.nlds * BLK, 0.0);
  for (int j = 0; j < oc.nlds; j++) {
    const int src = tab[oc.o_fill + 3 * j], slot = tab[oc.o_fill + 3 * j + 1], neg = tab[oc.o_fill + 3 * j + 2];
    if (slot < 0 || slot >= oc.nlds || src < 0 || src >= pl.nblk) return 1;
    for (int r = 0; r < BS; r++) for (int c = 0; c < BS; c++) BL[(size_t)slot * BLK + swz(r, c)] = (neg ? -1.0 : 1.0) * S[(size_t)src * BLK + r * BS + c];
  }
  std::vector<std::vector<Wave>> G(nw, std::vector<Wave>(NG, zero())), HF(nw, std::vector<Wave>(std::max(NH, 1), zero())), HT = HF;
  std::vector<std::vector<int>> vpos(nw, std::vector<int>(NG)), hslot = vpos, okp = vpos;
  const int zero_blk = pl.npad / BS + nw + 1;
  for (int w = 0; w < nw; w++) for (int s = 0; s < NG; s++) {
    const int p = w + nw * s, pe = p < oc.nbc ? p : oc.nbc - 1;
    okp[w][s] = p < oc.nbc; vpos[w][s] = p < oc.nbc ? p : zero_blk;
    const int hs = tab[oc.o_pos + 5 * pe + 4]; hslot[w][s] = hs >= 0 ? hs : 0;
    if (p < oc.nbc) {
      G[w][s] = ldA4g(&S[(size_t)tab[oc.o_pos + 5 * p] * BLK], 1.0);       // (operands of the 4-block MFMA of the hub / diagonal phases)
      const int hsrc = tab[oc.o_pos + 5 * p + 2];
      if (s < NH && hsrc >= 0) { HF[w][s] = ldA4g(&S[(size_t)hsrc * BLK], -1.0); HT[w][s] = ldD4g(&S[(size_t)hsrc * BLK], -1.0); }
    }
  }
  // ---- oc_solve
  const int np = (int)oc.pairs.size();
  if (2 * np > nw) return 1;
  std::vector<double> R((size_t)pl.npad + oc_rext(nw, np), 0.0);
  for (int j = 0; j < n; j++) R[pl.pos[j]] = rhs[j];
  double *EXT = &R[pl.npad];
  const int H = oc.nbc;
  // chain waves 2 i (E) and 2 i + 1 (F) walk pair i; its junction term waits in vector block 0 (pair 0) / nw + 1 + i behind the solve vector
  auto P_LE = [&](int i) { return tab[oc.o_pair + 12 * i + 5]; };
  auto P_LF = [&](int i) { return tab[oc.o_pair + 12 * i + 6]; };
  auto P_cb = [&](int i, int role) { return tab[oc.o_pair + 12 * i + 7 + role]; };
  auto P_junc = [&](int i) { return tab[oc.o_pair + 12 * i + 4]; };
  auto P_f = [&](int i) { return (P_junc(i) && P_LF(i) > 0) ? tab[P_cb(i, 1) + 2 * (P_LF(i) - 1)] : -1; };
  auto P_xs = [&](int i) { return i == 0 ? 0 : nw + 1 + i; };
  const bool HUB = NH > 0;
  if (HUB != (oc.has_hub != 0)) return 5;
  std::vector<int> wr(pl.nb + nw + 2 + np, -1), rd(pl.nb + nw + 2 + np, 0);      // hazard tracking per vector block within a phase
  bool hazard = false;
  auto touch = [&](int w, int blk, bool write) {
    if (wr[blk] >= 0 && wr[blk] != w) hazard = true;
    if (write) { if (rd[blk] & ~(1 << w)) hazard = true; wr[blk] = w; } else rd[blk] |= 1 << w;
  };
  auto barrier = [&]() { std::fill(wr.begin(), wr.end(), -1); std::fill(rd.begin(), rd.end(), 0); };
  const int ext0 = pl.npad / BS;
  // F1 (4-block MFMA: the result element of a stage goes to LDS and, through the quad broadcast of ds_swizzle, into the next stage's operand)
  auto bc4 = [](const Wave1 &r) { Wave x; for (int l = 0; l < 64; l++) for (int K = 0; K < 4; K++) x.v[l][K] = r.v[(l & 0x33) | (K << 2)]; return x; };
  auto ldE4 = [](const double *vec, int p) { Wave1 r; for (int l = 0; l < 64; l++) r.v[l] = vec[BS * p + 4 * ((l >> 2) & 3) + (l >> 4)]; return r; };
  bool quads1 = true;
  for (int w = 0; w < 2 * np; w++) {
    const int pi = w >> 1, role = w & 1;
    const int len = role == 0 ? P_LE(pi) : P_LF(pi), cb = P_cb(pi, role);
    if (len == 0) continue;
    int e = 0; touch(w, tab[cb], false);
    Wave x = ldB4(R.data(), tab[cb]);
    for (int k = 0; k + 1 < len; k++) {
      const int slot = tab[cb + 2 * k + 1], pn = tab[cb + 2 * (k + 1)];
      touch(w, pn, false);
      Wave1 c = ldE4(R.data(), pn);
      mv4(ldF4(&BL[(size_t)slot * BLK]), x, c);
      touch(w, pn, true);
      quads1 = stB4(R.data(), pn, c) && quads1;
      x = bc4(c); e = k + 1;
    }
    if (role == 0 && P_junc(pi)) { Wave1 c = zero1(); mv4(ldF4(&BL[(size_t)tab[cb + 2 * e + 1] * BLK]), x, c); touch(w, ext0 + P_xs(pi), true); quads1 = stB4(EXT, P_xs(pi), c) && quads1; }
  }
  if (!quads1) return 1;
  barrier();
  for (int pi = 0; pi < np; pi++) { const int f = P_f(pi); if (f >= 0) { const int w = f & (nw - 1); touch(w, f, false); touch(w, ext0 + P_xs(pi), false); Wave t = add(ldB(R.data(), f), ldB(EXT, P_xs(pi))); touch(w, f, true); stB(R.data(), f, t); } }
  std::vector<Wave> xh(nw, zero());
  std::vector<Wave1> xhd(nw, zero1());
  bool quads = true;
  if (HUB) {
    for (int w = 0; w < nw; w++) {
      Wave1 hsum = zero1();
      for (int s = 0; s < NG; s++) {                          // (one accumulator per position on the device, summed in slot order: the same value up to rounding)
        touch(w, vpos[w][s], false);
        const Wave t = ldB4(R.data(), vpos[w][s]);
        Wave1 acc = zero1();
        if (s < NH) mv4(HF[w][s], t, acc); else mv4(ldF4(&BL[(size_t)hslot[w][s] * BLK]), t, acc);
        for (int l = 0; l < 64; l++) hsum.v[l] += acc.v[l];
      }
      touch(w, ext0 + 1 + w, true); quads = stB4(EXT, 1 + w, hsum) && quads;
    }
    barrier();
    for (int w = 0; w < nw; w++) {
      touch(w, H, false);
      Wave th = ldB4(R.data(), H);
      for (int v = 0; v < nw; v++) { touch(w, ext0 + 1 + v, false); th = add(th, ldB4(EXT, 1 + v)); }
      mv4(ldF4(&BL[(size_t)oc.ghub_slot * BLK]), th, xhd[w]);
    }
    // every wave stores the same x_hub to the junction block and reads it back as operand pieces (same data: not a hazard)
    for (int w = 1; w < nw; w++) for (int l = 0; l < 64; l++) if (xhd[w].v[l] != xhd[0].v[l]) return 1;
    quads = stB4(EXT, 0, xhd[0]) && quads;
    for (int w = 0; w < nw; w++) xh[w] = ldB4(EXT, 0);
  }
  for (int w = 0; w < nw; w++) for (int s = 0; s < NG; s++) {
    touch(w, vpos[w][s], false);
    const Wave t = ldB4(R.data(), vpos[w][s]);
    Wave1 d = zero1(); mv4(G[w][s], t, d);
    if (HUB) { if (s < NH) mv4(HT[w][s], xh[w], d); else mv4(ldT4(&BL[(size_t)hslot[w][s] * BLK]), xh[w], d); }
    if (okp[w][s]) { touch(w, vpos[w][s], true); quads = stB4(R.data(), vpos[w][s], d) && quads; }
  }
  barrier();
  if (HUB) { touch(nw - 1, H, true); quads = stB4(R.data(), H, xhd[nw - 1]) && quads; }
  if (!quads) return 1;
  for (int w = 0; w < 2 * np; w++) {
    const int pi = w >> 1, role = w & 1, f = P_f(pi);
    const int len = role == 0 ? P_LE(pi) : P_LF(pi), cb = P_cb(pi, role);
    if (len == 0) continue;
    int k = len - 2; Wave x;
    if (role == 0 && P_junc(pi)) { touch(w, f, false); x = ldB4(R.data(), f); k = len - 1; } else { touch(w, tab[cb + 2 * (len - 1)], false); x = ldB4(R.data(), tab[cb + 2 * (len - 1)]); }
    for (; k >= 0; k--) {
      const int p = tab[cb + 2 * k], slot = tab[cb + 2 * k + 1];
      touch(w, p, false);
      Wave1 c = ldE4(R.data(), p);
      mv4(ldT4(&BL[(size_t)slot * BLK]), x, c);
      touch(w, p, true); if (!stB4(R.data(), p, c)) return 1;
      x = bc4(c);
    }
  }
  if (hazard) return 3;
  for (int j = 0; j < n; j++) sol[j] = R[pl.pos[j]];
  return 0;
}

extern "C" int plan_execute_oc(int n, int m, const int *Pp, const int *Pi, const int *Ap, const int *Ai, int NG, int NH, int ldl,
                               const double *Pval, const double *Aval, const double *rho, double sigma, const double *rhs, double *sol, long *info) {
  return plan_execute_oc_nw(n, m, Pp, Pi, Ap, Ai, 4, NG, NH, ldl, 0, Pval, Aval, rho, sigma, rhs, sol, info);
}

// Dense tiles of A for the iteration's two sweeps (plan.hpp build_tile_plan): A x and A' w from the tiles -- through the 4-block MFMA in the
// lane layouts the kernel uses (tile storage [lane][K] read as it is for A x, as its transpose for A' w; results in the lanes o4) -- plus the
// remainder ELL layouts, against the plain CSC products.  ordering / pad as the on-chip plans (2: four-wave, 3: eight-wave).
// returns 0 ok, 1 plan error, 4 no tiles found, 6 table inconsistency, 7 products differ; out = [ntile, entries in tiles, Ar slots, Atr slots, A slots, At slots, max tiles per chunk, max per block]
extern "C" int plan_tile_check(int n, int m, const int *Pp, const int *Pi, const int *Ap, const int *Ai, int ordering, const double *Aval, const double *x, const double *w,
                               double *err, long *out) {
  Plan pl = build_plan(n, m, Pp, Pi, Ap, Ai, ordering, 2);
  if (!pl.error.empty()) return 1;
  TilePlan tp = build_tile_plan(pl, n, m, Ap, Ai, 2);
  if (out) { out[0] = tp.ntile; out[1] = tp.entries_in_tiles; out[2] = tp.on ? tp.Ar.slots() : 0; out[3] = tp.on ? tp.Atr.slots() : 0; out[4] = pl.A.slots(); out[5] = pl.At.slots();
             out[6] = tp.max_per_chunk; out[7] = tp.max_per_block; }
  if (!tp.on) return 4;
  // every entry of A exactly once: in a tile or in each remainder layout
  std::vector<int> seenA(pl.nnzA_in, 0), seenT(pl.nnzA_in, 0);
  for (size_t e = 0; e < tp.tsrc.size(); e++) if (tp.tsrc[e] >= 0) { seenA[tp.tsrc[e]]++; seenT[tp.tsrc[e]]++; }
  for (size_t e = 0; e < tp.Ar.src.size(); e++) if (tp.Ar.src[e] >= 0) seenA[tp.Ar.src[e]]++;
  for (size_t e = 0; e < tp.Atr.src.size(); e++) if (tp.Atr.src[e] >= 0) seenT[tp.Atr.src[e]]++;
  for (int k = 0; k < pl.nnzA_in; k++) if (seenA[k] != 1 || seenT[k] != 1) return 6;
  for (int t = 0; t < tp.ntile; t++) for (int r = 0; r < BS; r++) {
    const int i = tp.rowid[(size_t)t * BS + r];
    if (i >= m) return 6;
    if (i >= 0) { bool listed = false; for (int q = tp.ta_ptr[i / WAVE]; q < tp.ta_ptr[i / WAVE + 1]; q++) listed |= tp.ta_tid[q] == t; if (!listed) return 6; }
  }
  std::vector<double> xp(pl.npad, 0.0), ax(m, 0.0), atw(pl.npad, 0.0);
  for (int j = 0; j < n; j++) xp[pl.pos[j]] = x[j];
  auto tval = [&](int t, int e) { const int sidx = tp.tsrc[(size_t)t * BLK + e]; return sidx >= 0 ? Aval[sidx] : 0.0; };
  // A x
  for (int t = 0; t < tp.ntile; t++) {
    Wave a, v = ldB4(xp.data(), tp.tJ[t]);
    for (int l = 0; l < 64; l++) for (int K = 0; K < 4; K++) a.v[l][K] = tval(t, l * 4 + K);
    Wave1 acc = zero1(); mv4(a, v, acc);
    for (int l = 0; l < 64; l++) if ((l & 3) == 0) { const int r = 4 * ((l >> 2) & 3) + (l >> 4), i = tp.rowid[(size_t)t * BS + r]; if (i >= 0) ax[i] += acc.v[l]; }
  }
  for (int c = 0; c < tp.Ar.nchunks; c++) for (int lane = 0; lane < WAVE; lane++) {
    const int i = c * WAVE + lane; if (i >= m) continue;
    for (int sl = tp.Ar.chunk_off[c]; sl < tp.Ar.chunk_off[c + 1]; sl++) { const long e = (long)sl * WAVE + lane; if (tp.Ar.src[e] >= 0) ax[i] += Aval[tp.Ar.src[e]] * xp[tp.Ar.idx[e]]; }
  }
  // A' w
  for (int J = 0; J < pl.nb; J++) for (int q = tp.tt_ptr[J]; q < tp.tt_ptr[J + 1]; q++) {
    const int t = tp.tt_tid[q];
    if (tp.tJ[t] != J) return 6;
    Wave a, v;
    for (int l = 0; l < 64; l++) for (int K = 0; K < 4; K++) {
      const int r = (l >> 4) + 4 * K, c = l & 15;
      a.v[l][K] = tval(t, (r + BS * (c & 3)) * 4 + (c >> 2));
      const int i = tp.rowid[(size_t)t * BS + r];
      v.v[l][K] = i >= 0 ? w[i] : 0.0;
    }
    Wave1 acc = zero1(); mv4(a, v, acc);
    for (int l = 0; l < 64; l++) if ((l & 3) == 0) atw[BS * J + 4 * ((l >> 2) & 3) + (l >> 4)] += acc.v[l];
  }
  for (int c = 0; c < tp.Atr.nchunks; c++) for (int lane = 0; lane < WAVE; lane++) {
    const int t = c * WAVE + lane; if (t >= pl.npad) continue;
    for (int sl = tp.Atr.chunk_off[c]; sl < tp.Atr.chunk_off[c + 1]; sl++) { const long e = (long)sl * WAVE + lane; if (tp.Atr.src[e] >= 0) atw[t] += Aval[tp.Atr.src[e]] * w[tp.Atr.idx[e]]; }
  }
  // reference products
  double worst = 0.0, scale = 1.0;
  std::vector<double> rax(m, 0.0), ratw(pl.npad, 0.0);
  for (int j = 0; j < n; j++) for (int k = Ap[j]; k < Ap[j + 1]; k++) { rax[Ai[k]] += Aval[k] * x[j]; ratw[pl.pos[j]] += Aval[k] * w[Ai[k]]; }
  for (int i = 0; i < m; i++) { worst = std::max(worst, std::fabs(ax[i] - rax[i])); scale = std::max(scale, std::fabs(rax[i])); }
  for (int t = 0; t < pl.npad; t++) { worst = std::max(worst, std::fabs(atw[t] - ratw[t])); scale = std::max(scale, std::fabs(ratw[t])); }
  if (err) *err = worst / scale;
  return worst <= 1e-12 * scale ? 0 : 7;
}

// chunk widths of the three ELL structures (diagnostic): out = [nA, widths..., nAt, widths..., nP, widths...]
extern "C" int plan_ell_widths(int n, int m, const int *Pp, const int *Pi, const int *Ap, const int *Ai, int pad4, int *out, int cap) {
  Plan pl = build_plan(n, m, Pp, Pi, Ap, Ai, 2, pad4 != 0);
  if (!pl.error.empty()) return 1;
  int k = 0;
  for (const Ell *e : {&pl.A, &pl.At, &pl.P}) {
    if (k + 1 + e->nchunks > cap) return 2;
    out[k++] = e->nchunks;
    for (int c = 0; c < e->nchunks; c++) out[k++] = e->chunk_off[c + 1] - e->chunk_off[c];
  }
  return 0;
}

// assembly recipe statistics (diagnostic): out = [nT, nblk, total gemm terms, max terms per block, blocks with P entries]
extern "C" int plan_asm_stats(int n, int m, const int *Pp, const int *Pi, const int *Ap, const int *Ai, int *out) {
  Plan pl = build_plan(n, m, Pp, Pi, Ap, Ai, 2, true);
  if (!pl.error.empty()) return 1;
  int mx = 0, withp = 0;
  for (int b = 0; b < pl.nblk; b++) {
    mx = std::max(mx, pl.asm_ptr[b + 1] - pl.asm_ptr[b]);
    bool hp = false; for (int e = 0; e < BLK; e++) if (pl.asm_pidx[(size_t)b * BLK + e] >= 0) hp = true;
    withp += hp;
  }
  out[0] = pl.nT; out[1] = pl.nblk; out[2] = pl.asm_ptr[pl.nblk]; out[3] = mx; out[4] = withp;
  for (int b = 0; b < pl.nblk && b < 100; b++) out[5 + b] = pl.asm_ptr[b + 1] - pl.asm_ptr[b];
  return 0;
}

// chain tables of the on-chip plan (diagnostic): out = [LE, LF, junc, chainE positions..., chainF positions...]
extern "C" int plan_oc_chains(int n, int m, const int *Pp, const int *Pi, const int *Ap, const int *Ai, int *out) {
  Plan pl = build_plan(n, m, Pp, Pi, Ap, Ai, 2, true);
  if (!pl.error.empty()) return 1;
  OcPlan oc = build_oc_plan(pl, 4, 1 << 20, 5, 3);
  if (!oc.ok) return 5;
  int k = 0; out[k++] = (int)oc.chainE.size(); out[k++] = (int)oc.chainF.size(); out[k++] = oc.junc;
  for (int p : oc.chainE) out[k++] = p;
  for (int p : oc.chainF) out[k++] = p;
  return 0;
}

// the on-chip plan's extras (diagnostic / tests): out = [at_poll, at_free, A widths ok under pad = 2 (max |width - exact| <= 3), records consistent]
extern "C" int plan_oc_extras(int n, int m, const int *Pp, const int *Pi, const int *Ap, const int *Ai, int *out) {
  Plan pl = build_plan(n, m, Pp, Pi, Ap, Ai, 2, 2), px = build_plan(n, m, Pp, Pi, Ap, Ai, 2, 0);
  if (!pl.error.empty() || !px.error.empty()) return 1;
  OcPlan oc = build_oc_plan(pl, 4, 1 << 20, 5, 3);
  if (!oc.ok) { oc = build_oc_plan(pl, 4, 1 << 20, 5, 0); if (!oc.ok) return 5; }
  oc_late_chunks(pl, oc, 4, 3, &out[0], &out[1]);
  int okw = 1;
  for (const auto &pr : {std::make_pair(&pl.A, &px.A), std::make_pair(&pl.At, &px.At)})
    for (int c = 0; c < pr.first->nchunks; c++) {
      const int w = pr.first->chunk_off[c + 1] - pr.first->chunk_off[c], e = pr.second->chunk_off[c + 1] - pr.second->chunk_off[c];
      if (w < e || w > e + 3 || ell_batches8(w) > ell_batches8(e) || (w > e && ell_batches8(w) >= ell_batches8(e))) okw = 0;
    }
  out[2] = okw;
  const std::vector<int> rec = oc_asm_records(pl);
  int okr = 1;
  for (int b = 0; b < pl.nblk; b++) {
    const int p0 = pl.asm_ptr[b], cnt = pl.asm_ptr[b + 1] - p0;
    if (rec[8 * b] != cnt || rec[8 * b + 1] != pl.blk_diag[b]) okr = 0;
    for (int t = 0; t < 3; t++) {
      const int a = rec[8 * b + 2 + 2 * t], bb = rec[8 * b + 3 + 2 * t];
      if (t < cnt) { if (a != pl.asm_a[p0 + t] || bb != pl.asm_b[p0 + t]) okr = 0; }
      else if (a != std::max(pl.nT, 1) || bb != std::max(pl.nT, 1)) okr = 0;        // the zero tile behind the T tiles
    }
  }
  out[3] = okr; out[4] = (int)(ws_layout(pl).T + ((long)std::max(pl.nT, 1) + 1) * BLK <= ws_layout(pl).l);
  return 0;
}
