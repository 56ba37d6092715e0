// plan.hpp -- host-side analysis of one QP sparsity pattern into the static "plan" the HIP kernels execute.
//
// Plays the role of the pattern-dependent half of osqp_setup that the reference reaches through
// CuCaQP::initSolver (reference src/sqp_solver/CuCaQP.cpp:183-197): ordering + symbolic factorisation.
// The reference redoes it for every QP; here it is done once per pattern and shared by the whole batch.
//
// What is planned (all index arrays are shared by every QP of the batch):
//   * a variable ordering that pushes "hub" variables (the reference's parameter block p, coupled to every
//     stage -- reference SQPOptimizationSolver.cpp:50) behind the stage-banded variables;
//   * ELL (64-row chunk) layouts of A (by row), A^T (by variable) and symmetric P (by variable);
//   * the 16x16 block pattern of the reduced KKT matrix  M = P + sigma I + A' diag(rho) A  with block fill,
//     its forward/backward block streams, and the left-looking block Cholesky op list;
//   * the assembly recipe of M from 16-row groups of A's general (non-singleton) rows.
// Pure C++ (no HIP) so the host logic is testable without a GPU.
#pragma once
#include <algorithm>
#include <cstdint>
#include <map>
#include <set>
#include <string>
#include <utility>
#include <vector>

namespace mpcqp {

constexpr int BS = 16;          // block edge
constexpr int BLK = BS * BS;    // doubles per block
constexpr int WAVE = 64;

struct Ell {
  int nrows = 0, nchunks = 0;
  std::vector<int> chunk_off;  // [nchunks+1], in units of 64-entry slots
  std::vector<int> idx;        // [slots*64] gather index (0 for padding)
  std::vector<int> src;        // [slots*64] index into the caller's value array (-1 padding)
  std::vector<int> flag;       // [slots*64] per-entry flag (A^T: 1 if the row is a singleton row)
  int slots() const { return chunk_off.empty() ? 0 : chunk_off.back(); }
  long entries() const { return (long)slots() * WAVE; }
};

struct EllEntry { int idx, src, flag; };

inline Ell build_ell(const std::vector<std::vector<EllEntry>> &rows) {
  Ell e;
  e.nrows = (int)rows.size();
  e.nchunks = (e.nrows + WAVE - 1) / WAVE;
  e.chunk_off.assign(e.nchunks + 1, 0);
  for (int c = 0; c < e.nchunks; c++) {
    size_t w = 0;
    for (int r = c * WAVE; r < std::min(e.nrows, (c + 1) * WAVE); r++) w = std::max(w, rows[r].size());
    e.chunk_off[c + 1] = e.chunk_off[c] + (int)w;
  }
  e.idx.assign(e.entries(), 0); e.src.assign(e.entries(), -1); e.flag.assign(e.entries(), 0);
  for (int r = 0; r < e.nrows; r++) {
    int c = r / WAVE, lane = r % WAVE;
    for (size_t s = 0; s < rows[r].size(); s++) {
      long p = ((long)e.chunk_off[c] + (long)s) * WAVE + lane;
      e.idx[p] = rows[r][s].idx; e.src[p] = rows[r][s].src; e.flag[p] = rows[r][s].flag;
    }
  }
  return e;
}

enum { OP_DIAG = 0, OP_OFF = 1 };
inline int pack_op(int kind, int src, int dst) { return kind | (src << 1) | (dst << 16); }

enum { FAC_SUB = 0, FAC_POTRF = 1, FAC_TRSM = 2 };
struct FacOp { int type, dst, a, b; };   // block indices in forward-stream order

struct Plan {
  int n = 0, m = 0, npad = 0, mpad = 0, nb = 0;
  int nnzP_in = 0, nnzA_in = 0, nnzP_triu = 0;
  int ordering = 0;
  std::vector<int> pos;        // [n]   variable -> permuted position
  std::vector<int> perm;       // [npad] position -> variable or -1
  std::vector<int> singleton;  // [m]
  Ell A, At, P;
  // blocks
  int nblk = 0;
  std::vector<int> blkI, blkJ;       // per block (forward order)
  std::vector<int> fwd_ops, bwd_ops; // packed ops, one per block, in stream order
  std::vector<int> bwd_of;           // [nblk] forward index -> backward-stream index
  std::vector<FacOp> fac;
  // assembly
  int nT = 0;                        // number of 16x16 blocks of T = sqrt(rho) A_general^T (transposed, 16-row groups)
  std::vector<int> tpos;             // [A.entries()] target double index in T, or -1
  std::vector<int> asm_ptr;          // [nblk+1]
  std::vector<int> asm_a, asm_b;     // T block ids
  std::vector<int> asm_pidx;         // [nblk*256], MFMA C layout [blk][g][lane] -> P ELL entry or -1
  std::vector<int> blk_diag;         // [nblk] J if diagonal block else -1
  std::string error;
};

namespace detail {

// symbolic block fill; pat[J] = set of I >= J (including J). returns total blocks
inline int block_fill(int nb, std::vector<std::set<int>> &pat) {
  int total = 0;
  for (int J = 0; J < nb; J++) {
    pat[J].insert(J);
    std::vector<int> rows(pat[J].begin(), pat[J].end());
    for (size_t a = 1; a < rows.size(); a++)
      for (size_t b = a; b < rows.size(); b++) pat[rows[a]].insert(rows[b]);
    total += (int)rows.size();
  }
  return total;
}

}  // namespace detail

// Build the plan. P: CSC n x n (entries with row > col ignored), A: CSC m x n.
inline Plan build_plan(int n, int m, const int *Pp, const int *Pi, const int *Ap, const int *Ai, int force_ordering = -1) {
  Plan pl;
  pl.n = n; pl.m = m;
  if (n <= 0 || m < 0 || !Pp || !Ap) { pl.error = "invalid dimensions"; return pl; }
  pl.nnzP_in = Pp[n]; pl.nnzA_in = Ap[n];
  for (int j = 0; j < n; j++) {
    if (Pp[j + 1] < Pp[j] || Ap[j + 1] < Ap[j]) { pl.error = "colptr not monotone"; return pl; }
    for (int k = Pp[j]; k < Pp[j + 1]; k++) if (Pi[k] < 0 || Pi[k] >= n) { pl.error = "P row index out of range"; return pl; }
    for (int k = Ap[j]; k < Ap[j + 1]; k++) if (Ai[k] < 0 || Ai[k] >= m) { pl.error = "A row index out of range"; return pl; }
  }
  // rows of A
  struct RE { int col, src; };
  std::vector<std::vector<RE>> arow(m);
  for (int j = 0; j < n; j++) for (int k = Ap[j]; k < Ap[j + 1]; k++) arow[Ai[k]].push_back({j, k});
  pl.singleton.assign(m, 0);
  for (int i = 0; i < m; i++) pl.singleton[i] = arow[i].size() <= 1 ? 1 : 0;
  // triu(P)
  struct PE { int i, j, src; };
  std::vector<PE> ptri;
  for (int j = 0; j < n; j++) for (int k = Pp[j]; k < Pp[j + 1]; k++) if (Pi[k] <= j) ptri.push_back({Pi[k], j, k});
  pl.nnzP_triu = (int)ptri.size();

  // adjacency of M = P + A'A (variables)
  std::vector<std::set<int>> adj(n);
  for (auto &e : ptri) if (e.i != e.j) { adj[e.i].insert(e.j); adj[e.j].insert(e.i); }
  for (int i = 0; i < m; i++) if (!pl.singleton[i])
    for (auto &a : arow[i]) for (auto &b : arow[i]) if (a.col != b.col) adj[a.col].insert(b.col);

  // candidate orderings
  auto eval = [&](const std::vector<int> &pos, std::vector<std::set<int>> *out) {
    int npad = ((n + BS - 1) / BS) * BS, nb = npad / BS;
    std::vector<std::set<int>> pat(nb);
    for (int v = 0; v < n; v++) for (int w : adj[v]) {
      int I = pos[v] / BS, J = pos[w] / BS;
      if (I >= J) pat[J].insert(I);
    }
    int tot = detail::block_fill(nb, pat);
    if (out) *out = pat;
    return tot;
  };
  std::vector<int> pos_nat(n);
  for (int v = 0; v < n; v++) pos_nat[v] = v;
  // hubs-last: repeatedly move the variable with the most "long" edges (|i-j| > 2 blocks) to the back
  std::vector<int> pos_hub(n);
  {
    const int LONG = 2 * BS;
    std::vector<char> hub(n, 0);
    std::vector<int> rank(n);  // position among non-hubs
    for (;;) {
      int r = 0;
      for (int v = 0; v < n; v++) rank[v] = hub[v] ? -1 : r++;
      int best = -1, bestc = 2;  // need at least 3 long edges
      for (int v = 0; v < n; v++) if (!hub[v]) {
        int c = 0;
        for (int w : adj[v]) if (!hub[w] && std::abs(rank[v] - rank[w]) > LONG) c++;
        if (c > bestc) { bestc = c; best = v; }
      }
      if (best < 0) break;
      hub[best] = 1;
    }
    int r = 0;
    for (int v = 0; v < n; v++) if (!hub[v]) pos_hub[v] = r++;
    int h = r;
    for (int v = 0; v < n; v++) if (hub[v]) pos_hub[v] = h++;
  }
  auto eval_gap = [&](const std::vector<int> &pos, int npad, std::vector<std::set<int>> *out) {
    int nb = npad / BS;
    std::vector<std::set<int>> pat(nb);
    for (int v = 0; v < n; v++) for (int w : adj[v]) {
      int I = pos[v] / BS, J = pos[w] / BS;
      if (I >= J) pat[J].insert(I);
    }
    int tot = detail::block_fill(nb, pat);
    if (out) *out = pat;
    return tot;
  };
  int npad_nat = ((n + BS - 1) / BS) * BS, npad_hub = npad_nat;
  int c_nat = eval(pos_nat, nullptr), c_hub = eval_gap(pos_hub, npad_hub, nullptr);
  bool use_hub = c_hub < c_nat;
  if (force_ordering == 0) use_hub = false;
  if (force_ordering == 1) use_hub = true;
  pl.ordering = use_hub ? 1 : 0;
  pl.pos = use_hub ? pos_hub : pos_nat;
  pl.npad = use_hub ? npad_hub : npad_nat;
  pl.nb = pl.npad / BS;
  pl.mpad = ((m + WAVE - 1) / WAVE) * WAVE;
  if (pl.mpad == 0) pl.mpad = WAVE;
  pl.perm.assign(pl.npad, -1);
  for (int v = 0; v < n; v++) pl.perm[pl.pos[v]] = v;
  if (pl.nb >= 32768) { pl.error = "too many blocks"; return pl; }

  // ---- ELL layouts
  {
    std::vector<std::vector<EllEntry>> rows(m);
    for (int i = 0; i < m; i++) for (auto &a : arow[i]) rows[i].push_back({pl.pos[a.col], a.src, 0});
    pl.A = build_ell(rows);
  }
  {
    std::vector<std::vector<EllEntry>> rows(pl.npad);
    for (int j = 0; j < n; j++) for (int k = Ap[j]; k < Ap[j + 1]; k++) rows[pl.pos[j]].push_back({Ai[k], k, pl.singleton[Ai[k]]});
    pl.At = build_ell(rows);
  }
  std::map<std::pair<int, int>, int> pent;  // (posrow, poscol) -> P ELL entry
  {
    std::vector<std::vector<EllEntry>> rows(pl.npad);
    for (auto &e : ptri) {
      rows[pl.pos[e.i]].push_back({pl.pos[e.j], e.src, 0});
      if (e.i != e.j) rows[pl.pos[e.j]].push_back({pl.pos[e.i], e.src, 0});
    }
    pl.P = build_ell(rows);
    for (int r = 0; r < pl.npad; r++) {
      int c = r / WAVE, lane = r % WAVE;
      for (size_t s = 0; s < rows[r].size(); s++) {
        int p = (pl.P.chunk_off[c] + (int)s) * WAVE + lane;
        auto key = std::make_pair(r, rows[r][s].idx);
        if (pent.count(key)) { pl.error = "duplicate entry in P"; return pl; }
        pent[key] = p;
      }
    }
  }

  // ---- block pattern with fill
  std::vector<std::set<int>> pat;
  eval_gap(pl.pos, pl.npad, &pat);
  std::map<std::pair<int, int>, int> bid;
  for (int J = 0; J < pl.nb; J++) {
    // diag first, then ascending I
    bid[{J, J}] = (int)pl.blkI.size(); pl.blkI.push_back(J); pl.blkJ.push_back(J);
    for (int I : pat[J]) if (I != J) { bid[{I, J}] = (int)pl.blkI.size(); pl.blkI.push_back(I); pl.blkJ.push_back(J); }
  }
  pl.nblk = (int)pl.blkI.size();
  pl.fwd_ops.resize(pl.nblk); pl.blk_diag.assign(pl.nblk, -1);
  for (int b = 0; b < pl.nblk; b++) {
    int I = pl.blkI[b], J = pl.blkJ[b];
    if (I == J) { pl.fwd_ops[b] = pack_op(OP_DIAG, J, J); pl.blk_diag[b] = J; }
    else pl.fwd_ops[b] = pack_op(OP_OFF, J, I);            // r_I -= L_IJ y_J
  }
  pl.bwd_of.assign(pl.nblk, -1);
  for (int J = pl.nb - 1; J >= 0; J--) {
    for (int I : pat[J]) if (I != J) { int b = bid[{I, J}]; pl.bwd_of[b] = (int)pl.bwd_ops.size(); pl.bwd_ops.push_back(pack_op(OP_OFF, I, J)); }  // r_J -= L_IJ' x_I
    int b = bid[{J, J}]; pl.bwd_of[b] = (int)pl.bwd_ops.size(); pl.bwd_ops.push_back(pack_op(OP_DIAG, J, J));
  }
  // ---- left-looking block Cholesky op list
  for (int J = 0; J < pl.nb; J++) {
    for (int I : pat[J]) {
      for (int K = 0; K < J; K++) {
        auto ia = bid.find({I, K}), ib = bid.find({J, K});
        if (ia != bid.end() && ib != bid.end()) pl.fac.push_back({FAC_SUB, bid[{I, J}], ia->second, ib->second});
      }
    }
    pl.fac.push_back({FAC_POTRF, bid[{J, J}], 0, 0});
    for (int I : pat[J]) if (I != J) pl.fac.push_back({FAC_TRSM, bid[{I, J}], bid[{J, J}], 0});
  }

  // ---- assembly: general rows in groups of 16 -> T blocks (transposed, [var_local][row_local])
  std::vector<int> gidx(m, -1);
  int mg = 0;
  for (int i = 0; i < m; i++) if (!pl.singleton[i]) gidx[i] = mg++;
  int nrb = (mg + BS - 1) / BS;
  std::map<std::pair<int, int>, int> tid;  // (RB, J) -> T block
  std::vector<std::set<int>> rbcols(nrb);
  for (int i = 0; i < m; i++) if (gidx[i] >= 0) for (auto &a : arow[i]) rbcols[gidx[i] / BS].insert(pl.pos[a.col] / BS);
  for (int rb = 0; rb < nrb; rb++) for (int J : rbcols[rb]) { int t = (int)tid.size(); tid[{rb, J}] = t; }
  pl.nT = (int)tid.size();
  pl.tpos.assign(pl.A.entries(), -1);
  for (int i = 0; i < m; i++) if (gidx[i] >= 0) {
    int c = i / WAVE, lane = i % WAVE;
    for (size_t s = 0; s < arow[i].size(); s++) {
      long p = ((long)pl.A.chunk_off[c] + (long)s) * WAVE + lane;
      int pc = pl.pos[arow[i][s].col];
      int t = tid[{gidx[i] / BS, pc / BS}];
      pl.tpos[p] = t * BLK + (pc % BS) * BS + (gidx[i] % BS);
    }
  }
  pl.asm_ptr.assign(pl.nblk + 1, 0);
  pl.asm_pidx.assign((size_t)pl.nblk * BLK, -1);
  for (int b = 0; b < pl.nblk; b++) {
    int I = pl.blkI[b], J = pl.blkJ[b];
    for (int rb = 0; rb < nrb; rb++) {
      auto ta = tid.find({rb, I}), tb = tid.find({rb, J});
      if (ta != tid.end() && tb != tid.end()) { pl.asm_a.push_back(ta->second); pl.asm_b.push_back(tb->second); }
    }
    pl.asm_ptr[b + 1] = (int)pl.asm_a.size();
    for (int g = 0; g < 4; g++) for (int lane = 0; lane < WAVE; lane++) {
      int row = (lane >> 4) + 4 * g, col = lane & 15;
      auto it = pent.find({I * BS + row, J * BS + col});
      if (it != pent.end()) pl.asm_pidx[(size_t)b * BLK + g * WAVE + lane] = it->second;
    }
  }
  return pl;
}

// workspace layout per QP, in doubles; every section starts on a 16-double (128 B) boundary
struct WsLayout {
  long ellA, ellAt, ellP, Lf, Lb, T, l, u, D, E, dx, dy, stride;
};
inline WsLayout ws_layout(const Plan &pl) {
  WsLayout w; long o = 0;
  auto take = [&](long cnt) { long r = o; o += (cnt + 15) / 16 * 16; return r; };
  w.ellA = take(pl.A.entries()); w.ellAt = take(pl.At.entries()); w.ellP = take(pl.P.entries());
  w.Lf = take((long)pl.nblk * BLK); w.Lb = take((long)pl.nblk * BLK); w.T = take((long)std::max(pl.nT, 1) * BLK);
  w.l = take(pl.mpad); w.u = take(pl.mpad); w.D = take(pl.npad); w.E = take(pl.mpad);
  w.dx = take(pl.npad); w.dy = take(pl.mpad);
  w.stride = o;
  return w;
}
inline long lds_bytes(const Plan &pl) {
  // x, q, r [npad]; z, y, w [mpad]; two padded 16x17 scratch tiles; 64 doubles of reduction scratch
  return (3L * pl.npad + 3L * pl.mpad + 2L * BS * (BS + 1) + 64) * 8L;
}

}  // namespace mpcqp
