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

// pad4: round chunk widths above 2 up to a multiple of 4 (zero entries) so that the latency-bound resident kernels
// need fewer load batches per chunk (17 slots = 16 + 4 instead of 8 + 8 + 1); the streaming kernel keeps exact widths
// pad = 2: for the kernels that keep 8 slots in flight -- round a width up (by at most 3) only where that saves a load batch of the
// 8 / 4 / 2 / 1 decomposition (14 -> 16: two batches instead of three; 17 stays: three either way, and 15 % fewer bytes than 20)
inline int ell_batches8(int w) { int r = w % 8; return w / 8 + (r & 1) + ((r >> 1) & 1) + ((r >> 2) & 1); }
inline Ell build_ell(const std::vector<std::vector<EllEntry>> &rows, int pad4 = 0) {
  Ell e;
  e.nrows = (int)rows.size();
  e.nchunks = (e.nrows + WAVE - 1) / WAVE;
  e.chunk_off.assign(e.nchunks + 1, 0);
  for (int c = 0; c < e.nchunks; c++) {
    size_t w = 0;
    for (int r = c * WAVE; r < std::min(e.nrows, (c + 1) * WAVE); r++) w = std::max(w, rows[r].size());
    if (pad4 == 1 && w > 2) w = (w + 3) / 4 * 4;
    if (pad4 == 2 && w > 2) { size_t best = w; for (size_t w2 = w + 1; w2 <= w + 3; w2++) if (ell_batches8((int)w2) < ell_batches8((int)best)) best = w2; w = best; }
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
  std::vector<std::vector<int>> colrows;  // [nb] off-diagonal block rows I > J of column J, ascending
  std::string error;
};

// ---- "resident" variant: block LDL' (M = W D W', D_J = S_JJ, G_J = D_J^-1, W_IJ = S_IJ G_J) held in LDS.
// Solve: forward t_I -= W_IJ t_J, diagonal x_J = G_J t_J, backward x_J -= W_IJ' x_I -- a single copy of each
// block serves both sweeps and the critical chain is one mat-vec per stage. Ops are scheduled into phases over
// nw waves (barrier between phases unless both neighbours run entirely on one and the same wave).
enum { SOP_SUB = 0, SOP_SUBT = 1, SOP_SET = 2 };
inline int pack_sop(int kind, int slot, int src, int dst) { return kind | (slot << 2) | (src << 14) | (dst << 23); }
struct ResPlan {
  int nw = 1, ntemp = 0, nphase = 0;
  // split accumulation runs: nconst constant blocks follow the nblk factor blocks (slot nblk = -I), rext doubles follow the npad
  // entries of the solve vector (per-wave partial sums, zeroed before every solve)
  int nconst = 0, rext = 0;
  // factorisation, per block column K
  std::vector<int> col_diag;             // [nb]
  std::vector<int> w_ptr, w_slot;        // W_IK = S_IK G_K -> temp tile (index within column)
  std::vector<int> u_ptr, u_dst, u_tmp, u_b;  // S_IJ -= temp(I) * S_JK'
  // the same factorisation by elimination-tree levels: independent block columns of one level are inverted by
  // different waves at once and share the W / Schur-update phases (same-destination updates stay on one wave)
  int nlev = 0;
  std::vector<int> lv_ptr, lv_diag;      // diagonal slots of the columns of each level
  std::vector<int> lw_ptr, lw_slot, lw_g;  // W ops per level: temp tile index = position within the level
  std::vector<int> lu_ptr, lu_dst, lu_tmp, lu_b;  // U ops per (level, wave): lu_ptr[lev * nw + w]
  // solve schedule: ops of phase p, wave w are s_ops[s_ptr[p*nw+w] .. s_ptr[p*nw+w+1])
  std::vector<int> s_ptr, s_ops, s_bar;  // s_bar[p] = 1 if a workgroup barrier must follow phase p
  // the same schedule flattened per wave: op words with SOP_BAR markers where the workgroup synchronises;
  // wave w executes l_ops[l_ptr[w] .. l_ptr[w+1]); the kernel copies this list into LDS once
  std::vector<int> l_ptr, l_ops;
  // and as 4-int records the kernel executes without decoding: {block byte offset, src byte offset in the
  // vector, dst byte offset, flags}; r_ptr[w] .. r_ptr[w+1] in records
  std::vector<int> r_ptr, r_rec;
  // ... and compressed into segments: maximal runs of records whose block / src / dst offsets form an arithmetic
  // progression and whose flags agree. 8 ints per segment {b0, s0, d0, flags, count, db, ds, dd};
  // g_ptr[w] .. g_ptr[w+1] in segments. The kernel runs one tight induction-variable loop per segment.
  std::vector<int> g_ptr, g_seg;
};
constexpr int SOP_BAR = 3;
enum { RF_T = 1, RF_SET = 2, RF_FLUSH = 4, RF_BAR = 8, RF_NOP = 16, RF_PRE = 32 };
// segment flags: SG_EACH = every op writes its own destination; otherwise the ops accumulate into one destination
// which is written at the segment's end iff SG_END (a run split over several segments carries the partial sum on)
// SG_IND: no op of an SG_EACH segment reads a vector block an earlier op of the segment wrote (ops may be interleaved)
enum { SG_T = 1, SG_SET = 2, SG_EACH = 4, SG_END = 8, SG_BAR = 16, SG_NOP = 32, SG_IND = 64 };

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
inline Plan build_plan(int n, int m, const int *Pp, const int *Pi, const int *Ap, const int *Ai, int force_ordering = -1, int pad4 = 0, int max_sep = 3) {
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
  if (force_ordering >= 1) use_hub = true;
  // ordering 2 ("twisted"): when the non-hub blocks form a path (block tridiagonal), interleave them from both ends
  // [0, k-1, 1, k-2, ...]: the elimination tree becomes two chains that meet in the middle -- same block count, half
  // the sequential depth (two waves eliminate / substitute concurrently).
  bool twisted = false;
  if (force_ordering >= 2 && force_ordering != 4) {
    int nonhub = 0;   // hubs sit behind all non-hubs; count the leading variables that kept relative order 0..r-1
    {
      std::vector<int> inv(n, -1);
      for (int v = 0; v < n; v++) inv[pos_hub[v]] = v;
      // non-hubs are those whose successor in position order has a larger variable index (monotone prefix)
      nonhub = n;
      for (int t = 1; t < n; t++) if (inv[t] < inv[t - 1]) { nonhub = t; break; }
    }
    // ordering 3: the same, for chains whose last block is only partly filled (cart-pole: 500 stage variables = 31 1/4 blocks): the hubs move
    // up to the next block boundary -- padding positions inside the vector, one more block row -- so that the chain part is whole blocks
    // and can be twisted; accepted when that costs at most four blocks
    std::vector<int> pos_pad(pos_hub);
    int npad_pad = npad_hub;
    const bool padded = force_ordering == 3 && nonhub % BS != 0 && nonhub < n;
    if (padded) {
      const int shift = (nonhub + BS - 1) / BS * BS - nonhub;
      for (int v = 0; v < n; v++) if (pos_hub[v] >= nonhub) pos_pad[v] = pos_hub[v] + shift;
      npad_pad = ((n + shift + BS - 1) / BS) * BS;
    }
    const int k = (nonhub + BS - 1) / BS;
    if ((nonhub % BS == 0 || padded) && k >= 4) {
      std::vector<std::set<int>> badj(k);
      bool path = true;
      for (int v = 0; v < n && path; v++) if (pos_hub[v] < nonhub) for (int w : adj[v]) if (pos_hub[w] < nonhub) {
        const int a = pos_hub[v] / BS, b = pos_hub[w] / BS;
        if (std::abs(a - b) > 1) { path = false; break; }
      }
      if (path) {
        std::vector<int> pos_tw(pos_pad);
        for (int v = 0; v < n; v++) if (pos_hub[v] < nonhub) {
          const int b = pos_hub[v] / BS, nbk = b < (k + 1) / 2 ? 2 * b : 2 * (k - 1 - b) + 1;
          pos_tw[v] = nbk * BS + pos_hub[v] % BS;
        }
        if (eval_gap(pos_tw, npad_pad, nullptr) <= c_hub + (padded ? 4 : 0)) { pos_hub = pos_tw; npad_hub = npad_pad; twisted = true; }
      }
    }
  }
  // ordering 4 ("dissected"): where the hub block has room beside the hub variables, vertex separators of the chain move into it -- the variables right of a cut
  // that a variable left of it couples to (the state x_s of one stage of an OCP) -- and the chain falls apart into segments that only the hub connects.  Every
  // segment is whole blocks, twisted by itself: twice as many elimination chains per separator, each a fraction of the length, and no block more (the fill
  // of a separator lands in the hub blocks W_hub,p, which every column has anyway).  Cart-pole N=100: 3 separators of 4 states beside the 4 parameters,
  // four segments of 8 blocks, eight chains of 4 where the twisted order has two of 16.
  bool dissected = false;
  if (force_ordering == 4 && use_hub) {
    int nonhub = n;
    std::vector<int> inv(n, -1);
    for (int v = 0; v < n; v++) inv[pos_hub[v]] = v;
    for (int t = 1; t < n; t++) if (inv[t] < inv[t - 1]) { nonhub = t; break; }
    const int nhubv = n - nonhub;
    // right boundary of every cut: rb[c] = non-hub variables at order >= c with a neighbour at order < c
    auto right_boundary = [&](int c) {
      std::vector<int> r;
      for (int t = c; t < nonhub && t < c + 4 * BS; t++) { const int w = inv[t]; for (int v : adj[w]) if (pos_hub[v] < c) { r.push_back(t); break; } }
      return r;
    };
    int s0 = BS + 1;
    for (int c = BS; c + BS <= nonhub; c++) s0 = std::min(s0, (int)right_boundary(c).size());
    const int want_sep = std::max(1, max_sep);      // (three separators = four twisted pairs for eight waves; one = two pairs for four)
    const int S = (nhubv > 0 && nhubv < BS && s0 >= 1 && s0 <= BS) ? std::min(want_sep, (BS - nhubv) / s0) : 0;
    if (S >= 1 && nonhub >= 4 * BS * (S + 1)) {
      std::vector<char> is_sep(nonhub, 0);
      std::vector<int> cuts;
      bool okc = true;
      for (int i = 1; i <= S && okc; i++) {
        const int ideal = (int)((long)i * nonhub / (S + 1));
        int best = -1;
        for (int d = 0; d <= 2 * BS && best < 0; d++) for (int c : {ideal - d, ideal + d}) {
          if (c <= (cuts.empty() ? BS : cuts.back() + 2 * BS) || c + BS > nonhub) continue;
          if ((int)right_boundary(c).size() == s0) { best = c; break; }
        }
        if (best < 0) { okc = false; break; }
        for (int t : right_boundary(best)) is_sep[t] = 1;
        cuts.push_back(best);
      }
      if (okc) {
        std::vector<int> pos_ds(n, -1);
        int base = 0;
        std::vector<std::pair<int, int>> segs;      // [first order index, one past the last) of every segment
        for (int i = 0; i <= S; i++) { const int a = i == 0 ? 0 : cuts[i - 1], b = i == S ? nonhub : cuts[i]; segs.push_back({a, b}); }
        for (auto &sg : segs) {
          std::vector<int> mem;
          for (int t = sg.first; t < sg.second; t++) if (!is_sep[t]) mem.push_back(inv[t]);
          const int kb = ((int)mem.size() + BS - 1) / BS;
          for (size_t r = 0; r < mem.size(); r++) {
            const int b = (int)r / BS, nbk = b < (kb + 1) / 2 ? 2 * b : 2 * (kb - 1 - b) + 1;
            pos_ds[mem[r]] = base + nbk * BS + (int)r % BS;
          }
          base += kb * BS;
        }
        int h = base;
        for (int t = nonhub; t < n; t++) pos_ds[inv[t]] = h++;
        for (int t = 0; t < nonhub; t++) if (is_sep[t]) pos_ds[inv[t]] = h++;
        if (h <= base + BS) { pos_hub = pos_ds; npad_hub = base + BS; twisted = true; dissected = true; }
      }
    }
    if (!dissected) { pl.error = "the dissected ordering does not apply"; return pl; }
  }
  pl.ordering = use_hub ? (dissected ? 4 : twisted ? 2 : 1) : 0;
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
    pl.A = build_ell(rows, pad4);
  }
  {
    std::vector<std::vector<EllEntry>> rows(pl.npad);
    for (int j = 0; j < n; j++) for (int k = Ap[j]; k < Ap[j + 1]; k++) rows[pl.pos[j]].push_back({Ai[k], k, pl.singleton[Ai[k]]});
    pl.At = build_ell(rows, pad4);
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
  pl.colrows.assign(pl.nb, {});
  for (int J = 0; J < pl.nb; J++) for (int I : pat[J]) if (I != J) pl.colrows[J].push_back(I);
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

// ---- dense tiles of A for the two sweeps of the ADMM iteration (on-chip kernels)
// The dynamics rows of a stage OCP carry a dense Jacobian block per stage (12 x 16 on the 12-state quadrotor): in the ELL layouts those rows
// are 17 slots wide, A and A' are two copies of the same numbers (77 KB per QP and iteration), and a sweep is a chain of dependent load
// batches.  Here general rows are grouped by the 16-column block that holds most of their entries; a group of up to 16 rows whose entries in
// that block are dense enough becomes a 16 x 16 tile, stored once, in the operand layout of the 4-block MFMA -- element (r, c) at
// [lane = r + 16 (c & 3)][K = c >> 2] -- which serves A x as it is (one 32-byte load per lane) and A' w through four 8-byte loads per lane
// (each instruction four whole 128-byte lines).  What is left of A (singleton rows, the entries outside the tiles) stays in ELL form
// (Ar by row, Atr by variable, same row / lane mapping as A / A').  Values are the same numbers as in the ELL arrays (same scaling
// expression); only the order of summation inside a row differs.
struct TilePlan {
  bool on = false;
  int ntile = 0;
  std::vector<int> tJ;            // [ntile + 1] column block; the last entry is the zero tile (never written, all rows padding)
  std::vector<int> rowid;         // [(ntile + 1) * 16] row of A behind row r of the tile, -1 = padding
  std::vector<int> tsrc;          // [(ntile + 1) * 256] index into the caller's A values, storage order [lane][K], -1 = structural zero
  Ell Ar, Atr;                    // A and A' without the tile entries
  std::vector<int> ta_ptr, ta_tid;   // tiles with a row in chunk c of A's rows: ta_tid[ta_ptr[c] .. ta_ptr[c + 1])
  std::vector<int> tt_ptr, tt_tid;   // tiles of column block J: tt_tid[tt_ptr[J] .. tt_ptr[J + 1])
  int max_per_chunk = 0, max_per_block = 0;
  bool rows_consecutive = true;   // the rows of every tile are consecutive rows of A (what the kernels' per-tile {first row, rows} records assume)
  long entries_in_tiles = 0;
};
constexpr int TILE_MIN_PER_ROW = 6;       // a group becomes a tile when its rows have on average this many entries in its block ...
constexpr int TILE_MIN_ENTRIES = 72;      // ... and at least this many together

inline TilePlan build_tile_plan(const Plan &pl, int n, int m, const int *Ap, const int *Ai, int pad) {
  TilePlan tp;
  struct RE { int col, src; };
  std::vector<std::vector<RE>> arow(m);
  for (int j = 0; j < n; j++) for (int k = Ap[j]; k < Ap[j + 1]; k++) arow[Ai[k]].push_back({j, k});
  // dominant block of every general row
  std::vector<int> dom(m, -1), cnt(m, 0);
  for (int i = 0; i < m; i++) if (!pl.singleton[i]) {
    std::map<int, int> c;
    for (auto &e : arow[i]) c[pl.pos[e.col] / BS]++;
    for (auto &kv : c) if (kv.second > cnt[i]) { cnt[i] = kv.second; dom[i] = kv.first; }
  }
  std::vector<char> in_tile(pl.nnzA_in, 0);
  std::vector<int> group;
  auto flush = [&]() {
    if (group.empty()) return;
    long tot = 0; for (int i : group) tot += cnt[i];
    if (tot >= TILE_MIN_ENTRIES && tot >= (long)TILE_MIN_PER_ROW * (long)group.size()) {
      const int J = dom[group[0]], t = tp.ntile++;
      tp.tJ.push_back(J);
      tp.rowid.resize((size_t)(t + 1) * BS, -1); tp.tsrc.resize((size_t)(t + 1) * BLK, -1);
      for (size_t r = 0; r < group.size(); r++) {
        const int i = group[r];
        tp.rowid[(size_t)t * BS + r] = i;
        for (auto &e : arow[i]) if (pl.pos[e.col] / BS == J) {
          const int c = pl.pos[e.col] % BS;
          tp.tsrc[(size_t)t * BLK + ((int)r + BS * (c & 3)) * 4 + (c >> 2)] = e.src;
          in_tile[e.src] = 1; tp.entries_in_tiles++;
        }
      }
    }
    group.clear();
  };
  for (int i = 0; i < m; i++) {
    if (pl.singleton[i] || dom[i] < 0) { flush(); continue; }
    if (!group.empty() && (dom[group[0]] != dom[i] || (int)group.size() == BS)) flush();
    group.push_back(i);
  }
  flush();
  if (tp.ntile == 0) return tp;
  tp.on = true;
  // the zero tile: block 0, no rows, no entries
  tp.tJ.push_back(0); tp.rowid.resize((size_t)(tp.ntile + 1) * BS, -1); tp.tsrc.resize((size_t)(tp.ntile + 1) * BLK, -1);
  // remainder ELL layouts (same row -> lane and variable -> lane mapping as A and A')
  {
    std::vector<std::vector<EllEntry>> rows(m);
    for (int i = 0; i < m; i++) for (auto &a : arow[i]) if (!in_tile[a.src]) rows[i].push_back({pl.pos[a.col], a.src, 0});
    tp.Ar = build_ell(rows, pad);
  }
  {
    std::vector<std::vector<EllEntry>> rows(pl.npad);
    for (int j = 0; j < n; j++) for (int k = Ap[j]; k < Ap[j + 1]; k++) if (!in_tile[k]) rows[pl.pos[j]].push_back({Ai[k], k, pl.singleton[Ai[k]]});
    tp.Atr = build_ell(rows, pad);
  }
  // which tiles a chunk of A's rows needs (a tile whose rows straddle two chunks is listed in both), which tiles a column block has
  const int nA = pl.A.nchunks;
  std::vector<std::vector<int>> per_chunk(nA), per_block(pl.nb);
  for (int t = 0; t < tp.ntile; t++) {
    std::set<int> cs;
    for (int r = 0; r < BS; r++) if (tp.rowid[(size_t)t * BS + r] >= 0) cs.insert(tp.rowid[(size_t)t * BS + r] / WAVE);
    for (int c : cs) per_chunk[c].push_back(t);
    per_block[tp.tJ[t]].push_back(t);
  }
  for (int t = 0; t < tp.ntile; t++) for (int r = 1; r < BS; r++) {
    const int a = tp.rowid[(size_t)t * BS + r - 1], b = tp.rowid[(size_t)t * BS + r];
    if (b >= 0 && b != a + 1) tp.rows_consecutive = false;
  }
  tp.ta_ptr.push_back(0);
  for (int c = 0; c < nA; c++) { for (int t : per_chunk[c]) tp.ta_tid.push_back(t); tp.ta_ptr.push_back((int)tp.ta_tid.size()); tp.max_per_chunk = std::max(tp.max_per_chunk, (int)per_chunk[c].size()); }
  tp.tt_ptr.push_back(0);
  for (int J = 0; J < pl.nb; J++) { for (int t : per_block[J]) tp.tt_tid.push_back(t); tp.tt_ptr.push_back((int)tp.tt_tid.size()); tp.max_per_block = std::max(tp.max_per_block, (int)per_block[J].size()); }
  return tp;
}

// workspace layout per QP, in doubles; every section starts on a 16-double (128 B) boundary
// INVARIANT the resident kernels rely on: the slab is zeroed once, at mpcqp_create, and inside the T region a kernel only ever writes the
// structural non-zeros (positions tpos >= 0); the other entries of the T tiles and the extra zero tile behind them stay 0 for the life of the
// handle (the on-chip assembly reads that zero tile unconditionally: oc_asm_records).  The streaming kernel rewrites T[0 .. nT) completely and
// never reads the zero tile.  The same holds for the zero tile behind the dense A tiles (TilePlan).  A kernel that wants scratch space in the
// slab must not take it from T.
struct WsLayout {
  long ellA = 0, ellAt = 0, ellP = 0, Lf = 0, Lb = 0, T = 0, l = 0, u = 0, D = 0, E = 0, dx = 0, dy = 0, Zg = 0, Yg = 0, stride = 0;
  long tile = 0, ellAr = 0, ellAtr = 0;      // dense tiles of A + the remainder ELL values (on-chip kernels with a TilePlan), behind everything else
};
inline WsLayout ws_layout(const Plan &pl, const TilePlan *tp = nullptr) {
  WsLayout w; long o = 0;
  auto take = [&](long cnt) { long r = o; o += (cnt + 15) / 16 * 16; return r; };
  w.ellA = take(pl.A.entries()); w.ellAt = take(pl.At.entries()); w.ellP = take(pl.P.entries());
  w.Lf = take(((long)pl.nblk + 1) * BLK);   // + the constant -I block of split runs
  w.Lb = take((long)pl.nblk * BLK); w.T = take(((long)std::max(pl.nT, 1) + 1) * BLK);   // + a tile of zeros (never written after creation)
  w.l = take(pl.mpad); w.u = take(pl.mpad); w.D = take(pl.npad); w.E = take(pl.mpad);
  w.dx = take(pl.npad); w.dy = take(pl.mpad);
  w.Zg = take(pl.mpad); w.Yg = take(pl.mpad);   // z, y of the kernels that keep them out of LDS (row-indexed only, like l and u)
  if (tp && tp->on) { w.tile = take(((long)tp->ntile + 1) * BLK); w.ellAr = take(tp->Ar.entries()); w.ellAtr = take(tp->Atr.entries()); }
  w.stride = o;
  return w;
}
inline int block_id(const Plan &pl, int I, int J) {
  for (int b = 0; b < pl.nblk; b++) if (pl.blkI[b] == I && pl.blkJ[b] == J) return b;
  return -1;
}

// split_runs: split long single-destination accumulation runs of the forward sweep over the waves (pays when the factor blocks are
// streamed from HBM -- cart-pole N=100 -10 %, quadrotor N=50 -7 %, N=20 -2.7 % -- not when they sit in LDS)
inline ResPlan build_res_plan(const Plan &pl, int nw, bool split_runs = false) {
  ResPlan rp; rp.nw = nw;
  const int nb = pl.nb;
  std::map<std::pair<int, int>, int> bid;
  for (int b = 0; b < pl.nblk; b++) bid[{pl.blkI[b], pl.blkJ[b]}] = b;
  // ---- factor plan (right-looking)
  rp.w_ptr.push_back(0); rp.u_ptr.push_back(0);
  for (int K = 0; K < nb; K++) {
    rp.col_diag.push_back(bid[{K, K}]);
    const std::vector<int> &rows = pl.colrows[K];
    rp.ntemp = std::max(rp.ntemp, (int)rows.size());
    for (size_t a = 0; a < rows.size(); a++) rp.w_slot.push_back(bid[{rows[a], K}]);
    rp.w_ptr.push_back((int)rp.w_slot.size());
    for (size_t a = 0; a < rows.size(); a++) for (size_t b = 0; b <= a; b++) {
      rp.u_dst.push_back(bid.at({rows[a], rows[b]})); rp.u_tmp.push_back((int)a); rp.u_b.push_back(bid[{rows[b], K}]);
    }
    rp.u_ptr.push_back((int)rp.u_dst.size());
  }
  // ---- level-parallel factor plan
  {
    std::vector<int> clev(nb, 0);
    for (int K = 0; K < nb; K++) for (int I : pl.colrows[K]) clev[I] = std::max(clev[I], clev[K] + 1);
    rp.nlev = nb ? *std::max_element(clev.begin(), clev.end()) + 1 : 0;
    rp.lv_ptr.push_back(0); rp.lw_ptr.push_back(0); rp.lu_ptr.push_back(0);
    int maxw = 0;
    for (int lev = 0; lev < rp.nlev; lev++) {
      struct U { int dst, tmp, b; };
      std::map<int, std::vector<U>> groups;
      int nwops = 0;
      for (int K = 0; K < nb; K++) if (clev[K] == lev) {
        rp.lv_diag.push_back(bid[{K, K}]);
        const std::vector<int> &rows = pl.colrows[K];
        const int t0 = nwops;
        for (size_t a = 0; a < rows.size(); a++) { rp.lw_slot.push_back(bid[{rows[a], K}]); rp.lw_g.push_back(bid[{K, K}]); nwops++; }
        for (size_t a = 0; a < rows.size(); a++) for (size_t b = 0; b <= a; b++) {
          const int dst = bid.at({rows[a], rows[b]});
          groups[dst].push_back({dst, t0 + (int)a, bid[{rows[b], K}]});
        }
      }
      maxw = std::max(maxw, nwops);
      rp.lv_ptr.push_back((int)rp.lv_diag.size());
      rp.lw_ptr.push_back((int)rp.lw_slot.size());
      std::vector<std::vector<U>> gs, per(nw);
      for (auto &kv : groups) gs.push_back(kv.second);
      std::stable_sort(gs.begin(), gs.end(), [](const std::vector<U> &a, const std::vector<U> &b) { return a.size() > b.size(); });
      for (auto &g : gs) {
        int best = 0;
        for (int w = 1; w < nw; w++) if (per[w].size() < per[best].size()) best = w;
        per[best].insert(per[best].end(), g.begin(), g.end());
      }
      for (int w = 0; w < nw; w++) {
        for (auto &u : per[w]) { rp.lu_dst.push_back(u.dst); rp.lu_tmp.push_back(u.tmp); rp.lu_b.push_back(u.b); }
        rp.lu_ptr.push_back((int)rp.lu_dst.size());
      }
    }
    rp.ntemp = std::max(rp.ntemp, maxw);
  }
  // ---- solve schedule: backward ASAP levels at op granularity, forward = mirror image
  std::vector<int> fin(nb, 0);
  int maxlev = -1;
  struct Op { int slot, I, J, lev; };
  std::vector<Op> ops;
  for (int J = nb - 1; J >= 0; J--) {
    int f = 0;
    for (int I : pl.colrows[J]) { ops.push_back({bid[{I, J}], I, J, fin[I]}); f = std::max(f, fin[I] + 1); maxlev = std::max(maxlev, fin[I]); }
    fin[J] = pl.colrows[J].empty() ? 0 : f;
  }
  const int nlev = maxlev + 1;
  // phases: forward nlev (levels descending), 1 diagonal, backward nlev (levels ascending).
  // Wave assignment: ops sharing a destination stay on one wave (register accumulation, no write races); a
  // single-op group follows the wave that produced its source block ("affinity") so that independent chains stay on
  // their own waves without synchronisation; everything else goes to the least-loaded wave.
  struct SOp { int word, src, dst; };
  std::vector<int> owner(nb + nw, -1);
  auto distribute = [&](std::vector<std::vector<SOp>> groups) {
    std::vector<std::vector<SOp>> per(nw);
    std::stable_sort(groups.begin(), groups.end(), [](const std::vector<SOp> &a, const std::vector<SOp> &b) { return a.size() > b.size(); });
    const bool few = (int)groups.size() <= nw;
    std::vector<char> taken(nw, 0), done(groups.size(), 0);
    auto place = [&](size_t gi, int w) {
      taken[w] = 1; done[gi] = 1;
      per[w].insert(per[w].end(), groups[gi].begin(), groups[gi].end());
      owner[groups[gi][0].dst] = w;
    };
    if (few)   // first the groups that can follow the wave that produced their source block
      for (size_t gi = 0; gi < groups.size(); gi++) {
        const auto &g = groups[gi];
        if (g.size() == 1 && owner[g[0].src] >= 0 && !taken[owner[g[0].src]]) place(gi, owner[g[0].src]);
      }
    for (size_t gi = 0; gi < groups.size(); gi++) if (!done[gi]) {
      int best = -1;
      for (int w = 0; w < nw; w++) if ((!few || !taken[w]) && (best < 0 || per[w].size() < per[best].size())) best = w;
      if (best < 0) { best = 0; for (int w = 1; w < nw; w++) if (per[w].size() < per[best].size()) best = w; }
      place(gi, best);
    }
    return per;
  };
  std::vector<std::vector<std::vector<SOp>>> phases;
  const int vslots = nb + nw;                       // vector blocks: nb of the solve vector + one partial-sum slot per wave
  for (int lev = nlev - 1; lev >= 0; lev--) {       // forward: t_I -= W_IJ t_J, grouped by dst I
    std::map<int, std::vector<SOp>> g;
    for (auto &o : ops) if (o.lev == lev) g[o.I].push_back({pack_sop(SOP_SUB, o.slot, o.J, o.I), o.J, o.I});
    std::vector<std::vector<SOp>> gs; for (auto &kv : g) gs.push_back(kv.second);
    // A destination that collects many contributions (the arrow block: one op per stage) would be one long run on one wave
    // while the others wait.  Split it: wave 0's share accumulates into the destination itself, the other shares into
    // per-wave partial-sum slots behind the solve vector (zero at the start of every solve, so they end up holding
    // -sum W t), and a combine phase adds them with the constant block -I:  t_I -= (-I) * partial.
    std::vector<SOp> combine;
    if (split_runs && nw > 1 && gs.size() == 1 && (int)gs[0].size() >= 4 * nw && vslots <= 511 && rp.rext == 0) {
      std::vector<SOp> big = gs[0]; gs.clear();
      const int I = big[0].dst, per = ((int)big.size() + nw - 1) / nw;
      for (int w = 0; w < nw; w++) {
        std::vector<SOp> part;
        for (int q = w * per; q < std::min((int)big.size(), (w + 1) * per); q++) {
          SOp o = big[q];
          if (w > 0) { unsigned wd = (unsigned)o.word; o.dst = nb + w; o.word = (int)((wd & 0x7fffffu) | ((unsigned)o.dst << 23)); }
          part.push_back(o);
        }
        if (!part.empty()) gs.push_back(part);
        if (w > 0 && !part.empty()) combine.push_back({pack_sop(SOP_SUB, pl.nblk, nb + w, I), nb + w, I});
      }
      rp.nconst = 1; rp.rext = nw * BS;
    }
    phases.push_back(distribute(gs));
    if (!combine.empty()) phases.push_back(distribute({combine}));
  }
  {
    std::vector<std::vector<SOp>> gs;
    for (int J = 0; J < nb; J++) gs.push_back({{pack_sop(SOP_SET, bid[{J, J}], J, J), J, J}});
    std::fill(owner.begin(), owner.end(), -1);
    phases.push_back(distribute(gs));
  }
  std::fill(owner.begin(), owner.end(), -1);
  for (int lev = 0; lev < nlev; lev++) {            // backward: x_J -= W_IJ' x_I, grouped by dst J
    std::map<int, std::vector<SOp>> g;
    for (auto &o : ops) if (o.lev == lev) g[o.J].push_back({pack_sop(SOP_SUBT, o.slot, o.I, o.J), o.I, o.J});
    std::vector<std::vector<SOp>> gs; for (auto &kv : g) gs.push_back(kv.second);
    phases.push_back(distribute(gs));
  }
  rp.nphase = (int)phases.size();
  rp.s_ptr.push_back(0);
  // a workgroup barrier is needed before a phase iff one of its ops touches a vector block that another wave has
  // written (RAW / WAW) or read-then-it-writes (WAR) since the last barrier
  std::vector<int> wby(nb + nw, -1), rby(nb + nw, 0);
  rp.s_bar.assign(rp.nphase, 0);
  for (int p = 0; p < rp.nphase; p++) {
    bool conflict = false;
    for (int w = 0; w < nw && !conflict; w++) for (auto &o : phases[p][w]) {
      if ((wby[o.src] >= 0 && wby[o.src] != w) || (wby[o.dst] >= 0 && wby[o.dst] != w) || (rby[o.dst] & ~(1 << w))) { conflict = true; break; }
    }
    if (conflict && p > 0) { rp.s_bar[p - 1] = 1; std::fill(wby.begin(), wby.end(), -1); std::fill(rby.begin(), rby.end(), 0); }
    for (int w = 0; w < nw; w++) for (auto &o : phases[p][w]) { wby[o.dst] = w; rby[o.src] |= 1 << w; rby[o.dst] |= 1 << w; }
    for (int w = 0; w < nw; w++) { for (auto &o : phases[p][w]) rp.s_ops.push_back(o.word); rp.s_ptr.push_back((int)rp.s_ops.size()); }
  }
  rp.s_bar[rp.nphase - 1] = 1;
  if (nw == 1) std::fill(rp.s_bar.begin(), rp.s_bar.end(), 0);
  rp.l_ptr.push_back(0);
  for (int w = 0; w < nw; w++) {
    for (int p = 0; p < rp.nphase; p++) {
      for (int q = rp.s_ptr[p * nw + w]; q < rp.s_ptr[p * nw + w + 1]; q++) rp.l_ops.push_back(rp.s_ops[q]);
      if (rp.s_bar[p]) rp.l_ops.push_back(SOP_BAR);
    }
    rp.l_ptr.push_back((int)rp.l_ops.size());
  }
  // records: runs (same dst, same kind, same phase) are flushed once; RF_PRE marks ops after whose vector read the
  // next op's block / old-destination fetch may already be issued (next op writes another block, no barrier between)
  rp.r_ptr.push_back(0);
  for (int w = 0; w < nw; w++) {
    struct R { int b, s, d, f, dstblk; };
    std::vector<R> recs;
    for (int p = 0; p < rp.nphase; p++) {
      const int a = rp.s_ptr[p * nw + w], e = rp.s_ptr[p * nw + w + 1];
      for (int q = a; q < e; q++) {
        unsigned op = (unsigned)rp.s_ops[q]; int kind = op & 3, slot = (op >> 2) & 0xfff, src = (op >> 14) & 0x1ff, dst = op >> 23;
        int f = (kind == SOP_SUBT ? RF_T : 0) | (kind == SOP_SET ? RF_SET : 0);
        bool cont = false;
        if (q + 1 < e && kind != SOP_SET) { unsigned nx = (unsigned)rp.s_ops[q + 1]; cont = (int)(nx & 3) == kind && (int)(nx >> 23) == dst; }
        if (!cont) f |= RF_FLUSH;
        recs.push_back({slot * BLK * 8, src * BS * 8, dst * BS * 8, f, dst});
      }
      if (rp.s_bar[p]) { if (a < e) recs.back().f |= RF_BAR; else recs.push_back({0, 0, 0, RF_NOP, -1}); }
    }
    for (size_t i = 0; i + 1 < recs.size(); i++) {
      const R &c = recs[i], &n = recs[i + 1];
      if ((c.f & (RF_NOP | RF_BAR)) || (n.f & RF_NOP)) continue;
      if ((c.f & RF_FLUSH) && n.dstblk == c.dstblk) continue;
      recs[i].f |= RF_PRE;
    }
    for (auto &r : recs) { rp.r_rec.push_back(r.b); rp.r_rec.push_back(r.s); rp.r_rec.push_back(r.d); rp.r_rec.push_back(r.f); }
    rp.r_ptr.push_back((int)rp.r_rec.size() / 4);
    // ---- segments
    if (w == 0) rp.g_ptr.push_back(0);
    size_t i = 0;
    while (i < recs.size()) {
      const R &c = recs[i];
      if (c.f & RF_NOP) { int sg[8] = {0, 0, 0, SG_NOP, 0, 0, 0, 0}; rp.g_seg.insert(rp.g_seg.end(), sg, sg + 8); i++; continue; }
      const int tflag = (c.f & RF_T ? SG_T : 0) | (c.f & RF_SET ? SG_SET : 0);
      size_t e = i + 1;
      int db = 0, ds = 0, dd = 0, flags;
      if (c.f & RF_FLUSH) {                       // ops that each write their own destination
        flags = tflag | SG_EACH;
        if (!(c.f & RF_BAR)) {
          while (e < recs.size()) {
            const R &n = recs[e], &p = recs[e - 1];
            if ((n.f & RF_NOP) || !(n.f & RF_FLUSH) || (n.f & (RF_T | RF_SET)) != (c.f & (RF_T | RF_SET))) break;
            if (e == i + 1) { db = n.b - p.b; ds = n.s - p.s; dd = n.d - p.d; if (dd == 0) break; }
            else if (n.b - p.b != db || n.s - p.s != ds || n.d - p.d != dd) break;
            e++;
            if (n.f & RF_BAR) break;
          }
        }
        if (recs[e - 1].f & RF_BAR) flags |= SG_BAR;
        bool ind = e - i >= 2;
        for (size_t a = i; a < e && ind; a++) for (size_t b = i; b < a; b++)
          if (recs[a].s == recs[b].d || recs[a].d == recs[b].d || (recs[a].d == recs[b].s && !(recs[a].s == recs[a].d && a == b))) { ind = false; break; }
        if (ind) flags |= SG_IND;
      } else {                                    // a run accumulating into one destination
        flags = tflag;
        while (e < recs.size()) {
          const R &n = recs[e], &p = recs[e - 1];
          if (e == i + 1) { db = n.b - p.b; ds = n.s - p.s; }
          else if (n.b - p.b != db || n.s - p.s != ds) break;
          e++;
          if (n.f & RF_FLUSH) break;
        }
        if (recs[e - 1].f & RF_FLUSH) flags |= SG_END;
        if (recs[e - 1].f & RF_BAR) flags |= SG_BAR;
      }
      int sg[8] = {c.b, c.s, c.d, flags, (int)(e - i), db, ds, dd};
      rp.g_seg.insert(rp.g_seg.end(), sg, sg + 8);
      i = e;
    }
    rp.g_ptr.push_back((int)rp.g_seg.size() / 8);
  }
  return rp;
}

// LDS footprint of the resident variant: blocks + temp tiles + x, q, r [npad] + z, y, w [mpad] + scratch
// doubles of the LDS region that holds the factor blocks + temp tiles during the solve and, before the first
// factorisation, the staged ELL values of A, A', P
inline long res_stage_doubles(const Plan &pl, const ResPlan &rp) {
  const long blocks = ((long)pl.nblk + rp.nconst + rp.ntemp) * BLK, ell = pl.A.entries() + pl.At.entries() + pl.P.entries();
  return (std::max(blocks, ell) + 15) / 16 * 16;
}
// the same kernels with the factor blocks left in the HBM slab: LDS holds temp tiles + vectors + schedule only
// The temp tiles are live only inside the level loop of the factorisation, where w (= rho z - y between factorisations, the rho
// vector during one) is dead if rho is recomputed from the bounds at the end: with mpad >= ntemp * 256 they alias w and cost no LDS.
inline bool gb_tmp_alias(const Plan &pl, const ResPlan &rp) { return (long)pl.mpad >= (long)rp.ntemp * BLK; }
inline long res_stage_doubles_gb(const Plan &pl, const ResPlan &rp) { return gb_tmp_alias(pl, rp) ? 0 : ((long)rp.ntemp * BLK + 15) / 16 * 16; }
// zy_global: z and y live in the slab as well (they are only ever indexed by their own row, so wave accesses are contiguous)
inline long lds_bytes_res_gb(const Plan &pl, const ResPlan &rp, bool zy_global = false) {
  const long sched_words = ((long)rp.g_seg.size() + rp.nw + 1 + 1) / 2 + 4;
  return (res_stage_doubles_gb(pl, rp) + 3L * pl.npad + rp.rext + (zy_global ? 1L : 3L) * pl.mpad + 16L * rp.nw + 16 + 16L * rp.nw + sched_words) * 8L;
}
inline long lds_bytes_res(const Plan &pl, const ResPlan &rp) {
  const long sched_words = ((long)rp.g_seg.size() + rp.nw + 1 + 1) / 2 + 4;   // int32 segments kept in LDS, in doubles
  return (res_stage_doubles(pl, rp) + 3L * pl.npad + rp.rext + 3L * pl.mpad + 16L * rp.nw + 16 + 16L * rp.nw + sched_words) * 8L;
}

// ---- "on-chip" variant (kernel_onchip.hpp): the whole factor stays on the CU at two workgroups per CU -- chain blocks and part
// of the hub blocks in LDS (one copy serves the forward sweep and, read transposed, the backward sweep), the inverse diagonal
// blocks G_p and the remaining hub blocks in registers -- and the solve is written for the one topology MPC problems have: block
// tridiagonal (one or two elimination chains: plain or twisted order) plus an optional arrow (the hub block, last).
//   position p < nbc: a block of the chain part; hub position = nbc (when has_hub).  succ(p) = the one non-hub off-diagonal block
//   row of column p.  Two chains may meet in their common last element f (twisted order): chainE ends with e, succ(e) = f = the
//   last element of chainF.
// Every mat-vec runs on the matrix cores (v_mfma_f64_16x16x4_f64, vector in the B operand's column 0): the result layout of one
// op is the operand layout of the next, so a chain runs register to register.
struct OcPlan {
  bool ok = false;
  int nbc = 0, has_hub = 0, junc = 0;
  std::vector<int> chainE, chainF;          // positions in elimination (forward) order
  // Several twisted pairs (the dissected ordering, build_plan ordering 4): pairs[0] = {chainE, chainF, junc}; the chains of one pair meet in their junction,
  // pairs meet nowhere but in the hub.  Chain waves 2 i, 2 i + 1 walk pair i in the solve; the factorisation takes the pairs one after the other.
  struct Pair { std::vector<int> E, F; int junc = 0; int oE = 0, oF = 0; };
  std::vector<Pair> pairs;
  int o_pair = 0;                            // in tab: per pair {LE, LF, oE, oF, junc, sLE, sLF, soE, soF, 0, 0, 0} (the s-entries: what the solve walks)
  int npw = 0;                              // positions per wave in the wave-parallel phases: wave w owns p = w + 4 s, s < npw
  int nhr = 0;                              // hub blocks of the positions with s < nhr live in registers (both layouts), the rest in LDS; = the instance's OCH
  int nlds = 0;                             // LDS block slots
  std::vector<int> gsrc, csrc, hsrc;        // [nbc] factor block ids (slab order): G_p, W_succ(p),p or -1, W_hub,p or -1
  std::vector<int> cslot, hslot;            // [nbc] LDS slot of the chain / hub block of column p, -1 = none (hub: -1 also when in registers)
  int ghub_src = -1, ghub_slot = -1;
  // flattened for the device: [0] LE [1] LF, then chainE and chainF as {position, LDS slot of W_succ(p),p} pairs (8-byte aligned),
  // then per position {gsrc, csrc, hsrc, cslot, hslot}, then the LDS fill list {src, slot, negate} x nlds
  std::vector<int> tab;
  int o_chainE = 0, o_chainF = 0, o_pos = 0, o_fill = 0;
  // Double stages of the solve (oc_add_doubles): where LDS has room, two consecutive chain stages  t_m = rhs_m - W_a t_a,  t_b = rhs_b - W_m t_m  become
  // ONE dependent stage  t_b = (rhs_b - W_m rhs_m) + (W_m W_a) t_a  with the product block kept in LDS beside the factor; the bracket and t_m are
  // independent mat-vecs that all waves share before and after the chain phase (the backward sweep likewise, with the same product block read
  // transposed).  The chains the solve walks (schainE / schainF: the positions it visits) are half as long; the factorisation is untouched.
  struct Dbl { int a, m, b, slot_a, slot_m, slot_pp, src_a, src_m; };
  std::vector<Dbl> dbl;
  std::vector<int> schainE, schainF, sslotE, sslotF;      // visited positions and the LDS slot of the block (factor or product) that leads to the next one
  int nfill = 0;                                          // LDS slots filled from the slab (the product slots behind them are computed on chip)
  int o_s = 0, o_dbl = 0, o_pp = 0;
};
inline int oc_rext(int nw, int npair = 1) { return (nw + 1 + npair) * BS; }     // behind the solve vector: the junction term, one hub partial sum per wave, a zero block, the junction terms of further pairs
constexpr int OC_CHAIN_SHORT = 17;          // the four-wave instances unroll their chain loops for 16 stages (kernel_onchip.hpp OC_MAXT); the eight-wave ones loop

// the flat table the kernels read: [0] LE [1] LF, the chains as {position, LDS slot of W_succ(p),p} pairs (8-byte aligned; what oc_ldl walks), at o_s
// {sLE, sLF} and the chains the SOLVE walks (visited positions, slot of the block -- factor or product -- that leads to the next), at o_dbl the double
// stages {a, m, b, slot of W_m,a .. , slot of W_b,m, 0, 0, 0} (8 ints each), then (o_pos) per position {gsrc, csrc, hsrc, cslot, hslot}, (o_fill) the
// LDS fill list {src, slot, negate} x nfill, and (o_pp) per double stage {slab block of W_b,m, slab block of W_m,a, LDS slot of their product}
inline void oc_build_tab(OcPlan &oc) {
  const int nbc = oc.nbc;
  if (oc.dbl.empty()) {
    oc.schainE = oc.chainE; oc.schainF = oc.chainF;
    oc.sslotE.clear(); oc.sslotF.clear();
    for (int p : oc.chainE) oc.sslotE.push_back(std::max(oc.cslot[p], 0));
    for (int p : oc.chainF) oc.sslotF.push_back(std::max(oc.cslot[p], 0));
  }
  oc.tab.clear();
  oc.tab.push_back((int)oc.chainE.size()); oc.tab.push_back((int)oc.chainF.size());
  // (a chain end without a block below it carries slot 0: the kernel's prefetch reads one stage past the end and drops the result)
  oc.o_chainE = (int)oc.tab.size(); for (int p : oc.chainE) { oc.tab.push_back(p); oc.tab.push_back(std::max(oc.cslot[p], 0)); }
  oc.o_chainF = (int)oc.tab.size(); for (int p : oc.chainF) { oc.tab.push_back(p); oc.tab.push_back(std::max(oc.cslot[p], 0)); }
  oc.o_s = (int)oc.tab.size();
  oc.tab.push_back((int)oc.schainE.size()); oc.tab.push_back((int)oc.schainF.size());
  for (size_t k = 0; k < oc.schainE.size(); k++) { oc.tab.push_back(oc.schainE[k]); oc.tab.push_back(oc.sslotE[k]); }
  for (size_t k = 0; k < oc.schainF.size(); k++) { oc.tab.push_back(oc.schainF[k]); oc.tab.push_back(oc.sslotF[k]); }
  oc.o_dbl = (int)oc.tab.size();
  for (const OcPlan::Dbl &d : oc.dbl) { int r[8] = {d.a, d.m, d.b, d.slot_a, d.slot_m, 0, 0, 0}; oc.tab.insert(oc.tab.end(), r, r + 8); }
  if (oc.pairs.empty()) { OcPlan::Pair p0; p0.E = oc.chainE; p0.F = oc.chainF; p0.junc = oc.junc; oc.pairs.push_back(p0); }
  {
    // the lists of the further pairs (one list serves factorisation and solve: no double stages there), then the records
    for (size_t i = 1; i < oc.pairs.size(); i++) {
      OcPlan::Pair &pr = oc.pairs[i];
      pr.oE = (int)oc.tab.size(); for (int p : pr.E) { oc.tab.push_back(p); oc.tab.push_back(std::max(oc.cslot[p], 0)); }
      pr.oF = (int)oc.tab.size(); for (int p : pr.F) { oc.tab.push_back(p); oc.tab.push_back(std::max(oc.cslot[p], 0)); }
    }
    oc.o_pair = (int)oc.tab.size();
    for (size_t i = 0; i < oc.pairs.size(); i++) {
      const OcPlan::Pair &pr = oc.pairs[i];
      int r[12] = {(int)pr.E.size(), (int)pr.F.size(), pr.oE, pr.oF, pr.junc, (int)pr.E.size(), (int)pr.F.size(), pr.oE, pr.oF, 0, 0, 0};
      if (i == 0) { r[2] = oc.o_chainE; r[3] = oc.o_chainF; r[5] = (int)oc.schainE.size(); r[6] = (int)oc.schainF.size(); r[7] = oc.o_s + 2; r[8] = oc.o_s + 2 + 2 * (int)oc.schainE.size(); }
      oc.tab.insert(oc.tab.end(), r, r + 12);
    }
  }
  oc.o_pos = (int)oc.tab.size();
  for (int p = 0; p < nbc; p++) { int r[5] = {oc.gsrc[p], oc.csrc[p], oc.hsrc[p], oc.cslot[p], oc.hslot[p]}; oc.tab.insert(oc.tab.end(), r, r + 5); }
  oc.o_fill = (int)oc.tab.size();
  for (int p = 0; p < nbc; p++) if (oc.cslot[p] >= 0) { int r[3] = {oc.csrc[p], oc.cslot[p], 1}; oc.tab.insert(oc.tab.end(), r, r + 3); }
  if (oc.has_hub) { int r[3] = {oc.ghub_src, oc.ghub_slot, 0}; oc.tab.insert(oc.tab.end(), r, r + 3); }
  for (int p = 0; p < nbc; p++) if (oc.hslot[p] >= 0) { int r[3] = {oc.hsrc[p], oc.hslot[p], 1}; oc.tab.insert(oc.tab.end(), r, r + 3); }
  oc.o_pp = (int)oc.tab.size();
  for (const OcPlan::Dbl &d : oc.dbl) { int r[3] = {d.src_m, d.src_a, d.slot_pp}; oc.tab.insert(oc.tab.end(), r, r + 3); }
}
// up to max_doubles double stages, handed out to whichever chain still has more dependent stages to walk; along a chain they are taken from its
// head, pair after pair, and never jump over a chain's last position (the junction term and the junction itself need it as it is)
inline void oc_add_doubles(OcPlan &oc, int max_doubles) {
  oc.dbl.clear(); oc.nlds = oc.nfill;
  if (!oc.ok || max_doubles <= 0) { oc_build_tab(oc); return; }
  int nE = 0, nF = 0;                                   // doubles taken per chain
  const int LE = (int)oc.chainE.size(), LF = (int)oc.chainF.size();
  auto can = [](int L, int nd) { return 2 * (nd + 1) <= L - 1; };     // the (nd + 1)-th double ends at index 2 (nd + 1) <= L - 1
  for (int k = 0; k < max_doubles; k++) {
    const int remE = LE - nE, remF = LF - nF;           // positions still visited
    const bool cE = can(LE, nE), cF = can(LF, nF);
    if (!cE && !cF) break;
    if (cE && (!cF || remE >= remF)) nE++; else nF++;
  }
  auto build = [&](const std::vector<int> &ch, int nd, std::vector<int> &vis, std::vector<int> &slot) {
    vis.clear(); slot.clear();
    size_t k = 0;
    for (int d = 0; d < nd; d++, k += 2) {
      OcPlan::Dbl e; e.a = ch[k]; e.m = ch[k + 1]; e.b = ch[k + 2]; e.slot_a = oc.cslot[e.a]; e.slot_m = oc.cslot[e.m]; e.src_a = oc.csrc[e.a]; e.src_m = oc.csrc[e.m];
      e.slot_pp = oc.nlds++;
      oc.dbl.push_back(e);
      vis.push_back(e.a); slot.push_back(e.slot_pp);
    }
    for (; k < ch.size(); k++) { vis.push_back(ch[k]); slot.push_back(std::max(oc.cslot[ch[k]], 0)); }
  };
  build(oc.chainE, nE, oc.schainE, oc.sslotE);
  build(oc.chainF, nF, oc.schainF, oc.sslotF);
  oc_build_tab(oc);
}
inline OcPlan build_oc_plan(const Plan &pl, int nw, int max_lds_blocks, int max_npw, int max_nhr, int max_chain = OC_CHAIN_SHORT) {
  OcPlan oc;
  const int nb = pl.nb;
  if (nb < 2 || (nw != 4 && nw != 8)) return oc;
  std::map<std::pair<int, int>, int> bid;
  for (int b = 0; b < pl.nblk; b++) bid[{pl.blkI[b], pl.blkJ[b]}] = b;
  const int H = nb - 1;
  // is the last block an arrow head?  It is when some column has it beside a chain successor; a last block that only its neighbour
  // couples to is the end of the chain
  int hub_with_chain = 0;
  for (int p = 0; p < H; p++) {
    const std::vector<int> &rows = pl.colrows[p];
    const bool hh = std::find(rows.begin(), rows.end(), H) != rows.end();
    if ((int)rows.size() - (hh ? 1 : 0) > 1) return oc;                 // more than one chain successor: not this topology
    if (hh && rows.size() > 1) hub_with_chain++;
  }
  oc.has_hub = hub_with_chain > 0;          // (a last block that is only ever a column's single neighbour is a chain end -- in the twisted order of both chains)
  oc.nbc = oc.has_hub ? nb - 1 : nb;
  const int nbc = oc.nbc;
  std::vector<int> succ(nbc, -1), npred(nbc, 0);
  for (int p = 0; p < nbc; p++)
    for (int I : pl.colrows[p]) { if (oc.has_hub && I == H) continue; succ[p] = I; npred[I]++; }
  std::vector<int> heads;
  for (int p = 0; p < nbc; p++) { if (npred[p] == 0) heads.push_back(p); if (npred[p] > 2) return oc; }
  if (heads.empty() || (int)heads.size() > nw) return oc;
  auto walk = [&](int h) { std::vector<int> c; for (int p = h; p >= 0; p = succ[p]) c.push_back(p); return c; };
  // chains that end in the same element form a twisted pair; a chain alone is a pair without junction.  One pair: the plain or twisted order; several:
  // the dissected order, whose segments meet only in the hub (without a hub they would not be one matrix)
  std::map<int, std::vector<std::vector<int>>> by_end;
  for (int h : heads) { std::vector<int> c = walk(h); by_end[c.back()].push_back(c); }
  if (by_end.size() > 1 && (!oc.has_hub || 2 * (int)by_end.size() > nw)) return oc;
  int total = 0;
  for (auto &kv : by_end) {
    std::vector<std::vector<int>> &cs = kv.second;
    if (cs.size() > 2) return oc;
    OcPlan::Pair pr;
    if (cs.size() == 2) {
      std::vector<int> c0 = cs[0], c1 = cs[1];
      for (size_t i = 0; i + 1 < c0.size(); i++) if (std::find(c1.begin(), c1.end(), c0[i]) != c1.end()) return oc;      // (they share exactly their last element)
      // E = the chain that hands its end over (drops the shared element), F keeps it; give F the longer one so both waves do the same work
      if (c0.size() > c1.size()) std::swap(c0, c1);
      c0.pop_back();
      pr.E = c0; pr.F = c1; pr.junc = 1;
      if (pr.E.empty()) { pr.junc = 0; pr.E = c1; pr.F.clear(); }
    } else pr.E = cs[0];
    if ((int)pr.E.size() > max_chain || (int)pr.F.size() > max_chain) return oc;        // (the four-wave kernels' chain loops are unrolled for 16 stages)
    total += (int)(pr.E.size() + pr.F.size());
    oc.pairs.push_back(pr);
  }
  std::sort(oc.pairs.begin(), oc.pairs.end(), [](const OcPlan::Pair &a, const OcPlan::Pair &b) { return a.E[0] < b.E[0]; });
  oc.chainE = oc.pairs[0].E; oc.chainF = oc.pairs[0].F; oc.junc = oc.pairs[0].junc;
  if (total != nbc) return oc;
  oc.npw = (nbc + nw - 1) / nw;
  if (oc.npw > max_npw) return oc;
  oc.gsrc.assign(nbc, -1); oc.csrc.assign(nbc, -1); oc.hsrc.assign(nbc, -1); oc.cslot.assign(nbc, -1); oc.hslot.assign(nbc, -1);
  int slots = 0;
  for (int p = 0; p < nbc; p++) {
    oc.gsrc[p] = bid.at({p, p});
    if (succ[p] >= 0) { oc.csrc[p] = bid.at({succ[p], p}); oc.cslot[p] = slots++; }
    if (oc.has_hub) { auto it = bid.find({H, p}); if (it != bid.end()) oc.hsrc[p] = it->second; }
  }
  if (oc.has_hub) { oc.ghub_src = bid.at({H, H}); oc.ghub_slot = slots++; }
  // hub blocks: the first nhr slots of every wave in registers -- exactly the kernel instance's count, the loops over a wave's
  // positions are straight-line code -- the rest in LDS.  The instances in use: patterns with an arrow head to which every column
  // couples (max_nhr register-resident hub blocks per wave), and patterns without one (the reduced form, no parameters).
  oc.nhr = 0;
  if (oc.has_hub) {
    for (int p = 0; p < nbc; p++) if (oc.hsrc[p] < 0) return oc;
    oc.nhr = max_nhr;
    for (int p = 0; p < nbc; p++) if (p / nw >= oc.nhr) oc.hslot[p] = slots++;
  }
  if (slots > max_lds_blocks) return oc;
  oc.nlds = slots;
  oc.nfill = oc.nlds;
  oc.ok = true;
  oc_build_tab(oc);
  return oc;
}
// LDS of the on-chip variant: block slots (the factorisation's temp tiles and the staged ELL values alias them), x, q, r (+ the
// junction / hub partial sums) [npad], z, y, w [mpad], reduction scratch, the table
// Which chunks of A' the iteration may compute late, during the chain phase of the solve (kernel_onchip.hpp oc_solve): with more than
// four chunks some wave would sweep two before the solve starts.  A chunk past the fourth qualifies as `free` when all its rows belong to
// the hub (read only behind the barrier that ends the chain phase), and as `poll` when all its rows are chain positions that neither chain
// fetches before the top of trip poll_trip: by then a chain has fetched its entries 0 .. 2 poll_trip + 1.
inline void oc_late_chunks(const Plan &pl, const OcPlan &oc, int nw, int poll_trip, int *at_poll, int *at_free) {
  *at_poll = -1; *at_free = -1;
  const int nc = pl.At.nchunks;
  if (nc <= nw || nc > nw + 2) return;
  int early = -1;                                    // highest position fetched before the poll
  for (const std::vector<int> *ch : {&oc.chainE, &oc.chainF})
    for (size_t i = 0; i < ch->size() && (int)i <= 2 * poll_trip + 1; i++) early = std::max(early, (*ch)[i]);
  int poll = -1, fre = -1;
  for (int c = nw; c < nc; c++) {
    const int t0 = c * WAVE, t1 = std::min(pl.npad, t0 + WAVE);
    const int p0 = t0 / BS, p1 = (t1 - 1) / BS;
    if (p0 >= oc.nbc) { if (fre >= 0) return; fre = c; }
    else if (p1 < oc.nbc && p0 > early) { if (poll >= 0) return; poll = c; }
    else return;
  }
  *at_poll = poll; *at_free = fre;
}
inline std::vector<int> oc_asm_records(const Plan &pl) {
  std::vector<int> r((size_t)8 * pl.nblk, std::max(pl.nT, 1));      // unused terms: the zero tile behind the T tiles (ws_layout)
  for (int b = 0; b < pl.nblk; b++) {
    const int p0 = pl.asm_ptr[b], n = pl.asm_ptr[b + 1] - p0;
    r[8 * b] = n; r[8 * b + 1] = pl.blk_diag[b];
    for (int t = 0; t < std::min(n, 3); t++) { r[8 * b + 2 + 2 * t] = pl.asm_a[p0 + t]; r[8 * b + 3 + 2 * t] = pl.asm_b[p0 + t]; }
  }
  return r;
}
inline long oc_stage_doubles(const OcPlan &oc, const ResPlan &rp, const Plan &pl) {
  // block slots / temp tiles, or the factorisation's scratch: 8 hand-over blocks + the assembly records (4 doubles per block)
  return std::max((long)std::max(oc.nlds, rp.ntemp) * BLK, 8L * BLK + ((4L * pl.nblk + 15) / 16) * 16);   // (8: kernel_onchip.hpp OC_LDL_SCR)
}
inline long lds_bytes_oc(const Plan &pl, const ResPlan &rp, const OcPlan &oc, bool zy_global = false, const TilePlan *tp = nullptr) {
  // the chain tables live in LDS (the per-position and fill tables are read from global memory), and so do the chunk offsets of A, A', P
  long tab_words = ((long)oc.o_pos + 1) / 2 + 4 + ((long)pl.A.nchunks + pl.At.nchunks + pl.P.nchunks + 3 + 1 + 1) / 2;     // (+ the ticket of the late right-hand side rows)
  if (tp && tp->on) tab_words += ((long)tp->Ar.nchunks + tp->Atr.nchunks + 2 + 1) / 2;      // chunk offsets of the two remainder layouts
  return (oc_stage_doubles(oc, rp, pl) + 3L * pl.npad + oc_rext(rp.nw, std::max<int>(1, (int)oc.pairs.size())) + (zy_global ? 1L : 3L) * pl.mpad + 16L * rp.nw + 16 + 16L * rp.nw + tab_words) * 8L;
}

inline long lds_bytes(const Plan &pl) {
  // x, q, r [npad]; z, y, w [mpad]; two padded 16x17 scratch tiles (aliased onto r when npad >= 544); 64 spare doubles
  const long tiles = pl.npad >= 2 * BS * (BS + 1) ? 0 : 2L * BS * (BS + 1);
  return (3L * pl.npad + 3L * pl.mpad + tiles + 64) * 8L;
}

}  // namespace mpcqp
