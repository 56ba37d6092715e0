// mpcqp.hip -- MI355X (gfx950) batched OSQP-style ADMM: device kernels + the C ABI of include/mpcqp.h.
//
// Replaces, for a batch of QPs sharing one sparsity, what the reference does per QP on the host through
// CuCaQP::setSystem -> initSolver -> solve -> getSolution (reference src/sqp_solver/CuCaQP.cpp:271-288,
// 183-224), i.e. OSQP's osqp_setup + osqp_solve (external to the reference, see oracle/osqp_oracle.h).
//
// One launch does the whole sequence for every QP of the batch (grid = batch, one workgroup per QP).  Kernel families
// (mpcqp_create picks one per sparsity and batch size from measured rules):
//   * mpcqp_res_kernel<NW, MINW, GB, REUSE, ZYG> -- NW wavefronts per QP (1 or 4), W-fused block LDL' of
//     M = P + sigma I + A' diag(rho) A in 16x16 blocks, solve driven by a host-built schedule of arithmetic-progression
//     segments (plan.hpp).  GB = false: the factor lives in LDS (small / mid-size problems, and any batch that fits one
//     resident round); GB = true: the factor stays in the per-QP HBM slab and LDS holds only vectors, temp tiles and the
//     schedule, so that 3-4 workgroups share a CU (the default for the 12-state quadrotor sizes; HBM-roofline-bound).
//     MINW selects the register budget (128 / 168 / 256 VGPRs), REUSE the kept-workspace entry (mpcqp_update_vectors),
//     ZYG keeps z and y in the slab too (long horizons).
//     OCG / OCH > 0: the on-chip mode of the global-block kernel (kernel_onchip.hpp) -- after each factorisation the factor is
//     brought on chip (chain blocks in LDS, inverse diagonal blocks and some hub blocks in registers; two workgroups per CU) and the
//     triangular solves run on the matrix cores, register to register along each chain.  The default for the 12-state quadrotor, N = 20.
//   * mpcqp_admm_kernel<PD> -- the first-generation streaming kernel, one QP per wavefront, block Cholesky streamed from
//     the slab; fallback when even the vectors exceed LDS, and a cross-check in the variant tests.
// Common to all: ADMM iterates in LDS; scaled A in two ELL orientations and scaled P in the slab, streamed with coalesced
// 512 B wave loads; the linear solve is a stream of 16x16 block mat-vecs (4 lanes per row, quad DPP reduction); the
// factorisation's block products are dense 16x16x16 GEMMs on the matrix cores (v_mfma_f64_16x16x4_f64); box projection,
// dual update and residual norms fused into the ELL sweeps; termination, infeasibility certificates and adaptive-rho
// re-factorisation in-kernel; workgroup-uniform state in scalar registers (uni()).
// Numerics are fp64 throughout and follow oracle/osqp_oracle.c step by step (same scaling rule, rho rule,
// termination / infeasibility tests and deterministic adaptive-rho schedule).
#include <map>
#include <mutex>
#include "kernels_all.hpp"
#include "kernels_util.hpp"
#include "reduced.hpp"

// ------------------------------------------------------------------------------------------ host side
static thread_local std::string g_last_error;
static thread_local const char *g_force_variant = nullptr;      // mpcqp_create_tuned: the family the next mpcqp_create on this thread must take ("" = the rule's choice)
static const char *variant_request() { return g_force_variant ? (g_force_variant[0] ? g_force_variant : nullptr) : getenv("MPCQP_VARIANT"); }
static int fail(int code, const std::string &msg) { g_last_error = msg; return code; }
int mpcqp_set_error(int code, const std::string &msg) { return fail(code, msg); }
int mpcqp_pick_device(int requested, int *device) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(MPCQP_ERR_NO_GPU, "hipGetDeviceCount found no device");
  int dev = requested;
  if (dev < 0) { if (hipGetDevice(&dev) != hipSuccess) dev = 0; }
  if (dev >= ndev) return fail(MPCQP_ERR_ARG, "device ordinal out of range");
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return fail(MPCQP_ERR_HIP, "hipGetDeviceProperties failed");
  if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos)
    return fail(MPCQP_ERR_NO_GPU, std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only");
  *device = dev;
  return MPCQP_OK;
}
#define HIPCHK(expr)                                                                              \
  do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(MPCQP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)

struct mpcqp_handle {
  int n = 0, m = 0, batch = 0, device = 0;
  hipEvent_t ev_guard = nullptr;              // mpcqp_order_after_last_solve
  mpcqp_settings st;
  Plan plan; WsLayout wl; long lds = 0;
  int variant = 0;              // 0 = streaming (1 wave / QP), NW > 0 = LDS-resident factor with NW waves / QP
  bool wide = false;            // resident kernel instance that may use the whole register file (one QP per CU)
  bool gblocks = false;         // multi-wave LDL' kernel with the factor blocks streamed from the HBM slab
  bool zyg = false;             // ... with z, y in the slab instead of LDS (lifts workgroups per CU for long horizons)
  bool occ3 = false;            // ... its 168-VGPR instance (exactly 3 workgroups per CU fit in LDS), 8 blocks in flight
  bool res1x = false;           // one-wave kernel, 128-VGPR instance: more than eight QPs per CU when the LDS footprint allows (double integrator N=10: 12.1 -> 13.8 M QP/s)
  int res3 = 0;                 // LDS-resident 4-wave kernel: 3 or 4 workgroups per CU (168- / 128-VGPR instances) when the LDS footprint allows, else 0
  bool stream_pd8 = false;      // streaming kernel instance (read from the environment once, at create)
  bool occ4 = false;            // ... its 128-VGPR instance (>= 3 workgroups per CU fit in LDS)
  bool oc = false;              // on-chip mode of the global-block kernel (kernel_onchip.hpp): two workgroups per CU, factor in LDS + registers
  int oc8 = 0;                  // ... its eight-wave instances for long chains (one workgroup per CU): 1 = <NG 4, NH 4>, 2 = <NG 7, NH 7>
  int resume_rounds = 1;        // two-kernel form: {re-factorisation, iteration} pairs queued behind a solve before the last pair (MPCQP_RESUME_ROUNDS)
  // two-kernel form: the set-up kernel's own launch shape.  Nothing of the factor is resident while it runs, so it does not need the iteration kernel's LDS
  // (the block slots) or its eight waves: four-wave workgroups (oc_ldl's chain waves and helpers are four in any case) with an LDS request of their own let
  // two or three QPs share a CU where the iteration kernel has one
  int setup_nw = 0; long lds_setup = 0; DevRes dres_setup; DevOc doc_setup;
  int *qctr = nullptr; int qslots = 0;        // two-kernel on-chip mode: ticket counters of the resident iteration workgroups (16 per solve in flight), and how many workgroups the GPU holds at once
  bool split = false;           // ... as two kernels, set-up and iteration (kernel_oc_split.hpp): the default; MPCQP_OC_MONO=1 and the tile experiment keep the single kernel
  OcPlan ocplan; DevOc doc;
  TilePlan tplan; bool tiles = false;   // on-chip kernels: dense tiles of A for the two sweeps of the iteration (plan.hpp build_tile_plan)
  bool vtiles = false;                  // ... in the two-kernel form, on the vector ALUs (kernel_oc_split.hpp; MPCQP_VTILES=1)
  ResPlan rplan; DevRes dres;
  DevPlan dp; DevIO io;
  std::vector<void *> dev_allocs;
  double *ws = nullptr;
  double *dP = nullptr, *dq = nullptr, *dA = nullptr, *dl = nullptr, *du = nullptr;  // owned copies (host-memory updates)
  double *dx0 = nullptr, *dy0 = nullptr, *drho0 = nullptr;
  bool keep = false, have_factor = false, reuse_next = false;
  int *order[2] = {nullptr, nullptr}; int order_cur = -1; bool lpt = true; hipEvent_t ev_order = nullptr;   // dispatch hint, double-buffered
  static constexpr int NPIPE = 8;
  hipStream_t pipe[NPIPE] = {};               // mpcqp_solve_host: compute streams, one per slice in flight
  hipStream_t pipe_copy = nullptr;            // ... and the one stream all host-to-device copies queue on, in slice order
  std::vector<hipEvent_t> pipe_ev;            // slice k's inputs have landed
  double *ox = nullptr, *oy = nullptr, *oz = nullptr, *oinfo = nullptr, *ocs = nullptr; int *ostatus = nullptr, *oiters = nullptr;
  long long *odbg = nullptr;
  bool have_data = false, solved = false;
  hipStream_t last_stream = nullptr;
  hipEvent_t ev_mid = nullptr;                // two-kernel on-chip mode: between the set-up and the iteration kernel (mpcqp_last_phase_ms)
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev0r = nullptr;   // (ev0r: reduced handles, ordering of the presolve behind a solve on another stream)
  // reduced form (mpcqp_create_reduced): this handle keeps the caller's dimensions, `inner` solves the QP without the eliminated variables
  mpcqp_handle *inner = nullptr; RedMaps red; DevRed dred; double *rx0 = nullptr, *ry0 = nullptr;
};

int mpcqp_order_after_last_solve(mpcqp_handle *h, hipStream_t s) {
  if (!h) return fail(MPCQP_ERR_ARG, "null handle");
  if (!h->solved || h->last_stream == s) return MPCQP_OK;
  if (!h->ev_guard) HIPCHK(hipEventCreateWithFlags(&h->ev_guard, hipEventDisableTiming));
  HIPCHK(hipEventRecord(h->ev_guard, h->last_stream));
  HIPCHK(hipStreamWaitEvent(s, h->ev_guard, 0));
  return MPCQP_OK;
}

template <class T>
static int upload(mpcqp_handle *h, const std::vector<T> &v, const T **out) {
  void *d = nullptr;
  size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
  HIPCHK(hipMalloc(&d, bytes));
  h->dev_allocs.push_back(d);
  if (!v.empty()) HIPCHK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  *out = (const T *)d;
  return MPCQP_OK;
}
static int upload_ell(mpcqp_handle *h, const Ell &e, DevEll *d) {
  d->nchunks = e.nchunks; d->entries = e.entries();
  int rc;
  if ((rc = upload(h, e.chunk_off, &d->chunk_off))) return rc;
  if ((rc = upload(h, e.idx, &d->idx))) return rc;
  if ((rc = upload(h, e.src, &d->src))) return rc;
  if ((rc = upload(h, e.flag, &d->flag))) return rc;
  return MPCQP_OK;
}
template <class T>
static int dalloc(mpcqp_handle *h, T **p, size_t count) {
  void *d = nullptr;
  HIPCHK(hipMalloc(&d, std::max<size_t>(count, 1) * sizeof(T)));
  h->dev_allocs.push_back(d);
  *p = (T *)d;
  return MPCQP_OK;
}

constexpr long OC_LDS_MAX = 80 * 1024;      // four-wave on-chip instances: two workgroups per CU
constexpr long OC8_LDS_MAX = 160 * 1024;    // eight-wave ones: one (kernel_table.hpp OC8_INST)
constexpr int OC8_MAX_CHAIN = 64;           // (oc_ldl keeps the chain's block ids one per lane)

// the kernel instance a handle runs (instantiated in the k_*.hip units, kernel_table.hpp): waves per QP, register budget, factor location, and --
// as its own instance so that the full-setup kernels carry no code for it -- the kept-workspace entry (mpcqp_update_vectors)
static const void *res_kernel_of(const mpcqp_handle *h, bool reuse) {
  auto lds = [&](int nw, int minw) { return reuse ? mpcqp_kernel_res_lds_r1(nw, minw) : mpcqp_kernel_res_lds_r0(nw, minw); };
  auto gb = [&](int nw, int minw, bool zyg) { return reuse ? mpcqp_kernel_res_gb_r1(nw, minw, zyg) : mpcqp_kernel_res_gb_r0(nw, minw, zyg); };
  auto mono = [&](int nw, int ng, int nh) { return reuse ? mpcqp_kernel_oc_mono_r1(nw, ng, nh, h->tiles) : mpcqp_kernel_oc_mono_r0(nw, ng, nh, h->tiles); };
  if (h->oc && h->oc8) return mono(8, OC8_INST[h->oc8 - 1].ng, OC8_INST[h->oc8 - 1].nh);      // (nullptr without an arrow head: such handles run the two-kernel form)
  if (h->oc) return mono(4, OC_NG, h->ocplan.has_hub ? OC_NH : 0);
  if (h->gblocks && h->variant == 2) return gb(2, 3, false);
  if (h->gblocks && h->zyg) return gb(4, h->occ3 ? 3 : 2, true);
  if (h->gblocks && h->occ3) return gb(4, 3, false);
  if (h->gblocks) return gb(4, h->occ4 ? 4 : 2, false);
  if (h->variant == 1) return lds(1, h->res1x ? 4 : 2);
  if (h->variant == 8) return lds(8, 2);
  if (h->variant == 2) return lds(2, 3);
  if (h->res3 == 4) return lds(4, 4);
  if (h->res3 == 3) return lds(4, 3);
  return lds(4, h->wide ? 1 : 2);
}
// the two kernels of the on-chip mode (kernel_oc_split.hpp): CuCaQP::initSolver and CuCaQP::solve
static const void *oc_setup_of(const mpcqp_handle *h, bool reuse) { return mpcqp_kernel_oc_setup(h->setup_nw, h->ocplan.has_hub != 0, reuse); }
static const void *oc_admm_of(const mpcqp_handle *h, bool rf = false) {
  const int nw = h->oc8 ? 8 : 4, ng = h->oc8 ? OC8_INST[h->oc8 - 1].ng : OC_NG, nh = !h->ocplan.has_hub ? 0 : h->oc8 ? OC8_INST[h->oc8 - 1].nh : OC_NH;
  if (!h->oc8 && h->ocplan.pairs.size() > 1) return mpcqp_kernel_oc_admm_p4(rf ? 1 : 0);      // (four waves, two twisted pairs: its own instances)
  if (h->vtiles && !rf) return mpcqp_kernel_oc_admm_tl(nw, ng, nh);      // (the last launch of a solve, rf, runs the ELL sweeps: the set-up writes both forms)
  return rf ? mpcqp_kernel_oc_admm_rf(nw, ng, nh) : mpcqp_kernel_oc_admm(nw, ng, nh);
}
// One solve of the two-kernel on-chip mode on stream s: set-up, iteration; then, for instances whose adaptive-rho step asked for a new factor
// (kernel_oc_split.hpp: they leave the iteration kernel marked OC_PENDING), `resume_rounds` pairs of {re-factorisation, iteration} in which every other
// workgroup returns at once, and a last pair whose iteration kernel re-factorises in place, so that any number of rho updates is served.
// The idle waves' L2 touch during the backward chains (kernel_oc_split.hpp idle_touch): off for the eight-wave instances (one QP's sweeps do not fit a CU's share of
// the L2; MPCQP_TOUCH8 turns it on) and, since round 4, for the four-wave instances of the two-kernel form as well -- with the sweeps' gathers batched and three set-up
// workgroups per CU the touch only competes with them (quadrotor N=20 x 8192 iteration kernel 5.25 -> 5.15 ms without it, N=15 3.98 -> 3.82, N=10 3.02 -> 2.94;
// cart-pole N=50 and double integrator N=60 unchanged; MPCQP_TOUCH4 turns it on).  The single-kernel form keeps it.
static int no_touch_of(const mpcqp_handle *h) {
  if (getenv("MPCQP_NO_TOUCH")) return 1;
  if (h->oc8) return getenv("MPCQP_TOUCH8") ? 0 : 1;
  if (h->split) return getenv("MPCQP_TOUCH4") ? 0 : 1;
  return 0;
}
static int launch_oc_split(mpcqp_handle *h, DevIO &io, int count, bool reuse, hipStream_t s, hipEvent_t after_setup, int qslot) {
  const dim3 grid(count), block(h->variant * WAVE), block_s(h->setup_nw * WAVE);
  // the iteration kernel as resident workgroups that draw instances from one counter (kernel_oc_split.hpp) when the batch is more than the GPU holds at once
  const bool queued = h->oc8 != 0;      // (the eight-wave instances are compiled as resident workgroups; the four-wave ones are not)
  int *qc = queued ? h->qctr + 16 * qslot : nullptr;
  if (queued) HIPCHK(hipMemsetAsync(qc, 0, 16 * sizeof(int), s));
  const dim3 grid_it(queued ? std::min(h->qslots, count) : count);
  int nlaunch = 0;
  DevIO ioq[10];
  auto io_of = [&]() -> void * { DevIO &q = ioq[nlaunch]; q = io; q.queue = queued ? qc + nlaunch : nullptr; q.count = count; return (void *)&ioq[nlaunch++]; };
  DevOc doc0 = h->doc; doc0.resume = 0;
  DevOc docr = doc0; docr.resume = 1;
  DevOc docs0 = h->doc_setup; docs0.resume = 0;
  DevOc docsr = docs0; docsr.resume = 1;
  void *args[] = {(void *)&h->dp, (void *)&h->dres, (void *)&h->st, (void *)&io, (void *)&doc0};
  void *argr[] = {(void *)&h->dp, (void *)&h->dres, (void *)&h->st, (void *)&io, (void *)&docr};
  void *sargs[] = {(void *)&h->dp, (void *)&h->dres_setup, (void *)&h->st, (void *)&io, (void *)&docs0};
  void *sargr[] = {(void *)&h->dp, (void *)&h->dres_setup, (void *)&h->st, (void *)&io, (void *)&docsr};
  HIPCHK(hipLaunchKernel(oc_setup_of(h, reuse), grid, block_s, sargs, (size_t)h->lds_setup, s));
  if (after_setup) HIPCHK(hipEventRecord(after_setup, s));
  const bool rho_updates = h->st.adaptive_rho != 0;
  args[3] = io_of();
  HIPCHK(hipLaunchKernel(oc_admm_of(h, !rho_updates), grid_it, block, args, (size_t)h->lds, s));     // (without adaptive rho nothing ever leaves: either instance serves)
  if (!rho_updates) return MPCQP_OK;
  for (int r = 0; r <= h->resume_rounds; r++) {
    HIPCHK(hipLaunchKernel(oc_setup_of(h, false), grid, block_s, sargr, (size_t)h->lds_setup, s));
    argr[3] = io_of();
    HIPCHK(hipLaunchKernel(oc_admm_of(h, r == h->resume_rounds), grid_it, block, argr, (size_t)h->lds, s));
  }
  return MPCQP_OK;
}
extern "C" {

void mpcqp_default_settings(mpcqp_settings *s) {
  if (!s) return;
  s->rho = 0.1; s->sigma = 1e-6; s->alpha = 1.6; s->eps_abs = 1e-3; s->eps_rel = 1e-3;
  s->eps_prim_inf = 1e-4; s->eps_dual_inf = 1e-4; s->adaptive_rho_tolerance = 5.0;
  s->max_iter = 10000; s->check_termination = 25; s->scaling = 10; s->adaptive_rho = 1;
  s->adaptive_rho_interval = 0; s->scaled_termination = 0; s->warm_start = 0; s->device = -1;
}

const char *mpcqp_strerror(int code) {
  static thread_local std::string buf;
  const char *base = "unknown error";
  switch (code) {
    case MPCQP_OK: base = "ok"; break;
    case MPCQP_ERR_ARG: base = "invalid argument"; break;
    case MPCQP_ERR_HIP: base = "HIP runtime error"; break;
    case MPCQP_ERR_NO_GPU: base = "no usable gfx950 GPU (this library has no CPU fallback)"; break;
    case MPCQP_ERR_STATE: base = "call order violated"; break;
    case MPCQP_ERR_LIMIT: base = "problem exceeds on-chip budget"; break;
  }
  buf = base;
  if (code != MPCQP_OK && !g_last_error.empty()) buf += ": " + g_last_error;
  return buf.c_str();
}

int mpcqp_create(int n, int m, int batch, const int *Pp, const int *Pi, const int *Ap, const int *Ai,
                 const mpcqp_settings *settings, mpcqp_handle **out) {
  if (!out) return fail(MPCQP_ERR_ARG, "out is null");
  *out = nullptr;
  if (n <= 0 || m < 0 || batch <= 0 || !Pp || !Pi || !Ap || !Ai) return fail(MPCQP_ERR_ARG, "Invalid dimensions.");
  if (!g_force_variant && !getenv("MPCQP_VARIANT") && getenv("MPCQP_AUTOTUNE") && getenv("MPCQP_AUTOTUNE")[0] == '1')
    return mpcqp_create_tuned(n, m, batch, Pp, Pi, Ap, Ai, settings, out);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(MPCQP_ERR_NO_GPU, "hipGetDeviceCount found no device");
  mpcqp_handle *h = new mpcqp_handle();
  if (settings) h->st = *settings; else mpcqp_default_settings(&h->st);
  int dev = h->st.device;
  if (dev < 0) { if (hipGetDevice(&dev) != hipSuccess) dev = 0; }
  if (dev >= ndev) { delete h; return fail(MPCQP_ERR_ARG, "device ordinal out of range"); }
  h->device = dev;
  auto bail = [&](int rc) { mpcqp_destroy(h); return rc; };
  if (hipSetDevice(dev) != hipSuccess) return bail(fail(MPCQP_ERR_HIP, "hipSetDevice failed"));
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return bail(fail(MPCQP_ERR_HIP, "hipGetDeviceProperties failed"));
  if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos)
    return bail(fail(MPCQP_ERR_NO_GPU, std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only"));
  h->n = n; h->m = m; h->batch = batch;
  h->plan = build_plan(n, m, Pp, Pi, Ap, Ai);
  if (!h->plan.error.empty()) return bail(fail(MPCQP_ERR_ARG, h->plan.error));
  const Plan &pl = h->plan;
  h->wl = ws_layout(pl);
  h->lds = lds_bytes(pl);
  // Kernel shape, from measured rules (profiles/r01_variant_grid.txt): factor in LDS with one, two or four waves per
  // QP while enough QPs fit a CU, else the factor streamed from the HBM slab.  MPCQP_VARIANT=stream|res1|res2|res4|res8|gres4 overrides.
  {
    const long LDS_MAX = 160 * 1024;
    int want = -1;
    if (const char *e = variant_request()) {
      std::string v(e);
      if (v == "stream") want = 0; else if (v == "res1") want = 1; else if (v == "res4") want = 4; else if (v == "res8") want = 8;
      else if (v == "gres4") { want = 4; h->gblocks = true; }
      else if (v == "res2") want = 2;
      else if (v == "gres2") { want = 2; h->gblocks = true; }
      else if (v == "oc4") { want = 4; h->gblocks = true; h->oc = true; }
      else if (v == "oc8") { want = 8; h->gblocks = true; h->oc = true; h->oc8 = -1; }
    }
    // candidate plans of the multi-wave kernels: ELL chunk widths padded to multiples of 4 (fewer load batches per chunk)
    // and the stage chain eliminated from both ends (two concurrent half-length chains)
    const bool twist = !getenv("MPCQP_NO_TWIST");
    Plan p1 = build_plan(n, m, Pp, Pi, Ap, Ai, -1, true);
    Plan p4 = build_plan(n, m, Pp, Pi, Ap, Ai, twist ? 2 : -1, true);
    if (!p1.error.empty() || p1.nblk > h->plan.nblk) p1 = h->plan;
    if (!p4.error.empty() || p4.nblk > h->plan.nblk) p4 = p1;
    const bool small_ok = pl.nblk < 4096 && pl.nb < 512;
    if (want < 0) {
      // measured on MI355X: one wave per QP with the factor in LDS when it is tiny; four waves per QP
      // with the factor in LDS when at least two QPs fit per CU; otherwise occupancy beats residency and the factor
      // blocks are streamed from the HBM slab by the same LDL' / segment machinery (several workgroups per CU)
      ResPlan r1 = build_res_plan(p1, 1), r4 = build_res_plan(p4, 4);
      const long l1 = lds_bytes_res(p1, r1), l4 = lds_bytes_res(p4, r4);
      // latency regime (measured, tools/graph_tick.py): when the whole batch is resident in one round of 4-wave workgroups
      // (two per CU by registers, one when the factor needs more than half the LDS), four waves per QP with the factor in
      // LDS finish a QP soonest (double integrator x256: 1.12 ms vs 1.41 ms with one wave per QP; quadrotor N=20 x256:
      // 1.21 ms vs 1.96 ms with the factor streamed from HBM)
      const long cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
      // (workgroups of the LDS-resident 4-wave kernel per CU: by LDS, and by the register budget of its instances -- 128 / 168 / 256 VGPRs)
      const long cap4 = (small_ok && l4 <= LDS_MAX) ? std::min<long>(LDS_MAX / l4, l4 <= 40 * 1024 ? 4 : l4 <= 53 * 1024 ? 3 : 2) : 0;
      // two waves per QP (168-VGPR instance: up to six per CU): the two half chains of the twisted order each get a wave and nothing idles in
      // the chain phases.  Taken where it fits more QPs per CU than the 4-wave kernel and at most one fewer than one wave per QP would:
      // double integrator N=20 (28 KiB, five per CU) 1.87 M QP/s against 1.55 M with one wave and 1.46 M with four; at 21-23 KiB +4...8 %
      // over one wave; below 15 KiB one wave per QP (11-13 per CU) wins, at 34 KiB and above the 4-wave kernel; same latency as the 4-wave
      // kernel on a batch of 64-1024
      const ResPlan r2 = build_res_plan(p4, 2);
      const long l2 = lds_bytes_res(p4, r2);
      const long q1 = small_ok && l1 <= LDS_MAX ? LDS_MAX / l1 : 0, q2 = small_ok && l2 <= 40 * 1024 ? std::min<long>(LDS_MAX / l2, 6) : 0;
      auto oc_takes = [&]() {
        if (getenv("MPCQP_NO_OC") || !small_ok) return false;
        auto ok4 = [&](const Plan &q) { const OcPlan o = build_oc_plan(q, 4, 1 << 20, OC_NG, OC_NH); return o.ok && lds_bytes_oc(q, build_res_plan(q, 4, false), o) <= OC_LDS_MAX; };
        if (ok4(p4)) return true;
        // (the padded twist -- plan.hpp ordering 3 -- where the hub variables would otherwise share a block with the last frame and spill into a second one: cart-pole N=22, 25)
        if (!twist) return false;
        const Plan q = build_plan(n, m, Pp, Pi, Ap, Ai, 3, 1);
        return q.error.empty() && ok4(q);
      };
      if (q2 > cap4 && q2 + 1 >= q1 && !getenv("MPCQP_NO_RES2")) want = 2;
      // the latency regime above; with three or four 4-wave workgroups per CU they stay ahead of one wave per QP up to about three resident
      // rounds (double integrator x2048 1.57 vs 1.81 ms, x4096 2.89 vs 2.76 ms)
      else if (cap4 > 0 && (long)batch <= cus * cap4 * (cap4 >= 3 ? 3 : 1)) {
        want = 4;
        // ... and in it the on-chip mode where it takes the pattern: up to two rounds of its two workgroups per CU it finishes a batch sooner than
        // the LDS-resident kernel at any occupancy (tools/small_batch_scan.py, x 64 ... 1024: quadrotor N = 5 / 10 / 20 0.21 / 0.43 / 0.85 ms against
        // 0.24 / 0.52 / 1.08, cart-pole N = 20 / 30 0.73 / 1.12 against 0.89 / 1.41, double integrator N = 30 / 50 1.11 / 1.64 against 1.31 / 2.18)
        if ((long)batch <= cus * 4 && oc_takes()) { h->gblocks = true; h->oc = true; }
      }
      // one wave per QP only where it puts more QPs on a CU than the 4-wave kernel has workgroups there (five against four at 28 KiB: +6 %;
      // four against four at 34-36 KiB: the 4-wave kernel is 23-31 % ahead -- double integrator N=24 / 26, cart-pole N=15)
      else if (small_ok && l1 <= 40 * 1024 && LDS_MAX / l1 > cap4) want = 1;
      else {
        // The on-chip mode (factor in LDS + registers at two workgroups per CU, solves and factorisation on the matrix cores) wherever the pattern is
        // a block chain with an arrow head that fits it AND the alternative is the LDS-resident 4-wave kernel at two workgroups per CU or a factor
        // streamed from the slab (measured, profiles/r02_final_variant_grid.txt, x 8192: quadrotor N = 6 ... 20 +8 ... +48 %, cart-pole N = 30 / 40 / 50
        // +20 / +26 / +37 %, double integrator N = 40 ... 80 +12 ... +64 %).  With three or more resident workgroups per CU the LDS-resident
        // kernels stay ahead (quadrotor N = 5 2.57 vs 2.06 M QP/s, cart-pole N = 20 818k vs 656k, double integrator N = 30 792k vs 611k).
        // LDS-resident 4-wave kernel while two fit a CU
        if (small_ok && l4 <= 80 * 1024) { want = 4; if (l4 > 53 * 1024 && oc_takes()) { h->gblocks = true; h->oc = true; } }
        // factor streamed from the slab, two waves per QP (168-VGPR instance, six workgroups per CU) while six fit the LDS: ahead of four waves x
        // four workgroups there (double integrator N=100 145k -> 156k QP/s; at 32 KiB and above four waves win)
        else if (small_ok && !getenv("MPCQP_NO_RES2") && lds_bytes_res_gb(p4, build_res_plan(p4, 2, true)) <= LDS_MAX / 6) {
          want = 2; h->gblocks = true;
          if (oc_takes()) { want = 4; h->oc = true; }
        }
        else if (small_ok && lds_bytes_res_gb(p4, build_res_plan(p4, 4, true), !getenv("MPCQP_NO_ZYG")) <= LDS_MAX) { want = 4; h->gblocks = true; h->oc = oc_takes(); }
        else want = 0;
      }
    }
    // Long chains: where the rules above arrive at a factor streamed from the slab, the eight-wave on-chip instances take the pattern if it is
    // a block chain with an arrow head of up to 56 blocks that fits one CU (MPCQP_NO_OC8 keeps the global-block kernels)
    if ((h->oc8 < 0 || (want > 0 && h->gblocks && !h->oc && !variant_request() && !getenv("MPCQP_NO_OC8") && !getenv("MPCQP_NO_OC"))) && small_ok) {
      Plan p8 = build_plan(n, m, Pp, Pi, Ap, Ai, twist ? 3 : -1, 2);
      h->oc8 = 0;
      if (twist && !getenv("MPCQP_NO_DISSECT")) {
        // the dissected order first (plan.hpp build_plan ordering 4): separators of the stage chain in the hub block where it has room, several twisted pairs of
        // short chains instead of one pair of long ones
        Plan pd = build_plan(n, m, Pp, Pi, Ap, Ai, 4, 2);
        if (pd.error.empty()) {
          const ResPlan rd = build_res_plan(pd, 8, false);
          for (int k = 0; k < 2 && !h->oc8; k++) {
            const OcPlan o = build_oc_plan(pd, 8, 1 << 20, OC8_INST[k].ng, OC8_INST[k].nh, OC8_MAX_CHAIN);
            if (o.ok && o.has_hub && o.pairs.size() > 1 && lds_bytes_oc(pd, rd, o, OC8_INST[k].zyg) <= OC8_LDS_MAX) {
              h->ocplan = o; h->oc8 = k + 1; h->oc = true; h->gblocks = true; h->zyg = OC8_INST[k].zyg; want = 8; p4 = pd;
            }
          }
        }
      }
      if (!h->oc8 && p8.error.empty()) {
        const ResPlan r8 = build_res_plan(p8, 8, false);
        for (int k = 0; k < 2 && !h->oc8; k++) {
          const OcPlan o = build_oc_plan(p8, 8, 1 << 20, OC8_INST[k].ng, OC8_INST[k].nh, OC8_MAX_CHAIN);
          // (a pattern without an arrow head -- the reduced form -- runs the same instances with no hub block: two-kernel form only)
          if (o.ok && (o.has_hub || !getenv("MPCQP_OC_MONO")) && lds_bytes_oc(p8, r8, o, OC8_INST[k].zyg) <= OC8_LDS_MAX) {
            h->ocplan = o; h->oc8 = k + 1; h->oc = true; h->gblocks = true; h->zyg = OC8_INST[k].zyg; want = 8; p4 = p8;
          }
        }
      }
      if (!h->oc8 && variant_request() && std::string(variant_request()) == "oc8")
        return bail(fail(MPCQP_ERR_LIMIT, "the eight-wave on-chip variant does not take this pattern / size"));
    }
    if (h->oc && !h->oc8) {
      // on-chip mode: block tridiagonal + arrow patterns whose factor fits LDS + the registers of the instance at two workgroups per CU
      const ResPlan r4 = build_res_plan(p4, 4, false);
      h->oc = false;
      OcPlan o = small_ok ? build_oc_plan(p4, 4, 1 << 20, OC_NG, OC_NH) : OcPlan();
      bool dissected4 = false, padded4 = false;
      if (small_ok && twist && !getenv("MPCQP_NO_PADTWIST")) {
        // the padded twist (ordering 3: the hub moved up to a block boundary, the chain part whole blocks) where the plain order is not taken -- the hub shares a
        // block with the last frame and spills into a second one: cart-pole N=22, 25 fell to the LDS-resident kernel, 18 ms against 12 -- or leaves one long chain
        // where the twist has two (cart-pole N=24: one chain of 7)
        Plan q = build_plan(n, m, Pp, Pi, Ap, Ai, 3, 2);
        if (q.error.empty()) {
          const OcPlan oq = build_oc_plan(q, 4, 1 << 20, OC_NG, OC_NH);
          auto longest = [](const OcPlan &x) { return std::max(x.chainE.size(), x.chainF.size()); };
          const bool fits = oq.ok && lds_bytes_oc(q, build_res_plan(q, 4, false), oq) <= OC_LDS_MAX, o_fits = o.ok && lds_bytes_oc(p4, r4, o) <= OC_LDS_MAX;
          if (fits && (!o_fits || longest(oq) < longest(o))) { o = oq; p4 = q; padded4 = true; }
        }
      }
      if (o.ok && o.has_hub && twist && !getenv("MPCQP_NO_DISSECT") && !getenv("MPCQP_OC_MONO") && !getenv("MPCQP_TILES") && !getenv("MPCQP_VTILES") && !getenv("MPCQP_DOUBLES")) {
        // the dissected order with one separator: two twisted pairs on the four waves (plan.hpp build_plan ordering 4; its own kernel instances, two-kernel form only)
        Plan pd = build_plan(n, m, Pp, Pi, Ap, Ai, 4, 2, 1);
        if (pd.error.empty()) {
          const OcPlan od = build_oc_plan(pd, 4, 1 << 20, OC_NG, OC_NH);
          if (od.ok && od.pairs.size() == 2 && lds_bytes_oc(pd, build_res_plan(pd, 4, false), od) <= OC_LDS_MAX) { o = od; p4 = pd; dissected4 = true; }
        }
      }
      if (o.ok && (dissected4 || padded4 || lds_bytes_oc(p4, r4, o) <= OC_LDS_MAX)) {
        h->ocplan = o; h->oc = true;
        // same ordering and blocks, ELL widths for this instance's 8 slots in flight (plan.hpp build_ell pad = 2)
        if (!dissected4 && !padded4 && !getenv("MPCQP_OC_PAD4")) { Plan poc = build_plan(n, m, Pp, Pi, Ap, Ai, twist ? 2 : -1, 2); if (poc.error.empty() && poc.nblk == p4.nblk) p4 = poc; }
      }
      if (!h->oc && want == 4 && variant_request() && std::string(variant_request()) == "oc4")
        return bail(fail(MPCQP_ERR_LIMIT, "the on-chip variant does not take this pattern / size"));
    }
    if (want > 0) { h->plan = want >= 2 ? p4 : p1; h->wl = ws_layout(h->plan); }
    if (want > 0) {
      h->rplan = build_res_plan(pl, want, h->gblocks && !h->oc);
      if (h->oc && h->ocplan.has_hub && getenv("MPCQP_TILES") && getenv("MPCQP_TILES")[0] == '1') {
        // EXPERIMENTAL, opt-in (MPCQP_TILES=1): dense tiles for the two sweeps of the iteration where the pattern has them (dense Jacobian
        // blocks: the quadrotor's 12 x 16 per stage) and the LDS still fits.  Parity-green, but measured SLOWER than the ELL sweeps in these
        // register-bound instances (quadrotor N = 20 x 8192: 12.0 - 13.7 ms against 8.68), so the default stays ELL.
        h->tplan = build_tile_plan(pl, n, m, Ap, Ai, 2);
        h->tiles = h->tplan.on && h->tplan.max_per_block <= 1 && h->tplan.max_per_chunk <= 8 && h->tplan.rows_consecutive &&
                   pl.A.nchunks <= 3 * want && pl.At.nchunks <= 2 * want &&      // (kernel_onchip.hpp OC_TILE_MAXA / OC_TILE_MAXT chunk records per wave)
                   lds_bytes_oc(pl, h->rplan, h->ocplan, h->oc8 && h->zyg, &h->tplan) <= (h->oc8 ? OC8_LDS_MAX : OC_LDS_MAX);
        if (h->tiles) h->wl = ws_layout(pl, &h->tplan);
      }
      if (h->oc && !h->tiles && getenv("MPCQP_VTILES") && getenv("MPCQP_VTILES")[0] == '1' && !getenv("MPCQP_OC_MONO")) {
        // EXPERIMENT, opt-in (MPCQP_VTILES=1): the two sweeps of the iteration on ONE copy of A's dense blocks -- 16 x 16 tiles, row-major in the slab, multiplied
        // on the vector ALUs (kernel_oc_split.hpp) -- plus the remainder ELL layouts.  No LDS beyond the ELL form's.  The set-up still writes the two ELL copies:
        // the residual sweeps of the termination checks and the factorisation read them.
        h->tplan = build_tile_plan(pl, n, m, Ap, Ai, 2);
        bool fits = h->tplan.on && h->tplan.max_per_block <= 1 && h->tplan.max_per_chunk <= 8 && h->tplan.rows_consecutive &&
                    pl.A.nchunks <= 8 * want && pl.At.nchunks <= 16 * want;          // (a wave's tile records ride in the lanes of registers: 8 tiles x 8 chunks of A, 4 blocks x 16 chunks of A')
        for (int t = 0; t < h->tplan.ntile && fits; t++) {
          int first = -1; for (int r = 0; r < BS; r++) if (h->tplan.rowid[(size_t)t * BS + r] >= 0) { first = h->tplan.rowid[(size_t)t * BS + r]; break; }
          fits = first >= 0 && first + BS <= pl.mpad;                                 // (a tile's sixteen rows of w are read as they lie: all inside the vector)
        }
        h->vtiles = fits;
        if (h->vtiles) h->wl = ws_layout(pl, &h->tplan);
      }
      long need = h->oc ? lds_bytes_oc(pl, h->rplan, h->ocplan, h->oc8 && h->zyg, h->tiles ? &h->tplan : nullptr) : h->gblocks ? lds_bytes_res_gb(pl, h->rplan) : lds_bytes_res(pl, h->rplan);
      if (h->oc && !h->tiles && getenv("MPCQP_DOUBLES")) {
        // EXPERIMENT, opt-in (MPCQP_DOUBLES=<n>): double stages of the solve (plan.hpp oc_add_doubles) where the CU's LDS has room for their product blocks
        // beside the factor: every one takes a dependent 16 x 16 mat-vec off the critical path of both triangular sweeps.  Parity-green, but measured
        // SLOWER (cart-pole N=100: 29.1 against 27.8 ms): the four wave-parallel phases it adds (two mat-vecs per double stage and direction, 48 cycles
        // of matrix pipe per MFMA, four more barriers) cost more than the halved chains save.
        const long cap = h->oc8 ? OC8_LDS_MAX : OC_LDS_MAX;
        int nd = (int)std::max<long>(0, (cap - need) / (BLK * 8));
        nd = std::min(nd, std::max(0, atoi(getenv("MPCQP_DOUBLES"))));
        for (; nd > 0; nd--) {       // (the table grows with the stages: take as many as still fit)
          OcPlan o2 = h->ocplan; oc_add_doubles(o2, nd);
          const long n2 = lds_bytes_oc(pl, h->rplan, o2, h->oc8 && h->zyg, nullptr);
          if (n2 <= cap) { h->ocplan = o2; need = n2; break; }
        }
      }
      if (h->gblocks && !h->oc && want == 4 && !getenv("MPCQP_NO_ZYG")) {     // (the two-wave global-block kernel has no such instance: forced on a long horizon it took this layout and returned garbage)
        // long horizons: with z and y in the slab one more workgroup fits per CU (2 -> 3 or 1 -> 2); measured on quadrotor N=50
        const long alt = lds_bytes_res_gb(pl, h->rplan, true);
        const long fit = LDS_MAX / need, fit_alt = std::min<long>(LDS_MAX / alt, 3);
        if (fit <= 2 && fit_alt > fit) { h->zyg = true; need = alt; }
      }
      h->occ4 = h->gblocks && !h->oc && !h->zyg && need <= 53 * 1024 && !getenv("MPCQP_GB_OCC2");
      // LDS between 40 and 53 KiB: three workgroups per CU fit, so the instance compiled for three waves per SIMD (168 VGPRs, no
      // spills, 8 blocks in flight) replaces the 128-VGPR one (at 42 KiB 92.9k -> 95.8k QP/s on cart-pole N=100, which now fits four per CU
      // because the temp tiles alias w, plan.hpp gb_tmp_alias: 76.2 -> 73.2 ms per 8192; at 32 KiB it loses, 589k -> 551k)
      // (with z and y in the slab the 168-VGPR instance at three per CU also beats the 128-VGPR one at four: quadrotor N=50 23.9 vs 26.0 ms)
      h->occ3 = h->gblocks && !h->oc && need <= 53 * 1024 && (need > 40 * 1024 || h->zyg || getenv("MPCQP_GB_OCC3")) && !getenv("MPCQP_GB_OCC2");
      if (!small_ok || need > LDS_MAX) return bail(fail(MPCQP_ERR_LIMIT, "resident variant needs " + std::to_string(need) + " B of LDS"));
      h->lds = need;
      h->res1x = want == 1 && !h->gblocks && LDS_MAX / need > 8 && !getenv("MPCQP_NO_RES1X");
      if (!h->gblocks && want == 4 && !getenv("MPCQP_NO_RES3")) h->res3 = need <= 40 * 1024 ? 4 : need <= 53 * 1024 ? 3 : 0;
      if (const char *pad = getenv("MPCQP_LDS_MIN")) h->lds = std::max<long>(h->lds, atol(pad));   // experiment: limit workgroups per CU
    }
    h->variant = want;
  }
  if (h->lds > 160 * 1024) return bail(fail(MPCQP_ERR_LIMIT, "LDS footprint " + std::to_string(h->lds) + " B exceeds 160 KiB per CU"));
  DevPlan &dp = h->dp;
  memset(&dp, 0, sizeof(dp));
  dp.n = n; dp.m = m; dp.npad = pl.npad; dp.mpad = pl.mpad; dp.nb = pl.nb; dp.nblk = pl.nblk; dp.nfac = (int)pl.fac.size(); dp.nT = pl.nT;
  int rc;
#define UP(expr) if ((rc = (expr))) return bail(rc)
  UP(upload_ell(h, pl.A, &dp.A)); UP(upload_ell(h, pl.At, &dp.At)); UP(upload_ell(h, pl.P, &dp.P));
  UP(upload(h, pl.pos, &dp.pos)); UP(upload(h, pl.perm, &dp.perm));
  UP(upload(h, pl.fwd_ops, &dp.fwd_ops)); UP(upload(h, pl.bwd_ops, &dp.bwd_ops)); UP(upload(h, pl.bwd_of, &dp.bwd_of));
  {
    std::vector<int4> f(pl.fac.size());
    for (size_t i = 0; i < f.size(); i++) f[i] = make_int4(pl.fac[i].type, pl.fac[i].dst, pl.fac[i].a, pl.fac[i].b);
    UP(upload(h, f, &dp.fac));
  }
  UP(upload(h, pl.tpos, &dp.tpos)); UP(upload(h, pl.asm_ptr, &dp.asm_ptr)); UP(upload(h, pl.asm_a, &dp.asm_a));
  UP(upload(h, pl.asm_b, &dp.asm_b)); UP(upload(h, pl.asm_pidx, &dp.asm_pidx)); UP(upload(h, pl.blk_diag, &dp.blk_diag));
  if (h->variant > 0) {
    const ResPlan &rp = h->rplan; DevRes &dr = h->dres;
    dr.nphase = rp.nphase; dr.ntemp = rp.ntemp; dr.nconst = rp.nconst; dr.rext = rp.rext;
    dr.nlev = rp.nlev;
    UP(upload(h, rp.lv_ptr, &dr.lv_ptr)); UP(upload(h, rp.lv_diag, &dr.lv_diag)); UP(upload(h, rp.lw_ptr, &dr.lw_ptr));
    UP(upload(h, rp.lw_slot, &dr.lw_slot)); UP(upload(h, rp.lw_g, &dr.lw_g)); UP(upload(h, rp.lu_ptr, &dr.lu_ptr));
    UP(upload(h, rp.lu_dst, &dr.lu_dst)); UP(upload(h, rp.lu_tmp, &dr.lu_tmp)); UP(upload(h, rp.lu_b, &dr.lu_b));
    UP(upload(h, rp.g_ptr, &dr.g_ptr)); UP(upload(h, rp.g_seg, &dr.g_seg)); dr.n_seg = (int)rp.g_seg.size() / 8; dr.stage = h->gblocks ? res_stage_doubles_gb(pl, rp) : res_stage_doubles(pl, rp);
    dr.tmp_alias = h->gblocks && !h->oc && gb_tmp_alias(pl, rp) ? 1 : 0;
    memset(&h->doc, 0, sizeof(h->doc));
    if (h->oc) {
      const OcPlan &o = h->ocplan; DevOc &d = h->doc;
      dr.stage = oc_stage_doubles(o, rp, pl); dr.rext = oc_rext(rp.nw, std::max<int>(1, (int)o.pairs.size())); dr.nconst = 0; dr.n_seg = 0;
      d.nbc = o.nbc; d.has_hub = o.has_hub; d.junc = o.junc; d.npw = o.npw; d.nhr = o.nhr; d.nlds = o.nlds; d.ntab = (int)o.tab.size();
      d.o_chainE = o.o_chainE; d.o_chainF = o.o_chainF; d.o_pos = o.o_pos; d.o_fill = o.o_fill; d.ghub_slot = o.ghub_slot; d.ghub_src = o.ghub_src;
      d.npair = (int)o.pairs.size(); d.o_pair = o.o_pair; d.nfill = o.nfill; d.o_s = o.o_s; d.o_dbl = o.o_dbl; d.ndbl = (int)o.dbl.size(); d.o_pp = o.o_pp;
      d.at_poll = d.at_free = -1;
      // (opt-in since the chains run on the 4-block MFMA: they now reach the ticket before wave 3 has the rows -- 913k with, 917k without)
      // (single-kernel four-wave instance only: the eight-wave solve, oc_solve_long, has no ticket wait, and the two-kernel form sweeps all of A' up front)
      h->split = !h->tiles && !getenv("MPCQP_OC_MONO") && pl.A.nchunks <= 32 * h->variant && pl.At.nchunks <= 32 * h->variant;
      if (!h->split || !mpcqp_kernel_oc_admm_tl(h->variant, h->oc8 ? OC8_INST[h->oc8 - 1].ng : OC_NG, !o.has_hub ? 0 : h->oc8 ? OC8_INST[h->oc8 - 1].nh : OC_NH)) h->vtiles = false;      // (a wave's chunk offsets ride in the lanes of one register: kernel_oc_split.hpp oc_my_chunks)
      if (const char *e = getenv("MPCQP_RESUME_ROUNDS")) h->resume_rounds = std::max(0, std::min(atoi(e), 8));
      if (getenv("MPCQP_LATE") && !h->tiles && !h->oc8 && !h->split) oc_late_chunks(pl, o, 4, 3 /* OC_POLL_TRIP */, &d.at_poll, &d.at_free);
      d.a_lds = (long)pl.A.entries() <= dr.stage ? 1 : 0;
      d.p_lds = d.a_lds && (long)pl.A.entries() + (long)pl.P.entries() <= dr.stage ? 1 : 0;
      h->setup_nw = h->variant; h->dres_setup = dr; h->lds_setup = 0;
      UP(upload(h, o.tab, &d.tab));
      UP(upload(h, oc_asm_records(pl), &d.asm_rec));
      if (h->tiles || h->vtiles) {
        const TilePlan &tp = h->tplan; DevTile &t = d.tl;
        t.on = 1; t.ntile = tp.ntile; t.nAr = tp.Ar.nchunks; t.nAtr = tp.Atr.nchunks; t.Ar_entries = tp.Ar.entries(); t.Atr_entries = tp.Atr.entries();
        UP(upload(h, tp.Ar.chunk_off, &t.Ar_off)); UP(upload(h, tp.Ar.idx, &t.Ar_idx)); UP(upload(h, tp.Ar.src, &t.Ar_src));
        UP(upload(h, tp.Atr.chunk_off, &t.Atr_off)); UP(upload(h, tp.Atr.idx, &t.Atr_idx)); UP(upload(h, tp.Atr.src, &t.Atr_src));
        UP(upload(h, tp.tJ, &t.tJ)); UP(upload(h, tp.rowid, &t.rowid));
        if (h->vtiles) {      // row-major tiles for the vector-ALU form: element (r, c) at 16 r + c (the plan keeps the MFMA operand order [r + 16 (c & 3)][c >> 2])
          std::vector<int> rm(tp.tsrc.size(), -1);
          for (size_t tt = 0; tt < tp.tsrc.size() / BLK; tt++) for (int r = 0; r < BS; r++) for (int c = 0; c < BS; c++)
            rm[tt * BLK + r * BS + c] = tp.tsrc[tt * BLK + (r + BS * (c & 3)) * 4 + (c >> 2)];
          UP(upload(h, rm, &t.tsrc));
        } else UP(upload(h, tp.tsrc, &t.tsrc));
        {   // per-chunk records of fixed size (kernel_onchip.hpp oc_tiles_a / oc_tiles_at): {tile, column block, first row, rows}, padded with the zero tile
          auto first = [&](int tt) { for (int r = 0; r < BS; r++) if (tp.rowid[(size_t)tt * BS + r] >= 0) return tp.rowid[(size_t)tt * BS + r]; return 0; };
          auto rows = [&](int tt) { int k = 0; for (int r = 0; r < BS; r++) k += tp.rowid[(size_t)tt * BS + r] >= 0; return k; };
          std::vector<int> ai(32 * (size_t)pl.A.nchunks, 0), ac(pl.A.nchunks, 0);
          for (int c = 0; c < pl.A.nchunks; c++) {
            ac[c] = tp.ta_ptr[c + 1] - tp.ta_ptr[c];
            for (int u = 0; u < 8; u++) {
              const int tt = u < ac[c] ? tp.ta_tid[tp.ta_ptr[c] + u] : tp.ntile;
              int rec[4] = {tt, tp.tJ[tt], first(tt), rows(tt)};
              std::copy(rec, rec + 4, ai.begin() + 32 * (size_t)c + 4 * u);
            }
          }
          UP(upload(h, ai, &t.ta_info)); UP(upload(h, ac, &t.ta_cnt));
          std::vector<int> ti(16 * (size_t)pl.At.nchunks, 0);
          for (int J = 0; J < 4 * pl.At.nchunks; J++) {
            const int tt = (J < pl.nb && tp.tt_ptr[J + 1] > tp.tt_ptr[J]) ? tp.tt_tid[tp.tt_ptr[J]] : tp.ntile;
            int rec[4] = {tt, first(tt), rows(tt), 0};
            std::copy(rec, rec + 4, ti.begin() + 4 * (size_t)J);
          }
          UP(upload(h, ti, &t.tt_info));
        }
        std::vector<unsigned long long> mask(pl.A.nchunks, 0ull);
        for (size_t k = 0; k < (size_t)tp.ntile * BS; k++) if (tp.rowid[k] >= 0) mask[tp.rowid[k] / WAVE] |= 1ull << (tp.rowid[k] % WAVE);
        UP(upload(h, mask, &t.ta_mask));
        t.o_tile = h->wl.tile; t.o_ellAr = h->wl.ellAr; t.o_ellAtr = h->wl.ellAtr;
      }
    }
  }
  if (h->oc) {      // (after the tables are uploaded: the set-up kernel's copies of the arguments)
    h->doc_setup = h->doc; h->lds_setup = h->lds;
    if (h->split) {
      // The set-up as four-wave workgroups with their own LDS request -- the factorisation's scratch blocks and assembly records, the staged values of A
      // and P where they fit, their 16-bit index tables where those fit too -- and their own vector layout: q stays in the slab, z is never touched, y
      // holds one n-vector of the Ruiz passes (kernel_oc_split.hpp oc_lds).  Three workgroups per CU (the kernel's 164 VGPRs allow no more) beat two
      // wherever A's values still fit beside them, and so does an unstaged third against a half-staged pair; a fully staged pair beats an unstaged
      // three.  Measured (x 8192 unless said, set-up kernel, ms): quadrotor N=20 A + P + index tables at two per CU 2.47, A alone at three 2.35, nothing
      // staged at three 3.26 (round-4 mid build); cart-pole N=50 2.14 / 1.95; cart-pole N=100 A staged at two 4.28, nothing staged at three 3.99;
      // quadrotor N=50 nothing fits: two per CU 8.05, squeezed to three 8.55 (not taken: the footprint is what the layout needs).  DESIGN.md 3.9
      const Plan &pq = h->plan; DevRes &ds = h->dres_setup; DevOc &dd = h->doc_setup;
      const long scratch = 8L * BLK + ((4L * pq.nblk + 15) / 16) * 16;                 // (plan.hpp oc_stage_doubles: OC_LDL_SCR blocks + the assembly records)
      const long vec = 2L * pq.npad + oc_rext(h->variant, std::max<int>(1, (int)h->ocplan.pairs.size())) + pq.mpad + pq.npad + 16L * 4 + 16 + 16L * 4;      // x, r; w; y (an n-vector here); the reduction scratch
      const long tabw = ((long)h->ocplan.o_pos + 1) / 2 + 4 + ((long)pq.A.nchunks + pq.At.nchunks + pq.P.nchunks + 3 + 1 + 1) / 2;
      const long cu = 160L * 1024, nA = (long)pq.A.entries(), nP = (long)pq.P.entries();
      struct Shape { long stage, bytes; int a, p, ix16, zpad, ixo_a, ixo_p; bool fits; };
      const long zoff = 2L * pq.npad + oc_rext(h->variant, std::max<int>(1, (int)h->ocplan.pairs.size()));      // (the z region starts behind x and r: kernel_oc_split.hpp oc_lds<NW, true>)
      auto shape = [&](const long cap_bytes) {
        const long cap = cap_bytes / 8;
        Shape r{scratch, 0, 0, 0, 0, 0, 0, 0, false};
        if (std::max(scratch, nA) + vec + tabw <= cap) { r.stage = std::max(scratch, nA); r.a = 1; }
        if (r.a && std::max(scratch, nA + nP) + vec + tabw <= cap) { r.stage = std::max(scratch, nA + nP); r.p = 1; }
        r.stage = (r.stage + 15) / 16 * 16;
        long total = r.stage + vec + tabw;
        const long zA = (nA / 4 + 15) / 16 * 16, zP = (nP / 4 + 15) / 16 * 16, zAP = ((nA + nP) / 4 + 15) / 16 * 16;
        const bool ix_ok = pq.npad < 65536 && !getenv("MPCQP_NO_IX16");
        if (r.a && r.p && ix_ok) {      // (the ten Ruiz passes then gather without a round trip to the L2 in front of every batch)
          if (total + zAP <= cap) { r.ix16 = 3; r.zpad = (int)zAP; } else if (total + zA <= cap) { r.ix16 = 1; r.zpad = (int)zA; }
          r.ixo_a = (int)(4 * (r.stage + zoff)); r.ixo_p = r.ixo_a + (int)nA;
          total += r.zpad;
        } else if (!r.a && ix_ok) {
          // values in the slab: the index tables alone (a quarter less to read per pass, the gathers' addresses from LDS) -- one of them in the factorisation's
          // scratch, which is idle until the factorisation starts, the other in the z region where that does not cost a workgroup per CU
          const bool a_scr = zA <= r.stage, p_scr = zP <= r.stage;
          auto z_fits = [&](long z) { return total + z <= cap && cu / ((total + z) * 8) == cu / (total * 8); };
          if (a_scr && z_fits(zP)) { r.ix16 = 3; r.ixo_a = 0; r.zpad = (int)zP; r.ixo_p = (int)(4 * (r.stage + zoff)); }
          else if (p_scr && z_fits(zA)) { r.ix16 = 3; r.ixo_p = 0; r.zpad = (int)zA; r.ixo_a = (int)(4 * (r.stage + zoff)); }
          else if (a_scr) { r.ix16 = 1; r.ixo_a = 0; }
          else if (z_fits(zA)) { r.ix16 = 1; r.zpad = (int)zA; r.ixo_a = (int)(4 * (r.stage + zoff)); }
          total += r.zpad;
        }
        r.bytes = total * 8; r.fits = total <= cap;
        return r;
      };
      Shape sh = shape(cu / 2);
      if (const char *e = getenv("MPCQP_SETUP_CAP")) sh = shape(atol(e));
      else { const Shape s3 = shape(cu / 3); if (s3.fits && (s3.a || !(sh.a && sh.p))) sh = s3; }
      dd.a_lds = sh.a; dd.p_lds = sh.p; dd.ix16 = sh.ix16; dd.zpad = sh.zpad; dd.ixo_a = sh.ixo_a; dd.ixo_p = sh.ixo_p;
      ds.stage = sh.stage; h->setup_nw = 4; h->lds_setup = sh.bytes;
      const long vecs = vec + sh.zpad;
      if (h->oc8 && !h->vtiles && !getenv("MPCQP_NO_ABALANCE")) {
        // row chunks of A to waves by longest-processing-time over their load batches (a batch = one round trip to memory; plan.hpp ell_batches8)
        std::vector<int> assign(8 * 32, -1), load(8, 0), cnt(8, 0), order_(pq.A.nchunks);
        for (int c = 0; c < pq.A.nchunks; c++) order_[c] = c;
        auto batches = [&](int c) { return ell_batches8(pq.A.chunk_off[c + 1] - pq.A.chunk_off[c]); };
        std::stable_sort(order_.begin(), order_.end(), [&](int a, int b) { return batches(a) > batches(b); });
        bool okA = true;
        for (int c : order_) {
          int w = 0;
          for (int v = 1; v < 8; v++) if (load[v] < load[w] || (load[v] == load[w] && cnt[v] < cnt[w])) w = v;
          if (cnt[w] >= 32) { okA = false; break; }
          assign[w * 32 + cnt[w]++] = c; load[w] += std::max(1, batches(c));
        }
        if (okA) UP(upload(h, assign, &h->doc.a_assign));
      }
      if (h->oc8) {
        int nb = 0; h->qslots = 256;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, oc_admm_of(h, false), h->variant * WAVE, (size_t)h->lds) == hipSuccess && nb > 0) {
          h->qslots = nb * (prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256);
        } else (void)hipGetLastError();
        UP(dalloc(h, &h->qctr, 16 * 16));
      }
      if (getenv("MPCQP_VERBOSE")) fprintf(stderr, "mpcqp: set-up kernel shape: 4 waves, %ld B of LDS (values of A %s, of P %s, 16-bit index tables %s; A %ld + P %ld entries, vectors %ld, tables %ld doubles), iteration kernel %ld B, %d resident workgroups\n",
                                           h->lds_setup, dd.a_lds ? "staged" : "in the slab", dd.p_lds ? "staged" : "in the slab", dd.ix16 == 3 ? "A and P" : dd.ix16 ? "A" : "off", (long)pq.A.entries(), (long)pq.P.entries(), vecs, tabw, (long)h->lds, h->qslots);
    }
  }
  const WsLayout &w = h->wl;
  dp.o_ellA = w.ellA; dp.o_ellAt = w.ellAt; dp.o_ellP = w.ellP; dp.o_Lf = w.Lf; dp.o_Lb = w.Lb; dp.o_T = w.T;
  dp.o_l = w.l; dp.o_u = w.u; dp.o_D = w.D; dp.o_E = w.E; dp.o_dx = w.dx; dp.o_dy = w.dy; dp.o_Zg = w.Zg; dp.o_Yg = w.Yg; dp.ws_stride = w.stride;
  UP(dalloc(h, &h->ws, (size_t)w.stride * batch));
  // the resident kernels only ever write the structural non-zeros of the T tiles (fixed positions, plan.hpp tpos): their zeros are set here, once
  UP([&]() -> int { HIPCHK(hipMemset(h->ws, 0, (size_t)w.stride * batch * sizeof(double))); HIPCHK(hipStreamSynchronize(0)); return MPCQP_OK; }());
  UP(dalloc(h, &h->ox, (size_t)batch * n)); UP(dalloc(h, &h->oy, (size_t)batch * std::max(m, 1))); UP(dalloc(h, &h->oz, (size_t)batch * std::max(m, 1)));
  UP(dalloc(h, &h->oinfo, (size_t)batch * 4)); UP(dalloc(h, &h->ocs, (size_t)batch));
  UP(dalloc(h, &h->ostatus, (size_t)batch)); UP(dalloc(h, &h->oiters, (size_t)batch));
#ifdef MPCQP_TIMING
  UP(dalloc(h, &h->odbg, (size_t)batch * 16 + 128));
#endif
#undef UP
  h->wide = h->variant == 4 && h->lds > 80 * 1024;
  h->stream_pd8 = h->lds > 40 * 1024 && !getenv("MPCQP_PD4");   // streaming kernel: 8 blocks in flight when one QP per SIMD is all that fits
  if (h->lds > 48 * 1024) {
    // MaxDynamicSharedMemorySize is a property of the kernel function, shared by every handle that launches it: keep a running
    // maximum per function so that a later handle with a smaller footprint never lowers the limit under an earlier one
    static std::mutex mu; static std::map<std::pair<const void *, int>, long> limit;   // (function, device)
    const void *fns[4] = {res_kernel_of(h, false), res_kernel_of(h, true), nullptr, nullptr};
    if (h->variant == 0) fns[0] = fns[1] = h->stream_pd8 ? (const void *)mpcqp_admm_kernel<8> : (const void *)mpcqp_admm_kernel<4>;
    if (h->split) { fns[0] = oc_setup_of(h, false); fns[1] = oc_setup_of(h, true); fns[2] = oc_admm_of(h, false); fns[3] = oc_admm_of(h, true); }
    std::lock_guard<std::mutex> lock(mu);
    for (int k = 0; k < 4; k++) {
      const void *fn = fns[k];
      if (!fn) continue;
      const long want_lds = (h->split && k < 2) ? h->lds_setup : h->lds;
      long &cur = limit[{fn, h->device}];
      if (want_lds <= cur) continue;
      if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)want_lds) != hipSuccess)
        return bail(fail(MPCQP_ERR_HIP, "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed"));
      cur = want_lds;
    }
  }
  if (h->variant > 0 && !(h->split ? oc_setup_of(h, false) && oc_setup_of(h, true) && oc_admm_of(h, false) && oc_admm_of(h, true) : res_kernel_of(h, false) && res_kernel_of(h, true)))
    return bail(fail(MPCQP_ERR_STATE, "no kernel instance for this handle (kernel_table.hpp)"));
  if (hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess || hipEventCreate(&h->ev_mid) != hipSuccess) return bail(fail(MPCQP_ERR_HIP, "hipEventCreate failed"));
  memset(&h->io, 0, sizeof(h->io));
  h->lpt = !getenv("MPCQP_NO_LPT");
  {   // dispatch-hint buffers up front: nothing is allocated inside mpcqp_solve, so a solve can be captured in a HIP graph
    int rc;
    if ((rc = dalloc(h, &h->order[0], (size_t)batch)) || (rc = dalloc(h, &h->order[1], (size_t)batch))) return bail(rc);
    if (hipEventCreateWithFlags(&h->ev_order, hipEventDisableTiming) != hipSuccess) return bail(fail(MPCQP_ERR_HIP, "hipEventCreate failed"));
  }
  *out = h;
  return MPCQP_OK;
}

// ---- kernel family by measurement (include/mpcqp.h mpcqp_create_tuned)
static std::mutex g_tune_mu;
static std::map<std::string, std::string> g_tune_cache;     // pattern + batch + device -> family
static std::string tune_key(int n, int m, int batch, int dev, const int *Pp, const int *Pi, const int *Ap, const int *Ai) {
  unsigned long long hsh = 1469598103934665603ULL;          // FNV-1a over the index arrays
  auto mix = [&](const int *p, long cnt) { for (long k = 0; k < cnt; k++) { hsh ^= (unsigned)p[k]; hsh *= 1099511628211ULL; } };
  mix(Pp, n + 1); mix(Pi, Pp[n]); mix(Ap, n + 1); mix(Ai, Ap[n]);
  return std::to_string(n) + "x" + std::to_string(m) + "x" + std::to_string(batch) + "@" + std::to_string(dev) + ":" + std::to_string(hsh);
}
int mpcqp_create_tuned(int n, int m, int batch, const int *Pp, const int *Pi, const int *Ap, const int *Ai,
                       const mpcqp_settings *settings, mpcqp_handle **out) {
  if (!out) return fail(MPCQP_ERR_ARG, "out is null");
  *out = nullptr;
  if (n <= 0 || m < 0 || batch <= 0 || !Pp || !Pi || !Ap || !Ai) return fail(MPCQP_ERR_ARG, "Invalid dimensions.");
  if (Pp[0] != 0 || Ap[0] != 0) return fail(MPCQP_ERR_ARG, "colptr must start at 0");
  for (int j = 0; j < n; j++) if (Pp[j + 1] < Pp[j] || Ap[j + 1] < Ap[j]) return fail(MPCQP_ERR_ARG, "colptr not monotone");
  for (int k = 0; k < Pp[n]; k++) if (Pi[k] < 0 || Pi[k] >= n) return fail(MPCQP_ERR_ARG, "P row index out of range");      // (before the pattern is hashed and a synthetic QP filled through it)
  for (int k = 0; k < Ap[n]; k++) if (Ai[k] < 0 || Ai[k] >= m) return fail(MPCQP_ERR_ARG, "A row index out of range");
  struct Force { const char *prev; explicit Force(const char *v) : prev(g_force_variant) { g_force_variant = v; } ~Force() { g_force_variant = prev; } };
  mpcqp_settings st; if (settings) st = *settings; else mpcqp_default_settings(&st);
  int dev = st.device; if (dev < 0 && hipGetDevice(&dev) != hipSuccess) dev = 0;
  const std::string key = tune_key(n, m, batch, dev, Pp, Pi, Ap, Ai);
  {
    std::string cached;
    { std::lock_guard<std::mutex> lock(g_tune_mu); auto it = g_tune_cache.find(key); if (it != g_tune_cache.end()) cached = it->second; }
    if (!cached.empty()) {
      int rc;
      { Force f(cached == "rule" ? "" : cached.c_str()); rc = mpcqp_create(n, m, batch, Pp, Pi, Ap, Ai, settings, out); }
      if (rc != MPCQP_ERR_LIMIT || cached == "rule") return rc;
      Force f("");       // the cached family no longer takes the size (other settings): the rule's choice instead of an error
      return mpcqp_create(n, m, batch, Pp, Pi, Ap, Ai, settings, out);
    }
  }
  // the synthetic QP on this pattern: P = unit diagonal (other entries 0: positive semidefinite whatever the pattern), A pseudo-random in
  // [-1, 1], q pseudo-random, -1 <= A x <= 1 (feasible at the origin); one instance shared by the batch (stride 0)
  const long nnzP = Pp[n], nnzA = Ap[n];
  std::vector<double> Pv(std::max<long>(nnzP, 1), 0.0), Av(std::max<long>(nnzA, 1), 0.0), qv(n), lv(std::max(m, 1), -1.0), uv(std::max(m, 1), 1.0);
  unsigned long long seed = 88172645463325252ULL;
  auto rnd = [&]() { seed ^= seed << 13; seed ^= seed >> 7; seed ^= seed << 17; return (double)(seed >> 11) / 9007199254740992.0 * 2.0 - 1.0; };
  for (int j = 0; j < n; j++) for (int k = Pp[j]; k < Pp[j + 1]; k++) if (Pi[k] == j) Pv[k] = 1.0;
  for (long k = 0; k < nnzA; k++) Av[k] = rnd();
  for (int j = 0; j < n; j++) qv[j] = rnd();
  // fixed work: set-up + TUNE_ITERS ADMM iterations, no early exit (eps = 0), no adaptive rho -- what a typical MPC solve costs (25 - 50 iterations)
  mpcqp_settings ts = st;
  ts.max_iter = 50; ts.eps_abs = ts.eps_rel = 0.0; ts.eps_prim_inf = ts.eps_dual_inf = 0.0; ts.adaptive_rho = 0; ts.warm_start = 0;
  const char *families[] = {"", "oc4", "oc8", "gres4", "gres2", "res4", "res2", "res1"};
  mpcqp_handle *best = nullptr; float best_ms = 0.f; std::string best_family; long seen_codes[8]; int nseen = 0;
  std::vector<double> xref;
  int last_rc = MPCQP_ERR_LIMIT;
  for (const char *fam : families) {
    mpcqp_handle *h = nullptr;
    int rc;
    { Force f(fam); rc = mpcqp_create(n, m, batch, Pp, Pi, Ap, Ai, &ts, &h); }
    if (rc != MPCQP_OK) { if (rc != MPCQP_ERR_LIMIT) last_rc = rc; continue; }
    long info[16]; mpcqp_plan_info(h, info);
    const long code = info[15] * 1000000 + info[7];                 // family + LDS footprint: the same kernel instance is timed once
    bool dup = false; for (int k = 0; k < nseen; k++) dup |= seen_codes[k] == code;
    if (dup) { mpcqp_destroy(h); continue; }
    seen_codes[nseen++] = code;
    float ms = 0.f; bool ok = true;
    mpcqp_set_dispatch_hint(h, 0);
    ok = mpcqp_update(h, Pv.data(), 0, qv.data(), 0, Av.data(), 0, lv.data(), 0, uv.data(), 0, MPCQP_MEM_HOST) == MPCQP_OK;
    for (int rep = 0; rep < 2 && ok; rep++) ok = mpcqp_solve(h, nullptr) == MPCQP_OK && mpcqp_last_kernel_ms(h, &ms) == MPCQP_OK;     // (the first launch warms the code object)
    // a candidate only counts if it did the work: every instance ran the full iteration count, and its x agrees with the first
    // candidate's (the families are the same algorithm: the first and the last instance of the batch are compared to 1e-6)
    if (ok) {
      std::vector<int> stt(batch), itr(batch); std::vector<double> xs((size_t)batch * n);
      ok = mpcqp_get(h, xs.data(), nullptr, nullptr, stt.data(), itr.data(), nullptr, MPCQP_MEM_HOST) == MPCQP_OK;
      for (int b = 0; b < batch && ok; b++) ok = itr[b] == ts.max_iter && (stt[b] == MPCQP_MAX_ITER_REACHED || stt[b] == MPCQP_SOLVED || stt[b] == MPCQP_SOLVED_INACCURATE);
      if (ok) {
        std::vector<double> x0(xs.begin(), xs.begin() + n), x1(xs.end() - n, xs.end());
        if (xref.empty()) xref = x0;
        double scale = 1.0, err = 0.0;
        for (int j = 0; j < n; j++) { scale = std::max(scale, std::fabs(xref[j])); err = std::max(err, std::max(std::fabs(x0[j] - xref[j]), std::fabs(x1[j] - xref[j]))); }
        ok = err == err && err <= 1e-6 * scale;
      }
    }
    if (!ok) { mpcqp_destroy(h); continue; }
    if (!best || ms < best_ms) { if (best) mpcqp_destroy(best); best = h; best_ms = ms; best_family = fam[0] ? fam : "rule"; }
    else mpcqp_destroy(h);
  }
  if (!best) return last_rc == MPCQP_ERR_LIMIT ? fail(MPCQP_ERR_LIMIT, "no kernel family takes this pattern / size") : last_rc;
  mpcqp_destroy(best);       // (its settings were the tuning run's: the handle that is returned is created afresh with the caller's)
  { std::lock_guard<std::mutex> lock(g_tune_mu); g_tune_cache[key] = best_family; }
  Force f(best_family == "rule" ? "" : best_family.c_str());
  return mpcqp_create(n, m, batch, Pp, Pi, Ap, Ai, settings, out);
}

int mpcqp_create_reduced(int n, int m, int batch, const int *Pp, const int *Pi, const int *Ap, const int *Ai,
                         int nfixed, const int *fixed_rows, const mpcqp_settings *settings, mpcqp_handle **out) {
  if (!out) return fail(MPCQP_ERR_ARG, "out is null");
  *out = nullptr;
  if (n <= 0 || m <= 0 || batch <= 0 || !Pp || !Pi || !Ap || !Ai || nfixed < 0 || (nfixed > 0 && !fixed_rows)) return fail(MPCQP_ERR_ARG, "Invalid dimensions.");
  for (int j = 0; j < n; j++) {
    if (Pp[j + 1] < Pp[j] || Ap[j + 1] < Ap[j]) return fail(MPCQP_ERR_ARG, "colptr not monotone");
    for (int k = Pp[j]; k < Pp[j + 1]; k++) if (Pi[k] < 0 || Pi[k] >= n) return fail(MPCQP_ERR_ARG, "P row index out of range");
    for (int k = Ap[j]; k < Ap[j + 1]; k++) if (Ai[k] < 0 || Ai[k] >= m) return fail(MPCQP_ERR_ARG, "A row index out of range");
  }
  RedMaps rm = build_red_maps(n, m, Pp, Pi, Ap, Ai, nfixed, fixed_rows);
  if (!rm.error.empty()) return fail(MPCQP_ERR_ARG, rm.error);
  mpcqp_handle *inner = nullptr;
  int rc = mpcqp_create(rm.nr, rm.mr, batch, rm.Ppr.data(), rm.Pir.data(), rm.Apr.data(), rm.Air.data(), settings, &inner);
  if (rc) return rc;
  mpcqp_handle *h = new mpcqp_handle();
  h->inner = inner; h->st = inner->st; h->device = inner->device; h->n = n; h->m = m; h->batch = batch; h->variant = -1;
  h->plan.n = n; h->plan.m = m; h->plan.nnzP_in = Pp[n]; h->plan.nnzA_in = Ap[n];
  h->red = rm;
  auto bail = [&](int code) { mpcqp_destroy(h); return code; };
  DevRed &d = h->dred;
  memset(&d, 0, sizeof(d));
  d.n = n; d.m = m; d.nr = rm.nr; d.mr = rm.mr; d.nfix = rm.nfix; d.nnzPr = (int)rm.Pir.size(); d.nnzAr = (int)rm.Air.size();
#define UP(expr) if ((rc = (expr))) return bail(rc)
  UP(upload(h, rm.Psrc, &d.Psrc)); UP(upload(h, rm.Asrc, &d.Asrc)); UP(upload(h, rm.fix_var, &d.fix_var)); UP(upload(h, rm.fix_row, &d.fix_row));
  UP(upload(h, rm.fix_src, &d.fix_src)); UP(upload(h, rm.free_var, &d.free_var)); UP(upload(h, rm.kept_row, &d.kept_row));
  UP(upload(h, rm.var_of, &d.var_of)); UP(upload(h, rm.row_of, &d.row_of));
  UP(upload(h, rm.qc_ptr, &d.qc_ptr)); UP(upload(h, rm.qc_k, &d.qc_k)); UP(upload(h, rm.qc_src, &d.qc_src));
  UP(upload(h, rm.lc_ptr, &d.lc_ptr)); UP(upload(h, rm.lc_k, &d.lc_k)); UP(upload(h, rm.lc_src, &d.lc_src));
  UP(upload(h, rm.yp_ptr, &d.yp_ptr)); UP(upload(h, rm.yp_var, &d.yp_var)); UP(upload(h, rm.yp_src, &d.yp_src));
  UP(upload(h, rm.ya_ptr, &d.ya_ptr)); UP(upload(h, rm.ya_row, &d.ya_row)); UP(upload(h, rm.ya_src, &d.ya_src));
  const size_t B = batch;
  UP(dalloc(h, &d.Pr, B * std::max(d.nnzPr, 1))); UP(dalloc(h, &d.qr, B * rm.nr)); UP(dalloc(h, &d.Ar, B * std::max(d.nnzAr, 1)));
  UP(dalloc(h, &d.lr, B * std::max(rm.mr, 1))); UP(dalloc(h, &d.ur, B * std::max(rm.mr, 1))); UP(dalloc(h, &d.xfix, B * std::max(rm.nfix, 1)));
  UP(dalloc(h, &d.bad, B));
  if (h->st.warm_start) { UP(dalloc(h, &h->rx0, B * rm.nr)); UP(dalloc(h, &h->ry0, B * std::max(rm.mr, 1))); }     // (nothing is allocated inside a solve: it stays graph-capturable)
  UP(dalloc(h, &h->ox, B * n)); UP(dalloc(h, &h->oy, B * m)); UP(dalloc(h, &h->oz, B * m));
  UP(dalloc(h, &h->oinfo, B * 4)); UP(dalloc(h, &h->ostatus, B)); UP(dalloc(h, &h->oiters, B));
#undef UP
  memset(&h->io, 0, sizeof(h->io));
  if (hipEventCreateWithFlags(&h->ev0r, hipEventDisableTiming) != hipSuccess) return bail(fail(MPCQP_ERR_HIP, "hipEventCreate failed"));
  *out = h;
  return MPCQP_OK;
}

int mpcqp_create_presolved(int n, int m, int batch, const int *Pp, const int *Pi, const int *Ap, const int *Ai,
                           const double *l, long sl, const double *u, long su, int mem,
                           const mpcqp_settings *settings, mpcqp_handle **out, int *nfixed_out) {
  if (!out) return fail(MPCQP_ERR_ARG, "out is null");
  *out = nullptr;
  if (nfixed_out) *nfixed_out = 0;
  if (n <= 0 || m <= 0 || batch <= 0 || !Pp || !Pi || !Ap || !Ai || !l || !u) return fail(MPCQP_ERR_ARG, "Invalid dimensions.");
  if (sl < 0 || su < 0 || (sl && sl < m) || (su && su < m)) return fail(MPCQP_ERR_ARG, "stride smaller than the array it strides (dimension mismatch)");
  if (mem != MPCQP_MEM_HOST && mem != MPCQP_MEM_DEVICE) return fail(MPCQP_ERR_ARG, "mem must be MPCQP_MEM_HOST or MPCQP_MEM_DEVICE");
  if (Ap[0] != 0) return fail(MPCQP_ERR_ARG, "colptr must start at 0");
  for (int j = 0; j < n; j++) {
    if (Ap[j + 1] < Ap[j]) return fail(MPCQP_ERR_ARG, "colptr not monotone");
    for (int k = Ap[j]; k < Ap[j + 1]; k++) if (Ai[k] < 0 || Ai[k] >= m) return fail(MPCQP_ERR_ARG, "A row index out of range");
  }
  // the bounds of the first update on the host (device arrays: one copy, at creation only)
  const size_t nl = sl ? (size_t)sl * (batch - 1) + m : (size_t)m, nu = su ? (size_t)su * (batch - 1) + m : (size_t)m;
  std::vector<double> hl, hu;
  if (mem == MPCQP_MEM_DEVICE) {
    hl.resize(nl); hu.resize(nu);
    HIPCHK(hipMemcpy(hl.data(), l, nl * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(hu.data(), u, nu * sizeof(double), hipMemcpyDeviceToHost));
    l = hl.data(); u = hu.data();
  }
  // rows with a single entry in A whose bounds coincide in EVERY instance; one row per variable (a second one stays an ordinary row)
  std::vector<int> cnt(m, 0), col(m, -1);
  for (int j = 0; j < n; j++) for (int k = Ap[j]; k < Ap[j + 1]; k++) { cnt[Ai[k]]++; col[Ai[k]] = j; }
  std::vector<char> taken(n, 0);
  std::vector<int> rows;
  for (int i = 0; i < m; i++) {
    if (cnt[i] != 1 || taken[col[i]]) continue;
    bool eq = true;
    for (int b = 0; b < (sl || su ? batch : 1) && eq; b++) {
      const double lo = l[(size_t)b * sl + i], up = u[(size_t)b * su + i];
      eq = std::fabs(up - lo) <= 1e-9 * std::max(1.0, std::fabs(lo)) && std::fabs(lo) < 1e20;      // (the presolve kernel's own test of the promise)
    }
    if (eq) { rows.push_back(i); taken[col[i]] = 1; }
  }
  if (nfixed_out) *nfixed_out = (int)rows.size();
  if ((int)rows.size() >= n) rows.resize(n - 1);       // (a QP with every variable fixed keeps one: the reduced pattern needs a variable)
  if (rows.empty()) return mpcqp_create(n, m, batch, Pp, Pi, Ap, Ai, settings, out);      // nothing to eliminate: the ordinary handle
  return mpcqp_create_reduced(n, m, batch, Pp, Pi, Ap, Ai, (int)rows.size(), rows.data(), settings, out);
}

// the solve of a reduced handle: substitute the fixed variables, hand the smaller QP to the inner handle, expand its result
static int solve_reduced(mpcqp_handle *h, hipStream_t s) {
  mpcqp_handle *in = h->inner; const DevRed &d = h->dred;
  const bool vectors = h->reuse_next;
  // the presolve overwrites the reduced arrays the previous solve of this handle read: if that one ran on another stream, wait for it
  if (h->solved && h->last_stream != s) { HIPCHK(hipEventRecord(h->ev0r, h->last_stream)); HIPCHK(hipStreamWaitEvent(s, h->ev0r, 0)); }
  hipLaunchKernelGGL(mpcqp_presolve_kernel, dim3(h->batch), dim3(256), 0, s, d, h->io, vectors ? 1 : 0);
  HIPCHK(hipGetLastError());
  int rc;
  if (vectors) rc = mpcqp_update_vectors(in, d.qr, d.nr, d.lr, d.mr, d.ur, d.mr, MPCQP_MEM_DEVICE);
  else rc = mpcqp_update(in, d.Pr, d.nnzPr, d.qr, d.nr, d.Ar, d.nnzAr, d.lr, d.mr, d.ur, d.mr, MPCQP_MEM_DEVICE);
  if (rc) return rc;
  if (h->st.warm_start && h->io.x0 && h->io.y0) {
    if (!h->rx0) return fail(MPCQP_ERR_STATE, "warm start on a reduced handle: settings.warm_start must be set when the handle is created");
    hipLaunchKernelGGL(mpcqp_red_gather_kernel, dim3(h->batch), dim3(256), 0, s, d, h->io.x0, h->io.y0, h->rx0, h->ry0);
    HIPCHK(hipGetLastError());
    if ((rc = mpcqp_warm_start(in, h->rx0, h->ry0, MPCQP_MEM_DEVICE))) return rc;
  }
  if ((rc = mpcqp_solve(in, (void *)s))) return rc;
  hipLaunchKernelGGL(mpcqp_postsolve_kernel, dim3(h->batch), dim3(256), 0, s, d, h->io, in->ox, in->oy, in->oz, in->ostatus, in->oiters, in->oinfo,
                     h->ox, h->oy, h->oz, h->ostatus, h->oiters, h->oinfo);
  HIPCHK(hipGetLastError());
  h->last_stream = s; h->solved = true; h->have_factor = h->keep; h->reuse_next = false;
  return MPCQP_OK;
}

static int stage(mpcqp_handle *h, double **own, const double *src, long stride, long width, const double **dst, long *dstride) {
  // host-memory update: copy into an owned device buffer
  size_t count = stride == 0 ? (size_t)width : (size_t)stride * (h->batch - 1) + width;
  if (!*own) { int rc = dalloc(h, own, (size_t)std::max<long>(width, 1) * h->batch); if (rc) return rc; }
  if (stride != 0 && stride != width) {
    for (int b = 0; b < h->batch; b++) HIPCHK(hipMemcpy(*own + (size_t)b * width, src + (size_t)b * stride, width * sizeof(double), hipMemcpyHostToDevice));
    *dstride = width;
  } else {
    HIPCHK(hipMemcpy(*own, src, count * sizeof(double), hipMemcpyHostToDevice));
    *dstride = stride;
  }
  *dst = *own;
  return MPCQP_OK;
}

int mpcqp_update(mpcqp_handle *h, const double *P, long sP, const double *q, long sq, const double *A, long sA,
                 const double *l, long sl, const double *u, long su, int mem) {
  if (!h) return fail(MPCQP_ERR_ARG, "null handle");
  if (!q || (h->plan.nnzP_in > 0 && !P) || (h->plan.nnzA_in > 0 && !A) || (h->m > 0 && (!l || !u))) return fail(MPCQP_ERR_ARG, "null data pointer");
  if (sP < 0 || sq < 0 || sA < 0 || sl < 0 || su < 0) return fail(MPCQP_ERR_ARG, "negative stride");
  if ((sP && sP < h->plan.nnzP_in) || (sq && sq < h->n) || (sA && sA < h->plan.nnzA_in) || (sl && sl < h->m) || (su && su < h->m))
    return fail(MPCQP_ERR_ARG, "stride smaller than the array it strides (dimension mismatch)");
  HIPCHK(hipSetDevice(h->device));
  DevIO &io = h->io;
  if (mem == MPCQP_MEM_DEVICE) {
    io.P = P; io.sP = sP; io.q = q; io.sq = sq; io.A = A; io.sA = sA; io.l = l; io.sl = sl; io.u = u; io.su = su;
  } else if (mem == MPCQP_MEM_HOST) {
    if (h->last_stream || h->solved) HIPCHK(hipStreamSynchronize(h->last_stream));
    int rc;
    if ((rc = stage(h, &h->dP, P ? P : q, sP, h->plan.nnzP_in, &io.P, &io.sP))) return rc;
    if ((rc = stage(h, &h->dq, q, sq, h->n, &io.q, &io.sq))) return rc;
    if ((rc = stage(h, &h->dA, A ? A : q, sA, h->plan.nnzA_in, &io.A, &io.sA))) return rc;
    if ((rc = stage(h, &h->dl, l ? l : q, sl, h->m, &io.l, &io.sl))) return rc;
    if ((rc = stage(h, &h->du, u ? u : q, su, h->m, &io.u, &io.su))) return rc;
  } else return fail(MPCQP_ERR_ARG, "mem must be MPCQP_MEM_HOST or MPCQP_MEM_DEVICE");
  h->have_data = true; h->reuse_next = false;
  return MPCQP_OK;
}

int mpcqp_warm_start(mpcqp_handle *h, const double *x0, const double *y0, int mem) {
  if (!h || !x0 || !y0) return fail(MPCQP_ERR_ARG, "null pointer");
  HIPCHK(hipSetDevice(h->device));
  if (mem == MPCQP_MEM_DEVICE) { h->io.x0 = x0; h->io.y0 = y0; return MPCQP_OK; }
  int rc;
  if (!h->dx0) { if ((rc = dalloc(h, &h->dx0, (size_t)h->batch * h->n))) return rc; if ((rc = dalloc(h, &h->dy0, (size_t)h->batch * std::max(h->m, 1)))) return rc; }
  HIPCHK(hipMemcpy(h->dx0, x0, (size_t)h->batch * h->n * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(h->dy0, y0, (size_t)h->batch * h->m * sizeof(double), hipMemcpyHostToDevice));
  h->io.x0 = h->dx0; h->io.y0 = h->dy0;
  return MPCQP_OK;
}

int mpcqp_set_dispatch_hint(mpcqp_handle *h, int enable) {
  if (!h) return fail(MPCQP_ERR_ARG, "null handle");
  if (h->inner) return mpcqp_set_dispatch_hint(h->inner, enable);
  h->lpt = enable != 0;
  if (!h->lpt) h->order_cur = -1;
  return MPCQP_OK;
}

int mpcqp_keep_workspace(mpcqp_handle *h, int enable) {
  if (!h) return fail(MPCQP_ERR_ARG, "null handle");
  if (h->inner) {
    const int rc = mpcqp_keep_workspace(h->inner, enable);
    if (rc) return rc;
    h->keep = enable != 0;
    if (!h->keep) { h->have_factor = false; h->reuse_next = false; }
    return MPCQP_OK;
  }
  if (enable && h->variant == 0) return fail(MPCQP_ERR_LIMIT, "the streaming kernel variant does not keep its workspace");
  h->keep = enable != 0;
  if (!h->keep) { h->have_factor = false; h->reuse_next = false; }
  return MPCQP_OK;
}

int mpcqp_update_vectors(mpcqp_handle *h, const double *q, long sq, const double *l, long sl, const double *u, long su, int mem) {
  if (!h) return fail(MPCQP_ERR_ARG, "null handle");
  if (!h->keep || !h->have_factor)
    return fail(MPCQP_ERR_STATE, "mpcqp_update_vectors needs mpcqp_keep_workspace(h, 1) and a completed mpcqp_update + mpcqp_solve before it");
  if (!q || (h->m > 0 && (!l || !u))) return fail(MPCQP_ERR_ARG, "null data pointer");
  if (sq < 0 || sl < 0 || su < 0 || (sq && sq < h->n) || (sl && sl < h->m) || (su && su < h->m)) return fail(MPCQP_ERR_ARG, "dimension mismatch: stride shorter than the array");
  HIPCHK(hipSetDevice(h->device));
  DevIO &io = h->io;
  if (mem == MPCQP_MEM_DEVICE) {
    io.q = q; io.sq = sq; io.l = l; io.sl = sl; io.u = u; io.su = su;
  } else if (mem == MPCQP_MEM_HOST) {
    if (h->last_stream || h->solved) HIPCHK(hipStreamSynchronize(h->last_stream));
    int rc;
    if ((rc = stage(h, &h->dq, q, sq, h->n, &io.q, &io.sq))) return rc;
    if ((rc = stage(h, &h->dl, l ? l : q, sl, h->m, &io.l, &io.sl))) return rc;
    if ((rc = stage(h, &h->du, u ? u : q, su, h->m, &io.u, &io.su))) return rc;
  } else return fail(MPCQP_ERR_ARG, "mem must be MPCQP_MEM_HOST or MPCQP_MEM_DEVICE");
  h->reuse_next = true;
  return MPCQP_OK;
}

int mpcqp_set_rho(mpcqp_handle *h, const double *rho0, int mem) {
  if (!h) return fail(MPCQP_ERR_ARG, "null handle");
  if (h->inner) return mpcqp_set_rho(h->inner, rho0, mem);
  HIPCHK(hipSetDevice(h->device));
  if (!rho0) { h->io.rho0 = nullptr; return MPCQP_OK; }
  if (mem == MPCQP_MEM_DEVICE) { h->io.rho0 = rho0; return MPCQP_OK; }
  if (mem != MPCQP_MEM_HOST) return fail(MPCQP_ERR_ARG, "mem must be MPCQP_MEM_HOST or MPCQP_MEM_DEVICE");
  int rc;
  if (!h->drho0) { if ((rc = dalloc(h, &h->drho0, (size_t)h->batch))) return rc; }
  HIPCHK(hipMemcpy(h->drho0, rho0, (size_t)h->batch * sizeof(double), hipMemcpyHostToDevice));
  h->io.rho0 = h->drho0;
  return MPCQP_OK;
}

int mpcqp_solve(mpcqp_handle *h, void *stream) {
  if (!h) return fail(MPCQP_ERR_ARG, "null handle");
  if (!h->have_data) return fail(MPCQP_ERR_STATE, "Solver not initialized. Call mpcqp_update() first.");
  HIPCHK(hipSetDevice(h->device));
  hipStream_t s = (hipStream_t)stream;
  if (h->inner) return solve_reduced(h, s);
  DevIO io = h->io;
  io.x = h->ox; io.y = h->oy; io.z = h->oz; io.status = h->ostatus; io.iters = h->oiters; io.info = h->oinfo; io.ws = h->ws; io.no_remap = getenv("MPCQP_NO_REMAP") ? 1 : 0; io.no_touch = no_touch_of(h); io.cscale = h->ocs; io.dbg = h->odbg;
  io.reuse = h->reuse_next ? 1 : 0; io.keep = h->keep ? 1 : 0;
  io.order = (h->lpt && h->order_cur >= 0) ? h->order[h->order_cur] : nullptr;
  if (io.order && h->last_stream != s) HIPCHK(hipStreamWaitEvent(s, h->ev_order, 0));    // the hint was written on another stream
  HIPCHK(hipEventRecord(h->ev0, s));
  if (h->variant > 0 && h->split) {     // CuCaQP::initSolver, then CuCaQP::solve
    int rc = launch_oc_split(h, io, h->batch, io.reuse != 0, s, h->ev_mid, 0);
    if (rc) return rc;
  }
  else if (h->variant > 0) {
    void *args[] = {(void *)&h->dp, (void *)&h->dres, (void *)&h->st, (void *)&io, (void *)&h->doc};
    HIPCHK(hipLaunchKernel(res_kernel_of(h, io.reuse != 0), dim3(h->batch), dim3(h->variant * WAVE), args, (size_t)h->lds, s));
  }
  else if (h->stream_pd8) hipLaunchKernelGGL(mpcqp_admm_kernel<8>, dim3(h->batch), dim3(WAVE), (size_t)h->lds, s, h->dp, h->st, io);
  else hipLaunchKernelGGL(mpcqp_admm_kernel<4>, dim3(h->batch), dim3(WAVE), (size_t)h->lds, s, h->dp, h->st, io);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(h->ev1, s));
  if (h->m > 0) {
    hipLaunchKernelGGL(mpcqp_validate_kernel, dim3((h->batch + 3) / 4), dim3(256), 0, s, h->batch, h->n, h->m, io.l, io.sl, io.u, io.su, io.x, io.y, io.z,
                       io.status, io.iters, io.info);
    HIPCHK(hipGetLastError());
  }
  if (h->lpt && h->batch > 1) {   // order of the next solve from this solve's iteration counts
    const int nxt = h->order_cur == 0 ? 1 : 0;
    hipLaunchKernelGGL(mpcqp_order_kernel, dim3(1), dim3(1024), 0, s, (const int *)h->oiters, h->order[nxt], h->batch, std::max(1, h->st.check_termination));
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(h->ev_order, s));
    h->order_cur = nxt;
  }
  h->last_stream = s; h->solved = true; h->have_factor = h->keep;
  return MPCQP_OK;
}

// launch of instances [b0, b0 + count) on stream s: every per-instance pointer of `io` is advanced, the kernels index by blockIdx
static int launch_slice(mpcqp_handle *h, DevIO io, int b0, int count, hipStream_t s, int qslot) {
  const long n = h->n, m = h->m;
  io.P += (long)b0 * io.sP; io.q += (long)b0 * io.sq; io.A += (long)b0 * io.sA; io.l += (long)b0 * io.sl; io.u += (long)b0 * io.su;
  if (io.x0) io.x0 += b0 * n;
  if (io.y0) io.y0 += b0 * m;
  if (io.rho0) io.rho0 += b0;
  io.x += b0 * n; io.y += b0 * m; io.z += b0 * m; io.status += b0; io.iters += b0; io.info += 4L * b0;
  io.ws += (long)b0 * h->dp.ws_stride; io.cscale += b0;
  if (io.dbg) io.dbg += 16L * b0;
  io.order = nullptr;
  if (h->variant > 0 && h->split) {
    int rc = launch_oc_split(h, io, count, false, s, nullptr, qslot);
    if (rc) return rc;
  }
  else if (h->variant > 0) {
    void *args[] = {(void *)&h->dp, (void *)&h->dres, (void *)&h->st, (void *)&io, (void *)&h->doc};
    HIPCHK(hipLaunchKernel(res_kernel_of(h, false), dim3(count), dim3(h->variant * WAVE), args, (size_t)h->lds, s));
  }
  else if (h->stream_pd8) hipLaunchKernelGGL(mpcqp_admm_kernel<8>, dim3(count), dim3(WAVE), (size_t)h->lds, s, h->dp, h->st, io);
  else hipLaunchKernelGGL(mpcqp_admm_kernel<4>, dim3(count), dim3(WAVE), (size_t)h->lds, s, h->dp, h->st, io);
  HIPCHK(hipGetLastError());
  if (m > 0) {
    hipLaunchKernelGGL(mpcqp_validate_kernel, dim3((count + 3) / 4), dim3(256), 0, s, count, h->n, h->m, io.l, io.sl, io.u, io.su, io.x, io.y, io.z,
                       io.status, io.iters, io.info);
    HIPCHK(hipGetLastError());
  }
  return MPCQP_OK;
}

int mpcqp_solve_host(mpcqp_handle *h, const double *P, long sP, const double *q, long sq, const double *A, long sA,
                     const double *l, long sl, const double *u, long su,
                     double *x, double *y, int *status, int *iters, int chunks) {
  if (!h) return fail(MPCQP_ERR_ARG, "null handle");
  if (!q || (h->plan.nnzP_in > 0 && !P) || (h->plan.nnzA_in > 0 && !A) || (h->m > 0 && (!l || !u))) return fail(MPCQP_ERR_ARG, "null data pointer");
  const long wP = h->plan.nnzP_in, wA = h->plan.nnzA_in, n = h->n, m = h->m;
  if (h->inner) return fail(MPCQP_ERR_STATE, "mpcqp_solve_host is not available on a reduced handle");
  if ((sP && sP != wP) || sq != n || (sA && sA != wA) || (m > 0 && (sl != m || su != m)))
    return fail(MPCQP_ERR_ARG, "dimension mismatch: mpcqp_solve_host takes dense instance-major arrays (stride = width; 0 shares P or A)");
  HIPCHK(hipSetDevice(h->device));
  if (h->last_stream || h->solved) HIPCHK(hipStreamSynchronize(h->last_stream));
  chunks = std::max(1, std::min(chunks > 0 ? chunks : 6, h->batch));
  int rc;
  const size_t B = h->batch;
  if (!h->dP) { if ((rc = dalloc(h, &h->dP, (size_t)std::max<long>(wP, 1) * B))) return rc; }
  if (!h->dq) { if ((rc = dalloc(h, &h->dq, (size_t)n * B))) return rc; }
  if (!h->dA) { if ((rc = dalloc(h, &h->dA, (size_t)std::max<long>(wA, 1) * B))) return rc; }
  if (!h->dl) { if ((rc = dalloc(h, &h->dl, (size_t)std::max<long>(m, 1) * B))) return rc; }
  if (!h->du) { if ((rc = dalloc(h, &h->du, (size_t)std::max<long>(m, 1) * B))) return rc; }
  // a slice's kernel ends with a tail (its slowest instance); two slices in flight fill each other's tails.  Two compute streams, not one per slice: the
  // runtime maps streams onto a few hardware queues (four by default), and with eight compute streams beside the copy stream the copies of a later slice
  // queued up behind kernels of earlier ones (rocprofv3 memory-copy trace: gaps of 0.9 - 1.7 ms in the transfer; 12.3 -> 10.3 ms per step on the north-star
  // batch, against a bound of ~10 ms = transfer of one slice + the launch).  MPCQP_PIPE_STREAMS overrides (1 .. 8).
  const int ns = std::min(chunks, getenv("MPCQP_PIPE_STREAMS") ? std::max(1, std::min(atoi(getenv("MPCQP_PIPE_STREAMS")), (int)mpcqp_handle::NPIPE)) : 2);
  for (int i = 0; i < ns; i++) if (!h->pipe[i]) HIPCHK(hipStreamCreateWithFlags(&h->pipe[i], hipStreamNonBlocking));
  DevIO io = h->io;
  io.P = h->dP; io.sP = sP; io.q = h->dq; io.sq = n; io.A = h->dA; io.sA = sA; io.l = h->dl; io.sl = m; io.u = h->du; io.su = m;
  io.x = h->ox; io.y = h->oy; io.z = h->oz; io.status = h->ostatus; io.iters = h->oiters; io.info = h->oinfo; io.ws = h->ws; io.no_remap = getenv("MPCQP_NO_REMAP") ? 1 : 0; io.no_touch = no_touch_of(h); io.cscale = h->ocs; io.dbg = h->odbg;
  io.reuse = 0; io.keep = h->keep ? 1 : 0; io.order = nullptr;
  if (sP == 0 && wP) HIPCHK(hipMemcpy(h->dP, P, wP * sizeof(double), hipMemcpyHostToDevice));      // shared matrices: once
  if (sA == 0 && wA) HIPCHK(hipMemcpy(h->dA, A, wA * sizeof(double), hipMemcpyHostToDevice));
  if (!h->pipe_copy) HIPCHK(hipStreamCreateWithFlags(&h->pipe_copy, hipStreamNonBlocking));
  while ((int)h->pipe_ev.size() < chunks) { hipEvent_t e; HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); h->pipe_ev.push_back(e); }
  for (int c = 0; c < chunks; c++) {
    const int b0 = (int)((long)h->batch * c / chunks), b1 = (int)((long)h->batch * (c + 1) / chunks), cnt = b1 - b0;
    if (cnt <= 0) continue;
    // inputs of all slices queue on ONE stream, in slice order, so that slice 0 is complete after 1/chunks of the transfer
    // (copies spread over several streams share the link and all finish late)
    hipStream_t cs = h->pipe_copy, s = h->pipe[c % ns];
    if (sP) HIPCHK(hipMemcpyAsync(h->dP + (size_t)b0 * wP, P + (size_t)b0 * wP, (size_t)cnt * wP * sizeof(double), hipMemcpyHostToDevice, cs));
    HIPCHK(hipMemcpyAsync(h->dq + (size_t)b0 * n, q + (size_t)b0 * n, (size_t)cnt * n * sizeof(double), hipMemcpyHostToDevice, cs));
    if (sA) HIPCHK(hipMemcpyAsync(h->dA + (size_t)b0 * wA, A + (size_t)b0 * wA, (size_t)cnt * wA * sizeof(double), hipMemcpyHostToDevice, cs));
    if (m) {
      HIPCHK(hipMemcpyAsync(h->dl + (size_t)b0 * m, l + (size_t)b0 * m, (size_t)cnt * m * sizeof(double), hipMemcpyHostToDevice, cs));
      HIPCHK(hipMemcpyAsync(h->du + (size_t)b0 * m, u + (size_t)b0 * m, (size_t)cnt * m * sizeof(double), hipMemcpyHostToDevice, cs));
    }
    HIPCHK(hipEventRecord(h->pipe_ev[c], cs));
    HIPCHK(hipStreamWaitEvent(s, h->pipe_ev[c], 0));
    if ((rc = launch_slice(h, io, b0, cnt, s, 1 + c % ns))) return rc;      // (one set of ticket counters per compute stream: the slices of a stream run one after the other)
    if (x) HIPCHK(hipMemcpyAsync(x + (size_t)b0 * n, h->ox + (size_t)b0 * n, (size_t)cnt * n * sizeof(double), hipMemcpyDeviceToHost, s));
    if (y && m) HIPCHK(hipMemcpyAsync(y + (size_t)b0 * m, h->oy + (size_t)b0 * m, (size_t)cnt * m * sizeof(double), hipMemcpyDeviceToHost, s));
    if (status) HIPCHK(hipMemcpyAsync(status + b0, h->ostatus + b0, (size_t)cnt * sizeof(int), hipMemcpyDeviceToHost, s));
    if (iters) HIPCHK(hipMemcpyAsync(iters + b0, h->oiters + b0, (size_t)cnt * sizeof(int), hipMemcpyDeviceToHost, s));
  }
  HIPCHK(hipStreamSynchronize(h->pipe_copy));
  for (int i = 0; i < ns; i++) HIPCHK(hipStreamSynchronize(h->pipe[i]));
  h->io.P = io.P; h->io.sP = io.sP; h->io.q = io.q; h->io.sq = io.sq; h->io.A = io.A; h->io.sA = io.sA; h->io.l = io.l; h->io.sl = io.sl; h->io.u = io.u; h->io.su = io.su;
  h->have_data = true; h->reuse_next = false; h->solved = true; h->have_factor = h->keep; h->last_stream = nullptr; h->order_cur = -1;
  return MPCQP_OK;
}

int mpcqp_get(mpcqp_handle *h, double *x, double *y, double *z, int *status, int *iters, double *info, int mem) {
  if (!h) return fail(MPCQP_ERR_ARG, "null handle");
  if (!h->solved) return fail(MPCQP_ERR_STATE, "no solve has been issued");
  HIPCHK(hipSetDevice(h->device));
  hipMemcpyKind k = mem == MPCQP_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
  hipStream_t s = h->last_stream;
  const size_t B = h->batch;
  if (x) HIPCHK(hipMemcpyAsync(x, h->ox, B * h->n * sizeof(double), k, s));
  if (y && h->m) HIPCHK(hipMemcpyAsync(y, h->oy, B * h->m * sizeof(double), k, s));
  if (z && h->m) HIPCHK(hipMemcpyAsync(z, h->oz, B * h->m * sizeof(double), k, s));
  if (status) HIPCHK(hipMemcpyAsync(status, h->ostatus, B * sizeof(int), k, s));
  if (iters) HIPCHK(hipMemcpyAsync(iters, h->oiters, B * sizeof(int), k, s));
  if (info) HIPCHK(hipMemcpyAsync(info, h->oinfo, B * 4 * sizeof(double), k, s));
  if (mem != MPCQP_MEM_DEVICE) HIPCHK(hipStreamSynchronize(s));
  return MPCQP_OK;
}

int mpcqp_sync(mpcqp_handle *h) {
  if (!h) return fail(MPCQP_ERR_ARG, "null handle");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipStreamSynchronize(h->last_stream));
  return MPCQP_OK;
}

void mpcqp_destroy(mpcqp_handle *h) {
  if (!h) return;
  if (h->inner) { mpcqp_destroy(h->inner); h->inner = nullptr; }
  (void)hipSetDevice(h->device);
  if (h->solved) (void)hipStreamSynchronize(h->last_stream);
  for (void *p : h->dev_allocs) (void)hipFree(p);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  if (h->ev_mid) (void)hipEventDestroy(h->ev_mid);
  if (h->ev_guard) (void)hipEventDestroy(h->ev_guard);
  if (h->ev_order) (void)hipEventDestroy(h->ev_order);
  if (h->ev0r) (void)hipEventDestroy(h->ev0r);
  for (int i = 0; i < mpcqp_handle::NPIPE; i++) if (h->pipe[i]) (void)hipStreamDestroy(h->pipe[i]);
  if (h->pipe_copy) (void)hipStreamDestroy(h->pipe_copy);
  for (hipEvent_t e : h->pipe_ev) (void)hipEventDestroy(e);
  delete h;
}

int mpcqp_last_kernel_ms(mpcqp_handle *h, float *ms) {
  if (!h || !ms) return fail(MPCQP_ERR_ARG, "null pointer");
  if (!h->solved) return fail(MPCQP_ERR_STATE, "no solve has been issued");
  if (h->inner) return mpcqp_last_kernel_ms(h->inner, ms);
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipEventSynchronize(h->ev1));
  HIPCHK(hipEventElapsedTime(ms, h->ev0, h->ev1));
  return MPCQP_OK;
}

int mpcqp_last_phase_ms(mpcqp_handle *h, float *setup_ms, float *solve_ms) {
  if (!h || !setup_ms || !solve_ms) return fail(MPCQP_ERR_ARG, "null pointer");
  if (!h->solved) return fail(MPCQP_ERR_STATE, "no solve has been issued");
  if (h->inner) return mpcqp_last_phase_ms(h->inner, setup_ms, solve_ms);
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipEventSynchronize(h->ev1));
  if (!h->split) { *setup_ms = 0.f; HIPCHK(hipEventElapsedTime(solve_ms, h->ev0, h->ev1)); return MPCQP_OK; }
  HIPCHK(hipEventElapsedTime(setup_ms, h->ev0, h->ev_mid));
  HIPCHK(hipEventElapsedTime(solve_ms, h->ev_mid, h->ev1));
  return MPCQP_OK;
}

int mpcqp_plan_info(const mpcqp_handle *h, long *o) {
  if (!h || !o) return fail(MPCQP_ERR_ARG, "null pointer");
  if (h->inner) return mpcqp_plan_info(h->inner, o);        // the plan that runs: the reduced pattern's
  const Plan &pl = h->plan;
  o[0] = h->n; o[1] = h->m; o[2] = h->batch; o[3] = pl.npad; o[4] = pl.mpad; o[5] = pl.nb; o[6] = pl.nblk; o[7] = h->lds;
  o[8] = h->wl.stride * 8; o[9] = pl.ordering; o[10] = pl.nnzP_triu; o[11] = pl.nnzA_in; o[12] = pl.nT; o[13] = h->oc ? ((h->tiles || h->vtiles) ? h->tplan.ntile : 0) : (long)pl.fac.size();
  o[14] = pl.A.slots() + pl.At.slots() + pl.P.slots(); o[15] = h->oc ? 200 + h->variant : h->gblocks ? 100 + h->variant : h->variant;
  return MPCQP_OK;
}

int mpcqp_oc_info(const mpcqp_handle *h, long *o) {
  if (!h || !o) return fail(MPCQP_ERR_ARG, "null pointer");
  if (h->inner) return mpcqp_oc_info(h->inner, o);
  for (int k = 0; k < 12; k++) o[k] = 0;
  o[8] = h->plan.A.slots(); o[9] = h->plan.At.slots(); o[10] = h->plan.P.slots();
  if (!h->oc) return MPCQP_OK;
  const OcPlan &p = h->ocplan;
  o[0] = p.nbc; o[1] = p.has_hub; o[2] = (long)p.chainE.size(); o[3] = (long)p.chainF.size(); o[4] = p.nlds; o[5] = p.npw; o[6] = p.nhr; o[7] = h->split ? 1 + h->resume_rounds : 0; o[11] = (long)std::max<size_t>(1, p.pairs.size());
  return MPCQP_OK;
}

int mpcqp_debug_scaling(mpcqp_handle *h, int b, double *D, double *E, double *c) {
  if (!h || b < 0 || b >= h->batch) return fail(MPCQP_ERR_ARG, "bad instance index");
  if (h->inner) return fail(MPCQP_ERR_STATE, "scaling of a reduced handle lives in the reduced dimensions");
  if (!h->solved) return fail(MPCQP_ERR_STATE, "no solve has been issued");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipStreamSynchronize(h->last_stream));
  const Plan &pl = h->plan;
  std::vector<double> Dp(pl.npad);
  const double *base = h->ws + (size_t)b * h->wl.stride;
  if (D) { HIPCHK(hipMemcpy(Dp.data(), base + h->wl.D, pl.npad * sizeof(double), hipMemcpyDeviceToHost)); for (int j = 0; j < h->n; j++) D[j] = Dp[pl.pos[j]]; }
  if (E && h->m) HIPCHK(hipMemcpy(E, base + h->wl.E, h->m * sizeof(double), hipMemcpyDeviceToHost));
  if (c) HIPCHK(hipMemcpy(c, h->ocs + b, sizeof(double), hipMemcpyDeviceToHost));
  return MPCQP_OK;
}

#ifdef MPCQP_TIMING
// timing build only: per-QP cycle counts of the 16 instrumented segments (copied to host)
int mpcqp_debug_timing(mpcqp_handle *h, long long *out) {
  if (!h || !out || !h->odbg) return fail(MPCQP_ERR_ARG, "no timing data");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipStreamSynchronize(h->last_stream));
  HIPCHK(hipMemcpy(out, h->odbg, ((size_t)h->batch * 16 + 128) * sizeof(long long), hipMemcpyDeviceToHost));
  return MPCQP_OK;
}
#endif

int mpcqp_debug_blockops(const double *A, const double *B, const double *C, const double *S, double *out_gemm, double *out_linv, int *potrf_fail) {
  if (!A || !B || !C || !S || !out_gemm || !out_linv || !potrf_fail) return fail(MPCQP_ERR_ARG, "null pointer");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(MPCQP_ERR_NO_GPU, "no device");
  double *d = nullptr; int *df = nullptr;
  HIPCHK(hipMalloc((void **)&d, 6 * BLK * sizeof(double)));
  HIPCHK(hipMalloc((void **)&df, sizeof(int)));
  HIPCHK(hipMemcpy(d, A, BLK * 8, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(d + BLK, B, BLK * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d + 2 * BLK, C, BLK * 8, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(d + 3 * BLK, S, BLK * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(mpcqp_blockops_kernel, dim3(1), dim3(WAVE), 0, 0, d, d + BLK, d + 2 * BLK, d + 3 * BLK, d + 4 * BLK, d + 5 * BLK, df);
  HIPCHK(hipGetLastError());
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(out_gemm, d + 4 * BLK, BLK * 8, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(out_linv, d + 3 * BLK, BLK * 8, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(potrf_fail, df, sizeof(int), hipMemcpyDeviceToHost));
  (void)hipFree(d); (void)hipFree(df);
  return MPCQP_OK;
}

}  // extern "C"
