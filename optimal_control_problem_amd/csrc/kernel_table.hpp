// kernel_table.hpp -- where mpcqp.hip finds the kernel instances that other translation units instantiate (k_*.hip); nullptr = no such instance
#pragma once
#define MPCQP_HIDDEN __attribute__((visibility("hidden")))
// register-resident blocks per wave of the four-wave on-chip instance: inverse diagonal blocks (positions per wave) and hub blocks
constexpr int OC_NG = 5, OC_NH = 3;
// Long chains (more than 20 chain blocks: quadrotor N > 20, cart-pole N > 60): eight waves per QP, one workgroup per CU -- the whole LDS and
// 8 x 256 VGPRs for one factor.  Two instances: up to 32 chain blocks with every hub block in registers (cart-pole N = 100: 62 KB of chain
// blocks + 36 KB of vectors in LDS), and up to 56 with seven positions per wave, G_p and every hub block (both orientations) in registers --
// 168 resident VGPRs -- and only the chain blocks and the hub's inverse in LDS (quadrotor N = 50: 50 blocks = 100 KB + 57 KB of vectors and
// tables = 159,880 B).  (Measured against <NG 7, NH 5> with z, y in the slab, 157,832 B and 136 resident VGPRs: 35.5 against 35.9 ms and an
// eighth less HBM traffic -- the slab vectors cost more than the extra spills.)
struct Oc8Inst { int ng, nh; bool zyg; };
constexpr Oc8Inst OC8_INST[2] = {{4, 4, false}, {7, 7, false}};

// (the instances with and without the kept-workspace entry are separate translation units: _r1 / _r0)
// mpcqp_res_kernel<NW, MINW, false, REUSE>: factor in LDS
MPCQP_HIDDEN const void *mpcqp_kernel_res_lds_r0(int nw, int minw);
MPCQP_HIDDEN const void *mpcqp_kernel_res_lds_r1(int nw, int minw);
// mpcqp_res_kernel<NW, MINW, true, REUSE, ZYG>: factor streamed from the slab
MPCQP_HIDDEN const void *mpcqp_kernel_res_gb_r0(int nw, int minw, bool zyg);
MPCQP_HIDDEN const void *mpcqp_kernel_res_gb_r1(int nw, int minw, bool zyg);
// mpcqp_res_kernel<NW, 2, true, REUSE, false, NG, NH, TL>: the on-chip mode as one kernel (the tile experiment; MPCQP_OC_MONO=1 for A/B runs)
MPCQP_HIDDEN const void *mpcqp_kernel_oc_mono_r0(int nw, int ng, int nh, bool tiles);
MPCQP_HIDDEN const void *mpcqp_kernel_oc_mono_r1(int nw, int ng, int nh, bool tiles);
// kernel_oc_split.hpp: the on-chip mode as set-up + iteration kernels
MPCQP_HIDDEN const void *mpcqp_kernel_oc_setup(int nw, bool hub, bool reuse);
MPCQP_HIDDEN const void *mpcqp_kernel_oc_admm(int nw, int ng, int nh);        // leaves for a re-factorisation
MPCQP_HIDDEN const void *mpcqp_kernel_oc_admm_rf(int nw, int ng, int nh);
MPCQP_HIDDEN const void *mpcqp_kernel_oc_admm_p4(int rf);                      // four waves, two twisted pairs of chains (dissected order)
MPCQP_HIDDEN const void *mpcqp_kernel_oc_admm_tl(int nw, int ng, int nh);     // sweeps on dense tiles of A (experiment, MPCQP_VTILES=1)     // re-factorises in place (the last launch of a solve)
