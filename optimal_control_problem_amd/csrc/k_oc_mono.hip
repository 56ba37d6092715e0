// k_oc_mono.hip -- the on-chip mode as ONE kernel (round 2 / 3 form): kept for the tile experiment (MPCQP_TILES=1) and for A/B runs against the
// two-kernel form (MPCQP_OC_MONO=1).  Compiled twice: -DKREUSE=0 / 1.
#include "kernels_all.hpp"
#ifndef KREUSE
#error "compile with -DKREUSE=0 or -DKREUSE=1"
#endif
#if KREUSE
#define KFN mpcqp_kernel_oc_mono_r1
#else
#define KFN mpcqp_kernel_oc_mono_r0
#endif
MPCQP_HIDDEN const void *KFN(int nw, int ng, int nh, bool tiles) {
  constexpr bool R = KREUSE != 0;
  if (nw == 8) {
    for (int k = 0; k < 2; k++) if (OC8_INST[k].ng == ng && OC8_INST[k].nh == nh) {
      if (k == 0) return tiles ? (const void *)mpcqp_res_kernel<8, 2, true, R, OC8_INST[0].zyg, OC8_INST[0].ng, OC8_INST[0].nh, true>
                               : (const void *)mpcqp_res_kernel<8, 2, true, R, OC8_INST[0].zyg, OC8_INST[0].ng, OC8_INST[0].nh>;
      return tiles ? (const void *)mpcqp_res_kernel<8, 2, true, R, OC8_INST[1].zyg, OC8_INST[1].ng, OC8_INST[1].nh, true>
                   : (const void *)mpcqp_res_kernel<8, 2, true, R, OC8_INST[1].zyg, OC8_INST[1].ng, OC8_INST[1].nh>;
    }
    return nullptr;      // (patterns without an arrow head: the two-kernel form only)
  }
  if (nw != 4 || ng != OC_NG) return nullptr;
  if (tiles) return nh == OC_NH ? (const void *)mpcqp_res_kernel<4, 2, true, R, false, OC_NG, OC_NH, true> : nullptr;
  if (nh == OC_NH) return (const void *)mpcqp_res_kernel<4, 2, true, R, false, OC_NG, OC_NH>;
  if (nh == 0) return (const void *)mpcqp_res_kernel<4, 2, true, R, false, OC_NG, 0>;
  return nullptr;
}
