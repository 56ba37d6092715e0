// k_res_gb.hip -- instances of mpcqp_res_kernel with the factor streamed from the slab (compiled twice: -DKREUSE=0 / 1)
#include "kernels_all.hpp"
#ifndef KREUSE
#error "compile with -DKREUSE=0 or -DKREUSE=1"
#endif
#if KREUSE
#define KFN mpcqp_kernel_res_gb_r1
#else
#define KFN mpcqp_kernel_res_gb_r0
#endif
MPCQP_HIDDEN const void *KFN(int nw, int minw, bool zyg) {
  constexpr bool R = KREUSE != 0;
  if (nw == 2) return (minw == 3 && !zyg) ? (const void *)mpcqp_res_kernel<2, 3, true, R> : nullptr;
  if (nw != 4) return nullptr;
  if (zyg) return minw == 3 ? (const void *)mpcqp_res_kernel<4, 3, true, R, true> : minw == 2 ? (const void *)mpcqp_res_kernel<4, 2, true, R, true> : nullptr;
  if (minw == 3) return (const void *)mpcqp_res_kernel<4, 3, true, R>;
  if (minw == 4) return (const void *)mpcqp_res_kernel<4, 4, true, R>;
  if (minw == 2) return (const void *)mpcqp_res_kernel<4, 2, true, R>;
  return nullptr;
}
