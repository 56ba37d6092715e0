// k_res_lds.hip -- instances of mpcqp_res_kernel with the factor in LDS (compiled twice: -DKREUSE=0 / 1, the kept-workspace entry)
#include "kernels_all.hpp"
#ifndef KREUSE
#error "compile with -DKREUSE=0 or -DKREUSE=1"
#endif
#if KREUSE
#define KFN mpcqp_kernel_res_lds_r1
#else
#define KFN mpcqp_kernel_res_lds_r0
#endif
MPCQP_HIDDEN const void *KFN(int nw, int minw) {
  constexpr bool R = KREUSE != 0;
  if (nw == 1) return minw == 4 ? (const void *)mpcqp_res_kernel<1, 4, false, R> : minw == 2 ? (const void *)mpcqp_res_kernel<1, 2, false, R> : nullptr;
  if (nw == 8) return minw == 2 ? (const void *)mpcqp_res_kernel<8, 2, false, R> : nullptr;
  if (nw == 2) return minw == 3 ? (const void *)mpcqp_res_kernel<2, 3, false, R> : nullptr;
  if (nw == 4) {
    if (minw == 4) return (const void *)mpcqp_res_kernel<4, 4, false, R>;
    if (minw == 3) return (const void *)mpcqp_res_kernel<4, 3, false, R>;
    if (minw == 1) return (const void *)mpcqp_res_kernel<4, 1, false, R>;
    if (minw == 2) return (const void *)mpcqp_res_kernel<4, 2, false, R>;
  }
  return nullptr;
}
