// k_oc_setup.hip -- instances of mpcqp_oc_setup_kernel (kernel_oc_split.hpp): CuCaQP::initSolver's half of the on-chip mode
#include "kernels_all.hpp"
MPCQP_HIDDEN const void *mpcqp_kernel_oc_setup(int nw, bool hub, bool reuse) {
  if (nw == 4) {
    if (hub) return reuse ? (const void *)mpcqp_oc_setup_kernel<4, true, true> : (const void *)mpcqp_oc_setup_kernel<4, false, true>;
    return reuse ? (const void *)mpcqp_oc_setup_kernel<4, true, false> : (const void *)mpcqp_oc_setup_kernel<4, false, false>;
  }
  return nullptr;
}
