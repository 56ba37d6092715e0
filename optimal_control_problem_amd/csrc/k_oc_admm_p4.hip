// k_oc_admm_p4.hip -- the four-wave iteration kernel for plans with two twisted pairs of chains (the dissected order, plan.hpp build_plan ordering 4): its own
// instances, so that the text of the one-pair instances (the north-star size's kernel) does not change
#include "kernels_all.hpp"
MPCQP_HIDDEN const void *mpcqp_kernel_oc_admm_p4(int rf) {
  return rf ? (const void *)mpcqp_oc_admm_kernel<4, OC_NG, OC_NH, 1, false, true> : (const void *)mpcqp_oc_admm_kernel<4, OC_NG, OC_NH, 0, false, true>;
}
