// k_oc_admm_tl.hip -- instances of mpcqp_oc_admm_kernel<..., RF = 0, TL = true> (kernel_oc_split.hpp): the iteration kernel with its two sweeps on dense
// tiles of A (one copy for both products, multiplied on the vector ALUs) -- an experiment, MPCQP_VTILES=1
#include "kernels_all.hpp"
MPCQP_HIDDEN const void *mpcqp_kernel_oc_admm_tl(int nw, int ng, int nh) {
  if (nw == 4 && ng == OC_NG && nh == OC_NH) return (const void *)mpcqp_oc_admm_kernel<4, OC_NG, OC_NH, 0, true>;
  if (nw == 8 && ng == OC8_INST[1].ng && nh == OC8_INST[1].nh) return (const void *)mpcqp_oc_admm_kernel<8, OC8_INST[1].ng, OC8_INST[1].nh, 0, true>;
  return nullptr;
}
