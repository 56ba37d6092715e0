// k_oc_admm.hip -- instances of mpcqp_oc_admm_kernel<..., RF = 0> (kernel_oc_split.hpp): CuCaQP::solve's half of the on-chip mode;
// an adaptive-rho step leaves the kernel for the set-up kernel's resume mode
#include "kernels_all.hpp"
MPCQP_HIDDEN const void *mpcqp_kernel_oc_admm(int nw, int ng, int nh) {
  constexpr int RF = 0;
  if (nw == 4 && ng == OC_NG) {
    if (nh == OC_NH) return (const void *)mpcqp_oc_admm_kernel<4, OC_NG, OC_NH, RF>;
    if (nh == 0) return (const void *)mpcqp_oc_admm_kernel<4, OC_NG, 0, RF>;
    return nullptr;
  }
  if (nw == 8) {
    if (ng == OC8_INST[0].ng && nh == OC8_INST[0].nh) return (const void *)mpcqp_oc_admm_kernel<8, OC8_INST[0].ng, OC8_INST[0].nh, RF>;
    if (ng == OC8_INST[1].ng && nh == OC8_INST[1].nh) return (const void *)mpcqp_oc_admm_kernel<8, OC8_INST[1].ng, OC8_INST[1].nh, RF>;
    if (ng == OC8_INST[0].ng && nh == 0) return (const void *)mpcqp_oc_admm_kernel<8, OC8_INST[0].ng, 0, RF>;      // no arrow head (the reduced form's long chains)
    if (ng == OC8_INST[1].ng && nh == 0) return (const void *)mpcqp_oc_admm_kernel<8, OC8_INST[1].ng, 0, RF>;
  }
  return nullptr;
}
