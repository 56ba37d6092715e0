// common.hpp -- helpers shared by the translation units of libmpcqp.so (not part of the ABI)
#pragma once
#include <string>
#include <hip/hip_runtime.h>
#include "../../include/mpcqp.h"

// records the message mpcqp_strerror() appends, returns `code` (defined in mpcqp.hip)
__attribute__((visibility("hidden"))) int mpcqp_set_error(int code, const std::string &msg);
// picks / validates the device like mpcqp_create: ordinal < 0 = current device; must be gfx950
__attribute__((visibility("hidden"))) int mpcqp_pick_device(int requested, int *device);

#define MPCQP_HIPCHK(expr)                                                                                              \
  do { hipError_t e_ = (expr); if (e_ != hipSuccess) return mpcqp_set_error(MPCQP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)
// makes work queued on `s` from here on wait for the handle's last solve if that ran on another stream (a caller that rewrites arrays the
// handle borrows -- mpcqp_stageqp_update's packed P and A -- calls this first); defined in mpcqp.hip
__attribute__((visibility("hidden"))) int mpcqp_order_after_last_solve(mpcqp_handle *h, hipStream_t s);
