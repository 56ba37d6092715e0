// kernels_all.hpp -- every device header of libmpcqp.so in the order they build on each other.  The library is several translation units (one per
// kernel family and entry, so that the instances compile in parallel: a single unit took six minutes); each includes this file and
// instantiates its own kernels, and mpcqp.hip -- the host side of the C ABI -- gets their addresses through kernel_table.hpp.
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mpcqp.h"
#include "plan.hpp"
#include "common.hpp"

using namespace mpcqp;
#include "kernels_common.hpp"
#include "kernel_stream.hpp"
#include "kernel_onchip.hpp"
#include "kernel_resident.hpp"
#include "kernel_oc_split.hpp"
#include "kernel_table.hpp"
