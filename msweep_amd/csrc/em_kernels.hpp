// em_kernels.hpp -- EM variant of the sweeps (--algorithm emgpu).
#pragma once
#include "common.hpp"
namespace msw {}
