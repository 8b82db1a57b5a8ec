// likelihood_kernels.hpp -- device build of the CSR-of-ECs likelihood (K0-K2).
#pragma once
#include "common.hpp"
namespace msw {}
