// bootstrap_kernels.hpp -- device bootstrap resampling (MT19937-64 stream + discrete draw).
#pragma once
#include "common.hpp"
namespace msw {}
