// kernels.hpp -- hand-written gfx950 kernels of the RCG abundance-estimation loop.
//
// Formulation: the log-responsibilities keep the closed form
//     gamma(g, j) = a * L(g, j) + u_g - lse_j ,   lse_j = logsumexp_g(a*L(g,j) + u_g)
// through every operation of rcgpar's RCG loop (reference call site
// src/mSWEEP.cpp:194-198), so the G x E matrices gamma / step / oldstep of the reference
// never exist.  One iteration = one "pass A" sweep (|g|^2 of the natural gradient) and
// one "pass B" sweep (per-EC softmax, column sums N_g, ELBO terms) over the read-only
// likelihood, plus O(G) kernels in between.
#pragma once
#include "common.hpp"
#include "device_util.hpp"
#include "sell.hpp"
#include "state_kernels.hpp"
#include "sweep_kernels.hpp"
#include "dense_kernels.hpp"
#include "gamma_kernels.hpp"
