// compress_kernels.hpp -- a dense likelihood that is "one background value + few listed cells" (what
// LL_WOR21::fill_ll_mat writes: log(zi) wherever an EC does not hit a group, a lookup-table value
// where it does -- include/Likelihood.hpp:92-107,176-185) is turned into the CSR-of-ECs + value
// table the SELL sweeps run on, bit for bit the same numbers.  The staged matrix is [G][E], rows =
// groups: thread j walks EC j down the rows, so every load is coalesced across the wavefront.
#pragma once
#include "common.hpp"

namespace msw {

__global__ __launch_bounds__(256) void k_dense_count(const uint64_t *bits, uint32_t G, uint32_t E, uint64_t bg,
                                                    uint32_t *cnt) {
  for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j <= E; j += gridDim.x * blockDim.x) {
    uint32_t c = 0;
    if (j < E)
      for (uint32_t g = 0; g < G; ++g) c += bits[(size_t)g * E + j] != bg;
    cnt[j] = c;  // entry E = 0: the scan turns the array into row pointers
  }
}

__global__ __launch_bounds__(256) void k_dense_extract(const uint64_t *bits, uint32_t G, uint32_t E, uint64_t bg,
                                                      const uint32_t *rowptr, uint32_t *grp, uint64_t *val,
                                                      uint32_t *pos) {
  for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < E; j += gridDim.x * blockDim.x) {
    uint32_t p = rowptr[j];
    for (uint32_t g = 0; g < G; ++g) {
      const uint64_t v = bits[(size_t)g * E + j];
      if (v != bg) {
        grp[p] = g;
        val[p] = v;
        pos[p] = p;
        ++p;
      }
    }
  }
}

// sorted values -> 1 where a new value starts (entry 0: 0), so that the inclusive sum is the value's rank
__global__ __launch_bounds__(256) void k_value_heads(const uint64_t *sorted, uint64_t n, uint32_t *head) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    head[i] = i > 0 && sorted[i] != sorted[i - 1];
}

// ex = exclusive scan of the heads: rank = ex + head; the first cell of every value writes the table
__global__ __launch_bounds__(256) void k_value_ranks(const uint64_t *sorted, const uint32_t *pos_sorted,
                                                    const uint32_t *ex, uint64_t n, uint32_t *idx, uint64_t *lut) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const bool head = i > 0 && sorted[i] != sorted[i - 1];
    const uint32_t rank = ex[i] + (head ? 1u : 0u);
    idx[pos_sorted[i]] = rank;
    if (head || i == 0) lut[rank] = sorted[i];
  }
}

}  // namespace msw
