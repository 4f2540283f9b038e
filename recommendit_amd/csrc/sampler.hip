// Device-side negative sampler -- replaces UserItemDataset._sample_negative (reference
// src/training/train_embeddings.py:58-63: uniform draw from the catalogue, re-drawn while the item is in the
// user's RATED set) for a whole batch in one launch.  The rated set is the sorted array of keys
// user*M + item (every rating, any value); membership = binary search.  Draws come from a counter-based
// generator keyed by (seed, sample index, attempt), so a batch is reproducible and order-independent.
// HBM/latency-bound integer work: ~log2(1e6)=20 dependent 8-byte reads per attempt, one thread per sample.
#include "common.h"
#include "recommendit_hip.h"

namespace {

__device__ __forceinline__ bool rated_contains(const int64_t* __restrict__ keys, int64_t n, int64_t key) {
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    const int64_t v = keys[mid];
    if (v < key) lo = mid + 1;
    else hi = mid;
  }
  return lo < n && keys[lo] == key;
}

__global__ __launch_bounds__(256) void sample_negatives_kernel(const int64_t* __restrict__ users, int64_t n,
                                                               const int64_t* __restrict__ catalog, int64_t n_catalog,
                                                               const int64_t* __restrict__ rated_keys, int64_t n_rated,
                                                               int64_t M, uint64_t seed, int max_attempts,
                                                               int64_t* neg, int* gave_up) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int64_t u = users[i];
  const uint64_t base = rihip_splitmix64(seed ^ rihip_splitmix64((uint64_t)i));
  int64_t pick = catalog[0];
  bool ok = false;
  for (int a = 0; a < max_attempts && !ok; ++a) {
    const uint64_t r = rihip_splitmix64(base + (uint64_t)a);
    // unbiased enough for catalogue sizes << 2^32: multiply-high of the top 32 bits
    pick = catalog[(int64_t)(((r >> 32) * (uint64_t)n_catalog) >> 32)];
    ok = !rated_contains(rated_keys, n_rated, u * M + pick);
  }
  neg[i] = pick;
  if (!ok && gave_up) atomicAdd(gave_up, 1);  // user rated (almost) the whole catalogue: the reference would spin
}

}  // namespace

extern "C" int rihip_sample_negatives(const int64_t* users, int64_t n, const int64_t* catalog, int64_t n_catalog,
                                      const int64_t* rated_keys, int64_t n_rated, int64_t key_stride, uint64_t seed,
                                      int max_attempts, int64_t* neg_out, int* gave_up, void* stream) {
  RIHIP_REQUIRE(users && catalog && rated_keys && neg_out, RIHIP_ERR_ARG, "sample_negatives: null pointer");
  RIHIP_REQUIRE(n >= 0 && n_catalog > 0 && n_catalog < (1ll << 32) && n_rated >= 0 && key_stride > 0 && max_attempts > 0,
                RIHIP_ERR_ARG, "sample_negatives: bad sizes");
  if (n == 0) return RIHIP_OK;
  hipLaunchKernelGGL(sample_negatives_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     users, n, catalog, n_catalog, rated_keys, n_rated, key_stride, seed, max_attempts, neg_out,
                     gave_up);
  RIHIP_CHECK_LAUNCH();
  return RIHIP_OK;
}
