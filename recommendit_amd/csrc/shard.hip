// Row-sharded embedding tables across ranks (BASELINE cfg4: 100M x 10M rows over 8 GPUs).
// The reference has one device and one table (src/training/train_embeddings.py:102-109, src/models/two_tower.py:27,54);
// with the table cut by rows over W ranks the gather at two_tower.py:40 / :69 becomes: route the ids of a batch to the
// ranks that own them (this file), exchange ids -> rows -> row gradients with RCCL all-to-alls (trainer.py), and gather /
// scatter locally.  Global row g >= 1 lives on rank (g-1) % W at local row (g-1) / W + 1 (cyclic: a Zipf-skewed id
// stream spreads evenly); local row 0 is an unused padding row on every rank.
#include "common.h"
#include "recommendit_hip.h"

#include <string.h>

#include <algorithm>

#include <rocprim/rocprim.hpp>

namespace {

__global__ void route_keys_kernel(const int64_t* __restrict__ ids, int64_t B, int W, int* keys, int* vals, int* err_flag) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  const int64_t g = ids[i];
  // the padding id 0 goes to rank 0's padding slot (local row 0: read like any row, gradient dropped by the row-sparse
  // optimiser -- nn.Embedding(padding_idx=0) semantics, the same as the replicated path); a negative id is an error
  if (g < 0 && err_flag) *err_flag = 1;
  keys[i] = g < 1 ? 0 : (int)((g - 1) % W);
  vals[i] = (int)i;
}

// sorted slot j holds pair perm[j]; pos[pair] = j; sorted_local[j] = owner-local row; counts[c] = #requests to rank c
__global__ void route_finish_kernel(const int64_t* __restrict__ ids, int64_t B, int W, const int* __restrict__ keys_sorted,
                                    const int* __restrict__ perm32, int64_t* sorted_local, int64_t* perm, int64_t* pos,
                                    int64_t* counts) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < B) {
    const int i = perm32[j];
    const int64_t g = ids[i];
    sorted_local[j] = g < 1 ? 0 : (g - 1) / W + 1;
    perm[j] = i;
    pos[i] = j;
  }
  if (j <= W && j > 0) {  // thread c+1 writes counts[c] = lower_bound(c+1) - lower_bound(c)
    const int c = (int)j - 1;
    int64_t lo0 = 0, hi0 = B, lo1 = 0, hi1 = B;
    while (lo0 < hi0) { const int64_t m = (lo0 + hi0) >> 1; if (keys_sorted[m] < c) lo0 = m + 1; else hi0 = m; }
    while (lo1 < hi1) { const int64_t m = (lo1 + hi1) >> 1; if (keys_sorted[m] < c + 1) lo1 = m + 1; else hi1 = m; }
    counts[c] = lo1 - lo0;
  }
}

// fixed-capacity form: `cap` send slots per owner.  Sorted slot j (owner c = keys_sorted[j], position j - first(c)) goes to
// slot c*cap + position; slot_ids (zeroed by the caller: 0 = the padding row) receives the owner-local row,
// slot_of_pair the slot of each pair.  More than cap requests for one owner: error bit 2, the pair aliases slot c*cap.
__global__ void route_fixed_kernel(const int64_t* __restrict__ ids, int64_t B, int W, int64_t cap,
                                   const int* __restrict__ keys_sorted, const int* __restrict__ perm32, int64_t* slot_ids,
                                   int64_t* slot_of_pair, int64_t* counts, int* err_flag) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < B) {
    const int c = keys_sorted[j];
    int64_t lo = 0, hi = j;   // first sorted slot of owner c (keys_sorted[j] == c, so the answer is <= j)
    while (lo < hi) { const int64_t m = (lo + hi) >> 1; if (keys_sorted[m] < c) lo = m + 1; else hi = m; }
    const int64_t posn = j - lo;
    const int i = perm32[j];
    const int64_t g = ids[i];
    if (posn < cap) {
      slot_ids[(int64_t)c * cap + posn] = g < 1 ? 0 : (g - 1) / W + 1;
      slot_of_pair[i] = (int64_t)c * cap + posn;
    } else {
      if (err_flag) atomicOr(err_flag, 2);
      slot_of_pair[i] = (int64_t)c * cap;
    }
  }
  if (j <= W && j > 0) {
    const int c = (int)j - 1;
    int64_t lo0 = 0, hi0 = B, lo1 = 0, hi1 = B;
    while (lo0 < hi0) { const int64_t m = (lo0 + hi0) >> 1; if (keys_sorted[m] < c) lo0 = m + 1; else hi0 = m; }
    while (lo1 < hi1) { const int64_t m = (lo1 + hi1) >> 1; if (keys_sorted[m] < c + 1) lo1 = m + 1; else hi1 = m; }
    counts[c] = (lo1 - lo0) < cap ? (lo1 - lo0) : cap;
  }
}

// out[slot[i]] = src[i] (the slots of one step are distinct)
__global__ __launch_bounds__(256) void scatter_rows4_kernel(const float* __restrict__ src, const int64_t* __restrict__ slot,
                                                            int64_t n, int64_t n_slots, int d4, float* out, int* err_flag) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * d4) return;
  const int64_t r = idx / d4;
  const int64_t sl = slot[r];
  if (sl < 0 || sl >= n_slots) { if (err_flag) *err_flag = 1; return; }
  reinterpret_cast<f32x4*>(out)[sl * d4 + (idx % d4)] = reinterpret_cast<const f32x4*>(src)[idx];
}

__global__ __launch_bounds__(256) void gather_rows4_kernel(const float* __restrict__ table, int64_t n_rows,
                                                           const int64_t* __restrict__ ids, int64_t n, int d4, float* out,
                                                           int* err_flag) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * d4) return;
  const int64_t r = idx / d4;
  int64_t id = ids[r];
  if (id < 0 || id >= n_rows) { if (err_flag) *err_flag = 1; id = 0; }
  reinterpret_cast<f32x4*>(out)[idx] = reinterpret_cast<const f32x4*>(table)[id * d4 + (idx % d4)];
}

size_t a256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

extern "C" int64_t rihip_route_workspace_bytes(int64_t B) {
  if (B <= 0) return -1;
  size_t sort_bytes = 0;
  if (rocprim::radix_sort_pairs(nullptr, sort_bytes, (const int*)nullptr, (int*)nullptr, (const int*)nullptr, (int*)nullptr,
                                (size_t)B) != hipSuccess)
    return -1;
  return (int64_t)(4 * a256(sizeof(int) * (size_t)B) + a256(sort_bytes) + 256);
}

extern "C" int rihip_route_rows(const int64_t* ids, int64_t B, int world, int64_t* sorted_local, int64_t* perm,
                                int64_t* pos, int64_t* counts, int* err_flag, void* workspace, int64_t workspace_bytes,
                                void* stream) {
  RIHIP_REQUIRE(ids && sorted_local && perm && pos && counts && workspace && B > 0, RIHIP_ERR_ARG, "route_rows: bad arguments");
  RIHIP_REQUIRE(world >= 1 && world <= 1024, RIHIP_ERR_ARG, "route_rows: world=%d", world);
  RIHIP_REQUIRE(B < (1ll << 31) && B >= world, RIHIP_ERR_ARG, "route_rows: B=%lld", (long long)B);
  RIHIP_REQUIRE(workspace_bytes >= rihip_route_workspace_bytes(B), RIHIP_ERR_ARG, "route_rows: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  char* b = (char*)workspace;
  const size_t nb = a256(sizeof(int) * (size_t)B);
  int* keys = (int*)b; int* vals = (int*)(b + nb); int* keys_s = (int*)(b + 2 * nb); int* perm32 = (int*)(b + 3 * nb);
  void* temp = b + 4 * nb;
  size_t tb = (size_t)workspace_bytes - 4 * nb;
  const unsigned g = (unsigned)((B + 255) / 256);
  hipLaunchKernelGGL(route_keys_kernel, dim3(g), dim3(256), 0, st, ids, B, world, keys, vals, err_flag);
  RIHIP_CHECK_LAUNCH();
  unsigned end_bit = 1;
  while ((1 << end_bit) < world) ++end_bit;
  RIHIP_CHECK_HIP(rocprim::radix_sort_pairs(temp, tb, (const int*)keys, keys_s, (const int*)vals, perm32, (size_t)B, 0, end_bit, st));
  // thread c+1 writes counts[c]: the grid must hold at least world+1 threads (B == world, B % 256 == 0 otherwise loses
  // the last count)
  const unsigned gf = (unsigned)((std::max<int64_t>(B, world + 1) + 255) / 256);
  hipLaunchKernelGGL(route_finish_kernel, dim3(gf), dim3(256), 0, st, ids, B, world, keys_s, perm32, sorted_local, perm, pos, counts);
  RIHIP_CHECK_LAUNCH();
  return RIHIP_OK;
}

extern "C" int rihip_route_rows_fixed(const int64_t* ids, int64_t B, int world, int64_t cap, int64_t* slot_ids,
                                      int64_t* slot_of_pair, int64_t* counts, int* err_flag, void* workspace,
                                      int64_t workspace_bytes, void* stream) {
  RIHIP_REQUIRE(ids && slot_ids && slot_of_pair && counts && workspace && B > 0 && cap > 0, RIHIP_ERR_ARG,
                "route_rows_fixed: bad arguments");
  RIHIP_REQUIRE(world >= 1 && world <= 1024, RIHIP_ERR_ARG, "route_rows_fixed: world=%d", world);
  RIHIP_REQUIRE(B < (1ll << 31), RIHIP_ERR_ARG, "route_rows_fixed: B=%lld", (long long)B);
  RIHIP_REQUIRE(workspace_bytes >= rihip_route_workspace_bytes(B), RIHIP_ERR_ARG, "route_rows_fixed: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  char* b = (char*)workspace;
  const size_t nb = a256(sizeof(int) * (size_t)B);
  int* keys = (int*)b; int* vals = (int*)(b + nb); int* keys_s = (int*)(b + 2 * nb); int* perm32 = (int*)(b + 3 * nb);
  void* temp = b + 4 * nb;
  size_t tb = (size_t)workspace_bytes - 4 * nb;
  RIHIP_CHECK_HIP(hipMemsetAsync(slot_ids, 0, sizeof(int64_t) * (size_t)world * (size_t)cap, st));
  hipLaunchKernelGGL(route_keys_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, st, ids, B, world, keys, vals, err_flag);
  RIHIP_CHECK_LAUNCH();
  unsigned end_bit = 1;
  while ((1 << end_bit) < world) ++end_bit;
  RIHIP_CHECK_HIP(rocprim::radix_sort_pairs(temp, tb, (const int*)keys, keys_s, (const int*)vals, perm32, (size_t)B, 0, end_bit, st));
  const unsigned gf = (unsigned)((std::max<int64_t>(B, world + 1) + 255) / 256);
  hipLaunchKernelGGL(route_fixed_kernel, dim3(gf), dim3(256), 0, st, ids, B, world, cap, keys_s, perm32, slot_ids, slot_of_pair,
                     counts, err_flag);
  RIHIP_CHECK_LAUNCH();
  return RIHIP_OK;
}

extern "C" int rihip_scatter_rows(const float* src, const int64_t* slot, int64_t n, int64_t n_slots, int d, float* out,
                                  int* err_flag, void* stream) {
  RIHIP_REQUIRE(src && slot && out && n >= 0 && n_slots > 0 && d > 0 && d % 4 == 0, RIHIP_ERR_ARG, "scatter_rows: bad arguments");
  RIHIP_REQUIRE(((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(out)) & 15) == 0, RIHIP_ERR_ARG,
                "scatter_rows: pointers must be 16-byte aligned");
  if (n == 0) return RIHIP_OK;
  const int64_t tot = n * (d / 4);
  hipLaunchKernelGGL(scatter_rows4_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, slot, n,
                     n_slots, d / 4, out, err_flag);
  RIHIP_CHECK_LAUNCH();
  return RIHIP_OK;
}

extern "C" int rihip_gather_rows(const float* table, int64_t n_rows, const int64_t* ids, int64_t n, int d, float* out,
                                 int* err_flag, void* stream) {
  RIHIP_REQUIRE(table && ids && out && n >= 0 && n_rows > 0 && d > 0 && d % 4 == 0, RIHIP_ERR_ARG, "gather_rows: bad arguments");
  RIHIP_REQUIRE(((reinterpret_cast<uintptr_t>(table) | reinterpret_cast<uintptr_t>(out)) & 15) == 0, RIHIP_ERR_ARG,
                "gather_rows: pointers must be 16-byte aligned");
  if (n == 0) return RIHIP_OK;
  const int64_t tot = n * (d / 4);
  hipLaunchKernelGGL(gather_rows4_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, table, n_rows,
                     ids, n, d / 4, out, err_flag);
  RIHIP_CHECK_LAUNCH();
  return RIHIP_OK;
}
