// IVF-Flat (inner product) build side of the index -- replaces faiss.IndexIVFFlat.train/add and
// faiss.write_index/read_index behind FAISSIndex (reference src/models/faiss_index.py:68-74, :164, :196).
//
//   k-means (Lloyd, IP assignment = the IndexFlatIP quantizer, mean update; empty lists keep their centroid):
//     assign   : ivf_assign_mfma_kernel (exact-f32 MFMA arg-max, lowest list id wins ties)
//     group    : stable radix sort of (list, row) pairs (rocPRIM) + per-list lower bounds
//     update   : per (list, split) partial sums in a fixed order, then one combine -> bitwise reproducible,
//                no float atomics, and no LDS-resident copy of the centroids (nlist is bounded only by the
//                2048-bit probe set of the search)
//   layout   : lists contiguous, each padded to a 64-row granule with zero rows (row id -1), rows ascending
//              inside a list; permuted on the device.
// The trained state can be read back / injected (rihip_ip_index_get_ivf / _set_ivf) so that the oracle tests and the
// FAISS file reader/writer (recommendit_amd/faiss_io.py) work from the very same centroids and list membership.
#include "common.h"
#include "recommendit_hip.h"
#include "ip_index.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include <rocprim/rocprim.hpp>

using namespace rihip_index;

namespace {

// MFMA form of the assignment (d in {32,64,128}): 4 waves x 32 register-stationary rows per workgroup, centroid tiles
// of 32 through LDS, S[centroid][row] on exact-f32 MFMA, arg-max over the accumulator rows (lowest index wins ties).
template <int D>
__global__ __launch_bounds__(256, 2) void ivf_assign_mfma_kernel(const float* __restrict__ X, int64_t N,
                                                                 const float* __restrict__ C, int nlist, int* assign) {
  constexpr int LDC = D + 4, KB = D / 8;
  constexpr int NV = (32 * (D / 4) + 255) / 256;
  __shared__ __attribute__((aligned(16))) float Cs[32 * LDC];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int r31 = lane & 31, hh = lane >> 5;
  const int ntile = (nlist + 31) / 32;
  for (int64_t blk = blockIdx.x; blk * 128 < N; blk += gridDim.x) {
    const int64_t row = blk * 128 + w * 32 + r31;
    const int64_t rowc = row < N ? row : N - 1;
    f32x4 xr[KB];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) xr[kb] = *reinterpret_cast<const f32x4*>(&X[rowc * D + kb * 8 + 4 * hh]);
    float best = -INFINITY;
    int bi = 0;
    for (int t = 0; t < ntile; ++t) {
      __syncthreads();  // previous tile fully consumed
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int idx = tid + i * 256;
        const int r = idx / (D / 4), c4 = idx % (D / 4);
        if (idx < 32 * (D / 4)) {
          const int c = t * 32 + r;
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
          if (c < nlist) v = reinterpret_cast<const f32x4*>(C + (size_t)c * D)[c4];
          *reinterpret_cast<f32x4*>(&Cs[r * LDC + c4 * 4]) = v;
        }
      }
      __syncthreads();
      f32x16 acc = zero16();
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        const f32x4 av = *reinterpret_cast<const f32x4*>(&Cs[r31 * LDC + kb * 8 + 4 * hh]);
        acc = mfma32(av.x, xr[kb].x, acc);
        acc = mfma32(av.y, xr[kb].y, acc);
        acc = mfma32(av.z, xr[kb].z, acc);
        acc = mfma32(av.w, xr[kb].w, acc);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {  // acc_row(r, lane) grows with r: '>' keeps the lowest centroid on ties
        const int c = t * 32 + acc_row(r, lane);
        if (c < nlist && acc[r] > best) { best = acc[r]; bi = c; }
      }
    }
    const float ob = __shfl_xor(best, 32, 64);
    const int oi = __shfl_xor(bi, 32, 64);
    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    if (hh == 0 && row < N) assign[row] = bi;
  }
}

__global__ void iota_int_kernel(int* p, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = (int)i;
}

// off[c] = first position of list c in the sorted keys (lower bound); off[nlist] = N
__global__ void list_bounds_kernel(const int* __restrict__ keys, int64_t N, int nlist, int* off) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c > nlist) return;
  int64_t lo = 0, hi = N;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (keys[mid] < c) lo = mid + 1; else hi = mid;
  }
  off[c] = (int)lo;
}

// partial sum of the member rows of list blockIdx.x, split blockIdx.y, in a fixed order
__global__ __launch_bounds__(256) void ivf_listsum_kernel(const float* __restrict__ X, int d, const int* __restrict__ rows,
                                                          const int* __restrict__ off, float* part) {
  __shared__ float red[256];
  const int c = blockIdx.x, s = blockIdx.y, SPL = gridDim.y, tid = threadIdx.x;
  const int RL = 256 / d, col = tid % d, rl = tid / d;
  const int b0 = off[c], n = off[c + 1] - b0;
  const int per = (n + SPL - 1) / SPL;
  const int b = b0 + s * per;
  const int e = (b + per < b0 + n) ? b + per : b0 + n;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int i = b + rl;
  for (; i + 3 * RL < e; i += 4 * RL) {
    a0 += X[(size_t)rows[i] * d + col];
    a1 += X[(size_t)rows[i + RL] * d + col];
    a2 += X[(size_t)rows[i + 2 * RL] * d + col];
    a3 += X[(size_t)rows[i + 3 * RL] * d + col];
  }
  for (; i < e; i += RL) a0 += X[(size_t)rows[i] * d + col];
  red[tid] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (rl == 0) {
    float acc = red[col];
    for (int r = 1; r < RL; ++r) acc += red[r * d + col];
    part[((size_t)c * SPL + s) * d + col] = acc;
  }
}

__global__ void ivf_mean_kernel(const float* __restrict__ part, int SPL, const int* __restrict__ off, int nlist, int d,
                                float* C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nlist * d) return;
  const int c = i / d, col = i % d;
  const int n = off[c + 1] - off[c];
  if (n <= 0) return;  // empty list keeps its previous centroid
  float s = 0.f;
  for (int k = 0; k < SPL; ++k) s += part[((size_t)c * SPL + k) * d + col];
  C[i] = s / (float)n;
}

// physical layout: row p of the padded, list-contiguous corpus <- member j = p - off_p[list] of its list
__global__ __launch_bounds__(256) void ivf_permute_kernel(const float* __restrict__ X, int d, const int* __restrict__ rows,
                                                          const int* __restrict__ off, const int64_t* __restrict__ off_p,
                                                          const int* __restrict__ tile_list, int64_t Np, float* Xn,
                                                          int64_t* row_ids) {
  const int d4 = d / 4;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= Np * d4) return;
  const int64_t p = idx / d4;
  const int c4 = (int)(idx % d4);
  const int c = tile_list[p / TR];
  const int64_t j = p - off_p[c];
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  int64_t rid = -1;
  if (j < off[c + 1] - off[c]) {
    rid = rows[off[c] + j];
    v = reinterpret_cast<const f32x4*>(X + (size_t)rid * d)[c4];
  }
  reinterpret_cast<f32x4*>(Xn + (size_t)p * d)[c4] = v;
  if (c4 == 0) row_ids[p] = rid;
}

// original-order rows out of the list-ordered corpus
__global__ void ivf_unpermute_kernel(const float* __restrict__ Xn, int d, const int64_t* __restrict__ row_ids, int64_t Np,
                                     float* X) {
  const int d4 = d / 4;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= Np * d4) return;
  const int64_t p = idx / d4;
  const int64_t rid = row_ids[p];
  if (rid < 0) return;
  reinterpret_cast<f32x4*>(X + (size_t)rid * d)[idx % d4] = reinterpret_cast<const f32x4*>(Xn + (size_t)p * d)[idx % d4];
}

int launch_assign(int d, const float* X, int64_t N, const float* C, int nlist, int* assign, hipStream_t st) {
  const int64_t nblk = (N + 127) / 128;
  const dim3 mg((unsigned)(nblk < 2048 ? nblk : 2048));
  if (d == 128) hipLaunchKernelGGL((ivf_assign_mfma_kernel<128>), mg, dim3(256), 0, st, X, N, C, nlist, assign);
  else if (d == 64) hipLaunchKernelGGL((ivf_assign_mfma_kernel<64>), mg, dim3(256), 0, st, X, N, C, nlist, assign);
  else if (d == 32) hipLaunchKernelGGL((ivf_assign_mfma_kernel<32>), mg, dim3(256), 0, st, X, N, C, nlist, assign);
  else { rihip_set_error("ip_index: unsupported embed_dim=%d (32/64/128)", d); return RIHIP_ERR_SHAPE; }
  RIHIP_CHECK_LAUNCH();
  return RIHIP_OK;
}

struct Grouper {  // (list,row) pairs sorted by list; rows ascending inside a list (the sort is stable)
  int* keys = nullptr; int* rows_in = nullptr; int* rows = nullptr; int* off = nullptr; void* temp = nullptr;
  size_t temp_bytes = 0;
  int64_t N = 0; int nlist = 0; int end_bit = 1;
  int init(int64_t N_, int nlist_, hipStream_t st) {
    N = N_; nlist = nlist_;
    end_bit = 1;
    while ((1 << end_bit) < nlist) ++end_bit;
    HIPCHK(hipMalloc((void**)&keys, sizeof(int) * N));
    HIPCHK(hipMalloc((void**)&rows_in, sizeof(int) * N));
    HIPCHK(hipMalloc((void**)&rows, sizeof(int) * N));
    HIPCHK(hipMalloc((void**)&off, sizeof(int) * (nlist + 1)));
    HIPCHK(rocprim::radix_sort_pairs(nullptr, temp_bytes, (const int*)nullptr, (int*)nullptr, (const int*)nullptr,
                                     (int*)nullptr, (size_t)N, 0, end_bit, st));
    HIPCHK(hipMalloc(&temp, temp_bytes ? temp_bytes : 16));
    hipLaunchKernelGGL(iota_int_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, rows_in, N);
    RIHIP_CHECK_LAUNCH();
    return RIHIP_OK;
  }
  int group(const int* assign, hipStream_t st) {
    size_t tb = temp_bytes;
    HIPCHK(rocprim::radix_sort_pairs(temp, tb, assign, keys, (const int*)rows_in, rows, (size_t)N, 0, end_bit, st));
    hipLaunchKernelGGL(list_bounds_kernel, dim3((unsigned)((nlist + 1 + 255) / 256)), dim3(256), 0, st, keys, N, nlist, off);
    RIHIP_CHECK_LAUNCH();
    return RIHIP_OK;
  }
  void release() { hipFree(keys); hipFree(rows_in); hipFree(rows); hipFree(off); hipFree(temp); keys = rows_in = rows = off = nullptr; temp = nullptr; }
};

// Lloyd iterations on device.  C: [nlist,d] device, initial centroids in, final centroids out; assign: [N] device, the
// assignment to the FINAL centroids; g holds that assignment grouped.
int kmeans(IpIndex* h, int nlist, int n_iter, float* C, int* assign, Grouper& g, hipStream_t st) {
  const int d = h->d;
  const int SPL = 16;
  float* part = nullptr;
  HIPCHK(hipMalloc((void**)&part, sizeof(float) * (size_t)nlist * SPL * d));
  int rc = RIHIP_OK;
  for (int it = 0; it <= n_iter && rc == RIHIP_OK; ++it) {
    rc = launch_assign(d, h->X, h->N, C, nlist, assign, st);
    if (rc == RIHIP_OK) rc = g.group(assign, st);
    if (it == n_iter || rc != RIHIP_OK) break;
    hipLaunchKernelGGL(ivf_listsum_kernel, dim3(nlist, SPL), dim3(256), 0, st, h->X, d, g.rows, g.off, part);
    hipLaunchKernelGGL(ivf_mean_kernel, dim3((nlist * d + 255) / 256), dim3(256), 0, st, part, SPL, g.off, nlist, d, C);
    if (hipGetLastError() != hipSuccess) { rihip_set_error("ivf k-means launch failed"); rc = RIHIP_ERR_HIP; }
  }
  if (rc == RIHIP_OK && hipStreamSynchronize(st) != hipSuccess) { rihip_set_error("ivf k-means: stream error"); rc = RIHIP_ERR_HIP; }
  hipFree(part);
  return rc;
}

// lay the lists out (takes ownership of C on success): h->X flat [N,d] -> list-ordered padded [Np,d]
int build_lists(IpIndex* h, int nlist, float* C, Grouper& g, hipStream_t st) {
  const int d = h->d;
  const int64_t N = h->N;
  std::vector<int> off(nlist + 1);
  HIPCHK(hipMemcpyAsync(off.data(), g.off, sizeof(int) * (nlist + 1), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  RIHIP_REQUIRE(off[0] == 0 && off[nlist] == N, RIHIP_ERR_ARG, "ip_index: list assignment outside [0,%d)", nlist);
  std::vector<int64_t> off_p(nlist + 1, 0);
  h->list_len.assign(nlist, 0);
  for (int c = 0; c < nlist; ++c) {
    h->list_len[c] = off[c + 1] - off[c];
    off_p[c + 1] = off_p[c] + (h->list_len[c] + TR - 1) / TR * TR;
  }
  const int64_t Np = off_p[nlist] > 0 ? off_p[nlist] : TR;
  std::vector<int> tl(Np / TR, 0);
  for (int c = 0; c < nlist; ++c)
    for (int64_t t = off_p[c] / TR; t < off_p[c + 1] / TR; ++t) tl[t] = c;
  float* Xn = nullptr;
  int64_t* d_offp = nullptr;
  HIPCHK(hipMalloc((void**)&Xn, sizeof(float) * (size_t)Np * d));
  HIPCHK(hipMalloc((void**)&h->row_ids, sizeof(int64_t) * Np));
  HIPCHK(hipMalloc((void**)&h->tile_list, sizeof(int) * (Np / TR)));
  HIPCHK(hipMalloc((void**)&d_offp, sizeof(int64_t) * (nlist + 1)));
  HIPCHK(hipMemcpyAsync(h->tile_list, tl.data(), sizeof(int) * (Np / TR), hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_offp, off_p.data(), sizeof(int64_t) * (nlist + 1), hipMemcpyHostToDevice, st));
  const int64_t tot = Np * (d / 4);
  hipLaunchKernelGGL(ivf_permute_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, h->X, d, g.rows, g.off,
                     d_offp, h->tile_list, Np, Xn, h->row_ids);
  RIHIP_CHECK_LAUNCH();
  HIPCHK(hipStreamSynchronize(st));
  hipFree(d_offp);
  hipFree(h->X);
  hipFree(h->Xb);  // the IVF scan never reads the bf16 filter copy
  h->Xb = nullptr;
  h->X = Xn; h->Np = Np; h->nlist = nlist; h->ivf = true; h->C = C;
  rihip_bump_generation();
  return derive_ivf_aux(h, st);
}

int train_impl(IpIndex* h, int nlist, int n_iter, float* C /*device, initial centroids; owned*/, const int* assign_given,
               hipStream_t st) {
  int* assign = nullptr;
  Grouper g;
  int rc = RIHIP_OK;
  if (hipMalloc((void**)&assign, sizeof(int) * h->N) != hipSuccess) { rihip_set_error("ip_index: allocation failed"); rc = RIHIP_ERR_HIP; }
  if (rc == RIHIP_OK) rc = g.init(h->N, nlist, st);
  if (rc == RIHIP_OK) {
    if (assign_given) rc = g.group(assign_given, st);
    else rc = kmeans(h, nlist, n_iter, C, assign, g, st);
  }
  if (rc == RIHIP_OK) rc = build_lists(h, nlist, C, g, st);
  g.release();
  hipFree(assign);
  if (rc != RIHIP_OK) {  // leave a usable flat index behind
    hipFree(C);
    hipFree(h->row_ids); hipFree(h->tile_list); hipFree(h->list_poff); hipFree(h->list_len_dev);
    h->row_ids = nullptr; h->tile_list = nullptr; h->list_poff = nullptr; h->list_len_dev = nullptr; h->C = nullptr;
  }
  return rc;
}

int check_trainable(IpIndex* h, int nlist, const char* what) {
  RIHIP_REQUIRE(h && h->X && !h->ivf, RIHIP_ERR_STATE, "%s: needs a flat, non-empty index", what);
  RIHIP_REQUIRE(nlist >= 1 && nlist <= h->N, RIHIP_ERR_ARG, "%s: nlist=%d for N=%lld", what, nlist, (long long)h->N);
  RIHIP_REQUIRE(nlist <= NLIST_MAX, RIHIP_ERR_SHAPE, "%s: nlist=%d > %d unsupported (probe set is a 2048-bit mask)", what,
                nlist, NLIST_MAX);
  return RIHIP_OK;
}

}  // namespace

namespace rihip_index {
int derive_ivf_aux(IpIndex* h, hipStream_t st) {
  const int nlist = h->nlist;
  std::vector<int64_t> poff(nlist + 1, 0);
  std::vector<int> len(nlist, 0);
  for (int c = 0; c < nlist; ++c) {
    len[c] = (int)h->list_len[c];
    poff[c + 1] = poff[c] + (h->list_len[c] + TR - 1) / TR * TR;
  }
  hipFree(h->list_poff); hipFree(h->list_len_dev);
  h->list_poff = nullptr; h->list_len_dev = nullptr;
  HIPCHK(hipMalloc((void**)&h->list_poff, sizeof(int64_t) * (nlist + 1)));
  HIPCHK(hipMalloc((void**)&h->list_len_dev, sizeof(int) * (nlist > 0 ? nlist : 1)));
  HIPCHK(hipMemcpyAsync(h->list_poff, poff.data(), sizeof(int64_t) * (nlist + 1), hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(h->list_len_dev, len.data(), sizeof(int) * nlist, hipMemcpyHostToDevice, st));
  HIPCHK(hipStreamSynchronize(st));
  return RIHIP_OK;
}
}  // namespace rihip_index

namespace {
// host-side column padding / slicing between the caller's width du and the kernel width d
std::vector<float> pad_host(const float* src, int64_t n, int du, int d) {
  std::vector<float> out((size_t)n * d, 0.f);
  for (int64_t i = 0; i < n; ++i) memcpy(out.data() + (size_t)i * d, src + (size_t)i * du, sizeof(float) * du);
  return out;
}
void unpad_host(const float* src, int64_t n, int d, int du, float* dst) {
  for (int64_t i = 0; i < n; ++i) memmove(dst + (size_t)i * du, src + (size_t)i * d, sizeof(float) * du);
}
}  // namespace

// k-means from `nlist` distinct seed rows picked by a seeded generator over a strided lattice.  Deterministic given seed.
extern "C" int rihip_ip_index_train_ivf(void* handle, int nlist, int n_iter, uint64_t seed, void* stream) {
  IpIndex* h = (IpIndex*)handle;
  RCCHK(check_trainable(h, nlist, "ip_index_train_ivf"));
  RIHIP_REQUIRE(n_iter >= 0, RIHIP_ERR_ARG, "ip_index_train_ivf: n_iter=%d", n_iter);
  hipStream_t st = (hipStream_t)stream;
  const int d = h->d;
  float* C = nullptr;
  HIPCHK(hipMalloc((void**)&C, sizeof(float) * (size_t)nlist * d));
  uint64_t s = rihip_splitmix64(seed);
  const int64_t step = h->N / nlist;
  for (int c = 0; c < nlist; ++c) {
    s = rihip_splitmix64(s);
    const int64_t pick = (int64_t)c * step + (int64_t)(s % (uint64_t)step);
    if (hipMemcpyAsync(C + (size_t)c * d, h->X + (size_t)pick * d, sizeof(float) * d, hipMemcpyDeviceToDevice, st) != hipSuccess) {
      hipFree(C);
      rihip_set_error("ip_index_train_ivf: seed copy failed");
      return RIHIP_ERR_HIP;
    }
  }
  return train_impl(h, nlist, n_iter, C, nullptr, st);
}

// k-means from caller-supplied initial centroids (host [nlist,d]); n_iter = 0 partitions by those centroids as they are.
extern "C" int rihip_ip_index_train_ivf_from(void* handle, int nlist, int n_iter, const float* init_centroids, void* stream) {
  IpIndex* h = (IpIndex*)handle;
  RCCHK(check_trainable(h, nlist, "ip_index_train_ivf_from"));
  RIHIP_REQUIRE(init_centroids && n_iter >= 0, RIHIP_ERR_ARG, "ip_index_train_ivf_from: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  float* C = nullptr;
  HIPCHK(hipMalloc((void**)&C, sizeof(float) * (size_t)nlist * h->d));
  const std::vector<float> cp = pad_host(init_centroids, nlist, h->du, h->d);
  if (hipMemcpyAsync(C, cp.data(), sizeof(float) * (size_t)nlist * h->d, hipMemcpyHostToDevice, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess) {
    hipFree(C);
    rihip_set_error("ip_index_train_ivf_from: centroid upload failed");
    return RIHIP_ERR_HIP;
  }
  return train_impl(h, nlist, n_iter, C, nullptr, st);
}

// inject a trained partition: centroids (host [nlist,d]) and the list of every row (host int32 [N]); what a FAISS
// IndexIVFFlat file holds (recommendit_amd/faiss_io.py)
extern "C" int rihip_ip_index_set_ivf(void* handle, int nlist, const float* centroids, const int32_t* assign, void* stream) {
  IpIndex* h = (IpIndex*)handle;
  RCCHK(check_trainable(h, nlist, "ip_index_set_ivf"));
  RIHIP_REQUIRE(centroids && assign, RIHIP_ERR_ARG, "ip_index_set_ivf: null pointer");
  for (int64_t i = 0; i < h->N; ++i)
    RIHIP_REQUIRE(assign[i] >= 0 && assign[i] < nlist, RIHIP_ERR_ARG, "ip_index_set_ivf: assign[%lld]=%d outside [0,%d)",
                  (long long)i, assign[i], nlist);
  hipStream_t st = (hipStream_t)stream;
  float* C = nullptr;
  int* a_dev = nullptr;
  HIPCHK(hipMalloc((void**)&C, sizeof(float) * (size_t)nlist * h->d));
  const std::vector<float> cp = pad_host(centroids, nlist, h->du, h->d);
  if (hipMalloc((void**)&a_dev, sizeof(int) * h->N) != hipSuccess ||
      hipMemcpyAsync(C, cp.data(), sizeof(float) * (size_t)nlist * h->d, hipMemcpyHostToDevice, st) != hipSuccess ||
      hipMemcpyAsync(a_dev, assign, sizeof(int) * h->N, hipMemcpyHostToDevice, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess) {
    hipFree(C); hipFree(a_dev);
    rihip_set_error("ip_index_set_ivf: upload failed");
    return RIHIP_ERR_HIP;
  }
  const int rc = train_impl(h, nlist, 0, C, a_dev, st);
  hipFree(a_dev);
  return rc;
}

extern "C" int rihip_ip_index_nlist(void* handle) { return handle ? ((IpIndex*)handle)->nlist : 0; }

// read the trained partition back: centroids host [nlist,d], assign host int32 [N] (either may be NULL); synchronous
extern "C" int rihip_ip_index_get_ivf(void* handle, float* centroids, int32_t* assign) {
  IpIndex* h = (IpIndex*)handle;
  RIHIP_REQUIRE(h && h->ivf, RIHIP_ERR_STATE, "ip_index_get_ivf: not an IVF index");
  if (centroids) {
    std::vector<float> c((size_t)h->nlist * h->d);
    HIPCHK(hipMemcpy(c.data(), h->C, sizeof(float) * c.size(), hipMemcpyDeviceToHost));
    unpad_host(c.data(), h->nlist, h->d, h->du, centroids);
  }
  if (assign) {
    std::vector<int64_t> rid(h->Np);
    std::vector<int> tl(h->Np / TR);
    HIPCHK(hipMemcpy(rid.data(), h->row_ids, sizeof(int64_t) * h->Np, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(tl.data(), h->tile_list, sizeof(int) * tl.size(), hipMemcpyDeviceToHost));
    for (int64_t p = 0; p < h->Np; ++p)
      if (rid[p] >= 0) assign[rid[p]] = tl[p / TR];
  }
  return RIHIP_OK;
}

// the stored vectors in insertion order (host [N,d]); synchronous.  faiss: index.reconstruct_n(0, ntotal)
extern "C" int rihip_ip_index_reconstruct(void* handle, float* out) {
  IpIndex* h = (IpIndex*)handle;
  RIHIP_REQUIRE(h && h->X && out, RIHIP_ERR_STATE, "ip_index_reconstruct: empty index or null output");
  const size_t bytes = sizeof(float) * (size_t)h->N * h->d;
  std::vector<float> host;
  float* dst = out;
  if (h->du != h->d) { host.resize((size_t)h->N * h->d); dst = host.data(); }
  if (!h->ivf) {
    HIPCHK(hipMemcpy(dst, h->X, bytes, hipMemcpyDeviceToHost));
  } else {
    float* tmp = nullptr;
    HIPCHK(hipMalloc((void**)&tmp, bytes));
    const int64_t tot = h->Np * (h->d / 4);
    hipLaunchKernelGGL(ivf_unpermute_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, 0, h->X, h->d, h->row_ids, h->Np, tmp);
    hipError_t e = hipMemcpy(dst, tmp, bytes, hipMemcpyDeviceToHost);
    hipFree(tmp);
    HIPCHK(e);
  }
  if (h->du != h->d) unpad_host(host.data(), h->N, h->d, h->du, out);
  return RIHIP_OK;
}

// list of each of n device rows under the index's centroids (the IndexFlatIP quantizer): assign device int32 [n]
extern "C" int rihip_ip_index_assign(void* handle, const float* X, int64_t n, int32_t* assign, void* stream) {
  IpIndex* h = (IpIndex*)handle;
  RIHIP_REQUIRE(h && h->ivf && h->C, RIHIP_ERR_STATE, "ip_index_assign: not an IVF index");
  RIHIP_REQUIRE(X && assign && n > 0, RIHIP_ERR_ARG, "ip_index_assign: bad arguments");
  if (h->du != h->d) {
    RCCHK(h->qpad.reserve(n * h->d));
    RCCHK(pad_rows(X, n, h->du, h->d, h->qpad.p, (hipStream_t)stream));
    X = h->qpad.p;
  }
  RIHIP_REQUIRE((reinterpret_cast<uintptr_t>(X) & 15) == 0, RIHIP_ERR_ARG, "ip_index_assign: X must be 16-byte aligned");
  return launch_assign(h->d, X, n, h->C, h->nlist, assign, (hipStream_t)stream);
}

// ---- persistence: own binary format ("RIHIPIDX" v1); the .meta.pkl sidecar stays with the Python wrapper
extern "C" int rihip_ip_index_save(void* handle, const char* path) {
  IpIndex* h = (IpIndex*)handle;
  RIHIP_REQUIRE(h && h->X && path, RIHIP_ERR_STATE, "ip_index_save: empty index");
  FILE* f = fopen(path, "wb");
  RIHIP_REQUIRE(f, RIHIP_ERR_IO, "ip_index_save: cannot open %s", path);
  const int64_t rows = h->ivf ? h->Np : h->N;
  std::vector<float> X((size_t)rows * h->d);
  hipMemcpy(X.data(), h->X, sizeof(float) * X.size(), hipMemcpyDeviceToHost);
  const char magic[8] = {'R', 'I', 'H', 'I', 'P', 'I', 'D', 'X'};
  int64_t hdr[8] = {1, h->d, h->N, h->ivf ? 1 : 0, h->nlist, h->nprobe, h->Np, h->du};   // hdr[7]: caller's width (0 = d)
  fwrite(magic, 1, 8, f); fwrite(hdr, sizeof(int64_t), 8, f); fwrite(X.data(), sizeof(float), X.size(), f);
  if (h->ivf) {
    std::vector<float> C((size_t)h->nlist * h->d); std::vector<int> tl(h->Np / TR); std::vector<int64_t> rid(h->Np);
    hipMemcpy(C.data(), h->C, sizeof(float) * C.size(), hipMemcpyDeviceToHost);
    hipMemcpy(tl.data(), h->tile_list, sizeof(int) * tl.size(), hipMemcpyDeviceToHost);
    hipMemcpy(rid.data(), h->row_ids, sizeof(int64_t) * rid.size(), hipMemcpyDeviceToHost);
    fwrite(C.data(), sizeof(float), C.size(), f); fwrite(tl.data(), sizeof(int), tl.size(), f);
    fwrite(rid.data(), sizeof(int64_t), rid.size(), f); fwrite(h->list_len.data(), sizeof(int64_t), h->nlist, f);
  }
  const bool ok = !ferror(f);
  fclose(f);
  RIHIP_REQUIRE(ok, RIHIP_ERR_IO, "ip_index_save: write error on %s", path);
  return RIHIP_OK;
}

extern "C" int rihip_ip_index_load(const char* path, void** handle) {
  RIHIP_REQUIRE(path && handle, RIHIP_ERR_ARG, "ip_index_load: bad arguments");
  FILE* f = fopen(path, "rb");
  RIHIP_REQUIRE(f, RIHIP_ERR_IO, "ip_index_load: cannot open %s", path);
  char magic[8]; int64_t hdr[8];
  if (fread(magic, 1, 8, f) != 8 || memcmp(magic, "RIHIPIDX", 8) != 0 || fread(hdr, sizeof(int64_t), 8, f) != 8 || hdr[0] != 1) {
    fclose(f); rihip_set_error("ip_index_load: %s is not a RIHIPIDX v1 file", path); return RIHIP_ERR_IO;
  }
  IpIndex* h = new IpIndex();
  h->d = (int)hdr[1]; h->du = hdr[7] > 0 ? (int)hdr[7] : h->d; h->N = hdr[2]; h->ivf = hdr[3] != 0; h->nlist = (int)hdr[4]; h->nprobe = (int)hdr[5]; h->Np = hdr[6];
  const int64_t rows = h->ivf ? h->Np : h->N;
  std::vector<float> X((size_t)rows * h->d);
  bool ok = fread(X.data(), sizeof(float), X.size(), f) == X.size();
  if (ok) { ok = hipMalloc((void**)&h->X, sizeof(float) * X.size()) == hipSuccess && hipMemcpy(h->X, X.data(), sizeof(float) * X.size(), hipMemcpyHostToDevice) == hipSuccess; }
  if (ok && h->ivf) {
    std::vector<float> C((size_t)h->nlist * h->d); std::vector<int> tl(h->Np / TR); std::vector<int64_t> rid(h->Np);
    h->list_len.assign(h->nlist, 0);
    ok = fread(C.data(), sizeof(float), C.size(), f) == C.size() && fread(tl.data(), sizeof(int), tl.size(), f) == tl.size() &&
         fread(rid.data(), sizeof(int64_t), rid.size(), f) == rid.size() &&
         fread(h->list_len.data(), sizeof(int64_t), h->nlist, f) == (size_t)h->nlist;
    if (ok) ok = hipMalloc((void**)&h->C, sizeof(float) * C.size()) == hipSuccess && hipMalloc((void**)&h->tile_list, sizeof(int) * tl.size()) == hipSuccess &&
                 hipMalloc((void**)&h->row_ids, sizeof(int64_t) * rid.size()) == hipSuccess &&
                 hipMemcpy(h->C, C.data(), sizeof(float) * C.size(), hipMemcpyHostToDevice) == hipSuccess &&
                 hipMemcpy(h->tile_list, tl.data(), sizeof(int) * tl.size(), hipMemcpyHostToDevice) == hipSuccess &&
                 hipMemcpy(h->row_ids, rid.data(), sizeof(int64_t) * rid.size(), hipMemcpyHostToDevice) == hipSuccess;
    if (ok) ok = derive_ivf_aux(h, 0) == RIHIP_OK;
  }
  if (ok && !h->ivf) ok = prepare_flat(h, 0) == RIHIP_OK;
  fclose(f);
  if (!ok) { rihip_ip_index_destroy(h); rihip_set_error("ip_index_load: truncated file or allocation failure: %s", path); return RIHIP_ERR_IO; }
  *handle = h;
  return RIHIP_OK;
}
