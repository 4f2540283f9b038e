// Ranking-feature assembly between retrieval and ranking, on the GPU -- replaces the 500-iteration Python
// loop of RecommendationPipeline._build_ranking_features (reference src/serving/recommender.py:213-263,
// duplicated at src/pipelines/run_pipeline.py:188-213) and the Redis MGET behind it.
//
// HBM-bound row work: feature tables are float64 (the reference computes these in Python floats and only
// casts to float32 inside LightGBMRanker.predict, ranker.py:173 -- computing in f64 and casting once keeps the
// result bit-identical).  user_tab [n_users+1, 24] = 6 scalars + 18 genre prefs; item_tab [n_items+1, 23] =
// 5 scalars + 18 genre flags; rows absent from the store hold the reference's defaults.
// One thread per (candidate, output column): the 50 canonical columns of
// src/features/feature_engineering.py:434-443 are remapped through col_map to the ranker's own feature order
// (missing columns -> 0.0, recommender.py:334-336).
#include "common.h"
#include "recommendit_hip.h"

namespace {

constexpr int UW = 24, IW = 23, NG = 18, NCANON = 50;

__device__ __forceinline__ double canon_feature(const double* __restrict__ u, const double* __restrict__ it, int c) {
  if (c < 6) return u[c];                       // avg_rating, log_rating_count, recency, gender, age, occupation
  if (c < 11) return it[c - 6];                 // item_avg_rating, item_log_rating_count, popularity, stddev, year
  if (c == 11) return u[0] - it[0];             // rating_diff
  if (c == 12) return u[1] / (it[1] + 1e-8);    // user_item_popularity_ratio
  if (c == 13) {                                // genre_affinity: left-to-right sum like Python's sum()
    double s = 0.0;
#pragma unroll
    for (int g = 0; g < NG; ++g) s += u[6 + g] * it[5 + g];
    return s;
  }
  if (c < 14 + NG) return u[6 + (c - 14)];      // user_genre_i
  return it[5 + (c - 14 - NG)];                 // item_genre_i
}

__global__ __launch_bounds__(256) void rank_features_kernel(const double* __restrict__ user_tab, int64_t n_urows,
                                                            const double* __restrict__ item_tab, int64_t n_irows,
                                                            const int64_t* __restrict__ user_ids,
                                                            const int64_t* __restrict__ cand, int64_t nq, int kc,
                                                            const int* __restrict__ col_map, int nf, float* X) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= nq * kc * nf) return;
  const int j = (int)(i % nf);
  const int64_t row = i / nf;
  const int64_t q = row / kc;
  int64_t uid = user_ids[q], iid = cand[row];
  const int c = col_map[j];
  float v = 0.f;
  if (iid >= 0 && c >= 0) {  // padded candidate (-1) -> all-zero row; unknown ranker column -> 0.0
    if (uid < 0 || uid >= n_urows) uid = 0;     // row 0 of both tables = the reference's defaults
    if (iid >= n_irows) iid = 0;
    v = (float)canon_feature(user_tab + uid * UW, item_tab + iid * IW, c);
  }
  X[i] = v;
}

// one wave per candidate row (nf <= 64 output columns): the user row (24 doubles) and the item row (23 doubles) are
// read once, coalesced, by lanes 0..46; every output column then picks its operand from those lanes.  Same float64
// operations in the same order as canon_feature (the genre sum runs left to right), so the bits are equal; the
// one-thread-per-element kernel above did 2-36 scattered loads per element and serialised on the genre-affinity column.
// Broadcasts from FIXED lanes go through v_readlane (scalar registers), only the per-column operand pick is a
// ds_bpermute: 25 double shuffles per row through the LDS crossbar made a first version as slow as the old kernel.
__device__ __forceinline__ double readlane_d(double x, int lane) {
  const long long b = __double_as_longlong(x);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(b & 0xFFFFFFFFll), lane);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(b >> 32), lane);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__global__ __launch_bounds__(256) void rank_features_wave_kernel(const double* __restrict__ user_tab, int64_t n_urows,
                                                                 const double* __restrict__ item_tab, int64_t n_irows,
                                                                 const int64_t* __restrict__ user_ids,
                                                                 const int64_t* __restrict__ cand, int64_t n_rows, int kc,
                                                                 const int* __restrict__ col_map, int nf, float* X) {
  const int lane = threadIdx.x & 63;
  const int c = lane < nf ? col_map[lane] : -1;
  // operand lane of a plain copy column (c < 11 or a genre column); other columns ignore it
  int src = 0;
  if (c >= 0) {
    if (c < 6) src = c;
    else if (c < 11) src = UW + (c - 6);
    else if (c >= 14 && c < 14 + NG) src = 6 + (c - 14);
    else if (c >= 14 + NG) src = UW + 5 + (c - 14 - NG);
  }
  for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < n_rows; row += (int64_t)gridDim.x * 4) {
    int64_t uid = user_ids[row / kc], iid = cand[row];
    const bool pad = iid < 0;                      // padded candidate (-1) -> all-zero row
    if (uid < 0 || uid >= n_urows) uid = 0;        // row 0 of both tables = the reference's defaults
    if (iid < 0 || iid >= n_irows) iid = 0;
    // lanes 0..23: u[lane]; lanes 24..46: it[lane - 24]
    double v = 0.0;
    if (lane < UW) v = user_tab[uid * UW + lane];
    else if (lane < UW + IW) v = item_tab[iid * IW + (lane - UW)];
    // genre_affinity: products on lanes 0..17 (their operands straight from the two rows), summed left to right
    double pg = 0.0;
    if (lane < NG) pg = user_tab[uid * UW + 6 + lane] * item_tab[iid * IW + 5 + lane];
    double aff = 0.0;
#pragma unroll
    for (int g = 0; g < NG; ++g) aff += readlane_d(pg, g);
    const double u0 = readlane_d(v, 0), u1 = readlane_d(v, 1), i0 = readlane_d(v, UW), i1 = readlane_d(v, UW + 1);
    double f = __shfl(v, src, 64);
    if (c == 11) f = u0 - i0;
    else if (c == 12) f = u1 / (i1 + 1e-8);
    else if (c == 13) f = aff;
    if (lane < nf) X[row * nf + lane] = (pad || c < 0) ? 0.f : (float)f;
  }
}

// final stage of the serving chain: DataFrame.nlargest(k, "score") (src/serving/recommender.py:346) for every request of
// a batch -- the k best ranker scores, ties keep the retrieval order, padded candidates (id < 0) last -- in one launch
// instead of the where / sort / gather sequence of tensor ops (5 dependent launches per batch)
__global__ __launch_bounds__(256) void rank_topk_kernel(const double* __restrict__ scores, const int64_t* __restrict__ cand,
                                                        const float* __restrict__ rs, int kc, int k, int P,
                                                        int64_t* out_ids, double* out_scores, float* out_rs) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
  unsigned long long* key = reinterpret_cast<unsigned long long*>(sm);       // [P] orderable score (larger = better)
  unsigned short* idx = reinterpret_cast<unsigned short*>(key + P);           // [P]
  const int64_t q = blockIdx.x;
  const int tid = threadIdx.x;
  for (int i = tid; i < P; i += 256) {
    unsigned long long kk = 0ull;   // slots past kc: below everything
    if (i < kc) {
      const bool real = cand[q * kc + i] >= 0;
      const double v = real ? scores[q * kc + i] : -INFINITY;
      unsigned long long u = (unsigned long long)__double_as_longlong(v);
      kk = (u >> 63) ? ~u : (u | 0x8000000000000000ull);
      // padded candidates last (key 1), NaN scores just before them (key 2: DataFrame.nlargest never ranks a NaN above
      // a number; a positive NaN would otherwise order above +inf), every number above both (key(-inf) = 2^52 - 1)
      if (!real) kk = 1ull;
      else if (v != v) kk = 2ull;
    }
    key[i] = kk; idx[i] = (unsigned short)i;
  }
  __syncthreads();
  for (int size = 2; size <= P; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int i = tid; i < P / 2; i += 256) {
        const int lo = (i / stride) * (stride << 1) + (i % stride), hi = lo + stride;
        const bool desc = ((lo & size) == 0);
        const unsigned long long a = key[lo], b = key[hi];
        const unsigned short ia = idx[lo], ib = idx[hi];
        const bool a_first = a > b || (a == b && ia < ib);   // total order: score desc, then retrieval position asc
        if (desc ? !a_first : a_first) { key[lo] = b; key[hi] = a; idx[lo] = ib; idx[hi] = ia; }
      }
      __syncthreads();
    }
  }
  for (int i = tid; i < k; i += 256) {
    const int j = idx[i];
    const bool ok = i < kc && j < kc;
    out_ids[q * k + i] = ok ? cand[q * kc + j] : -1;
    out_scores[q * k + i] = ok ? (cand[q * kc + j] >= 0 ? scores[q * kc + j] : -INFINITY) : -INFINITY;
    out_rs[q * k + i] = ok ? rs[q * kc + j] : -INFINITY;
  }
}

}  // namespace

extern "C" int rihip_rank_topk(const double* scores, const int64_t* cand, const float* retrieval_scores, int64_t nq, int kc,
                               int k, int64_t* out_ids, double* out_scores, float* out_retrieval_scores, void* stream) {
  RIHIP_REQUIRE(scores && cand && retrieval_scores && out_ids && out_scores && out_retrieval_scores, RIHIP_ERR_ARG,
                "rank_topk: null pointer");
  RIHIP_REQUIRE(nq >= 0 && kc >= 1 && kc <= 16384 && k >= 1, RIHIP_ERR_ARG, "rank_topk: kc=%d (1..16384), k=%d", kc, k);
  if (nq == 0) return RIHIP_OK;
  int P = 64;
  while (P < kc || P < k) P <<= 1;
  RIHIP_REQUIRE(P <= 16384, RIHIP_ERR_ARG, "rank_topk: k=%d too large", k);   // = the index's K_MAX: 160 KiB of LDS
  const size_t lds = (size_t)P * 10;
  static bool granted = false;
  if (!granted) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(rank_topk_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 10);
    granted = true;
  }
  hipLaunchKernelGGL(rank_topk_kernel, dim3((unsigned)nq), dim3(256), lds, (hipStream_t)stream, scores, cand, retrieval_scores,
                     kc, k, P, out_ids, out_scores, out_retrieval_scores);
  RIHIP_CHECK_LAUNCH();
  return RIHIP_OK;
}

extern "C" int rihip_rank_features_widths(int* user_width, int* item_width, int* n_canonical) {
  if (user_width) *user_width = UW;
  if (item_width) *item_width = IW;
  if (n_canonical) *n_canonical = NCANON;
  return RIHIP_OK;
}

extern "C" int rihip_rank_features_build(const double* user_tab, int64_t n_user_rows, const double* item_tab,
                                         int64_t n_item_rows, const int64_t* user_ids, const int64_t* cand_ids,
                                         int64_t nq, int kc, const int* col_map, int nf, float* X, void* stream) {
  RIHIP_REQUIRE(user_tab && item_tab && user_ids && cand_ids && col_map && X, RIHIP_ERR_ARG,
                "rank_features_build: null pointer");
  RIHIP_REQUIRE(nq >= 0 && kc > 0 && nf > 0 && n_user_rows > 0 && n_item_rows > 0, RIHIP_ERR_ARG,
                "rank_features_build: bad sizes");
  const int64_t n = nq * kc * nf;
  if (n == 0) return RIHIP_OK;
  if (nf <= 64) {
    const int64_t n_rows = nq * kc, nb = (n_rows + 3) / 4;
    hipLaunchKernelGGL(rank_features_wave_kernel, dim3((unsigned)(nb < 16384 ? nb : 16384)), dim3(256), 0,
                       (hipStream_t)stream, user_tab, n_user_rows, item_tab, n_item_rows, user_ids, cand_ids, n_rows, kc,
                       col_map, nf, X);
  } else {
    hipLaunchKernelGGL(rank_features_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       user_tab, n_user_rows, item_tab, n_item_rows, user_ids, cand_ids, nq, kc, col_map, nf, X);
  }
  RIHIP_CHECK_LAUNCH();
  return RIHIP_OK;
}
