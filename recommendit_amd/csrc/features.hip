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
    unsigned long long kk = 0ull;   // below every real score, -inf included
    if (i < kc) {
      const double v = cand[q * kc + i] >= 0 ? scores[q * kc + i] : -INFINITY;
      unsigned long long u = (unsigned long long)__double_as_longlong(v);
      kk = (u >> 63) ? ~u : (u | 0x8000000000000000ull);
      if (kk == 0ull) kk = 1ull;
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
  RIHIP_REQUIRE(nq >= 0 && kc >= 1 && kc <= 8192 && k >= 1, RIHIP_ERR_ARG, "rank_topk: kc=%d (1..8192), k=%d", kc, k);
  if (nq == 0) return RIHIP_OK;
  int P = 64;
  while (P < kc || P < k) P <<= 1;
  RIHIP_REQUIRE(P <= 8192, RIHIP_ERR_ARG, "rank_topk: k=%d too large", k);
  const size_t lds = (size_t)P * 10;
  static bool granted = false;
  if (!granted) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(rank_topk_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 10);
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
  hipLaunchKernelGGL(rank_features_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     user_tab, n_user_rows, item_tab, n_item_rows, user_ids, cand_ids, nq, kc, col_map, nf, X);
  RIHIP_CHECK_LAUNCH();
  return RIHIP_OK;
}
