// LambdaMART training on gfx950 -- the body of the reference's LightGBMRanker.train (src/models/ranker.py:52-155:
// lgb.train(objective=lambdarank, num_leaves 63, learning_rate 0.05, min_child_samples 20, colsample_bytree 0.8,
// reg_alpha 0.1, reg_lambda 0.1, label_gain [0,1,3,7,15], eval_at [5,10,20], early stopping 30).  SURVEY.md §8f-4.
//
// lightgbm is third-party and absent here (parity unpinned): this is LightGBM's published algorithm -- histogram
// based, leaf-wise GBDT with the LambdarankNDCG objective -- restated, and pinned bit for bit to its NumPy restatement
// oracle/lambdamart_np.py (tests/test_gpu_lambdamart.py).  What makes "bit for bit" possible: gradients and hessians
// are quantised to integers (2^20 levels of the largest magnitude) before they enter the histograms, so every
// histogram entry is an integer sum, independent of the order of the atomics; gains are then evaluated in double.
// Fidelity switches towards LightGBM's defaults (rihip_lambdamart_params, all off by default = the round-2 behaviour):
//   hist_bits = 40   float-histogram fidelity: gradients enter at 2^-40 of the round's maximum (LightGBM accumulates float
//                    gradients in double; 2^-40 is 65 536x finer than the float32 rounding of a gradient) and the sums
//                    are still exact integers, i.e. independent of summation order (an f64 accumulation is not);
//   use_missing = 1  NaN is a value of its own: a feature with NaN in the bin sample gets a last "missing" bin, every
//                    threshold is tried with the missing rows on either side, the node stores the better default
//                    direction (decision_type missing=NaN, default_left); 0: NaN is read as 0.0;
//   split_order = 1  ties between equal-gain thresholds of a feature are resolved as FeatureHistogram::
//                    FindBestThreshold does (right-to-left scan, strict '>': the HIGHEST threshold; then the
//                    left-to-right scan with the missing rows on the right); 0: the lowest threshold.
//
//   binning        <= 255 bins per feature, upper bounds from a strided sample of <= 200 000 rows (host), rows binned
//                  on the device (one byte per value: 145 MB for ML-1M's 2.9 M x 50 ranking rows)
//   gradients      one stable segmented radix sort of all query groups by score (rocPRIM), then one workgroup per
//                  query: pairs (i < truncation_level, j > i) of different labels, fp64
//   histograms     (gq, hq, count) per (feature, bin) of a leaf in LDS (int32, 1024-row chunks), summed in int64;
//                  smaller child built, larger child = parent - smaller
//   split search   one wave per feature, prefix over the bins, gain = ThresholdL1(G)^2/(H+l2) in double
//   growth         best-first on the host (one small read-back per split), rows partitioned on the device
// The result is a LightGBM-format text model (what Booster.save_model writes, src/models/ranker.py:203-209) that
// rihip_gbdt_create_from_text and a real lightgbm.Booster(model_file=...) load.
#include "common.h"
#include "recommendit_hip.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <type_traits>
#include <vector>

#include <rocprim/rocprim.hpp>

// No FMA contraction in this file: split gains are compared for EQUALITY (ties between thresholds with identical
// partitions are resolved by evaluation order), so two evaluations of the same sums must round identically -- and like the
// IEEE double arithmetic of the NumPy restatement the trees are compared against.
#pragma clang fp contract(off)

namespace {

#define TCHK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { rihip_set_error("%s:%d %s", __FILE__, __LINE__, hipGetErrorString(_e)); return RIHIP_ERR_HIP; } } while (0)
#define TRC(e) do { int _rc = (e); if (_rc) return _rc; } while (0)

constexpr int NBIN = 256;
constexpr int HCH = 1024;        // rows per histogram workgroup: 1024 x 2^20 < 2^31 keeps int32 sums exact
constexpr int FCH = 10;          // features per histogram WORKGROUP (10 x 256 x 3 x 4 B = 30 KB of LDS; 5 with 8-byte cells):
                                 // grid = (row chunks, feature chunks), one launch per histogram.  (All 50 features in one
                                 // 150 KB workgroup -- 1 per CU -- took 55 us per call: zeroing and flushing 38 400 cells
                                 // cost as much as the 1 024 rows, with nothing else resident to overlap them.)
constexpr int MAX_GROUP = 16384; // documents per query (sorted scores + labels of a query live in LDS)
constexpr int MAX_T = 32;        // truncation level supported by the per-thread pair accumulators
constexpr double QLEVELS = 1048576.0;
constexpr double K_EPS = 1e-15;

struct SplitInfo { double gain; int feature, bin; long long glq, hlq, cl; long long gq, hq, c; int default_left, pad; };

__device__ __forceinline__ uint64_t d2ord_desc(double v) {  // ascending radix order == descending score
  uint64_t u = (uint64_t)__double_as_longlong(v);
  u = (u >> 63) ? ~u : (u | 0x8000000000000000ull);
  return ~u;
}

// nanbin[f] >= 0: the feature has a "missing" bin (the last one) and NaN goes there; else NaN is read as 0.0
__global__ void bin_rows_kernel(const float* __restrict__ X, int64_t n, int F, const double* __restrict__ ub,
                                const int* __restrict__ nb, const int* __restrict__ nanbin, uint8_t* Xb) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * F) return;
  const int f = (int)(i % F);
  double x = (double)X[i];
  const int nbn = nanbin[f];
  if (x != x) {
    if (nbn >= 0) { Xb[i] = (uint8_t)nbn; return; }
    x = 0.0;
  }
  const double* u = ub + (size_t)f * NBIN;
  int lo = 0, hi = (nbn >= 0 ? nbn : nb[f]) - 1;   // first real bin with x <= upper bound (the last real bound is +inf)
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (x <= u[mid]) hi = mid; else lo = mid + 1;
  }
  Xb[i] = (uint8_t)lo;
}

__global__ void iota_rows_kernel(int* p, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = (int)i;
}

__global__ void sort_keys_kernel(const double* __restrict__ s, int64_t n, uint64_t* key, int* val) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { key[i] = d2ord_desc(s[i]); val[i] = (int)i; }
}

__device__ __forceinline__ double block_sum_d(double v, double* red, int tid) {  // fixed order: wave tree, then waves 0..3
  v = wave_sum_d(v);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  return ((red[0] + red[1]) + red[2]) + red[3];
}

struct GradArgs {
  const double* score; const float* label; const int* sorted; const int64_t* goff; int ng;
  const double* gain_tab; int n_gain; double sigmoid; int T; int norm;
  double* lam; double* hes;
};
// LambdarankNDCG::GetGradientsForOneQuery, one workgroup per query
__global__ __launch_bounds__(256) void lambdarank_kernel(GradArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
  const int q = blockIdx.x, tid = threadIdx.x;
  const int64_t b = a.goff[q];
  const int cnt = (int)(a.goff[q + 1] - b);
  double* ss = reinterpret_cast<double*>(sm);                 // [cnt] scores in rank order
  signed char* ll = reinterpret_cast<signed char*>(ss + cnt);  // [cnt] labels in rank order
  __shared__ double red[4];
  __shared__ int lab_hist[32];
  if (tid < 32) lab_hist[tid] = 0;
  __syncthreads();
  for (int r = tid; r < cnt; r += 256) {
    const int d = a.sorted[b + r];
    ss[r] = a.score[d];
    int l = (int)a.label[d];
    l = l < 0 ? 0 : (l >= a.n_gain ? a.n_gain - 1 : l);
    ll[r] = (signed char)l;
    atomicAdd(&lab_hist[l], 1);
  }
  __syncthreads();
  // max DCG at the truncation level: gains in descending label order
  double mx = 0.0;
  {
    int pos = 0;
    for (int l = a.n_gain - 1; l >= 0 && pos < a.T; --l)
      for (int c = 0; c < lab_hist[l] && pos < a.T; ++c, ++pos) mx += a.gain_tab[l] / log2((double)pos + 2.0);
  }
  const double inv = mx > 0.0 ? 1.0 / mx : 0.0;
  const int ni = (cnt - 1 < a.T) ? cnt - 1 : a.T;
  const bool do_norm = a.norm && cnt > 0 && ss[0] != ss[cnt - 1];
  double li[MAX_T], hi[MAX_T];   // this thread's share of lambda_i / hessian_i, i < truncation level
#pragma unroll
  for (int i = 0; i < MAX_T; ++i) { li[i] = 0.0; hi[i] = 0.0; }
  __shared__ double disc[MAX_T];   // 1 / log2(i + 2) of the truncation positions (the same expression as for the partner r)
  if (tid < MAX_T) disc[tid] = 1.0 / log2((double)tid + 2.0);
  __syncthreads();
  double sum_l = 0.0;
  for (int r = tid; r < cnt; r += 256) {
    const double sr = ss[r];
    const int lr = ll[r];
    const double gr = a.gain_tab[lr], dr = 1.0 / log2((double)r + 2.0);
    double lam_r = 0.0, hes_r = 0.0;
    const int iend = r < ni ? r : ni;
    // The loop over the truncation positions is unrolled with a STATIC index: every lane is at the same i in the same
    // iteration, so li[i] / hi[i] are plain registers (a dynamic index cost a 32-way select of two doubles per pair);
    // the order of the additions is the one of the rolled loop (i ascending), so the sums are bit for bit the same.
#pragma unroll
    for (int i = 0; i < MAX_T; ++i) {           // pair (i, r), i ranked above r
      if (i < iend) {
        const int lab_i = ll[i];
        if (lab_i != lr) {
          const bool hi_is_i = lab_i > lr;
          const double si = ss[i];
          const double ds = hi_is_i ? si - sr : sr - si;
          double dn = fabs(a.gain_tab[lab_i] - gr) * fabs(disc[i] - dr) * inv;
          if (do_norm) dn = dn / (0.01 + fabs(ds));
          const double rho = 1.0 / (1.0 + exp(a.sigmoid * ds));
          const double pl = -a.sigmoid * dn * rho;
          const double ph = a.sigmoid * a.sigmoid * dn * rho * (1.0 - rho);
          lam_r += hi_is_i ? -pl : pl;
          hes_r += ph;
          li[i] += hi_is_i ? pl : -pl; hi[i] += ph;
          sum_l += -2.0 * pl;
        }
      }
    }
    a.lam[b + r] = lam_r;                       // rank-order scratch; positions < ni are completed below
    a.hes[b + r] = hes_r;
  }
  sum_l = block_sum_d(sum_l, red, tid);
  const double nf = (a.norm && sum_l > 0.0) ? log2(1.0 + sum_l) / sum_l : 1.0;
#pragma unroll 1
  for (int i = 0; i < ni; ++i) {
    double l_i = 0.0, h_i = 0.0;
#pragma unroll
    for (int k = 0; k < MAX_T; ++k) if (k == i) { l_i = li[k]; h_i = hi[k]; }
    l_i = block_sum_d(l_i, red, tid);
    h_i = block_sum_d(h_i, red, tid);
    if (tid == 0) { a.lam[b + i] += l_i; a.hes[b + i] += h_i; }
  }
  __syncthreads();
  __threadfence_block();
  for (int r = tid; r < cnt; r += 256) { a.lam[b + r] *= nf; a.hes[b + r] *= nf; }
}

// rank-order lambdas -> document order, and the largest magnitudes (for the quantisation scale)
__global__ __launch_bounds__(256) void unsort_absmax_kernel(const double* __restrict__ ls, const double* __restrict__ hs, const int* __restrict__ sorted,
                                     int64_t n, double* lam, double* hes, unsigned long long* mx) {
  // (one atomic pair per workgroup of a grid-stride loop: one per WAVE of a 9 000-workgroup grid serialised 75 000
  // same-address atomics in L2 -- 0.85 ms per tree for 38 MB of traffic)
  __shared__ double wg[4], wh[4];
  double gm = 0.0, hm = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const double g = ls[i], h = hs[i];
    const int d = sorted[i];
    lam[d] = g; hes[d] = h;
    gm = fmax(gm, fabs(g)); hm = fmax(hm, h);
  }
  for (int o = 32; o > 0; o >>= 1) { gm = fmax(gm, __shfl_xor(gm, o, 64)); hm = fmax(hm, __shfl_xor(hm, o, 64)); }
  if ((threadIdx.x & 63) == 0) { wg[threadIdx.x >> 6] = gm; wh[threadIdx.x >> 6] = hm; }
  __syncthreads();
  if (threadIdx.x == 0) {   // non-negative doubles order like their bit patterns
    gm = fmax(fmax(wg[0], wg[1]), fmax(wg[2], wg[3])); hm = fmax(fmax(wh[0], wh[1]), fmax(wh[2], wh[3]));
    atomicMax(&mx[0], (unsigned long long)__double_as_longlong(gm));
    atomicMax(&mx[1], (unsigned long long)__double_as_longlong(hm));
  }
}
template <typename T>
__global__ void quantize_kernel(const double* __restrict__ lam, const double* __restrict__ hes, int64_t n, double sg, double sh,
                                T* gq, T* hq) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { gq[i] = (T)llrint(lam[i] * sg); hq[i] = (T)llrint(hes[i] * sh); }
}

// partial histograms of rows[b0 .. b0+len) over features [f0, f0+nf): cells of type T in LDS (int32 for 2^20 levels: 1024
// rows x 2^20 < 2^31; int64 for 2^40 levels), added into the int64 histogram
template <typename T>
__global__ __launch_bounds__(256) void hist_kernel(const uint8_t* __restrict__ Xb, int F, const int* __restrict__ rows,
                                                   int64_t b0, int64_t len, const T* __restrict__ gq,
                                                   const T* __restrict__ hq, int fch, long long* hist) {
  extern __shared__ __attribute__((aligned(16))) unsigned char hs_raw[];   // [nf][NBIN][3] of T
  const int f0 = blockIdx.y * fch;
  const int nf = F - f0 < fch ? F - f0 : fch;
  T* hs = reinterpret_cast<T*>(hs_raw);
  using U = typename std::conditional<sizeof(T) == 8, unsigned long long, int>::type;
  const int tid = threadIdx.x;
  for (int i = tid; i < nf * NBIN * 3; i += 256) hs[i] = 0;
  __syncthreads();
  const int64_t c0 = (int64_t)blockIdx.x * HCH;
  const int64_t c1 = (c0 + HCH < len) ? c0 + HCH : len;
  // (one row at a time per thread.  Handling a thread's four rows side by side -- ids, then gradients and bin bytes, then
  // the atomics -- was measured slower in both orders: 58 and 70 us per call against 43)
  for (int64_t i = c0 + tid; i < c1; i += 256) {
    const int r = rows[b0 + i];
    const T g = gq[r], h = hq[r];
    const uint8_t* xb = Xb + (size_t)r * F + f0;
    for (int f = 0; f < nf; ++f) {
      U* cell = reinterpret_cast<U*>(hs + ((size_t)f * NBIN + xb[f]) * 3);
      atomicAdd(cell, (U)g); atomicAdd(cell + 1, (U)h); atomicAdd(cell + 2, (U)1);
    }
  }
  __syncthreads();
  for (int i = tid; i < nf * NBIN * 3; i += 256) {
    const T v = hs[i];
    if (v != 0) atomicAdd(reinterpret_cast<unsigned long long*>(hist) + (size_t)f0 * NBIN * 3 + i, (unsigned long long)(long long)v);
  }
}

__device__ __forceinline__ double thr_l1(double g, double l1) {
  const double t = fabs(g) - l1;
  return t > 0.0 ? (g > 0.0 ? t : -t) : 0.0;
}
__device__ __forceinline__ double leaf_gain(double G, double H, double l1, double l2) {
  const double t = thr_l1(G, l1);
  return t * t / (H + l2);
}
// best split of one leaf per workgroup (blockIdx.x = 0 / 1: the two children of a split in ONE launch): wave w takes
// features w, w+16, ...; every lane evaluates the thresholds of its 4 bins, the wave keeps the best by (gain, evaluation
// order); ties between features go to the lower feature index (LightGBM: SplitInfo::operator>).
// A feature with a missing bin (nanbin[f] = its last bin) is tried both ways at every threshold: missing rows on the left
// (default_left = 1) and on the right (default_left = 0, which also offers "every real value left | missing right").
// Evaluation order = who wins among EQUAL gains (a leaf with few rows has many empty bins, i.e. runs of thresholds with
// identical partitions): order 0 = lowest threshold first; order 1 = FeatureHistogram::FindBestThreshold: the
// right-to-left scan first (highest threshold first, missing rows left), then the left-to-right scan (missing right),
// a later candidate replaces an earlier one only if its gain is strictly larger.
// One wave per (child, feature): grid = (ceil(F / 4), children), 4 waves per workgroup (the first form ran ONE workgroup
// per child over all features: 24-38 us on 2 of 256 CUs, a quarter of a tree's time).  Every wave writes its feature's
// best candidate; the last workgroup to finish (device counter, reset by it) picks each child's best feature -- lower
// feature index among equal gains, whatever the order the workgroups ran in.
// `sub_small` (the larger child of a split): its histogram is parent - smaller child, formed here bin by bin in the
// registers of the lane that owns the bin and written back in place, in the parent's slot.
__global__ __launch_bounds__(256) void split_kernel(const long long* __restrict__ hist0, const long long* __restrict__ hist1,
                                                    int F, const int* __restrict__ nb, const int* __restrict__ nanbin,
                                                    const unsigned char* __restrict__ used, double sg, double sh, double l1,
                                                    double l2, int min_child, double min_hess, int order, SplitInfo* out0,
                                                    const long long* __restrict__ sub_small, int sub_child,
                                                    SplitInfo* feat_best, unsigned* done_counter) {
  __shared__ int is_last;
  const int child = blockIdx.y;
  const long long* hist = child == 0 ? hist0 : hist1;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int f = blockIdx.x * 4 + w;
  SplitInfo best; best.gain = 0.0; best.feature = -1; best.bin = 0; best.glq = best.hlq = best.cl = 0; best.gq = best.hq = best.c = 0;
  best.default_left = 1; best.pad = 0;
  if (hist && f < F && used[f] && nb[f] >= 2) {   // (a null histogram: this child is not split further)
    const int nbf = nb[f];
    const int nbn = nanbin[f];
    const int nr = nbn >= 0 ? nbf - 1 : nbf;      // real bins
    long long* h = const_cast<long long*>(hist) + (size_t)f * NBIN * 3;
    const bool sub = sub_small && child == sub_child;
    const long long* hsm = sub ? sub_small + (size_t)f * NBIN * 3 : nullptr;
    // lane l owns bins 4l..4l+3: local sums, then an exclusive prefix over the lanes
    long long g4[4], h4[4], c4[4];
    long long gs = 0, hsum = 0, cs = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int bb = 4 * lane + k;
      g4[k] = bb < nbf ? h[bb * 3] : 0; h4[k] = bb < nbf ? h[bb * 3 + 1] : 0; c4[k] = bb < nbf ? h[bb * 3 + 2] : 0;
      if (sub && bb < nbf) {
        g4[k] -= hsm[bb * 3]; h4[k] -= hsm[bb * 3 + 1]; c4[k] -= hsm[bb * 3 + 2];
        h[bb * 3] = g4[k]; h[bb * 3 + 1] = h4[k]; h[bb * 3 + 2] = c4[k];
      }
    }
    // the missing bin (the last one) from its owner's registers: the value in memory may not be the subtracted one yet
    long long NG = 0, NH = 0, NC = 0;
    if (nbn >= 0) {
      const int k0 = nbn & 3, l0 = nbn >> 2;
      const long long sg4 = k0 == 0 ? g4[0] : k0 == 1 ? g4[1] : k0 == 2 ? g4[2] : g4[3];
      const long long sh4 = k0 == 0 ? h4[0] : k0 == 1 ? h4[1] : k0 == 2 ? h4[2] : h4[3];
      const long long sc4 = k0 == 0 ? c4[0] : k0 == 1 ? c4[1] : k0 == 2 ? c4[2] : c4[3];
      NG = __shfl(sg4, l0, 64); NH = __shfl(sh4, l0, 64); NC = __shfl(sc4, l0, 64);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { gs += g4[k]; hsum += h4[k]; cs += c4[k]; }
    long long pg = gs, phh = hsum, pc = cs;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const long long tg = __shfl_up(pg, o, 64), th = __shfl_up(phh, o, 64), tc = __shfl_up(pc, o, 64);
      if (lane >= o) { pg += tg; phh += th; pc += tc; }
    }
    const long long TG = __shfl(pg, 63, 64), TH = __shfl(phh, 63, 64), TC = __shfl(pc, 63, 64);
    long long cg = pg - gs, ch = phh - hsum, cc = pc - cs;   // exclusive
    const double G = sg > 0.0 ? (double)TG / sg : 0.0, H = sh > 0.0 ? (double)TH / sh : 0.0;
    const double parent = leaf_gain(G, H, l1, l2);
    double bg = 0.0; int bb_best = -1, bord = 0x7fffffff, bdl = 1; long long bgl = 0, bhl = 0, bcl = 0;
    auto consider = [&](long long lg, long long lh, long long lc, int bb, int dl, int ord) {
      const double GL = sg > 0.0 ? (double)lg / sg : 0.0, HL = sh > 0.0 ? (double)lh / sh : 0.0;
      const double GR = G - GL, HR = H - HL;
      const long long CR = TC - lc;
      if (lc >= min_child && CR >= min_child && HL >= min_hess && HR >= min_hess) {
        const double gain = leaf_gain(GL, HL, l1, l2) + leaf_gain(GR, HR, l1, l2) - parent;
        if (gain > K_EPS && (bb_best < 0 || gain > bg || (gain == bg && ord < bord))) {
          bg = gain; bb_best = bb; bord = ord; bdl = dl; bgl = lg; bhl = lh; bcl = lc;
        }
      }
    };
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int bb = 4 * lane + k;
      cg += g4[k]; ch += h4[k]; cc += c4[k];            // sums of the real bins <= bb (the missing bin is the last one)
      if (nbn < 0) {
        if (bb < nr - 1) consider(cg, ch, cc, bb, 1, order ? (nr - 2 - bb) : bb);
      } else {
        // missing left -- only with at least one real row on the left: "missing rows alone | every real row" is the
        // mirror image of the last missing-right candidate (equal gain up to rounding), offered once, there
        if (bb < nr - 1 && cc > 0) consider(cg + NG, ch + NH, cc + NC, bb, 1, order ? (nr - 2 - bb) : 2 * bb);
        if (bb < nr) consider(cg, ch, cc, bb, 0, order ? (1000 + bb) : 2 * bb + 1);                    // missing right
      }
    }
    // larger gain wins, equal gains: the earlier candidate of the evaluation order
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const double og = __shfl_xor(bg, o, 64);
      const int ob = __shfl_xor(bb_best, o, 64), oo = __shfl_xor(bord, o, 64), od = __shfl_xor(bdl, o, 64);
      const long long ogl = __shfl_xor(bgl, o, 64), ohl = __shfl_xor(bhl, o, 64), ocl = __shfl_xor(bcl, o, 64);
      const bool take = ob >= 0 && (bb_best < 0 || og > bg || (og == bg && oo < bord));
      if (take) { bg = og; bb_best = ob; bord = oo; bdl = od; bgl = ogl; bhl = ohl; bcl = ocl; }
    }
    if (bb_best >= 0) {
      best.gain = bg; best.feature = f; best.bin = bb_best; best.glq = bgl; best.hlq = bhl; best.cl = bcl;
      best.gq = TG; best.hq = TH; best.c = TC; best.default_left = bdl;
    }
  }
  if (f < F && lane == 0) feat_best[(size_t)child * F + f] = best;
  // ---- the last workgroup to arrive reduces over the features
  __threadfence();
  __syncthreads();
  if (tid == 0) {
    const unsigned total = gridDim.x * gridDim.y;
    is_last = atomicAdd(done_counter, 1u) == total - 1u;
  }
  __syncthreads();
  if (!is_last) return;
  __threadfence();
  if (w < (int)gridDim.y) {   // wave c: child c
    SplitInfo b = best;        // (any value: replaced below)
    b.feature = -1; b.gain = 0.0;
    double bgain = 0.0; int bfeat = -1;
    for (int f0 = 0; f0 < F; f0 += 64) {
      const int ff = f0 + lane;
      double gsel = 0.0; int fsel = -1;
      if (ff < F) {
        const SplitInfo* q = &feat_best[(size_t)w * F + ff];
        const int qf = __builtin_nontemporal_load(&q->feature);
        if (qf >= 0) { fsel = qf; gsel = __builtin_nontemporal_load(&q->gain); }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const double og = __shfl_xor(gsel, o, 64);
        const int of = __shfl_xor(fsel, o, 64);
        if (of >= 0 && (fsel < 0 || og > gsel || (og == gsel && of < fsel))) { gsel = og; fsel = of; }
      }
      if (fsel >= 0 && (bfeat < 0 || gsel > bgain)) { bgain = gsel; bfeat = fsel; }   // chunks ascend: ties keep the lower feature
    }
    if (lane == 0) {
      if (bfeat >= 0) {
        const SplitInfo* q = &feat_best[(size_t)w * F + bfeat];
        SplitInfo r;
        r.gain = __builtin_nontemporal_load(&q->gain); r.feature = bfeat; r.bin = __builtin_nontemporal_load(&q->bin);
        r.glq = __builtin_nontemporal_load(&q->glq); r.hlq = __builtin_nontemporal_load(&q->hlq); r.cl = __builtin_nontemporal_load(&q->cl);
        r.gq = __builtin_nontemporal_load(&q->gq); r.hq = __builtin_nontemporal_load(&q->hq); r.c = __builtin_nontemporal_load(&q->c);
        r.default_left = __builtin_nontemporal_load(&q->default_left); r.pad = 0;
        out0[w] = r;
      } else {
        out0[w] = b;
      }
    }
  }
  if (tid == 0) *done_counter = 0u;   // ready for the next launch (launches on one stream do not overlap)
}

// Partition of a leaf's rows in ONE launch: left = bin <= threshold bin, the rows of the feature's missing bin follow the
// node's default direction.  Every workgroup takes 8192 rows, counts its left / right rows (wave scans + 16 wave totals),
// reserves its two output ranges with one atomic each on the split's cursor pair, and writes src -> dst (the two row
// buffers alternate from parent to children: no copy back).  The order of the rows INSIDE a child depends on the order
// the workgroups arrive -- nothing downstream depends on it: histograms and leaf sums are integer sums.  (It replaced
// flags + a two-kernel rocPRIM scan + scatter + copy: five launches per split.)
constexpr int PART_ROWS = 8192;
__global__ __launch_bounds__(1024) void part_kernel(const uint8_t* __restrict__ Xb, int F, const int* __restrict__ src,
                                                    int* __restrict__ dst, int64_t b0, int64_t len, int f, int bin, int nanbin,
                                                    int default_left, int* cursors, int64_t n_left) {
  __shared__ int wl[16], wr[16];
  __shared__ int baseL, baseR;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int64_t c0 = (int64_t)blockIdx.x * PART_ROWS;
  int row[8];
  bool lft[8], ok[8];
  int nl = 0, nr = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int64_t i = c0 + (int64_t)k * 1024 + tid;
    ok[k] = i < len;
    row[k] = src[b0 + (ok[k] ? i : len - 1)];
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int x = Xb[(size_t)row[k] * F + f];
    lft[k] = (x == nanbin) ? (default_left != 0) : (x <= bin);
    nl += (ok[k] && lft[k]) ? 1 : 0;
    nr += (ok[k] && !lft[k]) ? 1 : 0;
  }
  int il = nl, ir = nr;                       // inclusive scans over the wave
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int tl = __shfl_up(il, o, 64), tr = __shfl_up(ir, o, 64);
    if (lane >= o) { il += tl; ir += tr; }
  }
  if (lane == 63) { wl[w] = il; wr[w] = ir; }
  __syncthreads();
  int offL = il - nl, offR = ir - nr;
  for (int k = 0; k < w; ++k) { offL += wl[k]; offR += wr[k]; }
  if (tid == 0) {
    int TL = 0, TR = 0;
    for (int k = 0; k < 16; ++k) { TL += wl[k]; TR += wr[k]; }
    baseL = TL ? atomicAdd(&cursors[0], TL) : 0;
    baseR = TR ? atomicAdd(&cursors[1], TR) : 0;
  }
  __syncthreads();
  int* dl = dst + b0 + baseL + offL;
  int* dr = dst + b0 + n_left + baseR + offR;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if (!ok[k]) continue;
    if (lft[k]) *dl++ = row[k]; else *dr++ = row[k];
  }
}
template <typename T>
__global__ void sum_leaf_kernel(const int* __restrict__ rows, int64_t b0, int64_t len, const T* __restrict__ gq,
                                const T* __restrict__ hq, long long* out) {
  long long g = 0, h = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = rows[b0 + i];
    g += gq[r]; h += hq[r];
  }
  for (int o = 32; o > 0; o >>= 1) { g += __shfl_xor(g, o, 64); h += __shfl_xor(h, o, 64); }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(reinterpret_cast<unsigned long long*>(out), (unsigned long long)g);
    atomicAdd(reinterpret_cast<unsigned long long*>(out) + 1, (unsigned long long)h);
  }
}
__global__ void add_leaf_kernel(const int* __restrict__ rows, int64_t b0, int64_t len, double v, double* score) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < len) score[rows[b0 + i]] += v;
}
// the new tree on binned rows (validation set): children >= 0 node, < 0 => ~leaf
// (bin[node] carries the threshold bin in its low 16 bits, default_left in bit 16)
__global__ void tree_add_kernel(const uint8_t* __restrict__ Xb, int64_t n, int F, const int* __restrict__ feat,
                                const int* __restrict__ bin, const int* __restrict__ lc, const int* __restrict__ rc,
                                const double* __restrict__ leaf, const int* __restrict__ nanbin, int n_nodes, double* score) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int node = n_nodes > 0 ? 0 : -1;
  const uint8_t* x = Xb + (size_t)i * F;
  while (node >= 0) {
    const int f = feat[node], b = bin[node];
    const int v = x[f];
    const bool left = (v == nanbin[f]) ? ((b >> 16) & 1) != 0 : v <= (b & 0xffff);
    node = left ? lc[node] : rc[node];
  }
  score[i] += leaf[~node];
}

// NDCG@k of every query from the sorted order: out[q * nk + t]; a query without a positive counts 1 (LightGBM)
__global__ __launch_bounds__(64) void ndcg_kernel(const float* __restrict__ label, const int* __restrict__ sorted,
                                                  const int64_t* __restrict__ goff, const double* __restrict__ gain_tab,
                                                  int n_gain, const int* __restrict__ ks, int nk, double* out) {
  const int q = blockIdx.x, lane = threadIdx.x;
  const int64_t b = goff[q];
  const int cnt = (int)(goff[q + 1] - b);
  __shared__ int lab_hist[32];
  if (lane < 32) lab_hist[lane] = 0;
  __syncthreads();
  for (int r = lane; r < cnt; r += 64) {
    int l = (int)label[b + r];
    l = l < 0 ? 0 : (l >= n_gain ? n_gain - 1 : l);
    atomicAdd(&lab_hist[l], 1);
  }
  __syncthreads();
  for (int t = 0; t < nk; ++t) {
    const int k = ks[t] < cnt ? ks[t] : cnt;
    double mx = 0.0;
    int pos = 0;
    for (int l = n_gain - 1; l >= 0 && pos < k; --l)
      for (int c = 0; c < lab_hist[l] && pos < k; ++c, ++pos) mx += gain_tab[l] / log2((double)pos + 2.0);
    double dcg = 0.0;
    for (int r = lane; r < k; r += 64) {
      int l = (int)label[sorted[b + r]];
      l = l < 0 ? 0 : (l >= n_gain ? n_gain - 1 : l);
      dcg += gain_tab[l] / log2((double)r + 2.0);
    }
    dcg = wave_sum_d(dcg);
    if (lane == 0) out[(size_t)q * nk + t] = mx > 0.0 ? dcg / mx : 1.0;
  }
}

uint64_t splitmix64_h(uint64_t x) { return rihip_splitmix64(x); }

struct Tree {
  std::vector<int> feat, bin, lc, rc, dtype; std::vector<long long> cnt_int;   // dtype: LightGBM decision_type per node
  std::vector<double> thr, gain, leaf, int_val, int_w, leaf_w; std::vector<long long> leaf_cnt;
};

struct Dataset {
  int64_t n = 0; int ng = 0;
  uint8_t* Xb = nullptr; float* y = nullptr; double* score = nullptr; int64_t* goff = nullptr;
  uint64_t *key = nullptr, *key2 = nullptr; int *val = nullptr, *sorted = nullptr; void* temp = nullptr; size_t temp_bytes = 0;
  int* goff32b = nullptr; int* goff32e = nullptr;
  double* ndcg = nullptr;
  void release() {
    hipFree(Xb); hipFree(y); hipFree(score); hipFree(goff); hipFree(key); hipFree(key2); hipFree(val); hipFree(sorted);
    hipFree(temp); hipFree(goff32b); hipFree(goff32e); hipFree(ndcg);
  }
};

int make_dataset(Dataset* D, const float* X, const float* y, const int32_t* groups, int64_t n, int F, int ng,
                 const double* d_ub, const int* d_nb, const int* d_nanbin, int nk, hipStream_t st) {
  D->n = n; D->ng = ng;
  std::vector<int64_t> off(ng + 1, 0);
  std::vector<int> ob(ng), oe(ng);
  for (int q = 0; q < ng; ++q) {
    RIHIP_REQUIRE(groups[q] > 0 && groups[q] <= MAX_GROUP, RIHIP_ERR_SHAPE, "lambdamart: query group of %d documents (1..%d supported)", groups[q], MAX_GROUP);
    off[q + 1] = off[q] + groups[q];
    ob[q] = (int)off[q]; oe[q] = (int)off[q + 1];
  }
  RIHIP_REQUIRE(off[ng] == n && n < (1ll << 31), RIHIP_ERR_ARG, "lambdamart: group sizes sum to %lld, n = %lld", (long long)off[ng], (long long)n);
  TCHK(hipMalloc((void**)&D->Xb, (size_t)n * F));
  TCHK(hipMalloc((void**)&D->y, sizeof(float) * n));
  TCHK(hipMalloc((void**)&D->score, sizeof(double) * n));
  TCHK(hipMalloc((void**)&D->goff, sizeof(int64_t) * (ng + 1)));
  TCHK(hipMalloc((void**)&D->goff32b, sizeof(int) * ng));
  TCHK(hipMalloc((void**)&D->goff32e, sizeof(int) * ng));
  TCHK(hipMalloc((void**)&D->key, sizeof(uint64_t) * n));
  TCHK(hipMalloc((void**)&D->key2, sizeof(uint64_t) * n));
  TCHK(hipMalloc((void**)&D->val, sizeof(int) * n));
  TCHK(hipMalloc((void**)&D->sorted, sizeof(int) * n));
  TCHK(hipMalloc((void**)&D->ndcg, sizeof(double) * (size_t)ng * nk));
  TCHK(hipMemcpyAsync(D->y, y, sizeof(float) * n, hipMemcpyDeviceToDevice, st));
  TCHK(hipMemsetAsync(D->score, 0, sizeof(double) * n, st));
  TCHK(hipMemcpyAsync(D->goff, off.data(), sizeof(int64_t) * (ng + 1), hipMemcpyHostToDevice, st));
  TCHK(hipMemcpyAsync(D->goff32b, ob.data(), sizeof(int) * ng, hipMemcpyHostToDevice, st));
  TCHK(hipMemcpyAsync(D->goff32e, oe.data(), sizeof(int) * ng, hipMemcpyHostToDevice, st));
  TCHK(hipStreamSynchronize(st));
  hipLaunchKernelGGL(bin_rows_kernel, dim3((unsigned)((n * F + 255) / 256)), dim3(256), 0, st, X, n, F, d_ub, d_nb, d_nanbin, D->Xb);
  TCHK(hipGetLastError());
  TCHK(rocprim::segmented_radix_sort_pairs(nullptr, D->temp_bytes, D->key, D->key2, D->val, D->sorted, (unsigned)n, (unsigned)ng,
                                           D->goff32b, D->goff32e, 0, 64, st));
  TCHK(hipMalloc(&D->temp, D->temp_bytes ? D->temp_bytes : 16));
  return RIHIP_OK;
}
// stable sort of every query group by descending score -> D->sorted (document index per rank position)
int sort_groups(Dataset* D, hipStream_t st) {
  hipLaunchKernelGGL(sort_keys_kernel, dim3((unsigned)((D->n + 255) / 256)), dim3(256), 0, st, D->score, D->n, D->key, D->val);
  size_t tb = D->temp_bytes;
  TCHK(rocprim::segmented_radix_sort_pairs(D->temp, tb, D->key, D->key2, D->val, D->sorted, (unsigned)D->n, (unsigned)D->ng,
                                           D->goff32b, D->goff32e, 0, 64, st));
  return RIHIP_OK;
}
int eval_ndcg(Dataset* D, const double* d_gain, int n_gain, const int* d_ks, int nk, std::vector<double>* out, hipStream_t st) {
  hipLaunchKernelGGL(ndcg_kernel, dim3(D->ng), dim3(64), 0, st, D->y, D->sorted, D->goff, d_gain, n_gain, d_ks, nk, D->ndcg);
  std::vector<double> h((size_t)D->ng * nk);
  TCHK(hipMemcpyAsync(h.data(), D->ndcg, sizeof(double) * h.size(), hipMemcpyDeviceToHost, st));
  TCHK(hipStreamSynchronize(st));
  out->assign(nk, 0.0);
  for (int q = 0; q < D->ng; ++q)
    for (int t = 0; t < nk; ++t) (*out)[t] += h[(size_t)q * nk + t];   // query order: reproducible
  for (int t = 0; t < nk; ++t) (*out)[t] /= (double)D->ng;
  return RIHIP_OK;
}

void fmt_double(std::string* s, double v) {
  char buf[40];
  snprintf(buf, sizeof(buf), "%.17g", v);
  s->append(buf);
}

std::string tree_to_text(const Tree& t, int index, double shrinkage) {
  std::string s;
  const int nl = (int)t.leaf.size();
  auto ints = [&](const char* k, const std::vector<int>& v) { s += k; s += "="; for (size_t i = 0; i < v.size(); ++i) { if (i) s += " "; s += std::to_string(v[i]); } s += "\n"; };
  auto lls = [&](const char* k, const std::vector<long long>& v) { s += k; s += "="; for (size_t i = 0; i < v.size(); ++i) { if (i) s += " "; s += std::to_string(v[i]); } s += "\n"; };
  auto dbl = [&](const char* k, const std::vector<double>& v) { s += k; s += "="; for (size_t i = 0; i < v.size(); ++i) { if (i) s += " "; fmt_double(&s, v[i]); } s += "\n"; };
  s += "Tree=" + std::to_string(index) + "\n";
  s += "num_leaves=" + std::to_string(nl) + "\n";
  s += "num_cat=0\n";
  if (nl > 1) {
    ints("split_feature", t.feat);
    dbl("split_gain", t.gain);
    dbl("threshold", t.thr);
    ints("decision_type", t.dtype);
    ints("left_child", t.lc);
    ints("right_child", t.rc);
  }
  dbl("leaf_value", t.leaf);
  if (nl > 1) {
    dbl("leaf_weight", t.leaf_w);
    lls("leaf_count", t.leaf_cnt);
    dbl("internal_value", t.int_val);
    dbl("internal_weight", t.int_w);
    lls("internal_count", t.cnt_int);
  }
  s += "is_linear=0\n";
  s += "shrinkage="; fmt_double(&s, shrinkage); s += "\n\n\n";
  return s;
}

}  // namespace

extern "C" void rihip_free(void* p) { free(p); }

extern "C" int rihip_lambdamart_train(const float* X, const float* y, const int32_t* groups, int64_t n, int F, int ng,
                                      const float* Xv, const float* yv, const int32_t* groups_v, int64_t nv, int ngv,
                                      const rihip_lambdamart_params* p, const char* feature_names, char** model_text,
                                      int* best_iteration, int* n_rounds, double* history, void* stream) {
  RIHIP_REQUIRE(X && y && groups && p && model_text && n > 0 && F > 0 && F <= 255 && ng > 0, RIHIP_ERR_ARG, "lambdamart_train: bad arguments");
  RIHIP_REQUIRE(p->num_leaves >= 2 && p->num_leaves <= 128 && p->n_estimators >= 1 && p->max_bin >= 2 && p->max_bin <= 255,
                RIHIP_ERR_ARG, "lambdamart_train: num_leaves in [2,128], max_bin in [2,255]");
  RIHIP_REQUIRE(p->truncation_level >= 1 && p->truncation_level <= MAX_T, RIHIP_ERR_ARG, "lambdamart_train: truncation_level in [1,%d]", MAX_T);
  RIHIP_REQUIRE(p->n_eval_at >= 1 && p->n_eval_at <= 8 && p->n_label_gain >= 2 && p->n_label_gain <= 32, RIHIP_ERR_ARG, "lambdamart_train: eval_at / label_gain sizes");
  RIHIP_REQUIRE(p->hist_bits == 0 || p->hist_bits == 20 || p->hist_bits == 40, RIHIP_ERR_ARG, "lambdamart_train: hist_bits = 20 or 40");
  hipStream_t st = (hipStream_t)stream;
  const bool has_valid = Xv && yv && groups_v && nv > 0 && ngv > 0;
  const int nk = p->n_eval_at;
  // 2^40 levels: the sums of n values stay below 2^62 (fewer levels for more than 2^22 rows)
  int bits = p->hist_bits == 40 ? 40 : 20;
  if (bits == 40) { int lg = 0; while ((1ll << lg) < n) ++lg; if (62 - lg < bits) bits = 62 - lg; }
  const bool wide = bits > 20;
  const double qlevels = ldexp(1.0, bits);
  const bool use_missing = p->use_missing != 0;
  const int split_order = p->split_order != 0 ? 1 : 0;

  // ---- bin upper bounds from a strided sample (host)
  const int64_t step = std::max<int64_t>(1, (n + p->bin_sample - 1) / p->bin_sample);
  const int64_t ns = (n + step - 1) / step;
  std::vector<float> sample((size_t)ns * F);
  TCHK(hipMemcpy2DAsync(sample.data(), sizeof(float) * F, X, sizeof(float) * F * step, sizeof(float) * F, ns, hipMemcpyDeviceToHost, st));
  TCHK(hipStreamSynchronize(st));
  std::vector<double> ub((size_t)F * NBIN, INFINITY);
  std::vector<int> nb(F, 1), nanbin(F, -1);
  {
    std::vector<double> col(ns), u; std::vector<int64_t> c;
    for (int f = 0; f < F; ++f) {
      size_t m = 0;
      for (int64_t i = 0; i < ns; ++i) { const double v = (double)sample[(size_t)i * F + f]; if (v == v) col[m++] = v; }
      const bool has_nan = use_missing && m < (size_t)ns;     // LightGBM: missing type NaN iff the bin sample holds one
      const int max_real = has_nan ? p->max_bin - 1 : p->max_bin;   // the missing rows take the last bin
      std::sort(col.begin(), col.begin() + m);
      u.clear(); c.clear();
      for (size_t i = 0; i < m; ++i) { if (u.empty() || col[i] != u.back()) { u.push_back(col[i]); c.push_back(1); } else c.back()++; }
      double* dst = ub.data() + (size_t)f * NBIN;
      int k = 0;
      if (u.size() > 1) {
        if ((int)u.size() <= max_real) {
          for (size_t i = 0; i + 1 < u.size(); ++i) dst[k++] = (u[i] + u[i + 1]) * 0.5;
        } else {
          std::vector<int64_t> cum(c.size());
          int64_t run = 0;
          for (size_t i = 0; i < c.size(); ++i) { run += c[i]; cum[i] = run; }
          const int64_t tot = run;
          int64_t last = -1;
          for (int b = 1; b < max_real; ++b) {
            const int64_t want = (tot * b + max_real - 1) / max_real;
            int64_t i = std::lower_bound(cum.begin(), cum.end(), want) - cum.begin();
            i = std::min<int64_t>(i, (int64_t)u.size() - 2);
            if (i > last) { dst[k++] = (u[i] + u[i + 1]) * 0.5; last = i; }
          }
        }
      }
      dst[k++] = INFINITY;
      if (has_nan) { nanbin[f] = k; dst[k++] = NAN; }
      nb[f] = k;
    }
  }
  double* d_ub = nullptr; int* d_nb = nullptr; double* d_gain = nullptr; int* d_ks = nullptr; int* d_nanbin = nullptr;
  TCHK(hipMalloc((void**)&d_ub, sizeof(double) * ub.size()));
  TCHK(hipMalloc((void**)&d_nb, sizeof(int) * F));
  TCHK(hipMalloc((void**)&d_nanbin, sizeof(int) * F));
  TCHK(hipMemcpy(d_nanbin, nanbin.data(), sizeof(int) * F, hipMemcpyHostToDevice));
  TCHK(hipMalloc((void**)&d_gain, sizeof(double) * p->n_label_gain));
  TCHK(hipMalloc((void**)&d_ks, sizeof(int) * nk));
  TCHK(hipMemcpy(d_ub, ub.data(), sizeof(double) * ub.size(), hipMemcpyHostToDevice));
  TCHK(hipMemcpy(d_nb, nb.data(), sizeof(int) * F, hipMemcpyHostToDevice));
  TCHK(hipMemcpy(d_gain, p->label_gain, sizeof(double) * p->n_label_gain, hipMemcpyHostToDevice));
  TCHK(hipMemcpy(d_ks, p->eval_at, sizeof(int) * nk, hipMemcpyHostToDevice));

  Dataset T, V;
  int rc = make_dataset(&T, X, y, groups, n, F, ng, d_ub, d_nb, d_nanbin, nk, st);
  if (rc == RIHIP_OK && has_valid) rc = make_dataset(&V, Xv, yv, groups_v, nv, F, ngv, d_ub, d_nb, d_nanbin, nk, st);
  if (rc) { T.release(); V.release(); hipFree(d_ub); hipFree(d_nb); hipFree(d_nanbin); hipFree(d_gain); hipFree(d_ks); return rc; }

  // ---- training scratch
  const int NL = p->num_leaves;
  const size_t HSZ = (size_t)F * NBIN * 3;
  double *ls = nullptr, *hs = nullptr, *lam = nullptr, *hes = nullptr, *d_leafv = nullptr;
  void *gq = nullptr, *hq = nullptr;      // int32 (2^20 levels) or int64 (up to 2^40 levels) per row
  int *rowsA = nullptr, *rowsB = nullptr, *d_cur = nullptr;
  int *d_feat = nullptr, *d_bin = nullptr, *d_lc = nullptr, *d_rc = nullptr;
  long long* hist = nullptr; long long* d_sum = nullptr; unsigned long long* d_mx = nullptr; unsigned char* d_used = nullptr;
  SplitInfo* d_split = nullptr; SplitInfo* d_featbest = nullptr; unsigned* d_done = nullptr;
  hipMalloc((void**)&ls, sizeof(double) * n); hipMalloc((void**)&hs, sizeof(double) * n);
  hipMalloc((void**)&lam, sizeof(double) * n); hipMalloc((void**)&hes, sizeof(double) * n);
  hipMalloc(&gq, (wide ? 8 : 4) * (size_t)n); hipMalloc(&hq, (wide ? 8 : 4) * (size_t)n);
  hipMalloc((void**)&rowsA, sizeof(int) * n); hipMalloc((void**)&rowsB, sizeof(int) * n);
  hipMalloc((void**)&d_cur, sizeof(int) * 2 * 128);      // a cursor pair per split of a tree (num_leaves <= 128)
  hipMalloc((void**)&hist, sizeof(long long) * HSZ * (size_t)NL); hipMalloc((void**)&d_sum, sizeof(long long) * 2);
  hipMalloc((void**)&d_mx, sizeof(unsigned long long) * 2); hipMalloc((void**)&d_used, F);
  hipMalloc((void**)&d_split, sizeof(SplitInfo) * 2);
  hipMalloc((void**)&d_featbest, sizeof(SplitInfo) * 2 * (size_t)F); hipMalloc((void**)&d_done, sizeof(unsigned));
  if (d_done) hipMemsetAsync(d_done, 0, sizeof(unsigned), st);
  hipMalloc((void**)&d_feat, sizeof(int) * NL); hipMalloc((void**)&d_bin, sizeof(int) * NL);
  hipMalloc((void**)&d_lc, sizeof(int) * NL); hipMalloc((void**)&d_rc, sizeof(int) * NL); hipMalloc((void**)&d_leafv, sizeof(double) * NL);
  if (hipGetLastError() != hipSuccess || !d_cur || !d_leafv) { rihip_set_error("lambdamart_train: device allocation failed"); return RIHIP_ERR_HIP; }
  const size_t grad_lds = [&] { int mg = 1; for (int q = 0; q < ng; ++q) mg = std::max(mg, (int)groups[q]); return (size_t)mg * 9 + 64; }();
  static bool granted = false;
  if (!granted) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lambdarank_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, MAX_GROUP * 9 + 64);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(hist_kernel<int>), hipFuncAttributeMaxDynamicSharedMemorySize, FCH * NBIN * 3 * 4);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(hist_kernel<long long>), hipFuncAttributeMaxDynamicSharedMemorySize, FCH * NBIN * 3 * 4);
    granted = true;
  }

  auto build_hist = [&](int* rows, int64_t b0, int64_t len, long long* dst) -> int {
    TCHK(hipMemsetAsync(dst, 0, sizeof(long long) * HSZ, st));
    if (len <= 0) return RIHIP_OK;
    const unsigned nchunk = (unsigned)((len + HCH - 1) / HCH);
    const int fch = wide ? FCH / 2 : FCH;     // the same 30 KB of LDS with 8-byte cells
    const dim3 grid(nchunk, (unsigned)((F + fch - 1) / fch));
    if (wide)
      hipLaunchKernelGGL(hist_kernel<long long>, grid, dim3(256), (size_t)fch * NBIN * 3 * 8, st, T.Xb, F, rows, b0, len,
                         (const long long*)gq, (const long long*)hq, fch, dst);
    else
      hipLaunchKernelGGL(hist_kernel<int>, grid, dim3(256), (size_t)fch * NBIN * 3 * 4, st, T.Xb, F, rows, b0, len,
                         (const int*)gq, (const int*)hq, fch, dst);
    TCHK(hipGetLastError());
    return RIHIP_OK;
  };
  double sg = 0.0, sh = 0.0;
  // best splits of up to two leaves (the children of a split) in one launch; a null histogram = leaf not tried
  auto find_splits = [&](const long long* h0, const long long* h1, SplitInfo* slots) {
    hipLaunchKernelGGL(split_kernel, dim3((unsigned)((F + 3) / 4), h1 ? 2 : 1), dim3(256), 0, st, h0, h1, F, d_nb, d_nanbin, d_used, sg, sh, p->reg_alpha,
                       p->reg_lambda, p->min_child_samples, p->min_sum_hessian, split_order, slots, (const long long*)nullptr, -1, d_featbest, d_done);
  };
  SplitInfo* h_split = nullptr;   // pinned: the per-split read-back is on the critical path of the tree growth
  if (hipHostMalloc((void**)&h_split, sizeof(SplitInfo) * 2) != hipSuccess) h_split = nullptr;

  std::vector<Tree> trees;
  std::vector<double> hist_rows;      // [round][2][nk]
  std::vector<double> best_val; std::vector<int> best_itr;
  int best_it = 0, rounds = 0;
  const int n_used = p->feature_fraction < 1.0 ? std::max(1, (int)(F * p->feature_fraction + 0.5)) : F;
  GradArgs ga;
  ga.score = T.score; ga.label = T.y; ga.sorted = T.sorted; ga.goff = T.goff; ga.ng = ng; ga.gain_tab = d_gain;
  ga.n_gain = p->n_label_gain; ga.sigmoid = p->sigmoid; ga.T = p->truncation_level; ga.norm = p->lambdarank_norm; ga.lam = ls; ga.hes = hs;
  bool stop = false;
  for (int it = 0; it <= p->n_estimators && rc == RIHIP_OK; ++it) {
    rc = sort_groups(&T, st);
    if (rc) break;
    if (it > 0) {   // metrics of the model with `it` trees
      std::vector<double> tr, va;
      rc = eval_ndcg(&T, d_gain, p->n_label_gain, d_ks, nk, &tr, st);
      if (rc == RIHIP_OK && has_valid) { rc = sort_groups(&V, st); if (rc == RIHIP_OK) rc = eval_ndcg(&V, d_gain, p->n_label_gain, d_ks, nk, &va, st); }
      if (rc) break;
      for (int t = 0; t < nk; ++t) hist_rows.push_back(tr[t]);
      for (int t = 0; t < nk; ++t) hist_rows.push_back(has_valid ? va[t] : NAN);
      rounds = it;
      if (has_valid) {
        // lightgbm.early_stopping semantics (callback.py, first_metric_only=False): every validation metric keeps its
        // own best score / best iteration; the metrics are visited in order and the FIRST one whose patience ran out
        // stops the run and names best_iteration; a run that reaches n_estimators reports metric 0's best iteration
        if (best_val.empty()) { best_val = va; best_itr.assign(nk, it); best_it = it; }
        else for (int t = 0; t < nk && !stop; ++t) {
          if (va[t] > best_val[t]) { best_val[t] = va[t]; best_itr[t] = it; }
          else if (it - best_itr[t] >= p->early_stopping_rounds) { stop = true; best_it = best_itr[t]; }
        }
        if (!stop) best_it = best_itr[0];
      } else best_it = it;
    }
    if (it == p->n_estimators || stop) break;
    // ---- gradients
    hipLaunchKernelGGL(lambdarank_kernel, dim3(ng), dim3(256), grad_lds, st, ga);
    hipMemsetAsync(d_mx, 0, sizeof(unsigned long long) * 2, st);
    hipLaunchKernelGGL(unsort_absmax_kernel, dim3((unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048)), dim3(256), 0, st, ls, hs, T.sorted, n, lam, hes, d_mx);
    unsigned long long mxb[2];
    if (hipMemcpyAsync(mxb, d_mx, sizeof(mxb), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { rihip_set_error("lambdamart_train: gradient pass failed: %s", hipGetErrorString(hipGetLastError())); rc = RIHIP_ERR_HIP; break; }
    double gm, hm; memcpy(&gm, &mxb[0], 8); memcpy(&hm, &mxb[1], 8);
    sg = gm > 0.0 ? qlevels / gm : 0.0; sh = hm > 0.0 ? qlevels / hm : 0.0;
    if (wide) hipLaunchKernelGGL(quantize_kernel<long long>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, lam, hes, n, sg, sh,
                                 (long long*)gq, (long long*)hq);
    else hipLaunchKernelGGL(quantize_kernel<int>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, lam, hes, n, sg, sh, (int*)gq, (int*)hq);
    // ---- features of this tree
    {
      std::vector<std::pair<uint64_t, int>> hv(F);
      for (int f = 0; f < F; ++f) hv[f] = {splitmix64_h(splitmix64_h(p->seed + 1000003ull * (uint64_t)it) ^ (uint64_t)f), f};
      std::stable_sort(hv.begin(), hv.end(), [](const std::pair<uint64_t, int>& a, const std::pair<uint64_t, int>& b) { return a.first < b.first; });
      std::vector<unsigned char> used(F, 0);
      for (int k = 0; k < n_used; ++k) used[hv[k].second] = 1;
      hipMemcpyAsync(d_used, used.data(), F, hipMemcpyHostToDevice, st);
      hipStreamSynchronize(st);
    }
    // ---- grow the tree (best-first): leaf l owns rows[b .. b+len)
    struct Leaf { int64_t b, len; int parent_node, side, slot, buf; SplitInfo s; bool has; long long gq, hq, c; };
    int* rowsBuf[2] = {rowsA, rowsB};    // a leaf's rows live in rowsBuf[leaf.buf]; children get the other buffer
    hipMemsetAsync(d_cur, 0, sizeof(int) * 2 * 128, st);
    std::vector<Leaf> leaves;
    Tree tree;
    hipLaunchKernelGGL(iota_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, rowsA, n);
    int n_slots = 1;
    auto read_splits = [&](SplitInfo* dst, int cntS) -> int {
      SplitInfo* via = h_split ? h_split : dst;
      TCHK(hipMemcpyAsync(via, d_split, sizeof(SplitInfo) * cntS, hipMemcpyDeviceToHost, st));
      TCHK(hipStreamSynchronize(st));
      if (via != dst) memcpy(dst, via, sizeof(SplitInfo) * cntS);
      return RIHIP_OK;
    };
    {
      Leaf root; root.b = 0; root.len = n; root.parent_node = -1; root.side = 0; root.slot = 0; root.buf = 0; root.has = false;
      rc = build_hist(rowsA, 0, n, hist);
      if (rc) break;
      find_splits(hist, nullptr, d_split);
      rc = read_splits(&root.s, 1);
      if (rc) break;
      root.has = root.s.feature >= 0;
      if (root.has) { root.gq = root.s.gq; root.hq = root.s.hq; root.c = root.s.c; }
      else {
        long long sums[2] = {0, 0};
        hipMemsetAsync(d_sum, 0, sizeof(long long) * 2, st);
        if (wide) hipLaunchKernelGGL(sum_leaf_kernel<long long>, dim3(256), dim3(256), 0, st, rowsA, 0, n, (const long long*)gq, (const long long*)hq, d_sum);
        else hipLaunchKernelGGL(sum_leaf_kernel<int>, dim3(256), dim3(256), 0, st, rowsA, 0, n, (const int*)gq, (const int*)hq, d_sum);
        hipMemcpyAsync(sums, d_sum, sizeof(sums), hipMemcpyDeviceToHost, st);
        hipStreamSynchronize(st);
        root.gq = sums[0]; root.hq = sums[1]; root.c = n;
      }
      leaves.push_back(root);
    }
    while ((int)leaves.size() < NL && rc == RIHIP_OK) {
      int pick = -1;
      for (int l = 0; l < (int)leaves.size(); ++l)
        if (leaves[l].has && (pick < 0 || leaves[l].s.gain > leaves[pick].s.gain)) pick = l;   // ties: smallest leaf index
      if (pick < 0) break;
      Leaf par = leaves[pick];
      const SplitInfo sp = par.s;
      const int node = (int)tree.feat.size();
      const int f_nan = nanbin[sp.feature];
      tree.feat.push_back(sp.feature); tree.bin.push_back(sp.bin | (sp.default_left ? (1 << 16) : 0));
      {
        // "every real value left | missing right" splits at the last real bin, whose bound is +inf: written as the largest
        // double (x <= DBL_MAX holds for every finite x; the text format has no inf)
        const double th = ub[(size_t)sp.feature * NBIN + sp.bin];
        tree.thr.push_back(th == INFINITY ? 1.7976931348623157e308 : th);
      }
      // decision_type: bit 1 = default_left, bits 2-3 = missing type (0 none, 2 NaN); a feature without a missing bin is
      // written like LightGBM writes it: default left, missing none (2)
      tree.dtype.push_back(f_nan >= 0 ? (8 | (sp.default_left ? 2 : 0)) : 2);
      tree.gain.push_back(sp.gain);
      tree.lc.push_back(~pick); tree.rc.push_back(~(int)leaves.size());
      tree.cnt_int.push_back(sp.c);
      {
        const double G = sg > 0.0 ? (double)sp.gq / sg : 0.0, H = sh > 0.0 ? (double)sp.hq / sh : 0.0;
        const double t1 = fabs(G) - p->reg_alpha;
        const double thr1 = t1 > 0.0 ? (G > 0.0 ? t1 : -t1) : 0.0;
        tree.int_val.push_back(-thr1 / (H + p->reg_lambda) * p->learning_rate);
        tree.int_w.push_back(H);
      }
      if (par.parent_node >= 0) (par.side == 0 ? tree.lc : tree.rc)[par.parent_node] = node;
      // partition the rows of the leaf into the other row buffer: left = bin <= threshold bin (one launch)
      hipLaunchKernelGGL(part_kernel, dim3((unsigned)((par.len + PART_ROWS - 1) / PART_ROWS)), dim3(1024), 0, st, T.Xb, F,
                         rowsBuf[par.buf], rowsBuf[1 - par.buf], par.b, par.len, sp.feature, sp.bin, f_nan, sp.default_left,
                         d_cur + 2 * node, (int64_t)sp.cl);
      Leaf L = par, R = par;
      L.len = sp.cl; R.b = par.b + sp.cl; R.len = par.len - sp.cl;
      L.parent_node = node; L.side = 0; R.parent_node = node; R.side = 1;
      L.buf = R.buf = 1 - par.buf;
      L.gq = sp.glq; L.hq = sp.hlq; L.c = sp.cl; R.gq = sp.gq - sp.glq; R.hq = sp.hq - sp.hlq; R.c = sp.c - sp.cl;
      // histograms: the smaller child is built, the larger one is parent - smaller (kept in the parent's slot)
      const bool left_small = L.len <= R.len;
      Leaf& S = left_small ? L : R;
      Leaf& Bg = left_small ? R : L;
      S.slot = n_slots++; Bg.slot = par.slot;
      rc = build_hist(rowsBuf[S.buf], S.b, S.len, hist + (size_t)S.slot * HSZ);
      if (rc) break;
      const bool tryL = L.len >= 2 * (int64_t)p->min_child_samples, tryR = R.len >= 2 * (int64_t)p->min_child_samples;
      SplitInfo two[2];
      memset(two, 0, sizeof(two));
      two[0].feature = two[1].feature = -1;
      if (tryL || tryR) {
        // (grid of 2 whenever the right child is tried: a null left histogram makes workgroup 0 return at once)
        const long long* hL = tryL ? hist + (size_t)L.slot * HSZ : nullptr;
        const long long* hR = tryR ? hist + (size_t)R.slot * HSZ : nullptr;
        // the larger child's histogram = parent - smaller child is formed by its own split workgroup (in place, in the
        // parent's slot); a larger child that is not tried is a final leaf: its histogram is never read again
        const long long* hS = hist + (size_t)S.slot * HSZ;
        const int big_child = left_small ? 1 : 0;
        hipLaunchKernelGGL(split_kernel, dim3((unsigned)((F + 3) / 4), tryR ? 2 : 1), dim3(256), 0, st, hL, hR, F, d_nb, d_nanbin, d_used, sg, sh, p->reg_alpha,
                           p->reg_lambda, p->min_child_samples, p->min_sum_hessian, split_order, d_split, hS, big_child, d_featbest, d_done);
        rc = read_splits(two, 2);
        if (rc) break;
      }
      L.s = two[0]; R.s = two[1];
      L.has = tryL && two[0].feature >= 0; R.has = tryR && two[1].feature >= 0;
      leaves[pick] = L;
      leaves.push_back(R);
    }
    if (rc) break;
    // ---- leaf values, score updates
    const int nl = (int)leaves.size();
    tree.leaf.resize(nl); tree.leaf_w.resize(nl); tree.leaf_cnt.resize(nl);
    for (int l = 0; l < nl; ++l) {
      const double G = sg > 0.0 ? (double)leaves[l].gq / sg : 0.0, H = sh > 0.0 ? (double)leaves[l].hq / sh : 0.0;
      const double t1 = fabs(G) - p->reg_alpha;
      const double thr1 = t1 > 0.0 ? (G > 0.0 ? t1 : -t1) : 0.0;
      tree.leaf[l] = -thr1 / (H + p->reg_lambda) * p->learning_rate;
      tree.leaf_w[l] = H; tree.leaf_cnt[l] = leaves[l].c;
      if (leaves[l].len > 0)
        hipLaunchKernelGGL(add_leaf_kernel, dim3((unsigned)((leaves[l].len + 255) / 256)), dim3(256), 0, st, rowsBuf[leaves[l].buf], leaves[l].b,
                           leaves[l].len, tree.leaf[l], T.score);
    }
    if (has_valid) {
      const int nn = (int)tree.feat.size();
      if (nn > 0) {
        hipMemcpyAsync(d_feat, tree.feat.data(), sizeof(int) * nn, hipMemcpyHostToDevice, st);
        hipMemcpyAsync(d_bin, tree.bin.data(), sizeof(int) * nn, hipMemcpyHostToDevice, st);
        hipMemcpyAsync(d_lc, tree.lc.data(), sizeof(int) * nn, hipMemcpyHostToDevice, st);
        hipMemcpyAsync(d_rc, tree.rc.data(), sizeof(int) * nn, hipMemcpyHostToDevice, st);
      }
      hipMemcpyAsync(d_leafv, tree.leaf.data(), sizeof(double) * nl, hipMemcpyHostToDevice, st);
      hipLaunchKernelGGL(tree_add_kernel, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, st, V.Xb, nv, F, d_feat, d_bin, d_lc, d_rc,
                         d_leafv, d_nanbin, nn, V.score);
      hipStreamSynchronize(st);   // the host vectors above must outlive the copies
    }
    if (hipGetLastError() != hipSuccess) { rihip_set_error("lambdamart_train: tree %d launch failed", it); rc = RIHIP_ERR_HIP; break; }
    trees.push_back(std::move(tree));
  }
  hipStreamSynchronize(st);
  hipFree(ls); hipFree(hs); hipFree(lam); hipFree(hes); hipFree(gq); hipFree(hq); hipFree(rowsA); hipFree(rowsB); hipFree(d_cur);
  hipFree(hist); hipFree(d_sum); hipFree(d_mx); hipFree(d_used); hipFree(d_split); hipFree(d_featbest); hipFree(d_done);
  if (h_split) hipHostFree(h_split);
  hipFree(d_feat); hipFree(d_bin); hipFree(d_lc); hipFree(d_rc); hipFree(d_leafv);
  T.release(); V.release(); hipFree(d_ub); hipFree(d_nb); hipFree(d_nanbin); hipFree(d_gain); hipFree(d_ks);
  if (rc) return rc;

  // ---- LightGBM text model (Booster.save_model layout, src/models/ranker.py:203-209)
  // with a validation set the model served and saved is the best iteration's (Booster.predict / save_model default to
  // best_iteration after lgb.early_stopping, ranker.py:129-138, :174, :209): drop the trees grown past it
  if (has_valid && best_it > 0 && (size_t)best_it < trees.size()) trees.resize((size_t)best_it);
  std::vector<std::string> blocks;
  for (size_t t = 0; t < trees.size(); ++t) blocks.push_back(tree_to_text(trees[t], (int)t, p->learning_rate));
  std::string names = feature_names ? feature_names : "";
  if (names.empty()) for (int f = 0; f < F; ++f) { if (f) names += " "; names += "Column_" + std::to_string(f); }
  std::string txt = "tree\nversion=v4\nnum_class=1\nnum_tree_per_iteration=1\nlabel_index=0\n";
  txt += "max_feature_idx=" + std::to_string(F - 1) + "\nobjective=lambdarank\nfeature_names=" + names + "\nfeature_infos=";
  for (int f = 0; f < F; ++f) { if (f) txt += " "; txt += "[-1e+30:1e+30]"; }
  txt += "\ntree_sizes=";
  for (size_t t = 0; t < blocks.size(); ++t) { if (t) txt += " "; txt += std::to_string(blocks[t].size()); }
  txt += "\n\n";
  for (const std::string& b : blocks) txt += b;
  txt += "end of trees\n\nfeature_importances:\n\nparameters:\n[boosting: gbdt]\n[objective: lambdarank]\n";
  txt += "[num_leaves: " + std::to_string(p->num_leaves) + "]\n[learning_rate: "; fmt_double(&txt, p->learning_rate);
  txt += "]\nend of parameters\n\npandas_categorical:null\n";
  char* out = (char*)malloc(txt.size() + 1);
  RIHIP_REQUIRE(out, RIHIP_ERR_ARG, "lambdamart_train: out of host memory");
  memcpy(out, txt.c_str(), txt.size() + 1);
  *model_text = out;
  if (best_iteration) *best_iteration = best_it > 0 ? best_it : (int)trees.size();
  if (n_rounds) *n_rounds = rounds;
  if (history) memcpy(history, hist_rows.data(), sizeof(double) * hist_rows.size());
  return RIHIP_OK;
}
