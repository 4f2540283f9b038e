// Handle of the inner-product index shared by topk.hip (search) and ivf.hip (k-means, list layout, state I/O).
#pragma once
#include "common.h"

#include <vector>

namespace rihip_index {

constexpr int TR = 64;         // list granule: every IVF list is padded to a multiple of TR physical rows
constexpr int TRS = 32;        // corpus rows per LDS tile of the exact-f32 scan kernel
constexpr int K_MAX = 16384;   // largest k served by the device select/sort buffer (128 KiB of LDS)
constexpr int NLIST_MAX = 2048;  // probe bitset: 64 words per query

template <typename T>
struct DevBuf {
  T* p = nullptr;
  int64_t n = 0;
  int reserve(int64_t want) {
    if (n >= want) return RIHIP_OK;
    if (p) { hipFree(p); rihip_bump_generation(); }   // a graph that captured the old pointer is stale now
    p = nullptr; n = 0;
    if (hipMalloc((void**)&p, sizeof(T) * (size_t)want) != hipSuccess) {
      rihip_set_error("ip_index: device allocation of %lld bytes failed", (long long)(sizeof(T) * (size_t)want));
      return RIHIP_ERR_HIP;
    }
    n = want;
    return RIHIP_OK;
  }
  void release() { if (p) { hipFree(p); rihip_bump_generation(); } p = nullptr; n = 0; }
};

struct IpIndex {
  int d = 0;             // kernel width: 32 / 64 / 128 (the template instantiations)
  int du = 0;            // caller's embedding width (1 .. 128): rows are zero-padded to d inside the handle, which
                         // changes no inner product (the extra terms are exact zeros of the fmaf chain)
  int64_t N = 0;         // real vectors
  float* X = nullptr;    // brute force: [N,d]; IVF: [Np,d] list-ordered, zero-padded to 64-row granules
  __bf16* Xb = nullptr;  // bf16 copy of X for the filter pass (flat index, N > 4*SAMPLE)
  float max_norm = 0.f;  // max row 2-norm (error bound of the bf16 filter)
  int two_precision = 1; // 1: bf16 filter + exact f32 re-score; 0: all-f32 search
  const int64_t* id_map = nullptr;   // optional device array: search results are id_map[row] instead of row (not owned)
  DevBuf<float> qnorm;
  // IVF
  int nlist = 0, nprobe = 1;
  bool ivf = false;
  int64_t Np = 0;              // padded physical rows
  float* C = nullptr;          // [nlist,d]
  int* tile_list = nullptr;    // [Np/64] list of each granule
  int64_t* list_poff = nullptr; // [nlist+1] first physical row of each list (device)
  int* list_len_dev = nullptr;  // [nlist] real rows of each list (device)
  int64_t* row_ids = nullptr;  // [Np] original row per physical row, -1 for padding
  std::vector<int64_t> list_len;  // host copy
  // scratch (grown on demand, owned by the handle)
  DevBuf<uint64_t> cand, scand, fcand, seg;   // seg: per (query, corpus split) survivor segments of the bf16 filter
  DevBuf<int> seg_cnt;
  DevBuf<int> count, fail_flags, fail_list, n_fail, fcount;
  DevBuf<float> thr, thr2, fQ, qpad;    // qpad: queries / rows zero-padded from du to d columns
  DevBuf<float> coarse;                                    // IVF: coarse scores [nq,nlist]
  DevBuf<int> probe_list, list_q, list_cnt, list_qoff, list_cur, work_off, plan;
  int* h_nfail = nullptr;  // pinned
  // deferred exactness check (rihip_ip_index_set_deferred_check): a thresholded IVF search of at most one internal chunk
  // returns without its host synchronisation; rihip_ip_index_search_finish() reads the failure count and re-does the
  // failed queries (unfiltered pass) into the same output rows
  bool defer_check = false, defer_ok = false;
  struct { bool active = false; const float* Q = nullptr; int64_t nq = 0; int k = 0; float* out_s = nullptr; int64_t* out_r = nullptr; } pending;
  const int* redo_slots = nullptr;   // internal: search_chunk runs only the unfiltered IVF pass, results to these rows
  hipEvent_t ev_fail = nullptr;      // recorded behind the failure count's copy of a deferred search (not while capturing)
  bool ev_recorded = false;
};

inline void free_index_arrays(IpIndex* h) {
  rihip_bump_generation();
  hipFree(h->X); hipFree(h->C); hipFree(h->tile_list); hipFree(h->list_poff); hipFree(h->list_len_dev); hipFree(h->row_ids); hipFree(h->Xb);
  h->X = nullptr; h->C = nullptr; h->tile_list = nullptr; h->list_poff = nullptr; h->list_len_dev = nullptr; h->row_ids = nullptr; h->Xb = nullptr;
  h->N = 0; h->Np = 0; h->ivf = false; h->nlist = 0; h->list_len.clear();
}

// zero-pad n rows of width du (device) into rows of width d (device)
int pad_rows(const float* src, int64_t n, int du, int d, float* dst, hipStream_t st);
// flat index: (re)build the bf16 filter copy and the row-norm bound (topk.hip)
int prepare_flat(IpIndex* h, hipStream_t st);
// upload the per-list offsets/lengths the list-major scan reads (from the host copy list_len; ivf.hip)
int derive_ivf_aux(IpIndex* h, hipStream_t st);

}  // namespace rihip_index

#define HIPCHK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { rihip_set_error("%s:%d %s", __FILE__, __LINE__, hipGetErrorString(_e)); return RIHIP_ERR_HIP; } } while (0)
#define RCCHK(e) do { int _rc = (e); if (_rc) return _rc; } while (0)
