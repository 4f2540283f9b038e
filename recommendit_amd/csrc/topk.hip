// Inner-product top-K retrieval for gfx950 -- replaces the faiss calls behind
// FAISSIndex.search / batch_search (reference src/models/faiss_index.py:113, :145).
//
// Exact brute force, three launches per query batch, no score matrix in HBM:
//   (0) threshold estimate: score a strided sample of the corpus, radix-select the r-th
//       largest sample score per query  (r chosen so that P(#{score>=thr} < k) ~ 1e-5)
//   (1) scan: 4 waves x 32 register-stationary queries per workgroup; corpus tiles stream
//       through LDS once per 128 queries; S tile on exact-f32 MFMA; scores >= thr[q] are
//       appended (score,row) to the query's candidate list (rare: ~0.1-0.3 % of scores)
//   (2) finalize: one workgroup per query radix-selects the k best 64-bit keys
//       (orderable score << 32 | ~row: ties -> lowest row, total order) and bitonic-sorts them.
//   Queries whose candidate list under- or overflows (heavy ties / adversarial data) are
//   re-done exactly with thr=-inf and capacity N ("fallback"), so the result is always exact.
//
// IVF-Flat (IP): k-means lists built on device, corpus re-ordered list-contiguous and padded
// to the 64-row tile so a tile belongs to one list; the scan skips tiles whose list no query
// of the block probes and masks per query.
#include "common.h"
#include "recommendit_hip.h"

#include <algorithm>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

#include "ip_index.h"

using namespace rihip_index;

namespace {

constexpr int QB = 128;         // queries per workgroup (32 per wave)
constexpr int SAMPLE = 16384;   // corpus rows scored for the threshold estimate
constexpr int CSTRIDE = 32;     // IVF candidate counters: one 128-byte line per query (same-line atomics serialise in L2)

__device__ __forceinline__ uint32_t f2ord(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t o) {
  uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
  return __uint_as_float(u);
}
__device__ __forceinline__ uint64_t make_key(float s, uint32_t row) {
  return ((uint64_t)f2ord(s) << 32) | (uint64_t)(0xFFFFFFFFu - row);
}

struct ScanArgs {
  const float* X;        // corpus [N,d] (list-ordered for IVF)
  const void* Xb;        // bf16 copy of the corpus [N,d] (filter pass of the two-precision search), or null
  int64_t n_virtual;     // virtual rows scanned: row(i) = i * row_stride
  int64_t row_stride;
  const float* Q;        // [nq,d]
  int64_t nq;
  const float* thr;      // [nq] or null (=> -inf)
  uint64_t* cand;        // [nq, cap]
  int64_t cap;
  int* count;            // [nq]
  int nsplit;            // splits of the tile sequence (gridDim.y)
  int dense;             // 1: slot = virtual row (no atomics, count preset); 0: atomic append
  int qgrid;                  // bf16 filter: number of query blocks (1-D XCD-aware launch)
  // bf16 filter: every (query, corpus split) pair has ONE writer (a wave), so its survivors go to a private segment
  // with the fill count kept in LDS -- no global atomic in the scan (returning global atomics cost 0.64 of 1.8 ms)
  uint64_t* seg;              // [nq, nsplit, seg_cap] keys
  int* seg_cnt;               // [nq, nsplit] survivors found (may exceed seg_cap: the query is then re-done exactly)
  int seg_cap;
  int cs;                     // ints between two queries' candidate counters (0/1 = dense; CSTRIDE = a 128-B line each)
};


// 4 waves x 32 register-stationary queries share each 32-row corpus tile.  Same software pipeline as the
// in-batch sweep: 3 LDS buffers, tile t+2 prefetched through registers, the S chain of tile t+1 interleaved with
// the threshold test / candidate emission of tile t, one barrier per tile.
template <int D>
__global__ __launch_bounds__(256, 2) void scan_kernel(ScanArgs a) {
  constexpr int LDX = D + 4, KB = D / 8;
  constexpr int EPK = 16 / KB > 0 ? 16 / KB : 1;
  constexpr int NV = (TRS * (D / 4) + 255) / 256;
  __shared__ __attribute__((aligned(16))) float Xs[3][TRS * LDX];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int r31 = lane & 31, hh = lane >> 5;
  const int64_t q = (int64_t)blockIdx.x * QB + w * 32 + r31;
  const bool q_ok = q < a.nq;
  const int64_t qrow = q_ok ? q : (a.nq - 1);

  f32x4 qf[KB];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) qf[kb] = *reinterpret_cast<const f32x4*>(&a.Q[qrow * D + kb * 8 + 4 * hh]);
  const float thr = (a.thr && q_ok) ? a.thr[q] : -INFINITY;
  uint64_t* my_cand = a.cand + (size_t)qrow * a.cap;

  const int64_t n_seq = (a.n_virtual + TRS - 1) / TRS;
  const int64_t per = (n_seq + a.nsplit - 1) / a.nsplit;
  const int64_t i0 = (int64_t)blockIdx.y * per;
  const int64_t i1 = (i0 + per < n_seq) ? i0 + per : n_seq;
  if (i0 >= i1) return;  // uniform across the workgroup

  f32x4 stage[NV];
  auto tile_at = [&](int64_t i) -> int64_t { return i; };
  auto load_tile = [&](int64_t tile) {
    const int64_t v_base = tile * TRS;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = tid + i * 256;
      const int r = idx / (D / 4), c4 = idx % (D / 4);
      const int64_t v = v_base + r;
      f32x4 val = {0.f, 0.f, 0.f, 0.f};
      if (idx < TRS * (D / 4) && v < a.n_virtual)
        val = reinterpret_cast<const f32x4*>(a.X + (size_t)(v * a.row_stride) * D)[c4];
      stage[i] = val;
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = tid + i * 256;
      const int r = idx / (D / 4), c4 = idx % (D / 4);
      if (idx < TRS * (D / 4)) *reinterpret_cast<f32x4*>(&Xs[buf][r * LDX + c4 * 4]) = stage[i];
    }
  };
  auto emit = [&](const f32x16& acc, int64_t tile) {
    const int64_t v_base = tile * TRS;
    if (!q_ok) return;
    if (a.dense) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t v = v_base + acc_row(r, lane);
        if (v < a.n_virtual) my_cand[v] = make_key(acc[r], (uint32_t)v);
      }
      return;
    }
    const int n_ok = (a.n_virtual - v_base) < TRS ? (int)(a.n_virtual - v_base) : TRS;
    unsigned hits = 0;  // per-lane aggregation: one atomic per (query, tile) that has survivors
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (acc_row(r, lane) < n_ok && acc[r] >= thr) hits |= (1u << r);
    if (hits) {
      int pos = atomicAdd(&a.count[q * (a.cs > 1 ? a.cs : 1)], __popc(hits));
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (hits & (1u << r)) {
          const int64_t v = v_base + acc_row(r, lane);
          if (pos < a.cap) my_cand[pos] = make_key(acc[r], (uint32_t)v);
          ++pos;
        }
      }
    }
  };
  auto s_chain = [&](const float* Xt, f32x16& acc) {
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      const f32x4 av = *reinterpret_cast<const f32x4*>(&Xt[r31 * LDX + kb * 8 + 4 * hh]);
      acc = mfma32(av.x, qf[kb].x, acc);
      acc = mfma32(av.y, qf[kb].y, acc);
      acc = mfma32(av.z, qf[kb].z, acc);
      acc = mfma32(av.w, qf[kb].w, acc);
    }
  };

  load_tile(tile_at(i0));
  store_tile(0);
  if (i0 + 1 < i1) {
    load_tile(tile_at(i0 + 1));
    store_tile(1);
  }
  __syncthreads();
  f32x16 st = zero16();
  s_chain(Xs[0], st);

#pragma unroll 1
  for (int64_t i = i0; i < i1; ++i) {
    const int it = (int)((i - i0) % 3);
    const int nxt = (it + 1) % 3, pre = (it + 2) % 3;
    const bool has_next = (i + 1 < i1), has_pre = (i + 2 < i1);
    if (has_pre) load_tile(tile_at(i + 2));
    f32x16 sn = zero16();
    if (has_next) s_chain(Xs[nxt], sn);  // the compiler interleaves the (independent) emit below into this chain
    emit(st, tile_at(i));
    if (has_pre) store_tile(pre);
    st = sn;
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------------
// Two-precision exact search for large corpora: FILTER on plain-bf16 MFMA (16x fewer matrix cycles than exact
// f32), then RE-SCORE the few survivors in exact f32.  |s - s_bf16| <= (2^-8 + 2^-18) |q| |x| (each operand
// rounded to bf16 with relative error <= 2^-9), so rows outside the candidate set {s_bf16 >= thr} have exact
// score < thr + eps; if the k-th exact score of the candidates is >= thr + eps the top-k is proven complete,
// otherwise the query takes the exact-f32 fallback.  Results are therefore bit-identical to the all-f32 search.
// ---------------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

__global__ void to_bf16_kernel(const float* __restrict__ x, int64_t n, __bf16* __restrict__ y) {
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i + 3 < n) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + i);
    y[i] = (__bf16)v.x; y[i + 1] = (__bf16)v.y; y[i + 2] = (__bf16)v.z; y[i + 3] = (__bf16)v.w;
  } else {
    for (int64_t j = i; j < n; ++j) y[j] = (__bf16)x[j];
  }
}
// max over rows of |x_row|^2 (order-independent: max of non-negative floats via int atomicMax)
__global__ __launch_bounds__(256) void rownorm_max_kernel(const float* __restrict__ X, int64_t N, int d, int* out_bits) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float best = 0.f;
  for (int64_t row = (int64_t)blockIdx.x * 4 + w; row < N; row += (int64_t)gridDim.x * 4) {
    float s = 0.f;
    for (int k = lane; k < d; k += 64) { const float v = X[row * d + k]; s += v * v; }
    s = wave_sum(s);
    best = fmaxf(best, s);
  }
  if (lane == 0) atomicMax(out_bits, __float_as_int(best));
}

#ifndef RIHIP_SCAN_ABLATE
#define RIHIP_SCAN_ABLATE 0
#endif
constexpr int QBB = 256;  // queries per workgroup of the bf16 filter (64 per wave: two 32-query groups)

constexpr int TRB = 64;     // corpus rows per pipeline stage of the bf16 filter (two 32-row MFMA sub-tiles)
constexpr int WQE = 192;    // survivor queue: 16-byte entries (score, row, query) per WAVE (LDS)
constexpr int SAMPLE_T = 8;  // threshold sample: scores kept per stream (query, corpus split, row half)

__device__ __forceinline__ float max3_raw(float a, float b, float c) {  // no NaN-canonicalising pre-ops
  float m;
  asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(a), "v"(b), "v"(c));
  return m;
}

// DENSE=false: survivors (score >= thr[q]) are queued in LDS and flushed to the per-query candidate lists now and then,
// so the hot loop contains no global store/atomic (those make hipcc drain the in-flight prefetch with vmcnt(0)).
// MODE 1 (dense): every score is stored at slot = virtual row (threshold-sample pass for large k).
// MODE 2 (top-T sample): every lane keeps the SAMPLE_T best scores of its stream (query, split, half of the rows) in
// registers and writes only those: the r-th largest of the union is a LOWER bound of the sample's r-th largest (a
// subset can only lose large scores), i.e. a safe threshold, and the sample pass writes 100x less.
// (launch bounds: the filter (MODE 0) wants 3 workgroups per CU even at the price of 72 spilled registers -- 2 per CU
// measured 2.27 instead of 2.04 ms per batch; the short sample passes spilled 130-250 registers at that bound and run
// 1.6x faster with 2 per CU and none)
template <int D, int MODE>
__global__ __launch_bounds__(256, MODE == 0 ? 3 : 2) void scan_bf16_kernel(ScanArgs a) {
  constexpr bool DENSE = MODE != 0;
  constexpr int LDB = D + 8, KB = D / 16;
  constexpr int NV = (TRB * (D / 8) + 255) / 256;  // 16-byte pieces staged per thread per stage
  __shared__ __attribute__((aligned(16))) __bf16 Xs[2][TRB * LDB];
  // A survivor is rare per lane but not per 64-lane wave (a wave meets one in ~90 % of its 16-score columns at k = 500),
  // so the hot path must neither wait nor synchronise.  A column whose maximum passes the threshold is scanned score by
  // score; the lanes holding a survivor write ONE 16-byte entry (score, row, query) each into their WAVE's private queue
  // at a slot computed from the compare mask (no LDS atomic, no returning operation: with a workgroup-wide atomic
  // counter this path cost more than the MFMA work).  Until round 3 the whole 16-score column was dumped (4 x
  // ds_write_b128 + 2 header words per column, thresholded again in the flush): those six LDS instructions per column
  // loaded the LDS pipe as much as the MFMA operand reads did, and the 72-byte columns filled the queue nine times
  // sooner.
  __shared__ __attribute__((aligned(16))) uint4 qent[DENSE ? 1 : 4][DENSE ? 1 : WQE];
  __shared__ int qcntS[DENSE ? 1 : QBB];    // survivors of each query in this corpus split (one writer wave each)
  __shared__ unsigned wcnt[2][4];           // queue lengths published at the stage barrier, double-buffered by parity
  int wpar = 0;
  unsigned wq_cnt = 0;            // wave-uniform: columns in this wave's queue
  const __bf16* Xb = reinterpret_cast<const __bf16*>(a.Xb);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int r31 = lane & 31, hh = lane >> 5;
  // XCD-aware block map (workgroups are dealt round-robin over the 8 XCDs, each with its own L2): all query blocks
  // that stream the SAME corpus split are placed on one XCD and next to each other in dispatch order, so a corpus
  // tile is fetched from HBM once per XCD instead of once per query block.  Pure speed: any placement is correct.
  const unsigned lin = blockIdx.x;
  const unsigned xcd = lin & 7u, kk = lin >> 3;
  const unsigned bx = kk % (unsigned)a.qgrid, by = (kk / (unsigned)a.qgrid) * 8u + xcd;
  if ((int)by >= a.nsplit) return;
  const int64_t qb0 = (int64_t)bx * QBB;
  const int ql0 = w * 64 + r31, ql1 = ql0 + 32;  // block-local query index of this lane's two query groups
  const bool ok0 = qb0 + ql0 < a.nq, ok1 = qb0 + ql1 < a.nq;
  const int64_t qr0 = ok0 ? qb0 + ql0 : a.nq - 1, qr1 = ok1 ? qb0 + ql1 : a.nq - 1;
  const float th0 = (a.thr && ok0) ? a.thr[qr0] : -INFINITY, th1 = (a.thr && ok1) ? a.thr[qr1] : -INFINITY;
  bf16x8_t qf0[KB], qf1[KB];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    const f32x4 u0 = *reinterpret_cast<const f32x4*>(&a.Q[qr0 * D + kb * 16 + 8 * hh]);
    const f32x4 u1 = *reinterpret_cast<const f32x4*>(&a.Q[qr0 * D + kb * 16 + 8 * hh + 4]);
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(&a.Q[qr1 * D + kb * 16 + 8 * hh]);
    const f32x4 v1 = *reinterpret_cast<const f32x4*>(&a.Q[qr1 * D + kb * 16 + 8 * hh + 4]);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      qf0[kb][j] = (__bf16)u0[j]; qf0[kb][4 + j] = (__bf16)u1[j];
      qf1[kb][j] = (__bf16)v0[j]; qf1[kb][4 + j] = (__bf16)v1[j];
    }
  }
  const int64_t n_seq = (a.n_virtual + TRB - 1) / TRB;
  const int64_t per = (n_seq + a.nsplit - 1) / a.nsplit;
  const int64_t i0 = (int64_t)by * per;
  const int64_t i1 = (i0 + per < n_seq) ? i0 + per : n_seq;
  if (i0 >= i1) {   // a split without rows (nsplit does not divide the stages): its segments are empty, and SAY so
    if (MODE == 0 && hh == 0) {
      if (ok0) a.seg_cnt[(size_t)qr0 * a.nsplit + by] = 0;
      if (ok1) a.seg_cnt[(size_t)qr1 * a.nsplit + by] = 0;
    }
    if (MODE == 2) {   // empty streams hold key 0 (below every score): no memset of the stream table needed
      const int64_t slot = ((int64_t)by * 2 + hh) * SAMPLE_T;
#pragma unroll
      for (int i = 0; i < SAMPLE_T; ++i) {
        if (ok0) a.cand[(size_t)qr0 * a.cap + slot + i] = 0ull;
        if (ok1) a.cand[(size_t)qr1 * a.cap + slot + i] = 0ull;
      }
    }
    return;
  }
  if (!DENSE) {   // visible to the flush after the first stage barrier
    if (hh == 0) { qcntS[ql0] = 0; qcntS[ql1] = 0; }
  }

  // staging: every thread owns the same (row-in-16, 16-byte column) slot of each 16-row slab of a stage, so a full
  // tile of the contiguous corpus is NV loads off ONE per-thread pointer with compile-time offsets
  bf16x8_t stA[NV], stB[NV];
  const int sr0 = tid / (D / 8), sc8 = tid % (D / 8);
  constexpr int SROWS = 256 / (D / 8);   // rows covered by one load instruction of the workgroup
  const __bf16* my_src = Xb + (size_t)sr0 * D + sc8 * 8;
  auto load_tile = [&](bf16x8_t* stage, int64_t tile) {
    const int64_t v_base = tile * TRB;
    if (a.row_stride == 1 && v_base + TRB <= a.n_virtual) {   // workgroup-uniform fast path
      const __bf16* src = my_src + (size_t)v_base * D;
#pragma unroll
      for (int i = 0; i < NV; ++i) stage[i] = *reinterpret_cast<const bf16x8_t*>(src + (size_t)i * SROWS * D);
      return;
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int64_t v = v_base + sr0 + i * SROWS;
      bf16x8_t val;
#pragma unroll
      for (int j = 0; j < 8; ++j) val[j] = (__bf16)0.f;
      if (sr0 + i * SROWS < TRB && v < a.n_virtual)
        val = *reinterpret_cast<const bf16x8_t*>(Xb + (size_t)(v * a.row_stride) * D + sc8 * 8);
      stage[i] = val;
    }
  };
  auto store_tile = [&](const bf16x8_t* stage, int buf) {
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (sr0 + i * SROWS < TRB) *reinterpret_cast<bf16x8_t*>(&Xs[buf][(sr0 + i * SROWS) * LDB + sc8 * 8]) = stage[i];
  };
  float top0[SAMPLE_T], top1[SAMPLE_T];
#pragma unroll
  for (int i = 0; i < SAMPLE_T; ++i) { top0[i] = -INFINITY; top1[i] = -INFINITY; }
  auto emit = [&](const f32x16& acc, int64_t v_base, int ql, bool ok, float th, int64_t qrow) {
    if (MODE != 0 && !ok) return;   // MODE 0: every lane takes part in the ballot below (wq_cnt must stay wave-uniform)
    if (MODE == 2) {
      float* top = (ql == ql0) ? top0 : top1;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float sc = (v_base + acc_row(r, lane) < a.n_virtual) ? acc[r] : -INFINITY;
        // insert into the descending list: t_i' = med3(t_{i-1}, s, t_i) (in place, from the tail)
#pragma unroll
        for (int i = SAMPLE_T - 1; i > 0; --i) top[i] = __builtin_amdgcn_fmed3f(top[i - 1], sc, top[i]);
        top[0] = fmaxf(top[0], sc);
      }
      return;
    }
    if (DENSE) {
      uint64_t* cnd = a.cand + (size_t)qrow * a.cap;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t v = v_base + acc_row(r, lane);
        if (v < a.n_virtual) cnd[v] = make_key(acc[r], (uint32_t)v);
      }
      return;
    }
    // The v_max3 ops are inline asm, which the compiler's MFMA->VALU hazard recognizer does not see: they must not be the
    // first readers of the accumulator (they read stale registers when they directly followed the MFMAs and lost
    // survivors).  Every group maximum starts with a compiler-visible fmaxf (hipcc pads it with the required wait
    // states), so every asm read is ordered behind one.
    float g[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) g[k] = max3_raw(fmaxf(acc[4 * k], acc[4 * k + 1]), acc[4 * k + 2], acc[4 * k + 3]);
    const float mx = max3_raw(fmaxf(g[0], g[1]), g[2], g[3]);
    const float te = ok ? th : INFINITY;
#if RIHIP_SCAN_ABLATE == 1   // experiments: the filter without survivor handling (scores computed, maxima taken, nothing kept)
    asm volatile("" :: "v"(mx));
    return;
#endif
    if (__ballot(mx >= te) == 0ull) return;   // wave-uniform: no survivor in the wave's 64 columns (1 in 6 at k = 500)
    // Survivors are found group of four by group of four, all branches wave-uniform: a wave's 1 024 scores hold ~1.7
    // survivors, so ~1.4 of the 4 groups and ~1.1 scores of such a group take the queue path.
    const unsigned vl = (unsigned)v_base + 4u * (unsigned)hh;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (__ballot(g[k] >= te) == 0ull) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float sc = acc[4 * k + j];
        const bool h = sc >= te;
        const unsigned long long m = __ballot(h);
        if (m == 0ull) continue;
        const unsigned pos = wq_cnt + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        wq_cnt += (unsigned)__popcll(m);
        if (h) {
          if (pos < (unsigned)WQE) qent[w][pos] = uint4{__float_as_uint(sc), vl + (unsigned)(j + 8 * k), (unsigned)ql, 0u};
          else atomicOr(&qcntS[ql], 1 << 30);   // queue full (a stage brought > 64 survivors to one wave): the segment is
                                                // marked overflowed and the query takes the exact re-do path
        }
      }
    }
  };
  // The wave empties its OWN queue (no barrier, no other wave involved): one entry per lane, slot in the query's segment
  // from a returning LDS atomic, all loads / atomics / stores of a pass in flight together.
  auto wave_flush = [&]() {
    const unsigned n = wq_cnt < (unsigned)WQE ? wq_cnt : (unsigned)WQE;   // entries beyond WQE went the slow path
    uint4 en[WQE / 64]; int pos[WQE / 64];
#pragma unroll
    for (int j = 0; j < WQE / 64; ++j) en[j] = qent[w][j * 64 + lane];
#pragma unroll
    for (int j = 0; j < WQE / 64; ++j) {
      const bool in = (unsigned)(j * 64 + lane) < n && (int64_t)en[j].y < a.n_virtual;
      pos[j] = in ? atomicAdd(&qcntS[en[j].z], 1) : a.seg_cap;   // LDS
    }
#pragma unroll
    for (int j = 0; j < WQE / 64; ++j)
      if (pos[j] < a.seg_cap)
        a.seg[((size_t)(qb0 + en[j].z) * a.nsplit + by) * a.seg_cap + pos[j]] = make_key(__uint_as_float(en[j].x), en[j].y);
    wq_cnt = 0;
  };
  // after every stage: the barrier hands the LDS tile buffer over.  ALL waves empty their queues at the same stage, as
  // soon as one of them is half full: a wave that flushes alone makes its three siblings wait at the next barrier, and
  // with four independent triggers the workgroup stalled four times as often (0.64 of 1.8 ms).
  auto stage_end = [&](bool last) {
    if (!DENSE && lane == 0) wcnt[wpar][w] = wq_cnt;
    __syncthreads();
    if (!DENSE) {
      // the slot read here is rewritten two barriers later at the earliest: every wave sees the same four values
      const unsigned m01 = wcnt[wpar][0] > wcnt[wpar][1] ? wcnt[wpar][0] : wcnt[wpar][1];
      const unsigned m23 = wcnt[wpar][2] > wcnt[wpar][3] ? wcnt[wpar][2] : wcnt[wpar][3];
      wpar ^= 1;
      if (last || (m01 > m23 ? m01 : m23) >= (unsigned)(WQE - 64)) wave_flush();   // workgroup-uniform decision (a stage adds ~10-20 entries per wave)
    }
  };
  auto compute = [&](int buf, int64_t i) {
#pragma unroll 1
    for (int sub = 0; sub < TRB / 32; ++sub) {
      const __bf16* Xt = &Xs[buf][sub * 32 * LDB];
      f32x16 a0 = zero16(), a1 = zero16();
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        const bf16x8_t av = *reinterpret_cast<const bf16x8_t*>(&Xt[r31 * LDB + kb * 16 + 8 * hh]);
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, qf0[kb], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, qf1[kb], a1, 0, 0, 0);
      }
      const int64_t v_base = i * TRB + sub * 32;
      emit(a0, v_base, ql0, ok0, th0, qr0);
      emit(a1, v_base, ql1, ok1, th1, qr1);
    }
  };

  // two stages in flight through registers (global latency under load is 2-4 stages of MFMA work), two LDS buffers
  load_tile(stA, i0);
  store_tile(stA, 0);
  if (i0 + 1 < i1) load_tile(stA, i0 + 1);
  __syncthreads();
#pragma unroll 1
  for (int64_t i = i0; i < i1; i += 2) {
    // even phase: buffer 0 = stage i, stA = stage i+1 (in flight)
    if (i + 2 < i1) load_tile(stB, i + 2);
    compute(0, i);
    if (i + 1 < i1) store_tile(stA, 1);
    stage_end(i + 1 >= i1);
    if (i + 1 >= i1) break;
    // odd phase: buffer 1 = stage i+1, stB = stage i+2 (in flight)
    if (i + 3 < i1) load_tile(stA, i + 3);
    compute(1, i + 1);
    if (i + 2 < i1) store_tile(stB, 0);
    stage_end(i + 2 >= i1);
  }
  if (MODE == 0 && hh == 0) {   // the wave's own LDS atomics are complete (in order): publish the segment fills
    if (ok0) a.seg_cnt[(size_t)qr0 * a.nsplit + by] = qcntS[ql0];
    if (ok1) a.seg_cnt[(size_t)qr1 * a.nsplit + by] = qcntS[ql1];
  }
  if (MODE == 2) {   // stream = (corpus split, row half): SAMPLE_T key slots each, [nq, 2 * nsplit * SAMPLE_T]
    const int64_t slot = ((int64_t)by * 2 + hh) * SAMPLE_T;
#pragma unroll
    for (int i = 0; i < SAMPLE_T; ++i) {
      if (ok0) a.cand[(size_t)qr0 * a.cap + slot + i] = make_key(top0[i], 0u);
      if (ok1) a.cand[(size_t)qr1 * a.cap + slot + i] = make_key(top1[i], 0u);
    }
  }
}

// RIHIP_FILTER_WIDE=1 sends batches of more than 512 queries at d = 128 to scan_bf16_wide_kernel (opt-in: see its header)
static bool filter_wide_enabled() {
  const char* ev = getenv("RIHIP_FILTER_WIDE");
  return ev && ev[0] == '1';
}

// ---- an OPT-IN filter for large batches at d = 128 (round 3 experiment, RIHIP_FILTER_WIDE=1) --------------------------
// What scan_bf16_kernel leaves on the table at 4 096 queries x 1 M rows, measured by switching parts off
// (profiles/README.md, "filter ablations"): without any survivor handling it runs 0.78-0.84 ms -- 1.3-1.4 PFLOP/s, at
// which the chip already holds its clock at ~2.0 GHz (bf16 MFMA on random data is power-limited well below the 2.5 PF
// of the data sheet) -- and the survivor handling ADDS 0.45-0.55 ms on top.
// This kernel tries the other end of the design space.  A workgroup is 8 waves = 2 per SIMD with 256 registers each,
// 128 queries per wave (four 32-query MFMA groups: an LDS fragment feeds four MFMAs; 1 024 queries per workgroup also
// quarter the L2 -> LDS traffic).  LDS tiles arrive by LDS-DMA (global_load_lds: no staging registers; the 32 registers
// saved hold k-chunks 0..3 of the NEXT sub-tile, re-read as soon as their MFMAs have issued, and k-chunks 4..7 of the
// current one), three stage buffers deep, so a DMA has two stages to land.  The LDS image is lane-linear per DMA piece
// (1 KiB = 4 rows of 256 B) and bank-conflict-free for the fragment reads through an XOR swizzle of the 16-byte chunk
// index, applied to the DMA's SOURCE address and to the read address alike.  The waves 0-3 and the waves 4-7 run the
// SAME program -- MFMAs of sub-tile s, barrier, survivors of sub-tile s, barrier -- one phase apart (waves 4-7 take one
// extra barrier first, waves 0-3 one extra at the end): in every phase each SIMD has one wave on the matrix pipe and
// one in the survivor code.
// Measured (tools/filter_ablate.sh, tools/wide_probe.py): MFMA-only 0.76-0.79 ms -- no better than the 3-wave kernel, both
// sit at the clock the chip holds; complete 1.28-1.33 ms against 1.28-1.33 ms for scan_bf16_kernel.  The survivor phase
// of one wave costs ~2 100 cycles (264 instructions, a wave alone issues ~1 per 8-10 cycles; the column dump is 20
// ds_write_b128 at 13 LDS cycles each) against ~1 300 for the partner's 32 MFMAs, so the phases do not balance.  With one
// barrier per stage instead of four (RIHIP_WIDE_PINGPONG=0) the two drifting waves hide each other's latencies and the
// kernel runs 1.17-1.23 ms -- but that form loses ~1 survivor in 10 of the queries (tools/filter_check.py; sporadic,
// any rank, any stage, not cured by full waits or nops: unexplained), so it is NOT offered.  Hence opt-in, not default.
#ifdef RIHIP_WIDE_PROBE   // diagnostic build: s_memtime stamps of one stage of workgroup 0, waves 0 and 4
__device__ unsigned long long g_wide_probe[64];
#define WIDE_STAMP(k) do { if (blockIdx.x == 0 && i == i0 + 20 && (threadIdx.x & 255) == 0) g_wide_probe[(threadIdx.x >> 8) * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int rihip_debug_wide_probe(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wide_probe), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : 1;
}
#define WIDE_STAMP2(k) do { if (probe_on && blockIdx.x == 0 && (threadIdx.x & 255) == 0) g_wide_probe[32 + (threadIdx.x >> 8) * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WIDE_STAMP(k)
#define WIDE_STAMP2(k)
#endif
// 1 (the only form offered): a barrier after EVERY phase holds the two wave halves one phase apart; 0: one barrier per
// stage, the waves drift -- faster, but it loses survivors (header above): diagnostic builds only.
#ifndef RIHIP_WIDE_PINGPONG
#define RIHIP_WIDE_PINGPONG 1
#endif
constexpr int QW = 128;          // queries per wave
constexpr int QB2 = 8 * QW;      // queries per workgroup
constexpr int ST2 = 64;          // corpus rows per stage (two 32-row MFMA sub-tiles)
constexpr int NB2 = 3;           // LDS stage buffers
constexpr int WQC = 64;          // survivor queue: 80-byte columns (16 scores + header) per wave
constexpr int WIDE_LDS = 8 * WQC * 80 + QB2 * 4;   // 24 576 bytes of dynamic LDS beside the 48 KB of stage buffers

__global__ __launch_bounds__(512, 2) void scan_bf16_wide_kernel(ScanArgs a) {
  constexpr int D = 128, KB = D / 16;
  // (the stage buffers are a static array and the queues dynamic LDS ON PURPOSE: hipcc puts `s_waitcnt vmcnt(0)` in front
  // of every LDS access that may alias a pending LDS-DMA, and it can only tell distinct objects apart)
  __shared__ __attribute__((aligned(1024))) __bf16 Xs[NB2 * ST2 * D];
  extern __shared__ __attribute__((aligned(16))) char wide_lds[];
  char* qcol = wide_lds;                                           // [8][WQC] columns of 80 bytes
  int* qcntS = reinterpret_cast<int*>(wide_lds + 8 * WQC * 80);    // [QB2] survivors per query in this split
  const __bf16* Xb = reinterpret_cast<const __bf16*>(a.Xb);
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);   // (w in a scalar register)
  const int r31 = lane & 31, hh = lane >> 5;
  // XCD-aware block map, as in scan_bf16_kernel: the query blocks that stream the same corpus split share an XCD's L2
  const unsigned lin = blockIdx.x;
  const unsigned xcd = lin & 7u, kk = lin >> 3;
  const unsigned bx = kk % (unsigned)a.qgrid, by = (kk / (unsigned)a.qgrid) * 8u + xcd;
  if ((int)by >= a.nsplit) return;
  const int64_t qb0 = (int64_t)bx * QB2;
  float te[4];
  bf16x8_t qf[4][KB];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int64_t q = qb0 + w * QW + g * 32 + r31;
    const int64_t qr = q < a.nq ? q : a.nq - 1;
    te[g] = q < a.nq ? a.thr[qr] : INFINITY;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      const f32x4 u0 = *reinterpret_cast<const f32x4*>(&a.Q[qr * D + kb * 16 + 8 * hh]);
      const f32x4 u1 = *reinterpret_cast<const f32x4*>(&a.Q[qr * D + kb * 16 + 8 * hh + 4]);
#pragma unroll
      for (int j = 0; j < 4; ++j) { qf[g][kb][j] = (__bf16)u0[j]; qf[g][kb][4 + j] = (__bf16)u1[j]; }
    }
  }
  const unsigned qlb = (unsigned)(w * QW + r31);   // block-local index of the lane's first query (group g: + 32 g)
  const int64_t n_seq = (a.n_virtual + ST2 - 1) / ST2;
  const int64_t per = (n_seq + a.nsplit - 1) / a.nsplit;
  const int64_t i0 = (int64_t)by * per;
  const int64_t i1 = (i0 + per < n_seq) ? i0 + per : n_seq;
  if (i0 >= i1) {   // a split without rows: its segments are empty, and SAY so
    if (hh == 0) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int64_t q = qb0 + w * QW + g * 32 + r31;
        if (q < a.nq) a.seg_cnt[(size_t)q * a.nsplit + by] = 0;
      }
    }
    return;
  }
  if (hh == 0) {
#pragma unroll
    for (int g = 0; g < 4; ++g) qcntS[qlb + g * 32] = 0;   // own queries only: visible to the wave's own flush in program order
  }

  // LDS-DMA: wave w issues pieces 2w, 2w+1 of a stage; piece j holds rows 4j .. 4j+3, lane L writes chunk L of the piece
  // = (row 4j + L/16, stored chunk L%16) and therefore FETCHES chunk (L%16) ^ (row & 15) of that row.  With j = 2w + t
  // the byte offset of lane L's source in the stage is ((base + 1024 t) ^ (64 t)): one register, re-derived per piece
  // (the opaque asm keeps hipcc from hoisting 64-bit addresses into registers this kernel does not have).
  const unsigned dma_base = (unsigned)((w * 8 + (lane >> 4)) * (D * 2)) + (unsigned)((((lane & 15) ^ (lane >> 4) ^ (8 * (w & 1))) & 15) << 4);
  auto issue_dma = [&](int64_t stage, int buf) {
    const int64_t v_base = stage * ST2;
    unsigned ob = dma_base;
    asm volatile("" : "+v"(ob));
    // (the bf16 copy is padded with zero rows to a whole stage -- prepare_flat -- and the flush drops rows >= n_virtual)
    const char* src = reinterpret_cast<const char*>(Xb + (size_t)v_base * D);
#pragma unroll
    for (int t = 0; t < 2; ++t)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + ((ob + 1024u * t) ^ (64u * t))),
                                       (__attribute__((address_space(3))) void*)(&Xs[(size_t)buf * (ST2 * D) + ((w * 2 + t) * 4) * D]), 16, 0, 0);
  };
  // fragment (row sub*32 + r31, k chunk 2 kb + hh) sits at chunk (2 kb + hh) ^ (r31 & 15) of its row:
  // byte offset = row * 256 + 16 * (hh ^ (r31 & 1)) + 32 * (kb ^ ((r31 & 15) >> 1)).
  // The reads are inline asm ON PURPOSE: hipcc (ROCm 7.2) puts `s_waitcnt vmcnt(0)` in front of every ds_read it can see
  // while an LDS-DMA that may alias it is pending (it cannot tell the stage buffers apart), which would drain the DMA
  // issued a few instructions earlier.  The compiler does not track these reads, so their `lgkmcnt` waits are explicit
  // too (wait_fa / wait_fb: asm that "rewrites" the fragment registers, so no MFMA can be scheduled above it).
  // The row part, the 16-byte half and the swizzle term leave bits 5..7 of the sum clear for (kb ^ y) << 5, so ONE
  // register p = base + (y << 5) (+ the stage buffer's offset) gives every fragment address as p ^ (kb << 5): this
  // kernel has no registers to spare, and a spilled value costs a `vmcnt(0)` -- i.e. a drained DMA -- at its reload.
  const unsigned xs_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)reinterpret_cast<char*>(&Xs[0]);
  const unsigned rd_p0 = xs_lds + (unsigned)(r31 * (D * 2)) + (unsigned)((hh ^ (r31 & 1)) << 4) + (unsigned)(((r31 & 15) >> 1) << 5);
  unsigned rd_p = rd_p0;   // + buffer offset of the stage being read
  auto read_frag = [&](bf16x8_t& dst, int sub, int kb) {
    unsigned pk = rd_p;
    asm volatile("" : "+v"(pk));         // (re-derive the 8 addresses instead of keeping 8 registers)
    const unsigned addr = pk ^ (unsigned)(kb << 5);
    if (sub == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr));
    else asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(dst) : "v"(addr));
  };
  // Survivor handling.  What it costs is INSTRUCTIONS: a wave alone issues one every >= 4 cycles (more beside the partner's
  // MFMAs) and a taken branch costs ~60, so the per-score branch tree this kernel first had took ~950 cycles per 32x32
  // tile and a one-tile-at-a-time column dump + a separate flush ~3 000 per phase against ~1 300 for the partner's 32
  // MFMAs (s_memtime stamps, tools/wide_probe.py).  This form:
  //  * the four tiles of a sub-tile are handled together (independent maximum chains and slot computations interleave);
  //  * a lane whose 16-score column holds a survivor dumps the column + a header (query, first row, threshold) into the
  //    wave's ring of WQC columns -- straight-line code, slot from the ballots; the per-score test happens in the flush,
  //    64 scores at a time with every lane busy;
  //  * every phase flushes up to 16 columns queued by EARLIER phases, its LDS reads / atomics / global stores issued
  //    ahead of, between and behind the steps of the dump so that none of their latencies is waited for idle;
  //  * all LDS operations of the hot loop are inline asm: hipcc (ROCm 7.2) puts `s_waitcnt vmcnt(0)` in front of every
  //    LDS access it can see that may alias a pending LDS-DMA, i.e. drains the DMA issued a few hundred cycles earlier.
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  char* qcol_w = qcol + w * (WQC * 80);
  const unsigned qcol_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)qcol_w;
  const unsigned qcnt_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)reinterpret_cast<char*>(qcntS);
  const int seg_sh = 31 - __builtin_clz((unsigned)a.seg_cap);   // (a power of two: see the host side)
  unsigned q_head = 0, q_tail = 0;   // wave-uniform ring indices (columns q_head .. q_tail-1 are queued, slot = index % WQC)
  bool q_over = false;               // the ring overflowed: all of the wave's queries take the exact re-do
  // (within every step the four tiles / four batch passes are computed side by side and only then come the predicated
  // LDS / global operations: a wave alone pays ~10 cycles per DEPENDENT instruction and ~4 per independent one)
  const unsigned nv32 = (unsigned)(a.n_virtual < 0xFFFFFFFFll ? a.n_virtual : 0xFFFFFFFFll);
  bool probe_on = false;
  auto survivor_phase = [&](const f32x16& a0, const f32x16& a1, const f32x16& a2, const f32x16& a3, unsigned vl, bool dump, bool flush) {
    WIDE_STAMP2(0);
    // ---- A: reads of the flush batch
    unsigned ln = (unsigned)lane;
    asm volatile("" : "+v"(ln));   // (keeps hipcc from holding lane-derived constants in registers across the MFMA phases)
    const unsigned e = ln & 15u, cj = ln >> 4;
    const unsigned ve = (e & 3u) + 8u * (e >> 2);
    const unsigned nb = q_tail - q_head < 16u ? q_tail - q_head : 16u;
    u32x4_t hd[4]; float sc[4]; int pos[4];
    if (flush) {
      unsigned cadr[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) cadr[j] = qcol_lds + ((q_head + 4u * j + cj) & (unsigned)(WQC - 1)) * 80u;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        asm volatile("ds_read_b128 %0, %2 offset:64\n\tds_read_b32 %1, %3" : "=v"(hd[j]), "=v"(sc[j]) : "v"(cadr[j]), "v"(cadr[j] + 4u * e));
    }
    WIDE_STAMP2(1);
    __builtin_amdgcn_sched_barrier(0);
    // ---- B: column maxima of the four new tiles (inline-asm v_max3 must not be the first reader of an MFMA result --
    // the hazard recognizer does not see it: each chain is seeded through a compiler-visible move)
    unsigned long long b0 = 0, b1 = 0, b2 = 0, b3 = 0;
    bool h0 = false, h1 = false, h2 = false, h3 = false;
    if (dump) {
      float m0 = a0[15], m1 = a1[15], m2 = a2[15], m3 = a3[15];
      asm volatile("" : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3));
#pragma unroll
      for (int r = 0; r < 14; r += 2) {
        m0 = max3_raw(m0, a0[r], a0[r + 1]);
        m1 = max3_raw(m1, a1[r], a1[r + 1]);
        m2 = max3_raw(m2, a2[r], a2[r + 1]);
        m3 = max3_raw(m3, a3[r], a3[r + 1]);
      }
      m0 = max3_raw(m0, a0[14], a0[14]); m1 = max3_raw(m1, a1[14], a1[14]);
      m2 = max3_raw(m2, a2[14], a2[14]); m3 = max3_raw(m3, a3[14], a3[14]);
#if RIHIP_SCAN_ABLATE != 1   // (1: experiments without survivor handling)
      h0 = m0 >= te[0]; h1 = m1 >= te[1]; h2 = m2 >= te[2]; h3 = m3 >= te[3];
#else
      asm volatile("" :: "v"(m0), "v"(m1), "v"(m2), "v"(m3));
#endif
      b0 = __ballot(h0); b1 = __ballot(h1); b2 = __ballot(h2); b3 = __ballot(h3);
    }
    WIDE_STAMP2(2);
    __builtin_amdgcn_sched_barrier(0);
    // ---- C: per-score test of the flush batch, slots in the queries' segments (returning LDS atomics)
    if (flush) {
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(hd[0]), "+v"(hd[1]), "+v"(hd[2]), "+v"(hd[3]), "+v"(sc[0]), "+v"(sc[1]), "+v"(sc[2]), "+v"(sc[3]));
      bool hit[4]; unsigned cadr[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        hit[j] = 4u * j + cj < nb && sc[j] >= __uint_as_float(hd[j].z) && hd[j].y + ve < nv32;
        cadr[j] = qcnt_lds + 4u * hd[j].w;
        pos[j] = a.seg_cap;
      }
      const unsigned one = 1u;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (hit[j]) asm volatile("ds_add_rtn_u32 %0, %1, %2" : "=v"(pos[j]) : "v"(cadr[j]), "v"(one) : "memory");
      q_head += nb;
    }
    WIDE_STAMP2(3);
    __builtin_amdgcn_sched_barrier(0);
    // ---- D: column dump of the new tiles
    if (dump) {
      const unsigned n0 = (unsigned)__popcll(b0), n1 = (unsigned)__popcll(b1), n2 = (unsigned)__popcll(b2), n3 = (unsigned)__popcll(b3);
      const unsigned base1 = q_tail + n0, base2 = base1 + n1, base3 = base2 + n2, tail2 = base3 + n3;
      q_over = q_over || tail2 - q_head > (unsigned)WQC;   // (q_head already advanced: the batch's reads are older LDS operations)
      unsigned qx = qlb;
      asm volatile("" : "+v"(qx));   // (derived here: hipcc would otherwise hold four query indices in registers)
      auto slot = [&](unsigned long long bal, unsigned base) -> unsigned {
        const unsigned p2 = base + __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
        return qcol_lds + (p2 & (unsigned)(WQC - 1)) * 80u;
      };
      const unsigned d0 = slot(b0, q_tail), d1 = slot(b1, base1), d2 = slot(b2, base2), d3 = slot(b3, base3);
      auto put = [&](const f32x16& acc, bool h, unsigned dst, int g) {
        if (h) {
          const f32x4 s0 = {acc[0], acc[1], acc[2], acc[3]}, s1 = {acc[4], acc[5], acc[6], acc[7]};
          const f32x4 s2 = {acc[8], acc[9], acc[10], acc[11]}, s3 = {acc[12], acc[13], acc[14], acc[15]};
          const u32x4_t hdr = {vl, vl, __float_as_uint(te[g]), qx + (unsigned)(g * 32)};
          asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %2 offset:16\n\tds_write_b128 %0, %3 offset:32\n\t"
                       "ds_write_b128 %0, %4 offset:48\n\tds_write_b128 %0, %5 offset:64"
                       :: "v"(dst), "v"(s0), "v"(s1), "v"(s2), "v"(s3), "v"(hdr) : "memory");
        }
      };
      put(a0, h0, d0, 0);
      put(a1, h1, d1, 1);
      put(a2, h2, d2, 2);
      put(a3, h3, d3, 3);
      q_tail = tail2;
    }
    WIDE_STAMP2(4);
    __builtin_amdgcn_sched_barrier(0);
    // ---- E: the flush batch's survivors to their segments
    if (flush) {
      uint64_t* dstp[4]; uint64_t key[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned seg_i = __umul24((unsigned)qb0 + hd[j].w, (unsigned)a.nsplit) + by;
        dstp[j] = a.seg + ((size_t)seg_i << seg_sh);
        key[j] = make_key(sc[j], hd[j].y + ve);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pos[0]), "+v"(pos[1]), "+v"(pos[2]), "+v"(pos[3]));
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (pos[j] < a.seg_cap) dstp[j][(unsigned)pos[j]] = key[j];
    }
    WIDE_STAMP2(5);
  };
  auto phase_barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };

  // prologue: stages i0 and i0+1 in flight, k-chunks 0..3 of (i0, sub-tile 0) read
  issue_dma(i0, 0);
  if (i0 + 1 < i1) {
    issue_dma(i0 + 1, 1);
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  phase_barrier();
  // Fragment registers: fa = k chunks 0..3 of the NEXT sub-tile (read while the current one's MFMAs issue, in flight
  // during its survivor handling), fb = k chunks 4..7 of the CURRENT one (read at the start of its MFMA phase, 512 MFMA
  // cycles before their use; dead during the survivor handling, which is where the register pressure peaks).
  bf16x8_t fa[4], fb[4];
  auto wait_fa = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3])); };
  auto wait_fb4 = [&]() { asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(fb[0]), "+v"(fb[1]), "+v"(fb[2]), "+v"(fb[3])); };
  auto wait_fb0 = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fb[0]), "+v"(fb[1]), "+v"(fb[2]), "+v"(fb[3])); };
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) read_frag(fa[kb], 0, kb);
#if RIHIP_WIDE_PINGPONG
  if (w >= 4) phase_barrier();   // waves 4-7 run one phase behind waves 0-3
#endif
  int buf = 0;
#pragma unroll 1
  for (int64_t i = i0; i < i1; ++i) {
    // (the survivor phases' global stores count in vmcnt like the DMA pieces; wherever they fall in the issue order, the
    // counted wait below still covers everything older than the two youngest operations, i.e. this wave's pieces of
    // stage i+1)
    // buffer (i+2) % 3 was last read in the phase that ended with the barrier this wave has just passed
    const int buf2 = buf >= 1 ? buf - 1 : 2;
    if (i + 2 < i1) issue_dma(i + 2, buf2);
    const unsigned vb = (unsigned)(i * ST2) + 4u * (unsigned)hh;
    f32x16 c0, c1, c2, c3;
    WIDE_STAMP(0);
    wait_fa();
    {   // MFMAs of sub-tile 0
      c0 = zero16(); c1 = zero16(); c2 = zero16(); c3 = zero16();
#if RIHIP_SCAN_ABLATE == 11   // experiments: waves 4-7 issue no MFMAs (the survivor phases of waves 0-3 run uncontended)
      if (w < 4)
#endif
      {
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) read_frag(fb[kb], 0, 4 + kb);
      __builtin_amdgcn_sched_barrier(0);   // (hipcc's scheduler otherwise sinks these reads to just before their wait)
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kb], qf[0][kb], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kb], qf[1][kb], c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kb], qf[2][kb], c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kb], qf[3][kb], c3, 0, 0, 0);
        read_frag(fa[kb], 1, kb);
        __builtin_amdgcn_sched_barrier(0);
      }
      wait_fb4();
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[kb], qf[0][4 + kb], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[kb], qf[1][4 + kb], c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[kb], qf[2][4 + kb], c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[kb], qf[3][4 + kb], c3, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      }
    }
    WIDE_STAMP(1);
#if RIHIP_WIDE_PINGPONG
    phase_barrier();
#endif
    WIDE_STAMP(2);
    probe_on = (i == i0 + 20);
    survivor_phase(c0, c1, c2, c3, vb, true, q_tail - q_head >= 12u);   // (a sub-tile brings ~7 columns at k = 500: a batch of <= 16 every other phase)
    probe_on = false;
    WIDE_STAMP(3);
    while (__builtin_expect(q_tail - q_head > 32u, 0)) survivor_phase(c0, c1, c2, c3, vb, false, true);   // (dense survivors)
    WIDE_STAMP(4);
#if RIHIP_WIDE_PINGPONG
    phase_barrier();
#endif
    WIDE_STAMP(5);
    wait_fa();
    {   // MFMAs of sub-tile 1
      c0 = zero16(); c1 = zero16(); c2 = zero16(); c3 = zero16();
#if RIHIP_SCAN_ABLATE == 11
      if (w < 4)
#endif
      {
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) read_frag(fb[kb], 1, 4 + kb);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kb], qf[0][kb], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kb], qf[1][kb], c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kb], qf[2][kb], c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kb], qf[3][kb], c3, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      wait_fb0();
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[kb], qf[0][4 + kb], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[kb], qf[1][4 + kb], c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[kb], qf[2][4 + kb], c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[kb], qf[3][4 + kb], c3, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      }
    }
    // this wave's pieces of stage i+1 have landed (the barriers add the other waves': the first reader of stage i+1 is
    // two barriers away for waves 0-3's pieces and one for waves 4-7's); the DMA of stage i+2 stays in flight
    if (i + 2 < i1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    WIDE_STAMP(6);
    phase_barrier();
    WIDE_STAMP(7);
    survivor_phase(c0, c1, c2, c3, vb + 32u, true, q_tail - q_head >= 12u);
    WIDE_STAMP(8);
    while (__builtin_expect(q_tail - q_head > 32u, 0)) survivor_phase(c0, c1, c2, c3, vb + 32u, false, true);
    WIDE_STAMP(9);
#if RIHIP_WIDE_PINGPONG
    phase_barrier();
#endif
    WIDE_STAMP(10);
    buf = buf == 2 ? 0 : buf + 1;
    rd_p = rd_p0 + (unsigned)buf * (unsigned)(ST2 * D * 2);
    if (i + 1 < i1) {
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) read_frag(fa[kb], 0, kb);
    }
  }
#if RIHIP_WIDE_PINGPONG
  if (w < 4) phase_barrier();    // (every wave passes the same number of barriers)
#endif
  {
    f32x16 z = zero16();
    while (q_tail != q_head) survivor_phase(z, z, z, z, 0u, false, true);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (q_over && hh == 0) {   // ring overflow (a sub-tile brought more columns than the ring had free: dense survivors)
#pragma unroll
    for (int g = 0; g < 4; ++g) atomicOr(&qcntS[qlb + g * 32], 1 << 30);
  }
  {   // the wave's own LDS atomics are complete (in order): publish the segment fills.  (Lane -> query mapping derived
      // again from the thread index: keeping the prologue's values alive through the loop costs registers.)
    unsigned t2 = threadIdx.x;
    asm volatile("" : "+v"(t2));
    if (((t2 >> 5) & 1u) == 0u) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int qlg = (int)(t2 >> 6) * QW + g * 32 + (int)(t2 & 31u);
        if (qb0 + qlg < a.nq) a.seg_cnt[(size_t)(qb0 + qlg) * a.nsplit + by] = qcntS[qlg];
      }
    }
  }
}

// segments -> the contiguous candidate list of query blockIdx.x (split order, then slot order: deterministic); a
// segment that overflowed marks the query as overflowed (count > cap => exact re-do)
__global__ __launch_bounds__(256) void compact_segments_kernel(const uint64_t* __restrict__ seg, const int* __restrict__ seg_cnt,
                                                               int nsplit, int seg_cap, uint64_t* cand, int64_t cap,
                                                               int* count, int cs) {
  __shared__ int off[1025];
  __shared__ int over;
  const int64_t q = blockIdx.x;
  const int tid = threadIdx.x;
  if (tid == 0) {
    int run = 0, ov = 0;
    for (int s = 0; s < nsplit; ++s) {
      int c = seg_cnt[q * nsplit + s];
      if (c > seg_cap) { ov = 1; c = seg_cap; }
      off[s] = run; run += c;
    }
    off[nsplit] = run;
    over = ov || run > cap;
    count[q * cs] = over ? (int)(cap + 1) : run;
  }
  __syncthreads();
  if (over) return;
  for (int s = tid >> 6; s < nsplit; s += 4) {          // one wave per segment
    const int n = off[s + 1] - off[s];
    const uint64_t* src = seg + ((size_t)q * nsplit + s) * seg_cap;
    uint64_t* dst = cand + (size_t)q * cap + off[s];
    for (int i = tid & 63; i < n; i += 64) dst[i] = src[i];
  }
}

// exact f32 re-score of the survivors: 16 lanes per candidate, fixed summation order
template <int D>
__global__ __launch_bounds__(256) void rerank_kernel(const float* __restrict__ X, const float* __restrict__ Q,
                                                     uint64_t* cand, int64_t cap, const int* __restrict__ count,
                                                     float* qnorm, int64_t N, const float* __restrict__ kth_approx,
                                                     float eps_scale, int cs) {
  constexpr int PER = D / 16;  // floats per lane
  const int64_t q = blockIdx.x;
  const int tid = threadIdx.x, l16 = tid & 15, grp = tid >> 4;
  float qv[PER];
  float cut;
#pragma unroll
  for (int j = 0; j < PER; ++j) qv[j] = Q[q * D + l16 * PER + j];
  {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < PER; ++j) s += qv[j] * qv[j];
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (tid == 0) qnorm[q] = sqrtf(s);
    // At least k candidates have an approximate score >= kth_approx[q], hence exact scores >= kth_approx - eps, so the
    // exact k-th score T* >= kth_approx - eps; a candidate whose approximate score is below kth_approx - 2 eps has an
    // exact score < kth_approx - eps <= T*: it cannot be in the top-k and its row is not fetched (key zeroed).
    cut = kth_approx ? kth_approx[q] - 2.f * eps_scale * sqrtf(s) - 1e-6f : -INFINITY;
  }
  const int cnt = count[q * cs];
  const int64_t n = cnt <= cap ? cnt : 0;  // overflowed list: the query is re-done exactly anyway
  uint64_t* keys = cand + (size_t)q * cap;
  for (int64_t i = grp; i < n; i += 16) {
    const uint64_t key = keys[i];
    const uint32_t row = 0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFull);
    if ((int64_t)row >= N) continue;
    if (ord2f((uint32_t)(key >> 32)) < cut) {
      if (l16 == 0) keys[i] = 0ull;
      continue;
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < PER; ++j) s = fmaf(qv[j], X[(size_t)row * D + l16 * PER + j], s);
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (l16 == 0) keys[i] = make_key(s, row);
  }
}

// ---- finalize: radix-select the k_sel best keys of query q, sort them, emit -------------------
struct FinArgs {
  const uint64_t* cand;  // [nq, cap]
  int64_t cap;
  const int* count;      // [nq * count_stride]
  int count_stride;      // 0/1 = dense
  const int* qmap;       // optional: output slot -> query index inside cand/count (fallback), or null
  int64_t nq;
  int k;                 // requested k (<= K_MAX)
  // mode 0: write top-k scores/rows ; mode 1: write thr[q] = score of the k_sel-th key
  int mode;
  int rank;              // mode 1: r
  float* out_scores;     // [nq_out, k]
  int64_t* out_rows;     // [nq_out, k]
  uint64_t* out_keys;    // mode 0, optional: write the k best KEYS (0-padded) instead of scores/rows (hierarchical select)
  const int* out_slot;   // optional: where query i's results go (fallback), or null
  float* thr_out;        // mode 1
  int* fail_flags;       // [nq] mode 0: 1 if count<need_min or count>cap
  int64_t need_min;      // min(k, N_effective): candidates required for exactness (0 => no check)
  // two-precision search: candidates were filtered by APPROXIMATE scores >= thr_chk[q]; the exact top-k is
  // proven complete iff its k-th exact score >= thr_chk[q] + eps_scale*qnorm[q] + 2e-6
  const float* thr_chk;
  const float* qnorm;
  float eps_scale;
  // IVF with a sampled threshold: fewer than k candidates is only acceptable when nothing was filtered (thr = -inf)
  const float* ivf_thr;
  const int64_t* id_map;  // optional: out_rows[i] = id_map[row] (the wrapper's faiss index -> item id), or null
  int* zero_me;           // optional: one int this launch resets (the failed-query counter of the kernels that follow)
  int* fail_list; int* n_fail;   // optional (mode 0): a failed query appends itself here (n_fail reset by an earlier launch's zero_me)
  int lds_keys;           // > 0: key slots in dynamic LDS behind the sort buffer (set by launch_finalize for small launches)
  int sort_slots;         // uint64 slots of the sort buffer in front of them
};

#ifdef RIHIP_FIN_PROBE
__device__ unsigned long long g_fin_probe[16];
#define FIN_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_fin_probe[k] = wall_clock64(); } while (0)
extern "C" int rihip_debug_fin_probe(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fin_probe), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : 1;
}
#else
#define FIN_STAMP(k)
#endif
// radix passes + compaction of finalize_kernel over the key list `kp` (the query's slice of the global candidate list, or
// its copy in LDS: the address space is inferred after inlining, so the LDS call compiles to ds_* instructions)
__device__ __forceinline__ uint64_t finalize_select(const FinArgs& a, const uint64_t* kp, const int64_t n, const int k_sel,
                                                    unsigned* hist, unsigned& s_bin, unsigned& s_above, unsigned& s_cnt,
                                                    uint64_t* sbuf, int P) {
  const int tid = threadIdx.x, lane = tid & 63;
  uint64_t T = 0;  // k_sel-th largest key
  if (k_sel > 0) {
    uint64_t prefix = 0, mask = 0;
    unsigned need = (unsigned)k_sel;
    for (int pass = 0; pass < 8; ++pass) {
      const int shift = 56 - 8 * pass;
      hist[tid] = 0;
      __syncthreads();
      {  // scores share their leading bytes: in the first passes almost every key lands in the same one or two bins, and
         // 12k atomics on one LDS word serialise (that, not the memory passes, was most of this kernel's time for a single
         // request).  Each thread counts runs of equal bins in a register and issues one atomic per run.
        unsigned run_bin = 0xFFFFFFFFu, run_cnt = 0;
        for (int64_t i = tid; i < n; i += 256) {
          const uint64_t key = kp[i];
          if ((key & mask) == prefix) {
            const unsigned b = (unsigned)(key >> shift) & 255u;
            if (b == run_bin) ++run_cnt;
            else { if (run_cnt) atomicAdd(&hist[run_bin], run_cnt); run_bin = b; run_cnt = 1; }
          }
        }
        if (run_cnt) atomicAdd(&hist[run_bin], run_cnt);
      }
      __syncthreads();
      if (tid < 64) {  // wave 0: suffix scan over bins 255..0, 4 bins per lane
        const int b0 = 255 - 4 * lane;
        const unsigned h0 = hist[b0], h1 = hist[b0 - 1], h2 = hist[b0 - 2], h3 = hist[b0 - 3];
        const unsigned mine = h0 + h1 + h2 + h3;
        unsigned incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const unsigned t = __shfl_up(incl, o, 64);
          if (lane >= o) incl += t;
        }
        const unsigned before = incl - mine;
        if (before < need && need <= incl) {
          unsigned c = before;
          int b = b0;
          if (c + h0 >= need) { b = b0; }
          else { c += h0; if (c + h1 >= need) { b = b0 - 1; }
          else { c += h1; if (c + h2 >= need) { b = b0 - 2; }
          else { c += h2; b = b0 - 3; } } }
          s_bin = (unsigned)b;
          s_above = c;
        }
      }
      __syncthreads();
      need -= s_above;
      prefix |= (uint64_t)s_bin << shift;
      mask |= 0xFFull << shift;
      // every key left in the chosen bucket is needed => the remaining low bytes cannot change the selection:
      // stop (typically after 3-4 of the 8 passes); T = prefix with zero low bytes still satisfies
      // #{key >= T} == k_sel
      const bool done = (hist[s_bin] == need);
      __syncthreads();
      FIN_STAMP(4 + pass);
      if (done) break;
    }
    T = prefix;
  }
  if (a.mode == 1) return T;
  // compact keys >= T (exactly k_sel of them: keys are unique), pad to pow2
  if (tid == 0) s_cnt = 0;
  for (int i = tid; i < P; i += 256) sbuf[i] = 0ull;
  __syncthreads();
  if (k_sel > 0) {
    for (int64_t i = tid; i < n; i += 256) {
      const uint64_t key = kp[i];
      if (key > T) {
        const unsigned pos = atomicAdd(&s_cnt, 1u);
        if (pos < (unsigned)P) sbuf[pos] = key;
      }
    }
  }
  __syncthreads();
  return T;
}

__global__ __launch_bounds__(256) void finalize_kernel(FinArgs a) {
  __shared__ unsigned hist[256];
  extern __shared__ __attribute__((aligned(16))) uint64_t sbuf[];  // [pow2 >= k] (mode 0 only), then [lds_keys] key copy
  __shared__ unsigned s_bin, s_above, s_cnt;
  const int tid = threadIdx.x;
  const int64_t qi = blockIdx.x;
  if (a.zero_me && qi == 0 && tid == 0) *a.zero_me = 0;
  const int64_t q = a.qmap ? a.qmap[qi] : qi;
  const int cnt_raw = a.count ? a.count[q * (a.count_stride > 1 ? a.count_stride : 1)] : (int)a.cap;   // null: full lists
  const int64_t n = cnt_raw < a.cap ? cnt_raw : a.cap;
  const uint64_t* keys = a.cand + (size_t)q * a.cap;
  const int64_t oslot = a.out_slot ? a.out_slot[qi] : qi;

  bool fail = (cnt_raw > a.cap) || (a.need_min > 0 && cnt_raw < a.need_min);
  if (a.ivf_thr && cnt_raw < a.k && a.ivf_thr[q] > -INFINITY) fail = true;
  int k_sel = (a.mode == 0) ? a.k : a.rank;
  if (k_sel > n) k_sel = (int)n;
  int P = 64;
  while (P < k_sel) P <<= 1;

  // Small launches (single requests: a handful of workgroups, each a chain of up to 8 dependent passes over its list)
  // copy the list into LDS once and select there: 48 -> 2x us for the 12k-slot lists of one request's probes.
  uint64_t T;
  FIN_STAMP(0);
  if (a.lds_keys > 0 && n <= a.lds_keys) {
    uint64_t* kS = sbuf + a.sort_slots;
    for (int64_t i0 = tid; i0 < n; i0 += 256 * 8) {     // 8 independent loads in flight per thread
      uint64_t v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int64_t i = i0 + u * 256; v[u] = keys[i < n ? i : n - 1]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int64_t i = i0 + u * 256; if (i < n) kS[i] = v[u]; }
    }
    __syncthreads();
    FIN_STAMP(1);
    T = finalize_select(a, kS, n, k_sel, hist, s_bin, s_above, s_cnt, sbuf, P);
  } else {
    T = finalize_select(a, keys, n, k_sel, hist, s_bin, s_above, s_cnt, sbuf, P);
  }

  if (a.mode == 1) {
    if (tid == 0) a.thr_out[q] = (k_sel > 0 && k_sel == a.rank) ? ord2f((uint32_t)(T >> 32)) : -INFINITY;
    return;
  }
  FIN_STAMP(2);
  {  // keys are unique except the all-zero padding key: the remaining slots all equal T
    const int cgt = (int)s_cnt;
    for (int i = cgt + tid; i < k_sel; i += 256) sbuf[i] = T;
  }
  __syncthreads();
  // (the hierarchical select's first level hands its k keys to a second finalize, which sorts: no sort here -- 45
  // barrier-separated stages, 12 us, for nothing)
  const bool need_sort = a.out_keys == nullptr || a.thr_chk != nullptr;
  for (int size = 2; need_sort && size <= P; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int i = tid; i < P / 2; i += 256) {
        const int lo = (i / stride) * (stride << 1) + (i % stride);
        const int hi = lo + stride;
        const bool desc = ((lo & size) == 0);
        const uint64_t x = sbuf[lo], y = sbuf[hi];
        if (desc ? (x < y) : (x > y)) { sbuf[lo] = y; sbuf[hi] = x; }
      }
      __syncthreads();
    }
  }
  FIN_STAMP(3);
  if (a.thr_chk && k_sel == a.k && k_sel > 0) {  // completeness proof of the approximate filter
    const float sk = ord2f((uint32_t)(sbuf[k_sel - 1] >> 32));
    if (sk < a.thr_chk[q] + a.eps_scale * a.qnorm[q] + 2e-6f) fail = true;
  }
  if (a.fail_flags && tid == 0) a.fail_flags[q] = fail ? 1 : 0;
  if (fail && a.fail_list && tid == 0) a.fail_list[atomicAdd(a.n_fail, 1)] = (int)q;   // (order immaterial: each is re-done on its own)
  if (a.out_keys) {
    for (int i = tid; i < a.k; i += 256) a.out_keys[oslot * a.k + i] = i < k_sel ? sbuf[i] : 0ull;
    return;
  }
  for (int i = tid; i < a.k; i += 256) {
    float sc = -INFINITY;
    int64_t row = -1;
    if (i < k_sel && sbuf[i] != 0ull) {
      const uint64_t key = sbuf[i];
      sc = ord2f((uint32_t)(key >> 32));
      row = (int64_t)(0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFull));
      if (a.id_map) row = a.id_map[row];
    }
    a.out_scores[oslot * a.k + i] = sc;
    a.out_rows[oslot * a.k + i] = row;
  }
}

// ---- fused refinement of the two-precision search (one workgroup per query, candidates in LDS) ------------------
// segments -> LDS | k-th largest APPROXIMATE score (radix select) | exact f32 re-score of the candidates that can still
// reach the top-k | top-k select + sort of the exact keys | completeness proof -> outputs.  Same selections and the same
// arithmetic as compact_segments_kernel + finalize_kernel(mode 1) + rerank_kernel + finalize_kernel(mode 0), without
// the three candidate-list round trips through HBM and three kernel boundaries.
struct RefineArgs {
  const uint64_t* seg; const int* seg_cnt; int nsplit, seg_cap;
  int64_t cap;            // candidate slots per query of the global scratch list `cand`
  int lds_slots;          // candidate slots in LDS; a longer list is refined in `cand` (same code, slower)
  uint64_t* cand;         // [nq, cap]
  const float* X; const float* Q; int64_t N; int k;
  const float* thr;       // approximate-score threshold the filter used (completeness proof)
  float eps_scale;
  float* out_scores; int64_t* out_rows; int* fail_flags;
  int* fail_list; int* n_fail;   // failed queries are appended here (n_fail zeroed by an earlier launch)
  const int64_t* id_map;  // optional: out_rows[i] = id_map[row] (the wrapper's faiss index -> item id), or null
};

// k_sel-th largest of n unique keys (LDS or global); returns the prefix T with zero low bytes once every key left in
// the chosen bucket is needed (#{key >= T} == k_sel).  All 256 threads call it; hist/s_bin/s_above are workgroup LDS.
__device__ __forceinline__ uint64_t radix_select_256(const uint64_t* keys, int64_t n, int k_sel, unsigned* hist,
                                                     unsigned* s_bin, unsigned* s_above) {
  const int tid = threadIdx.x, lane = tid & 63;
  uint64_t prefix = 0, mask = 0;
  unsigned need = (unsigned)k_sel;
  for (int pass = 0; pass < 8; ++pass) {
    const int shift = 56 - 8 * pass;
    hist[tid] = 0;
    __syncthreads();
    {  // one atomic per run of equal bins (see finalize_select)
      unsigned run_bin = 0xFFFFFFFFu, run_cnt = 0;
      for (int64_t i = tid; i < n; i += 256) {
        const uint64_t key = keys[i];
        if ((key & mask) == prefix) {
          const unsigned b = (unsigned)(key >> shift) & 255u;
          if (b == run_bin) ++run_cnt;
          else { if (run_cnt) atomicAdd(&hist[run_bin], run_cnt); run_bin = b; run_cnt = 1; }
        }
      }
      if (run_cnt) atomicAdd(&hist[run_bin], run_cnt);
    }
    __syncthreads();
    if (tid < 64) {  // wave 0: suffix scan over bins 255..0, 4 bins per lane
      const int b0 = 255 - 4 * lane;
      const unsigned h0 = hist[b0], h1 = hist[b0 - 1], h2 = hist[b0 - 2], h3 = hist[b0 - 3];
      const unsigned mine = h0 + h1 + h2 + h3;
      unsigned incl = mine;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
      }
      const unsigned before = incl - mine;
      if (before < need && need <= incl) {
        unsigned c = before;
        int b = b0;
        if (c + h0 >= need) { b = b0; }
        else { c += h0; if (c + h1 >= need) { b = b0 - 1; }
        else { c += h1; if (c + h2 >= need) { b = b0 - 2; }
        else { c += h2; b = b0 - 3; } } }
        *s_bin = (unsigned)b;
        *s_above = c;
      }
    }
    __syncthreads();
    need -= *s_above;
    prefix |= (uint64_t)(*s_bin) << shift;
    mask |= 0xFFull << shift;
    const bool done = (hist[*s_bin] == need);
    __syncthreads();
    if (done) break;
  }
  return prefix;
}

// the part of refine_kernel after the candidate count is known; `ck` is the LDS list (address space inferred after
// inlining: ds_* instructions) or the query's slice of the global scratch list
template <int D>
__device__ __forceinline__ void refine_body(const RefineArgs& a, uint64_t* ck, uint64_t* sbuf, const int n, const int* off,
                                            unsigned* hist, unsigned& s_bin, unsigned& s_above, unsigned& s_cnt,
                                            const float qn, const float (&qv)[D / 16], const int64_t q) {
  constexpr int PER = D / 16;
  const int tid = threadIdx.x, l16 = tid & 15, grp = tid >> 4;
  for (int s = tid >> 6; s < a.nsplit; s += 4) {          // one wave per segment
    const int cnt = off[s + 1] - off[s];
    const uint64_t* src = a.seg + ((size_t)q * a.nsplit + s) * a.seg_cap;
    for (int i = tid & 63; i < cnt; i += 64) ck[off[s] + i] = src[i];
  }
  __syncthreads();
  // ---- k-th largest approximate score (a lower bound of it: the select stops early)
  float kth = -INFINITY;
  if (n >= a.k) kth = ord2f((uint32_t)(radix_select_256(ck, n, a.k, hist, &s_bin, &s_above) >> 32));
  // At least k candidates have an approximate score >= kth, hence exact scores >= kth - eps, so the exact k-th score
  // T* >= kth - eps; a candidate whose approximate score is below kth - 2 eps has an exact score < kth - eps <= T*: it
  // cannot be in the top-k and its row is not fetched (key zeroed).
  const float cut = kth - 2.f * a.eps_scale * qn - 1e-6f;
  // four candidates per lane group and iteration: their row loads are in flight together
  typedef float rowvec __attribute__((ext_vector_type(PER)));
  for (int i0 = grp * 4; i0 < n; i0 += 64) {
    uint32_t row[4];
    bool go[4];
    rowvec xv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u;
      const uint64_t key = i < n ? ck[i] : 0ull;
      row[u] = 0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFull);
      go[u] = i < n && (int64_t)row[u] < a.N && !(ord2f((uint32_t)(key >> 32)) < cut);
      if (go[u]) xv[u] = *reinterpret_cast<const rowvec*>(a.X + (size_t)row[u] * D + l16 * PER);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      uint64_t nk = 0ull;
      if (go[u]) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < PER; ++j) s = fmaf(qv[j], xv[u][j], s);
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        nk = make_key(s, row[u]);
      }
      if (l16 == 0 && i0 + u < n) ck[i0 + u] = nk;
    }
  }
  __syncthreads();
  // ---- top-k of the exact keys, sorted
  const int64_t need_min = a.k < a.N ? a.k : a.N;
  bool fail = n < need_min;
  int k_sel = a.k < n ? a.k : n;
  uint64_t T = 0;
  if (k_sel > 0) T = radix_select_256(ck, n, k_sel, hist, &s_bin, &s_above);
  int P = 64;
  while (P < k_sel) P <<= 1;
  if (tid == 0) s_cnt = 0;
  for (int i = tid; i < P; i += 256) sbuf[i] = 0ull;
  __syncthreads();
  if (k_sel > 0) {
    for (int i = tid; i < n; i += 256) {
      const uint64_t key = ck[i];
      if (key > T) {
        const unsigned pos = atomicAdd(&s_cnt, 1u);
        if (pos < (unsigned)P) sbuf[pos] = key;
      }
    }
  }
  __syncthreads();
  {  // keys are unique except the all-zero padding key: the remaining slots all equal T
    const int cgt = (int)s_cnt;
    for (int i = cgt + tid; i < k_sel; i += 256) sbuf[i] = T;
  }
  __syncthreads();
  {  // exchanges at distance <= 64 stay inside the 128-key segment one wave owns: barrier only around the others
    int prev = 128;
    for (int size = 2; size <= P; size <<= 1) {
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
        if (stride >= 128 || prev >= 128) __syncthreads();
        else __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        prev = stride;
        for (int i = tid; i < P / 2; i += 256) {
          const int lo = (i / stride) * (stride << 1) + (i % stride);
          const int hi = lo + stride;
          const bool desc = ((lo & size) == 0);
          const uint64_t x = sbuf[lo], y = sbuf[hi];
          if (desc ? (x < y) : (x > y)) { sbuf[lo] = y; sbuf[hi] = x; }
        }
      }
    }
    __syncthreads();
  }
  if (k_sel == a.k && k_sel > 0) {  // completeness proof of the approximate filter
    const float sk = ord2f((uint32_t)(sbuf[k_sel - 1] >> 32));
    if (sk < a.thr[q] + a.eps_scale * qn + 2e-6f) fail = true;
  }
  if (tid == 0) {
    a.fail_flags[q] = fail ? 1 : 0;
    if (fail) a.fail_list[atomicAdd(a.n_fail, 1)] = (int)q;   // (order immaterial: every failed query is re-done on its own)
  }
  for (int i = tid; i < a.k; i += 256) {
    float sc = -INFINITY;
    int64_t row = -1;
    if (i < k_sel && sbuf[i] != 0ull) {
      const uint64_t key = sbuf[i];
      sc = ord2f((uint32_t)(key >> 32));
      row = (int64_t)(0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFull));
      if (a.id_map) row = a.id_map[row];
    }
    a.out_scores[q * a.k + i] = sc;
    a.out_rows[q * a.k + i] = row;
  }
}

template <int D>
__global__ __launch_bounds__(256) void refine_kernel(RefineArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint64_t rbuf[];   // [cap] candidates | [P] sort buffer
  __shared__ unsigned hist[256];
  __shared__ int off[1025];
  __shared__ unsigned s_bin, s_above, s_cnt;
  __shared__ int s_over;
  __shared__ float s_qn;
  constexpr int PER = D / 16;
  const int tid = threadIdx.x, l16 = tid & 15, grp = tid >> 4;
  const int64_t q = blockIdx.x;
  uint64_t* sbuf = rbuf + a.lds_slots;
  {  // exclusive prefix of the (clamped) segment counts: 4 segments per thread, wave scan, 4 wave totals
    int c[4], ov = 0, mine = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int sg = tid * 4 + u;
      int v = sg < a.nsplit ? a.seg_cnt[q * a.nsplit + sg] : 0;
      if (v > a.seg_cap) { ov = 1; v = a.seg_cap; }
      c[u] = v; mine += v;
    }
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(incl, o, 64);
      if ((tid & 63) >= o) incl += t;
    }
    if ((tid & 63) == 63) hist[tid >> 6] = (unsigned)incl;
    if (tid == 0) s_over = 0;
    __syncthreads();
    int base = incl - mine;
    for (int w2 = 0; w2 < (tid >> 6); ++w2) base += (int)hist[w2];
    if (ov) s_over = 1;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int sg = tid * 4 + u;
      if (sg <= a.nsplit) off[sg] = base;
      base += c[u];
    }
    if (tid == 255) {   // base = the total here
      off[a.nsplit] = base;
      if (base > a.cap) s_over = 1;
    }
    __syncthreads();
  }
  float qv[PER];
#pragma unroll
  for (int j = 0; j < PER; ++j) qv[j] = a.Q[q * D + l16 * PER + j];
  {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < PER; ++j) s += qv[j] * qv[j];
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (tid == 0) s_qn = sqrtf(s);
  }
  __syncthreads();
  if (s_over) {   // a segment or the list overflowed: exact re-do of this query (outputs are overwritten by it)
    if (tid == 0) {
      a.fail_flags[q] = 1;
      a.fail_list[atomicAdd(a.n_fail, 1)] = (int)q;
    }
    return;
  }
  const int n = off[a.nsplit];
  if (n <= a.lds_slots) refine_body<D>(a, rbuf, sbuf, n, off, hist, s_bin, s_above, s_cnt, s_qn, qv, q);
  else refine_body<D>(a, a.cand + (size_t)q * a.cap, sbuf, n, off, hist, s_bin, s_above, s_cnt, s_qn, qv, q);
}

__global__ void collect_fail_kernel(const int* __restrict__ flags, int64_t nq, int* list, int* n_fail) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nq && flags[i]) list[atomicAdd(n_fail, 1)] = (int)i;
}

__global__ void gather_rows_kernel(const float* __restrict__ Q, const int* __restrict__ idx, int n, int d, float* out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (int64_t)n * d) out[i] = Q[(size_t)idx[i / d] * d + (i % d)];
}

__global__ void map_rows_kernel(int64_t* rows, int64_t n, const int64_t* __restrict__ ids) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const int64_t r = rows[i]; rows[i] = (r >= 0) ? ids[r] : -1; }
}

// ------------------------------- IVF search: list-major scan ------------------------------------
// faiss IndexIVFFlat.search (reference src/models/faiss_index.py:113,:145): coarse top-nprobe lists per query by inner
// product with the centroids, then an exact scan of the probed lists only.  Batched the list-major way: the
// (query, probed list) pairs are grouped by LIST, and every wave takes (one list, 32 of the queries that probe it,
// a range of the list's 32-row tiles): the 32 queries sit in registers (MFMA B operand), the tile rows are loaded
// straight into the MFMA A-operand registers (no LDS, no barrier: waves are independent, the plan decides how many of
// them a list gets), exact-f32 MFMA, scores >= thr[q] appended to the query's candidate list.  Work = nprobe/nlist of
// the brute force, whatever the batch size; rows of a list are re-read by its query groups from L2.
struct LmArgs {
  const float* X;            // [Np,d] list-ordered corpus
  const float* Q;            // [nq,d]
  const float* thr;          // [nq] or null (every probed row is a candidate)
  uint64_t* cand;            // [nq, cap]
  int64_t cap;
  int* count;                // [nq]
  const int64_t* row_ids;    // [Np] original row of each physical row
  const int64_t* list_poff;  // [nlist+1] first physical row of each list (multiples of 64)
  const int* list_len;       // [nlist] real rows of each list
  const int* list_qoff;      // [nlist+1] first slot of each list in list_q
  const int* list_q;         // [nq*nprobe] query indices grouped by list
  const int* work_off;       // [nlist+1] first work item of each list
  const int* plan;           // [0] = number of work items, [1] = tiles per work item
  int nlist;
  int tile_step;             // visit every tile_step-th tile of a list (threshold sample), 1 = all
  int nprobe;                // list_q holds pair indices q * nprobe + p
  int count_stride;          // ints between two queries' candidate counters (32 = one 128-B line each: same-line
                             // atomics serialise in L2)
  int64_t dense_cap;         // > 0: dense slots, cand = [nq*nprobe, dense_cap] pre-zeroed keys (no atomics)
  int dense_ids;             // dense slots carry the original row id (unfiltered search) instead of 0 (threshold sample)
};

// coarse scores cs[q, c] = <Q[q], C[c]> on exact-f32 MFMA (4 waves x 32 register-stationary queries, centroid tiles
// of 32 through LDS) -- the IndexFlatIP quantizer
template <int D>
__device__ __forceinline__ void ivf_coarse_body(const float* __restrict__ Q, int64_t nq, const float* __restrict__ C,
                                                int nlist, float* cs, int64_t block) {
  constexpr int LDC = D + 4, KB = D / 8;
  constexpr int NV = (32 * (D / 4) + 255) / 256;
  __shared__ __attribute__((aligned(16))) float Cs[32 * LDC];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int r31 = lane & 31, hh = lane >> 5;
  const int64_t q = block * 128 + w * 32 + r31;
  const int64_t qc = q < nq ? q : nq - 1;
  f32x4 xr[KB];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) xr[kb] = *reinterpret_cast<const f32x4*>(&Q[qc * D + kb * 8 + 4 * hh]);
  const int ntile = (nlist + 31) / 32;
  for (int t = 0; t < ntile; ++t) {
    __syncthreads();
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
    if (q < nq) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = t * 32 + acc_row(r, lane);
        if (c < nlist) cs[(size_t)q * nlist + c] = acc[r];
      }
    }
  }
}
template <int D>
__global__ __launch_bounds__(256, 2) void ivf_coarse_kernel(const float* __restrict__ Q, int64_t nq,
                                                            const float* __restrict__ C, int nlist, float* cs) {
  ivf_coarse_body<D>(Q, nq, C, nlist, cs, blockIdx.x);
}

// top-nprobe lists of each query (ties -> lowest list id), one wave per query; counts the probes of every list
__device__ __forceinline__ void ivf_select_body(const float* __restrict__ cs, int64_t nq, int nlist, int nprobe,
                                                int* probe_list, int* list_cnt, float* sc, int64_t block) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t q = block * 4 + w;
  if (q >= nq) return;
  float* my = sc + (size_t)w * nlist;
  for (int c = lane; c < nlist; c += 64) my[c] = cs[(size_t)q * nlist + c];
  __builtin_amdgcn_wave_barrier();
  for (int p = 0; p < nprobe; ++p) {
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int c = lane; c < nlist; c += 64) {
      const float v = my[c];
      if (v > best || (v == best && c < bi)) { best = v; bi = c; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ob = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (bi == 0x7fffffff) bi = -1;  // fewer than nprobe lists (or only NaN scores left)
    if (lane == 0) {
      probe_list[(size_t)q * nprobe + p] = bi;
      if (bi >= 0) { atomicAdd(&list_cnt[bi], 1); my[bi] = -INFINITY; }
    }
    __builtin_amdgcn_wave_barrier();
  }
}
__global__ __launch_bounds__(256) void ivf_select_kernel(const float* __restrict__ cs, int64_t nq, int nlist, int nprobe,
                                                         int* probe_list, int* list_cnt) {
  extern __shared__ float sc[];  // [4][nlist]
  ivf_select_body(cs, nq, nlist, nprobe, probe_list, list_cnt, sc, blockIdx.x);
}

// one workgroup: slot offsets of the lists, the tile split and the work-item offsets; zeroes the candidate counters
__device__ __forceinline__ void ivf_plan_body(const int* __restrict__ list_cnt, const int64_t* __restrict__ list_poff,
                                              int nlist, int tile_step, int target_items, int* list_qoff,
                                              int* list_cur, int* work_off, int* plan, int* count, int64_t nq) {
  __shared__ int part[256];
  __shared__ int s_total;
  const int tid = threadIdx.x;
  const int per = (nlist + 255) / 256;
  const int c0 = tid * per, c1 = (c0 + per < nlist) ? c0 + per : nlist;
  for (int64_t i = tid; i < nq; i += 256) count[i] = 0;
  // exclusive prefix of `mine` over the 256 threads; s_total = sum.  Wave scans + four wave totals (a serial scan by one
  // thread was 3 x 256 dependent LDS round trips = most of this kernel's 10 us)
  auto block_excl = [&](int mine) -> int {
    const int lane = tid & 63, wv = tid >> 6;
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(incl, o, 64);
      if (lane >= o) incl += t;
    }
    if (lane == 63) part[wv] = incl;
    __syncthreads();
    int base = 0;
    for (int k = 0; k < wv; ++k) base += part[k];
    if (tid == 0) s_total = part[0] + part[1] + part[2] + part[3];
    __syncthreads();
    const int out = base + incl - mine;
    __syncthreads();     // part / s_total are reused by the next call
    return out;
  };
  // pass 1: slots (queries per list) and the total work = sum over (list, 32-query group) of the list's sampled tiles
  int m_sum = 0, g_sum = 0;
  for (int c = c0; c < c1; ++c) {
    const int64_t tiles = (list_poff[c + 1] - list_poff[c]) / TRS;
    const int64_t n_seq = (tiles + tile_step - 1) / tile_step;
    m_sum += list_cnt[c];
    g_sum += ((list_cnt[c] + 31) / 32) * (int)n_seq;
  }
  int q_off = block_excl(m_sum);
  const int m_total = s_total;
  (void)block_excl(g_sum);
  const int total_tiles = s_total;
  // tiles per work item: every item gets about the same number of tiles, whatever the length of its list
  int tpi = (total_tiles + target_items - 1) / (target_items > 0 ? target_items : 1);
  if (tpi < 1) tpi = 1;
  // pass 2: work items
  int w_sum = 0;
  for (int c = c0; c < c1; ++c) {
    const int64_t tiles = (list_poff[c + 1] - list_poff[c]) / TRS;
    const int64_t n_seq = (tiles + tile_step - 1) / tile_step;
    w_sum += ((list_cnt[c] + 31) / 32) * (int)((n_seq + tpi - 1) / tpi);
  }
  int w_off = block_excl(w_sum);
  const int n_work = s_total;
  for (int c = c0; c < c1; ++c) {
    list_qoff[c] = q_off; list_cur[c] = q_off; work_off[c] = w_off;
    q_off += list_cnt[c];
    const int64_t tiles = (list_poff[c + 1] - list_poff[c]) / TRS;
    const int64_t n_seq = (tiles + tile_step - 1) / tile_step;
    w_off += ((list_cnt[c] + 31) / 32) * (int)((n_seq + tpi - 1) / tpi);
  }
  if (tid == 0) { list_qoff[nlist] = m_total; work_off[nlist] = n_work; plan[0] = n_work; plan[1] = tpi; }
}
__global__ __launch_bounds__(256) void ivf_plan_kernel(const int* __restrict__ list_cnt, const int64_t* __restrict__ list_poff,
                                                       int nlist, int tile_step, int target_items, int* list_qoff,
                                                       int* list_cur, int* work_off, int* plan, int* count, int64_t nq) {
  ivf_plan_body(list_cnt, list_poff, nlist, tile_step, target_items, list_qoff, list_cur, work_off, plan, count, nq);
}

__global__ void ivf_scatter_kernel(const int* __restrict__ probe_list, int64_t n_pairs, int nprobe, int* list_cur, int* list_q) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pairs) return;
  const int c = probe_list[i];
  if (c >= 0) list_q[atomicAdd(&list_cur[c], 1)] = (int)i;   // the pair index: query = i / nprobe, probe rank = i % nprobe
}

// Small query batches (single requests above all): coarse scores -> probed lists -> plan -> scatter by ONE workgroup in
// one launch instead of a memset and four dependent launches of a few microseconds of work each; the other workgroups of
// the grid zero the dense candidate slots of the unfiltered scan that follows (what was a fifth launch).  The stages
// hand over through global memory: every array is written before the workgroup barrier that precedes its first read.
struct PrepSmallArgs {
  const float* Q; int64_t nq; const float* C; int nlist, nprobe;
  float* cs; int* probe_list; int* list_cnt; const int64_t* list_poff; int tile_step, target_items;
  int *list_qoff, *list_cur, *work_off, *plan, *count; int64_t n_count; int* list_q;
  uint64_t* zero_buf; int64_t zero_n;
};
template <int D>
__global__ __launch_bounds__(256, 2) void ivf_prepare_small_kernel(PrepSmallArgs a) {
  extern __shared__ float sc[];  // [4][nlist]
  const int tid = threadIdx.x;
  if (blockIdx.x > 0) {
    const int64_t stride = (int64_t)(gridDim.x - 1) * 256;
    for (int64_t i = (int64_t)(blockIdx.x - 1) * 256 + tid; i < a.zero_n; i += stride) a.zero_buf[i] = 0ull;
    return;
  }
  for (int c = tid; c < a.nlist; c += 256) a.list_cnt[c] = 0;
  __syncthreads();
  ivf_coarse_body<D>(a.Q, a.nq, a.C, a.nlist, a.cs, 0);      // nq <= 128: one block of the coarse product
  __syncthreads();
  for (int64_t b = 0; b * 4 < a.nq; ++b) ivf_select_body(a.cs, a.nq, a.nlist, a.nprobe, a.probe_list, a.list_cnt, sc, b);
  __syncthreads();
  ivf_plan_body(a.list_cnt, a.list_poff, a.nlist, a.tile_step, a.target_items, a.list_qoff, a.list_cur, a.work_off, a.plan,
                a.count, a.n_count);
  __syncthreads();
  const int64_t n_pairs = a.nq * a.nprobe;
  for (int64_t i = tid; i < n_pairs; i += 256) {
    const int c = a.probe_list[i];
    if (c >= 0) a.list_q[atomicAdd(&a.list_cur[c], 1)] = (int)i;
  }
}

template <int D>
__global__ __launch_bounds__(256, 2) void ivf_scan_lm_kernel(LmArgs a) {
  constexpr int KB = D / 8, LDX = D + 4;
  constexpr int NL = (TRS * (D / 4)) / 64;  // 16-byte pieces per lane per tile (fully coalesced 1-KiB wave loads)
  // wave-private staging tile: no workgroup barrier anywhere (a wave's LDS operations execute in order)
  __shared__ __attribute__((aligned(16))) float Xs_all[4][TRS * LDX];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int r31 = lane & 31, hh = lane >> 5;
  float* Xs = Xs_all[w];
  const int wi = blockIdx.x * 4 + w;
  if (wi >= a.plan[0]) return;  // wave-uniform
  // which list: largest c with work_off[c] <= wi
  int lo = 0, hi = a.nlist;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (a.work_off[mid] <= wi) lo = mid; else hi = mid;
  }
  const int c = lo;
  const int rem = wi - a.work_off[c];
  const int64_t p0 = a.list_poff[c];
  const int64_t tiles = (a.list_poff[c + 1] - p0) / TRS;
  const int tstep = a.tile_step > 1 ? a.tile_step : 1;
  const int64_t n_seq = (tiles + tstep - 1) / tstep;
  const int tpi = a.plan[1];                              // tiles per work item
  const int s_c = (int)((n_seq + tpi - 1) / tpi);
  const int grp = rem / s_c, split = rem % s_c;
  const int q0 = a.list_qoff[c], m = a.list_qoff[c + 1] - q0;
  const int slot = grp * 32 + r31;
  const bool q_ok = slot < m;
  const int pair = a.list_q[q0 + (q_ok ? slot : grp * 32)];   // pair = query * nprobe + probe rank
  const int64_t q = pair / a.nprobe;
  const int len = a.list_len[c];
  f32x4 qf[KB];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) qf[kb] = *reinterpret_cast<const f32x4*>(&a.Q[q * D + kb * 8 + 4 * hh]);
  const float thr = a.thr ? a.thr[q] : -INFINITY;
  uint64_t* my_cand = a.cand + (size_t)q * a.cap;
  // dense mode (threshold sample): slot = (pair, sampled tile, row): no atomics, no row-id gather
  uint64_t* my_dense = a.dense_cap > 0 ? a.cand + (size_t)pair * a.dense_cap : nullptr;
  const int64_t per = tpi;
  const int64_t i0 = (int64_t)split * per;
  const int64_t i1 = (i0 + per < n_seq) ? i0 + per : n_seq;
  if (i0 >= i1) return;

  f32x4 stage[NL];
  auto load_tile = [&](int64_t i) {  // rows of the tile are contiguous: lane l takes bytes [1024 j + 16 l, +16)
    const f32x4* src = reinterpret_cast<const f32x4*>(a.X + (size_t)(p0 + i * tstep * TRS) * D) + lane;
#pragma unroll
    for (int j = 0; j < NL; ++j) stage[j] = src[j * 64];
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      const int idx = j * 64 + lane;                 // 16-byte piece of the tile
      const int r = idx / (D / 4), c4 = idx % (D / 4);
      *reinterpret_cast<f32x4*>(&Xs[r * LDX + c4 * 4]) = stage[j];
    }
  };
  auto chain = [&]() -> f32x16 {
    f32x16 acc = zero16();
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      const f32x4 av = *reinterpret_cast<const f32x4*>(&Xs[r31 * LDX + kb * 8 + 4 * hh]);
      acc = mfma32(av.x, qf[kb].x, acc);
      acc = mfma32(av.y, qf[kb].y, acc);
      acc = mfma32(av.z, qf[kb].z, acc);
      acc = mfma32(av.w, qf[kb].w, acc);
    }
    return acc;
  };
  auto emit = [&](const f32x16& acc, int64_t i) {
    if (!q_ok) return;
    const int64_t t_row0 = i * tstep * TRS;              // first row of the tile inside the list
    const int64_t left = (int64_t)len - t_row0;           // list padding rows are never candidates
    const int n_ok = left >= TRS ? TRS : (left > 0 ? (int)left : 0);
    if (my_dense) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rr = acc_row(r, lane);
        const int64_t sl = i * TRS + rr;
        if (sl < a.dense_cap)
          my_dense[sl] = rr < n_ok ? make_key(acc[r], a.dense_ids ? (uint32_t)a.row_ids[p0 + t_row0 + rr] : 0u) : 0ull;
      }
      return;
    }
    unsigned hits = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (acc_row(r, lane) < n_ok && acc[r] >= thr) hits |= (1u << r);
    if (hits) {
      int pos = atomicAdd(&a.count[q * a.count_stride], __popc(hits));
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (hits & (1u << r)) {
          const int64_t v = p0 + t_row0 + acc_row(r, lane);
          if (pos < a.cap) my_cand[pos] = make_key(acc[r], (uint32_t)a.row_ids[v]);
          ++pos;
        }
      }
    }
  };
  load_tile(i0);
  store_tile();
#pragma unroll 1
  for (int64_t i = i0; i < i1; ++i) {
    const bool more = i + 1 < i1;
    if (more) load_tile(i + 1);          // in flight during this tile's MFMA chain
    const f32x16 acc = chain();
    if (more) store_tile();              // after the chain's LDS reads (same wave: in order)
    emit(acc, i);
  }
}

// ------------------------------------------ handle ---------------------------------------------
__global__ void fill_int_kernel(int* p, int64_t n, int v) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}


template <int D>
void launch_scan(const ScanArgs& a, dim3 grid, hipStream_t st) {
  hipLaunchKernelGGL((scan_kernel<D>), grid, dim3(256), 0, st, a);
}
int dispatch_scan(int d, const ScanArgs& a, dim3 grid, hipStream_t st) {
  if (d == 32) launch_scan<32>(a, grid, st);
  else if (d == 64) launch_scan<64>(a, grid, st);
  else if (d == 128) launch_scan<128>(a, grid, st);
  else { rihip_set_error("ip_index: unsupported embed_dim=%d (32/64/128)", d); return RIHIP_ERR_SHAPE; }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { rihip_set_error("scan launch: %s", hipGetErrorString(e)); return RIHIP_ERR_HIP; }
  return RIHIP_OK;
}
int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { rihip_set_error("%s launch: %s", what, hipGetErrorString(e)); return RIHIP_ERR_HIP; }
  return RIHIP_OK;
}

// finalize launch: the sort buffer is dynamic LDS sized to the power of two >= k (mode 0); mode 1 needs none
int launch_finalize(const FinArgs& f0, unsigned n, hipStream_t st) {
  FinArgs f = f0;
  size_t lds = 0;
  static bool granted = false;
  if (!granted) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(finalize_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)(sizeof(uint64_t) * K_MAX));
    granted = true;
  }
  int P = 0;
  if (f.mode == 0) {
    P = 64;
    while (P < f.k) P <<= 1;
    lds = sizeof(uint64_t) * (size_t)P;
  }
  // few workgroups (single requests / small batches) whose lists fit: select in LDS instead of 4-8 dependent passes over
  // global memory; large launches keep their occupancy (the LDS copy would cut it to one workgroup per CU)
  f.lds_keys = 0; f.sort_slots = P;
  if (n <= 2u * RIHIP_NCU && f.cap > 0 && (size_t)(P + f.cap) <= (size_t)K_MAX) {
    f.lds_keys = (int)f.cap;
    lds = sizeof(uint64_t) * (size_t)(P + f.cap);
  }
  hipLaunchKernelGGL(finalize_kernel, dim3(n), dim3(256), lds, st, f);
  return check_launch("finalize");
}

int pick_nsplit(int64_t nq, int64_t n_tiles) {
  const int64_t qblocks = (nq + QB - 1) / QB;
  int64_t ns = (2 * RIHIP_NCU + qblocks - 1) / qblocks;  // aim at >= 2 workgroups per CU
  if (ns > n_tiles) ns = n_tiles;
  if (ns < 1) ns = 1;
  if (ns > 65535) ns = 65535;
  return (int)ns;
}

int search_chunk(IpIndex* h, const float* Q, int64_t nq, int k, float* out_s, int64_t* out_r, hipStream_t st) {
  const int d = h->d;
  const int64_t Nphys = h->ivf ? h->Np : h->N;
  const int64_t n_tiles = (Nphys + TRS - 1) / TRS;
  const unsigned nqb = (unsigned)((nq + 255) / 256);
  const unsigned qgrid = (unsigned)((nq + QB - 1) / QB);
  RCCHK(h->count.reserve(nq * CSTRIDE));
  RCCHK(h->fail_flags.reserve(nq));
  RCCHK(h->fail_list.reserve(nq));
  RCCHK(h->thr.reserve(nq));
  RCCHK(h->n_fail.reserve(1));
  if (!h->h_nfail) HIPCHK(hipHostMalloc((void**)&h->h_nfail, sizeof(int)));

  ScanArgs sa;
  memset(&sa, 0, sizeof(sa));
  sa.X = h->X; sa.Q = Q; sa.nq = nq; sa.count = h->count.p;
  FinArgs fa;
  memset(&fa, 0, sizeof(fa));
  fa.nq = nq; fa.k = k; fa.count = h->count.p; fa.out_scores = out_s; fa.out_rows = out_r; fa.id_map = h->id_map;

  if (h->ivf) {
    // population upper bound of one query = the nprobe longest lists (padded to the 64-row granule)
    std::vector<int64_t> ll = h->list_len;
    std::sort(ll.begin(), ll.end(), [](int64_t x, int64_t y) { return x > y; });
    int64_t cap_full = 0;
    for (int i = 0; i < h->nprobe && i < (int)ll.size(); ++i) cap_full += ll[i];
    if (cap_full < 1) cap_full = 1;
    const int nlist = h->nlist;
    const int nprobe = h->nprobe < nlist ? h->nprobe : nlist;
    RIHIP_REQUIRE(nlist <= NLIST_MAX, RIHIP_ERR_SHAPE, "ip_index: nlist=%d > %d unsupported", nlist, NLIST_MAX);
    // one IVF pass over `n` queries: plan (tile split for this sampling step) -> list-major scan
    const int target = 16 * RIHIP_NCU;  // wave work items aimed at: 2 waves per SIMD on every CU, twice over
    auto ivf_scan = [&](const float* Qp, int64_t n, const float* thr, uint64_t* cand, int64_t cap, int tile_step,
                        int64_t dense_cap, bool planned, int dense_ids) -> int {
      if (!planned)   // (ivf_prepare already planned for the first scan that follows it)
        hipLaunchKernelGGL(ivf_plan_kernel, dim3(1), dim3(256), 0, st, h->list_cnt.p, h->list_poff, nlist, tile_step, target,
                           h->list_qoff.p, h->list_cur.p, h->work_off.p, h->plan.p, h->count.p, n * CSTRIDE);
      LmArgs x;
      memset(&x, 0, sizeof(x));
      x.X = h->X; x.Q = Qp; x.thr = thr; x.cand = cand; x.cap = cap; x.count = h->count.p; x.row_ids = h->row_ids;
      x.list_poff = h->list_poff; x.list_len = h->list_len_dev; x.list_qoff = h->list_qoff.p; x.list_q = h->list_q.p;
      x.work_off = h->work_off.p; x.plan = h->plan.p; x.nlist = nlist; x.tile_step = tile_step; x.nprobe = nprobe;
      x.dense_cap = dense_cap; x.count_stride = CSTRIDE; x.dense_ids = dense_ids;
      // n_work <= sum over (list, group) of (tiles/tpi + 1) <= target + #(list, query group) pairs
      const int64_t bound = (int64_t)target + nlist + (n * nprobe + 31) / 32 + 4;
      const dim3 grid((unsigned)((bound + 3) / 4));
      if (d == 32) hipLaunchKernelGGL((ivf_scan_lm_kernel<32>), grid, dim3(256), 0, st, x);
      else if (d == 64) hipLaunchKernelGGL((ivf_scan_lm_kernel<64>), grid, dim3(256), 0, st, x);
      else hipLaunchKernelGGL((ivf_scan_lm_kernel<128>), grid, dim3(256), 0, st, x);
      return check_launch("ivf scan");
    };
    // coarse quantizer -> probed lists -> (query, list) pairs grouped by list
    auto ivf_prepare = [&](const float* Qp, int64_t n, int first_tile_step, uint64_t* zero_buf, int64_t zero_n) -> int {
      RCCHK(h->coarse.reserve(n * nlist));
      RCCHK(h->probe_list.reserve(n * nprobe));
      RCCHK(h->list_q.reserve(n * nprobe));
      RCCHK(h->list_cnt.reserve(nlist)); RCCHK(h->list_qoff.reserve(nlist + 1)); RCCHK(h->list_cur.reserve(nlist));
      RCCHK(h->work_off.reserve(nlist + 1)); RCCHK(h->plan.reserve(2));
      if (n <= 8) {   // one launch (see ivf_prepare_small_kernel): the single workgroup takes the queries 4 at a time, so beyond a
                      // handful of queries the four parallel kernels are faster (64 queries: 0.48 vs 0.37 ms per batch)
        PrepSmallArgs p;
        p.Q = Qp; p.nq = n; p.C = h->C; p.nlist = nlist; p.nprobe = nprobe; p.cs = h->coarse.p; p.probe_list = h->probe_list.p;
        p.list_cnt = h->list_cnt.p; p.list_poff = h->list_poff; p.tile_step = first_tile_step; p.target_items = target;
        p.list_qoff = h->list_qoff.p; p.list_cur = h->list_cur.p; p.work_off = h->work_off.p; p.plan = h->plan.p;
        p.count = h->count.p; p.n_count = n * CSTRIDE; p.list_q = h->list_q.p; p.zero_buf = zero_buf; p.zero_n = zero_buf ? zero_n : 0;
        int64_t zb = zero_buf ? (zero_n + 256 * 16 - 1) / (256 * 16) : 0;      // ~16 stores per thread
        if (zb > 2 * RIHIP_NCU) zb = 2 * RIHIP_NCU;
        const dim3 pg((unsigned)(1 + zb));
        const size_t lds = sizeof(float) * 4 * nlist;
        if (d == 32) hipLaunchKernelGGL((ivf_prepare_small_kernel<32>), pg, dim3(256), lds, st, p);
        else if (d == 64) hipLaunchKernelGGL((ivf_prepare_small_kernel<64>), pg, dim3(256), lds, st, p);
        else hipLaunchKernelGGL((ivf_prepare_small_kernel<128>), pg, dim3(256), lds, st, p);
        return check_launch("ivf prepare (small)");
      }
      if (zero_buf) HIPCHK(hipMemsetAsync(zero_buf, 0, sizeof(uint64_t) * (size_t)zero_n, st));   // key 0 = below every score
      HIPCHK(hipMemsetAsync(h->list_cnt.p, 0, sizeof(int) * nlist, st));
      const dim3 cg((unsigned)((n + 127) / 128));
      if (d == 32) hipLaunchKernelGGL((ivf_coarse_kernel<32>), cg, dim3(256), 0, st, Qp, n, h->C, nlist, h->coarse.p);
      else if (d == 64) hipLaunchKernelGGL((ivf_coarse_kernel<64>), cg, dim3(256), 0, st, Qp, n, h->C, nlist, h->coarse.p);
      else hipLaunchKernelGGL((ivf_coarse_kernel<128>), cg, dim3(256), 0, st, Qp, n, h->C, nlist, h->coarse.p);
      hipLaunchKernelGGL(ivf_select_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), sizeof(float) * 4 * nlist, st,
                         h->coarse.p, n, nlist, nprobe, h->probe_list.p, h->list_cnt.p);
      // slot offsets + the work split of the first scan (tile step `first_tile_step`) + zeroed candidate counters
      hipLaunchKernelGGL(ivf_plan_kernel, dim3(1), dim3(256), 0, st, h->list_cnt.p, h->list_poff, nlist, first_tile_step, target,
                         h->list_qoff.p, h->list_cur.p, h->work_off.p, h->plan.p, h->count.p, n * CSTRIDE);
      hipLaunchKernelGGL(ivf_scatter_kernel, dim3((unsigned)((n * nprobe + 255) / 256)), dim3(256), 0, st, h->probe_list.p,
                         n * nprobe, nprobe, h->list_cur.p, h->list_q.p);
      return check_launch("ivf prepare");
    };
    // unfiltered pass (every probed vector is a candidate): small populations, small batches and the exact fallback.
    // Every (query, probed list) pair owns a dense slot range (slot = row inside the list): no atomics, 7 launches.
    int64_t max_tiles = 0;
    for (int c = 0; c < nlist; ++c) max_tiles = std::max<int64_t>(max_tiles, (h->list_len[c] + TR - 1) / TR * (TR / TRS));
    const int64_t cap_lf = std::max<int64_t>(max_tiles * TRS, TRS);   // slots per (query, probe): the longest list
    const int64_t cap_df = cap_lf * nprobe;
    auto ivf_full = [&](const float* Qp, int64_t n, const int* out_slot) -> int {
      RCCHK(h->fcand.reserve(n * cap_df));
      RCCHK(ivf_prepare(Qp, n, 1, h->fcand.p, n * cap_df));      // (also zeroes the dense slots: key 0 = below every score)
      RCCHK(ivf_scan(Qp, n, nullptr, h->fcand.p, cap_df, 1, cap_lf, true, 1));
      // two-level select: the k best keys of every (query, probe) pair (one workgroup per pair), then the k best of a
      // query's nprobe * k survivors -- a single workgroup over all ~100k slots of a query took 150 us
      const int64_t np_ = n * nprobe;
      RCCHK(h->scand.reserve(np_ * k));
      FinArgs f1;      // (count = null: every list is full -- all cap_lf dense slots of a pair, all nprobe * k survivors)
      memset(&f1, 0, sizeof(f1));
      f1.nq = np_; f1.k = k; f1.count = nullptr; f1.cand = h->fcand.p; f1.cap = cap_lf; f1.mode = 0; f1.out_keys = h->scand.p;
      RCCHK(launch_finalize(f1, (unsigned)np_, st));
      FinArgs f2;
      memset(&f2, 0, sizeof(f2));
      f2.nq = n; f2.k = k; f2.count = nullptr; f2.out_scores = out_s; f2.out_rows = out_r; f2.cand = h->scand.p;
      f2.cap = (int64_t)nprobe * k; f2.mode = 0; f2.out_slot = out_slot; f2.id_map = h->id_map;
      RCCHK(launch_finalize(f2, (unsigned)n, st));
      return check_launch("finalize");
    };
    if (h->redo_slots) return ivf_full(Q, nq, h->redo_slots);   // (rihip_ip_index_search_finish: the exact re-do only)
    const int SS = 16;  // threshold sample: every 16th probed tile
    // small populations or small batches (single requests): the unfiltered pass needs no threshold sample, no
    // exactness check and no host sync
    if (cap_full <= 16384 || (double)k * 4.0 > (double)cap_full / SS || nq * cap_df <= (int64_t)(1 << 23)) return ivf_full(Q, nq, nullptr);

    const double m = (double)k / SS;
    const int rank = (int)ceil(m + 4.0 * sqrt(m) + 4.0);
    int64_t cap = 4096;
    while ((double)cap < 2.5 * rank * SS) cap <<= 1;
    if (cap > cap_full) cap = cap_full;
    // pass A: every SS-th tile of each probed list, all of its scores kept in dense per-(query, probe) slots
    const int64_t cap_l = (max_tiles + SS - 1) / SS * TRS;             // sampled rows of the longest list
    const int64_t cap_s = cap_l * nprobe;
    RCCHK(h->scand.reserve(nq * cap_s));
    RCCHK(h->cand.reserve(nq * cap));
    RCCHK(ivf_prepare(Q, nq, SS, h->scand.p, nq * cap_s));      // (also zeroes the sample's dense slots: key 0 = below every score)
    RCCHK(ivf_scan(Q, nq, nullptr, h->scand.p, cap_s, SS, cap_l, true, 0));
    // (the sample's lists are dense: count = null means cap_s keys each; this launch also resets the failed-query counter
    // that the final select appends to -- no fill / collect launches of their own)
    fa.cand = h->scand.p; fa.cap = cap_s; fa.mode = 1; fa.rank = rank; fa.thr_out = h->thr.p; fa.count = nullptr;
    fa.zero_me = h->n_fail.p;
    RCCHK(launch_finalize(fa, (unsigned)nq, st));
    RCCHK(ivf_scan(Q, nq, h->thr.p, h->cand.p, cap, 1, 0, false, 0));                         // pass B: all probed tiles, filtered
    fa.cand = h->cand.p; fa.cap = cap; fa.mode = 0; fa.thr_out = nullptr; fa.fail_flags = h->fail_flags.p;
    fa.count = h->count.p; fa.zero_me = nullptr; fa.fail_list = h->fail_list.p; fa.n_fail = h->n_fail.p;
    fa.ivf_thr = h->thr.p; fa.count_stride = CSTRIDE;
    RCCHK(launch_finalize(fa, (unsigned)nq, st));
    RCCHK(check_launch("finalize"));
    HIPCHK(hipMemcpyAsync(h->h_nfail, h->n_fail.p, sizeof(int), hipMemcpyDeviceToHost, st));
    if (h->defer_check && h->defer_ok) {   // the caller checks later (rihip_ip_index_search_finish): no host sync here
      // finish then waits for THIS point of the stream only: whatever the caller enqueues behind the search keeps the GPU
      // busy while the host is already back (a capturing stream records no event: replays use _last_fail_count)
      hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
      (void)hipStreamIsCapturing(st, &cs);
      h->ev_recorded = false;
      if (cs == hipStreamCaptureStatusNone) {
        if (!h->ev_fail) HIPCHK(hipEventCreateWithFlags(&h->ev_fail, hipEventDisableTiming));
        HIPCHK(hipEventRecord(h->ev_fail, st));
        h->ev_recorded = true;
      }
      h->pending.active = true; h->pending.Q = Q; h->pending.nq = nq; h->pending.k = k; h->pending.out_s = out_s; h->pending.out_r = out_r;
      return RIHIP_OK;
    }
    HIPCHK(hipStreamSynchronize(st));
    const int nf = *h->h_nfail;
    if (nf > 0) {  // threshold too aggressive (or candidate overflow) for these queries: unfiltered re-do
      const int FCH = 64;
      RCCHK(h->fQ.reserve((int64_t)FCH * d));
      for (int f0 = 0; f0 < nf; f0 += FCH) {
        const int nfc = (nf - f0 < FCH) ? nf - f0 : FCH;
        hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((nfc * d + 255) / 256)), dim3(256), 0, st, Q,
                           h->fail_list.p + f0, nfc, d, h->fQ.p);
        RCCHK(ivf_full(h->fQ.p, nfc, h->fail_list.p + f0));
      }
    }
    return RIHIP_OK;
  }

  if (h->N <= 4 * (int64_t)SAMPLE) {
    // small corpus: every score is a candidate (dense slots, no atomics)
    RCCHK(h->cand.reserve(nq * h->N));
    hipLaunchKernelGGL(fill_int_kernel, dim3(nqb), dim3(256), 0, st, h->count.p, nq, (int)h->N);
    sa.n_virtual = h->N; sa.row_stride = 1; sa.thr = nullptr; sa.cand = h->cand.p; sa.cap = h->N; sa.dense = 1;
    sa.nsplit = pick_nsplit(nq, n_tiles);
    RCCHK(dispatch_scan(d, sa, dim3(qgrid, sa.nsplit), st));
    fa.cand = h->cand.p; fa.cap = h->N; fa.mode = 0;
    RCCHK(launch_finalize(fa, (unsigned)nq, st));
    return check_launch("finalize");
  }

  // ---- pass 0: threshold from a strided sample
  const bool two_prec = h->two_precision && h->Xb != nullptr;
  // the two-precision path samples twice as many rows: the threshold estimate tightens (expected survivors per query
  // 2 200 -> 1 700 at k = 500, N = 1 M), which saves more in the filter's emission and in the re-score than the longer
  // sample pass costs (measured 2.33 -> 2.16 ms per 4 096 queries; 3x, 4x the same, 6x slower again)
  const int64_t n_sample = (int64_t)SAMPLE * (two_prec ? 2 : 1);
  const int64_t stride = h->N / n_sample > 0 ? h->N / n_sample : 1;
  const int64_t S = (h->N + stride - 1) / stride;  // virtual rows i*stride < N
  const double m = (double)k * (double)S / (double)h->N;
  // rank of the sample score used as threshold; the two-precision filter needs a little more head-room because
  // the completeness proof asks for s_k >= thr + eps
  const int rank = (int)ceil((m + 4.0 * sqrt(m) + 4.0) * (two_prec ? 1.5 : 1.0));
  const double expect = (double)rank * (double)h->N / (double)S;
  int64_t cap = 4096;
  while ((double)cap < 2.5 * expect) cap <<= 1;
  if (cap > h->N) cap = h->N;
  RCCHK(h->scand.reserve(nq * S));
  RCCHK(h->cand.reserve(nq * cap));
  const unsigned qgrid_b = (unsigned)((nq + QBB - 1) / QBB);
  int main_nsplit = 0;
  int seg_cap = 64;   // per (query, corpus split) segment: 4x the expected survivors, a power of two, set below
  auto bf16_nsplit = [&](int64_t tiles, int wgs_per_cu = 3) -> int {
    int64_t ns = (wgs_per_cu * RIHIP_NCU + qgrid_b - 1) / qgrid_b;  // resident workgroups per CU (launch bounds: filter 3, sample 2)
    if (ns > tiles) ns = tiles;
    if (ns < 1) ns = 1;
    if (ns > 65535) ns = 65535;
    return (int)ns;
  };
  // the filter of a large batch at d = 128 takes the 1 024-query workgroups of scan_bf16_wide_kernel (1 resident per CU)
  const bool wide = two_prec && d == 128 && nq > 2 * QBB && filter_wide_enabled();
  const unsigned qgrid_w = (unsigned)((nq + QB2 - 1) / QB2);
  auto wide_nsplit = [&](int64_t tiles) -> int {
    int64_t ns = (RIHIP_NCU + qgrid_w - 1) / qgrid_w;
    if (ns > tiles) ns = tiles;
    if (ns < 1) ns = 1;
    if (ns > 1024) ns = 1024;
    return (int)ns;
  };
  if (two_prec) {
    const int ns_main = wide ? wide_nsplit((h->N + ST2 - 1) / ST2) : bf16_nsplit((h->N + TRB - 1) / TRB);
    while ((double)seg_cap < 4.0 * expect / ns_main) seg_cap <<= 1;
    if ((int64_t)seg_cap > cap) seg_cap = (int)cap;
    if (wide) while (seg_cap & (seg_cap - 1)) seg_cap &= seg_cap - 1;   // (the wide filter addresses segments by a shift)
  }
  const int64_t sample_tiles = (S + (two_prec ? TRB : TRS) - 1) / (two_prec ? TRB : TRS);
  // The sample pass keeps the SAMPLE_T best scores of every stream (query x corpus split x row half) in registers
  // instead of writing all S scores per query.  The threshold is the rank-th best of their union: exact while no stream
  // holds more than SAMPLE_T of the sample's top `rank` -- with rank <= streams * SAMPLE_T / 4 (mean <= 2 per stream) a
  // stream overflows with probability ~2e-4, and an overflow only lowers the threshold (more survivors, same result).
  // Larger ranks keep the dense sample.
  const bool sample_top = two_prec && (int64_t)rank * 4 <= (int64_t)2 * bf16_nsplit(sample_tiles, 2) * SAMPLE_T;
  const int64_t cap_s = sample_top ? (int64_t)2 * bf16_nsplit(sample_tiles, 2) * SAMPLE_T : S;  // streams = 2 * nsplit
  auto run_scan = [&](const ScanArgs& args, int64_t tiles) -> int {
    ScanArgs x = args;
    if (two_prec) {
      x.Xb = h->Xb;
      const bool use_wide = wide && !x.dense;
      x.nsplit = use_wide ? wide_nsplit(tiles) : bf16_nsplit(tiles, x.dense ? 2 : 3);
      x.qgrid = use_wide ? (int)qgrid_w : (int)qgrid_b;
      if (!x.dense) {
        RIHIP_REQUIRE(x.nsplit <= 1024, RIHIP_ERR_SHAPE, "ip_index: %d corpus splits", x.nsplit);
        RCCHK(h->seg.reserve(nq * x.nsplit * seg_cap));
        RCCHK(h->seg_cnt.reserve(nq * x.nsplit));
        x.seg = h->seg.p; x.seg_cnt = h->seg_cnt.p; x.seg_cap = seg_cap; main_nsplit = x.nsplit;
      }
      if (use_wide) {
        const dim3 gridw(qgrid_w * 8u * (unsigned)((x.nsplit + 7) / 8));
        static bool wide_granted = false;
        if (!wide_granted) {
          HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(scan_bf16_wide_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, WIDE_LDS));
          wide_granted = true;
        }
        hipLaunchKernelGGL(scan_bf16_wide_kernel, gridw, dim3(512), WIDE_LDS, st, x);
        return check_launch("scan_bf16_wide");
      }
      const dim3 grid(qgrid_b * 8u * (unsigned)((x.nsplit + 7) / 8));
      const int mode = x.dense ? (sample_top ? 2 : 1) : 0;
      if (mode == 2) {
        if (d == 32) hipLaunchKernelGGL((scan_bf16_kernel<32, 2>), grid, dim3(256), 0, st, x);
        else if (d == 64) hipLaunchKernelGGL((scan_bf16_kernel<64, 2>), grid, dim3(256), 0, st, x);
        else hipLaunchKernelGGL((scan_bf16_kernel<128, 2>), grid, dim3(256), 0, st, x);
      } else if (mode == 1) {
        if (d == 32) hipLaunchKernelGGL((scan_bf16_kernel<32, 1>), grid, dim3(256), 0, st, x);
        else if (d == 64) hipLaunchKernelGGL((scan_bf16_kernel<64, 1>), grid, dim3(256), 0, st, x);
        else hipLaunchKernelGGL((scan_bf16_kernel<128, 1>), grid, dim3(256), 0, st, x);
      } else {
        if (d == 32) hipLaunchKernelGGL((scan_bf16_kernel<32, 0>), grid, dim3(256), 0, st, x);
        else if (d == 64) hipLaunchKernelGGL((scan_bf16_kernel<64, 0>), grid, dim3(256), 0, st, x);
        else hipLaunchKernelGGL((scan_bf16_kernel<128, 0>), grid, dim3(256), 0, st, x);
      }
      return check_launch("scan_bf16");
    }
    x.nsplit = pick_nsplit(nq, tiles);
    return dispatch_scan(d, x, dim3(qgrid, x.nsplit), st);
  };
  sa.n_virtual = S; sa.row_stride = stride; sa.thr = nullptr; sa.cand = h->scand.p; sa.cap = cap_s; sa.dense = 1;
  RCCHK(run_scan(sa, sample_tiles));   // (the register top-T sample writes every stream slot, empty streams as key 0)
  fa.cand = h->scand.p; fa.cap = cap_s; fa.mode = 1; fa.rank = rank; fa.thr_out = h->thr.p;
  fa.count = nullptr;                  // dense sample lists: cap_s keys each
  fa.zero_me = h->n_fail.p;            // (reset here: refine_kernel appends the failed queries itself)
  RCCHK(launch_finalize(fa, (unsigned)nq, st));
  fa.count = h->count.p; fa.zero_me = nullptr;
  // ---- pass 1: thresholded scan (survivors into per-(query, split) segments; the f32 scan appends through `count`)
  if (!two_prec)
    hipLaunchKernelGGL(fill_int_kernel, dim3((unsigned)((nq * CSTRIDE + 255) / 256)), dim3(256), 0, st, h->count.p, nq * CSTRIDE, 0);
  sa.n_virtual = h->N; sa.row_stride = 1; sa.thr = h->thr.p; sa.cand = h->cand.p; sa.cap = cap; sa.dense = 0;
  sa.cs = CSTRIDE; fa.count_stride = CSTRIDE;
  RCCHK(run_scan(sa, two_prec ? (h->N + TRB - 1) / TRB : n_tiles));
  bool refined = false;
  if (two_prec && k <= 2048) {
    // one workgroup per query takes the survivors from the segments through approximate select, exact re-score,
    // top-k select + sort and the completeness proof (refine_kernel)
    RefineArgs r;
    memset(&r, 0, sizeof(r));
    r.seg = h->seg.p; r.seg_cnt = h->seg_cnt.p; r.nsplit = main_nsplit; r.seg_cap = seg_cap; r.cap = cap;
    r.lds_slots = 2048; r.cand = h->cand.p; r.X = h->X; r.Q = Q; r.N = h->N; r.k = k; r.thr = h->thr.p;
    r.eps_scale = (float)((1.0 / 256.0 + 1.0 / 262144.0) * (double)h->max_norm * 1.001);
    r.out_scores = out_s; r.out_rows = out_r; r.fail_flags = h->fail_flags.p; r.id_map = h->id_map;
    r.fail_list = h->fail_list.p; r.n_fail = h->n_fail.p;
    int P = 64;
    while (P < k) P <<= 1;
    const size_t lds = sizeof(uint64_t) * (size_t)(r.lds_slots + P);
    static bool granted = false;
    if (!granted) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(refine_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * (2048 + 2048));
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(refine_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * (2048 + 2048));
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(refine_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * (2048 + 2048));
      granted = true;
    }
    if (d == 32) hipLaunchKernelGGL((refine_kernel<32>), dim3((unsigned)nq), dim3(256), lds, st, r);
    else if (d == 64) hipLaunchKernelGGL((refine_kernel<64>), dim3((unsigned)nq), dim3(256), lds, st, r);
    else hipLaunchKernelGGL((refine_kernel<128>), dim3((unsigned)nq), dim3(256), lds, st, r);
    RCCHK(check_launch("refine"));
    refined = true;
  }
  if (!refined) {
  if (two_prec) {  // exact f32 re-score of the survivors (keys rewritten in place)
      hipLaunchKernelGGL(compact_segments_kernel, dim3((unsigned)nq), dim3(256), 0, st, h->seg.p, h->seg_cnt.p, main_nsplit,
                         seg_cap, h->cand.p, cap, h->count.p, CSTRIDE);
      RCCHK(h->qnorm.reserve(nq));
      RCCHK(h->thr2.reserve(nq));
      {  // k-th largest APPROXIMATE score per query (a lower bound of it: the select stops early): rerank_kernel uses it
         // to skip survivors that provably cannot reach the top-k -- about half of them
        FinArgs f2 = fa;
        f2.cand = h->cand.p; f2.cap = cap; f2.mode = 1; f2.rank = k; f2.thr_out = h->thr2.p; f2.fail_flags = nullptr;
        f2.thr_chk = nullptr; f2.qnorm = nullptr; f2.ivf_thr = nullptr; f2.need_min = 0;
        RCCHK(launch_finalize(f2, (unsigned)nq, st));
      }
      const float eps_sc = (float)((1.0 / 256.0 + 1.0 / 262144.0) * (double)h->max_norm * 1.001);
      if (d == 32) hipLaunchKernelGGL((rerank_kernel<32>), dim3((unsigned)nq), dim3(256), 0, st, h->X, Q, h->cand.p, cap, h->count.p, h->qnorm.p, h->N, h->thr2.p, eps_sc, CSTRIDE);
      else if (d == 64) hipLaunchKernelGGL((rerank_kernel<64>), dim3((unsigned)nq), dim3(256), 0, st, h->X, Q, h->cand.p, cap, h->count.p, h->qnorm.p, h->N, h->thr2.p, eps_sc, CSTRIDE);
      else hipLaunchKernelGGL((rerank_kernel<128>), dim3((unsigned)nq), dim3(256), 0, st, h->X, Q, h->cand.p, cap, h->count.p, h->qnorm.p, h->N, h->thr2.p, eps_sc, CSTRIDE);
      fa.thr_chk = h->thr.p; fa.qnorm = h->qnorm.p;
      fa.eps_scale = (float)((1.0 / 256.0 + 1.0 / 262144.0) * (double)h->max_norm * 1.001);
    }
    // ---- pass 2: finalize + exactness flags
    fa.cand = h->cand.p; fa.cap = cap; fa.mode = 0; fa.fail_flags = h->fail_flags.p;
    fa.need_min = k < h->N ? k : h->N; fa.thr_out = nullptr;
    RCCHK(launch_finalize(fa, (unsigned)nq, st));
  }
  if (!refined) {
    hipLaunchKernelGGL(fill_int_kernel, dim3(1), dim3(64), 0, st, h->n_fail.p, 1, 0);
    hipLaunchKernelGGL(collect_fail_kernel, dim3(nqb), dim3(256), 0, st, h->fail_flags.p, nq, h->fail_list.p, h->n_fail.p);
  }
  RCCHK(check_launch("finalize"));
  HIPCHK(hipMemcpyAsync(h->h_nfail, h->n_fail.p, sizeof(int), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  const int nf = *h->h_nfail;
  if (nf > 0) {  // exact re-do of under/overflowed queries (heavy ties, adversarial data)
    const int FCH = 8;
    RCCHK(h->fcand.reserve((int64_t)FCH * h->N));
    RCCHK(h->fQ.reserve((int64_t)FCH * d));
    RCCHK(h->fcount.reserve(FCH));
    for (int f0 = 0; f0 < nf; f0 += FCH) {
      const int nfc = (nf - f0 < FCH) ? nf - f0 : FCH;
      hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((nfc * d + 255) / 256)), dim3(256), 0, st, Q,
                         h->fail_list.p + f0, nfc, d, h->fQ.p);
      hipLaunchKernelGGL(fill_int_kernel, dim3(1), dim3(64), 0, st, h->fcount.p, nfc, (int)h->N);
      ScanArgs fs;
      memset(&fs, 0, sizeof(fs));
      fs.X = h->X; fs.n_virtual = h->N; fs.row_stride = 1; fs.Q = h->fQ.p; fs.nq = nfc; fs.thr = nullptr;
      fs.cand = h->fcand.p; fs.cap = h->N; fs.count = h->fcount.p; fs.dense = 1; fs.nsplit = pick_nsplit(nfc, n_tiles);
      RCCHK(dispatch_scan(d, fs, dim3(1, fs.nsplit), st));
      FinArgs ff;
      memset(&ff, 0, sizeof(ff));
      ff.cand = h->fcand.p; ff.cap = h->N; ff.count = h->fcount.p; ff.nq = nfc; ff.k = k; ff.mode = 0;
      ff.out_scores = out_s; ff.out_rows = out_r; ff.out_slot = h->fail_list.p + f0; ff.id_map = h->id_map;
      RCCHK(launch_finalize(ff, (unsigned)nfc, st));
    }
    RCCHK(check_launch("fallback"));
  }
  return RIHIP_OK;
}

}  // namespace

namespace rihip_index {
// flat index with N > 4*SAMPLE: the bf16 filter copy of the two-precision search + the row-norm bound it needs
int prepare_flat(IpIndex* h, hipStream_t st) {
  if (h->ivf || h->N <= 4 * (int64_t)SAMPLE) return RIHIP_OK;
  const int64_t n = h->N * h->d;
  hipFree(h->Xb);
  h->Xb = nullptr;
  // (rows padded with zeros to a whole 64-row stage: the LDS-DMA filter has no partial-stage path)
  const int64_t n_pad = ((h->N + ST2 - 1) / ST2) * ST2 * h->d;
  HIPCHK(hipMalloc((void**)&h->Xb, sizeof(__bf16) * (size_t)n_pad));
  if (n_pad > n) HIPCHK(hipMemsetAsync(h->Xb + n, 0, sizeof(__bf16) * (size_t)(n_pad - n), st));
  hipLaunchKernelGGL(to_bf16_kernel, dim3((unsigned)((n / 4 + 255) / 256 + 1)), dim3(256), 0, st, h->X, n, h->Xb);
  int* bits = nullptr;
  HIPCHK(hipMalloc((void**)&bits, sizeof(int)));
  HIPCHK(hipMemsetAsync(bits, 0, sizeof(int), st));
  hipLaunchKernelGGL(rownorm_max_kernel, dim3(1024), dim3(256), 0, st, h->X, h->N, h->d, bits);
  int hb = 0;
  HIPCHK(hipMemcpyAsync(&hb, bits, sizeof(int), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  hipFree(bits);
  float sq;
  memcpy(&sq, &hb, sizeof(float));
  h->max_norm = sqrtf(sq);
  return RIHIP_OK;
}
}  // namespace rihip_index

namespace rihip_index {
__global__ void pad_rows_kernel(const float* __restrict__ src, int64_t n, int du, int d, float* __restrict__ dst) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * d) return;
  const int64_t r = i / d;
  const int k = (int)(i % d);
  dst[i] = k < du ? src[r * du + k] : 0.f;
}
int pad_rows(const float* src, int64_t n, int du, int d, float* dst, hipStream_t st) {
  const int64_t tot = n * d;
  hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, src, n, du, d, dst);
  RIHIP_CHECK_LAUNCH();
  return RIHIP_OK;
}
}  // namespace rihip_index

extern "C" int rihip_ip_index_create(int d, void** handle) {
  RIHIP_REQUIRE(handle, RIHIP_ERR_ARG, "ip_index_create: null handle");
  RIHIP_REQUIRE(d >= 1 && d <= 128, RIHIP_ERR_SHAPE, "ip_index_create: unsupported embed_dim=%d (1..128)", d);
  IpIndex* h = new IpIndex();
  h->du = d;
  h->d = d <= 32 ? 32 : (d <= 64 ? 64 : 128);   // the scan kernels are instantiated for these widths: rows are zero-padded
  *handle = h;
  return RIHIP_OK;
}

extern "C" int rihip_ip_index_destroy(void* handle) {
  IpIndex* h = (IpIndex*)handle;
  if (!h) return RIHIP_OK;
  free_index_arrays(h);
  h->cand.release(); h->scand.release(); h->fcand.release(); h->count.release(); h->fail_flags.release();
  h->fail_list.release(); h->n_fail.release(); h->fcount.release(); h->thr.release(); h->thr2.release(); h->fQ.release();
  h->coarse.release(); h->probe_list.release(); h->list_q.release(); h->list_cnt.release(); h->list_qoff.release();
  h->list_cur.release(); h->work_off.release(); h->plan.release(); h->qnorm.release(); h->seg.release(); h->seg_cnt.release();
  h->qpad.release();
  if (h->h_nfail) hipHostFree(h->h_nfail);
  if (h->ev_fail) (void)hipEventDestroy(h->ev_fail);
  delete h;
  return RIHIP_OK;
}

// Replace the index content with N vectors (host or device pointer; copied, caller keeps ownership).
extern "C" int rihip_ip_index_set_vectors(void* handle, const float* X, int64_t N, int x_on_device, void* stream) {
  IpIndex* h = (IpIndex*)handle;
  RIHIP_REQUIRE(h && X && N > 0 && N < (1ll << 31), RIHIP_ERR_ARG, "ip_index_set_vectors: bad arguments");
  free_index_arrays(h);
  HIPCHK(hipMalloc((void**)&h->X, sizeof(float) * (size_t)N * h->d));
  if (h->du == h->d) {
    HIPCHK(hipMemcpyAsync(h->X, X, sizeof(float) * (size_t)N * h->d, x_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                          (hipStream_t)stream));
  } else {   // zero-pad the rows to the kernel width
    const float* src = X;
    float* tmp = nullptr;
    if (!x_on_device) {
      HIPCHK(hipMalloc((void**)&tmp, sizeof(float) * (size_t)N * h->du));
      if (hipMemcpyAsync(tmp, X, sizeof(float) * (size_t)N * h->du, hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess) {
        hipFree(tmp); rihip_set_error("ip_index_set_vectors: upload failed"); return RIHIP_ERR_HIP;
      }
      src = tmp;
    }
    const int rc = pad_rows(src, N, h->du, h->d, h->X, (hipStream_t)stream);
    (void)hipStreamSynchronize((hipStream_t)stream);
    if (tmp) hipFree(tmp);
    if (rc) return rc;
  }
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  h->N = N;
  RCCHK(prepare_flat(h, (hipStream_t)stream));
  return RIHIP_OK;
}

extern "C" int rihip_ip_index_set_two_precision(void* handle, int enable) {
  IpIndex* h = (IpIndex*)handle;
  RIHIP_REQUIRE(h, RIHIP_ERR_ARG, "ip_index_set_two_precision: null handle");
  h->two_precision = enable ? 1 : 0;
  rihip_bump_generation();
  return RIHIP_OK;
}

extern "C" int64_t rihip_ip_index_ntotal(void* handle) { return handle ? ((IpIndex*)handle)->N : 0; }
extern "C" int rihip_ip_index_is_ivf(void* handle) { return handle ? (((IpIndex*)handle)->ivf ? 1 : 0) : 0; }
extern "C" int rihip_ip_index_max_k(void) { return K_MAX; }

extern "C" int rihip_ip_index_set_nprobe(void* handle, int nprobe) {
  IpIndex* h = (IpIndex*)handle;
  RIHIP_REQUIRE(h && nprobe >= 1, RIHIP_ERR_ARG, "ip_index_set_nprobe: bad arguments");
  if (h->nprobe != nprobe) rihip_bump_generation();
  h->nprobe = nprobe;
  return RIHIP_OK;
}

// queries: device [nq,d]; outputs: device scores f32[nq,k] (desc, -inf pad), rows i64[nq,k] (-1 pad)
extern "C" int rihip_ip_index_search(void* handle, const float* Q, int64_t nq, int k, float* out_scores,
                                     int64_t* out_rows, void* stream) {
  IpIndex* h = (IpIndex*)handle;
  RIHIP_REQUIRE(h && h->X && h->N > 0, RIHIP_ERR_STATE, "ip_index_search: index is empty");
  RIHIP_REQUIRE(Q && out_scores && out_rows && nq > 0, RIHIP_ERR_ARG, "ip_index_search: bad arguments");
  RIHIP_REQUIRE(k >= 1 && k <= K_MAX, RIHIP_ERR_ARG, "ip_index_search: k=%d outside [1,%d]", k, K_MAX);
  if (h->du != h->d) {   // zero-pad the queries to the kernel width (scratch of the handle)
    RCCHK(h->qpad.reserve(nq * h->d));
    RCCHK(pad_rows(Q, nq, h->du, h->d, h->qpad.p, (hipStream_t)stream));
    Q = h->qpad.p;
  }
  RIHIP_REQUIRE((reinterpret_cast<uintptr_t>(Q) & 15) == 0, RIHIP_ERR_ARG, "ip_index_search: Q must be 16-byte aligned");
  const int64_t CH = 4096;  // queries per internal pass (bounds scratch)
  h->pending.active = false;
  h->defer_ok = nq <= CH;
  for (int64_t q0 = 0; q0 < nq; q0 += CH) {
    const int64_t n = (nq - q0 < CH) ? nq - q0 : CH;
    int rc = search_chunk(h, Q + q0 * h->d, n, k, out_scores + q0 * k, out_rows + q0 * k, (hipStream_t)stream);
    if (rc) return rc;
  }
  return RIHIP_OK;
}

// Deferred exactness check (serving chains): with enable = 1 a thresholded IVF search of <= 4 096 queries enqueues its
// work and returns WITHOUT the host synchronisation that reads the count of queries whose candidate threshold was too
// aggressive; the caller enqueues whatever consumes the results, then calls rihip_ip_index_search_finish (one sync):
// n_redone > 0 means that many queries were re-done exactly into the same output rows AFTER the consumers ran -- run them
// again.  finish must be called before the next search of the handle.  Other search paths are unaffected (n_redone = 0).
extern "C" int rihip_ip_index_set_deferred_check(void* handle, int enable) {
  IpIndex* h = (IpIndex*)handle;
  RIHIP_REQUIRE(h, RIHIP_ERR_ARG, "ip_index_set_deferred_check: null handle");
  h->defer_check = enable != 0;
  h->pending.active = false;   // (also drops a pending check: what a caller does after CAPTURING a chain, which ran nothing)
  return RIHIP_OK;
}
// 1 while a deferred search awaits its finish (what a hipGraph capture of a chain needs to know about itself)
extern "C" int rihip_ip_index_search_pending(void* handle) { return handle && ((IpIndex*)handle)->pending.active ? 1 : 0; }
// The failure count the LAST enqueued deferred search wrote (after a synchronisation of `stream`): for replays of a
// captured chain, which run no host code -- n > 0: run the chain again eagerly, not deferred.
extern "C" int rihip_ip_index_last_fail_count(void* handle, int* n, void* stream) {
  IpIndex* h = (IpIndex*)handle;
  RIHIP_REQUIRE(h && n && h->h_nfail, RIHIP_ERR_ARG, "ip_index_last_fail_count: no deferred search has run");
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  *n = *h->h_nfail;
  return RIHIP_OK;
}
extern "C" int rihip_ip_index_search_finish(void* handle, int* n_redone, void* stream) {
  IpIndex* h = (IpIndex*)handle;
  RIHIP_REQUIRE(h && n_redone, RIHIP_ERR_ARG, "ip_index_search_finish: bad arguments");
  *n_redone = 0;
  if (!h->pending.active) return RIHIP_OK;
  hipStream_t st = (hipStream_t)stream;
  h->pending.active = false;
  if (h->ev_recorded) HIPCHK(hipEventSynchronize(h->ev_fail));
  else HIPCHK(hipStreamSynchronize(st));
  const int nf = *h->h_nfail;
  if (nf <= 0) return RIHIP_OK;
  const int FCH = 64, d = h->d;
  RCCHK(h->fQ.reserve((int64_t)FCH * d));
  for (int f0 = 0; f0 < nf; f0 += FCH) {
    const int nfc = (nf - f0 < FCH) ? nf - f0 : FCH;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((nfc * d + 255) / 256)), dim3(256), 0, st, h->pending.Q,
                       h->fail_list.p + f0, nfc, d, h->fQ.p);
    h->redo_slots = h->fail_list.p + f0;
    const int rc = search_chunk(h, h->fQ.p, nfc, h->pending.k, h->pending.out_s, h->pending.out_r, st);
    h->redo_slots = nullptr;
    if (rc) return rc;
  }
  HIPCHK(hipStreamSynchronize(st));
  *n_redone = nf;
  return RIHIP_OK;
}

extern "C" int rihip_ip_index_set_id_map(void* handle, const int64_t* item_ids_dev) {
  IpIndex* h = (IpIndex*)handle;
  RIHIP_REQUIRE(h, RIHIP_ERR_ARG, "ip_index_set_id_map: null handle");
  if (h->id_map != item_ids_dev) rihip_bump_generation();
  h->id_map = item_ids_dev;   // not owned: must stay valid (>= ntotal entries) while searches run; NULL switches it off
  return RIHIP_OK;
}

extern "C" int rihip_map_rows_to_ids(int64_t* rows, int64_t n, const int64_t* item_ids, void* stream) {
  RIHIP_REQUIRE(rows && item_ids && n >= 0, RIHIP_ERR_ARG, "map_rows_to_ids: bad arguments");
  if (n == 0) return RIHIP_OK;
  hipLaunchKernelGGL(map_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rows, n, item_ids);
  RIHIP_CHECK_LAUNCH();
  return RIHIP_OK;
}
