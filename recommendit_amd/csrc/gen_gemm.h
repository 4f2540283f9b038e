// Runtime-shape workgroup GEMM shared by the generic tower kernels (tower_generic.hip) and the generic in-batch sweep
// (loss_generic.hip): a 32-row A tile in LDS times a row-major (or transposed) global matrix streamed through a 32-wide
// k-panel in LDS, exact-f32 MFMA (v_mfma_f32_32x32x2_f32), any sizes (zero-padded to the MFMA tile).
#pragma once
#include "common.h"

namespace rihip_gen {

constexpr int GTM = 32;    // rows per tile
constexpr int GKC = 32;    // k-panel width
constexpr int GLDP = GKC + 4;
constexpr int GNT = 2;     // 32-column output tiles per wave (4 waves x 2 x 32 = 256 columns)

__host__ __device__ inline int up8(int x) { return (x + 7) & ~7; }
__host__ __device__ inline int up32(int x) { return (x + 31) & ~31; }

// One k-panel of B: Wp[n][kk] = B[n][kc + kk] for n < Npad, kk < GKC, zero outside (N, K).  TRANS: B[n][k] = W[k*ldw + n],
// else W[n*ldw + k].  The panel travels global -> registers -> LDS in two steps so that the loads of panel p+1 are in
// flight while panel p is multiplied: these kernels run one or two tiles per workgroup and are bound by dependent
// memory latency (an L2 miss is ~1-2 us), not by bandwidth.  Every load is unconditional (clamped address, value
// selected afterwards): no branch separates the loads of a batch.
constexpr int GPB = 32;    // panel elements per thread at Npad = 256 (256 x 32 / 256 threads)
// element u of thread tid is flat index idx = u*256 + tid of the panel; (n, kk) advance incrementally from u to u+1 (a
// runtime division per element would cost more than the MFMAs of these small tiles: one wave per SIMD hides nothing)
struct PanelIdx {
  int n, kk, dn, dk, Npad;
  template <bool TRANS>
  __device__ __forceinline__ void init(int Npad_, int tid) {
    Npad = Npad_;
    if (TRANS) { n = tid % Npad; kk = tid / Npad; dn = 256 % Npad; dk = 256 / Npad; }   // consecutive threads -> consecutive n
    else { kk = tid & (GKC - 1); n = tid >> 5; dn = 8; dk = 0; }                       // consecutive threads -> consecutive k
  }
  template <bool TRANS>
  __device__ __forceinline__ void next() {
    n += dn; kk += dk;
    if (TRANS && n >= Npad) { n -= Npad; kk += 1; }
  }
};
template <bool TRANS>
__device__ __forceinline__ void panel_load(float (&v)[GPB], const float* __restrict__ W, int ldw, int N, int K, int Npad,
                                           int kc, int tid) {
  const int per = Npad >> 3;     // elements per thread: Npad * GKC / 256
  PanelIdx ix;
  ix.init<TRANS>(Npad, tid);
#pragma unroll
  for (int u = 0; u < GPB; ++u) {
    if (u < per) {
      const int k = kc + ix.kk;
      const int nc = ix.n < N ? ix.n : N - 1, kcl = k < K ? k : K - 1;
      // raw value of the clamped address: nothing consumes it before panel_store, so no wait separates the loads
      v[u] = TRANS ? W[(size_t)kcl * ldw + nc] : W[(size_t)nc * ldw + kcl];
      ix.next<TRANS>();
    }
  }
}
template <bool TRANS>
__device__ __forceinline__ void panel_store(const float (&v)[GPB], float* Wp, int N, int K, int Npad, int kc, int tid) {
  const int per = Npad >> 3;
  PanelIdx ix;
  ix.init<TRANS>(Npad, tid);
#pragma unroll
  for (int u = 0; u < GPB; ++u) {
    if (u < per) {
      Wp[ix.n * GLDP + ix.kk] = (ix.n < N && kc + ix.kk < K) ? v[u] : 0.f;
      ix.next<TRANS>();
    }
  }
}

// acc[t] (tile nt = w + 4t) (+)= As[32][K] . B^T panel-wise.  As: LDS, row stride lda, zero-padded to up8(K); B rows
// n >= N and columns k >= Kw (default K) read as zero.  ACCUM keeps the incoming accumulators.
// Contains workgroup barriers: every wave must call it.
template <bool TRANS, bool ACCUM = false>
__device__ __forceinline__ void wg_gemm(const float* As, int lda, int K, const float* __restrict__ W, int ldw, int N,
                                        float* Wp, f32x16* acc, int tid, int Kw = -1) {
  const int lane = tid & 63, w = tid >> 6;
  const int Kp = up8(K), Npad = up32(N);
  if (Kw < 0) Kw = K;
  if (!ACCUM) {
#pragma unroll
    for (int t = 0; t < GNT; ++t) acc[t] = zero16();
  }
  float pv[GPB];
  panel_load<TRANS>(pv, W, ldw, N, Kw, Npad, 0, tid);
  for (int kc = 0; kc < Kp; kc += GKC) {
    __syncthreads();   // the previous panel is consumed (first pass: the A tile is complete)
    panel_store<TRANS>(pv, Wp, N, Kw, Npad, kc, tid);
    __syncthreads();
    if (kc + GKC < Kp) panel_load<TRANS>(pv, W, ldw, N, Kw, Npad, kc + GKC, tid);   // in flight under the MFMAs below
    const int nb = ((Kp - kc < GKC) ? (Kp - kc) : GKC) >> 3;
    for (int b = 0; b < nb; ++b) {
      const f32x4 av = *reinterpret_cast<const f32x4*>(&As[(lane & 31) * lda + kc + 8 * b + 4 * (lane >> 5)]);
#pragma unroll
      for (int t = 0; t < GNT; ++t) {
        const int nt = w + 4 * t;
        if (nt * 32 < Npad) {
          const f32x4 wv = *reinterpret_cast<const f32x4*>(&Wp[(nt * 32 + (lane & 31)) * GLDP + 8 * b + 4 * (lane >> 5)]);
          acc[t] = mfma32(av.x, wv.x, acc[t]);
          acc[t] = mfma32(av.y, wv.y, acc[t]);
          acc[t] = mfma32(av.z, wv.z, acc[t]);
          acc[t] = mfma32(av.w, wv.w, acc[t]);
        }
      }
    }
  }
}

}  // namespace rihip_gen
