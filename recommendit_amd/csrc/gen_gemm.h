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

// Wp[n][kk] = B[n][kc + kk] for n < Npad, kk < GKC, zero outside (N, K).  TRANS: B[n][k] = W[k*ldw + n], else W[n*ldw + k].
template <bool TRANS>
__device__ __forceinline__ void stage_panel(float* Wp, const float* __restrict__ W, int ldw, int N, int K, int Npad, int kc,
                                            int tid) {
  const int tot = Npad * GKC;
  for (int idx = tid; idx < tot; idx += 256) {
    int n, kk;
    if (TRANS) { n = idx % Npad; kk = idx / Npad; }      // consecutive threads -> consecutive n (contiguous in W)
    else { kk = idx % GKC; n = idx / GKC; }              // consecutive threads -> consecutive k
    const int k = kc + kk;
    float v = 0.f;
    if (n < N && k < K) v = TRANS ? W[(size_t)k * ldw + n] : W[(size_t)n * ldw + k];
    Wp[n * GLDP + kk] = v;
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
  for (int kc = 0; kc < Kp; kc += GKC) {
    __syncthreads();   // the previous panel is consumed (first pass: the A tile is complete)
    stage_panel<TRANS>(Wp, W, ldw, N, Kw, Npad, kc, tid);
    __syncthreads();
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
