// Two-Tower MLP kernels for gfx950: fused  gather -> Linear+ReLU+dropout -> Linear -> L2-normalise
// (forward) and its hand-derived backward, on exact-f32 MFMA (v_mfma_f32_32x32x2_f32).
//
// Replaces the bodies of UserTower.forward / ItemTower.forward
// (reference src/models/two_tower.py:39-42, :68-72) and their autograd backward.
//
// Structure (both kernels): one 256-thread workgroup (4 waves, one per SIMD) walks 64-row
// tiles of the batch (persistent grid-stride loop).  The weight fragments each wave needs as
// MFMA B-operands are *register-stationary* for the whole kernel (loaded once per wave);
// only batch rows move through LDS.  Embedding rows are gathered by id as whole rows
// (256/512-B coalesced float4 segments) straight into the LDS tile.
//
// k-permutation: one ds_read_b128 fetches 4 consecutive k of a row; MFMA step s of an 8-wide
// k-block uses k = 8*kb + 4*(lane>>5) + s for BOTH operands, so A needs one b128 per 4 MFMAs.
#include <stdlib.h>
#include "common.h"
#include "recommendit_hip.h"
#include "tower_args.h"

namespace {

constexpr int TM = 64;  // batch rows per tile

template <int N>
struct WaveTiles {  // how the (TM/32) x (N/32) output tiles of a [64 x N] GEMM map to 4 waves
  static constexpr int CT = N / 32;
  static constexpr int NR = (CT >= 4) ? 2 : 1;
  __device__ static __forceinline__ int ct(int w) { return CT >= 4 ? w : (CT == 2 ? (w & 1) : 0); }
  __device__ static __forceinline__ int rt0(int w) { return CT >= 4 ? 0 : (CT == 2 ? (w >> 1) : w); }
  __device__ static __forceinline__ bool active(int w) { return CT == 1 ? (w < 2) : true; }
};

// acc[t] += A[rows of tile rt0+t][k] * Bfrag ; A row-major in LDS (k contiguous)
template <int KB, int NR>
__device__ __forceinline__ void gemm_rowA_regB(const float* As, int lda, int rt0, const f32x4 (&bf)[KB],
                                               f32x16 (&acc)[NR], int lane) {
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
    for (int t = 0; t < NR; ++t) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(&As[((rt0 + t) * 32 + r) * lda + kb * 8 + 4 * h]);
      acc[t] = mfma32(a.x, bf[kb].x, acc[t]);
      acc[t] = mfma32(a.y, bf[kb].y, acc[t]);
      acc[t] = mfma32(a.z, bf[kb].z, acc[t]);
      acc[t] = mfma32(a.w, bf[kb].w, acc[t]);
    }
  }
}

// B fragment for "out = A . W^T" (W row-major [N][K], k contiguous): lane (j=l&31,h) <- W[ct*32+j][8kb+4h+s]
template <int KB>
__device__ __forceinline__ void load_frag_rows(f32x4 (&bf)[KB], const float* __restrict__ W, int ldw, int K, int row,
                                               int lane) {
  const int h = lane >> 5;
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int k = kb * 8 + 4 * h + s;
      bf[kb][s] = (k < K) ? W[(size_t)row * ldw + k] : 0.f;
    }
  }
}
// B fragment for "out = A . W" (W row-major [K][N]): lane (j,h) <- W[8kb+4h+s][col]
template <int KB>
__device__ __forceinline__ void load_frag_cols(f32x4 (&bf)[KB], const float* __restrict__ W, int ldw, int col,
                                               int lane) {
  const int h = lane >> 5;
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
    for (int s = 0; s < 4; ++s) bf[kb][s] = W[(size_t)(kb * 8 + 4 * h + s) * ldw + col];
  }
}

// fragment-major copy of a row-major weight W[N][K] (k contiguous): Wp[(ct*KB + kb)*64 + lane] = the float4 that
// lane `lane` of the wave owning column tile ct feeds to the 4 MFMAs of k-block kb  => one coalesced 1 KB load per
// wave-instruction instead of 64 strided 16-B pieces
// (both weight matrices of a tower in one launch: blocks [0, nb1) pack W1, the rest W2)
__global__ void pack_frag_rows_kernel(const float* __restrict__ Wa, int Na, int Ka, int KBa, f32x4* __restrict__ Wpa,
                                      int nb1, const float* __restrict__ Wb, int Nb, int Kb, int KBb,
                                      f32x4* __restrict__ Wpb) {
  const bool first = (int)blockIdx.x < nb1;
  const float* __restrict__ W = first ? Wa : Wb;
  f32x4* __restrict__ Wp = first ? Wpa : Wpb;
  const int N = first ? Na : Nb, K = first ? Ka : Kb, KB = first ? KBa : KBb;
  const int i = (first ? blockIdx.x : blockIdx.x - nb1) * blockDim.x + threadIdx.x;
  if (i >= (N / 32) * KB * 64) return;
  const int lane = i & 63, kb = (i >> 6) % KB, ct = (i >> 6) / KB;
  const int row = ct * 32 + (lane & 31), h = lane >> 5;
  f32x4 v;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int k = kb * 8 + 4 * h + s;
    v[s] = (k < K) ? W[(size_t)row * K + k] : 0.f;
  }
  Wp[i] = v;
}


struct PackDesc { const float* W; int N, K, KB; f32x4* Wp; int nb; };
struct PackArgs { PackDesc m[4]; int n; };
// the same packing for up to four matrices (both towers of a step) in one launch
__global__ void pack_frag_rows_multi_kernel(PackArgs a) {
  int b = blockIdx.x, t = 0;
  while (t < a.n - 1 && b >= a.m[t].nb) { b -= a.m[t].nb; ++t; }
  const PackDesc m = a.m[t];
  const int i = b * blockDim.x + threadIdx.x;
  if (i >= (m.N / 32) * m.KB * 64) return;
  const int lane = i & 63, kb = (i >> 6) % m.KB, ct = (i >> 6) / m.KB;
  const int row = ct * 32 + (lane & 31), h = lane >> 5;
  f32x4 v;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int k = kb * 8 + 4 * h + s;
    v[s] = (k < m.K) ? m.W[(size_t)row * m.K + k] : 0.f;
  }
  m.Wp[i] = v;
}

template <int D, int K1P, bool ITEM>
__device__ __forceinline__ void gather_tile(float* Xs, int ldx, const float* __restrict__ table, int64_t n_rows,
                                            const int64_t* __restrict__ ids, const float* __restrict__ genres,
                                            int64_t row_base, int64_t B, int tid, int* err_flag) {
  constexpr int V = D / 4;
  for (int idx = tid; idx < TM * V; idx += 256) {
    const int r = idx / V, c4 = idx % V;
    const int64_t grow = row_base + r;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (grow < B) {
      int64_t id = ids[grow];
      if (id < 0 || id >= n_rows) {
        if (err_flag) *err_flag = 1;
        id = 0;
      }
      v = reinterpret_cast<const f32x4*>(table + (size_t)id * D)[c4];
    }
    *reinterpret_cast<f32x4*>(&Xs[r * ldx + c4 * 4]) = v;
  }
  if (ITEM) {
    constexpr int GW = K1P - D;  // 18 genres + zero pad
    for (int idx = tid; idx < TM * GW; idx += 256) {
      const int r = idx / GW, c = idx % GW;
      const int64_t grow = row_base + r;
      Xs[r * ldx + D + c] = (c < 18 && grow < B) ? genres[grow * 18 + c] : 0.f;
    }
  }
}

// body of the forward kernel; `block` of `nblocks` workgroups work on this tower (the pair kernel below runs the user
// and the item tower of a small batch in ONE launch: a step of B = 256 is bounded by its dependent launches)
template <int D, int H, bool ITEM>
__device__ __forceinline__ void tower_fwd_body(const TowerFwdArgs& a, float* Xs, float* Hs, const int block,
                                               const int nblocks) {
  constexpr int K1 = D + (ITEM ? 18 : 0);
  constexpr int K1P = (K1 + 7) / 8 * 8;
  constexpr int LDX = K1P + 4, LDH = H + 4, LDY = D + 4;
  constexpr int KB1 = K1P / 8, KB2 = H / 8;
  using T1 = WaveTiles<H>;
  using T2 = WaveTiles<D>;
  static_assert(LDY <= LDX, "Y tile must fit in the X tile");

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int ct1 = T1::ct(w), ct2 = T2::ct(w);

  f32x4 w1f[KB1], w2f[KB2];
  if (a.W1p) {
#pragma unroll
    for (int kb = 0; kb < KB1; ++kb) w1f[kb] = a.W1p[(ct1 * KB1 + kb) * 64 + lane];
#pragma unroll
    for (int kb = 0; kb < KB2; ++kb) w2f[kb] = a.W2p[(ct2 * KB2 + kb) * 64 + lane];
  } else {
    load_frag_rows<KB1>(w1f, a.W1, K1, K1, ct1 * 32 + (lane & 31), lane);
    load_frag_rows<KB2>(w2f, a.W2, H, H, ct2 * 32 + (lane & 31), lane);
  }
  const float b1v = a.b1[ct1 * 32 + (lane & 31)];
  const float b2v = a.b2[ct2 * 32 + (lane & 31)];
  const uint64_t seed_mul = a.seed_step ? rihip_splitmix64(a.seed_mul + (uint64_t)(*a.seed_step)) : a.seed_mul;
  const uint32_t inner0 = rihip_lowbias32((uint32_t)(seed_mul >> 32) + 0x9E3779B9u);

  const int64_t ntiles = (a.B + TM - 1) / TM;
  for (int64_t tile = block; tile < ntiles; tile += nblocks) {
    const int64_t row_base = tile * TM;
    gather_tile<D, K1P, ITEM>(Xs, LDX, a.table, a.n_rows, a.ids, a.genres, row_base, a.B, tid, a.err_flag);
    __syncthreads();

    // ---- Linear 1 + ReLU + dropout -> Hs
    if (T1::active(w)) {
      f32x16 acc[T1::NR];
#pragma unroll
      for (int t = 0; t < T1::NR; ++t) acc[t] = zero16();
      gemm_rowA_regB<KB1, T1::NR>(Xs, LDX, T1::rt0(w), w1f, acc, lane);
      const int col = ct1 * 32 + (lane & 31);
      // dropout counters of this tile fit 32 bits (always, short of 2^32 hidden elements): one hash per element
      const uint64_t idx0 = (uint64_t)(a.row0 + row_base) * H;
      const bool idx32 = (idx0 + (uint64_t)TM * H) < (1ull << 32);
      const uint32_t idx_base = (uint32_t)idx0;
#pragma unroll
      for (int t = 0; t < T1::NR; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (T1::rt0(w) + t) * 32 + acc_row(r, lane);
          const int64_t grow = row_base + row;
          float v = fmaxf(acc[t][r] + b1v, 0.f);
          if (a.training) {
            bool keep;
            if (idx32) keep = rihip_keep32((uint32_t)seed_mul, inner0, idx_base + (uint32_t)(row * H + col), a.thresh24);
            else keep = rihip_keep(seed_mul, (uint64_t)(a.row0 + grow) * H + col, a.thresh24);
            v = keep ? v * a.scale : 0.f;
          }
          Hs[row * LDH + col] = v;
          if (a.hid && grow < a.B) a.hid[grow * H + col] = v;
        }
      }
    }
    __syncthreads();

    // ---- Linear 2 -> Ys (aliases Xs; every wave is past its last Xs read)
    float* Ys = Xs;
    if (T2::active(w)) {
      f32x16 acc[T2::NR];
#pragma unroll
      for (int t = 0; t < T2::NR; ++t) acc[t] = zero16();
      gemm_rowA_regB<KB2, T2::NR>(Hs, LDH, T2::rt0(w), w2f, acc, lane);
      const int col = ct2 * 32 + (lane & 31);
#pragma unroll
      for (int t = 0; t < T2::NR; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (T2::rt0(w) + t) * 32 + acc_row(r, lane);
          Ys[row * LDY + col] = acc[t][r] + b2v;
        }
      }
    }
    __syncthreads();

    // ---- row L2-normalise: 4 threads per row, coalesced float4 stores
    {
      const int row = tid >> 2, q = tid & 3;
      constexpr int V = D / 16;  // float4 per thread
      f32x4 y[V];
      float ss = 0.f;
#pragma unroll
      for (int i = 0; i < V; ++i) {
        y[i] = *reinterpret_cast<const f32x4*>(&Ys[row * LDY + (q * V + i) * 4]);
        ss += y[i].x * y[i].x + y[i].y * y[i].y + y[i].z * y[i].z + y[i].w * y[i].w;
      }
      ss += __shfl_xor(ss, 1, 64);
      ss += __shfl_xor(ss, 2, 64);
      const float dn = fmaxf(sqrtf(ss), 1e-12f);
      const int64_t grow = row_base + row;
      if (grow < a.B) {
#pragma unroll
        for (int i = 0; i < V; ++i) {
          f32x4 o = {y[i].x / dn, y[i].y / dn, y[i].z / dn, y[i].w / dn};
          reinterpret_cast<f32x4*>(a.out + grow * D)[q * V + i] = o;
        }
        if (q == 0 && a.denom) a.denom[grow] = dn;
      }
    }
    __syncthreads();
  }
}

template <int D, int H, bool ITEM>
__global__ __launch_bounds__(256, 2) void tower_fwd_kernel(TowerFwdArgs a) {
  constexpr int K1P = (D + (ITEM ? 18 : 0) + 7) / 8 * 8;
  __shared__ __attribute__((aligned(16))) float Xs[TM * (K1P + 4)];  // X tile, later aliased by the Y tile
  __shared__ __attribute__((aligned(16))) float Hs[TM * (H + 4)];
  tower_fwd_body<D, H, ITEM>(a, Xs, Hs, (int)blockIdx.x, (int)gridDim.x);
}
// user tower on blocks [0, grid_u), item tower on the rest
template <int D, int H>
__global__ __launch_bounds__(256, 2) void tower_fwd_pair_kernel(TowerFwdArgs au, TowerFwdArgs ai, int grid_u) {
  constexpr int K1P = (D + 18 + 7) / 8 * 8;
  __shared__ __attribute__((aligned(16))) float Xs[TM * (K1P + 4)];
  __shared__ __attribute__((aligned(16))) float Hs[TM * (H + 4)];
  if ((int)blockIdx.x < grid_u) tower_fwd_body<D, H, false>(au, Xs, Hs, (int)blockIdx.x, grid_u);
  else tower_fwd_body<D, H, true>(ai, Xs, Hs, (int)blockIdx.x - grid_u, (int)gridDim.x - grid_u);
}

// -----------------------------------------------------------------------------------------
// Backward
// -----------------------------------------------------------------------------------------

template <int D, int H, bool ITEM>
__device__ __forceinline__ void tower_bwd_body(const TowerBwdArgs& a, float* Xs, float* Hs, float* Gs, const int block,
                                               const int nblocks) {
  constexpr int K1 = D + (ITEM ? 18 : 0);
  constexpr int K1P = (K1 + 7) / 8 * 8;
  constexpr int LDX = K1P + 4, LDH = H + 4, LDG = D + 4;
  constexpr int CTD = D / 32, CTH = H / 32, NX = (K1P + 31) / 32;
  using TH = WaveTiles<H>;  // dh  = Gy . W2   [64 x H]
  using TD = WaveTiles<D>;  // dx  = dPre . W1 [64 x D]
  // dW2 [D x H] tiles: T2 = CTD*CTH, TPW2 per wave, all of a wave's tiles share the D-tile
  constexpr int T2 = CTD * CTH;
  constexpr int TPW2 = (T2 >= 4) ? T2 / 4 : 1;
  // dW1 [H x K1P] tiles: CTH x NX; CTH==4: wave w owns h-tile w and all NX x-tiles;
  // CTH==2: wave w owns h-tile (w&1) and x-tiles (w>>1), (w>>1)+2, ...
  constexpr int TPW1 = (CTH == 4) ? NX : (NX + 1) / 2;

  // Xs [TM * LDX + 32] (+32: partial last x-tile over-read) | Hs [TM * LDH] hid tile, then dPre tile | Gs [TM * LDG] gy tile
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int r31 = lane & 31, hh = lane >> 5;

  // register-stationary B fragments
  f32x4 w2b[D / 8];  // dh = Gy . W2 : B[k=d][j=h] = W2[k][cth*32+j]
  f32x4 w1b[H / 8];  // dx = dPre . W1[:, :D] : B[k=h][j=x] = W1[k][ctx*32+j]
  load_frag_cols<D / 8>(w2b, a.W2, H, TH::ct(w) * 32 + r31, lane);
  load_frag_cols<H / 8>(w1b, a.W1, K1, TD::ct(w) * 32 + r31, lane);

  // persistent weight-grad accumulators
  f32x16 aW2[TPW2], aW1[TPW1];
#pragma unroll
  for (int t = 0; t < TPW2; ++t) aW2[t] = zero16();
#pragma unroll
  for (int t = 0; t < TPW1; ++t) aW1[t] = zero16();
  float ab1 = 0.f, ab2 = 0.f;  // thread tid<H owns db1[tid]; tid<D owns db2[tid]

  const bool w2_active = (w * TPW2) < T2;
  const int ctd2 = (w * TPW2) / CTH, cth2_0 = (w * TPW2) % CTH;
  const int cth1 = (CTH == 4) ? w : (w & 1);
  const int ctx1_0 = (CTH == 4) ? 0 : (w >> 1);
  constexpr int ctx1_step = (CTH == 4) ? 1 : 2;

  const int64_t ntiles = (a.B + TM - 1) / TM;
  for (int64_t tile = block; tile < ntiles; tile += nblocks) {
    const int64_t row_base = tile * TM;
    // ---- stage: gy (normalise backward), hid, gathered x
    {
      const int row = tid >> 2, q = tid & 3;
      constexpr int V = D / 16;
      const int64_t grow = row_base + row;
      f32x4 g[V], o[V];
      float dot = 0.f;
      float dn = 1.f;
      if (grow < a.B) {
        dn = a.denom[grow];
#pragma unroll
        for (int i = 0; i < V; ++i) {
          g[i] = reinterpret_cast<const f32x4*>(a.gout + grow * D)[q * V + i];
          o[i] = reinterpret_cast<const f32x4*>(a.out + grow * D)[q * V + i];
          dot += g[i].x * o[i].x + g[i].y * o[i].y + g[i].z * o[i].z + g[i].w * o[i].w;
        }
      } else {
#pragma unroll
        for (int i = 0; i < V; ++i) {
          g[i] = f32x4{0.f, 0.f, 0.f, 0.f};
          o[i] = g[i];
        }
      }
      dot += __shfl_xor(dot, 1, 64);
      dot += __shfl_xor(dot, 2, 64);
      if (dn <= 1e-12f) dot = 0.f;  // clamp branch of F.normalize: out = y/eps, d out/dy = 1/eps
#pragma unroll
      for (int i = 0; i < V; ++i) {
        f32x4 gy = {(g[i].x - o[i].x * dot) / dn, (g[i].y - o[i].y * dot) / dn, (g[i].z - o[i].z * dot) / dn,
                    (g[i].w - o[i].w * dot) / dn};
        *reinterpret_cast<f32x4*>(&Gs[row * LDG + (q * V + i) * 4]) = gy;
      }
    }
    for (int idx = tid; idx < TM * (H / 4); idx += 256) {
      const int r = idx / (H / 4), c4 = idx % (H / 4);
      const int64_t grow = row_base + r;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (grow < a.B) v = reinterpret_cast<const f32x4*>(a.hid + grow * H)[c4];
      *reinterpret_cast<f32x4*>(&Hs[r * LDH + c4 * 4]) = v;
    }
    gather_tile<D, K1P, ITEM>(Xs, LDX, a.table, a.n_rows, a.ids, a.genres, row_base, a.B, tid, nullptr);
    __syncthreads();

    // ---- db2 += colsum(gy) ; dW2 += gy^T . hid
    if (tid < D) {
      float s = 0.f;
#pragma unroll 8
      for (int r = 0; r < TM; ++r) s += Gs[r * LDG + tid];
      ab2 += s;
    }
    if (w2_active) {
#pragma unroll 4
      for (int s = 0; s < 32; ++s) {
        const int row = 2 * s + hh;
        const float av = Gs[row * LDG + ctd2 * 32 + r31];
#pragma unroll
        for (int t = 0; t < TPW2; ++t) {
          const float bv = Hs[row * LDH + (cth2_0 + t) * 32 + r31];
          aW2[t] = mfma32(av, bv, aW2[t]);
        }
      }
    }
    // ---- dh = gy . W2  (accumulators only; Hs is overwritten after the barrier)
    f32x16 dh[TH::NR];
#pragma unroll
    for (int t = 0; t < TH::NR; ++t) dh[t] = zero16();
    if (TH::active(w)) gemm_rowA_regB<D / 8, TH::NR>(Gs, LDG, TH::rt0(w), w2b, dh, lane);
    __syncthreads();
    if (TH::active(w)) {
      const int col = TH::ct(w) * 32 + r31;
#pragma unroll
      for (int t = 0; t < TH::NR; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (TH::rt0(w) + t) * 32 + acc_row(r, lane);
          const float hv = Hs[row * LDH + col];
          Hs[row * LDH + col] = (hv > 0.f) ? dh[t][r] * a.scale : 0.f;
        }
      }
    }
    __syncthreads();

    // ---- db1 += colsum(dPre) ; dW1 += dPre^T . x ; dx = dPre . W1[:, :D]
    if (tid < H) {
      float s = 0.f;
#pragma unroll 8
      for (int r = 0; r < TM; ++r) s += Hs[r * LDH + tid];
      ab1 += s;
    }
#pragma unroll 4
    for (int s = 0; s < 32; ++s) {
      const int row = 2 * s + hh;
      const float av = Hs[row * LDH + cth1 * 32 + r31];
#pragma unroll
      for (int t = 0; t < TPW1; ++t) {
        const int ctx = ctx1_0 + t * ctx1_step;
        if (ctx < NX) {
          const float bv = Xs[row * LDX + ctx * 32 + r31];
          aW1[t] = mfma32(av, bv, aW1[t]);
        }
      }
    }
    if (TD::active(w)) {
      f32x16 dx[TD::NR];
#pragma unroll
      for (int t = 0; t < TD::NR; ++t) dx[t] = zero16();
      gemm_rowA_regB<H / 8, TD::NR>(Hs, LDH, TD::rt0(w), w1b, dx, lane);
      const int col = TD::ct(w) * 32 + r31;
#pragma unroll
      for (int t = 0; t < TD::NR; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t grow = row_base + (TD::rt0(w) + t) * 32 + acc_row(r, lane);
          if (grow < a.B) a.dX[grow * D + col] = dx[t][r];
        }
      }
    }
    __syncthreads();
  }

  // ---- write this workgroup's partial weight grads: slab = [dW1 (H*K1) | db1 (H) | dW2 (D*H) | db2 (D)]
  constexpr int P = H * K1 + H + D * H + D;
  float* sl = a.slab + (size_t)block * P;
#pragma unroll
  for (int t = 0; t < TPW1; ++t) {
    const int ctx = ctx1_0 + t * ctx1_step;
    if (ctx < NX) {
      const int xc = ctx * 32 + r31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int hr = cth1 * 32 + acc_row(r, lane);
        if (xc < K1) sl[hr * K1 + xc] = aW1[t][r];
      }
    }
  }
  if (tid < H) sl[H * K1 + tid] = ab1;
  if (w2_active) {
#pragma unroll
    for (int t = 0; t < TPW2; ++t) {
      const int hc = (cth2_0 + t) * 32 + r31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int dr = ctd2 * 32 + acc_row(r, lane);
        sl[H * K1 + H + dr * H + hc] = aW2[t][r];
      }
    }
  }
  if (tid < D) sl[H * K1 + H + D * H + tid] = ab2;
}

template <int D, int H, bool ITEM>
__global__ __launch_bounds__(256) void tower_bwd_kernel(TowerBwdArgs a) {
  constexpr int K1P = (D + (ITEM ? 18 : 0) + 7) / 8 * 8;
  __shared__ __attribute__((aligned(16))) float Xs[TM * (K1P + 4) + 32];
  __shared__ __attribute__((aligned(16))) float Hs[TM * (H + 4)];
  __shared__ __attribute__((aligned(16))) float Gs[TM * (D + 4)];
  tower_bwd_body<D, H, ITEM>(a, Xs, Hs, Gs, (int)blockIdx.x, (int)gridDim.x);
}
// user tower on blocks [0, grid_u), item tower on the rest (each writes the slabs of its own workspace)
template <int D, int H>
__global__ __launch_bounds__(256) void tower_bwd_pair_kernel(TowerBwdArgs au, TowerBwdArgs ai, int grid_u) {
  constexpr int K1P = (D + 18 + 7) / 8 * 8;
  __shared__ __attribute__((aligned(16))) float Xs[TM * (K1P + 4) + 32];
  __shared__ __attribute__((aligned(16))) float Hs[TM * (H + 4)];
  __shared__ __attribute__((aligned(16))) float Gs[TM * (D + 4)];
  if ((int)blockIdx.x < grid_u) tower_bwd_body<D, H, false>(au, Xs, Hs, Gs, (int)blockIdx.x, grid_u);
  else tower_bwd_body<D, H, true>(ai, Xs, Hs, Gs, (int)blockIdx.x - grid_u, (int)gridDim.x - grid_u);
}

// grads (+)= sum over slabs, two levels, fixed order => deterministic
//   level 1: group g sums slabs g, g+G, g+2G, ...  -> part[g][P]      (grid.y = G)
//   level 2: sums the G partials and routes element i to dW1 | db1 | dW2 | db2
constexpr int SLAB_GROUPS = 16;

// ---- dense embedding grad: grad[ids[b]] += dX[b]  (padding row 0 gets no gradient), bitwise reproducible.
// Workgroup w of a table owns the rows with (id - 1) % nwg == w (popular rows of a skewed batch spread over the
// workgroups): it walks the batch in passes of 16384 positions, compacts the positions whose id it owns (in batch
// order), sorts the (row, position) keys in LDS and adds every run of equal rows to the table row sample after sample
// in batch order, starting from the row's current contents -- the float32 chain of index_add_ / np.add.at on a CPU.
// A short run is summed by one lane group straight from memory; a long one (a hot row) has its samples staged into LDS
// by the whole workgroup, 1024 / (d/4) rows per round, and one lane group adds them in order.  No float atomics.
constexpr int SCAT_CHUNK = 16384;     // batch positions per pass (14 bits of the key)
constexpr int SCAT_RBITS = 18;        // rows per workgroup <= 2^18 (18 bits of the key)
constexpr int SCAT_LONG = 48;         // runs longer than this are staged through LDS
struct ScatterDesc {
  float* grad; int64_t n_rows; const int64_t* ids; const float* dX; int64_t B;
};
struct ScatterArgs { ScatterDesc t[2]; int D; int nwg; };

__device__ __forceinline__ int scat_run_end(const uint32_t* keys, int i, int m, uint32_t rid) {
  // first index > i whose row differs (keys sorted): lower_bound of (rid + 1) << 14
  const uint32_t bound = (rid + 1u) << 14;
  int lo_ = i + 1, hi_ = m;
  while (lo_ < hi_) {
    const int mid = (lo_ + hi_) >> 1;
    if (keys[mid] < bound) lo_ = mid + 1; else hi_ = mid;
  }
  return lo_;
}

struct SlabDesc { const float* slab; float* part; int nslab, P, K1; float *dW1, *db1, *dW2, *db2; };
struct SlabArgs { SlabDesc t[2]; int H, D, accumulate, nt; };
// one element of a slab-reduction level for every tower of `a` (the arithmetic of slab_reduce1/2_multi_kernel): used by
// the scatter kernel's extra workgroups -- scatter-add and slab reduction both wait only for the tower backward, so a
// small-batch step runs them in one launch
__device__ __forceinline__ void slab_level_element(const SlabArgs& a, int level, int64_t e) {
  for (int z = 0; z < a.nt; ++z) {
    const SlabDesc& t = a.t[z];
    const int G = t.nslab < SLAB_GROUPS ? t.nslab : SLAB_GROUPS;
    if (level == 1) {
      if (t.nslab <= SLAB_GROUPS || e >= (int64_t)t.P * G) continue;
      const int g = (int)(e / t.P), i = (int)(e % t.P);
      float s = 0.f;
      for (int k = g; k < t.nslab; k += G) s += t.slab[(size_t)k * t.P + i];
      t.part[(size_t)g * t.P + i] = s;
    } else {
      if (e >= t.P) continue;
      const int i = (int)e;
      const float* part = t.nslab > SLAB_GROUPS ? t.part : t.slab;
      float s = 0.f;
      for (int g = 0; g < G; ++g) s += part[(size_t)g * t.P + i];
      const int H = a.H, D = a.D, K1 = t.K1;
      float* dst;
      int off;
      if (i < H * K1) { dst = t.dW1; off = i; }
      else if (i < H * K1 + H) { dst = t.db1; off = i - H * K1; }
      else if (i < H * K1 + H + D * H) { dst = t.dW2; off = i - H * K1 - H; }
      else { dst = t.db2; off = i - H * K1 - H - D * H; }
      dst[off] = a.accumulate ? dst[off] + s : s;
    }
  }
}

template <int LPR>   // lanes per row (float4 each); 0 = generic width, one lane per (run, column)
__global__ __launch_bounds__(1024) void scatter_range_kernel(ScatterArgs a, SlabArgs sl, int slab_level) {
  extern __shared__ uint32_t skeys[];
  if ((int)blockIdx.x >= a.nwg) {   // extra workgroups (grid row 0 only): a slab-reduction level beside the scatter
    if (blockIdx.y == 0 && slab_level > 0)
      slab_level_element(sl, slab_level, (int64_t)((int)blockIdx.x - a.nwg) * 1024 + threadIdx.x);
    return;
  }
  __shared__ int wave_cnt[2][16];
  __shared__ int long_list[SCAT_CHUNK / SCAT_LONG + 1];
  __shared__ int n_long;
  __shared__ __attribute__((aligned(16))) float stage[LPR > 0 ? 4096 : 4];   // 1024 threads x float4
  const ScatterDesc t = a.t[blockIdx.y];
  const int nwg = a.nwg;
  const int64_t own = blockIdx.x;
  if (own + 1 >= t.n_rows) return;   // no row with (id - 1) % nwg == own   (workgroup-uniform)
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int64_t c0 = 0; c0 < t.B; c0 += SCAT_CHUNK) {
    const int n = (int)((t.B - c0 < SCAT_CHUNK) ? t.B - c0 : SCAT_CHUNK);
    // ---- stable compaction of the pass's positions whose row this workgroup owns; all id loads are in flight together
    int64_t idv[SCAT_CHUNK / 1024];
#pragma unroll
    for (int it = 0; it < SCAT_CHUNK / 1024; ++it) {
      const int i = it * 1024 + tid;
      idv[it] = (i < n) ? t.ids[c0 + i] : 0;
    }
    int m = 0;
#pragma unroll
    for (int it = 0; it < SCAT_CHUNK / 1024; ++it) {
      if (it * 1024 >= n) break;   // uniform
      const int64_t id = idv[it];
      const bool in = id >= 1 && id < t.n_rows && (id - 1) % nwg == own;
      const uint64_t bal = __ballot(in);
      if (lane == 0) wave_cnt[it & 1][w] = __popcll(bal);
      __syncthreads();
      int before = 0, total = 0;
#pragma unroll
      for (int x = 0; x < 16; ++x) {
        const int c = wave_cnt[it & 1][x];
        before += (x < w) ? c : 0;
        total += c;
      }
      if (in)
        skeys[m + before + __popcll(bal & ((1ull << lane) - 1ull))] =
            ((uint32_t)((id - 1) / nwg) << 14) | (uint32_t)(it * 1024 + tid);
      m += total;
    }
    if (tid == 0) n_long = 0;
    if (m == 0) { __syncthreads(); continue; }   // uniform
    int np2 = 2;
    while (np2 < m) np2 <<= 1;
    for (int i = m + tid; i < np2; i += 1024) skeys[i] = 0xFFFFFFFFu;
    // ---- bitonic sort of the keys (unique: the position is part of the key).  Exchanges at distance <= 64 stay inside
    // the 128-key segment owned by one wave (LDS operations of a wave execute in order): only the passes that reach
    // across segments, and the first pass after one, need the workgroup barrier.
    int prev_j = 128;
    for (int k = 2; k <= np2; k <<= 1) {
      for (int j = k >> 1; j > 0; j >>= 1) {
        if (j >= 128 || prev_j >= 128) __syncthreads();
        else __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        prev_j = j;
        for (int x = tid; x < (np2 >> 1); x += 1024) {
          const int l = ((x & ~(j - 1)) << 1) | (x & (j - 1));
          const int h = l | j;
          const uint32_t u = skeys[l], v = skeys[h];
          const bool up = (l & k) == 0;
          if ((u > v) == up) { skeys[l] = v; skeys[h] = u; }
        }
      }
    }
    __syncthreads();
    // ---- runs of equal rows: sequential float32 chain onto the table row
    const float* base = t.dX + (size_t)c0 * a.D;
    if (LPR > 0) {
      constexpr int LP = LPR > 0 ? LPR : 1;
      constexpr int D = LP * 4, GPB = 1024 / LP;
      const int g = tid / LP, c4 = tid % LP;
      for (int i = g; i < m; i += GPB) {
        const uint32_t key = skeys[i];
        const uint32_t rid = key >> 14;
        if (i > 0 && (skeys[i - 1] >> 14) == rid) continue;   // not the head of a run
        const int end = scat_run_end(skeys, i, m, rid);
        if (end - i > SCAT_LONG) {
          if (c4 == 0) long_list[atomicAdd(&n_long, 1)] = i;   // different rows: the order of this list is immaterial
          continue;
        }
        f32x4* dst = reinterpret_cast<f32x4*>(t.grad + (size_t)((int64_t)rid * nwg + own + 1) * D) + c4;
        f32x4 acc = *dst;
        for (int j = i; j < end; ++j) {
          const f32x4 v = reinterpret_cast<const f32x4*>(base + (size_t)(skeys[j] & 16383u) * D)[c4];
          acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        *dst = acc;
      }
      __syncthreads();
      const int nl = n_long;
      for (int q = 0; q < nl; ++q) {   // hot rows: uniform loop, the whole workgroup stages, lane group 0 adds
        const int i = long_list[q];
        const uint32_t rid = skeys[i] >> 14;
        const int end = scat_run_end(skeys, i, m, rid);
        f32x4* dst = reinterpret_cast<f32x4*>(t.grad + (size_t)((int64_t)rid * nwg + own + 1) * D) + c4;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (g == 0) acc = *dst;
        for (int r0 = i; r0 < end; r0 += GPB) {
          const int r = r0 + g;
          if (r < end)
            reinterpret_cast<f32x4*>(stage)[g * LP + c4] =
                reinterpret_cast<const f32x4*>(base + (size_t)(skeys[r] & 16383u) * D)[c4];
          __syncthreads();
          if (g == 0) {
            const int cnt = (end - r0 < GPB) ? end - r0 : GPB;
            for (int x = 0; x < cnt; ++x) {
              const f32x4 v = reinterpret_cast<const f32x4*>(stage)[x * LP + c4];
              acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
          }
          __syncthreads();
        }
        if (g == 0) *dst = acc;
      }
    } else {
      const int D = a.D;
      for (int e = tid; e < m * D; e += 1024) {
        const int i = e / D, c = e % D;
        const uint32_t rid = skeys[i] >> 14;
        if (i > 0 && (skeys[i - 1] >> 14) == rid) continue;
        float* dst = t.grad + (size_t)((int64_t)rid * nwg + own + 1) * D + c;
        float acc = *dst;
        for (int j = i; j < m; ++j) {
          const uint32_t kj = skeys[j];
          if ((kj >> 14) != rid) break;
          acc += base[(size_t)(kj & 16383u) * D + c];
        }
        *dst = acc;
      }
    }
    __syncthreads();   // the next pass's keys overwrite skeys; its row updates follow this pass's (same workgroup)
  }
}

template <int D, int H>
int launch_fwd(bool item, const TowerFwdArgs& a, int grid, hipStream_t st) {
  if (item) hipLaunchKernelGGL((tower_fwd_kernel<D, H, true>), dim3(grid), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((tower_fwd_kernel<D, H, false>), dim3(grid), dim3(256), 0, st, a);
  return 0;
}
template <int D, int H>
int launch_bwd(bool item, const TowerBwdArgs& a, int grid, hipStream_t st) {
  if (item) hipLaunchKernelGGL((tower_bwd_kernel<D, H, true>), dim3(grid), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((tower_bwd_kernel<D, H, false>), dim3(grid), dim3(256), 0, st, a);
  return 0;
}

template <int D, int H>
int launch_fwd_pair(const TowerFwdArgs& au, const TowerFwdArgs& ai, int grid_u, int grid_i, hipStream_t st) {
  hipLaunchKernelGGL((tower_fwd_pair_kernel<D, H>), dim3(grid_u + grid_i), dim3(256), 0, st, au, ai, grid_u);
  return 0;
}
template <int D, int H>
int launch_bwd_pair(const TowerBwdArgs& au, const TowerBwdArgs& ai, int grid_u, int grid_i, hipStream_t st) {
  hipLaunchKernelGGL((tower_bwd_pair_kernel<D, H>), dim3(grid_u + grid_i), dim3(256), 0, st, au, ai, grid_u);
  return 0;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int rihip_tower_supported(int d, int hidden) {
  return (d == 32 && hidden == 64) || (d == 64 && hidden == 128) || (d == 128 && hidden == 128) ||
         (d == 64 && hidden == 64) || (d == 32 && hidden == 128);
}

// tower3.hip: data + weight gradients of a tile in one kernel (d = hidden = 128); returns the slabs written (0: not covered)
int rihip_launch_tower_bwd3(int d, int hidden, bool item, const TowerBwdArgs& a, int max_slabs, hipStream_t st);
// tower_generic.hip: runtime-shape kernels behind the tuned instantiations
bool rihip_tower_generic_ok(int d, int hidden);
void rihip_launch_tower_fwd_generic(int d, int hidden, bool item, const TowerFwdArgs& a, hipStream_t st);
int rihip_launch_tower_bwd_generic(int d, int hidden, bool item, const TowerBwdArgs& a, float* act, int max_slabs,
                                   hipStream_t st, hipEvent_t dx_event);

extern "C" int rihip_tower_shape_ok(int d, int hidden) {
  return (rihip_tower_supported(d, hidden) || rihip_tower_generic_ok(d, hidden)) ? 1 : 0;
}

#define DISPATCH_DH(FN, ...)                                         \
  if (d == 32 && hidden == 64) FN<32, 64>(__VA_ARGS__);              \
  else if (d == 64 && hidden == 128) FN<64, 128>(__VA_ARGS__);       \
  else if (d == 128 && hidden == 128) FN<128, 128>(__VA_ARGS__);     \
  else if (d == 64 && hidden == 64) FN<64, 64>(__VA_ARGS__);         \
  else if (d == 32 && hidden == 128) FN<32, 128>(__VA_ARGS__);

extern "C" int rihip_tower_forward(const float* table, int64_t n_rows, const int64_t* ids, const float* genres,
                                   int64_t B, int d, int hidden, const float* W1, const float* b1, const float* W2,
                                   const float* b2, int training, float dropout_p, uint64_t seed, int64_t row0,
                                   float* out, float* hid, float* denom, int* err_flag, float* workspace,
                                   const int64_t* seed_step_dev, void* stream) {
  RIHIP_REQUIRE(rihip_tower_shape_ok(d, hidden), RIHIP_ERR_SHAPE,
                "tower_forward: unsupported (embed_dim=%d, hidden_dim=%d): both must be multiples of 16 up to 256", d, hidden);
  RIHIP_REQUIRE(B >= 0 && n_rows > 0, RIHIP_ERR_ARG, "tower_forward: bad sizes B=%lld n_rows=%lld", (long long)B,
                (long long)n_rows);
  if (B == 0) return RIHIP_OK;  // empty batch: nothing to read or write
  RIHIP_REQUIRE(table && ids && W1 && b1 && W2 && b2 && out, RIHIP_ERR_ARG, "tower_forward: null pointer");
  RIHIP_REQUIRE(aligned16(table) && aligned16(out) && (!hid || aligned16(hid)), RIHIP_ERR_ARG,
                "tower_forward: table/out/hid must be 16-byte aligned");
  RIHIP_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, RIHIP_ERR_ARG, "tower_forward: dropout_p=%f", dropout_p);
  if (B == 0) return RIHIP_OK;
  TowerFwdArgs a;
  a.table = table; a.n_rows = n_rows; a.ids = ids; a.genres = genres; a.B = B;
  a.W1 = W1; a.b1 = b1; a.W2 = W2; a.b2 = b2; a.out = out; a.hid = hid; a.denom = denom;
  a.training = (training && dropout_p > 0.f) ? 1 : 0;
  a.seed_mul = rihip_seed_mul(seed); a.thresh24 = rihip_thresh24(dropout_p);
  a.scale = 1.f / (1.f - dropout_p); a.row0 = row0; a.err_flag = err_flag; a.seed_step = seed_step_dev;
  const int64_t ntiles = (B + TM - 1) / TM;
  const int wgs_per_cu = 2;  // the forward kernel is built for 2 workgroups per CU (launch bounds)
  const int grid = (int)(ntiles < wgs_per_cu * RIHIP_NCU ? ntiles : wgs_per_cu * RIHIP_NCU);
  const bool item = genres != nullptr;
  hipStream_t st = (hipStream_t)stream;
  a.W1p = nullptr; a.W2p = nullptr;
  if (!rihip_tower_supported(d, hidden)) {   // no tuned instantiation for this pair: the runtime-shape kernel
    rihip_launch_tower_fwd_generic(d, hidden, item, a, st);
    RIHIP_CHECK_LAUNCH();
    return RIHIP_OK;
  }
  {  // wave-per-32-rows kernel (tower2.hip): weights in LDS, activations in registers, no barrier in the row loop
    const char* ev = getenv("RIHIP_TOWER_FWD");  // 1: 64-row-tile kernel, 3: wave-per-32-rows kernel at any size (tests)
    const int which = ev ? atoi(ev) : 2;
    // it needs >= 8 row tiles of 32 per CU to fill the chip (one 147 KB weight fill per workgroup): below ~48k rows
    // the 64-row-tile kernel with its 2 small workgroups per CU is faster
    if (((which == 2 && B >= 49152) || which == 3) && aligned16(W2) && (reinterpret_cast<uintptr_t>(W1) & 7) == 0 &&
        rihip_launch_tower_fwd2(d, hidden, item, a, st)) {
      RIHIP_CHECK_LAUNCH();
      return RIHIP_OK;
    }
  }
  // re-pack the (just updated) weights fragment-major: one tiny launch, coalesced loads in the kernel -- only worth
  // it when many workgroups load them; a small batch is bounded by dependent kernel boundaries instead
  if (workspace && ntiles > 64) {
    const int K1 = d + (item ? 18 : 0), KB1 = (K1 + 7) / 8, KB2 = hidden / 8;
    RIHIP_REQUIRE(aligned16(workspace), RIHIP_ERR_ARG, "tower_forward: workspace must be 16-byte aligned");
    f32x4* w1p = reinterpret_cast<f32x4*>(workspace);
    f32x4* w2p = w1p + (size_t)(hidden / 32) * KB1 * 64;
    const int n1 = (hidden / 32) * KB1 * 64, n2 = (d / 32) * KB2 * 64;
    const int nb1 = (n1 + 255) / 256, nb2 = (n2 + 255) / 256;
    hipLaunchKernelGGL(pack_frag_rows_kernel, dim3(nb1 + nb2), dim3(256), 0, st, W1, hidden, K1, KB1, w1p, nb1, W2, d,
                       hidden, KB2, w2p);
    a.W1p = w1p; a.W2p = w2p;
  }
  DISPATCH_DH(launch_fwd, item, a, grid, st)
  RIHIP_CHECK_LAUNCH();
  return RIHIP_OK;
}

extern "C" int64_t rihip_tower_forward_workspace_floats(int d, int hidden, int item) {
  const int K1 = d + (item ? 18 : 0), KB1 = (K1 + 7) / 8;
  return (int64_t)hidden * KB1 * 8 + (int64_t)d * hidden;
}

extern "C" int64_t rihip_tower_backward_workspace_floats(int64_t B, int d, int hidden, int item) {
  const int K1 = d + (item ? 18 : 0);
  const int64_t P = (int64_t)hidden * K1 + hidden + (int64_t)d * hidden + d;
  const int64_t nt32 = (B + 31) / 32;
  const int64_t grid = nt32 < RIHIP_NCU ? nt32 : RIHIP_NCU;   // most slabs any backward variant writes
  // slabs + level-1 partials + (two-kernel backward) gy [B,d] and dPre [B,hidden]
  return ((grid > 0 ? grid : 1) + SLAB_GROUPS) * P + B * (int64_t)(d + hidden);
}

namespace {
// the two levels of slab_reduce1/2_kernel for up to two towers per launch (blockIdx.z picks the tower)
__global__ void slab_reduce1_multi_kernel(SlabArgs a) {
  const SlabDesc& t = a.t[blockIdx.z];
  const int G = t.nslab < SLAB_GROUPS ? t.nslab : SLAB_GROUPS;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int g = blockIdx.y;
  if (i >= t.P || g >= G || t.nslab <= SLAB_GROUPS) return;
  float s = 0.f;
  for (int k = g; k < t.nslab; k += G) s += t.slab[(size_t)k * t.P + i];
  t.part[(size_t)g * t.P + i] = s;
}
__global__ void slab_reduce2_multi_kernel(SlabArgs a) {
  const SlabDesc& t = a.t[blockIdx.z];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= t.P) return;
  const int G = t.nslab < SLAB_GROUPS ? t.nslab : SLAB_GROUPS;
  const float* part = t.nslab > SLAB_GROUPS ? t.part : t.slab;   // <= SLAB_GROUPS slabs: level 1 would be a copy
  float s = 0.f;
  for (int g = 0; g < G; ++g) s += part[(size_t)g * t.P + i];
  const int H = a.H, D = a.D, K1 = t.K1;
  float* dst;
  int off;
  if (i < H * K1) { dst = t.dW1; off = i; }
  else if (i < H * K1 + H) { dst = t.db1; off = i - H * K1; }
  else if (i < H * K1 + H + D * H) { dst = t.dW2; off = i - H * K1 - H; }
  else { dst = t.db2; off = i - H * K1 - H - D * H; }
  dst[off] = a.accumulate ? dst[off] + s : s;
}
inline float* slab_part_of(float* workspace, int64_t B, int P) {
  const int64_t nt32 = (B + 31) / 32;
  const int grid_ws = (int)(nt32 < RIHIP_NCU ? nt32 : RIHIP_NCU);   // slab slots in the workspace layout
  return workspace + (size_t)(grid_ws > 0 ? grid_ws : 1) * P;
}
}  // namespace

extern "C" int rihip_tower_backward_partial(const float* table, int64_t n_rows, const int64_t* ids, const float* genres,
                                            int64_t B, int d, int hidden, const float* W1, const float* W2,
                                            const float* grad_out, const float* out, const float* denom,
                                            const float* hid, float dropout_scale, float* dX, float* workspace,
                                            void* stream, void* dx_event, int* n_slabs) {
  RIHIP_REQUIRE(rihip_tower_shape_ok(d, hidden), RIHIP_ERR_SHAPE,
                "tower_backward: unsupported (embed_dim=%d, hidden_dim=%d): both must be multiples of 16 up to 256", d, hidden);
  RIHIP_REQUIRE(n_slabs, RIHIP_ERR_ARG, "tower_backward: null pointer");
  *n_slabs = 0;
  if (B <= 0) return RIHIP_OK;
  RIHIP_REQUIRE(table && ids && W1 && W2 && grad_out && out && denom && hid && dX && workspace, RIHIP_ERR_ARG,
                "tower_backward: null pointer");
  RIHIP_REQUIRE(aligned16(table) && aligned16(grad_out) && aligned16(out) && aligned16(hid), RIHIP_ERR_ARG,
                "tower_backward: table/grad_out/out/hid must be 16-byte aligned");
  TowerBwdArgs a;
  a.table = table; a.n_rows = n_rows; a.ids = ids; a.genres = genres; a.B = B; a.W1 = W1; a.W2 = W2;
  a.gout = grad_out; a.out = out; a.denom = denom; a.hid = hid; a.scale = dropout_scale; a.dX = dX; a.slab = workspace;
  const int64_t ntiles = (B + TM - 1) / TM;
  const int grid = (int)(ntiles < RIHIP_NCU ? ntiles : RIHIP_NCU);
  const bool item = genres != nullptr;
  hipStream_t st = (hipStream_t)stream;
  const int K1 = d + (item ? 18 : 0);
  const int P = hidden * K1 + hidden + d * hidden + d;
  float* part = slab_part_of(workspace, B, P);
  float* act = part + (size_t)SLAB_GROUPS * P;
  int nslab = 0;
  if (!rihip_tower_supported(d, hidden)) {   // runtime-shape kernels (same slab layout, one slab per batch split)
    const int64_t nt32 = (B + 31) / 32;
    nslab = rihip_launch_tower_bwd_generic(d, hidden, item, a, act, (int)(nt32 < RIHIP_NCU ? nt32 : RIHIP_NCU), st,
                                           (hipEvent_t)dx_event);
    RIHIP_CHECK_LAUNCH();
    *n_slabs = nslab;
    return RIHIP_OK;
  }
  {  // chip-filling batches: the two-kernel backward (tower2.hip).  RIHIP_TOWER_BWD = 3: that form at any size (tests),
     // 4: the one-kernel form (tower3.hip: data + weight gradients per tile, no gy / dPre round trip through HBM) at any
     // size -- measured slower on MI355X (457 vs 415 us per 196 608 rows: DESIGN.md §9), kept as a tested variant;
     // 1: the fused 64-row-tile kernel
    const char* ev = getenv("RIHIP_TOWER_BWD");
    const int which = ev ? atoi(ev) : 2;
    const int64_t nt32 = (B + 31) / 32;
    if (which == 4 && aligned16(W2) && aligned16(dX)) {
      nslab = rihip_launch_tower_bwd3(d, hidden, item, a, (int)(nt32 < RIHIP_NCU ? nt32 : RIHIP_NCU), st);
      if (nslab > 0 && dx_event) (void)hipEventRecord((hipEvent_t)dx_event, st);
    }
    if (nslab == 0 && ((which == 2 && B >= 49152) || which == 3) && aligned16(W2) && aligned16(dX))
      nslab = rihip_launch_tower_bwd2(d, hidden, item, a, act, st, (hipEvent_t)dx_event);
  }
  if (nslab == 0) {
    DISPATCH_DH(launch_bwd, item, a, grid, st)
    nslab = grid;
    if (dx_event) (void)hipEventRecord((hipEvent_t)dx_event, st);
  }
  RIHIP_CHECK_LAUNCH();
  *n_slabs = nslab;
  return RIHIP_OK;
}

namespace {
// descriptors of the slab reduction of one or two towers; Pmax / nmax = largest slab size / slab count among them
int slab_args(int d, int hidden, float* ws_a, int64_t B_a, int item_a, int n_slabs_a, float* dW1_a, float* db1_a,
              float* dW2_a, float* db2_a, float* ws_b, int64_t B_b, int item_b, int n_slabs_b, float* dW1_b, float* db1_b,
              float* dW2_b, float* db2_b, int accumulate, SlabArgs* out, int* Pmax, int* nmax) {
  RIHIP_REQUIRE(ws_a && dW1_a && db1_a && dW2_a && db2_a && n_slabs_a > 0 && B_a > 0, RIHIP_ERR_ARG,
                "tower_backward_reduce2: bad arguments (tower a)");
  const bool two = ws_b != nullptr && n_slabs_b > 0;
  RIHIP_REQUIRE(!two || (dW1_b && db1_b && dW2_b && db2_b && B_b > 0 && ws_b != ws_a), RIHIP_ERR_ARG,
                "tower_backward_reduce2: bad arguments (tower b)");
  SlabArgs& a = *out;
  a.H = hidden; a.D = d; a.accumulate = accumulate; a.nt = two ? 2 : 1;
  const int K1a = d + (item_a ? 18 : 0), Pa = hidden * K1a + hidden + d * hidden + d;
  a.t[0] = SlabDesc{ws_a, slab_part_of(ws_a, B_a, Pa), n_slabs_a, Pa, K1a, dW1_a, db1_a, dW2_a, db2_a};
  a.t[1] = a.t[0];
  *Pmax = Pa; *nmax = n_slabs_a;
  if (two) {
    const int K1b = d + (item_b ? 18 : 0), Pb = hidden * K1b + hidden + d * hidden + d;
    a.t[1] = SlabDesc{ws_b, slab_part_of(ws_b, B_b, Pb), n_slabs_b, Pb, K1b, dW1_b, db1_b, dW2_b, db2_b};
    *Pmax = Pb > *Pmax ? Pb : *Pmax;
    *nmax = n_slabs_b > *nmax ? n_slabs_b : *nmax;
  }
  return RIHIP_OK;
}
}  // namespace

extern "C" int rihip_tower_backward_reduce2(int d, int hidden, float* ws_a, int64_t B_a, int item_a, int n_slabs_a,
                                            float* dW1_a, float* db1_a, float* dW2_a, float* db2_a, float* ws_b,
                                            int64_t B_b, int item_b, int n_slabs_b, float* dW1_b, float* db1_b,
                                            float* dW2_b, float* db2_b, int accumulate, void* stream) {
  SlabArgs a;
  int Pmax = 0, nmax = 0;
  const int rc = slab_args(d, hidden, ws_a, B_a, item_a, n_slabs_a, dW1_a, db1_a, dW2_a, db2_a, ws_b, B_b, item_b, n_slabs_b,
                           dW1_b, db1_b, dW2_b, db2_b, accumulate, &a, &Pmax, &nmax);
  if (rc != RIHIP_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (nmax > SLAB_GROUPS)
    hipLaunchKernelGGL(slab_reduce1_multi_kernel, dim3((Pmax + 255) / 256, SLAB_GROUPS, a.nt), dim3(256), 0, st, a);
  hipLaunchKernelGGL(slab_reduce2_multi_kernel, dim3((Pmax + 255) / 256, 1, a.nt), dim3(256), 0, st, a);
  RIHIP_CHECK_LAUNCH();
  return RIHIP_OK;
}

extern "C" int rihip_tower_backward_ev(const float* table, int64_t n_rows, const int64_t* ids, const float* genres,
                                       int64_t B, int d, int hidden, const float* W1, const float* W2,
                                       const float* grad_out, const float* out, const float* denom, const float* hid,
                                       float dropout_scale, float* dX, float* dW1, float* db1, float* dW2, float* db2,
                                       int accumulate, float* workspace, void* stream, void* dx_event) {
  RIHIP_REQUIRE(B <= 0 || (dW1 && db1 && dW2 && db2), RIHIP_ERR_ARG, "tower_backward: null pointer");
  int nslab = 0;
  const int rc = rihip_tower_backward_partial(table, n_rows, ids, genres, B, d, hidden, W1, W2, grad_out, out, denom, hid,
                                              dropout_scale, dX, workspace, stream, dx_event, &nslab);
  if (rc != RIHIP_OK || nslab == 0) return rc;
  return rihip_tower_backward_reduce2(d, hidden, workspace, B, genres != nullptr, nslab, dW1, db1, dW2, db2, nullptr, 0, 0,
                                      0, nullptr, nullptr, nullptr, nullptr, accumulate, stream);
}

extern "C" int rihip_tower_backward(const float* table, int64_t n_rows, const int64_t* ids, const float* genres,
                                    int64_t B, int d, int hidden, const float* W1, const float* W2,
                                    const float* grad_out, const float* out, const float* denom, const float* hid,
                                    float dropout_scale, float* dX, float* dW1, float* db1, float* dW2, float* db2,
                                    int accumulate, float* workspace, void* stream) {
  return rihip_tower_backward_ev(table, n_rows, ids, genres, B, d, hidden, W1, W2, grad_out, out, denom, hid, dropout_scale,
                                 dX, dW1, db1, dW2, db2, accumulate, workspace, stream, nullptr);
}

// ---- both towers of one step in one launch (small batches: a step of B = 256 is ~10 dependent launches of ~10 us) ----
extern "C" int rihip_tower_forward_pair(const rihip_tower_io* user, const rihip_tower_io* item, int d, int hidden,
                                        int training, float dropout_p, int* err_flag, const int64_t* seed_step_dev,
                                        void* stream) {
  RIHIP_REQUIRE(user && item, RIHIP_ERR_ARG, "tower_forward_pair: null pointer");
  RIHIP_REQUIRE(rihip_tower_shape_ok(d, hidden), RIHIP_ERR_SHAPE,
                "tower_forward: unsupported (embed_dim=%d, hidden_dim=%d)", d, hidden);
  const rihip_tower_io* io[2] = {user, item};
  const char* ev = getenv("RIHIP_TOWER_FWD");
  bool pair = rihip_tower_supported(d, hidden) && user->B > 0 && item->B > 0 && user->B < 49152 && item->B < 49152 && !(ev && atoi(ev) == 3) &&
              user->genres == nullptr && item->genres != nullptr && dropout_p >= 0.f && dropout_p < 1.f;
  for (int t = 0; t < 2 && pair; ++t)
    pair = io[t]->table && io[t]->ids && io[t]->W1 && io[t]->b1 && io[t]->W2 && io[t]->b2 && io[t]->out &&
           io[t]->n_rows > 0 && aligned16(io[t]->table) && aligned16(io[t]->out) && (!io[t]->hid || aligned16(io[t]->hid)) &&
           (!io[t]->fwd_workspace || aligned16(io[t]->fwd_workspace));
  if (!pair) {   // anything unusual (or a chip-filling batch): the two single calls, with their own checks
    for (int t = 0; t < 2; ++t) {
      const int rc = rihip_tower_forward(io[t]->table, io[t]->n_rows, io[t]->ids, io[t]->genres, io[t]->B, d, hidden,
                                         io[t]->W1, io[t]->b1, io[t]->W2, io[t]->b2, training, dropout_p, io[t]->seed,
                                         io[t]->row0, io[t]->out, io[t]->hid, io[t]->denom, err_flag, io[t]->fwd_workspace,
                                         seed_step_dev, stream);
      if (rc != RIHIP_OK) return rc;
    }
    return RIHIP_OK;
  }
  hipStream_t st = (hipStream_t)stream;
  TowerFwdArgs a[2];
  int grid[2];
  PackArgs pk;
  pk.n = 0;
  for (int t = 0; t < 2; ++t) {
    const rihip_tower_io& o = *io[t];
    a[t].table = o.table; a[t].n_rows = o.n_rows; a[t].ids = o.ids; a[t].genres = o.genres; a[t].B = o.B;
    a[t].W1 = o.W1; a[t].b1 = o.b1; a[t].W2 = o.W2; a[t].b2 = o.b2; a[t].out = o.out; a[t].hid = o.hid; a[t].denom = o.denom;
    a[t].training = (training && dropout_p > 0.f) ? 1 : 0;
    a[t].seed_mul = rihip_seed_mul(o.seed); a[t].thresh24 = rihip_thresh24(dropout_p);
    a[t].scale = 1.f / (1.f - dropout_p); a[t].row0 = o.row0; a[t].err_flag = err_flag; a[t].seed_step = seed_step_dev;
    a[t].W1p = nullptr; a[t].W2p = nullptr;
    const int64_t ntiles = (o.B + TM - 1) / TM;
    grid[t] = (int)(ntiles < 2 * RIHIP_NCU ? ntiles : 2 * RIHIP_NCU);
    if (o.fwd_workspace && ntiles > 64) {   // same rule as rihip_tower_forward: fragment-major weights for many workgroups
      const int K1 = d + (t ? 18 : 0), KB1 = (K1 + 7) / 8, KB2 = hidden / 8;
      f32x4* w1p = reinterpret_cast<f32x4*>(o.fwd_workspace);
      f32x4* w2p = w1p + (size_t)(hidden / 32) * KB1 * 64;
      const int n1 = (hidden / 32) * KB1 * 64, n2 = (d / 32) * KB2 * 64;
      pk.m[pk.n++] = PackDesc{o.W1, hidden, K1, KB1, w1p, (n1 + 255) / 256};
      pk.m[pk.n++] = PackDesc{o.W2, d, hidden, KB2, w2p, (n2 + 255) / 256};
      a[t].W1p = w1p; a[t].W2p = w2p;
    }
  }
  if (pk.n > 0) {
    int nb = 0;
    for (int i = 0; i < pk.n; ++i) nb += pk.m[i].nb;
    for (int i = pk.n; i < 4; ++i) pk.m[i] = pk.m[0];
    hipLaunchKernelGGL(pack_frag_rows_multi_kernel, dim3(nb), dim3(256), 0, st, pk);
  }
  DISPATCH_DH(launch_fwd_pair, a[0], a[1], grid[0], grid[1], st)
  RIHIP_CHECK_LAUNCH();
  return RIHIP_OK;
}

extern "C" int rihip_tower_backward_partial_pair(const rihip_tower_io* user, const rihip_tower_io* item, int d, int hidden,
                                                 float dropout_scale, void* stream, void* dx_event_user,
                                                 void* dx_event_item, int* n_slabs_user, int* n_slabs_item) {
  RIHIP_REQUIRE(user && item && n_slabs_user && n_slabs_item, RIHIP_ERR_ARG, "tower_backward_partial_pair: null pointer");
  RIHIP_REQUIRE(rihip_tower_shape_ok(d, hidden), RIHIP_ERR_SHAPE,
                "tower_backward: unsupported (embed_dim=%d, hidden_dim=%d)", d, hidden);
  const rihip_tower_io* io[2] = {user, item};
  int* ns[2] = {n_slabs_user, n_slabs_item};
  void* evs[2] = {dx_event_user, dx_event_item};
  const char* ev = getenv("RIHIP_TOWER_BWD");
  bool pair = rihip_tower_supported(d, hidden) && user->B > 0 && item->B > 0 && user->B < 49152 && item->B < 49152 && !(ev && atoi(ev) == 3) &&
              user->genres == nullptr && item->genres != nullptr;
  for (int t = 0; t < 2 && pair; ++t)
    pair = io[t]->table && io[t]->ids && io[t]->W1 && io[t]->W2 && io[t]->grad_out && io[t]->out && io[t]->denom &&
           io[t]->hid && io[t]->dX && io[t]->bwd_workspace && aligned16(io[t]->table) && aligned16(io[t]->grad_out) &&
           aligned16(io[t]->out) && aligned16(io[t]->hid);
  pair = pair && user->bwd_workspace != item->bwd_workspace;
  if (!pair) {
    for (int t = 0; t < 2; ++t) {
      const int rc = rihip_tower_backward_partial(io[t]->table, io[t]->n_rows, io[t]->ids, io[t]->genres, io[t]->B, d, hidden,
                                                  io[t]->W1, io[t]->W2, io[t]->grad_out, io[t]->out, io[t]->denom, io[t]->hid,
                                                  dropout_scale, io[t]->dX, io[t]->bwd_workspace, stream, evs[t], ns[t]);
      if (rc != RIHIP_OK) return rc;
    }
    return RIHIP_OK;
  }
  hipStream_t st = (hipStream_t)stream;
  TowerBwdArgs a[2];
  int grid[2];
  for (int t = 0; t < 2; ++t) {
    const rihip_tower_io& o = *io[t];
    a[t].table = o.table; a[t].n_rows = o.n_rows; a[t].ids = o.ids; a[t].genres = o.genres; a[t].B = o.B; a[t].W1 = o.W1;
    a[t].W2 = o.W2; a[t].gout = o.grad_out; a[t].out = o.out; a[t].denom = o.denom; a[t].hid = o.hid;
    a[t].scale = dropout_scale; a[t].dX = o.dX; a[t].slab = o.bwd_workspace;
    const int64_t ntiles = (o.B + TM - 1) / TM;
    grid[t] = (int)(ntiles < RIHIP_NCU ? ntiles : RIHIP_NCU);
    *ns[t] = grid[t];
  }
  DISPATCH_DH(launch_bwd_pair, a[0], a[1], grid[0], grid[1], st)
  RIHIP_CHECK_LAUNCH();
  for (int t = 0; t < 2; ++t)
    if (evs[t]) (void)hipEventRecord((hipEvent_t)evs[t], st);
  return RIHIP_OK;
}

namespace {
int scatter_launch(ScatterArgs a, int n_tables, hipStream_t st, const SlabArgs* slab = nullptr, int slab_level = 0,
                   int slab_blocks = 0) {
  int64_t maxB = 0;
  int nwg = 4;
  bool al = a.D == 32 || a.D == 64 || a.D == 128;
  for (int t = 0; t < n_tables; ++t) {
    maxB = a.t[t].B > maxB ? a.t[t].B : maxB;
    al = al && aligned16(a.t[t].grad) && aligned16(a.t[t].dX);
  }
  // workgroups per table: ~128 batch positions each, and at most 2^18 rows each (the key holds 18 row bits)
  int64_t want = maxB / 128;
  want = want < 4 ? 4 : (want > 64 ? 64 : want);
  for (int t = 0; t < n_tables; ++t) {
    const int64_t need = ((a.t[t].n_rows >> SCAT_RBITS) + 1);
    want = need > want ? need : want;
  }
  nwg = (int)want;
  a.nwg = nwg;
  int np2 = 64;
  const int64_t cap = maxB < SCAT_CHUNK ? maxB : SCAT_CHUNK;
  while (np2 < cap) np2 <<= 1;
  const size_t lds = (size_t)np2 * 4;
  SlabArgs sl = SlabArgs();
  if (slab) sl = *slab;
  const dim3 grid((unsigned)(nwg + (slab ? slab_blocks : 0)), (unsigned)n_tables);
  static bool granted = false;
  if (!granted) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(scatter_range_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, SCAT_CHUNK * 4);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(scatter_range_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, SCAT_CHUNK * 4);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(scatter_range_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, SCAT_CHUNK * 4);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(scatter_range_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, SCAT_CHUNK * 4);
    granted = true;
  }
  const int lvl = slab ? slab_level : 0;
  if (al && a.D == 128) hipLaunchKernelGGL((scatter_range_kernel<32>), grid, dim3(1024), lds, st, a, sl, lvl);
  else if (al && a.D == 64) hipLaunchKernelGGL((scatter_range_kernel<16>), grid, dim3(1024), lds, st, a, sl, lvl);
  else if (al && a.D == 32) hipLaunchKernelGGL((scatter_range_kernel<8>), grid, dim3(1024), lds, st, a, sl, lvl);
  else hipLaunchKernelGGL((scatter_range_kernel<0>), grid, dim3(1024), lds, st, a, sl, lvl);
  RIHIP_CHECK_LAUNCH();
  return RIHIP_OK;
}
}  // namespace

extern "C" int rihip_embedding_scatter_add(float* grad_table, int64_t n_rows, const int64_t* ids, const float* dX,
                                           int64_t B, int d, void* stream) {
  RIHIP_REQUIRE(grad_table && ids && dX && d > 0, RIHIP_ERR_ARG, "embedding_scatter_add: null pointer");
    if (B <= 0) return RIHIP_OK;
  ScatterArgs a;
  a.D = d;
  a.t[0] = ScatterDesc{grad_table, n_rows, ids, dX, B};
  a.t[1] = a.t[0];
  return scatter_launch(a, 1, (hipStream_t)stream);
}

extern "C" int rihip_embedding_scatter_add2(float* grad_a, int64_t n_rows_a, const int64_t* ids_a, const float* dX_a,
                                            int64_t B_a, float* grad_b, int64_t n_rows_b, const int64_t* ids_b,
                                            const float* dX_b, int64_t B_b, int d, void* stream) {
  RIHIP_REQUIRE(grad_a && ids_a && dX_a && grad_b && ids_b && dX_b && d > 0, RIHIP_ERR_ARG,
                "embedding_scatter_add2: null pointer");
  RIHIP_REQUIRE(grad_a != grad_b, RIHIP_ERR_ARG, "embedding_scatter_add2: the two tables must differ");
  if (B_a <= 0 && B_b <= 0) return RIHIP_OK;
  ScatterArgs a;
  a.D = d;
  a.t[0] = ScatterDesc{grad_a, n_rows_a, ids_a, dX_a, B_a > 0 ? B_a : 0};
  a.t[1] = ScatterDesc{grad_b, n_rows_b, ids_b, dX_b, B_b > 0 ? B_b : 0};
  return scatter_launch(a, 2, (hipStream_t)stream);
}

// weight-gradient slab reduction of both towers AND the dense embedding scatter-add of both tables: both wait only for the
// tower backward, so the scatter launch carries one reduction level in extra workgroups (the first level when there are
// more than 16 slabs, followed by the second as its own launch; else the only level)
extern "C" int rihip_backward_reduce2_scatter2(int d, int hidden, float* ws_a, int64_t B_a, int item_a, int n_slabs_a,
                                               float* dW1_a, float* db1_a, float* dW2_a, float* db2_a, float* ws_b,
                                               int64_t B_b, int item_b, int n_slabs_b, float* dW1_b, float* db1_b,
                                               float* dW2_b, float* db2_b, int accumulate, float* grad_a,
                                               int64_t n_rows_a, const int64_t* ids_a, const float* dX_a, int64_t Bs_a,
                                               float* grad_b, int64_t n_rows_b, const int64_t* ids_b, const float* dX_b,
                                               int64_t Bs_b, void* stream) {
  RIHIP_REQUIRE(grad_a && ids_a && dX_a && grad_b && ids_b && dX_b && d > 0 && grad_a != grad_b && Bs_a > 0 && Bs_b > 0,
                RIHIP_ERR_ARG, "backward_reduce2_scatter2: bad scatter arguments");
  SlabArgs sl;
  int Pmax = 0, nmax = 0;
  const int rc = slab_args(d, hidden, ws_a, B_a, item_a, n_slabs_a, dW1_a, db1_a, dW2_a, db2_a, ws_b, B_b, item_b, n_slabs_b,
                           dW1_b, db1_b, dW2_b, db2_b, accumulate, &sl, &Pmax, &nmax);
  if (rc != RIHIP_OK) return rc;
  ScatterArgs a;
  a.D = d;
  a.t[0] = ScatterDesc{grad_a, n_rows_a, ids_a, dX_a, Bs_a};
  a.t[1] = ScatterDesc{grad_b, n_rows_b, ids_b, dX_b, Bs_b};
  hipStream_t st = (hipStream_t)stream;
  const bool two_levels = nmax > SLAB_GROUPS;
  const int64_t elems = two_levels ? (int64_t)Pmax * SLAB_GROUPS : Pmax;
  const int rc2 = scatter_launch(a, 2, st, &sl, two_levels ? 1 : 2, (int)((elems + 1023) / 1024));
  if (rc2 != RIHIP_OK) return rc2;
  if (two_levels) hipLaunchKernelGGL(slab_reduce2_multi_kernel, dim3((Pmax + 255) / 256, 1, sl.nt), dim3(256), 0, st, sl);
  RIHIP_CHECK_LAUNCH();
  return RIHIP_OK;
}
