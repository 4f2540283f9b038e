// Device-function bodies of the runtime-shape tower kernels, shared by tower_generic.hip (one launch per stage) and
// step_persistent.hip (the whole small-batch training step in one cooperative launch).
#pragma once
#include "common.h"
#include "gen_gemm.h"
#include "tower_args.h"

#ifndef RIHIP_TILE_STAMP
#define RIHIP_TILE_STAMP(k)      // (tools/persist_phases.py probe builds stamp the wall clock here)
#endif

namespace rihip_gen {

struct GenFwd {
  TowerFwdArgs a;
  int D, H, K1;
};

// forward of one 32-row tile by the whole workgroup (256 threads); smem: fwd_lds_floats(D, H, K1) floats; step_seed =
// the value of *a.seed_step read by the caller (0 when a.seed_step is null)
__device__ __forceinline__ void gen_fwd_tile(const GenFwd& g, float* smem, int64_t tile, uint64_t seed_mul) {
  const TowerFwdArgs& a = g.a;
  const int D = g.D, H = g.H, K1 = g.K1;
  const int K1p = up8(K1), ldx = K1p + 4, ldh = H + 4, ldy = D + 4;
  float* Xs = smem;                                   // [32][ldx] (later Y: [32][ldy], ldy <= ldx)
  float* Hs = Xs + GTM * ldx;                         // [32][ldh]
  float* Wp = Hs + GTM * ldh;                         // [256][GLDP]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  {
    const int64_t row_base = tile * GTM;
    __syncthreads();   // the previous tile's Y reads are done
    RIHIP_TILE_STAMP(0);
    // ---- gather: x = table[id] (|| genres), zero-padded to K1p; rows past B are zero.  8 threads per row; every load
    // is unconditional (clamped address) and nothing consumes a value before all loads of the thread are issued: the
    // tile costs two dependent memory latencies (id, row), not one per element
    {
      const int r = tid >> 3, q = tid & 7;
      const int64_t grow = row_base + r;
      const bool ok = grow < a.B;
      const int64_t gr = ok ? grow : a.B - 1;
      int64_t id = a.ids[gr];
      if (id < 0 || id >= a.n_rows) {
        if (ok && a.err_flag) *a.err_flag = 1;
        id = 0;
      }
      const f32x4* trow = reinterpret_cast<const f32x4*>(a.table + id * D);
      const int d4 = D >> 2;
      f32x4 xv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { const int c4 = q + 8 * j; xv[j] = trow[c4 < d4 ? c4 : 0]; }
      float gv[3] = {0.f, 0.f, 0.f};
      if (K1 > D) {
#pragma unroll
        for (int j = 0; j < 3; ++j) { const int gk = q + 8 * j; gv[j] = a.genres[gr * 18 + (gk < 18 ? gk : 0)]; }
      }
      const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c4 = q + 8 * j;
        if (c4 < d4) *reinterpret_cast<f32x4*>(&Xs[r * ldx + 4 * c4]) = ok ? xv[j] : z4;
      }
      if (K1 > D) {
#pragma unroll
        for (int j = 0; j < 3; ++j) { const int gk = q + 8 * j; if (gk < 18) Xs[r * ldx + D + gk] = ok ? gv[j] : 0.f; }
      }
      if (K1 + q < K1p) Xs[r * ldx + K1 + q] = 0.f;
    }
    // ---- Linear 1 + ReLU + dropout -> Hs (+ saved hidden)
    f32x16 acc[GNT];
    RIHIP_TILE_STAMP(1);
    wg_gemm<false>(Xs, ldx, K1, a.W1, K1, H, Wp, acc, tid);
    RIHIP_TILE_STAMP(2);
#pragma unroll
    for (int t = 0; t < GNT; ++t) {
      const int col = (w + 4 * t) * 32 + (lane & 31);
      if (col < H) {
        const float b1v = a.b1[col];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = acc_row(r, lane);
          const int64_t grow = row_base + row;
          float v = fmaxf(acc[t][r] + b1v, 0.f);
          if (a.training) v = rihip_keep(seed_mul, (uint64_t)(a.row0 + grow) * H + col, a.thresh24) ? v * a.scale : 0.f;
          Hs[row * ldh + col] = v;
          if (a.hid && grow < a.B) a.hid[grow * H + col] = v;
        }
      }
    }
    // ---- Linear 2 -> Y (aliases Xs: the first barrier inside wg_gemm orders it after every Xs read of GEMM 1)
    RIHIP_TILE_STAMP(3);
    wg_gemm<false>(Hs, ldh, H, a.W2, H, D, Wp, acc, tid);
    RIHIP_TILE_STAMP(4);
    float* Ys = Xs;
#pragma unroll
    for (int t = 0; t < GNT; ++t) {
      const int col = (w + 4 * t) * 32 + (lane & 31);
      if (col < D) {
        const float b2v = a.b2[col];
#pragma unroll
        for (int r = 0; r < 16; ++r) Ys[acc_row(r, lane) * ldy + col] = acc[t][r] + b2v;
      }
    }
    __syncthreads();
    RIHIP_TILE_STAMP(5);
    // ---- row L2-normalise: 8 threads per row
    {
      const int row = tid >> 3, q = tid & 7;
      float ss = 0.f;
      for (int c = q; c < D; c += 8) { const float y = Ys[row * ldy + c]; ss += y * y; }
      ss += __shfl_xor(ss, 1, 64);
      ss += __shfl_xor(ss, 2, 64);
      ss += __shfl_xor(ss, 4, 64);
      const float dn = fmaxf(sqrtf(ss), 1e-12f);
      const int64_t grow = row_base + row;
      if (grow < a.B) {
        for (int c = q; c < D; c += 8) a.out[grow * D + c] = Ys[row * ldy + c] / dn;
        if (q == 0 && a.denom) a.denom[grow] = dn;
      }
    }
  }
}

struct GenBwd {
  TowerBwdArgs a;
  float* gy;     // [B,D]
  float* dpre;   // [B,H]
  int D, H, K1;
  int rows_per_slab, nslab, P;
};

// (a) data gradients of one 32-row tile: gy = (g - out (out.g)) / den ; dh = gy.W2 ; dPre = dh * [hid > 0] * scale ;
//     dX = dPre.W1[:, :D].  Whole workgroup; smem: bwd_lds_floats(D, H) floats.
__device__ __forceinline__ void gen_bwd_data_tile(const GenBwd& g, float* smem, int64_t tile) {
  const TowerBwdArgs& a = g.a;
  const int D = g.D, H = g.H, K1 = g.K1;
  const int ldg = D + 4, ldh = H + 4;
  float* Gs = smem;                  // [32][ldg]  gy
  float* Ps = Gs + GTM * ldg;        // [32][ldh]  dPre
  float* Wp = Ps + GTM * ldh;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  {
    const int64_t row_base = tile * GTM;
    __syncthreads();
    {   // normalise-backward, 8 threads per row, float4 loads all issued before the first use (see the gather above)
      const int row = tid >> 3, q = tid & 7;
      const int64_t grow = row_base + row;
      const bool ok = grow < a.B;
      const int64_t gr = ok ? grow : a.B - 1;
      const int d4 = D >> 2;
      const f32x4* orow = reinterpret_cast<const f32x4*>(a.out + gr * D);
      const f32x4* grw = reinterpret_cast<const f32x4*>(a.gout + gr * D);
      f32x4 ov[8], gv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c4 = q + 8 * j, cc = c4 < d4 ? c4 : 0;
        ov[j] = orow[cc]; gv[j] = grw[cc];
      }
      const float den = a.denom[gr];
      float dot = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (q + 8 * j < d4) dot += (ov[j].x * gv[j].x + ov[j].y * gv[j].y) + (ov[j].z * gv[j].z + ov[j].w * gv[j].w);
      dot += __shfl_xor(dot, 1, 64);
      dot += __shfl_xor(dot, 2, 64);
      dot += __shfl_xor(dot, 4, 64);
      const float inv = ok ? 1.f / den : 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c4 = q + 8 * j;
        if (c4 < d4) {
          f32x4 v;
          v.x = (gv[j].x - ov[j].x * dot) * inv; v.y = (gv[j].y - ov[j].y * dot) * inv;
          v.z = (gv[j].z - ov[j].z * dot) * inv; v.w = (gv[j].w - ov[j].w * dot) * inv;
          if (ok) reinterpret_cast<f32x4*>(g.gy + grow * D)[c4] = v;
          *reinterpret_cast<f32x4*>(&Gs[row * ldg + 4 * c4]) = v;
        }
      }
    }
    f32x16 acc[GNT];
    // dh[row][h] = sum_d gy[row][d] W2[d][h]   (B[n = h][k = d] = W2[d*H + h]: transposed panel)
    wg_gemm<true>(Gs, ldg, D, a.W2, H, H, Wp, acc, tid);
#pragma unroll
    for (int t = 0; t < GNT; ++t) {
      const int col = (w + 4 * t) * 32 + (lane & 31);
      if (col < H) {
        float hv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {     // the 16 mask loads first, unconditional
          const int64_t grow = row_base + acc_row(r, lane);
          hv[r] = a.hid[(grow < a.B ? grow : a.B - 1) * H + col];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = acc_row(r, lane);
          const int64_t grow = row_base + row;
          const bool ok = grow < a.B;
          const float v = (ok && hv[r] > 0.f) ? acc[t][r] * a.scale : 0.f;
          if (ok) g.dpre[grow * H + col] = v;
          Ps[row * ldh + col] = v;
        }
      }
    }
    // dX[row][k] = sum_h dPre[row][h] W1[h][k], k < D   (B[n = k][kk = h] = W1[h*K1 + k]: transposed panel)
    wg_gemm<true>(Ps, ldh, H, a.W1, K1, D, Wp, acc, tid);
#pragma unroll
    for (int t = 0; t < GNT; ++t) {
      const int col = (w + 4 * t) * 32 + (lane & 31);
      if (col < D) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t grow = row_base + acc_row(r, lane);
          if (grow < a.B) a.dX[grow * D + col] = acc[t][r];
        }
      }
    }
  }
}

// (b) weight gradients: output tiles of  [dW1 | db1]: ceil(H/32) x ceil((K1+1)/32), M = h, N = k1 (column K1 = ones -> db1)
//                                       [dW2 | db2]: ceil(D/32) x ceil((H+1)/32),  M = dcol, N = h (column H = ones -> db2).
// Operand layout of mfma32: lane l supplies A[i = l&31][k] and B[k][j = l&31] with k = the batch row picked by the
// step and l>>5 -- so both operands are plain coalesced reads of row-major [batch, feature] arrays.
__host__ __device__ inline int gen_wgrad_tiles(int D, int H, int K1) {
  return ((H + 31) / 32) * ((K1 + 1 + 31) / 32) + ((D + 31) / 32) * ((H + 1 + 31) / 32);
}
struct GenWTile { bool first; int mt, nt; };
__device__ __forceinline__ GenWTile gen_wgrad_tile_of(const GenBwd& g, int tile) {
  const int nt1 = (g.K1 + 1 + 31) / 32, nt2 = (g.H + 1 + 31) / 32;
  const int n1 = ((g.H + 31) / 32) * nt1;
  GenWTile t;
  t.first = tile < n1;
  const int tt = t.first ? tile : tile - n1;
  t.mt = t.first ? tt / nt1 : tt / nt2;
  t.nt = t.first ? tt % nt1 : tt % nt2;
  return t;
}
// one wave: the partial output tile over the batch rows rb = r_begin, r_begin + r_step, ... < r1 (32 rows each).  Per
// 32-row block all operand loads (ids, then A and B values) are issued unconditionally from clamped addresses before the
// 16 MFMAs consume them: the block costs two dependent latencies, not one per step.
__device__ __forceinline__ f32x16 gen_wgrad_acc(const GenBwd& g, const GenWTile& t, int64_t r_begin, int64_t r_step, int64_t r1,
                                                int lane) {
  const TowerBwdArgs& a = g.a;
  const int D = g.D, H = g.H, K1 = g.K1;
  const bool first = t.first;
  const int i = t.mt * 32 + (lane & 31);          // A feature: h (dW1) or dcol (dW2)
  const int j = t.nt * 32 + (lane & 31);          // B feature: k1 (dW1) or h (dW2)
  const float* Asrc = first ? g.dpre : g.gy;
  const int lda = first ? H : D, Mdim = first ? H : D, Ndim = first ? K1 : H;
  const bool a_ok = i < Mdim;
  const int ic = a_ok ? i : 0;
  // B operand of this lane: 0 = zero column, 1 = the ones column (bias gradient), 2 = a strided array, 3 = table gather
  int mode = 0;
  const float* bbase = Asrc;
  int bstride = 0;
  if (j == Ndim) mode = 1;
  else if (j < Ndim) {
    if (!first) { mode = 2; bbase = a.hid + j; bstride = H; }
    else if (j < D) { mode = 3; }
    else { mode = 2; bbase = a.genres + (j - D); bstride = 18; }
  }
  const bool any_gather = __ballot(mode == 3) != 0ull;
  f32x16 acc = zero16();
  for (int64_t rb = r_begin; rb < r1; rb += r_step) {
    int64_t rowc[16];
    float av[16], bv[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int64_t row = rb + acc_row(s, lane);    // k index of step s for this lane half (any bijection works)
      rowc[s] = row < r1 ? row : r1 - 1;
    }
    int64_t idv[16];
    if (any_gather) {
#pragma unroll
      for (int s = 0; s < 16; ++s) idv[s] = a.ids[rowc[s]];
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) av[s] = Asrc[rowc[s] * lda + ic];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const float* bp = bbase + rowc[s] * bstride;
      if (any_gather && mode == 3) {
        int64_t id = idv[s];
        if (id < 0 || id >= a.n_rows) id = 0;
        bp = a.table + id * D + j;
      }
      bv[s] = *bp;
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const bool rok = rb + acc_row(s, lane) < r1;
      const float x = (rok && a_ok) ? av[s] : 0.f;
      const float y = !rok ? 0.f : (mode == 1 ? 1.f : (mode == 0 ? 0.f : bv[s]));
      acc = mfma32(x, y, acc);
    }
  }
  return acc;
}
// element (m, j) of the tile -> its destination: dW1[m][j] / db1[m] / dW2[m][j] / db2[m]; null if outside
__device__ __forceinline__ float* gen_wgrad_dst(const GenBwd& g, const GenWTile& t, int r, int lane, float* dW1, float* db1,
                                                float* dW2, float* db2) {
  const int m = t.mt * 32 + acc_row(r, lane), j = t.nt * 32 + (lane & 31);
  if (t.first) {
    if (m >= g.H) return nullptr;
    if (j < g.K1) return dW1 + (size_t)m * g.K1 + j;
    return j == g.K1 ? db1 + m : nullptr;
  }
  if (m >= g.D) return nullptr;
  if (j < g.H) return dW2 + (size_t)m * g.H + j;
  return j == g.H ? db2 + m : nullptr;
}

__host__ __device__ inline size_t gen_fwd_lds_floats(int D, int H, int K1) {
  return (size_t)GTM * (up8(K1) + 4) + (size_t)GTM * (H + 4) + 256 * GLDP;
}
__host__ __device__ inline size_t gen_bwd_lds_floats(int D, int H) {
  return (size_t)GTM * (D + 4) + (size_t)GTM * (H + 4) + 256 * GLDP;
}

}  // namespace rihip_gen
