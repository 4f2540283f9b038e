// Device-function bodies of the runtime-shape tower kernels, shared by tower_generic.hip (one launch per stage) and
// step_persistent.hip (the whole small-batch training step in one cooperative launch).
#pragma once
#include "common.h"
#include "gen_gemm.h"
#include "tower_args.h"

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
    // ---- gather: x = table[id] (|| genres), zero-padded to K1p; rows past B are zero
    for (int idx = tid; idx < GTM * K1p; idx += 256) {
      const int r = idx / K1p, k = idx % K1p;
      const int64_t grow = row_base + r;
      float v = 0.f;
      if (grow < a.B) {
        if (k < D) {
          int64_t id = a.ids[grow];
          if (id < 0 || id >= a.n_rows) {
            if (a.err_flag) *a.err_flag = 1;
            id = 0;
          }
          v = a.table[id * D + k];
        } else if (k < K1) {
          v = a.genres[grow * 18 + (k - D)];
        }
      }
      Xs[r * ldx + k] = v;
    }
    // ---- Linear 1 + ReLU + dropout -> Hs (+ saved hidden)
    f32x16 acc[GNT];
    wg_gemm<false>(Xs, ldx, K1, a.W1, K1, H, Wp, acc, tid);
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
    wg_gemm<false>(Hs, ldh, H, a.W2, H, D, Wp, acc, tid);
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
    {   // normalise-backward, 8 threads per row (D is a multiple of 16: no padding columns)
      const int row = tid >> 3, q = tid & 7;
      const int64_t grow = row_base + row;
      const bool ok = grow < a.B;
      float dot = 0.f;
      if (ok)
        for (int c = q; c < D; c += 8) dot += a.out[grow * D + c] * a.gout[grow * D + c];
      dot += __shfl_xor(dot, 1, 64);
      dot += __shfl_xor(dot, 2, 64);
      dot += __shfl_xor(dot, 4, 64);
      const float inv = ok ? 1.f / a.denom[grow] : 0.f;
      for (int c = q; c < D; c += 8) {
        float v = 0.f;
        if (ok) {
          v = (a.gout[grow * D + c] - a.out[grow * D + c] * dot) * inv;
          g.gy[grow * D + c] = v;
        }
        Gs[row * ldg + c] = v;
      }
    }
    f32x16 acc[GNT];
    // dh[row][h] = sum_d gy[row][d] W2[d][h]   (B[n = h][k = d] = W2[d*H + h]: transposed panel)
    wg_gemm<true>(Gs, ldg, D, a.W2, H, H, Wp, acc, tid);
#pragma unroll
    for (int t = 0; t < GNT; ++t) {
      const int col = (w + 4 * t) * 32 + (lane & 31);
      if (col < H) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = acc_row(r, lane);
          const int64_t grow = row_base + row;
          float v = 0.f;
          if (grow < a.B) {
            v = a.hid[grow * H + col] > 0.f ? acc[t][r] * a.scale : 0.f;
            g.dpre[grow * H + col] = v;
          }
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
// one wave: the partial output tile over the batch rows rb = r_begin, r_begin + r_step, ... < r1 (32 rows each)
__device__ __forceinline__ f32x16 gen_wgrad_acc(const GenBwd& g, const GenWTile& t, int64_t r_begin, int64_t r_step, int64_t r1,
                                                int lane) {
  const TowerBwdArgs& a = g.a;
  const int D = g.D, H = g.H, K1 = g.K1;
  const bool first = t.first;
  const int i = t.mt * 32 + (lane & 31);          // A feature: h (dW1) or dcol (dW2)
  const int j = t.nt * 32 + (lane & 31);          // B feature: k1 (dW1) or h (dW2)
  const float* Asrc = first ? g.dpre : g.gy;
  const int lda = first ? H : D, Mdim = first ? H : D, Ndim = first ? K1 : H;
  f32x16 acc = zero16();
  for (int64_t rb = r_begin; rb < r1; rb += r_step) {
#pragma unroll 4
    for (int s = 0; s < 16; ++s) {
      const int64_t row = rb + acc_row(s, lane);    // k index of step s for this lane half (any bijection works)
      float av = 0.f, bv = 0.f;
      if (row < r1) {
        if (i < Mdim) av = Asrc[row * lda + i];
        if (j == Ndim) bv = 1.f;
        else if (j < Ndim) {
          if (!first) bv = a.hid[row * H + j];
          else if (j < D) {
            int64_t id = a.ids[row];
            if (id < 0 || id >= a.n_rows) id = 0;
            bv = a.table[id * D + j];
          } else bv = a.genres[row * 18 + (j - D)];
        }
      }
      acc = mfma32(av, bv, acc);
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
