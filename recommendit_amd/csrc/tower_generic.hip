// Generic-shape tower kernels: ANY (embed_dim, hidden_dim) with both multiples of 16 up to 256.
//
// The reference builds its towers from whatever (embed_dim, hidden_dim) it is given (src/models/two_tower.py:27-33, :54-62,
// :80-95; EMBEDDING_DIM is env-overridable at src/config.py:13).  The tuned kernels (tower.hip, tower2.hip) are template
// instantiations for five pairs; this file is the path behind them for every other pair: runtime sizes, same math, same
// dropout counters, same outputs and the same slab layout for the weight gradients, so everything downstream (slab
// reduction, scatter, optimiser) is shared.
//
//   forward   : one workgroup (4 waves) per 32-row tile; the gathered rows (|| genres) and the hidden tile live in LDS,
//               the weights stream through a 32-wide k-panel in LDS (zero-padded to the MFMA tile), exact-f32 MFMA.
//   backward  : (a) data-gradient kernel, same tiling: gy (normalise-backward), dh -> dPre, dX; gy and dPre are also
//               written row-major to the workspace;  (b) weight-gradient kernel: dW2 = gy^T.hid, dW1 = dPre^T.x as split-K
//               GEMMs over the batch with both operands read straight from their row-major arrays (k = batch row is the
//               register index of the MFMA operand layout, so no transpose is needed); the bias gradients ride as one
//               extra "ones" column of each product.  One slab per batch split.
#include "common.h"
#include "gen_gemm.h"
#include "tower_args.h"

using namespace rihip_gen;

namespace {

struct GenFwd {
  TowerFwdArgs a;
  int D, H, K1;
};

__global__ __launch_bounds__(256) void tower_fwd_generic_kernel(GenFwd g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const TowerFwdArgs& a = g.a;
  const int D = g.D, H = g.H, K1 = g.K1;
  const int K1p = up8(K1), ldx = K1p + 4, ldh = H + 4, ldy = D + 4;
  float* Xs = smem;                                   // [32][ldx] (later Y: [32][ldy], ldy <= ldx)
  float* Hs = Xs + GTM * ldx;                         // [32][ldh]
  float* Wp = Hs + GTM * ldh;                         // [256][GLDP]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const uint64_t seed_mul = a.seed_step ? rihip_splitmix64(a.seed_mul + (uint64_t)(*a.seed_step)) : a.seed_mul;
  const int64_t ntiles = (a.B + GTM - 1) / GTM;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
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
//     dX = dPre.W1[:, :D]
__global__ __launch_bounds__(256) void tower_bwd_data_generic_kernel(GenBwd g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const TowerBwdArgs& a = g.a;
  const int D = g.D, H = g.H, K1 = g.K1;
  const int ldg = D + 4, ldh = H + 4;
  float* Gs = smem;                  // [32][ldg]  gy
  float* Ps = Gs + GTM * ldg;        // [32][ldh]  dPre
  float* Wp = Ps + GTM * ldh;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int64_t ntiles = (a.B + GTM - 1) / GTM;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
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

// (b) weight gradients.  blockIdx.y = slab (a range of batch rows), blockIdx.x = a group of 4 output tiles (one per
// wave) out of  [dW1 | db1]: ceil(H/32) x ceil((K1+1)/32) tiles, M = h, N = k1 (column K1 = ones -> db1)
//               [dW2 | db2]: ceil(D/32) x ceil((H+1)/32) tiles,  M = dcol, N = h (column H = ones -> db2).
// Operand layout of mfma32: lane l supplies A[i = l&31][k] and B[k][j = l&31] with k = the batch row picked by the
// step and l>>5 -- so both operands are plain coalesced reads of row-major [batch, feature] arrays.
__global__ __launch_bounds__(256) void tower_wgrad_generic_kernel(GenBwd g) {
  const TowerBwdArgs& a = g.a;
  const int D = g.D, H = g.H, K1 = g.K1;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int mt1 = (H + 31) / 32, nt1 = (K1 + 1 + 31) / 32, mt2 = (D + 31) / 32, nt2 = (H + 1 + 31) / 32;
  const int n1 = mt1 * nt1, n2 = mt2 * nt2;
  const int tile = blockIdx.x * 4 + w;
  if (tile >= n1 + n2) return;
  const bool first = tile < n1;                 // dW1 product
  const int tt = first ? tile : tile - n1;
  const int mt = first ? tt / nt1 : tt / nt2, nt = first ? tt % nt1 : tt % nt2;
  const int i = mt * 32 + (lane & 31);          // A feature: h (dW1) or dcol (dW2)
  const int j = nt * 32 + (lane & 31);          // B feature: k1 (dW1) or h (dW2)
  const float* Asrc = first ? g.dpre : g.gy;
  const int lda = first ? H : D, Mdim = first ? H : D, Ndim = first ? K1 : H;
  const int64_t r0 = (int64_t)blockIdx.y * g.rows_per_slab;
  int64_t r1 = r0 + g.rows_per_slab;
  if (r1 > a.B) r1 = a.B;
  f32x16 acc = zero16();
  for (int64_t rb = r0; rb < r1; rb += 32) {
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
  // slab layout (shared with the tuned kernels): [dW1 H*K1 | db1 H | dW2 D*H | db2 D]
  float* slab = a.slab + (size_t)blockIdx.y * g.P;
  const int oW1 = 0, ob1 = H * K1, oW2 = ob1 + H, ob2 = oW2 + D * H;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = mt * 32 + acc_row(r, lane);
    if (m >= Mdim) continue;
    if (first) {
      if (j < K1) slab[oW1 + m * K1 + j] = acc[r];
      else if (j == K1) slab[ob1 + m] = acc[r];
    } else {
      if (j < H) slab[oW2 + m * H + j] = acc[r];
      else if (j == H) slab[ob2 + m] = acc[r];
    }
  }
}

size_t fwd_lds(int D, int H, int K1) { return sizeof(float) * ((size_t)GTM * (up8(K1) + 4) + (size_t)GTM * (H + 4) + 256 * GLDP); }
size_t bwd_lds(int D, int H) { return sizeof(float) * ((size_t)GTM * (D + 4) + (size_t)GTM * (H + 4) + 256 * GLDP); }

}  // namespace

bool rihip_tower_generic_ok(int d, int hidden) {
  return d >= 16 && d <= 256 && hidden >= 16 && hidden <= 256 && d % 16 == 0 && hidden % 16 == 0;
}

void rihip_launch_tower_fwd_generic(int d, int hidden, bool item, const TowerFwdArgs& a, hipStream_t st) {
  GenFwd g;
  g.a = a; g.D = d; g.H = hidden; g.K1 = d + (item ? 18 : 0);
  const size_t lds = fwd_lds(d, hidden, g.K1);
  static bool granted = false;
  if (!granted) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tower_fwd_generic_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)fwd_lds(256, 256, 274));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tower_bwd_data_generic_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)bwd_lds(256, 256));
    granted = true;
  }
  const int64_t ntiles = (a.B + GTM - 1) / GTM;
  const int grid = (int)(ntiles < 4 * RIHIP_NCU ? ntiles : 4 * RIHIP_NCU);
  hipLaunchKernelGGL(tower_fwd_generic_kernel, dim3(grid), dim3(256), lds, st, g);
}

// returns the number of slabs written; `act` = [B,d] gy followed by [B,hidden] dPre
int rihip_launch_tower_bwd_generic(int d, int hidden, bool item, const TowerBwdArgs& a, float* act, int max_slabs,
                                   hipStream_t st, hipEvent_t dx_event) {
  GenBwd g;
  g.a = a; g.D = d; g.H = hidden; g.K1 = d + (item ? 18 : 0);
  g.gy = act; g.dpre = act + (size_t)a.B * d;
  g.P = hidden * g.K1 + hidden + d * hidden + d;
  static bool granted = false;
  if (!granted) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tower_fwd_generic_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)fwd_lds(256, 256, 274));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tower_bwd_data_generic_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)bwd_lds(256, 256));
    granted = true;
  }
  const int64_t ntiles = (a.B + GTM - 1) / GTM;
  const int grid = (int)(ntiles < 4 * RIHIP_NCU ? ntiles : 4 * RIHIP_NCU);
  hipLaunchKernelGGL(tower_bwd_data_generic_kernel, dim3(grid), dim3(256), bwd_lds(d, hidden), st, g);
  if (dx_event) (void)hipEventRecord(dx_event, st);
  // batch splits: enough workgroups to fill the chip, at most max_slabs, whole 32-row tiles each
  const int n_tiles_out = ((hidden + 31) / 32) * ((g.K1 + 1 + 31) / 32) + ((d + 31) / 32) * ((hidden + 1 + 31) / 32);
  const int gx = (n_tiles_out + 3) / 4;
  int64_t want = (2 * RIHIP_NCU + gx - 1) / gx;
  if (want > max_slabs) want = max_slabs;
  if (want > ntiles) want = ntiles;
  if (want < 1) want = 1;
  const int64_t tiles_per = (ntiles + want - 1) / want;
  g.rows_per_slab = (int)(tiles_per * GTM);
  g.nslab = (int)((ntiles + tiles_per - 1) / tiles_per);
  hipLaunchKernelGGL(tower_wgrad_generic_kernel, dim3(gx, g.nslab), dim3(256), 0, st, g);
  return g.nslab;
}
