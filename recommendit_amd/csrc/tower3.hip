// Tower backward in ONE kernel (d = hidden = 128, chip-filling batches): data gradients AND weight gradients of a 32-row
// tile by one 8-wave workgroup, without the row-major gy / dPre round trip through HBM of the two-kernel form
// (tower2.hip: tower_bwd_data_kernel + tower_wgrad_kernel move ~5 KB per row; this kernel ~2.7 KB).
// STATUS: correct (tests/test_gpu_towers.py, "coop" kernel family) and selectable with RIHIP_TOWER_BWD=4, but NOT the
// default: on MI355X it takes 154 + 312 us for 65 536 + 131 072 rows against 138 + 277 us of the two-kernel form.  The
// LDS holds both weight matrices, so the workgroup is 8 waves = 2 per SIMD = 256 registers per lane; the weight-gradient
// accumulators (80) + the operands staged for them (96) + gy in the row layout (128 while it is formed) leave no room to
// request a tile's operands one phase ahead -- a software-pipelined variant spilled 250 B per lane and ran at 405 us --
// and without that three memory latencies per tile are exposed (15-19 us per tile against 3.8 us of MFMA work).
// Reference: the implicit autograd backward of UserTower / ItemTower (src/models/two_tower.py:39-42, :68-72; called at
// src/training/train_embeddings.py:190).
//
// What makes the fusion fit: the MFMA operand layouts.  An accumulator tile computed as D[row][feature] holds, per lane,
// ONE feature and 16 batch rows -- exactly the A operand (A[i = feature][k = batch row]) of a weight-gradient product
// that contracts over the batch.  So
//   waves 0-3 (one hidden tile each):  dh[:, tile] = gy . W2[:, tile]   (A = gy rows in registers, B = W2 from LDS)
//                                      dPre = (hid > 0) ? dh * scale : 0 -> written to LDS (for dX) and kept in registers,
//                                      where it IS the A operand of  dW1[tile, :] += dPre^T . x  (B = the gathered rows
//                                      (|| genres), read straight from HBM, coalesced);  db1 = column sums.
//   waves 4-7 (one d tile each):       dW2[tile, :] += gy^T . hid  (A = gy recomputed in the feature-per-lane layout from
//                                      coalesced gout / out loads + the row scalars waves 0-3 left in LDS, B = hid from
//                                      HBM);  db2;  then  dX[:, tile] = dPre . W1[:, tile]  (A = the dPre tile in LDS,
//                                      B = W1 from LDS)  -> coalesced stores.
// Per tile both halves issue 128-144 MFMAs between the same two barriers; waves w and w+4 share a SIMD, so the two
// halves interleave on every SIMD.  LDS: W2 64 KB + W1[:, :128] 64 KB + the dPre tile 16.5 KB + row scalars.
// Weight-gradient accumulators stay in registers for the whole kernel (80 / 64 per lane); one slab per workgroup in the
// layout of the other backward kernels ([dW1 | db1 | dW2 | db2]), summed in fixed order by the slab-reduce kernels.
#include "common.h"
#include "tower_args.h"

namespace {

constexpr int D3 = 128, H3 = 128;
constexpr int LDP3 = H3 + 4;      // dPre tile rows: ds_read_b128 A operand of dX

template <bool ITEM>
__global__ __launch_bounds__(512, 2) void tower_bwd3_kernel(TowerBwdArgs a) {
  constexpr int D = D3, H = H3;
  constexpr int K1 = D + (ITEM ? 18 : 0);
  constexpr int NX = ITEM ? 5 : 4;              // 32-column tiles of x (the 5th: 18 genre columns, rest zero)
  constexpr int P = H * K1 + H + D * H + D;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* W2s = smem;                 // [D][H]   W2s[d][h] = W2[d][h]          (B operand of dh: lanes = consecutive h)
  float* W1s = W2s + D * H;          // [H][D]   W1s[h][d] = W1[h][d], d < D   (B operand of dX: lanes = consecutive d)
  float* Ps = W1s + H * D;           // [32][LDP3] dPre tile
  float* rowS = Ps + 32 * LDP3;      // [2][2][32] (dot, 1/denom) of the tile's rows, double-buffered
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int r31 = lane & 31, hh = lane >> 5;
  for (int i = tid; i < D * H / 4; i += 512) reinterpret_cast<f32x4*>(W2s)[i] = reinterpret_cast<const f32x4*>(a.W2)[i];
  for (int i = tid; i < H * D; i += 512) {
    const int hrow = i / D, dc = i % D;
    W1s[i] = a.W1[(size_t)hrow * K1 + dc];
  }
  const int64_t ntiles = (a.B + 31) / 32;
  const bool front = w < 4;          // waves 0-3: dh / dW1 ; waves 4-7: dW2 / dX
  const int tw = w & 3;              // this wave's 32-wide tile (hidden tile / d tile)

  f32x16 wacc[5];                    // front: dW1[tile tw, 4-5 k1 tiles]; back: dW2[tile tw, 4 hidden tiles]
#pragma unroll
  for (int t = 0; t < 5; ++t) wacc[t] = zero16();
  float bsum = 0.f;                  // db1 (front) / db2 (back) of feature tw*32 + r31: sum over this lane's rows
  __syncthreads();

  int buf = 0;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x, buf ^= 1) {
    const int64_t row_base = tile * 32;
    float* dotS = rowS + buf * 64;
    float* invS = dotS + 32;
    f32x16 dacc = zero16();
    if (front) {
      // ---- gy of the tile in the row-per-lane layout (lane (row, hh) holds its half of the row's k-blocks)
      const int64_t grow = row_base + r31;
      const bool ok = grow < a.B;
      const int64_t gr = ok ? grow : a.B - 1;
      f32x4 gy[16];
      {
        const float* gp = a.gout + gr * D + 4 * hh;
        const float* op = a.out + gr * D + 4 * hh;
        f32x4 o[16];
#pragma unroll
        for (int kb = 0; kb < 16; ++kb) {
          gy[kb] = *reinterpret_cast<const f32x4*>(gp + kb * 8);
          o[kb] = *reinterpret_cast<const f32x4*>(op + kb * 8);
        }
        float dot = 0.f;
#pragma unroll
        for (int kb = 0; kb < 16; ++kb)
          dot += gy[kb].x * o[kb].x + gy[kb].y * o[kb].y + gy[kb].z * o[kb].z + gy[kb].w * o[kb].w;
        dot += __shfl_xor(dot, 32, 64);
        const float dn = a.denom[gr];
        if (dn <= 1e-12f) dot = 0.f;  // clamp branch of F.normalize: out = y/eps, d out/dy = 1/eps
        const float inv = ok ? 1.f / dn : 0.f;
        if (w == 0 && hh == 0) { dotS[r31] = dot; invS[r31] = inv; }
#pragma unroll
        for (int kb = 0; kb < 16; ++kb) gy[kb] = (gy[kb] - o[kb] * dot) * inv;
      }
      // the mask of this wave's hidden tile, feature-per-lane: hid[row_k][tw*32 + r31] for the 16 rows of this lane half
      float hv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t rr = row_base + acc_row(r, lane);
        hv[r] = a.hid[(rr < a.B ? rr : a.B - 1) * H + tw * 32 + r31];
      }
      // ---- dh[:, tile tw] = gy . W2[:, tile]  (k = d; step s of k-block kb uses d = 8kb + 4hh + s for both operands)
      const float* w2p = W2s + tw * 32 + r31;
#pragma unroll
      for (int kb = 0; kb < 16; ++kb) {
        const int d0 = 8 * kb + 4 * hh;
        dacc = mfma32(gy[kb].x, w2p[(d0 + 0) * H], dacc);
        dacc = mfma32(gy[kb].y, w2p[(d0 + 1) * H], dacc);
        dacc = mfma32(gy[kb].z, w2p[(d0 + 2) * H], dacc);
        dacc = mfma32(gy[kb].w, w2p[(d0 + 3) * H], dacc);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const bool rok = row_base + acc_row(r, lane) < a.B;
        dacc[r] = (rok && hv[r] > 0.f) ? dacc[r] * a.scale : 0.f;
        bsum += dacc[r];
      }
    }
    __syncthreads();     // (A) the back waves are done reading the previous tile's dPre from Ps
    if (front) {
#pragma unroll
      for (int r = 0; r < 16; ++r) Ps[acc_row(r, lane) * LDP3 + tw * 32 + r31] = dacc[r];
    }
    __syncthreads();     // (B) Ps and the row scalars of this tile are visible
    if (front) {
      // ---- dW1[tile tw, :] += dPre^T . x   (A = dacc: lane = hidden unit, register = batch row; B = x[row][k1])
      // every x value of the tile is requested before the first MFMA: one memory latency per tile, not one per k1 tile
      int64_t idv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t rr = row_base + acc_row(r, lane);
        int64_t id = a.ids[rr < a.B ? rr : a.B - 1];
        if (id < 0 || id >= a.n_rows) id = 0;
        idv[r] = id;
      }
      float xv[NX][16];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) xv[nt][r] = a.table[idv[r] * D + nt * 32 + r31];
      if (ITEM) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t rr = row_base + acc_row(r, lane);
          xv[NX - 1][r] = a.genres[(rr < a.B ? rr : a.B - 1) * 18 + (r31 < 18 ? r31 : 0)];
        }
      }
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) wacc[nt] = mfma32(dacc[r], xv[nt][r], wacc[nt]);   // (dacc = 0 for rows past B)
      if (ITEM) {
#pragma unroll
        for (int r = 0; r < 16; ++r) wacc[4] = mfma32(dacc[r], r31 < 18 ? xv[NX - 1][r] : 0.f, wacc[4]);
      }
    } else {
      // ---- gy of the tile in the feature-per-lane layout for this wave's d tile; all of the tile's loads (gout, out and
      // the four hidden tiles of hid) are requested before anything consumes them
      float gyf[16], hv[4][16];
      {
        float gv[16], ov[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t rr = row_base + acc_row(r, lane);
          const int64_t rc = rr < a.B ? rr : a.B - 1;
          gv[r] = a.gout[rc * D + tw * 32 + r31];
          ov[r] = a.out[rc * D + tw * 32 + r31];
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) hv[nt][r] = a.hid[rc * H + nt * 32 + r31];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rl = acc_row(r, lane);
          gyf[r] = (gv[r] - ov[r] * dotS[rl]) * invS[rl];       // (1/denom is 0 for rows past B)
          bsum += gyf[r];
        }
      }
      // ---- dW2[tile tw, :] += gy^T . hid
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) wacc[nt] = mfma32(gyf[r], hv[nt][r], wacc[nt]);
      // ---- dX[:, tile tw] = dPre . W1[:, tile]   (A = the dPre tile in LDS, B = W1s[h][d])
      f32x16 xacc = zero16();
      const float* w1p = W1s + tw * 32 + r31;
#pragma unroll
      for (int kb = 0; kb < 16; ++kb) {
        const f32x4 av = *reinterpret_cast<const f32x4*>(&Ps[r31 * LDP3 + 8 * kb + 4 * hh]);
        const int h0 = 8 * kb + 4 * hh;
        xacc = mfma32(av.x, w1p[(h0 + 0) * D], xacc);
        xacc = mfma32(av.y, w1p[(h0 + 1) * D], xacc);
        xacc = mfma32(av.z, w1p[(h0 + 2) * D], xacc);
        xacc = mfma32(av.w, w1p[(h0 + 3) * D], xacc);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t orow = row_base + acc_row(r, lane);
        if (orow < a.B) a.dX[orow * D + tw * 32 + r31] = xacc[r];
      }
    }
  }

  // ---- this workgroup's slab: [dW1 H*K1 | db1 H | dW2 D*H | db2 D]
  float* slab = a.slab + (size_t)blockIdx.x * P;
  bsum += __shfl_xor(bsum, 32, 64);
  if (front) {
#pragma unroll
    for (int nt = 0; nt < NX; ++nt) {
      const int k1 = nt * 32 + r31;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (k1 < K1) slab[(size_t)(tw * 32 + acc_row(r, lane)) * K1 + k1] = wacc[nt][r];
    }
    if (hh == 0) slab[H * K1 + tw * 32 + r31] = bsum;
  } else {
    float* s2 = slab + H * K1 + H;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) s2[(size_t)(tw * 32 + acc_row(r, lane)) * H + nt * 32 + r31] = wacc[nt][r];
    if (hh == 0) s2[D * H + tw * 32 + r31] = bsum;
  }
}

constexpr size_t bwd3_lds() { return sizeof(float) * ((size_t)D3 * H3 * 2 + 32 * LDP3 + 128); }

}  // namespace

// returns the number of slabs written (0: shape not covered)
int rihip_launch_tower_bwd3(int d, int hidden, bool item, const TowerBwdArgs& a, int max_slabs, hipStream_t st) {
  if (d != D3 || hidden != H3) return 0;
  static bool granted = false;
  if (!granted) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tower_bwd3_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)bwd3_lds());
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tower_bwd3_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)bwd3_lds());
    granted = true;
  }
  const int64_t ntiles = (a.B + 31) / 32;
  int grid = (int)(ntiles < RIHIP_NCU ? ntiles : RIHIP_NCU);
  if (grid > max_slabs) grid = max_slabs;
  if (grid < 1) grid = 1;
  if (item) hipLaunchKernelGGL(tower_bwd3_kernel<true>, dim3(grid), dim3(512), bwd3_lds(), st, a);
  else hipLaunchKernelGGL(tower_bwd3_kernel<false>, dim3(grid), dim3(512), bwd3_lds(), st, a);
  return grid;
}
