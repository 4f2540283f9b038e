// Tower forward, second kernel family: one WAVE owns 32 batch rows end to end, the weights live in LDS (shared by the
// 8 waves of the workgroup) and the activations never leave registers -- no workgroup barrier in the row loop, so the
// two waves of a SIMD overlap freely (gather latency of one under the MFMAs of the other).
//
//   gather   lane (row, half) reads its half of the row's k-blocks straight into the MFMA B-operand registers
//   GEMM 1   H^T[hid][row] = W1 . X^T     A = W1 rows from LDS (one ds_read_b128 per 4 MFMAs), B = the row registers
//   epilogue bias + ReLU + dropout on the accumulators (lane <-> row, register <-> hidden unit)
//   GEMM 2   Y[row][d]    = H . W2^T     A = the H^T accumulator registers themselves (k <-> their row index),
//                                          B = W2^T from LDS (consecutive d per lane: conflict-free ds_read_b32)
//   epilogue bias, row norm by a 32-lane reduction, coalesced 128-B row stores
// Same arithmetic as tower_fwd_kernel (tower.hip): exact-f32 MFMA chains over k in the same k-block order.
#include "common.h"
#include "recommendit_hip.h"
#include "tower_args.h"

namespace {

template <int D, int H, bool ITEM>
__global__ __launch_bounds__(512, 2) void tower_fwd2_kernel(TowerFwdArgs a) {
  constexpr int K1 = D + (ITEM ? 18 : 0);
  constexpr int K1P = (K1 + 7) / 8 * 8;
  constexpr int KB1 = K1P / 8;
  constexpr int LDW1 = K1P + 4;   // floats per W1 row in LDS  (b128 reads of 16 lanes land on distinct 4-bank groups)
  constexpr int LDW2 = D + 4;     // floats per W2^T row in LDS
  constexpr int HT = H / 32, DT = D / 32;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* W1s = smem;                 // [H][LDW1]
  float* W2Ts = W1s + H * LDW1;      // [H][LDW2]   W2Ts[hid][d] = W2[d][hid]
  float* b1s = W2Ts + H * LDW2;      // [H]

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int r31 = lane & 31, hh = lane >> 5;

  // ---- weights -> LDS (once per workgroup)
  for (int i = tid; i < H * K1P; i += 512) {
    const int hrow = i / K1P, k = i % K1P;
    W1s[hrow * LDW1 + k] = (k < K1) ? a.W1[(size_t)hrow * K1 + k] : 0.f;
  }
  for (int i = tid; i < D * H; i += 512) {
    const int drow = i / H, hcol = i % H;
    W2Ts[hcol * LDW2 + drow] = a.W2[i];
  }
  for (int i = tid; i < H; i += 512) b1s[i] = a.b1[i];
  float b2v[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) b2v[dt] = a.b2[dt * 32 + r31];
  const uint64_t seed_mul = a.seed_step ? rihip_splitmix64(a.seed_mul + (uint64_t)(*a.seed_step)) : a.seed_mul;
  const uint32_t inner0 = rihip_lowbias32((uint32_t)(seed_mul >> 32) + 0x9E3779B9u);
  __syncthreads();

  const int64_t ntiles = (a.B + 31) / 32;
  const int64_t tstride = (int64_t)gridDim.x * 8;

  // B-operand registers of a 32-row tile: lane (row, hh) <- X[row][8kb + 4hh .. +3]
  f32x4 xr[KB1];
  // Rows past the end are clamped to the last row (their results are never stored; a row only feeds its own output
  // row): every load is unconditional, so the 16-19 independent 16-B requests of a lane are all in flight together.
  auto gather = [&](int64_t tile) {
    int64_t grow = tile * 32 + r31;
    grow = grow < a.B ? grow : a.B - 1;
    int64_t id = a.ids[grow];
    if (id < 0 || id >= a.n_rows) {
      if (a.err_flag) *a.err_flag = 1;
      id = 0;
    }
    const float* row = a.table + (size_t)id * D + 4 * hh;
    const float* grw = ITEM ? a.genres + grow * 18 : nullptr;
#pragma unroll
    for (int kb = 0; kb < KB1; ++kb) {
      if (kb * 8 + 8 <= D) {
        xr[kb] = *reinterpret_cast<const f32x4*>(row + kb * 8);
      } else {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (ITEM) {
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const int gc = kb * 8 + 4 * hh + s - D;          // genre column of this lane (depends on hh only)
            const float gv = grw[gc < 18 ? gc : 17];
            v[s] = gc < 18 ? gv : 0.f;
          }
        }
        xr[kb] = v;
      }
    }
  };

  int64_t tile = (int64_t)blockIdx.x * 8 + w;
  if (tile < ntiles) gather(tile);
  for (; tile < ntiles; tile += tstride) {
    const int64_t row_base = tile * 32;
    // ---- GEMM 1: H^T tiles
    f32x16 hacc[HT];
#pragma unroll
    for (int ht = 0; ht < HT; ++ht) hacc[ht] = zero16();
#pragma unroll
    for (int kb = 0; kb < KB1; ++kb) {
#pragma unroll
      for (int ht = 0; ht < HT; ++ht) {
        const f32x4 av = *reinterpret_cast<const f32x4*>(&W1s[(ht * 32 + r31) * LDW1 + kb * 8 + 4 * hh]);
        hacc[ht] = mfma32(av.x, xr[kb].x, hacc[ht]);
        hacc[ht] = mfma32(av.y, xr[kb].y, hacc[ht]);
        hacc[ht] = mfma32(av.z, xr[kb].z, hacc[ht]);
        hacc[ht] = mfma32(av.w, xr[kb].w, hacc[ht]);
      }
      if ((kb & 1) == 1) __builtin_amdgcn_sched_barrier(0);  // bound how far the LDS reads are hoisted (registers)
    }
    // ---- bias + ReLU + dropout (lane <-> row r31, register r of tile ht <-> hidden unit ht*32 + acc_row(r)).
    // acc_row(4q..4q+3) are 4 consecutive hidden units: biases come as one ds_read_b128 per 4 registers; the three
    // wave-uniform cases (eval / train with 32-bit counters / 64-bit counters) are separate straight-line loops.
    const int64_t grow = row_base + r31;
    const uint64_t idx0 = (uint64_t)(a.row0 + row_base) * H;
    const bool idx32 = (idx0 + 32ull * H) < (1ull << 32);
    const uint32_t idx_lane = (uint32_t)idx0 + (uint32_t)(r31 * H + 4 * hh);
    auto relu_bias = [&](int ht, int q, f32x4& v) {
      const f32x4 bq = *reinterpret_cast<const f32x4*>(&b1s[ht * 32 + 8 * q + 4 * hh]);
#pragma unroll
      for (int s = 0; s < 4; ++s) v[s] = fmaxf(hacc[ht][4 * q + s] + bq[s], 0.f);
    };
    if (!a.training) {
#pragma unroll
      for (int ht = 0; ht < HT; ++ht)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 v;
          relu_bias(ht, q, v);
#pragma unroll
          for (int s = 0; s < 4; ++s) hacc[ht][4 * q + s] = v[s];
        }
    } else if (idx32) {
      const uint32_t seed_lo = (uint32_t)seed_mul;
#pragma unroll
      for (int ht = 0; ht < HT; ++ht)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 v;
          relu_bias(ht, q, v);
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const bool keep = rihip_keep32(seed_lo, inner0, idx_lane + (uint32_t)(ht * 32 + 8 * q + s), a.thresh24);
            hacc[ht][4 * q + s] = keep ? v[s] * a.scale : 0.f;
          }
        }
    } else {
#pragma unroll
      for (int ht = 0; ht < HT; ++ht)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 v;
          relu_bias(ht, q, v);
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const int hu = ht * 32 + 8 * q + 4 * hh + s;
            const bool keep = rihip_keep(seed_mul, (uint64_t)(a.row0 + grow) * H + hu, a.thresh24);
            hacc[ht][4 * q + s] = keep ? v[s] * a.scale : 0.f;
          }
        }
    }
    if (a.hid && grow < a.B) {  // [B][H] row-major (what the backward reads): 16-B pieces, 4 consecutive hidden units
      float* hp = a.hid + grow * H + 4 * hh;
#pragma unroll
      for (int ht = 0; ht < HT; ++ht)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 v = {hacc[ht][4 * q], hacc[ht][4 * q + 1], hacc[ht][4 * q + 2], hacc[ht][4 * q + 3]};
          *reinterpret_cast<f32x4*>(hp + ht * 32 + 8 * q) = v;
        }
    }

    // ---- GEMM 2: Y tiles; A = hacc registers (k = hidden unit of that register), B = W2^T rows from LDS
    f32x16 yacc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) yacc[dt] = zero16();
#pragma unroll
    for (int ht = 0; ht < HT; ++ht) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int hu = ht * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) yacc[dt] = mfma32(hacc[ht][r], W2Ts[hu * LDW2 + dt * 32 + r31], yacc[dt]);
        if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
    }

    // ---- bias, row norm (row = acc_row(r) of every register; 32 lanes x DT tiles hold one row), normalised store.
    // The 16 row sums are reduced together (independent chains): 4 DPP stages inside 16-lane rows, one cross-row swap.
    float ss[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float acc2 = 0.f;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const float y = yacc[dt][r] + b2v[dt];
        yacc[dt][r] = y;
        acc2 += y * y;
      }
      ss[r] = acc2;
    }
#define RIHIP_DPP_ADD(CTRL)                                                                                      \
  _Pragma("unroll") for (int r = 0; r < 16; ++r) ss[r] +=                                                        \
      __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, ss[r]), CTRL, 0xF, 0xF, true));
    RIHIP_DPP_ADD(0xB1)    // quad_perm [1,0,3,2]
    RIHIP_DPP_ADD(0x4E)    // quad_perm [2,3,0,1]
    RIHIP_DPP_ADD(0x141)   // row_half_mirror: lane i <-> 7-i of its 8-lane group (pairs the two quads)
    RIHIP_DPP_ADD(0x140)   // row_mirror: lane i <-> 15-i of its 16-lane row (pairs the two halves)
#undef RIHIP_DPP_ADD
#pragma unroll
    for (int r = 0; r < 16; ++r) ss[r] += __shfl_xor(ss[r], 16, 64);  // the other 16-lane row of this 32-lane half
    const int64_t obase = row_base + 4 * hh;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float dn = fmaxf(sqrtf(ss[r]), 1e-12f);
      const int64_t orow = obase + (r & 3) + 8 * (r >> 2);
      if (orow < a.B) {
        const float inv = 1.f / dn;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) a.out[orow * D + dt * 32 + r31] = yacc[dt][r] * inv;
        if (r31 == 0 && a.denom) a.denom[orow] = dn;
      }
    }
    // next tile's rows: requested once the output registers are dead (earlier would spill); the latency is covered
    // by the other wave of the SIMD
    if (tile + tstride < ntiles) gather(tile + tstride);
  }
}

template <int D, int H, bool ITEM>
constexpr size_t fwd2_lds() {
  constexpr int K1P = (D + (ITEM ? 18 : 0) + 7) / 8 * 8;
  return ((size_t)H * (K1P + 4) + (size_t)H * (D + 4) + H) * sizeof(float);
}

template <int D, int H, bool ITEM>
void launch_one(const TowerFwdArgs& a, hipStream_t st) {
  auto k = tower_fwd2_kernel<D, H, ITEM>;
  static bool granted = false;
  if (!granted) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)fwd2_lds<D, H, ITEM>());
    granted = true;
  }
  const int64_t nchunks = (a.B + 255) / 256;  // 8 waves x 32 rows per workgroup pass
  const int grid = (int)(nchunks < RIHIP_NCU ? nchunks : RIHIP_NCU);
  const size_t lds = fwd2_lds<D, H, ITEM>();
  hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, st, a);
}
template <int D, int H>
void launch_dh(bool item, const TowerFwdArgs& a, hipStream_t st) {
  if (item) launch_one<D, H, true>(a, st);
  else launch_one<D, H, false>(a, st);
}

// ==================================================================================================================
// Backward, two kernels.
//
// (A) tower_bwd_data_kernel: the forward's structure run backwards -- a wave owns 32 rows, W2^T and W1[:, :D] live in
//     LDS, activations stay in registers:  gy = normalise-backward(gout, out, denom)  ->  dh^T = W2^T . gy^T (A = W2^T
//     rows, B = gy row registers)  ->  dPre = (hid > 0) ? dh * scale : 0  ->  dx = dPre . W1[:, :D] (A = the dh^T
//     accumulator registers, B = W1 rows from LDS)  -> coalesced dX stores.  gy [B,D] and dPre [B,H] are also written
//     row-major for (B).
// (B) tower_wgrad_kernel: dW2 = gy^T . hid, dW1 = dPre^T . x, db2, db1 -- contractions over the batch -- as an
//     LDS-tiled split-K product: 8 waves, 32-row tiles double-buffered through registers; waves 0-3 own one d-tile of
//     dW2 each (x 4 hidden tiles, plus the column sums), waves 4-7 one hidden tile of dW1 (x 4-5 input tiles); every
//     workgroup writes one slab, summed in fixed order by the slab-reduce kernels of tower.hip.
// ==================================================================================================================
template <int D, int H>
__global__ __launch_bounds__(512, 2) void tower_bwd_data_kernel(TowerBwdArgs a, float* __restrict__ gy_out,
                                                                float* __restrict__ dpre_out, int K1) {
  constexpr int KBD = D / 8;
  constexpr int LDW2 = D + 4;  // W2^T rows (k = d contiguous): ds_read_b128 A operand
  constexpr int LDW1 = D + 1;  // W1[:, :D] rows: ds_read_b32 B operand (consecutive d per lane)
  constexpr int HT = H / 32, DT = D / 32;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* W2Ts = smem;                // [H][LDW2]   W2Ts[hid][d] = W2[d][hid]
  float* W1s = W2Ts + H * LDW2;      // [H][LDW1]   W1s[hid][d]  = W1[hid][d], d < D

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int r31 = lane & 31, hh = lane >> 5;
  for (int i = tid; i < D * H; i += 512) {
    const int drow = i / H, hcol = i % H;
    W2Ts[hcol * LDW2 + drow] = a.W2[i];
  }
  for (int i = tid; i < H * D; i += 512) {
    const int hrow = i / D, dc = i % D;
    W1s[hrow * LDW1 + dc] = a.W1[(size_t)hrow * K1 + dc];
  }
  __syncthreads();

  const int64_t ntiles = (a.B + 31) / 32;
  const int64_t tstride = (int64_t)gridDim.x * 8;
  for (int64_t tile = (int64_t)blockIdx.x * 8 + w; tile < ntiles; tile += tstride) {
    const int64_t row_base = tile * 32;
    const int64_t grow = row_base + r31;
    const bool ok = grow < a.B;
    const int64_t gr = ok ? grow : a.B - 1;  // clamped: every load unconditional
    // the saved hidden activations of this lane's row are requested first: they are only needed after the first GEMM
    f32x4 hv[HT * 4];
    {
      const float* hp = a.hid + gr * H + 4 * hh;
#pragma unroll
      for (int ht = 0; ht < HT; ++ht)
#pragma unroll
        for (int q = 0; q < 4; ++q) hv[ht * 4 + q] = *reinterpret_cast<const f32x4*>(hp + ht * 32 + 8 * q);
    }
    // ---- gy = (gout - out * <gout, out>) / denom   (lane (row, hh) holds its half of the row's k-blocks)
    f32x4 gy[KBD];
    {
      const float* gp = a.gout + gr * D + 4 * hh;
      const float* op = a.out + gr * D + 4 * hh;
      f32x4 o[KBD];
      float dot = 0.f;
#pragma unroll
      for (int kb = 0; kb < KBD; ++kb) {
        gy[kb] = *reinterpret_cast<const f32x4*>(gp + kb * 8);
        o[kb] = *reinterpret_cast<const f32x4*>(op + kb * 8);
      }
#pragma unroll
      for (int kb = 0; kb < KBD; ++kb)
        dot += gy[kb].x * o[kb].x + gy[kb].y * o[kb].y + gy[kb].z * o[kb].z + gy[kb].w * o[kb].w;
      dot += __shfl_xor(dot, 32, 64);
      const float dn = a.denom[gr];
      if (dn <= 1e-12f) dot = 0.f;  // clamp branch of F.normalize: out = y/eps, d out/dy = 1/eps
      const float inv = ok ? 1.f / dn : 0.f;
      float* yp = gy_out + gr * D + 4 * hh;
#pragma unroll
      for (int kb = 0; kb < KBD; ++kb) {
        gy[kb] = (gy[kb] - o[kb] * dot) * inv;
        if (ok) *reinterpret_cast<f32x4*>(yp + kb * 8) = gy[kb];
      }
    }
    // ---- dh^T tiles
    f32x16 hacc[HT];
#pragma unroll
    for (int ht = 0; ht < HT; ++ht) hacc[ht] = zero16();
#pragma unroll
    for (int kb = 0; kb < KBD; ++kb) {
#pragma unroll
      for (int ht = 0; ht < HT; ++ht) {
        const f32x4 av = *reinterpret_cast<const f32x4*>(&W2Ts[(ht * 32 + r31) * LDW2 + kb * 8 + 4 * hh]);
        hacc[ht] = mfma32(av.x, gy[kb].x, hacc[ht]);
        hacc[ht] = mfma32(av.y, gy[kb].y, hacc[ht]);
        hacc[ht] = mfma32(av.z, gy[kb].z, hacc[ht]);
        hacc[ht] = mfma32(av.w, gy[kb].w, hacc[ht]);
      }
      if ((kb & 1) == 1) __builtin_amdgcn_sched_barrier(0);
    }
    // ---- dPre (register r of tile ht <-> hidden unit ht*32 + acc_row(r); 4 consecutive units per float4)
    {
      float* pp = dpre_out + gr * H + 4 * hh;
#pragma unroll
      for (int ht = 0; ht < HT; ++ht)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 v;
#pragma unroll
          for (int s2 = 0; s2 < 4; ++s2) {
            v[s2] = (hv[ht * 4 + q][s2] > 0.f && ok) ? hacc[ht][4 * q + s2] * a.scale : 0.f;
            hacc[ht][4 * q + s2] = v[s2];
          }
          if (ok) *reinterpret_cast<f32x4*>(pp + ht * 32 + 8 * q) = v;
        }
    }
    // ---- dx = dPre . W1[:, :D]
    f32x16 xacc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) xacc[dt] = zero16();
#pragma unroll
    for (int ht = 0; ht < HT; ++ht) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int hu = ht * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) xacc[dt] = mfma32(hacc[ht][r], W1s[hu * LDW1 + dt * 32 + r31], xacc[dt]);
        if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
    }
    const int64_t obase = row_base + 4 * hh;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t orow = obase + (r & 3) + 8 * (r >> 2);
      if (orow < a.B) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) a.dX[orow * D + dt * 32 + r31] = xacc[dt][r];
      }
    }
  }
}

template <int D, int H, bool ITEM>
__global__ __launch_bounds__(512, 2) void tower_wgrad_kernel(TowerBwdArgs a, const float* __restrict__ gy,
                                                             const float* __restrict__ dpre) {
  constexpr int K1 = D + (ITEM ? 18 : 0);
  constexpr int NX = (K1 + 31) / 32;       // 32-column tiles of x
  constexpr int XW = NX * 32;              // padded x width
  constexpr int CTD = D / 32, CTH = H / 32;
  static_assert(CTD == 4 && CTH == 4, "weight-gradient kernel: d = hidden = 128");
  constexpr int TROWS = 32;
  constexpr int BUF = TROWS * (D + H + H + XW);  // floats per buffer: gy | hid | dpre | x
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int r31 = lane & 31, hh = lane >> 5;

  // staging map: 512 threads x float4; row-major tiles
  constexpr int V_G = TROWS * D / 4, V_H = TROWS * H / 4, V_X = TROWS * (D / 4);
  constexpr int NV = (V_G + 2 * V_H + V_X + 511) / 512;
  f32x4 stg[NV];
  float stg_gen = 0.f;  // one genre value per thread (32 rows x 18 <= 576: threads 0..575? -> 2 passes below)
  float stg_gen2 = 0.f;
  // the x rows are gathered by id: the ids of a tile are requested one tile earlier than its rows, so that no row load
  // waits for its address inside the tile loop (an id -> row chain stalled every outstanding load of the step)
  constexpr int NXV = (V_X + 511) / 512;   // float4 slots of the x part per thread (they are the last slots)
  static_assert(V_G + 2 * V_H == (NV - NXV) * 512, "x part must start on a slot boundary");
  int64_t xid[NXV];
  auto load_ids = [&](int64_t tile) {
    const int64_t row_base = tile * TROWS;
#pragma unroll
    for (int q = 0; q < NXV; ++q) {
      const int j = tid + q * 512, r = j / (D / 4);
      const int64_t g = row_base + r;
      int64_t id = 0;
      if (j < V_X && g < a.B) id = a.ids[g];
      if (id < 0 || id >= a.n_rows) id = 0;
      xid[q] = id;
    }
  };
  auto load_tile = [&](int64_t tile) {
    const int64_t row_base = tile * TROWS;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = tid + i * 512;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (idx < V_G) {
        const int r = idx / (D / 4), c4 = idx % (D / 4);
        const int64_t g = row_base + r;
        if (g < a.B) v = reinterpret_cast<const f32x4*>(gy + g * D)[c4];
      } else if (idx < V_G + V_H) {
        const int j = idx - V_G, r = j / (H / 4), c4 = j % (H / 4);
        const int64_t g = row_base + r;
        if (g < a.B) v = reinterpret_cast<const f32x4*>(a.hid + g * H)[c4];
      } else if (idx < V_G + 2 * V_H) {
        const int j = idx - V_G - V_H, r = j / (H / 4), c4 = j % (H / 4);
        const int64_t g = row_base + r;
        if (g < a.B) v = reinterpret_cast<const f32x4*>(dpre + g * H)[c4];
      } else if (idx < V_G + 2 * V_H + V_X) {
        const int j = idx - V_G - 2 * V_H, r = j / (D / 4), c4 = j % (D / 4);
        const int64_t g = row_base + r;
        if (g < a.B) v = reinterpret_cast<const f32x4*>(a.table + (size_t)xid[i - (NV - NXV)] * D)[c4];
      }
      stg[i] = v;
    }
    if (ITEM) {
      const int r = tid / 18, c = tid % 18;   // threads 0..511 cover rows 0..28 (28*18+17 = 521 > 511): two passes
      const int64_t g = row_base + r;
      stg_gen = (r < TROWS && g < a.B) ? a.genres[g * 18 + c] : 0.f;
      const int t2 = tid + 512, r2 = t2 / 18, c2 = t2 % 18;
      const int64_t g2 = row_base + r2;
      stg_gen2 = (r2 < TROWS && g2 < a.B) ? a.genres[g2 * 18 + c2] : 0.f;
    }
  };
  auto store_tile = [&](float* buf) {
    float* Gs = buf;
    float* Hs = Gs + TROWS * D;
    float* Ps = Hs + TROWS * H;
    float* Xs = Ps + TROWS * H;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = tid + i * 512;
      if (idx < V_G) {
        *reinterpret_cast<f32x4*>(&Gs[idx * 4]) = stg[i];
      } else if (idx < V_G + V_H) {
        *reinterpret_cast<f32x4*>(&Hs[(idx - V_G) * 4]) = stg[i];
      } else if (idx < V_G + 2 * V_H) {
        *reinterpret_cast<f32x4*>(&Ps[(idx - V_G - V_H) * 4]) = stg[i];
      } else if (idx < V_G + 2 * V_H + V_X) {
        const int j = idx - V_G - 2 * V_H, r = j / (D / 4), c4 = j % (D / 4);
        *reinterpret_cast<f32x4*>(&Xs[r * XW + c4 * 4]) = stg[i];
      }
    }
    if (ITEM) {
      const int r = tid / 18, c = tid % 18;
      if (r < TROWS) Xs[r * XW + D + c] = stg_gen;
      const int t2 = tid + 512, r2 = t2 / 18, c2 = t2 % 18;
      if (r2 < TROWS) Xs[r2 * XW + D + c2] = stg_gen2;
    }
  };
  // x columns beyond K1 (tile padding) must be zero in both buffers: written once, never overwritten
  if (XW > K1) {
    constexpr int PADW = XW - K1;
    for (int i = tid; i < 2 * TROWS * PADW; i += 512) {
      const int b = i / (TROWS * PADW), j = i % (TROWS * PADW);
      smem[b * BUF + TROWS * (D + H + H) + (j / PADW) * XW + K1 + (j % PADW)] = 0.f;
    }
  }
  __syncthreads();

  constexpr int NACC = (NX > 4) ? NX : 4;
  f32x16 acc[NACC];
#pragma unroll
  for (int t = 0; t < NACC; ++t) acc[t] = zero16();
  float colsum = 0.f;  // waves 0-1: db2[tid], waves 2-3: db1[tid-128]

  const int64_t ntiles = (a.B + TROWS - 1) / TROWS;
  int64_t tile = blockIdx.x;
  if (tile < ntiles) {
    load_ids(tile);
    load_tile(tile);
    store_tile(smem);
    load_ids(tile + gridDim.x);   // (rows past the batch read id 0: never used)
  }
  __syncthreads();
  int cur = 0;
  for (; tile < ntiles; tile += gridDim.x, cur ^= 1) {
    const bool more = tile + gridDim.x < ntiles;
    if (more) {
      load_tile(tile + gridDim.x);
      load_ids(tile + 2 * (int64_t)gridDim.x);
    }
    const float* Gs = smem + cur * BUF;
    const float* Hs = Gs + TROWS * D;
    const float* Ps = Hs + TROWS * H;
    const float* Xs = Ps + TROWS * H;
    if (w < 4) {
      // dW2[d][hid] tiles (d-tile w, hidden tiles 0..3): A = gy columns, B = hid columns; k = row
      // operands of k-step s+1 are requested before the MFMAs of k-step s are issued (the compiler's own schedule
      // waited for every LDS read right before the two MFMAs that use it)
      float av[2], bv[2][4];
      av[0] = Gs[hh * D + w * 32 + r31];
#pragma unroll
      for (int t = 0; t < 4; ++t) bv[0][t] = Hs[hh * H + t * 32 + r31];
#pragma unroll
      for (int s = 0; s < TROWS / 2; ++s) {
        const int c = s & 1, nrow = 2 * (s + 1) + hh;
        if (s + 1 < TROWS / 2) {
          av[c ^ 1] = Gs[nrow * D + w * 32 + r31];
#pragma unroll
          for (int t = 0; t < 4; ++t) bv[c ^ 1][t] = Hs[nrow * H + t * 32 + r31];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = mfma32(av[c], bv[c][t], acc[t]);
        __builtin_amdgcn_sched_barrier(0);
      }
      // column sums: threads 0..127 -> db2 (gy), 128..255 -> db1 (dPre)
      const float* src = (tid < D) ? Gs + tid : Ps + (tid - D);
      constexpr int LDS_ = D;  // D == H
      float cs = 0.f;
#pragma unroll 8
      for (int r = 0; r < TROWS; ++r) cs += src[r * LDS_];
      colsum += cs;
    } else {
      // dW1[hid][k] tiles (hidden tile w-4, x tiles 0..NX-1): A = dPre columns, B = x columns
      float av[2], bv[2][NX];
      av[0] = Ps[hh * H + (w - 4) * 32 + r31];
#pragma unroll
      for (int t = 0; t < NX; ++t) bv[0][t] = Xs[hh * XW + t * 32 + r31];
#pragma unroll
      for (int s = 0; s < TROWS / 2; ++s) {
        const int c = s & 1, nrow = 2 * (s + 1) + hh;
        if (s + 1 < TROWS / 2) {
          av[c ^ 1] = Ps[nrow * H + (w - 4) * 32 + r31];
#pragma unroll
          for (int t = 0; t < NX; ++t) bv[c ^ 1][t] = Xs[nrow * XW + t * 32 + r31];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NX; ++t) acc[t] = mfma32(av[c], bv[c][t], acc[t]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (more) store_tile(smem + (cur ^ 1) * BUF);
    __syncthreads();
  }

  // ---- slab = [dW1 (H*K1) | db1 (H) | dW2 (D*H) | db2 (D)]
  constexpr int P = H * K1 + H + D * H + D;
  float* sl = a.slab + (size_t)blockIdx.x * P;
  if (w < 4) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) sl[H * K1 + H + (w * 32 + acc_row(r, lane)) * H + t * 32 + r31] = acc[t][r];
    if (tid < D) sl[H * K1 + H + D * H + tid] = colsum;
    else sl[H * K1 + (tid - D)] = colsum;
  } else {
#pragma unroll
    for (int t = 0; t < NX; ++t) {
      const int xc = t * 32 + r31;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (xc < K1) sl[((w - 4) * 32 + acc_row(r, lane)) * K1 + xc] = acc[t][r];
    }
  }
}

template <int D, int H, bool ITEM>
int launch_bwd2(const TowerBwdArgs& a, float* act, hipStream_t st, hipEvent_t dx_event) {
  constexpr int K1 = D + (ITEM ? 18 : 0);
  constexpr int XW = ((K1 + 31) / 32) * 32;
  float* gy = act;
  float* dpre = act + (size_t)a.B * D;
  {
    auto k = tower_bwd_data_kernel<D, H>;
    const size_t lds = ((size_t)H * (D + 4) + (size_t)H * (D + 1)) * sizeof(float);
    static bool granted = false;
    if (!granted) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); granted = true; }
    const int64_t nchunks = (a.B + 255) / 256;
    const int grid = (int)(nchunks < RIHIP_NCU ? nchunks : RIHIP_NCU);
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, st, a, gy, dpre, K1);
    if (dx_event) (void)hipEventRecord(dx_event, st);   // dX is complete: the row-gradient reduce may start beside the weight gradients
  }
  auto k = tower_wgrad_kernel<D, H, ITEM>;
  const size_t lds = 2 * (size_t)32 * (D + H + H + XW) * sizeof(float);
  static bool granted2 = false;
  if (!granted2) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); granted2 = true; }
  const int64_t ntiles = (a.B + 31) / 32;
  const int grid = (int)(ntiles < RIHIP_NCU ? ntiles : RIHIP_NCU);
  hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, st, a, (const float*)gy, (const float*)dpre);
  return grid;
}

}  // namespace

int rihip_launch_tower_bwd2(int d, int hidden, bool item, const TowerBwdArgs& a, float* act, hipStream_t st,
                            hipEvent_t dx_event) {
  if (d == 128 && hidden == 128)
    return item ? launch_bwd2<128, 128, true>(a, act, st, dx_event) : launch_bwd2<128, 128, false>(a, act, st, dx_event);
  return 0;
}

bool rihip_launch_tower_fwd2(int d, int hidden, bool item, const TowerFwdArgs& a, hipStream_t st) {
  if (d == 128 && hidden == 128) launch_dh<128, 128>(item, a, st);
  else if (d == 64 && hidden == 128) launch_dh<64, 128>(item, a, st);
  else if (d == 64 && hidden == 64) launch_dh<64, 64>(item, a, st);
  else if (d == 32 && hidden == 64) launch_dh<32, 64>(item, a, st);
  else if (d == 32 && hidden == 128) launch_dh<32, 128>(item, a, st);
  else return false;
  return true;
}
