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

}  // namespace

bool rihip_launch_tower_fwd2(int d, int hidden, bool item, const TowerFwdArgs& a, hipStream_t st) {
  if (d == 128 && hidden == 128) launch_dh<128, 128>(item, a, st);
  else if (d == 64 && hidden == 128) launch_dh<64, 128>(item, a, st);
  else if (d == 64 && hidden == 64) launch_dh<64, 64>(item, a, st);
  else if (d == 32 && hidden == 64) launch_dh<32, 64>(item, a, st);
  else if (d == 32 && hidden == 128) launch_dh<32, 128>(item, a, st);
  else return false;
  return true;
}
