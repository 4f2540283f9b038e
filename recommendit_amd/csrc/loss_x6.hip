// In-batch BPR passes on split-bf16 MFMA with fp32-level accuracy ("bf16x6").
//
// Every fp32 operand is split EXACTLY into three bf16 pieces by truncation, x = h + m + l (8 + 8 + 8 significant bits
// = the 24 of fp32; each residual is exactly representable), and a product a.b is taken as the six partial products
//   ah.bh + ah.bm + am.bh + ah.bl + al.bh + am.bm
// on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  The dropped terms (am.bl, al.bm, al.bl) are <= 2^-23 |a||b|,
// i.e. at the rounding level of an fp32 product, so the results pass the SAME tolerances as the exact-f32 kernels
// (tests/test_gpu_towers.py) while the matrix pipe does 6 x 32 cycles per 16 k instead of 8 x 64 (2.7x fewer), and
// VALU work rides beside the bf16 MFMAs (it does not beside v_mfma_f32_32x32x2_f32, tools/mfma_peak).
//
// Same algorithm, ownership and determinism as loss.hip: user pass = score sweep + dU (+ the fp32 G store), item
// pass = G^T.U from the stored G.  The swept tile arrives as fp32 and is split inside the workgroup into
//   row planes  [3][32 rows][d+8]   (A operand of S^T: 8 consecutive k per lane, one ds_read_b128) and
//   col planes  [3][d][32 rows+8]   (B operand of dOwner += G^T.Y: the 8 swept rows of one k-step, contiguous after
//                                    a row permutation that matches the accumulator-register order; one ds_read_b128).
#include "common.h"
#include "recommendit_hip.h"
#include "loss_sweep_args.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x16 mfma_b(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ uint32_t fbits(float x) { return __builtin_bit_cast(uint32_t, x); }
__device__ __forceinline__ float bitsf(uint32_t x) { return __builtin_bit_cast(float, x); }
// upper halves of two fp32 bit patterns -> one dword {lo16 = e0 >> 16, hi16 = e1 >> 16}  (truncating bf16 pack)
__device__ __forceinline__ uint32_t pack_hi(uint32_t e1, uint32_t e0) { return __builtin_amdgcn_perm(e1, e0, 0x07060302u); }
// exact 3-way split of x: bit patterns whose upper halves are the bf16 pieces h, m, l
__device__ __forceinline__ void split3(float x, uint32_t& b, uint32_t& c, uint32_t& s) {
  b = fbits(x);
  const float r1 = x - bitsf(b & 0xFFFF0000u);
  c = fbits(r1);
  const float r2 = r1 - bitsf(c & 0xFFFF0000u);
  s = fbits(r2);
}
// the six partial products of one k-step
__device__ __forceinline__ f32x16 mfma6(const u32x4 (&a)[3], const u32x4 (&b)[3], f32x16 acc) {
  acc = mfma_b(a[2], b[0], acc);  // smallest terms first
  acc = mfma_b(a[0], b[2], acc);
  acc = mfma_b(a[1], b[1], acc);
  acc = mfma_b(a[1], b[0], acc);
  acc = mfma_b(a[0], b[1], acc);
  acc = mfma_b(a[0], b[0], acc);
  return acc;
}

// position of swept row `row` (0..31) inside a column of the col planes.  PERM (user/item sweep): the 8 rows that
// accumulator registers 8s..8s+7 of half hh hold (16s + 8(j>>2) + 4hh + (j&3)) are contiguous at 16s + 8hh + j.
// !PERM (item pass from stored G): natural order (lane half hh, k-step s <-> users 16hh + 8s + j).
template <bool PERM>
__device__ __forceinline__ int colpos(int row) {
  if (!PERM) return row;
  const int s = row >> 4, rem = row & 15;
  return s * 16 + ((rem >> 2) & 1) * 8 + ((rem >> 3) << 2) + (rem & 3);
}

template <int D>
struct Planes {
  static constexpr int LDR = D + 8;         // bf16 per row of a row plane (16-B pad: conflict-free b128 reads)
  static constexpr int LDC = TSW + 8;       // bf16 per column of a col plane (80-B pitch: conflict-free b128 reads)
  static constexpr int ROWP = TSW * LDR;    // bf16 per row plane
  static constexpr int COLP = D * LDC;      // bf16 per col plane
};

// Split a staged fp32 tile into LDS planes.  Thread -> row pair rp (rows 2rp, 2rp+1) x CW consecutive columns; the row
// pair is the fastest index across lanes, so the dword writes into the col planes (one column each, 80-B pitch) land
// on 16 consecutive dwords x 4 columns = 64 distinct banks per wave.
template <int D, int NT, bool ROWS, bool PERM>
struct TileStager {
  static constexpr int CW = (TSW * D) / (2 * NT) >= 4 ? (TSW * D) / (2 * NT) : 4;  // columns per thread (4 or 8)
  static constexpr int NACT = (TSW / 2) * (D / CW);                                 // active staging threads
  static constexpr int NV = CW / 4;                                                 // float4 per row per thread
  f32x4 v[2][NV];
  __device__ __forceinline__ void load(const float* __restrict__ Ys, int64_t Ns, int64_t s_base, int tid) {
    if (tid < NACT) {
      const int rp = tid % (TSW / 2), cg = tid / (TSW / 2);
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int64_t srow = s_base + 2 * rp + e;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          f32x4 x = {0.f, 0.f, 0.f, 0.f};
          if (srow < Ns) x = *reinterpret_cast<const f32x4*>(&Ys[srow * D + cg * CW + 4 * i]);
          v[e][i] = x;
        }
      }
    }
  }
  __device__ __forceinline__ void store(__bf16* rowp, __bf16* colp, int tid) const {
    using P = Planes<D>;
    if (tid < NACT) {
      const int rp = tid % (TSW / 2), cg = tid / (TSW / 2);
      uint32_t b[2][CW], c[2][CW], s[2][CW];
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int j = 0; j < CW; ++j) split3(v[e][j >> 2][j & 3], b[e][j], c[e][j], s[e][j]);
      if (ROWS) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
#pragma unroll
          for (int q = 0; q < CW / 4; ++q) {
            const int off = (2 * rp + e) * P::LDR + cg * CW + 4 * q;
            const u32x2 wh = {pack_hi(b[e][4 * q + 1], b[e][4 * q]), pack_hi(b[e][4 * q + 3], b[e][4 * q + 2])};
            const u32x2 wm = {pack_hi(c[e][4 * q + 1], c[e][4 * q]), pack_hi(c[e][4 * q + 3], c[e][4 * q + 2])};
            const u32x2 wl = {pack_hi(s[e][4 * q + 1], s[e][4 * q]), pack_hi(s[e][4 * q + 3], s[e][4 * q + 2])};
            *reinterpret_cast<u32x2*>(&rowp[0 * P::ROWP + off]) = wh;
            *reinterpret_cast<u32x2*>(&rowp[1 * P::ROWP + off]) = wm;
            *reinterpret_cast<u32x2*>(&rowp[2 * P::ROWP + off]) = wl;
          }
        }
      }
      const int cp = colpos<PERM>(2 * rp);  // rows 2rp and 2rp+1 are adjacent in both orders
#pragma unroll
      for (int j = 0; j < CW; ++j) {
        const int off = (cg * CW + j) * P::LDC + cp;
        *reinterpret_cast<uint32_t*>(&colp[0 * P::COLP + off]) = pack_hi(b[1][j], b[0][j]);
        *reinterpret_cast<uint32_t*>(&colp[1 * P::COLP + off]) = pack_hi(c[1][j], c[0][j]);
        *reinterpret_cast<uint32_t*>(&colp[2 * P::COLP + off]) = pack_hi(s[1][j], s[0][j]);
      }
    }
  }
};

// 16 fp32 values (accumulator-register order) -> the A fragments of the two k-steps, three planes
__device__ __forceinline__ void split_regs(const float (&g)[16], u32x4 (&a0)[3], u32x4 (&a1)[3]) {
  uint32_t b[16], c[16], s[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) split3(g[r], b[r], c[r], s[r]);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    a0[0][q] = pack_hi(b[2 * q + 1], b[2 * q]);
    a0[1][q] = pack_hi(c[2 * q + 1], c[2 * q]);
    a0[2][q] = pack_hi(s[2 * q + 1], s[2 * q]);
    a1[0][q] = pack_hi(b[8 + 2 * q + 1], b[8 + 2 * q]);
    a1[1][q] = pack_hi(c[8 + 2 * q + 1], c[8 + 2 * q]);
    a1[2][q] = pack_hi(s[8 + 2 * q + 1], s[8 + 2 * q]);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// score sweep (user mode: + loss, r, optional G store; item mode: recompute form)
// ------------------------------------------------------------------------------------------------------------------
template <int D, bool MODE_USER, bool GOUT, int NW>
__global__ __launch_bounds__(NW * 64, 2) void inbatch_sweep_x6_kernel(SweepArgs a) {
  using P = Planes<D>;
  constexpr int NT = NW * 64;
  constexpr int KB = D / 16, CT = D / 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16* rowp = reinterpret_cast<__bf16*>(smem_raw);                    // [2][3][ROWP]
  __bf16* colp = rowp + 2 * 3 * P::ROWP;                                 // [2][3][COLP]
  float* posS = reinterpret_cast<float*>(colp + 2 * 3 * P::COLP);        // [2][TSW]
  float* rS = posS + 2 * TSW;                                            // [2][TSW]
  float* rsum = rS + 2 * TSW;                                            // [NW][32]
  double* red_loss = reinterpret_cast<double*>(rsum + NW * 32);          // [NW]

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int r31 = lane & 31, hh = lane >> 5;
  const int64_t o_base = (int64_t)blockIdx.x * (NW * 32) + w * 32;
  const int64_t o_loc = o_base + r31;
  const bool o_ok = o_loc < a.No;
  const bool owners_full = (o_base + 32 <= a.No);

  // register-stationary owner fragments (B operand of S^T), pre-scaled by log2(e): lane (o, h) <- Xo[o][16kb + 8h + j]
  u32x4 xo[KB][3];
  {
    const int64_t orow = o_ok ? o_loc : (a.No - 1);
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(&a.Xo[orow * D + kb * 16 + 8 * hh]) * RIHIP_LOG2E;
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(&a.Xo[orow * D + kb * 16 + 8 * hh + 4]) * RIHIP_LOG2E;
      uint32_t b[8], c[8], s[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        split3(v0[j], b[j], c[j], s[j]);
        split3(v1[j], b[4 + j], c[4 + j], s[4 + j]);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        xo[kb][0][q] = pack_hi(b[2 * q + 1], b[2 * q]);
        xo[kb][1][q] = pack_hi(c[2 * q + 1], c[2 * q]);
        xo[kb][2][q] = pack_hi(s[2 * q + 1], s[2 * q]);
      }
    }
  }
  const float pos_o = (MODE_USER && o_ok) ? a.pos[o_loc] * RIHIP_LOG2E : 0.f;
  const float inv_c = 1.f / a.c;

  f32x16 out[CT];
#pragma unroll
  for (int t = 0; t < CT; ++t) out[t] = zero16();
  float r_acc = 0.f, loss_acc = 0.f;

  const int64_t ntiles = (a.Ns + TSW - 1) / TSW;
  const int64_t per = (ntiles + a.nsplit - 1) / a.nsplit;
  const int64_t t0 = (int64_t)blockIdx.y * per;
  const int64_t t1 = (t0 + per < ntiles) ? t0 + per : ntiles;

  TileStager<D, NT, true, true> stg;
  float st_pos = 0.f, st_r = 0.f;
  auto load_tile = [&](int64_t tile) {
    stg.load(a.Ys, a.Ns, tile * TSW, tid);
    if (!MODE_USER && tid < TSW) {
      const int64_t srow = tile * TSW + tid;
      st_pos = (srow < a.Ns) ? a.pos[srow] * RIHIP_LOG2E : 0.f;
      st_r = (srow < a.Ns) ? a.r_in[srow] : 0.f;
    }
  };
  auto store_tile = [&](int buf) {
    stg.store(rowp + buf * 3 * P::ROWP, colp + buf * 3 * P::COLP, tid);
    if (!MODE_USER && tid < TSW) {
      posS[buf * TSW + tid] = st_pos;
      rS[buf * TSW + tid] = st_r;
    }
  };

  if (t0 < t1) {
    load_tile(t0);
    store_tile(0);
  } else {
    if (hh == 0) rsum[w * 32 + r31] = 0.f;
    if (MODE_USER && lane == 0) red_loss[w] = 0.0;
  }
  __syncthreads();

#pragma unroll 1
  for (int64_t tile = t0; tile < t1; ++tile) {
    const int cur = (int)((tile - t0) & 1);
    const int64_t s_base = tile * TSW;
    const bool more = (tile + 1 < t1);
    if (more) load_tile(tile + 1);
    const __bf16* rp_c = rowp + cur * 3 * P::ROWP;
    const __bf16* cp_c = colp + cur * 3 * P::COLP;

    // ---- S^T[s][o] = Y.Xo^T
    f32x16 st = zero16();
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      u32x4 av[3];
#pragma unroll
      for (int p = 0; p < 3; ++p)
        av[p] = *reinterpret_cast<const u32x4*>(&rp_c[p * P::ROWP + r31 * P::LDR + kb * 16 + 8 * hh]);
      st = mfma6(av, xo[kb], st);
    }
    // ---- weights sigma(z)
    const int64_t sg0 = a.s_goff + s_base, og0 = a.o_goff + o_base;
    const bool slow = !(owners_full && (s_base + TSW <= a.Ns)) || (sg0 < og0 + 32 && og0 < sg0 + TSW);
    const int64_t dd = og0 - sg0;
    const int ddi = (dd > -64 && dd < 64) ? (int)dd : 1000;
    const int64_t left = a.Ns - s_base;
    const int n_valid = left < TSW ? (int)left : TSW;
    float g[16];
    float den_prod = 1.f;
    if (!slow) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pos = MODE_USER ? pos_o : posS[cur * TSW + acc_row(r, lane)];
        g[r] = sweep_elem<MODE_USER, true>(st[r], pos, true, false, 0.f, loss_acc, den_prod, r_acc);
        if (MODE_USER && (r & 7) == 7) {
          loss_acc += __builtin_amdgcn_logf(den_prod);
          den_prod = 1.f;
        }
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int sl = acc_row(r, lane);
        const bool valid = o_ok && (sl < n_valid);
        const bool diag = (sl - r31 == ddi);
        const float pos = MODE_USER ? pos_o : posS[cur * TSW + sl];
        const float rd = MODE_USER ? 0.f : -rS[cur * TSW + sl] * inv_c;
        g[r] = sweep_elem<MODE_USER, false>(st[r], pos, valid, diag, rd, loss_acc, den_prod, r_acc);
        if (MODE_USER && (r & 7) == 7) {
          loss_acc += __builtin_amdgcn_logf(den_prod);
          den_prod = 1.f;
        }
      }
    }
    if (GOUT) {  // fp32 weights, 32x32-blocked G^T (same layout as loss.hip: two full 128-B lines per wave store)
      float* gp = a.gmat + ((size_t)tile * a.g_ub + (size_t)blockIdx.x * NW + w) * 1024 + (4 * hh) * 32 + r31;
#pragma unroll
      for (int r = 0; r < 16; ++r) gp[((r & 3) + 8 * (r >> 2)) * 32] = g[r];
    }
    // ---- dOwner[o][c] += sum_s G[s][o] Y[s][c]: registers 8s..8s+7 are the A fragment of k-step s
    u32x4 ga[2][3];
    split_regs(g, ga[0], ga[1]);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int t = 0; t < CT; ++t) {
        u32x4 bv[3];
#pragma unroll
        for (int p = 0; p < 3; ++p)
          bv[p] = *reinterpret_cast<const u32x4*>(&cp_c[p * P::COLP + (t * 32 + r31) * P::LDC + 16 * s + 8 * hh]);
        out[t] = mfma6(ga[s], bv, out[t]);
      }
    }
    if (more) store_tile(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue (as in loss.hip)
  if (t0 < t1) {
    const float rr = (r_acc + __shfl_xor(r_acc, 32, 64)) * a.c;
    if (hh == 0) rsum[w * 32 + r31] = rr;
    if (MODE_USER) {
      const float ls = wave_sum(loss_acc);
      if (lane == 0) red_loss[w] = (double)ls * (double)RIHIP_LN2;
    }
  }
  __syncthreads();
  const bool final_pass = (a.nsplit == 1);
  float* dst = final_pass ? a.dOwner : a.slab + (size_t)blockIdx.y * a.No * D;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int o = acc_row(r, lane);
    const int64_t orow = o_base + o;
    if (orow < a.No) {
      const int64_t drow = a.o_goff + orow - a.s_goff;
      const bool fix = final_pass && MODE_USER && drow >= 0 && drow < a.Ns;
      const float rs = rsum[w * 32 + o];
#pragma unroll
      for (int t = 0; t < CT; ++t) {
        float v = out[t][r] * a.c;
        if (fix) v -= rs * a.Ys[drow * D + t * 32 + r31];
        dst[orow * D + t * 32 + r31] = v;
      }
    }
  }
  if (MODE_USER && hh == 0 && o_ok) {
    if (final_pass) a.r_out[o_loc] = rsum[w * 32 + r31];
    else a.r_part[(size_t)blockIdx.y * a.No + o_loc] = rsum[w * 32 + r31];
  }
  if (MODE_USER && tid < NW / 4) {
    const int64_t gx128 = (a.No + OW - 1) / OW;
    const int64_t slot = (int64_t)blockIdx.x * (NW / 4) + tid;
    if (slot < gx128)
      a.loss_part[(size_t)blockIdx.y * gx128 + slot] =
          ((red_loss[4 * tid] + red_loss[4 * tid + 1]) + red_loss[4 * tid + 2]) + red_loss[4 * tid + 3];
  }
}

template <int D>
constexpr size_t sweep_x6_lds(int nw) {
  return 2 * 3 * (size_t)(Planes<D>::ROWP + Planes<D>::COLP) * 2 + (4 * TSW + (size_t)nw * 32) * 4 + (size_t)nw * 8 + 16;
}

// ------------------------------------------------------------------------------------------------------------------
// item pass from the stored fp32 G:  dI[j][:] = c * sum_i G[i][j] U[i][:] - r_j U_j
// ------------------------------------------------------------------------------------------------------------------
template <int D, int NW>
__global__ __launch_bounds__(NW * 64, 2) void inbatch_gt_x6_kernel(SweepArgs a) {
  using P = Planes<D>;
  constexpr int NT = NW * 64;
  constexpr int CT = D / 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16* colp = reinterpret_cast<__bf16*>(smem_raw);  // [2][3][COLP]

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int r31 = lane & 31, hh = lane >> 5;
  const int64_t o_base = (int64_t)blockIdx.x * (NW * 32) + w * 32;

  f32x16 out[CT];
#pragma unroll
  for (int t = 0; t < CT; ++t) out[t] = zero16();

  const int64_t ntiles = (a.Ns + TSW - 1) / TSW;
  const int64_t per = (ntiles + a.nsplit - 1) / a.nsplit;
  const int64_t t0 = (int64_t)blockIdx.y * per;
  const int64_t t1 = (t0 + per < ntiles) ? t0 + per : ntiles;

  TileStager<D, NT, false, false> stg;
  const f32x4* gp = reinterpret_cast<const f32x4*>(a.gmat + ((size_t)(blockIdx.x * NW + w) * a.g_ub) * 1024 +
                                                   r31 * 32 + 16 * hh);
  // G blocks are requested NPF tiles ahead (16 registers per tile in flight): the per-tile compute is short here
  // (48 bf16 MFMAs), so one tile of lookahead does not cover the HBM latency of the 2+ TB/s stream.
  constexpr int NPF = 4;
  f32x4 gq[NPF][4];  // lane (item, hh): users 16hh .. 16hh+15 of tiles tile .. tile+NPF-1 (slot 0 = current)
#pragma unroll
  for (int f = 0; f < NPF; ++f)
#pragma unroll
    for (int q = 0; q < 4; ++q) gq[f][q] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (t0 < t1) {
#pragma unroll
    for (int f = 0; f < NPF; ++f)
      if (t0 + f < t1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) gq[f][q] = gp[(size_t)(t0 + f) * 256 + q];
      }
    stg.load(a.Ys, a.Ns, t0 * TSW, tid);
    stg.store(nullptr, colp, tid);
  }
  __syncthreads();

#pragma unroll 1
  for (int64_t tile = t0; tile < t1; ++tile) {
    const int cur = (int)((tile - t0) & 1);
    const bool more = (tile + 1 < t1);
    if (more) stg.load(a.Ys, a.Ns, (tile + 1) * TSW, tid);
    const __bf16* cp_c = colp + cur * 3 * P::COLP;
    float g[16];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int s = 0; s < 4; ++s) g[4 * q + s] = gq[0][q][s];
    // rotate the ring and request tile + NPF
#pragma unroll
    for (int f = 0; f + 1 < NPF; ++f)
#pragma unroll
      for (int q = 0; q < 4; ++q) gq[f][q] = gq[f + 1][q];
    if (tile + NPF < t1) {
#pragma unroll
      for (int q = 0; q < 4; ++q) gq[NPF - 1][q] = gp[(size_t)(tile + NPF) * 256 + q];
    }
    u32x4 ga[2][3];  // k-step s <-> users 16hh + 8s + j
    split_regs(g, ga[0], ga[1]);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int t = 0; t < CT; ++t) {
        u32x4 bv[3];
#pragma unroll
        for (int p = 0; p < 3; ++p)
          bv[p] = *reinterpret_cast<const u32x4*>(&cp_c[p * P::COLP + (t * 32 + r31) * P::LDC + 16 * hh + 8 * s]);
        out[t] = mfma6(ga[s], bv, out[t]);
      }
    }
    if (more) stg.store(nullptr, colp + (cur ^ 1) * 3 * P::COLP, tid);
    __syncthreads();
  }

  const bool final_pass = (a.nsplit == 1);
  float* dst = final_pass ? a.dOwner : a.slab + (size_t)blockIdx.y * a.No * D;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int64_t orow = o_base + acc_row(r, lane);
    if (orow < a.No) {
      const int64_t drow = a.o_goff + orow - a.s_goff;
      const bool fix = final_pass && drow >= 0 && drow < a.Ns;
      const float rs = fix ? a.r_in[drow] : 0.f;
#pragma unroll
      for (int t = 0; t < CT; ++t) {
        float v = out[t][r] * a.c;
        if (fix) v -= rs * a.Ys[drow * D + t * 32 + r31];
        dst[orow * D + t * 32 + r31] = v;
      }
    }
  }
}

template <int D>
constexpr size_t gt_x6_lds() { return 2 * 3 * (size_t)Planes<D>::COLP * 2 + 16; }

// > 64 KB of dynamic LDS must be granted per kernel once (the flag lives in the caller: one per instantiation)
template <typename K>
void grant_lds(K kernel, size_t bytes, bool& done) {
  if (!done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    done = true;
  }
}

template <int D, bool MODE_USER, bool GOUT, int NW>
void launch_one(const SweepArgs& a, dim3 grid, hipStream_t st) {
  auto k = inbatch_sweep_x6_kernel<D, MODE_USER, GOUT, NW>;
  const size_t lds = sweep_x6_lds<D>(NW);
  static bool granted = false;
  grant_lds(k, lds, granted);
  hipLaunchKernelGGL(k, grid, dim3(NW * 64), lds, st, a);
}
template <int D, int NW>
void launch_sweep_nw(bool mode_user, const SweepArgs& a, dim3 grid, hipStream_t st) {
  if (mode_user && a.gmat) launch_one<D, true, true, NW>(a, grid, st);
  else if (mode_user) launch_one<D, true, false, NW>(a, grid, st);
  else launch_one<D, false, false, NW>(a, grid, st);
}
template <int D>
void launch_sweep_d(bool mode_user, const SweepArgs& a, dim3 grid, int nw, hipStream_t st) {
  if (nw == 8) launch_sweep_nw<D, 8>(mode_user, a, grid, st);
  else launch_sweep_nw<D, 4>(mode_user, a, grid, st);
}
template <int D, int NW>
void launch_gt_one(const SweepArgs& a, dim3 grid, hipStream_t st) {
  auto k = inbatch_gt_x6_kernel<D, NW>;
  static bool granted = false;
  grant_lds(k, gt_x6_lds<D>(), granted);
  hipLaunchKernelGGL(k, grid, dim3(NW * 64), gt_x6_lds<D>(), st, a);
}
template <int D>
void launch_gt_d(const SweepArgs& a, dim3 grid, int nw, hipStream_t st) {
  if (nw == 8) launch_gt_one<D, 8>(a, grid, st);
  else launch_gt_one<D, 4>(a, grid, st);
}

}  // namespace

void rihip_launch_sweep_x6(int d, bool mode_user, const SweepArgs& a, dim3 grid, int nw, hipStream_t st) {
  if (d == 32) launch_sweep_d<32>(mode_user, a, grid, nw, st);
  else if (d == 64) launch_sweep_d<64>(mode_user, a, grid, nw, st);
  else launch_sweep_d<128>(mode_user, a, grid, nw, st);
}
void rihip_launch_gt_x6(int d, const SweepArgs& a, dim3 grid, int nw, hipStream_t st) {
  if (d == 32) launch_gt_d<32>(a, grid, nw, st);
  else if (d == 64) launch_gt_d<64>(a, grid, nw, st);
  else launch_gt_d<128>(a, grid, nw, st);
}
