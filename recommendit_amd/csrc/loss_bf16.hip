// In-batch BPR sweep on split-bf16 MFMA ("bf16x3"): every fp32 operand is split x = hi + lo (two bf16) and every
// product a.b is taken as hi.hi + hi.lo + lo.hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation -- relative
// product error ~2^-16, 5.3x fewer matrix-pipe cycles than the exact-f32 path (3 x 32 cycles per 16 k instead of
// 8 x 64).  Same algorithm, ownership and determinism as inbatch_sweep_kernel (loss.hip); optional precision mode
// (the exact-f32 kernel stays the default and the parity reference).
//
// Tile images in LDS (bf16, hi and lo planes): row-major [32 swept rows][d] for S^T = Y.Xo^T (A operand: 8
// consecutive k per lane, one ds_read_b128), and TRANSPOSED [d][32 rows] for dOwner += G^T.Y, whose contraction runs
// over the swept-row index (B operand: rows 16s+4h+{0..3} and 16s+8+4h+{0..3} of one column = two ds_read_b64,
// exactly the k-order in which the S^T accumulator registers 8s..8s+7 serve as the A operand).
#include "common.h"
#include "recommendit_hip.h"
#include "loss_sweep_args.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ void split2(float x, __bf16& hi, __bf16& lo) {
  hi = (__bf16)x;
  lo = (__bf16)(x - (float)hi);
}

template <int D, bool MODE_USER>
__global__ __launch_bounds__(256, 2) void inbatch_sweep_bf16_kernel(SweepArgs a) {
  constexpr int LDB = D + 8;      // row-major plane: bf16 per row (16-B pad => conflict-free b128 reads)
  constexpr int RLT = TSW + 4;    // transposed plane: bf16 per column (72-B columns, 8-B aligned)
  constexpr int KB = D / 16, CT = D / 32;
  constexpr int NT = 2 * D;       // staging threads: 16 row pairs x D/8 column groups
  __shared__ __attribute__((aligned(16))) __bf16 Yh[2][TSW * LDB];
  __shared__ __attribute__((aligned(16))) __bf16 Yl[2][TSW * LDB];
  __shared__ __attribute__((aligned(16))) __bf16 Th[2][D * RLT];
  __shared__ __attribute__((aligned(16))) __bf16 Tl[2][D * RLT];
  __shared__ float posS[2][TSW];
  __shared__ float rS[2][TSW];
  __shared__ float rsum[4][32];
  __shared__ double red_loss[4];

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int r31 = lane & 31, hh = lane >> 5;
  const int64_t o_base = (int64_t)blockIdx.x * OW + w * 32;
  const int64_t o_loc = o_base + r31;
  const bool o_ok = o_loc < a.No;
  const bool owners_full = (o_base + 32 <= a.No);

  // register-stationary owner fragments (B operand of S^T): lane (o, h) <- Xo[o][16kb + 8h + j]
  bf16x8 xo_h[KB], xo_l[KB];
  {
    const int64_t orow = o_ok ? o_loc : (a.No - 1);
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      // pre-scaled by log2(e): scores leave the MFMA chain in log2 units (sweep_elem)
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(&a.Xo[orow * D + kb * 16 + 8 * hh]) * RIHIP_LOG2E;
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(&a.Xo[orow * D + kb * 16 + 8 * hh + 4]) * RIHIP_LOG2E;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        __bf16 h0, l0, h1, l1;
        split2(v0[j], h0, l0);
        split2(v1[j], h1, l1);
        xo_h[kb][j] = h0; xo_l[kb][j] = l0; xo_h[kb][4 + j] = h1; xo_l[kb][4 + j] = l1;
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

  // staging: thread -> row pair rp (rows 2rp, 2rp+1) x 8 columns cg*8..cg*8+7
  const int rp = tid / (D / 8), cg = tid % (D / 8);
  f32x4 stage[4];
  float st_pos = 0.f, st_r = 0.f;
  auto load_tile = [&](int64_t tile) {
    const int64_t s_base = tile * TSW;
    if (tid < NT) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int64_t srow = s_base + 2 * rp + e;
        f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
        if (srow < a.Ns) {
          v0 = *reinterpret_cast<const f32x4*>(&a.Ys[srow * D + cg * 8]);
          v1 = *reinterpret_cast<const f32x4*>(&a.Ys[srow * D + cg * 8 + 4]);
        }
        stage[2 * e] = v0;
        stage[2 * e + 1] = v1;
      }
    }
    if (!MODE_USER && tid < TSW) {
      const int64_t srow = s_base + tid;
      st_pos = (srow < a.Ns) ? a.pos[srow] * RIHIP_LOG2E : 0.f;
      st_r = (srow < a.Ns) ? a.r_in[srow] : 0.f;
    }
  };
  auto store_tile = [&](int buf) {
    if (tid < NT) {
      bf16x8 h[2], l[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          __bf16 hi, lo;
          split2(stage[2 * e + (j >> 2)][j & 3], hi, lo);
          h[e][j] = hi;
          l[e][j] = lo;
        }
        *reinterpret_cast<bf16x8*>(&Yh[buf][(2 * rp + e) * LDB + cg * 8]) = h[e];
        *reinterpret_cast<bf16x8*>(&Yl[buf][(2 * rp + e) * LDB + cg * 8]) = l[e];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {  // transposed planes: one dword = rows (2rp, 2rp+1) of column cg*8+j
        bf16x2 ph = {h[0][j], h[1][j]}, pl = {l[0][j], l[1][j]};
        *reinterpret_cast<bf16x2*>(&Th[buf][(cg * 8 + j) * RLT + 2 * rp]) = ph;
        *reinterpret_cast<bf16x2*>(&Tl[buf][(cg * 8 + j) * RLT + 2 * rp]) = pl;
      }
    }
    if (!MODE_USER && tid < TSW) {
      posS[buf][tid] = st_pos;
      rS[buf][tid] = st_r;
    }
  };

  if (t0 < t1) {
    load_tile(t0);
    store_tile(0);
  } else {
    if (hh == 0) rsum[w][r31] = 0.f;
    if (MODE_USER && lane == 0) red_loss[w] = 0.0;
  }
  __syncthreads();

#pragma unroll 1
  for (int64_t tile = t0; tile < t1; ++tile) {
    const int cur = (int)((tile - t0) & 1);
    const int64_t s_base = tile * TSW;
    const bool more = (tile + 1 < t1);
    if (more) load_tile(tile + 1);

    // ---- S^T[s][o] = Y.Xo^T with hi.hi + hi.lo + lo.hi
    f32x16 st = zero16();
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      const bf16x8 ah = *reinterpret_cast<const bf16x8*>(&Yh[cur][r31 * LDB + kb * 16 + 8 * hh]);
      const bf16x8 al = *reinterpret_cast<const bf16x8*>(&Yl[cur][r31 * LDB + kb * 16 + 8 * hh]);
      st = mfma_bf16(ah, xo_h[kb], st);
      st = mfma_bf16(ah, xo_l[kb], st);
      st = mfma_bf16(al, xo_h[kb], st);
    }
    // ---- G = sigma(z) * c
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
        const float pos = MODE_USER ? pos_o : posS[cur][acc_row(r, lane)];
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
        const float pos = MODE_USER ? pos_o : posS[cur][sl];
        const float rd = MODE_USER ? 0.f : -rS[cur][sl] * inv_c;
        g[r] = sweep_elem<MODE_USER, false>(st[r], pos, valid, diag, rd, loss_acc, den_prod, r_acc);
        if (MODE_USER && (r & 7) == 7) {
          loss_acc += __builtin_amdgcn_logf(den_prod);
          den_prod = 1.f;
        }
      }
    }
    // ---- dOwner[o][c] += sum_s G[s][o] Y[s][c]: registers 8s..8s+7 are the A fragment of k-step s
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 gh, gl;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        __bf16 hi, lo;
        split2(g[8 * s + j], hi, lo);
        gh[j] = hi;
        gl[j] = lo;
      }
#pragma unroll
      for (int t = 0; t < CT; ++t) {
        const int col = t * 32 + r31;
        const bf16x4 bh0 = *reinterpret_cast<const bf16x4*>(&Th[cur][col * RLT + 16 * s + 4 * hh]);
        const bf16x4 bh1 = *reinterpret_cast<const bf16x4*>(&Th[cur][col * RLT + 16 * s + 8 + 4 * hh]);
        const bf16x4 bl0 = *reinterpret_cast<const bf16x4*>(&Tl[cur][col * RLT + 16 * s + 4 * hh]);
        const bf16x4 bl1 = *reinterpret_cast<const bf16x4*>(&Tl[cur][col * RLT + 16 * s + 8 + 4 * hh]);
        bf16x8 bh, bl;
#pragma unroll
        for (int j = 0; j < 4; ++j) { bh[j] = bh0[j]; bh[4 + j] = bh1[j]; bl[j] = bl0[j]; bl[4 + j] = bl1[j]; }
        out[t] = mfma_bf16(gh, bh, out[t]);
        out[t] = mfma_bf16(gh, bl, out[t]);
        out[t] = mfma_bf16(gl, bh, out[t]);
      }
    }
    if (more) store_tile(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue (identical to the f32 kernel)
  if (t0 < t1) {
    const float rr = (r_acc + __shfl_xor(r_acc, 32, 64)) * a.c;
    if (hh == 0) rsum[w][r31] = rr;
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
      const float rs = rsum[w][o];
#pragma unroll
      for (int t = 0; t < CT; ++t) {
        float v = out[t][r] * a.c;
        if (fix) v -= rs * a.Ys[drow * D + t * 32 + r31];
        dst[orow * D + t * 32 + r31] = v;
      }
    }
  }
  if (MODE_USER && hh == 0 && o_ok) {
    if (final_pass) a.r_out[o_loc] = rsum[w][r31];
    else a.r_part[(size_t)blockIdx.y * a.No + o_loc] = rsum[w][r31];
  }
  if (MODE_USER && tid == 0)
    a.loss_part[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = ((red_loss[0] + red_loss[1]) + red_loss[2]) + red_loss[3];
}

template <int D>
void launch_bf16(bool mode_user, const SweepArgs& a, dim3 grid, hipStream_t st) {
  if (mode_user) hipLaunchKernelGGL((inbatch_sweep_bf16_kernel<D, true>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((inbatch_sweep_bf16_kernel<D, false>), grid, dim3(256), 0, st, a);
}

}  // namespace

void rihip_launch_sweep_bf16x3(int d, bool mode_user, const SweepArgs& a, dim3 grid, hipStream_t st) {
  if (d == 32) launch_bf16<32>(mode_user, a, grid, st);
  else if (d == 64) launch_bf16<64>(mode_user, a, grid, st);
  else launch_bf16<128>(mode_user, a, grid, st);
}
