// BPR losses for gfx950.
//  * bpr_pair:   sampled-negative BPR (reference src/models/two_tower.py:117-130), loss + grads fused.
//  * inbatch sweep: in-batch-negative BPR (reference src/models/two_tower.py:132-160) as a
//    flash-style fused score/loss/gradient sweep on exact-f32 MFMA.  The B x B score matrix is
//    never materialised: a workgroup owns 32 "owner" rows (users in MODE_USER, items in
//    MODE_ITEM), keeps their embedding fragments register-stationary, and sweeps tiles of the
//    other side through LDS.  For each 32x32 tile:  S^T = Y.Xo^T (MFMA) -> sigma/softplus (VALU)
//    -> dOwner += G^T-as-A-operand . Y (MFMA; the accumulator registers ARE the next A operand,
//    no LDS round trip).  Owner gradients never cross workgroups => no atomics, bitwise
//    reproducible.  dU and dI come from two launches of the same kernel with roles swapped
//    (S is recomputed; 8 B^2 d FLOP instead of 6 B^2 d, but zero atomic traffic).
#include <stdlib.h>
#include "common.h"
#include "recommendit_hip.h"
#include "loss_sweep_args.h"

void rihip_launch_sweep_bf16x3(int d, bool mode_user, const SweepArgs& a, dim3 grid, hipStream_t st);
void rihip_launch_sweep_x6(int d, bool mode_user, const SweepArgs& a, dim3 grid, int nw, hipStream_t st);
// loss_generic.hip: runtime-width sweep (any embed_dim multiple of 16 up to 256) behind the tuned instantiations
bool rihip_inbatch_generic_ok(int d);
void rihip_launch_sweep_generic(int d, bool mode_user, const SweepArgs& a, int nsplit_slots, hipStream_t st);
void rihip_launch_gt_x6(int d, const SweepArgs& a, dim3 grid, int nw, hipStream_t st);

namespace {

// ------------------------------------------------------------------------------------------
// sampled-negative BPR: one wave per row
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bpr_pair_kernel(const float* __restrict__ U, const float* __restrict__ P,
                                                       const float* __restrict__ N, int64_t B, int d, float inv_B,
                                                       float* dU, float* dP, float* dN, double* part) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  double acc = 0.0;
  for (int64_t b = (int64_t)blockIdx.x * 4 + w; b < B; b += (int64_t)gridDim.x * 4) {
    float sp = 0.f, sn = 0.f;
    for (int c = lane; c < d; c += 64) {
      const float u = U[b * d + c];
      sp += u * P[b * d + c];
      sn += u * N[b * d + c];
    }
    sp = wave_sum(sp);
    sn = wave_sum(sn);
    const float delta = sp - sn;
    // -logsigmoid(delta) = softplus(-delta); d/d delta = -sigmoid(-delta)
    const float e = expf(-fabsf(delta));
    const float loss = fmaxf(-delta, 0.f) + log1pf(e);
    const float sig_neg = (delta >= 0.f) ? e / (1.f + e) : 1.f / (1.f + e);  // sigmoid(-delta)
    const float wgt = -sig_neg * inv_B;
    for (int c = lane; c < d; c += 64) {
      const float u = U[b * d + c], p = P[b * d + c], n = N[b * d + c];
      dU[b * d + c] = wgt * (p - n);
      dP[b * d + c] = wgt * u;
      dN[b * d + c] = -wgt * u;
    }
    if (lane == 0) acc += (double)loss;
  }
  __shared__ double sh[4];
  if (lane == 0) sh[w] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

// float4 form for d in {32, 64, 128}: LPR = d/4 lanes per row, 64/LPR rows per wave side by side
template <int LPR>
__global__ __launch_bounds__(256) void bpr_pair_v4_kernel(const float* __restrict__ U, const float* __restrict__ P,
                                                          const float* __restrict__ N, int64_t B, float inv_B,
                                                          float* dU, float* dP, float* dN, double* part) {
  constexpr int RPW = 64 / LPR, d = LPR * 4;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int sub = lane / LPR, c4 = lane % LPR;
  double acc = 0.0;
  for (int64_t b0 = ((int64_t)blockIdx.x * 4 + w) * RPW; b0 < B; b0 += (int64_t)gridDim.x * 4 * RPW) {
    const int64_t b = b0 + sub;
    const bool ok = b < B;
    const int64_t bc = ok ? b : B - 1;
    const f32x4 u = reinterpret_cast<const f32x4*>(U + bc * d)[c4];
    const f32x4 p = reinterpret_cast<const f32x4*>(P + bc * d)[c4];
    const f32x4 n = reinterpret_cast<const f32x4*>(N + bc * d)[c4];
    float sp = (u.x * p.x + u.y * p.y) + (u.z * p.z + u.w * p.w);
    float sn = (u.x * n.x + u.y * n.y) + (u.z * n.z + u.w * n.w);
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) {
      sp += __shfl_xor(sp, o, 64);
      sn += __shfl_xor(sn, o, 64);
    }
    const float delta = sp - sn;
    const float e = expf(-fabsf(delta));
    const float loss = fmaxf(-delta, 0.f) + log1pf(e);
    const float sig_neg = (delta >= 0.f) ? e / (1.f + e) : 1.f / (1.f + e);  // sigmoid(-delta)
    const float wgt = -sig_neg * inv_B;
    if (ok) {
      reinterpret_cast<f32x4*>(dU + b * d)[c4] = f32x4{wgt * (p.x - n.x), wgt * (p.y - n.y), wgt * (p.z - n.z), wgt * (p.w - n.w)};
      reinterpret_cast<f32x4*>(dP + b * d)[c4] = f32x4{wgt * u.x, wgt * u.y, wgt * u.z, wgt * u.w};
      reinterpret_cast<f32x4*>(dN + b * d)[c4] = f32x4{-wgt * u.x, -wgt * u.y, -wgt * u.z, -wgt * u.w};
      if (c4 == 0) acc += (double)loss;
    }
  }
  acc = wave_sum_d(acc);
  __shared__ double sh[4];
  if (lane == 0) sh[w] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

// loss = scale * sum(part[0..n))  -- single wave, fixed order => deterministic
__global__ void finalize_sum_kernel(const double* __restrict__ part, int n, double scale, float* out) {
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 64) s += part[i];
  s = wave_sum_d(s);
  if (threadIdx.x == 0) *out = (float)(s * scale);
}

// pos[i] = U[i] . I[i + off]
__global__ __launch_bounds__(256) void rowdot_kernel(const float* __restrict__ U, const float* __restrict__ I,
                                                     int64_t B, int64_t off, int d, float* pos) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int64_t b = (int64_t)blockIdx.x * 4 + w; b < B; b += (int64_t)gridDim.x * 4) {
    float s = 0.f;
    for (int c = lane; c < d; c += 64) s += U[b * d + c] * I[(b + off) * d + c];
    s = wave_sum(s);
    if (lane == 0) pos[b] = s;
  }
}

// ------------------------------------------------------------------------------------------
// in-batch sweep
// ------------------------------------------------------------------------------------------
// GOUT (user mode only): additionally write the G tile (diagonal and ragged entries = 0) to a.gmat, so that the item
// gradients come from a plain G^T.U product (inbatch_gt_kernel) instead of a second score sweep.
template <int D, bool MODE_USER, bool GOUT = false, int NW = 4>
__global__ __launch_bounds__(NW * 64, 2) void inbatch_sweep_kernel(SweepArgs a) {
  constexpr int NT = NW * 64;  // NW = 8: one 512-thread workgroup per CU, half the swept-tile traffic per owner
  static_assert(!GOUT || MODE_USER, "G is stored by the user-mode sweep");
  constexpr int LDY = D + 4;
  constexpr int KB = D / 8, CT = D / 32;
  constexpr int EPK = 16 / KB > 0 ? 16 / KB : 1;   // score elements processed per k-block of the next S chain
  constexpr int NV = (TSW * (D / 4) + NT - 1) / NT;  // float4 staged per thread per tile
  static_assert(KB <= 16, "embed_dim <= 128");
  __shared__ __attribute__((aligned(16))) float Ysh[3][TSW * LDY];
  __shared__ float posS[3][TSW];
  __shared__ float rS[3][TSW];
  __shared__ float rsum[NW][32];
  __shared__ double red_loss[NW];

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int r31 = lane & 31, hh = lane >> 5;
  const int64_t o_base = (int64_t)blockIdx.x * (NW * 32) + w * 32;
  const int64_t o_loc = o_base + r31;  // this lane's owner (S^T accumulator column)
  const bool o_ok = o_loc < a.No;
  const int64_t o_gidx = a.o_goff + o_loc;
  const bool owners_full = (o_base + 32 <= a.No);

  // register-stationary owner fragments: B[k][n=o] = Xo[o][k]
  f32x4 xo[KB];
  {
    const int64_t orow = o_ok ? o_loc : (a.No - 1);
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      xo[kb] = *reinterpret_cast<const f32x4*>(&a.Xo[orow * D + kb * 8 + 4 * hh]);
      xo[kb] *= RIHIP_LOG2E;  // scores come out of the MFMA chain in log2 units (sweep_elem)
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

  f32x4 stage[NV];
  float st_pos = 0.f, st_r = 0.f;
  auto load_tile = [&](int64_t tile) {
    const int64_t s_base = tile * TSW;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = tid + i * NT;
      const int r = idx / (D / 4), c4 = idx % (D / 4);
      const int64_t srow = s_base + r;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (idx < TSW * (D / 4) && srow < a.Ns) v = reinterpret_cast<const f32x4*>(a.Ys + srow * D)[c4];
      stage[i] = v;
    }
    if (!MODE_USER && tid < TSW) {
      const int64_t srow = s_base + tid;
      st_pos = (srow < a.Ns) ? a.pos[srow] * RIHIP_LOG2E : 0.f;
      st_r = (srow < a.Ns) ? a.r_in[srow] : 0.f;
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = tid + i * NT;
      const int r = idx / (D / 4), c4 = idx % (D / 4);
      if (idx < TSW * (D / 4)) *reinterpret_cast<f32x4*>(&Ysh[buf][r * LDY + c4 * 4]) = stage[i];
    }
    if (!MODE_USER && tid < TSW) {
      posS[buf][tid] = st_pos;
      rS[buf][tid] = st_r;
    }
  };

  if (t0 >= t1) {  // empty split (can only happen for degenerate splits): contribute zeros
    if (hh == 0) rsum[w][r31] = 0.f;
    if (MODE_USER && lane == 0) red_loss[w] = 0.0;
  }
  f32x16 st = zero16();
  if (t0 < t1) {
    load_tile(t0);
    store_tile(0);
    if (t0 + 1 < t1) {
      load_tile(t0 + 1);
      store_tile(1);
    }
    __syncthreads();
    const float* Y0 = Ysh[0];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {  // S^T of the first tile (not overlapped)
      const f32x4 av = *reinterpret_cast<const f32x4*>(&Y0[r31 * LDY + kb * 8 + 4 * hh]);
      st = mfma32(av.x, xo[kb].x, st);
      st = mfma32(av.y, xo[kb].y, st);
      st = mfma32(av.z, xo[kb].z, st);
      st = mfma32(av.w, xo[kb].w, st);
    }
  }

#pragma unroll 1
  for (int64_t tile = t0; tile < t1; ++tile) {
    const int it = (int)((tile - t0) % 3);
    const int cur = it, nxt = (it + 1) % 3, pre = (it + 2) % 3;
    const int64_t s_base = tile * TSW;
    const bool has_next = (tile + 1 < t1);
    const bool has_pre = (tile + 2 < t1);
    if (has_pre) load_tile(tile + 2);  // global loads stay in flight under the MFMA blocks below

    // does this tile touch the diagonal of this wave's owners or a ragged edge?
    const int64_t sg0 = a.s_goff + s_base, og0 = a.o_goff + o_base;
    const bool slow = !(owners_full && (s_base + TSW <= a.Ns)) || (sg0 < og0 + 32 && og0 < sg0 + TSW);
    // 32-bit forms of the per-element tests (only evaluated on slow tiles)
    const int64_t dd = og0 - sg0;                                        // diag  <=>  sl - r31 == dd
    const int ddi = (dd > -64 && dd < 64) ? (int)dd : 1000;
    const int64_t left = a.Ns - s_base;
    const int n_valid = left < TSW ? (int)left : TSW;                    // valid <=>  sl < n_valid (and o_ok)
    const float* Yn = Ysh[nxt];
    const float* Yc = Ysh[cur];
    f32x16 sn = zero16();
    float g[16];
    float den_prod = 1.f;  // running product of (1 + e^-z): one log per eight elements
    if (!slow) {
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        {  // unconditional: on the last tile this multiplies a stale buffer and the result is dropped -- a branch
           // here splits the block and the scheduler then parks all the VALU work behind the MFMA chain
          const f32x4 av = *reinterpret_cast<const f32x4*>(&Yn[r31 * LDY + kb * 8 + 4 * hh]);
          sn = mfma32(av.x, xo[kb].x, sn);
          sn = mfma32(av.y, xo[kb].y, sn);
          sn = mfma32(av.z, xo[kb].z, sn);
          sn = mfma32(av.w, xo[kb].w, sn);
        }
#pragma unroll
        for (int e = 0; e < EPK; ++e) {
          const int r = kb * EPK + e;
          const float pos = MODE_USER ? pos_o : posS[cur][acc_row(r, lane)];
          g[r] = sweep_elem<MODE_USER, true>(st[r], pos, true, false, 0.f, loss_acc, den_prod, r_acc);
          if (MODE_USER && (r & 7) == 7) {
            loss_acc += __builtin_amdgcn_logf(den_prod);
            den_prod = 1.f;
          }
        }
      }
    } else {
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        {  // unconditional: on the last tile this multiplies a stale buffer and the result is dropped -- a branch
           // here splits the block and the scheduler then parks all the VALU work behind the MFMA chain
          const f32x4 av = *reinterpret_cast<const f32x4*>(&Yn[r31 * LDY + kb * 8 + 4 * hh]);
          sn = mfma32(av.x, xo[kb].x, sn);
          sn = mfma32(av.y, xo[kb].y, sn);
          sn = mfma32(av.z, xo[kb].z, sn);
          sn = mfma32(av.w, xo[kb].w, sn);
        }
#pragma unroll
        for (int e = 0; e < EPK; ++e) {
          const int r = kb * EPK + e;
          const int sl = acc_row(r, lane);
          const bool valid = o_ok && (sl < n_valid);
          const bool diag = (sl - r31 == ddi);
          const float pos = MODE_USER ? pos_o : posS[cur][sl];
          const float rd = MODE_USER ? 0.f : -rS[cur][sl] * inv_c;  // weights are unscaled until the epilogue
          g[r] = sweep_elem<MODE_USER, false>(st[r], pos, valid, diag, rd, loss_acc, den_prod, r_acc);
          if (MODE_USER && (r & 7) == 7) {
            loss_acc += __builtin_amdgcn_logf(den_prod);
            den_prod = 1.f;
          }
        }
      }
    }
    // ---- dOwner[o][c] += sum_s G[s][o] * Y[s][c]   (A operand = g registers, k = acc_row(r))
    float* gp = nullptr;
    if (GOUT) gp = a.gmat + ((size_t)tile * a.g_ub + (size_t)blockIdx.x * NW + w) * 1024 + (4 * hh) * 32 + r31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int krow = (r & 3) + 8 * (r >> 2) + 4 * hh;
      if (GOUT) gp[((r & 3) + 8 * (r >> 2)) * 32] = g[r];  // 2 full 128 B lines per wave store
#pragma unroll
      for (int t = 0; t < CT; ++t) {
        const float bv = Yc[krow * LDY + t * 32 + r31];
        out[t] = mfma32(g[r], bv, out[t]);
      }
    }
    if (has_pre) store_tile(pre);
    st = sn;
    __syncthreads();
  }

  // ---- epilogue
  if (t0 < t1) {
    const float rr = (r_acc + __shfl_xor(r_acc, 32, 64)) * a.c;  // both halves hold the same owner column
    if (hh == 0) rsum[w][r31] = rr;
    if (MODE_USER) {
      const float ls = wave_sum(loss_acc);
      if (lane == 0) red_loss[w] = (double)ls * (double)RIHIP_LN2;  // log2 units -> nats
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
      const int64_t drow = a.o_goff + orow - a.s_goff;  // owner's positive partner inside the swept set
      const bool fix = final_pass && MODE_USER && drow >= 0 && drow < a.Ns;
      const float rs = rsum[w][o];
#pragma unroll
      for (int t = 0; t < CT; ++t) {
        float v = out[t][r] * a.c;
        if (fix) v -= rs * a.Ys[drow * D + t * 32 + r31];  // G_ii = -sum_{j!=i} G_ij
        dst[orow * D + t * 32 + r31] = v;
      }
    }
  }
  if (MODE_USER && hh == 0 && o_ok) {
    if (final_pass) a.r_out[o_loc] = rsum[w][r31];
    else a.r_part[(size_t)blockIdx.y * a.No + o_loc] = rsum[w][r31];
  }
  // loss partials keep the 128-owner granularity of rihip_inbatch_loss_parts() whatever the workgroup size
  if (MODE_USER && tid < NW / 4) {
    const int64_t gx128 = (a.No + OW - 1) / OW;
    const int64_t slot = (int64_t)blockIdx.x * (NW / 4) + tid;
    if (slot < gx128)
      a.loss_part[(size_t)blockIdx.y * gx128 + slot] =
          ((red_loss[4 * tid] + red_loss[4 * tid + 1]) + red_loss[4 * tid + 2]) + red_loss[4 * tid + 3];
  }
}

// nsplit>1: dOwner = sum_s slab[s] (- r * Y[diag] in user mode); r_out = sum_s r_part[s]   (fixed order)
template <bool MODE_USER>
__global__ __launch_bounds__(256) void sweep_finish_kernel(SweepArgs a, int d) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // one float4 of one owner row
  const int v4 = d / 4;
  if (i >= a.No * v4) return;
  const int64_t orow = i / v4;
  const int c4 = (int)(i % v4);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < a.nsplit; ++s) {
    const f32x4 u = reinterpret_cast<const f32x4*>(a.slab + ((size_t)s * a.No + orow) * d)[c4];
    acc.x += u.x; acc.y += u.y; acc.z += u.z; acc.w += u.w;
  }
  if (MODE_USER) {
    float rs = 0.f;
    for (int s = 0; s < a.nsplit; ++s) rs += a.r_part[(size_t)s * a.No + orow];
    const int64_t drow = a.o_goff + orow - a.s_goff;
    if (drow >= 0 && drow < a.Ns) {
      const f32x4 y = reinterpret_cast<const f32x4*>(a.Ys + drow * d)[c4];
      acc.x -= rs * y.x; acc.y -= rs * y.y; acc.z -= rs * y.z; acc.w -= rs * y.w;
    }
    if (c4 == 0) a.r_out[orow] = rs;
  }
  reinterpret_cast<f32x4*>(a.dOwner + orow * d)[c4] = acc;
}


// ------------------------------------------------------------------------------------------
// item gradients from the stored G:  dI[j][:] = sum_i G[i][j] * U[i][:]  - r_j U_j   (plain exact-f32 GEMM)
// owners = items (4 waves x 32 per workgroup, one 32x32 G^T block per wave and swept tile, loaded straight into the
// MFMA A-operand registers: lane (item, hh) holds users 16*hh .. 16*hh+15 of its item = 64 contiguous bytes);
// swept = the local users, staged through the same 3-buffer LDS pipeline as the sweep kernel.
// ------------------------------------------------------------------------------------------
template <int D, int NW>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 3 : 2) void inbatch_gt_kernel(SweepArgs a) {
  constexpr int NT = NW * 64;
  constexpr int LDY = D + 4;
  constexpr int CT = D / 32;
  constexpr int NV = (TSW * (D / 4) + NT - 1) / NT;
  __shared__ __attribute__((aligned(16))) float Ysh[3][TSW * LDY];

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

  f32x4 stage[NV];
  auto load_tile = [&](int64_t tile) {
    const int64_t s_base = tile * TSW;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = tid + i * NT;
      const int r = idx / (D / 4), c4 = idx % (D / 4);
      const int64_t srow = s_base + r;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (idx < TSW * (D / 4) && srow < a.Ns) v = reinterpret_cast<const f32x4*>(a.Ys + srow * D)[c4];
      stage[i] = v;
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = tid + i * NT;
      const int r = idx / (D / 4), c4 = idx % (D / 4);
      if (idx < TSW * (D / 4)) *reinterpret_cast<f32x4*>(&Ysh[buf][r * LDY + c4 * 4]) = stage[i];
    }
  };
  // this wave's G^T block row; consecutive swept tiles are consecutive 4 KB blocks
  const f32x4* gp = reinterpret_cast<const f32x4*>(a.gmat + ((size_t)(blockIdx.x * NW + w) * a.g_ub) * 1024 +
                                                   r31 * 32 + 16 * hh);
  // Pipeline: the user tile t+2 and the G block t+2 are requested at the top of iteration t; the tile sits in
  // registers for one whole iteration and goes to LDS at the top of iteration t+1 (published by that iteration's
  // barrier, read in iteration t+2).  vmcnt retires in order, so the wait for a tile only covers requests that are
  // at least one full iteration (64 MFMAs per wave) old -- the HBM stream of G never stalls the L2-resident tile.
  // G of tile, tile+1, ..., tile+GA.  With d <= 64 a tile is only 32 MFMAs per wave (0.85 us): the HBM stream of G needs
  // four tiles of lead there, two at d = 128.
  constexpr int GA = (D >= 128) ? 2 : 4;
  f32x4 gs[GA + 1][4];
#pragma unroll
  for (int j = 0; j <= GA; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q) gs[j][q] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (t0 < t1) {
#pragma unroll
    for (int j = 0; j < GA; ++j)
      if (t0 + j < t1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) gs[j][q] = gp[(size_t)(t0 + j) * 256 + q];
      }
    load_tile(t0);
    store_tile(0);
    if (t0 + 1 < t1) load_tile(t0 + 1);  // stays in registers until the top of the first iteration
    __syncthreads();
  }
  int it = 0;
#pragma unroll 1
  for (int64_t tile = t0; tile < t1; ++tile, it = (it == 2 ? 0 : it + 1)) {
    const int cur = it, nxt = (it + 1) % 3;
    if (tile + 1 < t1) store_tile(nxt);  // buffer nxt was last read two iterations ago
    if (tile + 2 < t1) load_tile(tile + 2);
    if (tile + GA < t1) {
#pragma unroll
      for (int q = 0; q < 4; ++q) gs[GA][q] = gp[(size_t)(tile + GA) * 256 + q];
    }
    const float* Yc = Ysh[cur];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int krow = 16 * hh + 4 * q + s;
#pragma unroll
        for (int t = 0; t < CT; ++t) out[t] = mfma32(gs[0][q][s], Yc[krow * LDY + t * 32 + r31], out[t]);
      }
    }
#pragma unroll
    for (int j = 0; j < GA; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) gs[j][q] = gs[j + 1][q];
    __syncthreads();
  }

  const bool final_pass = (a.nsplit == 1);
  float* dst = final_pass ? a.dOwner : a.slab + (size_t)blockIdx.y * a.No * D;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int64_t orow = o_base + acc_row(r, lane);
    if (orow < a.No) {
      const int64_t drow = a.o_goff + orow - a.s_goff;  // the item's own user inside the swept (local) users
      const bool fix = final_pass && drow >= 0 && drow < a.Ns;
      const float rs = fix ? a.r_in[drow] : 0.f;
#pragma unroll
      for (int t = 0; t < CT; ++t) {
        float v = out[t][r] * a.c;                          // the stored weights are unscaled sigma(z)
        if (fix) v -= rs * a.Ys[drow * D + t * 32 + r31];  // G_jj = -r_j
        dst[orow * D + t * 32 + r31] = v;
      }
    }
  }
}

// nsplit>1 finish of the stored-G item pass: dOwner = sum_s slab[s] - r * U[diag]   (fixed order)
__global__ __launch_bounds__(256) void gt_finish_kernel(SweepArgs a, int d) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int v4 = d / 4;
  if (i >= a.No * v4) return;
  const int64_t orow = i / v4;
  const int c4 = (int)(i % v4);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < a.nsplit; ++s) {
    const f32x4 u = reinterpret_cast<const f32x4*>(a.slab + ((size_t)s * a.No + orow) * d)[c4];
    acc.x += u.x; acc.y += u.y; acc.z += u.z; acc.w += u.w;
  }
  const int64_t drow = a.o_goff + orow - a.s_goff;
  if (drow >= 0 && drow < a.Ns) {
    const float rs = a.r_in[drow];
    const f32x4 y = reinterpret_cast<const f32x4*>(a.Ys + drow * d)[c4];
    acc.x -= rs * y.x; acc.y -= rs * y.y; acc.z -= rs * y.z; acc.w -= rs * y.w;
  }
  reinterpret_cast<f32x4*>(a.dOwner + orow * d)[c4] = acc;
}

template <int D>
void launch_sweep(bool mode_user, const SweepArgs& a, dim3 grid, int nw, hipStream_t st) {
  if (nw == 8) {
    if (mode_user && a.gmat) hipLaunchKernelGGL((inbatch_sweep_kernel<D, true, true, 8>), grid, dim3(512), 0, st, a);
    else if (mode_user) hipLaunchKernelGGL((inbatch_sweep_kernel<D, true, false, 8>), grid, dim3(512), 0, st, a);
    else hipLaunchKernelGGL((inbatch_sweep_kernel<D, false, false, 8>), grid, dim3(512), 0, st, a);
    return;
  }
  if (mode_user && a.gmat) hipLaunchKernelGGL((inbatch_sweep_kernel<D, true, true>), grid, dim3(256), 0, st, a);
  else if (mode_user) hipLaunchKernelGGL((inbatch_sweep_kernel<D, true>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((inbatch_sweep_kernel<D, false>), grid, dim3(256), 0, st, a);
}

// workgroup shape: 8 waves x 32 owners (one workgroup per CU) once the problem is large enough to fill the chip that
// way -- the swept tile is then fetched once per 256 owners instead of once per 128; 4 waves otherwise.
int sweep_nw(int64_t n_owner, int64_t n_swept) {
  static const char* ev = getenv("RIHIP_SWEEP_NW");
  if (ev) return atoi(ev) == 8 ? 8 : 4;
  return (n_owner >= 4096 && n_swept >= 16384) ? 8 : 4;   // (8192 x 8192 at d = 64 measured 2 % faster with 4 waves)
}

// swept-range splits (bounded by the 128-owner formula: the workspace / loss-part sizes are computed from it)
int sweep_nsplit(int64_t n_owner, int64_t n_swept, int nw = 4) {
  const int64_t tiles = (n_swept + TSW - 1) / TSW;
  const int64_t gx4 = (n_owner + OW - 1) / OW;
  int64_t ns4 = (2 * RIHIP_NCU + gx4 - 1) / gx4;  // 4-wave workgroups: aim at >= 2 workgroups per CU
  if (ns4 > 16) ns4 = 16;
  if (ns4 > tiles) ns4 = tiles;
  if (ns4 < 1) ns4 = 1;
  if (nw == 4) return (int)ns4;
  const int64_t gx8 = (n_owner + 2 * OW - 1) / (2 * OW);
  int64_t ns = (RIHIP_NCU + gx8 - 1) / gx8;      // 8-wave workgroups: >= 1 per CU
  if (ns > ns4) ns = ns4;
  if (ns < 1) ns = 1;
  return (int)ns;
}

}  // namespace

extern "C" int rihip_bpr_pair_loss(const float* U, const float* P, const float* N, int64_t B, int d, float* loss,
                                   float* dU, float* dP, float* dN, double* workspace, void* stream) {
  RIHIP_REQUIRE(U && P && N && dU && dP && dN && workspace, RIHIP_ERR_ARG, "bpr_pair_loss: null pointer");
  RIHIP_REQUIRE(B > 0 && d > 0, RIHIP_ERR_ARG, "bpr_pair_loss: bad sizes B=%lld d=%d", (long long)B, d);
  hipStream_t st = (hipStream_t)stream;
  const int64_t nb = (B + 3) / 4;
  const int grid = (int)(nb < 1024 ? nb : 1024);
  const bool al = ((reinterpret_cast<uintptr_t>(U) | reinterpret_cast<uintptr_t>(P) | reinterpret_cast<uintptr_t>(N) |
                    reinterpret_cast<uintptr_t>(dU) | reinterpret_cast<uintptr_t>(dP) | reinterpret_cast<uintptr_t>(dN)) & 15) == 0;
  const float inv_B = 1.f / (float)B;
  if (al && d == 128) hipLaunchKernelGGL((bpr_pair_v4_kernel<32>), dim3(grid), dim3(256), 0, st, U, P, N, B, inv_B, dU, dP, dN, workspace);
  else if (al && d == 64) hipLaunchKernelGGL((bpr_pair_v4_kernel<16>), dim3(grid), dim3(256), 0, st, U, P, N, B, inv_B, dU, dP, dN, workspace);
  else if (al && d == 32) hipLaunchKernelGGL((bpr_pair_v4_kernel<8>), dim3(grid), dim3(256), 0, st, U, P, N, B, inv_B, dU, dP, dN, workspace);
  else hipLaunchKernelGGL(bpr_pair_kernel, dim3(grid), dim3(256), 0, st, U, P, N, B, d, inv_B, dU, dP, dN, workspace);
  RIHIP_CHECK_LAUNCH();
  // loss == NULL: the caller sums workspace[0 .. rihip_bpr_pair_nparts(B)) * (1/B) itself (rihip_sum_partials or
  // rihip_clip_coef_step) -- one dependent launch less on the step's critical path
  if (loss) hipLaunchKernelGGL(finalize_sum_kernel, dim3(1), dim3(64), 0, st, workspace, grid, 1.0 / (double)B, loss);
  RIHIP_CHECK_LAUNCH();
  return RIHIP_OK;
}

extern "C" int64_t rihip_bpr_pair_nparts(int64_t B) {
  const int64_t nb = (B + 3) / 4;
  return nb < 1024 ? (nb > 0 ? nb : 1) : 1024;
}

extern "C" int rihip_rowdot(const float* U, const float* I, int64_t B, int64_t i_offset, int d, float* pos,
                            void* stream) {
  RIHIP_REQUIRE(U && I && pos && B > 0 && d > 0, RIHIP_ERR_ARG, "rowdot: bad arguments");
  const int64_t nb = (B + 3) / 4;
  const int grid = (int)(nb < 1024 ? nb : 1024);
  hipLaunchKernelGGL(rowdot_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, U, I, B, i_offset, d, pos);
  RIHIP_CHECK_LAUNCH();
  return RIHIP_OK;
}

// loss partials written per sweep (doubles) and float workspace (split slabs) needed by rihip_inbatch_sweep
extern "C" int64_t rihip_inbatch_workspace_doubles(int64_t n_owner) { return ((n_owner + OW - 1) / OW) * 16 + 1; }
extern "C" int64_t rihip_inbatch_loss_parts(int64_t n_owner, int64_t n_swept) {
  return ((n_owner + OW - 1) / OW) * sweep_nsplit(n_owner, n_swept, sweep_nw(n_owner, n_swept));
}
extern "C" int64_t rihip_inbatch_workspace_floats(int64_t n_owner, int64_t n_swept, int d) {
  const int ns = sweep_nsplit(n_owner, n_swept);
  return ns > 1 ? (int64_t)ns * n_owner * (d + 1) : 1;
}

extern "C" int rihip_inbatch_sweep(int mode_user, const float* owners, int64_t n_owner, int64_t owner_goff,
                                   const float* swept, int64_t n_swept, int64_t swept_goff, int d, const float* pos,
                                   const float* r_in, int64_t n_global, float* d_owner, float* r_out,
                                   double* loss_part, float* workspace, int precision, void* stream) {
  const bool tuned = d == 32 || d == 64 || d == 128;
  RIHIP_REQUIRE(tuned || rihip_inbatch_generic_ok(d), RIHIP_ERR_SHAPE,
                "inbatch_sweep: unsupported embed_dim=%d (multiples of 16 up to 256)", d);
  RIHIP_REQUIRE(tuned || precision == 0, RIHIP_ERR_SHAPE,
                "inbatch_sweep: the split-bf16 modes exist for embed_dim 32/64/128 only (embed_dim=%d)", d);
  RIHIP_REQUIRE(precision >= 0 && precision <= 2, RIHIP_ERR_ARG,
                "inbatch_sweep: precision=%d (0=f32 MFMA, 1=bf16x3, 2=bf16x6)", precision);
  RIHIP_REQUIRE(owners && swept && pos && d_owner, RIHIP_ERR_ARG, "inbatch_sweep: null pointer");
  RIHIP_REQUIRE(mode_user ? (r_out && loss_part) : (r_in != nullptr), RIHIP_ERR_ARG,
                "inbatch_sweep: mode-specific pointer missing");
  RIHIP_REQUIRE(n_owner > 0 && n_swept > 0 && n_global >= 2, RIHIP_ERR_ARG,
                "inbatch_sweep: sizes No=%lld Ns=%lld B=%lld", (long long)n_owner, (long long)n_swept,
                (long long)n_global);
  SweepArgs a;
  a.Xo = owners; a.No = n_owner; a.o_goff = owner_goff; a.Ys = swept; a.Ns = n_swept; a.s_goff = swept_goff;
  a.pos = pos; a.r_in = r_in; a.c = (float)(1.0 / ((double)n_global * (double)(n_global - 1)));
  a.dOwner = d_owner; a.r_out = r_out; a.loss_part = loss_part;
  a.gmat = nullptr; a.g_ub = 0;
  const int nw_f32 = sweep_nw(n_owner, n_swept);
  const int nw = precision == 1 ? 4 : nw_f32;   // the split-bf16 kernel has 4-wave workgroups only
  a.nsplit = sweep_nsplit(n_owner, n_swept, nw_f32);
  RIHIP_REQUIRE(a.nsplit == 1 || workspace, RIHIP_ERR_ARG, "inbatch_sweep: workspace required (nsplit=%d)", a.nsplit);
  a.slab = workspace;
  a.r_part = workspace ? workspace + (size_t)a.nsplit * n_owner * d : nullptr;
  const dim3 grid((unsigned)((n_owner + nw * 32 - 1) / (nw * 32)), (unsigned)a.nsplit);
  hipStream_t st = (hipStream_t)stream;
  if (!tuned) {   // runtime-width kernel: final results in one launch (it fills every loss slot of the tuned layout)
    rihip_launch_sweep_generic(d, mode_user != 0, a, a.nsplit, st);
    RIHIP_CHECK_LAUNCH();
    return RIHIP_OK;
  }
  if (precision == 1) rihip_launch_sweep_bf16x3(d, mode_user != 0, a, grid, st);
  else if (precision == 2) rihip_launch_sweep_x6(d, mode_user != 0, a, grid, nw, st);
  else if (d == 32) launch_sweep<32>(mode_user != 0, a, grid, nw, st);
  else if (d == 64) launch_sweep<64>(mode_user != 0, a, grid, nw, st);
  else launch_sweep<128>(mode_user != 0, a, grid, nw, st);
  RIHIP_CHECK_LAUNCH();
  if (a.nsplit > 1) {
    const int64_t n4 = n_owner * (d / 4);
    if (mode_user) hipLaunchKernelGGL((sweep_finish_kernel<true>), dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, a, d);
    else hipLaunchKernelGGL((sweep_finish_kernel<false>), dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, a, d);
    RIHIP_CHECK_LAUNCH();
  }
  return RIHIP_OK;
}

// ---- stored-G variant: user pass writes G, item pass is a plain G^T.U product --------------------------------------
extern "C" int64_t rihip_inbatch_gmat_floats(int64_t n_users, int64_t n_items) {
  const int64_t ub = 8 * ((n_users + 2 * OW - 1) / (2 * OW)), ib = 8 * ((n_items + 2 * OW - 1) / (2 * OW));
  return ub * ib * 1024;
}

extern "C" int rihip_inbatch_user_pass(const float* users, int64_t n_users, int64_t user_goff, const float* items,
                                       int64_t n_items, int64_t item_goff, int d, const float* pos, int64_t n_global,
                                       float* d_users, float* r_out, double* loss_part, float* workspace, float* gmat,
                                       int precision, void* stream) {
  RIHIP_REQUIRE(d == 32 || d == 64 || d == 128, RIHIP_ERR_SHAPE,
                "inbatch_user_pass: the stored-G form exists for embed_dim 32/64/128 only (embed_dim=%d: use rihip_inbatch_sweep)", d);
  RIHIP_REQUIRE(users && items && pos && d_users && r_out && loss_part && gmat, RIHIP_ERR_ARG,
                "inbatch_user_pass: null pointer");
  RIHIP_REQUIRE(precision == 0 || precision == 2, RIHIP_ERR_ARG,
                "inbatch_user_pass: precision=%d (0=f32 MFMA, 2=bf16x6)", precision);
  RIHIP_REQUIRE(n_users > 0 && n_items > 0 && n_global >= 2, RIHIP_ERR_ARG,
                "inbatch_user_pass: sizes users=%lld items=%lld B=%lld", (long long)n_users, (long long)n_items,
                (long long)n_global);
  SweepArgs a;
  a.Xo = users; a.No = n_users; a.o_goff = user_goff; a.Ys = items; a.Ns = n_items; a.s_goff = item_goff;
  a.pos = pos; a.r_in = nullptr; a.c = (float)(1.0 / ((double)n_global * (double)(n_global - 1)));
  a.dOwner = d_users; a.r_out = r_out; a.loss_part = loss_part;
  a.gmat = gmat; a.g_ub = 8 * ((n_users + 2 * OW - 1) / (2 * OW));
  const int nw = sweep_nw(n_users, n_items);
  a.nsplit = sweep_nsplit(n_users, n_items, nw);
  RIHIP_REQUIRE(a.nsplit == 1 || workspace, RIHIP_ERR_ARG, "inbatch_user_pass: workspace required (nsplit=%d)", a.nsplit);
  a.slab = workspace;
  a.r_part = workspace ? workspace + (size_t)a.nsplit * n_users * d : nullptr;
  const dim3 grid((unsigned)((n_users + nw * 32 - 1) / (nw * 32)), (unsigned)a.nsplit);
  hipStream_t st = (hipStream_t)stream;
  if (precision == 2) rihip_launch_sweep_x6(d, true, a, grid, nw, st);
  else if (d == 32) launch_sweep<32>(true, a, grid, nw, st);
  else if (d == 64) launch_sweep<64>(true, a, grid, nw, st);
  else launch_sweep<128>(true, a, grid, nw, st);
  RIHIP_CHECK_LAUNCH();
  if (a.nsplit > 1) {
    const int64_t n4 = n_users * (d / 4);
    hipLaunchKernelGGL((sweep_finish_kernel<true>), dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, a, d);
    RIHIP_CHECK_LAUNCH();
  }
  return RIHIP_OK;
}

extern "C" int rihip_inbatch_item_pass(const float* gmat, const float* users, int64_t n_users, int64_t user_goff,
                                       int64_t n_items, int64_t item_goff, int d, const float* r, int64_t n_global,
                                       float* d_items, float* workspace, int precision, void* stream) {
  RIHIP_REQUIRE(d == 32 || d == 64 || d == 128, RIHIP_ERR_SHAPE, "inbatch_item_pass: unsupported embed_dim=%d", d);
  RIHIP_REQUIRE(gmat && users && r && d_items, RIHIP_ERR_ARG, "inbatch_item_pass: null pointer");
  RIHIP_REQUIRE(precision == 0 || precision == 2, RIHIP_ERR_ARG,
                "inbatch_item_pass: precision=%d (0=f32 MFMA, 2=bf16x6)", precision);
  RIHIP_REQUIRE(n_users > 0 && n_items > 0 && n_global >= 2, RIHIP_ERR_ARG,
                "inbatch_item_pass: sizes users=%lld items=%lld B=%lld", (long long)n_users, (long long)n_items,
                (long long)n_global);
  SweepArgs a;
  a.Xo = nullptr; a.No = n_items; a.o_goff = item_goff; a.Ys = users; a.Ns = n_users; a.s_goff = user_goff;
  a.pos = nullptr; a.r_in = r; a.c = (float)(1.0 / ((double)n_global * (double)(n_global - 1))); a.dOwner = d_items; a.r_out = nullptr; a.loss_part = nullptr;
  a.gmat = const_cast<float*>(gmat); a.g_ub = 8 * ((n_users + 2 * OW - 1) / (2 * OW));
  // 8-wave workgroups (256 items) halve the L2->L1 traffic of the swept user tiles; used when they still fill the chip
  const int nw = sweep_nw(n_items, n_users);
  const int ow = nw * 32;
  const int64_t gx = (n_items + ow - 1) / ow;
  a.nsplit = sweep_nsplit(n_items, n_users, nw);
  RIHIP_REQUIRE(a.nsplit == 1 || workspace, RIHIP_ERR_ARG, "inbatch_item_pass: workspace required (nsplit=%d)", a.nsplit);
  a.slab = workspace; a.r_part = nullptr;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)gx, (unsigned)a.nsplit);
  if (precision == 2) {
    rihip_launch_gt_x6(d, a, grid, nw, st);
  } else if (nw == 8) {
    if (d == 32) hipLaunchKernelGGL((inbatch_gt_kernel<32, 8>), grid, dim3(512), 0, st, a);
    else if (d == 64) hipLaunchKernelGGL((inbatch_gt_kernel<64, 8>), grid, dim3(512), 0, st, a);
    else hipLaunchKernelGGL((inbatch_gt_kernel<128, 8>), grid, dim3(512), 0, st, a);
  } else {
    if (d == 32) hipLaunchKernelGGL((inbatch_gt_kernel<32, 4>), grid, dim3(256), 0, st, a);
    else if (d == 64) hipLaunchKernelGGL((inbatch_gt_kernel<64, 4>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((inbatch_gt_kernel<128, 4>), grid, dim3(256), 0, st, a);
  }
  RIHIP_CHECK_LAUNCH();
  if (a.nsplit > 1) {
    const int64_t n4 = n_items * (d / 4);
    hipLaunchKernelGGL(gt_finish_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, a, d);
    RIHIP_CHECK_LAUNCH();
  }
  return RIHIP_OK;
}

extern "C" int rihip_sum_partials(const double* part, int64_t n, double scale, float* out, void* stream) {
  RIHIP_REQUIRE(part && out && n > 0, RIHIP_ERR_ARG, "sum_partials: bad arguments");
  hipLaunchKernelGGL(finalize_sum_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, part, (int)n, scale, out);
  RIHIP_CHECK_LAUNCH();
  return RIHIP_OK;
}
