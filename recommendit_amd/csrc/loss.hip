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
#include "common.h"
#include "recommendit_hip.h"

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
struct SweepArgs {
  const float* Xo;   // owners [No,d]
  int64_t No;
  int64_t o_goff;    // global index of owner 0
  const float* Ys;   // swept [Ns,d]
  int64_t Ns;
  int64_t s_goff;    // global index of swept 0
  const float* pos;  // MODE_USER: [No] by owner ; MODE_ITEM: [Ns] by swept (user) index
  const float* r_in; // MODE_ITEM: [Ns] rowsum of G per user
  float c;           // 1/(B(B-1))
  float* dOwner;     // [No,d]
  float* r_out;      // MODE_USER: [No]
  double* loss_part; // MODE_USER: [grid]
};

constexpr int TS = 128;  // swept rows staged per iteration (32 per wave)

template <int D, bool MODE_USER>
__global__ __launch_bounds__(256) void inbatch_sweep_kernel(SweepArgs a) {
  constexpr int LDY = D + 4;
  constexpr int KB = D / 8, CT = D / 32;
  __shared__ __attribute__((aligned(16))) float Ysh[TS * LDY];  // swept tile; reused for the cross-wave reduction
  __shared__ float posS[TS];
  __shared__ float rS[TS];
  __shared__ float red_r[4][32];
  __shared__ double red_loss[4];

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int r31 = lane & 31, hh = lane >> 5;
  const int64_t o_base = (int64_t)blockIdx.x * 32;
  const int64_t o_loc = o_base + r31;           // this lane's owner (S^T accumulator column)
  const bool o_ok = o_loc < a.No;
  const int64_t o_gidx = a.o_goff + o_loc;

  // register-stationary owner fragments: B[k][n=o] = Xo[o][k]
  f32x4 xo[KB];
  {
    const int64_t orow = o_ok ? o_loc : (a.No - 1);
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(&a.Xo[orow * D + kb * 8 + 4 * hh]);
      xo[kb] = v;
    }
  }
  const float pos_o = (MODE_USER && o_ok) ? a.pos[o_loc] : 0.f;

  f32x16 out[CT];
#pragma unroll
  for (int t = 0; t < CT; ++t) out[t] = zero16();
  float r_acc = 0.f;     // MODE_USER: sum_s G[o][s] over this wave's share (per lane = per owner, this half's rows)
  float loss_acc = 0.f;

  const int64_t ntiles = (a.Ns + TS - 1) / TS;
  for (int64_t tile = 0; tile < ntiles; ++tile) {
    const int64_t s_base = tile * TS;
    // ---- stage swept tile (zero rows beyond Ns)
    for (int idx = tid; idx < TS * (D / 4); idx += 256) {
      const int r = idx / (D / 4), c4 = idx % (D / 4);
      const int64_t srow = s_base + r;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (srow < a.Ns) v = reinterpret_cast<const f32x4*>(a.Ys + srow * D)[c4];
      *reinterpret_cast<f32x4*>(&Ysh[r * LDY + c4 * 4]) = v;
    }
    if (!MODE_USER && tid < TS) {
      const int64_t srow = s_base + tid;
      posS[tid] = (srow < a.Ns) ? a.pos[srow] : 0.f;
      rS[tid] = (srow < a.Ns) ? a.r_in[srow] : 0.f;
    }
    __syncthreads();

    // ---- S^T[s][o] for this wave's 32 swept rows
    const float* Yw = &Ysh[(w * 32) * LDY];
    f32x16 st = zero16();
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      const f32x4 av = *reinterpret_cast<const f32x4*>(&Yw[r31 * LDY + kb * 8 + 4 * hh]);
      st = mfma32(av.x, xo[kb].x, st);
      st = mfma32(av.y, xo[kb].y, st);
      st = mfma32(av.z, xo[kb].z, st);
      st = mfma32(av.w, xo[kb].w, st);
    }
    // ---- G = sigma(z) * c  (diagonal / out-of-range masked)
    float g[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int sl = w * 32 + acc_row(r, lane);
      const int64_t srow = s_base + sl;
      const int64_t s_gidx = a.s_goff + srow;
      const bool valid = o_ok && (srow < a.Ns);
      const bool diag = (s_gidx == o_gidx);
      const float z = st[r] - (MODE_USER ? pos_o : posS[sl]);
      const float e = expf(-fabsf(z));
      const float sig = ((z >= 0.f) ? 1.f : e) / (1.f + e);
      float gv = sig * a.c;
      if (MODE_USER) {
        const float sp = fmaxf(z, 0.f) + log1pf(e);
        if (valid && !diag) {
          loss_acc += sp;
          r_acc += gv;
        } else {
          gv = 0.f;
        }
      } else {
        if (!valid) gv = 0.f;
        else if (diag) gv = -rS[sl];
      }
      g[r] = gv;
    }
    // ---- dOwner[o][c] += sum_s G[s][o] * Y[s][c]   (A operand = g registers, k = acc_row(r))
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int krow = (r & 3) + 8 * (r >> 2) + 4 * hh;
#pragma unroll
      for (int t = 0; t < CT; ++t) {
        const float bv = Yw[krow * LDY + t * 32 + r31];
        out[t] = mfma32(g[r], bv, out[t]);
      }
    }
    __syncthreads();
  }

  // ---- cross-wave reduction through LDS (fixed order w=0..3 => deterministic)
  float* red = Ysh;  // [4][32][LDY]
#pragma unroll
  for (int t = 0; t < CT; ++t) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(w * 32 + acc_row(r, lane)) * LDY + t * 32 + r31] = out[t][r];
  }
  if (MODE_USER) {
    const float rr = r_acc + __shfl_xor(r_acc, 32, 64);  // both halves hold the same owner column
    if (hh == 0) red_r[w][r31] = rr;
    const float ls = wave_sum(loss_acc);
    if (lane == 0) red_loss[w] = (double)ls;
  }
  __syncthreads();
  {
    const int o = tid >> 3, q = tid & 7;  // 8 threads per owner row
    const int64_t orow = o_base + o;
    float rsum = 0.f;
    if (MODE_USER) rsum = red_r[0][o] + red_r[1][o] + red_r[2][o] + red_r[3][o];
    if (orow < a.No) {
      const int64_t drow = a.o_goff + orow - a.s_goff;  // owner's positive partner in the swept set
      const bool has_diag = MODE_USER && drow >= 0 && drow < a.Ns;
      for (int c4 = q; c4 < D / 4; c4 += 8) {
        f32x4 v = *reinterpret_cast<const f32x4*>(&red[(0 * 32 + o) * LDY + c4 * 4]);
#pragma unroll
        for (int ww = 1; ww < 4; ++ww) {
          const f32x4 u = *reinterpret_cast<const f32x4*>(&red[(ww * 32 + o) * LDY + c4 * 4]);
          v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        }
        if (has_diag) {  // G_ii = -sum_{j!=i} G_ij
          const f32x4 y = reinterpret_cast<const f32x4*>(a.Ys + drow * D)[c4];
          v.x -= rsum * y.x; v.y -= rsum * y.y; v.z -= rsum * y.z; v.w -= rsum * y.w;
        }
        reinterpret_cast<f32x4*>(a.dOwner + orow * D)[c4] = v;
      }
      if (MODE_USER && q == 0) a.r_out[orow] = rsum;
    }
  }
  if (MODE_USER && tid == 0) a.loss_part[blockIdx.x] = red_loss[0] + red_loss[1] + red_loss[2] + red_loss[3];
}

template <int D>
void launch_sweep(bool mode_user, const SweepArgs& a, int grid, hipStream_t st) {
  if (mode_user) hipLaunchKernelGGL((inbatch_sweep_kernel<D, true>), dim3(grid), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((inbatch_sweep_kernel<D, false>), dim3(grid), dim3(256), 0, st, a);
}

}  // namespace

extern "C" int rihip_bpr_pair_loss(const float* U, const float* P, const float* N, int64_t B, int d, float* loss,
                                   float* dU, float* dP, float* dN, double* workspace, void* stream) {
  RIHIP_REQUIRE(U && P && N && loss && dU && dP && dN && workspace, RIHIP_ERR_ARG, "bpr_pair_loss: null pointer");
  RIHIP_REQUIRE(B > 0 && d > 0, RIHIP_ERR_ARG, "bpr_pair_loss: bad sizes B=%lld d=%d", (long long)B, d);
  hipStream_t st = (hipStream_t)stream;
  const int64_t nb = (B + 3) / 4;
  const int grid = (int)(nb < 1024 ? nb : 1024);
  hipLaunchKernelGGL(bpr_pair_kernel, dim3(grid), dim3(256), 0, st, U, P, N, B, d, 1.f / (float)B, dU, dP, dN,
                     workspace);
  RIHIP_CHECK_LAUNCH();
  hipLaunchKernelGGL(finalize_sum_kernel, dim3(1), dim3(64), 0, st, workspace, grid, 1.0 / (double)B, loss);
  RIHIP_CHECK_LAUNCH();
  return RIHIP_OK;
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

extern "C" int64_t rihip_inbatch_workspace_doubles(int64_t n_owner) { return (n_owner + 31) / 32 + 1; }

extern "C" int rihip_inbatch_sweep(int mode_user, const float* owners, int64_t n_owner, int64_t owner_goff,
                                   const float* swept, int64_t n_swept, int64_t swept_goff, int d, const float* pos,
                                   const float* r_in, int64_t n_global, float* d_owner, float* r_out,
                                   double* loss_part, void* stream) {
  RIHIP_REQUIRE(d == 32 || d == 64 || d == 128, RIHIP_ERR_SHAPE, "inbatch_sweep: unsupported embed_dim=%d", d);
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
  const int grid = (int)((n_owner + 31) / 32);
  hipStream_t st = (hipStream_t)stream;
  if (d == 32) launch_sweep<32>(mode_user != 0, a, grid, st);
  else if (d == 64) launch_sweep<64>(mode_user != 0, a, grid, st);
  else launch_sweep<128>(mode_user != 0, a, grid, st);
  RIHIP_CHECK_LAUNCH();
  return RIHIP_OK;
}

extern "C" int rihip_sum_partials(const double* part, int64_t n, double scale, float* out, void* stream) {
  RIHIP_REQUIRE(part && out && n > 0, RIHIP_ERR_ARG, "sum_partials: bad arguments");
  hipLaunchKernelGGL(finalize_sum_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, part, (int)n, scale, out);
  RIHIP_CHECK_LAUNCH();
  return RIHIP_OK;
}
