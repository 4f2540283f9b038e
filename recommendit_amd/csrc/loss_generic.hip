// In-batch-negative BPR (TwoTowerModel.in_batch_bpr_loss, src/models/two_tower.py:132-160) for embedding widths without
// a tuned sweep instantiation: any multiple of 16 up to 256.  Same contract as inbatch_sweep_kernel (loss.hip), recompute
// form: mode user = owners are users, swept are items (loss, r, dU); mode item = owners are items, swept are users (dI).
//   S = O.Y^T ; z_ij = s_ij - pos ; g = sigma(z) ; dOwner = c (G.Y) (- r_i Y_i) ; loss = sum softplus(z)
// One workgroup (4 waves) owns 128 owners = one loss slot, 32 at a time in LDS; the swept rows go by in tiles of 128
// (one 32x32 score tile per wave), the weights G of a tile are written to LDS and multiplied back against the same
// swept rows; both products run through the runtime-shape workgroup GEMM of gen_gemm.h (exact-f32 MFMA).
#include "common.h"
#include "gen_gemm.h"
#include "loss_sweep_args.h"

using namespace rihip_gen;

namespace {

constexpr int SWT = 128;   // swept rows per tile

struct GenSweep {
  SweepArgs a;
  int D;
};

template <bool MODE_USER>
__global__ __launch_bounds__(256) void inbatch_sweep_generic_kernel(GenSweep g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const SweepArgs& a = g.a;
  const int D = g.D;
  const int ldo = D + 4, ldg = SWT + 4;
  float* Os = smem;                    // [32][ldo] owners
  float* Gs = Os + 32 * ldo;           // [32][ldg] weights of the current swept tile
  float* Wp = Gs + 32 * ldg;           // [256][GLDP]
  float* red = Wp + 256 * GLDP;        // [4][32] per-wave row sums
  __shared__ double red_loss[4];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r31 = lane & 31;
  const int64_t gx128 = (a.No + OW - 1) / OW;
  if (blockIdx.y > 0) {                // the tuned layout reserves a loss slot per swept split: unused here
    if (MODE_USER && tid == 0) a.loss_part[(size_t)blockIdx.y * gx128 + blockIdx.x] = 0.0;
    return;
  }
  double loss_wg = 0.0;
  for (int og = 0; og < 4; ++og) {
    const int64_t o_base = (int64_t)blockIdx.x * OW + og * 32;
    if (o_base >= a.No) break;
    __syncthreads();
    for (int idx = tid; idx < 32 * D; idx += 256) {
      const int r = idx / D, k = idx % D;
      Os[r * ldo + k] = (o_base + r < a.No) ? a.Xo[(o_base + r) * D + k] : 0.f;
    }
    f32x16 out[GNT];
#pragma unroll
    for (int t = 0; t < GNT; ++t) out[t] = zero16();
    float racc[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) racc[r] = 0.f;
    float loss_acc = 0.f;
    for (int64_t sb = 0; sb < a.Ns; sb += SWT) {
      const int nsw = (a.Ns - sb < SWT) ? (int)(a.Ns - sb) : SWT;
      f32x16 sacc[GNT];
      // S[owner][swept] = O . Y^T : wave w holds the 32 swept rows sb + 32w .. (tile nt = w)
      wg_gemm<false>(Os, ldo, D, a.Ys + sb * D, D, nsw, Wp, sacc, tid);
      const int64_t srow = sb + w * 32 + r31;
      const bool s_ok = srow < a.Ns;
      const int64_t s_g = a.s_goff + srow;
      const float pos_s = (!MODE_USER && s_ok) ? a.pos[srow] : 0.f;
      const float rd_s = (!MODE_USER && s_ok) ? -a.r_in[srow] / a.c : 0.f;   // weights are unscaled until the epilogue
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ol = acc_row(r, lane);
        const int64_t orow = o_base + ol;
        const bool valid = s_ok && orow < a.No;
        const bool diag = (a.o_goff + orow) == s_g;
        const float z = sacc[0][r] - (MODE_USER ? (orow < a.No ? a.pos[orow] : 0.f) : pos_s);
        float gv = 1.f / (1.f + __expf(-z));
        if (MODE_USER) {
          if (valid && !diag) {
            loss_acc += fmaxf(z, 0.f) + log1pf(__expf(-fabsf(z)));
            racc[r] += gv;
          } else gv = 0.f;
        } else {
          if (!valid) gv = 0.f;
          else if (diag) gv = rd_s;
        }
        Gs[ol * ldg + w * 32 + r31] = gv;
      }
      // dOwner[owner][c] += sum_s G[owner][s] Y[s][c]   (B[n = c][k = s] = Y[(sb + s)*D + c]: transposed panel)
      wg_gemm<true, true>(Gs, ldg, SWT, a.Ys + sb * D, D, D, Wp, out, tid, nsw);
    }
    // ---- row sums of G per owner (user mode) and the loss of this owner group
    __syncthreads();
    if (MODE_USER) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = racc[r];
        v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
        v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64);
        if (r31 == 0) red[w * 32 + acc_row(r, lane)] = v;
      }
      const float ls = wave_sum(loss_acc);
      if (lane == 0) red_loss[w] = (double)ls;
    }
    __syncthreads();
    if (MODE_USER) loss_wg += ((red_loss[0] + red_loss[1]) + red_loss[2]) + red_loss[3];
#pragma unroll
    for (int t = 0; t < GNT; ++t) {
      const int col = (w + 4 * t) * 32 + r31;
      if (col < D) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ol = acc_row(r, lane);
          const int64_t orow = o_base + ol;
          if (orow < a.No) {
            float v = out[t][r] * a.c;
            if (MODE_USER) {
              const float rs = (((red[ol] + red[32 + ol]) + red[64 + ol]) + red[96 + ol]) * a.c;
              const int64_t drow = a.o_goff + orow - a.s_goff;    // the owner's positive partner inside the swept set
              if (drow >= 0 && drow < a.Ns) v -= rs * a.Ys[drow * D + col];   // G_ii = -sum_{j != i} G_ij
              if (col == 0) a.r_out[orow] = rs;
            }
            a.dOwner[orow * D + col] = v;
          }
        }
      }
    }
  }
  if (MODE_USER && tid == 0) a.loss_part[blockIdx.x] = loss_wg;
}

size_t sweep_lds(int D) { return sizeof(float) * ((size_t)32 * (D + 4) + 32 * (SWT + 4) + 256 * GLDP + 128); }

}  // namespace

bool rihip_inbatch_generic_ok(int d) { return d >= 16 && d <= 256 && d % 16 == 0; }

void rihip_launch_sweep_generic(int d, bool mode_user, const SweepArgs& a, int nsplit_slots, hipStream_t st) {
  GenSweep g;
  g.a = a; g.D = d;
  static bool granted = false;
  if (!granted) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(inbatch_sweep_generic_kernel<true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)sweep_lds(256));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(inbatch_sweep_generic_kernel<false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)sweep_lds(256));
    granted = true;
  }
  const dim3 grid((unsigned)((a.No + OW - 1) / OW), (unsigned)(mode_user ? nsplit_slots : 1));
  if (mode_user) hipLaunchKernelGGL(inbatch_sweep_generic_kernel<true>, grid, dim3(256), sweep_lds(d), st, g);
  else hipLaunchKernelGGL(inbatch_sweep_generic_kernel<false>, grid, dim3(256), sweep_lds(d), st, g);
}
