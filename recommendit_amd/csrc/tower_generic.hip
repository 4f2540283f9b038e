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
#include "tower_generic_body.h"

using namespace rihip_gen;

namespace {

__global__ __launch_bounds__(256) void tower_fwd_generic_kernel(GenFwd g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const TowerFwdArgs& a = g.a;
  const uint64_t seed_mul = a.seed_step ? rihip_splitmix64(a.seed_mul + (uint64_t)(*a.seed_step)) : a.seed_mul;
  const int64_t ntiles = (a.B + GTM - 1) / GTM;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) gen_fwd_tile(g, smem, tile, seed_mul);
}

__global__ __launch_bounds__(256) void tower_bwd_data_generic_kernel(GenBwd g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int64_t ntiles = (g.a.B + GTM - 1) / GTM;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) gen_bwd_data_tile(g, smem, tile);
}

// blockIdx.y = slab (a range of batch rows), blockIdx.x = a group of 4 output tiles (one per wave)
__global__ __launch_bounds__(256) void tower_wgrad_generic_kernel(GenBwd g) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int tile = blockIdx.x * 4 + w;
  if (tile >= gen_wgrad_tiles(g.D, g.H, g.K1)) return;
  const GenWTile t = gen_wgrad_tile_of(g, tile);
  const int64_t r0 = (int64_t)blockIdx.y * g.rows_per_slab;
  int64_t r1 = r0 + g.rows_per_slab;
  if (r1 > g.a.B) r1 = g.a.B;
  const f32x16 acc = gen_wgrad_acc(g, t, r0, 32, r1, lane);
  // slab layout (shared with the tuned kernels): [dW1 H*K1 | db1 H | dW2 D*H | db2 D]
  float* slab = g.a.slab + (size_t)blockIdx.y * g.P;
  float* dW1 = slab, *db1 = dW1 + g.H * g.K1, *dW2 = db1 + g.H, *db2 = dW2 + g.D * g.H;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float* dst = gen_wgrad_dst(g, t, r, lane, dW1, db1, dW2, db2);
    if (dst) *dst = acc[r];
  }
}

size_t fwd_lds(int D, int H, int K1) { return sizeof(float) * gen_fwd_lds_floats(D, H, K1); }
size_t bwd_lds(int D, int H) { return sizeof(float) * gen_bwd_lds_floats(D, H); }

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
  const int n_tiles_out = gen_wgrad_tiles(d, hidden, g.K1);
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
