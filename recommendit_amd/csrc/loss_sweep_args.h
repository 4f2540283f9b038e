// Shared between the exact-f32 (loss.hip) and split-bf16 (loss_bf16.hip) in-batch sweep kernels.
#pragma once
#include "common.h"

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
  double* loss_part; // MODE_USER: [grid.x*grid.y]
  float* slab;       // nsplit>1: [nsplit][No][d] partial owner gradients
  float* r_part;     // nsplit>1, MODE_USER: [nsplit][No]
  int nsplit;
  float* gmat;       // stored-G variant: G^T in 32x32 blocks, block (item_blk, user_blk) at (item_blk*g_ub + user_blk)*1024,
  int64_t g_ub;      //   element (item_local, user_local) at item_local*32 + user_local ; g_ub = user blocks per item block row
};

constexpr int OW = 128;  // owners per workgroup (32 per wave)
constexpr int TSW = 32;  // swept rows per LDS tile (shared by the 4 waves)

// Workgroup = 4 waves x 32 register-stationary owners; every wave multiplies the SAME 32-row swept tile,
// so one 16 KB (d=128) tile feeds 4 x 128 MFMAs.  Software pipeline (3 LDS tile buffers, one barrier per tile):
//   iteration t:  global loads of tile t+2 -> registers
//                 S^T(t+1) MFMA chain  INTERLEAVED with the sigma/softplus VALU work on S^T(t)
//                 (the chain is latency-paced at 64 cycles per MFMA, so the VALU instructions ride in its shadow)
//                 dOwner += G(t)^T . Y(t)   (accumulator registers are the A operand)
//                 registers -> LDS buffer of tile t+2 ; barrier
// Tiles that contain neither the diagonal nor a ragged edge take a branch-free element path.
// gridDim.y splits the swept range so that small batches still fill the chip; partial owner gradients of the
// splits are combined in fixed order by sweep_finish_kernel.
// One score element.  Inputs are in log2 units (owners and pos are pre-scaled by log2(e)), so
//   z2 = log2(e) (s_ij - s_ii),  e = 2^-z2 = e^-z,  sigma(z) = 1/(1+e),  softplus(z) = z + log(1+e).
// VALU instructions are NOT hidden beside v_mfma_f32_32x32x2_f32 (measured: ~2.8 cycles per simple op, ~7.3 per
// transcendental, on top of the MFMA's 64), so the element is kept at 7 instructions: sub, exp, add, rcp and three
// accumulates; the scale c = 1/(B(B-1)) is applied once in the epilogue, the logs are taken on products of eight
// factors (1+e <= 1+e^|z|: eight of them overflow f32 only for mean |z| > 11; L2-normalised rows have |z| <= 2).
// e = +inf (z << 0) gives sigma = 0, e = 0 gives sigma = 1: the right limits without a branch.
// Returns the UNSCALED gradient weight sigma(z) (0 for masked elements; r_diag on the diagonal in item mode).
constexpr float RIHIP_LOG2E = 1.4426950408889634f;
constexpr float RIHIP_LN2 = 0.6931471805599453f;
template <bool MODE_USER, bool FAST>
__device__ __forceinline__ float sweep_elem(float s2, float pos2, bool valid, bool diag, float r_diag,
                                            float& loss2_acc, float& den_prod, float& r_acc) {
  const float z2 = s2 - pos2;
  const float e = __builtin_amdgcn_exp2f(-z2);
  const float den = 1.f + e;
  float gv = __builtin_amdgcn_rcpf(den);
  if (MODE_USER) {
    if (FAST || (valid && !diag)) {
      loss2_acc += z2;
      den_prod *= den;
      r_acc += gv;
    } else {
      gv = 0.f;
    }
  } else if (!FAST) {
    if (!valid) gv = 0.f;
    else if (diag) gv = r_diag;
  }
  return gv;
}
