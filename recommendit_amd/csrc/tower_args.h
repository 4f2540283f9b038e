// Argument block shared by the two forward-kernel families (tower.hip, tower2.hip).
#pragma once
#include "common.h"

struct TowerFwdArgs {
  const f32x4 *W1p, *W2p;  // fragment-major weights (null => strided loads from W1/W2)
  const float* table;
  int64_t n_rows;
  const int64_t* ids;
  const float* genres;  // [B,18] or null
  int64_t B;
  const float *W1, *b1, *W2, *b2;
  float* out;    // [B,D]
  float* hid;    // [B,H] post-dropout hidden (nullable)
  float* denom;  // [B] max(|y|,eps) (nullable)
  int training;
  uint64_t seed_mul;
  uint32_t thresh24;
  float scale;      // 1/(1-p)
  int64_t row0;     // global row offset for the dropout counter
  const int64_t* seed_step;  // optional device step counter mixed into the dropout seed (graph replay)
  int* err_flag;    // set to 1 on out-of-range id (nullable)
};

struct TowerBwdArgs {
  const float* table;
  int64_t n_rows;
  const int64_t* ids;
  const float* genres;
  int64_t B;
  const float *W1, *W2;
  const float* gout;   // [B,D] dL/d out
  const float* out;    // [B,D]
  const float* denom;  // [B]
  const float* hid;    // [B,H]
  float scale;         // dropout 1/(1-p) (1 when not training)
  float* dX;           // [B,D] per-sample embedding-row grads
  float* slab;         // [grid][P] partial weight grads, P = H*K1 + H + D*H + D
};

// two-kernel backward (tower2.hip): data-gradient kernel (wave per 32 rows, weights in LDS) that also writes gy [B,D]
// and dPre [B,H] to `act`, then the weight-gradient kernel (LDS-tiled split-K over the batch) that fills the slabs.
// Returns the number of slabs written (0: no instantiation for this (d, hidden)).
// dx_event (nullable) is recorded on st as soon as dX is complete (before the weight-gradient kernel).
int rihip_launch_tower_bwd2(int d, int hidden, bool item, const TowerBwdArgs& a, float* act, hipStream_t st,
                            hipEvent_t dx_event);

// wave-per-32-rows forward (tower2.hip); returns false when the (d, hidden) pair has no instantiation
bool rihip_launch_tower_fwd2(int d, int hidden, bool item, const TowerFwdArgs& a, hipStream_t st);
