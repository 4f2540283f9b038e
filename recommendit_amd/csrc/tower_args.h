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

// wave-per-32-rows forward (tower2.hip); returns false when the (d, hidden) pair has no instantiation
bool rihip_launch_tower_fwd2(int d, int hidden, bool item, const TowerFwdArgs& a, hipStream_t st);
