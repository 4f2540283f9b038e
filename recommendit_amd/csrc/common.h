// Shared device/host helpers for the recommendit gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define RIHIP_OK 0
#define RIHIP_ERR_ARG 1
#define RIHIP_ERR_HIP 2
#define RIHIP_ERR_SHAPE 3
#define RIHIP_ERR_IO 4
#define RIHIP_ERR_STATE 5

void rihip_set_error(const char* fmt, ...);
// Library-wide generation of everything a captured hipGraph may have baked in (handle-owned scratch pointers, nprobe,
// id maps, index / forest content): bumped whenever such state is reallocated, freed or changed.
void rihip_bump_generation(void);

#define RIHIP_CHECK_HIP(expr)                                                          \
  do {                                                                                 \
    hipError_t _e = (expr);                                                            \
    if (_e != hipSuccess) {                                                            \
      rihip_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return RIHIP_ERR_HIP;                                                            \
    }                                                                                  \
  } while (0)

#define RIHIP_CHECK_LAUNCH()                                                           \
  do {                                                                                 \
    hipError_t _e = hipGetLastError();                                                 \
    if (_e != hipSuccess) {                                                            \
      rihip_set_error("%s:%d kernel launch -> %s", __FILE__, __LINE__, hipGetErrorString(_e)); \
      return RIHIP_ERR_HIP;                                                            \
    }                                                                                  \
  } while (0)

#define RIHIP_REQUIRE(cond, code, ...)                                                 \
  do {                                                                                 \
    if (!(cond)) {                                                                     \
      rihip_set_error(__VA_ARGS__);                                                    \
      return (code);                                                                   \
    }                                                                                  \
  } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define RIHIP_NCU 256

// ---- exact-f32 MFMA 32x32x2: D[i][j] += sum_{k<2} A[i][k] B[k][j] -------------------
// lane l supplies A[i=l&31][k=l>>5] and B[k=l>>5][j=l&31]; accumulator register r of lane
// l holds D[row=(r&3)+8*(r>>2)+4*(l>>5)][col=l&31].
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}

// ---- counter-based dropout mask (bit-identical to oracle/two_tower_np.py: dropout_keep_mask) ----------
// 32-bit avalanche hash ("lowbias32") of the 64-bit element counter folded with the 64-bit seed: two 32-bit
// multiplies per element instead of splitmix64's 64-bit ones (the mask sits in the MFMA epilogue).
__host__ __device__ __forceinline__ uint64_t rihip_splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x ^= x >> 30;
  x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27;
  x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return x;
}
__host__ __device__ __forceinline__ uint32_t rihip_lowbias32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7feb352du;
  x ^= x >> 15;
  x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}
// keep element `idx` (global row * n_cols + col) with probability 1-p; thresh24 = floor(p*2^24)
__host__ __device__ __forceinline__ bool rihip_keep(uint64_t seed_mix, uint64_t idx, uint32_t thresh24) {
  const uint32_t lo = (uint32_t)idx ^ (uint32_t)seed_mix;
  const uint32_t hi = (uint32_t)(idx >> 32) ^ (uint32_t)(seed_mix >> 32);
  const uint32_t h = rihip_lowbias32(lo ^ rihip_lowbias32(hi + 0x9E3779B9u));
  return (h >> 8) >= thresh24;
}
// same decision for element counters below 2^32 (the upper word of the counter is 0, so the inner hash is a constant
// of the launch): inner = rihip_lowbias32((uint32_t)(seed_mix >> 32) + 0x9E3779B9u)
__host__ __device__ __forceinline__ bool rihip_keep32(uint32_t seed_lo, uint32_t inner, uint32_t idx, uint32_t thresh24) {
  return (rihip_lowbias32((idx ^ seed_lo) ^ inner) >> 8) >= thresh24;
}
// the 64-bit seed is scrambled once on the host
__host__ __device__ __forceinline__ uint64_t rihip_seed_mul(uint64_t seed) { return rihip_splitmix64(seed); }
static inline uint32_t rihip_thresh24(float p) {
  double t = (double)p * 16777216.0;
  if (t < 0) t = 0;
  if (t > 16777215.0) t = 16777215.0;
  return (uint32_t)t;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
