// Price of VALU / transcendental / LDS instructions issued beside v_mfma_f32_32x32x2_f32 (2 waves per SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NFMA, int NEXP, int NLDS>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters, float a0) {
  __shared__ float sh[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) sh[i] = 1e-6f * i;
  __syncthreads();
  f32x16 acc[4];
  for (int n = 0; n < 4; ++n)
    for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
  float a = a0 + threadIdx.x * 1e-7f, b = 1.0f;
  float v[8], e[8];
  for (int i = 0; i < 8; ++i) { v[i] = a0 * i; e[i] = a0 + i; }
  float l = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[n], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NFMA; ++j) v[j] = __builtin_fmaf(v[j], 1.0001f, 0.5f);
#pragma unroll
        for (int j = 0; j < NEXP; ++j) e[j] = __builtin_amdgcn_exp2f(e[j]);
#pragma unroll
        for (int j = 0; j < NLDS; ++j) l += sh[(threadIdx.x + 64 * j + 7 * u + n) & 4095];
      }
    }
  }
  float s = l;
  for (int n = 0; n < 4; ++n)
    for (int i = 0; i < 16; ++i) s += acc[n][i];
  for (int i = 0; i < 8; ++i) s += v[i] + e[i];
  if (s == 123.456f) out[0] = s;
}

template <int NFMA, int NEXP, int NLDS>
void run() {
  float* out;
  hipMalloc(&out, 4);
  const int iters = 2000, grid = 512;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<NFMA, NEXP, NLDS><<<grid, 256>>>(out, 10, 0.5f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<NFMA, NEXP, NLDS><<<grid, 256>>>(out, iters, 0.5f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double fl = (double)grid * 4 * iters * 64 * 4096.0;
  printf("per MFMA: %d fma + %d exp + %d ds_read_b32 : %.3f ms  %.1f TF/s\n", NFMA, NEXP, NLDS, ms, fl / ms / 1e9);
  hipFree(out);
}

int main() {
  run<0, 0, 0>(); run<2, 0, 0>(); run<4, 0, 0>(); run<6, 0, 0>(); run<8, 0, 0>();
  run<0, 1, 0>(); run<0, 2, 0>(); run<0, 4, 0>();
  run<2, 1, 0>(); run<4, 2, 0>();
  run<0, 0, 1>(); run<0, 0, 2>(); run<2, 1, 1>();
  return 0;
}
