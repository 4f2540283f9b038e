// Practical ceiling of v_mfma_f32_32x32x2_f32 on gfx950: pure MFMA loops at 1/2/3 waves per SIMD, with and without an
// LDS-fed B operand.  Build: hipcc -O3 --offload-arch=gfx950 mfma_peak.hip -o mfma_peak ; run: ./mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, bool LDS>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0) {
  __shared__ float sh[32 * 132];
  for (int i = threadIdx.x; i < 32 * 132; i += 256) sh[i] = 1e-6f * i;
  __syncthreads();
  f32x16 acc[NACC];
  for (int n = 0; n < NACC; ++n)
    for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
  float a = a0 + threadIdx.x * 1e-7f, b = 1.0f;
  const int lane = threadIdx.x & 63;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
#pragma unroll
      for (int n = 0; n < NACC; ++n) {
        if (LDS) b = sh[((u + (lane >> 5) * 16) * 132 + n * 32 + (lane & 31))];
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[n], 0, 0, 0);
      }
    }
  }
  float s = 0.f;
  for (int n = 0; n < NACC; ++n)
    for (int i = 0; i < 16; ++i) s += acc[n][i];
  if (s == 123.456f) out[0] = s;
}

template <int NACC, bool LDS>
void run(const char* name, int wg_per_cu) {
  float* out;
  hipMalloc(&out, 4);
  const int iters = 4000;
  const int grid = 256 * wg_per_cu;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<NACC, LDS><<<grid, 256>>>(out, 10, 0.5f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<NACC, LDS><<<grid, 256>>>(out, iters, 0.5f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double fl = (double)grid * 4 * iters * 16 * NACC * 4096.0;
  printf("%-28s wg/cu=%d  %.3f ms  %.1f TF/s\n", name, wg_per_cu, ms, fl / ms / 1e9);
  hipFree(out);
}

int main() {
  for (int w = 1; w <= 3; ++w) {
    run<1, false>("1 acc (dependent chain)", w);
    run<4, false>("4 acc, register operands", w);
    run<4, true>("4 acc, B operand from LDS", w);
  }
  return 0;
}
