// Optimiser-side kernels for gfx950 (HBM-bound streaming / row work):
//   * global grad-norm (clip_grad_norm_, reference src/training/train_embeddings.py:191) as
//     deterministic two-stage partial sums -> one device scalar, folded into Adam (no host sync)
//   * dense Adam with coupled L2 (torch.optim.Adam(weight_decay), train_embeddings.py:160,192)
//   * row-wise sparse path for tables that cannot take a dense pass per step (10M/100M rows):
//     sort ids -> segment-reduce per unique row (bitwise reproducible, no float atomics)
//     -> fused row Adam on touched rows only (deviation from the reference stated in DESIGN.md).
#include "common.h"
#include "recommendit_hip.h"

#include <cstring>
#include <string.h>
#include <rocprim/rocprim.hpp>

namespace {

constexpr int NPART = 64;  // partial sums written per rihip_sumsq call

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, int64_t n, double* part) {
  double acc = 0.0;
  const int64_t stride = (int64_t)gridDim.x * 256;
  const int64_t n4 = n / 4;
  const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    const f32x4 v = x4[i];
    acc += (double)(v.x * v.x + v.y * v.y) + (double)(v.z * v.z + v.w * v.w);
  }
  if (blockIdx.x == 0) {
    for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += 256) acc += (double)x[i] * (double)x[i];
  }
  acc = wave_sum_d(acc);
  __shared__ double sh[4];
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

// coef = min(1, max_norm / (sqrt(sum part) + 1e-6))  (torch clip_grad_norm_ semantics)
__global__ void clip_coef_kernel(const double* __restrict__ part, int n, float max_norm, float* coef, float* norm) {
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 64) s += part[i];
  s = wave_sum_d(s);
  if (threadIdx.x == 0) {
    const float tn = (float)sqrt(s);
    const float c = max_norm / (tn + 1e-6f);
    *coef = c < 1.f ? c : 1.f;
    if (norm) *norm = tn;
  }
}

// clip coefficient of this step + the step clock: hyper = {lr / (1 - b1^t), sqrt(1 - b2^t)} for t = *step (the step
// that is running), then *step = t + 1 for the next step's kernels (dropout seed offset, this routine).  One thread.
__device__ __forceinline__ void clip_and_clock(double sumsq, float max_norm, float* coef, float* norm, int64_t* step,
                                               const float* lr, float b1, float b2, float* hyper) {
  const float tn = (float)sqrt(sumsq);
  const float c = max_norm / (tn + 1e-6f);
  *coef = c < 1.f ? c : 1.f;
  if (norm) *norm = tn;
  if (step) {
    const int64_t t = *step;
    const double bc1 = 1.0 - pow((double)b1, (double)t);
    const double bc2 = 1.0 - pow((double)b2, (double)t);
    hyper[0] = (float)((double)(*lr) / bc1);
    hyper[1] = (float)sqrt(bc2);
    *step = t + 1;
  }
}
// 1024 threads: a step has up to a few thousand partials (MLP + two row-sparse tables) and this single workgroup sits
// on the step's critical path -- 64 threads took 14-22 us for them, latency-bound
__global__ __launch_bounds__(1024) void clip_coef_step_kernel(const double* __restrict__ part, int n, float max_norm,
                                                              float* coef, float* norm, int64_t* step, const float* lr,
                                                              float b1, float b2, float* hyper,
                                                              const double* __restrict__ loss_part, int n_loss,
                                                              double loss_scale, float* loss) {
  __shared__ double sh[2][16];
  const int tid = threadIdx.x;
  double s = 0.0, l = 0.0;
  for (int i = tid; i < n; i += 1024) s += part[i];
  if (loss_part)
    for (int i = tid; i < n_loss; i += 1024) l += loss_part[i];
  s = wave_sum_d(s);
  l = wave_sum_d(l);
  if ((tid & 63) == 0) { sh[0][tid >> 6] = s; sh[1][tid >> 6] = l; }
  __syncthreads();
  if (tid == 0) {
    double st = 0.0, lt = 0.0;
#pragma unroll
    for (int w = 0; w < 16; ++w) { st += sh[0][w]; lt += sh[1][w]; }   // fixed order
    clip_and_clock(st, max_norm, coef, norm, step, lr, b1, b2, hyper);
    if (loss_part) *loss = (float)(lt * loss_scale);   // the loss value of the step, off the critical path
  }
}

struct AdamHyper {
  float lr_over_bc1;   // lr / (1 - b1^t)
  float sqrt_bc2;      // sqrt(1 - b2^t)
  float b1, b2, eps, wd;
};

__device__ __forceinline__ void adam_elem(float& p, float g, float& m, float& v, float coef, const AdamHyper& h) {
  g = g * coef;
  if (h.wd != 0.f) g = g + h.wd * p;
  m = h.b1 * m + (1.f - h.b1) * g;
  v = h.b2 * v + (1.f - h.b2) * g * g;
  const float denom = sqrtf(v) / h.sqrt_bc2 + h.eps;
  p = p - h.lr_over_bc1 * (m / denom);
}

// device-resident step clock for graph replay: the bias-corrected step size of THIS step is computed on the device
// from a step counter that the kernel itself advances (host constants baked into a captured graph would freeze t)
__global__ void adam_hyper_kernel(int64_t* step, const float* lr, float b1, float b2, float* hyper) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const int64_t t = *step + 1;
    *step = t;
    const double bc1 = 1.0 - pow((double)b1, (double)t);
    const double bc2 = 1.0 - pow((double)b2, (double)t);
    hyper[0] = (float)((double)(*lr) / bc1);
    hyper[1] = (float)sqrt(bc2);
  }
}

__global__ __launch_bounds__(256) void adam_dense_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                         float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                         const float* coef_dev, AdamHyper h, const float* hyper_dev) {
  if (hyper_dev) { h.lr_over_bc1 = hyper_dev[0]; h.sqrt_bc2 = hyper_dev[1]; }
  const float coef = coef_dev ? *coef_dev : 1.f;
  const int64_t stride = (int64_t)gridDim.x * 256;
  const int64_t n4 = n / 4;
  f32x4* p4 = reinterpret_cast<f32x4*>(p);
  const f32x4* g4 = reinterpret_cast<const f32x4*>(g);
  f32x4* m4 = reinterpret_cast<f32x4*>(m);
  f32x4* v4 = reinterpret_cast<f32x4*>(v);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    f32x4 pp = p4[i], gg = g4[i], mm = m4[i], vv = v4[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float pk = pp[k], mk = mm[k], vk = vv[k];
      adam_elem(pk, gg[k], mk, vk, coef, h);
      pp[k] = pk; mm[k] = mk; vv[k] = vk;
    }
    p4[i] = pp; m4[i] = mm; v4[i] = vv;
  }
  if (blockIdx.x == 0) {
    for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += 256) adam_elem(p[i], g[i], m[i], v[i], coef, h);
  }
}

// ---------------------------------- row-sparse path ---------------------------------------
__global__ void iota_kernel(int* v, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] = (int)i;
}
__global__ void head_flags_kernel(const int64_t* __restrict__ sorted, int64_t n, int* flags) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) flags[i] = (i == 0 || sorted[i] != sorted[i - 1]) ? 1 : 0;
}
// seg[i] = inclusive scan of head flags (1-based segment number); write segment starts + unique ids
// an id outside [0, n_rows) (already flagged by the tower forward's err word) becomes the padding row 0: its gradient is
// dropped and the row-sparse Adam never touches memory outside the table
__global__ void seg_starts_kernel(const int64_t* __restrict__ sorted, const int* __restrict__ seg, int64_t n,
                                  int64_t n_rows, int* seg_start, int64_t* uniq, int* n_unique) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int s = seg[i] - 1;
  if (i == 0 || seg[i] != seg[i - 1]) {
    seg_start[s] = (int)i;
    const int64_t id = sorted[i];
    uniq[s] = (id >= 0 && (n_rows <= 0 || id < n_rows)) ? id : 0;
  }
  if (i == n - 1) {
    seg_start[s + 1] = (int)n;
    *n_unique = s + 1;
  }
}

// Segmented row sum in two deterministic passes (a popular item can own thousands of samples of a batch;
// one wave per segment would serialise on it):
//  (1) one wave per block of RB consecutive SORTED positions sums each run of equal ids inside its block
//      and stores the partial at P[position where the run starts inside the block];
//  (2) one wave per unique row adds its partials (s0, then every multiple of RB inside the segment) in order.
constexpr int RB = 32;

__global__ __launch_bounds__(256) void rows_partial_kernel(const float* __restrict__ dX, const int* __restrict__ perm,
                                                           const int64_t* __restrict__ sorted, int64_t B, int d,
                                                           float* __restrict__ P) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t nblk = (B + RB - 1) / RB;
  for (int64_t b = (int64_t)blockIdx.x * 4 + w; b < nblk; b += (int64_t)gridDim.x * 4) {
    const int64_t p0 = b * RB, p1 = (p0 + RB < B) ? p0 + RB : B;
    for (int c = lane; c < d; c += 64) {
      float acc = 0.f;
      int64_t run = p0;
      int64_t prev = sorted[p0];
#pragma unroll 4
      for (int64_t i = p0; i < p1; ++i) {
        const int64_t k = sorted[i];
        const float v = dX[(size_t)perm[i] * d + c];
        if (k != prev) {
          P[(size_t)run * d + c] = acc;
          run = i;
          acc = v;
          prev = k;
        } else {
          acc += v;
        }
      }
      P[(size_t)run * d + c] = acc;
    }
  }
}

__global__ __launch_bounds__(256) void rows_combine_kernel(const float* __restrict__ P, const int* __restrict__ seg_start,
                                                           const int64_t* __restrict__ uniq,
                                                           const int* __restrict__ n_unique, int d, float* Gc,
                                                           double* part) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int nu = *n_unique;
  double acc = 0.0;
  for (int u = blockIdx.x * 4 + w; u < nu; u += gridDim.x * 4) {
    const int s0 = seg_start[u], s1 = seg_start[u + 1];
    const bool pad = uniq[u] == 0;  // padding_idx row: gradient forced to zero (nn.Embedding semantics)
    const int q0 = (s0 / RB + 1) * RB;
    for (int c = lane; c < d; c += 64) {
      float s = P[(size_t)s0 * d + c];
#pragma unroll 8
      for (int q = pad ? s1 : q0; q < s1; q += RB) s += P[(size_t)q * d + c];
      if (pad) s = 0.f;
      Gc[(size_t)u * d + c] = s;
      acc += (double)s * (double)s;
    }
  }
  acc = wave_sum_d(acc);
  __shared__ double sh[4];
  if (lane == 0) sh[w] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ __launch_bounds__(256) void adam_rows_kernel(float* __restrict__ table, float* __restrict__ m,
                                                        float* __restrict__ v, const int64_t* __restrict__ uniq,
                                                        const float* __restrict__ Gc,
                                                        const int* __restrict__ n_unique, int d,
                                                        const float* coef_dev, AdamHyper h, const float* hyper_dev) {
  if (hyper_dev) { h.lr_over_bc1 = hyper_dev[0]; h.sqrt_bc2 = hyper_dev[1]; }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int nu = *n_unique;
  const float coef = coef_dev ? *coef_dev : 1.f;
  for (int u = blockIdx.x * 4 + w; u < nu; u += gridDim.x * 4) {
    const size_t row = (size_t)uniq[u] * d;
    for (int c = lane; c < d; c += 64) {
      float p = table[row + c], mm = m[row + c], vv = v[row + c];
      adam_elem(p, Gc[(size_t)u * d + c], mm, vv, coef, h);
      table[row + c] = p; m[row + c] = mm; v[row + c] = vv;
    }
  }
}

// ---- float4 forms for d in {32, 64, 128}: LPR = d/4 lanes per row, 64/LPR rows (or blocks of sorted positions) per
// wave side by side; per element the same operations in the same order as the scalar kernels above.
template <int LPR>
__global__ __launch_bounds__(256) void rows_partial_v4_kernel(const float* __restrict__ dX, const int* __restrict__ perm,
                                                              const int64_t* __restrict__ sorted, int64_t B,
                                                              float* __restrict__ P) {
  constexpr int RPW = 64 / LPR, d = LPR * 4;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int sub = lane / LPR, c4 = lane % LPR;
  const int64_t nblk = (B + RB - 1) / RB;
  for (int64_t b = ((int64_t)blockIdx.x * 4 + w) * RPW + sub; b < nblk; b += (int64_t)gridDim.x * 4 * RPW) {
    const int64_t p0 = b * RB, p1 = (p0 + RB < B) ? p0 + RB : B;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int64_t run = p0;
    int64_t prev = sorted[p0];
#pragma unroll 4
    for (int64_t i = p0; i < p1; ++i) {
      const int64_t k = sorted[i];
      const f32x4 v = reinterpret_cast<const f32x4*>(dX + (size_t)perm[i] * d)[c4];
      if (k != prev) {
        reinterpret_cast<f32x4*>(P + (size_t)run * d)[c4] = acc;
        run = i;
        acc = v;
        prev = k;
      } else {
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
    }
    reinterpret_cast<f32x4*>(P + (size_t)run * d)[c4] = acc;
  }
}

template <int LPR>
__global__ __launch_bounds__(256) void rows_combine_v4_kernel(const float* __restrict__ P,
                                                              const int* __restrict__ seg_start,
                                                              const int64_t* __restrict__ uniq,
                                                              const int* __restrict__ n_unique, float* Gc,
                                                              double* part) {
  constexpr int RPW = 64 / LPR, d = LPR * 4;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int sub = lane / LPR, c4 = lane % LPR;
  const int nu = *n_unique;
  double acc = 0.0;
  for (int u = (blockIdx.x * 4 + w) * RPW + sub; u < nu; u += gridDim.x * 4 * RPW) {
    const int s0 = seg_start[u], s1 = seg_start[u + 1];
    const bool pad = uniq[u] == 0;  // padding_idx row: gradient forced to zero (nn.Embedding semantics)
    const int q0 = (s0 / RB + 1) * RB;
    f32x4 s = reinterpret_cast<const f32x4*>(P + (size_t)s0 * d)[c4];
    // (the padding row's segment can be tens of thousands of slots long in the fixed-capacity row exchange: not summed)
#pragma unroll 4
    for (int q = pad ? s1 : q0; q < s1; q += RB) {
      const f32x4 v = reinterpret_cast<const f32x4*>(P + (size_t)q * d)[c4];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (pad) s = f32x4{0.f, 0.f, 0.f, 0.f};
    reinterpret_cast<f32x4*>(Gc + (size_t)u * d)[c4] = s;
    acc += ((double)s.x * (double)s.x + (double)s.y * (double)s.y) + ((double)s.z * (double)s.z + (double)s.w * (double)s.w);
  }
  acc = wave_sum_d(acc);
  __shared__ double sh[4];
  if (lane == 0) sh[w] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

template <int LPR>
__global__ __launch_bounds__(256) void adam_rows_v4_kernel(float* __restrict__ table, float* __restrict__ m,
                                                           float* __restrict__ v, const int64_t* __restrict__ uniq,
                                                           const float* __restrict__ Gc,
                                                           const int* __restrict__ n_unique, const float* coef_dev,
                                                           AdamHyper h, const float* hyper_dev) {
  constexpr int RPW = 64 / LPR, d = LPR * 4;
  if (hyper_dev) { h.lr_over_bc1 = hyper_dev[0]; h.sqrt_bc2 = hyper_dev[1]; }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int sub = lane / LPR, c4 = lane % LPR;
  const int nu = *n_unique;
  const float coef = coef_dev ? *coef_dev : 1.f;
  for (int u = (blockIdx.x * 4 + w) * RPW + sub; u < nu; u += gridDim.x * 4 * RPW) {
    const size_t row = (size_t)uniq[u] * d;
    f32x4 pp = reinterpret_cast<f32x4*>(table + row)[c4], mm = reinterpret_cast<f32x4*>(m + row)[c4],
          vv = reinterpret_cast<f32x4*>(v + row)[c4];
    const f32x4 gg = reinterpret_cast<const f32x4*>(Gc + (size_t)u * d)[c4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float pk = pp[k], mk = mm[k], vk = vv[k];
      adam_elem(pk, gg[k], mk, vk, coef, h);
      pp[k] = pk; mm[k] = mk; vv[k] = vk;
    }
    reinterpret_cast<f32x4*>(table + row)[c4] = pp;
    reinterpret_cast<f32x4*>(m + row)[c4] = mm;
    reinterpret_cast<f32x4*>(v + row)[c4] = vv;
  }
}

AdamHyper make_hyper(float lr, float b1, float b2, float eps, float wd, int64_t step) {
  AdamHyper h;
  const double bc1 = 1.0 - pow((double)b1, (double)step);
  const double bc2 = 1.0 - pow((double)b2, (double)step);
  h.lr_over_bc1 = (float)((double)lr / bc1);
  h.sqrt_bc2 = (float)sqrt(bc2);
  h.b1 = b1; h.b2 = b2; h.eps = eps; h.wd = wd;
  return h;
}

constexpr int ROWS_GRID = 1024;

}  // namespace

// ---- multi-tensor forms (small-batch steps are bounded by dependent kernel boundaries: one launch for up to 4
// tensors; blockIdx.y picks the tensor, the arithmetic and the partial layout equal the single-tensor calls)
struct MultiDesc {
  float* p[4];
  float* g[4];
  float* m[4];
  float* v[4];
  int64_t n[4];
  int zero_grad_mask;
};
__global__ __launch_bounds__(256) void sumsq_multi_kernel(MultiDesc d, double* part) {
  const int t = blockIdx.y;
  const float* __restrict__ x = d.g[t];
  const int64_t n = d.n[t];
  double acc = 0.0;
  const int64_t stride = (int64_t)gridDim.x * 256;
  const int64_t n4 = n / 4;
  const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    const f32x4 v = x4[i];
    acc += (double)(v.x * v.x + v.y * v.y) + (double)(v.z * v.z + v.w * v.w);
  }
  if (blockIdx.x == 0) {
    for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += 256) acc += (double)x[i] * (double)x[i];
  }
  acc = wave_sum_d(acc);
  __shared__ double sh[4];
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[(size_t)t * NPART + blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}
__global__ __launch_bounds__(256) void adam_dense_multi_kernel(MultiDesc d, const float* coef_dev, AdamHyper h,
                                                               const float* hyper_dev) {
  const int t = blockIdx.y;
  if (hyper_dev) { h.lr_over_bc1 = hyper_dev[0]; h.sqrt_bc2 = hyper_dev[1]; }
  const float coef = coef_dev ? *coef_dev : 1.f;
  const bool zg = (d.zero_grad_mask >> t) & 1;
  float* __restrict__ p = d.p[t];
  float* __restrict__ g = d.g[t];
  float* __restrict__ m = d.m[t];
  float* __restrict__ v = d.v[t];
  const int64_t n = d.n[t];
  const int64_t stride = (int64_t)gridDim.x * 256;
  const int64_t n4 = n / 4;
  f32x4* p4 = reinterpret_cast<f32x4*>(p);
  f32x4* g4 = reinterpret_cast<f32x4*>(g);
  f32x4* m4 = reinterpret_cast<f32x4*>(m);
  f32x4* v4 = reinterpret_cast<f32x4*>(v);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    f32x4 pp = p4[i], gg = g4[i], mm = m4[i], vv = v4[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float pk = pp[k], mk = mm[k], vk = vv[k];
      adam_elem(pk, gg[k], mk, vk, coef, h);
      pp[k] = pk; mm[k] = mk; vv[k] = vk;
    }
    p4[i] = pp; m4[i] = mm; v4[i] = vv;
    if (zg) g4[i] = f32x4{0.f, 0.f, 0.f, 0.f};  // the dense table gradient is consumed: next step scatters into zeros
  }
  if (blockIdx.x == 0) {
    for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += 256) {
      adam_elem(p[i], g[i], m[i], v[i], coef, h);
      if (zg) g[i] = 0.f;
    }
  }
}

extern "C" int rihip_sumsq_nparts(void) { return NPART; }
extern "C" int rihip_rows_nparts(void) { return ROWS_GRID; }

extern "C" int rihip_sumsq(const float* x, int64_t n, double* part, void* stream) {
  RIHIP_REQUIRE(x && part && n >= 0, RIHIP_ERR_ARG, "sumsq: bad arguments");
  RIHIP_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0, RIHIP_ERR_ARG, "sumsq: x must be 16-byte aligned");
  hipLaunchKernelGGL(sumsq_kernel, dim3(NPART), dim3(256), 0, (hipStream_t)stream, x, n, part);
  RIHIP_CHECK_LAUNCH();
  return RIHIP_OK;
}

extern "C" int rihip_sumsq_multi(int n_tensors, const float* const* x, const int64_t* n, double* part, void* stream) {
  RIHIP_REQUIRE(n_tensors >= 1 && n_tensors <= 4 && x && n && part, RIHIP_ERR_ARG, "sumsq_multi: bad arguments");
  MultiDesc d;
  memset(&d, 0, sizeof(d));
  for (int t = 0; t < n_tensors; ++t) {
    RIHIP_REQUIRE(x[t] && n[t] >= 0 && (reinterpret_cast<uintptr_t>(x[t]) & 15) == 0, RIHIP_ERR_ARG,
                  "sumsq_multi: tensor %d null, negative size or not 16-byte aligned", t);
    d.g[t] = const_cast<float*>(x[t]);
    d.n[t] = n[t];
  }
  hipLaunchKernelGGL(sumsq_multi_kernel, dim3(NPART, n_tensors), dim3(256), 0, (hipStream_t)stream, d, part);
  RIHIP_CHECK_LAUNCH();
  return RIHIP_OK;
}

extern "C" int rihip_adam_dense_multi(int n_tensors, float* const* p, float* const* g, float* const* m,
                                      float* const* v, const int64_t* n, int zero_grad_mask, float lr, float beta1,
                                      float beta2, float eps, float weight_decay, int64_t step,
                                      const float* clip_coef, const float* hyper_dev, void* stream) {
  RIHIP_REQUIRE(n_tensors >= 1 && n_tensors <= 4 && p && g && m && v && n && (step >= 1 || hyper_dev), RIHIP_ERR_ARG,
                "adam_dense_multi: bad arguments");
  MultiDesc d;
  memset(&d, 0, sizeof(d));
  int64_t nmax = 0;
  for (int t = 0; t < n_tensors; ++t) {
    RIHIP_REQUIRE(p[t] && g[t] && m[t] && v[t] && n[t] >= 0, RIHIP_ERR_ARG, "adam_dense_multi: tensor %d null", t);
    RIHIP_REQUIRE(((reinterpret_cast<uintptr_t>(p[t]) | reinterpret_cast<uintptr_t>(g[t]) |
                    reinterpret_cast<uintptr_t>(m[t]) | reinterpret_cast<uintptr_t>(v[t])) & 15) == 0,
                  RIHIP_ERR_ARG, "adam_dense_multi: pointers must be 16-byte aligned");
    d.p[t] = p[t]; d.g[t] = g[t]; d.m[t] = m[t]; d.v[t] = v[t]; d.n[t] = n[t];
    if (n[t] > nmax) nmax = n[t];
  }
  d.zero_grad_mask = zero_grad_mask;
  if (nmax == 0) return RIHIP_OK;
  const int64_t nb = (nmax / 4 + 255) / 256;
  const int grid = (int)(nb < 1 ? 1 : (nb < 2048 ? nb : 2048));
  hipLaunchKernelGGL(adam_dense_multi_kernel, dim3(grid, n_tensors), dim3(256), 0, (hipStream_t)stream, d, clip_coef,
                     make_hyper(lr, beta1, beta2, eps, weight_decay, step >= 1 ? step : 1), hyper_dev);
  RIHIP_CHECK_LAUNCH();
  return RIHIP_OK;
}

extern "C" int rihip_clip_coef(const double* part, int64_t n_part, float max_norm, float* coef, float* total_norm,
                               void* stream) {
  RIHIP_REQUIRE(part && coef && n_part > 0, RIHIP_ERR_ARG, "clip_coef: bad arguments");
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, part, (int)n_part, max_norm, coef,
                     total_norm);
  RIHIP_CHECK_LAUNCH();
  return RIHIP_OK;
}

extern "C" int rihip_clip_coef_step(const double* part, int64_t n_part, float max_norm, float* coef, float* total_norm,
                                    int64_t* step_dev, const float* lr_dev, float beta1, float beta2, float* hyper_dev,
                                    const double* loss_part, int64_t n_loss_part, double loss_scale, float* loss,
                                    void* stream) {
  RIHIP_REQUIRE(part && coef && n_part > 0 && step_dev && lr_dev && hyper_dev, RIHIP_ERR_ARG, "clip_coef_step: bad arguments");
  RIHIP_REQUIRE(!loss_part || (loss && n_loss_part > 0), RIHIP_ERR_ARG, "clip_coef_step: loss partials without an output");
  hipLaunchKernelGGL(clip_coef_step_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, part, (int)n_part, max_norm, coef,
                     total_norm, step_dev, lr_dev, beta1, beta2, hyper_dev, loss_part, (int)n_loss_part, loss_scale, loss);
  RIHIP_CHECK_LAUNCH();
  return RIHIP_OK;
}

extern "C" int rihip_adam_dense(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                                float beta2, float eps, float weight_decay, int64_t step, const float* clip_coef,
                                const float* hyper_dev, void* stream) {
  RIHIP_REQUIRE(p && g && m && v && n >= 0 && (step >= 1 || hyper_dev), RIHIP_ERR_ARG, "adam_dense: bad arguments");
  RIHIP_REQUIRE(((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                  reinterpret_cast<uintptr_t>(v)) & 15) == 0,
                RIHIP_ERR_ARG, "adam_dense: pointers must be 16-byte aligned");
  if (n == 0) return RIHIP_OK;
  const int64_t nb = (n / 4 + 255) / 256;
  const int grid = (int)(nb < 1 ? 1 : (nb < 2048 ? nb : 2048));
  hipLaunchKernelGGL(adam_dense_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, clip_coef,
                     make_hyper(lr, beta1, beta2, eps, weight_decay, step >= 1 ? step : 1), hyper_dev);
  RIHIP_CHECK_LAUNCH();
  return RIHIP_OK;
}

extern "C" int rihip_adam_hyper_step(int64_t* step_dev, const float* lr_dev, float beta1, float beta2, float* hyper_dev,
                                     void* stream) {
  RIHIP_REQUIRE(step_dev && lr_dev && hyper_dev, RIHIP_ERR_ARG, "adam_hyper_step: null pointer");
  hipLaunchKernelGGL(adam_hyper_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, step_dev, lr_dev, beta1, beta2, hyper_dev);
  RIHIP_CHECK_LAUNCH();
  return RIHIP_OK;
}

// ---- row-sparse path ------------------------------------------------------------------------
// workspace layout (bytes), all 256-B aligned:
//   keys_out int64[B] | vals_in int32[B] | perm int32[B] | flags int32[B] | seg int32[B] |
//   seg_start int32[B+1] | n_unique int32 | rocprim temp | P float[B,d] (block partials)
static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct RowsWs {
  int64_t* keys_out; int* vals_in; int* perm; int* flags; int* seg; int* seg_start; int* n_unique; void* temp;
  float* P; size_t temp_bytes; size_t total;
};

static int rows_ws_layout(int64_t B, int d, void* base, RowsWs* ws) {
  size_t sort_bytes = 0, scan_bytes = 0;
  hipError_t e = rocprim::radix_sort_pairs(nullptr, sort_bytes, (const int64_t*)nullptr, (int64_t*)nullptr,
                                           (const int*)nullptr, (int*)nullptr, (size_t)B);
  if (e != hipSuccess) return RIHIP_ERR_HIP;
  e = rocprim::inclusive_scan(nullptr, scan_bytes, (const int*)nullptr, (int*)nullptr, (size_t)B, rocprim::plus<int>());
  if (e != hipSuccess) return RIHIP_ERR_HIP;
  size_t off = 0;
  char* b = (char*)base;
  ws->keys_out = (int64_t*)(b + off); off += align256(sizeof(int64_t) * B);
  ws->vals_in = (int*)(b + off); off += align256(sizeof(int) * B);
  ws->perm = (int*)(b + off); off += align256(sizeof(int) * B);
  ws->flags = (int*)(b + off); off += align256(sizeof(int) * B);
  ws->seg = (int*)(b + off); off += align256(sizeof(int) * B);
  ws->seg_start = (int*)(b + off); off += align256(sizeof(int) * (B + 1));
  ws->n_unique = (int*)(b + off); off += 256;
  ws->temp = (void*)(b + off);
  ws->temp_bytes = sort_bytes > scan_bytes ? sort_bytes : scan_bytes;
  off += align256(ws->temp_bytes);
  ws->P = (float*)(b + off); off += align256(sizeof(float) * (size_t)B * d);
  ws->total = off;
  return RIHIP_OK;
}

extern "C" int64_t rihip_rows_workspace_bytes(int64_t B, int d) {
  RowsWs ws;
  if (B <= 0 || d <= 0 || rows_ws_layout(B, d, nullptr, &ws) != RIHIP_OK) return -1;
  return (int64_t)ws.total;
}

// Groups the B (id, sample) pairs by id.  Outputs (device): uniq[<=B] unique ids ascending,
// and inside the workspace perm / seg_start / n_unique used by rihip_rows_reduce / rihip_adam_rows.
extern "C" int rihip_rows_group(const int64_t* ids, int64_t B, int d, int64_t n_rows, int64_t* uniq, void* workspace,
                                int64_t workspace_bytes, void* stream) {
  RIHIP_REQUIRE(ids && uniq && workspace && B > 0 && B < (1ll << 31), RIHIP_ERR_ARG, "rows_group: bad arguments");
  RowsWs ws;
  RIHIP_REQUIRE(rows_ws_layout(B, d, workspace, &ws) == RIHIP_OK, RIHIP_ERR_HIP, "rows_group: rocprim size query failed");
  RIHIP_REQUIRE((int64_t)ws.total <= workspace_bytes, RIHIP_ERR_ARG, "rows_group: workspace too small (%zu > %lld)",
                ws.total, (long long)workspace_bytes);
  hipStream_t st = (hipStream_t)stream;
  const unsigned nb = (unsigned)((B + 255) / 256);
  hipLaunchKernelGGL(iota_kernel, dim3(nb), dim3(256), 0, st, ws.vals_in, B);
  RIHIP_CHECK_LAUNCH();
  size_t tb = ws.temp_bytes;
  unsigned end_bit = 64;  // ids are non-negative row numbers < n_rows: sort only the bits that can differ
  if (n_rows > 0) {
    end_bit = 1;
    while (end_bit < 63 && (1ll << end_bit) < n_rows) ++end_bit;
  }
  RIHIP_CHECK_HIP(rocprim::radix_sort_pairs(ws.temp, tb, ids, ws.keys_out, ws.vals_in, ws.perm, (size_t)B, 0, end_bit, st));
  hipLaunchKernelGGL(head_flags_kernel, dim3(nb), dim3(256), 0, st, ws.keys_out, B, ws.flags);
  RIHIP_CHECK_LAUNCH();
  tb = ws.temp_bytes;
  RIHIP_CHECK_HIP(rocprim::inclusive_scan(ws.temp, tb, ws.flags, ws.seg, (size_t)B, rocprim::plus<int>(), st));
  hipLaunchKernelGGL(seg_starts_kernel, dim3(nb), dim3(256), 0, st, ws.keys_out, ws.seg, B, n_rows, ws.seg_start, uniq,
                     ws.n_unique);
  RIHIP_CHECK_LAUNCH();
  return RIHIP_OK;
}

extern "C" int rihip_rows_n_unique_ptr(void* workspace, int64_t B, int d, const int** out) {
  RowsWs ws;
  RIHIP_REQUIRE(rows_ws_layout(B, d, workspace, &ws) == RIHIP_OK, RIHIP_ERR_HIP, "rows: size query failed");
  *out = ws.n_unique;
  return RIHIP_OK;
}

// Gc[u] = sum of dX rows of unique row u; part[rihip_rows_nparts()] = partial sums of |Gc|^2
extern "C" int rihip_rows_reduce(const float* dX, int64_t B, int d, const int64_t* uniq, void* workspace, float* Gc,
                                 double* part, void* stream) {
  RIHIP_REQUIRE(dX && uniq && workspace && Gc && part && B > 0, RIHIP_ERR_ARG, "rows_reduce: bad arguments");
  RowsWs ws;
  RIHIP_REQUIRE(rows_ws_layout(B, d, workspace, &ws) == RIHIP_OK, RIHIP_ERR_HIP, "rows_reduce: size query failed");
  const int64_t nblk = (B + RB - 1) / RB;
  const int g1 = (int)((nblk + 3) / 4 < 4096 ? (nblk + 3) / 4 : 4096);
  hipStream_t st = (hipStream_t)stream;
  const bool al = ((reinterpret_cast<uintptr_t>(dX) | reinterpret_cast<uintptr_t>(ws.P) | reinterpret_cast<uintptr_t>(Gc)) & 15) == 0;
  if (al && d == 128) {
    hipLaunchKernelGGL((rows_partial_v4_kernel<32>), dim3(g1), dim3(256), 0, st, dX, ws.perm, ws.keys_out, B, ws.P);
    hipLaunchKernelGGL((rows_combine_v4_kernel<32>), dim3(ROWS_GRID), dim3(256), 0, st, ws.P, ws.seg_start, uniq, ws.n_unique, Gc, part);
  } else if (al && d == 64) {
    hipLaunchKernelGGL((rows_partial_v4_kernel<16>), dim3(g1), dim3(256), 0, st, dX, ws.perm, ws.keys_out, B, ws.P);
    hipLaunchKernelGGL((rows_combine_v4_kernel<16>), dim3(ROWS_GRID), dim3(256), 0, st, ws.P, ws.seg_start, uniq, ws.n_unique, Gc, part);
  } else if (al && d == 32) {
    hipLaunchKernelGGL((rows_partial_v4_kernel<8>), dim3(g1), dim3(256), 0, st, dX, ws.perm, ws.keys_out, B, ws.P);
    hipLaunchKernelGGL((rows_combine_v4_kernel<8>), dim3(ROWS_GRID), dim3(256), 0, st, ws.P, ws.seg_start, uniq, ws.n_unique, Gc, part);
  } else {
    hipLaunchKernelGGL(rows_partial_kernel, dim3(g1), dim3(256), 0, st, dX, ws.perm, ws.keys_out, B, d, ws.P);
    hipLaunchKernelGGL(rows_combine_kernel, dim3(ROWS_GRID), dim3(256), 0, st, ws.P, ws.seg_start, uniq, ws.n_unique, d, Gc, part);
  }
  RIHIP_CHECK_LAUNCH();
  return RIHIP_OK;
}

extern "C" int rihip_adam_rows(float* table, float* m, float* v, const int64_t* uniq, const float* Gc, int64_t B,
                               int d, void* workspace, float lr, float beta1, float beta2, float eps,
                               float weight_decay, int64_t step, const float* clip_coef, const float* hyper_dev,
                               void* stream) {
  RIHIP_REQUIRE(table && m && v && uniq && Gc && workspace && B > 0 && (step >= 1 || hyper_dev), RIHIP_ERR_ARG,
                "adam_rows: bad arguments");
  RowsWs ws;
  RIHIP_REQUIRE(rows_ws_layout(B, d, workspace, &ws) == RIHIP_OK, RIHIP_ERR_HIP, "adam_rows: size query failed");
  const AdamHyper hy = make_hyper(lr, beta1, beta2, eps, weight_decay, step >= 1 ? step : 1);
  hipStream_t st = (hipStream_t)stream;
  const bool al = ((reinterpret_cast<uintptr_t>(table) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v) |
                    reinterpret_cast<uintptr_t>(Gc)) & 15) == 0;
  if (al && d == 128) hipLaunchKernelGGL((adam_rows_v4_kernel<32>), dim3(ROWS_GRID), dim3(256), 0, st, table, m, v, uniq, Gc, ws.n_unique, clip_coef, hy, hyper_dev);
  else if (al && d == 64) hipLaunchKernelGGL((adam_rows_v4_kernel<16>), dim3(ROWS_GRID), dim3(256), 0, st, table, m, v, uniq, Gc, ws.n_unique, clip_coef, hy, hyper_dev);
  else if (al && d == 32) hipLaunchKernelGGL((adam_rows_v4_kernel<8>), dim3(ROWS_GRID), dim3(256), 0, st, table, m, v, uniq, Gc, ws.n_unique, clip_coef, hy, hyper_dev);
  else hipLaunchKernelGGL(adam_rows_kernel, dim3(ROWS_GRID), dim3(256), 0, st, table, m, v, uniq, Gc, ws.n_unique, d, clip_coef, hy, hyper_dev);
  RIHIP_CHECK_LAUNCH();
  return RIHIP_OK;
}
