// The whole small-batch training step of the reference's hot loop in ONE cooperative launch
// (src/training/train_embeddings.py:178-195: towers -> bpr_loss -> backward -> clip_grad_norm_ -> Adam.step, batch
// size BATCH_SIZE = 1024 by default, src/config.py:25; 256 in BASELINE configs[0]).
//
// At these sizes a step is a few microseconds of arithmetic behind 7 dependent launches of ~10 us each.  Here one
// persistent grid (one workgroup per CU) walks the step in four phases separated by THREE grid barriers:
//   A  towers forward: one 32-row tile per workgroup (user tiles, then the pos || neg item tiles)
//   -- barrier --
//   B  per tower tile: loss gradient of its rows (sampled BPR is row-local: dU_b, dP_b, dN_b need only U_b, P_b, N_b),
//      then the data gradients gy, dPre, dX; the user tiles also sum softplus(-delta)
//   -- barrier --
//   C  weight gradients (one output tile per workgroup, its four waves split the batch, fixed-order LDS combine, written
//      straight into the gradient tensors) and the dense embedding-gradient scatter (table rows owned modulo the grid,
//      every row's samples added in batch order); each workgroup leaves the squared norm of what it produced
//   -- barrier --
//   D  clip coefficient from the per-workgroup partial norms (every workgroup, same order), dense Adam + coupled L2 over
//      the MLPs and BOTH tables (the reference's optimiser: every row decays every step), table gradients left zero,
//      step clock advanced, loss written.
// The tower arithmetic is the runtime-shape code of tower_generic_body.h (any embed_dim / hidden_dim it covers).
// Barriers: one arrival counter per barrier; a counter is reset by workgroup 0 once the NEXT barrier has been passed
// (nobody can still be spinning on it; the last one is reset by the next launch).  A spin that does not complete within ~1 s sets err_flag bit 3 and lets every
// wave leave, so a grid that was not co-resident cannot hang the device.
#include "recommendit_hip.h"
#ifdef RIHIP_STEP_PROBE
#include <hip/hip_runtime.h>
__device__ unsigned long long g_rihip_probe[8];
#define RIHIP_TILE_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_rihip_probe[k] = wall_clock64(); } while (0)
#endif
#include "tower_generic_body.h"

using namespace rihip_gen;

namespace {

constexpr int STEP_WG = 256;        // threads per workgroup
constexpr int SCAT_LIST = 4096;     // positions one workgroup can own in the scatter phase (LDS list)

struct AdamH { float b1, b2, eps, wd; };

struct StepArgs {
  GenFwd fu, fi;
  GenBwd bu, bi;
  int64_t B;                 // pairs; the item tower sees 2B rows (pos || neg)
  float *gW1u, *gb1u, *gW2u, *gb2u, *gW1i, *gb1i, *gW2i, *gb2i;
  float *flat_p, *flat_g, *flat_m, *flat_v; int64_t n_flat;
  float *utab, *utab_g, *utab_m, *utab_v; int64_t n_urows;
  float *itab, *itab_g, *itab_m, *itab_v; int64_t n_irows;
  AdamH h; float max_norm;
  const float* lr_dev; int64_t* step_dev; float* hyper_dev; float* coef; float* gnorm; float* loss; int* err_flag;
  double* loss_part;   // [n user tiles]
  double* sq_part;     // [gridDim.x]
  double* stamps;      // [8] phase time stamps of workgroup 0 (100 MHz wall clock ticks): start, A, b0, B, b1, C, b2, D
  unsigned* bar;       // [4]
  size_t lds_floats;
};

// Grid barrier.  The 8 XCDs have private L2s and ordinary (coarse-grained) device memory is only made coherent between
// them by an explicit L2 write-back (release) / invalidate (acquire), which are expensive: so exactly ONE thread per
// workgroup fences, after the workgroup barrier has drained every wave's stores into the L2 and before the other waves
// read anything; the spin itself uses relaxed loads (an acquire load per poll would invalidate the L2 every iteration).
__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned n_wg, int* err_flag) {
  __syncthreads();       // all waves: stores of this phase issued and complete (s_waitcnt vmcnt(0) before s_barrier)
  __shared__ int ok_s;
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int ok = 1;
    long long spins = 0;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < n_wg) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1ll << 24)) { ok = 0; break; }     // ~1 s: the grid is not co-resident -- give up instead of hanging
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (!ok && err_flag) atomicOr(err_flag, 8);
    ok_s = ok;
  }
  __syncthreads();
  return ok_s != 0;
}

__device__ __forceinline__ void adam_elem_p(float& p, float g, float& m, float& v, float coef, float lr_over_bc1,
                                            float sqrt_bc2, const AdamH& h) {
  g = g * coef;
  if (h.wd != 0.f) g = g + h.wd * p;
  m = h.b1 * m + (1.f - h.b1) * g;
  v = h.b2 * v + (1.f - h.b2) * g * g;
  const float denom = sqrtf(v) / sqrt_bc2 + h.eps;
  p = p - lr_over_bc1 * (m / denom);
}

// loss gradient rows of one tower tile (tower 0 = user rows b; tower 1 = item rows r: pos r < B, neg r >= B).  Each wave
// owns 8 of the 32 rows and handles them TOGETHER: all U / P / N loads of the 8 rows are issued before the first
// reduction (three dependent latencies per tile instead of three per row).  User tiles also return the sum of
// softplus(-delta) of the wave's rows (lane 0).
__device__ __forceinline__ double loss_rows(const StepArgs& s, int tower, int64_t tile) {
  const int D = s.fu.D;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const float* U = s.fu.a.out;
  const float* I = s.fi.a.out;
  const float invB = 1.f / (float)s.B;
  const int64_t n_rows = tower == 0 ? s.B : 2 * s.B;
  float dlt[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) dlt[k] = 0.f;
  int64_t bb[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    int64_t r = tile * GTM + w * 8 + k;
    if (r >= n_rows) r = n_rows - 1;
    bb[k] = (tower == 0 || r < s.B) ? r : r - s.B;
  }
  for (int c0 = 0; c0 < D; c0 += 64) {
    const int c = c0 + lane;
    const int cc = c < D ? c : 0;
    float u[8], p[8], n[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      u[k] = U[bb[k] * D + cc]; p[k] = I[bb[k] * D + cc]; n[k] = I[(s.B + bb[k]) * D + cc];
    }
    if (c < D) {
#pragma unroll
      for (int k = 0; k < 8; ++k) dlt[k] += u[k] * (p[k] - n[k]);
    }
  }
  double lsum = 0.0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int64_t r = tile * GTM + w * 8 + k;
    const float delta = wave_sum(dlt[k]);
    if (r >= n_rows) continue;               // wave-uniform
    const int64_t b = bb[k];
    // d/d delta of softplus(-delta) = -sigma(-delta)
    const float wgt = -1.f / (1.f + __expf(delta)) * invB;
    if (tower == 0) {
      float* go = const_cast<float*>(s.bu.a.gout);
      for (int c = lane; c < D; c += 64) go[b * D + c] = wgt * (I[b * D + c] - I[(s.B + b) * D + c]);
      if (lane == 0) lsum += (double)(fmaxf(-delta, 0.f) + log1pf(__expf(-fabsf(delta))));
    } else {
      const float sg = r < s.B ? wgt : -wgt;
      float* go = const_cast<float*>(s.bi.a.gout);
      for (int c = lane; c < D; c += 64) go[r * D + c] = sg * U[b * D + c];
    }
  }
  return lsum;
}

__global__ __launch_bounds__(STEP_WG) void bpr_step_persistent_kernel(StepArgs s) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ double red_d[4];
  __shared__ float bc_s[4];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const unsigned nwg = gridDim.x;
  const int wg = blockIdx.x;
  const int D = s.fu.D;
  // the step clock is read once, before anything of this step can advance it
  const int64_t t_step = *s.step_dev;
  const float lr = *s.lr_dev;
  const uint64_t seed_u = s.fu.a.seed_step ? rihip_splitmix64(s.fu.a.seed_mul + (uint64_t)t_step) : s.fu.a.seed_mul;
  const uint64_t seed_i = s.fi.a.seed_step ? rihip_splitmix64(s.fi.a.seed_mul + (uint64_t)t_step) : s.fi.a.seed_mul;
  const int64_t ntu = (s.B + GTM - 1) / GTM, nti = (2 * s.B + GTM - 1) / GTM;
  auto stamp = [&](int k) { if (wg == 0 && tid == 0) s.stamps[k] = (double)wall_clock64(); };
  stamp(0);

  // ---- A: towers forward
  for (int64_t it = wg; it < ntu + nti; it += nwg) {
    if (it < ntu) gen_fwd_tile(s.fu, smem, it, seed_u);
    else gen_fwd_tile(s.fi, smem, it - ntu, seed_i);
  }
#ifdef RIHIP_STEP_PROBE   // (tools: the same tiles a second time -- warm instruction / data caches -- timed as phase A)
  stamp(0);
  for (int64_t it = wg; it < ntu + nti; it += nwg) {
    if (it < ntu) gen_fwd_tile(s.fu, smem, it, seed_u);
    else gen_fwd_tile(s.fi, smem, it - ntu, seed_i);
  }
#endif
  stamp(1);
  if (!grid_barrier(s.bar + 0, nwg, s.err_flag)) return;
  stamp(2);
  if (wg == 0 && tid == 0) s.bar[2] = 0;     // the previous launch's last barrier: that grid has drained

  // ---- B: loss gradient + data gradients per tower tile
  for (int64_t it = wg; it < ntu + nti; it += nwg) {
    const int tower = it < ntu ? 0 : 1;
    const int64_t tile = tower == 0 ? it : it - ntu;
    const double ls = loss_rows(s, tower, tile);
    if (tower == 0) {
      if (lane == 0) red_d[w] = ls;
      __syncthreads();
      if (tid == 0) s.loss_part[tile] = ((red_d[0] + red_d[1]) + red_d[2]) + red_d[3];
    }
    __syncthreads();      // the gout rows written above are read back by other waves of this workgroup (same CU, same L1)
    gen_bwd_data_tile(tower == 0 ? s.bu : s.bi, smem, tile);
  }
  stamp(3);
  if (!grid_barrier(s.bar + 1, nwg, s.err_flag)) return;
  stamp(4);
  if (wg == 0 && tid == 0) s.bar[0] = 0;

  // ---- C: weight gradients + dense embedding scatter; squared norm of everything this workgroup produced
  double sq = 0.0;
  {
    const int nwu = gen_wgrad_tiles(D, s.bu.H, s.bu.K1), nwi = gen_wgrad_tiles(D, s.bi.H, s.bi.K1);
    float* red = smem;    // [4][1024] partial tiles
    for (int it = wg; it < nwu + nwi; it += (int)nwg) {
      const bool user = it < nwu;
      const GenBwd& g = user ? s.bu : s.bi;
      const GenWTile t = gen_wgrad_tile_of(g, user ? it : it - nwu);
      const int64_t rows = user ? s.B : 2 * s.B;
      const f32x16 acc = gen_wgrad_acc(g, t, 32 * w, 128, rows, lane);
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 16; ++r) red[w * 1024 + r * 64 + lane] = acc[r];
      __syncthreads();
      if (w == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = ((red[r * 64 + lane] + red[1024 + r * 64 + lane]) + red[2048 + r * 64 + lane]) + red[3072 + r * 64 + lane];
          float* dst = user ? gen_wgrad_dst(g, t, r, lane, s.gW1u, s.gb1u, s.gW2u, s.gb2u)
                            : gen_wgrad_dst(g, t, r, lane, s.gW1i, s.gb1i, s.gW2i, s.gb2i);
          if (dst) { *dst = v; sq += (double)v * (double)v; }
        }
      }
    }
    // scatter: this workgroup owns the table rows r with r % nwg == wg (row 0 = padding_idx and ids outside the table are
    // skipped: nn.Embedding backward).  Pass 1 lists its batch positions in order; pass 2: the first position of every
    // row adds up the row's samples in batch order.
    int* list = reinterpret_cast<int*>(smem);            // [SCAT_LIST] positions
    int* cnt = list + SCAT_LIST;                         // [1]
    for (int tb = 0; tb < 2; ++tb) {
      const int64_t* ids = tb == 0 ? s.bu.a.ids : s.bi.a.ids;
      const float* dX = tb == 0 ? s.bu.a.dX : s.bi.a.dX;
      float* gt = tb == 0 ? s.utab_g : s.itab_g;
      const int64_t nrows = tb == 0 ? s.n_urows : s.n_irows;
      const int64_t n = tb == 0 ? s.B : 2 * s.B;
      __syncthreads();
      if (tid == 0) *cnt = 0;
      __syncthreads();
      for (int64_t base = 0; base < n; base += STEP_WG) {
        const int64_t pos = base + tid;
        bool mine = false;
        if (pos < n) {
          const int64_t id = ids[pos];
          mine = id >= 1 && id < nrows && (int)(id % nwg) == wg;
        }
        // stable append: waves in order, lanes in order
        const unsigned long long bal = __ballot(mine);
        __shared__ int wcnt[4];
        if (lane == 0) wcnt[w] = __popcll(bal);
        __syncthreads();
        int off = *cnt;
        for (int k = 0; k < w; ++k) off += wcnt[k];
        if (mine) {
          const int slot = off + __popcll(bal & ((1ull << lane) - 1ull));
          if (slot < SCAT_LIST) list[slot] = (int)pos;
        }
        __syncthreads();
        if (tid == 0) *cnt = *cnt + wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        __syncthreads();
      }
      int nl = *cnt;
      if (nl > SCAT_LIST) { if (tid == 0 && s.err_flag) atomicOr(s.err_flag, 16); nl = SCAT_LIST; }
      for (int e = w; e < nl; e += 4) {      // one wave per list entry
        const int64_t id = ids[list[e]];
        bool leader = true;
        for (int k = lane; k < e; k += 64) leader = leader && (ids[list[k]] != id);
        if (__ballot(!leader) != 0ull) continue;
        for (int c = lane; c < D; c += 64) {
          float acc = 0.f;
          for (int k = e; k < nl; ++k)
            if (ids[list[k]] == id) acc += dX[(int64_t)list[k] * D + c];
          gt[id * D + c] = acc;
          sq += (double)acc * (double)acc;
        }
      }
    }
  }
  {
    const double ws = wave_sum_d(sq);
    __syncthreads();
    if (lane == 0) red_d[w] = ws;
    __syncthreads();
    if (tid == 0) s.sq_part[wg] = ((red_d[0] + red_d[1]) + red_d[2]) + red_d[3];
  }
  stamp(5);
  if (!grid_barrier(s.bar + 2, nwg, s.err_flag)) return;
  stamp(6);
  if (wg == 0 && tid == 0) s.bar[1] = 0;

  // ---- D: clip coefficient (same arithmetic in every workgroup) + dense Adam over MLPs and both tables
  {
    double tot = 0.0;
    for (unsigned i = tid; i < nwg; i += STEP_WG) tot += s.sq_part[i];
    tot = wave_sum_d(tot);
    if (lane == 0) red_d[w] = tot;
    __syncthreads();
    if (tid == 0) {
      const double st = ((red_d[0] + red_d[1]) + red_d[2]) + red_d[3];
      const float tn = (float)sqrt(st);
      const float c = s.max_norm / (tn + 1e-6f);
      const double bc1 = 1.0 - pow((double)s.h.b1, (double)t_step);
      const double bc2 = 1.0 - pow((double)s.h.b2, (double)t_step);
      bc_s[0] = c < 1.f ? c : 1.f;
      bc_s[1] = (float)((double)lr / bc1);
      bc_s[2] = (float)sqrt(bc2);
      bc_s[3] = tn;
    }
    __syncthreads();
    const float coef = bc_s[0], lr1 = bc_s[1], sb2 = bc_s[2];
    // float4 walk over [MLP flat | user table | item table] (every segment is a multiple of 4 floats, 16-B aligned)
    const int64_t seg_n[3] = {s.n_flat / 4, s.n_urows * D / 4, s.n_irows * D / 4};
    float* seg_p[3] = {s.flat_p, s.utab, s.itab};
    float* seg_g[3] = {s.flat_g, s.utab_g, s.itab_g};
    float* seg_m[3] = {s.flat_m, s.utab_m, s.itab_m};
    float* seg_v[3] = {s.flat_v, s.utab_v, s.itab_v};
#pragma unroll
    for (int sg = 0; sg < 3; ++sg) {
      f32x4* p4 = reinterpret_cast<f32x4*>(seg_p[sg]);
      f32x4* g4 = reinterpret_cast<f32x4*>(seg_g[sg]);
      f32x4* m4 = reinterpret_cast<f32x4*>(seg_m[sg]);
      f32x4* v4 = reinterpret_cast<f32x4*>(seg_v[sg]);
      for (int64_t i = (int64_t)wg * STEP_WG + tid; i < seg_n[sg]; i += (int64_t)nwg * STEP_WG) {
        f32x4 pp = p4[i], gg = g4[i], mm = m4[i], vv = v4[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float pk = pp[k], mk = mm[k], vk = vv[k];
          adam_elem_p(pk, gg[k], mk, vk, coef, lr1, sb2, s.h);
          pp[k] = pk; mm[k] = mk; vv[k] = vk;
        }
        p4[i] = pp; m4[i] = mm; v4[i] = vv;
        if (sg > 0) g4[i] = f32x4{0.f, 0.f, 0.f, 0.f};   // the dense table gradient is consumed: the next step scatters into zeros
      }
    }
    if (wg == 0) {
      double l = 0.0;
      for (int64_t i = tid; i < ntu; i += STEP_WG) l += s.loss_part[i];
      l = wave_sum_d(l);
      __syncthreads();
      if (lane == 0) red_d[w] = l;
      __syncthreads();
      if (tid == 0) {
        *s.loss = (float)((((red_d[0] + red_d[1]) + red_d[2]) + red_d[3]) / (double)s.B);
        *s.coef = coef; *s.gnorm = bc_s[3];
        s.hyper_dev[0] = lr1; s.hyper_dev[1] = sb2;
        *s.step_dev = t_step + 1;
      }
    }
  }
  stamp(7);
#ifdef RIHIP_STEP_PROBE
  if (wg == 0 && tid == 0) {
    g_rihip_probe[6] = wall_clock64();
    for (int k = 0; k < 6; ++k) s.stamps[k] = (double)g_rihip_probe[k];     // sub-phases of the (warm) forward tile
    s.stamps[6] = s.stamps[7] = (double)g_rihip_probe[5];
  }
#endif
}

}  // namespace

extern "C" int rihip_bpr_step_persistent_supported(int64_t B, int d, int hidden) {
  return (B >= 1 && 2 * B <= SCAT_LIST && d >= 16 && d <= 256 && hidden >= 16 && hidden <= 256 && d % 16 == 0 &&
          hidden % 16 == 0) ? 1 : 0;
}

extern "C" int64_t rihip_bpr_step_scratch_doubles(int64_t B) { return (B + 31) / 32 + RIHIP_NCU + 8; }

extern "C" int rihip_bpr_step_persistent(const rihip_step_args* a, void* stream) {
  RIHIP_REQUIRE(a, RIHIP_ERR_ARG, "bpr_step_persistent: null arguments");
  const rihip_tower_io &u = a->user, &it = a->item;
  RIHIP_REQUIRE(rihip_bpr_step_persistent_supported(u.B, a->d, a->hidden), RIHIP_ERR_SHAPE,
                "bpr_step_persistent: unsupported (B=%lld, embed_dim=%d, hidden_dim=%d): B <= %d, dims multiples of 16 up to 256",
                (long long)u.B, a->d, a->hidden, SCAT_LIST / 2);
  RIHIP_REQUIRE(it.B == 2 * u.B && u.genres == nullptr && it.genres != nullptr, RIHIP_ERR_ARG,
                "bpr_step_persistent: item.B must be 2*user.B (pos || neg), genres on the item tower only");
  RIHIP_REQUIRE(u.table && u.ids && it.table && it.ids && u.out && u.hid && u.denom && it.out && it.hid && it.denom &&
                    u.grad_out && it.grad_out && u.dX && it.dX && u.bwd_workspace && it.bwd_workspace, RIHIP_ERR_ARG,
                "bpr_step_persistent: null tower buffer");
  RIHIP_REQUIRE(a->dW1_u && a->db1_u && a->dW2_u && a->db2_u && a->dW1_i && a->db1_i && a->dW2_i && a->db2_i && a->flat_p &&
                    a->flat_g && a->flat_m && a->flat_v && a->utab_g && a->utab_m && a->utab_v && a->itab_g && a->itab_m &&
                    a->itab_v && a->lr_dev && a->step_dev && a->hyper_dev && a->coef && a->gnorm && a->loss &&
                    a->scratch_doubles && a->barrier, RIHIP_ERR_ARG, "bpr_step_persistent: null optimiser buffer");
  RIHIP_REQUIRE(a->n_scratch_doubles >= rihip_bpr_step_scratch_doubles(u.B), RIHIP_ERR_ARG,
                "bpr_step_persistent: scratch too small");
  RIHIP_REQUIRE(a->dropout_p >= 0.f && a->dropout_p < 1.f, RIHIP_ERR_ARG, "bpr_step_persistent: dropout_p=%f", a->dropout_p);
  RIHIP_REQUIRE(a->n_flat > 0 && a->n_flat % 4 == 0 &&
                    ((reinterpret_cast<uintptr_t>(a->flat_p) | reinterpret_cast<uintptr_t>(a->flat_g) |
                      reinterpret_cast<uintptr_t>(a->flat_m) | reinterpret_cast<uintptr_t>(a->flat_v) |
                      reinterpret_cast<uintptr_t>(u.table) | reinterpret_cast<uintptr_t>(it.table) |
                      reinterpret_cast<uintptr_t>(a->utab_g) | reinterpret_cast<uintptr_t>(a->itab_g) |
                      reinterpret_cast<uintptr_t>(u.out) | reinterpret_cast<uintptr_t>(it.out) |
                      reinterpret_cast<uintptr_t>(u.grad_out) | reinterpret_cast<uintptr_t>(it.grad_out) |
                      reinterpret_cast<uintptr_t>(u.bwd_workspace) | reinterpret_cast<uintptr_t>(it.bwd_workspace)) & 15) == 0,
                RIHIP_ERR_ARG, "bpr_step_persistent: buffers must be 16-byte aligned and n_flat a multiple of 4");
  const int d = a->d, H = a->hidden;
  StepArgs s;
  const rihip_tower_io* io[2] = {&u, &it};
  GenFwd* f[2] = {&s.fu, &s.fi};
  GenBwd* b[2] = {&s.bu, &s.bi};
  const float scale = 1.f / (1.f - a->dropout_p);
  const bool training = a->training && a->dropout_p > 0.f;
  for (int t = 0; t < 2; ++t) {
    const rihip_tower_io& o = *io[t];
    TowerFwdArgs& fa = f[t]->a;
    fa.W1p = nullptr; fa.W2p = nullptr; fa.table = o.table; fa.n_rows = o.n_rows; fa.ids = o.ids; fa.genres = o.genres; fa.B = o.B;
    fa.W1 = o.W1; fa.b1 = o.b1; fa.W2 = o.W2; fa.b2 = o.b2; fa.out = o.out; fa.hid = o.hid; fa.denom = o.denom;
    fa.training = training ? 1 : 0; fa.seed_mul = rihip_seed_mul(o.seed); fa.thresh24 = rihip_thresh24(a->dropout_p);
    fa.scale = scale; fa.row0 = o.row0; fa.seed_step = a->step_dev; fa.err_flag = a->err_flag;
    f[t]->D = d; f[t]->H = H; f[t]->K1 = d + (t ? 18 : 0);
    TowerBwdArgs& ba = b[t]->a;
    ba.table = o.table; ba.n_rows = o.n_rows; ba.ids = o.ids; ba.genres = o.genres; ba.B = o.B; ba.W1 = o.W1; ba.W2 = o.W2;
    ba.gout = o.grad_out; ba.out = o.out; ba.denom = o.denom; ba.hid = o.hid; ba.scale = training ? scale : 1.f; ba.dX = o.dX;
    ba.slab = nullptr;
    b[t]->gy = o.bwd_workspace; b[t]->dpre = o.bwd_workspace + (size_t)o.B * d;
    b[t]->D = d; b[t]->H = H; b[t]->K1 = d + (t ? 18 : 0); b[t]->rows_per_slab = 0; b[t]->nslab = 1;
    b[t]->P = H * b[t]->K1 + H + d * H + d;
  }
  s.B = u.B;
  s.gW1u = a->dW1_u; s.gb1u = a->db1_u; s.gW2u = a->dW2_u; s.gb2u = a->db2_u;
  s.gW1i = a->dW1_i; s.gb1i = a->db1_i; s.gW2i = a->dW2_i; s.gb2i = a->db2_i;
  s.flat_p = a->flat_p; s.flat_g = a->flat_g; s.flat_m = a->flat_m; s.flat_v = a->flat_v; s.n_flat = a->n_flat;
  s.utab = const_cast<float*>(u.table); s.utab_g = a->utab_g; s.utab_m = a->utab_m; s.utab_v = a->utab_v; s.n_urows = u.n_rows;
  s.itab = const_cast<float*>(it.table); s.itab_g = a->itab_g; s.itab_m = a->itab_m; s.itab_v = a->itab_v; s.n_irows = it.n_rows;
  s.h = AdamH{a->beta1, a->beta2, a->eps, a->weight_decay}; s.max_norm = a->max_norm;
  s.lr_dev = a->lr_dev; s.step_dev = a->step_dev; s.hyper_dev = a->hyper_dev; s.coef = a->coef; s.gnorm = a->gnorm;
  s.loss = a->loss; s.err_flag = a->err_flag;
  s.loss_part = a->scratch_doubles; s.sq_part = a->scratch_doubles + (u.B + 31) / 32;
  s.stamps = s.sq_part + RIHIP_NCU;
  s.bar = a->barrier;
  size_t lf = gen_fwd_lds_floats(d, H, d + 18);
  const size_t lb = gen_bwd_lds_floats(d, H);
  if (lb > lf) lf = lb;
  if (lf < 4096 + 64) lf = 4096 + 64;    // phase C: 4 partial tiles / the scatter list
  s.lds_floats = lf;
  static bool granted = false;
  if (!granted) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(bpr_step_persistent_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(float) * gen_fwd_lds_floats(256, 256, 274)));
    granted = true;
  }
  // one workgroup per CU: co-resident by construction on an otherwise idle device; the barrier gives up (error bit 3)
  // rather than hang if it is not
  hipLaunchKernelGGL(bpr_step_persistent_kernel, dim3(RIHIP_NCU), dim3(STEP_WG), sizeof(float) * lf, (hipStream_t)stream, s);
  RIHIP_CHECK_LAUNCH();
  return RIHIP_OK;
}
