/* recommendit_hip.h -- C ABI of librecommendit_hip.so (gfx950 / MI355X).
 *
 * The reference (sarihammad/recommendit) has no FFI of its own: its seam is three Python
 * classes (src/models/__init__.py:1-3).  Each entry point below names the reference code
 * whose *body* it replaces (paths relative to the reference root); the Python classes in
 * recommendit_amd/ keep the reference's class surface and call these through ctypes.
 * INTEGRATION.md shows the binding a reference maintainer would add.
 *
 * Conventions
 *   - every function returns 0 (RIHIP_OK) or a non-zero status; rihip_last_error() gives the
 *     message (thread-local).  Nothing aborts the process: the callers' fallbacks
 *     (src/serving/recommender.py:202-207, src/serving/app.py:182-185) must keep working.
 *   - all array pointers are DEVICE pointers unless a comment says host; the caller allocates
 *     inputs, outputs and workspaces; the library never frees caller memory and never keeps a
 *     caller pointer after returning.  Opaque handles (ip_index, gbdt) own their device memory
 *     and are destroyed by their paired *_destroy.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls are
 *     asynchronous on that stream except where documented.
 *   - float = IEEE binary32, ids/rows = int64_t.
 */
#ifndef RECOMMENDIT_HIP_H
#define RECOMMENDIT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RIHIP_ABI_VERSION 1

int rihip_abi_version(void);
/* "gfx950": the only ISA this library carries code objects for */
const char* rihip_target_arch(void);
/* last error message of the calling thread ("" if none) */
const char* rihip_last_error(void);
/* Generation of the library state a captured hipGraph can have baked in: handle-owned scratch buffers of the index and
 * the forest (grown on demand: the old allocation is freed), nprobe, the id map, the index / forest content.  Bumped on
 * every such change, library-wide.  A caller that replays a captured chain (GpuRecommendationPipeline, the batched form
 * of recommender.py:269-387) records the value after capture and re-captures when it differs. */
uint64_t rihip_scratch_generation(void);
/* name of device 0's ISA as reported by the runtime (host buffer); needs a GPU */
int rihip_device_arch(char* buf, int buf_len);

/* ---- Two-Tower towers ----------------------------------------------------------------------
 * rihip_tower_forward replaces UserTower.forward (src/models/two_tower.py:39-42; genres==NULL)
 * and ItemTower.forward (:68-72; genres = [B,18] multi-hot):
 *   x = table[ids] (|| genres) ; h = dropout(relu(x W1^T + b1)) ; y = h W2^T + b2 ;
 *   out = y / max(|y|_2, 1e-12).
 * table [n_rows,d]; W1 [hidden, d(+18)]; b1 [hidden]; W2 [d,hidden]; b2 [d]  (nn.Linear layout).
 * training!=0 && dropout_p>0: keep-mask from the counter-based generator keyed by
 * (seed, (row0+row)*hidden+col) -- see oracle/two_tower_np.py:dropout_keep_mask.
 * hid [B,hidden] (post-dropout activations) and denom [B] are saved for backward (nullable).
 * err_flag (device int, nullable) is set to 1 if an id is outside [0,n_rows) (row 0 is used).
 * workspace (nullable): rihip_tower_forward_workspace_floats(d, hidden, item) floats, 16-B aligned; holds the
 * MFMA-fragment-major copy of W1/W2 rebuilt each call (coalesced weight loads).
 * seed_step_dev (nullable): device int64 mixed into the dropout seed, so a captured hipGraph draws a fresh
 * mask on every replay (the counter is advanced by rihip_adam_hyper_step).
 * Shapes: the reference takes any (embed_dim, hidden_dim) (two_tower.py:80-95).  rihip_tower_shape_ok(d, hidden) = 1 for
 * every pair of multiples of 16 up to 256; rihip_tower_supported(d, hidden) = 1 for the pairs with tuned template
 * instantiations (the fast path) -- every other pair runs the runtime-shape kernels of csrc/tower_generic.hip. */
int rihip_tower_supported(int d, int hidden);
int rihip_tower_shape_ok(int d, int hidden);
int64_t rihip_tower_forward_workspace_floats(int d, int hidden, int item);
int rihip_tower_forward(const float* table, int64_t n_rows, const int64_t* ids, const float* genres, int64_t B,
                        int d, int hidden, const float* W1, const float* b1, const float* W2, const float* b2,
                        int training, float dropout_p, uint64_t seed, int64_t row0, float* out, float* hid,
                        float* denom, int* err_flag, float* workspace, const int64_t* seed_step_dev, void* stream);

/* Backward of the above (autograd of two_tower.py:39-42/:68-72, run by
 * src/training/train_embeddings.py:190).  grad_out = dL/d out [B,d].
 * Outputs: dX [B,d] per-sample embedding-row gradients (scatter them with
 * rihip_embedding_scatter_add or the row-sparse path); dW1/db1/dW2/db2 in nn.Linear layout,
 * overwritten (accumulate=0) or added to (accumulate=1).  dropout_scale = 1/(1-p) if the
 * forward ran in training mode with p>0, else 1.  workspace: floats, size from
 * rihip_tower_backward_workspace_floats. */
int64_t rihip_tower_backward_workspace_floats(int64_t B, int d, int hidden, int item);
int rihip_tower_backward(const float* table, int64_t n_rows, const int64_t* ids, const float* genres, int64_t B,
                         int d, int hidden, const float* W1, const float* W2, const float* grad_out,
                         const float* out, const float* denom, const float* hid, float dropout_scale, float* dX,
                         float* dW1, float* db1, float* dW2, float* db2, int accumulate, float* workspace,
                         void* stream);
/* Same, plus dx_event (a hipEvent_t, nullable): recorded on `stream` as soon as dX is complete -- before the
 * weight-gradient kernels -- so a caller can start the embedding-row gradient reduction on another stream beside them. */
int rihip_tower_backward_ev(const float* table, int64_t n_rows, const int64_t* ids, const float* genres, int64_t B,
                            int d, int hidden, const float* W1, const float* W2, const float* grad_out,
                            const float* out, const float* denom, const float* hid, float dropout_scale, float* dX,
                            float* dW1, float* db1, float* dW2, float* db2, int accumulate, float* workspace,
                            void* stream, void* dx_event);
/* The backward in two halves, for a step with two towers: _partial launches the gradient kernels only (dX complete,
 * weight-gradient slabs left in `workspace`, their count in *n_slabs); _reduce2 then sums the slabs of BOTH towers
 * (b may be absent: ws_b = NULL) in one pair of launches -- the same sums in the same order as rihip_tower_backward,
 * so the results are bit-identical.  The two towers need separate workspaces. */
int rihip_tower_backward_partial(const float* table, int64_t n_rows, const int64_t* ids, const float* genres, int64_t B,
                                 int d, int hidden, const float* W1, const float* W2, const float* grad_out,
                                 const float* out, const float* denom, const float* hid, float dropout_scale,
                                 float* dX, float* workspace, void* stream, void* dx_event, int* n_slabs);
int rihip_tower_backward_reduce2(int d, int hidden, float* ws_a, int64_t B_a, int item_a, int n_slabs_a, float* dW1_a,
                                 float* db1_a, float* dW2_a, float* db2_a, float* ws_b, int64_t B_b, int item_b,
                                 int n_slabs_b, float* dW1_b, float* db1_b, float* dW2_b, float* db2_b, int accumulate,
                                 void* stream);

/* Both towers of one training step in one launch each way (a step of batch 256 is bounded by its ~10 dependent
 * launches).  rihip_tower_io carries what differs per tower in rihip_tower_forward / rihip_tower_backward_partial;
 * user->genres must be NULL and item->genres non-NULL.  Batches that take the chip-filling kernels (>= 49 152 rows)
 * or arguments the pair kernels do not cover fall back to the two single calls -- the results are bit-identical either
 * way.  The two towers need separate workspaces. */
typedef struct rihip_tower_io {
  const float* table; int64_t n_rows; const int64_t* ids; const float* genres; int64_t B;
  const float *W1, *b1, *W2, *b2;
  uint64_t seed; int64_t row0;
  float *out, *hid, *denom;          /* forward outputs = backward inputs */
  float* fwd_workspace;              /* rihip_tower_forward_workspace_floats, nullable */
  const float* grad_out; float* dX;  /* backward */
  float* bwd_workspace;              /* rihip_tower_backward_workspace_floats */
} rihip_tower_io;
int rihip_tower_forward_pair(const rihip_tower_io* user, const rihip_tower_io* item, int d, int hidden, int training,
                             float dropout_p, int* err_flag, const int64_t* seed_step_dev, void* stream);
int rihip_tower_backward_partial_pair(const rihip_tower_io* user, const rihip_tower_io* item, int d, int hidden,
                                      float dropout_scale, void* stream, void* dx_event_user, void* dx_event_item,
                                      int* n_slabs_user, int* n_slabs_item);

/* The whole sampled-negative training step with the reference's optimiser (towers -> bpr_loss -> backward ->
 * clip_grad_norm_ -> dense Adam + coupled L2 on the MLPs and BOTH tables: src/training/train_embeddings.py:178-195,
 * :160-161) in ONE launch: a persistent grid walks the step behind three grid barriers (csrc/step_persistent.hip).  For
 * the batch sizes the reference trains at (BATCH_SIZE = 1024, src/config.py:25; B <= 2048 here) a step is otherwise
 * seven dependent launches.  Any (embed_dim, hidden_dim) of rihip_tower_shape_ok.  item.B = 2*user.B (pos || neg).
 * grad_out / dX / out / hid / denom of both rihip_tower_io are scratch the step fills; bwd_workspace needs B*(d+hidden)
 * floats.  dW1_u ... db2_i: the MLP gradient tensors (views of flat_g).  *_tab_g: dense table gradients, zero on entry and
 * left zero.  step_dev / lr_dev / hyper_dev / coef / gnorm / loss as in rihip_clip_coef_step (the clock is advanced).
 * barrier: 4 zero-initialised words owned by the caller for the life of the trainer.  Error bit 8 of *err_flag: the grid
 * was not co-resident (another kernel held CUs) and the step was abandoned; bit 1: an id outside its table. */
typedef struct rihip_step_args {
  rihip_tower_io user, item;
  float *dW1_u, *db1_u, *dW2_u, *db2_u, *dW1_i, *db1_i, *dW2_i, *db2_i;
  float *flat_p, *flat_g, *flat_m, *flat_v; int64_t n_flat;
  float *utab_g, *utab_m, *utab_v, *itab_g, *itab_m, *itab_v;
  int d, hidden, training; float dropout_p;
  float beta1, beta2, eps, weight_decay, max_norm;
  const float* lr_dev; int64_t* step_dev; float* hyper_dev; float* coef; float* gnorm; float* loss; int* err_flag;
  double* scratch_doubles; int64_t n_scratch_doubles;   /* >= rihip_bpr_step_scratch_doubles(B) */
  unsigned* barrier;
} rihip_step_args;
int rihip_bpr_step_persistent_supported(int64_t B, int d, int hidden);
int64_t rihip_bpr_step_scratch_doubles(int64_t B);
int rihip_bpr_step_persistent(const rihip_step_args* a, void* stream);

/* nn.Embedding backward (dense): grad_table[ids[b]] += dX[b]; row 0 (padding_idx, two_tower.py:27,54) and ids outside
 * [1, n_rows) are skipped.  Bitwise reproducible: every row receives its samples one after the other in batch order,
 * starting from its current contents (the float32 chain of index_add_ on a CPU) -- no floating-point atomics.
 * _add2 handles two tables (user + item) in one launch. */
int rihip_embedding_scatter_add(float* grad_table, int64_t n_rows, const int64_t* ids, const float* dX, int64_t B,
                                int d, void* stream);
int rihip_embedding_scatter_add2(float* grad_a, int64_t n_rows_a, const int64_t* ids_a, const float* dX_a, int64_t B_a,
                                 float* grad_b, int64_t n_rows_b, const int64_t* ids_b, const float* dX_b,
                                 int64_t B_b, int d, void* stream);
/* rihip_tower_backward_reduce2 and rihip_embedding_scatter_add2 together: both wait only for the tower backward, so
 * the scatter launch carries one slab-reduction level in extra workgroups (a dense small-batch step saves a dependent
 * launch).  Results are bit-identical to the two separate calls.  Bs_* = samples scattered into each table. */
int rihip_backward_reduce2_scatter2(int d, int hidden, float* ws_a, int64_t B_a, int item_a, int n_slabs_a, float* dW1_a,
                                    float* db1_a, float* dW2_a, float* db2_a, float* ws_b, int64_t B_b, int item_b,
                                    int n_slabs_b, float* dW1_b, float* db1_b, float* dW2_b, float* db2_b, int accumulate,
                                    float* grad_a, int64_t n_rows_a, const int64_t* ids_a, const float* dX_a, int64_t Bs_a,
                                    float* grad_b, int64_t n_rows_b, const int64_t* ids_b, const float* dX_b, int64_t Bs_b,
                                    void* stream);

/* ---- losses --------------------------------------------------------------------------------
 * rihip_bpr_pair_loss replaces TwoTowerModel.bpr_loss (two_tower.py:117-130) and its backward:
 * loss = mean softplus(-(u.p - u.n)); dU,dP,dN = d loss / d inputs.  workspace: >=1024 doubles.
 * loss may be NULL: the value is then (1/B) * sum(workspace[0 .. rihip_bpr_pair_nparts(B))), summed later by the
 * caller (rihip_sum_partials / rihip_clip_coef_step). */
int rihip_bpr_pair_loss(const float* U, const float* P, const float* N, int64_t B, int d, float* loss, float* dU,
                        float* dP, float* dN, double* workspace, void* stream);
int64_t rihip_bpr_pair_nparts(int64_t B);

/* In-batch-negative BPR (TwoTowerModel.in_batch_bpr_loss, two_tower.py:132-160, closed form
 *   L = 1/(B(B-1)) sum_i sum_{j!=i} softplus(s_ij - s_ii)) is three calls:
 *   rihip_rowdot        pos[i] = U[i].I[i+i_offset]
 *   rihip_inbatch_sweep mode_user=1: owners=users, swept=items -> d_owner=dU, r_out, loss_part
 *   rihip_inbatch_sweep mode_user=0: owners=items, swept=users (+pos, r_in=r) -> d_owner=dI
 *   rihip_sum_partials  loss = scale * sum(loss_part[0 .. rihip_inbatch_loss_parts))  with scale = 1/(B(B-1))
 * workspace: floats, rihip_inbatch_workspace_floats(n_owner, n_swept, d) (slabs of the swept-range splits that
 * keep small batches chip-filling; combined in fixed order => bitwise reproducible).
 * precision: 0 = exact-f32 MFMA (v_mfma_f32_32x32x2_f32, an fmaf chain); 2 = "bf16x6": every fp32 operand split
 * EXACTLY into three bf16 pieces (8+8+8 bits), six of the nine partial products on bf16 MFMA with f32 accumulation
 * (dropped terms <= 2^-23 |a||b|: fp32-level accuracy, same test tolerances as precision 0, ~2.5x faster);
 * 1 = "bf16x3": two pieces, products hi.hi+hi.lo+lo.hi (relative product error ~2^-16).  The stored-G passes take
 * precision 0 or 2.
 * Global indices (owner_goff / swept_goff) place a rank's local rows inside the all-gathered
 * batch for multi-GPU in-batch negatives; n_global = B.  d in {32,64,128}. */
int rihip_rowdot(const float* U, const float* I, int64_t B, int64_t i_offset, int d, float* pos, void* stream);
int64_t rihip_inbatch_workspace_doubles(int64_t n_owner);
int64_t rihip_inbatch_loss_parts(int64_t n_owner, int64_t n_swept);
int64_t rihip_inbatch_workspace_floats(int64_t n_owner, int64_t n_swept, int d);
int rihip_inbatch_sweep(int mode_user, const float* owners, int64_t n_owner, int64_t owner_goff, const float* swept,
                        int64_t n_swept, int64_t swept_goff, int d, const float* pos, const float* r_in,
                        int64_t n_global, float* d_owner, float* r_out, double* loss_part, float* workspace,
                        int precision, void* stream);
int rihip_sum_partials(const double* part, int64_t n, double scale, float* out, void* stream);

/* Stored-G form of the same loss (two_tower.py:132-160), the default when memory allows: the user pass is the
 * mode_user=1 sweep that ALSO writes the weights sigma(s_ij - s_ii) (0 on the diagonal; G = weight/(B(B-1))) to `gmat`
 * (rihip_inbatch_gmat_floats(n_users, n_items) floats, 32x32-blocked G^T); the item pass is then a plain exact-f32
 * product d_items[j] = sum_i G[i][j] users[i] - r_j users[j] over the LOCAL users for ALL n_items items -- no second
 * score sweep (6 B^2 d FLOP per step instead of 8 B^2 d).  Multi-GPU: every rank calls both with its local users and
 * the all-gathered items, then reduce-scatters d_items.  Workspaces: rihip_inbatch_workspace_floats(n_users, n_items, d)
 * for the user pass and (n_items, n_users, d) for the item pass. */
int64_t rihip_inbatch_gmat_floats(int64_t n_users, int64_t n_items);
int rihip_inbatch_user_pass(const float* users, int64_t n_users, int64_t user_goff, const float* items,
                            int64_t n_items, int64_t item_goff, int d, const float* pos, int64_t n_global,
                            float* d_users, float* r_out, double* loss_part, float* workspace, float* gmat,
                            int precision, void* stream);
int rihip_inbatch_item_pass(const float* gmat, const float* users, int64_t n_users, int64_t user_goff,
                            int64_t n_items, int64_t item_goff, int d, const float* r, int64_t n_global,
                            float* d_items, float* workspace, int precision, void* stream);

/* ---- optimiser -----------------------------------------------------------------------------
 * clip_grad_norm_(max_norm) (train_embeddings.py:191): rihip_sumsq writes rihip_sumsq_nparts()
 * partial sums of x^2 per call; rihip_clip_coef reduces any number of partials to
 * coef = min(1, max_norm/(norm+1e-6)) ON DEVICE (no host sync), consumed by the Adam kernels. */
int rihip_sumsq_nparts(void);
/* Multi-tensor forms of rihip_sumsq / rihip_adam_dense for steps that are bounded by dependent kernel boundaries
 * (batch 256 ... 8192 at ML-1M scale): up to 4 tensors per launch, HOST arrays of device pointers and sizes; the
 * arithmetic and the partial layout (rihip_sumsq_nparts() doubles per tensor, consecutively) equal the single calls.
 * zero_grad_mask bit t: tensor t's gradient is overwritten with zeros after the update (dense table gradients). */
int rihip_sumsq_multi(int n_tensors, const float* const* x, const int64_t* n, double* part, void* stream);
int rihip_adam_dense_multi(int n_tensors, float* const* p, float* const* g, float* const* m, float* const* v,
                           const int64_t* n, int zero_grad_mask, float lr, float beta1, float beta2, float eps,
                           float weight_decay, int64_t step, const float* clip_coef, const float* hyper_dev,
                           void* stream);
int rihip_sumsq(const float* x, int64_t n, double* part, void* stream);
int rihip_clip_coef(const double* part, int64_t n_part, float max_norm, float* coef, float* total_norm, void* stream);
/* torch.optim.Adam(lr, betas, eps, weight_decay) single step with coupled L2
 * (train_embeddings.py:160,192); g is scaled by *clip_coef (nullable) first.  step >= 1. */
int rihip_adam_dense(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                     float eps, float weight_decay, int64_t step, const float* clip_coef, const float* hyper_dev,
                     void* stream);
/* Graph-replay clock: *step_dev += 1, then hyper_dev[0] = *lr_dev / (1 - beta1^t), hyper_dev[1] = sqrt(1 - beta2^t).
 * Passing hyper_dev (non-NULL) to rihip_adam_dense / rihip_adam_rows overrides their host-side lr/step arguments. */
int rihip_adam_hyper_step(int64_t* step_dev, const float* lr_dev, float beta1, float beta2, float* hyper_dev,
                          void* stream);
/* The same clock folded into the launches a step makes anyway (a small-batch step costs ~5 us per dependent launch):
 * rihip_clip_coef_step = rihip_clip_coef, then hyper_dev for t = *step_dev (the step that is running: initialise
 * *step_dev to 1), then *step_dev = t + 1.  loss_part (nullable): *loss = loss_scale * sum(loss_part[0..n)) as well --
 * the loss partials of rihip_bpr_pair_loss(loss = NULL) / the in-batch passes, summed off the critical path. */
int rihip_clip_coef_step(const double* part, int64_t n_part, float max_norm, float* coef, float* total_norm,
                         int64_t* step_dev, const float* lr_dev, float beta1, float beta2, float* hyper_dev,
                         const double* loss_part, int64_t n_loss_part, double loss_scale, float* loss, void* stream);

/* Row-sparse path for tables too large for a dense pass per step (SURVEY.md §7 hard part 1):
 * group (id,sample) pairs by id (radix sort), sum each row's contributions in sorted order
 * (bitwise reproducible), then Adam on touched rows only.  workspace bytes from
 * rihip_rows_workspace_bytes(B, d); uniq int64[B]; Gc float[B,d]; part double[rihip_rows_nparts()].
 * n_rows: number of table rows (ids < n_rows) -- only the significant key bits are sorted; 0 = unknown (all 63). */
int64_t rihip_rows_workspace_bytes(int64_t B, int d);
int rihip_rows_nparts(void);
int rihip_rows_group(const int64_t* ids, int64_t B, int d, int64_t n_rows, int64_t* uniq, void* workspace,
                     int64_t workspace_bytes, void* stream);
int rihip_rows_n_unique_ptr(void* workspace, int64_t B, int d, const int** n_unique_dev);
int rihip_rows_reduce(const float* dX, int64_t B, int d, const int64_t* uniq, void* workspace, float* Gc,
                      double* part, void* stream);
int rihip_adam_rows(float* table, float* m, float* v, const int64_t* uniq, const float* Gc, int64_t B, int d,
                    void* workspace, float lr, float beta1, float beta2, float eps, float weight_decay,
                    int64_t step, const float* clip_coef, const float* hyper_dev, void* stream);

/* ---- row-sharded tables (multi-GPU; BASELINE cfg4) ---------------------------------------------
 * The reference keeps each embedding table on one device (src/models/two_tower.py:27,:54; device picked at
 * src/training/train_embeddings.py:102-109).  Cut by rows over `world` ranks, global row g >= 1 lives on rank
 * (g-1) % world at local row (g-1)/world + 1 (local row 0 = unused padding).  route_rows sorts a batch of global
 * ids by owner (stable): sorted_local int64[B] = owner-local rows in send order, perm int64[B] = pair of each send
 * slot, pos int64[B] = send slot of each pair (the ids the tower kernels use on the received-row staging table),
 * counts int64[world] = requests per owner (the all-to-all split sizes).  gather_rows: out[i] = table[ids[i]]. */
int64_t rihip_route_workspace_bytes(int64_t B);
int rihip_route_rows(const int64_t* ids, int64_t B, int world, int64_t* sorted_local, int64_t* perm, int64_t* pos,
                     int64_t* counts, int* err_flag, void* workspace, int64_t workspace_bytes, void* stream);
/* Fixed-capacity form (no split size ever crosses to the host: equal-split all-to-alls of `cap` slots per peer):
 * slot_ids int64[world*cap] = owner-local rows in slot (owner*cap + position), 0 (the padding row: gradient dropped by
 * the row-sparse optimiser) in unused slots; slot_of_pair int64[B] = slot of each pair = its row in the
 * [world*cap, d] received-rows staging table; counts int64[world] (device).  cap = B can never overflow; a smaller cap
 * that does sets bit 2 of *err_flag.  scatter_rows: out[slot[i]] = src[i] (row gradients into their send slots). */
int rihip_route_rows_fixed(const int64_t* ids, int64_t B, int world, int64_t cap, int64_t* slot_ids,
                           int64_t* slot_of_pair, int64_t* counts, int* err_flag, void* workspace,
                           int64_t workspace_bytes, void* stream);
int rihip_scatter_rows(const float* src, const int64_t* slot, int64_t n, int64_t n_slots, int d, float* out,
                       int* err_flag, void* stream);
int rihip_gather_rows(const float* table, int64_t n_rows, const int64_t* ids, int64_t n, int d, float* out,
                      int* err_flag, void* stream);

/* ---- LambdaMART training -------------------------------------------------------
 * Replaces the body of LightGBMRanker.train (src/models/ranker.py:52-155): lgb.train(objective="lambdarank", ...) with
 * the reference's parameters (defaults below = ranker.py:107-121).  X device f32 [n,F] row-major, y device f32 [n]
 * (integer relevance grades), groups HOST int32 [ng] (documents per query, in row order; <= 16384 each); the
 * validation set is optional (NULL / 0): with it, training stops when an eval metric has not improved for
 * early_stopping_rounds.  model_text receives a malloc'd LightGBM-format text model (free with rihip_free):
 * rihip_gbdt_create_from_text loads it.  history (host, nullable): [n_rounds][2][n_eval_at] NDCG of train / valid
 * (valid = NaN without a validation set).  Synchronous.  Algorithm and its NumPy restatement: csrc/gbdt_train.hip,
 * oracle/lambdamart_np.py (lightgbm itself is not available offline: parity unpinned). */
typedef struct rihip_lambdamart_params {
  int num_leaves, n_estimators, min_child_samples, max_bin, truncation_level, early_stopping_rounds, lambdarank_norm,
      bin_sample;
  int n_eval_at;
  int eval_at[8];
  int n_label_gain;
  double label_gain[32];
  double learning_rate, reg_alpha, reg_lambda, feature_fraction, min_sum_hessian, sigmoid;
  uint64_t seed;
  /* fidelity switches towards LightGBM's defaults (0 = the integer-2^20 / NaN-as-zero / lowest-threshold behaviour):
   * hist_bits 20|40 (40: float-histogram fidelity, sums still order-independent integers); use_missing 1: NaN gets its
   * own bin and every node learns a default direction (decision_type missing = NaN); split_order 1: equal-gain
   * thresholds resolved in FeatureHistogram::FindBestThreshold's scan order */
  int hist_bits, use_missing, split_order, reserved;
} rihip_lambdamart_params;
int rihip_lambdamart_train(const float* X, const float* y, const int32_t* groups, int64_t n, int F, int ng,
                           const float* Xv, const float* yv, const int32_t* groups_v, int64_t nv, int ngv,
                           const rihip_lambdamart_params* params, const char* feature_names, char** model_text,
                           int* best_iteration, int* n_rounds, double* history, void* stream);
void rihip_free(void* p);

/* ---- inner-product index -------------------------------------------------------------------
 * Replaces faiss.IndexFlatIP / IndexIVFFlat(METRIC_INNER_PRODUCT) behind FAISSIndex
 * (src/models/faiss_index.py:68-74 build, :113/:145 search, :164/:196 write/read).
 * Normalisation of vectors/queries stays in the wrapper (faiss_index.py:64-65,:108-110).
 * search: Q device [nq,d]; out_scores f32[nq,k] descending, -inf padded; out_rows i64[nq,k]
 * row numbers in insertion order, -1 padded (faiss convention kept by faiss_index.py:148-152).
 * Exact for a flat index (ties -> lowest row); synchronises the stream once per 4096 queries
 * (exactness check).  k <= rihip_ip_index_max_k() (16384).  1 <= d <= 128: the handle zero-pads rows, queries and
 * centroids to its kernel width (32 / 64 / 128), which changes no inner product; every array that crosses the ABI is
 * [*, d] at the caller's width.  nlist <= 2048.
 * IVF (faiss_index.py:68-74): train_ivf = k-means (Lloyd, IP assignment, mean update, empty lists keep their
 * centroid) from seeded rows, then list-contiguous layout; train_ivf_from = the same from caller-supplied
 * initial centroids (host [nlist,d]; n_iter = 0 partitions by them as they are); set_ivf injects centroids AND
 * the list of every row (host int32 [N]) -- what a FAISS IndexIVFFlat file holds; get_ivf reads both back
 * (host, either may be NULL); reconstruct returns the stored vectors in insertion order (host [N,d]);
 * assign = list of each of n device rows under the index's centroids (device int32 [n]). */
int rihip_ip_index_create(int d, void** handle);
int rihip_ip_index_destroy(void* handle);
int rihip_ip_index_set_vectors(void* handle, const float* X, int64_t N, int x_on_device, void* stream);
int64_t rihip_ip_index_ntotal(void* handle);
int rihip_ip_index_is_ivf(void* handle);
int rihip_ip_index_nlist(void* handle);
int rihip_ip_index_max_k(void);
int rihip_ip_index_train_ivf(void* handle, int nlist, int n_iter, uint64_t seed, void* stream);
int rihip_ip_index_train_ivf_from(void* handle, int nlist, int n_iter, const float* init_centroids, void* stream);
int rihip_ip_index_set_ivf(void* handle, int nlist, const float* centroids, const int32_t* assign, void* stream);
int rihip_ip_index_get_ivf(void* handle, float* centroids, int32_t* assign);   /* synchronous */
int rihip_ip_index_reconstruct(void* handle, float* out);                       /* synchronous */
int rihip_ip_index_assign(void* handle, const float* X, int64_t n, int32_t* assign, void* stream);
int rihip_ip_index_set_nprobe(void* handle, int nprobe);
/* flat indexes with N > 65536: 1 (default) = bf16-MFMA filter with a rigorous error bound + exact f32 re-score of
 * the survivors (results identical to the all-f32 search, proven per query, exact fallback otherwise); 0 = all-f32 */
int rihip_ip_index_set_two_precision(void* handle, int enable);
int rihip_ip_index_search(void* handle, const float* Q, int64_t nq, int k, float* out_scores, int64_t* out_rows,
                          void* stream);
/* Deferred exactness check for serving chains (the reference's recommender.py:269-387 runs retrieval -> features ->
 * ranker per request; here the chain is enqueued without a host round trip in its middle): with enable = 1 a thresholded
 * IVF search of <= 4096 queries returns WITHOUT the host synchronisation that reads how many queries need the exact
 * re-do; enqueue the consumers of the result, then call rihip_ip_index_search_finish (one synchronisation): *n_redone > 0
 * means that many queries were re-done exactly into the same output rows after the consumers ran -- run them again.
 * finish must be called before the next search of the handle; all other search paths are unaffected (*n_redone = 0). */
int rihip_ip_index_set_deferred_check(void* handle, int enable);
int rihip_ip_index_search_finish(void* handle, int* n_redone, void* stream);
/* for hipGraph replays of a captured chain (no host code runs inside a replay): _pending = 1 while a deferred search
 * awaits its finish (ask right after capture); _last_fail_count synchronises `stream` and returns the failure count the
 * last enqueued deferred search wrote -- n > 0: run the chain again eagerly with the check not deferred */
int rihip_ip_index_search_pending(void* handle);
int rihip_ip_index_last_fail_count(void* handle, int* n, void* stream);
int rihip_ip_index_save(void* handle, const char* path);          /* host path; synchronous */
int rihip_ip_index_load(const char* path, void** handle);         /* host path; synchronous */
/* rows[i] = rows[i] >= 0 ? item_ids[rows[i]] : -1   (faiss_index.py:123, :148-152) */
int rihip_map_rows_to_ids(int64_t* rows, int64_t n, const int64_t* item_ids, void* stream);
/* The same mapping inside rihip_ip_index_search (no second pass over the result): with a non-NULL device array of
 * >= ntotal item ids, out_rows receives item_ids[row] (-1 padding unchanged).  The array is not copied and must stay
 * valid while searches run; NULL restores row numbers. */
int rihip_ip_index_set_id_map(void* handle, const int64_t* item_ids_dev);

/* ---- LambdaMART forward --------------------------------------------------------------------
 * Replaces lgb.Booster(model_file=...) (src/models/ranker.py:219) and Booster.predict
 * (ranker.py:174): raw score = sum over trees of the reached leaf value, float64.
 * X device f32 [n, ldx]; out device f64 [n]. */
int rihip_gbdt_load_text(const char* path, void** handle);
int rihip_gbdt_create_from_text(const char* text, int64_t len, void** handle);
int rihip_gbdt_destroy(void* handle);
int rihip_gbdt_num_trees(void* handle);
int rihip_gbdt_num_features(void* handle);
int64_t rihip_gbdt_feature_names(void* handle, char* buf, int64_t buf_len); /* '\n'-joined, host */
int rihip_gbdt_feature_importance(void* handle, int importance_type, double* out_host);
int rihip_gbdt_predict(void* handle, const float* X, int64_t n, int ldx, double* out, void* stream);

/* ---- ranking-feature assembly ----------------------------------------------------------------
 * Replaces RecommendationPipeline._build_ranking_features (src/serving/recommender.py:213-263) and the
 * feature-store fetch in front of it (recommender.py:319-322) with GPU-resident float64 tables:
 * user_tab [n_user_rows, 24] = avg_rating, log_rating_count, recency_score, gender_encoded, age_normalized,
 * occupation_normalized, genre_pref[18]; item_tab [n_item_rows, 23] = avg_rating, log_rating_count,
 * popularity_score, rating_stddev, year_normalized, genre_vector[18]; row 0 and absent rows = the reference's
 * defaults.  cand_ids [nq,kc] (-1 = padding -> zero row).  col_map[nf]: canonical column index (order of
 * src/features/feature_engineering.py:434-443) of each ranker feature, -1 => 0.0 (recommender.py:334-336).
 * X f32 [nq*kc, nf]: values computed in float64 and cast once, like ranker.py:173. */
int rihip_rank_features_widths(int* user_width, int* item_width, int* n_canonical);
int rihip_rank_features_build(const double* user_tab, int64_t n_user_rows, const double* item_tab,
                              int64_t n_item_rows, const int64_t* user_ids, const int64_t* cand_ids, int64_t nq,
                              int kc, const int* col_map, int nf, float* X, void* stream);
/* nlargest(k, "score") of every request (src/serving/recommender.py:346): scores f64 [nq,kc] (ranker output), cand i64
 * [nq,kc] (-1 = padding, ranked last), retrieval_scores f32 [nq,kc]; outputs [nq,k], ties keep the retrieval order. */
int rihip_rank_topk(const double* scores, const int64_t* cand, const float* retrieval_scores, int64_t nq, int kc, int k,
                    int64_t* out_ids, double* out_scores, float* out_retrieval_scores, void* stream);

/* ---- negative sampler -----------------------------------------------------------------------
 * Replaces UserItemDataset._sample_negative (src/training/train_embeddings.py:58-63) for a batch: neg_out[i] =
 * uniform draw from catalog[], re-drawn (up to max_attempts) while users[i]*key_stride + item is in the sorted
 * array rated_keys[] (one key per rating of any value, :48-50).  Counter-based draws keyed by (seed, i, attempt).
 * gave_up (device int, nullable) counts samples whose attempts were exhausted. */
int rihip_sample_negatives(const int64_t* users, int64_t n, const int64_t* catalog, int64_t n_catalog,
                           const int64_t* rated_keys, int64_t n_rated, int64_t key_stride, uint64_t seed,
                           int max_attempts, int64_t* neg_out, int* gave_up, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RECOMMENDIT_HIP_H */
