/*
 * collie_hip.h — C ABI of libcollie_hip.so: the MI355X (gfx950) implementation of Collie's
 * matrix-factorization train / eval hot path.
 *
 * The reference (ShopRunner/collie, /root/reference) is pure Python and has no FFI layer; the
 * drop-in boundary is its Python object protocol.  Each entry point below
 * therefore cites the reference *Python* interface whose inside it replaces.  The host-side
 * mirror of those Python interfaces lives in collie_amd/ and binds these symbols with ctypes
 * (collie_amd/_lib.py); INTEGRATION.md shows the binding a Collie maintainer would add.
 *
 * Conventions
 *   - every pointer except `*_cfg`/struct arguments is a DEVICE pointer; the caller owns all
 *     memory (outputs and workspaces are allocated by the caller, e.g. by torch);
 *   - all work is enqueued on `stream` (a hipStream_t); nothing here synchronises the device
 *     or allocates memory, so every entry point may be called under HIP graph capture;
 *   - return value 0 = ok, negative = error; `collie_last_error()` gives the text
 *     (thread-local);
 *   - tables are row-major contiguous fp32: embeddings [rows, dim], biases [rows] ([rows,1]);
 *   - index arrays are int32 or int64, selected by a `COLLIE_I32/I64` tag;
 *   - out-of-range ids never fault: the sample is skipped and bit 0 of `*status` (device int32,
 *     may be NULL) is set; the fused loss entry points (collie_mf_train_forward,
 *     collie_mf_explicit_forward) also set bit 1 when the loss they wrote is not finite (what
 *     trainer.py:333's detect_anomaly looks for on the autograd route).
 */
#ifndef COLLIE_HIP_H
#define COLLIE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* collie_stream_t; /* hipStream_t */

enum { COLLIE_I32 = 0, COLLIE_I64 = 1 };

/* collie/loss/{bpr,hinge,warp}.py of the reference; the adaptive variants are what base_pipeline.py:307-338 selects
 * whenever num_negative_samples > 1 */
enum {
  COLLIE_LOSS_BPR = 0,            /* collie/loss/bpr.py:8-89      */
  COLLIE_LOSS_HINGE = 1,          /* collie/loss/hinge.py:8-85    */
  COLLIE_LOSS_ADAPTIVE_BPR = 2,   /* collie/loss/bpr.py:92-167    */
  COLLIE_LOSS_ADAPTIVE_HINGE = 3, /* collie/loss/hinge.py:88-171  */
  COLLIE_LOSS_WARP = 4            /* collie/loss/warp.py:8-154    */
};

/* torch.optim algorithms base_pipeline.py:418-433 can construct */
enum {
  COLLIE_OPT_SGD = 0,         /* torch.optim.SGD (momentum 0), sparse or dense grads  */
  COLLIE_OPT_ADAGRAD = 1,     /* torch.optim.Adagrad, sparse or dense grads           */
  COLLIE_OPT_SPARSE_ADAM = 2, /* torch.optim.SparseAdam (touched rows only)           */
  COLLIE_OPT_ADAM = 3         /* torch.optim.Adam (dense: every row, every step)      */
};

enum { COLLIE_SIDE_USER = 0, COLLIE_SIDE_ITEM = 1 };

/* explicit-feedback losses (collie/model/base/base_pipeline.py:287-292) */
enum { COLLIE_EXPLICIT_MSE = 0, COLLIE_EXPLICIT_MAE = 1 };

#define COLLIE_MAX_META 16
#define COLLIE_MAX_NEG 63 /* fused training path: one wave lane per (positive|negative) score */

/* The four parameter tables of MatrixFactorizationModel._setup_model
 * (collie/model/matrix_factorization.py:106-117). */
typedef struct collie_mf_tables {
  float* user_emb;  /* user_embeddings.weight [num_users, dim] */
  float* item_emb;  /* item_embeddings.weight [num_items, dim] */
  float* user_bias; /* user_biases.weight     [num_users, 1]   */
  float* item_bias; /* item_biases.weight     [num_items, 1]   */
  int64_t num_users;
  int64_t num_items;
  int32_t dim;
} collie_mf_tables;

/* One implicit batch ((users, pos_items), neg_items) as the loaders emit it
 * (collie/interactions/datasets.py:382-388,439-443; base_pipeline.py:608-613).
 * neg element (b, k) is read at neg[b * neg_stride_b + k * neg_stride_k] (element strides). */
typedef struct collie_mf_batch {
  const void* users;
  const void* pos_items;
  const void* neg_items;
  int32_t users_dtype;
  int32_t pos_dtype;
  int32_t neg_dtype;
  int32_t num_neg; /* K */
  int64_t batch;   /* B */
  int64_t neg_stride_b;
  int64_t neg_stride_k;
} collie_mf_batch;

/* Loss selection + partial-credit metadata (collie/loss/metadata_utils.py:6-71): meta[m] is an
 * int32 [num_items] array of category codes; weights must sum to <= 1 (checked by the host). */
typedef struct collie_loss_cfg {
  int32_t kind;
  int32_t n_meta;
  int64_t num_items; /* numerator of the WARP weight ln(num_items / tries), warp.py:120 */
  const int32_t* meta[COLLIE_MAX_META];
  float meta_weight[COLLIE_MAX_META];
} collie_loss_cfg;

/* Hyper-parameters of one torch.optim param group, read by the host from
 * optimizer.param_groups at every step (LR schedulers mutate them between epochs). */
typedef struct collie_optim_cfg {
  int32_t algo;
  int32_t step; /* 1-based step count of this parameter (bias correction / lr_decay) */
  float lr;
  float beta1;
  float beta2;
  float eps;
  float weight_decay;
  float lr_decay;
} collie_optim_cfg;

const char* collie_last_error(void);
int collie_version(void);

/* ---------------------------------------------------------------------------------------------
 * A2  MatrixFactorizationModel.forward(users, items)   matrix_factorization.py:120-159
 * scores[j] = <U[users[j]], I[items[j]]> + bu[users[j]] + bi[items[j]], optional y_range sigmoid.
 * ------------------------------------------------------------------------------------------- */
int collie_mf_score(const collie_mf_tables* t, const void* users, int32_t users_dtype,
                    const void* items, int32_t items_dtype, int64_t m, int32_t has_y_range,
                    float y_lo, float y_hi, float* scores, int32_t* status,
                    collie_stream_t stream);

/* autograd of the above (nn.Embedding backward): per pair j
 *   grad_user_rows[j,:] = g'_j * I[items[j],:],  grad_item_rows[j,:] = g'_j * U[users[j],:],
 *   grad_bias[j] = g'_j, where g'_j = grad_out[j] (times the y_range sigmoid derivative).
 * These are the *values* of the uncoalesced sparse COO gradients whose indices are users/items. */
int collie_mf_score_backward(const collie_mf_tables* t, const void* users, int32_t users_dtype,
                             const void* items, int32_t items_dtype, int64_t m,
                             int32_t has_y_range, float y_lo, float y_hi, const float* grad_out,
                             float* grad_user_rows, float* grad_item_rows, float* grad_bias,
                             int32_t* status, collie_stream_t stream);

/* The same two calls in TRAINING mode with dropout_p > 0 (matrix_factorization.py:143-147:
 * nn.Dropout on the gathered user rows and, independently, on the gathered item rows, before the
 * product).  Element d of pair j is kept with probability 1 - p and scaled by 1 / (1 - p); the
 * multipliers come from Philox4x32-10 with counter (j, d, call) and key `seed`, so the backward
 * call reproduces the forward call's masks from (seed, call) alone -- nothing is stored.  Pass a
 * fresh `call` number for every forward call (the reference draws fresh masks for the positive and
 * the negative pass of one step).  torch's own generator stream cannot be reproduced on a device:
 * the DISTRIBUTION and the arithmetic given a mask are what match the reference. */
int collie_mf_score_dropout(const collie_mf_tables* t, const void* users, int32_t users_dtype,
                            const void* items, int32_t items_dtype, int64_t m,
                            int32_t has_y_range, float y_lo, float y_hi, float dropout_p,
                            uint64_t seed, uint64_t call, float* scores, int32_t* status,
                            collie_stream_t stream);
int collie_mf_score_dropout_backward(const collie_mf_tables* t, const void* users,
                                     int32_t users_dtype, const void* items, int32_t items_dtype,
                                     int64_t m, int32_t has_y_range, float y_lo, float y_hi,
                                     float dropout_p, uint64_t seed, uint64_t call,
                                     const float* grad_out, float* grad_user_rows,
                                     float* grad_item_rows, float* grad_bias, int32_t* status,
                                     collie_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * A4-A7  stand-alone loss callables  f(positive_scores, negative_scores, num_items=, ...)
 * collie/loss/{bpr,hinge,warp,metadata_utils}.py.  s_neg is [K,B] row-major (K = 1 for the
 * non-adaptive kinds).  Writes the scalar loss and (if non-NULL) d loss / d scores.
 * pos_items / neg_items ([B] / [K,B], int64) are only read when cfg->n_meta > 0.
 * ------------------------------------------------------------------------------------------- */
int collie_loss_forward_backward(const collie_loss_cfg* cfg, const float* s_pos,
                                 const float* s_neg, int64_t batch, int32_t num_neg,
                                 const int64_t* pos_items, const int64_t* neg_items,
                                 float* loss_out, float* grad_pos, float* grad_neg,
                                 float* partials /* >= 2*collie_loss_partials(batch) floats */,
                                 collie_stream_t stream);
int64_t collie_loss_partials(int64_t batch);

/* ---------------------------------------------------------------------------------------------
 * A3+A4..A8  BasePipeline.calculate_loss(batch) + loss.backward()
 * base_pipeline.py:603-634, trainer.py:462-463.  One fused kernel: gathers, K+1 dot products per
 * sample, loss, and the per-sample gradient rows staged in `workspace` for the optimizers.
 * need_grad = 0 (validation_step) skips the staging.
 * s_pos_out [B] and s_neg_out [K,B] (reference layout) are optional (NULL to skip).
 *
 * The workspace belongs to ONE model: it must be zero-filled when created and is then reused by
 * every step.  step_id (>= 1, increasing by EXACTLY one per forward with need_grad on a workspace)
 * tags the per-item "looked up in this step" stamps kept inside it (one byte per item, cleared by
 * this call every 255 steps).
 * ------------------------------------------------------------------------------------------- */
size_t collie_mf_train_workspace_bytes(int64_t batch, int32_t num_neg, int32_t dim,
                                       int64_t num_users, int64_t num_items);
int collie_mf_train_forward(const collie_mf_tables* t, const collie_mf_batch* b,
                            const collie_loss_cfg* loss, void* workspace, size_t workspace_bytes,
                            int32_t need_grad, int32_t step_id, float* loss_out, float* s_pos_out,
                            float* s_neg_out, int32_t* status, collie_stream_t stream);

/* A8 (grad.coalesce(), torch/optim/_functional.py:44): ONE stable LSD radix sort over the 3B
 * gradient-carrying lookups of the step (B user rows, B positive and B selected-negative item
 * rows; keys = row id, item rows offset by num_users) + segment heads.  Deterministic: no atomics
 * on floating-point data anywhere.  Must follow collie_mf_train_forward on the same workspace.
 * Looked-up rows that carry no gradient (the K-1 other negatives) are not sorted; they are found
 * through the per-item stamps when an optimizer needs them (SparseAdam). */
int collie_mf_train_build_segments(const collie_mf_tables* t, const collie_mf_batch* b,
                                   void* workspace, size_t workspace_bytes,
                                   collie_stream_t stream);

/* A10/A11  optimizer.step() for the embedding tables (is_bias = 0) or the bias tables
 * (is_bias = 1; 2 after collie_mf_explicit_forward: both bias tables carry a gradient) of one or
 * both sides in one launch.  table_mask: bit 0 = user table, bit 1 = item
 * table; cfg / state1 / state2 / row_stamp are indexed [COLLIE_SIDE_USER], [COLLIE_SIDE_ITEM]
 * (entries of sides not in the mask are ignored).
 * Sparse algorithms (sparse=True models, dense = 0): SGD / ADAGRAD / SPARSE_ADAM touch exactly the
 * looked-up rows, zero-gradient rows included.  Dense algorithms (sparse=False, dense = 1): ADAM /
 * SGD / ADAGRAD additionally sweep every other row with a zero gradient (+ weight decay).
 * state1 = exp_avg | state_sum, state2 = exp_avg_sq (NULL where the algorithm has none).
 * row_stamp: int32 [rows] scratch owned by the optimizer, required when dense != 0 and the
 * zero-gradient update is not the identity (Adam, or weight_decay != 0); NULL otherwise.
 * grad_scale: optional device float the staged gradient is multiplied by (the grad_output that
 * autograd hands to backward; NULL = 1).  step_id: the value given to collie_mf_train_forward. */
int collie_mf_train_update(const collie_mf_tables* t, const collie_mf_batch* b, void* workspace,
                           size_t workspace_bytes, int32_t table_mask, int32_t is_bias,
                           int32_t dense, int32_t step_id, const collie_optim_cfg* cfg /*[2]*/,
                           float* const* state1 /*[2]*/, float* const* state2 /*[2]*/,
                           int32_t* const* row_stamp /*[2]*/, const float* grad_scale,
                           collie_stream_t stream);

/* The trainers' loop body for one batch (trainer.py:458-470: calculate_loss, backward, every
 * optimizer.step) in ONE call: collie_mf_train_forward + collie_mf_train_build_segments (where
 * needed) + collie_mf_train_update_all, with grad_output = 1.  Arrays indexed [user_emb, item_emb,
 * user_bias, item_bias]; cfg[i].algo = -1: table i is not updated; cfg[i].step: that table's
 * 1-based step count.  For launch-bound batch sizes the host side of a step is one FFI call. */
int collie_mf_train_step(const collie_mf_tables* t, const collie_mf_batch* b,
                         const collie_loss_cfg* loss, void* workspace, size_t workspace_bytes,
                         int32_t step_id, int32_t dense, float* loss_out, int32_t* status,
                         const collie_optim_cfg* cfg /*[4]*/, float* const* state1 /*[4]*/,
                         float* const* state2 /*[4]*/, int32_t* const* row_stamp /*[4]*/,
                         collie_stream_t stream);

/* ---- replaying a captured training step -----------------------------------------------------------
 * The loop body of the trainers (collie/model/base/trainer.py:458-470: zero_grad, training_step,
 * backward, optimizer.step for every optimizer) is, on this path, a fixed sequence of kernel
 * launches: at the reference's default batch size (1,024) the host cannot issue them as fast as the
 * GPU runs them.  A caller may therefore capture the sequence into a HIP graph once and replay it.
 * Kernel arguments are frozen in a captured graph, so the arguments that change from step to step
 * -- step_id and the step-dependent scalars the optimizers derive from cfg[].step (bias
 * corrections, Adagrad's decayed rate) -- are read from device memory instead when step_id is
 * COLLIE_STEP_DEVICE:
 *   collie_mf_train_schedule_steps  (NOT capturable: blocks the host, copies from host memory)
 *       writes the schedule of the next n_steps <= COLLIE_STEP_SCHEDULE_MAX steps into the
 *       workspace, computed from cfg exactly as collie_mf_train_update computes them by value (the
 *       replayed trajectory is bit-identical to the one stepped by value).  cfg[4]: the optimizer of
 *       user_emb, item_emb, user_bias, item_bias at the FIRST scheduled step (cfg[i].step = that
 *       table's 1-based step count then; algo = -1: table not updated).  Call it again before the
 *       schedule runs out, after any change of lr / betas / lr_decay, and after any step that was
 *       made by value in between.
 *   collie_mf_train_forward / collie_mf_train_update with step_id = COLLIE_STEP_DEVICE
 *       use the current schedule entry (cfg[].step is ignored, every other cfg field is read at
 *       the call, i.e. frozen at capture).
 *   collie_mf_train_advance_step    the last launch of the step: moves on to the next entry (and
 *       restarts the stamp-tag cycle when that step is due to); also writes the loss if the forward
 *       deferred it (COLLIE_DEFER_LOSS).
 * No counterpart in the reference.
 * ------------------------------------------------------------------------------------------- */
#define COLLIE_STEP_DEVICE (-1)
#define COLLIE_STEP_SCHEDULE_MAX 1024
/* need_grad | COLLIE_DEFER_LOSS for collie_mf_train_forward: the loss partials stay in the workspace
 * and loss_out (+ the non-finite bit of *status) is written by the collie_mf_train_advance_step that
 * ends the step: one launch less in a replayed step, where every launch costs more than the work. */
#define COLLIE_DEFER_LOSS 2
/* need_grad | COLLIE_NO_STAMPS: the forward does not stamp the looked-up items in the workspace.
 * For callers that do not run the library's SparseAdam "looked up, zero gradient" pass on these
 * tables: the row-sharded step computes on per-step compact tables and its OWNERS stamp the rows
 * they serve (collie_shard_serve_rows).  collie_mf_train_update with SPARSE_ADAM on the item side
 * must not follow such a forward. */
#define COLLIE_NO_STAMPS 4
/* Copies one batch (collie_mf_batch: any strides, int32 / int64 ids) into another of the same shape:
 * the batch of the step into the buffers a captured step was recorded on, in ONE launch.  The
 * loaders of the reference hand out fresh tensors per batch (collie/interactions/dataloaders.py). */
int collie_mf_batch_copy(const collie_mf_batch* src, const collie_mf_batch* dst,
                         collie_stream_t stream);
int collie_mf_train_schedule_steps(const collie_mf_tables* t, const collie_mf_batch* b,
                                   void* workspace, size_t workspace_bytes, int32_t first_step_id,
                                   int32_t n_steps, const collie_optim_cfg* cfg /*[4]*/,
                                   collie_stream_t stream);
int collie_mf_train_advance_step(const collie_mf_tables* t, const collie_mf_batch* b,
                                 void* workspace, size_t workspace_bytes,
                                 float* loss_out /* NULL unless the forward deferred the loss */,
                                 int32_t* status, collie_stream_t stream);

/* Whether collie_mf_train_update of this batch reads the segments (1) or not (0): sparse
 * algorithms on batches of at most 4,096 samples (the reference's default is 1,024) with
 * embedding_dim % 4 == 0 and <= 256 are coalesced inside the update launch itself, and
 * collie_mf_train_build_segments may be skipped for them (it is harmless to call it).
 * collie_mf_train_export_grad always needs the segments. */
int collie_mf_train_update_needs_segments(const collie_mf_tables* t, const collie_mf_batch* b,
                                          int32_t dense);

/* collie_mf_train_update for the embedding tables AND the bias tables in one call -- one launch
 * where the sort-free path above applies (a captured step is bound by its number of launches),
 * otherwise the two ordinary calls.  cfg / state1 / state2 / row_stamp are indexed [user_emb,
 * item_emb, user_bias, item_bias]; cfg[i].algo = -1: table i is not updated.  bias_mode: the
 * is_bias value of the bias update (1, or 2 after collie_mf_explicit_forward). */
int collie_mf_train_update_all(const collie_mf_tables* t, const collie_mf_batch* b, void* workspace,
                               size_t workspace_bytes, int32_t bias_mode, int32_t dense,
                               int32_t step_id, const collie_optim_cfg* cfg /*[4]*/,
                               float* const* state1 /*[4]*/, float* const* state2 /*[4]*/,
                               int32_t* const* row_stamp /*[4]*/, const float* grad_scale,
                               collie_stream_t stream);

/* Materialise the coalesced gradient of one table (tests, and `.grad` for callers that ask):
 * rows_out [cap] int64 (sorted unique rows that carry gradient entries), vals_out [cap, dim or 1],
 * count_out [1] int32 (device); cap >= 2 * batch.  touched_out (optional, uint8 [num_items], item
 * side only) = 1 for every item looked up in step_id, zero-gradient negatives included. */
int collie_mf_train_export_grad(const collie_mf_tables* t, const collie_mf_batch* b,
                                void* workspace, size_t workspace_bytes, int32_t side,
                                int32_t is_bias, int32_t step_id, int64_t* rows_out,
                                float* vals_out, int32_t* count_out, uint8_t* touched_out,
                                collie_stream_t stream);

/* Scratch of the stand-alone row-list updates (collie_dense_rows_update): sort + segment buffers
 * for m (row, value) entries of width dim (need not be zeroed). */
size_t collie_sparse_rows_workspace_bytes(int64_t m, int32_t dim);

/* ---------------------------------------------------------------------------------------------
 * Row-sharded training (no counterpart in the reference, which has no multi-GPU path for the
 * tables: SURVEY.md 8e).  One step on a rank, in call order:
 *   collie_shard_plan          requester: distinct rows of the batch, compact numbering
 *   collie_shard_serve_rows    owner: the requested rows as the peers' blocks of the lookup buffer
 *   collie_shard_unpack_bias   requester: bias values of the received buffer, indexed like its rows
 *   collie_mf_train_forward    requester: fused forward on the compact tables (COLLIE_NO_STAMPS)
 *   collie_shard_export_grads  requester: coalesced gradient rows as the owners' blocks
 *   collie_shard_owner_update  owner: coalesce over the requesters, SparseAdam / Adagrad / SGD
 *
 * Exchange buffers ("wire" buffers): matrices of dim-float rows.  The block of one peer holds its
 * c embedding (or gradient) rows, then ceil(c / dim) rows that carry the c bias values and -- in
 * the gradient direction -- as many rows again with the c owner-side row numbers (int32 bits): ONE
 * all-to-all with row splits moves everything.  collie_shard_wire_rows(counts [world], world, dim,
 * gradient) = rows of a buffer whose blocks hold counts[p] entries (-1: bad arguments).
 *
 * collie_shard_plan replaces, for one rank's batch ((users, pos_items), neg_items)
 * (base_pipeline.py:603-613), the host-side de-duplication a sharded nn.Embedding lookup would
 * need: finds the distinct user / item rows the batch looks up (stable radix sort of the (2 + K) B
 * ids: cost proportional to the lookups, not to the tables; or, when the planned tables hold at
 * most 64 rows per lookup, one mark per looked-up row + a popcount scan over the marks -- same
 * outputs), orders them by (owner = id mod
 * world, kind = user|item, id) -- the order in which they are requested from and returned by the
 * owners -- and rewrites the batch in compact row numbers = rows of the lookup buffer the owners will
 * fill (so the numbers skip the bias rows between two owners' blocks).
 *   compact_users [B] int32, compact_pos [B] int32, compact_neg [B, K] int64 (row major):
 *                  compact row numbers, in the id dtypes the loaders use (so that the forward
 *                  kernel takes the same route as on one GPU)
 *   send_rows [n]: row to request, as a row of the owner's combined table (its user rows first,
 *                  then its item rows), dense (no gaps); send_capacity >= min((2+K)B, users+items)
 *   route_counts [2*world + 2] int64: rows per (owner, kind) route; [2*world] = n;
 *                  [2*world + 1] = 1 if an id was outside the tables (it then stands for the
 *                  nearest valid row; the caller raises)
 * local_rank >= 0 ("users are local"): the caller guarantees that this rank owns every user of the
 * batch (a loader that hands each rank the interactions of its own users, or
 * sharded.route_batch_to_user_owners): user rows are then read and updated in place, only the item
 * ids are planned, compact_users [b] = users[b] / world (the local row), and a user owned by
 * another rank is reported through route_counts [2*world + 1] like an id outside the tables.
 * local_rank = -1: users are requested from their owners like items.
 * workspace: collie_shard_plan_workspace_bytes(batch, num_neg, world, num_users, num_items,
 * local_rank) bytes for the same arguments (need not be zeroed; 0: bad arguments).  world <= 16.
 * ------------------------------------------------------------------------------------------- */
int64_t collie_shard_wire_rows(const int64_t* counts, int32_t world, int32_t dim, int32_t gradient);
size_t collie_shard_plan_workspace_bytes(int64_t batch, int32_t num_neg, int32_t world,
                                         int64_t num_users, int64_t num_items, int32_t local_rank);
int collie_shard_plan(const collie_mf_batch* b, int64_t num_users, int64_t num_items,
                      int32_t world, int32_t dim, int32_t local_rank, void* workspace,
                      size_t workspace_bytes,
                      int32_t* compact_users, int32_t* compact_pos, int64_t* compact_neg,
                      int64_t* send_rows, int64_t send_capacity, int64_t* route_counts,
                      collie_stream_t stream);

/* Owner side of the lookup exchange: the rows the peers requested out of this rank's combined
 * table -- the nn.Embedding lookups of matrix_factorization.py:143-150 executed where the rows
 * live -- written as the peers' blocks of the lookup buffer `out` [out_rows, dim].  rows: int64,
 * the peers' request lists back to back (peer_counts [world] entries each, host array; rows of the
 * combined table as collie_shard_plan numbers them); a row outside [0, num_rows) reads as zeros
 * and sets bit 0 of *status.
 * item_touch (optional): uint8 [num_rows - item_base], zero-filled at creation, owned by the
 * model: every served row >= item_base (the item rows of the combined table) is stamped "looked up
 * in step_id" (one byte, cycle of 255 steps: cleared by this call when step_id % 255 == 0; call it
 * once per step with step_id = the optimizers' step of that training step, also with no rows).
 * collie_shard_owner_update finds the looked-up rows that received no gradient through these stamps. */
int collie_shard_serve_rows(const float* emb, const float* bias, int64_t num_rows, int32_t dim,
                            const int64_t* rows, const int64_t* peer_counts, int32_t world,
                            float* out, int64_t out_rows, uint8_t* item_touch, int64_t item_base,
                            int32_t step_id, int32_t* status, collie_stream_t stream);

/* Requester: bias_out[r] = the bias value of row r of the received lookup buffer `buf`
 * [buf_rows, dim] (peer_counts [world]: rows requested from each peer, host array); bias_out
 * [buf_rows] is written at embedding rows only. */
int collie_shard_unpack_bias(const float* buf, int64_t buf_rows, int32_t dim,
                             const int64_t* peer_counts, int32_t world, float* bias_out,
                             collie_stream_t stream);

/* Requester, after collie_mf_train_forward(need_grad = 1 | COLLIE_NO_STAMPS) on the compact
 * tables (t: user_emb == item_emb == the lookup buffer, user_bias == item_bias == the unpacked
 * biases, num_users == num_items >= its rows; b: the batch in compact numbers; train_workspace:
 * that forward's workspace): the backward pass of base_pipeline.py:613-634 as the sparse gradient
 * nn.Embedding(sparse=True) would hold after coalesce() -- one row per distinct user, positive item
 * and selected negative, duplicates summed in batch order, times *grad_scale (B_local / B_global;
 * NULL = 1) -- written as the OWNERS' blocks of the gradient wire buffer (rows, bias gradients, and
 * the rows of the owners' tables they belong to: send_rows of collie_shard_plan).
 *   lookup_counts [world] (host): rows requested from each peer
 *   wire [wire_rows, dim], wire_rows >= 3B + 2 (ceil(3B / dim) + world)
 *   peer_counts [world + 1] int64 (DEVICE, out): gradient rows per owner; [world] = rows of `wire`
 *                  in use.  The caller exchanges the counts and reads them back: the row splits of
 *                  the gradient all-to-all are collie_shard_wire_rows of one count each.
 * users_local != 0 (collie_shard_plan with local_rank >= 0): t->user_emb / user_bias are this
 * rank's own user rows [num_users = local users], t->item_emb / item_bias the lookup buffer; only
 * the item rows are exported, and the call leaves the segments of the whole batch in
 * train_workspace, so that collie_mf_train_update(t, b, ..., table_mask = 1 << COLLIE_SIDE_USER,
 * is_bias = 0, ..., grad_scale) applied next updates the user rows in place (no
 * collie_mf_train_build_segments in between).
 * workspace: collie_shard_export_workspace_bytes(batch). */
size_t collie_shard_export_workspace_bytes(int64_t batch);
int collie_shard_export_grads(const collie_mf_tables* t, const collie_mf_batch* b,
                              void* train_workspace, size_t train_workspace_bytes, int32_t world,
                              int32_t users_local, const int64_t* lookup_counts,
                              const int64_t* send_rows,
                              const float* grad_scale, float* wire, int64_t wire_rows,
                              int64_t* peer_counts, void* workspace, size_t workspace_bytes,
                              collie_stream_t stream);

/* Owner: optimizer.step() of the combined table and its bias table on the gradient blocks received
 * from the peers (torch/optim/_functional.py:41-84 SparseAdam; adagrad.py:361-385; sgd): the
 * blocks are coalesced in peer order (stable sort by row: the sum over the requesters is formed
 * in rank order, bit-reproducible), every row with an entry is updated, and -- SparseAdam -- every
 * item row that collie_shard_serve_rows stamped in this step but that received no entry has its
 * moments decayed and its parameter moved (the "looked up, zero gradient" rows).
 *   wire [wire_rows, dim]: the received buffer; peer_counts [world] (host): entries per peer
 *   row_seg: int32 [num_rows], zero-filled at creation, owned by the model
 *   cfg / bias_cfg: same step; state1 / state2 as collie_sparse_rows_update
 * workspace: collie_shard_owner_workspace_bytes(entries, wire_rows, dim).  Call it every step, also
 * with no entries. */
size_t collie_shard_owner_workspace_bytes(int64_t entries, int64_t wire_rows, int32_t dim);
int collie_shard_owner_update(float* emb, float* bias, int64_t num_rows, int32_t dim,
                              int64_t item_base, const float* wire, int64_t wire_rows,
                              const int64_t* peer_counts, int32_t world, const uint8_t* item_touch,
                              int32_t* row_seg, const collie_optim_cfg* cfg,
                              const collie_optim_cfg* bias_cfg, float* state1, float* state2,
                              float* bias_state1, float* bias_state2, void* workspace,
                              size_t workspace_bytes, int32_t* status, collie_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * A13-A17  get_preds + mapk / mrr / auc   collie/metrics.py:77-107,144-282
 *
 * The [users, items] score matrix of get_preds (metrics.py:102-107) is never written.  Scores are
 * produced in 32x32 tiles by exact-fp32 MFMA (v_mfma_f32_32x32x2_f32) and consumed in registers.
 * All entry points take a *batch* of users (user_ids [n_users], int64) and an item shard
 * [item_begin, item_end) of the catalog, so the same calls serve single-GPU evaluation (one shard
 * = whole catalog) and catalog-sharded evaluation (one shard per GPU, results merged).
 * The user's positives are a CSR slice over the batch: indptr [n_users+1], indices [nnz] (item
 * ids, ascending inside a row as scipy's tocsr() gives them, metrics.py:368).
 * Shards: row r of the local item table is the global item  r * id_stride + id_offset  (single
 * GPU: stride 1, offset 0; round-robin row sharding over G GPUs: stride G, offset rank).  Ids in
 * outputs, `indices` and `best_id` are always GLOBAL ids.
 * ------------------------------------------------------------------------------------------- */

/* largest k the register-resident top-k supports */
int32_t collie_mf_eval_topk_capacity(void);

size_t collie_mf_eval_workspace_bytes(int64_t n_users, int64_t n_items_shard);

/* preds.topk(k) (metrics.py:171) over the shard: topk_ids / topk_scores [n_users, k], ordered by
 * descending score, ascending item id on exact ties.  Scores are y_range-transformed if asked. */
int collie_mf_eval_topk(const collie_mf_tables* t, const int64_t* user_ids, int64_t n_users,
                        int64_t item_begin, int64_t item_end, int64_t id_stride,
                        int64_t id_offset, int32_t k, int32_t has_y_range, float y_lo, float y_hi,
                        float* topk_scores, int64_t* topk_ids, void* workspace,
                        size_t workspace_bytes, collie_stream_t stream);

/* The same call for callers that cannot allocate (code under HIP graph capture, tight serving
 * loops that answer "top-k items for these users", harnesses that forbid hipMalloc).  It has no
 * counterpart in the reference.  The candidate lists live in a file-scope device array inside the
 * library (16 MiB per GPU, held until the process ends) instead of a caller workspace; the call
 * itself allocates nothing.  Limits: n_users <= 1024 per call (larger batches are rejected with an
 * error: use collie_mf_eval_topk).  All calls on one GPU share that GPU's array: calls that
 * overlap on one GPU (different streams or threads) race; serialise them.  The first call on a
 * device looks up the array's device address (hipGetSymbolAddress): make it outside graph
 * capture.  Outputs are bit-identical to collie_mf_eval_topk's. */
int collie_mf_eval_topk_static(const collie_mf_tables* t, const int64_t* user_ids, int64_t n_users,
                               int64_t item_begin, int64_t item_end, int64_t id_stride,
                               int64_t id_offset, int32_t k, int32_t has_y_range, float y_lo,
                               float y_hi, float* topk_scores, int64_t* topk_ids,
                               collie_stream_t stream);

/* collie_mf_eval_topk with the catalogue pass on bf16 MFMA (v_mfma_f32_32x32x16_bf16), OPT-IN: every
 * f32 value is split into three bf16 terms (hi + mid + lo = 24 significant bits) and the six products
 * of total order <= 2 are accumulated in f32 -- 2.7x fewer MFMA cycles than the exact-f32 instruction
 * -- which only RANKS: the 16 best candidates per user in that order are re-scored with the exact
 * chain of collie_mf_eval_topk and the k best of them are emitted, so scores are the same bits and
 * ids differ only if more than 16 - k items lie within ~1e-7 (relative) of the k-th score.
 * Reference: preds.topk(k) of collie/metrics.py:171 on the scores of :102-107.  Limits
 * (collie_mf_eval_bf16x3_ok): embedding_dim 64 or 128, k <= 12, one round; callers fall back to
 * collie_mf_eval_topk otherwise.
 *   collie_mf_eval_split_bf16x3: out = bf16 [rows, 3, dim] (6 * rows * dim bytes, 16-byte aligned),
 *       the item table's split image; valid until the table changes (once per evaluate call).
 *   workspace: collie_mf_eval_bf16x3_workspace_bytes(n_users, item_end - item_begin). */
int32_t collie_mf_eval_bf16x3_ok(int32_t dim, int32_t k);
int collie_mf_eval_split_bf16x3(const float* table, int64_t rows, int32_t dim, void* out,
                                collie_stream_t stream);
size_t collie_mf_eval_bf16x3_workspace_bytes(int64_t n_users, int64_t n_items_shard);
int collie_mf_eval_topk_bf16x3(const collie_mf_tables* t, const void* item_split,
                               const int64_t* user_ids, int64_t n_users, int64_t item_begin,
                               int64_t item_end, int64_t id_stride, int64_t id_offset, int32_t k,
                               int32_t has_y_range, float y_lo, float y_hi, float* topk_scores,
                               int64_t* topk_ids, void* workspace, size_t workspace_bytes,
                               collie_stream_t stream);

/* Catalog-sharded evaluation: merge per-shard top-k candidate lists (cand_* [n_users, nslots,
 * slot_len]: every shard's own top-k in slot_len >= its length entries, unused entries = (-inf,
 * INT32_MAX)) into the global top-k, any k <= nslots * slot_len (metrics.py:170-175 allows any
 * k <= num_items).  No counterpart in the reference (it has no sharded evaluation). */
int collie_mf_eval_merge_topk(const float* cand_scores, const int32_t* cand_ids, int64_t n_users,
                              int32_t nslots, int32_t slot_len, int32_t k, float* topk_scores,
                              int64_t* topk_ids, collie_stream_t stream);

/* Scores of the positives, computed with the same fmaf chain as the MFMA tiles so that they are
 * bit-identical to the catalog scores of the same pairs:
 *   pos_scores [nnz]  raw model score (before y_range)
 *   thr_lo/thr_hi [nnz] (optional) the raw-score interval whose fp32 sigmoid(preds) equals the
 *       positive's: AUC ranks sigmoid(preds) (metrics.py:277), which ties distinct raw scores
 *   best_score / best_id [n_users] (optional) the user's best positive = first relevant item of
 *       the descending order (metrics.py:224-238)
 * compute_scores = 1: pos_scores are computed (entries whose item lives on another shard get 0, so
 * that a sum all-reduce over the shards completes them); 0: pos_scores are given (already
 * all-reduced) and only the thresholds / best positives are derived from them. */
int collie_mf_eval_pos_scores(const collie_mf_tables* t, const int64_t* user_ids, int64_t n_users,
                              const int64_t* indptr, const int64_t* indices, int64_t nnz,
                              int32_t has_y_range, float y_lo, float y_hi, int32_t compute_scores,
                              int64_t id_stride, int64_t id_offset, float* pos_scores,
                              float* thr_lo, float* thr_hi, float* best_score, int64_t* best_id,
                              collie_stream_t stream);

/* Second MFMA pass: rank counts of the positives against the shard, ADDED into counts
 * [n_users, 3] (uint64; zero before the first shard):
 *   [0] #{j : s_j > best  or (s_j == best and j < best_id)}                (MRR)
 *   [1] sum over the user's positives e of #{j : sigmoid(s_j) <  sigmoid(s_e)}   (AUC)
 *   [2] sum over e of #{j : sigmoid(s_j) <= sigmoid(s_e)}                        (AUC ties) */
int collie_mf_eval_rank_counts(const collie_mf_tables* t, const int64_t* user_ids,
                               int64_t n_users, const int64_t* indptr, const float* thr_lo,
                               const float* thr_hi, const float* best_score,
                               const int64_t* best_id, int64_t item_begin, int64_t item_end,
                               int64_t id_stride, int64_t id_offset, uint64_t* counts,
                               collie_stream_t stream);

/* collie_mf_eval_topk and collie_mf_eval_rank_counts of the same users and item range in ONE pass
 * over the catalogue: `evaluate_in_batches([mapk, mrr, auc], ...)` (metrics.py:379-384 calls every
 * metric on the same score matrix) then pays for one scoring GEMM instead of two.  Same outputs,
 * bit for bit, as the two calls; the thresholds come from collie_mf_eval_pos_scores, which needs no
 * top-k.  Shapes the LDS-DMA kernel does not cover run the two passes internally. */
int collie_mf_eval_topk_rank_counts(const collie_mf_tables* t, const int64_t* user_ids,
                                    int64_t n_users, int64_t item_begin, int64_t item_end,
                                    int64_t id_stride, int64_t id_offset, int32_t k,
                                    int32_t has_y_range, float y_lo, float y_hi,
                                    const int64_t* indptr, const float* thr_lo,
                                    const float* thr_hi, const float* best_score,
                                    const int64_t* best_id, float* topk_scores, int64_t* topk_ids,
                                    uint64_t* counts, void* workspace, size_t workspace_bytes,
                                    collie_stream_t stream);

/* Per-user AP@k (float64, metrics.py:178-196), reciprocal rank (fp32, :237-238) and tie-averaged
 * AUC of sigmoid(preds) (:266-280) from the top-k ids and the rank counts.  Any of ap / rr / auc
 * may be NULL. */
int collie_mf_eval_metrics(const int64_t* topk_ids, int32_t k, int64_t n_users, int64_t n_items,
                           const int64_t* indptr, const int64_t* indices, const float* pos_scores,
                           const uint64_t* counts, int32_t has_y_range, float y_lo, float y_hi,
                           double* ap, float* rr, double* auc, collie_stream_t stream);

/* ---- metric callables on a materialised score matrix (generic route) ----------------------------
 * mapk / mrr / auc(targets, user_ids, preds, k) (collie/metrics.py:144-282) receive
 * preds [n_rows, n_cols] from get_preds (metrics.py:77-107) for ANY model and catalogue size.
 * preds is row-major with row stride ld (elements); n_cols < 2^31.
 *
 * collie_preds_topk: preds.topk(k, dim=1) (metrics.py:171) for any 1 <= k <= n_cols: topk_scores /
 * topk_ids [n_rows, k], descending score, ascending column on exact ties (-0.0 == +0.0; NaN above
 * +inf, as torch.topk).  No workspace. */
int collie_preds_topk(const float* preds, int64_t n_rows, int64_t n_cols, int64_t ld, int32_t k,
                      float* topk_scores, int64_t* topk_ids, collie_stream_t stream);

/* The positives' scores gathered from preds (CSR over the rows: indptr [n_rows+1], indices [nnz]
 * ascending inside a row, metrics.py:135-141) plus what collie_mf_eval_pos_scores derives from
 * them: thr_lo / thr_hi [nnz] (raw-score interval with the same fp32 sigmoid, metrics.py:277) and
 * the best positive per row (metrics.py:224-238).  Column ids outside [0, n_cols) set bit 0 of
 * *status and score 0. */
int collie_preds_pos_scores(const float* preds, int64_t n_rows, int64_t n_cols, int64_t ld,
                            const int64_t* indptr, const int64_t* indices, int64_t nnz,
                            float* pos_scores, float* thr_lo, float* thr_hi, float* best_score,
                            int64_t* best_id, int32_t* status, collie_stream_t stream);

/* Rank counters of collie_mf_eval_rank_counts (same three definitions) over preds, ADDED into
 * counts [n_rows, 3] (uint64, zero before the call): what mrr (metrics.py:224-240: full descending
 * order) and auc (metrics.py:266-282) need, without sorting the rows. */
int collie_preds_rank_counts(const float* preds, int64_t n_rows, int64_t n_cols, int64_t ld,
                             const int64_t* indptr, const float* thr_lo, const float* thr_hi,
                             const float* best_score, const int64_t* best_id, uint64_t* counts,
                             collie_stream_t stream);

/* item_item_similarity / user_user_similarity (base_pipeline.py:785-864): cosine similarity of every
 * row of one embedding table [rows, dim] with row `id`; out [rows]. */
int collie_embedding_similarity(const float* emb, int64_t rows, int32_t dim, int64_t id, float* out,
                                collie_stream_t stream);

/* ---- explicit feedback ----------------------------------------------------
 * Replaces the explicit branch of BasePipeline.calculate_loss (base_pipeline.py:635-650):
 * preds = forward(users, items) (matrix_factorization.py:143-157, y_range sigmoid rescale
 * included), loss = torch.nn.MSELoss / L1Loss(reduction='mean')(preds, ratings), and the
 * score-level backward, staged in the training workspace (collie_mf_train_workspace_bytes(batch,
 * 1, dim, ...)) exactly as collie_mf_train_forward stages the implicit step.  Follow with
 * collie_mf_train_build_segments and collie_mf_train_update on a batch descriptor whose
 * pos_items / neg_items both point at `items` (num_neg = 1); the bias update must be called with
 * is_bias = 2 (the user-bias gradient is not identically zero for an explicit loss).
 * preds_out (optional): [batch] predictions.  need_grad = 0: loss (and preds) only. */
int collie_mf_explicit_forward(const collie_mf_tables* t, const void* users, int32_t users_dtype,
                               const void* items, int32_t items_dtype, const float* ratings,
                               int64_t batch, int32_t loss_kind, int32_t has_y_range, float y_lo,
                               float y_hi, void* workspace, size_t workspace_bytes,
                               int32_t need_grad, float* loss_out, float* preds_out,
                               int32_t* status, collie_stream_t stream);

/* ---- device-side batch assembly + negative sampling ---------------------
 * Replaces Interactions.__getitem__ / _negative_sample (collie/interactions/datasets.py:375-445)
 * and the index shuffling of the samplers (collie/interactions/samplers.py:43-60) for one batch:
 * sample t is dataset entry order[start + t] (order == NULL: dataset order) of the COO arrays
 * rows / cols [nnz]; outputs are the loaders' batch format, users int32 [batch], pos int32
 * [batch], neg int64 [batch, num_neg] (datasets.py:382-388), written in HBM.
 *   indptr / indices: CSR of the positives (column ids ascending inside a row), or both NULL for
 *       approximate sampling (max_number_of_samples_to_consider == 0, datasets.py:427-443);
 *   max_tries: max_number_of_samples_to_consider (rejections per sample, datasets.py:411);
 *   seed, position_base: the draws of sample t come from Philox4x32-10 with key = seed and counter
 *       = (position_base + start + t, draw number): pass position_base = epoch * nnz so that every
 *       epoch draws fresh negatives.  (The reference's stream is CPython random / numpy global
 *       state: only the distribution and the rejection rule are reproduced.) */
int collie_sample_batch(const int32_t* rows, const int32_t* cols, int64_t nnz,
                        const int64_t* order, int64_t start, int64_t batch,
                        const int64_t* indptr, const int32_t* indices, int64_t num_items,
                        int32_t num_neg, int32_t max_tries, uint64_t seed, uint64_t position_base,
                        int32_t* users_out, int32_t* pos_out, int64_t* neg_out,
                        collie_stream_t stream);

/* ---- hybrid head: dense layers (BASELINE configs[4]) ------------------------
 * HybridPretrainedModel.forward (collie/model/hybrid_pretrained_matrix_factorization.py:327-464):
 * metadata[ids, :] -> Linear + LeakyReLU stack (:350-361); torch.cat((user_meta_out, U[users],
 * I[items], item_meta_out), 1) (:430-459) -> Linear + LeakyReLU stack -> Linear(.., 1) + user bias
 * + item bias (:387-399); and the autograd of those ops.  Exact-fp32 MFMA GEMMs.
 *
 * An operand [m, width] is the column-wise concatenation of up to COLLIE_DENSE_MAX_SEGS segments
 * and is never materialised.  Segment s contributes `width` columns: row j of the operand reads
 * row idx[j] of src (idx == NULL: row j) -- the nn.Embedding / metadata[ids, :] gathers.  If
 * act_ref != NULL the value is multiplied by LeakyReLU'(act_ref[j, c]) (1 if > 0 else slope): the
 * activation backward (F.leaky_relu, :357) fused into the operand load; act_ref is the layer's saved
 * OUTPUT (same sign as its input). */
#define COLLIE_DENSE_MAX_SEGS 4
typedef struct collie_dense_seg {
  const float* src;     /* [rows, ld] */
  const void* idx;      /* [m] int32/int64 row numbers into src, or NULL */
  const float* act_ref; /* [m, act_ld] or NULL */
  int64_t ld;
  int64_t act_ld;
  int64_t rows; /* rows of src; ids outside [0, rows) read as zeros and set bit 0 of *status */
  int32_t idx_dtype;
  int32_t width;
} collie_dense_seg;

typedef struct collie_dense_operand {
  int32_t n_seg;
  collie_dense_seg seg[COLLIE_DENSE_MAX_SEGS];
} collie_dense_operand;

/* Kernel family behind collie_dense_forward / collie_dense_forward_add / collie_dense_weight_grad:
 * COLLIE_DENSE_AUTO (default) picks the LDS-DMA kernels for the shapes they take (the layers of the
 * reference's default head, hybrid_pretrained_matrix_factorization.py:110) and the register-staged
 * kernels for everything else; COLLIE_DENSE_REGISTER_STAGED always takes the latter.  The two families
 * give the same bits (except n_in = 384 forwards, 32 x 64 and > 65,536-row weight gradients: 1e-5), so
 * this exists for comparison runs and tests only.  Process-wide; not to be changed while launches of
 * another host thread are being issued.  No counterpart in the reference. */
#define COLLIE_DENSE_AUTO 0
#define COLLIE_DENSE_REGISTER_STAGED 1
int collie_dense_set_kernels(int32_t family);

/* nn.Linear (+ F.leaky_relu): y[j, n] = act(sum_k x[j, k] * W[n, k] + bias[n]), W[n, k] read at
 * w[n * w_ld_out + k * w_ld_in] (a Linear weight: w_ld_out = n_in, w_ld_in = 1; its transpose for
 * the data backward dX = dZ W: w_ld_out = 1, w_ld_in = row length of the weight).  bias may be
 * NULL; leaky = 0: no activation.  If user_bias / item_bias are given the result additionally gets
 * + user_bias[users[j]] + item_bias[items[j]] (:394-398, the final layer).  y is [m, y_ld]. */
int collie_dense_forward(const collie_dense_operand* x, int64_t m, const float* w, int64_t w_ld_out,
                         int64_t w_ld_in, const float* bias, int32_t n_in, int32_t n_out,
                         int32_t leaky, float slope, const void* users, int32_t users_dtype,
                         const float* user_bias, int64_t num_users, const void* items,
                         int32_t items_dtype, const float* item_bias, int64_t num_items, float* y,
                         int64_t y_ld, int32_t* status, collie_stream_t stream);

/* collie_dense_forward with one more term before the activation: y = act(x W^T + b + addend[m mod
 * add_period, :]), addend [add_period, >= n_out] floats with row pitch add_ld, add_period >= 1 (the
 * LDS-DMA kernels take add_period >= 256, shorter periods go to the register-staged ones).
 * For operands whose leading column segments only depend on m mod add_period -- calculate_loss feeds
 * users.repeat(K + 1) (base_pipeline.py:616-621), so the user embedding's share of the first combined
 * layer (hybrid_pretrained_matrix_factorization.py:430-446) is the same for the 1 + K rows of a
 * sample: it is computed once per sample and added here, and the layer's GEMM runs over the
 * remaining columns of W.  addend == NULL: exactly collie_dense_forward. */
int collie_dense_forward_add(const collie_dense_operand* x, int64_t m, const float* w,
                             int64_t w_ld_out, int64_t w_ld_in, const float* bias, int32_t n_in,
                             int32_t n_out, int32_t leaky, float slope, const void* users,
                             int32_t users_dtype, const float* user_bias, int64_t num_users,
                             const void* items, int32_t items_dtype, const float* item_bias,
                             int64_t num_items, const float* addend, int64_t add_ld,
                             int64_t add_period, float* y, int64_t y_ld, int32_t* status,
                             collie_stream_t stream);

/* The rows of a training batch whose score gradient can differ from zero, for the backward of the
 * head.  torch.autograd pushes all (1 + K) B rows of `preds` through every backward GEMM of
 * hybrid_pretrained_matrix_factorization.py:401-464; under adaptive BPR / hinge and WARP
 * (loss/bpr.py:153-167, hinge.py:157-171, warp.py:73-154) only the positive and ONE negative of a
 * sample have a non-zero d loss / d score.  g [(1 + K) B]: the score gradient in calculate_loss's
 * order (base_pipeline.py:616-622: B positives, then negative k of sample b at B + k B + b);
 * sample b keeps row b and row B + k* B + b, k* = the first k with the largest |g| (k = 0 if all
 * are zero).  Outputs, 2 B long, positive and negative of a sample side by side: act (the row
 * numbers), g_out, users_out, items_out (g, users, items at those rows; ids int64). */
int collie_active_rows(const float* g, int64_t batch, int32_t num_neg, const int64_t* users,
                       const int64_t* items, int64_t* act, float* g_out, int64_t* users_out,
                       int64_t* items_out, collie_stream_t stream);

/* calculate_loss's expansion of a batch into its (1 + K) B (user, item) pairs
 * (base_pipeline.py:603-622: users.repeat(K + 1); cat(pos_items, neg_items.T.flatten())), ids
 * widened to int64: pair k B + b is (users[b], k == 0 ? pos[b] : neg[b, k - 1]).  One launch
 * instead of the five torch ops; all_users / all_items are [(1 + K) B] int64. */
int collie_pair_ids(const collie_mf_batch* b, int64_t* all_users, int64_t* all_items,
                    collie_stream_t stream);

/* outs[t][j, :] = srcs[t][idx[j], :] for up to 4 row-major float tensors at once (the saved
 * activations of the head gathered down to the rows above: torch.index_select, one launch instead
 * of one per tensor).  srcs[t] is [m_src, lds[t]] with widths[t] used columns, outs[t] is
 * [m_out, widths[t]] contiguous; the arrays srcs / lds / widths / outs are HOST arrays of n
 * entries.  idx outside [0, m_src) reads as zeros and sets bit 0 of *status (may be NULL). */
int collie_gather_rows(int32_t n, const float* const* srcs, const int64_t* lds, const int32_t* widths,
                       float* const* outs, const int64_t* idx, int64_t m_out, int64_t m_src,
                       int32_t* status, collie_stream_t stream);

/* The first combined layer on get_preds' users x catalogue grid (metrics.py:77-107: row j of the
 * expanded pairs is user j / period, item j mod period) once the layer's two shares are known:
 * y[j, :] = act((outer[j / period, :] + bias) + inner[j mod period, :]), j < n_outer * period;
 * outer [n_outer, n] = the user rows times their columns of W, inner [period, n] = the item side
 * (both from collie_dense_forward with bias = NULL, leaky = 0).  Replaces the nn.Linear + leaky_relu of
 * hybrid_pretrained_matrix_factorization.py:430-446 on that grid; same rounding order as
 * collie_dense_forward_add's epilogue.  bias may be NULL. */
int collie_dense_outer_sum(const float* outer, int64_t outer_ld, int64_t n_outer, const float* inner,
                           int64_t inner_ld, int64_t period, const float* bias, int32_t n,
                           int32_t leaky, float slope, float* y, int64_t y_ld,
                           collie_stream_t stream);

/* get_preds' users x catalogue grid through the reference's DEFAULT head (combined_layers_dims =
 * [128, 64, 32], hybrid_pretrained_matrix_factorization.py:110) in one launch: for u < n_users,
 * i < period: x1 = leaky((user_share[u] + b1) + item_share[i]) (as collie_dense_outer_sum),
 * x2 = leaky(W2 x1 + b2) (W2 [64, 128]), x3 = leaky(W3 x2 + b3) (W3 [32, 64]),
 * y[u * y_ld + i] = w4 . x3 + b4 + user_bias[users[u]] + item_bias[item0 + i] (:430-464, :394-398).
 * No activation is written to memory.  users: the n_users ids (for the bias table only); user_bias /
 * item_bias may both be NULL; slope <= 1.  Same arithmetic as the layer-by-layer route up to
 * summation order. */
int collie_dense_grid_tail(const float* user_share, int64_t us_ld, int64_t n_users,
                           const float* item_share, int64_t is_ld, int64_t period, const float* b1,
                           const float* w2, const float* b2, const float* w3, const float* b3,
                           const float* w4, const float* b4, float slope, const void* users,
                           int32_t users_dtype, const float* user_bias, int64_t num_users,
                           int64_t item0, const float* item_bias, int64_t num_items, float* y,
                           int64_t y_ld, int32_t* status, collie_stream_t stream);

/* autograd of nn.Linear w.r.t. its parameters: dw[n, k] = sum_j dz[j, n] * x[j, k] ([n_out, n_in]
 * contiguous), db[n] = sum_j dz[j, n]; dz is usually the operand (dY, act_ref = Y).  Either output
 * may be NULL.  Deterministic (fixed chunks of rows, partials added in fixed order). */
size_t collie_dense_weight_grad_workspace_bytes(int64_t m, int32_t n_out, int32_t n_in);
int collie_dense_weight_grad(const collie_dense_operand* dz, const collie_dense_operand* x,
                             int64_t m, int32_t n_out, int32_t n_in, float slope, float* dw,
                             float* db, void* workspace, size_t workspace_bytes, int32_t* status,
                             collie_stream_t stream);

/* torch.optim.{Adam,SGD,Adagrad}.step on one dense parameter with a dense gradient (the head's
 * Linear weights; base_pipeline.py:406-437 builds one optimizer over all parameters of the hybrid
 * model).  state1 = exp_avg | state_sum, state2 = exp_avg_sq.  grad_scale: optional device float. */
int collie_dense_optim_step(float* param, const float* grad, int64_t n,
                            const collie_optim_cfg* cfg, float* state1, float* state2,
                            const float* grad_scale, collie_stream_t stream);

/* The same step for up to 16 parameters of ONE optimizer group (same cfg) in one launch: the head's
 * ten small Linear weights and biases.  params / grads / numels / state1 / state2 are HOST arrays of
 * n entries (state arrays may be NULL where the algorithm keeps no such state). */
int collie_dense_optim_step_multi(int32_t n, float* const* params, const float* const* grads,
                                  const int64_t* numels, const collie_optim_cfg* cfg,
                                  float* const* state1, float* const* state2,
                                  collie_stream_t stream);

/* The same optimizers on a table whose dense gradient is given as an UNCOALESCED COO list (rows [m],
 * vals [m, dim]): what nn.Embedding(sparse=False).backward + a dense optimizer do to the bias /
 * embedding tables of the hybrid model.  Rows with entries get the summed gradient (input order,
 * deterministic); every other row gets the zero-gradient update where that is not the identity
 * (Adam's moment decay, weight decay), which needs row_stamp = int32 [num_rows] scratch owned by the
 * optimizer (zero-filled at creation).  workspace: collie_sparse_rows_workspace_bytes(m, dim).
 * m = 0 (rows / vals / workspace may be NULL): an all-zero gradient, every row takes the
 * zero-gradient update. */
int collie_dense_rows_update(float* table, int64_t num_rows, int32_t dim, const void* rows,
                             int32_t rows_dtype, const float* vals, int64_t m,
                             const collie_optim_cfg* cfg, float* state1, float* state2,
                             int32_t* row_stamp, void* workspace, size_t workspace_bytes,
                             int32_t* status, collie_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* COLLIE_HIP_H */
