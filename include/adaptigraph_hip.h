/*
 * adaptigraph_hip.h — C ABI of libadaptigraph_hip.so, the MI355X (gfx950) engine for AdaptiGraph's
 * message-passing rollout hot path.
 *
 * The reference has no FFI/plugin layer: its boundary is a Python call surface.  Each
 * entry point below names the reference symbol it stands behind; INTEGRATION.md shows the ctypes stub a
 * reference maintainer would add.  Conventions:
 *   - every data pointer is a DEVICE pointer (tensor.data_ptr()) unless marked host; fp32 / int32 / uint8
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing synchronises the host
 *   - the caller owns all memory, including the scratch `workspace` sized by the *_workspace_bytes queries;
 *     the library allocates only the packed weights inside ag_model
 *   - return 0 on success, negative on error; ag_last_error() gives the message (thread-local)
 *   - one ag_model may be used from one host thread at a time; distinct models/streams may run concurrently
 *   - inputs are never mutated; outputs are fully overwritten
 */
#ifndef ADAPTIGRAPH_HIP_H
#define ADAPTIGRAPH_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: exactly the functions declared between this push and the pop at the end of the
 * header are exported (tests/test_abi.py holds `nm -D` to that list). */
#pragma GCC visibility push(default)

typedef struct ag_model ag_model;
typedef void *ag_stream_t; /* hipStream_t */

enum { AG_VARIANT_SINGLE = 0, AG_VARIANT_BATCH = 1 };
enum { AG_HEIGHT_MIN = 0, AG_HEIGHT_MASKED_MEAN = 1 };

/* Derived from model_config / material_config / dataset_config exactly as DynamicsPredictor.__init__ does
 * (src/dynamics/gnn/model.py:63-126). */
typedef struct ag_model_config {
    int32_t nf;           /* nf_particle == nf_relation == nf_effect (150)                  model.py:79-81   */
    int32_t n_his;        /* dataset_config['n_his'] (4)                                    model.py:77      */
    int32_t attr_dim;     /* model_config['attr_dim'] == rel_attr_dim (2)                   model.py:98,110  */
    int32_t phys_dim;     /* #physics_params with use: True                                 model.py:91-94   */
    int32_t action_dim;   /* 3                                                              model.py:99      */
    int32_t pstep;        /* propagation steps (3)                                          model.py:278     */
    float motion_clamp;   /* 100                                                            model.py:85      */
} ag_model_config;

const char *ag_last_error(void);
int ag_version(void);

/* DynamicsPredictor.__init__ + load_state_dict (model.py:63-126; checkpoint layout SURVEY.md §8b):
 * `weights` = 22 HOST pointers to fp32 tensors in state_dict order
 *   particle_encoder.model.{0,2,4}.{weight,bias}, relation_encoder.model.{0,2,4}.{weight,bias},
 *   particle_propagator.linear.{weight,bias}, relation_propagator.linear.{weight,bias},
 *   non_rigid_predictor.linear_{0,1,2}.{weight,bias}           (nn.Linear weight = (out,in) row-major).
 * Packs them into MFMA-ready chunk streams on the current HIP device. */
int ag_model_create(const ag_model_config *cfg, const float *const *weights, ag_model **out);
int ag_model_update_weights(ag_model *m, const float *const *weights);
int ag_model_destroy(ag_model *m);

/* Engine knobs (all have sane defaults; used by bench.py for A/B passes):
 *   "rollout_streams"  0..4  ag_rollout runs the batch as this many independent parts on separate streams; 0 (default): the engine's choice by
 *                            workload, see ag_rollout_streams_for
 *   "fuse_aggregate"   0/2   segment reduce as its own HBM-streaming kernel (0, default) or inside node_update through an LDS stage
 *                            (2, precision 2 only: no `agg` table; measured equal solo, -4 % in the two-stream rollout); bit-identical results
 *   "max_blocks"       n     persistent grid size (default 2 x #CUs)
 *   "edge_products"    2/3   precision 2 only: arithmetic of the EDGE stack: 2 = fp16 (default: split-fp16 weights x fp16 activations + an e5m2
 *                            residual byte per activation, three fp16 MFMAs per fp32 product; models whose edge weights exceed the fp16 range
 *                            keep 3), 3 = split-bf16 like precision 1
 *   "edge_stationary"  0/1   with edge_products 2: 1 = weight-stationary kernel (default: weights in registers, activations handed from wave
 *                            to wave through LDS), 0 = streaming kernel (weights through LDS per 128 edges); bit-identical results
 *   "node_stationary"  0/1   precision 1 / 2, rounds before the last: 1 = weight-stationary node update (default: one workgroup per CU keeps the three
 *                            layers' split-bf16 weights in registers, 32-row blocks pipelined through LDS), 0 = streaming kernel (weights through
 *                            LDS per 128 rows); bit-identical results
 *   "node_dedup"       0/1/2 particle_encoder / hoisted Pn / the first round's Hr, Hs computed once per DISTINCT [attrs | phys | action] row of a sample
 *                            (the node encoder sees no positions, model.py:168-195) and read through an index, once per ag_rollout call: 1 (default) =
 *                            where it pays (>= 32 768 node-rows x steps per call), 2 = always, 0 = never (once per node and model step).
 *                            Bit-identical results
 *   "self_edges"       0/1   ag_rollout (the engine's own edge builder): 1 (default) = a particle's self-loop (graph.py:68-75; its edge inputs are
 *                            [a_n, a_n, 0, 0 ...], model.py:228-253 with r = s, a function of the node's attribute pair only) is left out of the edge
 *                            encoder and of the per-edge table for the attribute classes (1, 0) and (0, 1) — every driver of the reference produces
 *                            nothing else — one table row per class is encoded instead and the segment reduce adds it at the self-loop's position
 *                            in the receiver's order (10 % / 17 % / 5 % fewer rows at rope-1k / cloth-4k / granular-2k); other attribute pairs keep
 *                            their self-loops as real edges.  0 = every edge through the pipeline.  Bit-identical results
 *   "shared_state"     0/1   ag_rollout: 1 = exploit that dynamics() rolls ONE cloud out under `B` sampled pushes (forward_dynamics.py:11-38;
 *                            config/planning/rope.yaml:39-42: 20 000 samples per planning step).  The trajectory of the cloud WITHOUT a tool (the caller's
 *                            sample 0 with its tool slots invalid) is rolled out once as an extra internal sample; per model step every sample's full edge
 *                            lists are still built, but the encoders and the propagation rounds run only over the rows whose result can differ from the
 *                            base's — nodes whose inputs or earlier predictions differ in any bit (tool slots always), rows whose edge list differs from
 *                            the base's, and their 3-hop closure (three propagation rounds) — and every other particle takes the base's prediction.
 *                            Results equal the plain rollout bit for bit for ANY input (samples whose states differ from sample 0's are simply all
 *                            private); one stream, node de-duplication forced on, workspace of ag_rollout_workspace_bytes_for(model, ...) with the option
 *                            set.  0 (default) = every sample in full — what the headline benchmark is quoted on.  Not combined with
 *                            "fuse_aggregate" 2 / "cu_split" (those calls take the plain path)
 *   "agg_q16"          0/1   precision 2 with the defaults of "fuse_aggregate" / "node_stationary": 1 = the per-node sums of a propagation round (`agg`,
 *                            model.py:295) travel from the segment reduce to node_update as 16-bit block-scaled rows (the per-edge table's format, 320 B
 *                            instead of 640 B per node and round).  One more rounding per node and round: NOT bit-identical to 0 (default) — measured
 *                            deviation and gain in docs/NEGATIVE_RESULTS.md R6.4; ignored where a kernel of the round does not read the format
 *   "cu_split"         0|8k  CU-partitioned rollout (off by default): the first `cu_split` CU-mask bits (cu_split / 8 CUs of every XCD) run the
 *                            MFMA-bound edge encoder, the other CUs the HBM-bound edge build / segment reduce / node update / state step, the batch
 *                            parts pipelined through the two partitions on two CU-masked queues (hipExtStreamCreateWithCUMask).  Bit-identical
 *                            results; measured NOT faster than sharing the chip (docs/NEGATIVE_RESULTS.md R5.1, profiles/r05_cu_split_sweep_rope.txt)
 *   "precision"        0/1/2 0 = exact fp32 MFMA; 1 = split-bf16 ("bf16x3": x = hi + lo, 3 bf16 MFMAs per product,
 *                            fp32 accumulate; 1e-6..8e-6 abs deviation from the reference forward, gate 1e-4);
 *                            2 (default) = mode 1 for the node-level stacks, the edge stack in fp16 with residual bytes ("edge_products" 2) and the
 *                            per-edge Eterm table in 16-bit block-scaled fixed point (q16: half the dominant HBM stream); the same deviation
 *                            class as mode 1 at any motion size (random sweeps, trained weights, actions up to +-0.5) */
int ag_set_option(ag_model *m, const char *name, int value);

/* The model's CURRENT value of an option of ag_set_option — whatever set it (default, AG_* environment at ag_model_create, ag_set_option):
 * what a caller that accounts for the engine's work (bench.py's roofline bytes) must ask instead of re-reading the environment.
 * No reference counterpart (the reference has no engine options). */
int ag_get_option(const ag_model *m, const char *name, int *value);

/* Sticky numeric status of a model, read-and-clear (synchronises `stream`): bit 0 (AG_STATUS_NONFINITE) = some forward on this
 * model left the range of its arithmetic.  In precision mode 2 that is (a) an activation of the fp16 edge stack beyond +-65504 — detected in
 * the epilogue that produces it, whatever later layers make of the inf — or (b) a non-finite value reaching the per-edge table or a message
 * sum; with finite inputs both mean a checkpoint with large activations: switch the model to precision 1 (fp32 range).  In every mode,
 * non-finite inputs raise it through the message sums.  The reference has no counterpart (fp32 throughout, model.py:283-295).
 * (Until r03 bit 1 flagged mode-2 forwards that predicted motions above 0.125, where that round's arithmetic could leave the 1e-4 gate; the
 * r04 arithmetic has no such range and the bit is never set.) */
enum { AG_STATUS_NONFINITE = 1 };
int ag_model_status(ag_model *m, int *flags /*host*/, ag_stream_t stream);

/* Upper bound on the edge count the builder can emit: B*N*(min(N,topk) + (connect_tools_all ? max_tools : 0)). */
int64_t ag_edge_capacity(int B, int N, int topk, int connect_tools_all, int max_tools);
size_t ag_edges_workspace_bytes(int B, int N, int topk, int connect_tools_all, int max_tools);

/* construct_edges_from_states (variant SINGLE, src/dynamics/dataset/graph.py:38-89) and
 * construct_edges_from_states_batch (variant BATCH, graph.py:91-156) for B samples at once.
 *   pos (B,N,3) f32; mask, tool_mask (B,N) u8; thr_sq (B) f32 = squared radius rounded as the variant does
 *   (SINGLE: (float)((double)r*r), BATCH: (float)r*(float)r); topk <= 64; max_tools >= #tool slots per sample.
 * Out: row_ptr (B*N+1) i32 over global rows b*N+i (row_ptr[B*N] = total edges, stays on device),
 *      edge_recv / edge_send (e_cap) i32 GLOBAL node ids in the reference's (receiver, sender) order.
 * Equivalent dense outputs: Rr[b, e - row_ptr[b*N], edge_recv[e] - b*N] = 1 (graph.py:152-155). */
int ag_build_edges(const float *pos, const uint8_t *mask, const uint8_t *tool_mask, const float *thr_sq, int topk,
                   int connect_tools_all, int variant, int B, int N, int max_tools, int32_t *row_ptr,
                   int32_t *edge_recv, int32_t *edge_send, int64_t e_cap, void *workspace, size_t workspace_bytes,
                   ag_stream_t stream);

/* Scratch sizes.  ag_*_workspace_bytes(...) holds for ANY model / option setting (per-edge table sized as fp32 rows);
 * ag_*_workspace_bytes_for(m, ...) is exact for model `m` as configured NOW — in precision mode 2 the per-edge table is 16-bit (q16) rows,
 * half the largest buffer: query right before the call (a call checks its own carving against workspace_bytes and fails with AG_ERR_WS,
 * never overruns).  The compact tables of the node-encoder de-duplication are bounded (8 shared rows per sample + ~B N / 16 private
 * rows): inputs with more distinct [attrs | phys | action] rows run through the per-node encoder, same bits.
 * Rollout, rope-1k x 256: 3.5 -> 2.1 GB (2.9 for any mode); the reference planner's 20 000 x 200: 56 -> 33 GB (46). */
size_t ag_forward_workspace_bytes(int B, int N, int64_t e_cap);
size_t ag_forward_workspace_bytes_for(const ag_model *m, int B, int N, int64_t e_cap);

/* DynamicsPredictor.forward (model.py:129-313) on a CSR adjacency instead of one-hot Rr/Rs.
 *   state (B,n_his,N,3), attrs (B,N,2), action (B,N,3), p_instance (B,n_p,n_instance), phys (B,phys_dim)
 *   row_ptr/edge_recv/edge_send as produced by ag_build_edges (edges sorted by receiver).
 * Out: pred_pos, pred_motion (B,n_p,3). */
int ag_forward(ag_model *m, const float *state, const float *attrs, const float *action, const float *p_instance,
               int n_instance, const float *phys, const int32_t *row_ptr, const int32_t *edge_recv,
               const int32_t *edge_send, int64_t e_cap, int B, int N, int n_p, float *pred_pos, float *pred_motion,
               void *workspace, size_t workspace_bytes, ag_stream_t stream);

/* Inner rollout loop of dynamics() / dynamics_masked() (src/planning/forward_dynamics.py:125-197 / :319-393):
 * edges -> forward -> record-on-repeat -> tool advance -> history shift -> edge rebuild, n_steps times,
 * with no host synchronisation. */
typedef struct ag_rollout_params {
    int32_t B, N, n_p;            /* samples, slots per sample (objects + tools), object slots                */
    int32_t n_instance;
    int32_t topk, connect_tools_all, max_tools;
    int32_t n_steps;              /* max(action_repeat), forward_dynamics.py:156                              */
    int32_t height_mode;          /* AG_HEIGHT_MIN (:163) or AG_HEIGHT_MASKED_MEAN (:359)                     */
    float gripper_raise;          /* 0.01 * sim_real_ratio if gripper_enable else 0 (:167-168)                */
} ag_rollout_params;

size_t ag_rollout_workspace_bytes(const ag_rollout_params *p);
size_t ag_rollout_workspace_bytes_for(const ag_model *m, const ag_rollout_params *p);
/* Number of batch parts (one HIP stream each) ag_rollout will use for these parameters: the "rollout_streams" option, or with its default 0 the engine's
 * choice by workload (one stream, two where the edge stack dominates: top-k >= 16), never more than B / 8. */
int ag_rollout_streams_for(const ag_model *m, const ag_rollout_params *p);

/*   state0 (B,n_his,N,3) initial history incl. tool slots; delta (B,N,3) per-step tool motion (graph["action"]);
 *   attrs (B,N,2); p_instance (B,n_p,n_instance); phys (B,phys_dim); mask/tool_mask (B,N) u8;
 *   obj_mask (B,n_p) u8 (only read in MASKED_MEAN mode, may be NULL otherwise); thr_sq (B);
 *   repeat (B) i32 = action_repeat.
 * Out: out_seq (B,n_p,3) = prediction of step repeat[b] (rows with repeat outside 1..n_steps are left untouched);
 *      state_final (B,n_his,N,3) optional (may be NULL). */
int ag_rollout(ag_model *m, const ag_rollout_params *p, const float *state0, const float *delta, const float *attrs,
               const float *p_instance, const float *phys, const uint8_t *mask, const uint8_t *tool_mask,
               const uint8_t *obj_mask, const float *thr_sq, const int32_t *repeat, float *out_seq,
               float *state_final, void *workspace, size_t workspace_bytes, ag_stream_t stream);

/* chamfer(x, y) of src/planning/losses.py:4-10 — the MPPI error term:
 *   x (B,N,3) predicted particles, y (B,M,3) if y_batched else (1,M,3) target cloud  ->  out (B)
 *   out[b] = mean_m min_n ||x[b,n]-y[.,m]|| + mean_n min_m ||x[b,n]-y[.,m]||; N + M <= 12800. */
int ag_chamfer(const float *x, const float *y, int B, int N, int M, int y_batched, float *out, ag_stream_t stream);

/* mean_chamfer(state_pred, state_real, pred_mask, real_mask) of src/planning/losses.py:12-24 — the sys-id objective
 *  — for the whole batch in one launch: chamfer over the points whose mask byte is non-zero.
 *   x (B,N,3), x_mask (B,N) u8, y (B|1,M,3), y_mask (B|1,M) u8  ->  out (B); NaN where a side has no valid point. */
int ag_chamfer_masked(const float *x, const uint8_t *x_mask, const float *y, const uint8_t *y_mask, int B, int N, int M,
                      int y_batched, float *out, ag_stream_t stream);

/* ---- training path: graph pieces of DynamicsPredictor.forward and their adjoints on the CSR adjacency.
 * Plain row-major fp32 tensors of arbitrary feature width D; every reduction runs in a fixed order (no atomics).
 *
 * ag_gather_rows:  out[e,:] = x[idx[e],:]                    — replaces Rr.bmm(X) / Rs.bmm(X), model.py:224-249,283-284
 * ag_segment_sum:  out[n,:] = sum_{k in [ptr[n],ptr[n+1])} vals[perm ? perm[k] : k, :]
 *                  — replaces Rr_t.bmm (model.py:295) with perm = NULL over the receiver-sorted edges, and is the adjoint of
 *                  a gather by any index given that index's (pointer, permutation) view (receivers: row_ptr/NULL;
 *                  senders: col_ptr/stable argsort of send)
 * ag_message_forward:  agg[n,:] = sum_{e in row n} relu((eterm[e,:] + hr[n,:]) + hs[send[e],:])
 *                  — relation_propagator + Rr_t.bmm of one round (model.py:283-295) after the W_rp column split
 * ag_message_backward: grad_edge[e,:] = grad_agg[recv(e),:] * [pre-activation > 0]  (= d loss / d eterm[e]),
 *                  grad_hr[n,:] = sum_{e in row n} grad_edge[e,:];  d loss / d hs = ag_segment_sum(grad_edge, col_ptr, perm) */
int ag_gather_rows(const float *x, const int32_t *idx, float *out, int64_t n_out, int D, ag_stream_t stream);
int ag_segment_sum(const float *vals, const int32_t *ptr, const int32_t *perm, float *out, int64_t n_seg, int D, ag_stream_t stream);
int ag_message_forward(const float *eterm, const float *hr, const float *hs, const int32_t *row_ptr, const int32_t *send, float *agg,
                       int64_t n_nodes, int D, ag_stream_t stream);
int ag_message_backward(const float *eterm, const float *hr, const float *hs, const int32_t *row_ptr, const int32_t *send,
                        const float *grad_agg, float *grad_edge, float *grad_hr, int64_t n_nodes, int D, ag_stream_t stream);

/* ---- training path, dense stacks (row n4): the Linear(+ReLU) chains of DynamicsPredictor.forward and their backward as fused
 * MFMA kernels (activations stay in registers between layers, in both directions; arithmetic per call: `precision` below — the
 * Python training path defaults to split-bf16).  Replaces the forward/backward of
 *   AG_CHAIN_EDGE     relation_encoder.model.{0,2,4} (+ReLU each) and relation_propagator.linear[:, :nf] (no ReLU)   model.py:274,289
 *   AG_CHAIN_NODE     particle_encoder.model.{0,2,4} (+ReLU each)                                                    model.py:268
 *   AG_CHAIN_DECODER  non_rigid_predictor.linear_{0,1} (+ReLU), linear_2                                             model.py:306
 * in src/dynamics/train/train.py:90-112 (forward + loss.backward()).
 *
 * ag_train_pack: device-side packing of one layer into the kernels' chunk-image format (weights change every optimiser step):
 *   op(W)[o][k] = transposed ? W[k*ld + col0 + o] : W[o*ld + col0 + k] (o < n_out, k < n_in), bias (nullable) as column n_in;
 *   `compact` = the single-chunk first-layer image (n_in + 1 <= 32), else `n_tiles` 32-row images; dst gets
 *   (compact ? 1 : n_tiles) * 5120 floats.  Forward streams hold the layers in order (first layer compact for EDGE/NODE);
 *   backward streams hold W_{L-1}^T .. W_1^T (5 images each, no bias) then W_0^T (1 image for EDGE/NODE, 5 for DECODER).
 * ag_train_chain: forward (backward = 0): x ([rows][d_in] dense for EDGE/NODE, [rows_pad][160] for DECODER) -> y[l], l < L, each
 *   [rows_pad][160] fp32 (rows_pad = rows rounded up to 128; columns >= 150 and padding rows are scratch).  backward = 1:
 *   dy ([rows_pad][160], gradient w.r.t. y[L-1]) + the saved y[l] -> dz[l] (pre-activation gradients, [rows_pad][160]) and
 *   dx (same shape as x; nullable for EDGE/NODE).  Weight gradients are dz[l]^T y[l-1] — plain library GEMMs left to the caller.
 *   `y` / `dz` are HOST arrays of L device pointers.
 * `precision`: 0 = exact fp32 MFMA, non-zero = split-bf16 (as the inference engine's mode 1); the packed stream and the chain call
 *   must use the same value. */
enum { AG_CHAIN_EDGE = 0, AG_CHAIN_NODE = 1, AG_CHAIN_DECODER = 2 };
int ag_train_pack(const float *W, const float *bias, int n_out, int n_in, int ld, int col0, int transposed, int compact, int n_tiles,
                  int precision, float *dst, ag_stream_t stream);
int ag_train_chain(int kind, int backward, int precision, const float *x, const float *packed, float *const *y, const float *dy,
                   float *const *dz, float *dx, int64_t rows, int d_in, ag_stream_t stream);

/* rel_inputs of DynamicsPredictor.forward (model.py:220-253) and its adjoint, from one per-node table
 *   tab (n_nodes, D) = [attrs (attr_dim) | group = cat(p_instance, 0) (group_dim) | state_norm (D - attr_dim - group_dim)]:
 *   out[e] = [tab[r][:A] | tab[s][:A] | sum |tab[r][A:A+G] - tab[s][A:A+G]| | tab[r][A+G:] - tab[s][A+G:]], width 2A + 1 + (D - A - G),
 * with r = recv[e], s = send[e] (global node ids).  Backward: grad_tab[n] = sum over the edges n receives (row_ptr order) and
 * sends (col_ptr / send_perm order) of the per-edge gradients; scratch_r / scratch_s are (n_edges, D) each. */
int ag_edge_inputs_forward(const float *tab, int D, int attr_dim, int group_dim, const int32_t *recv, const int32_t *send, float *out, int64_t n_edges,
                           ag_stream_t stream);
int ag_edge_inputs_backward(const float *tab, int D, int attr_dim, int group_dim, const int32_t *recv, const int32_t *send, const int32_t *row_ptr,
                            const int32_t *col_ptr, const int32_t *send_perm, const float *grad_out, float *scratch_r, float *scratch_s, float *grad_tab,
                            int64_t n_edges, int64_t n_nodes, ag_stream_t stream);

/* The node update's residual, Propagator.forward with res (model.py:36-40): y = relu((a + b) + c) over n contiguous floats
 * (n % 4 == 0, 16-byte aligned), and its adjoint out = g * [y > 0] (the same for all three inputs). */
int ag_add3_relu(const float *a, const float *b, const float *c, float *y, int64_t n, ag_stream_t stream);
int ag_relu_mask(const float *g, const float *y, float *out, int64_t n, ag_stream_t stream);

/* Weight and bias gradients of up to 4 dense layers in two launches: for layer l,
 *   out[l][o][k] = sum_rows dz[l][row][o] * prev[l][row][k]   (k < n_in[l])      = d loss / d W_l[o][k]
 *   out[l][o][n_in[l]] = sum_rows dz[l][row][o]                                   = d loss / d b_l[o]
 * dz[l] has row stride dz_ld[l] <= 160 (160 for the tables of ag_train_chain backward; n_out for a dense (rows, n_out) gradient),
 * prev[l] is the layer's input with row stride prev_ld[l] (>= n_in[l]); out is [n_layers][160][160] fp32.  Rows are split into slabs, partial sums meet in a fixed order: bit-reproducible.
 * dz / dz_ld / prev / prev_ld / n_in are HOST arrays. */
size_t ag_train_weight_grads_workspace_bytes(int64_t rows, int n_layers);
int ag_train_weight_grads(int n_layers, const float *const *dz, const int32_t *dz_ld, const float *const *prev, const int32_t *prev_ld, const int32_t *n_in,
                          int64_t rows, float *out, void *workspace, size_t workspace_bytes, ag_stream_t stream);

/* The same, ACCUMULATING layer l's gradients straight into caller storage (a parameter's .grad): w_grad[l][o * w_grad_ld[l] + k] +=
 * dW_l[o][k] for o < n_out[l], k < n_in[l], and b_grad[l][o] += db_l[o] (b_grad or b_grad[l] may be NULL).  A layer whose w_grad[l]
 * is NULL goes to `out` as above.  This is what removes autograd's per-parameter slice / clone / accumulate kernels from the
 * training step (train.py:110-112 loss.backward() accumulates into .grad the same way). */
int ag_train_weight_grads_into(int n_layers, const float *const *dz, const int32_t *dz_ld, const float *const *prev, const int32_t *prev_ld,
                               const int32_t *n_in, int64_t rows, float *out, float *const *w_grad, const int32_t *w_grad_ld, float *const *b_grad,
                               const int32_t *n_out, void *workspace, size_t workspace_bytes, ag_stream_t stream);

/* Optional per-kernel timing with HIP events recorded on the caller's stream around every launch of each
 * kernel class (used by bench.py for the roofline line; off by default, costs two event records per launch).
 * ag_profile_read synchronises on the recorded events and returns, per class, the summed milliseconds, the
 * launch count, and for AG_K_EDGE_ENCODE the summed number of edges processed (units for the roofline). */
enum { AG_K_EDGES = 0, AG_K_NODE_ENCODE = 1, AG_K_EDGE_ENCODE = 2, AG_K_AGGREGATE = 3, AG_K_NODE_UPDATE = 4,
       AG_K_ROLLOUT_STEP = 5, AG_K_COUNT = 6 };
int ag_profile_enable(ag_model *m, int enable);
int ag_profile_read(ag_model *m, double *ms /*AG_K_COUNT*/, int64_t *launches /*AG_K_COUNT*/, int64_t *edges);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* ADAPTIGRAPH_HIP_H */
