// ag_common.h — shared constants, layouts and internal kernel-launch prototypes (gfx950 only).
//
// Data layouts in HBM (all fp32 unless noted):
//   row-major activation tables   [rows][AG_FP]      AG_FP = 160 (= nf_effect 150 padded to 5 x 32),
//                                                    640-byte rows, columns 150..159 are zero
//   packed ("fragment-image") node tables [rows/32][5 tiles][4 q][2 h][32 j][4 p]:
//       element (row g, feature f) lives at (g/32)*5120 + (((f/32)*4 + (f%32)/8)*2 + (f%8)/4)*128 + (g%32)*4 + f%4,
//       i.e. exactly the register image of the MFMA operand a lane holds, so a wave moves it with
//       fully coalesced 16-byte accesses (used for tensors only the MLP kernels touch: h, Pn)
//   CSR/COO adjacency: row_ptr[B*N+1] (global rows = b*N+i), edge_recv[E], edge_send[E] as GLOBAL node
//       ids, edges sorted by (receiver, sender) == the reference's nonzero() order (graph.py:151)
//   weight streams: per fused kernel, a sequence of 20480-byte chunk images (one 32-feature out-tile each),
//       input columns [0,K) weights, column K the bias (K = fan-in), rest zero; copied linearly into LDS:
//       F32: [32 out][160] floats, element (i, k) at i*160 + 4*((k/4) ^ ((i>>1)&7)) + k%4 (16-byte XOR swizzle)
//       B3 : [10 steps u][hi|lo][64 lanes l=(i,h)][8 slots e] bf16, slot e = column 16u + 8(e>>2) + 4h + (e&3)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define AG_F 150             // nf_particle = nf_relation = nf_effect (config/dynamics/*.yaml)
#define AG_FP 160            // padded feature stride (5 MFMA tiles of 32)
#define AG_NT 5              // 32-wide feature tiles
#define AG_WSTRIDE 160       // row stride of a weight chunk image (floats); 16-byte columns XOR-swizzled
#define AG_CHUNK_FLOATS 5120 // 32 out-features x 160 = 1280 float4 = 5 per thread of a 256-thread block
#define AG_CHUNK_F4 1280
#define AG_ROWS_PER_WAVE 32
#ifndef AG_MLP_THREADS
#define AG_MLP_THREADS 256      // threads per MLP workgroup (256: two workgroups per CU; 512: one, sharing one weight ring)
#endif
#define AG_MLP_WAVES (AG_MLP_THREADS / 64)
#define AG_ROWS_PER_BLOCK (32 * AG_MLP_WAVES)
#define AG_MLP_WG_PER_CU (512 / AG_MLP_THREADS)
#define AG_PACK_BLOCK (32 * AG_FP)
#define AG_NHIS 4            // n_his compiled into the edge-feature prologue (config/dynamics/*.yaml n_his: 4)
#define AG_ATTR 2            // attr_dim = rel_attr_dim = 2
#define AG_EDGE_IN 17        // 2*attr + group + 3*n_his  (model.py:109-113)
#define AG_NODE_IN_MAX 8     // attr + phys + action <= 8 (model.py:96-101)

enum { AG_PREC_F32 = 0, AG_PREC_B3 = 1 };
enum { AG_OK = 0, AG_ERR_ARG = -1, AG_ERR_HIP = -2, AG_ERR_WS = -3, AG_ERR_CONFIG = -4 };

struct AgWeights {           // device pointers into the packed weight streams (float4-aligned)
    const float4 *node_encode;   // PE0(1 compact chunk) PE1 PE2 | PPa(+b_pp) | Wr | Ws   26 chunks
    const float4 *edge_encode;   // RE0(1 compact chunk) RE1 RE2 | We(+b_rp)                16 chunks
    const float4 *node_mid;      // PPb | Wr | Ws                               15 chunks
    const float4 *node_last;     // PPb | D0 | D1 | D2(1 chunk)                 16 chunks
    // the same streams as split-bf16 fragment images (precision AG_PREC_B3)
    const float4 *node_encode_b3, *edge_encode_b3, *node_mid_b3, *node_last_b3;
    // the edge stream of the fp16 edge stack (precision mode 2, PrecH3): chunk 0 = the first layer as a split-fp16 fragment image, chunks 1..15 =
    // the wide units as [hi fp16 fragments | block-scaled fp8 A operands] (ag_api.hip pack_layer_h3), and their block scales (128 dwords per unit)
    const float4 *edge_encode_h2;
    const uint32_t *edge_scale_h3;
};

struct AgFwdArgs {
    // caller tensors (device)
    const float *state;      // (B, H, N, 3)
    const float *attrs;      // (B, N, 2)
    const float *action;     // (B, N, 3)
    const float *p_instance; // (B, n_p, n_inst)
    const float *phys;       // (B, phys_dim)
    const int32_t *row_ptr;  // (B*N + 1)
    const int32_t *edge_recv, *edge_send;  // (E_cap) global node ids
    float *pred_pos, *pred_motion;         // (B, n_p, 3)
    // workspace tables
    float *h, *pn;           // packed node tables
    float *hr, *hs, *agg;    // row-major node tables
    float *eterm;            // row-major edge table
    int B, N, n_p, n_inst, phys_dim, e_cap, pstep;
    float clamp;
    unsigned long long *edge_counter;   // optional (profiling): += number of edges per edge_encode launch
    float *hr_out, *hs_out;   // where node_update writes the NEXT round's Hr/Hs (ping-pong with hr/hs)
    int agg_reverse;          // segment reduce walks the nodes (and with them the per-edge table) from the END: see run_propagate
    int hs_q16, hs_out_q16;   // precision mode 2: the sender table `hs` read / the one written by this launch holds q16 rows (320 B, the Eterm format) instead
                              // of fp32 rows — the rounds after the first; round 0 gathers the node encoder's fp32 rows
    int precision;     // AG_PREC_F32 (exact fp32 MFMA) or AG_PREC_B3 (hi/lo bf16 split, 3 MFMAs per product)
    int eterm_half;    // 1: the Eterm table is 16-bit block-scaled fixed point in accumulator order (q16, precision mode 2)
    int fuse_agg;      // 1: node_update does the segment reduce itself (no aggregate launch, no agg table)
    int max_blocks;    // persistent grid size = resident workgroups (2 per CU)
    float *edge_node_tab;  // (rows_pad, 16): per-node inputs of the edge features [attr0, attr1, group0, 0, v0(3), v1(3), v2(3), x_cur(3)] (weight-stationary edge encoder)
    int *tile_ctr;     // zeroed int: row-tile claim counter of this forward's edge_encode launch (NULL: static grid stride)
    int *status;       // sticky device word of the model: bit 0 = a forward left the range of its arithmetic (ag_model_status)
    int edge_products; // precision mode 2 only: 2 = fp16 edge stack with residual bytes (PrecH3, default), 3 = split-bf16 like mode 1
    int ws_blocks;     // workgroups of the weight-stationary edge encoder for this launch
    int edge_ws;       // with edge_products == 2: 1 = weight-stationary kernel (default), 0 = streaming kernel
    int node_ws;       // split-bf16 node_update of the rounds before the last on the weight-stationary kernel (ag_mlp.hip: node_update_nws_kernel)
    int agg_q16;       // precision mode 2, option "agg_q16" (default 1): the segment reduce stores `agg` as q16 rows (320 B, the Eterm row layout with unsigned
                       // values, ag_q16_encode_segment) and node_update decodes them; the reduce fused into node_update rounds its sums the same way
    // ---- node-encoder de-duplication.  The node encoder sees [attrs | phys | action] only (positions do not enter:
    // model.py:168-173 is skipped for state_dim = 0), and every rollout driver of the reference gives all object particles of a sample
    // the same row (forward_dynamics.py:83-123: attrs (1,0), the sample's physics parameter, zero action), so particle_encode, the hoisted
    // Pn and the first round's Hr / Hs are computed ONCE per distinct row (node_classify_kernel) into compact tables and read through
    // node_row; rows that match none of a sample's first AG_DEDUP_REPS distinct rows simply get a private compact row.
    int dedup;                       // 1: the fields below are live
    int32_t *node_row;               // (rows_pad) compact row of every node: b * (N + AG_DEDUP_REPS) + k (shared) / + AG_DEDUP_REPS + i (private)
    int32_t *enc_row, *enc_src;      // (rows_c) work list of node_encode: compact row to produce, a node that has this input row
    int *enc_count;                  // work-list length (device word, zeroed per forward)
    float *h0c, *pnc, *hrc, *hsc;    // compact row-major tables [rows_c + 128][160]: particle_encode (= h of round 0), hoisted Pn, round 0's Hr / Hs
    int32_t *send_c;                 // (e_pad) edge_send mapped to compact rows: the first round's sender gathers
    int rows_c;                      // compact rows incl. padding; rows [rows_c, rows_c + 128) are dump rows of out-of-range lanes
    // The compact tables are BOUNDED (r05): B * AG_DEDUP_REPS shared rows + a private budget of ~B N / 16 rows handed out through `priv_count`.
    // A call whose inputs need more private rows raises `*ovf` (node_classify_kernel) and runs as if de-duplication were off: the compact
    // encoder returns at once, the per-node encoder (launched every step behind a test of the flag) fills the full-size tables, and the first
    // round's kernels read those (hr_full / hs_full, Pn and h from the packed tables).  Both paths give the same bits.
    int *priv_count, *ovf;           // device words, zeroed with enc_count; ovf == NULL: no de-duplication in this call
    int shared_rows;                 // = B * AG_DEDUP_REPS: first private row
    const float *hr_full, *hs_full;  // round 0 only: the full-size tables the per-node encoder writes (read instead of hr / hs when *ovf)
    // per-launch views set by the sequencer (ag_api.hip: run_propagate)
    const int32_t *hr_row;           // segment reduce: row of Hr to read for node g (NULL: g) — node_row in round 0
    const float *pn_rows;            // node_update: Pn from compact rows pn_rows[node_row[g]] (NULL: packed table pn)
    const float *h_rows;             // node_update, round 0: residual h from compact rows (NULL: packed table h)
    // set per model step by ag_rollout: the edge builder's launches of this step already produced edge_node_tab / send_c (AgEdgeArgs riders)
    int tab_done, remap_done;
    // ---- self-edge elision (r06; ag_rollout with the engine's own edge builder; option "self_edges").  A particle's self-loop (graph.py:68-75: d = 0 is
    // always in radius and, barring >= top-k exact duplicates, in the top-k) has the edge inputs [a_n, a_n, |g_n - g_n| = 0, x_n - x_n = 0 ...]
    // (model.py:228-253 with r = s): its relation_encode / Eterm row depends on the node's ATTRIBUTE CLASS only.  The builder leaves the self-loops of
    // the two classes every driver of the reference produces — (1, 0) object, (0, 1) tool — out of the COO list the edge encoder walks and out of the
    // per-edge table; the encoder computes ONE row per class instead (AG_SELF_ROWS synthetic edges appended behind the list: table rows E, E + 1), and
    // the segment reduce adds that row at the self-loop's position in the receiver's (ascending-sender) order, so every sum keeps its order: same bits.
    // self_info[g] = (class << 16) | position of the elided self-loop in g's row, or -1 (no self-loop, or another attribute pair: kept as a real edge);
    // row_ptr / edge_recv / edge_send then describe the graph WITHOUT the elided edges.  NULL: nothing is elided (ag_forward on a caller's CSR).
    const int32_t *self_info;
    // ---- shared-state rollout (r06, ag_shared.hip; option "shared_state"): the propagation kernels run over a COMPACT row set — the base sample's nodes
    // followed by every sample's private nodes — whose size only the device knows.  Non-NULL n_rows_dev / e_count_dev replace B * N / row_ptr[B * N]
    // as the row and edge counts (B * N stays the upper bound the launches are sized by); row_orig[row] is the node (sample * N + particle) a row
    // stands for — the decoder round needs the sample's state and writes pred_pos / pred_motion BY ROW (rows x 3) instead of by (sample, particle).
    const int *n_rows_dev, *e_count_dev;
    const int32_t *row_orig;
    int self_class_row0;      // row of class 0 in edge_node_tab (= the layout's rows_pad) = the node id the synthetic edges carry as receiver and sender
    int self_rows;            // synthetic class edges behind the COO list (AG_SELF_ROWS with elision, else 0): the edge encoders walk row_ptr[B N] + self_rows edges
};
#define AG_SELF_CLASSES 2
#define AG_SELF_REPL 32            // copies of each class row in the per-edge table: every node reads one (by node id), so that 256 k nodes do not hammer one cache line
#define AG_SELF_ROWS (AG_SELF_CLASSES * AG_SELF_REPL)
__device__ __forceinline__ int ag_rows(const AgFwdArgs &a) { return a.n_rows_dev ? *a.n_rows_dev : a.B * a.N; }
__device__ __forceinline__ int ag_edges(const AgFwdArgs &a) { return a.e_count_dev ? *a.e_count_dev : a.row_ptr[a.B * a.N]; }
// rows of the per-edge table / entries of the COO arrays for a graph of at most e_cap edges (+ the class rows), in whole 256-row tiles
__host__ __device__ __forceinline__ size_t ag_edge_rows_pad(long long e_cap) { return ((size_t)(e_cap > 0 ? e_cap : 1) + AG_SELF_ROWS + 255) / 256 * 256; }
// attribute class of a node for self-edge elision: 0 = (1, 0), 1 = (0, 1), -1 = anything else
__device__ __forceinline__ int ag_self_class(float a0, float a1) { return (a0 == 1.0f && a1 == 0.0f) ? 0 : ((a0 == 0.0f && a1 == 1.0f) ? 1 : -1); }
#define AG_TILE_CTRS 4
#define AG_DEDUP_REPS 8            // distinct node-encoder input rows shared within a sample (more than that: private rows)

// ---- the per-edge table of precision mode 2: "q16", block-scaled 16-bit fixed point --------------------------------------------
// A row of Eterm is 320 bytes = [5 out-tiles t][2 lane halves h][16 values] in accumulator order (feature 32t + 8q + 4h + p at 16-bit
// index 32t + 16h + 4q + p: a lane of the producing MFMA kernel writes 32 contiguous bytes per tile).  The 32 values of one out-tile
// share a power-of-two scale: with m = max |v| over the tile and eb = the biased fp32 exponent of m (2^(eb-127) <= m < 2^(eb-126)),
//     stored  q = round_to_nearest_even(v * 2^(126-eb) * 32767)   (v_cvt_pknorm_i16_f32),     value = q * 2^(eb-126) / 32767,
// i.e. an absolute error <= 2^(eb-142) ~ m / 46 000 for EVERY value of the tile (fp16, the r01-r03 format, rounds each value to 2^-12 of
// ITSELF: the same 16 bits measured 7 x further from the exact forward, tools/scheme_err.py).  eb is clamped to [AG_Q16_EB_MIN, 252].
// The five exponent bytes live in the row's own padding (features 150..159 never exist: 20 bytes): bytes 280 + t (t < 4) and 284 (t = 4)
// from the h = 0 lanes, and once more at 316 + t / 312 from the h = 1 lanes, so that BOTH 16-byte segments that hold padding carry the
// byte a consumer lane needs in their THIRD dword: segment 17 (bytes 272..287) tiles 0..3, segment 19 (bytes 304..319) tile 4.
#define AG_Q16_EB_MIN 20
#define AG_Q16_EB_MAX 252
__device__ __forceinline__ int ag_q16_exp_byte_offset(int t, int h) { return h ? (t == 4 ? 312 : 316 + t) : (t == 4 ? 284 : 280 + t); }
__device__ __forceinline__ float ag_q16_scale(int eb) { return ldexpf(1.0f / 32767.0f, eb - 126); }

// ---- segment reduce of ONE node over the q16 table, shared by aggregate_half_kernel (ag_aggregate.hip) and the reduce fused into
//      node_update (ag_mlp.hip) ------------------------------------------------------------------------------------------------------
// Twenty adjacent lanes of ONE wave own a node (three nodes per wave, lanes 60..63 idle); lane c (0..19) owns the 16-byte segment c of
// every edge row = features f0 + {0..3} and f0 + 8 + {0..3}, f0 = ag_half_lane_feature(c), of out-tile c >> 2.  The tile's exponent byte
// arrives by ONE ds_bpermute per edge from the third dword of lane 17's (tiles 0..3) or lane 19's (tile 4) own load — no extra memory
// instruction (the reduce is bound by the texture pipe).
// A node has ~10 edges and every edge costs a dependent index -> row round trip, so FOUR edges are kept in flight per lane and
// the sender indices of the next four are fetched one iteration ahead (the adds still run in ascending edge order: bit-identical to a
// sequential loop).
// round 0 of a de-duplicated call that overflowed the compact tables: the per-node encoder's full-size tables by node id
__device__ __forceinline__ void ag_overflow_view(AgFwdArgs &a)
{
    if (a.ovf && a.hr_full && *a.ovf != 0) { a.hr = const_cast<float *>(a.hr_full); a.hs = const_cast<float *>(a.hs_full); a.hr_row = nullptr; }
}
#ifndef AG_AGG_IN_FLIGHT
#define AG_AGG_IN_FLIGHT 4
#endif
#define AG_AGG_GROUP 20             // lanes per node
#define AG_AGG_NODES_PER_WAVE 3
__device__ __forceinline__ int ag_half_lane_feature(int c) { return 32 * (c >> 2) + 8 * (2 * (c & 1)) + 4 * ((c >> 1) & 1); }

// HSQ: the sender table is q16 too (rows of the Eterm format written by node_update's RowStoreQ16Epi): ONE 16-byte load per edge and lane instead of
// two, half the gathered bytes.  Sender terms are independent per edge, so their 16-bit rounding averages out over a receiver's edges (float64
// emulation on the random sweep's cases: worst deviation 8.3e-6 -> 8.5e-6); the receiver term Hr, common to all edges of a node, stays fp32
// (as q16 it adds coherently: 1.5e-5).
// Non-temporal hints (the `nt` bit of global_load / global_store) on streams that are touched ONCE per launch and are far larger than L2 + Infinity
// Cache: the per-edge table in the segment reduce, the reduce's `agg` store, `h` in node_update (loaded once, stored once per round).  They change no
// value, only what the caches keep: measured at C2 (r05, profiles/r05_nt_hints.txt) the reduce goes 0.245 -> 0.207 ms per launch (4.6 -> 5.4 TB/s),
// +4 % graph-steps/s.  NOT everything read once gains: `nt` on node_update's `agg` loads (rows the reduce has just written) costs that kernel 24 %,
// on the edge encoder's per-edge table stores 4 %, on the Hr / Hs stores 1-2 %; the sc0 / sc1 scope bits make no difference on any of them.
typedef int ag_i32x4 __attribute__((ext_vector_type(4)));
typedef float ag_f32x4 __attribute__((ext_vector_type(4)));
#ifdef AG_NO_NT     // A/B builds (tools/ab_build.sh base="-DAG_NO_NT"): plain loads / stores
__device__ __forceinline__ int4 ag_ld_nt(const int4 *p) { return *p; }
__device__ __forceinline__ float4 ag_ld_nt(const float4 *p) { return *p; }
__device__ __forceinline__ void ag_st_nt(float4 *p, const float4 &v) { *p = v; }
__device__ __forceinline__ void ag_st_nt(int4 *p, const int4 &v) { *p = v; }
#else
__device__ __forceinline__ int4 ag_ld_nt(const int4 *p) { const ag_i32x4 v = __builtin_nontemporal_load(reinterpret_cast<const ag_i32x4 *>(p)); return make_int4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ float4 ag_ld_nt(const float4 *p) { const ag_f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const ag_f32x4 *>(p)); return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void ag_st_nt(float4 *p, const float4 &v) { __builtin_nontemporal_store(ag_f32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<ag_f32x4 *>(p)); }
__device__ __forceinline__ void ag_st_nt(int4 *p, const int4 &v) { __builtin_nontemporal_store(ag_i32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<ag_i32x4 *>(p)); }
#endif

// ---- `agg` as q16 rows (option "agg_q16"): the reduce's lanes already hold a row in the table's own order — lane c of a node's twenty owns the 16-bit
// positions 8c .. 8c + 7 (acc0 = the first four, acc1 = the last four; ag_half_lane_feature), the four lanes of an aligned quad one out-tile.  The
// tile maximum is two quad-permute DPP moves away; the five exponent bytes reach the two lanes whose segments hold them (17: bytes 280..284, 19:
// bytes 312, 316..319 — the Eterm row format above, byte for byte) by four ds_bpermute.  Values are sums of ReLU outputs (>= +0), padding positions are
// already zero (ag_reduce_node_q16's tail).  Because nothing is negative the 16 bits are UNSIGNED here — q = rne(v 2^(126-eb) 65535), value = q 2^(eb-126) / 65535:
// half the rounding step of the signed per-edge rows (the one difference to that format; only node_update reads these rows, ag_mlp.hip: agg_q16_*).
__device__ __forceinline__ float ag_q16u_scale(int eb) { return ldexpf(1.0f / 65535.0f, eb - 126); }
// the eight values of a lane as 16-bit words (no exponent bytes yet) + the tile's exponent byte
__device__ __forceinline__ int4 ag_q16_quantize_segment(const float4 &acc0, const float4 &acc1, int &eb)
{
    typedef unsigned short ag_u16x2 __attribute__((ext_vector_type(2)));
    auto bits = [](float v) { return __float_as_uint(v) & 0x7fffffffu; };
    unsigned m = max(max(max(bits(acc0.x), bits(acc0.y)), max(bits(acc0.z), bits(acc0.w))), max(max(bits(acc1.x), bits(acc1.y)), max(bits(acc1.z), bits(acc1.w))));
    m = max(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0xB1, 0xf, 0xf, false));      // quad_perm [1, 0, 3, 2]
    m = max(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0x4E, 0xf, 0xf, false));      // quad_perm [2, 3, 0, 1]
    eb = (int)(m >> 23);
    eb = eb < AG_Q16_EB_MIN ? AG_Q16_EB_MIN : (eb > AG_Q16_EB_MAX ? AG_Q16_EB_MAX : eb);
    const int inv = 126 - eb;
    auto pack = [&](float x, float y) {
        return (int)__builtin_bit_cast(unsigned, (ag_u16x2)__builtin_amdgcn_cvt_pknorm_u16(__builtin_ldexpf(x, inv), __builtin_ldexpf(y, inv)));
    };
    return make_int4(pack(acc0.x, acc0.y), pack(acc0.z, acc0.w), pack(acc1.x, acc1.y), pack(acc1.z, acc1.w));
}
// eight 16-bit words -> values (node_update's side of the table; the fused reduce's round trip)
__device__ __forceinline__ void ag_q16u_decode8(const int4 &w, float sc, float (&x)[8])
{
    const unsigned a = (unsigned)w.x, b = (unsigned)w.y, c = (unsigned)w.z, d = (unsigned)w.w;
    x[0] = (float)(a & 0xffffu) * sc; x[1] = (float)(a >> 16) * sc; x[2] = (float)(b & 0xffffu) * sc; x[3] = (float)(b >> 16) * sc;
    x[4] = (float)(c & 0xffffu) * sc; x[5] = (float)(c >> 16) * sc; x[6] = (float)(d & 0xffffu) * sc; x[7] = (float)(d >> 16) * sc;
}
// the segment as stored: values + the row's exponent bytes in the two lanes whose segments hold them
__device__ __forceinline__ int4 ag_q16_encode_segment(const float4 &acc0, const float4 &acc1, int c, int group_lane0)
{
    int eb;
    int4 w = ag_q16_quantize_segment(acc0, acc1, eb);
    unsigned ebs = 0;
#pragma unroll
    for (int t = 0; t < 4; ++t) ebs |= (unsigned)__builtin_amdgcn_ds_bpermute((group_lane0 + 4 * t) << 2, eb) << (8 * t);
    if (c == 17) { w.z = (int)ebs; w.w = eb; }              // (this lane is in the fifth tile's quad: `eb` is tile 4's)
    if (c == 19) { w.z = eb; w.w = (int)ebs; }
    return w;
}
// what node_update will read back from the stored segment, without the table (the reduce fused into node_update: same bits as the separate kernels)
__device__ __forceinline__ void ag_q16_roundtrip_segment(float4 &acc0, float4 &acc1)
{
    int eb;
    const int4 w = ag_q16_quantize_segment(acc0, acc1, eb);
    float x[8];
    ag_q16u_decode8(w, ag_q16u_scale(eb), x);
    acc0 = make_float4(x[0], x[1], x[2], x[3]);
    acc1 = make_float4(x[4], x[5], x[6], x[7]);
}

// An elided self-loop (AgFwdArgs::self_info) is a VIRTUAL edge: node g has n = (e1 - e0) + 1 of them, virtual edge j is table row / COO entry
// e0 + j - (j > kself), except j == kself: the class row (table row E + class) and the node itself as the sender.
//
// SELF (compile time: the kernels are instantiated with and without elision — a run-time `self_info ? ... : ...` is a wave-uniform BRANCH around a
// load, and hipcc answers every such join with a vmcnt(0): the four prologue loads of the reduce ran one after the other, 0.208 -> 0.232 ms per launch):
// `E` = number of real edges; class k's row stands AG_SELF_REPL times in the table (rows E + k AG_SELF_REPL + r) and a node reads copy r = node id mod
// AG_SELF_REPL with the same load as any other row — EVERY node reads a class row, and one copy would be 256 k x 20 lanes x 3 rounds on the same three
// cache lines.  (Staging the class rows in LDS — per workgroup behind a barrier, or per wave without one — and selecting them into the self slot
// measured 3-6 % SLOWER than no elision at all: eight v_cndmask per slot in a kernel that is also VALU-heavy; profiles/r06_self_edges.txt.)  The sender
// indices of the SELF variant are loaded UNCONDITIONALLY from a clamped position and selected afterwards (a load under `j < n && j != kself` is sunk
// into a branch and waited for on the spot).
template <int kInFlight = AG_AGG_IN_FLIGHT, bool HSQ = false, bool SELF = false>
__device__ __forceinline__ void ag_reduce_node_q16(const AgFwdArgs &a, int g, int c, int group_lane0, float4 &acc0, float4 &acc1, int E = 0)
{
    const int f0 = ag_half_lane_feature(c);
    const int e0 = a.row_ptr[g], e1 = a.row_ptr[g + 1];
    int n = e1 - e0, kself = 0x7fffffff, eself = 0;
    const int elast = E > 0 ? E - 1 : 0;
    if constexpr (SELF) {
        const int si = a.self_info[g];
        kself = si >= 0 ? (si & 0xffff) : 0x7fffffff;
        n += si >= 0 ? 1 : 0;
        eself = E + (si >= 0 ? si >> 16 : 0) * AG_SELF_REPL + (g & (AG_SELF_REPL - 1));      // this node's copy of its class row
    }
    const int4 *et = reinterpret_cast<const int4 *>(a.eterm) + c;          // segment c of row e: et[e * 20]
    const float *hs = a.hs + f0;
    const int4 *hq = reinterpret_cast<const int4 *>(a.hs) + c;             // HSQ: segment c of sender row s: hq[s * 20]
    const int exp_src = (group_lane0 + (c < 16 ? 17 : 19)) << 2;           // ds_bpermute byte address of the lane that loaded the exponent bytes
    const int exp_shift = c < 16 ? 8 * (c >> 2) : 0;
    const size_t gr = a.hr_row ? (size_t)a.hr_row[g] : (size_t)g;      // (round 0 with node de-duplication: the node's compact row)
    // COO position of virtual edge j (any j: clamped into the array), its table row, and its sender from the raw COO entry: the node itself (in the
    // numbering edge_send uses: its own Hr row index) for the elided self-loop
    auto pos = [&](int j) { return SELF ? min(e0 + j - (j > kself ? 1 : 0), elast) : e0 + j; };
    auto trow = [&](int j) { return (SELF && j == kself) ? eself : pos(j); };
    auto pick = [&](int j, int raw) { return j < n ? ((SELF && j == kself) ? (int)gr : raw) : -1; };
    int s[kInFlight];
    if constexpr (SELF) {
        int raw[kInFlight];
#pragma unroll
        for (int i = 0; i < kInFlight; ++i) raw[i] = a.edge_send[pos(i)];
#pragma unroll
        for (int i = 0; i < kInFlight; ++i) { asm volatile("" : "+v"(raw[i])); s[i] = pick(i, raw[i]); }
    } else {
#pragma unroll
        for (int i = 0; i < kInFlight; ++i) s[i] = e0 + i < e1 ? a.edge_send[e0 + i] : -1;
    }
    const float4 hr0 = *reinterpret_cast<const float4 *>(a.hr + gr * AG_FP + f0);
    const float4 hr1 = *reinterpret_cast<const float4 *>(a.hr + gr * AG_FP + f0 + 8);
    acc0 = make_float4(0.f, 0.f, 0.f, 0.f);
    acc1 = acc0;
    for (int e = 0; e < n; e += kInFlight) {       // n and with it every branch below are uniform over the node's 20 lanes
        int sn[kInFlight];
#pragma unroll
        for (int i = 0; i < kInFlight; ++i) {
            if constexpr (SELF) sn[i] = a.edge_send[pos(e + kInFlight + i)];      // raw; selected at the end of the trip
            else sn[i] = e0 + e + kInFlight + i < e1 ? a.edge_send[e0 + e + kInFlight + i] : -1;
        }
        int4 t[kInFlight], v[HSQ ? kInFlight : 1];
        float4 u0[HSQ ? 1 : kInFlight], u1[HSQ ? 1 : kInFlight];
#pragma unroll
        for (int i = 0; i < kInFlight; ++i)
            if (s[i] >= 0) {
                t[i] = ag_ld_nt(&et[(size_t)trow(e + i) * (AG_FP / 8)]);      // (round 0 included: plain loads of the table the edge encoder has just written measured +2 %)
                if constexpr (HSQ) v[i] = hq[(size_t)s[i] * (AG_FP / 8)];
                else {
                    u0[i] = *reinterpret_cast<const float4 *>(hs + (size_t)s[i] * AG_FP);
                    u1[i] = *reinterpret_cast<const float4 *>(hs + (size_t)s[i] * AG_FP + 8);
                }
            }
#pragma unroll
        for (int i = 0; i < kInFlight; ++i)
            if (s[i] >= 0) {
                const int eb = (__builtin_amdgcn_ds_bpermute(exp_src, t[i].z) >> exp_shift) & 0xff;
                const float sc = ag_q16_scale(eb);
                const float q0 = (float)(short)(t[i].x & 0xffff), q1 = (float)(t[i].x >> 16), q2 = (float)(short)(t[i].y & 0xffff), q3 = (float)(t[i].y >> 16);
                const float q4 = (float)(short)(t[i].z & 0xffff), q5 = (float)(t[i].z >> 16), q6 = (float)(short)(t[i].w & 0xffff), q7 = (float)(t[i].w >> 16);
                if constexpr (HSQ) {
                    const int ebs = (__builtin_amdgcn_ds_bpermute(exp_src, v[i].z) >> exp_shift) & 0xff;
                    const float ss = ag_q16_scale(ebs);
                    const float p0 = (float)(short)(v[i].x & 0xffff), p1 = (float)(v[i].x >> 16), p2 = (float)(short)(v[i].y & 0xffff), p3 = (float)(v[i].y >> 16);
                    const float p4 = (float)(short)(v[i].z & 0xffff), p5 = (float)(v[i].z >> 16), p6 = (float)(short)(v[i].w & 0xffff), p7 = (float)(v[i].w >> 16);
                    acc0.x += fmaxf(fmaf(p0, ss, fmaf(q0, sc, hr0.x)), 0.f); acc0.y += fmaxf(fmaf(p1, ss, fmaf(q1, sc, hr0.y)), 0.f);
                    acc0.z += fmaxf(fmaf(p2, ss, fmaf(q2, sc, hr0.z)), 0.f); acc0.w += fmaxf(fmaf(p3, ss, fmaf(q3, sc, hr0.w)), 0.f);
                    acc1.x += fmaxf(fmaf(p4, ss, fmaf(q4, sc, hr1.x)), 0.f); acc1.y += fmaxf(fmaf(p5, ss, fmaf(q5, sc, hr1.y)), 0.f);
                    acc1.z += fmaxf(fmaf(p6, ss, fmaf(q6, sc, hr1.z)), 0.f); acc1.w += fmaxf(fmaf(p7, ss, fmaf(q7, sc, hr1.w)), 0.f);
                } else {
                    acc0.x += fmaxf(fmaf(q0, sc, hr0.x) + u0[i].x, 0.f); acc0.y += fmaxf(fmaf(q1, sc, hr0.y) + u0[i].y, 0.f);
                    acc0.z += fmaxf(fmaf(q2, sc, hr0.z) + u0[i].z, 0.f); acc0.w += fmaxf(fmaf(q3, sc, hr0.w) + u0[i].w, 0.f);
                    acc1.x += fmaxf(fmaf(q4, sc, hr1.x) + u1[i].x, 0.f); acc1.y += fmaxf(fmaf(q5, sc, hr1.y) + u1[i].y, 0.f);
                    acc1.z += fmaxf(fmaf(q6, sc, hr1.z) + u1[i].z, 0.f); acc1.w += fmaxf(fmaf(q7, sc, hr1.w) + u1[i].w, 0.f);
                }
            }
#pragma unroll
        for (int i = 0; i < kInFlight; ++i) {
            if constexpr (SELF) { asm volatile("" : "+v"(sn[i])); s[i] = pick(e + kInFlight + i, sn[i]); }
            else s[i] = sn[i];
        }
    }
    // the padding positions (features 150..159) decoded the exponent bytes: they are not features
    if (c == 17) acc1 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c == 19) { acc0.z = 0.f; acc0.w = 0.f; acc1 = make_float4(0.f, 0.f, 0.f, 0.f); }
    // non-finite node terms (Hr / Hs) surface here as a non-finite sum: raise the model's sticky status bit (ag_model_status) instead of
    // passing it on silently — the decoder's clamp (model.py:309) would otherwise turn it into a plausible +-100 motion
    if (a.status && !isfinite(((acc0.x + acc0.y) + (acc0.z + acc0.w)) + ((acc1.x + acc1.y) + (acc1.z + acc1.w)))) atomicOr(a.status, 1);
}

// kernel launchers (one translation unit each)
void ag_launch_node_encode(const AgWeights &w, const AgFwdArgs &a, hipStream_t s);
void ag_launch_node_encode_fallback(const AgWeights &w, const AgFwdArgs &a, hipStream_t s);
void ag_launch_send_remap(const AgFwdArgs &a, hipStream_t s);
void ag_launch_edge_encode(const AgWeights &w, const AgFwdArgs &a, hipStream_t s);
void ag_launch_aggregate(const AgFwdArgs &a, hipStream_t s);
void ag_launch_node_update(const AgWeights &w, const AgFwdArgs &a, int last, hipStream_t s);

struct AgEdgeArgs {
    const float *pos;            // (B, N, 3) with pos_stride floats between samples
    size_t pos_stride;
    const uint8_t *mask, *tool;  // (B, N)
    const float *thr_sq;         // (B) squared radius, already rounded per the builder variant
    int topk, connect, variant, B, N, max_tools;
    int cap0, cap;     // slots per row: cap0 = min(N, topk) after top-k; cap = cap0 + max_tools after connect_tools_all
    int32_t *row_ptr, *edge_recv, *edge_send;
    // workspace
    int32_t *sel0;     // (B*N, cap0) top-k/radius senders (local ids), ascending
    int32_t *sel;      // (B*N, cap)  final senders when connect_tools_all (else unused)
    int32_t *deg;      // (B*N)
    int32_t *flag;     // (B) connect_tools_all batch_mask
    int32_t *blk_sum;  // scan partials
    // uniform-grid candidate search (ag_edges.hip: bin_kernel / select_cells_kernel)
    struct GridParamsPOD { float x0, y0, z0, inv; int nx, ny, nz, total; };
    void *grid_raw;        // (B) GridParams
    int32_t *cell_start;   // (B, 8193)
    float4 *sorted;        // (B, N) x, y, z, bits(j | tool << 30) in cell order
    // Riders (ag_rollout only; all NULL elsewhere): two pieces of the MODEL step that depend on nothing but what the builder's own launches read or
    // write, carried by those launches instead of a launch of their own (a 15 us latency chain per model step on a nearly idle chip):
    //  * the per-node input rows of the edge features (AgFwdArgs::edge_node_tab: a function of the state only) by extra workgroups of bin_kernel;
    //  * the sender column mapped to compact rows (AgFwdArgs::send_c, node-encoder de-duplication) by rowptr_scatter_kernel where it writes edge_send.
    const float *tab_state, *tab_attrs, *tab_pinst;   // (B, AG_NHIS, N, 3), (B, N, 2), (B, tab_n_p, tab_n_inst)
    float *tab_out;                                   // (B*N, 16); NULL: no rider
    int tab_n_inst, tab_n_p;
    int *tab_status;                                  // the model's sticky status word (non-finite raw inputs), or NULL
    const int32_t *map_node_row;                      // (B*N) compact row of every node
    const int *map_ovf;                               // device flag: this call runs without de-duplication (identity map)
    int32_t *map_send_c;                              // (E); NULL: no rider
    // Self-edge elision (ag_rollout only; AgFwdArgs::self_info): with `self_attrs` set, the self-loops of nodes of attribute class 0 / 1 are left out of
    // row_ptr / edge_recv / edge_send, their position in the row goes to self_info, and AG_SELF_ROWS synthetic entries (recv = send = self_class_row0 + k:
    // the class rows of the per-node input table) are appended behind the list.
    const float *self_attrs;                          // (B, N, 2); NULL: off
    int32_t *self_info;                               // (B*N) out
    int32_t *self_pos;                                // (B*N) workspace: position of the elidable self-loop in the row, or -1 (scan_partial -> rowptr_scatter)
    int self_class_row0;                              // node-table row of class 0 (= rows_pad of the forward layout)
    // Shared-state rollout (ag_shared.hip): samples whose `active` word is 0 this step — no dirty particle, no tool within the radius of any particle:
    // their graph IS the base sample's — are skipped by every launch of the builder (their rows get degree 0, their per-node input rows are not written).
    const int32_t *active;                            // (B); NULL: every sample
};
enum { AG_RIDER_TAB = 1, AG_RIDER_MAP = 2 };
int ag_launch_build_edges(const AgEdgeArgs &a, hipStream_t s);      // returns the riders its launches carried (AG_RIDER_*)

// Per-node inputs of the edge features (model.py:155-165, 220-253), 64 bytes per node, so that the weight-stationary edge encoder's gather is
// two indexed 64-byte rows per edge: [attr0, attr1, group0, 0 | v0 | v1 | v2 | x_cur], v_i = state[i+1] - state[i].  The per-edge features are
// then plain differences of two rows — the same subtractions in the same order as (pr[i+1] - pr[i]) - (ps[i+1] - ps[i]) in edge_features.
// One thread per node g; shared by edge_node_tab_kernel (ag_mlp.hip) and the rider workgroups of bin_kernel (ag_edges.hip).
// `class_row0` >= 0 (self-edge elision): rows class_row0 + k, k < AG_SELF_ROWS, are the two endpoints of class k's synthetic self-edge — the class's
// attribute pair, everything else zero, so that the edge features come out as a real self-loop's: [a, a, |0 - 0|, 0 - 0 ...] (written by thread k).
__device__ __forceinline__ void ag_edge_node_tab_row(const float *state, const float *attrs, const float *p_instance, int n_inst, int n_p,
                                                     int B, int N, float *tab, int *status, int g, long long class_row0 = -1)
{
    if (class_row0 >= 0 && g < AG_SELF_CLASSES) {
        float4 *dst = reinterpret_cast<float4 *>(tab + (size_t)(class_row0 + g) * 16);
        dst[0] = make_float4(g == 0 ? 1.0f : 0.0f, g == 0 ? 0.0f : 1.0f, 0.0f, 0.0f);
        dst[1] = dst[2] = dst[3] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    if (g >= B * N) return;
    const int b = g / N, i = g - b * N;
    float p[AG_NHIS][3];
#pragma unroll
    for (int hh = 0; hh < AG_NHIS; ++hh)
#pragma unroll
        for (int c = 0; c < 3; ++c) p[hh][c] = state[(((size_t)b * AG_NHIS + hh) * N + i) * 3 + c];
    float o[16];
    o[0] = attrs[(size_t)g * 2]; o[1] = attrs[(size_t)g * 2 + 1];
    o[2] = (n_inst > 0 && i < n_p) ? p_instance[((size_t)b * n_p + i) * n_inst] : 0.0f;
    o[3] = 0.0f;
#pragma unroll
    for (int hh = 0; hh + 1 < AG_NHIS; ++hh)
#pragma unroll
        for (int c = 0; c < 3; ++c) o[4 + hh * 3 + c] = p[hh + 1][c] - p[hh][c];
#pragma unroll
    for (int c = 0; c < 3; ++c) o[4 + (AG_NHIS - 1) * 3 + c] = p[AG_NHIS - 1][c];
    float4 *dst = reinterpret_cast<float4 *>(tab + (size_t)g * 16);
#pragma unroll
    for (int q = 0; q < 4; ++q) dst[q] = make_float4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
    // Non-finite raw inputs raise the status bit HERE: the weight-stationary kernel builds its first-layer operands without the range check the
    // streaming kernel applies to them (h3_pair<true>), and a NaN that reaches a hidden activation with its sign bit set is ReLU'd to 0.
    // (A finite difference beyond fp16's range becomes +-inf there and is caught by the hidden layers' own check.)
    float sum = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; ++k) sum += o[k] * 0.0f;      // 0 for finite rows, NaN otherwise
    if (status && !(sum == 0.0f)) atomicOr(status, 1);
}

struct AgStepArgs {
    float *state;             // (B, H, N, 3) working copy, shifted in place
    const float *delta;       // (B, N, 3) per-step tool motion ("action")
    const float *pred_pos;    // (B, n_p, 3); shared-state rollout: (rows, 3) by compact row, read through cmap
    // shared-state rollout (ag_shared.hip): sample 0 is the base; a particle's prediction is row cmap[b * N + i] of pred_pos (its private row, or the
    // base sample's row i); `dirty` turns on where a private prediction differs from the base's in any bit; samples b >= 1 record into out_seq[b - 1]
    const int32_t *cmap;      // NULL: the plain rollout
    uint8_t *dirty;
    int32_t *sample_dirty;    // (B) set with `dirty`
    const uint8_t *obj_mask;  // (B, n_p) or nullptr; used by height mode 1
    const int32_t *repeat;    // (B) action_repeat
    float *out_seq;           // (B, n_p, 3) recorded when repeat == step
    int B, N, n_p, H, step, height_mode;
    float raise;              // gripper raise (0 when disabled)
};
void ag_launch_rollout_step(const AgStepArgs &a, hipStream_t s);
void ag_launch_zero_words(int32_t *p, int n, hipStream_t s);      // (a kernel, not a memset node: HIP-graph replay)

// ---- shared-state rollout (ag_shared.hip) ------------------------------------------------------------------------------------
// dynamics() rolls ONE cloud out under `bsz` sampled pushes (forward_dynamics.py:11-38; 20 000 samples per planning step in config/planning/rope.yaml):
// wherever a sample's tool has not (yet) had any influence, its particles follow the trajectory of the cloud without a tool — the BASE — bit for bit.
// The engine rolls the base out once, as internal sample 0 (the caller's sample 0 with its tool slots invalid), and per model step computes
// privately only the rows whose result can differ: the 3-hop closure (three propagation rounds) of the rows that differ from the base in an input, an
// edge list or a sender.  Everything per-node that is cheap stays full size (states, edge lists); the heavy kernels run over a COMPACT row set
// [base rows | private rows of sample 1 | sample 2 ...] through index arrays built per step.
struct AgSharedArgs {
    int B1, N, n_p, n_inst, phys_dim, H;      // B1 = caller samples + 1
    // caller inputs (B1 - 1 samples) and their staged copies (B1 samples; sample 0 = the base)
    const float *state0, *delta, *attrs, *p_instance, *phys, *thr_sq;
    const uint8_t *mask, *tool, *obj_mask;
    const int32_t *repeat;
    float *s_state, *s_delta, *s_attrs, *s_pinst, *s_phys, *s_thr;
    uint8_t *s_mask, *s_tool, *s_obj_mask;
    int32_t *s_repeat;
    // flags per internal node
    uint8_t *dirty;                 // sticky: an input, or an earlier prediction, of this node differs from the base's
    uint8_t *sel_a, *sel_b;         // this step's private set while it grows (touched rows -> +1 hop -> +2 hops)
    int32_t *sample_dirty;          // (B1) sticky: some NON-tool node of the sample is dirty
    int32_t *active;                // (B1) this step: the sample can differ from the base at all (sample_dirty, or a tool within the radius of a particle)
    int max_tools;
    // this step's full graph (the edge builder over B1 samples) and the node encoder's compact rows
    const int32_t *row_ptr, *edge_send, *self_info, *node_row;
    // this step's compact graph
    int32_t *cmap;                  // (B1 N) node -> compact row (private row, or the base sample's row of the same particle)
    int32_t *orig;                  // (rows) compact row -> node
    int32_t *row_ptr_c, *self_info_c, *node_row_c;      // (rows [+ 1])
    int32_t *recv_o, *send_o;       // (edges) endpoints as NODES: the edge encoder's gathers of per-node inputs
    int32_t *send_r0, *send_cm;     // (edges) sender as a row of the node encoder's compact tables (round 0) / as a compact row (later rounds)
    int32_t *blk_cnt, *blk_deg;     // scan partials per 256 nodes
    int *n_rows, *n_edges;          // device words: compact rows / edges of this step
    int self_rows, self_class_row0;
};
void ag_launch_shared_stage(const AgSharedArgs &a, hipStream_t s);
void ag_launch_shared_active(const AgSharedArgs &a, hipStream_t s);      // before the step's edge build
void ag_launch_shared_compact(const AgSharedArgs &a, hipStream_t s);     // behind it
void ag_launch_gather_rows(const float *x, const int *idx, float *out, long long E, int D, hipStream_t s);
void ag_launch_segment_sum(const float *vals, const int *ptr, const int *perm, float *out, long long N, int D, hipStream_t s);
void ag_launch_message_fwd(const float *eterm, const float *hr, const float *hs, const int *row_ptr, const int *send, float *agg,
                           long long N, int D, hipStream_t s);
void ag_launch_message_bwd(const float *eterm, const float *hr, const float *hs, const int *row_ptr, const int *send,
                           const float *g_agg, float *g_e, float *g_hr, long long N, int D, hipStream_t s);
// training chains (ag_mlp.hip): kind 0 = relation_encoder + W_rp[:, :F], 1 = particle_encoder, 2 = non_rigid_predictor
struct AgChainArgsPOD { const float *x; const float *w; float *y[4]; const float *dy; float *dz[4]; float *dx; long long rows; int d_in; };
void ag_launch_train_pack(const float *W, const float *bias, int n_out, int n_in, int ld, int col0, int transposed, int compact,
                          int n_tiles, int b3, float *dst, hipStream_t s);
void ag_launch_chain(int kind, int backward, int b3, const AgChainArgsPOD &p, int max_blocks, hipStream_t s);
void ag_launch_edge_inputs_fwd(const float *tab, int D, int A, int G, const int *recv, const int *send, float *out, long long E, hipStream_t s);
void ag_launch_edge_inputs_bwd(const float *tab, int D, int A, int G, const int *recv, const int *send, const int *row_ptr, const int *col_ptr,
                               const int *perm, const float *gout, float *g_r, float *g_s, float *gtab, long long E, long long M, hipStream_t s);
void ag_launch_add3_relu(const float *a, const float *b, const float *c, float *y, long long n, hipStream_t s);
void ag_launch_relu_mask(const float *g, const float *y, float *out, long long n, hipStream_t s);
size_t ag_weight_grads_ws_floats(long long rows, int n_layers);
void ag_launch_weight_grads(int n_layers, const float *const *dz, const int *dz_ld, const float *const *prev, const int *prev_ld, const int *n_in,
                            long long rows, float *partial, float *out, float *const *w_dst, const int *w_ld, float *const *b_dst, const int *n_out,
                            hipStream_t s);
int ag_launch_chamfer(const float *x, const float *y, const unsigned char *xmask, const unsigned char *ymask, int B, int N, int M,
                      int y_batched, float *out, hipStream_t s);
