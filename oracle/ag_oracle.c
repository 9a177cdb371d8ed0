/*
 * ag_oracle.c — CPU restatement of AdaptiGraph's message-passing rollout hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity checker for the HIP engine: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The product path
 * (adaptigraph_amd/) never calls it and has no CPU fallback.
 *
 * Parity status: PINNED.  Every function below is checked against golden vectors produced by
 * running the imported reference in the build container (tools/gen_golden.py -> tests/golden/,
 * tests/test_oracle_golden.py).  The reference has no tests/fixtures of its own.
 *
 * Each function restates one reference symbol, in the reference's formulation (NOT the engine's
 * restructured one): un-split 450/300-wide propagator inputs, O(N^2) pair tests, per-step edge
 * rebuild.  The only representational change is that the one-hot matrices Rr/Rs are carried as the
 * index lists they encode (Rr.bmm(X) == X[recv] exactly, model.py:224-225).
 *
 * Arithmetic: fp32 storage, fp32 k-ordered fma chains for the dense layers; the pair distance is
 * ((dx*dx + dy*dy) + dz*dz) with separately rounded products, which is bit-identical to
 * torch.sum(s_diff ** 2, -1) on CPU (verified at survey time) — build with -ffp-contract=off.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define AGO_OK 0
#define AGO_ERR_CAP (-2)

/* index of the 22 state_dict tensors, in nn.Module registration order (model.py:103-122) */
enum {
    W_PE0, B_PE0, W_PE1, B_PE1, W_PE2, B_PE2,   /* particle_encoder.model.{0,2,4}   */
    W_RE0, B_RE0, W_RE1, B_RE1, W_RE2, B_RE2,   /* relation_encoder.model.{0,2,4}   */
    W_PP, B_PP,                                   /* particle_propagator.linear       */
    W_RP, B_RP,                                   /* relation_propagator.linear       */
    W_D0, B_D0, W_D1, B_D1, W_D2, B_D2,         /* non_rigid_predictor.linear_{0,1,2} */
    AGO_N_TENSORS
};

typedef struct {
    int n_his;       /* dataset_config['n_his'] */
    int attr_dim;    /* model_config['attr_dim'] == rel_attr_dim */
    int phys_dim;    /* number of material params with use: True (model.py:91-94) */
    int action_dim;  /* 3 */
    int nf;          /* nf_particle == nf_relation == nf_effect */
    int pstep;
    int n_instance;  /* p_instance.size(2) */
    float motion_clamp; /* model.py:85 */
} ago_config;

/* y[0..n_out) = act(W x + b [+ res]); W is (n_out, n_in) row-major as nn.Linear stores it.
 * Restates nn.Linear + ReLU (model.py:9-14, 34-40, 57-60) as one fp32 fma chain per output, k ascending. */
static void dense(const float *W, const float *b, const float *x, int n_in, int n_out, const float *res, int relu,
                  float *y)
{
    for (int o = 0; o < n_out; ++o) {
        const float *w = W + (size_t)o * n_in;
        float acc = b[o];
        for (int k = 0; k < n_in; ++k) acc = fmaf(w[k], x[k], acc);
        if (res) acc += res[o];
        y[o] = (relu && acc < 0.0f) ? 0.0f : acc;
    }
}

/* Same layer on a block of rows, vectorised over outputs with a transposed weight copy (used so the
 * cpu_baseline leg is not needlessly slow; identical arithmetic: per-output k-ascending fma chain). */
static void dense_rows(const float *Wt /* (n_in, n_out) */, const float *b, const float *X, int rows, int n_in,
                       int n_out, const float *RES, int relu, float *Y)
{
    for (int r = 0; r < rows; ++r) {
        const float *x = X + (size_t)r * n_in;
        float *y = Y + (size_t)r * n_out;
        for (int o = 0; o < n_out; ++o) y[o] = b[o];
        for (int k = 0; k < n_in; ++k) {
            const float xk = x[k];
            const float *w = Wt + (size_t)k * n_out;
#pragma omp simd
            for (int o = 0; o < n_out; ++o) y[o] = fmaf(w[o], xk, y[o]);
        }
        if (RES) {
            const float *res = RES + (size_t)r * n_out;
            for (int o = 0; o < n_out; ++o) y[o] += res[o];
        }
        if (relu)
            for (int o = 0; o < n_out; ++o) y[o] = y[o] < 0.0f ? 0.0f : y[o];
    }
}

static float *transpose(const float *W, int n_out, int n_in)
{
    float *T = (float *)malloc(sizeof(float) * (size_t)n_out * n_in);
    for (int o = 0; o < n_out; ++o)
        for (int k = 0; k < n_in; ++k) T[(size_t)k * n_out + o] = W[(size_t)o * n_in + k];
    return T;
}

/*
 * DynamicsPredictor.forward — src/dynamics/gnn/model.py:129-313.
 *   state (B,H,N,3)  attrs (B,N,A)  action (B,N,3)  p_instance (B,n_p,I)  phys (B,phys_dim)
 *   recv/send (B,e_stride) int32, first n_rel[b] entries valid  (== rows of Rr/Rs, model.py:143)
 *   out: pred_pos, pred_motion (B,n_p,3)
 * Supports the shipped model_config family (state_dim=offset_dim=density_dim=rel_particle_dim=
 * rel_density_dim=0, rel_group_dim=1, rel_distance_dim=3; config/dynamics/{rope,granular,cloth}.yaml).
 */
int ago_forward(const ago_config *cfg, const float *const *w, const float *state, const float *attrs,
                const float *action, const float *p_instance, const float *phys, const int32_t *recv,
                const int32_t *send, const int32_t *n_rel, int e_stride, int B, int N, int n_p, float *pred_pos,
                float *pred_motion)
{
    const int H = cfg->n_his, A = cfg->attr_dim, Pd = cfg->phys_dim, F = cfg->nf, I = cfg->n_instance;
    const int d_node = A + Pd + cfg->action_dim;      /* model.py:96-101 */
    const int d_edge = 2 * A + 1 + 3 * H;             /* model.py:109-113 */
    const int S = 3 * H;

    float *pe0t = transpose(w[W_PE0], F, d_node), *pe1t = transpose(w[W_PE1], F, F), *pe2t = transpose(w[W_PE2], F, F);
    float *re0t = transpose(w[W_RE0], F, d_edge), *re1t = transpose(w[W_RE1], F, F), *re2t = transpose(w[W_RE2], F, F);
    float *ppt = transpose(w[W_PP], F, 2 * F), *rpt = transpose(w[W_RP], F, 3 * F);
    float *d0t = transpose(w[W_D0], F, F), *d1t = transpose(w[W_D1], F, F);

#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < B; ++b) {
        const int E = n_rel[b];
        const int32_t *rc = recv + (size_t)b * e_stride, *sd = send + (size_t)b * e_stride;
        const float *st = state + (size_t)b * H * N * 3;
        float *snorm = (float *)malloc(sizeof(float) * (size_t)N * S);   /* state_norm_t, model.py:155-165 */
        float *p_in = (float *)malloc(sizeof(float) * (size_t)N * d_node);
        float *t0 = (float *)malloc(sizeof(float) * (size_t)(N > E ? N : E) * F);
        float *t1 = (float *)malloc(sizeof(float) * (size_t)(N > E ? N : E) * F);
        float *penc = (float *)malloc(sizeof(float) * (size_t)N * F);
        float *peff = (float *)malloc(sizeof(float) * (size_t)N * F);
        float *renc = (float *)malloc(sizeof(float) * (size_t)(E ? E : 1) * F);
        float *cat = (float *)malloc(sizeof(float) * (size_t)(N > E ? N : E) * 3 * F);
        float *erel = (float *)malloc(sizeof(float) * (size_t)(E ? E : 1) * F);
        float *agg = (float *)malloc(sizeof(float) * (size_t)N * F);

        for (int n = 0; n < N; ++n) {
            for (int h = 0; h + 1 < H; ++h)
                for (int c = 0; c < 3; ++c)   /* state_res = state[:,1:] - state[:,:-1], model.py:155 */
                    snorm[n * S + h * 3 + c] = st[((size_t)(h + 1) * N + n) * 3 + c] - st[((size_t)h * N + n) * 3 + c];
            for (int c = 0; c < 3; ++c) snorm[n * S + (H - 1) * 3 + c] = st[((size_t)(H - 1) * N + n) * 3 + c];
            /* p_inputs = [attrs | physics_param (0 for shape particles) | action], model.py:168,184-195 */
            float *pi = p_in + (size_t)n * d_node;
            for (int a = 0; a < A; ++a) pi[a] = attrs[((size_t)b * N + n) * A + a];
            for (int q = 0; q < Pd; ++q) pi[A + q] = n < n_p ? phys[(size_t)b * Pd + q] : 0.0f;
            for (int a = 0; a < cfg->action_dim; ++a) pi[A + Pd + a] = action[((size_t)b * N + n) * cfg->action_dim + a];
        }
        /* particle_encode = Encoder(p_inputs), model.py:268 */
        dense_rows(pe0t, w[B_PE0], p_in, N, d_node, F, NULL, 1, t0);
        dense_rows(pe1t, w[B_PE1], t0, N, F, F, NULL, 1, t1);
        dense_rows(pe2t, w[B_PE2], t1, N, F, F, NULL, 1, penc);
        memcpy(peff, penc, sizeof(float) * (size_t)N * F);            /* particle_effect = particle_encode :269 */

        /* rel_inputs = [attrs_r | attrs_s | sum|g_r - g_s| | state_norm_r - state_norm_s], model.py:220-253 */
        float *r_in = (float *)malloc(sizeof(float) * (size_t)(E ? E : 1) * d_edge);
        for (int e = 0; e < E; ++e) {
            const int r = rc[e], s = sd[e];
            float *ri = r_in + (size_t)e * d_edge;
            for (int a = 0; a < A; ++a) ri[a] = attrs[((size_t)b * N + r) * A + a];
            for (int a = 0; a < A; ++a) ri[A + a] = attrs[((size_t)b * N + s) * A + a];
            float gd = 0.0f;
            for (int i = 0; i < I; ++i) {   /* g = cat([p_instance, zeros]) model.py:235 */
                const float gr = r < n_p ? p_instance[((size_t)b * n_p + r) * I + i] : 0.0f;
                const float gs = s < n_p ? p_instance[((size_t)b * n_p + s) * I + i] : 0.0f;
                gd += fabsf(gr - gs);
            }
            ri[2 * A] = gd;
            for (int c = 0; c < S; ++c) ri[2 * A + 1 + c] = snorm[r * S + c] - snorm[s * S + c];
        }
        /* relation_encode = Encoder(rel_inputs), model.py:274 */
        dense_rows(re0t, w[B_RE0], r_in, E, d_edge, F, NULL, 1, t0);
        dense_rows(re1t, w[B_RE1], t0, E, F, F, NULL, 1, t1);
        dense_rows(re2t, w[B_RE2], t1, E, F, F, NULL, 1, renc);

        for (int p = 0; p < cfg->pstep; ++p) {   /* model.py:278-303 */
            for (int e = 0; e < E; ++e) {        /* cat([relation_encode, effect_r, effect_s]) :283-289 */
                memcpy(cat + (size_t)e * 3 * F, renc + (size_t)e * F, sizeof(float) * F);
                memcpy(cat + (size_t)e * 3 * F + F, peff + (size_t)rc[e] * F, sizeof(float) * F);
                memcpy(cat + (size_t)e * 3 * F + 2 * F, peff + (size_t)sd[e] * F, sizeof(float) * F);
            }
            dense_rows(rpt, w[B_RP], cat, E, 3 * F, F, NULL, 1, erel);
            memset(agg, 0, sizeof(float) * (size_t)N * F);  /* effect_rel_agg = Rr_t.bmm(effect_rel) :295 */
            for (int e = 0; e < E; ++e) {
                float *a = agg + (size_t)rc[e] * F;
                const float *m = erel + (size_t)e * F;
                for (int f = 0; f < F; ++f) a[f] += m[f];
            }
            for (int n = 0; n < N; ++n) {        /* cat([particle_encode, effect_rel_agg]) :300 */
                memcpy(cat + (size_t)n * 2 * F, penc + (size_t)n * F, sizeof(float) * F);
                memcpy(cat + (size_t)n * 2 * F + F, agg + (size_t)n * F, sizeof(float) * F);
            }
            dense_rows(ppt, w[B_PP], cat, N, 2 * F, F, peff, 1, t0);   /* residual before ReLU, model.py:36-40 */
            memcpy(peff, t0, sizeof(float) * (size_t)N * F);
        }
        /* non_rigid_predictor on the object slots, clamp + integrate, model.py:306-309 */
        dense_rows(d0t, w[B_D0], peff, n_p, F, F, NULL, 1, t0);
        dense_rows(d1t, w[B_D1], t0, n_p, F, F, NULL, 1, t1);
        for (int n = 0; n < n_p; ++n) {
            float m[3];
            dense(w[W_D2], w[B_D2], t1 + (size_t)n * F, F, 3, NULL, 0, m);
            for (int c = 0; c < 3; ++c) {
                float cl = m[c] > cfg->motion_clamp ? cfg->motion_clamp : (m[c] < -cfg->motion_clamp ? -cfg->motion_clamp : m[c]);
                pred_motion[((size_t)b * n_p + n) * 3 + c] = m[c];
                pred_pos[((size_t)b * n_p + n) * 3 + c] = st[((size_t)(H - 1) * N + n) * 3 + c] + cl;
            }
        }
        free(snorm); free(p_in); free(t0); free(t1); free(penc); free(peff); free(renc); free(cat); free(erel);
        free(agg); free(r_in);
    }
    free(pe0t); free(pe1t); free(pe2t); free(re0t); free(re1t); free(re2t); free(ppt); free(rpt); free(d0t); free(d1t);
    return AGO_OK;
}

/*
 * construct_edges_from_states (variant 0, src/dynamics/dataset/graph.py:38-89) and
 * construct_edges_from_states_batch (variant 1, graph.py:91-156), one call for a batch of samples.
 *   pos (B,N,3)  mask,tool_mask (B,N) uint8  radius (B,) double
 *   out: recv/send (B,e_cap) int32 in the reference's row-major nonzero() order (receiver, then sender),
 *        n_rel (B,).  Returns AGO_ERR_CAP if a sample needs more than e_cap slots.
 * Threshold rounding: single = fp32(double r*r) (graph.py:53,68); batch = fp32 r * fp32 r (graph.py:106-108).
 * Top-k ties (exactly equal fp32 distances) are broken by lower sender index; torch.topk leaves the
 * choice unspecified, so golden generators avoid duplicate points.
 * The row top-k is restated on the in-radius candidates only: every in-radius entry is smaller than every
 * out-of-radius (or 1e10-masked) entry, so "in top-k of the row and in radius" == "among the k nearest
 * in-radius senders".
 */
int ago_build_edges(const float *pos, const uint8_t *mask, const uint8_t *tool_mask, const double *radius, int topk,
                    int connect_tools_all, int variant, int B, int N, int e_cap, int32_t *recv, int32_t *send,
                    int32_t *n_rel)
{
    int status = AGO_OK;
    const int k = N < topk ? N : topk;   /* graph.py:71 / :128 */
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < B; ++b) {
        const float *x = pos + (size_t)b * N * 3;
        const uint8_t *m = mask + (size_t)b * N, *tm = tool_mask + (size_t)b * N;
        float thr;
        if (variant == 0) thr = (float)(radius[b] * radius[b]);
        else { const float r = (float)radius[b]; thr = r * r; }
        uint8_t *adj = (uint8_t *)calloc((size_t)N * N, 1);
        float *dis = (float *)malloc(sizeof(float) * N);
        int *cand = (int *)malloc(sizeof(int) * N);
        for (int i = 0; i < N; ++i) {
            int nc = 0;
            for (int j = 0; j < N; ++j) {
                const float dx = x[i * 3] - x[j * 3], dy = x[i * 3 + 1] - x[j * 3 + 1], dz = x[i * 3 + 2] - x[j * 3 + 2];
                float d = (dx * dx + dy * dy) + dz * dz;           /* graph.py:54-55 / :109-110 */
                if (!(m[i] && m[j])) d = 1e10f;                    /* graph.py:59 / :114 */
                if (tm[i] && tm[j]) d = 1e10f;                     /* graph.py:63 / :118 */
                dis[j] = d;
                if (d - thr < 0.0f) cand[nc++] = j;                /* graph.py:68 / :125 */
            }
            if (nc <= k) {
                for (int c = 0; c < nc; ++c) adj[(size_t)i * N + cand[c]] = 1;
            } else {                                               /* row top-k, graph.py:72-75 / :129-132 */
                for (int c = 0; c < nc; ++c) {
                    const int j = cand[c];
                    int rank = 0;
                    for (int c2 = 0; c2 < nc; ++c2) {
                        const int j2 = cand[c2];
                        rank += (dis[j2] < dis[j]) || (dis[j2] == dis[j] && j2 < j);
                    }
                    if (rank < k) adj[(size_t)i * N + j] = 1;
                }
            }
        }
        if (connect_tools_all) {
            if (variant == 0) {                                    /* graph.py:77-80 */
                for (int i = 0; i < N; ++i)
                    for (int j = 0; j < N; ++j) {
                        uint8_t *a = &adj[(size_t)i * N + j];
                        if (tm[i] && m[j]) *a = 0;                 /* obj_tool_mask_1 */
                        if (tm[j] && m[i]) *a = 1;                 /* obj_tool_mask_2 */
                        if (tm[i] && tm[j]) *a = 0;                /* tool_mask_12 */
                    }
            } else {                                               /* graph.py:134-144 */
                int flag = 0;                                      /* batch_mask: tool receiver with a non-tool sender */
                for (int i = 0; i < N && !flag; ++i)
                    if (tm[i])
                        for (int j = 0; j < N; ++j)
                            if (!tm[j] && adj[(size_t)i * N + j]) { flag = 1; break; }
                for (int i = 0; i < N; ++i)
                    for (int j = 0; j < N; ++j) {
                        uint8_t *a = &adj[(size_t)i * N + j];
                        if (tm[i] && m[j]) *a = 0;                 /* (neg_)batch_obj_tool_mask_1 -> 0 either way */
                        if (tm[j] && m[i]) *a = (uint8_t)flag;     /* batch_obj_tool_mask_2 -> 1 / neg_ -> 0 */
                    }
            }
        }
        int ne = 0, over = 0;
        for (int i = 0; i < N && !over; ++i)                       /* adj_matrix.nonzero(): row-major, graph.py:84 / :151 */
            for (int j = 0; j < N; ++j)
                if (adj[(size_t)i * N + j]) {
                    if (ne >= e_cap) { over = 1; break; }
                    recv[(size_t)b * e_cap + ne] = i;
                    send[(size_t)b * e_cap + ne] = j;
                    ++ne;
                }
        n_rel[b] = ne;
        if (over) {
#pragma omp atomic write
            status = AGO_ERR_CAP;
        }
        free(adj); free(dis); free(cand);
    }
    return status;
}

int ago_num_tensors(void) { return AGO_N_TENSORS; }
