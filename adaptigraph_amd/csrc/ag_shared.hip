// ag_shared.hip — shared-state rollout: which rows of a model step have to be computed per sample, and the compact graph over them.
//
// The reference's planner calls dynamics(state, action_samples, ...) with ONE state and thousands of sampled pushes
// (src/planning/forward_dynamics.py:11-38, src/config/planning/rope.yaml:39-42, src/planning/real_world/planner.py:246).  Until a sample's tool
// touches a particle — and afterwards outside the light cone of the touched particles, three hops per model step — the sample's particles follow the
// trajectory of the cloud WITHOUT a tool, bit for bit (profiles/r06_shared_state_probe.txt).  Internal sample 0 is that base (the caller's sample 0
// with its tool slots invalid).  Per model step, after the edge builder has produced the full graph of every sample (cheap, and exact by
// construction):
//   touch    a row is PRIVATE if the node is dirty (an input or an earlier prediction differs from the base's in any bit; tool slots always), if its
//            edge list differs from the base's row of the same particle, or if one of its senders is dirty: its first-round message sum can differ
//   hop x 2  ... or if one of its senders is private: rounds 2 and 3 (model.py:277-301: three propagation rounds)
//   scan     compact numbering [base rows | private rows in node order], row_ptr of the compact graph
//   scatter  COO arrays of the compact graph: endpoints as nodes (edge-feature gathers), senders as compact rows (a private row, else the base row of
//            the same particle) and as rows of the node encoder's compact tables (round 0)
// Every row that is not private has the base's edge list, the base's inputs and only senders whose values equal the base's: its result IS the base
// row's, and ag_rollout.hip's state update copies it from there.  Summation orders are untouched, so the rollout equals the plain one bit for bit
// (tests/test_gpu_parity.py: test_shared_state_*).
#include "ag_common.h"

namespace {

__device__ __forceinline__ bool bits_differ(float a, float b) { return __float_as_uint(a) != __float_as_uint(b); }

// Stage the caller's B samples behind the base (internal sample 0 = caller sample 0 without its tools) and mark what differs from the base.
__global__ __launch_bounds__(256) void shared_stage_kernel(AgSharedArgs a)
{
    const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
    if (g >= (long long)a.B1 * a.N) return;
    const int b1 = (int)(g / a.N), i = (int)(g - (long long)b1 * a.N);
    const int sb = b1 == 0 ? 0 : b1 - 1;      // the caller sample this internal sample copies
    const size_t src = (size_t)sb * a.N + i, ref = (size_t)i;
    bool diff = false;
    for (int h = 0; h < a.H; ++h)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = a.state0[(((size_t)sb * a.H + h) * a.N + i) * 3 + c], r = a.state0[((size_t)h * a.N + i) * 3 + c];
            a.s_state[(((size_t)b1 * a.H + h) * a.N + i) * 3 + c] = v;
            diff |= bits_differ(v, r);
        }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v = a.delta[src * 3 + c], r = a.delta[ref * 3 + c];
        a.s_delta[(size_t)g * 3 + c] = v;
        diff |= bits_differ(v, r);
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const float v = a.attrs[src * 2 + c], r = a.attrs[ref * 2 + c];
        a.s_attrs[(size_t)g * 2 + c] = v;
        diff |= bits_differ(v, r);
    }
    if (i < a.n_p) {
        for (int ii = 0; ii < a.n_inst; ++ii) {
            const float v = a.p_instance[((size_t)sb * a.n_p + i) * a.n_inst + ii], r = a.p_instance[(size_t)i * a.n_inst + ii];
            a.s_pinst[((size_t)b1 * a.n_p + i) * a.n_inst + ii] = v;
            diff |= bits_differ(v, r);
        }
        if (a.obj_mask) a.s_obj_mask[(size_t)b1 * a.n_p + i] = a.obj_mask[(size_t)sb * a.n_p + i];
    }
    for (int k = 0; k < a.phys_dim; ++k) diff |= bits_differ(a.phys[(size_t)sb * a.phys_dim + k], a.phys[k]);
    diff |= bits_differ(a.thr_sq[sb], a.thr_sq[0]);
    const bool mk = a.mask[src] != 0, tl = a.tool[src] != 0, mk0 = a.mask[ref] != 0, tl0 = a.tool[ref] != 0;
    const bool base_valid = mk0 && !tl0;      // the base has no tools: their slots are invalid there (no edge touches them)
    if (b1 == 0) {
        a.s_mask[g] = base_valid ? 1 : 0;
        a.s_tool[g] = tl0 ? 1 : 0;
        a.dirty[g] = 0;
    } else {
        a.s_mask[g] = mk ? 1 : 0;
        a.s_tool[g] = tl ? 1 : 0;
        const bool dn = diff || tl || tl0 || mk != base_valid;
        a.dirty[g] = dn ? 1 : 0;
        if (dn && !tl) a.sample_dirty[b1] = 1;      // (zeroed by the launcher)
    }
    if (i == 0) {
        for (int k = 0; k < a.phys_dim; ++k) a.s_phys[(size_t)b1 * a.phys_dim + k] = a.phys[(size_t)sb * a.phys_dim + k];
        a.s_thr[b1] = a.thr_sq[sb];
        a.s_repeat[b1] = b1 == 0 ? 0 : a.repeat[sb];      // (the base records nothing: model steps count from 1)
    }
}

// Can this sample differ from the base at all in this step?  Only through a dirty particle, or through a tool that has a particle within the
// interaction radius — the SAME arithmetic as the edge builder's pair test (separately rounded products, (d - thr) < 0): no pair in radius, no tool
// edge, whatever top-k does.  One workgroup per sample; base: always.
__global__ __launch_bounds__(256) void shared_active_kernel(AgSharedArgs a)
{
    constexpr int kTools = 32;
    __shared__ float tx[kTools], ty[kTools], tz[kTools];
    __shared__ int n_tools;
    const int b = blockIdx.x, N = a.N;
    if (b == 0) { if (threadIdx.x == 0) a.active[0] = 1; return; }
    if (a.sample_dirty[b] != 0) { if (threadIdx.x == 0) a.active[b] = 1; return; }      // (uniform)
    if (threadIdx.x == 0) n_tools = 0;
    __syncthreads();
    const float *pos = a.s_state + (((size_t)b * a.H + (a.H - 1)) * N) * 3;
    const uint8_t *mk = a.s_mask + (size_t)b * N, *tl = a.s_tool + (size_t)b * N;
    for (int i = threadIdx.x; i < N; i += 256)
        if (tl[i] && mk[i]) {
            const int k = atomicAdd(&n_tools, 1);
            if (k < kTools) { tx[k] = pos[i * 3]; ty[k] = pos[i * 3 + 1]; tz[k] = pos[i * 3 + 2]; }
        }
    __syncthreads();
    const int nt = n_tools;
    bool any = nt > kTools;      // (more tools than the list holds: treat the sample as active)
    const float thr = a.s_thr[b];
    if (!any)
        for (int i = threadIdx.x; i < N; i += 256) {
            if (!mk[i] || tl[i]) continue;
            const float x = pos[i * 3], y = pos[i * 3 + 1], z = pos[i * 3 + 2];
            for (int k = 0; k < nt; ++k) {
                const float dx = x - tx[k], dy = y - ty[k], dz = z - tz[k];
                const float d = (dx * dx + dy * dy) + dz * dz;
                any = any || (d - thr) < 0.0f;
            }
        }
    const int r = __syncthreads_or(any ? 1 : 0);
    if (threadIdx.x == 0) a.active[b] = r ? 1 : 0;
}

// first-round set: dirty nodes, rows whose edge list is not the base's, rows with a dirty sender
__global__ __launch_bounds__(256) void shared_touch_kernel(AgSharedArgs a)
{
    const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
    if (g >= (long long)a.B1 * a.N) return;
    const int b1 = (int)(g / a.N), i = (int)(g - (long long)b1 * a.N);
    if (!a.active[b1]) { a.sel_a[g] = 0; return; }      // (its rows were not even built: they are the base's)
    bool sel = b1 == 0 || a.dirty[g] != 0;
    if (!sel) {
        const int e0 = a.row_ptr[g], e1 = a.row_ptr[g + 1], f0 = a.row_ptr[i], f1 = a.row_ptr[i + 1];
        sel = (e1 - e0 != f1 - f0) || (a.self_info && a.self_info[g] != a.self_info[i]);
        const int off = b1 * a.N;
        for (int k = 0; !sel && k < e1 - e0; ++k) {
            const int s = a.edge_send[e0 + k];
            sel = (s - off != a.edge_send[f0 + k]) || a.dirty[s] != 0;
        }
    }
    a.sel_a[g] = sel ? 1 : 0;
}

// one more propagation round: rows with a private sender
__global__ __launch_bounds__(256) void shared_hop_kernel(AgSharedArgs a, const uint8_t *in, uint8_t *out)
{
    const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
    if (g >= (long long)a.B1 * a.N) return;
    if (!a.active[g / a.N]) { out[g] = 0; return; }
    bool sel = in[g] != 0;
    if (!sel) {
        const int e0 = a.row_ptr[g], e1 = a.row_ptr[g + 1];
        for (int k = e0; !sel && k < e1; ++k) sel = in[a.edge_send[k]] != 0;
    }
    out[g] = sel ? 1 : 0;
}

__device__ int2 block_exclusive_scan2(int2 v, int2 *total)
{
    __shared__ int2 wsum[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int2 x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int y0 = __shfl_up(x.x, o), y1 = __shfl_up(x.y, o);
        if (lane >= o) { x.x += y0; x.y += y1; }
    }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    int2 base = make_int2(0, 0);
    for (int w = 0; w < wave; ++w) { base.x += wsum[w].x; base.y += wsum[w].y; }
    *total = make_int2(wsum[0].x + wsum[1].x + wsum[2].x + wsum[3].x, wsum[0].y + wsum[1].y + wsum[2].y + wsum[3].y);
    __syncthreads();
    return make_int2(base.x + x.x - v.x, base.y + x.y - v.y);
}

__global__ __launch_bounds__(256) void shared_scan_partial_kernel(AgSharedArgs a, const uint8_t *sel)
{
    const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
    const bool on = g < (long long)a.B1 * a.N && sel[g] != 0;
    int2 total;
    block_exclusive_scan2(make_int2(on ? 1 : 0, on ? a.row_ptr[g + 1] - a.row_ptr[g] : 0), &total);
    if (threadIdx.x == 0) { a.blk_cnt[blockIdx.x] = total.x; a.blk_deg[blockIdx.x] = total.y; }
}

__global__ __launch_bounds__(256) void shared_offsets_kernel(AgSharedArgs a, const uint8_t *sel)
{
    int2 part = make_int2(0, 0);
    for (int i = threadIdx.x; i < (int)blockIdx.x; i += 256) { part.x += a.blk_cnt[i]; part.y += a.blk_deg[i]; }
    int2 base;
    block_exclusive_scan2(part, &base);      // (the TOTAL of the partial sums in front of this block)
    const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
    const bool in = g < (long long)a.B1 * a.N;
    const bool on = in && sel[g] != 0;
    const int e0 = in ? a.row_ptr[g] : 0, d = on ? a.row_ptr[g + 1] - e0 : 0;
    int2 total;
    const int2 ex = block_exclusive_scan2(make_int2(on ? 1 : 0, d), &total);
    if (in) {
        if (on) {
            const int c = base.x + ex.x;
            a.cmap[g] = c;
            a.orig[c] = (int)g;
            a.row_ptr_c[c] = base.y + ex.y;
            a.node_row_c[c] = a.node_row[g];
            a.self_info_c[c] = a.self_info ? a.self_info[g] : -1;
        } else a.cmap[g] = (int)(g % a.N);      // the base sample's row of the same particle
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        const int rows = base.x + total.x, edges = base.y + total.y;
        *a.n_rows = rows;
        *a.n_edges = edges;
        a.row_ptr_c[rows] = edges;
        for (int k = 0; k < a.self_rows; ++k) a.recv_o[edges + k] = a.send_o[edges + k] = a.self_class_row0 + k / AG_SELF_REPL;      // the class rows' synthetic edges
    }
}

// COO arrays of the compact graph: eight lanes per compact row
__global__ __launch_bounds__(256) void shared_scatter_kernel(AgSharedArgs a)
{
    const int rows = *a.n_rows;
    const int sub = threadIdx.x & 7;
    for (int c = blockIdx.x * 32 + (threadIdx.x >> 3); c < rows; c += gridDim.x * 32) {
        const int g = a.orig[c];
        const int e0 = a.row_ptr[g], d = a.row_ptr[g + 1] - e0, o = a.row_ptr_c[c];
        for (int k = sub; k < d; k += 8) {
            const int s = a.edge_send[e0 + k];
            a.recv_o[o + k] = g;
            a.send_o[o + k] = s;
            a.send_cm[o + k] = a.cmap[s];
            a.send_r0[o + k] = a.node_row[s];
        }
    }
}

}  // namespace

void ag_launch_shared_stage(const AgSharedArgs &a, hipStream_t s)
{
    const long long n = (long long)a.B1 * a.N;
    ag_launch_zero_words(a.sample_dirty, a.B1, s);
    hipLaunchKernelGGL(shared_stage_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a);
}

void ag_launch_shared_active(const AgSharedArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(shared_active_kernel, dim3(a.B1), dim3(256), 0, s, a);
}

void ag_launch_shared_compact(const AgSharedArgs &a, hipStream_t s)
{
    const long long n = (long long)a.B1 * a.N;
    const unsigned nb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(shared_touch_kernel, dim3(nb), dim3(256), 0, s, a);
    hipLaunchKernelGGL(shared_hop_kernel, dim3(nb), dim3(256), 0, s, a, (const uint8_t *)a.sel_a, a.sel_b);
    hipLaunchKernelGGL(shared_hop_kernel, dim3(nb), dim3(256), 0, s, a, (const uint8_t *)a.sel_b, a.sel_a);
    hipLaunchKernelGGL(shared_scan_partial_kernel, dim3(nb), dim3(256), 0, s, a, (const uint8_t *)a.sel_a);
    hipLaunchKernelGGL(shared_offsets_kernel, dim3(nb), dim3(256), 0, s, a, (const uint8_t *)a.sel_a);
    const unsigned ns = nb * 8 < 4096 ? nb * 8 : 4096;
    hipLaunchKernelGGL(shared_scatter_kernel, dim3(ns > 0 ? ns : 1), dim3(256), 0, s, a);
}
