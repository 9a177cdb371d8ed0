// ag_rollout.hip — per-step state update of the batched rollout, entirely on device.
//
// Replaces the tail of the inner loop of dynamics() / dynamics_masked()
// (src/planning/forward_dynamics.py:160-176 / :356-372):
//     pred_state_seq[repeat == ai] = pred_state                       record-on-repeat
//     y_cur   = min_i pred_y            (dynamics, :163)   |  masked mean (dynamics_masked, :359)
//     eef_cur = state[-1, tools] + action[tools]; eef_cur.y = y_cur (+ gripper raise)   (:164-168)
//     state   = cat([state[1:], cat([pred_state, eef_cur])])           history shift     (:170,176)
// A sample is updated by ceil(3N / 2048) workgroups (grid.y), each of which re-derives the sample's tool height from the predicted
// positions itself (a 4-16 KB L2-resident read: cheaper than a second launch or a grid-wide hand-off; with ONE workgroup per sample the
// 64-sample cloth-4k launch used a quarter of the CUs: 0.077 ms); no host synchronisation (the reference syncs via .item() every step).
#include "ag_common.h"

namespace {

constexpr int kStepChunk = 2048;     // plane elements per workgroup

// SHARED (compile time: the plain rollout keeps its direct indexing — the map costs it 25 % of this launch): the shared-state rollout of ag_shared.hip
template <bool SHARED>
__global__ __launch_bounds__(256) void rollout_step_kernel(AgStepArgs a)
{
    __shared__ float red[8];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // prediction of particle i, component c: by (sample, particle), or — shared-state rollout — by compact row through cmap (a private row or the base's)
    const float *pred = a.pred_pos + (SHARED ? (size_t)0 : (size_t)b * a.n_p * 3);
    const int32_t *cm = SHARED ? a.cmap + (size_t)b * a.N : nullptr;
    auto at = [&](int i, int c) { return SHARED ? pred[(size_t)cm[i] * 3 + c] : pred[i * 3 + c]; };

    float y;
    if (a.height_mode == 0) {
        float m = INFINITY;
        for (int i = tid; i < a.n_p; i += 256) m = fminf(m, at(i, 1));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fminf(m, __shfl_xor(m, o));
        if (lane == 0) red[wave] = m;
        __syncthreads();
        y = fminf(fminf(red[0], red[1]), fminf(red[2], red[3]));
    } else {
        const uint8_t *mk = a.obj_mask + (size_t)b * a.n_p;
        float s = 0.f, c = 0.f;
        for (int i = tid; i < a.n_p; i += 256) {
            const float w = mk[i] ? 1.f : 0.f;
            s += at(i, 1) * w;
            c += w;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); c += __shfl_xor(c, o); }
        if (lane == 0) { red[wave] = s; red[4 + wave] = c; }
        __syncthreads();
        y = ((red[0] + red[1]) + (red[2] + red[3])) / ((red[4] + red[5]) + (red[6] + red[7]));
    }
    y += a.raise;

    const int plane = a.N * 3;
    const int k0 = blockIdx.y * kStepChunk, k1 = min(k0 + kStepChunk, plane);
    if (a.repeat[b] == a.step && !(SHARED && b == 0)) {
        float *o = a.out_seq + (size_t)(SHARED ? b - 1 : b) * a.n_p * 3;      // (shared-state rollout: internal sample b is the caller's b - 1; the base records nothing)
        for (int k = k0 + tid; k < min(k1, a.n_p * 3); k += 256) o[k] = SHARED ? at(k / 3, k % 3) : pred[k];
    }
    float *st = a.state + (size_t)b * a.H * a.N * 3;
    const float *dl = a.delta + (size_t)b * a.N * 3;
    // shared-state rollout: a private prediction that differs from the base's in any bit makes the particle dirty from the next step on
    auto mark = [&](int n, int c, float v) {
        if constexpr (SHARED)
            if (b > 0 && cm[n] != n && __float_as_uint(v) != __float_as_uint(pred[(size_t)n * 3 + c])) { a.dirty[(size_t)b * a.N + n] = 1; a.sample_dirty[b] = 1; }
    };
    if (a.H == AG_NHIS) {
        // all of this thread's loads first, then its stores: `st` is read and written, so a plain loop orders every iteration's loads behind the
        // previous iteration's stores — eight dependent memory round trips per thread (17 us per launch at C2; 8 us this way)
        constexpr int kPer = kStepChunk / 256;
        float v[kPer][AG_NHIS], nv[kPer];
#pragma unroll
        for (int i = 0; i < kPer; ++i) {
            const int k = k0 + tid + 256 * i;
            if (k < k1) {
#pragma unroll
                for (int h = 1; h < AG_NHIS; ++h) v[i][h] = st[(size_t)h * plane + k];
                const int n = k / 3, c = k - n * 3;
                nv[i] = n < a.n_p ? (SHARED ? at(n, c) : pred[k]) : (c == 1 ? y : v[i][AG_NHIS - 1] + dl[k]);
                if (n < a.n_p) mark(n, c, nv[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < kPer; ++i) {
            const int k = k0 + tid + 256 * i;
            if (k < k1) {
#pragma unroll
                for (int h = 0; h + 1 < AG_NHIS; ++h) st[(size_t)h * plane + k] = v[i][h + 1];
                st[(size_t)(AG_NHIS - 1) * plane + k] = nv[i];
            }
        }
        return;
    }
    for (int k = k0 + tid; k < k1; k += 256) {
        const int n = k / 3, c = k - n * 3;
        const float last = st[(size_t)(a.H - 1) * plane + k];
        for (int h = 0; h + 1 < a.H; ++h) st[(size_t)h * plane + k] = st[(size_t)(h + 1) * plane + k];
        float nv;
        if (n < a.n_p) { nv = SHARED ? at(n, c) : pred[k]; mark(n, c, nv); }
        else nv = (c == 1) ? y : last + dl[k];
        st[(size_t)(a.H - 1) * plane + k] = nv;
    }
}

}  // namespace

// Zero a few device words with a KERNEL instead of hipMemsetAsync: inside a captured HIP graph the memset NODES of the shared-state rollout were not
// reliably replayed (second replay of tests' graph: garbage; eager calls and the first replay fine), kernel nodes are.
__global__ void zero_words_kernel(int32_t *p, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = 0;
}
void ag_launch_zero_words(int32_t *p, int n, hipStream_t s)
{
    if (n > 0) hipLaunchKernelGGL(zero_words_kernel, dim3((n + 255) / 256), dim3(256), 0, s, p, n);
}

void ag_launch_rollout_step(const AgStepArgs &a, hipStream_t s)
{
    const dim3 grid(a.B, (a.N * 3 + kStepChunk - 1) / kStepChunk);
    if (a.cmap) hipLaunchKernelGGL(rollout_step_kernel<true>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(rollout_step_kernel<false>, grid, dim3(256), 0, s, a);
}
