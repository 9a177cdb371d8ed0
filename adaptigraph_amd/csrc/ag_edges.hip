// ag_edges.hip — radius-graph + per-receiver top-k adjacency, emitted as receiver-sorted CSR/COO.
//
// Replaces construct_edges_from_states (variant 0, src/dynamics/dataset/graph.py:38-89) and
// construct_edges_from_states_batch (variant 1, graph.py:91-156), without the dense (B,N,N,3) difference
// tensor, the (B,N,N) distance/top-k matrices or the one-hot (B,E,N) outputs, and with no host sync.
//
// Exactness contract (bit-for-bit edge sets; a single flipped edge moves the outputs by O(1e-2)):
//   d_ij  = ((dx*dx + dy*dy) + dz*dz) in fp32 with separately rounded products (this file is compiled with
//           -ffp-contract=off) — bit-identical to torch.sum(s_diff ** 2, -1) on the reference's CPU path
//   masks : d = 1e10 unless mask[i] && mask[j]; d = 1e10 for tool-tool pairs            (graph.py:114,118)
//   in radius  <=>  (d - thr) < 0                                                         (graph.py:125)
//   row top-k : the k = min(N, topk) smallest d of the row; restricted to in-radius candidates this is the
//           k nearest in-radius senders (every in-radius entry is smaller than every other entry);
//           exact-distance ties go to the lower sender index (torch.topk leaves them unspecified)
//   connect_tools_all : the two variants' override rules, including the batch variant's per-sample
//           batch_mask and the tool->tool edges it keeps (graph.py:77-80 vs :134-144)
//   order : (receiver, sender) ascending == adj_matrix.nonzero() row-major order (graph.py:151)
#include "ag_common.h"
#include <cstdlib>

namespace {

constexpr int kRowsPerBlock = 16;
constexpr int kCand = 384;     // per-wave candidate list capacity; pruned to top-k whenever it could overflow

__device__ __forceinline__ void wave_lds_fence()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// keep only the k best (d, j)-ordered candidates, preserving ascending-j order; returns new count
__device__ int prune_topk(float *cd, int *cj, int cnt, int k, int lane)
{
    unsigned keepbits = 0;
    for (int base = 0, ch = 0; base < cnt; base += 64, ++ch) {
        const int idx = base + lane;
        const bool have = idx < cnt;
        const float d = have ? cd[idx] : 0.f;
        const int jj = have ? cj[idx] : 0;
        int rank = 0;
        for (int t = 0; t < cnt; ++t) {
            const float dt = cd[t];
            const int jt = cj[t];
            rank += (dt < d) || (dt == d && jt < jj);
        }
        if (have && rank < k) keepbits |= 1u << ch;
    }
    wave_lds_fence();
    int newcnt = 0;
    for (int base = 0, ch = 0; base < cnt; base += 64, ++ch) {
        const int idx = base + lane;
        const bool have = idx < cnt;
        const float d = have ? cd[idx] : 0.f;
        const int jj = have ? cj[idx] : 0;
        const bool keep = (keepbits >> ch) & 1u;
        const unsigned long long bal = __ballot(keep);
        wave_lds_fence();   // all lanes hold their (d, j) before anyone overwrites a slot
        if (keep) {
            const int pos = newcnt + __popcll(bal & ((1ull << lane) - 1ull));
            cd[pos] = d;
            cj[pos] = jj;
        }
        newcnt += __popcll(bal);
        wave_lds_fence();
    }
    return newcnt;
}

__global__ __launch_bounds__(256) void select_kernel(AgEdgeArgs a)
{
    // dynamic LDS: the sample's particle table staged once per block (SoA x|y|z + a 2-bit flag per particle),
    // then per-wave candidate lists.  Every receiver row of the block re-reads it N/64 times.
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = blockIdx.y, N = a.N;
    if (a.active && !a.active[b]) return;      // (shared-state rollout: this sample's graph is the base sample's)
    const int Np = (N + 63) & ~63;
    float *sx = reinterpret_cast<float *>(smem), *sy = sx + Np, *sz = sy + Np;
    unsigned char *sf = reinterpret_cast<unsigned char *>(sz + Np);          // bit0 = mask, bit1 = tool
    float *s_d = reinterpret_cast<float *>(sf + Np);
    int *s_j = reinterpret_cast<int *>(s_d + 4 * kCand);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *cd = s_d + wave * kCand;
    int *cj = s_j + wave * kCand;
    const float *pos = a.pos + (size_t)b * a.pos_stride;
    const uint8_t *mk = a.mask + (size_t)b * N, *tl = a.tool + (size_t)b * N;
    for (int j = threadIdx.x; j < Np; j += 256) {
        const bool in = j < N;
        sx[j] = in ? pos[j * 3] : 0.f;
        sy[j] = in ? pos[j * 3 + 1] : 0.f;
        sz[j] = in ? pos[j * 3 + 2] : 0.f;
        sf[j] = in ? (unsigned char)((mk[j] ? 1 : 0) | (tl[j] ? 2 : 0)) : 0;   // padding: invalid sender
    }
    __syncthreads();
    const float thr = a.thr_sq[b];
    const int k = N < a.topk ? N : a.topk;

    for (int rr = wave; rr < kRowsPerBlock; rr += 4) {
        const int i = blockIdx.x * kRowsPerBlock + rr;
        if (i >= N) break;   // wave-uniform
        const float xi = sx[i], yi = sy[i], zi = sz[i];
        const bool mi = sf[i] & 1, ti = sf[i] & 2;
        int cnt = 0;
        for (int j0 = 0; j0 < Np; j0 += 64) {
            const int j = j0 + lane;
            const unsigned fj = sf[j];
            const float dx = xi - sx[j], dy = yi - sy[j], dz = zi - sz[j];
            float d = (dx * dx + dy * dy) + dz * dz;
            if (!(mi && (fj & 1))) d = 1e10f;          // also kills the j >= N padding (flag 0)
            if (ti && (fj & 2)) d = 1e10f;
            const bool c = (d - thr) < 0.0f;
            const unsigned long long bal = __ballot(c);
            if (bal) {
                if (cnt + 64 > kCand) {   // wave-uniform; keeps the append below in bounds
                    cnt = prune_topk(cd, cj, cnt, k, lane);
                }
                if (c) {
                    const int p = cnt + __popcll(bal & ((1ull << lane) - 1ull));
                    cd[p] = d;
                    cj[p] = j;
                }
                cnt += __popcll(bal);
                wave_lds_fence();
            }
        }
        if (cnt > k) cnt = prune_topk(cd, cj, cnt, k, lane);
        const size_t row = (size_t)b * N + i;
        const int jsel = lane < cnt ? cj[lane] : -1;
        if (lane < cnt) a.sel0[row * a.cap0 + lane] = jsel;
        if (lane == 0) a.deg[row] = cnt;       // provisional when connect_tools_all: finalize_connect_kernel rewrites it
        if (a.connect) {
            // batch_mask (graph.py:123,135): some tool receiver keeps an edge from a non-tool sender
            const bool hit = ti && lane < cnt && !(sf[jsel < 0 ? 0 : jsel] & 2);
            if (a.variant == 1 && __ballot(hit) && lane == 0) atomicOr(&a.flag[b], 1);
        }
        wave_lds_fence();
    }
}

__device__ int block_exclusive_scan(int v, int *total)
{
    __shared__ int wsum[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int y = __shfl_up(x, o);
        if (lane >= o) x += y;
    }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += wsum[w];
    *total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
    return base + x - v;
}

// ---- uniform-grid ("cell list") candidate search -------------------------------------------------------------
// For N in the thousands the O(N^2) scan above dominates the whole step (cloth-4k: 16.8 M pair tests per graph).
// bin_kernel sorts the valid particles of a sample into cubic cells of edge >= radius * 1.002 (one workgroup per
// sample, counting sort in LDS, x fastest so the 3 x-neighbour cells are one contiguous range); select_cells_kernel
// then tests only the <= 9 ranges around the receiver.  The candidate SET is exactly the in-radius set of the brute
// force scan (any pair with fp32 d < thr lies in adjacent cells: |dx|/cs <= 0.998 plus < 1e-3 of fp32 rounding in the
// cell coordinate), the distance arithmetic and the (d, j) ranking are the same code, and the <= k survivors are
// re-sorted by sender index, so the emitted edge lists are bit-identical (tests compare both paths to the oracle).
constexpr int kCellMax = 8192;

struct GridParams { float x0, y0, z0, inv; int nx, ny, nz, total; };

__device__ __forceinline__ int cell_coord(float x, float x0, float inv, int n)
{
    const int c = (int)((x - x0) * inv);
    return c < 0 ? 0 : (c >= n ? n - 1 : c);
}

__global__ __launch_bounds__(256) void bin_kernel(AgEdgeArgs a)
{
    __shared__ int cnt[kCellMax];
    __shared__ float red[6][4];
    __shared__ GridParams G;
    if ((int)blockIdx.x >= a.B) {       // rider workgroups (ag_rollout): the model step's per-node input rows of the edge features, a function of the state only
        const int gt = ((int)blockIdx.x - a.B) * 256 + (int)threadIdx.x;
        if (a.active && gt < a.B * a.N && !a.active[gt / a.N]) return;      // (no private edge names a node of an inactive sample)
        ag_edge_node_tab_row(a.tab_state, a.tab_attrs, a.tab_pinst, a.tab_n_inst, a.tab_n_p, a.B, a.N, a.tab_out, a.tab_status, gt,
                             a.self_attrs ? (long long)a.self_class_row0 : -1);
        return;
    }
    if (a.active && !a.active[blockIdx.x]) return;
    const int b = blockIdx.x, N = a.N, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *pos = a.pos + (size_t)b * a.pos_stride;
    const uint8_t *mk = a.mask + (size_t)b * N, *tl = a.tool + (size_t)b * N;
    // The sample's particles are read ONCE, all loads of a thread in flight together (up to kBinK per thread: N <= 4 352), and the three passes below
    // run from registers; a plain `if (mk[j]) ... pos[j]` loop per pass is a chain of dependent round trips (r05: 16 -> 8 us at 256 x 1 001).
    constexpr int kBinK = 17;
    const bool cached = N <= 256 * kBinK;
    float px[kBinK], py[kBinK], pz[kBinK];
    unsigned vbits = 0, tbits = 0;
    if (cached) {
        unsigned char mv[kBinK], tv[kBinK];
#pragma unroll
        for (int i = 0; i < kBinK; ++i) {
            const int j = tid + 256 * i, jc = j < N ? j : 0;
            mv[i] = mk[jc]; tv[i] = tl[jc];
            px[i] = pos[jc * 3]; py[i] = pos[jc * 3 + 1]; pz[i] = pos[jc * 3 + 2];
        }
#pragma unroll
        for (int i = 0; i < kBinK; ++i) {
            if (tid + 256 * i < N && mv[i]) vbits |= 1u << i;
            if (tv[i]) tbits |= 1u << i;
        }
    }
    // f(j, x, y, z, is_tool) for every valid particle of this thread
    auto for_valid = [&](auto &&f) {
        if (cached) {
#pragma unroll
            for (int i = 0; i < kBinK; ++i)
                if (vbits >> i & 1u) f(tid + 256 * i, px[i], py[i], pz[i], (tbits >> i & 1u) != 0);
        } else {
            for (int j = tid; j < N; j += 256)
                if (mk[j]) f(j, pos[j * 3], pos[j * 3 + 1], pos[j * 3 + 2], tl[j] != 0);
        }
    };
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for_valid([&](int, float x, float y, float z, bool) {
        lo[0] = fminf(lo[0], x); hi[0] = fmaxf(hi[0], x);
        lo[1] = fminf(lo[1], y); hi[1] = fmaxf(hi[1], y);
        lo[2] = fminf(lo[2], z); hi[2] = fmaxf(hi[2], z);
    });
    for (int c = 0; c < 3; ++c) {
        for (int o = 32; o > 0; o >>= 1) { lo[c] = fminf(lo[c], __shfl_xor(lo[c], o)); hi[c] = fmaxf(hi[c], __shfl_xor(hi[c], o)); }
        if (lane == 0) { red[c][wave] = lo[c]; red[3 + c][wave] = hi[c]; }
    }
    __syncthreads();
    if (tid == 0) {
        float mn[3], mx[3];
        for (int c = 0; c < 3; ++c) {
            mn[c] = fminf(fminf(red[c][0], red[c][1]), fminf(red[c][2], red[c][3]));
            mx[c] = fmaxf(fmaxf(red[3 + c][0], red[3 + c][1]), fmaxf(red[3 + c][2], red[3 + c][3]));
            if (!(mx[c] >= mn[c])) { mn[c] = 0.f; mx[c] = 0.f; }     // no valid particle
        }
        float cs = sqrtf(a.thr_sq[b]) * 1.002f;
        if (!(cs > 1e-20f)) cs = 1e-20f;
        int nx = 1, ny = 1, nz = 1;
        // coarsen until the grid fits the LDS histogram (larger cells stay exact, only less selective).  Non-finite
        // extents (an Inf coordinate of a diverged rollout; NaNs never enter the min/max) or a cell size that overflows
        // fall back to ONE cell: every valid particle is then a candidate of every receiver and the pair test itself
        // decides, exactly like the reference, which yields no edge for a NaN/Inf distance (graph.py:121: `dis < thresh`).
        bool ok = true;
        for (int c = 0; c < 3; ++c) ok = ok && isfinite(mn[c]) && isfinite(mx[c]) && isfinite(mx[c] - mn[c]);
        for (int it = 0; ok; ++it) {
            const float fx = (mx[0] - mn[0]) / cs, fy = (mx[1] - mn[1]) / cs, fz = (mx[2] - mn[2]) / cs;
            if (fx < 8000.f && fy < 8000.f && fz < 8000.f) {
                nx = (int)fx + 1; ny = (int)fy + 1; nz = (int)fz + 1;
                if ((long long)nx * ny * nz <= kCellMax) break;
            }
            cs *= 2.0f;
            if (it > 300 || !isfinite(cs)) ok = false;
        }
        if (!ok) { nx = ny = nz = 1; cs = 1.0f; for (int c = 0; c < 3; ++c) if (!isfinite(mn[c])) mn[c] = 0.f; }
        G.x0 = mn[0]; G.y0 = mn[1]; G.z0 = mn[2]; G.inv = 1.0f / cs; G.nx = nx; G.ny = ny; G.nz = nz; G.total = 0;
    }
    __syncthreads();
    if (a.connect && tid == 0) a.flag[b] = 0;        // batch_mask of this sample (set by the selection kernels, read by finalize_connect)
    const GridParams g = G;
    const int ncell = g.nx * g.ny * g.nz;
    for (int c = tid; c < ncell; c += 256) cnt[c] = 0;
    __syncthreads();
    for_valid([&](int, float x, float y, float z, bool) {
        const int cid = (cell_coord(z, g.z0, g.inv, g.nz) * g.ny + cell_coord(y, g.y0, g.inv, g.ny)) * g.nx + cell_coord(x, g.x0, g.inv, g.nx);
        atomicAdd(&cnt[cid], 1);
    });
    __syncthreads();
    // exclusive scan of cnt[0..ncell): thread t owns cells [t*32, t*32+32)
    int local = 0;
    for (int c = tid * 32; c < tid * 32 + 32 && c < ncell; ++c) local += cnt[c];
    int total;
    int run = block_exclusive_scan(local, &total);
    int32_t *cs_out = a.cell_start + (size_t)b * (kCellMax + 1);
    for (int c = tid * 32; c < tid * 32 + 32 && c < ncell; ++c) {
        const int v = cnt[c];
        cnt[c] = run;            // becomes the running insertion offset
        cs_out[c] = run;
        run += v;
    }
    if (tid == 0) { cs_out[ncell] = total; GridParams out = g; out.total = total; reinterpret_cast<GridParams *>(a.grid_raw)[b] = out; }
    __syncthreads();
    float4 *sorted = a.sorted + (size_t)b * N;
    for_valid([&](int j, float x, float y, float z, bool tool) {
        const int cid = (cell_coord(z, g.z0, g.inv, g.nz) * g.ny + cell_coord(y, g.y0, g.inv, g.ny)) * g.nx + cell_coord(x, g.x0, g.inv, g.nx);
        const int p = atomicAdd(&cnt[cid], 1);
        sorted[p] = make_float4(x, y, z, __int_as_float(j | (tool ? 0x40000000 : 0)));
    });
}

__global__ __launch_bounds__(256) void select_cells_kernel(AgEdgeArgs a)
{
    __shared__ float s_d[4][kCand];
    __shared__ int s_j[4][kCand];
    const int b = blockIdx.y, N = a.N;
    if (a.active && !a.active[b]) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *cd = s_d[wave];
    int *cj = s_j[wave];
    const float *pos = a.pos + (size_t)b * a.pos_stride;
    const uint8_t *mk = a.mask + (size_t)b * N, *tl = a.tool + (size_t)b * N;
    const GridParams g = reinterpret_cast<const GridParams *>(a.grid_raw)[b];
    const int32_t *cstart = a.cell_start + (size_t)b * (kCellMax + 1);
    const float4 *sorted = a.sorted + (size_t)b * N;
    const float thr = a.thr_sq[b];
    const int k = N < a.topk ? N : a.topk;

    for (int rr = wave; rr < kRowsPerBlock; rr += 4) {
        const int i = blockIdx.x * kRowsPerBlock + rr;
        if (i >= N) break;   // wave-uniform
        const size_t row = (size_t)b * N + i;
        const bool mi = mk[i], ti = tl[i];
        int cnt = 0;
        if (mi) {
            const float xi = pos[i * 3], yi = pos[i * 3 + 1], zi = pos[i * 3 + 2];
            const int ix = cell_coord(xi, g.x0, g.inv, g.nx), iy = cell_coord(yi, g.y0, g.inv, g.ny), iz = cell_coord(zi, g.z0, g.inv, g.nz);
            const int x_lo = ix > 0 ? ix - 1 : 0, x_hi = ix + 1 < g.nx ? ix + 1 : g.nx - 1;
            // lanes 0..8 fetch the (dz, dy) neighbour ranges in ONE round trip; the concatenation of the ranges is
            // then swept 64 candidates at a time (a serial loop over the 9 ranges is a chain of dependent loads)
            int len = 0, p0 = 0;
            if (lane < 9) {
                const int z2 = iz + lane / 3 - 1, y2 = iy + lane % 3 - 1;
                if (z2 >= 0 && z2 < g.nz && y2 >= 0 && y2 < g.ny) {
                    const int base = (z2 * g.ny + y2) * g.nx;
                    p0 = cstart[base + x_lo];
                    len = cstart[base + x_hi + 1] - p0;
                }
            }
            int incl = len;
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) {
                const int y = __shfl_up(incl, o);
                if (lane >= o) incl += y;
            }
            const int total = __shfl(incl, 8);
            const int excl = incl - len;
            for (int tb = 0; tb < total; tb += 64) {
                const int t = tb + lane;
                int p = -1;
#pragma unroll
                for (int r = 0; r < 9; ++r) {
                    const int e_r = __shfl(excl, r), n_r = __shfl(len, r), p_r = __shfl(p0, r);
                    if (t >= e_r && t < e_r + n_r) p = p_r + (t - e_r);
                }
                bool c = false;
                float d = 0.f;
                int j = 0;
                if (p >= 0) {
                    const float4 sj = sorted[p];
                    const int w = __float_as_int(sj.w);
                    j = w & 0x3fffffff;
                    const float dx = xi - sj.x, dy2 = yi - sj.y, dz2 = zi - sj.z;
                    d = (dx * dx + dy2 * dy2) + dz2 * dz2;
                    if (ti && (w & 0x40000000)) d = 1e10f;   // tool-tool (graph.py:118); invalid senders are not binned
                    c = (d - thr) < 0.0f;
                }
                const unsigned long long bal = __ballot(c);
                if (bal) {
                    if (cnt + 64 > kCand) cnt = prune_topk(cd, cj, cnt, k, lane);
                    if (c) {
                        const int q = cnt + __popcll(bal & ((1ull << lane) - 1ull));
                        cd[q] = d;
                        cj[q] = j;
                    }
                    cnt += __popcll(bal);
                    wave_lds_fence();
                }
            }
            if (cnt > k) cnt = prune_topk(cd, cj, cnt, k, lane);
        }
        // the survivors arrive in cell order: emit them in ascending sender index (k <= 64: one per lane)
        const int jsel = lane < cnt ? cj[lane] : 0x7fffffff;
        int rank = 0;
        for (int t = 0; t < cnt; ++t) rank += __shfl(jsel, t) < jsel;
        if (lane < cnt) a.sel0[row * a.cap0 + rank] = jsel;
        if (lane == 0) a.deg[row] = cnt;
        if (a.connect) {
            const bool hit = ti && lane < cnt && !tl[jsel == 0x7fffffff ? 0 : jsel];
            if (a.variant == 1 && __ballot(hit) && lane == 0) atomicOr(&a.flag[b], 1);
        }
        wave_lds_fence();
    }
}

// Lane-per-receiver variant of the cell search for the shipped top-k values: every lane walks the <= 9 cell ranges of
// its own receiver and keeps its K best (d, j) in registers (compare-exchange insertion), so 64 receivers are in
// flight per wave instead of one — the wave-per-receiver kernel above is a chain of dependent round trips per row.
template <int K>
__device__ __forceinline__ int lanes_exact(bool active, float xi, float yi, float zi, bool ti, const GridParams &g, const int32_t *cstart,
                                           const float4 *sorted, float thr, float (&bd)[K], int (&bj)[K])
{
#pragma unroll
    for (int s = 0; s < K; ++s) { bd[s] = 3.0e38f; bj[s] = 0x7fffffff; }
    int cnt = 0;
    if (active) {
        const int ix = cell_coord(xi, g.x0, g.inv, g.nx), iy = cell_coord(yi, g.y0, g.inv, g.ny), iz = cell_coord(zi, g.z0, g.inv, g.nz);
        const int x_lo = ix > 0 ? ix - 1 : 0, x_hi = ix + 1 < g.nx ? ix + 1 : g.nx - 1;
        for (int r = 0; r < 9; ++r) {
            const int z2 = iz + r / 3 - 1, y2 = iy + r % 3 - 1;
            if (z2 < 0 || z2 >= g.nz || y2 < 0 || y2 >= g.ny) continue;
            const int base = (z2 * g.ny + y2) * g.nx;
            const int p1 = cstart[base + x_hi + 1];
            for (int p = cstart[base + x_lo]; p < p1; ++p) {
                const float4 sj = sorted[p];
                const int w = __float_as_int(sj.w);
                const float dx = xi - sj.x, dy = yi - sj.y, dz = zi - sj.z;
                float d = (dx * dx + dy * dy) + dz * dz;
                if (ti && (w & 0x40000000)) d = 1e10f;   // tool-tool (graph.py:118); invalid senders are not binned
                if ((d - thr) < 0.0f) {
                    float cdv = d;
                    int cjv = w & 0x3fffffff;
#pragma unroll
                    for (int s = 0; s < K; ++s) {   // keep bd/bj ascending in (d, j); the loser falls through
                        const bool lt = (cdv < bd[s]) || (cdv == bd[s] && cjv < bj[s]);
                        const float td = lt ? bd[s] : cdv;
                        const int tj = lt ? bj[s] : cjv;
                        bd[s] = lt ? cdv : bd[s];
                        bj[s] = lt ? cjv : bj[s];
                        cdv = td;
                        cjv = tj;
                    }
                    cnt = cnt < K ? cnt + 1 : K;
                }
            }
        }
    }
    return cnt;
}

template <int K>
__global__ __launch_bounds__(256) void select_lanes_kernel(AgEdgeArgs a)
{
    const int b = blockIdx.y, N = a.N;
    if (a.active && !a.active[b]) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const float *pos = a.pos + (size_t)b * a.pos_stride;
    const uint8_t *mk = a.mask + (size_t)b * N, *tl = a.tool + (size_t)b * N;
    const GridParams g = reinterpret_cast<const GridParams *>(a.grid_raw)[b];
    const int32_t *cstart = a.cell_start + (size_t)b * (kCellMax + 1);
    const float4 *sorted = a.sorted + (size_t)b * N;
    const float thr = a.thr_sq[b];
    const size_t row = (size_t)b * N + i;
    const bool mi = mk[i], ti = tl[i];
    float bd[K];
    int bj[K];
    const float xi = mi ? pos[i * 3] : 0.f, yi = mi ? pos[i * 3 + 1] : 0.f, zi = mi ? pos[i * 3 + 2] : 0.f;
    const int cnt = lanes_exact<K>(mi, xi, yi, zi, ti, g, cstart, sorted, thr, bd, bj);
    bool hit = false;
#pragma unroll
    for (int s = 0; s < K; ++s)
        if (s < cnt) {
            int rank = 0;
#pragma unroll
            for (int t = 0; t < K; ++t) rank += (t < cnt && bj[t] < bj[s]) ? 1 : 0;
            a.sel0[row * a.cap0 + rank] = bj[s];        // ascending sender index
            hit = hit || (ti && !tl[bj[s]]);
        }
    a.deg[row] = cnt;
    if (a.connect && a.variant == 1 && hit) atomicOr(&a.flag[b], 1);   // batch_mask (graph.py:123,135)
}

// The same search with packed keys (the default for all three shipped top-k values; it matters most at top-k 20, the granular
// configuration, where the insertion network IS the cost): ~40 in-radius
// candidates per receiver x 20 compare-exchange steps x 7 VALU ops, executed by the whole wave whenever any lane accepts.
//  * keys are ONE 32-bit word, (d as fixed point: floor(d * 2^(32-jb) / thr), in-radius d lies in [0, thr)) << jb | j, so a
//    compare-exchange step is v_min_u32 + v_max_u32 (2 ops instead of 7).  The quantisation is monotone (fp32 multiply and
//    floor are), so the K smallest keys are the K nearest senders EXACTLY unless the K-th and (K+1)-th candidates fall into the
//    same quantum; the network therefore carries K+1 entries and a receiver whose boundary is ambiguous is redone with the
//    exact (d, j) network.  21 bits of d at N ~ 2k: relative resolution 5e-7 of thr against ~1/40 between neighbouring order
//    statistics, i.e. ~1e-5 of the receivers (keeping the fp32 bit pattern's leading bits instead — 12 mantissa bits — sent
//    0.5 % of the receivers, hence every fourth WAVE, through the exact network: 0.21 ms instead of 0.12).  The in-radius test
//    itself is made on the full fp32 d.
//  * receivers are taken in CELL order (thread t = t-th binned particle), so the lanes of a wave walk the same cell ranges:
//    equal trip counts and broadcast candidate loads instead of 64 unrelated walks.
// Results are bit-identical to select_lanes_kernel / the brute-force scan (tests compare all paths with the oracle).
template <int K>
__global__ __launch_bounds__(256, 4) void select_lanes_packed_kernel(AgEdgeArgs a, int jb)
{
    const int b = blockIdx.y, N = a.N;
    if (a.active && !a.active[b]) return;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const uint8_t *mk = a.mask + (size_t)b * N, *tl = a.tool + (size_t)b * N;
    if (t < N && !mk[t]) a.deg[(size_t)b * N + t] = 0;            // receivers that were not binned have no edges
    const GridParams g = reinterpret_cast<const GridParams *>(a.grid_raw)[b];
    const int32_t *cstart = a.cell_start + (size_t)b * (kCellMax + 1);
    const float4 *sorted = a.sorted + (size_t)b * N;
    const float thr = a.thr_sq[b];
    const bool active = t < g.total;
    const float4 me = active ? sorted[t] : make_float4(0.f, 0.f, 0.f, 0.f);
    const int wi = __float_as_int(me.w);
    const int i = wi & 0x3fffffff;
    const bool ti = (wi & 0x40000000) != 0;
    const float xi = me.x, yi = me.y, zi = me.z;
    const unsigned jmask = (1u << jb) - 1u;
    const float qmax = (float)(1u << (32 - jb));                 // quanta in [0, thr): a power of two, exact in fp32
    const float qscale = qmax / thr;
    const unsigned qtop = (1u << (32 - jb)) - 1u;
    unsigned kl[K + 1];
#pragma unroll
    for (int s = 0; s <= K; ++s) kl[s] = 0xffffffffu;
    int tot = 0;
    if (active) {
        const int ix = cell_coord(xi, g.x0, g.inv, g.nx), iy = cell_coord(yi, g.y0, g.inv, g.ny), iz = cell_coord(zi, g.z0, g.inv, g.nz);
        const int x_lo = ix > 0 ? ix - 1 : 0, x_hi = ix + 1 < g.nx ? ix + 1 : g.nx - 1;
        for (int r = 0; r < 9; ++r) {
            const int z2 = iz + r / 3 - 1, y2 = iy + r % 3 - 1;
            if (z2 < 0 || z2 >= g.nz || y2 < 0 || y2 >= g.ny) continue;
            const int base = (z2 * g.ny + y2) * g.nx;
            const int p1 = cstart[base + x_hi + 1];
            // candidates four at a time, their loads in flight together (one load, then its tests, per trip was a memory round trip per candidate)
            for (int p = cstart[base + x_lo]; p < p1; p += 4) {
                float4 sq[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) sq[u] = sorted[p + u < p1 ? p + u : p1 - 1];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (p + u >= p1) break;
                    const float4 sj = sq[u];
                    const int w = __float_as_int(sj.w);
                    const float dx = xi - sj.x, dy = yi - sj.y, dz = zi - sj.z;
                    float d = (dx * dx + dy * dy) + dz * dz;
                    if (ti && (w & 0x40000000)) d = 1e10f;
                    if ((d - thr) < 0.0f) {
                        const unsigned qd = min((unsigned)(d * qscale), qtop);      // monotone in d (d >= 0; a NaN never gets here)
                        unsigned key = (qd << jb) | (unsigned)(w & 0x3fffffff);
#pragma unroll
                        for (int s = 0; s <= K; ++s) {
                            const unsigned lo = min(kl[s], key);
                            key = max(kl[s], key);
                            kl[s] = lo;
                        }
                        ++tot;
                    }
                }
            }
        }
    }
    int cnt = tot < K ? tot : K;
    const bool ambiguous = tot > K && ((kl[K - 1] ^ kl[K]) & ~jmask) == 0u;
    if (__ballot(ambiguous)) {          // rare: the exact network for those lanes (the others idle through it)
        float bd[K];
        int bj[K];
        const int c2 = lanes_exact<K>(ambiguous, xi, yi, zi, ti, g, cstart, sorted, thr, bd, bj);
        if (ambiguous) {
            cnt = c2;
#pragma unroll
            for (int s = 0; s < K; ++s) kl[s] = (unsigned)bj[s];
        }
    }
    // senders in ascending index order: odd-even transposition sort of the K sender ids (unused slots sort to the end)
    unsigned js[K];
#pragma unroll
    for (int s = 0; s < K; ++s) js[s] = s < cnt ? (kl[s] & (ambiguous ? 0x7fffffffu : jmask)) : 0x7fffffffu;
#pragma unroll
    for (int round = 0; round < K; ++round)
#pragma unroll
        for (int s = round & 1; s + 1 < K; s += 2) {
            const unsigned lo = min(js[s], js[s + 1]), hi = max(js[s], js[s + 1]);
            js[s] = lo;
            js[s + 1] = hi;
        }
    if (!active) return;
    const size_t row = (size_t)b * N + i;
    bool hit = false;
#pragma unroll
    for (int s = 0; s < K; ++s)
        if (s < cnt) {
            a.sel0[row * a.cap0 + s] = (int)js[s];
            hit = hit || (ti && !tl[js[s]]);
        }
    a.deg[row] = cnt;
    if (a.connect && a.variant == 1 && hit) atomicOr(&a.flag[b], 1);   // batch_mask (graph.py:123,135)
}

// connect_tools_all overrides: merge {kept object senders} with {all tool senders} in ascending order.
//   single (graph.py:77-80): tool receivers end with no edges; batch (:134-144): gated by batch_mask, and tool receivers
//   keep tool->tool edges (incl. the self loop) when it is set.
// One THREAD per receiver: the kept senders are already ascending and at most top-k long, the sample's tool indices are
// compacted once per workgroup into LDS (ascending), so a row is a two-pointer merge of two short sorted lists.
// (A wave per receiver sweeping all N candidate senders was 45 % of the whole cloth-4k step.)
constexpr int kKeepFast = 8;          // finalize_connect_kernel: kept-sender lists of top-k <= 8 are merged from LDS
constexpr int kToolListMax = 8192;     // LDS-resident tool list; more tools than this take the sweep kernel below

__global__ __launch_bounds__(256) void finalize_connect_kernel(AgEdgeArgs a)
{
    extern __shared__ int tools[];
    __shared__ int wcount[4];
    __shared__ int kept[256 * kKeepFast];           // per-thread list of the kept NON-tool senders (fast path: top-k <= kKeepFast)
    const int b = blockIdx.y, N = a.N, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (a.active && !a.active[b]) return;
    const uint8_t *mk = a.mask + (size_t)b * N, *tl = a.tool + (size_t)b * N;
    // the tool flags of the sample, all of a thread's loads in flight together (N <= 256 * kFlagK; beyond that they are read in the loop)
    constexpr int kFlagK = 17;
    unsigned tflag = 0;
    const bool flags_cached = N <= 256 * kFlagK;
    if (flags_cached) {
        unsigned char tv[kFlagK];
#pragma unroll
        for (int u = 0; u < kFlagK; ++u) tv[u] = tl[tid + 256 * u < N ? tid + 256 * u : 0];
#pragma unroll
        for (int u = 0; u < kFlagK; ++u) tflag |= (tid + 256 * u < N && tv[u]) ? 1u << u : 0u;
    }
    // this receiver's kept senders and their tool flags: two batches of loads instead of a chain of two dependent loads per sender
    const int i = blockIdx.x * 256 + tid;
    const size_t row = (size_t)b * N + (i < N ? i : 0);
    const int deg0 = i < N ? a.deg[row] : 0;
    const bool fast = a.cap0 <= kKeepFast;
    int nkept = 0;
    if (fast) {
        int sv[kKeepFast];
        unsigned char sf[kKeepFast];
#pragma unroll
        for (int u = 0; u < kKeepFast; ++u) sv[u] = u < deg0 ? a.sel0[row * a.cap0 + u] : 0;
#pragma unroll
        for (int u = 0; u < kKeepFast; ++u) sf[u] = tl[sv[u]];
#pragma unroll
        for (int u = 0; u < kKeepFast; ++u)
            if (u < deg0 && !sf[u]) kept[tid * kKeepFast + nkept++] = sv[u];
    }
    int nt = 0;
    if (flags_cached) {          // ordered compaction of the sample's tool indices with two barriers in all (the loop below: two per 256 slots)
        __shared__ int wc[kFlagK][4], wbase[kFlagK][4], s_total;
        unsigned long long bal[kFlagK];
#pragma unroll
        for (int u = 0; u < kFlagK; ++u) {
            bal[u] = __ballot((tflag >> u & 1u) != 0);
            if (lane == 0) wc[u][wave] = __popcll(bal[u]);
        }
        __syncthreads();
        if (tid == 0) {
            int run = 0;
            for (int u = 0; u < kFlagK; ++u)
                for (int w = 0; w < 4; ++w) { wbase[u][w] = run; run += wc[u][w]; }
            s_total = run;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < kFlagK; ++u)
            if (tflag >> u & 1u) {
                const int slot = wbase[u][wave] + __popcll(bal[u] & ((1ull << lane) - 1ull));
                if (slot < a.cap) tools[slot] = tid + 256 * u;      // only the first `cap` tools can reach a (cap-long) output row
            }
        nt = s_total;
        __syncthreads();
    } else
    for (int j0 = 0, it = 0; j0 < N; j0 += 256, ++it) {
        const int j = j0 + tid;
        const bool t = j < N && tl[j];
        const unsigned long long bal = __ballot(t);
        if (lane == 0) wcount[wave] = __popcll(bal);
        __syncthreads();
        int base = nt;
        for (int w = 0; w < wave; ++w) base += wcount[w];
        const int slot = base + __popcll(bal & ((1ull << lane) - 1ull));
        if (t && slot < a.cap) tools[slot] = j;      // only the first `cap` tools can reach a (cap-long) output row
        nt += wcount[0] + wcount[1] + wcount[2] + wcount[3];
        __syncthreads();
    }
    if (i >= N) return;
    const bool mi = mk[i], ti = tl[i];
    const bool add_tools = mi && (a.variant == 1 ? (a.flag[b] != 0) : !ti);
    const bool keep_sel = !ti;
    const int32_t *in = a.sel0 + row * a.cap0;
    int32_t *dst = a.sel + row * a.cap;
    const int na = keep_sel ? deg0 : 0, nb = add_tools ? (nt < a.cap ? nt : a.cap) : 0;
    int ia = 0, ib = 0, out = 0;
    if (fast) {          // both lists in LDS: the merge is a few dozen LDS reads and the row's stores
        const int *ka = kept + tid * kKeepFast;
        const int nk = keep_sel ? nkept : 0;
        while (ia < nk || ib < nb) {
            const bool take_a = ia < nk && (ib >= nb || ka[ia] < tools[ib]);
            const int j = take_a ? ka[ia] : tools[ib];
            if (out < a.cap) dst[out] = j;
            ++out;
            if (take_a) ++ia; else ++ib;
        }
        a.deg[row] = out < a.cap ? out : a.cap;
        return;
    }
    int sa = -1;
    while (ia < na) { sa = in[ia]; if (!tl[sa]) break; ++ia; }          // tool senders among the kept ones come from list B
    while (ia < na || ib < nb) {
        const bool take_a = ia < na && (ib >= nb || sa < tools[ib]);
        const int j = take_a ? sa : tools[ib];
        if (out < a.cap) dst[out] = j;
        ++out;
        if (take_a) { ++ia; while (ia < na) { sa = in[ia]; if (!tl[sa]) break; ++ia; } }
        else ++ib;
    }
    a.deg[row] = out < a.cap ? out : a.cap;
}

// same result for samples with more than kToolListMax tools: one wave per receiver sweeps every candidate sender
__global__ __launch_bounds__(256) void finalize_connect_sweep_kernel(AgEdgeArgs a)
{
    const int b = blockIdx.y, N = a.N;
    if (a.active && !a.active[b]) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint8_t *mk = a.mask + (size_t)b * N, *tl = a.tool + (size_t)b * N;
    const int i = blockIdx.x * 4 + wave;
    if (i >= N) return;
    const size_t row = (size_t)b * N + i;
    const bool mi = mk[i], ti = tl[i];
    const int deg0 = a.deg[row];
    const int mysel = lane < deg0 ? a.sel0[row * a.cap0 + lane] : -1;
    const bool add_tools = mi && (a.variant == 1 ? (a.flag[b] != 0) : !ti);
    const bool keep_sel = !ti;
    int out = 0;
    for (int j0 = 0; j0 < N; j0 += 64) {
        const int j = j0 + lane;
        const bool inb = j < N;
        const bool tj = inb && tl[j];
        bool member = false;
        if (keep_sel) {
            for (int t = 0; t < deg0; ++t) {
                const int st = __shfl(mysel, t);
                member |= (st == j);
            }
            member = member && inb && !tj;
        }
        if (add_tools && tj) member = true;
        const unsigned long long bal = __ballot(member);
        if (member) {
            const int p = out + __popcll(bal & ((1ull << lane) - 1ull));
            if (p < a.cap) a.sel[row * a.cap + p] = j;
        }
        out += __popcll(bal);
    }
    if (lane == 0) a.deg[row] = out < a.cap ? out : a.cap;
}

// ---- exclusive scan of the per-row degrees -> row_ptr, then COO fill -----------------------------
constexpr int kScanRows = 256;    // rows per block (one per thread; ag_api.hip sizes blk_sum with the same number)

// Self-edge elision (AgEdgeArgs::self_attrs): a row's self-loop is left out of the lists when the node's attribute pair is class 0 / 1
// (ag_self_class).  Its position in the (ascending) sender row is found here — the workgroup's 256 rows staged through LDS with coalesced
// loads, each thread then walks its own row — and handed to rowptr_scatter_kernel in self_pos; the scanned degree excludes it.
__device__ __forceinline__ int self_position(const AgEdgeArgs &a, const int32_t *sel, int cap, int *stage, int row0, int nloc, int deg)
{
    const int n = nloc * cap;
    const int32_t *src = sel + (size_t)row0 * cap;
    for (int i0 = threadIdx.x; i0 < n; i0 += 4 * 256) {
        int v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = i0 + 256 * u < n ? src[i0 + 256 * u] : 0;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i0 + 256 * u < n) stage[i0 + 256 * u] = v[u];
    }
    __syncthreads();
    int pos = -1;
    if ((int)threadIdx.x < nloc) {
        const int r = row0 + threadIdx.x, me = r % a.N;
        const int cls = ag_self_class(a.self_attrs[(size_t)r * 2], a.self_attrs[(size_t)r * 2 + 1]);
        if (cls >= 0)
            for (int k = 0; k < deg; ++k)
                if (stage[threadIdx.x * cap + k] == me) pos = k;
        if (pos > 0xffff) pos = -1;      // (self_info keeps the position in 16 bits; a longer row keeps its self-loop as a real edge)
    }
    return pos;
}
constexpr int kSelfStageMax = 256 * 40;      // ints of LDS for the staged rows; rows longer than 40 slots (top-k + tools) are read from global memory

// shared-state rollout: does any of the 256 rows of this block belong to an active sample?  (NULL `active`: yes)
// `with_prev`: ... or the row in front of the block (rowptr_scatter_kernel: row_ptr[row0] is also the END of that row)
__device__ __forceinline__ bool block_rows_active(const AgEdgeArgs &a, int row0, int rows, bool with_prev = false)
{
    if (!a.active) return true;
    const int last = (row0 + kScanRows <= rows ? row0 + kScanRows : rows) - 1;
    for (int b = (with_prev && row0 > 0 ? row0 - 1 : row0) / a.N; b <= last / a.N; ++b)
        if (a.active[b]) return true;
    return false;
}

__global__ __launch_bounds__(256) void scan_partial_kernel(AgEdgeArgs a, const int32_t *sel, int cap)
{
    __shared__ int stage[kSelfStageMax];
    const int rows = a.B * a.N;
    if (!block_rows_active(a, blockIdx.x * kScanRows, rows)) {      // (uniform) nothing to count: no row is read
        if (threadIdx.x == 0) a.blk_sum[blockIdx.x] = 0;
        return;
    }
    const int r = blockIdx.x * kScanRows + threadIdx.x;
    int d = r < rows ? a.deg[r] : 0;
    if (a.active && r < rows && !a.active[r / a.N]) d = 0;      // (shared-state rollout: skipped sample)
    if (a.self_attrs) {
        const int row0 = blockIdx.x * kScanRows, nloc = rows - row0 < kScanRows ? rows - row0 : kScanRows;
        int pos = -1;
        if (cap * 256 <= kSelfStageMax) pos = self_position(a, sel, cap, stage, row0, nloc, d);
        else if (r < rows && ag_self_class(a.self_attrs[(size_t)r * 2], a.self_attrs[(size_t)r * 2 + 1]) >= 0) {
            const int me = r % a.N;
            for (int k = 0; k < d && k <= 0xffff; ++k)
                if (sel[(size_t)r * cap + k] == me) pos = k;
        }
        if (r < rows) a.self_pos[r] = pos;
        d -= pos >= 0 ? 1 : 0;
    }
    int total;
    block_exclusive_scan(d, &total);
    if (threadIdx.x == 0) a.blk_sum[blockIdx.x] = total;
}

// Second and last pass: a block's base offset is the sum of the partial sums in front of it (every block adds them up itself: nblk values, 1 001 at
// 256 x 1 001 rows — cheaper than a one-workgroup launch that scans them), then row_ptr of its 256 rows and, from LDS, their COO entries: one
// thread per (row, slot), coalesced reads of the per-row sender lists, near-coalesced writes, four slots' loads in flight per thread.
// (r05: was scan_blocks + rowptr + scatter, three launches of 5 + 5 + 11 us at C2.)
__global__ __launch_bounds__(256) void rowptr_scatter_kernel(AgEdgeArgs a, const int32_t *sel, int cap)
{
    __shared__ int s_ptr[kScanRows], s_deg[kScanRows], s_self[kScanRows];
    const int rows = a.B * a.N;
    if (blockIdx.x != gridDim.x - 1 && !block_rows_active(a, blockIdx.x * kScanRows, rows, true)) return;      // (uniform; the last block also writes the totals)
    // rider (ag_rollout, de-duplicated node encoder): the sender column mapped to compact rows for round 0's reduce, written where edge_send is.
    // (The overflow word is read here, with the partial sums, not in front of the loop that uses it: one memory round trip less on the chain.)
    const bool map = a.map_send_c != nullptr;
    const bool ident = map && *a.map_ovf != 0;          // the call overflowed the compact tables: round 0 gathers the full-size table by node id
    int part = 0;
    for (int i = threadIdx.x; i < (int)blockIdx.x; i += 256) part += a.blk_sum[i];
    int base;
    block_exclusive_scan(part, &base);
    const int r = blockIdx.x * kScanRows + threadIdx.x;
    const bool live = r < rows && !(a.active && !a.active[r / a.N]);      // (rows of a skipped sample: degree 0, nothing of theirs is read — not even self_pos,
    const int d = live ? a.deg[r] : 0;                                    //  which the degree scan's early-exit blocks never wrote)
    // self-edge elision: the row's self-loop (slot ks) is not stored; self_info tells the segment reduce where it belongs
    const int ks = (a.self_attrs && live) ? a.self_pos[r] : -1;
    if (a.self_attrs && r < rows)
        a.self_info[r] = ks >= 0 ? ((ag_self_class(a.self_attrs[(size_t)r * 2], a.self_attrs[(size_t)r * 2 + 1]) << 16) | ks) : -1;
    int total;
    const int off = base + block_exclusive_scan(d - (ks >= 0 ? 1 : 0), &total);
    s_ptr[threadIdx.x] = off;
    s_deg[threadIdx.x] = d;
    s_self[threadIdx.x] = ks >= 0 ? ks : 0x7fffffff;
    if (r < rows) a.row_ptr[r] = off;
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) a.row_ptr[rows] = base + total;
    if (a.self_attrs && blockIdx.x == gridDim.x - 1 && (int)threadIdx.x < AG_SELF_ROWS) {      // the class rows' synthetic edges, behind the list
        a.edge_recv[base + total + threadIdx.x] = a.self_class_row0 + (int)threadIdx.x / AG_SELF_REPL;      // (AG_SELF_REPL copies per class)
        a.edge_send[base + total + threadIdx.x] = a.self_class_row0 + (int)threadIdx.x / AG_SELF_REPL;
    }
    __syncthreads();
    const int row0 = blockIdx.x * kScanRows;
    const int nloc = rows - row0 < kScanRows ? rows - row0 : kScanRows;
    const int32_t *src = sel + (size_t)row0 * cap;
    const int n = nloc * cap;
    for (int i0 = threadIdx.x; i0 < n; i0 += 4 * 256) {
        int v[4], e[4], sg[4], rw[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = i0 + 256 * u < n ? src[i0 + 256 * u] : 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = i0 + 256 * u;
            e[u] = -1; sg[u] = 0;
            if (idx >= n) continue;
            const int lr = idx / cap, slot = idx - lr * cap;
            if (slot >= s_deg[lr] || slot == s_self[lr]) continue;
            const int row = row0 + lr;
            e[u] = s_ptr[lr] + slot - (slot > s_self[lr] ? 1 : 0);
            sg[u] = (row / a.N) * a.N + v[u];
            a.edge_recv[e[u]] = row;
            a.edge_send[e[u]] = sg[u];
        }
        if (map) {
#pragma unroll
            for (int u = 0; u < 4; ++u) rw[u] = ident ? sg[u] : a.map_node_row[sg[u]];      // (slot 0 of an unused lane: row 0, in range, not stored)
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (e[u] >= 0) a.map_send_c[e[u]] = rw[u];
        }
    }
}

}  // namespace

int ag_launch_build_edges(const AgEdgeArgs &a, hipStream_t s)
{
    const int rows = a.B * a.N;
    int riders = 0;
    // (connect_tools_all: the per-sample batch_mask word is cleared by bin_kernel on the cell path — one fill launch per step less)
    static const int force = getenv("AG_EDGE_CELLS") ? atoi(getenv("AG_EDGE_CELLS")) : -1;   // -1 auto, 0 brute force, 1 cells
    const bool cells = force < 0 ? a.N >= 256 : force != 0;
    if (a.connect && !cells) ag_launch_zero_words(a.flag, a.B, s);
    if (cells) {
        const int nb_tab = a.tab_out ? (rows + 255) / 256 : 0;        // rider workgroups follow the B binning ones
        hipLaunchKernelGGL(bin_kernel, dim3(a.B + nb_tab), dim3(256), 0, s, a);
        if (nb_tab) riders |= AG_RIDER_TAB;
        const dim3 lgrid((a.N + 255) / 256, a.B);
        static const int packed = getenv("AG_EDGE_PACKED") ? atoi(getenv("AG_EDGE_PACKED")) : 1;      // 0: the exact (d, j) network for every receiver
        int jb = 1;
        while ((1 << jb) < a.N) ++jb;
        const bool pk = packed && jb <= 16;          // >= 16 bits of d in the key (else nearly every boundary would be ambiguous)
        if (a.cap0 == 5 && a.topk == 5) {
            if (pk) hipLaunchKernelGGL(select_lanes_packed_kernel<5>, lgrid, dim3(256), 0, s, a, jb);
            else hipLaunchKernelGGL(select_lanes_kernel<5>, lgrid, dim3(256), 0, s, a);
        } else if (a.cap0 == 10 && a.topk == 10) {
            if (pk) hipLaunchKernelGGL(select_lanes_packed_kernel<10>, lgrid, dim3(256), 0, s, a, jb);
            else hipLaunchKernelGGL(select_lanes_kernel<10>, lgrid, dim3(256), 0, s, a);
        } else if (a.cap0 == 20 && a.topk == 20) {
            if (pk) hipLaunchKernelGGL(select_lanes_packed_kernel<20>, lgrid, dim3(256), 0, s, a, jb);
            else hipLaunchKernelGGL(select_lanes_kernel<20>, lgrid, dim3(256), 0, s, a);
        }
        else hipLaunchKernelGGL(select_cells_kernel, dim3((a.N + kRowsPerBlock - 1) / kRowsPerBlock, a.B), dim3(256), 0, s, a);
    } else {
        const int Np = (a.N + 63) & ~63;
        const size_t smem = (size_t)Np * 13 + 4 * kCand * 8;
        hipLaunchKernelGGL(select_kernel, dim3((a.N + kRowsPerBlock - 1) / kRowsPerBlock, a.B), dim3(256), smem, s, a);
    }
    const int32_t *sel = a.sel0;
    int cap = a.cap0;
    if (a.connect) {
        if (a.cap <= kToolListMax)             // cap = top-k + the caller's bound on tools per sample
            hipLaunchKernelGGL(finalize_connect_kernel, dim3((a.N + 255) / 256, a.B), dim3(256), (size_t)a.cap * sizeof(int), s, a);
        else
            hipLaunchKernelGGL(finalize_connect_sweep_kernel, dim3((a.N + 3) / 4, a.B), dim3(256), 0, s, a);
        sel = a.sel;
        cap = a.cap;
    }
    const int nblk = (rows + kScanRows - 1) / kScanRows;
    hipLaunchKernelGGL(scan_partial_kernel, dim3(nblk), dim3(256), 0, s, a, sel, cap);
    hipLaunchKernelGGL(rowptr_scatter_kernel, dim3(nblk), dim3(256), 0, s, a, sel, cap);
    if (a.map_send_c) riders |= AG_RIDER_MAP;
    return riders;
}
