// ag_api.hip — C-ABI entry points of libadaptigraph_hip.so (see include/adaptigraph_hip.h).
// Host-side only: argument checks, weight packing, workspace carving, kernel sequencing on the caller's stream.
#include "../../include/adaptigraph_hip.h"
#include "ag_common.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define AG_HIP(x)                                                                                    \
    do {                                                                                             \
        hipError_t e_ = (x);                                                                         \
        if (e_ != hipSuccess) return fail(AG_ERR_HIP, "%s: %s", #x, hipGetErrorString(e_));          \
    } while (0)

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct Carver {   // bump allocator over the caller's workspace, 256-byte aligned
    char *base;
    size_t off = 0, cap;
    Carver(void *p, size_t c) : base(static_cast<char *>(p)), cap(c) {}
    template <class T> T *take(size_t n)
    {
        off = align_up(off, 256);
        T *r = reinterpret_cast<T *>(base + off);
        off += n * sizeof(T);
        return r;
    }
    bool ok() const { return off <= cap && (base != nullptr || off == 0); }
};

// state_dict order (model.py:103-122)
enum { W_PE0, B_PE0, W_PE1, B_PE1, W_PE2, B_PE2, W_RE0, B_RE0, W_RE1, B_RE1, W_RE2, B_RE2, W_PP, B_PP, W_RP, B_RP,
       W_D0, B_D0, W_D1, B_D1, W_D2, B_D2, N_TENSORS };

uint16_t bf16_rne(float x)   // round-to-nearest-even, as v_cvt_pk_bf16_f32 does (finite inputs)
{
    uint32_t u;
    memcpy(&u, &x, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
// split-fp16 weight pair (edge stack of precision mode 2): hi = fp16(v) (RNE), lo = fp16(v - hi), subnormals kept
void f16_split(float v, uint16_t &hi, uint16_t &lo)
{
    const _Float16 h = (_Float16)v;
    const _Float16 l = (_Float16)(v - (float)h);
    memcpy(&hi, &h, 2);
    memcpy(&lo, &l, 2);
}
// OCP e4m3fn (1-4-3, bias 7, subnormals, max 448, no inf), round-to-nearest-even, saturating: the A-operand format of the block-scaled MFMA
uint8_t e4m3_rne(float v)
{
    const uint8_t sign = std::signbit(v) ? 0x80 : 0;
    const float a = fabsf(v);
    if (!(a > 0.0f)) return sign;
    if (a >= 448.0f) return sign | 0x7e;
    int e;
    (void)frexpf(a, &e);
    e -= 1;                                            // a = f * 2^e, f in [1, 2)
    if (e < -6) {                                      // subnormal: multiples of 2^-9 (8 of them reach the smallest normal, whose pattern is 8)
        return sign | (uint8_t)nearbyintf(ldexpf(a, 9));
    }
    int m = (int)nearbyintf((ldexpf(a, -e) - 1.0f) * 8.0f);
    if (m == 8) { m = 0; e += 1; }
    const int bits = ((e + 7) << 3) | m;
    return sign | (uint8_t)(bits > 0x7e ? 0x7e : bits);
}

float bf16_to_f32(uint16_t b)
{
    const uint32_t u = (uint32_t)b << 16;
    float x;
    memcpy(&x, &u, 4);
    return x;
}

// Narrow first layer (fan-in K <= 24, + bias column K): all five out-tiles in ONE chunk image.
//   F32: [5 tiles][32 rows][32 floats], 16-byte XOR swizzle;  B3: [5 tiles][NU steps][hi|lo][64 lanes][8 bf16]
// b3: 0 = fp32 image, 1 = split-bf16 fragment image, 2 = split-fp16 fragment image (same layout)
// lo_feat0 >= 0 (split-fp16 edge stack): image columns K + 1 + i repeat input column lo_feat0 + i, i < K - lo_feat0 — the kernels put the
// fp16 rounding residual of that input there (ag_mlp.hip, f16_residual), so the first layer multiplies hi + lo of those inputs.
void pack_first_layer(std::vector<float> &dst, int b3, const float *W, int K, int n_out, const float *bias, int lo_feat0 = -1)
{
    const size_t base = dst.size();
    dst.resize(base + AG_CHUNK_FLOATS, 0.0f);
    float *c = dst.data() + base;
    uint16_t *cb = reinterpret_cast<uint16_t *>(c);
    const int NU = (K + 1 + 15) / 16;
    for (int ti = 0; ti < AG_NT; ++ti)
        for (int i = 0; i < 32; ++i) {
            const int o = 32 * ti + i;
            if (o >= n_out) continue;
            const int k_end = lo_feat0 >= 0 ? K + 1 + (K - lo_feat0) : K + 1;
            for (int k = 0; k < k_end; ++k) {
                const float v = k < K ? W[(size_t)o * K + k] : (k == K ? bias[o] : W[(size_t)o * K + (k - K - 1 + lo_feat0)]);
                if (!b3) {
                    c[ti * 1024 + i * 32 + 4 * ((k >> 2) ^ ((i >> 1) & 7)) + (k & 3)] = v;
                } else {
                    const int u = k >> 4, r = k & 15, h = (r >> 2) & 1, e = (r >> 3) * 4 + (r & 3), lane = h * 32 + i;
                    uint16_t hi = bf16_rne(v), lo = bf16_rne(v - bf16_to_f32(hi));
                    if (b3 == 2) f16_split(v, hi, lo);
                    cb[((size_t)((ti * NU + u) * 2 + 0) * 64 + lane) * 8 + e] = hi;
                    cb[((size_t)((ti * NU + u) * 2 + 1) * 64 + lane) * 8 + e] = lo;
                }
            }
        }
}

// Append one layer as n_tiles chunk images (AG_CHUNK_FLOATS floats each): input columns [col0, col0+K) of W in
// image columns [0, K), the bias (if any) in image column K; layout per `b3` as described in ag_common.h.
void pack_layer(std::vector<float> &dst, int b3, const float *W, int ld, int col0, int K, int n_out, const float *bias,
                int n_tiles)
{
    for (int ti = 0; ti < n_tiles; ++ti) {
        const size_t base = dst.size();
        dst.resize(base + AG_CHUNK_FLOATS, 0.0f);
        float *c = dst.data() + base;
        uint16_t *cb = reinterpret_cast<uint16_t *>(c);
        for (int i = 0; i < 32; ++i) {
            const int o = 32 * ti + i;
            if (o >= n_out) continue;
            for (int k = 0; k <= K; ++k) {
                if (k == K && !bias) break;
                const float v = k < K ? W[(size_t)o * ld + col0 + k] : bias[o];
                if (!b3) {
                    c[i * AG_WSTRIDE + 4 * ((k >> 2) ^ ((i >> 1) & 7)) + (k & 3)] = v;
                } else {
                    const int u = k >> 4, r = k & 15, h = (r >> 2) & 1, e = (r >> 3) * 4 + (r & 3), lane = h * 32 + i;
                    uint16_t hi = bf16_rne(v), lo = bf16_rne(v - bf16_to_f32(hi));
                    if (b3 >= 2) f16_split(v, hi, lo);
                    if (b3 == 3) {      // PrecH6: fp16 hi fragments only, [k16-step][64][8]; the fp6 lo part is packed on the device (pack_lo6_kernel)
                        cb[((size_t)u * 64 + lane) * 8 + e] = hi;
                        continue;
                    }
                    cb[((size_t)(2 * u + 0) * 64 + lane) * 8 + e] = hi;
                    cb[((size_t)(2 * u + 1) * 64 + lane) * 8 + e] = lo;
                }
            }
        }
    }
}

// One wide layer of the fp16 edge stack (PrecH3, ag_mlp.hip) as n_tiles chunk images of 20 480 bytes + their block scales:
//   bytes [0, 10240):      hi = fp16(W) fragments, [10 k16-steps u][64 lanes (i, h)][8 fp16], slot e = column 16u + 8(e>>2) + 4h + (e&3)
//   bytes [10240, 20480):  the A operands of the block-scaled fp8 MFMA (v_mfma_scale_f32_32x32x64_f8f6f4), [5 input tiles t][64 lanes (i, h)][32 B]:
//                          bytes 0..15 = e4m3(W_lo / s_lo), bytes 16..31 = e4m3(W_hi / s_hi) of columns 32t + 8q + 4h + p at byte 4q + p
//                          (W_lo = W - hi; the instruction pairs byte k of an A lane with byte k of the B lane of the same half, and the two
//                          16-byte groups are its two K blocks: the kernels put [top byte of x16 | e5m2(x - x16)] of the same columns there)
//   scales (appended to `scales`, 128 dwords per tile): [2 dwords][64 lanes]: lane (i, h) holds the E8M0 exponents of its row's five input tiles,
//                          h = 0: s_lo, h = 1: s_hi (a K block takes its A scale from the lanes of the half with the block's number):
//                          dword 0 = tiles 0..3 (one byte each, selected by op_sel), dword 1 byte 0 = tile 4.  s = 2^(floor(log2(block max)) - 7).
void pack_layer_h3(std::vector<float> &dst, std::vector<uint32_t> &scales, const float *W, int ld, int col0, int K, int n_out, const float *bias,
                   int n_tiles)
{
    for (int ti = 0; ti < n_tiles; ++ti) {
        const size_t base = dst.size();
        dst.resize(base + AG_CHUNK_FLOATS, 0.0f);
        uint16_t *hi16 = reinterpret_cast<uint16_t *>(dst.data() + base);
        uint8_t *mx = reinterpret_cast<uint8_t *>(dst.data() + base) + 10240;
        const size_t sbase = scales.size();
        scales.resize(sbase + 128, 0x7f7f7f7fu);
        for (int i = 0; i < 32; ++i) {
            const int o = 32 * ti + i;
            float w[AG_FP], lo[AG_FP], hif[AG_FP];
            for (int k = 0; k < AG_FP; ++k) {
                w[k] = 0.0f;
                if (o < n_out) w[k] = k < K ? W[(size_t)o * ld + col0 + k] : ((k == K && bias) ? bias[o] : 0.0f);
                uint16_t hb, lb;
                f16_split(w[k], hb, lb);
                const _Float16 hh = (_Float16)w[k];
                hif[k] = (float)hh;
                lo[k] = w[k] - hif[k];
                const int u = k >> 4, r = k & 15, h = (r >> 2) & 1, e = (r >> 3) * 4 + (r & 3), lane = h * 32 + i;
                hi16[((size_t)u * 64 + lane) * 8 + e] = hb;
            }
            for (int t = 0; t < AG_NT; ++t) {
                float mlo = 0.0f, mhi = 0.0f;
                for (int k = 32 * t; k < 32 * t + 32; ++k) { mlo = fmaxf(mlo, fabsf(lo[k])); mhi = fmaxf(mhi, fabsf(hif[k])); }
                auto expo = [](float m) {      // E8M0 byte of 2^(floor(log2 m) - 7): the block maximum lands in [128, 256) <= 448
                    int e = 0;
                    (void)frexpf(m, &e);
                    const int b = (e - 1) - 7 + 127;
                    return b < 1 ? 1 : (b > 254 ? 254 : b);
                };
                const int elo = mlo > 0.0f ? expo(mlo) : 127, ehi = mhi > 0.0f ? expo(mhi) : 127;
                for (int h = 0; h < 2; ++h) {
                    const int lane = h * 32 + i;
                    uint32_t &word = scales[sbase + (t >> 2) * 64 + lane];
                    const uint32_t byte = (uint32_t)(h ? ehi : elo);
                    word = (word & ~(0xffu << (8 * (t & 3)))) | (byte << (8 * (t & 3)));
                    for (int q = 0; q < 4; ++q)
                        for (int pp = 0; pp < 4; ++pp) {
                            const int k = 32 * t + 8 * q + 4 * h + pp;
                            uint8_t *dstb = mx + ((size_t)t * 64 + lane) * 32 + 4 * q + pp;
                            dstb[0] = e4m3_rne(ldexpf(lo[k], 127 - elo));
                            dstb[16] = e4m3_rne(ldexpf(hif[k], 127 - ehi));
                        }
                }
            }
        }
    }
}

}  // namespace

#define AG_MAX_PARTS 4      // batch parts (streams) of a rollout
struct ag_model {
    ag_model_config cfg;
    float *dev = nullptr;       // all packed streams, one allocation
    size_t dev_floats = 0;
    AgWeights w{};
    // optional profiling (ag_profile_enable): event pairs per kernel class, recorded on the caller's stream
    int fuse_agg = 0;           // segment reduce inside node_update (env AG_FUSE_AGG / "fuse_aggregate"): 2 = cooperative LDS-staged reduce
                                // (precision mode 2 only; other modes keep the launch), 0 = separate aggregate launch (default)
    int precision = AG_PREC_B3; // env AG_PRECISION=f32|bf16x3|fast / ag_set_option("precision", 0|1|2)
    int eterm_half = 1;         // precision mode 2 ("fast"): 16-bit (q16) Eterm table, fp16 edge stack
    int max_blocks = 512;       // persistent grid: 2 workgroups per CU
    int edge_products = 2;      // precision mode 2: 2 = fp16 edge stack (split-fp16 weights x fp16 activations + e5m2 residual bytes: PrecH3),
                                // 3 = split-bf16 like mode 1 (env AG_EDGE_PRODUCTS / "edge_products")
    bool h2_ok = true;          // every edge-stack weight fits fp16 (else mode 2 keeps the split-bf16 edge stack)
    int agg_q16 = 1;            // precision mode 2: `agg` as q16 rows between the segment reduce and node_update (env AG_AGG_Q16 / "agg_q16"; default 1, r06): one
                                // more 16-bit rounding per node and round (unsigned, block-scaled), half the bytes of that table; 0 = fp32 rows
    int node_ws = 1;            // split-bf16 node_update of the rounds before the last on the weight-stationary kernel (default; env AG_NODE_WS /
                                // "node_stationary" 0 = the streaming kernel); bit-identical
    int edge_ws = 1;            // fp16 edge stack (PrecH3) on the weight-stationary kernel (default) or, 0, the streaming one (env AG_EDGE_WS / "edge_stationary")
    int self_edges = 1;         // ag_rollout: leave the self-loops of attribute classes (1, 0) / (0, 1) out of the per-edge pipeline — one table row per class, added by
                                // the segment reduce at the self-loop's position (AgFwdArgs::self_info; env AG_SELF_EDGES / "self_edges" 0 = every edge through
                                // the pipeline); bit-identical
    int shared_state = 0;       // ag_rollout: roll the tool-less base trajectory out once and compute per sample only the rows that can differ from it
                                // (ag_shared.hip; env AG_SHARED_STATE / "shared_state"); bit-identical; off by default (the headline benchmark is quoted on the
                                // full per-sample work)
    int node_dedup = 1;         // encode each distinct node-encoder input row of a sample once (env AG_NODE_DEDUP / "node_dedup"): 0 = never (every node,
                                // every step), 1 = where it pays (default: >= 32 768 node-rows x steps per call; below that the two extra small launches cost
                                // more than the shorter kernels save: 0.126 vs 0.115 ms for one 100-particle forward), 2 = always
    int stagger = 1;            // offset the rollout streams by one encode stage (env AG_STAGGER=0 disables)
    int split = 0;              // rollout batch parts on separate streams ("rollout_streams" / env AG_SPLIT): 1..4, or 0 = by the workload (rollout_want)
    hipStream_t aux_stream[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[4] = {nullptr, nullptr, nullptr, nullptr};
    // CU partitioning of the rollout (env AG_CU_SPLIT / "cu_split"): the first `cu_split` CU-mask bits — cu_split / 8 CUs of EVERY
    // XCD (tools/ubench/cu_mask_map.hip: bit b = CU slot b / 8 of XCD b % 8) — run the MFMA-bound edge encoder, the other CUs the HBM-bound
    // edge build / segment reduce / node update / state step, the batch parts pipelined through the two partitions.  0 = off.
    int cu_split = 0;
    int n_cus = 256;
    int part_cus = 0;           // cu_split the two masked streams below were created for
    hipStream_t mfma_stream = nullptr, hbm_stream = nullptr;
    hipEvent_t ev_ready[4] = {nullptr, nullptr, nullptr, nullptr}, ev_enc[4] = {nullptr, nullptr, nullptr, nullptr}, ev_join_mfma = nullptr;
    bool profiling = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev[AG_K_COUNT];
    size_t ev_used[AG_K_COUNT] = {0, 0, 0, 0, 0, 0};
    unsigned long long *edge_counter = nullptr;
    int *status = nullptr;      // device word, sticky: bit 0 = a forward left the range of its arithmetic (fp16 activation overflow, non-finite values)
};

namespace {

int pack_and_upload(ag_model *m, const float *const *t)
{
    const int F = m->cfg.nf, dn = m->cfg.attr_dim + m->cfg.phys_dim + m->cfg.action_dim;
    const int de = 2 * m->cfg.attr_dim + 1 + 3 * m->cfg.n_his;
    std::vector<float> s;
    s.reserve((size_t)(2 * 81 + 16) * AG_CHUNK_FLOATS);
    size_t off[2][4];
    for (int b3 = 0; b3 < 2; ++b3) {
        off[b3][0] = s.size();                                             // node_encode stream
        pack_first_layer(s, b3, t[W_PE0], dn, F, t[B_PE0]);
        pack_layer(s, b3, t[W_PE1], F, 0, F, F, t[B_PE1], AG_NT);
        pack_layer(s, b3, t[W_PE2], F, 0, F, F, t[B_PE2], AG_NT);
        pack_layer(s, b3, t[W_PP], 2 * F, 0, F, F, t[B_PP], AG_NT);        // Pn  = W_pp[:, :F] enc + b_pp
        pack_layer(s, b3, t[W_RP], 3 * F, F, F, F, nullptr, AG_NT);        // Hr  = W_rp[:, F:2F] h
        pack_layer(s, b3, t[W_RP], 3 * F, 2 * F, F, F, nullptr, AG_NT);    // Hs  = W_rp[:, 2F:3F] h
        off[b3][1] = s.size();                                             // edge_encode stream
        pack_first_layer(s, b3, t[W_RE0], de, F, t[B_RE0]);
        pack_layer(s, b3, t[W_RE1], F, 0, F, F, t[B_RE1], AG_NT);
        pack_layer(s, b3, t[W_RE2], F, 0, F, F, t[B_RE2], AG_NT);
        pack_layer(s, b3, t[W_RP], 3 * F, 0, F, F, t[B_RP], AG_NT);        // Eterm = W_rp[:, :F] enc_e + b_rp
        off[b3][2] = s.size();                                             // node_update (not last)
        pack_layer(s, b3, t[W_PP], 2 * F, F, F, F, nullptr, AG_NT);        // W_pp[:, F:2F] agg
        pack_layer(s, b3, t[W_RP], 3 * F, F, F, F, nullptr, AG_NT);
        pack_layer(s, b3, t[W_RP], 3 * F, 2 * F, F, F, nullptr, AG_NT);
        off[b3][3] = s.size();                                             // node_update (last) + decoder
        pack_layer(s, b3, t[W_PP], 2 * F, F, F, F, nullptr, AG_NT);
        pack_layer(s, b3, t[W_D0], F, 0, F, F, t[B_D0], AG_NT);
        pack_layer(s, b3, t[W_D1], F, 0, F, F, t[B_D1], AG_NT);
        pack_layer(s, b3, t[W_D2], F, 0, F, 3, t[B_D2], 1);
    }
    const size_t off_h2 = s.size();                                        // edge_encode stream of the fp16 edge stack (PrecH3)
    std::vector<uint32_t> sc;                                              // block scales of its 15 wide units, 128 dwords each
    pack_first_layer(s, 2, t[W_RE0], de, F, t[B_RE0], 2 * m->cfg.attr_dim + 1);      // + residual columns for the 12 state differences, inputs 5..16 (AG_EDGE_LO_FEAT0)
    pack_layer_h3(s, sc, t[W_RE1], F, 0, F, F, t[B_RE1], AG_NT);
    pack_layer_h3(s, sc, t[W_RE2], F, 0, F, F, t[B_RE2], AG_NT);
    pack_layer_h3(s, sc, t[W_RP], 3 * F, 0, F, F, t[B_RP], AG_NT);
    const size_t off_sc = s.size();
    s.resize(off_sc + sc.size());
    memcpy(s.data() + off_sc, sc.data(), sc.size() * sizeof(uint32_t));
    if (!m->dev) {
        AG_HIP(hipMalloc(reinterpret_cast<void **>(&m->dev), s.size() * sizeof(float)));
        m->dev_floats = s.size();
    } else {
        // Re-packing a LIVE model: kernels still running on any stream (torch side streams are non-blocking, so the
        // null-stream copy below is not ordered against them) may be reading the old streams.  Weight updates are rare
        // (once per checkpoint / optimiser step evaluated through the engine): drain the device first.
        AG_HIP(hipDeviceSynchronize());
    }
    AG_HIP(hipMemcpy(m->dev, s.data(), s.size() * sizeof(float), hipMemcpyHostToDevice));
    // fp16 edge stack (precision mode 2, PrecH3): only if every edge-stack weight and bias is representable in fp16
    {
        bool ok = true;
        auto fits = [&](const float *p, size_t n, size_t ld = 0, size_t cols = 0) {
            for (size_t i = 0; i < n; ++i) {
                const float v = ld ? p[(i / cols) * ld + i % cols] : p[i];
                ok = ok && fabsf(v) <= 65504.0f;
            }
        };
        fits(t[W_RE0], (size_t)F * de); fits(t[B_RE0], F); fits(t[W_RE1], (size_t)F * F); fits(t[B_RE1], F);
        fits(t[W_RE2], (size_t)F * F); fits(t[B_RE2], F); fits(t[W_RP], (size_t)F * F, 3 * F, F); fits(t[B_RP], F);
        m->h2_ok = ok;
    }
    auto at = [&](int b3, int k) { return reinterpret_cast<const float4 *>(m->dev + off[b3][k]); };
    m->w.node_encode = at(0, 0); m->w.edge_encode = at(0, 1); m->w.node_mid = at(0, 2); m->w.node_last = at(0, 3);
    m->w.node_encode_b3 = at(1, 0); m->w.edge_encode_b3 = at(1, 1); m->w.node_mid_b3 = at(1, 2); m->w.node_last_b3 = at(1, 3);
    m->w.edge_encode_h2 = reinterpret_cast<const float4 *>(m->dev + off_h2);
    m->w.edge_scale_h3 = reinterpret_cast<const uint32_t *>(m->dev + off_sc);
    return AG_OK;
}

struct FwdLayout {
    size_t rows_pad, e_pad, rows_c;
};

FwdLayout fwd_layout(int B, int N, int64_t e_cap, bool full_dedup = false)
{
    FwdLayout L;
    L.rows_pad = align_up((size_t)B * N, AG_ROWS_PER_BLOCK);
    L.e_pad = ag_edge_rows_pad(e_cap);   // whole row tiles of either edge encoder (128 / 256 edges), incl. the class rows of elided self-loops
    // compact rows of the de-duplicated node encoder: AG_DEDUP_REPS shared rows per sample + a bounded private budget (the reference's drivers
    // use 2 rows per sample; until r04 this was sized for "every node private": 4 x B (N + 8) rows, 10.7 GB at the planner's 20 000 x 200)
    L.rows_c = align_up((size_t)B * AG_DEDUP_REPS + std::max<size_t>((size_t)B * N / 16, 1024), AG_ROWS_PER_BLOCK);
    // shared-state rollout: the propagation tables are numbered by compact row, so the per-node fallback of an overflowing call has nowhere to
    // write — a private row for every node instead (the budget cannot overflow)
    if (full_dedup) L.rows_c = align_up((size_t)B * (AG_DEDUP_REPS + (size_t)N), AG_ROWS_PER_BLOCK);
    return L;
}

void carve_forward(Carver &c, AgFwdArgs &a, int B, int N, int64_t e_cap, bool eterm16 = false, bool full_dedup = false)
{
    const FwdLayout L = fwd_layout(B, N, e_cap, full_dedup);
    const size_t rc = L.rows_c + AG_ROWS_PER_BLOCK;              // + dump rows of the encoder's out-of-range lanes
    a.h = c.take<float>(L.rows_pad * AG_FP);
    a.pn = c.take<float>(L.rows_pad * AG_FP);
    a.hr = c.take<float>(L.rows_pad * AG_FP);
    a.hs = c.take<float>(L.rows_pad * AG_FP);
    a.hr_out = c.take<float>(L.rows_pad * AG_FP);
    a.hs_out = c.take<float>(L.rows_pad * AG_FP);
    a.h0c = c.take<float>(rc * AG_FP);
    a.pnc = c.take<float>(rc * AG_FP);
    a.hrc = c.take<float>(rc * AG_FP);
    a.hsc = c.take<float>(rc * AG_FP);
    a.node_row = c.take<int32_t>(L.rows_pad);
    a.enc_row = c.take<int32_t>(rc);
    a.enc_src = c.take<int32_t>(rc);
    a.send_c = c.take<int32_t>(L.e_pad);
    a.rows_c = (int)L.rows_c;
    a.agg = c.take<float>(L.rows_pad * AG_FP);
    // per-edge table: fp32 rows (640 B), or — sized for a model in precision mode 2 (ag_*_workspace_bytes_for) — q16 rows (320 B) + the
    // weight-stationary kernel's dump rows behind them
    a.eterm = eterm16 ? c.take<float>((L.e_pad + 256) * (AG_FP / 2)) : c.take<float>(L.e_pad * AG_FP);
    a.edge_node_tab = c.take<float>((L.rows_pad + AG_SELF_ROWS) * 16);      // + the class rows of elided self-loops (ag_edge_node_tab_row)
    a.self_class_row0 = (int)L.rows_pad;
    a.tile_ctr = c.take<int>(AG_TILE_CTRS);
    a.enc_count = a.tile_ctr + 1;                                // (zeroed by run_node_encode before the classification fills the work list)
    a.priv_count = a.tile_ctr + 2;
    a.ovf = a.tile_ctr + 3;
    a.shared_rows = B * AG_DEDUP_REPS;
}

void carve_edges(Carver &c, AgEdgeArgs &a)
{
    const size_t rows = (size_t)a.B * a.N;
    a.sel0 = c.take<int32_t>(rows * a.cap0);
    a.sel = a.connect ? c.take<int32_t>(rows * a.cap) : nullptr;
    a.deg = c.take<int32_t>(rows);
    a.flag = c.take<int32_t>(a.B);
    a.blk_sum = c.take<int32_t>(rows / 256 + 2);       // one partial sum per 256 rows (ag_edges.hip: kScanRows)
    a.grid_raw = c.take<int32_t>((size_t)a.B * 8);
    a.cell_start = c.take<int32_t>((size_t)a.B * 8193);
    a.sorted = c.take<float4>(rows);
}

void edge_caps(int N, int topk, int connect, int max_tools, int *cap0, int *cap)
{
    *cap0 = N < topk ? N : topk;
    *cap = *cap0 + (connect ? max_tools : 0);
}

struct Timed {   // RAII: bracket one kernel launch with an event pair when profiling is on
    ag_model *m;
    int k;
    hipStream_t s;
    hipEvent_t stop = nullptr;
    Timed(ag_model *m_, int k_, hipStream_t s_) : m(m_), k(k_), s(s_)
    {
        if (!m->profiling) return;
        if (m->ev_used[k] == m->ev[k].size()) {
            hipEvent_t a, b;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
            m->ev[k].emplace_back(a, b);
        }
        auto &pr = m->ev[k][m->ev_used[k]++];
        (void)hipEventRecord(pr.first, s);
        stop = pr.second;
    }
    ~Timed() { if (stop) (void)hipEventRecord(stop, s); }
};

void setup_args(ag_model *m, AgFwdArgs &a, int max_blocks, int steps = 1)      // the model's options -> this call's kernel arguments
{
    a.edge_counter = m->profiling ? m->edge_counter : nullptr;
    a.precision = m->precision;
    a.eterm_half = m->eterm_half;
    a.fuse_agg = (m->fuse_agg == 2 && !(a.precision == AG_PREC_B3 && a.eterm_half)) ? 0 : m->fuse_agg;   // mode 2 of the option needs the fp16 table
    a.max_blocks = max_blocks;     // per call, not per model: a model shared by two callers is not mutated
    a.edge_products = m->h2_ok ? m->edge_products : 3;      // a checkpoint with edge-stack weights beyond fp16's range keeps the split-bf16 edge stack
    a.edge_ws = m->edge_ws;
    a.node_ws = m->node_ws;
    // `agg` as q16 rows (mode 2): the q16 reduce writes them, every split-bf16 node_update reads them, the fused reduce rounds its sums the same way
    a.agg_q16 = (m->agg_q16 && a.precision == AG_PREC_B3 && a.eterm_half) ? 1 : 0;
    a.dedup = m->node_dedup && (long long)a.B * (a.N + AG_DEDUP_REPS) < 0x7fffff00LL &&
              (m->node_dedup >= 2 || (long long)a.B * a.N * (steps > 0 ? steps : 1) >= 32768);
    a.hr_row = nullptr; a.pn_rows = nullptr; a.h_rows = nullptr; a.hr_full = nullptr; a.hs_full = nullptr;
    if (!a.dedup) a.ovf = nullptr;
    {   // workgroups of the weight-stationary edge encoder (one per CU): a launch that shares the chip with the other rollout streams
        // takes 1.5x its share of the CUs, capped at all of them (two-stream rollout, C2, r04: 128 CUs -> 127.8 k, 160 -> 133.3 k, 192 -> 134.2 k, 224 -> 132.1 k,
        // 256 -> 132.0 k graph-steps/s)
        const int full = m->max_blocks / AG_MLP_WG_PER_CU, share = max_blocks / AG_MLP_WG_PER_CU * 3 / 2;
        a.ws_blocks = share < full ? (share > 0 ? share : 1) : (full > 0 ? full : 1);
    }
    a.status = m->status;
}

// particle_encoder + hoisted Pn + the first round's Hr / Hs.  Their inputs (attrs, phys, action) do not change during a rollout
// (forward_dynamics.py:177-180: graph["action"] is constant across the inner steps), so with node de-duplication — whose compact
// tables no later kernel overwrites — ag_rollout runs this ONCE per call instead of once per model step.
void run_node_encode(ag_model *m, AgFwdArgs &a, hipStream_t s)
{
    if (a.dedup) ag_launch_zero_words(a.enc_count, 3, s);      // enc_count, priv_count, ovf (a kernel: memset nodes of a captured graph were not replayed reliably)
    { Timed t(m, AG_K_NODE_ENCODE, s); ag_launch_node_encode(m->w, a, s); }
}

// De-duplicated calls only: the per-node encoder behind a test of the overflow flag (a handful of workgroups that read one word and return,
// unless this call's inputs did not fit the compact tables).  Every model step: the full-size tables it fills are overwritten by the rounds.
void run_node_encode_fallback(ag_model *m, AgFwdArgs &a, hipStream_t s)
{
    if (a.dedup) ag_launch_node_encode_fallback(m->w, a, s);
}

// does this call's edge encoder run on the weight-stationary kernel (the condition ag_launch_edge_encode tests, ag_mlp.hip)
bool edge_ws_path(const AgFwdArgs &a)
{
    return a.precision == AG_PREC_B3 && a.eterm_half && a.edge_products == 2 && a.edge_ws && a.n_inst <= 1 && (long long)a.B * a.N * 4 < 0x7fffffffLL;
}

void run_edge_encode(ag_model *m, AgFwdArgs &a, hipStream_t s)
{
    // row-tile claim counter of the STREAMING edge encoders; the weight-stationary kernel (default mode) deals its blocks statically: no fill launch
    const bool ws = edge_ws_path(a);
    if (a.tile_ctr && !ws) ag_launch_zero_words(a.tile_ctr, 1, s);
    { Timed t(m, AG_K_EDGE_ENCODE, s); ag_launch_edge_encode(m->w, a, s); if (a.dedup && !ws && !a.remap_done) ag_launch_send_remap(a, s); }      // (ws: mapped by the node-table launch)
}

void run_propagate(ag_model *m, AgFwdArgs &a, hipStream_t s)
{
    for (int p = 0; p < a.pstep; ++p) {
        AgFwdArgs r = a;                 // this round's view of the tables
        if (a.dedup) {
            r.pn_rows = a.pnc;           // Pn: compact rows in every round
            if (p == 0) {                // round 0 reads the encoder's compact rows: Hr through node_row, Hs through send_c, h from h0c
                r.hr_full = a.hr;        // (or, when the call overflowed the compact tables, the per-node encoder's full-size ones)
                r.hs_full = a.hs;
                r.hr = a.hrc;
                r.hs = a.hsc;
                r.hr_row = a.node_row;
                r.edge_send = a.send_c;
                r.h_rows = a.h0c;
            }
        }
        // The per-edge table (0.8 GB at C2) is streamed once per round and is larger than the 256 MB memory-side cache, which after a pass
        // holds the END of what that pass touched: alternate the direction — the edge encoder writes the table front to back, so round 0
        // starts at the back, round 1 at the front, ... — and every pass begins where the last one ended.  Measured: segment reduce 0.2472 ->
        // 0.2452 ms per launch at C2 (-0.8 %: the cache holds a quarter of one pass, and the reduce is not latency-bound enough to care); same bits.
        // (With the non-temporal hints on the table's loads, ag_common.h, the direction no longer matters: never / always / alternating within 0.5 %.)
        r.agg_reverse = (p & 1) == 0 ? 1 : 0;
        r.hs_q16 = (a.eterm_half && p > 0) ? 1 : 0;                     // rounds after the first gather the q16 rows the previous node_update wrote
        r.hs_out_q16 = (a.eterm_half && p + 1 < a.pstep) ? 1 : 0;
        if (!r.fuse_agg) { Timed t(m, AG_K_AGGREGATE, s); ag_launch_aggregate(r, s); }
        { Timed t(m, AG_K_NODE_UPDATE, s); ag_launch_node_update(m->w, r, p == a.pstep - 1, s); }
        std::swap(a.hr, a.hr_out);
        std::swap(a.hs, a.hs_out);
    }
}

void run_forward(ag_model *m, AgFwdArgs &a, hipStream_t s)
{
    setup_args(m, a, m->max_blocks);
    run_node_encode(m, a, s);
    run_node_encode_fallback(m, a, s);
    run_edge_encode(m, a, s);
    run_propagate(m, a, s);
}

}  // namespace

extern "C" {

const char *ag_last_error(void) { return g_err.c_str(); }
int ag_version(void) { return 1; }

int ag_model_create(const ag_model_config *cfg, const float *const *weights, ag_model **out)
{
    if (!cfg || !weights || !out) return fail(AG_ERR_ARG, "ag_model_create: null argument");
    // The kernels are compiled for the shipped model_config family (config/dynamics/{rope,granular,cloth}.yaml).
    if (cfg->nf != AG_F) return fail(AG_ERR_CONFIG, "nf=%d unsupported (kernels built for %d)", cfg->nf, AG_F);
    if (cfg->n_his != AG_NHIS) return fail(AG_ERR_CONFIG, "n_his=%d unsupported (built for %d)", cfg->n_his, AG_NHIS);
    if (cfg->attr_dim != AG_ATTR || cfg->action_dim != 3)
        return fail(AG_ERR_CONFIG, "attr_dim=%d action_dim=%d unsupported", cfg->attr_dim, cfg->action_dim);
    if (cfg->phys_dim < 0 || cfg->attr_dim + cfg->phys_dim + cfg->action_dim >= AG_NODE_IN_MAX)
        return fail(AG_ERR_CONFIG, "phys_dim=%d unsupported", cfg->phys_dim);
    if (cfg->pstep < 1) return fail(AG_ERR_CONFIG, "pstep=%d unsupported", cfg->pstep);
    for (int i = 0; i < N_TENSORS; ++i)
        if (!weights[i]) return fail(AG_ERR_ARG, "ag_model_create: weight %d is null", i);
    ag_model *m = new ag_model();
    m->cfg = *cfg;
    if (const char *v = getenv("AG_FUSE_AGG")) m->fuse_agg = atoi(v);
    if (m->fuse_agg != 0 && m->fuse_agg != 2) m->fuse_agg = 0;
    if (const char *v = getenv("AG_PRECISION")) {
        const int mode = (!strcmp(v, "f32") || !strcmp(v, "0")) ? 0 : (!strcmp(v, "bf16x3") || !strcmp(v, "1")) ? 1 : 2;
        m->precision = mode ? AG_PREC_B3 : AG_PREC_F32;
        m->eterm_half = mode == 2;
    }
    if (const char *v = getenv("AG_SPLIT")) m->split = atoi(v);
    if (const char *v = getenv("AG_EDGE_WS")) m->edge_ws = atoi(v) != 0;
    if (const char *v = getenv("AG_NODE_WS")) m->node_ws = atoi(v) != 0;
    if (const char *v = getenv("AG_AGG_Q16")) m->agg_q16 = atoi(v) != 0;
    if (const char *v = getenv("AG_EDGE_PRODUCTS")) m->edge_products = atoi(v) == 3 ? 3 : 2;
    if (const char *v = getenv("AG_STAGGER")) m->stagger = atoi(v);
    if (const char *v = getenv("AG_NODE_DEDUP")) m->node_dedup = atoi(v);
    if (const char *v = getenv("AG_SELF_EDGES")) m->self_edges = atoi(v) != 0;
    if (const char *v = getenv("AG_SHARED_STATE")) m->shared_state = atoi(v) != 0;
    {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) {
            m->max_blocks = AG_MLP_WG_PER_CU * prop.multiProcessorCount;
            m->n_cus = prop.multiProcessorCount;
        }
        if (const char *v = getenv("AG_CU_SPLIT")) m->cu_split = atoi(v);
        if (const char *v = getenv("AG_MAX_BLOCKS")) m->max_blocks = atoi(v);
    }
    int rc = pack_and_upload(m, weights);
    if (rc == AG_OK && (hipMalloc(reinterpret_cast<void **>(&m->status), sizeof(int)) != hipSuccess ||
                        hipMemset(m->status, 0, sizeof(int)) != hipSuccess))
        rc = fail(AG_ERR_HIP, "ag_model_create: status word allocation failed");
    if (rc != AG_OK) {
        if (m->status) (void)hipFree(m->status);
        if (m->dev) (void)hipFree(m->dev);
        delete m;
        return rc;
    }
    *out = m;
    return AG_OK;
}

int ag_model_update_weights(ag_model *m, const float *const *weights)
{
    if (!m || !weights) return fail(AG_ERR_ARG, "ag_model_update_weights: null argument");
    return pack_and_upload(m, weights);
}

int ag_model_destroy(ag_model *m)
{
    if (!m) return AG_OK;
    for (auto &v : m->ev)
        for (auto &pr : v) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    if (m->edge_counter) (void)hipFree(m->edge_counter);
    for (int k = 0; k < 4; ++k) {
        if (m->aux_stream[k]) (void)hipStreamDestroy(m->aux_stream[k]);
        if (m->ev_join[k]) (void)hipEventDestroy(m->ev_join[k]);
    }
    if (m->ev_fork) (void)hipEventDestroy(m->ev_fork);
    if (m->mfma_stream) (void)hipStreamDestroy(m->mfma_stream);
    if (m->hbm_stream) (void)hipStreamDestroy(m->hbm_stream);
    for (int k = 0; k < 4; ++k) {
        if (m->ev_ready[k]) (void)hipEventDestroy(m->ev_ready[k]);
        if (m->ev_enc[k]) (void)hipEventDestroy(m->ev_enc[k]);
    }
    if (m->ev_join_mfma) (void)hipEventDestroy(m->ev_join_mfma);
    if (m->status) (void)hipFree(m->status);
    if (m->dev) (void)hipFree(m->dev);
    delete m;
    return AG_OK;
}

static int chamfer_common(const char *who, const float *x, const uint8_t *xm, const float *y, const uint8_t *ym, int B, int N, int M,
                          int y_batched, float *out, ag_stream_t stream)
{
    if (!x || !y || !out) return fail(AG_ERR_ARG, "%s: null argument", who);
    if (B < 1 || N < 1 || M < 1) return fail(AG_ERR_ARG, "%s: bad sizes B=%d N=%d M=%d", who, B, N, M);
    if (ag_launch_chamfer(x, y, xm, ym, B, N, M, y_batched ? 1 : 0, out, static_cast<hipStream_t>(stream)) != 0)
        return fail(AG_ERR_ARG, "%s: N + M = %d exceeds the LDS-resident limit (12800 points)", who, N + M);
    AG_HIP(hipGetLastError());
    return AG_OK;
}

int ag_chamfer(const float *x, const float *y, int B, int N, int M, int y_batched, float *out, ag_stream_t stream)
{
    return chamfer_common("ag_chamfer", x, nullptr, y, nullptr, B, N, M, y_batched, out, stream);
}

int ag_chamfer_masked(const float *x, const uint8_t *x_mask, const float *y, const uint8_t *y_mask, int B, int N, int M,
                      int y_batched, float *out, ag_stream_t stream)
{
    if (!x_mask || !y_mask) return fail(AG_ERR_ARG, "ag_chamfer_masked: null mask");
    return chamfer_common("ag_chamfer_masked", x, x_mask, y, y_mask, B, N, M, y_batched, out, stream);
}

int ag_gather_rows(const float *x, const int32_t *idx, float *out, int64_t n_out, int D, ag_stream_t stream)
{
    if (n_out < 0 || D < 1) return fail(AG_ERR_ARG, "ag_gather_rows: bad sizes n_out=%lld D=%d", (long long)n_out, D);
    if (n_out > 0 && (!x || !idx || !out)) return fail(AG_ERR_ARG, "ag_gather_rows: null argument");
    ag_launch_gather_rows(x, idx, out, n_out, D, static_cast<hipStream_t>(stream));
    AG_HIP(hipGetLastError());
    return AG_OK;
}

int ag_segment_sum(const float *vals, const int32_t *ptr, const int32_t *perm, float *out, int64_t n_seg, int D, ag_stream_t stream)
{
    if (n_seg < 0 || D < 1) return fail(AG_ERR_ARG, "ag_segment_sum: bad sizes n_seg=%lld D=%d", (long long)n_seg, D);
    if (n_seg > 0 && (!vals || !ptr || !out)) return fail(AG_ERR_ARG, "ag_segment_sum: null argument");
    ag_launch_segment_sum(vals, ptr, perm, out, n_seg, D, static_cast<hipStream_t>(stream));
    AG_HIP(hipGetLastError());
    return AG_OK;
}

int ag_message_forward(const float *eterm, const float *hr, const float *hs, const int32_t *row_ptr, const int32_t *send, float *agg,
                       int64_t n_nodes, int D, ag_stream_t stream)
{
    if (n_nodes < 0 || D < 1) return fail(AG_ERR_ARG, "ag_message_forward: bad sizes");
    if (n_nodes > 0 && (!eterm || !hr || !hs || !row_ptr || !send || !agg)) return fail(AG_ERR_ARG, "ag_message_forward: null argument");
    ag_launch_message_fwd(eterm, hr, hs, row_ptr, send, agg, n_nodes, D, static_cast<hipStream_t>(stream));
    AG_HIP(hipGetLastError());
    return AG_OK;
}

int ag_message_backward(const float *eterm, const float *hr, const float *hs, const int32_t *row_ptr, const int32_t *send,
                        const float *grad_agg, float *grad_edge, float *grad_hr, int64_t n_nodes, int D, ag_stream_t stream)
{
    if (n_nodes < 0 || D < 1) return fail(AG_ERR_ARG, "ag_message_backward: bad sizes");
    if (n_nodes > 0 && (!eterm || !hr || !hs || !row_ptr || !send || !grad_agg || !grad_edge || !grad_hr))
        return fail(AG_ERR_ARG, "ag_message_backward: null argument");
    ag_launch_message_bwd(eterm, hr, hs, row_ptr, send, grad_agg, grad_edge, grad_hr, n_nodes, D, static_cast<hipStream_t>(stream));
    AG_HIP(hipGetLastError());
    return AG_OK;
}

int ag_train_pack(const float *W, const float *bias, int n_out, int n_in, int ld, int col0, int transposed, int compact, int n_tiles,
                  int precision, float *dst, ag_stream_t stream)
{
    if (!W || !dst) return fail(AG_ERR_ARG, "ag_train_pack: null argument");
    if (n_out < 1 || n_in < 1 || n_in > AG_F || ld < 1 || col0 < 0 || n_tiles < 1 || n_tiles > AG_NT || n_out > 32 * n_tiles * (compact ? AG_NT : 1))
        return fail(AG_ERR_ARG, "ag_train_pack: bad sizes n_out=%d n_in=%d ld=%d n_tiles=%d", n_out, n_in, ld, n_tiles);
    if (compact && n_in + (bias ? 1 : 0) > 32) return fail(AG_ERR_ARG, "ag_train_pack: compact image holds <= 32 input columns");
    ag_launch_train_pack(W, bias, n_out, n_in, ld, col0, transposed ? 1 : 0, compact ? 1 : 0, n_tiles, precision ? 1 : 0, dst, static_cast<hipStream_t>(stream));
    AG_HIP(hipGetLastError());
    return AG_OK;
}

int ag_train_chain(int kind, int backward, int precision, const float *x, const float *packed, float *const *y, const float *dy,
                   float *const *dz, float *dx, int64_t rows, int d_in, ag_stream_t stream)
{
    if (kind < AG_CHAIN_EDGE || kind > AG_CHAIN_DECODER) return fail(AG_ERR_ARG, "ag_train_chain: bad kind %d", kind);
    const int L = kind == AG_CHAIN_EDGE ? 4 : 3;
    if (!packed || !y || rows < 0 || (!backward && !x) || (backward && (!dy || !dz))) return fail(AG_ERR_ARG, "ag_train_chain: null argument");
    if ((kind == AG_CHAIN_EDGE && d_in != AG_EDGE_IN) || (kind == AG_CHAIN_NODE && (d_in < 1 || d_in >= AG_NODE_IN_MAX)) ||
        (kind == AG_CHAIN_DECODER && d_in != AG_F))
        return fail(AG_ERR_ARG, "ag_train_chain: d_in=%d does not fit kind %d", d_in, kind);
    AgChainArgsPOD p{};
    p.x = x; p.w = packed; p.dy = dy; p.dx = dx; p.rows = rows; p.d_in = d_in;
    for (int l = 0; l < L; ++l) {
        if (!y[l] || (backward && !dz[l])) return fail(AG_ERR_ARG, "ag_train_chain: table %d is null", l);
        p.y[l] = y[l];
        p.dz[l] = backward ? dz[l] : nullptr;
    }
    if (rows == 0) return AG_OK;
    static int cus_of[64];      // CU count per device, queried once (hipGetDeviceProperties is far too slow for a per-launch call)
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
        if (cus_of[dev] == 0) {
            hipDeviceProp_t prop;
            cus_of[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
        }
        cus = cus_of[dev];
    }
    ag_launch_chain(kind, backward ? 1 : 0, precision ? 1 : 0, p, AG_MLP_WG_PER_CU * cus, static_cast<hipStream_t>(stream));
    AG_HIP(hipGetLastError());
    return AG_OK;
}

int ag_edge_inputs_forward(const float *tab, int D, int attr_dim, int group_dim, const int32_t *recv, const int32_t *send, float *out, int64_t n_edges,
                           ag_stream_t stream)
{
    if (D < 1 || attr_dim < 0 || group_dim < 0 || attr_dim + group_dim > D || n_edges < 0) return fail(AG_ERR_ARG, "ag_edge_inputs_forward: bad sizes");
    if (n_edges > 0 && (!tab || !recv || !send || !out)) return fail(AG_ERR_ARG, "ag_edge_inputs_forward: null argument");
    ag_launch_edge_inputs_fwd(tab, D, attr_dim, group_dim, recv, send, out, n_edges, static_cast<hipStream_t>(stream));
    AG_HIP(hipGetLastError());
    return AG_OK;
}

int ag_edge_inputs_backward(const float *tab, int D, int attr_dim, int group_dim, const int32_t *recv, const int32_t *send, const int32_t *row_ptr,
                            const int32_t *col_ptr, const int32_t *send_perm, const float *grad_out, float *scratch_r, float *scratch_s, float *grad_tab,
                            int64_t n_edges, int64_t n_nodes, ag_stream_t stream)
{
    if (D < 1 || attr_dim < 0 || group_dim < 0 || attr_dim + group_dim > D || n_edges < 0 || n_nodes < 0) return fail(AG_ERR_ARG, "ag_edge_inputs_backward: bad sizes");
    if (n_nodes > 0 && (!row_ptr || !col_ptr || !grad_tab)) return fail(AG_ERR_ARG, "ag_edge_inputs_backward: null argument");
    if (n_edges > 0 && (!tab || !recv || !send || !send_perm || !grad_out || !scratch_r || !scratch_s)) return fail(AG_ERR_ARG, "ag_edge_inputs_backward: null argument");
    ag_launch_edge_inputs_bwd(tab, D, attr_dim, group_dim, recv, send, row_ptr, col_ptr, send_perm, grad_out, scratch_r, scratch_s, grad_tab, n_edges, n_nodes,
                              static_cast<hipStream_t>(stream));
    AG_HIP(hipGetLastError());
    return AG_OK;
}

int ag_add3_relu(const float *a, const float *b, const float *c, float *y, int64_t n, ag_stream_t stream)
{
    if (n < 0 || (n & 3) || (n > 0 && (!a || !b || !c || !y))) return fail(AG_ERR_ARG, "ag_add3_relu: n=%lld must be a non-negative multiple of 4, tensors non-null", (long long)n);
    ag_launch_add3_relu(a, b, c, y, n, static_cast<hipStream_t>(stream));
    AG_HIP(hipGetLastError());
    return AG_OK;
}

int ag_relu_mask(const float *g, const float *y, float *out, int64_t n, ag_stream_t stream)
{
    if (n < 0 || (n & 3) || (n > 0 && (!g || !y || !out))) return fail(AG_ERR_ARG, "ag_relu_mask: n=%lld must be a non-negative multiple of 4, tensors non-null", (long long)n);
    ag_launch_relu_mask(g, y, out, n, static_cast<hipStream_t>(stream));
    AG_HIP(hipGetLastError());
    return AG_OK;
}

size_t ag_train_weight_grads_workspace_bytes(int64_t rows, int n_layers)
{
    return ag_weight_grads_ws_floats(rows, n_layers < 1 ? 1 : n_layers) * sizeof(float);
}

int ag_train_weight_grads(int n_layers, const float *const *dz, const int32_t *dz_ld, const float *const *prev, const int32_t *prev_ld, const int32_t *n_in,
                          int64_t rows, float *out, void *workspace, size_t workspace_bytes, ag_stream_t stream)
{
    return ag_train_weight_grads_into(n_layers, dz, dz_ld, prev, prev_ld, n_in, rows, out, nullptr, nullptr, nullptr, nullptr, workspace, workspace_bytes,
                                      stream);
}

int ag_train_weight_grads_into(int n_layers, const float *const *dz, const int32_t *dz_ld, const float *const *prev, const int32_t *prev_ld,
                               const int32_t *n_in, int64_t rows, float *out, float *const *w_grad, const int32_t *w_grad_ld, float *const *b_grad,
                               const int32_t *n_out, void *workspace, size_t workspace_bytes, ag_stream_t stream)
{
    if (n_layers < 1 || n_layers > 4 || !dz || !dz_ld || !prev || !prev_ld || !n_in || rows < 0) return fail(AG_ERR_ARG, "ag_train_weight_grads: bad argument");
    if (!w_grad && !out) return fail(AG_ERR_ARG, "ag_train_weight_grads: no destination");
    for (int l = 0; l < n_layers; ++l) {
        if (!dz[l] || !prev[l] || n_in[l] < 1 || n_in[l] > AG_F || prev_ld[l] < n_in[l] || dz_ld[l] < 1 || dz_ld[l] > AG_FP)
            return fail(AG_ERR_ARG, "ag_train_weight_grads: layer %d: null table or n_in=%d ld=%d", l, n_in[l], prev_ld[l]);
        if (w_grad && (!w_grad_ld || !n_out || (w_grad[l] && (n_out[l] < 1 || n_out[l] > AG_FP || w_grad_ld[l] < n_in[l]))))
            return fail(AG_ERR_ARG, "ag_train_weight_grads_into: layer %d: bad destination", l);
        if (w_grad && !w_grad[l] && !out) return fail(AG_ERR_ARG, "ag_train_weight_grads_into: layer %d has no destination", l);
    }
    if (!workspace || workspace_bytes < ag_train_weight_grads_workspace_bytes(rows, n_layers))
        return fail(AG_ERR_WS, "ag_train_weight_grads: workspace %zu < %zu bytes", workspace_bytes, ag_train_weight_grads_workspace_bytes(rows, n_layers));
    ag_launch_weight_grads(n_layers, dz, dz_ld, prev, prev_ld, n_in, rows, static_cast<float *>(workspace), out, w_grad, w_grad_ld, b_grad, n_out,
                           static_cast<hipStream_t>(stream));
    AG_HIP(hipGetLastError());
    return AG_OK;
}

int ag_set_option(ag_model *m, const char *name, int value)
{
    if (!m || !name) return fail(AG_ERR_ARG, "ag_set_option: null argument");
    if (!strcmp(name, "rollout_streams")) {
        if (value < 0 || value > AG_MAX_PARTS) return fail(AG_ERR_ARG, "ag_set_option: rollout_streams takes 0 (by the workload) or 1..%d, not %d", AG_MAX_PARTS, value);
        m->split = value;
    }
    else if (!strcmp(name, "fuse_aggregate")) {
        if (value != 0 && value != 2) return fail(AG_ERR_ARG, "ag_set_option: fuse_aggregate takes 0 (separate launch) or 2 (reduce inside node_update), not %d", value);
        m->fuse_agg = value;
    }
    else if (!strcmp(name, "precision")) { m->precision = value ? AG_PREC_B3 : AG_PREC_F32; m->eterm_half = value == 2; }
    else if (!strcmp(name, "max_blocks")) m->max_blocks = value;
    else if (!strcmp(name, "edge_products")) {
        if (value != 2 && value != 3) return fail(AG_ERR_ARG, "ag_set_option: edge_products takes 2 (fp16 edge stack, default) or 3 (split-bf16), not %d", value);
        m->edge_products = value;
    }
    else if (!strcmp(name, "edge_stationary")) m->edge_ws = value != 0;
    else if (!strcmp(name, "node_stationary")) m->node_ws = value != 0;
    else if (!strcmp(name, "agg_q16")) m->agg_q16 = value != 0;
    else if (!strcmp(name, "node_dedup")) m->node_dedup = value < 0 ? 0 : (value > 2 ? 2 : value);
    else if (!strcmp(name, "self_edges")) m->self_edges = value != 0;
    else if (!strcmp(name, "shared_state")) m->shared_state = value != 0;
    else if (!strcmp(name, "cu_split")) {
        if (value != 0 && (value < 8 || value > m->n_cus - 8 || (value & 7)))
            return fail(AG_ERR_ARG, "ag_set_option: cu_split takes 0 (off) or a multiple of 8 in [8, %d], not %d", m->n_cus - 8, value);
        m->cu_split = value;
    }
    else return fail(AG_ERR_ARG, "ag_set_option: unknown option '%s'", name);
    return AG_OK;
}

int ag_get_option(const ag_model *m, const char *name, int *value)
{
    if (!m || !name || !value) return fail(AG_ERR_ARG, "ag_get_option: null argument");
    if (!strcmp(name, "rollout_streams")) *value = m->split;
    else if (!strcmp(name, "fuse_aggregate")) *value = m->fuse_agg;
    else if (!strcmp(name, "precision")) *value = m->precision == AG_PREC_F32 ? 0 : (m->eterm_half ? 2 : 1);
    else if (!strcmp(name, "max_blocks")) *value = m->max_blocks;
    else if (!strcmp(name, "edge_products")) *value = m->edge_products;
    else if (!strcmp(name, "edge_stationary")) *value = m->edge_ws;
    else if (!strcmp(name, "node_stationary")) *value = m->node_ws;
    else if (!strcmp(name, "agg_q16")) *value = m->agg_q16;
    else if (!strcmp(name, "node_dedup")) *value = m->node_dedup;
    else if (!strcmp(name, "self_edges")) *value = m->self_edges;
    else if (!strcmp(name, "shared_state")) *value = m->shared_state;
    else if (!strcmp(name, "cu_split")) *value = m->cu_split;
    else return fail(AG_ERR_ARG, "ag_get_option: unknown option '%s'", name);
    return AG_OK;
}

int ag_model_status(ag_model *m, int *flags, ag_stream_t stream)
{
    if (!m || !flags) return fail(AG_ERR_ARG, "ag_model_status: null argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    AG_HIP(hipMemcpyAsync(flags, m->status, sizeof(int), hipMemcpyDeviceToHost, s));
    AG_HIP(hipMemsetAsync(m->status, 0, sizeof(int), s));
    AG_HIP(hipStreamSynchronize(s));
    return AG_OK;
}

int ag_profile_enable(ag_model *m, int enable)
{
    if (!m) return fail(AG_ERR_ARG, "ag_profile_enable: null model");
    if (enable && !m->edge_counter) {
        AG_HIP(hipMalloc(reinterpret_cast<void **>(&m->edge_counter), sizeof(unsigned long long)));
    }
    if (enable) AG_HIP(hipMemset(m->edge_counter, 0, sizeof(unsigned long long)));
    for (auto &u : m->ev_used) u = 0;
    m->profiling = enable != 0;
    return AG_OK;
}

int ag_profile_read(ag_model *m, double *ms, int64_t *launches, int64_t *edges)
{
    if (!m || !ms || !launches || !edges) return fail(AG_ERR_ARG, "ag_profile_read: null argument");
    for (int k = 0; k < AG_K_COUNT; ++k) {
        double tot = 0.0;
        for (size_t i = 0; i < m->ev_used[k]; ++i) {
            float t = 0.f;
            AG_HIP(hipEventSynchronize(m->ev[k][i].second));
            AG_HIP(hipEventElapsedTime(&t, m->ev[k][i].first, m->ev[k][i].second));
            tot += t;
        }
        ms[k] = tot;
        launches[k] = (int64_t)m->ev_used[k];
    }
    unsigned long long e = 0;
    if (m->edge_counter) AG_HIP(hipMemcpy(&e, m->edge_counter, sizeof e, hipMemcpyDeviceToHost));
    *edges = (int64_t)e;
    return AG_OK;
}

int64_t ag_edge_capacity(int B, int N, int topk, int connect_tools_all, int max_tools)
{
    int cap0, cap;
    edge_caps(N, topk, connect_tools_all, max_tools, &cap0, &cap);
    return (int64_t)B * N * (connect_tools_all ? cap : cap0);
}

size_t ag_edges_workspace_bytes(int B, int N, int topk, int connect_tools_all, int max_tools)
{
    AgEdgeArgs a{};
    a.B = B; a.N = N; a.connect = connect_tools_all;
    edge_caps(N, topk, connect_tools_all, max_tools, &a.cap0, &a.cap);
    Carver c(nullptr, 0);
    carve_edges(c, a);
    return align_up(c.off, 256);
}

int ag_build_edges(const float *pos, const uint8_t *mask, const uint8_t *tool_mask, const float *thr_sq, int topk,
                   int connect_tools_all, int variant, int B, int N, int max_tools, int32_t *row_ptr,
                   int32_t *edge_recv, int32_t *edge_send, int64_t e_cap, void *workspace, size_t workspace_bytes,
                   ag_stream_t stream)
{
    if (!pos || !mask || !tool_mask || !thr_sq || !row_ptr || !edge_recv || !edge_send || !workspace)
        return fail(AG_ERR_ARG, "ag_build_edges: null argument");
    if (B < 1 || N < 1 || topk < 1 || topk > 64) return fail(AG_ERR_ARG, "ag_build_edges: B=%d N=%d topk=%d (1..64)", B, N, topk);
    if (variant != AG_VARIANT_SINGLE && variant != AG_VARIANT_BATCH) return fail(AG_ERR_ARG, "bad variant %d", variant);
    if ((int64_t)B * N >= (1ll << 31) / 64) return fail(AG_ERR_ARG, "B*N too large for int32 edge offsets");
    if (e_cap < ag_edge_capacity(B, N, topk, connect_tools_all, max_tools))
        return fail(AG_ERR_ARG, "ag_build_edges: e_cap %lld < ag_edge_capacity()", (long long)e_cap);
    AgEdgeArgs a{};
    a.pos = pos; a.mask = mask; a.tool = tool_mask; a.thr_sq = thr_sq;
    a.topk = topk; a.connect = connect_tools_all ? 1 : 0; a.variant = variant; a.B = B; a.N = N; a.max_tools = max_tools;
    edge_caps(N, topk, a.connect, max_tools, &a.cap0, &a.cap);
    a.row_ptr = row_ptr; a.edge_recv = edge_recv; a.edge_send = edge_send;
    a.pos_stride = (size_t)N * 3;
    Carver c(workspace, workspace_bytes);
    carve_edges(c, a);
    if (!c.ok()) return fail(AG_ERR_WS, "ag_build_edges: workspace %zu < %zu bytes", workspace_bytes, c.off);
    ag_launch_build_edges(a, static_cast<hipStream_t>(stream));
    AG_HIP(hipGetLastError());
    return AG_OK;
}

static bool table16(const ag_model *m) { return m && m->precision == AG_PREC_B3 && m->eterm_half; }      // precision mode 2: q16 rows (320 B)

size_t ag_forward_workspace_bytes(int B, int N, int64_t e_cap) { return ag_forward_workspace_bytes_for(nullptr, B, N, e_cap); }

size_t ag_forward_workspace_bytes_for(const ag_model *m, int B, int N, int64_t e_cap)
{
    AgFwdArgs a{};
    Carver c(nullptr, 0);
    carve_forward(c, a, B, N, e_cap, table16(m));
    return align_up(c.off, 256);
}

int ag_forward(ag_model *m, const float *state, const float *attrs, const float *action, const float *p_instance,
               int n_instance, const float *phys, const int32_t *row_ptr, const int32_t *edge_recv,
               const int32_t *edge_send, int64_t e_cap, int B, int N, int n_p, float *pred_pos, float *pred_motion,
               void *workspace, size_t workspace_bytes, ag_stream_t stream)
{
    if (!m || !state || !attrs || !action || !p_instance || !row_ptr || !edge_recv || !edge_send || !pred_pos ||
        !pred_motion || !workspace)
        return fail(AG_ERR_ARG, "ag_forward: null argument");
    if (m->cfg.phys_dim > 0 && !phys) return fail(AG_ERR_ARG, "ag_forward: phys is null");
    if (B < 1 || N < 1 || n_p < 0 || n_p > N || n_instance < 0 || e_cap < 0 || e_cap > 0x7fffffff)
        return fail(AG_ERR_ARG, "ag_forward: bad sizes B=%d N=%d n_p=%d e_cap=%lld", B, N, n_p, (long long)e_cap);
    AgFwdArgs a{};
    a.state = state; a.attrs = attrs; a.action = action; a.p_instance = p_instance; a.phys = phys;
    a.row_ptr = row_ptr; a.edge_recv = edge_recv; a.edge_send = edge_send;
    a.pred_pos = pred_pos; a.pred_motion = pred_motion;
    a.B = B; a.N = N; a.n_p = n_p; a.n_inst = n_instance; a.phys_dim = m->cfg.phys_dim; a.e_cap = (int)e_cap;
    a.pstep = m->cfg.pstep; a.clamp = m->cfg.motion_clamp;
    Carver c(workspace, workspace_bytes);
    carve_forward(c, a, B, N, e_cap, table16(m));
    if (!c.ok()) return fail(AG_ERR_WS, "ag_forward: workspace %zu < %zu bytes", workspace_bytes, c.off);
    run_forward(m, a, static_cast<hipStream_t>(stream));
    AG_HIP(hipGetLastError());
    return AG_OK;
}

static void carve_rollout(Carver &c, const ag_rollout_params *p, int B, AgFwdArgs &f, AgEdgeArgs &e, float **state,
                          float **pred_pos, float **pred_motion, int n_his, bool eterm16)
{
    const int64_t e_cap = ag_edge_capacity(B, p->N, p->topk, p->connect_tools_all, p->max_tools);
    const size_t rows = (size_t)B * p->N;
    *state = c.take<float>(rows * n_his * 3);
    *pred_pos = c.take<float>((size_t)B * p->n_p * 3 + 4);
    *pred_motion = c.take<float>((size_t)B * p->n_p * 3 + 4);
    e.row_ptr = c.take<int32_t>(rows + 1);
    e.edge_recv = c.take<int32_t>((size_t)e_cap + 1 + AG_SELF_ROWS);      // (+ the synthetic class edges behind the list: self-edge elision)
    e.edge_send = c.take<int32_t>((size_t)e_cap + 1 + AG_SELF_ROWS);
    e.self_info = c.take<int32_t>(rows);
    e.self_pos = c.take<int32_t>(rows);
    e.B = B; e.N = p->N; e.connect = p->connect_tools_all ? 1 : 0;
    edge_caps(p->N, p->topk, e.connect, p->max_tools, &e.cap0, &e.cap);
    carve_edges(c, e);
    carve_forward(c, f, B, p->N, e_cap, eterm16);
    f.e_cap = (int)e_cap;
}

// The batch is rolled out as up to AG_MAX_PARTS independent parts on separate streams: graphs never interact,
// and co-running kernel streams de-phase the memory-bound stages (segment reduce, edge-feature gathers, edge
// build) of one part against the MFMA-bound stages of another — each part's persistent kernels take an equal
// share of the resident-workgroup slots.
// "rollout_streams" 0: two streams hide launch gaps and overlap the MFMA-bound edge encoder of one half with the HBM-bound kernels of the other, but the halves
// also evict each other's tables between a reduce and its node_update.  Measured with the r05 kernels (profiles/r05_nt_hints.txt, ab_streams): one stream wins where
// the edge encoder is a third of the step (rope, top-k 10: +1-2 % at 256 graphs, +4 % at 1 024; cloth, top-k 5: even), two where it is 40 % (granular, top-k 20: +1.7 %).
static int rollout_want(const ag_model *m, const ag_rollout_params *p) { return m && m->split > 0 ? m->split : (p->topk >= 16 ? 2 : 1); }
static int rollout_parts(int B, int want)
{
    int parts = want < 1 ? 1 : (want > AG_MAX_PARTS ? AG_MAX_PARTS : want);
    while (parts > 1 && B / parts < 8) --parts;
    return parts;
}
static void part_range(int B, int parts, int k, int *b0, int *nb)
{
    const int per = (B + parts - 1) / parts;
    *b0 = k * per < B ? k * per : B;
    *nb = (*b0 + per <= B) ? per : B - *b0;
}

// The two CU-masked streams of the partitioned rollout (created once per cu_split value).  Mask bit b is CU slot b / 8 of XCD b % 8
// (tools/ubench/cu_mask_map.hip -> profiles/r05_cu_mask_map.txt), so bits [0, X) and [X, n_cus) with X a multiple of 8 are disjoint sets of
// X / 8 and (n_cus - X) / 8 CUs of EVERY XCD: both partitions keep all eight L2s and all memory channels.
static int ensure_partition(ag_model *m)
{
    if (m->mfma_stream && m->hbm_stream && m->part_cus == m->cu_split) return AG_OK;
    if (m->mfma_stream) { (void)hipStreamSynchronize(m->mfma_stream); (void)hipStreamDestroy(m->mfma_stream); m->mfma_stream = nullptr; }
    if (m->hbm_stream) { (void)hipStreamSynchronize(m->hbm_stream); (void)hipStreamDestroy(m->hbm_stream); m->hbm_stream = nullptr; }
    const int words = (m->n_cus + 31) / 32;
    std::vector<uint32_t> lo(words, 0u), hi(words, 0u);
    for (int b = 0; b < m->n_cus; ++b) (b < m->cu_split ? lo : hi)[b >> 5] |= 1u << (b & 31);
    AG_HIP(hipExtStreamCreateWithCUMask(&m->mfma_stream, (uint32_t)words, lo.data()));
    AG_HIP(hipExtStreamCreateWithCUMask(&m->hbm_stream, (uint32_t)words, hi.data()));
    for (int k = 0; k < 4; ++k) {
        if (!m->ev_ready[k]) AG_HIP(hipEventCreateWithFlags(&m->ev_ready[k], hipEventDisableTiming));
        if (!m->ev_enc[k]) AG_HIP(hipEventCreateWithFlags(&m->ev_enc[k], hipEventDisableTiming));
    }
    if (!m->ev_join_mfma) AG_HIP(hipEventCreateWithFlags(&m->ev_join_mfma, hipEventDisableTiming));
    m->part_cus = m->cu_split;
    return AG_OK;
}

// ---- shared-state rollout (ag_shared.hip): one stream, the batch as ONE part behind the base sample -------------------------------------------------
struct SharedLayout {
    AgSharedArgs sh{};
    AgFwdArgs f{};
    AgEdgeArgs e{};
    float *pred_pos = nullptr, *pred_motion = nullptr;
};

static bool shared_applicable(const ag_model *m, const ag_rollout_params *p)
{
    if (!m || !p || !m->shared_state || p->B < 2 || p->n_steps < 1) return false;
    if (m->fuse_agg != 0 || m->cu_split != 0) return false;      // (the fused reduce and the CU-partitioned pipeline keep the plain path)
    if ((long long)(p->B + 1) * (p->N + AG_DEDUP_REPS) >= 0x7fffff00LL) return false;
    if (ag_edge_capacity(p->B + 1, p->N, p->topk, p->connect_tools_all, p->max_tools) >= 0x7fffff00LL) return false;
    return true;
}

static void carve_shared(Carver &c, const ag_model *m, const ag_rollout_params *p, SharedLayout &L, bool eterm16)
{
    const int B1 = p->B + 1, N = p->N, n_p = p->n_p, H = AG_NHIS, Pd = m->cfg.phys_dim, I = p->n_instance;
    const int64_t e_cap = ag_edge_capacity(B1, N, p->topk, p->connect_tools_all, p->max_tools);
    const size_t rows = (size_t)B1 * N, ecoo = (size_t)e_cap + 1 + AG_SELF_ROWS;
    AgSharedArgs &s = L.sh;
    s.s_state = c.take<float>(rows * H * 3);
    s.s_delta = c.take<float>(rows * 3);
    s.s_attrs = c.take<float>(rows * 2);
    s.s_pinst = c.take<float>((size_t)B1 * n_p * (I > 0 ? I : 1));
    s.s_phys = c.take<float>((size_t)B1 * (Pd > 0 ? Pd : 1));
    s.s_thr = c.take<float>(B1);
    s.s_mask = c.take<uint8_t>(rows);
    s.s_tool = c.take<uint8_t>(rows);
    s.s_obj_mask = c.take<uint8_t>((size_t)B1 * n_p);
    s.s_repeat = c.take<int32_t>(B1);
    s.dirty = c.take<uint8_t>(rows);
    s.sel_a = c.take<uint8_t>(rows);
    s.sel_b = c.take<uint8_t>(rows);
    s.sample_dirty = c.take<int32_t>(B1);
    s.active = c.take<int32_t>(B1);
    s.cmap = c.take<int32_t>(rows);
    s.orig = c.take<int32_t>(rows);
    s.row_ptr_c = c.take<int32_t>(rows + 1);
    s.self_info_c = c.take<int32_t>(rows);
    s.node_row_c = c.take<int32_t>(rows + AG_ROWS_PER_BLOCK);
    s.recv_o = c.take<int32_t>(ecoo);
    s.send_o = c.take<int32_t>(ecoo);
    s.send_r0 = c.take<int32_t>(ecoo);
    s.send_cm = c.take<int32_t>(ecoo);
    s.blk_cnt = c.take<int32_t>(rows / 256 + 2);
    s.blk_deg = c.take<int32_t>(rows / 256 + 2);
    s.n_rows = c.take<int>(4);
    s.n_edges = s.n_rows + 1;
    L.pred_pos = c.take<float>(rows * 3 + 4);
    L.pred_motion = c.take<float>(rows * 3 + 4);
    AgEdgeArgs &e = L.e;
    e.row_ptr = c.take<int32_t>(rows + 1);
    e.edge_recv = c.take<int32_t>(ecoo);
    e.edge_send = c.take<int32_t>(ecoo);
    e.self_info = c.take<int32_t>(rows);
    e.self_pos = c.take<int32_t>(rows);
    e.B = B1; e.N = N; e.connect = p->connect_tools_all ? 1 : 0;
    edge_caps(N, p->topk, e.connect, p->max_tools, &e.cap0, &e.cap);
    carve_edges(c, e);
    carve_forward(c, L.f, B1, N, e_cap, eterm16, true);
    L.f.e_cap = (int)e_cap;
}

static int rollout_shared(ag_model *m, const ag_rollout_params *p, const float *state0, const float *delta, const float *attrs,
                          const float *p_instance, const float *phys, const uint8_t *mask, const uint8_t *tool_mask, const uint8_t *obj_mask,
                          const float *thr_sq, const int32_t *repeat, float *out_seq, float *state_final, void *workspace, size_t workspace_bytes,
                          hipStream_t s)
{
    Carver c(workspace, workspace_bytes);
    SharedLayout L;
    carve_shared(c, m, p, L, table16(m));
    if (!c.ok()) return fail(AG_ERR_WS, "ag_rollout (shared state): workspace %zu < %zu bytes", workspace_bytes, c.off);
    AgSharedArgs &sh = L.sh;
    AgFwdArgs &f = L.f;
    AgEdgeArgs &e = L.e;
    const int B1 = p->B + 1, N = p->N, n_p = p->n_p, H = m->cfg.n_his, Pd = m->cfg.phys_dim;
    const size_t plane = (size_t)N * 3;
    sh.B1 = B1; sh.N = N; sh.n_p = n_p; sh.n_inst = p->n_instance; sh.phys_dim = Pd; sh.H = H;
    sh.state0 = state0; sh.delta = delta; sh.attrs = attrs; sh.p_instance = p_instance; sh.phys = phys; sh.thr_sq = thr_sq;
    sh.mask = mask; sh.tool = tool_mask; sh.obj_mask = obj_mask; sh.repeat = repeat;
    sh.max_tools = p->max_tools;
    ag_launch_shared_stage(sh, s);      // internal sample 0 = the base (caller sample 0 without its tools), 1 .. B = the caller's samples; first dirty flags
    e.mask = sh.s_mask; e.tool = sh.s_tool; e.thr_sq = sh.s_thr; e.topk = p->topk; e.variant = AG_VARIANT_BATCH; e.max_tools = p->max_tools;
    e.pos = sh.s_state + (size_t)(H - 1) * plane;
    e.pos_stride = (size_t)H * plane;
    e.active = sh.active;
    f.state = sh.s_state; f.attrs = sh.s_attrs; f.action = sh.s_delta; f.p_instance = sh.s_pinst; f.phys = Pd > 0 ? sh.s_phys : nullptr;
    f.row_ptr = e.row_ptr; f.edge_recv = e.edge_recv; f.edge_send = e.edge_send;
    f.pred_pos = L.pred_pos; f.pred_motion = L.pred_motion;
    f.B = B1; f.N = N; f.n_p = n_p; f.n_inst = p->n_instance; f.phys_dim = Pd; f.pstep = m->cfg.pstep; f.clamp = m->cfg.motion_clamp;
    setup_args(m, f, m->max_blocks, p->n_steps);
    f.dedup = 1;                        // the propagation rounds read the node encoder through its compact rows whatever "node_dedup" says (carve: no overflow possible)
    f.ovf = f.tile_ctr + 3;
    const int self_rows = m->self_edges ? AG_SELF_ROWS : 0;
    if (self_rows) { e.self_attrs = f.attrs; e.self_class_row0 = f.self_class_row0; }
    if (edge_ws_path(f)) {              // rider: the per-node input rows of the edge features (all B1 N nodes: the compact edges name their endpoints as nodes)
        e.tab_state = f.state; e.tab_attrs = f.attrs; e.tab_pinst = f.p_instance; e.tab_out = f.edge_node_tab;
        e.tab_n_inst = f.n_inst; e.tab_n_p = f.n_p; e.tab_status = f.status;
    }
    run_node_encode(m, f, s);           // classification + compact encoder, once per call
    sh.row_ptr = e.row_ptr; sh.edge_send = e.edge_send; sh.self_info = self_rows ? e.self_info : nullptr; sh.node_row = f.node_row;
    sh.self_rows = self_rows; sh.self_class_row0 = f.self_class_row0;
    AgFwdArgs fE = f, fP = f;           // the edge encoder's and the propagation rounds' views of the compact graph
    fE.edge_recv = sh.recv_o; fE.edge_send = sh.send_o; fE.e_count_dev = sh.n_edges; fE.remap_done = 1; fE.self_rows = self_rows;
    fP.row_ptr = sh.row_ptr_c; fP.edge_send = sh.send_cm; fP.send_c = sh.send_r0; fP.node_row = sh.node_row_c;
    fP.self_info = self_rows ? sh.self_info_c : nullptr; fP.self_rows = self_rows;
    fP.n_rows_dev = sh.n_rows; fP.e_count_dev = sh.n_edges; fP.row_orig = sh.orig;
    AgStepArgs st{};
    st.state = sh.s_state; st.delta = sh.s_delta; st.pred_pos = L.pred_pos; st.obj_mask = obj_mask ? sh.s_obj_mask : nullptr;
    st.repeat = sh.s_repeat; st.out_seq = out_seq; st.B = B1; st.N = N; st.n_p = n_p; st.H = H;
    st.height_mode = p->height_mode; st.raise = p->gripper_raise; st.cmap = sh.cmap; st.dirty = sh.dirty; st.sample_dirty = sh.sample_dirty;
    for (int ai = 1; ai <= p->n_steps; ++ai) {
        int riders;
        { Timed tm(m, AG_K_EDGES, s); ag_launch_shared_active(sh, s); riders = ag_launch_build_edges(e, s); ag_launch_shared_compact(sh, s); }
        fE.tab_done = (riders & AG_RIDER_TAB) != 0;
        run_edge_encode(m, fE, s);
        run_propagate(m, fP, s);
        st.step = ai;
        { Timed tm(m, AG_K_ROLLOUT_STEP, s); ag_launch_rollout_step(st, s); }
    }
    if (state_final)
        AG_HIP(hipMemcpyAsync(state_final, sh.s_state + (size_t)H * plane, (size_t)p->B * H * plane * sizeof(float), hipMemcpyDeviceToDevice, s));
    AG_HIP(hipGetLastError());
    return AG_OK;
}

size_t ag_rollout_workspace_bytes(const ag_rollout_params *p) { return ag_rollout_workspace_bytes_for(nullptr, p); }
int ag_rollout_streams_for(const ag_model *m, const ag_rollout_params *p) { return m && p ? (shared_applicable(m, p) ? 1 : rollout_parts(p->B, rollout_want(m, p))) : 0; }

size_t ag_rollout_workspace_bytes_for(const ag_model *m, const ag_rollout_params *p)
{
    if (!p) return 0;
    // ag_rollout carves one layout per batch part and the number of parts is a model option ("rollout_streams", 1..4) the
    // caller may change between this query and the call: size for the largest of the four possible carvings, exactly.
    size_t need = 0;
    for (int want = 1; want <= AG_MAX_PARTS; ++want) {
        const int parts = rollout_parts(p->B, want);
        Carver c(nullptr, 0);
        for (int k = 0; k < parts; ++k) {
            AgFwdArgs f{};
            AgEdgeArgs e{};
            float *a, *b, *d;
            int b0, nb;
            part_range(p->B, parts, k, &b0, &nb);
            carve_rollout(c, p, nb, f, e, &a, &b, &d, AG_NHIS, table16(m));
        }
        need = c.off > need ? c.off : need;
    }
    if (shared_applicable(m, p)) {      // (an option the caller may switch off again before the call: the larger of the two layouts)
        Carver c(nullptr, 0);
        SharedLayout L;
        carve_shared(c, m, p, L, table16(m));
        need = c.off > need ? c.off : need;
    }
    return align_up(need, 256);
}

int ag_rollout(ag_model *m, const ag_rollout_params *p, const float *state0, const float *delta, const float *attrs,
               const float *p_instance, const float *phys, const uint8_t *mask, const uint8_t *tool_mask,
               const uint8_t *obj_mask, const float *thr_sq, const int32_t *repeat, float *out_seq,
               float *state_final, void *workspace, size_t workspace_bytes, ag_stream_t stream)
{
    if (!m || !p || !state0 || !delta || !attrs || !p_instance || !mask || !tool_mask || !thr_sq || !repeat ||
        !out_seq || !workspace)
        return fail(AG_ERR_ARG, "ag_rollout: null argument");
    if (p->height_mode == AG_HEIGHT_MASKED_MEAN && !obj_mask) return fail(AG_ERR_ARG, "ag_rollout: obj_mask is null");
    if (p->B < 1 || p->N < 1 || p->n_p < 1 || p->n_p > p->N || p->topk < 1 || p->topk > 64 || p->n_steps < 0)
        return fail(AG_ERR_ARG, "ag_rollout: bad sizes");
    if (m->cfg.phys_dim > 0 && !phys) return fail(AG_ERR_ARG, "ag_rollout: phys is null");
    hipStream_t s0 = static_cast<hipStream_t>(stream);
    if (shared_applicable(m, p))
        return rollout_shared(m, p, state0, delta, attrs, p_instance, phys, mask, tool_mask, obj_mask, thr_sq, repeat, out_seq, state_final, workspace,
                              workspace_bytes, s0);
    const int H = m->cfg.n_his, N = p->N, n_p = p->n_p, Pd = m->cfg.phys_dim;
    const int parts = rollout_parts(p->B, rollout_want(m, p));
    for (int k = 1; k < parts; ++k)
        if (!m->aux_stream[k]) {
            AG_HIP(hipStreamCreateWithFlags(&m->aux_stream[k], hipStreamNonBlocking));
            AG_HIP(hipEventCreateWithFlags(&m->ev_join[k], hipEventDisableTiming));
        }
    if (parts > 1 && !m->ev_fork) AG_HIP(hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming));
    Carver c(workspace, workspace_bytes);
    struct Part { AgFwdArgs f{}; AgEdgeArgs e{}; float *state, *pp, *pm; int b0, B; } part[AG_MAX_PARTS];
    for (int k = 0; k < parts; ++k) {
        part_range(p->B, parts, k, &part[k].b0, &part[k].B);
        carve_rollout(c, p, part[k].B, part[k].f, part[k].e, &part[k].state, &part[k].pp, &part[k].pm, H, table16(m));
    }
    if (!c.ok()) return fail(AG_ERR_WS, "ag_rollout: workspace %zu < %zu bytes", workspace_bytes, c.off);
    if (parts > 1) {
        AG_HIP(hipEventRecord(m->ev_fork, s0));
        for (int k = 1; k < parts; ++k) AG_HIP(hipStreamWaitEvent(m->aux_stream[k], m->ev_fork, 0));
    }
    const size_t plane = (size_t)N * 3;
    const int part_blocks = m->max_blocks / parts > 0 ? m->max_blocks / parts : 1;   // each part's persistent kernels take an equal share
    // CU-partitioned pipeline (cu_split > 0; needs >= 2 batch parts and the weight-stationary edge encoder, whose grid is one workgroup per CU)
    const bool partitioned = m->cu_split >= 8 && m->cu_split <= m->n_cus - 8 && parts >= 2 && m->precision == AG_PREC_B3 && m->eterm_half &&
                             m->edge_products == 2 && m->h2_ok && m->edge_ws && p->n_instance <= 1 && (long long)p->B * N * 4 < 0x7fffffffLL;
    hipStream_t sE = nullptr, sR = nullptr;
    if (partitioned) {
        const int prc = ensure_partition(m);
        if (prc != AG_OK) return prc;
        sE = m->mfma_stream; sR = m->hbm_stream;
        AG_HIP(hipStreamWaitEvent(sE, m->ev_fork, 0));
        AG_HIP(hipStreamWaitEvent(sR, m->ev_fork, 0));
    }
    int rc = AG_OK;
    // issue the steps round-robin over the parts so every stream always has work queued
    struct Run { AgStepArgs st{}; hipStream_t s; } run[AG_MAX_PARTS];
    for (int k = 0; k < parts && rc == AG_OK; ++k) {
        Part &q = part[k];
        hipStream_t s = partitioned ? sR : (k == 0 ? s0 : m->aux_stream[k]);
        run[k].s = s;
        const size_t b0 = (size_t)q.b0;
        const size_t state_bytes = (size_t)q.B * H * plane * sizeof(float);
        if (hipMemcpyAsync(q.state, state0 + b0 * H * plane, state_bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) { rc = AG_ERR_HIP; break; }
        AgEdgeArgs &e = q.e;
        AgFwdArgs &f = q.f;
        e.mask = mask + b0 * N; e.tool = tool_mask + b0 * N; e.thr_sq = thr_sq + b0; e.topk = p->topk;
        e.variant = AG_VARIANT_BATCH; e.max_tools = p->max_tools;
        e.pos = q.state + (size_t)(H - 1) * plane;     // edges on the current frame = state[:, -1] (forward_dynamics.py:125 / :171)
        e.pos_stride = (size_t)H * plane;
        f.state = q.state; f.attrs = attrs + b0 * N * 2; f.action = delta + b0 * plane;
        f.p_instance = p_instance + b0 * n_p * p->n_instance; f.phys = phys ? phys + b0 * Pd : nullptr;
        f.row_ptr = e.row_ptr; f.edge_recv = e.edge_recv; f.edge_send = e.edge_send;
        f.pred_pos = q.pp; f.pred_motion = q.pm;
        f.B = q.B; f.N = N; f.n_p = n_p; f.n_inst = p->n_instance; f.phys_dim = Pd;
        f.pstep = m->cfg.pstep; f.clamp = m->cfg.motion_clamp;
        if (m->self_edges) {      // self-edge elision: the builder leaves class-0 / class-1 self-loops out of the lists, the encoder adds one row per class
            e.self_attrs = f.attrs; e.self_class_row0 = f.self_class_row0;
            f.self_info = e.self_info; f.self_rows = AG_SELF_ROWS;
        }
        AgStepArgs &st = run[k].st;
        st.state = q.state; st.delta = delta + b0 * plane; st.pred_pos = q.pp;
        st.obj_mask = obj_mask ? obj_mask + b0 * n_p : nullptr; st.repeat = repeat + b0;
        st.out_seq = out_seq + b0 * n_p * 3; st.B = q.B; st.N = N; st.n_p = n_p; st.H = H;
        st.height_mode = p->height_mode; st.raise = p->gripper_raise;
    }
    if (partitioned) {
        // Two in-order queues with disjoint CU masks.  HBM queue (sR): edge build -> [ready] ... [encoded] -> three propagation rounds -> state step;
        // MFMA queue (sE): [ready] -> per-node input rows + edge encoder + sender remap -> [encoded].  Part k's step ai + 1 cannot start before its
        // step ai has finished (the edges are rebuilt from the predicted positions), so with two parts the steady state is
        //     sE:  E(A, i+1)  E(B, i+1)  E(A, i+2) ...          sR:  R(B, i)  R(A, i+1)  R(B, i+1) ...
        // each partition always busy with its own kind of work, the other part's.  Every table is per part, and E(k, i+1) is ordered behind
        // R(k, i) through `ready`, so nothing is overwritten while it is read.
        // Enqueue order matters (both queues are in order): a part's NEXT edge build follows its own state step directly, and the wait for the
        // encoder stands in front of the rounds that need it — not in front of the other part's edge build.
        auto encode = [&](int k) {       // sR: [edges built] -> ready;  sE: ready -> encoder -> encoded
            if (hipEventRecord(m->ev_ready[k], sR) != hipSuccess || hipStreamWaitEvent(sE, m->ev_ready[k], 0) != hipSuccess) return AG_ERR_HIP;
            run_edge_encode(m, part[k].f, sE);
            return hipEventRecord(m->ev_enc[k], sE) == hipSuccess ? AG_OK : AG_ERR_HIP;
        };
        for (int k = 0; k < parts && rc == AG_OK && p->n_steps > 0; ++k) {
            AgFwdArgs &f = part[k].f;
            { Timed tm(m, AG_K_EDGES, sR); ag_launch_build_edges(part[k].e, sR); }
            setup_args(m, f, AG_MLP_WG_PER_CU * (m->n_cus - m->cu_split), p->n_steps);     // persistent node kernels: the HBM partition's CUs
            f.ws_blocks = m->cu_split;                                                     // edge encoder: one workgroup per CU of the MFMA partition
            run_node_encode(m, f, sR);
            run_node_encode_fallback(m, f, sR);
            rc = encode(k);
        }
        for (int ai = 1; ai <= p->n_steps && rc == AG_OK; ++ai)
            for (int k = 0; k < parts && rc == AG_OK; ++k) {
                AgFwdArgs &f = part[k].f;
                if (hipStreamWaitEvent(sR, m->ev_enc[k], 0) != hipSuccess) { rc = AG_ERR_HIP; break; }
                run_propagate(m, f, sR);
                run[k].st.step = ai;
                { Timed tm(m, AG_K_ROLLOUT_STEP, sR); ag_launch_rollout_step(run[k].st, sR); }
                if (ai < p->n_steps) {
                    { Timed tm(m, AG_K_EDGES, sR); ag_launch_build_edges(part[k].e, sR); }
                    if (!f.dedup) run_node_encode(m, f, sR);
                    run_node_encode_fallback(m, f, sR);
                    rc = encode(k);
                }
            }
    } else
    for (int ai = 1; ai <= p->n_steps && rc == AG_OK; ++ai)
        for (int k = 0; k < parts && rc == AG_OK; ++k) {
            hipStream_t s = run[k].s;
            AgFwdArgs &f = part[k].f;
            AgEdgeArgs &e = part[k].e;
            if (ai == 1) {
                setup_args(m, f, part_blocks, p->n_steps);
                // Riders of the edge builder's launches (AgEdgeArgs): the per-node input rows of the weight-stationary edge encoder and, for a
                // de-duplicated node encoder, the sender column mapped to compact rows — one 15 us launch per model step less.  The map needs
                // node_row, so the node encoder (which does not read the edges) goes in front of the first step's edge build.
                if (edge_ws_path(f)) {
                    e.tab_state = f.state; e.tab_attrs = f.attrs; e.tab_pinst = f.p_instance; e.tab_out = f.edge_node_tab;
                    e.tab_n_inst = f.n_inst; e.tab_n_p = f.n_p; e.tab_status = f.status;
                }
                if (f.dedup) { e.map_node_row = f.node_row; e.map_ovf = f.ovf; e.map_send_c = f.send_c; }
            }
            if (ai == 1 || !f.dedup) run_node_encode(m, f, s);       // step-invariant when de-duplicated (see run_node_encode)
            int riders;
            { Timed tm(m, AG_K_EDGES, s); riders = ag_launch_build_edges(e, s); }
            f.tab_done = (riders & AG_RIDER_TAB) != 0;
            f.remap_done = (riders & AG_RIDER_MAP) != 0;
            run_node_encode_fallback(m, f, s);
            run_edge_encode(m, f, s);
            if (ai == 1 && k == 0 && parts > 1 && m->stagger) {
                // phase offset: the other parts start once part 0 has finished its first MFMA-bound encode stage, so
                // from then on one stream's HBM-bound segment reduce co-runs with another stream's MFMA-bound stage
                if (hipEventRecord(m->ev_fork, s0) != hipSuccess) { rc = AG_ERR_HIP; break; }
                for (int kk = 1; kk < parts; ++kk)
                    if (hipStreamWaitEvent(m->aux_stream[kk], m->ev_fork, 0) != hipSuccess) rc = AG_ERR_HIP;
                if (rc != AG_OK) break;
            }
            run_propagate(m, part[k].f, s);
            run[k].st.step = ai;
            { Timed tm(m, AG_K_ROLLOUT_STEP, s); ag_launch_rollout_step(run[k].st, s); }
        }
    for (int k = 0; k < parts && rc == AG_OK && state_final; ++k) {
        const size_t b0 = (size_t)part[k].b0;
        if (hipMemcpyAsync(state_final + b0 * H * plane, part[k].state, (size_t)part[k].B * H * plane * sizeof(float),
                           hipMemcpyDeviceToDevice, run[k].s) != hipSuccess)
            rc = AG_ERR_HIP;
    }
    // every exit path joins the auxiliary streams back into the caller's stream
    if (partitioned) {
        if (hipEventRecord(m->ev_join_mfma, sE) != hipSuccess || hipStreamWaitEvent(s0, m->ev_join_mfma, 0) != hipSuccess ||
            hipEventRecord(m->ev_join[1], sR) != hipSuccess || hipStreamWaitEvent(s0, m->ev_join[1], 0) != hipSuccess)
            rc = AG_ERR_HIP;
    } else
    for (int k = 1; k < parts; ++k)
        if (hipEventRecord(m->ev_join[k], m->aux_stream[k]) != hipSuccess || hipStreamWaitEvent(s0, m->ev_join[k], 0) != hipSuccess)
            rc = AG_ERR_HIP;
    if (rc != AG_OK) return fail(rc, "ag_rollout: a HIP runtime call failed: %s", hipGetErrorString(hipGetLastError()));
    AG_HIP(hipGetLastError());
    return AG_OK;
}

}  // extern "C"
