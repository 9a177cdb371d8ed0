// ag_mlp.hip — fused dense-MLP kernels on the gfx950 matrix cores, three arithmetics:
//   F32 : v_mfma_f32_32x32x2_f32   — exact fp32 (a k-ordered fmaf chain), 157 TFLOP/s class
//   B3  : v_mfma_f32_32x32x16_bf16 — every fp32 operand split x = hi + lo (two bf16), products
//         lo*hi + hi*lo + hi*hi accumulated in fp32 ("bf16x3"): ~2^-17 relative operand error, measured
//         1e-6..6e-6 max-abs on the reference forwards (gate 1e-4), 16/3 = 5.3x the fp32 MFMA rate.
//   H2  : v_mfma_f32_32x32x16_f16  — EDGE stack of precision mode 2 only: activations rounded to one fp16, weights split
//         hi + lo (two fp16), lo*x + hi*x ("fp16x2", struct PrecH3); streaming kernel edge_encode_kernel<PrecH3> and the
//         weight-stationary edge_encode_ws_kernel (the default: weights in registers, activations through LDS).
//
// Replaces the reference's Encoder / Propagator / ParticlePredictor stacks
// (src/dynamics/gnn/model.py:4-60) and the one-hot gathers feeding them (model.py:214-253, 283-295).
//
// Design (CDNA4-first):
//  * One wave owns 32 rows (edges or nodes).  The product is computed TRANSPOSED, D^T = W . X^T: the weight
//    matrix is the MFMA A operand (32 out-features x k), the activations are the B operand (k x 32 rows).
//    The 32x32 accumulator layout then gives lane (j = lane&31, h = lane>>5) the features
//    {32t + 8q + 4h + p} of row j — exactly the B-operand image the NEXT layer needs if its k-loop visits k
//    in that order (the k order of a dot product is free as long as A and B agree; the host packs the
//    weights to match).  So activations never leave registers between layers: bias, ReLU, the bf16 split
//    and the layer-to-layer hand-off are register-only.  No LDS round trip, no transposes.
//  * Weights stream through LDS in 20 KB chunk images (one 32-feature out-tile; bias stored as input column
//    150 against a constant-1 activation so it rides the MFMA chain), double-buffered by LDS-DMA
//    (global_load_lds_dwordx4) issued a full tile ahead, one barrier per tile.
//      F32 image: [32 out][160] floats, 16-byte XOR swizzle (col16 ^= (row>>1)&7) -> conflict-free ds_read_b128
//      B3  image: [10 k16-steps][hi|lo][64 lanes][8 bf16] fragment-major -> every ds_read_b128 is lane-linear
//  * All kernels are persistent (<= 2 workgroups per CU walk the 128-row tiles with a grid stride); the weight
//    ring keeps turning across row tiles.  256-thread workgroups, one wave per SIMD, 2 workgroups per CU.
#include "ag_common.h"
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

// max(x, 0) as ONE op: fp32 bit patterns order like int32 for x >= 0 and every negative float is a negative int, so
// relu(x) = as_float(max(as_int(x), 0)) (v_max_i32).  Through fmaxf / fmed3 the compiler adds a canonicalising
// `v_max_f32 x, x, x` per value.  (Not inline asm: the MFMA -> VALU read hazard is software-managed and the hazard
// recogniser does not look inside asm.)
__device__ __forceinline__ float relu1(float x)
{
    const int b = __float_as_int(x);
    return __int_as_float(b > 0 ? b : 0);
}
// two fp32 -> packed bf16, round-to-nearest-even (v_cvt_pk_bf16_f32), low half = a
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b)
{
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

// compile-time loop: f(std::integral_constant<int, I>) for I in [I0, N) — inline-asm immediates need constant expressions
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// Weight-fragment reads of the split-bf16 kernels, issued from inline asm with hand-counted waits.  Left to the
// compiler, the software-pipelined reads of a tile are re-serialised by the machine scheduler in most tiles (one
// register, `s_waitcnt lgkmcnt(0)` after every ds_read: each k16-step then eats a full LDS round trip).  LDS returns
// data in order, so `lgkmcnt(n)` with n = number of fragment reads issued AFTER the wanted pair is exact for them;
// compiler-issued LDS/SMEM traffic in between can only make the wait stricter.  The wait is tied to the fragment
// registers ("+v") so their MFMAs cannot be scheduled above it.
template <int OFF>
__device__ __forceinline__ void lds_read16(bf16x8 &d, unsigned lds_byte_addr)
{
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(lds_byte_addr), "n"(OFF));
}
template <int PENDING>
__device__ __forceinline__ void lds_wait_pair(bf16x8 &a, bf16x8 &b)
{
    static_assert(PENDING >= 0 && PENDING <= 15, "lgkmcnt is 4 bits");
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(PENDING));
}
__device__ __forceinline__ unsigned lds_addr_of(const void *p)
{
    return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void *)p;
}

#ifndef AG_H3_WG_PER_CU
#define AG_H3_WG_PER_CU 2    // edge_encode_kernel<PrecH3>: two activation images (fp16 + expanded residual) per Act, 40 KB LDS per workgroup
#endif

struct ChunkPipe {
    const float4 *g;   // weight stream (global), chunk k at g + k*AG_CHUNK_F4; the stream is walked cyclically
    int total;         // chunks in the stream (= chunks per row tile)
    int fetch;         // next stream chunk to fetch (wraps at total)
    int buf;           // LDS buffer holding the current chunk (0/1)
    float *lds;        // 2 * AG_CHUNK_FLOATS
    const uint32_t *scales = nullptr;   // PrecH3 only: block scales of the stream's wide units (128 dwords per chunk after the first)
};

// Asynchronous global -> LDS copy of the next weight chunk (global_load_lds_dwordx4: LDS-DMA, no VGPR staging,
// no ds_write in the wave's LDS queue).  Each wave-instruction lands 64 x 16 B at a wave-uniform LDS base (M0), so
// the chunk image is copied linearly: thread t moves float4 t + 256u, u = 0..4.
// Issued from inline asm on purpose: through the builtin, hipcc (ROCm 7.2) treats the DMA as a pending LDS write
// and puts s_waitcnt vmcnt(0) in front of the very next ds_read, i.e. it waits ~1 us for the copy at the top of
// every tile.  With asm the copy stays in flight under the tile's MFMAs and is drained by pipe_wait() right
// before the tile's barrier (cdna_hip_programming.md §5 "Pipelining across barriers").  vmcnt retires in order,
// so compiler-counted waits for its own loads can only over-wait because of these extra entries, never under-wait.
__device__ __forceinline__ void dma16(const void *gsrc, unsigned lds_byte_addr)
{
    asm volatile("s_mov_b32 m0, %0\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off"
                 :: "s"(lds_byte_addr), "v"(gsrc) : "memory", "m0");
}

__device__ __forceinline__ void pipe_dma(ChunkPipe &P, int buf)
{
    // The chunk index is laundered through an SGPR so the optimiser cannot prove the (cyclic) address sequence
    // loop-invariant: otherwise LICM hoists ~100 64-bit addresses out of the persistent loop and spills them.
    int f = P.fetch;
    asm volatile("" : "+s"(f));
    const float4 *g = P.g + (size_t)f * AG_CHUNK_F4 + threadIdx.x;
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)P.lds;
    const unsigned dst = __builtin_amdgcn_readfirstlane(base + (buf * AG_CHUNK_FLOATS + (threadIdx.x >> 6) * 256) * 4);
#pragma unroll
    for (int u = 0; u * AG_MLP_THREADS < AG_CHUNK_F4; ++u)        // 1280 float4 per chunk: 5 pieces per wave at 256 threads, 3 / 2 at 512
        if ((u + 1) * AG_MLP_THREADS <= AG_CHUNK_F4 || (int)threadIdx.x + u * AG_MLP_THREADS < AG_CHUNK_F4)   // wave-uniform (64 | 1280)
            dma16(g + AG_MLP_THREADS * u, dst + 16 * AG_MLP_THREADS * u);
    P.fetch = P.fetch + 1 == P.total ? 0 : P.fetch + 1;
}

// Drain the LDS-DMA of the next chunk before the tile's barrier: a full `vmcnt(0)`.  A counted wait that leaves the
// tile's own epilogue stores in flight (vmcnt(S)) measured the same in the split-bf16 kernels (their epilogue is deferred
// by a tile, so the stores are ~1 us old here) and 1.5 % faster in one exact-fp32 kernel, but it is only correct if a
// younger store can never retire before an older load; LLVM's own waitcnt pass does not assume that on gfx9-class
// targets (mixed load/store events make the counter "out of order"), so neither does this code.
__device__ __forceinline__ void pipe_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__device__ __forceinline__ void pipe_start(ChunkPipe &P)
{
    pipe_dma(P, 0);         // chunk 0
    pipe_wait();
    __syncthreads();
}

// ---- per-tile epilogues (run right after a 32-feature out-tile is finished, so its stores overlap the next
//      tile's MFMAs instead of piling up behind the layer) -------------------------------------------------
struct NoEpi {
    __device__ __forceinline__ void operator()(int, const f32x16 &) const {}
};
// Epilogue stores are unconditional: a row past the valid range writes into the table's padding rows (every table is
// allocated in whole row tiles), which keeps the epilogue branch-free.
struct RowStoreEpi {        // one tile of the row-major [rows][160] table; row = table + row*160 + 4h
    float *row;
    __device__ __forceinline__ void operator()(int ti, const f32x16 &v) const
    {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4 *>(row + 32 * ti + 8 * q) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
    }
};

// ---- q16: the 16-bit per-edge table of precision mode 2 (format: ag_common.h).  The pieces below are shared by the streaming kernels'
//      epilogue (RowStoreQ16Epi) and the weight-stationary kernel's epilogue pieces, so both write the same bits. ------------------------------
// largest |v| of two values against a running maximum (as a bit pattern; m >= 0).  NANSAFE: compared as unsigned integers — |x| orders
// like one, and inf / NaN sort above every finite value, so a non-finite accumulator ends up in the block exponent and raises the
// status bit (the split-bf16 edge stack has no other check).  Otherwise ONE v_maximum3_f32 with |.| source modifiers (the IEEE-754-2019
// maximum of gfx950: a NaN operand PROPAGATES, inf is kept; until r04 this was v_max3_f32, which drops a NaN — a NaN accumulator that no
// activation check had caught was then stored as 0 with status 0): both flag the same tiles and give the same maximum for finite ones.
// (Inline asm on accumulators: callers read them >= 4 MFMAs after their last write.)
template <bool NANSAFE>
__device__ __forceinline__ unsigned q16_max2(unsigned m, float a, float b)
{
    if constexpr (NANSAFE) {
        const unsigned ua = __float_as_uint(a) & 0x7fffffffu, ub = __float_as_uint(b) & 0x7fffffffu;
        return max(m, max(ua, ub));
    } else {
        unsigned r;
        asm("v_maximum3_f32 %0, %1, |%2|, |%3|" : "=v"(r) : "v"(m), "v"(a), "v"(b));
        return r;
    }
}
// block exponent of an out-tile from the lane's own maximum: the other half of the tile's rows sits in lane j + 32 (v_permlane32_swap)
__device__ __forceinline__ int q16_tile_exp(unsigned m, bool &nonfinite)
{
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 sw = __builtin_amdgcn_permlane32_swap(m, m, false, false);
    m = max(sw.x, sw.y);
    nonfinite = m >= 0x7e800000u;               // biased exponent > AG_Q16_EB_MAX (|v| >= 2^126), inf or NaN: the clamped scale would saturate the tile
    const int eb = (int)(m >> 23);
    return eb < AG_Q16_EB_MIN ? AG_Q16_EB_MIN : (eb > AG_Q16_EB_MAX ? AG_Q16_EB_MAX : eb);
}
__device__ __forceinline__ int q16_inv_scale(int eb) { return 126 - eb; }          // the tile's values are scaled by 2^(126 - eb)
// two values -> packed snorm16 (round to nearest).  The scaling is v_ldexp_f32, one per value, NOT one v_pk_mul_f32 per pair: the packed fp32
// instructions take ~39 cycles beside a busy matrix pipe against ~10 for an ordinary VALU instruction (tools/ubench/valu_beside_mfma.hip), and
// this runs in the shadow of MFMAs in every edge kernel.  (ldexp by 2^k and the multiplication by 2^k round identically: same bits.)
__device__ __forceinline__ unsigned q16_pack(float a, float b, int inv)
{
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, (s16x2)__builtin_amdgcn_cvt_pknorm_i16(__builtin_ldexpf(a, inv), __builtin_ldexpf(b, inv)));
}
// stores of one out-tile of lane (j, h): `row` = table + e * 320 bytes; the lane's 32 bytes start at 64 ti + 32 h.  In tile 4 the last eight
// bytes of the lane's chunk are padding that holds exponent bytes written by OTHER lanes / waves: they are not touched.
__device__ __forceinline__ void q16_store_half(unsigned char *row, int ti, int h, int s, const unsigned (&w)[4])
{
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    unsigned char *p = row + 64 * ti + 32 * h + 16 * s;
    if (ti == 4 && s == 1) *reinterpret_cast<u32x2 *>(p) = u32x2{w[0], w[1]};       // (plain stores: `nt` here costs the edge encoder 4 %, ag_common.h)
    else *reinterpret_cast<u32x4 *>(p) = u32x4{w[0], w[1], w[2], w[3]};
}
__device__ __forceinline__ void q16_store_exp(unsigned char *row, int ti, int h, int eb) { row[ag_q16_exp_byte_offset(ti, h)] = (unsigned char)eb; }

struct RowStoreQ16Epi {     // Eterm as q16 (precision mode 2).  The maximum runs on compiler-visible integer instructions here: `v` comes straight
                            // from builtin MFMAs, and an inline-asm reader gets no MFMA -> VALU wait states from the compiler (the asm version,
                            // hoisted above the tile barrier, read accumulators the last scaled MFMA had not written yet: a block exponent off by
                            // one in 1e-4 of the tiles).  Same result as the float maximum of the weight-stationary kernel for finite tiles.
    unsigned char *row;     // table + e * 320
    int h;
    int *status = nullptr;  // model status word: bit 0 is raised when a tile holds a non-finite value (or one beyond 2^127)
    __device__ __forceinline__ void operator()(int ti, const f32x16 &v) const
    {
        unsigned m = 0;
#pragma unroll
        for (int r = 0; r < 16; r += 2) m = q16_max2<true>(m, v[r], v[r + 1]);
        bool bad;
        const int eb = q16_tile_exp(m, bad);
        const int inv = q16_inv_scale(eb);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            unsigned w[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) w[k] = q16_pack(v[8 * s + 2 * k], v[8 * s + 2 * k + 1], inv);
            q16_store_half(row, ti, h, s, w);
        }
        q16_store_exp(row, ti, h, eb);
        if (bad && status) atomicOr(status, 1);     // AG_STATUS_NONFINITE
    }
};
struct PackStoreEpi {       // same for the fragment-image tables (h, Pn); blk_lane = table + block*5120 + h*128 + j*4
    float *blk_lane;
    __device__ __forceinline__ void operator()(int ti, const f32x16 &v) const
    {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            ag_st_nt(reinterpret_cast<float4 *>(blk_lane + ((ti * 4 + q) * 2) * 128), make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]));
    }
};

// ---- accumulator initialisers ------------------------------------------------------------------------------
struct ZeroInit {
    __device__ __forceinline__ f32x16 operator()(int /*ti*/) const
    {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        return acc;
    }
};
struct ResidInit {  // accumulator := Pn + h, i.e. W_pp[:, :F].enc + b_pp + residual (model.py:36-40,299-301).  Each source is either a
                    // packed (fragment-image) table, pointer already offset to this wave's 32-row block and this lane's (h, j), or — node
                    // de-duplication — a row-major row of a compact table, pointer = row + 4h.
                    // The loads run ahead of their use (r05; loaded where a tile needs them, each of the layer's five tiles waited a full memory
                    // latency two MFMAs into its chain): prefetch() issues h of ALL five out-tiles and Pn of tile 0 at the row tile's top (h's
                    // registers are the ones the layer's output image occupies tile by tile: disjoint live ranges), operator()(ti) adds what
                    // has arrived and issues Pn of tile ti + 1.  Measured -2.6 % (node_update is bound by bytes through L2, not by these
                    // latencies: docs/NEGATIVE_RESULTS.md R5.2).
    const float *pn, *hh;
    bool pn_rowmajor, h_rowmajor;
    mutable f32x16 hraw[AG_NT];       // h of all five out-tiles, issued at the row tile's top
    mutable f32x16 pnext;             // Pn of the NEXT out-tile
    template <bool NT = false>
    __device__ __forceinline__ static void load_tile(const float *p, bool rowmajor, int ti, f32x16 &d)
    {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            // ONE unconditional 16-byte load per quad (address by select; a load under `if (rowmajor)` is split into predicated dword loads)
            const int off = ((ti * 4 + q) * 2) * 128, offr = 32 * ti + 8 * q;
            const bool pad = offr >= 152;                             // row-major rows: columns >= 152 + 4h are not read (zero)
            const float4 *src = reinterpret_cast<const float4 *>(p + (rowmajor ? (pad ? 0 : offr) : off));
            const float4 a = NT ? ag_ld_nt(src) : *src;
            d[4 * q + 0] = a.x; d[4 * q + 1] = a.y; d[4 * q + 2] = a.z; d[4 * q + 3] = a.w;      // (padding quad: zeroed where the tile is consumed, zero_pad)
        }
    }
    __device__ __forceinline__ static void zero_pad(bool rowmajor, int ti, f32x16 &d)      // a VALU op on a loaded value waits for the load: not in load_tile
    {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (32 * ti + 8 * q >= 152 && rowmajor) { d[4 * q + 0] = 0.f; d[4 * q + 1] = 0.f; d[4 * q + 2] = 0.f; d[4 * q + 3] = 0.f; }
    }
    __device__ __forceinline__ void prefetch() const
    {
#pragma unroll
        for (int t = 0; t < AG_NT; ++t) load_tile<true>(hh, h_rowmajor, t, hraw[t]);
        load_tile(pn, pn_rowmajor, 0, pnext);
    }
    __device__ __forceinline__ f32x16 operator()(int ti) const
    {
        f32x16 acc;
        // opaque: without it the scheduler pulls this add up into the PREVIOUS out-tile, right behind the load (to free the register), where it
        // waits for the load — and, vmcnt being in order, for whatever else was issued before it
        asm volatile("" : "+v"(pnext));
        zero_pad(pn_rowmajor, ti, pnext);
        zero_pad(h_rowmajor, ti, hraw[ti]);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = pnext[r] + hraw[ti][r];
        if (ti + 1 < AG_NT) load_tile(pn, pn_rowmajor, ti + 1, pnext);
        return acc;
    }
};

// =====================================================================================================
// Precision policies.  Both expose
//   Act                      register image of a 160-wide activation row block (the MFMA B operands)
//   from_tiles(f32 tiles)    build an Act from fp32 accumulator tiles
//   layer<K,NT,RELU,BIAS>    out-tile loop: acc = init(ti); acc += W_chunk . in; relu; epi(ti, acc);
//                            the finished tile is handed to `sink(ti, acc)` (next layer's Act, or raw tiles)
// K = number of input columns visited (k >= K is zero padding).  With BIAS the layer's bias is input column K of
// the packed weights and the matching activation "feature K" is forced to 1.0, so the bias rides the MFMA chain
// (columns >= AG_F of every activation table are padding, nothing else reads them).
// =====================================================================================================
struct PrecF32 {
    struct Act { f32x16 t[AG_NT]; };
    __device__ __forceinline__ static void set_tile(Act &a, int ti, const f32x16 &v) { a.t[ti] = v; }

    template <int K, int NT, bool RELU, bool BIAS, class Init, class Epi, class Sink>
    __device__ __forceinline__ static void layer(ChunkPipe &P, const Act &in, const Init &init, const Epi &epi, Sink &&sink)
    {
        constexpr int KE = K + (BIAS ? 1 : 0);
        constexpr int PT = (KE + 7) / 8;      // quads (= 4 k-steps = one ds_read_b128 per lane) per tile
        const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
        // per-lane fragment addresses: row i, 16-byte column (8t + 2q + h) ^ ((i >> 1) & 7) (host pre-swizzled)
        const int sw = (i >> 1) & 7;
        int qoff[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) qoff[q] = i * AG_WSTRIDE + 4 * ((2 * q + h) ^ sw);
#pragma unroll
        for (int ti = 0; ti < NT; ++ti) {
            const float *buf = P.lds + P.buf * AG_CHUNK_FLOATS;
            f32x16 acc = init(ti);      // BEFORE the chunk DMA: vmcnt retires in order, so a wait for a load issued behind the DMA waits for the DMA too
            pipe_dma(P, P.buf ^ 1);
#pragma unroll
            for (int m = 0; m < PT; ++m) {
                const int t = m / 4, q = m % 4;
                const float4 w = *reinterpret_cast<const float4 *>(buf + qoff[q] + 32 * t);
                const float wv[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const int k0 = 32 * t + 8 * q + p;        // column seen by the h = 0 half (h = 1: k0 + 4)
                    if (k0 < KE) {
                        float x = in.t[t][4 * q + p];
                        if (BIAS && (k0 == K || k0 + 4 == K)) x = (h == (k0 == K ? 0 : 1)) ? 1.0f : x;
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[p], x, acc, 0, 0, 0);
                    }
                }
            }
            if (RELU) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = relu1(acc[r]);
            }
            epi(ti, acc);
            sink(ti, acc);
            pipe_wait();
            __syncthreads();
            P.buf ^= 1;
        }
    }

    // First layers (fan-in <= 24): all five out-tiles are packed into ONE chunk ([5][32 rows][32 floats], same
    // 16-byte swizzle), so the layer costs one DMA and one barrier instead of five.
    template <int K, class Sink>
    __device__ __forceinline__ static void layer_first(ChunkPipe &P, const Act &in, Sink &&sink)
    {
        constexpr int PT = (K + 7) / 8;
        static_assert(K <= 32, "compact first layer");
        const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
        const int sw = (i >> 1) & 7;
        const float *buf = P.lds + P.buf * AG_CHUNK_FLOATS;
        pipe_dma(P, P.buf ^ 1);
#pragma unroll
        for (int ti = 0; ti < AG_NT; ++ti) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
            for (int q = 0; q < PT; ++q) {
                const float4 w = *reinterpret_cast<const float4 *>(buf + ti * 1024 + i * 32 + 4 * ((2 * q + h) ^ sw));
                const float wv[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                for (int p = 0; p < 4; ++p)
                    if (8 * q + p < K) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[p], in.t[0][4 * q + p], acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = relu1(acc[r]);
            sink(ti, acc);
        }
        pipe_wait();
        __syncthreads();
        P.buf ^= 1;
    }
};



struct PrecB3 {
    // step u = 2t + s covers features [16u, 16u+16): lane (j,h) slot e holds feature 16u + 8(e>>2) + 4h + (e&3),
    // which is accumulator register 8s + e of out-tile t — so a finished tile converts in place, no shuffles.
    struct Act { bf16x8 hi[2 * AG_NT], lo[2 * AG_NT]; };
    // hi = bf16(x) (RNE), lo = bf16(x - hi), two values per packed convert: 6 VALU ops per value pair.  (Written on pairs
    // with explicit converts: from per-element `(__bf16)x` the compiler emitted ~3x as many ops, and the relu+split
    // epilogues were co-limiting the kernels with the MFMAs.)
    __device__ __forceinline__ static void set_tile(Act &a, int ti, const f32x16 &v)
    {
#pragma unroll
        for (int s = 0; s < 2; ++s) set_half(a, ti, s, v);
    }
    // k16-step 2ti + s of the next layer's operand = accumulator registers 8s..8s+7 of out-tile ti
    __device__ __forceinline__ static void set_half(Act &a, int ti, int s, const f32x16 &v)
    {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        u32x4 H, L;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float x0 = v[8 * s + 2 * w], x1 = v[8 * s + 2 * w + 1];
            const unsigned hp = cvt_pk_bf16(x0, x1);
            const float h0 = __uint_as_float(hp << 16), h1 = __uint_as_float(hp & 0xffff0000u);
            H[w] = hp;
            L[w] = cvt_pk_bf16(x0 - h0, x1 - h1);
        }
        a.hi[2 * ti + s] = __builtin_bit_cast(bf16x8, H);
        a.lo[2 * ti + s] = __builtin_bit_cast(bf16x8, L);
    }

    template <int K, int NT, bool RELU, bool BIAS, class Init, class Epi, class Sink>
    __device__ __forceinline__ static void layer(ChunkPipe &P, const Act &in, const Init &init, const Epi &epi, Sink &&sink)
    {
        constexpr int KE = K + (BIAS ? 1 : 0);
        constexpr int NU = (KE + 15) / 16;    // k16-steps per tile
        constexpr int PF = 2;                 // weight fragments are read PF steps ahead of their MFMAs
        const int lane = threadIdx.x & 63, h = lane >> 5;
        // The epilogue of tile ti (ReLU, hi/lo split for the next layer, stores) is DEFERRED into tile ti+1, behind
        // that tile's barrier and fragment prefetch: its ~100 VALU ops then issue in the shadow of tile ti+1's MFMAs
        // instead of sitting between the last MFMA of a tile and the barrier.
        f32x16 prev;
        auto finish = [&](int ti, f32x16 &v) {
            if (RELU) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = relu1(v[r]);
            }
            epi(ti, v);
            sink(ti, v);
        };
#pragma unroll
        for (int ti = 0; ti < NT; ++ti) {
            const unsigned la = lds_addr_of(P.lds) + (unsigned)(P.buf * AG_CHUNK_FLOATS * 4 + lane * 16);
            f32x16 acc = init(ti);      // BEFORE the chunk DMA: vmcnt retires in order, so a wait for a load issued behind the DMA waits for the DMA too
            pipe_dma(P, P.buf ^ 1);
            bf16x8 wq[PF + 1][2];
            static_for<0, (PF < NU ? PF : NU)>([&](auto U) {
                constexpr int u = decltype(U)::value;
                lds_read16<(2 * u) * 1024>(wq[u][0], la);
                lds_read16<(2 * u + 1) * 1024>(wq[u][1], la);
            });
            if (ti > 0) finish(ti - 1, prev);
            static_for<0, NU>([&](auto U) {
                constexpr int u = decltype(U)::value;
                if constexpr (u + PF < NU) {
                    lds_read16<(2 * (u + PF)) * 1024>(wq[(u + PF) % (PF + 1)][0], la);
                    lds_read16<(2 * (u + PF) + 1) * 1024>(wq[(u + PF) % (PF + 1)][1], la);
                }
                constexpr int ahead = (NU - 1 - u) < PF ? (NU - 1 - u) : PF;     // k16-steps whose reads were issued after step u's
                lds_wait_pair<2 * ahead>(wq[u % (PF + 1)][0], wq[u % (PF + 1)][1]);
                const bf16x8 wh = wq[u % (PF + 1)][0], wl = wq[u % (PF + 1)][1];
                bf16x8 xh = in.hi[u], xl = in.lo[u];
                if constexpr (BIAS && K / 16 == u) {        // feature K = 16u + 8(e>>2) + 4h + (e&3)
                    constexpr int o = K % 16, e = (o >> 3) * 4 + (o & 3), hb = (o >> 2) & 1;
                    if (h == hb) { xh[e] = (__bf16)1.0f; xl[e] = (__bf16)0.0f; }
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, xh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xh, acc, 0, 0, 0);
            });
            prev = acc;
            pipe_wait();
            __syncthreads();
            P.buf ^= 1;
        }
        finish(NT - 1, prev);
    }

    // First layers (fan-in <= 32): the NU k16-steps of all five out-tiles are packed into ONE chunk
    // ([5 tiles][NU][hi|lo][64 lanes][8 bf16]), one DMA and one barrier for the whole layer.
    template <int K, class Sink>
    __device__ __forceinline__ static void layer_first(ChunkPipe &P, const Act &in, Sink &&sink)
    {
        constexpr int NU = (K + 15) / 16;
        static_assert(NU <= 2, "compact first layer");
        const int lane = threadIdx.x & 63;
        const unsigned la = lds_addr_of(P.lds) + (unsigned)(P.buf * AG_CHUNK_FLOATS * 4 + lane * 16);
        pipe_dma(P, P.buf ^ 1);
        // fragments of out-tile ti+1 are read while tile ti's MFMAs run (<= 4*NU reads in flight)
        bf16x8 wq[2][NU][2];
        static_for<0, NU>([&](auto U) {
            constexpr int u = decltype(U)::value;
            lds_read16<(u * 2) * 1024>(wq[0][u][0], la);
            lds_read16<(u * 2 + 1) * 1024>(wq[0][u][1], la);
        });
        static_for<0, AG_NT>([&](auto T) {
            constexpr int ti = decltype(T)::value;
            if constexpr (ti + 1 < AG_NT)
                static_for<0, NU>([&](auto U) {
                    constexpr int u = decltype(U)::value;
                    lds_read16<(((ti + 1) * NU + u) * 2) * 1024>(wq[(ti + 1) & 1][u][0], la);
                    lds_read16<(((ti + 1) * NU + u) * 2 + 1) * 1024>(wq[(ti + 1) & 1][u][1], la);
                });
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            static_for<0, NU>([&](auto U) {
                constexpr int u = decltype(U)::value;
                constexpr int later = (NU - 1 - u) + (ti + 1 < AG_NT ? NU : 0);      // pairs issued after this one
                lds_wait_pair<2 * later>(wq[ti & 1][u][0], wq[ti & 1][u][1]);
                const bf16x8 wh = wq[ti & 1][u][0], wl = wq[ti & 1][u][1];
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, in.hi[u], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, in.lo[u], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, in.hi[u], acc, 0, 0, 0);
            });
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = relu1(acc[r]);
            sink(ti, acc);
        });
        pipe_wait();
        __syncthreads();
        P.buf ^= 1;
    }

};


// ---------------------------------------------------------------------------------------------------------------------
// H3: the arithmetic of the EDGE stack in precision mode 2 — fp16 with byte-sized corrections on the block-scaled fp8 MFMA.
//   weight      W = hi + lo:  hi = fp16(W);  the corrections use e4m3(lo / s_lo) and e4m3(hi / s_hi) with one power-of-two scale per
//               (output row, 32-column input tile) (packed on the host: ag_api.hip pack_layer_h3)
//   activation  x = x16 + r:  x16 = fp16(x) (RNE);  r8 = e5m2(x - x16) (RNE): ONE byte per value;  x8 = the top byte of x16 (= e5m2, truncated)
//   per 32-column input tile t of an out-tile:
//       acc += hi . x16   two v_mfma_f32_32x32x16_f16 (k16-steps 2t, 2t + 1)
//       acc += s_lo (lo8 . x8) + s_hi (hi8 . r8)   ONE v_mfma_scale_f32_32x32x64_f8f6f4 (A e4m3, B e5m2; its two K blocks are the two terms)
//   i.e. W.x = hi.x16 + lo.x16 + hi.r (+ lo.r ~ 2^-23, dropped) with the two 2^-11-sized terms at 3-4 significant bits.
// History: r02-r03 ran hi.x16 + lo.x16 on twenty fp16 MFMAs per out-tile.  On weights trained by the reference the fp16 rounding of
// the ACTIVATIONS (2^-12 relative, every layer contributing alike) then costs 2-5e-5 of the 1e-4 gate and grows with the predicted motion
// (1.4e-4 at |motion| 0.2 in tools/fuzz_parity.py).  A third fp16 product hi.r fixes that (float64 emulation on the fuzz cases,
// tools/scheme_err.py: 4.9e-5 -> 5.6e-6 with the q16 table) but the chip is POWER-limited under MFMA load (1 630 TFLOP/s of fp16 MFMA
// sustained): thirty MFMAs per out-tile measured 0.91 ms for the edge encoder instead of 0.55.  The block-scaled instruction does K = 64 for
// 1.25 fp16-MFMA-times (tools/ubench/mx_mfma.hip): ten fp16 MFMAs + five scaled ones = 16.25 MFMA-times per out-tile, LESS than the r03
// scheme's 20, for the same 5.6e-6 -> 6.0e-6 emulated deviation.
// RANGE: a hidden activation beyond +-65504 converts to +inf.  Every epilogue keeps the largest fp16 bit pattern it produced
// (`bad`, one packed integer maximum per value pair) and raises status bit 0 when it reaches 0x7c00 (inf / NaN): the overflow is
// reported WHERE it happens, whatever later layers make of it.  For checkpoints with larger activations use precision 1
// (split-bf16, fp32 range).  Measured head-room: the trained goldens rescaled to 64x larger edge-stack activations
// (tests/golden/*act64*, tools/gen_trained.py) still match within the mode's tolerance with status 0.
// FIRST layer of the edge stack: its 17 inputs + bias column use 18 of the 32 K slots of two k16-steps.  Twelve of the inputs are
// position / velocity differences of any size (a tool joined to every cloth particle by connect_tools_all sits metres away: |x| ~ 50
// rounds to fp16 with an error of 0.01).  Spare slots 18..29 carry the fp16 rounding residuals of inputs 5..16 against the same weight
// columns (ag_api.hip pack_first_layer), so the first layer sees them to 2^-22 on two plain fp16 products (split-fp16 weights).
#define AG_EDGE_LO_SLOT0 (AG_EDGE_IN + 1)       // first residual slot
#define AG_EDGE_LO_FEAT0 (2 * AG_ATTR + 1)      // first input with a residual: the state differences (model.py:241-253)
#define AG_EDGE_LO_COUNT (AG_EDGE_IN - AG_EDGE_LO_FEAT0)
static_assert(AG_EDGE_LO_SLOT0 == 18 && AG_EDGE_LO_FEAT0 == 5 && AG_EDGE_LO_COUNT == 12, "edge_encode_ws_kernel builds slots 18..29 by hand");
__device__ __forceinline__ float f16_residual(float v) { return v - (float)(_Float16)v; }

typedef unsigned h3_u32x4 __attribute__((ext_vector_type(4)));
typedef int h3_i32x8 __attribute__((ext_vector_type(8)));
#define AG_H3_HI_BYTES 10240      // a wide unit's chunk image: [0, 10240) fp16 hi fragments, [10240, 20480) the scaled MFMA's A operands
// two (already ReLU'd) fp32 activations -> the packed fp16 pair and their two residual bytes (into the low or high half of R)
template <bool SIGNED = false>      // SIGNED: the values may be negative (raw first-layer inputs): the range check then ignores the sign bits
__device__ __forceinline__ unsigned h3_pair(float x0, float x1, int &R, bool hi_word, unsigned &bad)
{
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    const f32x2 x = {x0, x1};
    const f16x2 hx = __builtin_convertvector(x, f16x2);
    const unsigned H = __builtin_bit_cast(unsigned, hx);
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    bad = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(u16x2, bad), __builtin_bit_cast(u16x2, SIGNED ? (H & 0x7fff7fffu) : H)));
    // r = x - float(x16), exact: one v_fma_mix_f32 per value (fp16 source read in place; the compiler's own selection is convert + subtract)
    float r0, r1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(H), "v"(x0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(H), "v"(x1));
    R = hi_word ? __builtin_amdgcn_cvt_pk_bf8_f32(r0, r1, R, true) : __builtin_amdgcn_cvt_pk_bf8_f32(r0, r1, R, false);
    return H;
}
// x8 of an input tile: the top bytes of the sixteen fp16 values of k16-steps 2t (xa) and 2t + 1 (xb), in element order
__device__ __forceinline__ h3_u32x4 h3_top_bytes(const h3_u32x4 &xa, const h3_u32x4 &xb)
{
    h3_u32x4 r;
    r[0] = __builtin_amdgcn_perm(xa[1], xa[0], 0x07050301u);
    r[1] = __builtin_amdgcn_perm(xa[3], xa[2], 0x07050301u);
    r[2] = __builtin_amdgcn_perm(xb[1], xb[0], 0x07050301u);
    r[3] = __builtin_amdgcn_perm(xb[3], xb[2], 0x07050301u);
    return r;
}
__device__ __forceinline__ h3_i32x8 h3_b_operand(const h3_u32x4 &x8, const h3_u32x4 &r8)
{
    return h3_i32x8{(int)x8[0], (int)x8[1], (int)x8[2], (int)x8[3], (int)r8[0], (int)r8[1], (int)r8[2], (int)r8[3]};
}
// status bit 0 when a lane produced an fp16 inf / NaN (bit patterns >= 0x7c00 in either half-word of `bad`)
__device__ __forceinline__ void h3_report(unsigned bad, int *status)
{
    if (((bad + 0x04000400u) & 0x80008000u) && status) atomicOr(status, 1);     // AG_STATUS_NONFINITE
}

struct PrecH3 {
    struct Act { f16x8 v[2 * AG_NT]; h3_u32x4 r8[AG_NT]; unsigned bad = 0; };      // fp16 values by k16-step, residual bytes by input tile
    __device__ __forceinline__ static void set_tile(Act &a, int ti, const f32x16 &v)
    {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            h3_u32x4 H;
            int R[2] = {0, 0};
#pragma unroll
            for (int w = 0; w < 4; ++w) H[w] = h3_pair<true>(v[8 * s + 2 * w], v[8 * s + 2 * w + 1], R[w >> 1], (w & 1) != 0, a.bad);
            a.v[2 * ti + s] = __builtin_bit_cast(f16x8, H);
            a.r8[ti][2 * s] = (unsigned)R[0];
            a.r8[ti][2 * s + 1] = (unsigned)R[1];
        }
    }

    // tile loop as PrecB3::layer (weight ring, deferred epilogue); per input tile two fp16 MFMAs and one block-scaled fp8 MFMA
    template <int K, int NT, bool RELU, bool BIAS, class Init, class Epi, class Sink>
    __device__ __forceinline__ static void layer(ChunkPipe &P, const Act &in, const Init &init, const Epi &epi, Sink &&sink)
    {
        static_assert(K == AG_F && BIAS, "the scaled-MFMA images are packed for 150 inputs + the bias column");
        const int lane = threadIdx.x & 63, h = lane >> 5;
        f32x16 prev;
        auto finish = [&](int ti, f32x16 &v) {
            if (RELU) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = relu1(v[r]);
            }
            epi(ti, v);
            sink(ti, v);
        };
#pragma unroll
        for (int ti = 0; ti < NT; ++ti) {
            const unsigned la = lds_addr_of(P.lds) + (unsigned)(P.buf * AG_CHUNK_FLOATS * 4 + lane * 16);
            const unsigned lm = lds_addr_of(P.lds) + (unsigned)(P.buf * AG_CHUNK_FLOATS * 4 + AG_H3_HI_BYTES + lane * 32);
            const int cur = P.fetch == 0 ? P.total - 1 : P.fetch - 1;         // stream chunk in P.buf (chunk 0 is the first layer)
            const unsigned sc0 = P.scales[(cur - 1) * 128 + lane], sc1 = P.scales[(cur - 1) * 128 + 64 + lane];
            pipe_dma(P, P.buf ^ 1);
            f32x16 acc = init(ti);
            bf16x8 wq[2][4];          // per input tile: hi fragments of its two k16-steps, the scaled operand's two 16-byte halves
            auto issue = [&](auto T) {
                constexpr int t = decltype(T)::value;
                lds_read16<(2 * t) * 1024>(wq[t & 1][0], la);
                lds_read16<(2 * t + 1) * 1024>(wq[t & 1][1], la);
                lds_read16<t * 2048>(wq[t & 1][2], lm);
                lds_read16<t * 2048 + 16>(wq[t & 1][3], lm);
            };
            issue(std::integral_constant<int, 0>{});
            if (ti > 0) finish(ti - 1, prev);
            static_for<0, AG_NT>([&](auto T) {
                constexpr int t = decltype(T)::value;
                if constexpr (t + 1 < AG_NT) issue(std::integral_constant<int, t + 1>{});
                lds_wait_pair<(t + 1 < AG_NT ? 4 : 0)>(wq[t & 1][0], wq[t & 1][1]);
                lds_wait_pair<(t + 1 < AG_NT ? 4 : 0)>(wq[t & 1][2], wq[t & 1][3]);
                f16x8 xa = in.v[2 * t], xb = in.v[2 * t + 1];
                if constexpr (K / 32 == t) {        // bias column: feature K = 16u + 8(e>>2) + 4h + (e&3) := 1.0 (its residual byte is 0: the feature is padding)
                    constexpr int u = K / 16, o = K % 16, e = (o >> 3) * 4 + (o & 3), hb = (o >> 2) & 1;
                    if (h == hb) { if constexpr (u & 1) xb[e] = (_Float16)1.0f; else xa[e] = (_Float16)1.0f; }
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wq[t & 1][0]), xa, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wq[t & 1][1]), xb, acc, 0, 0, 0);
                const h3_u32x4 a0 = __builtin_bit_cast(h3_u32x4, wq[t & 1][2]), a1 = __builtin_bit_cast(h3_u32x4, wq[t & 1][3]);
                const h3_i32x8 A = h3_b_operand(a0, a1);
                const h3_i32x8 B = h3_b_operand(h3_top_bytes(__builtin_bit_cast(h3_u32x4, xa), __builtin_bit_cast(h3_u32x4, xb)), in.r8[t]);
                acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc, 0, 1, t & 3, (int)(t < 4 ? sc0 : sc1), 0, 0x7f7f7f7f);
            });
            prev = acc;
            pipe_wait();
            __syncthreads();
            P.buf ^= 1;
        }
        finish(NT - 1, prev);
    }

    // narrow first layer: two plain fp16 products on split-fp16 weights (its inputs carry their own residuals in spare K slots, see above)
    template <int K, class Sink>
    __device__ __forceinline__ static void layer_first(ChunkPipe &P, const Act &in, Sink &&sink)
    {
        constexpr int NU = (K + 15) / 16;
        static_assert(NU <= 2, "compact first layer");
        const int lane = threadIdx.x & 63;
        const unsigned la = lds_addr_of(P.lds) + (unsigned)(P.buf * AG_CHUNK_FLOATS * 4 + lane * 16);
        pipe_dma(P, P.buf ^ 1);
        bf16x8 wq[2][NU][2];
        static_for<0, NU>([&](auto U) {
            constexpr int u = decltype(U)::value;
            lds_read16<(u * 2) * 1024>(wq[0][u][0], la);
            lds_read16<(u * 2 + 1) * 1024>(wq[0][u][1], la);
        });
        static_for<0, AG_NT>([&](auto T) {
            constexpr int ti = decltype(T)::value;
            if constexpr (ti + 1 < AG_NT)
                static_for<0, NU>([&](auto U) {
                    constexpr int u = decltype(U)::value;
                    lds_read16<(((ti + 1) * NU + u) * 2) * 1024>(wq[(ti + 1) & 1][u][0], la);
                    lds_read16<(((ti + 1) * NU + u) * 2 + 1) * 1024>(wq[(ti + 1) & 1][u][1], la);
                });
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            static_for<0, NU>([&](auto U) {
                constexpr int u = decltype(U)::value;
                constexpr int later = (NU - 1 - u) + (ti + 1 < AG_NT ? NU : 0);
                lds_wait_pair<2 * later>(wq[ti & 1][u][0], wq[ti & 1][u][1]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wq[ti & 1][u][1]), in.v[u], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wq[ti & 1][u][0]), in.v[u], acc, 0, 0, 0);
            });
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = relu1(acc[r]);
            sink(ti, acc);
        });
        pipe_wait();
        __syncthreads();
        P.buf ^= 1;
    }
};


// layer -> next Act
template <class Prec, int K, bool RELU, bool BIAS, class Init, class Epi = NoEpi>
__device__ __forceinline__ void dense(ChunkPipe &P, const typename Prec::Act &in, typename Prec::Act &out, const Init &init,
                                      const Epi &epi = Epi{})
{
    Prec::template layer<K, AG_NT, RELU, BIAS>(P, in, init, epi, [&](int ti, const f32x16 &v) { Prec::set_tile(out, ti, v); });
}
// narrow first layer (ReLU, bias column already in the input features) -> next Act
template <class Prec, int K>
__device__ __forceinline__ void dense_first(ChunkPipe &P, const typename Prec::Act &in, typename Prec::Act &out)
{
    Prec::template layer_first<K>(P, in, [&](int ti, const f32x16 &v) { Prec::set_tile(out, ti, v); });
}
// layer whose output is only stored (by `epi`)
template <class Prec, int K, bool RELU, bool BIAS, class Init, class Epi>
__device__ __forceinline__ void dense_store(ChunkPipe &P, const typename Prec::Act &in, const Init &init, const Epi &epi)
{
    Prec::template layer<K, AG_NT, RELU, BIAS>(P, in, init, epi, [](int, const f32x16 &) {});
}

__device__ __forceinline__ void load_rowmajor(const float *row, f32x16 (&v)[AG_NT], int h)
{
#pragma unroll
    for (int t = 0; t < AG_NT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (32 * t + 8 * q < 152) x = *reinterpret_cast<const float4 *>(row + 32 * t + 8 * q + 4 * h);
            v[t][4 * q] = x.x; v[t][4 * q + 1] = x.y; v[t][4 * q + 2] = x.z; v[t][4 * q + 3] = x.w;
        }
}

// `agg` as q16 rows (option "agg_q16", ag_common.h): lane (j, h) of an MFMA kernel needs, per out-tile t, the features 32t + 8q + 4h + p — the sixteen
// CONTIGUOUS 16-bit positions 32t + 16h .. + 15 of row j (two 16-byte loads instead of four) — and the row's exponent bytes 280..284 (one 8-byte load).
struct AggRowQ16 { int4 v[AG_NT][2]; uint2 ex; };
__device__ __forceinline__ void load_rowmajor_q16(const unsigned char *row, AggRowQ16 &r, int h)
{
#pragma unroll
    for (int t = 0; t < AG_NT; ++t) {
        r.v[t][0] = *reinterpret_cast<const int4 *>(row + 64 * t + 32 * h);
        r.v[t][1] = *reinterpret_cast<const int4 *>(row + 64 * t + 32 * h + 16);
    }
    r.ex = *reinterpret_cast<const uint2 *>(row + 280);
}
__device__ __forceinline__ float agg_q16_tile_scale(const uint2 &ex, int t) { return ag_q16u_scale(t < 4 ? (int)((ex.x >> (8 * t)) & 0xffu) : (int)(ex.y & 0xffu)); }
__device__ __forceinline__ void agg_q16_decode8(const int4 &w, float sc, float (&x)[8]) { ag_q16u_decode8(w, sc, x); }
__device__ __forceinline__ void agg_q16_tile(const AggRowQ16 &r, int t, int h, f32x16 &v)
{
    const float sc = agg_q16_tile_scale(r.ex, t);
    float lo[8], hi[8];
    agg_q16_decode8(r.v[t][0], sc, lo);
    agg_q16_decode8(r.v[t][1], sc, hi);
#pragma unroll
    for (int k = 0; k < 8; ++k) { v[k] = lo[k]; v[8 + k] = hi[k]; }
    if (t == 4) {      // features 150..159 do not exist: their positions hold the exponent bytes (and zeros)
        if (h) { v[10] = 0.0f; v[11] = 0.0f; }
        v[12] = 0.0f; v[13] = 0.0f; v[14] = 0.0f; v[15] = 0.0f;
    }
}

#define AG_LDS_DECL __shared__ __attribute__((aligned(16))) float lds[2 * AG_CHUNK_FLOATS]; __shared__ int s_next_tile[2];

// Row tiles are CLAIMED from a per-launch counter instead of walked with a static grid stride: the two workgroups that
// share a CU do not progress at the same rate (issue arbitration is oldest-first, so the workgroup launched second runs
// ~40 % slower per row tile, s_memtime trace), and with a static split the early finishers leave their CU half empty for
// the tail.  The atomic is issued together with the row tile's first loads (whose wait it shares) and the claimed index
// travels through LDS under the tile's own barriers, so the queue costs no extra round trip or barrier.
struct TileQueue {
    int *ctr, *slot;
    int tile, par, claimed;
    __device__ __forceinline__ TileQueue(int *c, int *s) : ctr(c), slot(s), tile(blockIdx.x), par(0), claimed(0) {}
    __device__ __forceinline__ void claim() { if (threadIdx.x == 0) claimed = ctr ? (int)gridDim.x + atomicAdd(ctr, 1) : tile + (int)gridDim.x; }
    __device__ __forceinline__ void publish() { if (threadIdx.x == 0) slot[par] = claimed; }   // >= 1 barrier before next()
    __device__ __forceinline__ void next() { tile = slot[par]; par ^= 1; }                      // after the row tile's last barrier
};

template <class Prec> __device__ __forceinline__ const float4 *pick(const float4 *f32, const float4 *b3);
template <> __device__ __forceinline__ const float4 *pick<PrecF32>(const float4 *f32, const float4 *) { return f32; }
template <> __device__ __forceinline__ const float4 *pick<PrecB3>(const float4 *, const float4 *b3) { return b3; }
template <> __device__ __forceinline__ const float4 *pick<PrecH3>(const float4 *, const float4 *b3) { return b3; }   // (the caller passes the fp16 image)

// ---------------------------------------------------------------------------------------------
// Node encoder + pstep-invariant node terms.
//   enc = Encoder([attrs | phys | action])                      model.py:168-195, 268
//   h0  = enc                                                     model.py:269
//   Pn  = W_pp[:, :F] . enc + b_pp     (first column block of particle_propagator, model.py:300)
//   Hr  = W_rp[:, F:2F] . h0,  Hs = W_rp[:, 2F:3F] . h0   (receiver / sender column blocks of
//          relation_propagator applied at NODE level instead of per edge, model.py:283-289)
// ---------------------------------------------------------------------------------------------
// Node-encoder de-duplication, step 1: one WAVE per sample walks the sample's nodes in index order and maps every node to a compact
// table row: the first AG_DEDUP_REPS distinct input rows [attrs | phys (0 for tool slots) | action] (bitwise comparison) become shared
// rows, a node that matches none of them gets a private row.  New rows are appended to the encoder's work list (global counter: the
// ORDER of the list does not matter, a row's MFMA chain does not depend on its position in a row tile).
__global__ __launch_bounds__(256) void node_classify_kernel(AgFwdArgs a)
{
    __shared__ unsigned rep[4][AG_DEDUP_REPS][AG_NODE_IN_MAX];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + wave;
    if (b >= a.B) return;                                  // wave-uniform; the kernel has no workgroup-level synchronisation
    const int N = a.N, A = AG_ATTR, Pd = a.phys_dim, D = A + Pd + 3;
    const int base = b * AG_DEDUP_REPS;          // this sample's shared rows
    int nrep = 0;
    constexpr int kPre = 8;                                // 64-node slices whose inputs are fetched together (one memory round trip per 512 nodes)
    for (int s0 = 0; s0 < N; s0 += 64 * kPre) {
        unsigned vv[kPre][AG_NODE_IN_MAX];
#pragma unroll
        for (int u = 0; u < kPre; ++u) {
            const int i = s0 + 64 * u + lane;
            const bool valid = i < N;
            const size_t g = (size_t)b * N + (valid ? i : 0);
#pragma unroll
            for (int k = 0; k < AG_NODE_IN_MAX; ++k) {
                float x = 0.0f;
                if (k < A) x = a.attrs[g * A + k];
                else if (k < A + Pd) x = (valid && i < a.n_p) ? a.phys[(size_t)b * Pd + (k - A)] : 0.0f;
                else if (k < D) x = a.action[g * 3 + (k - A - Pd)];
                vv[u][k] = __float_as_uint(x);
            }
        }
#pragma unroll
        for (int u = 0; u < kPre; ++u) {
            const int i = s0 + 64 * u + lane;
            if (s0 + 64 * u >= N) break;                       // wave-uniform
            const bool valid = i < N;
            const size_t g = (size_t)b * N + (valid ? i : 0);
            unsigned (&v)[AG_NODE_IN_MAX] = vv[u];
            int match = -1;
            for (int r = 0; r < nrep; ++r) {
                bool eq = true;
#pragma unroll
                for (int k = 0; k < AG_NODE_IN_MAX; ++k) eq = eq && v[k] == rep[wave][r][k];
                if (eq && match < 0) match = r;
            }
            while (nrep < AG_DEDUP_REPS) {
                const unsigned long long un = __ballot(valid && match < 0);
                if (!un) break;
                const int leader = __ffsll((long long)un) - 1;
                bool eq = true;
#pragma unroll
                for (int k = 0; k < AG_NODE_IN_MAX; ++k) {
                    const unsigned lv = (unsigned)__shfl((int)v[k], leader);
                    if (lane == 0) rep[wave][nrep][k] = lv;
                    eq = eq && v[k] == lv;
                }
                if (valid && match < 0 && eq) match = nrep;
                if (lane == leader) {
                    const int slot = atomicAdd(a.enc_count, 1);
                    if (slot < a.rows_c) { a.enc_row[slot] = base + nrep; a.enc_src[slot] = (int)g; }
                    else *a.ovf = 1;
                }
                ++nrep;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
            }
            int row = base + match;
            const unsigned long long priv = __ballot(valid && match < 0);       // rows of their own: one pair of atomics per wave, not per lane
            if (priv) {
                const int first = __ffsll((long long)priv) - 1;
                int slot0 = 0, prow0 = 0;
                if (lane == first) { slot0 = atomicAdd(a.enc_count, __popcll(priv)); prow0 = atomicAdd(a.priv_count, __popcll(priv)); }
                slot0 = __shfl(slot0, first);
                prow0 = __shfl(prow0, first);
                if (valid && match < 0) {
                    const int rank = __popcll(priv & ((1ull << lane) - 1ull));
                    row = a.shared_rows + prow0 + rank;                      // private rows follow the B x AG_DEDUP_REPS shared ones
                    const bool fits = row < a.rows_c && slot0 + rank < a.rows_c;
                    if (!fits) { *a.ovf = 1; row = a.rows_c - 1; }           // budget exhausted: this call runs without de-duplication (every consumer tests ovf)
                    else { a.enc_row[slot0 + rank] = row; a.enc_src[slot0 + rank] = (int)g; }
                }
            }
            if (valid) a.node_row[g] = row;
        }
    }
}

// step 2 (independent of the encoders): the first round's sender gathers go to compact rows, so the sender column is mapped once
__device__ __forceinline__ void send_remap_body(const AgFwdArgs &a, int block, int nblocks)
{
    const int E = a.row_ptr[a.B * a.N];
    const bool ovf = *a.ovf != 0;                // the call overflowed the compact tables: round 0 gathers the full-size sender table by node id
    // four edges per thread and trip with their loads batched (index, then row, then store): a plain strided loop orders every trip's two dependent
    // loads behind the previous trip's store
    const int stride = nblocks * 256;
    for (int e0 = block * 256 + threadIdx.x; e0 < E; e0 += 4 * stride) {
        int sd[4], rw[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) sd[u] = e0 + u * stride < E ? a.edge_send[e0 + u * stride] : 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) rw[u] = ovf ? sd[u] : a.node_row[sd[u]];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (e0 + u * stride < E) a.send_c[e0 + u * stride] = rw[u];
    }
}
__global__ __launch_bounds__(256) void send_remap_kernel(AgFwdArgs a) { send_remap_body(a, blockIdx.x, gridDim.x); }

template <class Prec, bool DEDUP>
__global__ __launch_bounds__(AG_MLP_THREADS, AG_MLP_WG_PER_CU) void node_encode_kernel(AgWeights w, AgFwdArgs a)
{
    AG_LDS_DECL
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5, wave = tid >> 6;
    // compact encoder: nothing to do when the call overflowed the compact tables; per-node encoder of a de-duplicated call (a.ovf set): only then
    if (a.ovf && (*a.ovf != 0) == DEDUP) return;
    const int Mn = DEDUP ? *a.enc_count : a.B * a.N;      // rows to encode: the work list of node_classify_kernel, or every node
    const int ntiles = (Mn + AG_ROWS_PER_BLOCK - 1) / AG_ROWS_PER_BLOCK;
    if ((int)blockIdx.x >= ntiles) return;                // (de-duplicated: a handful of row tiles)
    ChunkPipe P{pick<Prec>(w.node_encode, w.node_encode_b3), 26, 0, 0, lds};
    pipe_start(P);
    TileQueue q(nullptr, s_next_tile);   // ~4 row tiles per workgroup: nothing to balance, static stride
#pragma unroll 1
    while (q.tile < ntiles) {
        const int tile = q.tile;
        q.claim();
        const int g = tile * AG_ROWS_PER_BLOCK + wave * 32 + j;
        const bool valid = g < Mn;
        const int gc = valid ? (DEDUP ? a.enc_src[g] : g) : 0;        // a node that carries this row's inputs
        const int b = gc / a.N, i = gc - b * a.N;

        // p_inputs = [attrs(2) | physics_param (0 for tool slots) | action(3) | 1 (bias column)], feature k = 4h + p
        f32x16 in0;
#pragma unroll
        for (int r = 0; r < 16; ++r) in0[r] = 0.0f;
        {
            const int A = AG_ATTR, Pd = a.phys_dim;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int k = 4 * h + p;
                float v = 0.0f;
                if (k < A) v = a.attrs[(size_t)gc * A + k];
                else if (k < A + Pd) v = i < a.n_p ? a.phys[(size_t)b * Pd + (k - A)] : 0.0f;
                else if (k < A + Pd + 3) v = a.action[(size_t)gc * 3 + (k - A - Pd)];
                else if (k == A + Pd + 3) v = 1.0f;   // bias column of particle_encoder.model.0
                in0[p] = v;
            }
        }
        typename Prec::Act x, y;
        Prec::set_tile(x, 0, in0);
        dense_first<Prec, AG_NODE_IN_MAX>(P, x, y);
        q.publish();
        dense<Prec, AG_F, true, true>(P, y, x, ZeroInit{});
        if constexpr (DEDUP) {
            // compact row-major tables at the row the work item names; lanes past the list write dump rows [rows_c, rows_c + 128)
            const size_t row = valid ? (size_t)a.enc_row[g] : (size_t)a.rows_c + wave * 32 + j;
            const size_t rowoff = row * AG_FP + 4 * h;
            dense<Prec, AG_F, true, true>(P, x, y, ZeroInit{}, RowStoreEpi{a.h0c + rowoff});                 // y = particle_encode = h0
            dense_store<Prec, AG_F, false, true>(P, y, ZeroInit{}, RowStoreEpi{a.pnc + rowoff});             // Pn
            dense_store<Prec, AG_F, false, false>(P, y, ZeroInit{}, RowStoreEpi{a.hrc + rowoff});            // Hr (round 0 reads it through node_row)
            dense_store<Prec, AG_F, false, false>(P, y, ZeroInit{}, RowStoreEpi{a.hsc + rowoff});            // Hs (round 0 gathers it through send_c)
        } else {
            const size_t blk = (size_t)(tile * AG_MLP_WAVES + wave) * AG_PACK_BLOCK + h * 128 + j * 4;
            const size_t rowoff = (size_t)g * AG_FP + 4 * h;   // own row even when past Mn (padding rows)
            dense<Prec, AG_F, true, true>(P, x, y, ZeroInit{}, PackStoreEpi{a.h + blk});               // y = particle_encode = h0
            dense_store<Prec, AG_F, false, true>(P, y, ZeroInit{}, PackStoreEpi{a.pn + blk});           // Pn
            dense_store<Prec, AG_F, false, false>(P, y, ZeroInit{}, RowStoreEpi{a.hr + rowoff});  // Hr
            dense_store<Prec, AG_F, false, false>(P, y, ZeroInit{}, RowStoreEpi{a.hs + rowoff});  // Hs
        }
        q.next();
    }
}

// ---------------------------------------------------------------------------------------------
// Edge encoder + pstep-invariant edge term.
//   rel_inputs = [attrs_r | attrs_s | sum|g_r - g_s| | state_norm_r - state_norm_s]   model.py:220-253
//   enc_e      = Encoder(rel_inputs)                                                   model.py:274
//   Eterm      = W_rp[:, :F] . enc_e + b_rp      (first column block of relation_propagator, model.py:289)
// The one-hot gathers Rr.bmm / Rs.bmm become indexed reads of the (L2-resident) raw node inputs.
// ---------------------------------------------------------------------------------------------
template <class Prec> constexpr int kEdgeWgPerCu = AG_MLP_WG_PER_CU;
template <> constexpr int kEdgeWgPerCu<PrecH3> = AG_H3_WG_PER_CU;
// weight stream of the edge stack per arithmetic, and whether its first-layer image carries the residual columns (f16_residual)
template <class Prec> constexpr bool kEdgeResidualSlots = false;
template <> constexpr bool kEdgeResidualSlots<PrecH3> = true;
template <class Prec> __device__ __forceinline__ const float4 *edge_stream(const AgWeights &w) { return pick<Prec>(w.edge_encode, std::is_same_v<Prec, PrecH3> ? w.edge_encode_h2 : w.edge_encode_b3); }
template <class Prec>
__global__ __launch_bounds__(AG_MLP_THREADS, (kEdgeWgPerCu<Prec>)) void edge_encode_kernel(AgWeights w, AgFwdArgs a)
{
    AG_LDS_DECL
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5, wave = tid >> 6;
    const int E = ag_edges(a) + a.self_rows;      // (+ the class rows of elided self-loops, AgFwdArgs::self_info: synthetic edges behind the list)
    if (a.edge_counter && blockIdx.x == 0 && tid == 0) atomicAdd(a.edge_counter, (unsigned long long)E);
    const int ntiles = (E + AG_ROWS_PER_BLOCK - 1) / AG_ROWS_PER_BLOCK;
    if ((int)blockIdx.x >= ntiles) return;
    ChunkPipe P{edge_stream<Prec>(w), 16, 0, 0, lds, w.edge_scale_h3};
    pipe_start(P);
    TileQueue q(a.tile_ctr, s_next_tile);   // ~38 row tiles per workgroup at C2
#pragma unroll 1
    while (q.tile < ntiles) {
        const int tile = q.tile;
        q.claim();
        const int e = tile * AG_ROWS_PER_BLOCK + wave * 32 + j;
        const bool valid = e < E;
        int r = valid ? a.edge_recv[e] : 0, s = valid ? a.edge_send[e] : 0;
        // synthetic self-edge of attribute class r - class_row0 (self-edge elision): [a, a, 0, 0 ...] — a real self-loop's inputs (x - x = +0)
        const int cls = (a.self_rows && r >= a.self_class_row0) ? r - a.self_class_row0 : -1;      // (every copy of a class row carries the class's node-table row)
        if (cls >= 0) r = s = 0;
        const int b = r / a.N, ri = r - b * a.N, si = s - b * a.N;

        float feat[24];
#pragma unroll
        for (int k = 0; k < 24; ++k) feat[k] = 0.0f;
        feat[0] = a.attrs[(size_t)r * 2]; feat[1] = a.attrs[(size_t)r * 2 + 1];
        feat[2] = a.attrs[(size_t)s * 2]; feat[3] = a.attrs[(size_t)s * 2 + 1];
        if (cls >= 0) { feat[0] = feat[2] = cls == 0 ? 1.0f : 0.0f; feat[1] = feat[3] = cls == 0 ? 0.0f : 1.0f; }
        {
            float gd = 0.0f;   // g = cat([p_instance, 0]) (model.py:235), group_diff = sum |g_r - g_s| (:238)
            for (int ii = 0; ii < a.n_inst; ++ii) {
                const float gr = ri < a.n_p ? a.p_instance[((size_t)b * a.n_p + ri) * a.n_inst + ii] : 0.0f;
                const float gs = si < a.n_p ? a.p_instance[((size_t)b * a.n_p + si) * a.n_inst + ii] : 0.0f;
                gd += fabsf(gr - gs);
            }
            feat[4] = gd;
        }
        feat[AG_EDGE_IN] = 1.0f;   // bias column of relation_encoder.model.0
        {
            const float *st = a.state + (size_t)b * AG_NHIS * a.N * 3;
            float pr[AG_NHIS][3], ps[AG_NHIS][3];
#pragma unroll
            for (int hh = 0; hh < AG_NHIS; ++hh)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    pr[hh][c] = st[((size_t)hh * a.N + ri) * 3 + c];
                    ps[hh][c] = st[((size_t)hh * a.N + si) * 3 + c];
                }
#pragma unroll
            for (int hh = 0; hh + 1 < AG_NHIS; ++hh)   // state_res = state[:,1:] - state[:,:-1]  (model.py:155)
#pragma unroll
                for (int c = 0; c < 3; ++c) feat[5 + hh * 3 + c] = (pr[hh + 1][c] - pr[hh][c]) - (ps[hh + 1][c] - ps[hh][c]);
#pragma unroll
            for (int c = 0; c < 3; ++c) feat[5 + (AG_NHIS - 1) * 3 + c] = pr[AG_NHIS - 1][c] - ps[AG_NHIS - 1][c];
            if (cls >= 0)      // (node 0 stood in for the class row's endpoints: its differences with itself are +0 unless it is non-finite)
#pragma unroll
                for (int k = 5; k < AG_EDGE_IN; ++k) feat[k] = 0.0f;
        }
        f32x16 in0;
#pragma unroll
        for (int r16 = 0; r16 < 16; ++r16) in0[r16] = 0.0f;
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int p = 0; p < 4; ++p) in0[4 * q + p] = h ? feat[8 * q + 4 + p] : feat[8 * q + p];
        if constexpr (kEdgeResidualSlots<Prec>) {      // fp16 residuals of the state differences in the spare K slots 18..29 (see f16_residual)
#pragma unroll
            for (int q = 2; q < 4; ++q)
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const int k0 = 8 * q + p, k1 = k0 + 4;      // this lane half's slot: k0 (h = 0) or k1 (h = 1)
                    const bool l0 = k0 >= AG_EDGE_LO_SLOT0 && k0 < AG_EDGE_LO_SLOT0 + AG_EDGE_LO_COUNT;
                    const bool l1 = k1 >= AG_EDGE_LO_SLOT0 && k1 < AG_EDGE_LO_SLOT0 + AG_EDGE_LO_COUNT;
                    const float v0 = l0 ? f16_residual(feat[k0 - AG_EDGE_LO_SLOT0 + AG_EDGE_LO_FEAT0]) : (k0 < 18 ? feat[k0] : 0.0f);
                    const float v1 = l1 ? f16_residual(feat[k1 - AG_EDGE_LO_SLOT0 + AG_EDGE_LO_FEAT0]) : (k1 < 18 ? feat[k1] : 0.0f);
                    in0[4 * q + p] = h ? v1 : v0;
                }
        }

        typename Prec::Act x, y;
        Prec::set_tile(x, 0, in0);
        dense_first<Prec, AG_EDGE_IN + 1>(P, x, y);
        q.publish();
        dense<Prec, AG_F, true, true>(P, y, x, ZeroInit{});
        dense<Prec, AG_F, true, true>(P, x, y, ZeroInit{});    // relation_encode
        if (a.eterm_half)    // Eterm (q16 table in precision mode 2)
            dense_store<Prec, AG_F, false, true>(P, y, ZeroInit{}, RowStoreQ16Epi{reinterpret_cast<unsigned char *>(a.eterm) + (size_t)e * (2 * AG_FP), h, a.status});
        else
            dense_store<Prec, AG_F, false, true>(P, y, ZeroInit{}, RowStoreEpi{a.eterm + (size_t)e * AG_FP + 4 * h});
        if constexpr (std::is_same_v<Prec, PrecH3>) { h3_report(x.bad, a.status); h3_report(y.bad, a.status); }      // a hidden activation left fp16's range
        q.next();
    }
}

struct EdgeRaw {           // raw gathered inputs of one edge (receiver r, sender s), model.py:220-253
    float ar[2], as[2], gr, gs;
    float pr[AG_NHIS][3], ps[AG_NHIS][3];
    int ri, si, b;
};

__device__ __forceinline__ void edge_gather(const AgFwdArgs &a, int r, int s, EdgeRaw &g)
{
    const int b = r / a.N, ri = r - b * a.N, si = s - b * a.N;
    g.ri = ri; g.si = si; g.b = b;
    g.ar[0] = a.attrs[(size_t)r * 2]; g.ar[1] = a.attrs[(size_t)r * 2 + 1];
    g.as[0] = a.attrs[(size_t)s * 2]; g.as[1] = a.attrs[(size_t)s * 2 + 1];
    // instance 0 of g = cat([p_instance, 0]) (model.py:235); further instances are read in edge_features
    g.gr = (a.n_inst > 0 && ri < a.n_p) ? a.p_instance[((size_t)b * a.n_p + ri) * a.n_inst] : 0.0f;
    g.gs = (a.n_inst > 0 && si < a.n_p) ? a.p_instance[((size_t)b * a.n_p + si) * a.n_inst] : 0.0f;
    const float *st = a.state + (size_t)b * AG_NHIS * a.N * 3;
#pragma unroll
    for (int hh = 0; hh < AG_NHIS; ++hh)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            g.pr[hh][c] = st[((size_t)hh * a.N + ri) * 3 + c];
            g.ps[hh][c] = st[((size_t)hh * a.N + si) * 3 + c];
        }
}

// rel_inputs = [attrs_r | attrs_s | sum|g_r - g_s| | state_res_r - state_res_s | cur_r - cur_s | 1]; lane half h keeps
// features 8q + 4h + p (the B-operand image of k16-steps 0 and 1)
__device__ __forceinline__ void edge_features(const AgFwdArgs &a, const EdgeRaw &g, int h, f32x16 &in0)
{
    float feat[24];
#pragma unroll
    for (int k = 0; k < 24; ++k) feat[k] = 0.0f;
    feat[0] = g.ar[0]; feat[1] = g.ar[1]; feat[2] = g.as[0]; feat[3] = g.as[1];
    float gd = fabsf(g.gr - g.gs);
    for (int ii = 1; ii < a.n_inst; ++ii) {
        const float gr = g.ri < a.n_p ? a.p_instance[((size_t)g.b * a.n_p + g.ri) * a.n_inst + ii] : 0.0f;
        const float gs = g.si < a.n_p ? a.p_instance[((size_t)g.b * a.n_p + g.si) * a.n_inst + ii] : 0.0f;
        gd += fabsf(gr - gs);
    }
    feat[4] = gd;
    feat[AG_EDGE_IN] = 1.0f;   // bias column of relation_encoder.model.0
#pragma unroll
    for (int hh = 0; hh + 1 < AG_NHIS; ++hh)   // state_res = state[:,1:] - state[:,:-1]  (model.py:155)
#pragma unroll
        for (int c = 0; c < 3; ++c) feat[5 + hh * 3 + c] = (g.pr[hh + 1][c] - g.pr[hh][c]) - (g.ps[hh + 1][c] - g.ps[hh][c]);
#pragma unroll
    for (int c = 0; c < 3; ++c) feat[5 + (AG_NHIS - 1) * 3 + c] = g.pr[AG_NHIS - 1][c] - g.ps[AG_NHIS - 1][c];
#pragma unroll
    for (int r16 = 0; r16 < 16; ++r16) in0[r16] = 0.0f;
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int p = 0; p < 4; ++p) in0[4 * q + p] = h ? feat[8 * q + 4 + p] : feat[8 * q + p];
}

// =====================================================================================================================
// Weight-STATIONARY edge encoder (precision mode 2, arithmetic PrecH3; the default there, ag_set_option("edge_stationary", 0) selects the streaming kernel).
//
// The streaming kernels above re-read the whole 320 KB weight image from L2 through LDS for every 128 edges: 2 560 B of
// L2->LDS traffic and 2 560 B of LDS fragment reads per edge, against 388 B of HBM traffic.  Here the dataflow is turned around:
//  * ONE 512-thread workgroup per CU, two waves per SIMD, 256 registers per lane.  The four layers are cut into 15 "units" of one
//    32-feature out-tile (30 matrix instructions per 32 edges) plus the narrow first layer (5 tiles x 4): a wave owns TWO units and keeps
//    their A operands (fp16 hi fragments + the scaled MFMA's [e4m3 lo | e4m3 hi] operands: 80 registers per unit) in REGISTERS for the
//    whole launch — the compiler splits a 256-register wave 128 + 128, so a wave's first unit and the fp16 half of its second sit in
//    accumulation registers (the MFMA reads its A operand from there directly), the rest and all accumulators in architectural ones:
//        wave 0, 1: RE1 tiles {0,1}, {2,3} + first-layer tile 0 / 1       wave 2, 3: RE2 tiles {0,1}, {2,3} + first-layer tile 2 / 3
//        wave 4, 5: We tiles {0,1}, {2,3}                                  wave 6: RE1 tile 4, RE2 tile 4
//        wave 7: We tile 4, per-edge input gather, first-layer tile 4          (waves w and w + 4 share a SIMD)
//  * 32-edge blocks flow through the waves as a software pipeline; a layer's 160 x 32 activation block is handed over through LDS
//    as a SET of two images already in the B-operand layout of the next layer (lane (j, h) writes exactly the bytes lane (j, h) of
//    the consumer reads): 10 KB of fp16 values and 5 KB of e5m2 residual bytes (PrecH3: x = x16 + r8).  One barrier per ROUND.  Block i:
//    indices / node rows / features in rounds i .. i + 2 (wave 7), first layer in round i + 3, RE1 i + 4, RE2 i + 5 (its pairs hand the
//    block over at the start of round i + 6), We i + 7 (its pairs store the rows at the start of round i + 8).  Rings: RE1 and RE2 inputs
//    two blocks, We input three; LDS 131 KB.
//  * What a second wave per SIMD buys (tools/ubench/mx_lone.hip, valu_beside_mfma.hip; profiles/r04_edge_ws8_trace.txt): a wave does not
//    overlap its own VALU work with its own matrix instructions, and while one wave of a SIMD issues MFMAs back to back the other's
//    instructions take ~10 cycles each (packed fp32 VALU 39: none are used here).  So a wave runs its 30 MFMAs, then its epilogues as plain
//    code, and the two waves of a SIMD are kept in OPPOSITE halves of their rounds: waves 2-5 start a round with the epilogue of the
//    accumulators they computed in the previous round, their partners start with their MFMAs.  Until r04 the kernel ran four waves of 512
//    registers with every epilogue cut into micro-chores pinned into MFMA shadows: 0.72 ms against 0.61 for this one.
//  * A dependent MFMA issued straight after its predecessor uses the pipe's accumulate path; results of asm MFMAs are not interlocked against
//    compiler-placed readers (ws_settle), a VALU-written B operand needs two wait states (s_nop 1), and the scaled MFMA reads its eight B
//    registers over several passes after issue (two register sets by tile parity).
//  * Each accumulator sees hi.x16 of k16-steps 2t, 2t + 1 and the scaled correction product by ascending input tile t, so results equal
//    edge_encode_kernel<PrecH3> bit for bit.
// =====================================================================================================================
#define AG_WS_IMG 10240          // bytes of one fp16 activation image: [10 k16-steps][64 lanes][8 fp16]
#define AG_WS_RES 5120           // bytes of its residual image: [10 k16-steps][64 lanes][8 e5m2]
#define AG_WS_SET (AG_WS_IMG + AG_WS_RES)
#define AG_WS_IN0 2048           // first-layer input image: 2 k16-steps
#define AG_WS_SLOTS 3

// A (layer, out-tile) unit's A operands: fp16 hi fragments by k16-step, the block-scaled MFMA's operands [e4m3 lo | e4m3 hi] by input tile, and
// the unit's block scales (lane (i, h): h = 0 the lo scales, h = 1 the hi scales; sc0 = tiles 0..3 by byte, sc1 byte 0 = tile 4)
struct WsUnit { f16x8 hi[10]; h3_i32x8 mx[AG_NT]; unsigned sc0, sc1; };
typedef h3_u32x4 ws_u32x4;
typedef int ws_i32x2 __attribute__((ext_vector_type(2)));

// ACC: keep the unit in the accumulation-register half of the file; a wave holds three units (240 registers) there.
template <bool ACC, bool ACC_MX = ACC>
__device__ __forceinline__ void ws_load_unit(WsUnit &W, const float4 *chunk, const uint32_t *scales, int lane)
{
    typedef int i32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int u = 0; u < 10; ++u) W.hi[u] = *reinterpret_cast<const f16x8 *>(chunk + u * 64 + lane);
#pragma unroll
    for (int t = 0; t < AG_NT; ++t) {
        const i32x4 a = *reinterpret_cast<const i32x4 *>(chunk + AG_H3_HI_BYTES / 16 + (t * 64 + lane) * 2);
        const i32x4 b = *reinterpret_cast<const i32x4 *>(chunk + AG_H3_HI_BYTES / 16 + (t * 64 + lane) * 2 + 1);
        W.mx[t] = h3_i32x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    }
    W.sc0 = scales[lane];
    W.sc1 = scales[64 + lane];
    // opaque values (after ALL the loads: the asm is a use, and a use waits for its load): they must live in registers and cannot be re-loaded
    // inside the persistent loop
#pragma unroll
    for (int u = 0; u < 10; ++u) { if (ACC) asm volatile("" : "+a"(W.hi[u])); else asm volatile("" : "+v"(W.hi[u])); }
#pragma unroll
    for (int t = 0; t < AG_NT; ++t) { if (ACC_MX) asm volatile("" : "+a"(W.mx[t])); else asm volatile("" : "+v"(W.mx[t])); }
    asm volatile("" : "+v"(W.sc0), "+v"(W.sc1));
}
__device__ __forceinline__ unsigned lds_addr3(const __attribute__((address_space(3))) void *p) { return (unsigned)(uintptr_t)p; }
template <int N>
__device__ __forceinline__ void ws_wait2(bf16x8 &a, bf16x8 &b) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N)); }
template <int N>
__device__ __forceinline__ void ws_wait3(bf16x8 &a, bf16x8 &b, bf16x8 &c)
{
    static_assert(N >= 0 && N <= 15, "lgkmcnt is 4 bits");
    asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a), "+v"(b), "+v"(c) : "n"(N));
}

// acc (+)= W . x from inline asm: accumulator and B operand in architectural registers, A operand where the unit lives
template <bool ACC, bool FIRST>
__device__ __forceinline__ void ws_mfma(f32x16 &acc, const f16x8 &w, const bf16x8 &x)
{
    if constexpr (FIRST) {
        if constexpr (ACC) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(acc) : "a"(w), "v"(x));
        else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(acc) : "v"(w), "v"(x));
    } else {
        if constexpr (ACC) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(w), "v"(x));
        else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(x));
    }
}
// the two fp16 MFMAs of an input tile on one accumulator, back to back (the second takes the pipe's accumulate path), as ONE statement: hipcc pads
// every asm statement whose outputs the next instruction reads with a wait state of its own
template <bool ACC, bool FIRST>
__device__ __forceinline__ void ws_mfma2(f32x16 &acc, const f16x8 &w0, const f16x8 &w1, const bf16x8 &x0, const bf16x8 &x1)
{
    if constexpr (FIRST) {
        if constexpr (ACC) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %3, 0\n\tv_mfma_f32_32x32x16_f16 %0, %2, %4, %0" : "=&v"(acc) : "a"(w0), "a"(w1), "v"(x0), "v"(x1));
        else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %3, 0\n\tv_mfma_f32_32x32x16_f16 %0, %2, %4, %0" : "=&v"(acc) : "v"(w0), "v"(w1), "v"(x0), "v"(x1));
    } else {
        if constexpr (ACC) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %3, %0\n\tv_mfma_f32_32x32x16_f16 %0, %2, %4, %0" : "+v"(acc) : "a"(w0), "a"(w1), "v"(x0), "v"(x1));
        else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %3, %0\n\tv_mfma_f32_32x32x16_f16 %0, %2, %4, %0" : "+v"(acc) : "v"(w0), "v"(w1), "v"(x0), "v"(x1));
    }
}
// acc += 2^(sa - 127) 2^(sb - 127) A8 . B8 over two 32-element K blocks: A e4m3 (cbsz 0), B e5m2 (blgp 1); the A scale is byte SEL of `sa` in the
// lanes of the half with the block's number, the B scale byte 0 of `sb`
// (the statement opens with the two wait states a VALU-written B operand needs before an MFMA reads it: the byte permutes that build it may be
// scheduled anywhere above)
template <bool ACC, int SEL>
__device__ __forceinline__ void ws_mfma_mx(f32x16 &acc, const h3_i32x8 &a, const h3_i32x8 &b, unsigned sa, unsigned sb)
{
#define AG_MX(OPS) \
    do { if constexpr (ACC) asm volatile("s_nop 1\n\tv_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 " OPS " cbsz:0 blgp:1" : "+v"(acc) : "a"(a), "v"(b), "v"(sa), "v"(sb)); \
         else asm volatile("s_nop 1\n\tv_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 " OPS " cbsz:0 blgp:1" : "+v"(acc) : "v"(a), "v"(b), "v"(sa), "v"(sb)); } while (0)
    if constexpr (SEL == 0) AG_MX("op_sel:[0,0,0] op_sel_hi:[0,0,0]");
    else if constexpr (SEL == 1) AG_MX("op_sel:[1,0,0] op_sel_hi:[0,0,0]");
    else if constexpr (SEL == 2) AG_MX("op_sel:[0,0,0] op_sel_hi:[1,0,0]");
    else AG_MX("op_sel:[1,0,0] op_sel_hi:[1,0,0]");
#undef AG_MX
}

// Epilogue of the hidden layers in eight pieces.  M = 0..7 of out-tile T: half S = M >> 2, output dword M & 3 (two accumulator values):
// ReLU, packed fp16 convert, largest-pattern tracking, the two residual bytes (h3_pair); the fourth dword stores the consumer's 16 bytes
// of k16-step 2T + S (bias column: feature 150 := 1.0) and its 8 residual bytes (residual image: [5 input tiles][64 lanes][16 bytes]).
struct WsEpi { ws_u32x4 H; int R[4]; unsigned bad; };
typedef __attribute__((address_space(3))) unsigned char lds_u8;      // LDS pointers stay in their address space: a store is one ds_write with an
                                                                     // immediate offset (through a generic pointer: two address instructions each)
template <int T, int M>
__device__ __forceinline__ void ws_act_micro(const f32x16 &acc, WsEpi &E, lds_u8 *set_lane, int h)
{
    constexpr int S = M >> 2, w = M & 3;
    unsigned untracked = 0;
    E.H[w] = h3_pair<false>(relu1(acc[8 * S + 2 * w]), relu1(acc[8 * S + 2 * w + 1]), E.R[2 * S + (w >> 1)], (w & 1) != 0, untracked);
    // largest fp16 pattern so far (inf / NaN = an overflow of this layer): the values are >= 0, so the float maximum is the integer one; NaN propagates
    if constexpr ((w & 1) == 1) asm("v_pk_maximum3_f16 %0, %0, %1, %2" : "+v"(E.bad) : "v"(E.H[w - 1]), "v"(E.H[w]));
    if constexpr (w == 3) {
        if constexpr (T == 4 && S == 1) {           // feature 150 = 16*9 + 6: element e = 2 of the h = 1 half (its residual byte is 0: the feature is padding)
            if (h == 1) E.H[1] = (E.H[1] & 0xffff0000u) | 0x3c00u;
        }
        *reinterpret_cast<__attribute__((address_space(3))) ws_u32x4 *>(set_lane + (2 * T + S) * 1024) = E.H;
        if constexpr (S == 1)
            *reinterpret_cast<__attribute__((address_space(3))) ws_u32x4 *>(set_lane + AG_WS_IMG + T * 1024) = ws_u32x4{(unsigned)E.R[0], (unsigned)E.R[1], (unsigned)E.R[2], (unsigned)E.R[3]};
    }
}
// We: one out-tile of the q16 table in seven chores (format and helpers: RowStoreQ16Epi above): 0, 1 the lane's maximum over its 16 values,
// 2 the tile exponent (partner half by v_permlane32_swap) and its byte, 3..6 two packed converts each, 4 and 6 store 16 (8) bytes.
// Branch-free on purpose: a store under `if (block is valid)` made the compiler sink the whole tile's converts into the conditional block.
// Rows of blocks outside the launch go to 32 dump rows behind the table (the 16-bit table uses half of its fp32-sized allocation).
struct WsQ16 { unsigned m; int eb; int inv; unsigned w[4]; unsigned nonfinite; };
template <int T, int C>
__device__ __forceinline__ void ws_q16_chore(const f32x16 &acc, WsQ16 &Q, unsigned char *row, int h)
{
    static_assert(C >= 0 && C < 7, "seven chores per out-tile");
    if constexpr (C == 0) {
        Q.m = 0;
#pragma unroll
        for (int r = 0; r < 8; r += 2) Q.m = q16_max2<false>(Q.m, acc[r], acc[r + 1]);
    } else if constexpr (C == 1) {
#pragma unroll
        for (int r = 8; r < 16; r += 2) Q.m = q16_max2<false>(Q.m, acc[r], acc[r + 1]);
    } else if constexpr (C == 2) {
        bool nf;
        Q.eb = q16_tile_exp(Q.m, nf);
        Q.inv = q16_inv_scale(Q.eb);
        Q.nonfinite |= nf ? 1u : 0u;
        q16_store_exp(row, T, h, Q.eb);
    } else {
        constexpr int s = (C - 3) >> 1, k0 = 2 * ((C - 3) & 1);
        Q.w[k0] = q16_pack(acc[8 * s + 2 * k0], acc[8 * s + 2 * k0 + 1], Q.inv);
        Q.w[k0 + 1] = q16_pack(acc[8 * s + 2 * k0 + 2], acc[8 * s + 2 * k0 + 3], Q.inv);
        if constexpr (((C - 3) & 1) == 1) q16_store_half(row, T, h, s, Q.w);
    }
}

// End of a round: LDS writes of this wave landed, then the workgroup barrier.  NOT __syncthreads(): its workgroup-scope fence also
// drains vmcnt, i.e. it would wait at every round for the Eterm stores (and gather loads) issued a few hundred cycles earlier.
__device__ __forceinline__ void ws_round_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Per-node inputs of the edge features: ag_edge_node_tab_row (ag_common.h), one thread per node.
// Workgroups past the node range (de-duplicated calls) map the sender column to compact rows for round 0's reduce (send_remap_body): one small
// launch per model step instead of two.  (In ag_rollout both ride the edge builder's launches instead — AgEdgeArgs riders — and this kernel is
// launched for what they did not cover: ag_forward, the brute-force edge path, the CU-partitioned rollout.)
__global__ __launch_bounds__(256) void edge_node_tab_kernel(AgFwdArgs a, int nb_tab)
{
    if ((int)blockIdx.x >= nb_tab) { send_remap_body(a, (int)blockIdx.x - nb_tab, (int)gridDim.x - nb_tab); return; }
    ag_edge_node_tab_row(a.state, a.attrs, a.p_instance, a.n_inst, a.n_p, a.B, a.N, a.edge_node_tab, a.status, blockIdx.x * 256 + threadIdx.x,
                         a.self_rows ? (long long)a.self_class_row0 : -1);
}

// First layer of one 32-edge block for out-tiles [T0, T0 + NT): per tile 2 k16-steps x (lo, hi) fp16 MFMAs with the A fragments read from the
// LDS image `wf` (this lane's 16 bytes of fragment 0; [5 tiles][2 steps][hi | lo][64 lanes][8 fp16]) and the block's input image at `lin` — NT
// independent chains, two plain products (the inputs carry their own residuals in spare K slots).  slot(IC<m>) runs after MFMA m = g * NT + t
// (g = 0..3: (step 0, lo), (step 0, hi), (step 1, lo), (step 1, hi)), m = 0 .. 4 NT - 1.
template <int T0, int NT, class Slot>
__device__ __forceinline__ void ws_first_layer(f32x16 (&accF)[NT], unsigned lin, unsigned wf, Slot &&slot)
{
    bf16x8 xq[2], fq[2][NT];
    __builtin_amdgcn_s_setprio(3);      // as in ws_phase
    lds_read16<0>(xq[0], lin);
    lds_read16<1024>(xq[1], lin);
    static_for<0, NT>([&](auto T) { constexpr int t = decltype(T)::value; lds_read16<(((T0 + t) * 2 + 0) * 2 + 1) * 1024>(fq[0][t], wf); });
    static_for<0, 4>([&](auto GG) {
        constexpr int g = decltype(GG)::value, u = g >> 1;
        if constexpr (g < 3) {
            constexpr int nu = (g + 1) >> 1, nhl = 1 - ((g + 1) & 1);
            static_for<0, NT>([&](auto T) { constexpr int t = decltype(T)::value; lds_read16<(((T0 + t) * 2 + nu) * 2 + nhl) * 1024>(fq[(g + 1) & 1][t], wf); });
        }
        static_for<0, NT>([&](auto T) {
            constexpr int t = decltype(T)::value;
            constexpr int later = (NT - 1 - t) + (g < 3 ? NT : 0);
            ws_wait2<later>(fq[g & 1][t], xq[u]);
            ws_mfma<false, (g == 0)>(accF[t], __builtin_bit_cast(f16x8, fq[g & 1][t]), xq[u]);
            slot(std::integral_constant<int, g * NT + t>{});
            __builtin_amdgcn_sched_barrier(0);
        });
    });
    __builtin_amdgcn_s_setprio(0);
}

#define AG_WS_LAG_F 3
#define AG_WS_LAG_1 4
#define AG_WS_LAG_2 5
#define AG_WS_LAG_3 7      // the RE2 pairs hand their block over at the start of the NEXT round
// An asm MFMA's result is not interlocked against the VALU instructions the compiler places after it: 16 passes + 4 states for the scaled one
__device__ __forceinline__ void ws_settle(f32x16 &a) { asm volatile("s_nop 15\n\ts_nop 7" : "+v"(a)); }
__device__ __forceinline__ void ws_settle(f32x16 &a, f32x16 &b) { asm volatile("s_nop 15\n\ts_nop 7" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void ws_settle(f32x16 &a, f32x16 &b, f32x16 &c) { asm volatile("s_nop 15\n\ts_nop 7" : "+v"(a), "+v"(b), "+v"(c)); }
// One MFMA phase of the eight-wave kernel: accumulators 0 .. NA-1 run units W[0 .. NA-1] on ONE input set (la: this lane's 16 bytes of k16-step 0).
// The compiler splits a 256-register wave into 128 + 128: a wave's first unit and the fp16 half of its second live in accumulation registers
// (NACC2 half-units: fp16 fragments of unit k = half-unit 2k, its scaled-MFMA operands = half-unit 2k + 1), the rest in architectural ones.
// No operand ring: the partner wave's matrix work covers the LDS latency.  The next tile's reads are issued after the fp16 MFMAs that read the
// current fragments and land during the scaled MFMAs; the scaled MFMA's B operand (read over several passes after issue) alternates between two
// register sets by tile parity.
template <int NA, int NACC2, int U0 = 0, int NW, int NACCS>
__device__ __forceinline__ void ws_phase(const WsUnit (&W)[NW], f32x16 (&acc)[NACCS], unsigned la)
{
    bf16x8 xa, xb, r;
    // The instruction arbiter serves the OLDER wave of a SIMD first: without a raised priority the younger wave's MFMAs wait behind every VALU
    // instruction of the older wave's epilogue (kernel 0.676 -> 0.615 ms with it)
    __builtin_amdgcn_s_setprio(3);
    lds_read16<0>(xa, la);
    lds_read16<1024>(xb, la);
    lds_read16<AG_WS_IMG>(r, la);
    const unsigned one = 0x7f7f7f7fu;      // E8M0 127 = 2^0: the activations' bytes are plain e5m2 numbers
    h3_i32x8 Bq[2];
    static_for<0, AG_NT>([&](auto TT) {
        constexpr int t = decltype(TT)::value;
        ws_wait3<0>(xa, xb, r);
        h3_i32x8 &B = Bq[t & 1];
        B = h3_b_operand(h3_top_bytes(__builtin_bit_cast(h3_u32x4, xa), __builtin_bit_cast(h3_u32x4, xb)), __builtin_bit_cast(h3_u32x4, r));
        static_for<0, NA>([&](auto KK) {
            constexpr int k = decltype(KK)::value;
            ws_mfma2<(2 * (U0 + k) < NACC2), (t == 0)>(acc[U0 + k], W[U0 + k].hi[2 * t], W[U0 + k].hi[2 * t + 1], xa, xb);
        });
        if constexpr (t + 1 < AG_NT) {
            lds_read16<(2 * t + 2) * 1024>(xa, la);
            lds_read16<(2 * t + 3) * 1024>(xb, la);
            lds_read16<AG_WS_IMG + (t + 1) * 1024>(r, la);
        }
        static_for<0, NA>([&](auto KK) {
            constexpr int k = decltype(KK)::value;
            ws_mfma_mx<(2 * (U0 + k) + 1 < NACC2), (t & 3)>(acc[U0 + k], W[U0 + k].mx[t], B, t < 4 ? W[U0 + k].sc0 : W[U0 + k].sc1, one);
        });
        if constexpr (t > 0) asm volatile("" :: "v"(Bq[(t - 1) & 1]));      // the previous tile's operand is released only now
    });
    asm volatile("" :: "v"(Bq[(AG_NT - 1) & 1]));
    __builtin_amdgcn_s_setprio(0);
}
template <int T>
__device__ __forceinline__ void ws_hidden_tile(const f32x16 &acc, WsEpi &E, lds_u8 *set_lane, int h)
{
    static_for<0, 8>([&](auto MM) { ws_act_micro<T, decltype(MM)::value>(acc, E, set_lane, h); });
}
template <int T>
__device__ __forceinline__ void ws_table_tile(const f32x16 &acc, WsQ16 &Q, unsigned char *row, int h)
{
    static_for<0, 7>([&](auto CC) { ws_q16_chore<T, decltype(CC)::value>(acc, Q, row, h); });
}

__global__ __launch_bounds__(512, 1) void edge_encode_ws_kernel(AgWeights w, AgFwdArgs a)
{
    __shared__ __attribute__((aligned(16))) unsigned char s_act[7][AG_WS_SET];                // input sets of RE1, RE2 (rings of two blocks), We (ring of three)
    __shared__ __attribute__((aligned(16))) unsigned char s_in0[AG_WS_SLOTS][AG_WS_IN0];      // first-layer inputs
    __shared__ __attribute__((aligned(16))) float4 s_wf[AG_CHUNK_F4];                         // first-layer fragments [5 tiles][2 steps][hi|lo][64][8]
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int E = ag_edges(a) + a.self_rows;      // (+ the class rows of elided self-loops: their endpoints are the class rows of the per-node table)
    if (a.edge_counter && blockIdx.x == 0 && tid == 0) atomicAdd(a.edge_counter, (unsigned long long)E);
    const int nblk = (E + 31) / 32;
    if ((int)blockIdx.x >= nblk) return;
    const int n_i = (nblk - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;      // blocks of this workgroup: blockIdx + i * gridDim
    const int rounds = n_i + AG_WS_LAG_3 + 1;      // + 1: the We pairs store a block's rows at the start of the next round
    const float4 *ws = w.edge_encode_h2;
    for (int i = tid; i < AG_CHUNK_F4; i += 512) s_wf[i] = ws[i];
    for (int i = tid; i < (int)(sizeof(s_act) / 16); i += 512) reinterpret_cast<float4 *>(&s_act[0][0])[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = tid; i < (int)(sizeof(s_in0) / 16); i += 512) reinterpret_cast<float4 *>(&s_in0[0][0])[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const size_t e_pad = ag_edge_rows_pad(a.e_cap);        // rows of the table (fwd_layout); dump rows start here
    auto gblock = [&](int i) { return (int)blockIdx.x + i * (int)gridDim.x; };
    auto slot_of = [](int i) { return (i + 4 * AG_WS_SLOTS) % AG_WS_SLOTS; };                 // i >= -12
    // this lane's 16 bytes of k16-step 0 of the fp16 image of input set `layer` (0: RE1, 1: RE2, 2: We), block i (i >= -8)
    auto img = [&](int layer, int i) -> lds_u8 * { return (lds_u8 *)&s_act[layer < 2 ? 2 * layer + ((i + 8) & 1) : 4 + (i + 9) % 3][lane * 16]; };
    auto eterm_row = [&](int i) {
        const size_t e = ((i >= 0 && i < n_i) ? (size_t)gblock(i) * 32 : e_pad) + j;
        return reinterpret_cast<unsigned char *>(a.eterm) + e * (2 * AG_FP);
    };
    WsEpi Ep{{0u, 0u, 0u, 0u}, {0, 0, 0, 0}, 0u};
    const uint32_t *wsc = w.edge_scale_h3;      // block scales of unit k (stream chunk 1 + k): wsc + 128 k
    auto load_unit = [&](WsUnit &U, int chunk) { ws_load_unit<true, true>(U, ws + (size_t)chunk * AG_CHUNK_F4, wsc + (size_t)(chunk - 1) * 128, lane); };
    auto load_unit2 = [&](WsUnit &U, int chunk) { ws_load_unit<true, false>(U, ws + (size_t)chunk * AG_CHUNK_F4, wsc + (size_t)(chunk - 1) * 128, lane); };
    auto noop = [](auto) {};
    const unsigned wf = lds_addr_of(s_wf) + lane * 16;

    // hidden layer L (1: RE1, 2: RE2), out-tiles T0 and T0 + 1; then first-layer tile TF (-1: none)
    auto hidden_pair = [&](auto LL, auto TT, auto FF, auto DD) {
        constexpr int L = decltype(LL)::value, T0 = decltype(TT)::value, TF = decltype(FF)::value;
        constexpr bool DEFER = decltype(DD)::value;      // the round starts with the PREVIOUS round's epilogue (the SIMD partner starts with its MFMAs)
        WsUnit W[2];
        load_unit(W[0], 1 + 5 * (L - 1) + T0);
        load_unit2(W[1], 2 + 5 * (L - 1) + T0);
        __syncthreads();
        f32x16 acc[2];
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[0][q] = acc[1][q] = 0.0f;
#pragma unroll 1
        for (int r = 0; r < rounds; ++r) {
            const int i = r - (AG_WS_LAG_F + L);
            if constexpr (DEFER) {
                lds_u8 *outp = img(L, i - 1);
                ws_hidden_tile<T0>(acc[0], Ep, outp, h);
                ws_hidden_tile<T0 + 1>(acc[1], Ep, outp, h);
            }
            const unsigned la = lds_addr3(img(L - 1, i));
            ws_phase<2, 3>(W, acc, la);
            ws_settle(acc[0], acc[1]);
            if constexpr (!DEFER) {
                lds_u8 *out = img(L, i);
                ws_hidden_tile<T0>(acc[0], Ep, out, h);
                ws_hidden_tile<T0 + 1>(acc[1], Ep, out, h);
            }
            if constexpr (TF >= 0) {
                const int i0 = r - AG_WS_LAG_F;
                f32x16 accF[1];
                ws_first_layer<TF, 1>(accF, lds_addr_of(&s_in0[slot_of(i0)][lane * 16]), wf, noop);
                ws_settle(accF[0]);
                ws_hidden_tile<TF>(accF[0], Ep, img(0, i0), h);
            }
            ws_round_barrier();
        }
    };
    // We tiles T0 and T0 + 1 and first-layer tile TF.  The round STARTS with the previous round's table epilogue (the SIMD partner starts with its
    // MFMAs: the two waves stay in opposite halves of their rounds), then the first-layer tile, then this round's MFMAs.
    auto table_pair = [&](auto FF, auto TT) {
        constexpr int TF = decltype(FF)::value, T0 = decltype(TT)::value;
        WsUnit W[2];
        load_unit(W[0], 11 + T0);
        load_unit2(W[1], 12 + T0);
        __syncthreads();
        f32x16 acc[2], accF[1];
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[0][q] = acc[1][q] = 0.0f;
        WsQ16 Q{0u, AG_Q16_EB_MIN, 0, {0u, 0u, 0u, 0u}, 0u};
#pragma unroll 1
        for (int r = 0; r < rounds; ++r) {
            const int i0 = r - AG_WS_LAG_F, i3 = r - AG_WS_LAG_3;
            unsigned char *rowp = eterm_row(i3 - 1);
            ws_table_tile<T0>(acc[0], Q, rowp, h);
            ws_table_tile<T0 + 1>(acc[1], Q, rowp, h);
            if constexpr (TF >= 0) {
                ws_first_layer<TF, 1>(accF, lds_addr_of(&s_in0[slot_of(i0)][lane * 16]), wf, noop);
                ws_settle(accF[0]);
                ws_hidden_tile<TF>(accF[0], Ep, img(0, i0), h);
            }
            const unsigned la = lds_addr3(img(2, i3));
            ws_phase<2, 3>(W, acc, la);
            ws_settle(acc[0], acc[1]);
            ws_round_barrier();
        }
        if (Q.nonfinite && a.status) atomicOr(a.status, 1);
    };

    if (wave == 0) hidden_pair(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::false_type{});
    else if (wave == 1) hidden_pair(std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{}, std::integral_constant<int, 1>{}, std::false_type{});
    else if (wave == 2) hidden_pair(std::integral_constant<int, 2>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{}, std::true_type{});
    else if (wave == 3) hidden_pair(std::integral_constant<int, 2>{}, std::integral_constant<int, 2>{}, std::integral_constant<int, 3>{}, std::true_type{});
    else if (wave == 4) table_pair(std::integral_constant<int, -1>{}, std::integral_constant<int, 0>{});
    else if (wave == 5) table_pair(std::integral_constant<int, -1>{}, std::integral_constant<int, 2>{});
    else if (wave == 6) {
        // ---------------------------------------------------------------- RE1 tile 4 and RE2 tile 4: two input sets, one after the other
        WsUnit W[1], W1[1];
        load_unit(W[0], 1 + 4);
        load_unit2(W1[0], 6 + 4);
        __syncthreads();
        f32x16 accA[1], accB[1];
#pragma unroll 1
        for (int r = 0; r < rounds; ++r) {
            const int i1 = r - AG_WS_LAG_1, i2 = r - AG_WS_LAG_2;
            const unsigned la1 = lds_addr3(img(0, i1)), la2 = lds_addr3(img(1, i2));
            ws_phase<1, 2>(W, accA, la1);
            ws_settle(accA[0]);
            ws_hidden_tile<4>(accA[0], Ep, img(1, i1), h);
            ws_phase<1, 1>(W1, accB, la2);
            ws_settle(accB[0]);
            ws_hidden_tile<4>(accB[0], Ep, img(2, i2), h);
            ws_round_barrier();
        }
    } else {
        // ---------------------------------------------------------------- per-edge input gather, first-layer tiles 2-4, We tile 4
        WsUnit W[1];
        load_unit(W[0], 11 + 4);
        __syncthreads();
        f32x16 accF[1], acc[1];
        WsQ16 Q{0u, AG_Q16_EB_MIN, 0, {0u, 0u, 0u, 0u}, 0u};
        // three blocks in flight: edge indices (this round) -> the two 64-byte node rows (next round) -> features (the round after)
        int er = 0, es = 0;                // indices of block r (loaded in round r, used in round r + 1)
        float4 R[4], S[4];                 // receiver / sender rows of block r - 1 (loaded in round r, used in round r + 1)
#pragma unroll
        for (int q = 0; q < 4; ++q) R[q] = S[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
        auto pk = [](float x0, float x1) { const f32x2 v = {x0, x1}; return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2)); };
        const float4 *tab = reinterpret_cast<const float4 *>(a.edge_node_tab);
        static_assert(AG_NHIS == 4 && AG_EDGE_IN == 17, "edge_node_tab rows and the feature pieces are laid out for four history frames");
#pragma unroll 1
        for (int r = 0; r < rounds; ++r) {
            const int i0 = r - AG_WS_LAG_F, i3 = r - AG_WS_LAG_3;
            {   // We tile 4 first: the SIMD partner (an RE2 pair) starts its round with an epilogue
                const unsigned la = lds_addr3(img(2, i3));
                ws_phase<1, 2>(W, acc, la);
                ws_settle(acc[0]);
                ws_table_tile<4>(acc[0], Q, eterm_row(i3), h);
            }
            {   // features of block r - 2 from the rows loaded last round: [attrs_r | attrs_s | |g_r - g_s| | row_r[4:16] - row_s[4:16] | 1];
                // lane half h keeps slots 8q + 4h + c -> B-operand image of k16-step 0 (features 0..15) and 1 (slot 16: feature 16, 17: the bias
                // 1.0, 18..29: the fp16 residuals of features 5..16, f16_residual; the same values in the same slots as edge_encode_kernel<PrecH3>)
                float feat[24];
#pragma unroll
                for (int k = 0; k < 24; ++k) feat[k] = 0.0f;
                feat[0] = R[0].x; feat[1] = R[0].y; feat[2] = S[0].x; feat[3] = S[0].y; feat[4] = fabsf(R[0].z - S[0].z); feat[AG_EDGE_IN] = 1.0f;
                feat[5] = R[1].x - S[1].x; feat[6] = R[1].y - S[1].y; feat[7] = R[1].z - S[1].z; feat[8] = R[1].w - S[1].w;
                feat[9] = R[2].x - S[2].x; feat[10] = R[2].y - S[2].y; feat[11] = R[2].z - S[2].z; feat[12] = R[2].w - S[2].w;
                feat[13] = R[3].x - S[3].x; feat[14] = R[3].y - S[3].y; feat[15] = R[3].z - S[3].z; feat[16] = R[3].w - S[3].w;
                ws_u32x4 X, X1;
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int c2 = 0; c2 < 2; ++c2)
                        X[2 * q + c2] = pk(h ? feat[8 * q + 4 + 2 * c2] : feat[8 * q + 2 * c2], h ? feat[8 * q + 5 + 2 * c2] : feat[8 * q + 1 + 2 * c2]);
                // slots 16, 17 | 20, 21 and 18, 19 | 22, 23, then 24, 25 | 28, 29 and 26, 27 | 30, 31   (h = 0 | h = 1; slots 30, 31 stay zero)
                X1[0] = h ? pk(f16_residual(feat[7]), f16_residual(feat[8])) : pk(feat[16], feat[AG_EDGE_IN]);
                X1[1] = h ? pk(f16_residual(feat[9]), f16_residual(feat[10])) : pk(f16_residual(feat[5]), f16_residual(feat[6]));
                X1[2] = h ? pk(f16_residual(feat[15]), f16_residual(feat[16])) : pk(f16_residual(feat[11]), f16_residual(feat[12]));
                X1[3] = h ? 0u : pk(f16_residual(feat[13]), f16_residual(feat[14]));
                *reinterpret_cast<ws_u32x4 *>(&s_in0[slot_of(r - 2)][lane * 16]) = X;
                *reinterpret_cast<ws_u32x4 *>(&s_in0[slot_of(r - 2)][lane * 16 + 1024]) = X1;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) { R[q] = tab[(unsigned)er * 4u + q]; S[q] = tab[(unsigned)es * 4u + q]; }      // node rows of block r - 1
            {                                                                                                          // edge indices of block r
                const int e = (r < n_i ? gblock(r) : 0) * 32 + j;
                const bool valid = r < n_i && e < E;
                er = valid ? a.edge_recv[e] : 0;
                es = valid ? a.edge_send[e] : 0;
            }
            ws_first_layer<4, 1>(accF, lds_addr_of(&s_in0[slot_of(i0)][lane * 16]), wf, noop);
            ws_settle(accF[0]);
            ws_hidden_tile<4>(accF[0], Ep, img(0, i0), h);
            ws_round_barrier();
        }
        if (Q.nonfinite && a.status) atomicOr(a.status, 1);
    }
    h3_report(Ep.bad, a.status);
}

// ---------------------------------------------------------------------------------------------
// One propagation round at node level: fused segment reduce (aggregate_rows) or a pre-computed `agg` table,
// then the node update (model.py:299-301), then either the next round's node-level relation terms (Hr, Hs)
// or — after the last round — the decoder + clamp + integrate (model.py:306-309).
// ---------------------------------------------------------------------------------------------
// FUSE (precision mode 2 only): the round's segment reduce runs INSIDE this kernel, so the `agg` table never exists in HBM
// (-328 MB of the 2.3 GB a round moves) and the aggregate launch disappears.  The reduce keeps the standalone kernel's memory
// pattern — 20 adjacent lanes stream one node's 320-byte Eterm rows, 12 nodes per pass of the workgroup — because that pattern,
// not the MFMA lane layout, is what coalesces (the r01 fusion gathered row-per-lane in the MFMA layout: 0.52 ms vs 0.30 + 0.20).
// Its sums cross to the owning wave's B-operand registers through a 32-row LDS stage (row stride 164 floats: conflict-free
// ds_read_b128), one wave's 32 rows at a time.  The two workgroups of a CU are in different phases, so one's latency-bound
// gather overlaps the other's MFMA chain; row tiles are dealt XCD-contiguously so a graph's sender rows stay in one L2.
// Measured (C2, r02): 0.474 ms per round vs 0.292 + 0.196 separate (-3 %), 1.97 GB at 4.2 TB/s instead of 2.3 GB at 4.7: the
// bytes saved are paid back in bandwidth, and in the two-stream rollout the separate kernels co-run better (99 k vs 103 k
// graph-steps/s), so this is ag_set_option("fuse_aggregate", 2), not the default.  Keeping 8 edges or three nodes per lane in
// flight changed nothing (0.473 / 0.475 ms): the round is bandwidth-bound at what this access mix reaches, not latency-bound.
#define AG_STAGE_LD 164
template <class Prec, bool LAST, bool FUSE, bool HSQ = false, bool AQ = false>      // HSQ: the next round's sender table is written as q16 rows (mode 2); AQ: `agg` is read as q16 rows
__global__ __launch_bounds__(AG_MLP_THREADS, AG_MLP_WG_PER_CU) void node_update_kernel(AgWeights w, AgFwdArgs a)
{
    AG_LDS_DECL
    __shared__ __attribute__((aligned(16))) float stage[FUSE ? 32 * AG_STAGE_LD : 4];
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5, wave = tid >> 6;
    const int Mn = ag_rows(a);
    const int ntiles = (Mn + AG_ROWS_PER_BLOCK - 1) / AG_ROWS_PER_BLOCK;
    ChunkPipe P{LAST ? pick<Prec>(w.node_last, w.node_last_b3) : pick<Prec>(w.node_mid, w.node_mid_b3), LAST ? 16 : 15, 0, 0, lds};
    pipe_start(P);
    TileQueue q(nullptr, s_next_tile);   // ~4 row tiles per workgroup: nothing to balance, static stride
    if (FUSE) {   // XCD-contiguous deal: block b sits on XCD b % 8; give XCD x the logical ids [x*nb/8, (x+1)*nb/8) so that in every
                  // round of the grid stride one XCD works on ~nb/8 CONSECUTIVE row tiles (whole graphs)
        const int nb = gridDim.x, bid = blockIdx.x, qq = nb >> 3, rr = nb & 7, xcd = bid & 7, idx = bid >> 3;
        q.tile = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + idx;
    }
    const bool ovf = a.ovf && *a.ovf != 0;      // de-duplicated call that overflowed the compact tables: Pn / h come from the packed tables
    const float *pn_rows = ovf ? nullptr : a.pn_rows, *h_rows = ovf ? nullptr : a.h_rows;
    f32x16 agg_next[AQ ? 1 : AG_NT];      // (!FUSE) the agg rows of the row tile about to start
    AggRowQ16 aggq_next;                  // (AQ: as loaded, decoded where the operand image is built)
    const unsigned char *aggq = reinterpret_cast<const unsigned char *>(a.agg);
    bool have_next = false;
#pragma unroll 1
    while (q.tile < ntiles) {
        const int tile = q.tile;
        q.claim();
        const int g = tile * AG_ROWS_PER_BLOCK + wave * 32 + j;
        const bool valid = g < Mn;
        const int gc = valid ? g : 0;

        typename Prec::Act x, y;
        if constexpr (FUSE) {
            AgFwdArgs ar = a;
            ag_overflow_view(ar);
            const int ng = lane / AG_AGG_GROUP, c = lane - ng * AG_AGG_GROUP, f0 = ag_half_lane_feature(c);      // twenty lanes of one wave per node
            const int slot = wave * AG_AGG_NODES_PER_WAVE + ng;                                                      // 12 node slots per pass
#pragma unroll 1
            for (int grp = 0; grp < AG_MLP_WAVES; ++grp) {
#pragma unroll 1
                for (int pass = 0; pass < 3; ++pass) {                   // 3 x 12 node slots >= 32 rows
                    const int r = pass * 12 + slot;
                    if (ng < AG_AGG_NODES_PER_WAVE && r < 32) {
                        const int gn = tile * AG_ROWS_PER_BLOCK + grp * 32 + r;
                        float4 acc0 = make_float4(0.f, 0.f, 0.f, 0.f), acc1 = acc0;
                        if (gn < Mn) {
                            const int E_ = ar.self_info ? ag_edges(ar) : 0;
                            if (ar.self_info) {
                                if (a.hs_q16) ag_reduce_node_q16<AG_AGG_IN_FLIGHT, true, true>(ar, gn, c, ng * AG_AGG_GROUP, acc0, acc1, E_);
                                else ag_reduce_node_q16<AG_AGG_IN_FLIGHT, false, true>(ar, gn, c, ng * AG_AGG_GROUP, acc0, acc1, E_);
                            } else {
                                if (a.hs_q16) ag_reduce_node_q16<AG_AGG_IN_FLIGHT, true, false>(ar, gn, c, ng * AG_AGG_GROUP, acc0, acc1);
                                else ag_reduce_node_q16<AG_AGG_IN_FLIGHT, false, false>(ar, gn, c, ng * AG_AGG_GROUP, acc0, acc1);
                            }
                        }
                        if (a.agg_q16) ag_q16_roundtrip_segment(acc0, acc1);      // (the 16-bit rounding the `agg` rows of the separate kernels go through: same bits)
                        *reinterpret_cast<float4 *>(stage + r * AG_STAGE_LD + f0) = acc0;
                        *reinterpret_cast<float4 *>(stage + r * AG_STAGE_LD + f0 + 8) = acc1;
                    }
                }
                __syncthreads();
                if (wave == grp) {
#pragma unroll
                    for (int t = 0; t < AG_NT; ++t) {
                        f32x16 v;
#pragma unroll
                        for (int qd = 0; qd < 4; ++qd) {
                            const float4 u = *reinterpret_cast<const float4 *>(stage + j * AG_STAGE_LD + 32 * t + 8 * qd + 4 * h);
                            v[4 * qd] = u.x; v[4 * qd + 1] = u.y; v[4 * qd + 2] = u.z; v[4 * qd + 3] = u.w;
                        }
                        Prec::set_tile(x, t, v);
                    }
                }
                __syncthreads();
            }
        } else {
            // this row tile's agg rows: loaded during the PREVIOUS row tile's last two layers (below), except for a workgroup's first tile
            if (!have_next) {
                if constexpr (AQ) load_rowmajor_q16(aggq + (size_t)gc * (2 * AG_FP), aggq_next, h);
                else load_rowmajor(a.agg + (size_t)gc * AG_FP, agg_next, h);
            }
        }
        const size_t blk = (size_t)(tile * AG_MLP_WAVES + wave) * AG_PACK_BLOCK + h * 128 + j * 4;
        const size_t rowoff = (size_t)g * AG_FP + 4 * h;   // own row even when past Mn (padding rows)
        // Pn (+ h in round 0) come from the compact rows of the de-duplicated node encoder when it is on
        const size_t crow = pn_rows ? (size_t)a.node_row[gc] * AG_FP + 4 * h : 0;
        const ResidInit resid{pn_rows ? pn_rows + crow : a.pn + blk, h_rows ? h_rows + crow : a.h + blk, pn_rows != nullptr, h_rows != nullptr};
        resid.prefetch();      // issued before the operand split of agg below, whose ~250 VALU instructions cover part of the latency
        if constexpr (!FUSE) {
#pragma unroll
            for (int t = 0; t < AG_NT; ++t) {
                if constexpr (AQ) { f32x16 v; agg_q16_tile(aggq_next, t, h, v); Prec::set_tile(x, t, v); }
                else Prec::set_tile(x, t, agg_next[t]);
            }
        }
        // The next row tile of this workgroup (static grid stride: TileQueue without a counter): its agg rows are fetched while this tile's second
        // and third layers run, into the registers the first layer's input image has just left.
        const int tile_n = tile + (int)gridDim.x;
        auto prefetch_agg = [&]() {
            if constexpr (!FUSE) {
                have_next = tile_n < ntiles;       // workgroup-uniform
                if (have_next) {
                    const int gn = tile_n * AG_ROWS_PER_BLOCK + wave * 32 + j;
                    if constexpr (AQ) load_rowmajor_q16(aggq + (size_t)(gn < Mn ? gn : 0) * (2 * AG_FP), aggq_next, h);
                    else load_rowmajor(a.agg + (size_t)(gn < Mn ? gn : 0) * AG_FP, agg_next, h);
                } else {      // (a defined value on this path too: otherwise the previous tile's rows stay live through the whole first layer)
                    if constexpr (AQ) {
#pragma unroll
                        for (int t = 0; t < AG_NT; ++t) { aggq_next.v[t][0] = make_int4(0, 0, 0, 0); aggq_next.v[t][1] = make_int4(0, 0, 0, 0); }
                        aggq_next.ex = make_uint2(0u, 0u);
                    } else {
#pragma unroll
                        for (int t = 0; t < AG_NT; ++t)
#pragma unroll
                            for (int r = 0; r < 16; ++r) agg_next[t][r] = 0.0f;
                    }
                }
            }
        };
        if (!LAST) {
            dense<Prec, AG_F, true, false>(P, x, y, resid, PackStoreEpi{a.h + blk});   // h'
            q.publish();
            prefetch_agg();
            // Hr/Hs of the NEXT round go to the alternate tables: other workgroups of this launch may still be
            // gathering this round's Hs rows (fused aggregation reads them inside this kernel).
            dense_store<Prec, AG_F, false, false>(P, y, ZeroInit{}, RowStoreEpi{a.hr_out + rowoff});
            if constexpr (HSQ) dense_store<Prec, AG_F, false, false>(P, y, ZeroInit{}, RowStoreQ16Epi{reinterpret_cast<unsigned char *>(a.hs_out) + (size_t)g * (2 * AG_FP), h, a.status});
            else dense_store<Prec, AG_F, false, false>(P, y, ZeroInit{}, RowStoreEpi{a.hs_out + rowoff});
        } else {
            dense<Prec, AG_F, true, false>(P, x, y, resid);   // particle_effect'
            q.publish();
            dense<Prec, AG_F, true, true>(P, y, x, ZeroInit{});    // linear_0 + ReLU
            dense<Prec, AG_F, true, true>(P, x, y, ZeroInit{});    // linear_1 + ReLU
            prefetch_agg();                                        // (both activation images are live until here: the decoder ping-pongs them)
            f32x16 m;
            Prec::template layer<AG_F, 1, false, true>(P, y, ZeroInit{}, NoEpi{},      // linear_2 -> rows 0..2 of tile 0
                                                       [&](int, const f32x16 &v) { m = v; });
            const int go = a.row_orig ? a.row_orig[gc] : gc;      // (shared-state rollout: the node this compact row stands for; predictions are stored by row)
            const int b = go / a.N, i = go - b * a.N;
            if (valid && h == 0 && i < a.n_p) {
                const float *cur = a.state + (((size_t)b * AG_NHIS + (AG_NHIS - 1)) * a.N + i) * 3;
                float *pm = a.pred_motion + (a.row_orig ? (size_t)gc : (size_t)b * a.n_p + i) * 3;
                float *pp = a.pred_pos + (a.row_orig ? (size_t)gc : (size_t)b * a.n_p + i) * 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float mv = m[c];
                    pm[c] = mv;
                    pp[c] = cur[c] + fminf(fmaxf(mv, -a.clamp), a.clamp);   // model.py:309
                }
            }
        }
        q.next();
    }
}

// =====================================================================================================================
// Weight-STATIONARY node update of the rounds before the last (precision modes 1 / 2; ag_set_option("node_stationary", 1)).
//
// The streaming node_update_kernel re-reads its 15 weight chunks (300 KB) from L2 for every 128 rows: 0.60 GB per launch at C2 beside 0.74 GB
// of tables, and the timing-only build without those copies is 17 % shorter (profiles/r05_node_update_ablation.txt).  Here ONE 256-thread
// workgroup per CU (one wave per SIMD, 496 registers per lane) keeps all 15 (layer, out-tile) units' split-bf16 A operands in registers for
// the whole launch — wave w owns out-tile w of each of the three layers, the fifth tiles go one each to waves 3 (first layer), 1 (Hr), 2 (Hs) — and 32-row blocks
// flow through as a two-stage pipeline with ONE barrier per block:
//     phase p:   first layer (h' = relu(W_pp[:, F:] agg + Pn + h)) of block p        input set X[p & 1]  -> h' to HBM + output set Y[p & 1]
//                Hr / Hs layers of block p - 1 (one pass over the set for both)        input set Y[(p - 1) & 1]
//                agg rows of block p + 1: loaded at the top of the phase, split into bf16 hi / lo and written to X[(p + 1) & 1] at its end
// A set is the 160 x 32 activation block in the B-operand layout of the next layer (lane (j, h) writes the bytes lane (j, h) of a consumer
// reads): [10 k16-steps][hi | lo][64 lanes][8 bf16] = 20 KB.  Every accumulator sees its k16-steps in ascending order with the products in
// the streaming kernel's order (lo.hi, hi.lo, hi.hi) and is initialised the same way, so the results equal node_update_kernel bit for bit.
// =====================================================================================================================
#define AG_NWS_SET 20480
struct NwsUnit { bf16x8 hi[10], lo[10]; };

// ACC: the unit lives in the accumulation-register half of the file (the matrix instructions read their A operand from there directly): a
// wave's first three units; the fourth (waves 1-3) sits in architectural registers.  Left to the compiler (builtin MFMAs), the 320 weight
// registers end up wherever it likes, with ~1 750 accumulation-register moves per block and spills.
template <bool ACC>
__device__ __forceinline__ void nws_load_unit(NwsUnit &W, const float4 *chunk, int lane)
{
#pragma unroll
    for (int u = 0; u < 10; ++u) {
        W.hi[u] = *reinterpret_cast<const bf16x8 *>(chunk + (2 * u) * 64 + lane);
        W.lo[u] = *reinterpret_cast<const bf16x8 *>(chunk + (2 * u + 1) * 64 + lane);
    }
#pragma unroll
    for (int u = 0; u < 10; ++u) {      // opaque (after ALL the loads): must stay in registers, cannot be re-loaded in the loop
        if constexpr (ACC) asm volatile("" : "+a"(W.hi[u]), "+a"(W.lo[u]));
        else asm volatile("" : "+v"(W.hi[u]), "+v"(W.lo[u]));
    }
}
// one k16-step of one unit: acc += lo.xh + hi.xl + hi.xh in the streaming kernel's order; inline asm so that the A operands are read where the
// unit lives.  (s_nop 1: two wait states between a VALU / LDS-load written operand and the matrix instruction that reads it.)
template <bool ACC>
__device__ __forceinline__ void nws_mfma3(f32x16 &acc, const bf16x8 &wh, const bf16x8 &wl, const bf16x8 &xh, const bf16x8 &xl)
{
    if constexpr (ACC)
        asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %3, %0\n\tv_mfma_f32_32x32x16_bf16 %0, %2, %4, %0\n\tv_mfma_f32_32x32x16_bf16 %0, %2, %3, %0"
                     : "+v"(acc) : "a"(wl), "a"(wh), "v"(xh), "v"(xl));
    else
        asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %3, %0\n\tv_mfma_f32_32x32x16_bf16 %0, %2, %4, %0\n\tv_mfma_f32_32x32x16_bf16 %0, %2, %3, %0"
                     : "+v"(acc) : "v"(wl), "v"(wh), "v"(xh), "v"(xl));
}
// results of asm MFMAs are not interlocked against the compiler's VALU readers: 8 passes + 4 states, with margin
__device__ __forceinline__ void nws_settle(f32x16 &a) { asm volatile("s_nop 15\n\ts_nop 7" : "+v"(a)); }
// up to three units over ONE input set (la: LDS byte address of this lane's 16 bytes of k16-step 0): the operand pair of step u + 2 is read
// while step u's matrix instructions run (inline-asm reads with counted waits: left to the compiler every step waited a full LDS round trip)
template <int NU, bool ACC0, bool ACC1, bool ACC2>
__device__ __forceinline__ void nws_layer(f32x16 (&acc)[NU], const NwsUnit &W0, const NwsUnit &W1, const NwsUnit &W2, unsigned la)
{
    bf16x8 xq[3][2];
    lds_read16<0>(xq[0][0], la);
    lds_read16<1024>(xq[0][1], la);
    lds_read16<2048>(xq[1][0], la);
    lds_read16<3072>(xq[1][1], la);
    static_for<0, 10>([&](auto UU) {
        constexpr int u = decltype(UU)::value;
        if constexpr (u + 2 < 10) {
            lds_read16<(2 * (u + 2)) * 1024>(xq[(u + 2) % 3][0], la);
            lds_read16<(2 * (u + 2) + 1) * 1024>(xq[(u + 2) % 3][1], la);
        }
        constexpr int ahead = (9 - u) < 2 ? (9 - u) : 2;
        lds_wait_pair<2 * ahead>(xq[u % 3][0], xq[u % 3][1]);
        nws_mfma3<ACC0>(acc[0], W0.hi[u], W0.lo[u], xq[u % 3][0], xq[u % 3][1]);
        if constexpr (NU >= 2) nws_mfma3<ACC1>(acc[1], W1.hi[u], W1.lo[u], xq[u % 3][0], xq[u % 3][1]);
        if constexpr (NU >= 3) nws_mfma3<ACC2>(acc[NU - 1], W2.hi[u], W2.lo[u], xq[u % 3][0], xq[u % 3][1]);
    });
#pragma unroll
    for (int k = 0; k < NU; ++k) nws_settle(acc[k]);
}
// k16-step `step` of a set from eight fp32 values in accumulator order (PrecB3::set_half's conversion)
__device__ __forceinline__ void nws_write_half(unsigned char *set_lane, int step, const float (&x)[8])
{
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 H, L;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const unsigned hp = cvt_pk_bf16(x[2 * w], x[2 * w + 1]);
        const float h0 = __uint_as_float(hp << 16), h1 = __uint_as_float(hp & 0xffff0000u);
        H[w] = hp;
        L[w] = cvt_pk_bf16(x[2 * w] - h0, x[2 * w + 1] - h1);
    }
    *reinterpret_cast<u32x4 *>(set_lane + (2 * step) * 1024) = H;
    *reinterpret_cast<u32x4 *>(set_lane + (2 * step + 1) * 1024) = L;
}

template <int WAVE, bool HSQ, bool AQ>
__device__ __forceinline__ void nws_wave(const AgWeights &w, const AgFwdArgs &a, unsigned char *sX, unsigned char *sY)
{
    constexpr int N1 = WAVE == 3 ? 2 : 1, N2 = WAVE == 1 ? 2 : 1, N3 = WAVE == 2 ? 2 : 1;      // the fifth out-tiles: layer 1's to wave 3, Hr's to wave 1, Hs's to wave 2
    constexpr int N23 = N2 + N3;
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    const int Mn = ag_rows(a);
    const int nblk = (Mn + 31) / 32;
    const int n_i = ((int)blockIdx.x < nblk) ? (nblk - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    const bool ovf = a.ovf && *a.ovf != 0;
    const float *pn_rows = ovf ? nullptr : a.pn_rows, *h_rows = ovf ? nullptr : a.h_rows;
    NwsUnit U1[N1], U2[N2], U3[N3];
    const float4 *ws = w.node_mid_b3;
    // placement: a wave's units U1[0], U2[0], U3[0] sit in accumulation registers (240), its fifth-tile unit (waves 1-3) in architectural ones
    nws_load_unit<true>(U1[0], ws + (size_t)(0 + WAVE) * AG_CHUNK_F4, lane);
    nws_load_unit<true>(U2[0], ws + (size_t)(5 + WAVE) * AG_CHUNK_F4, lane);
    nws_load_unit<true>(U3[0], ws + (size_t)(10 + WAVE) * AG_CHUNK_F4, lane);
    if constexpr (N1 == 2) nws_load_unit<false>(U1[1], ws + (size_t)(0 + 4) * AG_CHUNK_F4, lane);
    if constexpr (N2 == 2) nws_load_unit<false>(U2[1], ws + (size_t)(5 + 4) * AG_CHUNK_F4, lane);
    if constexpr (N3 == 2) nws_load_unit<false>(U3[1], ws + (size_t)(10 + 4) * AG_CHUNK_F4, lane);
    const int t1[2] = {WAVE, 4}, t2[2] = {WAVE, 4}, t3[2] = {WAVE, 4};
    auto gblock = [&](int i) { return (int)blockIdx.x + i * (int)gridDim.x; };
    // input staging: this wave converts k16-steps WAVE, WAVE + 4, WAVE + 8 (< 10) of the next block's agg rows
    constexpr int NS = WAVE < 2 ? 3 : 2;
    float4 raw[AQ ? 1 : NS][2];
    int4 rawq[AQ ? NS : 1];      // AQ (`agg` as q16 rows): k16-step u of lane (j, h) is the 16 bytes at 64 (u >> 1) + 32 h + 16 (u & 1) of row j
    uint2 rawex;
    auto stage_load = [&](int i) {
        const size_t g0 = (size_t)(i < n_i ? gblock(i) : 0) * 32 + j;
        const size_t g = g0 < (size_t)Mn ? g0 : 0;       // rows past B*N (last block): the reduce never wrote them — node 0's row instead of stale workspace bytes
        if constexpr (AQ) {
            const unsigned char *rowb = reinterpret_cast<const unsigned char *>(a.agg) + g * (2 * AG_FP);
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                const int u = WAVE + 4 * k;
                rawq[k] = *reinterpret_cast<const int4 *>(rowb + 64 * (u >> 1) + 32 * h + 16 * (u & 1));
            }
            rawex = *reinterpret_cast<const uint2 *>(rowb + 280);
        } else {
            const float *row = a.agg + g * AG_FP + 4 * h;
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                const int u = WAVE + 4 * k;
                raw[k][0] = *reinterpret_cast<const float4 *>(row + 16 * u);
                raw[k][1] = *reinterpret_cast<const float4 *>(row + 16 * u + 8);
            }
        }
    };
    auto stage_write = [&](unsigned char *set) {
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            if constexpr (AQ) {
                const int u = WAVE + 4 * k;
                float x[8];
                agg_q16_decode8(rawq[k], agg_q16_tile_scale(rawex, u >> 1), x);
                if (u == 9) {      // features 150..159 do not exist (their positions hold the exponent bytes): h = 0 owns 144..147 | 152..155, h = 1 148..151 | 156..159
                    if (h) { x[2] = 0.0f; x[3] = 0.0f; }
                    x[4] = 0.0f; x[5] = 0.0f; x[6] = 0.0f; x[7] = 0.0f;
                }
                nws_write_half(set + lane * 16, u, x);
            } else {
                const float x[8] = {raw[k][0].x, raw[k][0].y, raw[k][0].z, raw[k][0].w, raw[k][1].x, raw[k][1].y, raw[k][1].z, raw[k][1].w};
                nws_write_half(set + lane * 16, WAVE + 4 * k, x);
            }
        }
    };
    // residual Pn + h of this wave's first-layer tiles, loaded one phase ahead; the compact-row index of a block (node de-duplication) one
    // phase earlier still, so that no load in the loop body depends on another load of the same phase
    f32x16 rh[N1];
    const float *pn_ptr = nullptr;
    int idx_raw = 0;                                     // node_row of block i + 1, in flight during a phase
    const int32_t *idx_tab = pn_rows ? a.node_row : a.row_ptr;      // (without de-duplication: any valid table, the value is not used)
    size_t crow_cur = 0;                                 // compact-row offset of the block whose residual is loaded next (a VALUE)
    auto idx_load = [&](int i) {
        const size_t g = (size_t)(i < n_i ? gblock(i) : 0) * 32 + j;
        idx_raw = idx_tab[g < (size_t)Mn ? g : 0];       // (unconditional: a `cond ? load : 0` costs a conservative wait at the top of the loop body)
    };
    auto resid_load = [&](int i) {
        const int blk = i < n_i ? gblock(i) : 0;
        const size_t pk = (size_t)blk * AG_PACK_BLOCK + h * 128 + j * 4;
        pn_ptr = pn_rows ? pn_rows + crow_cur : a.pn + pk;      // Pn (two L2-hot compact rows per sample in the reference's rollouts) is loaded where it is
#pragma unroll                                                 // added: 16 registers per tile less in flight through the phase
        for (int k = 0; k < N1; ++k) ResidInit::load_tile<true>(h_rows ? h_rows + crow_cur : a.h + pk, h_rows != nullptr, t1[k], rh[k]);
    };
    // Nothing that is still in flight is carried over the loop's back edge (the compiler's wait-count analysis answers a loop-carried pending load
    // with a full vmcnt(0) at the top of the body, previous phase's stores included): a phase issues the next block's loads at its top and turns
    // them into VALUES at its end — the staged operand set in LDS and the first layer's accumulator initialisation Pn + h in `acc1`.
    f32x16 acc1[N1];
    auto resid_to_acc = [&]() {
#pragma unroll
        for (int k = 0; k < N1; ++k) {
            f32x16 rp;      // (loaded HERE: hoisted in front of the Hr / Hs pass it costs 16 registers per tile through that pass — spills, 0.171 ms)
            ResidInit::load_tile(pn_ptr, pn_rows != nullptr, t1[k], rp);
            ResidInit::zero_pad(pn_rows != nullptr, t1[k], rp);
            ResidInit::zero_pad(h_rows != nullptr, t1[k], rh[k]);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[k][r] = rp[r] + rh[k][r];
        }
    };
    idx_load(0);
    crow_cur = (size_t)idx_raw * AG_FP + 4 * h;
    stage_load(0);
    resid_load(0);
    idx_load(1);
    stage_write(sX);
    resid_to_acc();
    crow_cur = (size_t)idx_raw * AG_FP + 4 * h;          // block 1
    __syncthreads();
#pragma unroll 1
    for (int p = 0; p <= n_i; ++p) {
        unsigned char *X = sX + (p & 1) * AG_NWS_SET, *Xn = sX + ((p + 1) & 1) * AG_NWS_SET;
        unsigned char *Y = sY + (p & 1) * AG_NWS_SET, *Yp = sY + ((p + 1) & 1) * AG_NWS_SET;
        if (p + 1 < n_i) { stage_load(p + 1); resid_load(p + 1); idx_load(p + 2); }
        if (p < n_i) {                                   // first layer of block p
            const int blk = gblock(p);
            nws_layer<N1, true, false, false>(acc1, U1[0], U1[N1 - 1], U1[N1 - 1], lds_addr_of(X) + lane * 16);
            const PackStoreEpi store{a.h + (size_t)blk * AG_PACK_BLOCK + h * 128 + j * 4};
#pragma unroll
            for (int k = 0; k < N1; ++k) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc1[k][r] = relu1(acc1[k][r]);
                store(t1[k], acc1[k]);
#pragma unroll
                for (int sh = 0; sh < 2; ++sh) {
                    const float x[8] = {acc1[k][8 * sh], acc1[k][8 * sh + 1], acc1[k][8 * sh + 2], acc1[k][8 * sh + 3],
                                        acc1[k][8 * sh + 4], acc1[k][8 * sh + 5], acc1[k][8 * sh + 6], acc1[k][8 * sh + 7]};
                    nws_write_half(Y + lane * 16, 2 * t1[k] + sh, x);
                }
            }
        }
        // Hr / Hs of block p - 1: one pass over the set for both layers' units.  The next block's loads are turned into values BETWEEN this pass and
        // its stores: the compiler's counted waits then see only the first layer's (by now old) stores behind the loads, not fresh ones.
        f32x16 acc[N23];
        const size_t g = (size_t)gblock(p >= 1 ? p - 1 : 0) * 32 + j;
        if (p >= 1) {
#pragma unroll
            for (int k = 0; k < N23; ++k)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[k][r] = 0.0f;
            const unsigned la = lds_addr_of(Yp) + lane * 16;
            // unit order in acc: U2[0], (U2[1]), U3[0], (U3[1])
            if constexpr (N2 == 2) nws_layer<3, true, false, true>(acc, U2[0], U2[1], U3[0], la);
            else if constexpr (N3 == 2) nws_layer<3, true, true, false>(acc, U2[0], U3[0], U3[1], la);
            else nws_layer<2, true, true, false>(acc, U2[0], U3[0], U3[0], la);
        }
        if (p + 1 < n_i) { stage_write(Xn); resid_to_acc(); crow_cur = (size_t)idx_raw * AG_FP + 4 * h; }
        if (p >= 1) {
            const RowStoreEpi sr{a.hr_out + g * AG_FP + 4 * h};
#pragma unroll
            for (int k = 0; k < N2; ++k) sr(t2[k], acc[k]);
            if constexpr (HSQ) {
                const RowStoreQ16Epi sq{reinterpret_cast<unsigned char *>(a.hs_out) + g * (2 * AG_FP), h, a.status};
#pragma unroll
                for (int k = 0; k < N3; ++k) sq(t3[k], acc[N2 + k]);
            } else {
                const RowStoreEpi ss{a.hs_out + g * AG_FP + 4 * h};
#pragma unroll
                for (int k = 0; k < N3; ++k) ss(t3[k], acc[N2 + k]);
            }
        }
        ws_round_barrier();      // LDS side only: __syncthreads() would also drain this phase's table stores (vmcnt(0)) with the whole CU waiting
    }
}

template <bool HSQ, bool AQ>
__global__ __launch_bounds__(256, 1) void node_update_nws_kernel(AgWeights w, AgFwdArgs a)
{
    __shared__ __attribute__((aligned(16))) unsigned char sX[2 * AG_NWS_SET], sY[2 * AG_NWS_SET];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave == 0) nws_wave<0, HSQ, AQ>(w, a, sX, sY);
    else if (wave == 1) nws_wave<1, HSQ, AQ>(w, a, sX, sY);
    else if (wave == 2) nws_wave<2, HSQ, AQ>(w, a, sX, sY);
    else nws_wave<3, HSQ, AQ>(w, a, sX, sY);
}

// =====================================================================================================
// Training path: the dense stacks of DynamicsPredictor.forward and their backward on the
// same fused-layer machinery, in either arithmetic: exact fp32 MFMA (PrecF32) or split-bf16 (PrecB3: 2^-17 relative operand
// error, measured well inside the 2e-4 gradient gate of tests/golden/train_rope.npz, ~5x the MFMA rate; default).
//   forward  y_l = act_l(W_l y_{l-1} + b_l), l = 0..L-1, every y_l stored (row-major [rows][160]) for the backward
//   backward dz_l = dy_l (.) [y_l > 0] (ReLU layers), dy_{l-1} = dz_l W_l — again a chain of fused layers, with the
//            TRANSPOSED weights as the MFMA A operand and the ReLU mask applied in registers from the saved y_{l-1};
//            every dz_l is stored: dW_l = dz_l^T y_{l-1} and db_l = sum_rows dz_l are plain library GEMMs / reductions
//            over those tables (torch.mm in adaptigraph_amd/train_ops.py).
// Replaces the F.linear chains of train.py:90-112's forward/backward for relation_encoder + W_rp[:, :F] (KIND_EDGE),
// particle_encoder (KIND_NODE) and non_rigid_predictor (KIND_DEC); the activations never leave registers between
// layers in either direction.
// =====================================================================================================
template <int KIND> struct ChainShape;
template <> struct ChainShape<0> { static constexpr int L = 4, KF = AG_EDGE_IN + 1, RELU = 0x7; static constexpr bool NARROW = true; };   // RE0 RE1 RE2 We
template <> struct ChainShape<1> { static constexpr int L = 3, KF = AG_NODE_IN_MAX, RELU = 0x7; static constexpr bool NARROW = true; };   // PE0 PE1 PE2
template <> struct ChainShape<2> { static constexpr int L = 3, RELU = 0x3; static constexpr bool NARROW = false; };                      // D0 D1 D2 (wide input: no KF)

struct AgChainArgs {
    const float *x;          // forward input: [rows][d_in] dense (narrow kinds) or [rows][160] (KIND_DEC)
    const float4 *w;         // packed fp32 chunk stream (ag_train_pack): forward order, or transposed in backward order
    float *y[4];             // per-layer outputs, row-major [rows_pad][160] (forward: written; backward: read)
    const float *dy;         // backward: gradient w.r.t. y[L-1], [rows_pad][160]
    float *dz[4];            // backward: per-layer pre-activation gradients, [rows_pad][160] (written)
    float *dx;               // backward: gradient w.r.t. x, same shape as x (written), may be null
    long long rows;
    int d_in;
};

struct MaskStore {   // sink of a backward layer: dz = dy (.) [y > 0] (if MASK), stored row-major and handed on
    const float *yrow; float *dzrow; bool mask;
    __device__ __forceinline__ f32x16 operator()(int ti, const f32x16 &v) const
    {
        f32x16 r = v;
        if (mask) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 y = *reinterpret_cast<const float4 *>(yrow + 32 * ti + 8 * q);
                r[4 * q] = y.x > 0.f ? r[4 * q] : 0.f; r[4 * q + 1] = y.y > 0.f ? r[4 * q + 1] : 0.f;
                r[4 * q + 2] = y.z > 0.f ? r[4 * q + 2] : 0.f; r[4 * q + 3] = y.w > 0.f ? r[4 * q + 3] : 0.f;
            }
        }
        RowStoreEpi{dzrow}(ti, r);
        return r;
    }
};

template <int KIND, class Prec>
__global__ __launch_bounds__(AG_MLP_THREADS, AG_MLP_WG_PER_CU) void chain_forward_kernel(AgChainArgs a)
{
    typedef ChainShape<KIND> S;
    AG_LDS_DECL
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5, wave = tid >> 6;
    const long long ntiles = (a.rows + AG_ROWS_PER_BLOCK - 1) / AG_ROWS_PER_BLOCK;
    ChunkPipe P{a.w, (S::NARROW ? 1 : AG_NT) + AG_NT * (S::L - 1), 0, 0, lds};
    pipe_start(P);
    (void)s_next_tile;
#pragma unroll 1
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long g = tile * AG_ROWS_PER_BLOCK + wave * 32 + j;
        const long long gc = g < a.rows ? g : 0;
        const size_t rowoff = (size_t)g * AG_FP + 4 * h;      // own row even past `rows` (tables are padded to whole row tiles)
        typename Prec::Act x, y;
        if constexpr (S::NARROW) {
            f32x16 in0;
#pragma unroll
            for (int r = 0; r < 16; ++r) in0[r] = 0.0f;
#pragma unroll
            for (int q = 0; q < (S::KF + 7) / 8; ++q)
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const int k = 8 * q + 4 * h + p;
                    in0[4 * q + p] = k < a.d_in ? a.x[(size_t)gc * a.d_in + k] : (k == a.d_in ? 1.0f : 0.0f);   // bias column
                }
            Prec::set_tile(x, 0, in0);
            Prec::template layer_first<S::KF>(P, x, [&](int ti, const f32x16 &v) { RowStoreEpi{a.y[0] + rowoff}(ti, v); Prec::set_tile(y, ti, v); });
        } else {
            f32x16 v[AG_NT];
            load_rowmajor(a.x + (size_t)gc * AG_FP, v, h);
#pragma unroll
            for (int t = 0; t < AG_NT; ++t) Prec::set_tile(x, t, v[t]);
            if constexpr (S::RELU & 1) dense<Prec, AG_F, true, true>(P, x, y, ZeroInit{}, RowStoreEpi{a.y[0] + rowoff});
            else dense<Prec, AG_F, false, true>(P, x, y, ZeroInit{}, RowStoreEpi{a.y[0] + rowoff});
        }
        static_for<1, S::L>([&](auto LI) {
            constexpr int l = decltype(LI)::value;
            auto &in = (l & 1) ? y : x;
            auto &out = (l & 1) ? x : y;
            if constexpr ((S::RELU >> l) & 1) dense<Prec, AG_F, true, true>(P, in, out, ZeroInit{}, RowStoreEpi{a.y[l] + rowoff});
            else dense<Prec, AG_F, false, true>(P, in, out, ZeroInit{}, RowStoreEpi{a.y[l] + rowoff});
        });
    }
}

template <int KIND, class Prec>
__global__ __launch_bounds__(AG_MLP_THREADS, AG_MLP_WG_PER_CU) void chain_backward_kernel(AgChainArgs a)
{
    typedef ChainShape<KIND> S;
    AG_LDS_DECL
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5, wave = tid >> 6;
    const long long ntiles = (a.rows + AG_ROWS_PER_BLOCK - 1) / AG_ROWS_PER_BLOCK;
    // transposed stream, in the order the backward consumes it: W_{L-1}^T ... W_1^T (5 chunks each), then W_0^T
    // (one 32-row tile when the input is narrow, else 5)
    ChunkPipe P{a.w, AG_NT * (S::L - 1) + (S::NARROW ? 1 : AG_NT), 0, 0, lds};
    pipe_start(P);
    (void)s_next_tile;
#pragma unroll 1
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long g = tile * AG_ROWS_PER_BLOCK + wave * 32 + j;
        const bool valid = g < a.rows;
        const size_t rowoff = (size_t)g * AG_FP + 4 * h;
        typename Prec::Act x, y;
        {   // dz_{L-1} = dy (.) [y_{L-1} > 0]
            f32x16 v[AG_NT];
            load_rowmajor(a.dy + (size_t)g * AG_FP, v, h);
            const MaskStore ms{a.y[S::L - 1] + rowoff, a.dz[S::L - 1] + rowoff, ((S::RELU >> (S::L - 1)) & 1) != 0};
#pragma unroll
            for (int t = 0; t < AG_NT; ++t) Prec::set_tile(x, t, ms(t, v[t]));
        }
        static_for<1, S::L>([&](auto LI) {           // dz_{l-1} = (dz_l W_l) (.) [y_{l-1} > 0],  l = L-1 .. 1
            constexpr int l = S::L - decltype(LI)::value;       // L-1, L-2, ..., 1
            constexpr bool odd = (S::L - 1 - l) & 1;
            auto &in = odd ? y : x;
            auto &out = odd ? x : y;
            const MaskStore ms{a.y[l - 1] + rowoff, a.dz[l - 1] + rowoff, ((S::RELU >> (l - 1)) & 1) != 0};
            Prec::template layer<AG_F, AG_NT, false, false>(P, in, ZeroInit{}, NoEpi{}, [&](int ti, const f32x16 &v) { Prec::set_tile(out, ti, ms(ti, v)); });
        });
        auto &last = ((S::L - 1) & 1) ? y : x;          // dz_0
        if constexpr (S::NARROW) {                       // dx = dz_0 W_0: the d_in <= 24 input columns are rows 0.. of ONE out-tile
            f32x16 m;
            Prec::template layer<AG_F, 1, false, false>(P, last, ZeroInit{}, NoEpi{}, [&](int, const f32x16 &v) { m = v; });
            if (a.dx && valid) {
#pragma unroll
                for (int q = 0; q < 3; ++q)
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const int k = 8 * q + 4 * h + p;
                        if (k < a.d_in) a.dx[(size_t)g * a.d_in + k] = m[4 * q + p];
                    }
            }
        } else {
            dense_store<Prec, AG_F, false, false>(P, last, ZeroInit{}, RowStoreEpi{a.dx + rowoff});
        }
    }
}

// Device-side weight packing for the training chains (weights change every optimiser step, so the host-side packer of
// ag_model_create is not an option): writes fp32 chunk images (layout: ag_common.h) of
//   op(W)[o][k] = transposed ? W[k * ld + col0 + o] : W[o * ld + col0 + k],  o < n_out, k < n_in;  column n_in = bias[o]
// `compact`: the one-chunk first-layer image [5 tiles][32 rows][32 floats]; else n_tiles standard images.
__global__ __launch_bounds__(256) void train_pack_kernel(const float *W, const float *bias, int n_out, int n_in, int ld, int col0,
                                                         int transposed, int compact, int n_tiles, float *dst)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int total = (compact ? 1 : n_tiles) * AG_CHUNK_FLOATS;
    if (t >= total) return;
    int tile, i, c;
    if (compact) { tile = t / 1024; i = (t % 1024) / 32; c = t % 32; }
    else { tile = t / AG_CHUNK_FLOATS; i = (t % AG_CHUNK_FLOATS) / AG_WSTRIDE; c = t % AG_WSTRIDE; }
    const int k = 4 * ((c >> 2) ^ ((i >> 1) & 7)) + (c & 3), o = 32 * tile + i;
    float v = 0.0f;
    if (o < n_out) {
        if (k < n_in) v = transposed ? W[(size_t)k * ld + col0 + o] : W[(size_t)o * ld + col0 + k];
        else if (k == n_in && bias) v = bias[o];
    }
    dst[t] = v;
}

// the same layers as split-bf16 fragment images (PrecB3): one thread per weight, writing its hi and lo halves;
//   standard image [10 steps u][hi|lo][64 lanes (i, h)][8 slots e], slot e = column 16u + 8(e>>2) + 4h + (e&3);
//   compact first-layer image [5 tiles][NU][hi|lo][64][8] with NU = ceil((n_in + 1) / 16) <= 2
__global__ __launch_bounds__(256) void train_pack_b3_kernel(const float *W, const float *bias, int n_out, int n_in, int ld, int col0,
                                                            int transposed, int compact, int n_tiles, float *dst)
{
    const int NU = compact ? (n_in + 1 + 15) / 16 : 10;
    const int per_tile = NU * 512;                            // (u, lane, e) triples per 32-row tile
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= (compact ? AG_NT : n_tiles) * per_tile) return;
    const int tile = t / per_tile, r = t % per_tile, u = r / 512, lane = (r % 512) / 8, e = r % 8;
    const int i = lane & 31, h = lane >> 5, k = 16 * u + 8 * (e >> 2) + 4 * h + (e & 3), o = 32 * tile + i;
    float v = 0.0f;
    if (o < n_out) {
        if (k < n_in) v = transposed ? W[(size_t)k * ld + col0 + o] : W[(size_t)o * ld + col0 + k];
        else if (k == n_in && bias) v = bias[o];
    }
    const unsigned hp = cvt_pk_bf16(v, 0.0f) & 0xffffu;
    const unsigned lp = cvt_pk_bf16(v - __uint_as_float(hp << 16), 0.0f) & 0xffffu;
    unsigned short *cb = reinterpret_cast<unsigned short *>(dst);
    const size_t base = compact ? (size_t)((tile * NU + u) * 2) * 512 : (size_t)tile * (AG_CHUNK_FLOATS * 2) + (size_t)(2 * u) * 512;
    cb[base + lane * 8 + e] = (unsigned short)hp;
    cb[base + 512 + lane * 8 + e] = (unsigned short)lp;
}

}  // namespace


static inline int grid_for(int rows, int max_blocks)
{
    const int tiles = (rows + AG_ROWS_PER_BLOCK - 1) / AG_ROWS_PER_BLOCK;
    return tiles < max_blocks ? (tiles > 0 ? tiles : 1) : max_blocks;
}

void ag_launch_send_remap(const AgFwdArgs &a, hipStream_t s)
{
    if (a.e_cap <= 0) return;
    const int blocks = (a.e_cap + 255) / 256;
    hipLaunchKernelGGL(send_remap_kernel, dim3(blocks < 4096 ? blocks : 4096), dim3(256), 0, s, a);
}

void ag_launch_node_encode(const AgWeights &w, const AgFwdArgs &a, hipStream_t s)
{
    if (a.dedup) {
        hipLaunchKernelGGL(node_classify_kernel, dim3((a.B + 3) / 4), dim3(256), 0, s, a);
        const dim3 gridc(grid_for(a.rows_c, a.max_blocks)), blockc(AG_MLP_THREADS);      // worst case every row is private; workgroups past the list exit
        if (a.precision == AG_PREC_B3) hipLaunchKernelGGL((node_encode_kernel<PrecB3, true>), gridc, blockc, 0, s, w, a);
        else hipLaunchKernelGGL((node_encode_kernel<PrecF32, true>), gridc, blockc, 0, s, w, a);
        return;
    }
    const dim3 grid(grid_for(a.B * a.N, a.max_blocks)), block(AG_MLP_THREADS);
    if (a.precision == AG_PREC_B3) hipLaunchKernelGGL((node_encode_kernel<PrecB3, false>), grid, block, 0, s, w, a);
    else hipLaunchKernelGGL((node_encode_kernel<PrecF32, false>), grid, block, 0, s, w, a);
}


void ag_launch_node_encode_fallback(const AgWeights &w, const AgFwdArgs &a, hipStream_t s)
{
    const dim3 grid(grid_for(a.B * a.N, a.max_blocks)), block(AG_MLP_THREADS);
    if (a.precision == AG_PREC_B3) hipLaunchKernelGGL((node_encode_kernel<PrecB3, false>), grid, block, 0, s, w, a);
    else hipLaunchKernelGGL((node_encode_kernel<PrecF32, false>), grid, block, 0, s, w, a);
}

void ag_launch_edge_encode(const AgWeights &w, const AgFwdArgs &a, hipStream_t s)
{
    if (a.e_cap <= 0) return;
    const dim3 block(AG_MLP_THREADS);
    const int e_max = a.e_cap + a.self_rows;      // upper bound of the rows this launch encodes (the true count is on the device)
    if (a.precision == AG_PREC_B3 && a.eterm_half && a.edge_products == 2) {     // mode 2: two fp16 products per k16-step, three workgroups per CU
        if (a.edge_ws && a.n_inst <= 1 && (long long)a.B * a.N * 4 < 0x7fffffffLL) {        // weight-stationary: one workgroup per CU, 32-edge blocks
            const int blocks = (e_max + 31) / 32, slots = a.ws_blocks;
            const int nb_tab = a.tab_done ? 0 : (a.B * a.N + 255) / 256;
            const int nb_map = a.dedup && !a.remap_done ? ((a.e_cap + 1023) / 1024 < 4096 ? (a.e_cap + 1023) / 1024 : 4096) : 0;      // four edges per thread
            if (nb_tab + nb_map > 0) hipLaunchKernelGGL(edge_node_tab_kernel, dim3(nb_tab + nb_map), dim3(256), 0, s, a, nb_tab);
            hipLaunchKernelGGL(edge_encode_ws_kernel, dim3(blocks < slots ? blocks : (slots > 0 ? slots : 1)), dim3(512), 0, s, w, a);   // (always eight waves, whatever AG_MLP_THREADS is)
            return;
        }
        const dim3 grid(grid_for(e_max, a.max_blocks / AG_MLP_WG_PER_CU * AG_H3_WG_PER_CU));
        hipLaunchKernelGGL(edge_encode_kernel<PrecH3>, grid, block, 0, s, w, a);
        return;
    }
    const dim3 grid(grid_for(e_max, a.max_blocks));
    if (a.precision == AG_PREC_B3) hipLaunchKernelGGL(edge_encode_kernel<PrecB3>, grid, block, 0, s, w, a);
    else hipLaunchKernelGGL(edge_encode_kernel<PrecF32>, grid, block, 0, s, w, a);
}

void ag_launch_node_update(const AgWeights &w, const AgFwdArgs &a, int last, hipStream_t s)
{
    const dim3 grid(grid_for(a.B * a.N, a.max_blocks)), block(AG_MLP_THREADS);
    if (a.precision == AG_PREC_B3 && !last && a.node_ws && !a.fuse_agg) {      // weight-stationary kernel: one workgroup per CU
        const int cus = a.max_blocks / AG_MLP_WG_PER_CU > 0 ? a.max_blocks / AG_MLP_WG_PER_CU : 1, nblk = (a.B * a.N + 31) / 32;
        const dim3 g2(nblk < cus ? nblk : cus);
        // (agg_q16 is a mode-2 option and mode 2 writes the next round's sender table as q16 rows: the combination <false, true> does not exist)
        if (a.agg_q16 && a.hs_out_q16) hipLaunchKernelGGL((node_update_nws_kernel<true, true>), g2, dim3(256), 0, s, w, a);
        else if (a.hs_out_q16) hipLaunchKernelGGL((node_update_nws_kernel<true, false>), g2, dim3(256), 0, s, w, a);
        else hipLaunchKernelGGL((node_update_nws_kernel<false, false>), g2, dim3(256), 0, s, w, a);
        return;
    }
    if (a.precision == AG_PREC_B3) {
        if (a.fuse_agg == 2 && a.eterm_half) {      // cooperative LDS-staged reduce inside the kernel (no aggregate launch, no agg table)
            if (last) hipLaunchKernelGGL((node_update_kernel<PrecB3, true, true>), grid, block, 0, s, w, a);
            else if (a.hs_out_q16) hipLaunchKernelGGL((node_update_kernel<PrecB3, false, true, true>), grid, block, 0, s, w, a);
            else hipLaunchKernelGGL((node_update_kernel<PrecB3, false, true>), grid, block, 0, s, w, a);
        } else if (a.agg_q16 && (last || a.hs_out_q16)) {      // (mode 2 only: `agg` arrives as q16 rows; its rounds before the last write Hs as q16 rows)
            if (last) hipLaunchKernelGGL((node_update_kernel<PrecB3, true, false, false, true>), grid, block, 0, s, w, a);
            else hipLaunchKernelGGL((node_update_kernel<PrecB3, false, false, true, true>), grid, block, 0, s, w, a);
        } else if (last) hipLaunchKernelGGL((node_update_kernel<PrecB3, true, false>), grid, block, 0, s, w, a);
        else if (a.hs_out_q16) hipLaunchKernelGGL((node_update_kernel<PrecB3, false, false, true>), grid, block, 0, s, w, a);
        else hipLaunchKernelGGL((node_update_kernel<PrecB3, false, false>), grid, block, 0, s, w, a);
    } else {
        if (last) hipLaunchKernelGGL((node_update_kernel<PrecF32, true, false>), grid, block, 0, s, w, a);
        else hipLaunchKernelGGL((node_update_kernel<PrecF32, false, false>), grid, block, 0, s, w, a);
    }
}

void ag_launch_train_pack(const float *W, const float *bias, int n_out, int n_in, int ld, int col0, int transposed, int compact,
                          int n_tiles, int b3, float *dst, hipStream_t s)
{
    if (b3) {
        if (compact) ag_launch_zero_words(reinterpret_cast<int32_t *>(dst), AG_CHUNK_FLOATS, s);      // the compact image has unused tail bytes when NU = 1
        const int total = (compact ? AG_NT * ((n_in + 1 + 15) / 16) : n_tiles * 10) * 512;
        hipLaunchKernelGGL(train_pack_b3_kernel, dim3((total + 255) / 256), dim3(256), 0, s, W, bias, n_out, n_in, ld, col0, transposed, compact, n_tiles, dst);
        return;
    }
    const int total = (compact ? 1 : n_tiles) * AG_CHUNK_FLOATS;
    hipLaunchKernelGGL(train_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, s, W, bias, n_out, n_in, ld, col0, transposed, compact, n_tiles, dst);
}

void ag_launch_chain(int kind, int backward, int b3, const AgChainArgsPOD &p, int max_blocks, hipStream_t s)
{
    AgChainArgs a;
    a.x = p.x; a.w = reinterpret_cast<const float4 *>(p.w); a.dy = p.dy; a.dx = p.dx; a.rows = p.rows; a.d_in = p.d_in;
    for (int l = 0; l < 4; ++l) { a.y[l] = p.y[l]; a.dz[l] = p.dz[l]; }
    const long long tiles = (p.rows + AG_ROWS_PER_BLOCK - 1) / AG_ROWS_PER_BLOCK;
    const dim3 grid((unsigned)(tiles < max_blocks ? (tiles > 0 ? tiles : 1) : max_blocks)), block(AG_MLP_THREADS);
#define AG_CHAIN_CASE(K, P) \
    if (backward) hipLaunchKernelGGL((chain_backward_kernel<K, P>), grid, block, 0, s, a); else hipLaunchKernelGGL((chain_forward_kernel<K, P>), grid, block, 0, s, a);
    switch (kind * 2 + (b3 ? 1 : 0)) {
    case 0: AG_CHAIN_CASE(0, PrecF32) break;
    case 1: AG_CHAIN_CASE(0, PrecB3) break;
    case 2: AG_CHAIN_CASE(1, PrecF32) break;
    case 3: AG_CHAIN_CASE(1, PrecB3) break;
    case 4: AG_CHAIN_CASE(2, PrecF32) break;
    case 5: AG_CHAIN_CASE(2, PrecB3) break;
    default: break;
    }
#undef AG_CHAIN_CASE
}
