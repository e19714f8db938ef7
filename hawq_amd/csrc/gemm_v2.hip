// Round 5: the 1x1 convolutions with a long K loop (QuantBnConv2d, quant_modules.py:489-494: the reduce convs that open a
// bottleneck and the expand conv + identity conv that close the first unit of a stage, q_resnet.py:231-260) as a streaming GEMM
// built on the machinery of band_v2.hip.
//
// Why.  conv_kernel (conv_igemm.hip) walks K in 64-byte chunks with one workgroup barrier per chunk: for these layers a chunk is
// 64-128 cycles of matrix pipe behind ~1000 cycles of LDS-DMA round trip + barrier, so a K = 2048 reduce conv on a 7 x 7 map spends
// 32 such round trips (15-20 us for 1.3 us of MFMA work), and its operand tiles are 64-byte row segments a row apart - half
// cache lines, the slow LDS-DMA shape (tools/ubench/dma_issue.hip).  A 1x1 conv has no operand reuse beyond its tile, so what bounds
// it is the CU's ingest: (BM + BN) * K bytes per tile at ~40-60 B/clk.  This kernel is organised around exactly that:
//   * K is walked in 128-byte chunks: a pixel row contributes one full 128-byte line per chunk (8 rows = one contiguous-line KiB per
//     LDS-DMA instruction); weights are packed on the host into [N/64][K/128][64 rows][128 B] (hawq_pack_w1x1_k128), pre-swizzled;
//   * a ring of WS stages {X tile [128 px][128 B], W tile [BN][128 B]} filled by 4 producer waves that run WS - 1 chunks ahead,
//     consumed by 4 MFMA waves (2 x 2: 64 px x BN/2 channels each); NO workgroup barrier in the K loop - landed[] / done[] counters
//     in LDS as in band_v2.hip (flag after data on the writer's side, data after flag on the reader's);
//   * LDS rows are 128 bytes; slot s (16 B) of row r is stored at s ^ ((r >> 1) & 7): the 16 lanes of a ds_read_b128 group read 16
//     rows whose (r & 1, (r >> 1) & 7) are all different - conflict-free for both operands;
//   * the identity conv of a resize unit (q_resnet.py:236: 1x1, stride 2, its own K) is a second PHASE of the same ring into a
//     second accumulator set - no second launch, no second pass over the output tile;
//   * epilogues straight from registers (a lane holds 16 consecutive channels of one pixel): REQUANT (int8, NHWC rows or planes),
//     RESIDUAL with a stored uint16 residual or with the identity conv's accumulators (quant_utils.py:415-456: two requants, the
//     un-clamped sum, ReLU, uint16 out with the sticky overflow flag, the next QuantAct's int8).
// int8 operands, fast-contract tables, K (and K2) multiples of 128, Cout a multiple of BN.
// Round 6: (a) hawq4 operands (NIB; both operands 4-bit, Cin % 256 == 0: a 128-byte chunk is 256 channels, every 16-byte fragment feeds
// two MFMA K-steps after the nibble unpack of band_v2.hip - weights as value * 16, accumulators shifted back by 4 once, exact) and hawq4
// outputs, so that the reduce convs of the 4-bit schedules (bit_config.py:806, 1512) stream half the bytes instead of running the generic
// register-staged kernel; (b) a RAW epilogue (int32 accumulators + bias, dense [M][Cout]) so that the parity tests pin these kernels'
// accumulators directly (north_star: "bit-exact on the int32 pre-requant accumulators").
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace {

struct G2P {
    const char *x, *w;       // [N][H1][W1][K] int8 rows (NHWC) read at stride s1 ; hawq_pack_w1x1_k128 stream
    const char *x2, *w2;     // second phase: block input [N][H2][W2][K2] read at stride s2 ; packed identity weights
    const int32_t *ctab, *ctab_id;
    const int32_t *bias;     // RAW only
    int32_t *out_acc;        // RAW only: [M][Cout] int32
    char *out;               // int8: NHWC [M][Cout] or planes [Cout/16][M][16 B]; hawq4: NHWC [M][Cout/2] or planes [Cout/32][M][16 B]; may be null (RESIDUAL)
    const char *res_in;      // [M][Cout] uint16 (single-branch RESIDUAL)
    char *res_out;           // [M][Cout] uint16 or null
    int *flags;
    int M, K, Cout, n1, n2;  // n1 / n2: 128-byte chunks of phase 1 / 2
    int Ho, Wo, H1, W1, s1, H2, W2, K2, s2;
    int out_planar, out_bits, q_lo, q_hi;
    int mq, eq, m_id, e_id;
    unsigned x_bytes, w_bytes, x2_bytes, w2_bytes;
    int dbg;
    long long *dbgbuf;
};

template <int CT_, int WS_, int MINW_>
struct G2Cfg {
    static constexpr int CT = CT_, WS = WS_, MINW = MINW_;   // CT: 32-channel MFMA tiles per wave
    static constexpr int BM = 128, BN = 64 * CT, NW = 4, NPROD = 4, NT = (NW + NPROD) * 64;
    static constexpr int XT = BM * 128, WT = BN * 128, STAGE = XT + WT;
    static constexpr int RING_END = WS * STAGE;
    static constexpr int OFF_CTAB = RING_END, OFF_SYNC = RING_END + 2 * BN * 16;   // ctab, ctab_id, then landed[4] / done[8]
    static constexpr int LDS_BYTES = OFF_SYNC + 64;
    static constexpr int XP = BM / 8 / NPROD, WP = BN / 8 / NPROD, P = XP + WP;     // 1-KiB pieces per producer wave per stage
    static constexpr int NR = CT + 2;                                                // fragment reads per k-step
    static constexpr int PD = CT == 1 ? 2 : 1, NBUF = 2 * PD;                        // k-steps the fragment requests run ahead; buffers (k-step & (NBUF - 1))
    static_assert(WS >= 3 && WS <= 6 && LDS_BYTES <= 160 * 1024, "ring");
};

// `buffer_load ... lds` (rows beyond M are out of range: the hardware writes zeros).  In a micro-benchmark `global_load_lds` ingests more
// per issuing wave (tools/ubench/dma_l2.hip: 53 against 33 B/clk/CU with 4 waves); inside these kernels it measured 10-20 % SLOWER
// (profiles/r05_band_v2.md) - the flat form's per-lane 64-bit addresses and zero-page selects sit in the producers' issue loop.
__device__ __forceinline__ void g2_dma(__amdgpu_buffer_rsrc_t r, char *lds, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)lds, 16, voff, soff, 0, 0);
}
template <int N>
__device__ __forceinline__ void g2_wait_vm_upto(int n) {
    if constexpr (N == 0) {
        wait_vmcnt<0>();
    } else {
        if (n >= N) wait_vmcnt<N>(); else g2_wait_vm_upto<N - 1>(n);
    }
}
__device__ __forceinline__ int g2_min4(const v4i &v) { return min(min(v.x, v.y), min(v.z, v.w)); }
__device__ __forceinline__ int g2_lds_min4_now(unsigned addr) {
    v4i v;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return __builtin_amdgcn_readfirstlane(g2_min4(v));
}
__device__ __forceinline__ void g2_lds_store_b32(unsigned addr, int v) { asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
// byte offset of 16-byte slot s of row r inside a [rows][128 B] tile
__device__ __forceinline__ unsigned g2_off(int r, int s) { return (unsigned)(r * 128 + ((s ^ ((r >> 1) & 7)) << 4)); }

// hawq4 -> int8 operand dwords (see band_v2.hip b2_unpack16): two packed dwords = 16 channels = one MFMA K-step of this lane
template <bool WEIGHT>
__device__ __forceinline__ v4i g2_unpack16(unsigned x0, unsigned x1) {
    v4i r;
    if (WEIGHT) {
        r.x = (int)((x0 << 4) & 0xF0F0F0F0u), r.y = (int)(x0 & 0xF0F0F0F0u), r.z = (int)((x1 << 4) & 0xF0F0F0F0u), r.w = (int)(x1 & 0xF0F0F0F0u);
    } else {
        r.x = (int)(x0 & 0x0F0F0F0Fu), r.y = (int)((x0 >> 4) & 0x0F0F0F0Fu), r.z = (int)(x1 & 0x0F0F0F0Fu), r.w = (int)((x1 >> 4) & 0x0F0F0F0Fu);
    }
    return r;
}

// EPI: HAWQ_EPI_REQUANT / HAWQ_EPI_RESIDUAL / HAWQ_EPI_RAW.  DUAL: the residual is the identity conv (second phase).  MODE: 0 tie-free,
// 2 exact ties.  NIB: hawq4 operands (both phases).
template <class C, int EPI, bool DUAL, int MODE, bool NIB = false>
__global__ __launch_bounds__(C::NT, C::MINW) void gemm1x1_v2_kernel(const G2P p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const bool prof = HAWQ_DBG_BIT(p.dbg, 128) && p.dbgbuf;
    const long long t_entry = prof ? (long long)__builtin_readcyclecounter() : 0;
    const int tiles_c = p.Cout / C::BN;
    const int nwg = ((p.M + C::BM - 1) / C::BM) * tiles_c;
    int wg = blockIdx.x;
    {   // each XCD (id mod 8) owns a contiguous run of tiles: the channel tiles of a pixel tile share its rows in that L2
        const int q = nwg >> 3, r = nwg & 7, xcd = wg & 7, idx = wg >> 3;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tc = wg % tiles_c, tm = wg / tiles_c;
    const int m0 = tm * C::BM, c0 = tc * C::BN;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int total = p.n1 + (DUAL ? p.n2 : 0);
    char *const ctab_lds = smem + C::OFF_CTAB;
    const unsigned sync_a = lds_addr(smem + C::OFF_SYNC);   // landed[4] at +0, done[8] at +16

    if (wave >= C::NW) {
        // ------------------------------------------------------------------ producer waves
        const int dw = wave - C::NW;
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)p.x, 0, (int)p.x_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void *)p.w, 0, (int)p.w_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rx2 = __builtin_amdgcn_make_buffer_rsrc((void *)(DUAL ? p.x2 : p.x), 0, (int)(DUAL ? p.x2_bytes : p.x_bytes), 0x00020000);
        const __amdgpu_buffer_rsrc_t rw2 = __builtin_amdgcn_make_buffer_rsrc((void *)(DUAL ? p.w2 : p.w), 0, (int)(DUAL ? p.w2_bytes : p.w_bytes), 0x00020000);
        // X piece j of this wave = tile rows (dw + 4 j) * 8 .. + 7; lane -> (row, stored slot); it fetches the logical slot that belongs there
        unsigned xo1[C::XP], xo2[C::XP];   // byte offset of this lane's 16 bytes inside the tensor (chunk 0), or out of range: row beyond M
#pragma unroll
        for (int j = 0; j < C::XP; ++j) {
            const int r = (dw + C::NPROD * j) * 8 + (lane >> 3), m = m0 + r;
            const unsigned so = (unsigned)(((lane & 7) ^ ((r >> 1) & 7)) << 4);
            const int x = m % p.Wo, gy = m / p.Wo, y = gy % p.Ho, n = gy / p.Ho;
            xo1[j] = m < p.M ? (unsigned)((n * p.H1 + y * p.s1) * p.W1 + x * p.s1) * (unsigned)p.K + so : 0x80000000u;
            xo2[j] = DUAL && m < p.M ? (unsigned)((n * p.H2 + y * p.s2) * p.W2 + x * p.s2) * (unsigned)p.K2 + so : 0x80000000u;
        }
        const unsigned wvo = (unsigned)(lane * 16);
        auto issue_stage = [&](int s) {   // stage s -> ring slot s % WS
            char *xd = smem + (s % C::WS) * C::STAGE, *wd = xd + C::XT;
            const bool ph2 = DUAL && s >= p.n1;
            const int ch = ph2 ? s - p.n1 : s;
            const int nch = ph2 ? p.n2 : p.n1;
#pragma unroll
            for (int j = 0; j < C::XP; ++j)
                g2_dma(ph2 ? rx2 : rx, xd + (dw + C::NPROD * j) * 1024, ph2 ? xo2[j] : xo1[j], (unsigned)(ch * 128));
#pragma unroll
            for (int j = 0; j < C::WP; ++j) {
                const int pc = dw + C::NPROD * j;   // piece of the W tile: 64-row group pc >> 3, rows (pc & 7) * 8 .. + 7
                const unsigned so = ((unsigned)((tc * C::CT + (pc >> 3)) * nch + ch) * 8192u) + (unsigned)((pc & 7) * 1024);
                g2_dma(ph2 ? rw2 : rw, wd + pc * 1024, wvo, so);
            }
        };
        auto publish = [&](int n) { g2_lds_store_b32(sync_a + (unsigned)(dw * 4), n); };
        int done_seen = 0;   // the last min(done[]) seen: a poll is only paid when it cannot already answer the question (see band_v2.hip)
        auto wait_done = [&](int n) {
            while (done_seen < n) {
                done_seen = g2_lds_min4_now(sync_a + 16);
                if (done_seen < n) __builtin_amdgcn_s_sleep(1);
            }
        };
        // requant constants first (oldest: landed with stage 0), then the first stages - before anything else happens in the workgroup
        if (EPI != HAWQ_EPI_RAW && dw < C::CT) {
            const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void *)p.ctab, 0, p.Cout * 16, 0x00020000);
            g2_dma(rc, ctab_lds + dw * 1024, wvo, (unsigned)((c0 + dw * 64) * 16));
        } else if (DUAL && dw >= 2 && dw - 2 < C::CT) {
            const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void *)p.ctab_id, 0, p.Cout * 16, 0x00020000);
            g2_dma(rc, ctab_lds + C::BN * 16 + (dw - 2) * 1024, wvo, (unsigned)((c0 + (dw - 2) * 64) * 16));
        }
        int issued = total < C::WS - 1 ? total : C::WS - 1;
        for (int s = 0; s < issued; ++s) issue_stage(s);
        __builtin_amdgcn_s_barrier();   // the synchronisation words are initialised (MFMA wave 0); bare: no vmcnt(0) in front of it
        g2_wait_vm_upto<(C::WS - 2) * C::P>((issued - 1) * C::P);   // stage 0 (and the constants)
        publish(1);
        int hw1 = issued >= 2 ? C::P : 0;   // pieces of the newest stage, if it may still be in flight
        for (;;) {
            g2_wait_vm_upto<C::P>(hw1);               // everything but the newest stage
            publish(hw1 ? issued - 1 : issued);
            if (issued < total && !HAWQ_DBG_BIT(p.dbg, 1)) {
                wait_done(issued - C::WS + 1);        // ring slot issued % WS: every MFMA wave is done with stage issued - WS
                issue_stage(issued);
                ++issued, hw1 = C::P;
            } else if (hw1) {
                hw1 = 0;
            } else {
                break;
            }
        }
        publish(total + 2);   // everything has landed; this wave touches the synchronisation words no more
        return;
    }

    // ---------------------------------------------------------------------- MFMA waves: 64 px x 32 CT channels
    if (t < 16) {
        const bool used = t < 4 ? true : (t < 12 ? t - 4 < C::NW : false);
        *reinterpret_cast<int *>(smem + C::OFF_SYNC + t * 4) = t >= 12 ? 0 : (used ? 0 : 0x7fffffff);
    }
    const int wm = wave & 1, wn = wave >> 1;
    const int l31 = lane & 31, h = lane >> 5;
    unsigned xo[2][4], wo[C::CT][4];   // fragment offsets inside a stage: [pixel tile / channel tile][k-step of the chunk]
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
        for (int q = 0; q < 2; ++q) xo[q][ks] = g2_off(wm * 64 + q * 32 + l31, 2 * ks + h);
#pragma unroll
        for (int c = 0; c < C::CT; ++c) wo[c][ks] = (unsigned)C::XT + g2_off(wn * (32 * C::CT) + c * 32 + cperm(l31), 2 * ks + h);
    }
    v16i acc[C::CT][2], acc2[DUAL ? C::CT : 1][2];
#pragma unroll
    for (int c = 0; c < C::CT; ++c)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[c][q][r] = 0;
                if (DUAL) acc2[c][q][r] = 0;
            }
    // fragment buffers by k-step parity (a chunk has four k-steps: no parity drift across chunks); the fragments of k-step ks + PD are
    // requested before the MFMAs of k-step ks (PD = 2 for the 64-channel tile, whose k-step is only two MFMAs long)
    v4i wf[C::NBUF][C::CT], af[C::NBUF][2];
#define G2_FETCH(KS, ST)                                                                              \
    if (!HAWQ_DBG_BIT(p.dbg, 4)) {                                                                    \
        _Pragma("unroll") for (int c = 0; c < C::CT; ++c) wf[(KS) % C::NBUF][c] = lds_read16<0>((ST) + wo[c][KS]); \
        _Pragma("unroll") for (int q = 0; q < 2; ++q) af[(KS) % C::NBUF][q] = lds_read16<0>((ST) + xo[q][KS]);    \
    }
#define G2_MMA(KS, ACC)                                                                               \
    {                                                                                                 \
        _Pragma("unroll") for (int c = 0; c < C::CT; ++c) pin(wf[(KS) % C::NBUF][c]);                  \
        _Pragma("unroll") for (int q = 0; q < 2; ++q) pin(af[(KS) % C::NBUF][q]);                      \
        if (!HAWQ_DBG_BIT(p.dbg, 2)) {                                                                \
            if constexpr (NIB) {                                                                      \
                _Pragma("unroll") for (int hf = 0; hf < 2; ++hf) {                                    \
                    v4i w8[C::CT], a8[2];                                                             \
                    _Pragma("unroll") for (int c = 0; c < C::CT; ++c) w8[c] = g2_unpack16<true>((unsigned)wf[(KS) % C::NBUF][c][2 * hf], (unsigned)wf[(KS) % C::NBUF][c][2 * hf + 1]); \
                    _Pragma("unroll") for (int q = 0; q < 2; ++q) a8[q] = g2_unpack16<false>((unsigned)af[(KS) % C::NBUF][q][2 * hf], (unsigned)af[(KS) % C::NBUF][q][2 * hf + 1]); \
                    _Pragma("unroll") for (int q = 0; q < 2; ++q)                                     \
                        _Pragma("unroll") for (int c = 0; c < C::CT; ++c)                             \
                            ACC[c][q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(w8[c], a8[q], ACC[c][q], 0, 0, 0); \
                }                                                                                     \
            } else {                                                                                  \
                _Pragma("unroll") for (int q = 0; q < 2; ++q)                                         \
                    _Pragma("unroll") for (int c = 0; c < C::CT; ++c)                                 \
                        ACC[c][q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[(KS) % C::NBUF][c], af[(KS) % C::NBUF][q], ACC[c][q], 0, 0, 0); \
            }                                                                                         \
        }                                                                                             \
    }
    const long long t_begin = prof ? (long long)__builtin_readcyclecounter() : 0;
    __syncthreads();   // the synchronisation words are initialised
    const unsigned done_a = sync_a + 16u + (unsigned)(wave * 4);
    const unsigned ring_a = lds_addr(smem);
    v4i pl;
    auto ensure = [&](int n) {
        pin(pl);
        if (__builtin_amdgcn_readfirstlane(g2_min4(pl)) >= n) return;
        while (g2_lds_min4_now(sync_a) < n) __builtin_amdgcn_s_sleep(1);
    };
    while (g2_lds_min4_now(sync_a) < 1) __builtin_amdgcn_s_sleep(1);   // stage 0
    const long long t_b0 = prof ? (long long)__builtin_readcyclecounter() : 0;
    __builtin_amdgcn_s_setprio(2);
    int s = 0, slot = 0;
    unsigned st = ring_a;
    G2_FETCH(0, st)
    if (C::PD == 2) { G2_FETCH(1, st) }
    auto run = [&](auto &ACC, int s_end) {   // chunks s .. s_end - 1 into ACC (the fragments of its first PD k-steps are on their way)
        for (; s < s_end; ++s) {
            asm volatile("ds_read_b128 %0, %1" : "=v"(pl) : "v"(sync_a) : "memory");
            if (++slot == C::WS) slot = 0;
            const unsigned nst = ring_a + (unsigned)(slot * C::STAGE);
            const bool more = s + 1 < total;
            if constexpr (C::PD == 2) {
                G2_FETCH(2, st) wait_lgkm<2 * C::NR>(); G2_MMA(0, ACC)
                G2_FETCH(3, st) wait_lgkm<2 * C::NR>(); G2_MMA(1, ACC)
                if (more) {
                    ensure(s + 2);
                    G2_FETCH(0, nst) wait_lgkm<2 * C::NR>(); G2_MMA(2, ACC)
                    G2_FETCH(1, nst) wait_lgkm<2 * C::NR>();
                } else {
                    wait_lgkm<0>();
                    G2_MMA(2, ACC)
                }
            } else {
                G2_FETCH(1, st) wait_lgkm<C::NR>(); G2_MMA(0, ACC)
                G2_FETCH(2, st) wait_lgkm<C::NR>(); G2_MMA(1, ACC)
                G2_FETCH(3, st) wait_lgkm<C::NR>(); G2_MMA(2, ACC)
                if (more) {
                    ensure(s + 2);
                    G2_FETCH(0, nst) wait_lgkm<C::NR>();
                } else {
                    wait_lgkm<0>();
                }
            }
            g2_lds_store_b32(done_a, s + 1);   // every fragment read of chunk s has returned
            G2_MMA(3, ACC)
            st = nst;
        }
    };
    run(acc, p.n1);
    if constexpr (DUAL) run(acc2, total);
#undef G2_FETCH
#undef G2_MMA
    __builtin_amdgcn_s_setprio(0);
    const long long t_loop_end = prof ? (long long)__builtin_readcyclecounter() : 0;

    if constexpr (NIB) {   // the weights were value * 16: every sum is an exact multiple of 16
#pragma unroll
        for (int c = 0; c < C::CT; ++c)
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    acc[c][q][r] >>= 4;
                    if (DUAL) acc2[c][q][r] >>= 4;
                }
    }
    // ---------------------------------------------------------------------- epilogue, straight from registers
    if constexpr (EPI == HAWQ_EPI_RAW) {   // int32 accumulators + bias, dense [M][Cout]: a lane's 16 channels are 64 contiguous bytes
#pragma unroll
        for (int c = 0; c < C::CT; ++c) {
            const int ch = c0 + wn * (32 * C::CT) + c * 32 + h * 16;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int m = m0 + wm * 64 + q * 32 + l31;
                if (m >= p.M) continue;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const v4i b4 = *reinterpret_cast<const v4i *>(p.bias + ch + 4 * gq);
                    const v4i v = {acc[c][q][4 * gq] + b4.x, acc[c][q][4 * gq + 1] + b4.y, acc[c][q][4 * gq + 2] + b4.z, acc[c][q][4 * gq + 3] + b4.w};
                    reinterpret_cast<v4i *>(p.out_acc + (size_t)m * p.Cout + ch)[gq] = v;
                }
            }
        }
        return;
    }
    v4i rin[2][C::CT][2];
    if constexpr (EPI == HAWQ_EPI_RESIDUAL && !DUAL) {   // the stored residual: 32 contiguous bytes per (pixel, 16 channels), requested now
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            int m = m0 + wm * 64 + q * 32 + l31;
            m = m < p.M ? m : p.M - 1;
#pragma unroll
            for (int c = 0; c < C::CT; ++c) {
                const v4i *rp = reinterpret_cast<const v4i *>(p.res_in + ((size_t)m * p.Cout + c0 + wn * (32 * C::CT) + c * 32 + h * 16) * 2);
                rin[q][c][0] = rp[0], rin[q][c][1] = rp[1];
            }
        }
    }
    while (g2_lds_min4_now(sync_a) < total + 2) __builtin_amdgcn_s_sleep(1);   // the producers are through: the constants have landed
    DyNt dids = dynt_prepare(p.m_id, p.e_id), dq = dynt_prepare(p.mq, p.eq);
    asm volatile("" : "+v"(dids.add), "+v"(dq.add));
    const int qhi2 = (p.q_hi & 0xffff) | (p.q_hi << 16);
    unsigned oor = 0;
#pragma unroll
    for (int c = 0; c < C::CT; ++c) {
        const int lch = wn * (32 * C::CT) + c * 32 + h * 16;   // tile-local first channel of this lane
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            // one (channel tile, pixel tile) at a time, its constants re-read from LDS: the dual form holds 2 x 64 accumulators and has no
            // registers for both pixel tiles' packed outputs (it spilled 50-150 of them with the loops the other way round)
            __builtin_amdgcn_sched_barrier(0);
            int w[4], rp[8];
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                DyNt d[4], di[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const v4i e = *reinterpret_cast<const v4i *>(ctab_lds + (lch + 4 * gq + j) * 16);
                    d[j].m = e.x, d[j].s = e.y & 31, d[j].k = e.y >> 8;
                    d[j].add = (long long)(((unsigned long long)(unsigned)e.w << 32) | (unsigned)e.z);
                    if constexpr (DUAL) {
                        const v4i u = *reinterpret_cast<const v4i *>(ctab_lds + C::BN * 16 + (lch + 4 * gq + j) * 16);
                        di[j].m = u.x, di[j].s = u.y & 31, di[j].k = u.y >> 8;
                        di[j].add = (long long)(((unsigned long long)(unsigned)u.w << 32) | (unsigned)u.z);
                    } else {
                        di[j] = dids;
                    }
                }
                int qv[4];
                if constexpr (EPI == HAWQ_EPI_REQUANT) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) qv[j] = med3i(dyadic_mode<MODE>(acc[c][q][4 * gq + j], d[j]), p.q_lo, p.q_hi);
                    w[gq] = pack4_fast(qv[0], qv[1], qv[2], qv[3]);
                } else {
                    int idin[4], o[4];
                    if constexpr (DUAL) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) idin[j] = acc2[DUAL ? c : 0][q][4 * gq + j];
                    } else {
                        const unsigned w0 = (unsigned)rin[q][c][gq >> 1][(gq & 1) * 2], w1 = (unsigned)rin[q][c][gq >> 1][(gq & 1) * 2 + 1];
                        idin[0] = (int)(w0 & 0xffffu), idin[1] = (int)(w0 >> 16), idin[2] = (int)(w1 & 0xffffu), idin[3] = (int)(w1 >> 16);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int a = dyadic_mode<MODE>(acc[c][q][4 * gq + j], d[j]);
                        const int b = DUAL ? dyadic_mode<MODE>(idin[j], di[j]) : dyadic_mode<MODE == 2 ? 2 : 0>(idin[j], di[j]);
                        o[j] = max(a + b, 0);                 // no clamp: quant_utils.py:456
                        qv[j] = dyadic_mode<MODE>(o[j], dq);  // o >= 0, m >= 0: q >= 0; clamped from above in the pack
                    }
                    if (m0 + wm * 64 + q * 32 + l31 < p.M) oor |= (unsigned)(o[0] | o[1]) | (unsigned)(o[2] | o[3]);
                    rp[2 * gq] = pack2_u16_sat(o[0], o[1]);
                    rp[2 * gq + 1] = pack2_u16_sat(o[2], o[3]);
                    w[gq] = pack4_min(qv[0], qv[1], qv[2], qv[3], qhi2);
                }
            }
            const int m = m0 + wm * 64 + q * 32 + l31;
            if (m < p.M && !HAWQ_DBG_BIT(p.dbg, 8)) {
                if (EPI == HAWQ_EPI_RESIDUAL && p.res_out) {
                    v4i *dst = reinterpret_cast<v4i *>(p.res_out + ((size_t)m * p.Cout + c0 + lch) * 2);
                    const v4i ra = {rp[0], rp[1], rp[2], rp[3]}, rb = {rp[4], rp[5], rp[6], rp[7]};
                    dst[0] = ra, dst[1] = rb;
                }
                if (p.out && p.out_bits == 8) {
                    const v4i ww = {w[0], w[1], w[2], w[3]};
                    char *dst = p.out_planar ? p.out + ((size_t)((c0 + lch) >> 4) * p.M + m) * 16 : p.out + (size_t)m * p.Cout + c0 + lch;
                    *reinterpret_cast<v4i *>(dst) = ww;
                } else if (p.out) {   // hawq4: bytes hold values 0 .. 15, channels 4k .. 4k+3 in w[k]; low nibbles = channels 0-3 of an 8-group, high = 4-7
                    const int ch = c0 + lch;
                    const v2i ww = {w[0] | (w[1] << 4), w[2] | (w[3] << 4)};
                    char *dst = p.out_planar ? p.out + ((size_t)(ch >> 5) * p.M + m) * 16 + ((ch >> 4) & 1) * 8 : p.out + (((size_t)m * p.Cout + ch) >> 1);
                    *reinterpret_cast<v2i *>(dst) = ww;
                }
            }
        }
    }
    if (EPI == HAWQ_EPI_RESIDUAL && (oor >> 16) != 0 && p.res_out && !HAWQ_DBG_BIT(p.dbg, ~0)) atomicOr(p.flags, 1);
    if (prof && blockIdx.x == 8 && t == 0) {
        p.dbgbuf[0] = t_begin - t_entry, p.dbgbuf[1] = t_loop_end - t_begin;
        p.dbgbuf[2] = (long long)__builtin_readcyclecounter() - t_loop_end, p.dbgbuf[3] = total;
        p.dbgbuf[4] = t_b0 - t_begin;
    }
}

// <CT, WS, MINW>
using G64D = G2Cfg<1, 5, 2>;    // 128 px x 64 ch, 5-stage ring (122 KiB): long K, few workgroups
using G64S = G2Cfg<1, 3, 2>;    // 128 px x 64 ch, 3-stage ring (74 KiB): two workgroups per CU - many workgroups, short K
using G128 = G2Cfg<2, 4, 2>;    // 128 px x 128 ch, 4-stage ring (132 KiB): wide outputs (the expand + identity launches)
constexpr int NUM_G2 = 3;

typedef void (*G2Fn)(const G2P);
struct G2Info { G2Fn fn[2][3][2]; G2Fn raw[2]; int bn, lds, nt; };   // fn[hawq4][REQUANT | RESIDUAL | RESIDUAL + identity conv][exact-tie]; raw[hawq4]
#define G2_FNS(CFG, N) {{gemm1x1_v2_kernel<CFG, HAWQ_EPI_REQUANT, false, 0, N>, gemm1x1_v2_kernel<CFG, HAWQ_EPI_REQUANT, false, 2, N>},    \
                        {gemm1x1_v2_kernel<CFG, HAWQ_EPI_RESIDUAL, false, 0, N>, gemm1x1_v2_kernel<CFG, HAWQ_EPI_RESIDUAL, false, 2, N>},  \
                        {gemm1x1_v2_kernel<CFG, HAWQ_EPI_RESIDUAL, true, 0, N>, gemm1x1_v2_kernel<CFG, HAWQ_EPI_RESIDUAL, true, 2, N>}}
#define G2_ENTRY(CFG) {{G2_FNS(CFG, false), G2_FNS(CFG, true)}, {gemm1x1_v2_kernel<CFG, HAWQ_EPI_RAW, false, 0, false>, gemm1x1_v2_kernel<CFG, HAWQ_EPI_RAW, false, 0, true>}, \
                       CFG::BN, CFG::LDS_BYTES, CFG::NT}
const G2Info kG2[NUM_G2] = {G2_ENTRY(G64D), G2_ENTRY(G64S), G2_ENTRY(G128)};

}  // namespace

int gemm_v2_count(void) { return NUM_G2; }

// [N][K] int8 -> [N/64][K/128][64 rows][128 B], 16-byte slot s of row r at r * 128 + ((s ^ ((r >> 1) & 7)) << 4)
extern "C" int hawq_pack_w1x1_k128(const int8_t *src, int8_t *dst, int32_t N, int32_t K) {
    HAWQ_REQUIRE(src && dst && N > 0 && K > 0 && N % 64 == 0 && K % 128 == 0, "hawq_pack_w1x1_k128: N must be a positive multiple of 64, K of 128");
    const int nch = K >> 7;
    for (int g = 0; g < (N >> 6); ++g)
        for (int ch = 0; ch < nch; ++ch) {
            int8_t *tile = dst + ((size_t)g * nch + ch) * 8192;
            for (int r = 0; r < 64; ++r) {
                const int8_t *row = src + (size_t)(g * 64 + r) * K + ch * 128;
                for (int sl = 0; sl < 8; ++sl)
                    for (int b = 0; b < 16; ++b) tile[r * 128 + ((sl ^ ((r >> 1) & 7)) << 4) + b] = row[sl * 16 + b];
            }
        }
    return 0;
}

bool gemm_v2_applies(const hawq_conv_args *a, int v) {
    if (v < 0 || v >= NUM_G2) return false;
    const G2Info &gi = kG2[v];
    const bool dual = a->in2 != nullptr;
    const bool nib = a->in_bits == 4 && a->w_bits == 4;
    const int rowb = nib ? a->Cin >> 1 : a->Cin, rowb2 = nib ? a->Cin2 >> 1 : a->Cin2;   // bytes per pixel / weight row
    const long long Ho = (a->H - 1) / a->stride + 1, Wo = (a->W - 1) / a->stride + 1, M = (long long)a->N * Ho * Wo, Min = (long long)a->N * a->H * a->W;
    const bool qout_ok = a->out_bits == 8 || (a->out_bits == 4 && a->q_lo >= 0 && a->q_hi <= 15 && a->Cout % 32 == 0);
    const bool raw = a->epilogue == HAWQ_EPI_RAW;
    const bool epi_ok = (raw && a->out_acc && a->bias && !dual) ||
                        (a->epilogue == HAWQ_EPI_REQUANT && a->out_q && qout_ok && (a->out_bits == 8 || a->relu || a->q_lo >= 0) && !dual) ||
                        (a->epilogue == HAWQ_EPI_RESIDUAL && (dual || (a->res_in && a->res_in_bits == 16)) && (!a->res_out || (a->res_out_bits == 16 && a->flags)) &&
                         !a->res_no_relu && !a->res_clamp16 && (a->res_out || a->out_q) && (!a->out_q || qout_ok));
    const bool dual_ok = !dual || (a->wgt2_k128 && a->ctab_id && a->in2_bits == a->in_bits && a->w2_bits == a->w_bits && rowb2 % 128 == 0 &&
                                   (long long)a->N * a->H2 * a->W2 * rowb2 < (1ll << 31) && (long long)a->Cout * rowb2 < (1ll << 31));
    return a->KH == 1 && a->KW == 1 && a->stride >= 1 && a->pad == 0 && (raw || (a->fast_tables != 0 && a->ctab)) && a->wgt_k128 != nullptr && !a->in_planar && epi_ok && dual_ok &&
           ((a->in_bits == 8 && a->w_bits == 8) || nib) && rowb % 128 == 0 && a->Cout % gi.bn == 0 && (a->in_pitch == 0 || a->in_pitch == rowb) &&
           (a->out_pitch == 0 || a->out_pitch == a->Cout) && Min * rowb < (1ll << 31) && (long long)a->Cout * rowb < (1ll << 31) && M * a->Cout < (1ll << 31);
}

int gemm_v2_launch(const hawq_conv_args *a, int v, int exact_tie, int dbg, void *stream) {
    const G2Info &gi = kG2[v];
    const bool dual = a->in2 != nullptr;
    const bool nib = a->in_bits == 4;
    const int rowb = nib ? a->Cin >> 1 : a->Cin, rowb2 = nib ? a->Cin2 >> 1 : a->Cin2;
    G2P p;
    p.x = (const char *)a->in, p.w = (const char *)a->wgt_k128, p.x2 = (const char *)a->in2, p.w2 = (const char *)a->wgt2_k128;
    p.ctab = a->ctab, p.ctab_id = a->ctab_id;
    p.bias = a->bias, p.out_acc = a->out_acc;
    p.out = (char *)a->out_q, p.res_in = (const char *)a->res_in, p.res_out = (char *)a->res_out, p.flags = a->flags;
    p.Ho = (a->H - 1) / a->stride + 1, p.Wo = (a->W - 1) / a->stride + 1, p.M = a->N * p.Ho * p.Wo, p.K = rowb, p.Cout = a->Cout;
    p.H1 = a->H, p.W1 = a->W, p.s1 = a->stride;
    p.n1 = rowb >> 7, p.n2 = dual ? rowb2 >> 7 : 0;
    p.H2 = a->H2, p.W2 = a->W2, p.K2 = rowb2, p.s2 = a->stride2;
    p.out_planar = a->out_planar, p.out_bits = a->out_bits;
    p.q_lo = a->relu && a->q_lo < 0 ? 0 : a->q_lo, p.q_hi = a->q_hi;
    p.mq = a->mq, p.eq = a->eq, p.m_id = a->m_id_scalar, p.e_id = a->e_id_scalar;
    if (a->epilogue != HAWQ_EPI_RESIDUAL || !a->out_q) p.mq = 0, p.eq = 33;
    if (a->epilogue != HAWQ_EPI_RESIDUAL || dual) p.m_id = 0, p.e_id = 33;
    p.x_bytes = (unsigned)((long long)a->N * a->H * a->W * rowb), p.w_bytes = (unsigned)((long long)a->Cout * rowb);
    p.x2_bytes = dual ? (unsigned)((long long)a->N * a->H2 * a->W2 * rowb2) : 0, p.w2_bytes = dual ? (unsigned)((long long)a->Cout * rowb2) : 0;
    p.dbg = dbg;
    static long long *dbg_dev = nullptr;
    if (HAWQ_DBG_BIT(dbg, 128) && !dbg_dev) (void)hipMalloc(&dbg_dev, 8 * sizeof(long long));
    p.dbgbuf = HAWQ_DBG_BIT(dbg, 128) ? dbg_dev : nullptr;
    static const bool attrs = [] {
        bool good = true;
        for (const G2Info &i : kG2) {
            for (int k = 0; k < 12; ++k) good &= hipFuncSetAttribute((const void *)i.fn[k / 6][(k % 6) >> 1][k & 1], hipFuncAttributeMaxDynamicSharedMemorySize, i.lds) == hipSuccess;
            for (int k = 0; k < 2; ++k) good &= hipFuncSetAttribute((const void *)i.raw[k], hipFuncAttributeMaxDynamicSharedMemorySize, i.lds) == hipSuccess;
        }
        return good;
    }();
    HAWQ_REQUIRE(attrs, "hawq_conv2d: hipFuncSetAttribute failed for the round-5 1x1 kernels");
    const int grid = ((p.M + 127) / 128) * (p.Cout / gi.bn);
    const int e = a->epilogue == HAWQ_EPI_REQUANT ? 0 : (dual ? 2 : 1);
    G2Fn fn = a->epilogue == HAWQ_EPI_RAW ? gi.raw[nib ? 1 : 0] : gi.fn[nib ? 1 : 0][e][exact_tie ? 1 : 0];
    hipLaunchKernelGGL(fn, dim3(grid), dim3(gi.nt), gi.lds, (hipStream_t)stream, p);
    HAWQ_CHECK_HIP(hipGetLastError());
    if (p.dbgbuf) {   // experiment hook (synchronises!)
        long long hb[5];
        (void)hipStreamSynchronize((hipStream_t)stream);
        (void)hipMemcpy(hb, p.dbgbuf, sizeof(hb), hipMemcpyDeviceToHost);
        fprintf(stderr, "[gemm-v2 %d bn=%d M=%d K=%d(+%d) Cout=%d grid=%d lds=%d] chunks %lld: prologue %lld | K loop %lld | epilogue %lld cycles (wave 0 of workgroup 8); first operands %lld\n",
                v, gi.bn, p.M, a->Cin, dual ? a->Cin2 : 0, p.Cout, grid, gi.lds, hb[3], hb[0], hb[1], hb[2], hb[4]);
    }
    return 0;
}
