// Round 5: 3x3 / stride 1 / pad 1 convolution (QuantBnConv2d + ReLU + QuantAct, quant_modules.py:489-494, 527-545, 233-260; the
// second conv of every ResNet50 bottleneck, q_resnet.py:241-243, and both convs of a ResNet18/34 basic block, :300-306) rebuilt
// around what this round measured on gfx950 (profiles/r05_band_v2.md, tools/ubench/dma_issue.hip, dma_exec.hip):
//   * an LDS-DMA instruction costs the CU's vector-memory path ~2 cycles per 128-byte LINE it touches (min 16 per KiB): sixteen
//     64-byte row segments a filter row apart - the shape every weight tile had so far - cap a CU at 28-32 B/clk, a contiguous
//     KiB runs at 56-64 B/clk; and ONE wave that issues its pieces back to back ingests 45-65 GB/s (rounds 1-4 believed a wave
//     was capped at 6-10 GB/s - the probe that said so divided 64-bit integers per issue - and spent 4-8 producer waves on it);
//   * a K loop of MFMAs alone keeps the matrix pipe ~89 % busy, the same loop with one s_barrier per step 66 %: twelve waves that
//     meet 12-24 times per tile wait for the slowest every time;
//   * once the barriers are gone the loop is bound by LDS bandwidth: 64 px x 64 ch wave tiles read one 1-KiB fragment per MFMA
//     (50 % of the LDS cycles before bank conflicts and the DMA's own writes), and a band with zero columns (row pitch Wo + 2)
//     cannot be read conflict-free at all on 14 x 14 / 7 x 7 maps (16 lanes of a ds_read_b128 group, 14 distinct bank groups).
// Therefore:
//   * weights are packed on the host into the exact byte stream a workgroup consumes: [Cout/64][slice][kh][kw][64 rows][64 B],
//     rows pre-swizzled for conflict-free fragment reads, so a filter-row step is 12 contiguous KiB (hawq_pack_w3x3_band);
//   * activations arrive as channel-group planes (hawq_conv_args.in_planar) and the band is DENSE: band pixel j is global pixel
//     mb0 + j, a tap is a constant pixel offset (kh - 1) * Wo + (kw - 1), lanes whose tap falls outside their image row / column
//     read one shared zero word instead (9 validity bits per lane, one v_cndmask per fragment address).  Consecutive lanes read
//     consecutive 16-byte units: conflict-free on every map size, every band piece is a contiguous, line-aligned KiB
//     (`buffer_load ... lds`: pixels before / behind the tensor are out of range and the hardware writes zeros), and the band is
//     a quarter smaller;
//   * 2-4 producer waves own every LDS-DMA; no workgroup barrier inside the K loop: landed[] / done[] counters in LDS (below);
//   * an MFMA wave owns PT x 2 MFMA tiles (PT = 4: 128 px x 64 ch, 128 accumulators, 0.75 fragment reads per MFMA; PT = 2:
//     64 px x 64 ch) and the waves of a workgroup split K: wave group g takes k-half g (channels 32g .. 32g+31 of every
//     64-channel slice).  The two partial accumulator sets meet once, after the K loop, through LDS, each wave keeping half of
//     its pixel tiles - which also halves the requantisation work per wave;
//   * the epilogue needs no staging tile: a lane holds 16 consecutive channels of one pixel = one 16-byte store (NHWC rows)
//     or 32 lanes x 16 B = 512 contiguous bytes (planes).
// One step = one filter row of one 64-channel slice (3 taps, 12 KiB of weights); W ring of WS stages, band double-buffered
// per slice.
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace {

struct B2P {
    const char *in;        // planes [Cin/16][M][16 B] int8
    const char *wgt;       // hawq_pack_w3x3_band stream
    const int32_t *ctab;   // [Cout][4]
    const int32_t *bias;   // RAW only
    int32_t *out_acc;      // RAW only: [M][Cout] int32 (accumulators + bias)
    char *out;             // NHWC [M][Cout] int8, or planes [Cout/16][M][16 B]
    const char *res_in;    // RESIDUAL: [M][Cout] uint16
    char *res_out;         // RESIDUAL: [M][Cout] uint16 or null
    int *flags;
    int M, Ho, Wo, Cin, Cout;
    int nsteps, cchunks;
    int out_planar, out_bits, q_lo, q_hi;
    int mq, eq, m_id, e_id;
    unsigned in_bytes, wgt_bytes;
    int dbg;
    long long *dbgbuf;
};

template <int PT_, int WM_, int NPROD_, int WS_, int BAND_PX_, int MINW_>
struct V2Cfg {
    static constexpr int PT = PT_, WM = WM_, NPROD = NPROD_, WS = WS_, BAND_PX = BAND_PX_, MINW = MINW_;   // MINW: waves per SIMD the register budget is cut for
    static constexpr int BM = 32 * PT * WM, NW = 2 * WM, NT = (NW + NPROD) * 64;
    static constexpr int PLANE = BAND_PX * 16, BAND_BYTES = 4 * PLANE;
    static constexpr int WTAP = 4096, WSTAGE = 3 * WTAP;
    static constexpr int OFF_W = 2 * BAND_BYTES, RING_END = OFF_W + WS * WSTAGE;
    // The last four pixels of plane 3 of either band stage are never read as pixels (launcher: band length <= BAND_PX - 4) and never
    // written by the band fill, which masks their four lanes off (an LDS-DMA writes nothing for lanes EXEC has switched off:
    // tools/ubench/dma_exec.hip).  Byte BAND_BYTES - 16 of either stage is the zero word the out-of-image taps read (the SAME offset in
    // both stages: a tap address is stage base + a per-lane constant), bytes BAND_BYTES - 64 .. - 17 of stage 1 are the
    // synchronisation words.  A tile that must fit 80 KiB twice per CU has no other byte to spare: its requant constants land in the
    // first ring stage the K loop releases for good; the larger tiles have the CU to themselves and own a KiB for them.
    static constexpr bool TIGHT = RING_END == 80 * 1024;
    static constexpr int OFF_CTAB = RING_END, OFF_SYNC = 2 * BAND_BYTES - 64, ZERO_OFF = BAND_BYTES - 16;
    static constexpr int LDS_BYTES = RING_END + (TIGHT ? 0 : 1024);
    static constexpr int PG = BAND_PX / 64;             // 64-pixel groups per plane
    static constexpr int PPP = 4 / NPROD;               // planes per producer wave
    static constexpr int BPI = PG * PPP, WPI = 12 / NPROD;
    static constexpr int HQ = PT / 2;                   // pixel tiles a wave keeps after the exchange
    static constexpr int XCH = HQ * 2 * 16 * 256;       // exchange area per MFMA wave: HQ x 2 tiles x 16 registers x 64 lanes x 4 B
    static_assert(PT == 2 || PT == 4, "pixel tiles per MFMA wave");
    static_assert(NPROD == 2 || NPROD == 4, "producer waves");
    static_assert(BAND_PX % 64 == 0 && WS >= 4 && WS <= 5, "ring");
    static_assert(NW * XCH <= (TIGHT ? 2 * BAND_BYTES : RING_END), "the partial-sum exchange reuses the band area (and the ring)");
};

__device__ __forceinline__ void bdma(__amdgpu_buffer_rsrc_t r, char *lds, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)lds, 16, voff, soff, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vm_upto(int n) {   // s_waitcnt vmcnt(n) for a wave-uniform n in [0, N]
    if constexpr (N == 0) {
        wait_vmcnt<0>();
    } else {
        if (n >= N) wait_vmcnt<N>(); else wait_vm_upto<N - 1>(n);
    }
}

// ---- flag synchronisation: no workgroup barrier inside the K loop.
//   landed[p] (producer wave p): number of steps whose operands - this producer's share - are in LDS; written after the covering vmcnt wait
//   done[w]   (MFMA wave w)    : number of steps whose fragment reads have all returned
// live in LDS.  An MFMA wave reads landed[] ahead of time (one broadcast ds_read_b128 per step, covered by the fragment waits) and only
// spins when data is really late; a producer refills ring stage (i - 1) % WS once min(done[]) >= i.  LDS serves the DS instructions of a
// CU in order and an LDS-DMA piece is in LDS when its wave's vmcnt says so, hence flag-after-data on the writer's side and
// data-after-flag on the reader's side is all the ordering there is.
__device__ __forceinline__ int min4(const v4i &v) { return min(min(v.x, v.y), min(v.z, v.w)); }
__device__ __forceinline__ int lds_min4_now(unsigned addr) {   // min of the 4 dwords at a wave-uniform LDS address, as a scalar
    v4i v;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return __builtin_amdgcn_readfirstlane(min4(v));
}
__device__ __forceinline__ void lds_store_b32(unsigned addr, int v) { asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory"); }

// hawq4 operands (both 4-bit, Cin % 128 == 0): 8 channels per dword, (c0 | c1 << 8 | c2 << 16 | c3 << 24) | (c4 .. c7 likewise) << 4
// (include/hawq_mi355.h).  A 64-byte slice is then 128 channels; the packed bytes travel through LDS untouched and every 16-byte
// fragment (32 channels) feeds TWO MFMA K-steps after unpacking in registers - activations zero-extended, weights as value * 16 (the
// nibble moved to the top of its byte: sign for free), the sums shifted back by 4 once, which is exact.
template <bool WEIGHT>
__device__ __forceinline__ v4i b2_unpack16(unsigned x0, unsigned x1) {
    v4i r;
    if (WEIGHT) {
        r.x = (int)((x0 << 4) & 0xF0F0F0F0u), r.y = (int)(x0 & 0xF0F0F0F0u), r.z = (int)((x1 << 4) & 0xF0F0F0F0u), r.w = (int)(x1 & 0xF0F0F0F0u);
    } else {
        r.x = (int)(x0 & 0x0F0F0F0Fu), r.y = (int)((x0 >> 4) & 0x0F0F0F0Fu), r.z = (int)(x1 & 0x0F0F0F0Fu), r.w = (int)((x1 >> 4) & 0x0F0F0F0Fu);
    }
    return r;
}
// this lane's 16 consecutive requantised channels (4 dwords of 4 bytes each, already clamped) of pixel m, first channel ch: int8 or hawq4,
// NHWC rows or channel-group planes (16 / 32 channels per 16-byte unit)
__device__ __forceinline__ void b2_store_q(const B2P &p, const int (&w)[4], int m, int ch) {
    if (p.out_bits == 8) {
        const v4i ww = {w[0], w[1], w[2], w[3]};
        char *dst = p.out_planar ? p.out + ((size_t)(ch >> 4) * p.M + m) * 16 : p.out + (size_t)m * p.Cout + ch;
        *reinterpret_cast<v4i *>(dst) = ww;
    } else {   // bytes hold values 0 .. 15: channels 4k .. 4k+3 in w[k]; low nibbles = channels 0-3 of an 8-group, high = 4-7
        const v2i ww = {w[0] | (w[1] << 4), w[2] | (w[3] << 4)};
        char *dst = p.out_planar ? p.out + ((size_t)(ch >> 5) * p.M + m) * 16 + ((ch >> 4) & 1) * 8 : p.out + (((size_t)m * p.Cout + ch) >> 1);
        *reinterpret_cast<v2i *>(dst) = ww;
    }
}

// EPI: HAWQ_EPI_REQUANT, HAWQ_EPI_RESIDUAL (single branch, uint16 residuals: the second conv of a basic block), or HAWQ_EPI_RAW (round 6:
// the int32 accumulators + bias as a dense [M][Cout] tensor - what the parity tests compare with the oracle's exact sums).
// MODE: 0 = tie-free tables, 2 = exact-tie correction on every requant (fast_tables bit 2).
// NIB: both operands hawq4.
template <class C, int EPI, int MODE, bool NIB>
__global__ __launch_bounds__(C::NT, C::MINW) void conv3x3_v2_kernel(const B2P p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const bool prof = HAWQ_DBG_BIT(p.dbg, 128) && p.dbgbuf;
    const long long t_entry = prof ? (long long)__builtin_readcyclecounter() : 0;
    const int tiles_c = p.Cout >> 6;
    const int nwg = ((p.M + C::BM - 1) / C::BM) * tiles_c;
    int wg = blockIdx.x;
    {   // each XCD (id mod 8) owns a contiguous run of tiles: the channel tiles of a pixel tile share its band in that L2
        const int q = nwg >> 3, r = nwg & 7, xcd = wg & 7, idx = wg >> 3;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tc = wg % tiles_c, tm = wg / tiles_c;
    const int m0 = tm * C::BM, c0 = tc << 6;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);   // scalar: role and k-half branches are scalar branches
    const int Wo = p.Wo;
    const int mb0 = (m0 - Wo - 1) & ~7;                                  // global pixel held by band pixel 0 (line-aligned; may be negative)
    const int mlast = (m0 + C::BM < p.M ? m0 + C::BM : p.M) - 1;
    const int npg = (mlast + Wo + 1 - mb0 + 64) >> 6;                    // 64-pixel groups of a plane that are filled at all
    const int nsteps = p.nsteps, cchunks = p.cchunks;
    char *const band = smem, *const wring = smem + C::OFF_W;
    char *const ctab_lds = C::TIGHT ? wring + (nsteps % C::WS) * C::WSTAGE : smem + C::OFF_CTAB;
    const unsigned sync_a = lds_addr(smem + C::OFF_SYNC);   // landed[4] at +0, done[8] at +16 (the zero word of band stage 1 at +48)

    if (wave >= C::NW) {
        // ------------------------------------------------------------------ producer waves: every LDS-DMA of the kernel
        const int dw = wave - C::NW;
        const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)p.in, 0, (int)p.in_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void *)p.wgt, 0, (int)p.wgt_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rct = __builtin_amdgcn_make_buffer_rsrc((void *)p.ctab, 0, EPI == HAWQ_EPI_RAW ? 0 : p.Cout * 16, 0x00020000);   // (RAW: no constants - zero records, the DMA writes zeros)
        unsigned bvo[C::PG];   // byte offset of this lane's pixel inside a plane, or out of range (the hardware then writes zeros)
#pragma unroll
        for (int g = 0; g < C::PG; ++g) {
            const int pix = mb0 + g * 64 + lane;
            bvo[g] = (unsigned)pix < (unsigned)p.M ? (unsigned)pix * 16u : 0x80000000u;
        }
        const unsigned wvo = (unsigned)(dw * 1024 + lane * 16);
        const unsigned plane_bytes = (unsigned)p.M * 16u;
        const unsigned wbase = (unsigned)tc * (unsigned)nsteps * (unsigned)C::WSTAGE;
        auto issue_band = [&](int cc) {   // returns the pieces issued
            char *dst = band + (cc & 1) * C::BAND_BYTES;
#pragma unroll
            for (int i = 0; i < C::PPP; ++i) {
                const int pl = dw + i * C::NPROD;
                const unsigned so = (unsigned)(cc * 4 + pl) * plane_bytes;
#pragma unroll
                for (int g = 0; g < C::PG; ++g)
                    if (g < npg) {
                        if (pl == 3 && g == C::PG - 1) {   // zero word / synchronisation words live in these four pixels
                            if (lane < 60) bdma(rin, dst + pl * C::PLANE + g * 1024, bvo[g], so);
                        } else {
                            bdma(rin, dst + pl * C::PLANE + g * 1024, bvo[g], so);
                        }
                    }
            }
            return npg * C::PPP;
        };
        auto issue_w = [&](int s) {
            char *dst = wring + (s % C::WS) * C::WSTAGE + dw * 1024;
            const unsigned so = wbase + (unsigned)s * (unsigned)C::WSTAGE;
#pragma unroll
            for (int i = 0; i < C::WPI; ++i) bdma(rw, dst + i * (C::NPROD * 1024), wvo, so + i * (C::NPROD * 1024));
        };
        auto publish = [&](int n) { lds_store_b32(sync_a + (unsigned)(dw * 4), n); };
        auto wait_done = [&](int n) {   // every MFMA wave has finished the fragment reads of steps < n
            while (true) {
                const int a = lds_min4_now(sync_a + 16), b = C::NW > 4 ? lds_min4_now(sync_a + 32) : 0x7fffffff;
                if (min(a, b) >= n) break;
                __builtin_amdgcn_s_sleep(1);
            }
        };
        // first operands on their way before anything else happens in this workgroup (W(0) and the band first: step 0 needs them)
        issue_w(0);
        const int nb0 = issue_band(0);
#pragma unroll
        for (int s = 1; s < C::WS - 1; ++s) issue_w(s);   // launcher: nsteps >= 6 > WS - 1
        const int nct = (!C::TIGHT && dw == 0) ? 1 : 0;
        if (nct) bdma(rct, ctab_lds, (unsigned)(lane * 16), (unsigned)(c0 * 16));
        (void)nb0;
        __builtin_amdgcn_s_barrier();   // the synchronisation words are initialised (MFMA wave 0); a bare barrier: no vmcnt(0) in front of it
        wait_vm_upto<(C::WS - 2) * C::WPI + 1>((C::WS - 2) * C::WPI + nct);   // W(0), band(0)
        publish(1);   // (the loop's first publish is WS - 2 >= 2)
        int cc = 0, kh = 0;
        int hw1 = C::WPI + nct;   // W (/ constants) pieces of the previous iteration - of the prologue's last stage at first
        for (int i = 0; i < nsteps; ++i) {
            // (1) Everything but the W pieces of the previous iteration has landed (LDS-DMA returns in order; an iteration issues its band
            //     pieces first): W(0 .. i + WS - 3), and every band issued so far - the band of slice c is issued in iteration 3c - 3, so the
            //     bands of steps <= i + 2 are there, which covers step i + WS - 3 as long as WS <= 5.  Publishing as far ahead as the ring
            //     allows matters: landed[] is what lets an MFMA wave run ahead of the slowest one, and with a count that trailed the ring by
            //     two stages the K loop ran at the pace of the flag round trips (1030 cycles per step with no arithmetic at all).
            wait_vm_upto<C::WPI + 1>(hw1);
            publish(i + C::WS - 2);
            // (2) refill: ring stage (i - 1) % WS and band stage (cc + 1) & 1 are free once every MFMA wave is done with step i - 1
            int nb = 0, nw = 0;
            const int s2 = i + C::WS - 1;
            const bool wantb = kh == 0 && cc + 1 < cchunks, wantw = s2 < nsteps, wantc = C::TIGHT && s2 == nsteps && dw == 0;
            if ((wantb || wantw || wantc) && !HAWQ_DBG_BIT(p.dbg, 1)) {
                wait_done(i);
                if (wantb && !HAWQ_DBG_BIT(p.dbg, 64)) nb = issue_band(cc + 1);
                if (wantw && !HAWQ_DBG_BIT(p.dbg, 32)) issue_w(s2), nw = C::WPI;
                if (wantc) bdma(rct, ctab_lds, (unsigned)(lane * 16), (unsigned)(c0 * 16)), nw = 1;
            }
            hw1 = nw;
            (void)nb;
            if (++kh == 3) kh = 0, ++cc;
        }
        wait_vmcnt<0>();
        publish(nsteps + 2);   // everything (the constants too) has landed; this wave touches the synchronisation words no more
        return;                // (the MFMA waves' epilogue barriers: ended waves are not counted by s_barrier)
    }

    // ---------------------------------------------------------------------- MFMA waves: PT x 2 tiles, k-half g of every slice
    if (t < 16) {
        const bool used = t < 4 ? t < C::NPROD : (t < 12 ? t - 4 < C::NW : false);
        *reinterpret_cast<int *>(smem + C::OFF_SYNC + t * 4) = t >= 12 ? 0 : (used ? 0 : 0x7fffffff);   // stage 1: counters + zero word
        *reinterpret_cast<int *>(smem + C::BAND_BYTES - 64 + t * 4) = 0;                                  // stage 0: zero word
    }
    const int wave_m = wave % C::WM, g = wave / C::WM;
    const int l31 = lane & 31, h = lane >> 5;
    // byte offset of tap (kh, kw) of this lane's pixel from the base of a band stage: its k-half's plane + the dense band pixel, or the
    // stage's zero word when the tap leaves the pixel's image (the only per-step address arithmetic left is one add per fragment)
    unsigned off[C::PT][9];
#pragma unroll
    for (int q = 0; q < C::PT; ++q) {
        int m = m0 + wave_m * (32 * C::PT) + q * 32 + l31;
        m = m < p.M ? m : p.M - 1;
        const int Gr = m / Wo, x = m - Gr * Wo, y = Gr % p.Ho;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const bool v = (unsigned)(y + kh - 1) < (unsigned)p.Ho && (unsigned)(x + kw - 1) < (unsigned)Wo;
                off[q][kh * 3 + kw] = v ? (unsigned)((2 * g + h) * C::PLANE + (m - mb0 + (kh - 1) * Wo + kw - 1) * 16) : (unsigned)C::ZERO_OFF;
            }
    }
    unsigned wofs[2];   // A-fragment byte offsets inside a tap tile (swizzled rows), this group's k-half
#pragma unroll
    for (int c = 0; c < 2; ++c) wofs[c] = lds_off(c * 32 + cperm(l31), 2 * g + h);
    const unsigned band_a = lds_addr(band), wring_a = lds_addr(wring);

    v16i acc[2][C::PT];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int q = 0; q < C::PT; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][q][r] = 0;

    // three fragment buffers, one per tap of a step (fixed roles: no buffer parity across steps); the fragments of the next tap -
    // after tap 2: tap 0 of the next step - are requested before the MFMAs of the current one
    v4i wf[3][2], af[3][C::PT];
#define V2_FETCH(KH, KW, BST, WST)                                                                    \
    if (!HAWQ_DBG_BIT(p.dbg, 4)) {                                                                    \
        _Pragma("unroll") for (int c = 0; c < 2; ++c) wf[KW][c] = lds_read16<(KW) * C::WTAP>((WST) + wofs[c]); \
        _Pragma("unroll") for (int q = 0; q < C::PT; ++q) af[KW][q] = lds_read16<0>((BST) + off[q][(KH) * 3 + (KW)]); \
    }
#define V2_MMA(KW)                                                                                    \
    {                                                                                                 \
        _Pragma("unroll") for (int c = 0; c < 2; ++c) pin(wf[KW][c]);                                  \
        _Pragma("unroll") for (int q = 0; q < C::PT; ++q) pin(af[KW][q]);                              \
        if (!HAWQ_DBG_BIT(p.dbg, 2)) {                                                                \
            if constexpr (NIB) {                                                                      \
                _Pragma("unroll") for (int hf = 0; hf < 2; ++hf) {                                    \
                    v4i w8[2], a8[C::PT];                                                             \
                    _Pragma("unroll") for (int c = 0; c < 2; ++c) w8[c] = b2_unpack16<true>((unsigned)wf[KW][c][2 * hf], (unsigned)wf[KW][c][2 * hf + 1]); \
                    _Pragma("unroll") for (int q = 0; q < C::PT; ++q) a8[q] = b2_unpack16<false>((unsigned)af[KW][q][2 * hf], (unsigned)af[KW][q][2 * hf + 1]); \
                    _Pragma("unroll") for (int q = 0; q < C::PT; ++q)                                 \
                        _Pragma("unroll") for (int c = 0; c < 2; ++c)                                 \
                            acc[c][q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(w8[c], a8[q], acc[c][q], 0, 0, 0); \
                }                                                                                     \
            } else {                                                                                  \
                _Pragma("unroll") for (int q = 0; q < C::PT; ++q)                                     \
                    _Pragma("unroll") for (int c = 0; c < 2; ++c)                                     \
                        acc[c][q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[KW][c], af[KW][q], acc[c][q], 0, 0, 0); \
            }                                                                                         \
        }                                                                                             \
    }
    constexpr int NF = 2 + C::PT;   // fragment reads per tap
    const long long t_begin = prof ? (long long)__builtin_readcyclecounter() : 0;
    __syncthreads();   // the synchronisation words are initialised
    const unsigned done_a = sync_a + 16u + (unsigned)(wave * 4);
    v4i pl;   // landed[] as requested at the top of a step
    auto ensure = [&](int n) {   // every producer's landed[] >= n; `pl` is covered by a fragment wait, the slow path spins
        pin(pl);
        if (__builtin_amdgcn_readfirstlane(min4(pl)) >= n) return;
        while (lds_min4_now(sync_a) < n) __builtin_amdgcn_s_sleep(1);
    };
    while (lds_min4_now(sync_a) < 1) __builtin_amdgcn_s_sleep(1);   // band(0), W(0)
    const long long t_b0 = prof ? (long long)__builtin_readcyclecounter() : 0;
    __builtin_amdgcn_s_setprio(2);
    {
        int s = 0, ws = 0;   // step, its ring stage
        unsigned bst = band_a, wst = wring_a;
        V2_FETCH(0, 0, bst, wst)
        // one step: taps kw = 0 / 1 / 2 of filter row KH; NKH / bnext: filter row and band stage of the next step
#define V2_STEP(KH, NKH, BNEXT, LAST)                                                                       \
        {                                                                                                   \
            asm volatile("ds_read_b128 %0, %1" : "=v"(pl) : "v"(sync_a) : "memory");   /* for the prefetch below */ \
            V2_FETCH(KH, 1, bst, wst) wait_lgkm<NF>(); V2_MMA(0)                                            \
            V2_FETCH(KH, 2, bst, wst) wait_lgkm<NF>(); V2_MMA(1)                                            \
            if (++ws == C::WS) ws = 0;                                                                      \
            const unsigned wnext = wring_a + (unsigned)(ws * C::WSTAGE);                                    \
            if (!(LAST)) {   /* first fragments of the next step */                                         \
                ensure(s + 2);                                                                              \
                V2_FETCH(NKH, 0, BNEXT, wnext)                                                              \
                wait_lgkm<NF>();                                                                            \
            } else {                                                                                        \
                wait_lgkm<0>();                                                                             \
            }                                                                                               \
            lds_store_b32(done_a, s + 1);   /* every fragment read of step s has returned: ring stage (and band) free */ \
            V2_MMA(2)                                                                                       \
            wst = wnext, ++s;                                                                               \
        }
        for (int cc = 0; cc < cchunks; ++cc) {
            const unsigned bother = band_a + (unsigned)(((cc + 1) & 1) * C::BAND_BYTES);
            const bool last = cc + 1 == cchunks;
            V2_STEP(0, 1, bst, false)
            V2_STEP(1, 2, bst, false)
            V2_STEP(2, 0, bother, last)
            bst = bother;
        }
#undef V2_STEP
    }
#undef V2_FETCH
#undef V2_MMA
    __builtin_amdgcn_s_setprio(0);
    const long long t_loop_end = prof ? (long long)__builtin_readcyclecounter() : 0;

    // RESIDUAL: this lane's old residual values (its HQ pixels x 2 x 16 channels, 32 contiguous bytes each) are requested now and
    // arrive under the exchange
    v4i rin[C::HQ][2][2];
    if constexpr (EPI == HAWQ_EPI_RESIDUAL) {
#pragma unroll
        for (int qq = 0; qq < C::HQ; ++qq) {
            int m = m0 + wave_m * (32 * C::PT) + (g * C::HQ + qq) * 32 + l31;
            m = m < p.M ? m : p.M - 1;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const v4i *rp = reinterpret_cast<const v4i *>(p.res_in + ((size_t)m * p.Cout + c0 + c * 32 + h * 16) * 2);
                rin[qq][c][0] = rp[0], rin[qq][c][1] = rp[1];
            }
        }
    }
    while (lds_min4_now(sync_a) < nsteps + 2) __builtin_amdgcn_s_sleep(1);   // the producers are through (requant constants landed)
    __syncthreads();   // MFMA waves only (ended waves do not count): every wave is done with band and ring, which the exchange reuses
    // ---------------------------------------------------------------------- partial sums of the two k-halves meet in LDS
    // wave (wave_m, g) keeps pixel tiles g * HQ .. g * HQ + HQ - 1 and hands the accumulators of the others to wave (wave_m, 1 - g)
    v16i sum[2][C::HQ];
    auto exchange = [&](auto G) {
        constexpr int gg = decltype(G)::value;
        char *dst = smem + (wave_m + (1 - gg) * C::WM) * C::XCH + lane * 16;
#pragma unroll
        for (int qq = 0; qq < C::HQ; ++qq)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const v16i &a = acc[c][(1 - gg) * C::HQ + qq];
                    const v4i v = {a[4 * i], a[4 * i + 1], a[4 * i + 2], a[4 * i + 3]};
                    *reinterpret_cast<v4i *>(dst + ((qq * 2 + c) * 4 + i) * 1024) = v;
                }
        __syncthreads();
        const char *src = smem + wave * C::XCH + lane * 16;
#pragma unroll
        for (int qq = 0; qq < C::HQ; ++qq)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const v4i v = *reinterpret_cast<const v4i *>(src + ((qq * 2 + c) * 4 + i) * 1024);
                    const v16i &a = acc[c][gg * C::HQ + qq];
#pragma unroll
                    for (int j = 0; j < 4; ++j) sum[c][qq][4 * i + j] = NIB ? (a[4 * i + j] + v[j]) >> 4 : a[4 * i + j] + v[j];   // (weights were value * 16)
                }
    };
    if (g == 0) exchange(std::integral_constant<int, 0>{}); else exchange(std::integral_constant<int, 1>{});
    const long long t_xch = prof ? (long long)__builtin_readcyclecounter() : 0;
    // ---------------------------------------------------------------------- requantisation + stores (no staging tile)
    if constexpr (EPI == HAWQ_EPI_RAW) {   // a lane's 16 consecutive channels of one pixel: 64 contiguous bytes
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int ch = c0 + c * 32 + h * 16;
#pragma unroll
            for (int qq = 0; qq < C::HQ; ++qq) {
                const int m = m0 + wave_m * (32 * C::PT) + (g * C::HQ + qq) * 32 + l31;
                if (m >= p.M) continue;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const v4i b4 = *reinterpret_cast<const v4i *>(p.bias + ch + 4 * gq);
                    const v4i v = {sum[c][qq][4 * gq] + b4.x, sum[c][qq][4 * gq + 1] + b4.y, sum[c][qq][4 * gq + 2] + b4.z, sum[c][qq][4 * gq + 3] + b4.w};
                    reinterpret_cast<v4i *>(p.out_acc + (size_t)m * p.Cout + ch)[gq] = v;
                }
            }
        }
    }
    if constexpr (EPI == HAWQ_EPI_REQUANT) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int lch = c * 32 + h * 16;
            int w[C::HQ][4];
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                DyNt d[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const v4i e = *reinterpret_cast<const v4i *>(ctab_lds + (lch + 4 * gq + j) * 16);
                    d[j].m = e.x, d[j].s = e.y & 31, d[j].k = e.y >> 8;
                    d[j].add = (long long)(((unsigned long long)(unsigned)e.w << 32) | (unsigned)e.z);
                }
#pragma unroll
                for (int qq = 0; qq < C::HQ; ++qq) {
                    int qv[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) qv[j] = med3i(dyadic_mode<MODE>(sum[c][qq][4 * gq + j], d[j]), p.q_lo, p.q_hi);
                    w[qq][gq] = pack4_fast(qv[0], qv[1], qv[2], qv[3]);
                }
            }
#pragma unroll
            for (int qq = 0; qq < C::HQ; ++qq) {
                const int m = m0 + wave_m * (32 * C::PT) + (g * C::HQ + qq) * 32 + l31;
                if (m < p.M && !HAWQ_DBG_BIT(p.dbg, 8)) b2_store_q(p, w[qq], m, c0 + lch);
            }
        }
    }
    if constexpr (EPI == HAWQ_EPI_RESIDUAL) {
        // second conv of a basic block (q_resnet.py:300-316): o = ReLU(requant(acc + bias) + requant(identity)), un-clamped
        // (quant_utils.py:415-456), stored as uint16 with the sticky overflow flag; q = the next block's QuantAct of o
        DyNt dids = dynt_prepare(p.m_id, p.e_id), dq = dynt_prepare(p.mq, p.eq);
        asm volatile("" : "+v"(dids.add), "+v"(dq.add));
        const int qhi2 = (p.q_hi & 0xffff) | (p.q_hi << 16);
        unsigned oor = 0;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int lch = c * 32 + h * 16;
            int w[C::HQ][4], rp[C::HQ][8];
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                DyNt d[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const v4i e = *reinterpret_cast<const v4i *>(ctab_lds + (lch + 4 * gq + j) * 16);
                    d[j].m = e.x, d[j].s = e.y & 31, d[j].k = e.y >> 8;
                    d[j].add = (long long)(((unsigned long long)(unsigned)e.w << 32) | (unsigned)e.z);
                }
#pragma unroll
                for (int qq = 0; qq < C::HQ; ++qq) {
                    const unsigned w0 = (unsigned)rin[qq][c][gq >> 1][(gq & 1) * 2], w1 = (unsigned)rin[qq][c][gq >> 1][(gq & 1) * 2 + 1];
                    const int idin[4] = {(int)(w0 & 0xffffu), (int)(w0 >> 16), (int)(w1 & 0xffffu), (int)(w1 >> 16)};
                    int o[4], qv[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int a = dyadic_mode<MODE>(sum[c][qq][4 * gq + j], d[j]);
                        const int b = dyadic_mode<MODE == 2 ? 2 : 0>(idin[j], dids);
                        o[j] = max(a + b, 0);
                        qv[j] = dyadic_mode<MODE>(o[j], dq);   // o >= 0, m >= 0: q >= 0 >= q_lo; clamped from above in the pack
                    }
                    const int mrow = m0 + wave_m * (32 * C::PT) + (g * C::HQ + qq) * 32 + l31;
                    if (mrow < p.M) oor |= (unsigned)(o[0] | o[1]) | (unsigned)(o[2] | o[3]);
                    rp[qq][2 * gq] = pack2_u16_sat(o[0], o[1]);
                    rp[qq][2 * gq + 1] = pack2_u16_sat(o[2], o[3]);
                    w[qq][gq] = pack4_min(qv[0], qv[1], qv[2], qv[3], qhi2);
                }
            }
#pragma unroll
            for (int qq = 0; qq < C::HQ; ++qq) {
                const int m = m0 + wave_m * (32 * C::PT) + (g * C::HQ + qq) * 32 + l31;
                if (m < p.M && !HAWQ_DBG_BIT(p.dbg, 8)) {
                    if (p.res_out) {
                        v4i *dst = reinterpret_cast<v4i *>(p.res_out + ((size_t)m * p.Cout + c0 + lch) * 2);
                        const v4i ra = {rp[qq][0], rp[qq][1], rp[qq][2], rp[qq][3]}, rb = {rp[qq][4], rp[qq][5], rp[qq][6], rp[qq][7]};
                        dst[0] = ra, dst[1] = rb;
                    }
                    if (p.out) b2_store_q(p, w[qq], m, c0 + lch);
                }
            }
        }
        if ((oor >> 16) != 0 && p.res_out && !HAWQ_DBG_BIT(p.dbg, ~0)) atomicOr(p.flags, 1);
    }
    if (prof && blockIdx.x == 8 && t == 0) {
        p.dbgbuf[0] = t_begin - t_entry, p.dbgbuf[1] = t_loop_end - t_begin;
        p.dbgbuf[2] = (long long)__builtin_readcyclecounter() - t_loop_end, p.dbgbuf[3] = nsteps;
        p.dbgbuf[4] = t_b0 - t_begin, p.dbgbuf[5] = t_xch - t_loop_end;
    }
}

// <PT, WM, NPROD, WS, BAND_PX, MINW>.  (PT = 4 - 128 px x 64 ch per MFMA wave, 0.75 fragment reads per MFMA - compiles and is exact, but
// 246 registers leave ONE MFMA wave per SIMD, whose own address / wait / flag instructions then sit between its MFMAs: 13.6 against
// 12.4 us on the 14 x 14 layer, profiles/r05_band_v2.md.  Not instantiated.)
using V128 = V2Cfg<2, 2, 2, 4, 256, 3>;    // 128 px x 64 ch: 4 MFMA waves (64 x 64 each, 2 k-halves) + 2 producers, 80 KiB: two workgroups per CU
using V256 = V2Cfg<2, 4, 4, 5, 384, 3>;    // 256 px x 64 ch: 8 MFMA waves + 4 producers, 5-stage ring, 109 KiB
constexpr int NUM_V2 = 2;

typedef void (*V2Fn)(const B2P);
struct V2Info { V2Fn fn[2][2][2]; V2Fn raw[2]; int bm, band_px, lds, nt, tight; };   // fn[hawq4][residual][exact-tie]; raw[hawq4]
#define V2_FNS(CFG, N) {{conv3x3_v2_kernel<CFG, HAWQ_EPI_REQUANT, 0, N>, conv3x3_v2_kernel<CFG, HAWQ_EPI_REQUANT, 2, N>}, \
                        {conv3x3_v2_kernel<CFG, HAWQ_EPI_RESIDUAL, 0, N>, conv3x3_v2_kernel<CFG, HAWQ_EPI_RESIDUAL, 2, N>}}
#define V2_ENTRY(CFG) {{V2_FNS(CFG, false), V2_FNS(CFG, true)}, {conv3x3_v2_kernel<CFG, HAWQ_EPI_RAW, 0, false>, conv3x3_v2_kernel<CFG, HAWQ_EPI_RAW, 0, true>}, \
                       CFG::BM, CFG::BAND_PX, CFG::LDS_BYTES, CFG::NT, CFG::TIGHT}
const V2Info kV2[NUM_V2] = {V2_ENTRY(V128), V2_ENTRY(V256)};

}  // namespace

int band_v2_count(void) { return NUM_V2; }

// [Cout][3][3][Cin] int8 -> [Cout/64][Cin/64][kh][kw][64 rows][64 B], slot s of row r at r * 64 + ((s ^ ((r >> 2) & 3)) << 4) (lds_off)
extern "C" int hawq_pack_w3x3_band(const int8_t *src, int8_t *dst, int32_t Cout, int32_t Cin) {
    HAWQ_REQUIRE(src && dst && Cout > 0 && Cin > 0 && Cout % 64 == 0 && Cin % 64 == 0, "hawq_pack_w3x3_band: Cout / Cin must be positive multiples of 64");
    const int cch = Cin >> 6;
    for (int ct = 0; ct < (Cout >> 6); ++ct)
        for (int cc = 0; cc < cch; ++cc)
            for (int tap = 0; tap < 9; ++tap) {
                int8_t *tile = dst + ((size_t)(ct * cch + cc) * 9 + tap) * 4096;
                for (int r = 0; r < 64; ++r) {
                    const int8_t *row = src + ((size_t)(ct * 64 + r) * 9 + tap) * Cin + cc * 64;
                    for (int sl = 0; sl < 4; ++sl)
                        for (int b = 0; b < 16; ++b) tile[r * 64 + ((sl ^ ((r >> 2) & 3)) << 4) + b] = row[sl * 16 + b];
                }
            }
    return 0;
}

bool band_v2_applies(const hawq_conv_args *a, int v) {
    if (v < 0 || v >= NUM_V2) return false;
    const V2Info &vi = kV2[v];
    const long long M = (long long)a->N * a->H * a->W;
    const int band_len = vi.bm + 2 * a->W + 2 + 7;   // pixels m0 - Wo - 1 .. m0 + BM + Wo, start rounded down to a line
    const bool qout_ok = a->out_bits == 8 || (a->out_bits == 4 && a->q_lo >= 0 && a->q_hi <= 15 && a->Cout % 32 == 0);
    const bool raw = a->epilogue == HAWQ_EPI_RAW;
    const bool epi_ok = (raw && a->out_acc && a->bias) ||
                        (a->epilogue == HAWQ_EPI_REQUANT && a->out_q && qout_ok && (a->out_bits == 8 || a->relu || a->q_lo >= 0)) ||
                        (a->epilogue == HAWQ_EPI_RESIDUAL && a->res_in && a->res_in_bits == 16 && (!a->res_out || (a->res_out_bits == 16 && a->flags)) &&
                         !a->res_no_relu && !a->res_clamp16 && (a->res_out || a->out_q) && (!a->out_q || qout_ok));
    const bool nib = a->in_bits == 4 && a->w_bits == 4;
    const int rowb = nib ? a->Cin >> 1 : a->Cin;   // bytes per pixel
    return a->KH == 3 && a->KW == 3 && a->stride == 1 && a->pad == 1 && a->in2 == nullptr && (raw || (a->fast_tables != 0 && a->ctab)) && a->wgt_band != nullptr &&
           a->in_planar == 1 && epi_ok && ((a->in_bits == 8 && a->w_bits == 8) || (nib && a->Cin % 128 == 0)) &&
           (a->in_pitch == 0 || a->in_pitch == rowb) && (a->out_pitch == 0 || a->out_pitch == a->Cout) &&
           band_len <= vi.band_px - 4 && rowb / 64 * 3 >= 6 && M * rowb < (1ll << 31) && (long long)a->Cout * rowb * 9 < (1ll << 31);
}

int band_v2_launch(const hawq_conv_args *a, int v, int exact_tie, int dbg, void *stream) {
    const V2Info &vi = kV2[v];
    B2P p;
    p.in = (const char *)a->in, p.wgt = (const char *)a->wgt_band, p.ctab = a->ctab, p.out = (char *)a->out_q;
    p.bias = a->bias, p.out_acc = a->out_acc;
    p.res_in = (const char *)a->res_in, p.res_out = (char *)a->res_out, p.flags = a->flags;
    p.M = a->N * a->H * a->W, p.Ho = a->H, p.Wo = a->W, p.Cin = a->Cin, p.Cout = a->Cout;
    const bool nib = a->in_bits == 4;
    const int rowb = nib ? a->Cin >> 1 : a->Cin;
    p.cchunks = rowb >> 6, p.nsteps = 3 * p.cchunks;
    p.out_planar = a->out_planar, p.out_bits = a->out_bits;
    p.q_lo = a->relu && a->q_lo < 0 ? 0 : a->q_lo, p.q_hi = a->q_hi;
    p.mq = a->mq, p.eq = a->eq, p.m_id = a->m_id_scalar, p.e_id = a->e_id_scalar;
    if (a->epilogue == HAWQ_EPI_RESIDUAL && !a->out_q) p.mq = 0, p.eq = 33;   // no next QuantAct: a harmless table
    p.in_bytes = (unsigned)((long long)p.M * rowb), p.wgt_bytes = (unsigned)((long long)a->Cout * rowb * 9);
    p.dbg = dbg;
    static long long *dbg_dev = nullptr;
    if (HAWQ_DBG_BIT(dbg, 128) && !dbg_dev) (void)hipMalloc(&dbg_dev, 8 * sizeof(long long));
    p.dbgbuf = HAWQ_DBG_BIT(dbg, 128) ? dbg_dev : nullptr;
    static const bool attrs = [] {
        bool good = true;
        for (const V2Info &i : kV2)
        {
            for (int k = 0; k < 8; ++k) good &= hipFuncSetAttribute((const void *)i.fn[k >> 2][(k >> 1) & 1][k & 1], hipFuncAttributeMaxDynamicSharedMemorySize, i.lds) == hipSuccess;
            for (int k = 0; k < 2; ++k) good &= hipFuncSetAttribute((const void *)i.raw[k], hipFuncAttributeMaxDynamicSharedMemorySize, i.lds) == hipSuccess;
        }
        return good;
    }();
    HAWQ_REQUIRE(attrs, "hawq_conv2d: hipFuncSetAttribute failed for the round-5 3x3 kernels");
    const int grid = ((p.M + vi.bm - 1) / vi.bm) * (p.Cout >> 6);
    V2Fn fn = a->epilogue == HAWQ_EPI_RAW ? vi.raw[nib ? 1 : 0] : vi.fn[nib ? 1 : 0][a->epilogue == HAWQ_EPI_RESIDUAL ? 1 : 0][exact_tie ? 1 : 0];
    hipLaunchKernelGGL(fn, dim3(grid), dim3(vi.nt), vi.lds, (hipStream_t)stream, p);
    HAWQ_CHECK_HIP(hipGetLastError());
    if (p.dbgbuf) {   // experiment hook (synchronises!)
        long long hb[6];
        (void)hipStreamSynchronize((hipStream_t)stream);
        (void)hipMemcpy(hb, p.dbgbuf, sizeof(hb), hipMemcpyDeviceToHost);
        fprintf(stderr, "[band-v2 %d bm=%d M=%d Cin=%d Cout=%d grid=%d lds=%d] steps %lld: prologue %lld | K loop %lld | epilogue %lld cycles (wave 0 of workgroup 8); first operands %lld, exchange %lld\n",
                v, vi.bm, p.M, p.Cin, p.Cout, grid, vi.lds, hb[3], hb[0], hb[1], hb[2], hb[4], hb[5]);
    }
    return 0;
}
