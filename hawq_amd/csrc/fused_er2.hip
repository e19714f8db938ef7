// Software-pipelined "expand -> reduce" launch for gfx950 (MI355X), round 4: the pair of fused_er.hip re-scheduled so that the
// matrix pipe works UNDER the requantisation arithmetic instead of beside it.
//
// Same arithmetic, same layers, same arguments as fused_er.hip (q_resnet.py:231-260, quant_utils.py:390-456): a workgroup owns BM
// pixels and walks the expand conv's C3 output channels in slices of 64,
//     GEMM1(j)  acc1[BM x 64] = x2[BM x C] . W3[slice j]^T
//     epi 1(j)  o = ReLU(requant(acc1) + requant(residual)); residual slice out (uint16); q(j) = QuantAct_{i+1}(o) -> LDS
//     GEMM2(j)  acc2[BM x C] += q(j)[BM x 64] . W1[:, slice j]^T
// fused_er.hip runs these phases in lock step with two workgroup barriers per slice: 5-6.5 k cycles per slice for ~1.0 k cycles of
// matrix pipe and ~1.5 k of VALU per SIMD (profiles/r02_fused_er_steps.md) - the waves wait at B1 for the weights, at B2 for q, and the
// 16 MFMAs of a slice never overlap the ~180 VALU instructions of its epilogue.  Here:
//   * ONE barrier per slice.  GEMM2 is delayed by a slice: iteration j runs GEMM1(j), then epilogue 1(j) with the MFMAs of GEMM2(j-1)
//     interleaved into its VALU stream (independent accumulators; the matrix pipe and the VALU issue side by side, MI355X_MICROARCH.md
//     "Wave scheduling"), so q(j-1) and W1(j-1) are consumed a whole iteration after they were produced and nobody waits for them;
//   * q, the residual staging tile and the ctab slice are double-buffered by slice parity; the coalesced residual stores of slice
//     j-1 leave at the top of iteration j (their staging tile was completed before barrier j);
//   * the old residual slice goes straight into registers (plain loads, 32 B per lane), each half re-loaded for slice j+1 as soon as
//     epilogue 1(j) has consumed it: the loads have the rest of the epilogue, the barrier, the stores and GEMM1(j+1) as cover;
//   * NP producer waves own every LDS-DMA instruction and run one slice ahead on a 4-slot weight ring (W3(j+1) and W1(j) land during
//     iteration j); compute waves issue no LDS-DMA at all, so hipcc keeps exact lgkmcnt bookkeeping for their fragment reads.
// int8 x int8, fast-contract tables (exact-tie mode as separate instantiations), uint16 residuals, single-branch residual units.
#include <stdio.h>
#include <stdlib.h>

#include "common.h"

namespace {

struct E2P {
    const uint8_t *x2, *w3, *w1;
    const int32_t *ctab3, *ctab1;
    const uint16_t *res_in;
    uint16_t *res_out;
    uint8_t *y;
    int M, C3;
    int m_id_s, e_id_s, mq, eq, q_hi;
    int y_lo, y_hi;
    int y_planar, y_nib;
    int32_t *flags;
    long long *dbgbuf;   // probe builds (HAWQ_ABLATE, HAWQ_DBG=128): per-phase cycle sums of compute wave 0 and producer wave 0 of workgroup 8
    int dbg;             // probe builds: HAWQ_DBG bits (256: phase-shift probe, see the slice loop)
};

__device__ __attribute__((aligned(16))) const int g_e2_zero16[4] = {0, 0, 0, 0};

// C: channels of the expand conv's input = of the reduce conv's output.  WM: 32-pixel MFMA tiles per workgroup (x 2 compute waves
// along the channel axis).  NP: producer waves.  MINB: waves per SIMD the register allocation must allow.
template <int C_, int WM_, int NP_, int MINB_, int SG_ = 4>
struct E2Cfg {
    static constexpr int C = C_, WM = WM_, NP = NP_, MINB = MINB_;
    static constexpr int SG = SG_;   // scheduling groups of epilogue 1: 4 = GEMM2's MFMAs pinned after every channel group, 2 = after every second one, 0 = the compiler places them
    static constexpr int BM = 32 * WM, NW = 2 * WM, NTC = 64 * NW, NT = NTC + 64 * NP;
    static constexpr int NPT = 64 * NP, RPP = NPT / 4;   // producer threads; operand rows per LDS-DMA pass (4 lanes x 16 B per 64-byte row)
    static constexpr int KC = C / 64;                    // 64-byte chunks of GEMM1's K
    static constexpr int CT2 = C / 64;                   // 32-channel MFMA tiles per compute wave in GEMM2 (2 waves across the C channels)
    static constexpr int WSLOT = 64 * C;                 // W3 slice [KC][64 rows][64 B]  ==  W1 slice [C rows][64 B]
    static constexpr int WPASS = WSLOT / (RPP * 64);     // LDS-DMA instructions per producer thread per ring slot
    static constexpr int XPASS = BM / RPP;               // per 64-byte chunk of the x2 tile
    static constexpr int X2_BYTES = BM * C, Q_BYTES = BM * 64, RES_BYTES = BM * 128;
    static constexpr int OFF_X2 = 0;
    static constexpr int OFF_RING = OFF_X2 + X2_BYTES;   // slots 0/1: W3(j) by parity, slots 2/3: W1(j) by parity
    static constexpr int OFF_Q = OFF_RING + 4 * WSLOT;   // [2][BM][64 B]
    static constexpr int OFF_RES = OFF_Q + 2 * Q_BYTES;  // [2][BM][64] uint16
    static constexpr int OFF_CT3 = OFF_RES + 2 * RES_BYTES;   // [2][64][16 B]
    static constexpr int LDS_BYTES = OFF_CT3 + 2048;
    static_assert(NP >= 1 && WSLOT % (RPP * 64) == 0 && BM % RPP == 0 && C % RPP == 0 && (BM * 8) % NTC == 0, "tiles must fill whole producer passes");
    static_assert(BM * C <= 4 * WSLOT, "the output tile is staged on the weight ring");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

__device__ __forceinline__ void e2_dma16(const char *src, char *dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src, (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
}
__device__ __forceinline__ void e2_dma4(const char *src, char *dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src, (__attribute__((address_space(3))) void *)dst, 4, 0, 0);
}
template <bool K0>
__device__ __forceinline__ DyNt e2_ctab_entry(const char *ctab, int ch) {
    const v4i t = *reinterpret_cast<const v4i *>(ctab + ch * 16);
    DyNt d;
    d.m = t.x, d.s = K0 ? t.y : (t.y & 31), d.k = K0 ? 0 : (t.y >> 8);
    d.add = (long long)(((unsigned long long)(unsigned)t.w << 32) | (unsigned)t.z);
    return d;
}

template <class F, bool TIE, bool CK0 = false, bool QK0 = false>
__global__ __launch_bounds__(F::NT, F::MINB) void expand_reduce_pipelined_kernel(const E2P p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int MODE = TIE ? 2 : 0;                    // scalar identity table (uniform pre-shift)
    constexpr int MODE_Q = TIE ? 2 : (QK0 ? 1 : 0);      // scalar table of the next QuantAct
    constexpr int MODE_C = TIE ? 2 : (CK0 ? 1 : 0);      // per-channel tables
    constexpr bool K0 = CK0 && !TIE;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int m0 = blockIdx.x * F::BM;
    const int n = p.C3 >> 6;   // slices (even: C3 is a multiple of 128 for every ResNet bottleneck)
    char *x2t = smem + F::OFF_X2, *ring = smem + F::OFF_RING, *qt = smem + F::OFF_Q, *rest = smem + F::OFF_RES, *ct3 = smem + F::OFF_CT3;
    const bool prof = HAWQ_DBG_BIT(~0, 128) && p.dbgbuf != nullptr;
    long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tprev = prof ? (long long)__builtin_readcyclecounter() : 0;
#define E2_STAMP(K)                                                    \
    if (prof) {                                                        \
        const long long now = (long long)__builtin_readcyclecounter(); \
        ph[K] += now - tprev;                                          \
        tprev = now;                                                   \
    }

    if (wave >= F::NW) {
        // ================================================================ producer waves: every LDS-DMA instruction, nothing else
        const int pt = t - F::NTC, pw = wave - F::NW;
        const int prow = pt >> 2, pslot = pt & 3;
        const int sw = (pslot ^ ((prow >> 2) & 3)) << 4;   // source-side swizzle of this thread's 16-byte slot (same for rows prow + k * RPP)
        const char *zero = reinterpret_cast<const char *>(g_e2_zero16);
        auto issue_w3 = [&](int j) {   // rows j*64 .. j*64+63 of W3 [C3][C], as KC chunks of [64 rows][64 B] -> slot j & 1
            char *dst = ring + (j & 1) * F::WSLOT + pw * 1024;
#pragma unroll
            for (int i = 0; i < F::WPASS; ++i) {
                const int idx = i * F::RPP + prow, chunk = idx >> 6, row = idx & 63;
                e2_dma16((const char *)p.w3 + (size_t)(j * 64 + row) * F::C + chunk * 64 + sw, dst + i * (F::RPP * 64));
            }
            if (pw < 4) e2_dma4((const char *)p.ctab3 + (size_t)j * 1024 + pw * 256 + lane * 4, ct3 + (j & 1) * 1024 + pw * 256);
        };
        auto issue_w1 = [&](int j) {   // columns j*64 .. j*64+63 of W1 [C][C3], as [C rows][64 B] -> slot 2 + (j & 1)
            char *dst = ring + (2 + (j & 1)) * F::WSLOT + pw * 1024;
#pragma unroll
            for (int i = 0; i < F::WPASS; ++i) {
                const int row = i * F::RPP + prow;
                e2_dma16((const char *)p.w1 + (size_t)row * p.C3 + j * 64 + sw, dst + i * (F::RPP * 64));
            }
        };
#pragma unroll
        for (int kc = 0; kc < F::KC; ++kc)   // resident x2 tile: KC chunks of [BM pixel rows][64 B]
#pragma unroll
            for (int i = 0; i < F::XPASS; ++i) {
                const int row = i * F::RPP + prow;
                e2_dma16(m0 + row < p.M ? (const char *)p.x2 + (size_t)(m0 + row) * F::C + kc * 64 + sw : zero,
                         x2t + kc * (F::BM * 64) + i * (F::RPP * 64) + pw * 1024);
            }
        issue_w3(0);
        wait_vmcnt<0>();
        E2_STAMP(0)
        __builtin_amdgcn_s_barrier();   // barrier(0)
        for (int j = 0; j < n; ++j) {   // during iteration j the compute waves read W3(j), W1(j-1), ctab3(j)
            E2_STAMP(1)
            if (j + 1 < n) issue_w3(j + 1);
            issue_w1(j);
            E2_STAMP(2)
            wait_vmcnt<0>();
            E2_STAMP(3)
            __builtin_amdgcn_s_barrier();   // barrier(j + 1)
        }
        if (prof && blockIdx.x == 8 && pt == 0)
            for (int k = 0; k < 4; ++k) p.dbgbuf[8 + k] = ph[k];
        __syncthreads();   // the two workgroup barriers of epilogue 2
        __syncthreads();
        return;
    }
    // ==================================================================== compute waves
    const int wave_m = wave % F::WM, wave_c = wave / F::WM;   // GEMM1: 32 px x 32 ch per wave; GEMM2: 32 px x C/2 ch
    const int l31 = lane & 31, h = lane >> 5;
    v16i acc2[F::CT2];
#pragma unroll
    for (int c = 0; c < F::CT2; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[c][r] = 0;
    const int arow = wave_m * 32 + l31;                           // this lane's pixel row (both GEMMs, both epilogues)
    const int wrow1 = wave_c * 32 + cperm(l31);                   // GEMM1: W3 slice row
    const int lch = wave_c * 32 + h * 16;                         // slice-local first channel of this lane's 16 outputs
    DyNt dids = dynt_prepare(p.m_id_s, p.e_id_s), dq = dynt_prepare(p.mq, p.eq);
    constexpr bool SC0 = QK0 && !TIE;   // round 6: the QK0 instantiations also take the identity table in its shift-free form (launcher: ids0_form)
    DyS0 dis0 = dys0_prepare(p.m_id_s, p.e_id_s);
    asm volatile("" : "+v"(dids.add), "+v"(dq.add), "+v"(dis0.add));   // opaque rounding constants in VGPR pairs: the scalar m already takes the constant bus of the v_mad_i64_i32, an SGPR addend would cost a v_mov_b64 per requant
    const unsigned rowmask = (m0 + arow < p.M) ? 0xffffffffu : 0u;
    const int qhi2 = (p.q_hi & 0xffff) | (p.q_hi << 16);
    const int res_row = (m0 + arow < p.M) ? m0 + arow : m0;       // rows beyond M read a valid row and are never stored
    const char *res_base = (const char *)p.res_in + ((size_t)res_row * p.C3 + lch) * 2;
    unsigned oor = 0;

    auto load_res = [&](int j, v4i (&r)[2]) {   // this lane's 16 residual values of slice j (32 contiguous bytes)
        const v4i *rp = reinterpret_cast<const v4i *>(res_base + (size_t)j * 128);
        r[0] = rp[0];
        r[1] = rp[1];
    };
    auto store_slice = [&](int j) {   // new residual slice j: staging tile -> memory, whole 128-byte rows
        const char *src = rest + (j & 1) * F::RES_BYTES;
#pragma unroll
        for (int i = 0; i < F::BM * 8 / F::NTC; ++i) {
            const int idx = t + F::NTC * i, row = idx >> 3, jj = idx & 7;
            if (m0 + row < p.M)
                *reinterpret_cast<v4i *>((char *)p.res_out + ((size_t)(m0 + row) * p.C3 + j * 64) * 2 + ((jj ^ (row & 7)) << 4)) =
                    *reinterpret_cast<const v4i *>(src + idx * 16);
        }
    };
    auto gemm1 = [&](int j, v16i &acc1) {
        const char *w3s = ring + (j & 1) * F::WSLOT;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[r] = 0;
#pragma unroll
        for (int kc = 0; kc < F::KC; ++kc)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const v4i wf = *reinterpret_cast<const v4i *>(w3s + kc * 4096 + lds_off(wrow1, 2 * ks + h));
                const v4i af = *reinterpret_cast<const v4i *>(x2t + kc * (F::BM * 64) + lds_off(arow, 2 * ks + h));
                acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf, af, acc1, 0, 0, 0);
                if (F::KC > 2 && ks == 1 && (kc & 1) && kc + 1 < F::KC) __builtin_amdgcn_sched_barrier(0);   // bounds the fragments in flight (registers)
            }
    };
    // MFMAs [i0, i1) of GEMM2(j): i -> k-half i / CT2, channel tile i % CT2.  Fragments are fetched a whole channel group of
    // epilogue VALU before the MFMAs that consume them (g2_fetch ... VALU ... g2_mma), so that no MFMA waits for LDS.
    constexpr int NG = (2 * F::CT2 + 3) / 4;   // MFMAs per channel group of the epilogue (at most)
    auto g2_fetch = [&](int j, int i0, int i1, v4i (&wfr)[NG], v4i (&afr)[NG]) {
        const char *w1s = ring + (2 + (j & 1)) * F::WSLOT, *qs = qt + (j & 1) * F::Q_BYTES;
#pragma unroll
        for (int i = i0; i < i1; ++i) {
            const int ks = i / F::CT2, c = i % F::CT2;
            afr[i - i0] = *reinterpret_cast<const v4i *>(qs + lds_off(arow, 2 * ks + h));
            wfr[i - i0] = *reinterpret_cast<const v4i *>(w1s + lds_off(wave_c * (F::CT2 * 32) + c * 32 + cperm(l31), 2 * ks + h));
        }
    };
    auto g2_mma = [&](int i0, int i1, const v4i (&wfr)[NG], const v4i (&afr)[NG]) {
#pragma unroll
        for (int i = i0; i < i1; ++i) {
            const int c = i % F::CT2;
            acc2[c] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wfr[i - i0], afr[i - i0], acc2[c], 0, 0, 0);
        }
    };
    auto gemm2_part = [&](int j, int i0, int i1) {   // un-pipelined form (the tail iteration)
        const char *w1s = ring + (2 + (j & 1)) * F::WSLOT, *qs = qt + (j & 1) * F::Q_BYTES;
#pragma unroll
        for (int i = i0; i < i1; ++i) {
            const int ks = i / F::CT2, c = i % F::CT2;
            const v4i af = *reinterpret_cast<const v4i *>(qs + lds_off(arow, 2 * ks + h));
            const v4i wf = *reinterpret_cast<const v4i *>(w1s + lds_off(wave_c * (F::CT2 * 32) + c * 32 + cperm(l31), 2 * ks + h));
            acc2[c] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf, af, acc2[c], 0, 0, 0);
        }
    };
    // epilogue 1 of slice j (residual add, ReLU, next QuantAct) with the MFMAs of GEMM2(j - 1) spread over its four channel groups
    auto epilogue1 = [&](int j, const v16i &acc1, v4i (&rin)[2], bool with_gemm2) {
        char *rb = rest + (j & 1) * F::RES_BYTES + arow * 128;
        const char *ctb = ct3 + (j & 1) * 1024;
        const v4i *rnext = reinterpret_cast<const v4i *>(res_base + (size_t)(j + 1 < n ? j + 1 : j) * 128);
        int rpack[8], qpack[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int i0 = g * (2 * F::CT2) / 4, i1 = (g + 1) * (2 * F::CT2) / 4;
            v4i wfr[NG], afr[NG];
            if (with_gemm2) {
                g2_fetch(j - 1, i0, i1, wfr, afr);
                if (F::SG == 4 || (F::SG == 2 && !(g & 1))) __builtin_amdgcn_sched_barrier(0);
            }
            const unsigned w0 = (unsigned)rin[g >> 1][(g & 1) * 2], w1 = (unsigned)rin[g >> 1][(g & 1) * 2 + 1];
            const int idin[4] = {(int)(w0 & 0xffffu), (int)(w0 >> 16), (int)(w1 & 0xffffu), (int)(w1 >> 16)};
            if (g & 1) rin[g >> 1] = rnext[g >> 1];   // this half is consumed: fetch the same half of slice j + 1 (the last slice re-reads itself)
            int o[4], qv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const DyNt dm = e2_ctab_entry<K0>(ctb, lch + 4 * g + k);
                const int a = dyadic_mode<MODE_C>(acc1[4 * g + k], dm);
                const int b = SC0 ? dyadic_s0(idin[k], dis0) : dyadic_mode<MODE>(idin[k], dids);
                o[k] = max(a + b, 0);                                  // no clamp: quant_utils.py:456
                qv[k] = dyadic_mode<MODE_Q>(o[k], dq);                 // o >= 0, m >= 0: q >= 0; clamped from above in the pack
            }
            oor |= ((unsigned)(o[0] | o[1]) | (unsigned)(o[2] | o[3])) & rowmask;
            rpack[2 * g] = pack2_u16_sat(o[0], o[1]);
            rpack[2 * g + 1] = pack2_u16_sat(o[2], o[3]);
            qpack[g] = pack4_min(qv[0], qv[1], qv[2], qv[3], qhi2);
            if (with_gemm2) {
                if (F::SG == 4 || (F::SG == 2 && (g & 1))) __builtin_amdgcn_sched_barrier(0);
                g2_mma(i0, i1, wfr, afr);
                if (F::SG == 4 || (F::SG == 2 && (g & 1))) __builtin_amdgcn_sched_barrier(0);
            }
        }
        const v4i ra = {rpack[0], rpack[1], rpack[2], rpack[3]}, rc = {rpack[4], rpack[5], rpack[6], rpack[7]};
        *reinterpret_cast<v4i *>(rb + (((lch >> 3) ^ (arow & 7)) << 4)) = ra;
        *reinterpret_cast<v4i *>(rb + ((((lch >> 3) + 1) ^ (arow & 7)) << 4)) = rc;
        const v4i qw = {qpack[0], qpack[1], qpack[2], qpack[3]};
        *reinterpret_cast<v4i *>(qt + (j & 1) * F::Q_BYTES + lds_off(arow, lch >> 4)) = qw;
    };

    v4i rin[2];   // residual slice of the iteration in flight; each half is re-loaded for the next slice as soon as it is consumed
    v16i acc1;
    load_res(0, rin);
    // ---- iteration 0: no GEMM2 yet
    __builtin_amdgcn_s_barrier();   // barrier(0): x2, W3(0), ctab3(0) have landed
    gemm1(0, acc1);
    epilogue1(0, acc1, rin, false);
    // ---- iterations 1 .. n-1
    E2_STAMP(0)
    for (int j = 1; j < n; ++j) {
        __builtin_amdgcn_s_barrier();   // barrier(j): W3(j), ctab3(j), W1(j-1) have landed; q(j-1) and the staging tile of slice j-1 are complete
        E2_STAMP(1)
        store_slice(j - 1);
        E2_STAMP(2)
        if (HAWQ_DBG_BIT(p.dbg, 256) && wave_c == 1) {
            // TIMING PROBE (probe build only, results wrong on purpose): the two compute waves of a SIMD are waves w and w + WM, i.e. the two
            // channel halves.  Here the upper half runs its epilogue (on the accumulators of the PREVIOUS slice) BEFORE its GEMM1, so that on
            // every SIMD one wave is in its MFMA phase while the other is in its VALU phase - what a phase-shifted schedule of this kernel
            // could gain at best (profiles/r06_epilogue_census.md: the parts ADD because both waves pass the slice barrier together)
            epilogue1(j, acc1, rin, true);
            gemm1(j, acc1);
        } else {
            gemm1(j, acc1);
            E2_STAMP(3)
            epilogue1(j, acc1, rin, true);
        }
        E2_STAMP(4)
    }
    // ---- iteration n: the last slice's stores and GEMM2
    __builtin_amdgcn_s_barrier();   // barrier(n)
    store_slice(n - 1);
    gemm2_part(n - 1, 0, 2 * F::CT2);
    E2_STAMP(5)
    if (prof && blockIdx.x == 8 && t == 0)
        for (int k = 0; k < 8; ++k) p.dbgbuf[k] = ph[k];
#undef E2_STAMP
    if ((oor >> 16) != 0) atomicOr(p.flags, 1);
    // ---------------------------------------------------------------- epilogue 2: the reduce conv's QuantAct
    __syncthreads();   // all GEMM2 fragment reads done: the ring becomes the output staging tile [BM][C B]
    {
        constexpr int CPR = F::C / 16;   // 16-byte chunks per output row
        char *yt = ring;
        if (p.y_nib) {
            // hawq4 output: 16 channels of a lane = 8 bytes, byte k of an 8-channel group = c_k | c_{k+4} << 4 (include/hawq_mi355.h)
            constexpr int CPN = F::C / 32;
#pragma unroll
            for (int c = 0; c < F::CT2; ++c) {
                const int ch0 = wave_c * (F::CT2 * 32) + c * 32 + h * 16;
                int qv[16];
#pragma unroll
                for (int k = 0; k < 16; ++k)
                    qv[k] = med3i(dyadic_mode<MODE_C>(acc2[c][k], e2_ctab_entry<K0>((const char *)p.ctab1, ch0 + k)), p.y_lo, p.y_hi);
                const v2i ww = {(int)pack8_u4(&qv[0]), (int)pack8_u4(&qv[8])};
                *reinterpret_cast<v2i *>(yt + (arow * CPN + ((ch0 >> 5) ^ (arow & (CPN - 1)))) * 16 + ((ch0 >> 4) & 1) * 8) = ww;
            }
            __syncthreads();
            if (p.y_planar) {   // planes [C / 32][M][16 B]
                for (int idx = t; idx < F::BM * CPN; idx += F::NTC) {
                    const int ch = idx / F::BM, row = idx % F::BM;
                    if (m0 + row < p.M)
                        *reinterpret_cast<v4i *>((char *)p.y + ((size_t)ch * p.M + (m0 + row)) * 16) =
                            *reinterpret_cast<const v4i *>(yt + (row * CPN + (ch ^ (row & (CPN - 1)))) * 16);
                }
            } else {
                for (int idx = t; idx < F::BM * CPN; idx += F::NTC) {
                    const int row = idx / CPN, jj = idx % CPN;
                    if (m0 + row < p.M)
                        *reinterpret_cast<v4i *>((char *)p.y + (size_t)(m0 + row) * (F::C / 2) + ((jj ^ (row & (CPN - 1))) << 4)) =
                            *reinterpret_cast<const v4i *>(yt + idx * 16);
                }
            }
            return;
        }
#pragma unroll
        for (int c = 0; c < F::CT2; ++c) {
            const int ch0 = wave_c * (F::CT2 * 32) + c * 32 + h * 16;
            int w[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                int qv[4];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    qv[k] = med3i(dyadic_mode<MODE_C>(acc2[c][4 * g + k], e2_ctab_entry<K0>((const char *)p.ctab1, ch0 + 4 * g + k)), p.y_lo, p.y_hi);
                w[g] = pack4_fast(qv[0], qv[1], qv[2], qv[3]);
            }
            const v4i ww = {w[0], w[1], w[2], w[3]};
            *reinterpret_cast<v4i *>(yt + (arow * CPR + ((ch0 >> 4) ^ (arow & (CPR - 1)))) * 16) = ww;
        }
        __syncthreads();
        if (p.y_planar) {   // channel-group planes [C / 16][M][16 B] (hawq_conv_args.out_planar)
#pragma unroll
            for (int i = 0; i < F::BM * CPR / F::NTC; ++i) {
                const int idx = t + F::NTC * i, ch = idx / F::BM, row = idx % F::BM;
                if (m0 + row < p.M)
                    *reinterpret_cast<v4i *>((char *)p.y + ((size_t)ch * p.M + (m0 + row)) * 16) =
                        *reinterpret_cast<const v4i *>(yt + (row * CPR + (ch ^ (row & (CPR - 1)))) * 16);
            }
        } else {
#pragma unroll
            for (int i = 0; i < F::BM * CPR / F::NTC; ++i) {
                const int idx = t + F::NTC * i, row = idx / CPR, jj = idx % CPR;
                if (m0 + row < p.M)
                    *reinterpret_cast<v4i *>((char *)p.y + (size_t)(m0 + row) * F::C + ((jj ^ (row & (CPR - 1))) << 4)) =
                        *reinterpret_cast<const v4i *>(yt + idx * 16);
            }
        }
    }
}

// (C = 256 with 8 producers would need 16 waves at <= 128 registers: the 64 GEMM2 accumulators + 16 of GEMM1 leave too few - 184 spills)
// scheduling groups (SG): pinning GEMM2's MFMAs after every channel group, after every second one or leaving them to the compiler
// measured the same (34.8 / 34.8 / 36.4 us at batch 64: the kernel is VALU-bound, profiles/r04_pair_kernels.md); SG = 2 is the
// form of C = 256 that allocates without a spill
using P256B = E2Cfg<256, 4, 4, 3, 2>;   // stage 3: 128 pixels, 8 compute + 4 producer waves, 146 KiB: one workgroup per CU, 162-168 registers
using P128A = E2Cfg<128, 4, 8, 4>;   // stage 2: 128 pixels, 8 + 8 waves, 98 KiB
using P128B = E2Cfg<128, 2, 4, 4>;   //          64 pixels, 4 + 4 waves, 66 KiB: two workgroups per CU
using P64A = E2Cfg<64, 4, 4, 3>;     // stage 1: 128 pixels, 8 + 4 waves, 74 KiB
constexpr int NUM_E2 = 4;

typedef void (*E2Fn)(const E2P);
struct E2Info { E2Fn fn[5]; int c, bm, nt, lds; };   // fn: {general, exact-tie, per-channel k all zero, + next-QuantAct k zero, only the latter}
#define E2_ENTRY(F) {{expand_reduce_pipelined_kernel<F, false>, expand_reduce_pipelined_kernel<F, true>, expand_reduce_pipelined_kernel<F, false, true>, \
                      expand_reduce_pipelined_kernel<F, false, true, true>, expand_reduce_pipelined_kernel<F, false, false, true>}, F::C, F::BM, F::NT, F::LDS_BYTES}
const E2Info kE2[NUM_E2] = {E2_ENTRY(P256B), E2_ENTRY(P128A), E2_ENTRY(P128B), E2_ENTRY(P64A)};

bool e2_conv_ok(const hawq_conv_args &a) {
    return (a.in_pitch == 0 || a.in_pitch == a.Cin) && (a.out_pitch == 0 || a.out_pitch == a.Cout) && a.KH == 1 && a.KW == 1 && a.stride == 1 && a.pad == 0 && a.in_bits == 8 && a.w_bits == 8 && a.fast_tables != 0 && !a.in2 && !a.in_planar;
}

// index into kE2 of the nth (1-based) variant that takes this pair, or -1
int e2_variant(const hawq_expand_reduce_args *a, int nth) {
    const hawq_conv_args &e = a->expand, &r = a->reduce;
    if (!r.wgt || !e2_conv_ok(e) || !e2_conv_ok(r)) return -1;
    if (e.epilogue != HAWQ_EPI_RESIDUAL || r.epilogue != HAWQ_EPI_REQUANT) return -1;
    if (!e.res_in || e.res_in_bits != 16 || !e.res_out || e.res_out_bits != 16 || !e.flags || !e.ctab || !r.ctab || !r.out_q) return -1;
    if ((r.out_bits != 8 && r.out_bits != 4) || e.out_bits != 8) return -1;
    if (r.out_bits == 4 && (r.q_lo < 0 || r.q_hi > 15)) return -1;   // hawq4 stores unsigned nibbles
    if (r.Cin != e.Cout || r.Cout != e.Cin || r.N != e.N || r.H != e.H || r.W != e.W || e.Cout % 128) return -1;
    int k = 0;
    for (int i = 0; i < NUM_E2; ++i)
        if (kE2[i].c == e.Cin && ++k == nth) return i;
    return -1;
}

}  // namespace

// number of pipelined variants that take this pair (numbered after the variants of fused_er.hip and fused_wp.hip)
int er2_num_variants(const hawq_expand_reduce_args *a) {
    int n = 0;
    while (e2_variant(a, n + 1) >= 0) ++n;
    return n;
}

int er2_launch(const hawq_expand_reduce_args *a, int nth, void *stream) {
    const int v = e2_variant(a, nth);
    HAWQ_REQUIRE(v >= 0, "hawq_conv_expand_reduce: no pipelined variant %d for this launch", nth);
    const hawq_conv_args &e = a->expand, &r = a->reduce;
    auto e_fast = [](int ek) { return (ek & 0xff) >= 33 && (ek & 0xff) <= 62; };
    HAWQ_REQUIRE(e.mq >= 0 && e_fast(e.eq) && e.m_id_scalar >= 0 && e_fast(e.e_id_scalar), "hawq_conv_expand_reduce: scalar tables outside the fast contract");
    HAWQ_REQUIRE(e.q_lo <= 0, "hawq_conv_expand_reduce: the block-input QuantAct clamp must admit 0");
    HAWQ_REQUIRE(e.q_hi >= 0 && e.q_hi <= 32767, "hawq_conv_expand_reduce: q_hi outside [0, 32767]");
    E2P p;
    p.x2 = (const uint8_t *)e.in, p.w3 = (const uint8_t *)e.wgt, p.w1 = (const uint8_t *)r.wgt;
    p.ctab3 = e.ctab, p.ctab1 = r.ctab;
    p.res_in = (const uint16_t *)e.res_in, p.res_out = (uint16_t *)e.res_out;
    p.y = (uint8_t *)r.out_q;
    const long long M = (long long)e.N * e.H * e.W;
    HAWQ_REQUIRE(M > 0 && M < (1ll << 30), "hawq_conv_expand_reduce: bad problem size");
    p.M = (int)M, p.C3 = e.Cout;
    p.m_id_s = e.m_id_scalar, p.e_id_s = e.e_id_scalar, p.mq = e.mq, p.eq = e.eq, p.q_hi = e.q_hi;
    p.y_lo = r.relu && r.q_lo < 0 ? 0 : r.q_lo, p.y_hi = r.q_hi;
    p.y_planar = r.out_planar;
    p.y_nib = r.out_bits == 4;
    p.flags = e.flags;
    p.dbgbuf = nullptr;
    p.dbg = 0;
#ifdef HAWQ_ABLATE
    static const int dbg_env = HAWQ_DBG_ENV();
    p.dbg = dbg_env;
    static long long *dbg_dev = nullptr;
    if ((dbg_env & 128) && !dbg_dev) (void)hipMalloc(&dbg_dev, 16 * sizeof(long long));
    if (dbg_env & 128) p.dbgbuf = dbg_dev;
#endif
    const E2Info &ei = kE2[v];
    static const bool attrs = [] {
        bool good = true;
        for (const E2Info &k : kE2)
            for (int i = 0; i < 5; ++i)
                good &= hipFuncSetAttribute((const void *)k.fn[i], hipFuncAttributeMaxDynamicSharedMemorySize, k.lds) == hipSuccess;
        return good;
    }();
    HAWQ_REQUIRE(attrs, "hawq_conv_expand_reduce: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
    const bool tie = ((e.fast_tables | r.fast_tables) & 4) != 0, ck0 = (e.fast_tables & 8) && (r.fast_tables & 8);
    const bool qk0 = (e.eq >> 8) == 0 && ids0_form(e.e_id_scalar);   // the QK0 instantiations: next-QuantAct table without pre-shift AND identity table in the shift-free form
    hipLaunchKernelGGL(ei.fn[tie ? 1 : (ck0 ? (qk0 ? 3 : 2) : (qk0 ? 4 : 0))], dim3((p.M + ei.bm - 1) / ei.bm), dim3(ei.nt), ei.lds, (hipStream_t)stream, p);
    HAWQ_CHECK_HIP(hipGetLastError());
    if (p.dbgbuf) {   // probe builds only (synchronises!)
        long long hb[12];
        (void)hipStreamSynchronize((hipStream_t)stream);
        (void)hipMemcpy(hb, p.dbgbuf, sizeof(hb), hipMemcpyDeviceToHost);
        fprintf(stderr, "[pipelined C=%d bm=%d M=%d C3=%d] cycles of compute wave 0 / workgroup 8: prologue + iteration 0 %lld | barrier wait %lld | stores %lld | GEMM1 %lld | "
                        "epilogue 1 + GEMM2 %lld | tail %lld || producer wave 0: prologue %lld | barrier wait %lld | issue %lld | vmcnt wait %lld\n",
                ei.c, ei.bm, p.M, p.C3, hb[0], hb[1], hb[2], hb[3], hb[4], hb[5], hb[8], hb[9], hb[10], hb[11]);
    }
    return 0;
}
