// One launch per MobileNetV2 linear-bottleneck unit (round 4).
//
// Reference path: Q_LinearBottleneck.forward (q_mobilenetv2.py:59-93): quant_act -> conv1 1x1 (+ReLU6 + quant_act1) -> conv2 depthwise
// 3x3 (+ReLU6 + quant_act2) -> conv3 1x1 -> quant_act_int32 (identity branch or not); the next unit's block-input QuantAct rides along
// as it does on hawq_conv2d's RESIDUAL epilogue.
//
// Why: as three launches the unit writes and re-reads its expanded ("hidden") tensor twice - 6x the width of what enters and leaves the
// unit.  On the 112 x 112 and 56 x 56 maps those two tensors ARE the unit's HBM traffic (profiles/r04_e_mbv2_perop_compact_gfast.txt:
// units 1-4 are 45 % of the network's launch time).  Here a workgroup owns an 8 x 16 tile of output pixels of one image and walks the
// hidden channels 32 at a time:
//   GEMM1   window pixels (the tile's (7 S + 3) x (15 S + 3) halo window, 32 per MFMA block) x K = Cin (<= 64, one or two
//           v_mfma_i32_32x32x32_i8) x 32 hidden channels; lane = pixel, 16 registers = 16 consecutive channels (cperm), requantised
//           with the fused per-channel constants (3 instructions) and written to LDS as int8 [window pixel][32]; pixels outside the
//           image become the depthwise conv's zero padding.  The 1x1 conv is recomputed on the halo (x 1.4 at stride 1) - it is K <= 64.
//   DW      thread = (4 channels, one output column, 4 output rows): the rows' taps come out of LDS once (6 / 9 window rows of 3
//           dwords), one v_dot4_i32_i8 per MAC against byte-masked weight dwords (as hawq_depthwise3x3_requant), requant, int8
//           [output pixel][32] to LDS.  A wave's lanes cover 8 adjacent pixels x 32 channels: contiguous LDS rows.
//   GEMM2   the 128 output pixels (one 32-pixel block per wave) x K = these 32 hidden channels x Cout (<= 64), accumulated in
//           registers over all slices.
//   closing the direct RESIDUAL arithmetic of hawq_conv2d (conv_igemm.hip, ConvP.gfast): per-channel requant + identity requant, no
//           ReLU, 16-bit clamp without an identity, the next QuantAct; int32 carrier and int8 q leave as 64 / 16 bytes per lane.
// Two workgroup barriers per slice; weights and table slices are staged through LDS one slice ahead (a few KiB per slice, from L2).
// HBM bytes per unit = its input + its outputs (+ the identity), e.g. unit 2 (16 -> 96 -> 24, 112^2 -> 56^2, batch 128): 26 MB in,
// 10 MB out against 360 MB through the three launches.
#include "common.h"
#include <type_traits>

namespace {

struct LbP {
    const int8_t *x;
    int N, H, W, Ho, Wo, in_pitch;
    const int8_t *w1;
    int w1_pitch;
    const int32_t *ct1;
    int lo1, hi1;
    const int8_t *w9;
    int w9_pitch;
    const int32_t *ct2;
    int lo2, hi2;
    const int8_t *w3;
    int w3_pitch;
    const int32_t *ct3;
    int nsl;
    const int32_t *res_in;
    int m_id, e_id;
    int32_t *res_out;
    int8_t *out_q;
    int mq, eq, q_lo, q_hi, clamp16, out_pitch;
    int tiles_x, tiles_y;
};

__device__ __forceinline__ v4i ldg4(const void *p) { return *reinterpret_cast<const v4i *>(p); }

__device__ __forceinline__ DyNt entry(const v4i t4) {
    DyNt d;
    d.m = t4.x, d.s = t4.y & 31, d.k = t4.y >> 8;
    d.add = (long long)(((unsigned long long)(unsigned)t4.w << 32) | (unsigned)t4.z);
    return d;
}

template <bool TIE>
__device__ __forceinline__ int requant(int v, const DyNt &d) {
    return TIE ? dyadic_tie(v, d) : dyadic_nt(v, d);
}

constexpr int LB_TH = 8, LB_TW = 16, LB_NT = 256;

// S: depthwise stride; KS1: 32-byte K steps of the expand conv (Cin <= 32 KS1); CT2: 32-channel blocks of the projection's output.
// Everything a slice needs besides activations - its 32 + 32 table rows, the W1 rows, the 9 depthwise tap rows and the W3 columns
// (3.3 .. 5.3 KiB) - is staged through LDS one slice ahead: each thread fetches at most two 16-byte items at the top of slice j
// and stores them between the two barriers of that slice, so that no global-memory latency sits on the per-slice dependency chain
// (the first version read them where they were used: ~2.8 us per slice on the 14 x 14 units, 12 slices each).
// register budget = waves per SIMD the compiler must leave room for (stride 1 / stride 2 instantiations); measured on the ten fused
// units of MobileNetV2 w1 at batch 128: unconstrained 981 us, 4 / 3 973 us (spills), 3 / 2 924 us (profiles/r04_e_mbv2_unit_occupancy.txt)
#ifndef LB_OCC1
#define LB_OCC1 3
#define LB_OCC2 2
#endif
#if LB_OCC1
#define LB_OCC_ATTR __attribute__((amdgpu_waves_per_eu(S == 1 ? LB_OCC1 : LB_OCC2)))
#else
#define LB_OCC_ATTR
#endif
template <int S, int KS1, int CT2, bool TIE>
__global__ __launch_bounds__(LB_NT) LB_OCC_ATTR void linear_bottleneck_kernel(const LbP p) {
    constexpr int WH = (LB_TH - 1) * S + 3, WW = (LB_TW - 1) * S + 3, WP = WH * WW, NB1 = (WP + 31) / 32;
    constexpr int NR = 3 * S + 3;   // window rows under 4 vertically adjacent outputs
    constexpr int N_W1 = 64 * KS1, N_W9 = 18, N_W3 = 64 * CT2, N_ITEMS = 64 + N_W1 + N_W9 + N_W3;   // 16-byte items per slice
    constexpr int OFF_CT1 = 0, OFF_CT2 = 512, OFF_W1 = 1024, OFF_W9 = OFF_W1 + 1024 * KS1, OFF_W3 = OFF_W9 + 288, BUF = OFF_W3 + 1024 * CT2;
    static_assert(N_ITEMS <= 2 * LB_NT, "two items per thread");
    __shared__ __attribute__((aligned(16))) char xs[NB1 * 32 * 32 * KS1];   // block input on the halo window [window pixel][K]
    __shared__ __attribute__((aligned(16))) char hid[NB1 * 32 * 32];        // quant_act1 output [window pixel][32 channels of the slice]
    __shared__ __attribute__((aligned(16))) char dwo[LB_TH * LB_TW * 32];
    __shared__ __attribute__((aligned(16))) char stg[2][BUF];   // [slice parity]
    __shared__ v4i ct3s[32 * CT2];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, h = lane >> 5;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    int bid = blockIdx.x;
    const int tx = bid % p.tiles_x;
    bid /= p.tiles_x;
    const int ty = bid % p.tiles_y, n = bid / p.tiles_y;
    const int oy0 = ty * LB_TH, ox0 = tx * LB_TW, iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;
    const int8_t *img = p.x + (size_t)n * p.H * p.W * p.in_pitch;

    // The expand conv's B operand (lane = pixel, half h = bytes 16 h .. 16 h + 15 of each 32-byte K step): the tile's halo window goes
    // to LDS once (every slice's GEMM1 reads it again).  Requant work per wave: NOWN whole window blocks (wave, wave + 4, ..) and a
    // quarter of the channels of the last two blocks (registers 4 wave .. 4 wave + 3 of both lane halves) - 6 / 18 blocks do not
    // divide by 4 waves and the MFMA itself is free; vmask = which of this wave's pixels lie inside the image.
    constexpr int NOWN = (NB1 - 2) / 4, NBW = NOWN + 2;
    static_assert((NB1 - 2) % 4 == 0, "window blocks = 4 NOWN + 2");
    unsigned vmask = 0;
#pragma unroll
    for (int i = 0; i < NBW; ++i) {
        const int blk = i < NOWN ? wave + 4 * i : NB1 - 2 + (i - NOWN), wp = blk * 32 + l31;
        const int wy = wp / WW, wx = wp - wy * WW, iy = iy0 + wy, ix = ix0 + wx;
        const bool ok = wp < WP && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        vmask |= (ok ? 1u : 0u) << i;
    }
#pragma unroll
    for (int i = 0; i < (NB1 + 3) / 4; ++i) {
        const int blk = wave + 4 * i, wp = blk * 32 + l31;
        if (blk < NB1) {
            const int wy = wp / WW, wx = wp - wy * WW, iy = iy0 + wy, ix = ix0 + wx;
            const bool ok = wp < WP && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const int8_t *src = img + (size_t)(ok ? iy * p.W + ix : 0) * p.in_pitch + h * 16;
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) *reinterpret_cast<v4i *>(xs + wp * (32 * KS1) + ks * 32 + h * 16) = ldg4(src + ks * 32);
        }
    }
    // staging roles of this thread: item -> (source of slice 0, bytes from one slice to the next, LDS offset inside a stage)
    const char *ssrc[2];
    int sstep[2], sdst[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int it = t + r * LB_NT;
        ssrc[r] = nullptr, sstep[r] = 0, sdst[r] = 0;
        if (it < 32) {
            ssrc[r] = (const char *)p.ct1 + it * 16, sstep[r] = 512, sdst[r] = OFF_CT1 + it * 16;
        } else if (it < 64) {
            ssrc[r] = (const char *)p.ct2 + (it - 32) * 16, sstep[r] = 512, sdst[r] = OFF_CT2 + (it - 32) * 16;
        } else if (it < 64 + N_W1) {
            const int k = it - 64, row = k / (2 * KS1), chunk = k % (2 * KS1);
            ssrc[r] = (const char *)p.w1 + (size_t)row * p.w1_pitch + chunk * 16, sstep[r] = 32 * p.w1_pitch, sdst[r] = OFF_W1 + row * (32 * KS1) + chunk * 16;
        } else if (it < 64 + N_W1 + N_W9) {
            const int k = it - 64 - N_W1;
            ssrc[r] = (const char *)p.w9 + (size_t)(k >> 1) * p.w9_pitch + (k & 1) * 16, sstep[r] = 32, sdst[r] = OFF_W9 + k * 16;
        } else if (it < N_ITEMS) {
            const int k = it - 64 - N_W1 - N_W9;
            ssrc[r] = (const char *)p.w3 + (size_t)(k >> 1) * p.w3_pitch + (k & 1) * 16, sstep[r] = 32, sdst[r] = OFF_W3 + k * 16;
        }
    }
    v4i sreg[2];
#pragma unroll
    for (int r = 0; r < 2; ++r)
        if (ssrc[r]) *reinterpret_cast<v4i *>(&stg[0][sdst[r]]) = ldg4(ssrc[r]);
    if (t < 32 * CT2) ct3s[t] = ldg4(p.ct3 + t * 4);
    v16i acc2[CT2];
#pragma unroll
    for (int c = 0; c < CT2; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[c][r] = 0;
    // depthwise thread mapping: 4 channels cg, output column dx, output rows dy0 .. dy0 + 3
    const int cg = t & 7, dx = (t >> 3) & 15, dy0 = (t >> 7) * 4;
    // closing: this lane's output pixel
    const int pl = wave * 32 + l31, gy = oy0 + (pl >> 4), gx = ox0 + (pl & 15);
    const bool out_ok = gy < p.Ho && gx < p.Wo;
    const size_t pix = ((size_t)n * p.Ho + gy) * p.Wo + gx;
    __syncthreads();

    for (int j = 0; j < p.nsl; ++j) {
        const char *sb = stg[j & 1];
        const bool more = j + 1 < p.nsl;
        if (more) {   // next slice's tables and weights: fetched now, stored to LDS between this slice's barriers
#pragma unroll
            for (int r = 0; r < 2; ++r)
                if (ssrc[r]) sreg[r] = ldg4(ssrc[r] + (size_t)(j + 1) * sstep[r]);
        }
        // ---------------------------------------------------------------- GEMM1 + quant_act1 -> hid
        {
            v4i wf[KS1];
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) wf[ks] = *reinterpret_cast<const v4i *>(sb + OFF_W1 + cperm(l31) * (32 * KS1) + ks * 32 + h * 16);
            const v4i *ct1 = reinterpret_cast<const v4i *>(sb + OFF_CT1) + h * 16;
            // passes of at most 3 accumulator tiles: OPP whole blocks + SPP shared blocks each
            constexpr int NPASS = NOWN >= 2 ? 2 : 1, OPP = NOWN / NPASS, SPP = 2 / NPASS, NA = OPP + SPP;
            static_assert(NOWN % NPASS == 0, "own blocks divide into the passes");
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
                v16i a[NA];
                int blk[NA], vbit[NA];
#pragma unroll
                for (int i = 0; i < NA; ++i) {
                    blk[i] = i < OPP ? wave + 4 * (ps * OPP + i) : NB1 - 2 + ps * SPP + (i - OPP);
                    vbit[i] = i < OPP ? ps * OPP + i : NOWN + ps * SPP + (i - OPP);
#pragma unroll
                    for (int r = 0; r < 16; ++r) a[i][r] = 0;
#pragma unroll
                    for (int ks = 0; ks < KS1; ++ks) {
                        const v4i xf = *reinterpret_cast<const v4i *>(xs + (blk[i] * 32 + l31) * (32 * KS1) + ks * 32 + h * 16);
                        a[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[ks], xf, a[i], 0, 0, 0);
                    }
                }
                // channel-outer: a channel's constants are fetched and unpacked once for all tiles of the pass.  The accumulator
                // tiles are only READ here (conditional in-place updates made hipcc copy them between AGPRs and VGPRs around every branch)
                int qo[OPP][4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    v4i e4[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) e4[k] = ct1[4 * g + k];
                    const bool mine = g == wave_s;   // this wave's quarter of the shared blocks (scalar condition)
                    int qg[OPP][4], qs[SPP][4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        DyNt d = entry(e4[k]);
#pragma unroll
                        for (int i = 0; i < OPP; ++i) qg[i][k] = med3i(requant<TIE>(a[i][4 * g + k], d), p.lo1, p.hi1);
                        if (mine) {
#pragma unroll
                            for (int i = 0; i < SPP; ++i) qs[i][k] = med3i(requant<TIE>(a[OPP + i][4 * g + k], d), p.lo1, p.hi1);
                        }
                    }
#pragma unroll
                    for (int i = 0; i < OPP; ++i) qo[i][g] = pack4_fast(qg[i][0], qg[i][1], qg[i][2], qg[i][3]);
                    if (mine) {
#pragma unroll
                        for (int i = 0; i < SPP; ++i) {
                            const int w = ((vmask >> vbit[OPP + i]) & 1) ? pack4_fast(qs[i][0], qs[i][1], qs[i][2], qs[i][3]) : 0;
                            *reinterpret_cast<int *>(hid + (blk[OPP + i] * 32 + l31) * 32 + h * 16 + 4 * g) = w;
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < OPP; ++i) {
                    const bool ok = (vmask >> vbit[i]) & 1;   // outside the image: the depthwise conv's zero padding
                    const v4i w = ok ? v4i{qo[i][0], qo[i][1], qo[i][2], qo[i][3]} : v4i{0, 0, 0, 0};
                    *reinterpret_cast<v4i *>(hid + (blk[i] * 32 + l31) * 32 + h * 16) = w;
                }
            }
        }
        __syncthreads();   // B1: hid complete; every wave is past GEMM2 of the previous slice
        // ---------------------------------------------------------------- depthwise 3x3 + quant_act2 -> dwo
        {
            int wm[9][4];
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) {
                const int ww = *reinterpret_cast<const int *>(sb + OFF_W9 + tp * 32 + cg * 4);
#pragma unroll
                for (int k = 0; k < 4; ++k) wm[tp][k] = ww & (0xff << (8 * k));
            }
            int acc[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[i][k] = 0;
            const char *base = hid + ((dy0 * S) * WW + dx * S) * 32 + cg * 4;
#pragma unroll
            for (int rr = 0; rr < NR; ++rr) {
                int r[3];
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) r[kw] = *reinterpret_cast<const int *>(base + (rr * WW + kw) * 32);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int kh = rr - i * S;   // compile-time after unrolling
                    if (kh >= 0 && kh < 3) {
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                            for (int k = 0; k < 4; ++k) acc[i][k] = __builtin_amdgcn_sdot4(r[kw], wm[kh * 3 + kw][k], acc[i][k], false);
                    }
                }
            }
            DyNt d[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                d[k] = entry(reinterpret_cast<const v4i *>(sb + OFF_CT2)[cg * 4 + k]);
                asm volatile("" : "+v"(d[k].add));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int qv[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) qv[k] = med3i(requant<TIE>(acc[i][k], d[k]), p.lo2, p.hi2);
                *reinterpret_cast<int *>(dwo + ((dy0 + i) * LB_TW + dx) * 32 + cg * 4) = pack4_fast(qv[0], qv[1], qv[2], qv[3]);
            }
        }
        if (more) {   // the other stage's last readers (GEMM2 of slice j - 1) are behind B1
#pragma unroll
            for (int r = 0; r < 2; ++r)
                if (ssrc[r]) *reinterpret_cast<v4i *>(&stg[(j & 1) ^ 1][sdst[r]]) = sreg[r];
        }
        __syncthreads();   // B2: dwo complete, hid free, next stage complete
        // ---------------------------------------------------------------- GEMM2 partial sum over this slice
        {
            const v4i af = *reinterpret_cast<const v4i *>(dwo + (wave * 32 + l31) * 32 + h * 16);
#pragma unroll
            for (int c = 0; c < CT2; ++c) {
                const v4i wf = *reinterpret_cast<const v4i *>(sb + OFF_W3 + (c * 32 + cperm(l31)) * 32 + h * 16);
                acc2[c] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf, af, acc2[c], 0, 0, 0);
            }
        }
    }

    // -------------------------------------------------------------------- closing: quant_act_int32 (+ identity), next QuantAct
    if (!out_ok) return;
    DyNt dids = dynt_prepare(p.m_id, p.e_id), dq = dynt_prepare(p.mq, p.eq);
    asm volatile("" : "+v"(dids.add), "+v"(dq.add));
    const int clo = p.clamp16 ? -32768 : (int)0x80000000, chi = p.clamp16 ? 32767 : 0x7fffffff;
    auto close = [&](auto with_identity) {
#pragma unroll
        for (int c = 0; c < CT2; ++c) {
            const int ch = c * 32 + h * 16;
            if (ch >= p.out_pitch) continue;
            const size_t elem = pix * p.out_pitch + ch;
            int qw[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                int o[4], qv[4];
                v4i rin = {0, 0, 0, 0};
                if (decltype(with_identity)::value) rin = ldg4(p.res_in + elem + 4 * g);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    DyNt d = entry(ct3s[ch + 4 * g + k]);
                    asm volatile("" : "+v"(d.add));
                    int ov = requant<TIE>(acc2[c][4 * g + k], d);
                    if (decltype(with_identity)::value) ov += requant<TIE>(rin[k], dids);
                    ov = med3i(ov, clo, chi);
                    o[k] = ov;
                    qv[k] = med3i(requant<TIE>(ov, dq), p.q_lo, p.q_hi);
                }
                if (p.res_out) *reinterpret_cast<v4i *>(p.res_out + elem + 4 * g) = v4i{o[0], o[1], o[2], o[3]};
                qw[g] = pack4_fast(qv[0], qv[1], qv[2], qv[3]);
            }
            if (p.out_q) *reinterpret_cast<v4i *>(p.out_q + elem) = v4i{qw[0], qw[1], qw[2], qw[3]};
        }
    };
    if (p.res_in) close(std::true_type{});
    else close(std::false_type{});
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Second organisation ("planar", the default): the first one above keeps every tensor [pixel][channel] - natural for the MFMAs, but
// it makes the launch VALU-bound twice over: a lane of GEMM1 owns 16 CHANNELS, so every requant needs its own table row (an LDS
// read and two unpack instructions per output), and the depthwise conv sees 4 channels per dword, so a MAC is one v_dot4 (9 per
// output).  Here the hidden tensor is CHANNEL-PLANAR between GEMM1 and the depthwise conv:
//   GEMM1   operands swapped - A = block input (rows = window positions), B = W1 (columns = channels): a lane owns ONE channel and 16
//           consecutive window positions; its requant constants are four registers for the whole slice (3-4 instructions per output)
//           and its 16 bytes go to hidp[channel][position] as they are.  Window rows are padded to a multiple of 4 positions so that a
//           dword never straddles two rows; positions outside the image are cleared through a byte mask map built with the window.
//   DW      thread = (channel, output row): a dword holds 4 horizontally adjacent positions of its channel, so one v_dot4 covers a
//           whole tap row of an output when the three taps share a dword and two when they straddle: 4.5 per output instead of 9, the
//           six shifted copies of the weight row are built once per slice.  Results go to dwo[pixel][channel] byte by byte - the
//           one transposition of the unit, GEMM2 wants K = channels contiguous.
//   GEMM2 / closing as above.
// K0H: no per-channel pre-shift on the hidden side (conv1 and depthwise tables, fast_tables bit 3 of both): one instruction less per requant
template <bool TIE, bool K0>
__device__ __forceinline__ int requant_h(int v, const DyNt &d) {
    return TIE ? dyadic_tie(v, d) : dyadic_nt_k<K0>(v, d);
}

// NG: slice groups.  Maps with few tiles (14 x 14 at batch 128: one workgroup per CU) leave one wave per SIMD and every latency of the
// per-slice chain exposed; with NG > 1 the workgroup has NG x 4 waves, group g walks slices g, g + NG, .. with its own hidp / dwo /
// stage buffers (the block-input window is shared) and the groups' projection accumulators are summed through LDS at the end.
// TW: tile width.  16 everywhere but on the 7 x 7 maps, where an 8 x 8 tile holds one whole image (77 % of its pixels real, against 38 % of an
// 8 x 16 tile) and the wide units there (160 -> 960 -> 160 / 320 channels) spread the projection over the waves by (pixel block,
// output-channel blocks) instead of pixel blocks alone: TH * TW / 32 pixel blocks x 4 / that many groups of output blocks.
template <int S, int KS1, int CT2, bool TIE, bool K0H, int NG, int TW = LB_TW>
__global__ __launch_bounds__(LB_NT * NG) void linear_bottleneck_planar_kernel(const LbP p) {
    static_assert(TW == 16 || TW == 8, "tile width");
    constexpr int NPB = LB_TH * TW / 32, NCG = 4 / NPB, CTW = (CT2 + NCG - 1) / NCG;   // pixel blocks, output-block groups, output blocks per wave
    constexpr int WH = (LB_TH - 1) * S + 3, WW = (TW - 1) * S + 3, WWP = (WW + 3) / 4 * 4, NPOS = WH * WWP, NB1 = (NPOS + 31) / 32;
    constexpr int MAXB = (NB1 + 3) / 4, MAXBP = (NB1 + 4 * NG - 1) / (4 * NG);
    constexpr int CHP = NB1 * 32 + 4;   // bytes per channel plane: an odd number of dwords (conflict-free dword accesses across channels)
    static_assert((CHP / 4) % 2 == 1, "plane pitch");
    constexpr int DWP = TW * 32 + 32;   // bytes per output row of dwo: odd and even rows fall on different banks
    constexpr int N_W1 = 64 * KS1, N_W9 = 18, N_W3 = 64 * CT2, N_ITEMS = 64 + N_W1 + N_W9 + N_W3;   // 16-byte items per slice
    constexpr int OFF_CT1 = 0, OFF_CT2 = 512, OFF_W1 = 1024, OFF_W9 = OFF_W1 + 1024 * KS1, OFF_W3 = OFF_W9 + 288, BUF = OFF_W3 + 1024 * CT2;
    constexpr int NSR = (N_ITEMS + LB_NT - 1) / LB_NT;   // staged 16-byte items per thread and slice
    __shared__ __attribute__((aligned(16))) char xs[NB1 * 32 * 32 * KS1];   // block input [window position][K]
    __shared__ __attribute__((aligned(16))) char vm[NB1 * 32];              // 0xff inside the image, 0 outside (and on padding positions)
    // per group: hidp = quant_act1 output [channel of the slice][window position], dwo = quant_act2 output [output pixel][channel of
    // the slice], two stages of tables + weights; the pool is reused for the cross-group sum of the projection accumulators
    constexpr int HIDB = (32 * CHP + 15) / 16 * 16, DWOB = LB_TH * DWP, GRPB = HIDB + DWOB + 2 * BUF;
    static_assert(NG == 1 || NG * GRPB >= (NG - 1) * LB_NT * 16 * 4, "reduction buffer fits the pool");
    __shared__ __attribute__((aligned(16))) char pool[NG * GRPB];
    __shared__ v4i ct3s[32 * CT2];
    const int tall = threadIdx.x, t = tall & (LB_NT - 1), grp = NG > 1 ? __builtin_amdgcn_readfirstlane(tall >> 8) : 0;
    const int lane = t & 63, wave = t >> 6, gwave = tall >> 6, l31 = lane & 31, h = lane >> 5;
    char *hidp = pool + grp * GRPB, *dwo = hidp + HIDB, *stg0 = dwo + DWOB;
    int bid = blockIdx.x;
    const int tx = bid % p.tiles_x;
    bid /= p.tiles_x;
    const int ty = bid % p.tiles_y, n = bid / p.tiles_y;
    const int oy0 = ty * LB_TH, ox0 = tx * TW, iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;
    const int8_t *img = p.x + (size_t)n * p.H * p.W * p.in_pitch;
#pragma unroll
    for (int i = 0; i < MAXBP; ++i) {
        const int blk = gwave + 4 * NG * i, pos = blk * 32 + l31;
        if (blk < NB1) {
            const int wy = pos / WWP, wx = pos - wy * WWP, iy = iy0 + wy, ix = ix0 + wx;
            const bool ok = wy < WH && wx < WW && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const int8_t *src = img + (size_t)(ok ? iy * p.W + ix : 0) * p.in_pitch + h * 16;
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) *reinterpret_cast<v4i *>(xs + pos * (32 * KS1) + ks * 32 + h * 16) = ldg4(src + ks * 32);
            if (h == 0) vm[pos] = ok ? (char)0xff : (char)0;
        }
    }
    const char *ssrc[NSR];
    int sstep[NSR], sdst[NSR];
#pragma unroll
    for (int r = 0; r < NSR; ++r) {
        const int it = t + r * LB_NT;
        ssrc[r] = nullptr, sstep[r] = 0, sdst[r] = 0;
        if (it < 32) {
            ssrc[r] = (const char *)p.ct1 + it * 16, sstep[r] = 512, sdst[r] = OFF_CT1 + it * 16;
        } else if (it < 64) {
            ssrc[r] = (const char *)p.ct2 + (it - 32) * 16, sstep[r] = 512, sdst[r] = OFF_CT2 + (it - 32) * 16;
        } else if (it < 64 + N_W1) {
            const int k = it - 64, row = k / (2 * KS1), chunk = k % (2 * KS1);
            ssrc[r] = (const char *)p.w1 + (size_t)row * p.w1_pitch + chunk * 16, sstep[r] = 32 * p.w1_pitch, sdst[r] = OFF_W1 + row * (32 * KS1) + chunk * 16;
        } else if (it < 64 + N_W1 + N_W9) {
            const int k = it - 64 - N_W1;
            ssrc[r] = (const char *)p.w9 + (size_t)(k >> 1) * p.w9_pitch + (k & 1) * 16, sstep[r] = 32, sdst[r] = OFF_W9 + k * 16;
        } else if (it < N_ITEMS) {
            const int k = it - 64 - N_W1 - N_W9;
            ssrc[r] = (const char *)p.w3 + (size_t)(k >> 1) * p.w3_pitch + (k & 1) * 16, sstep[r] = 32, sdst[r] = OFF_W3 + k * 16;
        }
    }
    v4i sreg[NSR];
    if (grp < p.nsl) {
#pragma unroll
        for (int r = 0; r < NSR; ++r)
            if (ssrc[r]) *reinterpret_cast<v4i *>(stg0 + sdst[r]) = ldg4(ssrc[r] + (size_t)grp * sstep[r]);
    }
    for (int i = tall; i < 32 * CT2; i += LB_NT * NG) ct3s[i] = ldg4(p.ct3 + i * 4);
    v16i acc2[CTW];
#pragma unroll
    for (int c = 0; c < CTW; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[c][r] = 0;
    const int dc = t & 31, doy = t >> 5;   // depthwise: this thread's channel (== l31) and output row
    // projection: this wave's pixel block and its first output block (then every NCG-th); this lane's output pixel
    const int pb = wave % NPB, cgw = wave / NPB, pl = pb * 32 + l31, py = pl / TW, px = pl % TW, gy = oy0 + py, gx = ox0 + px;
    const bool out_ok = gy < p.Ho && gx < p.Wo;
    const size_t pix = ((size_t)n * p.Ho + gy) * p.Wo + gx;
    __syncthreads();

    for (int jj = 0; jj * NG < p.nsl; ++jj) {
        const int j = jj * NG + grp;
        const bool act = j < p.nsl, more = j + NG < p.nsl;   // (uniform per group; the barriers below are the whole workgroup's)
        const char *sb = stg0 + (jj & 1) * BUF;
        if (more) {
#pragma unroll
            for (int r = 0; r < NSR; ++r)
                if (ssrc[r]) sreg[r] = ldg4(ssrc[r] + (size_t)(j + NG) * sstep[r]);
        }
        // ---------------------------------------------------------------- GEMM1 + quant_act1 -> hidp (this lane: channel l31)
        if (act) {
            v4i wf[KS1];
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) wf[ks] = *reinterpret_cast<const v4i *>(sb + OFF_W1 + l31 * (32 * KS1) + ks * 32 + h * 16);
            DyNt d1 = entry(reinterpret_cast<const v4i *>(sb + OFF_CT1)[l31]);
            asm volatile("" : "+v"(d1.add));
#pragma unroll
            for (int i = 0; i < MAXB; ++i) {
                const int blk = wave + 4 * i;
                if (blk < NB1) {
                    v16i a;
#pragma unroll
                    for (int r = 0; r < 16; ++r) a[r] = 0;
#pragma unroll
                    for (int ks = 0; ks < KS1; ++ks) {
                        const v4i xf = *reinterpret_cast<const v4i *>(xs + (blk * 32 + cperm(l31)) * (32 * KS1) + ks * 32 + h * 16);
                        a = __builtin_amdgcn_mfma_i32_32x32x32_i8(xf, wf[ks], a, 0, 0, 0);
                    }
                    const v4i m4 = *reinterpret_cast<const v4i *>(vm + blk * 32 + 16 * h);
                    char *dst = hidp + l31 * CHP + blk * 32 + 16 * h;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        int qv[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) qv[k] = med3i(requant_h<TIE, K0H>(a[4 * g + k], d1), p.lo1, p.hi1);
                        *reinterpret_cast<int *>(dst + 4 * g) = pack4_fast(qv[0], qv[1], qv[2], qv[3]) & m4[g];
                    }
                }
            }
        }
        __syncthreads();   // B1: hidp complete; every wave is past GEMM2 of the previous slice
        // ---------------------------------------------------------------- depthwise 3x3 + quant_act2 -> dwo (this thread: channel dc, row doy)
        if (act) {
            int acc[TW];
#pragma unroll
            for (int x = 0; x < TW; ++x) acc[x] = 0;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const unsigned char *wp = reinterpret_cast<const unsigned char *>(sb + OFF_W9 + kh * 96 + dc);
                const unsigned wa = (unsigned)wp[0] | ((unsigned)wp[32] << 8) | ((unsigned)wp[64] << 16);   // taps kw = 0, 1, 2 in bytes 0, 1, 2
                const int *rowp = reinterpret_cast<const int *>(hidp + dc * CHP + (doy * S + kh) * WWP);
                if constexpr (S == 1) {
                    // output x = 4 q + r reads bytes x .. x + 2 of the row: inside dword q for r = 0, 1; across q and q + 1 for r = 2, 3
                    const int w0 = (int)wa, w1 = (int)(wa << 8), w2a = (int)(wa << 16), w2b = (int)(wa >> 16), w3a = (int)(wa << 24), w3b = (int)(wa >> 8);
                    int r[TW / 4 + 1];
#pragma unroll
                    for (int q = 0; q < TW / 4 + 1; ++q) r[q] = rowp[q];
#pragma unroll
                    for (int q = 0; q < TW / 4; ++q) {
                        acc[4 * q] = __builtin_amdgcn_sdot4(r[q], w0, acc[4 * q], false);
                        acc[4 * q + 1] = __builtin_amdgcn_sdot4(r[q], w1, acc[4 * q + 1], false);
                        acc[4 * q + 2] = __builtin_amdgcn_sdot4(r[q], w2a, acc[4 * q + 2], false);
                        acc[4 * q + 2] = __builtin_amdgcn_sdot4(r[q + 1], w2b, acc[4 * q + 2], false);
                        acc[4 * q + 3] = __builtin_amdgcn_sdot4(r[q], w3a, acc[4 * q + 3], false);
                        acc[4 * q + 3] = __builtin_amdgcn_sdot4(r[q + 1], w3b, acc[4 * q + 3], false);
                    }
                } else {
                    // output x reads bytes 2 x .. 2 x + 2: inside dword x / 2 for even x; across x / 2 and x / 2 + 1 for odd x
                    const int w0 = (int)wa, w1a = (int)(wa << 16), w1b = (int)(wa >> 16);
                    int r[TW / 2 + 1];
#pragma unroll
                    for (int q = 0; q < TW / 2 + 1; ++q) r[q] = rowp[q];
#pragma unroll
                    for (int q = 0; q < TW / 2; ++q) {
                        acc[2 * q] = __builtin_amdgcn_sdot4(r[q], w0, acc[2 * q], false);
                        acc[2 * q + 1] = __builtin_amdgcn_sdot4(r[q], w1a, acc[2 * q + 1], false);
                        acc[2 * q + 1] = __builtin_amdgcn_sdot4(r[q + 1], w1b, acc[2 * q + 1], false);
                    }
                }
            }
            DyNt d2 = entry(reinterpret_cast<const v4i *>(sb + OFF_CT2)[dc]);
            asm volatile("" : "+v"(d2.add));
            char *dst = dwo + doy * DWP + dc;
#pragma unroll
            for (int x = 0; x < TW; ++x) dst[x * 32] = (char)med3i(requant_h<TIE, K0H>(acc[x], d2), p.lo2, p.hi2);
        }
        if (more) {
#pragma unroll
            for (int r = 0; r < NSR; ++r)
                if (ssrc[r]) *reinterpret_cast<v4i *>(stg0 + ((jj & 1) ^ 1) * BUF + sdst[r]) = sreg[r];
        }
        __syncthreads();   // B2: dwo complete, hidp free, next stage complete
        // ---------------------------------------------------------------- GEMM2 partial sum over this slice
        if (act) {
            const v4i af = *reinterpret_cast<const v4i *>(dwo + py * DWP + px * 32 + h * 16);
#pragma unroll
            for (int ci = 0; ci < CTW; ++ci) {
                const int c = cgw + ci * NCG;
                if (c < CT2) {
                    const v4i wf = *reinterpret_cast<const v4i *>(sb + OFF_W3 + (c * 32 + cperm(l31)) * 32 + h * 16);
                    acc2[ci] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf, af, acc2[ci], 0, 0, 0);
                }
            }
        }
    }

    if constexpr (NG > 1) {   // sum of the groups' accumulators, one 32-channel block at a time: [group - 1][register][thread] ints in the pool
        int *red = reinterpret_cast<int *>(pool);
#pragma unroll
        for (int c = 0; c < CTW; ++c) {
            __syncthreads();   // (first pass: every group is past its last GEMM2; later: the adds of the previous block are done)
            if (grp > 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) red[((grp - 1) * 16 + r) * LB_NT + t] = acc2[c][r];
            }
            __syncthreads();
            if (grp == 0) {
#pragma unroll
                for (int g = 1; g < NG; ++g)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc2[c][r] += red[((g - 1) * 16 + r) * LB_NT + t];
            }
        }
        if (grp > 0) return;
    }
    // -------------------------------------------------------------------- closing: quant_act_int32 (+ identity), next QuantAct
    if (!out_ok) return;
    DyNt dids = dynt_prepare(p.m_id, p.e_id), dq = dynt_prepare(p.mq, p.eq);
    asm volatile("" : "+v"(dids.add), "+v"(dq.add));
    const int clo = p.clamp16 ? -32768 : (int)0x80000000, chi = p.clamp16 ? 32767 : 0x7fffffff;
    auto close = [&](auto with_identity) {
#pragma unroll
        for (int ci = 0; ci < CTW; ++ci) {
            const int c = cgw + ci * NCG, ch = c * 32 + h * 16;
            if (c >= CT2 || ch >= p.out_pitch) continue;
            const size_t elem = pix * p.out_pitch + ch;
            int qw[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                int o[4], qv[4];
                v4i rin = {0, 0, 0, 0};
                if (decltype(with_identity)::value) rin = ldg4(p.res_in + elem + 4 * g);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    DyNt d = entry(ct3s[ch + 4 * g + k]);
                    asm volatile("" : "+v"(d.add));
                    int ov = requant<TIE>(acc2[ci][4 * g + k], d);
                    if (decltype(with_identity)::value) ov += requant<TIE>(rin[k], dids);
                    ov = med3i(ov, clo, chi);
                    o[k] = ov;
                    qv[k] = med3i(requant<TIE>(ov, dq), p.q_lo, p.q_hi);
                }
                if (p.res_out) *reinterpret_cast<v4i *>(p.res_out + elem + 4 * g) = v4i{o[0], o[1], o[2], o[3]};
                qw[g] = pack4_fast(qv[0], qv[1], qv[2], qv[3]);
            }
            if (p.out_q) *reinterpret_cast<v4i *>(p.out_q + elem) = v4i{qw[0], qw[1], qw[2], qw[3]};
        }
    };
    if (p.res_in) close(std::true_type{});
    else close(std::false_type{});
}

typedef void (*LbFn)(const LbP);
template <int S, int KS1, int CT2, int NG>
LbFn pick_planar(bool tie, bool k0h) {
    if (k0h && !tie) return linear_bottleneck_planar_kernel<S, KS1, CT2, false, true, NG>;
    return tie ? linear_bottleneck_planar_kernel<S, KS1, CT2, true, false, NG> : linear_bottleneck_planar_kernel<S, KS1, CT2, false, false, NG>;
}
// organisation 1 ([pixel][channel]) exists for K steps / output blocks up to 2; the planar one up to 3 (96-channel inputs and outputs);
// slice groups (ng 2, 4) exist for stride 1 (the 14 x 14 maps are where workgroups are scarce)
template <int S, int KS1, int CT2>
LbFn pick_variant(bool tie, bool planar, bool k0h, int ng) {
    if constexpr (KS1 <= 2 && CT2 <= 2) {
        if (!planar) return tie ? linear_bottleneck_kernel<S, KS1, CT2, true> : linear_bottleneck_kernel<S, KS1, CT2, false>;
    }
    if constexpr (S == 1) {
        if (ng == 2) return pick_planar<S, KS1, CT2, 2>(tie, k0h);
        if (ng == 4) return pick_planar<S, KS1, CT2, 4>(tie, k0h);
    }
    return pick_planar<S, KS1, CT2, 1>(tie, k0h);
}
// the 8 x 8 tile family: the wide units on maps of at most 8 x 8 pixels (MobileNetV2 w1 units 14-17), two slice groups
template <int S, int KS1, int CT2, int NG>
LbFn pick_t8_variant(bool tie, bool k0h) {
    if (k0h && !tie) return linear_bottleneck_planar_kernel<S, KS1, CT2, false, true, NG, 8>;
    return tie ? linear_bottleneck_planar_kernel<S, KS1, CT2, true, false, NG, 8> : linear_bottleneck_planar_kernel<S, KS1, CT2, false, false, NG, 8>;
}
LbFn pick_t8(int stride, int ks1, int ct2, bool tie, bool k0h, int *ng) {
    int dummy;
    if (!ng) ng = &dummy;
    if (stride == 1 && ks1 == 5 && ct2 == 5) return *ng = 4, pick_t8_variant<1, 5, 5, 4>(tie, k0h);
    if (stride == 1 && ks1 == 5 && ct2 == 10) return *ng = 2, pick_t8_variant<1, 5, 10, 2>(tie, k0h);
    if (stride == 2 && ks1 == 3 && ct2 == 5) return *ng = 2, pick_t8_variant<2, 3, 5, 2>(tie, k0h);
    return nullptr;
}
bool lb_wide(int ip, int op) { return ip > 96 || op > 96; }

template <int S, int KS1>
LbFn pick_ct2(int ct2, bool tie, bool planar, bool k0h, int ng) {
    return ct2 == 1 ? pick_variant<S, KS1, 1>(tie, planar, k0h, ng) : ct2 == 2 ? pick_variant<S, KS1, 2>(tie, planar, k0h, ng) : pick_variant<S, KS1, 3>(tie, planar, k0h, ng);
}
template <int S>
LbFn pick(int ks1, int ct2, bool tie, bool planar, bool k0h, int ng) {
    return ks1 == 1 ? pick_ct2<S, 1>(ct2, tie, planar, k0h, ng) : ks1 == 2 ? pick_ct2<S, 2>(ct2, tie, planar, k0h, ng) : pick_ct2<S, 3>(ct2, tie, planar, k0h, ng);
}

bool e_fast(int ek) { return (ek & 0xff) >= 33 && (ek & 0xff) <= 62 && (ek >> 8) >= 0 && (ek >> 8) < 31; }

// nullptr when the launch takes the unit, else why not
const char *lb_refusal(const hawq_bottleneck_args *a) {
    const hawq_conv_args &e = a->expand, &q = a->project;
    if (!e.in || !e.wgt || !e.ctab || !q.wgt || !q.ctab || !a->dw_wgt9c || !a->dw_ctab) return "null pointer (in / wgt / ctab of the three layers)";
    if (e.N <= 0 || e.H <= 0 || e.W <= 0) return "empty input";
    if (e.KH != 1 || e.KW != 1 || e.stride != 1 || e.pad != 0 || q.KH != 1 || q.KW != 1 || q.stride != 1 || q.pad != 0) return "conv1 / conv3 must be 1x1, stride 1";
    if (e.in_bits != 8 || e.w_bits != 8 || q.in_bits != 8 || q.w_bits != 8 || e.in2 || q.in2 || e.in_planar || q.out_planar) return "int8 NHWC single-branch layers only";
    if (e.epilogue != HAWQ_EPI_REQUANT || !e.fast_tables || !e.relu) return "expand: REQUANT epilogue with ReLU and fast_tables";
    if (q.epilogue != HAWQ_EPI_RESIDUAL || !q.fast_tables || !q.res_no_relu) return "project: signed RESIDUAL epilogue (res_no_relu) with fast_tables";
    if (!a->dw_fast_tables || (a->dw_stride != 1 && a->dw_stride != 2)) return "depthwise: fast tables, stride 1 or 2";
    const int ip = e.in_pitch ? e.in_pitch : e.Cin, op = q.out_pitch ? q.out_pitch : q.Cout;
    if (e.Cin % 64 || e.Cin > 192 || (ip != 16 && ip != 32 && ip != 64 && ip != 96 && ip != 160) || ip > e.Cin) return "expand: K = 64 / 128 / 192 packed weights, in_pitch 16 / 32 / 64 / 96 / 160";
    if (e.Cout <= 0 || e.Cout % 64 || q.Cin != e.Cout || a->c_mid <= 0 || a->c_mid > e.Cout) return "hidden width: expand.Cout == project.Cin, a multiple of 64, c_mid inside it";
    if (q.Cout % 64 || q.Cout > 320 || (op != 16 && op != 32 && op != 64 && op != 96 && op != 160 && op != 320) || op > q.Cout) return "project: Cout = 64 .. 320 packed rows, out_pitch 16 / 32 / 64 / 96 / 160 / 320";
    if (a->tile == 1 && (ip > 64 || op > 64)) return "tile 1 (the [pixel][channel] organisation) takes at most 64-channel inputs and outputs";
    if (e.q_hi < 0 || e.q_hi > 127 || a->dw_q_lo < 0 || a->dw_q_hi < a->dw_q_lo || a->dw_q_hi > 127) return "hidden activations must be 0 .. 127 int8 (ReLU in the clamp)";
    if (q.out_q && (q.out_bits != 8 || q.q_lo < -128 || q.q_hi > 127 || q.q_lo > q.q_hi || q.mq < 0 || !e_fast(q.eq))) return "project: int8 out_q with a fast (mq, eq)";
    if (q.res_in && (q.res_in_bits != 32 || q.m_id_scalar < 0 || !e_fast(q.e_id_scalar) || a->dw_stride != 1)) return "identity: int32 carrier, fast scalar table, stride 1";
    if (q.res_out && q.res_out_bits != 32) return "res_out must be the int32 carrier";
    if (!q.out_q && !q.res_out) return "nothing to write";
    const int H = e.H, W = e.W, Ho = (H - 1) / a->dw_stride + 1, Wo = (W - 1) / a->dw_stride + 1;
    if (q.N != e.N || q.H != Ho || q.W != Wo) return "project geometry must be the depthwise conv's output grid";
    if (a->tile < 0 || a->tile > 4) return "tile: 0 (default) .. 4";
    if (lb_wide(ip, op)) {   // more than 96 channels in or out: the 8 x 8 tile family, maps of at most 8 x 8 output pixels, at least four slices (one per slice group)
        if (a->tile == 1) return "tile 1 (the [pixel][channel] organisation) takes at most 64-channel inputs and outputs";
        if (Ho > 8 || Wo > 8 || a->c_mid <= 96) return "inputs / outputs wider than 96 channels: output maps of at most 8 x 8 pixels only";
        if (!pick_t8(a->dw_stride, (ip + 31) / 32, (op + 31) / 32, false, false, nullptr)) return "no instantiation for this wide unit (160 -> 160 / 320 at stride 1, 96 -> 160 at stride 2)";
    }
    if ((long long)e.N * ((Ho + LB_TH - 1) / LB_TH) * ((Wo + LB_TW - 1) / LB_TW) > 0x7fffffffll) return "grid too large";
    return nullptr;
}

}  // namespace

extern "C" int hawq_linear_bottleneck_ok(const hawq_bottleneck_args *a) { return a && lb_refusal(a) == nullptr ? 1 : 0; }

extern "C" int hawq_linear_bottleneck(const hawq_bottleneck_args *a, void *stream) {
    HAWQ_REQUIRE(a, "hawq_linear_bottleneck: null args");
    const char *why = lb_refusal(a);
    HAWQ_REQUIRE(!why, "hawq_linear_bottleneck: %s", why);
    const hawq_conv_args &e = a->expand, &q = a->project;
    LbP p;
    p.x = (const int8_t *)e.in;
    p.N = e.N, p.H = e.H, p.W = e.W, p.Ho = q.H, p.Wo = q.W;
    p.in_pitch = e.in_pitch ? e.in_pitch : e.Cin;
    p.w1 = (const int8_t *)e.wgt, p.w1_pitch = e.Cin, p.ct1 = e.ctab, p.lo1 = e.q_lo < 0 ? 0 : e.q_lo, p.hi1 = e.q_hi;
    p.w9 = a->dw_wgt9c, p.w9_pitch = e.Cout, p.ct2 = a->dw_ctab, p.lo2 = a->dw_q_lo, p.hi2 = a->dw_q_hi;
    p.w3 = (const int8_t *)q.wgt, p.w3_pitch = q.Cin, p.ct3 = q.ctab;
    p.nsl = (a->c_mid + 31) / 32;
    p.res_in = (const int32_t *)q.res_in, p.m_id = q.res_in ? q.m_id_scalar : 0, p.e_id = q.res_in ? q.e_id_scalar : 33;
    p.res_out = (int32_t *)q.res_out, p.out_q = (int8_t *)q.out_q;
    p.mq = q.out_q ? q.mq : 0, p.eq = q.out_q ? q.eq : 33, p.q_lo = q.q_lo, p.q_hi = q.q_hi, p.clamp16 = q.res_clamp16;
    p.out_pitch = q.out_pitch ? q.out_pitch : q.Cout;
    const bool wide = lb_wide(p.in_pitch, p.out_pitch);
    const int tw = wide ? 8 : LB_TW;
    p.tiles_x = (p.Wo + tw - 1) / tw, p.tiles_y = (p.Ho + LB_TH - 1) / LB_TH;
    const bool tie = ((e.fast_tables | a->dw_fast_tables | q.fast_tables) & 4) != 0;
    const int ks1 = (p.in_pitch + 31) / 32, ct2 = (p.out_pitch + 31) / 32;
    const bool planar = a->tile != 1;   // tile 1: the [pixel][channel] organisation (A/B measurements)
    const bool k0h = (e.fast_tables & 8) && (a->dw_fast_tables & 8);   // the caller's promise: no per-channel pre-shift in ctab / dw_ctab
    // slice groups: tile 2 / 3 / 4 force 1 / 2 / 4; by default as many as bring the launch to about three waves per SIMD
    const long long wgs = (long long)p.N * p.tiles_y * p.tiles_x;
    int ng = a->tile == 2 ? 1 : a->tile == 3 ? 2 : a->tile == 4 ? 4 : (wgs >= 768 ? 1 : (wgs >= 384 ? 2 : 4));
    if (!planar || a->dw_stride != 1) ng = 1;
    while (ng > 1 && ng > p.nsl) ng >>= 1;
    LbFn fn;
    if (wide) fn = pick_t8(a->dw_stride, ks1, ct2, tie, k0h, &ng);
    else fn = a->dw_stride == 1 ? pick<1>(ks1, ct2, tie, planar, k0h, ng) : pick<2>(ks1, ct2, tie, planar, k0h, ng);
    HAWQ_REQUIRE(fn, "hawq_linear_bottleneck: no kernel for this unit");
    hipLaunchKernelGGL(fn, dim3((unsigned)wgs), dim3(LB_NT * ng), 0, (hipStream_t)stream, p);
    HAWQ_CHECK_HIP(hipGetLastError());
    return 0;
}
