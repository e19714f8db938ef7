// Implicit-GEMM integer convolution for gfx950 (MI355X) with fused HAWQ epilogues.
//
// GEMM view: D[Cout][M] = W[Cout][K] * X[K][M],  M = N*Ho*Wo output pixels,
// K = KH*KW*Cin.  Weights are the MFMA "A" operand (rows = output channels), pixels the "B"
// operand (cols), so that after v_mfma_i32_32x32x32_i8 every lane owns ONE pixel and, thanks
// to a row permutation applied when the weight fragment is read from LDS, 16 CONSECUTIVE
// output channels of it: the epilogue then stores 16 int8 (one dwordx4) / 16 uint16 (two
// dwordx4) per lane straight into the NHWC output.
//
// K is walked in chunks of 64 input channels of one filter tap.  Each chunk of both operands is
// fetched global->registers (16 B per thread per row, im2col addressing with zero fill for
// padding), unpacked to int8 if the tensor is 4-bit (hawq4 nibble format), and written to a
// double-buffered, XOR-swizzled LDS tile [rows][64 B]; fragments are read back with
// conflict-free ds_read_b128.  All MACs are exact int32.
//
// Reference arithmetic replaced: quant_modules.py:489-494 (conv), q_resnet.py:242-258 (ReLU,
// residual add), quant_utils.py:390-456 (fixedpoint_fn case 0 / case 1).
#include "common.h"

namespace {

struct ConvP {
    const uint8_t *in, *wgt;
    const int32_t *bias;
    int N, H, W, Cin, Cout, KH, KW, stride, pad, Ho, Wo, M;
    int in_bits, w_bits;
    const uint8_t *in2, *wgt2;
    const int32_t *bias2;
    int H2, W2, Cin2, stride2, in2_bits, w2_bits;
    int relu;
    const int32_t *m, *e, *m_id, *e_id;
    int m_id_s, e_id_s;
    const void *res_in;
    int res_in_bits;
    void *res_out;
    int res_out_bits;
    void *out_q;
    int out_bits, q_lo, q_hi, mq, eq;
    int32_t *out_acc;
    float *out_f32;
    const float *fscale;
    int ldo, n_valid;
    int32_t *flags;
};

template <int BM_, int BN_, int WM_, int WN_>
struct Cfg {
    static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_;
    static constexpr int PT = BM / WM / 32;  // pixel MFMA tiles per wave
    static constexpr int CT = BN / WN / 32;  // channel MFMA tiles per wave
    static constexpr int AL = BM / 64;       // 16-B A loads per thread per chunk
    static constexpr int WL = BN / 64;
    static constexpr int LDS_BYTES = 2 * (BM + BN) * 64;
    static_assert(WM * WN == 4, "4 waves per workgroup");
};

// MFMA C/D row i of a 32x32 tile lives in (reg, half) with i = (reg&3) + 8*(reg>>2) + 4*half.
// Reading weight row cperm(i) as MFMA row i makes lane-half h own channels 16h..16h+15.
__device__ __forceinline__ int cperm(int i) { return (((i >> 2) & 1) << 4) + (i & 3) + ((i >> 3) << 2); }

__device__ __forceinline__ int lds_off(int row, int slot) { return row * 64 + ((slot ^ ((row >> 2) & 3)) << 4); }

// 8 bytes of hawq4 (16 channels) -> 16 int8.  Unsigned: zero-extended.  Signed weights come
// out as value*16 (nibble moved to the top of its byte); the accumulator is shifted back by 4
// after the K loop, which is exact.
template <bool SIGNED16>
__device__ __forceinline__ v4i unpack16(unsigned x0, unsigned x1) {
    v4i r;
    if (SIGNED16) {
        r.x = (int)((x0 << 4) & 0xF0F0F0F0u);
        r.y = (int)(x0 & 0xF0F0F0F0u);
        r.z = (int)((x1 << 4) & 0xF0F0F0F0u);
        r.w = (int)(x1 & 0xF0F0F0F0u);
    } else {
        r.x = (int)(x0 & 0x0F0F0F0Fu);
        r.y = (int)((x0 >> 4) & 0x0F0F0F0Fu);
        r.z = (int)(x1 & 0x0F0F0F0Fu);
        r.w = (int)((x1 >> 4) & 0x0F0F0F0Fu);
    }
    return r;
}

template <int BITS>
__device__ __forceinline__ v4i load_chunk16(const uint8_t *p, bool valid) {
    v4i z = {0, 0, 0, 0};
    if (!valid) return z;
    if (BITS == 8) return *reinterpret_cast<const v4i *>(p);
    v2i t = *reinterpret_cast<const v2i *>(p);
    z.x = t.x;
    z.y = t.y;
    return z;
}

// One GEMM segment: acc[ct][pt] += W_tile * X_tile over all taps and input channels.
template <class C, int A_BITS, int W_BITS>
__device__ __forceinline__ void gemm_segment(v16i (&acc)[C::CT][C::PT], const uint8_t *__restrict__ in,
                                             const uint8_t *__restrict__ wgt, int H, int W, int Cin, int KH,
                                             int KW, int stride, int pad, int Ho, int Wo, int M, int Cout,
                                             int m0, int c0, char *smem) {
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wave_m = wave % C::WM, wave_c = wave / C::WM;
    const int lrow = t >> 2, lslot = t & 3;
    constexpr int ABYTES = A_BITS == 8 ? 16 : 8;  // bytes per 16 channels in global memory
    constexpr int WBYTES = W_BITS == 8 ? 16 : 8;

    // per-thread im2col row bookkeeping
    int pix_base[C::AL], iy0[C::AL], ix0[C::AL];
    bool mval[C::AL];
#pragma unroll
    for (int i = 0; i < C::AL; ++i) {
        const int m = m0 + lrow + 64 * i;
        mval[i] = m < M;
        const int mm = mval[i] ? m : 0;
        const int n = mm / (Ho * Wo);
        const int r = mm - n * (Ho * Wo);
        const int oy = r / Wo, ox = r - oy * Wo;
        iy0[i] = oy * stride - pad;
        ix0[i] = ox * stride - pad;
        pix_base[i] = (n * H + iy0[i]) * W + ix0[i];
    }
    const int taps = KH * KW;
    const int cchunks = Cin >> 6;
    const int nk = taps * cchunks;
    const size_t wrow_bytes = (size_t)taps * Cin * W_BITS / 8;

    char *ldsA = smem;                 // [2][BM][64]
    char *ldsW = smem + 2 * C::BM * 64;  // [2][BN][64]

    v4i ra[C::AL], rw[C::WL];
    int kh = 0, kw = 0, cc = 0;  // coordinates of the chunk being LOADED

    auto load_regs = [&]() {
#pragma unroll
        for (int i = 0; i < C::AL; ++i) {
            const int iy = iy0[i] + kh, ix = ix0[i] + kw;
            const bool v = mval[i] && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            const size_t off = ((size_t)(pix_base[i] + kh * W + kw) * Cin + (cc << 6)) * A_BITS / 8 + lslot * ABYTES;
            ra[i] = load_chunk16<A_BITS>(in + (v ? off : 0), v);
        }
#pragma unroll
        for (int j = 0; j < C::WL; ++j) {
            const int c = c0 + lrow + 64 * j;
            const bool v = c < Cout;
            const size_t off = (size_t)c * wrow_bytes + ((size_t)((kh * KW + kw) * Cin + (cc << 6))) * W_BITS / 8 +
                               lslot * WBYTES;
            rw[j] = load_chunk16<W_BITS>(wgt + (v ? off : 0), v);
        }
        if (++cc == cchunks) {
            cc = 0;
            if (++kw == KW) {
                kw = 0;
                ++kh;
            }
        }
    };
    auto store_lds = [&](int buf) {
#pragma unroll
        for (int i = 0; i < C::AL; ++i) {
            v4i v = ra[i];
            if (A_BITS == 4) v = unpack16<false>((unsigned)v.x, (unsigned)v.y);
            *reinterpret_cast<v4i *>(ldsA + buf * C::BM * 64 + lds_off(lrow + 64 * i, lslot)) = v;
        }
#pragma unroll
        for (int j = 0; j < C::WL; ++j) {
            v4i v = rw[j];
            if (W_BITS == 4) v = unpack16<true>((unsigned)v.x, (unsigned)v.y);
            *reinterpret_cast<v4i *>(ldsW + buf * C::BN * 64 + lds_off(lrow + 64 * j, lslot)) = v;
        }
    };

    const int l31 = lane & 31, h = lane >> 5;
    int arow[C::PT], wrow[C::CT];
#pragma unroll
    for (int p = 0; p < C::PT; ++p) arow[p] = wave_m * (C::PT * 32) + p * 32 + l31;
#pragma unroll
    for (int c = 0; c < C::CT; ++c) wrow[c] = wave_c * (C::CT * 32) + c * 32 + cperm(l31);

    load_regs();
    store_lds(0);
    __syncthreads();
    for (int k = 0; k < nk; ++k) {
        const int buf = k & 1;
        if (k + 1 < nk) load_regs();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int slot = 2 * ks + h;
            v4i wf[C::CT], af[C::PT];
#pragma unroll
            for (int c = 0; c < C::CT; ++c)
                wf[c] = *reinterpret_cast<const v4i *>(ldsW + buf * C::BN * 64 + lds_off(wrow[c], slot));
#pragma unroll
            for (int p = 0; p < C::PT; ++p)
                af[p] = *reinterpret_cast<const v4i *>(ldsA + buf * C::BM * 64 + lds_off(arow[p], slot));
#pragma unroll
            for (int c = 0; c < C::CT; ++c)
#pragma unroll
                for (int p = 0; p < C::PT; ++p)
                    acc[c][p] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[c], af[p], acc[c][p], 0, 0, 0);
        }
        if (k + 1 < nk) store_lds(buf ^ 1);
        __syncthreads();
    }
    if (W_BITS == 4) {
#pragma unroll
        for (int c = 0; c < C::CT; ++c)
#pragma unroll
            for (int p = 0; p < C::PT; ++p)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[c][p][r] >>= 4;
    }
}

template <class C>
__device__ __forceinline__ void run_segment(v16i (&acc)[C::CT][C::PT], const uint8_t *in, const uint8_t *wgt,
                                            int a_bits, int w_bits, int H, int W, int Cin, int KH, int KW,
                                            int stride, int pad, int Ho, int Wo, int M, int Cout, int m0, int c0,
                                            char *smem) {
    if (a_bits == 8 && w_bits == 8)
        gemm_segment<C, 8, 8>(acc, in, wgt, H, W, Cin, KH, KW, stride, pad, Ho, Wo, M, Cout, m0, c0, smem);
    else if (a_bits == 4 && w_bits == 4)
        gemm_segment<C, 4, 4>(acc, in, wgt, H, W, Cin, KH, KW, stride, pad, Ho, Wo, M, Cout, m0, c0, smem);
    else if (a_bits == 8 && w_bits == 4)
        gemm_segment<C, 8, 4>(acc, in, wgt, H, W, Cin, KH, KW, stride, pad, Ho, Wo, M, Cout, m0, c0, smem);
    else
        gemm_segment<C, 4, 8>(acc, in, wgt, H, W, Cin, KH, KW, stride, pad, Ho, Wo, M, Cout, m0, c0, smem);
}

__device__ __forceinline__ void load16(const int32_t *p, int (&v)[16]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v4i t = reinterpret_cast<const v4i *>(p)[i];
        v[4 * i] = t.x;
        v[4 * i + 1] = t.y;
        v[4 * i + 2] = t.z;
        v[4 * i + 3] = t.w;
    }
}

// store 16 clamped ints as int8 (16 B) or hawq4 (8 B) at channel offset ch of pixel pix
__device__ __forceinline__ void store_q16(void *out, int bits, size_t elem, const int (&q)[16]) {
    if (bits == 8) {
        v4i w;
        w.x = (int)pack4_i8(q[0], q[1], q[2], q[3]);
        w.y = (int)pack4_i8(q[4], q[5], q[6], q[7]);
        w.z = (int)pack4_i8(q[8], q[9], q[10], q[11]);
        w.w = (int)pack4_i8(q[12], q[13], q[14], q[15]);
        *reinterpret_cast<v4i *>((char *)out + elem) = w;
    } else {
        v2i w;
        w.x = (int)pack8_u4(&q[0]);
        w.y = (int)pack8_u4(&q[8]);
        *reinterpret_cast<v2i *>((char *)out + (elem >> 1)) = w;
    }
}

template <class C, int EPI, bool DUAL>
__global__ __launch_bounds__(256) void conv_kernel(const ConvP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // XCD-aware tile order: consecutive workgroup ids land on different XCDs (id % 8), so give
    // each XCD a contiguous run of pixel tiles that share the same weight tile in its L2.
    const int tiles_m = (p.M + C::BM - 1) / C::BM;
    const int tiles_c = p.Cout / C::BN;
    const int nwg = tiles_m * tiles_c;
    int wg = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = wg & 7, idx = wg >> 3;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tc = wg % tiles_c, tm = wg / tiles_c;  // channel tiles of one pixel tile are adjacent
    const int m0 = tm * C::BM, c0 = tc * C::BN;

    v16i acc[C::CT][C::PT];
#pragma unroll
    for (int c = 0; c < C::CT; ++c)
#pragma unroll
        for (int q = 0; q < C::PT; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][q][r] = 0;
    run_segment<C>(acc, p.in, p.wgt, p.in_bits, p.w_bits, p.H, p.W, p.Cin, p.KH, p.KW, p.stride, p.pad, p.Ho,
                   p.Wo, p.M, p.Cout, m0, c0, smem);
    v16i acc2[DUAL ? C::CT : 1][DUAL ? C::PT : 1];
    if (DUAL) {
#pragma unroll
        for (int c = 0; c < C::CT; ++c)
#pragma unroll
            for (int q = 0; q < C::PT; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[DUAL ? c : 0][DUAL ? q : 0][r] = 0;
        run_segment<C>(reinterpret_cast<v16i(&)[C::CT][C::PT]>(acc2), p.in2, p.wgt2, p.in2_bits, p.w2_bits, p.H2,
                       p.W2, p.Cin2, 1, 1, p.stride2, 0, p.Ho, p.Wo, p.M, p.Cout, m0, c0, smem);
    }

    // ---------------------------------------------------------------- epilogue
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wave_m = wave % C::WM, wave_c = wave / C::WM;
    const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
    for (int c = 0; c < C::CT; ++c) {
        const int ch = c0 + wave_c * (C::CT * 32) + c * 32 + h * 16;  // first of this lane's 16 channels
        int bias[16], mm[16], ee[16], bias2[16], m1[16], e1[16];
        load16(p.bias + ch, bias);
        if (EPI == HAWQ_EPI_REQUANT || EPI == HAWQ_EPI_RESIDUAL) {
            load16(p.m + ch, mm);
            load16(p.e + ch, ee);
        }
        if (DUAL) {
            load16(p.bias2 + ch, bias2);
            load16(p.m_id + ch, m1);
            load16(p.e_id + ch, e1);
        }
#pragma unroll
        for (int q = 0; q < C::PT; ++q) {
            const int pix = m0 + wave_m * (C::PT * 32) + q * 32 + l31;
            if (pix >= p.M) continue;
            const size_t elem = (size_t)pix * p.Cout + ch;
            int v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = acc[c][q][r] + bias[r];
            if (EPI == HAWQ_EPI_RAW) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    v4i w = {v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]};
                    reinterpret_cast<v4i *>(p.out_acc + elem)[i] = w;
                }
            } else if (EPI == HAWQ_EPI_DEQUANT) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (ch + r < p.n_valid) p.out_f32[(size_t)pix * p.ldo + ch + r] = (float)v[r] * p.fscale[ch + r];
            } else if (EPI == HAWQ_EPI_REQUANT) {
                int qv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int x = p.relu ? max(v[r], 0) : v[r];
                    qv[r] = clampi(dyadic_rne(x, mm[r], ee[r]), p.q_lo, p.q_hi);
                }
                store_q16(p.out_q, p.out_bits, elem, qv);
            } else {  // RESIDUAL
                int idv[16];
                if (DUAL) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        idv[r] = dyadic_rne(acc2[DUAL ? c : 0][DUAL ? q : 0][r] + bias2[r], m1[r], e1[r]);
                } else {
                    if (p.res_in_bits == 16) {
                        const v4i *src = reinterpret_cast<const v4i *>((const uint16_t *)p.res_in + elem);
                        v4i a = src[0], b = src[1];
                        const int w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            idv[2 * i] = w[i] & 0xffff;
                            idv[2 * i + 1] = (unsigned)w[i] >> 16;
                        }
                    } else {
                        load16((const int32_t *)p.res_in + elem, idv);
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) idv[r] = dyadic_rne(idv[r], p.m_id_s, p.e_id_s);
                }
                int o[16];
                bool ovf = false;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    o[r] = max(dyadic_rne(v[r], mm[r], ee[r]) + idv[r], 0);  // no clamp: quant_utils.py:456
                    ovf |= o[r] > 65535;
                }
                if (p.res_out) {
                    if (p.res_out_bits == 16) {
                        if (ovf) atomicOr(p.flags, 1);
                        v4i a, b;
                        int w[8];
#pragma unroll
                        for (int i = 0; i < 8; ++i)
                            w[i] = min(o[2 * i], 65535) | (min(o[2 * i + 1], 65535) << 16);
                        a.x = w[0], a.y = w[1], a.z = w[2], a.w = w[3];
                        b.x = w[4], b.y = w[5], b.z = w[6], b.w = w[7];
                        v4i *dst = reinterpret_cast<v4i *>((uint16_t *)p.res_out + elem);
                        dst[0] = a;
                        dst[1] = b;
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            v4i w = {o[4 * i], o[4 * i + 1], o[4 * i + 2], o[4 * i + 3]};
                            reinterpret_cast<v4i *>((int32_t *)p.res_out + elem)[i] = w;
                        }
                    }
                }
                if (p.out_q) {
                    int qv[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) qv[r] = clampi(dyadic_rne(o[r], p.mq, p.eq), p.q_lo, p.q_hi);
                    store_q16(p.out_q, p.out_bits, elem, qv);
                }
            }
        }
    }
}

using T0 = Cfg<128, 128, 2, 2>;
using T1 = Cfg<256, 64, 4, 1>;
using T2 = Cfg<64, 64, 2, 2>;
using T3 = Cfg<128, 64, 2, 2>;
constexpr int NUM_TILES = 4;

typedef void (*KernelFn)(const ConvP);
struct TileInfo {
    int BM, BN, lds;
    KernelFn fn[5];  // RAW, REQUANT, RESIDUAL, DEQUANT, RESIDUAL+DUAL
};
#define TILE_ENTRY(T)                                                                                   \
    {                                                                                                   \
        T::BM, T::BN, T::LDS_BYTES, {                                                                   \
            conv_kernel<T, HAWQ_EPI_RAW, false>, conv_kernel<T, HAWQ_EPI_REQUANT, false>,               \
                conv_kernel<T, HAWQ_EPI_RESIDUAL, false>, conv_kernel<T, HAWQ_EPI_DEQUANT, false>,      \
                conv_kernel<T, HAWQ_EPI_RESIDUAL, true>                                                 \
        }                                                                                               \
    }
const TileInfo kTiles[NUM_TILES] = {TILE_ENTRY(T0), TILE_ENTRY(T1), TILE_ENTRY(T2), TILE_ENTRY(T3)};

int pick_tile(int M, int Cout, bool dual) {
    // Enough workgroups to fill 256 CUs a few times over, the largest tile that allows it.
    auto nwg = [&](int t) { return ((M + kTiles[t].BM - 1) / kTiles[t].BM) * (Cout / kTiles[t].BN); };
    if (Cout % 128 == 0 && nwg(0) >= 1024) return 0;
    if (nwg(1) >= 1024 && !dual) return 1;
    if (Cout % 128 == 0 && nwg(0) >= 512) return 0;
    if (nwg(3) >= 512) return 3;
    return 2;
}

}  // namespace

extern "C" int hawq_conv2d_num_tiles(void) { return NUM_TILES; }

extern "C" int hawq_conv2d(const hawq_conv_args *a, void *stream) {
    HAWQ_REQUIRE(a != nullptr, "hawq_conv2d: null args");
    HAWQ_REQUIRE(a->in && a->wgt && a->bias, "hawq_conv2d: in/wgt/bias must be non-null");
    HAWQ_REQUIRE(a->Cin > 0 && a->Cin % 64 == 0, "hawq_conv2d: Cin=%d must be a positive multiple of 64", a->Cin);
    HAWQ_REQUIRE(a->Cout > 0 && a->Cout % 64 == 0, "hawq_conv2d: Cout=%d must be a positive multiple of 64", a->Cout);
    HAWQ_REQUIRE((a->in_bits == 8 || a->in_bits == 4) && (a->w_bits == 8 || a->w_bits == 4),
                 "hawq_conv2d: in_bits/w_bits must be 4 or 8 (got %d/%d)", a->in_bits, a->w_bits);
    HAWQ_REQUIRE(a->KH > 0 && a->KW > 0 && a->stride > 0 && a->pad >= 0 && a->N > 0 && a->H > 0 && a->W > 0,
                 "hawq_conv2d: bad geometry");
    ConvP p;
    p.in = (const uint8_t *)a->in;
    p.wgt = (const uint8_t *)a->wgt;
    p.bias = a->bias;
    p.N = a->N, p.H = a->H, p.W = a->W, p.Cin = a->Cin, p.Cout = a->Cout;
    p.KH = a->KH, p.KW = a->KW, p.stride = a->stride, p.pad = a->pad;
    p.Ho = (a->H + 2 * a->pad - a->KH) / a->stride + 1;
    p.Wo = (a->W + 2 * a->pad - a->KW) / a->stride + 1;
    HAWQ_REQUIRE(p.Ho > 0 && p.Wo > 0, "hawq_conv2d: empty output");
    const long long M = (long long)a->N * p.Ho * p.Wo;
    HAWQ_REQUIRE(M * (long long)a->Cout < (1ll << 40) && M < (1ll << 30), "hawq_conv2d: problem too large");
    p.M = (int)M;
    p.in_bits = a->in_bits, p.w_bits = a->w_bits;
    const bool dual = a->in2 != nullptr;
    p.in2 = (const uint8_t *)a->in2, p.wgt2 = (const uint8_t *)a->wgt2, p.bias2 = a->bias2;
    p.H2 = a->H2, p.W2 = a->W2, p.Cin2 = a->Cin2, p.stride2 = a->stride2;
    p.in2_bits = a->in2_bits, p.w2_bits = a->w2_bits;
    if (dual) {
        HAWQ_REQUIRE(a->epilogue == HAWQ_EPI_RESIDUAL, "hawq_conv2d: second branch needs HAWQ_EPI_RESIDUAL");
        HAWQ_REQUIRE(a->wgt2 && a->bias2 && a->m_id && a->e_id, "hawq_conv2d: second branch tables missing");
        HAWQ_REQUIRE(a->Cin2 > 0 && a->Cin2 % 64 == 0, "hawq_conv2d: Cin2 must be a multiple of 64");
        HAWQ_REQUIRE((a->in2_bits == 8 || a->in2_bits == 4) && (a->w2_bits == 8 || a->w2_bits == 4),
                     "hawq_conv2d: in2_bits/w2_bits must be 4 or 8");
        HAWQ_REQUIRE((a->H2 - 1) / a->stride2 + 1 == p.Ho && (a->W2 - 1) / a->stride2 + 1 == p.Wo,
                     "hawq_conv2d: second branch output grid differs");
    }
    p.relu = a->relu;
    p.m = a->m, p.e = a->e, p.m_id = a->m_id, p.e_id = a->e_id;
    p.m_id_s = a->m_id_scalar, p.e_id_s = a->e_id_scalar;
    p.res_in = a->res_in, p.res_in_bits = a->res_in_bits;
    p.res_out = a->res_out, p.res_out_bits = a->res_out_bits;
    p.out_q = a->out_q, p.out_bits = a->out_bits, p.q_lo = a->q_lo, p.q_hi = a->q_hi, p.mq = a->mq, p.eq = a->eq;
    p.out_acc = a->out_acc, p.out_f32 = a->out_f32, p.fscale = a->fscale, p.ldo = a->ldo, p.n_valid = a->n_valid;
    p.flags = a->flags;
    int slot = -1;
    switch (a->epilogue) {
        case HAWQ_EPI_RAW:
            HAWQ_REQUIRE(a->out_acc, "hawq_conv2d: RAW needs out_acc");
            slot = 0;
            break;
        case HAWQ_EPI_REQUANT:
            HAWQ_REQUIRE(a->out_q && a->m && a->e, "hawq_conv2d: REQUANT needs out_q, m, e");
            HAWQ_REQUIRE(a->out_bits == 8 || a->out_bits == 4, "hawq_conv2d: out_bits must be 4 or 8");
            slot = 1;
            break;
        case HAWQ_EPI_RESIDUAL:
            HAWQ_REQUIRE(a->m && a->e, "hawq_conv2d: RESIDUAL needs m, e");
            HAWQ_REQUIRE(dual || a->res_in, "hawq_conv2d: RESIDUAL needs res_in or a second branch");
            HAWQ_REQUIRE(dual || a->res_in_bits == 16 || a->res_in_bits == 32, "hawq_conv2d: res_in_bits 16/32");
            HAWQ_REQUIRE(!a->res_out || a->res_out_bits == 32 || (a->res_out_bits == 16 && a->flags),
                         "hawq_conv2d: res_out_bits 16 (with flags) or 32");
            HAWQ_REQUIRE(!a->out_q || a->out_bits == 8 || a->out_bits == 4, "hawq_conv2d: out_bits must be 4 or 8");
            HAWQ_REQUIRE(a->res_out || a->out_q, "hawq_conv2d: RESIDUAL needs res_out and/or out_q");
            slot = dual ? 4 : 2;
            break;
        case HAWQ_EPI_DEQUANT:
            HAWQ_REQUIRE(a->out_f32 && a->fscale && a->ldo > 0, "hawq_conv2d: DEQUANT needs out_f32, fscale, ldo");
            slot = 3;
            break;
        default:
            HAWQ_REQUIRE(false, "hawq_conv2d: unknown epilogue %d", a->epilogue);
    }
    int tile = a->tile > 0 ? a->tile - 1 : pick_tile(p.M, p.Cout, dual);
    HAWQ_REQUIRE(tile >= 0 && tile < NUM_TILES, "hawq_conv2d: bad tile id %d", a->tile);
    if (p.Cout % kTiles[tile].BN != 0) tile = 2;
    const TileInfo &ti = kTiles[tile];
    const int grid = ((p.M + ti.BM - 1) / ti.BM) * (p.Cout / ti.BN);
    hipLaunchKernelGGL(ti.fn[slot], dim3(grid), dim3(256), ti.lds, (hipStream_t)stream, p);
    HAWQ_CHECK_HIP(hipGetLastError());
    return 0;
}
