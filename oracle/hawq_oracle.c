/*
 * hawq_oracle.c - TEST INFRASTRUCTURE ONLY (never linked into or called by the product).
 *
 * Plain-C, pure-integer CPU restatement of the arithmetic of HAWQ's frozen quantized
 * ResNet forward.  Layout is the reference's own (NCHW activations, OIHW weights) so that
 * this checker shares no layout/packing code with the HIP path it checks.
 *
 * Reference lines restated (paths relative to the HAWQ tree):
 *   hq_frexp_me        utils/quantization_utils/quant_utils.py:188-213  (batch_frexp)
 *   hq_dyadic          utils/quantization_utils/quant_utils.py:390-413, 416-456 (fixedpoint_fn)
 *   hq_quantize_f32    utils/quantization_utils/quant_utils.py:73-97, 237-258, 281-308
 *   hq_conv2d_nchw     utils/quantization_utils/quant_modules.py:489-494 (F.conv2d on integers)
 *   hq_linear          utils/quantization_utils/quant_modules.py:125-130
 *   hq_maxpool_nchw    utils/models/q_resnet.py:93,119 (nn.MaxPool2d(3,2,1))
 *   hq_avgpool_floor   utils/quantization_utils/quant_modules.py:596-600 + quant_utils.py:334-337
 *
 * Parity pin: tests/test_oracle_vs_golden.py checks every function against fixtures that
 * tests/golden/make_golden.py produced by running the live reference in the build container.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

int hq_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* batch_frexp: r = mant * 2^ex, mant in [0.5,1);  m = ROUND_HALF_UP(mant * 2^31), e = 31 - ex.
 * mant*2^31 is exact in binary64 and has <= 22 fractional bits, so floor(v + 0.5) is exact
 * and equals Decimal(v).quantize(1, ROUND_HALF_UP) for v >= 0. */
void hq_frexp_me(const double *r, int64_t n, int64_t *m, int32_t *e) {
    for (int64_t i = 0; i < n; ++i) {
        int ex;
        double mant = frexp(r[i], &ex);
        double v = mant * 2147483648.0;
        m[i] = (int64_t)(v >= 0 ? floor(v + 0.5) : -floor(-v + 0.5));
        e[i] = 31 - ex;
    }
}

/* round_half_even((acc * m) / 2^e) in exact integer arithmetic (torch.round on an exactly
 * representable quotient).  e may be <= 0 (then it is an exact left shift). */
static inline int64_t dyadic1(int64_t acc, int64_t m, int32_t e) {
    __int128 p = (__int128)acc * (__int128)m;
    if (e <= 0) return (int64_t)(p << (-e));
    __int128 f = p >> e; /* floor */
    __int128 rem = p - (f << e);
    __int128 half = (__int128)1 << (e - 1);
    if (rem > half || (rem == half && (f & 1))) f += 1;
    return (int64_t)f;
}

/* acc: [N][C][HW] int64.  m/e have nch entries (1 = per tensor, C = per channel).
 * out = dyadic(acc) optionally clamped to [lo,hi]. */
void hq_dyadic_nchw(const int64_t *acc, int64_t N, int64_t C, int64_t HW, const int64_t *m,
                    const int32_t *e, int64_t nch, int do_clamp, int64_t lo, int64_t hi,
                    int64_t *out) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int64_t n = 0; n < N; ++n)
        for (int64_t c = 0; c < C; ++c) {
            int64_t mm = m[nch == 1 ? 0 : c];
            int32_t ee = e[nch == 1 ? 0 : c];
            const int64_t *a = acc + (n * C + c) * HW;
            int64_t *o = out + (n * C + c) * HW;
            for (int64_t i = 0; i < HW; ++i) {
                int64_t q = dyadic1(a[i], mm, ee);
                if (do_clamp) q = q < lo ? lo : (q > hi ? hi : q);
                o[i] = q;
            }
        }
}

/* linear_quantize + clamp: q = clamp(rint(fl(inv_scale * x)), lo, hi); inv_scale = fl(1/S) is
 * formed by the caller in binary32 exactly as `1. / scale` does. rintf = round-half-even. */
void hq_quantize_f32(const float *x, int64_t n, float inv_scale, float lo, float hi, int64_t *q) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        volatile float t = inv_scale * x[i]; /* one binary32 rounding, no contraction */
        float r = rintf(t);
        r = r < lo ? lo : (r > hi ? hi : r);
        q[i] = (int64_t)r;
    }
}

/* Exact integer convolution, zero padding, dilation 1, groups 1.
 * x [N][Ci][H][W] int8-range values held in int16, w [Co][Ci][KH][KW] int8, bias [Co] int64.
 * out [N][Co][Ho][Wo] int64 (never overflows: |sum| < 2^15*2^7*K). */
void hq_conv2d_nchw(const int16_t *x, int64_t N, int64_t Ci, int64_t H, int64_t W, const int8_t *w,
                    const int64_t *bias, int64_t Co, int64_t KH, int64_t KW, int64_t stride,
                    int64_t pad, int64_t *out) {
    const int64_t Ho = (H + 2 * pad - KH) / stride + 1;
    const int64_t Wo = (W + 2 * pad - KW) / stride + 1;
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
    for (int64_t n = 0; n < N; ++n)
        for (int64_t co = 0; co < Co; ++co) {
            int64_t *o = out + (n * Co + co) * Ho * Wo;
            for (int64_t i = 0; i < Ho * Wo; ++i) o[i] = bias ? bias[co] : 0;
            for (int64_t ci = 0; ci < Ci; ++ci) {
                const int16_t *xp = x + (n * Ci + ci) * H * W;
                const int8_t *wp = w + ((co * Ci + ci) * KH) * KW;
                for (int64_t kh = 0; kh < KH; ++kh)
                    for (int64_t kw = 0; kw < KW; ++kw) {
                        const int64_t wv = wp[kh * KW + kw];
                        if (wv == 0) continue;
                        for (int64_t oy = 0; oy < Ho; ++oy) {
                            const int64_t iy = oy * stride - pad + kh;
                            if (iy < 0 || iy >= H) continue;
                            const int16_t *xr = xp + iy * W;
                            int64_t *orow = o + oy * Wo;
                            /* valid ox range: 0 <= ox*stride - pad + kw < W */
                            int64_t ox0 = 0, ox1 = Wo;
                            while (ox0 < Wo && ox0 * stride - pad + kw < 0) ++ox0;
                            while (ox1 > ox0 && (ox1 - 1) * stride - pad + kw >= W) --ox1;
                            const int64_t base = -pad + kw;
                            for (int64_t ox = ox0; ox < ox1; ++ox)
                                orow[ox] += wv * (int64_t)xr[ox * stride + base];
                        }
                    }
            }
        }
}

/* x [B][K] (int16 holding 8-bit values), w [O][K] int8, bias [O] int64 -> out [B][O] int64 */
void hq_linear(const int16_t *x, int64_t B, int64_t K, const int8_t *w, const int64_t *bias,
               int64_t O, int64_t *out) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int64_t b = 0; b < B; ++b)
        for (int64_t o = 0; o < O; ++o) {
            int64_t s = bias ? bias[o] : 0;
            for (int64_t k = 0; k < K; ++k) s += (int64_t)x[b * K + k] * (int64_t)w[o * K + k];
            out[b * O + o] = s;
        }
}

/* MaxPool2d(k, stride, pad) with -inf padding on int64 planes [NC][H][W]. */
void hq_maxpool_nchw(const int64_t *x, int64_t NC, int64_t H, int64_t W, int64_t k, int64_t stride,
                     int64_t pad, int64_t *out) {
    const int64_t Ho = (H + 2 * pad - k) / stride + 1;
    const int64_t Wo = (W + 2 * pad - k) / stride + 1;
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < NC; ++p)
        for (int64_t oy = 0; oy < Ho; ++oy)
            for (int64_t ox = 0; ox < Wo; ++ox) {
                int64_t best = INT64_MIN;
                for (int64_t ky = 0; ky < k; ++ky)
                    for (int64_t kx = 0; kx < k; ++kx) {
                        int64_t iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
                        if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                        int64_t v = x[(p * H + iy) * W + ix];
                        if (v > best) best = v;
                    }
                out[(p * Ho + oy) * Wo + ox] = best;
            }
}

/* QuantAveragePool2d on integers: trunc(sum/HW + 0.01).  For the post-ReLU (>= 0) inputs
 * of the ResNets this is floor(sum / HW); negative sums follow trunc toward zero with
 * the reference's +0.01 fudge, restated in exact rationals:  trunc((100*sum + HW) / (100*HW)). */
void hq_avgpool_trunc(const int64_t *x, int64_t NC, int64_t HW, int64_t *out) {
    for (int64_t p = 0; p < NC; ++p) {
        int64_t s = 0;
        for (int64_t i = 0; i < HW; ++i) s += x[p * HW + i];
        int64_t num = 100 * s + HW, den = 100 * HW;
        out[p] = num / den; /* C division truncates toward zero */
    }
}
