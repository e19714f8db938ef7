// Shared device helpers for libhawq_mi355 (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hawq_mi355.h"

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v16i __attribute__((ext_vector_type(16)));

// ---- error plumbing (host) ---------------------------------------------------------------
void hawq_set_error(const char *fmt, ...);
#define HAWQ_CHECK_HIP(expr)                                                              \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess) {                                                           \
            hawq_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                           __LINE__);                                                     \
            return 1;                                                                     \
        }                                                                                 \
    } while (0)
#define HAWQ_REQUIRE(cond, ...)       \
    do {                              \
        if (!(cond)) {                \
            hawq_set_error(__VA_ARGS__); \
            return 2;                 \
        }                             \
    } while (0)

// ---- timing ablations / cycle stamps (probe builds only) ------------------------------------
// HAWQ_DBG=<bits> makes kernels skip loads / MFMAs / epilogues / stores (results wrong on purpose) or stamp
// s_memtime phases.  That code exists only in a library built with -DHAWQ_ABLATE (`make ABLATE=1` ->
// lib/libhawq_mi355_ablate.so, selected with HAWQ_LIB=...); the shipped library never reads the variable and
// the bit tests below fold to constants.
#ifdef HAWQ_ABLATE
#include <stdlib.h>
#define HAWQ_DBG_ENV() (getenv("HAWQ_DBG") ? atoi(getenv("HAWQ_DBG")) : 0)
#define HAWQ_DBG_BIT(v, bits) ((v) & (bits))
#else
#define HAWQ_DBG_ENV() 0
#define HAWQ_DBG_BIT(v, bits) 0
#endif

// ---- dyadic requantisation ---------------------------------------------------------------
// Table entry = (m, ek) with ek = e | k << 8:   q = round_half_even(((v << k) * m) / 2^e),
// 0 <= m < 2^31, 1 <= e <= 62, k >= 0 small (host guarantees |v << k| < 2^31).  The pre-shift k
// lets the host lift every e to >= 33 so that the conv epilogues can use the high-word fast path
// below; mathematically (v*2^k*m)/2^(e_orig+k) is the same rational, so rounding is unchanged.
// Restates fixedpoint_fn's float64 emulation (quant_utils.py:404-408) in exact integers:
// half-up via +2^(e-1), then an exact tie (all e low bits of the biased product zero) is pulled
// back to the even neighbour.
__device__ __forceinline__ int32_t dyadic_rne(int32_t v, int32_t m, int32_t ek) {
    const int e = ek & 0xff, k = ek >> 8;
    const long long half = 1ll << (e - 1);
    const long long t = (long long)(v << k) * (long long)m + half;
    long long f = t >> e;
    const long long mask = (1ll << e) - 1;
    if ((t & mask) == 0) f &= ~1ll;
    return (int32_t)f;
}

// Fast path used by the conv epilogues when the HOST has proved, for every table entry, that
//   (a) e >= 33 (s = e - 32 in [1,30]), |v << k| < 2^31, and
//   (b) an exact tie cannot occur: a tie needs 2^(e-1) | (v << k) * m, i.e. tz(v) >= e-1-k-tz(m);
//       with |v| < 2^vbits known that is impossible when tz(m) <= e - 1 - k - vbits
//       (hawq_amd.quant_utils.tables_are_fast).
// Then round-half-even == round-half-up == floor((v*m + 2^(e-1)) / 2^e), the bias 2^(e-1) lives
// entirely in the high word and can be the 64-bit addend of ONE v_mad_i64_i32, and the quotient is
// an arithmetic shift of the high word:  2 VALU instructions.
// (callers extract s from a table word with `& 31`, not `& 0xff`: s = e - 32 is in [1, 30], and a 5-bit mask folds into the shift
// instruction - the hardware ignores the upper bits of a shift amount - while `& 0xff` is a VALU instruction per output)
struct DyNt {
    int m, s, k;
    long long add;  // 2^(e-1) = (1 << (s-1)) << 32   (+ (bias << k) * m when the bias is folded in)
};
__device__ __forceinline__ DyNt dynt_prepare(int m, int ek) {
    DyNt c;
    c.m = m;
    c.s = (ek & 0xff) - 32;
    c.k = ek >> 8;
    c.add = (long long)(1u << (c.s - 1)) << 32;
    return c;
}
__device__ __forceinline__ int32_t dyadic_nt(int32_t v, const DyNt &c) {
    const long long t = (long long)(v << c.k) * (long long)c.m + c.add;
    return (int)(t >> 32) >> c.s;
}
// K0: the host additionally guarantees k == 0 for this table (true for every conv table whose e is >= 33 without
// lifting, i.e. whenever the requant ratio is below 2^-2): one VALU instruction less per element
template <bool K0>
__device__ __forceinline__ int32_t dyadic_nt_k(int32_t v, const DyNt &c) {
    const long long t = (long long)(K0 ? v : (v << c.k)) * (long long)c.m + c.add;
    return (int)(t >> 32) >> c.s;
}
// TIE-aware form for tables whose tie-freedom the host could NOT prove (a few channels with many trailing zeros in
// m and a small e): half-up first, then the exact-tie correction of round-half-even.  x*m = (2j+1)*2^(e-1) makes
// t = x*m + 2^(e-1) a multiple of 2^e and half-up returns j+1; RNE wants the even one of {j, j+1}, i.e. one less
// when the half-up result is odd (holds for negative values too: arithmetic shifts floor).
__device__ __forceinline__ int32_t dyadic_tie(int32_t v, const DyNt &c) {
    const long long t = (long long)(v << c.k) * (long long)c.m + c.add;
    const int hi = (int)(t >> 32);
    const int q = hi >> c.s;
    const bool tie = (unsigned)t == 0u && (hi & ((1 << c.s) - 1)) == 0;
    return q - (tie ? (q & 1) : 0);
}
// MODE 0: tie-free table with per-entry pre-shift, 1: tie-free and k == 0, 2: ties possible (exact RNE)
template <int MODE>
__device__ __forceinline__ int32_t dyadic_mode(int32_t v, const DyNt &c) {
    if (MODE == 2) return dyadic_tie(v, c);
    return dyadic_nt_k<MODE == 1>(v, c);
}
// Round 6 (profiles/r06_epilogue_census.md): the scalar identity table of a pass-through unit - the ratio of two residual scales, 0.25 <= r < 2 -
// arrives lifted to (e = 33, k = 33 - e_orig >= 1).  The same rational is ((v << (k - 1)) * m) / 2^32: with the rounding constant 2^31 in the
// addend the quotient IS the high word - no shift behind the v_mad_i64_i32 (one VALU instruction per residual output less).  Kernels
// instantiated for it (`SC0`) are only launched when the host table has that form (ids0_form below); identical results by construction:
// floor((v 2^(k-1) m + 2^31) / 2^32) == floor((v 2^k m + 2^32) / 2^33).
__host__ __device__ __forceinline__ bool ids0_form(int ek) { return (ek & 0xff) == 33 && (ek >> 8) >= 1; }
struct DyS0 {
    int m, k;
    long long add;   // 2^31
};
__device__ __forceinline__ DyS0 dys0_prepare(int m, int ek) {
    DyS0 c;
    c.m = m, c.k = (ek >> 8) - 1, c.add = 1ll << 31;
    return c;
}
__device__ __forceinline__ int32_t dyadic_s0(int32_t v, const DyS0 &c) { return (int32_t)(((long long)(v << c.k) * (long long)c.m + c.add) >> 32); }

// clamp(v, lo, hi) for lo <= hi in one instruction
__device__ __forceinline__ int32_t med3i(int32_t v, int32_t lo, int32_t hi) {
    int32_t r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(v), "v"(lo), "v"(hi));
    return r;
}

__device__ __forceinline__ int32_t clampi(int32_t v, int32_t lo, int32_t hi) {
    return v < lo ? lo : (v > hi ? hi : v);
}

// pack 4 small ints (already clamped to int8 range) into one dword, element 0 in byte 0
__device__ __forceinline__ uint32_t pack4_i8(int a, int b, int c, int d) {
    return (uint32_t)(a & 0xff) | ((uint32_t)(b & 0xff) << 8) | ((uint32_t)(c & 0xff) << 16) |
           ((uint32_t)(d & 0xff) << 24);
}
// hawq4: 8 channels c0..c7 (values 0..15) -> one dword (see include/hawq_mi355.h)
__device__ __forceinline__ uint32_t pack8_u4(const int *q) {
    return pack4_i8(q[0], q[1], q[2], q[3]) | (pack4_i8(q[4], q[5], q[6], q[7]) << 4);
}

// ---- shared device helpers of the conv kernels (conv_igemm.hip, fused_er.hip) ---------------------------
// MFMA C/D row i of a 32x32 tile lives in (reg, half) with i = (reg&3) + 8*(reg>>2) + 4*half.
// Reading weight row cperm(i) as MFMA row i makes lane-half h own channels 16h..16h+15.
__device__ __forceinline__ int cperm(int i) { return (((i >> 2) & 1) << 4) + (i & 3) + ((i >> 3) << 2); }

__device__ __forceinline__ int lds_off(int row, int slot) { return row * 64 + ((slot ^ ((row >> 2) & 3)) << 4); }

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// 4 ints already clamped to the int8 range -> one dword (byte 0 = first): 2x v_cvt_pk_i16_i32 + v_perm
__device__ __forceinline__ int pack4_fast(int a, int b, int c, int d) {
    typedef short s2 __attribute__((ext_vector_type(2)));
    const s2 lo = __builtin_amdgcn_cvt_pk_i16(a, b), hi = __builtin_amdgcn_cvt_pk_i16(c, d);
    return (int)__builtin_amdgcn_perm(__builtin_bit_cast(unsigned, hi), __builtin_bit_cast(unsigned, lo), 0x06040200u);
}
// 4 NON-NEGATIVE ints -> min(., hi) -> one dword of bytes: the upper clamp runs on the packed int16 pairs (v_pk_min_i16: one
// instruction per two values; v_cvt_pk_i16_i32 saturates at 32767 >= hi, so min(sat16(x), hi) == min(x, hi) for x >= 0, 0 <= hi <= 32767)
__device__ __forceinline__ int pack4_min(int a, int b, int c, int d, int hi2 /* hi | hi << 16 */) {
    typedef short s2 __attribute__((ext_vector_type(2)));
    const s2 h2 = __builtin_bit_cast(s2, hi2);
    const s2 lo = __builtin_elementwise_min(__builtin_amdgcn_cvt_pk_i16(a, b), h2), hi = __builtin_elementwise_min(__builtin_amdgcn_cvt_pk_i16(c, d), h2);
    return (int)__builtin_amdgcn_perm(__builtin_bit_cast(unsigned, hi), __builtin_bit_cast(unsigned, lo), 0x06040200u);
}
// two non-negative ints -> saturating uint16 pair (v_cvt_pk_u16_u32)
__device__ __forceinline__ int pack2_u16_sat(int a, int b) {
    typedef unsigned short u2 __attribute__((ext_vector_type(2)));
    const u2 r = __builtin_amdgcn_cvt_pk_u16((unsigned)a, (unsigned)b);
    return __builtin_bit_cast(int, r);
}

// Hand-issued LDS fragment reads.  hipcc treats every LDS-DMA instruction as a pending FLAT access and from then
// on only ever emits `s_waitcnt lgkmcnt(0)` (measured on a toy kernel: lgkmcnt(2) without, lgkmcnt(0) with one
// global_load_lds in the loop), so compiler-visible ds_reads cannot be software-pipelined in these kernels.
// Reads issued through lds_read16 are invisible to its bookkeeping: the caller waits with wait_lgkm<N>() (N =
// reads allowed to stay in flight; LDS returns in order, and a scalar load that happens to be in flight can
// only make the wait stricter, never too lax, as long as N counts LDS reads issued AFTER the ones needed) and
// then passes every fragment through pin() before its first use.
template <int OFF>
__device__ __forceinline__ v4i lds_read16(unsigned addr) {
    v4i r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
    return r;
}
template <int N>
__device__ __forceinline__ void wait_lgkm() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void pin(v4i &f) { asm volatile("" : "+v"(f)); }
__device__ __forceinline__ unsigned lds_addr(const char *p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const char *)p;
}
