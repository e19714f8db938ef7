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

// ---- dyadic requantisation ---------------------------------------------------------------
// q = round_half_even(v * m / 2^e), exact in 64-bit integers (|v| < 2^31, 0 <= m < 2^31,
// 1 <= e <= 62).  Restates fixedpoint_fn's float64 emulation (quant_utils.py:404-408):
// half-up via the +2^(e-1) bias folded into the 64-bit multiply-add, then the exact-tie case
// (all e low bits of the biased product zero) is pulled back to the even neighbour.
__device__ __forceinline__ int32_t dyadic_rne(int32_t v, int32_t m, int32_t e) {
    const long long half = 1ll << (e - 1);
    const long long t = (long long)v * (long long)m + half;
    long long f = t >> e;
    const long long mask = (1ll << e) - 1;
    if ((t & mask) == 0) f &= ~1ll;
    return (int32_t)f;
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
