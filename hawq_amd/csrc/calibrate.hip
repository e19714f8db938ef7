// Range statistics of QuantAct's un-frozen forward (calibration / QAT range tracking) on the MI355X:
//   hawq_minmax_f32     x.data.min(), x.data.max()                                  (quant_modules.py:233-236)
//   hawq_kthvalue_f32   torch.kthvalue(+-x.view(-1), k) of get_percentile_min_max   (quant_utils.py:38-70)
// Both exact: min / max are order-free, and the k-th smallest value is found by a 4-pass radix select on the
// order-preserving integer image of the floats (no sort, no approximation): pass d histograms byte d (most significant
// first) of the keys that match the prefix found so far, a one-wave kernel picks the byte that contains rank k.
// HBM-bound streaming kernels: one read of the tensor per pass, 16 B per lane, grid-stride, per-workgroup LDS histograms
// merged with 256 global atomics per workgroup.
#include "common.h"

namespace {

__device__ __forceinline__ unsigned key_of(float v, bool negate) {
    unsigned u = __float_as_uint(negate ? -v : v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);   // monotone: a < b  <=>  key(a) < key(b)  (-0 < +0; NaNs sort to the ends)
}
__device__ __forceinline__ float value_of(unsigned k, bool negate) {
    const unsigned u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    const float v = __uint_as_float(u);
    return negate ? -v : v;
}

__global__ void minmax_kernel(const float *__restrict__ x, long long n, unsigned *keys) {   // keys[0] = min key, keys[1] = max key
    unsigned lo = 0xffffffffu, hi = 0u;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long n4 = n >> 2;
    const float4 *x4 = reinterpret_cast<const float4 *>(x);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 v = x4[i];
        const unsigned k0 = key_of(v.x, false), k1 = key_of(v.y, false), k2 = key_of(v.z, false), k3 = key_of(v.w, false);
        lo = min(min(lo, k0), min(min(k1, k2), k3));
        hi = max(max(hi, k0), max(max(k1, k2), k3));
    }
    for (long long i = (n4 << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const unsigned k = key_of(x[i], false);
        lo = min(lo, k), hi = max(hi, k);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        lo = min(lo, (unsigned)__shfl_xor((int)lo, off, 64));
        hi = max(hi, (unsigned)__shfl_xor((int)hi, off, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&keys[0], lo);
        atomicMax(&keys[1], hi);
    }
}
__global__ void minmax_init(unsigned *keys) { keys[0] = 0xffffffffu, keys[1] = 0u; }
__global__ void minmax_finish(const unsigned *keys, float *out) { out[0] = value_of(keys[0], false), out[1] = value_of(keys[1], false); }

// state[0] = prefix (the key bytes fixed so far, in place), state[1..2] = remaining rank k (64-bit), hist[256]
__global__ void kth_init(unsigned *state, unsigned *hist, long long k) {
    const int t = threadIdx.x;
    hist[t] = 0;
    if (t == 0) state[0] = 0, state[1] = (unsigned)(k & 0xffffffffll), state[2] = (unsigned)(k >> 32);
}
template <int PASS>   // PASS 0: most significant byte
__global__ void kth_hist(const float *__restrict__ x, long long n, int negate, const unsigned *state, unsigned *hist) {
    __shared__ unsigned lh[256];
    lh[threadIdx.x & 255] = 0;
    __syncthreads();
    constexpr int SH = 24 - 8 * PASS;
    const unsigned prefix = state[0];
    const unsigned pmask = PASS == 0 ? 0u : (0xffffffffu << (SH + 8));
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const unsigned k = key_of(x[i], negate != 0);
        if ((k & pmask) == prefix) atomicAdd(&lh[(k >> SH) & 0xff], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 256 && lh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], lh[threadIdx.x]);
}
template <int PASS>
__global__ void kth_pick(unsigned *state, unsigned *hist, int negate, float *out) {
    __shared__ unsigned long long cum[256];
    const int t = threadIdx.x;
    cum[t] = hist[t];
    __syncthreads();
    if (t == 0) {
        unsigned long long k = ((unsigned long long)state[2] << 32) | state[1], run = 0;
        int d = 255;
        for (int b = 0; b < 256; ++b) {
            if (run + cum[b] >= k) { d = b; break; }
            run += cum[b];
        }
        k -= run;
        const unsigned prefix = state[0] | ((unsigned)d << (24 - 8 * PASS));
        state[0] = prefix, state[1] = (unsigned)(k & 0xffffffffull), state[2] = (unsigned)(k >> 32);
        if (PASS == 3) out[0] = value_of(prefix, negate != 0);
    }
    __syncthreads();
    hist[t] = 0;
}

}  // namespace

extern "C" int hawq_minmax_f32(const float *x, int64_t n, float *out2, void *scratch, void *stream) {
    HAWQ_REQUIRE(x && out2 && scratch && n > 0, "hawq_minmax_f32: null argument or empty tensor");
    HAWQ_REQUIRE(((uintptr_t)x & 15) == 0, "hawq_minmax_f32: x must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    unsigned *keys = (unsigned *)scratch;
    const int grid = (int)std::min<int64_t>(2048, (n / 4 + 255) / 256 + 1);
    hipLaunchKernelGGL(minmax_init, dim3(1), dim3(1), 0, s, keys);
    hipLaunchKernelGGL(minmax_kernel, dim3(grid), dim3(256), 0, s, x, (long long)n, keys);
    hipLaunchKernelGGL(minmax_finish, dim3(1), dim3(1), 0, s, keys, out2);
    HAWQ_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int hawq_kthvalue_f32(const float *x, int64_t n, int64_t k, int32_t negate, float *out, void *scratch, void *stream) {
    HAWQ_REQUIRE(x && out && scratch && n > 0, "hawq_kthvalue_f32: null argument or empty tensor");
    HAWQ_REQUIRE(k >= 1 && k <= n, "hawq_kthvalue_f32: k = %lld outside [1, n = %lld]", (long long)k, (long long)n);
    hipStream_t s = (hipStream_t)stream;
    unsigned *state = (unsigned *)scratch, *hist = state + 4;   // scratch: 4 + 256 uint32
    const int grid = (int)std::min<int64_t>(2048, (n + 255) / 256);
    hipLaunchKernelGGL(kth_init, dim3(1), dim3(256), 0, s, state, hist, (long long)k);
#define HAWQ_KTH_PASS(P)                                                                              \
    hipLaunchKernelGGL(kth_hist<P>, dim3(grid), dim3(256), 0, s, x, (long long)n, negate, state, hist); \
    hipLaunchKernelGGL(kth_pick<P>, dim3(1), dim3(256), 0, s, state, hist, negate, out);
    HAWQ_KTH_PASS(0)
    HAWQ_KTH_PASS(1)
    HAWQ_KTH_PASS(2)
    HAWQ_KTH_PASS(3)
#undef HAWQ_KTH_PASS
    HAWQ_CHECK_HIP(hipGetLastError());
    return 0;
}
