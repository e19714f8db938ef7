// Micro-benchmark (round 6): can the per-launch fixed cost of the stage-3 / stage-4 layers (kernel boundary + ramp +
// tail: 3.6-4.9 us per 3x3 launch, profiles/r05_band2_cin_sweep.txt) be overlapped by TILE-LEVEL DATAFLOW instead of paying one kernel
// boundary per layer?  A chain of L 1x1 "layers"  X_{l+1} = clamp((X_l . W_l^T) >> 8)  (int8 [M][C] activations, int8 [C][C] weights, int8 MFMA,
// 128 px x 64 ch tiles, the same tile routine everywhere) is run three ways:
//   A  one launch per layer (what the engine's hipGraph does today), L dependent launches on one stream;
//   B  ONE persistent launch: workgroups take (layer, tile) items from an atomic ticket counter IN TOPOLOGICAL ORDER and wait on per-pixel-tile
//      counters in global memory written by the producing tiles (the landed[] / done[] protocol of band_v2.hip lifted from LDS to L2:
//      release = stores, workgroup barrier, agent-scope fence, atomic add; acquire = spin on an agent-scope atomic load, fence) - a tile
//      only ever waits for tickets drawn before its own, so the scheme cannot deadlock whatever the number of resident workgroups;
//   C  one persistent launch with a GRID barrier between layers (ticket counter per layer + arrival counter), the structure
//      MI355X_MICROARCH.md's price list puts at 4-7 us per barrier.
// All three produce the same bytes (checked).  Output: us per layer for each mode and shape.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/ubench/tile_dataflow.hip -o /tmp/tile_dataflow && /tmp/tile_dataflow
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

static void check_(hipError_t e, const char *what, int line) {
    if (e != hipSuccess) {
        fprintf(stderr, "%s: %s (line %d)\n", what, hipGetErrorString(e), line);
        exit(1);
    }
}
#define CHECK(x) check_((x), #x, __LINE__)

__device__ __forceinline__ int cperm(int i) { return (((i >> 2) & 1) << 4) + (i & 3) + ((i >> 3) << 2); }

// one 128 px x 64 ch tile: 4 waves, wave w owns pixels 32 w .. 32 w + 31 and all 64 channels (two 32 x 32 MFMA tiles); operands go straight
// from global memory (L2-resident) into registers - the GEMM's own efficiency is not the question here, the launch structure around it is.
// KREP repeats the K loop (same result: the accumulators are reset) to scale the tile's duration.
// COH: the activation rows are read and written with system-scope accesses (`buffer_load / buffer_store ... sc0 sc1`: served at the memory
// side, no line left in this XCD's L2) - per-access coherence instead of whole-L2 write-backs / invalidations; the weights stay cached.
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
template <int KREP, bool COH = false>
__device__ __forceinline__ void tile(const int8_t *__restrict__ x, const int8_t *__restrict__ w, int8_t *__restrict__ y, int M, int C, int m0, int c0) {
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, M * C, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void *)y, 0, M * C, 0x00020000);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, h = lane >> 5;
    int m = m0 + wave * 32 + l31;
    const bool live = m < M;
    m = live ? m : M - 1;
    v16i acc[2];
    for (int rep = 0; rep < KREP; ++rep) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][r] = 0;
        for (int k = 0; k < C; k += 32) {
            v4i b;
            if constexpr (COH) b = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rx, m * C + k + h * 16, 0, 17));
            else b = *reinterpret_cast<const v4i *>(x + (size_t)m * C + k + h * 16);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const v4i a = *reinterpret_cast<const v4i *>(w + (size_t)(c0 + c * 32 + cperm(l31)) * C + k + h * 16);
                acc[c] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[c], 0, 0, 0);
            }
        }
    }
    if (!live) return;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        int q[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            int v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int t = acc[c][4 * g + j] >> 8;
                v[j] = t < -127 ? -127 : (t > 127 ? 127 : t);
            }
            q[g] = (v[0] & 0xff) | ((v[1] & 0xff) << 8) | ((v[2] & 0xff) << 16) | ((v[3] & 0xff) << 24);
        }
        const v4i o = {q[0], q[1], q[2], q[3]};
        if constexpr (COH) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, o), ry, m * C + c0 + c * 32 + h * 16, 0, 17);
        else *reinterpret_cast<v4i *>(y + (size_t)m * C + c0 + c * 32 + h * 16) = o;
    }
}

// ---- A: one launch per layer
template <int KREP>
__global__ __launch_bounds__(256) void layer_kernel(const int8_t *x, const int8_t *w, int8_t *y, int M, int C) {
    const int tiles_c = C >> 6;
    tile<KREP>(x, w, y, M, C, (blockIdx.x / tiles_c) * 128, (blockIdx.x % tiles_c) << 6);
}

struct Chain {
    int8_t *act[2];       // ping-pong activations (layer l reads act[l & 1], writes act[(l + 1) & 1]); layer 0 reads x0
    const int8_t *x0;
    const int8_t *w;      // [L][C][C]
    int *ticket;          // B: one counter; C: [L] counters + [L] arrival counters behind it
    int *done;            // B: [L][tiles_m]
    int M, C, L;
};

// ---- B: persistent, tile-level dataflow
// Control flow is kept WAVE-UNIFORM on purpose: every branch around a barrier is a scalar branch (readfirstlane'd values), the ticket is
// drawn by all lanes of wave 0 together (lane 0 adds 1, the others 0) and every wave polls the flag word itself.  The first version drew the
// ticket under `if (threadIdx.x == 0)`: the structurizer split the item loop around that lane-0 block and the waves of a workgroup then
// executed different numbers of s_barrier - the kernel hung.
template <int KREP, bool COH>
__global__ __launch_bounds__(256) void dataflow_kernel(Chain p) {
    __shared__ int s_item;
    const int tiles_c = p.C >> 6, tiles_m = (p.M + 127) >> 7, per_layer = tiles_c * tiles_m, total = per_layer * p.L;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    int *abort_flag = p.ticket + 2 * p.L;
    for (;;) {
        if (wave == 0) {
            const int old = __hip_atomic_fetch_add(p.ticket, lane == 0 ? 1 : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_item = __builtin_amdgcn_readfirstlane(old);
        }
        __syncthreads();
        const int item = __builtin_amdgcn_readfirstlane(s_item);
        if (item >= total) return;
        const int l = item / per_layer, t = item % per_layer, tm = t / tiles_c, tc = t % tiles_c;
        if (l > 0) {
            const int *f = p.done + (size_t)(l - 1) * tiles_m + tm;
            int spins = 0;   // (bounded: a protocol bug must end as a MISMATCH line, not as a hung GPU)
            // relaxed polls (an acquire load would invalidate this XCD's L2 on EVERY poll); ONE acquire fence behind the loop
            while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < tiles_c) {
                if (++spins > (1 << 12) || __builtin_amdgcn_readfirstlane(__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                    __hip_atomic_store(abort_flag, 1 + item, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                __builtin_amdgcn_s_sleep(4);
            }
            if constexpr (!COH) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // buffer_inv sc1: this XCD's L2 forgets everything
        }
        const int8_t *x = l == 0 ? p.x0 : p.act[l & 1];
        tile<KREP, COH>(x, p.w + (size_t)l * p.C * p.C, p.act[(l + 1) & 1], p.M, p.C, tm * 128, tc << 6);
        if constexpr (COH) {
            // the sc0 sc1 stores are write-through: once vmcnt says they are done they are at the memory side; workgroup-scope ordering only
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (wave == 0) __hip_atomic_fetch_add(p.done + (size_t)l * tiles_m + tm, lane == 0 ? 1 : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // buffer_wbl2 sc1: this XCD's L2 writes back every dirty line
            __syncthreads();   // every wave's stores are released; (also: s_item may be overwritten)
            if (wave == 0) __hip_atomic_fetch_add(p.done + (size_t)l * tiles_m + tm, lane == 0 ? 1 : 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
// NOTE on B's buffers: layer l + 1 overwrites the buffer layer l read (ping-pong).  A tile of layer l + 1 for pixel tile tm may run while a
// tile of layer l for ANOTHER pixel tile is still reading act[l & 1] - but it writes rows of pixel tile tm only, and every layer-l reader of
// those rows (the tiles (l, tm, *)) has finished before done[l][tm] reaches tiles_c.  Rows are private to a pixel tile: no hazard.

// ---- C: persistent with a grid barrier per layer
template <int KREP>
__global__ __launch_bounds__(256) void gridbar_kernel(Chain p) {
    __shared__ int s_item;
    const int tiles_c = p.C >> 6, tiles_m = (p.M + 127) >> 7, per_layer = tiles_c * tiles_m;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    int *abort_flag = p.ticket + 2 * p.L;
    for (int l = 0; l < p.L; ++l) {
        const int8_t *x = l == 0 ? p.x0 : p.act[l & 1];
        for (;;) {
            if (wave == 0) {
                const int old = __hip_atomic_fetch_add(p.ticket + l, lane == 0 ? 1 : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_item = __builtin_amdgcn_readfirstlane(old);
            }
            __syncthreads();
            const int t = __builtin_amdgcn_readfirstlane(s_item);
            __syncthreads();
            if (t >= per_layer) break;
            tile<KREP>(x, p.w + (size_t)l * p.C * p.C, p.act[(l + 1) & 1], p.M, p.C, (t / tiles_c) * 128, (t % tiles_c) << 6);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        int *arr = p.ticket + p.L + l;
        if (wave == 0) __hip_atomic_fetch_add(arr, lane == 0 ? 1 : 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;   // (bounded: workgroups that are not co-resident would otherwise wait for ever)
        while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(arr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < (int)gridDim.x) {
            if (++spins > (1 << 12) || __builtin_amdgcn_readfirstlane(__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                __hip_atomic_store(abort_flag, 1000000 + l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
            __builtin_amdgcn_s_sleep(4);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
}

template <int KREP>
static void run_shape(int M, int C, int L, int resident_per_cu) {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int tiles_c = C >> 6, tiles_m = (M + 127) >> 7, per_layer = tiles_c * tiles_m;
    std::vector<int8_t> hx((size_t)M * C), hw((size_t)L * C * C);
    srand(1);
    for (auto &v : hx) v = (int8_t)(rand() % 255 - 127);
    for (auto &v : hw) v = (int8_t)(rand() % 15 - 7);
    int8_t *x0, *w, *act[2], *ref;
    int *ctr;
    CHECK(hipMalloc(&x0, hx.size())), CHECK(hipMalloc(&w, hw.size()));
    CHECK(hipMalloc(&act[0], hx.size())), CHECK(hipMalloc(&act[1], hx.size())), CHECK(hipMalloc(&ref, hx.size()));
    const size_t nctr = 2 * L + 1 + (size_t)L * tiles_m;
    CHECK(hipMalloc(&ctr, nctr * sizeof(int)));
    CHECK(hipMemcpy(x0, hx.data(), hx.size(), hipMemcpyHostToDevice)), CHECK(hipMemcpy(w, hw.data(), hw.size(), hipMemcpyHostToDevice));
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)), CHECK(hipEventCreate(&e1));
    Chain p;
    p.act[0] = act[0], p.act[1] = act[1], p.x0 = x0, p.w = w, p.ticket = ctr, p.done = ctr + 2 * L + 1, p.M = M, p.C = C, p.L = L;
    const int reps = 20;
    float ms[4] = {0, 0, 0, 0};
    auto mode_a = [&] {
        for (int l = 0; l < L; ++l)
            hipLaunchKernelGGL(layer_kernel<KREP>, dim3(per_layer), dim3(256), 0, st, l == 0 ? x0 : act[l & 1], w + (size_t)l * C * C, act[(l + 1) & 1], M, C);
    };
    // A as a captured graph (what the engine replays)
    hipGraph_t g;
    hipGraphExec_t ge;
    CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    mode_a();
    CHECK(hipStreamEndCapture(st, &g));
    CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int r = 0; r < 3; ++r) CHECK(hipGraphLaunch(ge, st));
    CHECK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r) CHECK(hipGraphLaunch(ge, st));
    CHECK(hipEventRecord(e1, st));
    CHECK(hipStreamSynchronize(st));
    CHECK(hipEventElapsedTime(&ms[0], e0, e1));
    fprintf(stderr, "[M %d C %d res %d] A done\n", M, C, resident_per_cu);
    CHECK(hipMemcpy(ref, act[L & 1], hx.size(), hipMemcpyDeviceToDevice));
    std::vector<int8_t> href(hx.size()), hout(hx.size());
    CHECK(hipMemcpy(href.data(), ref, hx.size(), hipMemcpyDeviceToHost));
    const int grid = cus * resident_per_cu;
    bool same[3] = {true, true, true};
    const char *only = getenv("TD_MODES");   // e.g. "1" / "2" / "13"
    for (int mode = 1; mode <= 3; ++mode) {
        if (only && !strchr(only, '0' + mode)) continue;
        for (int r = -3; r < reps; ++r) {
            if (r == 0) CHECK(hipEventRecord(e0, st));
            CHECK(hipMemsetAsync(ctr, 0, nctr * sizeof(int), st));
            if (mode == 1) hipLaunchKernelGGL((dataflow_kernel<KREP, false>), dim3(grid), dim3(256), 0, st, p);
            else if (mode == 2) hipLaunchKernelGGL(gridbar_kernel<KREP>, dim3(grid), dim3(256), 0, st, p);
            else hipLaunchKernelGGL((dataflow_kernel<KREP, true>), dim3(grid), dim3(256), 0, st, p);
        }
        CHECK(hipEventRecord(e1, st));
        CHECK(hipStreamSynchronize(st));
        CHECK(hipEventElapsedTime(&ms[mode], e0, e1));
        {
            std::vector<int> hc(nctr);
            CHECK(hipMemcpy(hc.data(), ctr, nctr * sizeof(int), hipMemcpyDeviceToHost));
            long long dsum = 0;
            for (size_t i = 2 * L + 1; i < nctr; ++i) dsum += hc[i];
            fprintf(stderr, "  mode %d done: %.3f ms per chain; ticket[0] %d abort %d sum(done) %lld (expected %d)\n", mode, ms[mode] / reps, hc[0], hc[2 * L], dsum, mode != 2 ? per_layer * L : 0);
        }
        CHECK(hipMemcpy(hout.data(), act[L & 1], hx.size(), hipMemcpyDeviceToHost));
        same[mode - 1] = memcmp(hout.data(), href.data(), hx.size()) == 0;
    }
    // cost of the counter reset launch alone (it is inside B's and C's timed region)
    float ms_set;
    CHECK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r) CHECK(hipMemsetAsync(ctr, 0, nctr * sizeof(int), st));
    CHECK(hipEventRecord(e1, st));
    CHECK(hipStreamSynchronize(st));
    CHECK(hipEventElapsedTime(&ms_set, e0, e1));
    printf("M %6d C %4d L %2d  tiles/layer %4d  grid %4d | A launches %7.2f us/layer | B dataflow, L2 fences %7.2f (%s) | C grid barrier %7.2f (%s) | D dataflow, sc0 sc1 accesses %7.2f (%s) | memset %5.2f us per chain\n",
           M, C, L, per_layer, grid, ms[0] / reps / L * 1e3, (ms[1] - ms_set) / reps / L * 1e3, same[0] ? "same bytes" : "MISMATCH",
           (ms[2] - ms_set) / reps / L * 1e3, same[1] ? "same bytes" : "MISMATCH", (ms[3] - ms_set) / reps / L * 1e3, same[2] ? "same bytes" : "MISMATCH", ms_set / reps * 1e3);
    CHECK(hipGraphExecDestroy(ge)), CHECK(hipGraphDestroy(g));
    CHECK(hipFree(x0)), CHECK(hipFree(w)), CHECK(hipFree(act[0])), CHECK(hipFree(act[1])), CHECK(hipFree(ref)), CHECK(hipFree(ctr));
    CHECK(hipStreamDestroy(st));
}

int main(int argc, char **argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    if (argc >= 5) {   // one shape: M C L workgroups-per-CU [krep]   (debugging / sweeps)
        const int M = atoi(argv[1]), C = atoi(argv[2]), L = atoi(argv[3]), res = atoi(argv[4]), krep = argc > 5 ? atoi(argv[5]) : 1;
        (void)krep;
        run_shape<1>(M, C, L, res);
        return 0;
    }
    // stage-3-like (14 x 14 x 64 images, C = 256), stage-4-like (7 x 7 x 64 images, C = 512), and both at 128 images; 12 layers
    for (int res = 1; res <= 4; res *= 2) {
        run_shape<1>(12544, 256, 12, res);
        run_shape<1>(3136, 512, 12, res);
        run_shape<1>(25088, 256, 12, res);
    }
    return 0;
}
