// gather_bench.hip -- random-gather microbenchmark for MI355X (gfx950).
//
// Two uses ("next round" item 1c):
//   * what HBM delivers for the access pattern of the walk kernels (scattered S-byte reads, S = 4..64, over a
//     buffer far larger than the 256 MiB Infinity Cache), independent loads and dependent chains;
//   * calibration of rocprofv3's FETCH_SIZE for that pattern: the number of accesses is known exactly, so
//     FETCH_SIZE / accesses = bytes the memory side moves per scattered access.
// Build:  hipcc --offload-arch=gfx950 -O3 -o gather_bench tools/gather_bench.hip
// Run:    ./gather_bench [GiB=8] [iters=64]            (prints one JSON line per variant)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}

template <int S> struct Vec;
template <> struct Vec<4> { typedef uint32_t T; static __device__ uint32_t fold(uint32_t v) { return v; } };
template <> struct Vec<8> { typedef uint2 T; static __device__ uint32_t fold(uint2 v) { return v.x ^ v.y; } };
template <> struct Vec<16> { typedef uint4 T; static __device__ uint32_t fold(uint4 v) { return v.x ^ v.y ^ v.z ^ v.w; } };
struct alignas(32) U8 { uint4 a, b; };
template <> struct Vec<32> { typedef U8 T; static __device__ uint32_t fold(U8 v) { return v.a.x ^ v.a.w ^ v.b.x ^ v.b.w; } };
struct alignas(64) U16 { uint4 a, b, c, d; };
template <> struct Vec<64> { typedef U16 T; static __device__ uint32_t fold(U16 v) { return v.a.x ^ v.b.x ^ v.c.x ^ v.d.w; } };

// independent gathers: every lane issues `iters` loads whose addresses do not depend on loaded data
template <int S>
__global__ void __launch_bounds__(256) gather_kernel(const typename Vec<S>::T *__restrict__ buf, uint64_t n_elems,
                                                     int iters, uint32_t *out) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
#pragma unroll 4
    for (int it = 0; it < iters; it++) {
        const uint64_t idx = mix(tid * 0x9E3779B97F4A7C15ull + (uint64_t)it) % n_elems;
        acc ^= Vec<S>::fold(buf[idx]);
    }
    if (acc == 0x12345678u) out[0] = acc;   // keep the loads alive
}

// dependent chains: the next address is a function of the loaded value (one outstanding load per lane)
template <int S>
__global__ void __launch_bounds__(256) chase_kernel(const typename Vec<S>::T *__restrict__ buf, uint64_t n_elems,
                                                    int iters, uint32_t *out) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t idx = mix(tid) % n_elems;
    uint32_t acc = 0;
    for (int it = 0; it < iters; it++) {
        const uint32_t v = Vec<S>::fold(buf[idx]);
        acc ^= v;
        idx = mix(idx * 0x9E3779B97F4A7C15ull + v + (uint64_t)it) % n_elems;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

__global__ void fill_kernel(uint32_t *buf, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        buf[i] = (uint32_t)mix(i);
}

template <int S> static void run(const void *buf, uint64_t bytes, int iters, int waves_per_simd, uint32_t *d_out, int n_cu) {
    const uint64_t n_elems = bytes / S;
    const int blocks = n_cu * waves_per_simd;   // 256-thread blocks: 4 waves each = one wave per SIMD
    for (int dep = 0; dep < 2; dep++) {
        hipEvent_t a, b;
        CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            CHECK(hipEventRecord(a));
            if (dep) hipLaunchKernelGGL(chase_kernel<S>, dim3(blocks), dim3(256), 0, 0, (const typename Vec<S>::T *)buf, n_elems, iters, d_out);
            else hipLaunchKernelGGL(gather_kernel<S>, dim3(blocks), dim3(256), 0, 0, (const typename Vec<S>::T *)buf, n_elems, iters, d_out);
            CHECK(hipEventRecord(b));
            CHECK(hipEventSynchronize(b));
            float ms; CHECK(hipEventElapsedTime(&ms, a, b));
            if (ms < best) best = ms;
        }
        const double acc = (double)blocks * 256.0 * iters;
        printf("{\"bench\": \"%s\", \"bytes_per_access\": %d, \"waves_per_simd\": %d, \"accesses\": %.0f, \"ms\": %.3f, "
               "\"G_accesses_per_s\": %.2f, \"useful_GBps\": %.1f, \"sector64_GBps\": %.1f}\n",
               dep ? "chase" : "gather", S, waves_per_simd, acc, best, acc / best / 1e6, acc * S / best / 1e6,
               acc * (S > 64 ? S : 64) / best / 1e6);
        fflush(stdout);
    }
}

int main(int argc, char **argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 8.0;
    const int iters = argc > 2 ? atoi(argv[2]) : 64;
    const uint64_t bytes = (uint64_t)(gib * (1ull << 30)) & ~4095ull;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    void *buf; uint32_t *d_out;
    CHECK(hipMalloc(&buf, bytes));
    CHECK(hipMalloc((void **)&d_out, 4));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, (uint32_t *)buf, bytes / 4);
    CHECK(hipDeviceSynchronize());
    for (int w : {8, 4}) {
        run<4>(buf, bytes, iters, w, d_out, prop.multiProcessorCount);
        run<8>(buf, bytes, iters, w, d_out, prop.multiProcessorCount);
        run<16>(buf, bytes, iters, w, d_out, prop.multiProcessorCount);
        run<32>(buf, bytes, iters, w, d_out, prop.multiProcessorCount);
        run<64>(buf, bytes, iters, w, d_out, prop.multiProcessorCount);
    }
    return 0;
}
