// lane_mem_bench.hip -- what the vector memory path of one MI355X charges for the access shapes of the lane kernel
// (round 5).  Every lane runs a chain of dependent STEPS over a buffer of 64-byte lines far larger than the caches (the edge
// lines of walk_lanes.hip.h); the variants differ in how a step touches its line and what else it loads:
//   A  one 16-byte load of the line's head
//   B  16 + 8 bytes (two instructions, same line)                     -- the record load of the lane kernel today
//   C  four 16-byte loads of the same line by the same lane            -- record + tail (TAILS form: LDS-DMA in the kernel)
//   D  A, then three DEPENDENT 2-byte loads inside the same line       -- bisection of an inline list / of the pivots
//   E  the whole line by a QUAD: four LDS-DMA instructions, lane l moves piece l & 3 of the line of lane 16 k + (l >> 2);
//      the step then reads its line from LDS                           -- one request per line instead of one per piece
//   G  A + one 8-byte load from a per-lane sequential stream           -- the draw
//   H  A, then ONE dependent 2-byte load from a second large array     -- an overflow-list probe
//   I  A, then a dependent QUAD fetch of a 64-byte node of the second array (as E) -- a sector-local list node
//   J  E + draws by quad-fetched sectors (one fetch per 8 steps)       -- everything sector-wise
// Build: hipcc --offload-arch=gfx950 -O3 -o lane_mem_bench tools/lane_mem_bench.hip ;  run: ./lane_mem_bench [GiB=6] [steps=200]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef const __attribute__((address_space(1))) void *glb_ptr_t;

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

__global__ void fill_kernel(uint32_t *buf, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        buf[i] = mix32((uint32_t)i * 2654435761u + (uint32_t)(i >> 32));
}

template <int V, int WAVES>
__global__ void __launch_bounds__(256, WAVES)
bench_kernel(const uint4 *__restrict__ lines, uint32_t n_lines, const uint4 *__restrict__ lists, uint32_t n_list_lines,
             const double *__restrict__ draws, int steps, uint32_t *out) {
    __shared__ uint4 s_line[(V == 4 || V == 9) ? 4 : 1][4][64];      // [wave][piece][lane]: quad-fetched lines land as 64 contiguous bytes per walk
    __shared__ uint4 s_node[V == 8 ? 4 : 1][4][64];
    __shared__ uint4 s_draw[V == 9 ? 4 : 1][4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t idx = mix32((uint32_t)tid * 747796405u + 1u) % n_lines;
    uint32_t acc = 0;
    const double *my_draws = draws + tid * (uint64_t)(steps + 8);
    double dacc = 0.0;
    for (int it = 0; it < steps; it++) {
        uint32_t v = 0;
        if (V == 0) {                                    // A
            const uint4 a = lines[(uint64_t)idx * 4];
            v = a.x ^ a.w;
        } else if (V == 1) {                             // B
            const uint4 a = lines[(uint64_t)idx * 4];
            const uint2 b = *(const uint2 *)(lines + (uint64_t)idx * 4 + 1);
            v = a.x ^ a.w ^ b.y;
        } else if (V == 2) {                             // C
            const uint4 *p = lines + (uint64_t)idx * 4;
            const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
            v = a.x ^ b.y ^ c.z ^ d.w;
        } else if (V == 3) {                             // D
            const uint4 a = lines[(uint64_t)idx * 4];
            const uint16_t *h = (const uint16_t *)(lines + (uint64_t)idx * 4);
            uint32_t o = 8u + (a.x & 7u);
            uint32_t x = h[o];
            o = 8u + ((x ^ a.y) & 15u);
            x ^= h[o];
            o = 12u + ((x ^ a.z) & 15u);
            x ^= h[o];
            v = a.w ^ x;
        } else if (V == 4 || V == 9) {                   // E / J
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t oi = (uint32_t)__shfl((int)idx, k * 16 + (lane >> 2), 64);
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(lines + (uint64_t)oi * 4 + (lane & 3)), (lds_ptr_t)&s_line[wv][k][0], 16, 0, 0);
            }
            if (V == 9 && (it & 7) == 0) {               // this lane's next 8 draws: one sector, by its quad
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint64_t ot = (uint64_t)blockIdx.x * blockDim.x + (uint64_t)(wv * 64 + k * 16 + (lane >> 2));
                    const double *src = draws + ot * (uint64_t)(steps + 8) + it;
                    __builtin_amdgcn_global_load_lds((glb_ptr_t)((const uint4 *)src + (lane & 3)), (lds_ptr_t)&s_draw[wv][k][0], 16, 0, 0);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const uint4 a = ((const uint4 *)&s_line[wv][0][0])[lane * 4];
            const uint4 d = ((const uint4 *)&s_line[wv][0][0])[lane * 4 + 3];
            v = a.x ^ a.w ^ d.y;
            if (V == 9) dacc += ((const double *)&s_draw[wv][0][0])[lane * 8 + (it & 7)];
        } else if (V == 6) {                             // G
            const uint4 a = lines[(uint64_t)idx * 4];
            dacc += my_draws[it];
            v = a.x ^ a.w;
        } else if (V == 7) {                             // H
            const uint4 a = lines[(uint64_t)idx * 4];
            const uint32_t li = mix32(a.x ^ idx) % n_list_lines;
            const uint16_t x = ((const uint16_t *)(lists + (uint64_t)li * 4))[a.y & 31u];
            v = a.w ^ x;
        } else if (V == 8) {                             // I
            const uint4 a = lines[(uint64_t)idx * 4];
            const uint32_t li = mix32(a.x ^ idx) % n_list_lines;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t oi = (uint32_t)__shfl((int)li, k * 16 + (lane >> 2), 64);
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(lists + (uint64_t)oi * 4 + (lane & 3)), (lds_ptr_t)&s_node[wv][k][0], 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const uint16_t x = ((const uint16_t *)&s_node[wv][0][0])[lane * 32 + (a.y & 31u)];
            v = a.w ^ x;
        }
        acc ^= v;
        idx = mix32(v + idx * 0x9E3779B9u + (uint32_t)it) % n_lines;
    }
    if (acc == 0x12345678u || dacc == 1.2345) out[0] = acc;
}

template <int V, int WAVES>
static void run(const char *name, const uint4 *lines, uint32_t n_lines, const uint4 *lists, uint32_t n_list_lines, const double *draws,
                int steps, uint32_t *d_out, int n_cu, int occ) {
    const int blocks = n_cu * occ;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL((bench_kernel<V, WAVES>), dim3(blocks), dim3(256), 0, 0, lines, n_lines, lists, n_list_lines, draws, steps, d_out);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    const double n = (double)blocks * 256.0 * steps;
    printf("{\"variant\": \"%s\", \"blocks_per_cu\": %d, \"steps\": %.0f, \"ms\": %.3f, \"G_steps_per_s\": %.2f, \"us_per_lane_step\": %.2f}\n", name, occ, n,
           best, n / best / 1e6, best * 1e3 / steps);
    fflush(stdout);
}

int main(int argc, char **argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 6.0;
    const int steps = argc > 2 ? atoi(argv[2]) : 200;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    const uint64_t bytes = (uint64_t)(gib * (1ull << 30)) & ~4095ull;
    const uint32_t n_lines = (uint32_t)(bytes / 64);
    uint4 *lines, *lists; uint32_t *d_out; double *draws;
    CHECK(hipMalloc((void **)&lines, bytes));
    CHECK(hipMalloc((void **)&lists, bytes));
    const uint64_t n_lanes_max = (uint64_t)n_cu * 8 * 256;
    CHECK(hipMalloc((void **)&draws, n_lanes_max * (uint64_t)(steps + 8) * sizeof(double)));
    CHECK(hipMalloc((void **)&d_out, 4));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, (uint32_t *)lines, bytes / 4);
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, (uint32_t *)lists, bytes / 4);
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, (uint32_t *)draws, n_lanes_max * (uint64_t)(steps + 8) * 2);
    CHECK(hipDeviceSynchronize());
    for (int occ : {5, 3}) {
        run<0, 5>("A 16B", lines, n_lines, lists, n_lines, draws, steps, d_out, n_cu, occ);
        run<1, 5>("B 16B+8B same line", lines, n_lines, lists, n_lines, draws, steps, d_out, n_cu, occ);
        run<2, 5>("C 4x16B same line", lines, n_lines, lists, n_lines, draws, steps, d_out, n_cu, occ);
        run<3, 5>("D 16B + 3 dependent 2B same line", lines, n_lines, lists, n_lines, draws, steps, d_out, n_cu, occ);
        run<4, 5>("E quad LDS-DMA whole line", lines, n_lines, lists, n_lines, draws, steps, d_out, n_cu, occ);
        run<8, 5>("I 16B + dependent quad node fetch", lines, n_lines, lists, n_lines, draws, steps, d_out, n_cu, occ);
        run<9, 5>("J quad line + quad draw sectors", lines, n_lines, lists, n_lines, draws, steps, d_out, n_cu, occ);
        run<6, 5>("G 16B + 8B sequential draw", lines, n_lines, lists, n_lines, draws, steps, d_out, n_cu, occ);
        run<7, 5>("H 16B + dependent 2B second array", lines, n_lines, lists, n_lines, draws, steps, d_out, n_cu, occ);
    }
    return 0;
}
