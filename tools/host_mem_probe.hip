// Can the lane kernel read the lists it has no device memory for from PINNED HOST memory?  Random 64-byte sectors (a quad of
// lanes, 16 bytes each: the QUAD form's fetch shape) of a host-mapped buffer, dependent chains, all CUs: sectors per second.
// usage: hipcc -O2 --offload-arch=gfx950 -o /tmp/hmp tools/host_mem_probe.hip && /tmp/hmp [GB]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void __launch_bounds__(256) probe(const uint4 *buf, uint64_t n_sectors, int steps, unsigned long long *sink) {
    const uint32_t lane = threadIdx.x & 63, quad = lane >> 2, piece = lane & 3;
    uint64_t x = ((uint64_t)blockIdx.x * 256 + (threadIdx.x & ~3u)) * 0x9E3779B97F4A7C15ull + quad;
    unsigned long long acc = 0;
    for (int s = 0; s < steps; s++) {
        x = x * 6364136223846793005ull + 1442695040888963407ull;
        const uint64_t sec = (x >> 20) % n_sectors;
        const uint4 v = buf[sec * 4 + piece];
        acc += v.x;
        x ^= (uint64_t)__shfl(v.y, lane & ~3u);   // the next address depends on the data (a quad stays together)
    }
    if (acc == 0x12345) sink[0] = acc;
}
int main(int argc, char **argv) {
    const double gb = argc > 1 ? atof(argv[1]) : 3.0;
    const uint64_t n_sectors = (uint64_t)(gb * 1e9) / 64;
    for (int host = 1; host >= 0; host--) {
        uint4 *buf = nullptr;
        hipError_t e = host ? hipHostMalloc((void **)&buf, n_sectors * 64, hipHostMallocMapped) : hipMalloc((void **)&buf, n_sectors * 64);
        if (e != hipSuccess) { printf("%s alloc: %s\n", host ? "host" : "device", hipGetErrorString(e)); continue; }
        if (host) for (uint64_t i = 0; i < n_sectors * 4; i += 64) ((uint32_t *)(buf + i))[1] = (uint32_t)(i * 2654435761u);
        else hipMemset(buf, 1, n_sectors * 64);
        unsigned long long *sink; hipMalloc((void **)&sink, 8);
        for (int wgs : {256, 1024, 4096}) {
            const int steps = host ? 64 : 512;
            hipLaunchKernelGGL(probe, dim3(wgs), dim3(256), 0, 0, buf, n_sectors, 8, sink);
            hipDeviceSynchronize();
            const double t0 = now();
            hipLaunchKernelGGL(probe, dim3(wgs), dim3(256), 0, 0, buf, n_sectors, steps, sink);
            hipDeviceSynchronize();
            const double dt = now() - t0;
            const double secs = (double)wgs * 64 * steps;   // 64 quads per workgroup
            printf("%s memory, %d workgroups: %.3f G sectors/s (%.1f GB/s), %.2f us per dependent step\n", host ? "pinned host" : "device", wgs,
                   secs / dt / 1e9, secs * 64 / dt / 1e9, dt / steps * 1e6);
        }
        if (host) hipHostFree(buf); else hipFree(buf);
        hipFree(sink);
    }
    return 0;
}
