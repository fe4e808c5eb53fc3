// hipMalloc timing probe (fresh process on a fresh box): is the SECOND multi-GB allocation slow, or any allocation after kernels ran?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void touch(char *p, size_t n) { size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4096; if (i < n) p[i] = 1; }
int main(int argc, char **argv) {
    const char *mode = argc > 1 ? argv[1] : "two";
    double t = now();
    hipFree(0);
    printf("[%s] init %.1f ms\n", mode, (now() - t) * 1e3);
    auto alloc = [&](double gb, bool use) {
        void *p = nullptr; double t0 = now();
        hipError_t e = hipMalloc(&p, (size_t)(gb * 1e9));
        double t1 = now();
        if (use && e == hipSuccess) { hipLaunchKernelGGL(touch, dim3((unsigned)((size_t)(gb * 1e9) / 4096 / 256 + 1)), dim3(256), 0, 0, (char *)p, (size_t)(gb * 1e9)); hipDeviceSynchronize(); }
        printf("[%s] hipMalloc %.1f GB: %.2f ms (%s)%s %.2f ms\n", mode, gb, (t1 - t0) * 1e3, hipGetErrorString(e), use ? ", touch" : "", (now() - t1) * 1e3);
        return p;
    };
    if (!strcmp(mode, "two")) { void *a = alloc(4.5, true); void *b = alloc(5.2, true); void *c = alloc(13.8, true); void *d = alloc(5.2, false); hipFree(a); hipFree(b); hipFree(c); hipFree(d); alloc(5.2, false); }
    else if (!strcmp(mode, "one")) { alloc(9.7, true); alloc(0.5, true); alloc(13.8, true); }
    else if (!strcmp(mode, "pre")) {   // does ONE big allocation, freed at once, make the later ones cheap? (pw_warmup's pretouch)
        const double gb = argc > 2 ? atof(argv[2]) : 40.0;
        void *a = alloc(gb, false); double t0 = now(); hipFree(a); printf("[pre] hipFree %.2f ms\n", (now() - t0) * 1e3);
        void *b = alloc(4.5, true); void *c = alloc(5.2, true); void *d = alloc(13.8, true); void *e2 = alloc(12.9, true);
        hipFree(b); hipFree(c); hipFree(d); hipFree(e2);
    }
    else if (!strcmp(mode, "small")) { for (int i = 0; i < 6; i++) alloc(1.0, false); alloc(5.2, false); }
    return 0;
}
