// tools/vram_alloc_cost.hip -- what does a large hipMalloc cost, and when is it paid (in the call, or by the first use)?
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/vac tools/vram_alloc_cost.hip && /tmp/vac [GiB]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void touch(uint4* p, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = uint4{1, 2, 3, 4}; }
int main(int argc, char** argv)
{
    const size_t bytes = (size_t)(argc > 1 ? atoll(argv[1]) : 45) << 30;
    (void)hipFree(nullptr);
    for (int rep = 0; rep < 3; ++rep) {
        void* p = nullptr;
        double t0 = now();
        if (hipMalloc(&p, bytes) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
        double t1 = now();
        (void)hipDeviceSynchronize();
        double t2 = now();
        hipLaunchKernelGGL(touch, dim3(4096), dim3(256), 0, 0, (uint4*)p, bytes / 16);
        (void)hipDeviceSynchronize();
        double t3 = now();
        hipLaunchKernelGGL(touch, dim3(4096), dim3(256), 0, 0, (uint4*)p, bytes / 16);
        (void)hipDeviceSynchronize();
        double t4 = now();
        (void)hipFree(p);
        double t5 = now();
        printf("%zu GiB: hipMalloc %.3f s, sync %.3f s, first pass over it %.3f s, second %.3f s, hipFree %.3f s\n", bytes >> 30, t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4);
    }
}
