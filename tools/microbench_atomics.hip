// tools/microbench_atomics.hip -- scattered atomic / load rates versus table size on MI355X (design input
// for the bucket counters and scatter cursors).  hipcc --offload-arch=gfx950 -O3 -o mb tools/microbench_atomics.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int MODE>   // 0: u64 atomic no return, 1: u64 atomic returning, 2: u32 atomic returning, 3: u64 load, 4: 32-B store, 5: u32 atomic no return
__global__ void k(unsigned long long* tab, uint64_t mask, uint64_t n_ops, unsigned long long* sink)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long acc = 0;
    for (; i < n_ops; i += stride) {
        uint64_t a = ((uint64_t)mix((uint32_t)i) | ((uint64_t)mix((uint32_t)(i >> 7) + 77u) << 32)) & mask;
        if (MODE == 0) atomicAdd(&tab[a], 1ull);
        else if (MODE == 1) acc += atomicAdd(&tab[a], 1ull);
        else if (MODE == 2) acc += atomicAdd(reinterpret_cast<unsigned int*>(tab) + a, 1u);
        else if (MODE == 3) acc += tab[a];
        else if (MODE == 5) atomicAdd(reinterpret_cast<unsigned int*>(tab) + a, 1u);
        else { uint4* p = reinterpret_cast<uint4*>(tab) + 2 * (a >> 2); p[0] = uint4{1, 2, 3, 4}; p[1] = uint4{5, 6, 7, 8}; }
    }
    if (acc == 0x123456789ull) *sink = acc;
}

template <int MODE> void run(const char* name, unsigned long long* tab, uint64_t words, uint64_t n_ops, unsigned long long* sink)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    k<MODE><<<8192, 256>>>(tab, words - 1, n_ops / 8, sink);
    CK(hipEventRecord(a));
    k<MODE><<<8192, 256>>>(tab, words - 1, n_ops, sink);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("  %-22s %7.2f G ops/s\n", name, n_ops / ms / 1e6);
}

int main()
{
    unsigned long long* sink; CK(hipMalloc(&sink, 8));
    for (uint64_t mb : {32, 128, 256, 512, 1024, 2048, 8192}) {
        const uint64_t words = mb * 1024 * 1024 / 8;
        unsigned long long* tab; CK(hipMalloc(&tab, words * 8)); CK(hipMemset(tab, 0, words * 8));
        printf("table %llu MB\n", (unsigned long long)mb);
        const uint64_t n = 1ull << 30;
        run<0>("atomic u64", tab, words, n, sink);
        run<1>("atomic u64 returning", tab, words, n, sink);
        run<2>("atomic u32 returning", tab, words, n, sink);
        run<5>("atomic u32", tab, words, n, sink);
        run<3>("load u64", tab, words, n, sink);
        run<4>("store 32 B", tab, words, n, sink);
        CK(hipFree(tab));
    }
    return 0;
}
