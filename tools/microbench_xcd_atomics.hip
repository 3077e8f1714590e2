// tools/microbench_xcd_atomics.hip -- can the counting scan's fire-and-forget atomics (6.9e9 on a 1 GB table: ~19 G/s, what the scan
// runs at) be served by the XCDs' own L2s instead of the memory side?  A device-scope atomic has to be performed where all eight
// XCDs see it; a WORKGROUP-scope one is performed in the L2 of the XCD its workgroup runs on -- which is only usable if every XCD
// counts into a table of its own (summed afterwards).  Rates of both, and of the summing pass, on tables of the scan's size.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mbx tools/microbench_xcd_atomics.hip && /tmp/mbx
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ __forceinline__ uint32_t xcc_id() { uint32_t v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 15u; }

template <int MODE>   // 0: agent scope, one table; 1: workgroup scope, a table per XCD; 2: workgroup scope, one table (rate only: not a usable count)
__global__ void k(unsigned long long* tab, uint64_t words, uint64_t n_ops, unsigned long long* per_xcd_ops)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint32_t x = xcc_id();
    unsigned long long* t = MODE == 1 ? tab + (uint64_t)x * words : tab;
    uint64_t done = 0;
    for (; i < n_ops; i += stride, ++done) {
        const uint64_t a = ((uint64_t)mix((uint32_t)i) | ((uint64_t)mix((uint32_t)(i >> 7) + 77u) << 32)) & (words - 1);
        if (MODE == 0) __hip_atomic_fetch_add(&t[a], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_fetch_add(&t[a], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (threadIdx.x == 0 && blockIdx.x < 64) atomicAdd(&per_xcd_ops[x], 1ull);    // which XCDs the first blocks landed on
}
__global__ void k_sum(const unsigned long long* tab, uint64_t words, int n_tabs, unsigned long long* out, unsigned long long* total)
{
    unsigned long long acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (uint64_t)gridDim.x * blockDim.x) {
        unsigned long long s = 0;
        for (int t = 0; t < n_tabs; ++t) s += tab[(uint64_t)t * words + i];
        out[i] = s; acc += s;
    }
    atomicAdd(total, acc);
}

int main()
{
    const uint64_t words = (1ull << 30) / 8, n = 1ull << 32;          // a 1 GB table, 4.3e9 operations
    unsigned long long *tab, *out, *ctr;
    CK(hipMalloc(&tab, 8 * words * 8)); CK(hipMalloc(&out, words * 8)); CK(hipMalloc(&ctr, 64 * 8));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const char* names[3] = {"agent scope, one 1-GB table", "workgroup scope, a 1-GB table per XCD", "workgroup scope, one table (rate only)"};
    for (int mode = 0; mode < 3; ++mode) {
        CK(hipMemset(tab, 0, 8 * words * 8)); CK(hipMemset(ctr, 0, 64 * 8));
        CK(hipEventRecord(a));
        if (mode == 0) k<0><<<8192, 256>>>(tab, words, n, ctr); else if (mode == 1) k<1><<<8192, 256>>>(tab, words, n, ctr); else k<2><<<8192, 256>>>(tab, words, n, ctr);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        CK(hipMemset(ctr + 32, 0, 8));
        CK(hipEventRecord(a));
        k_sum<<<4096, 256>>>(tab, words, mode == 1 ? 8 : 1, out, ctr + 32);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms2; CK(hipEventElapsedTime(&ms2, a, b));
        unsigned long long h[33]; CK(hipMemcpy(h, ctr, 33 * 8, hipMemcpyDeviceToHost));
        printf("%-42s %7.2f G atomics/s   sum of all counters %llu (%s), summing pass %.2f ms;  first 64 blocks on XCDs:", names[mode], n / ms / 1e6, h[32],
               h[32] == n ? "every operation counted" : "OPERATIONS LOST", ms2);
        for (int x = 0; x < 8; ++x) printf(" %llu", h[x]);
        printf("\n");
    }
    return 0;
}
