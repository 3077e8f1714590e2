// tools/microbench_sweep.hip -- what a scatter sweep costs per record on MI355X, alone on the chip: (A) as k_scatter_runs
// does it -- a RETURNING 64-bit atomic on the bucket's cursor (a 107-MB table for a tenth of 2^27 buckets), then two
// 16-byte stores at the slot it returned -- against (B) slot = bucket base (a plain load from a table of that size) +
// a rank the lane already holds, then the same stores.  DESIGN.md section 9, "the next lever".
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mbs tools/microbench_sweep.hip && /tmp/mbs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

__global__ void k_init(unsigned long long* cur, uint64_t nb, uint64_t per) { for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += (uint64_t)gridDim.x * blockDim.x) cur[i] = i * per; }

template <int MODE>
__global__ void k(unsigned long long* cur, uint64_t nb_mask, uint4* rec, uint64_t n_rec, uint64_t n_ops)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_ops; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t b = mix((uint32_t)i ^ (uint32_t)(i >> 32) * 0x9E3779B1u) & nb_mask;
        uint64_t dst;
        if (MODE == 0) dst = atomicAdd(&cur[b], 1ull);
        else dst = cur[b] + (i & 31u);
        if (dst < n_rec) { rec[2 * dst] = uint4{(uint32_t)i, 2, 3, 4}; rec[2 * dst + 1] = uint4{5, 6, 7, (uint32_t)b}; }
    }
}

int main()
{
    const uint64_t nb = 1ull << 24, per = 52, n_rec = nb * per, n_ops = n_rec - nb * 32;   // 16.7 M buckets (134 MB of cursors), 8.7e8 records (27.9 GB)
    unsigned long long* cur; uint4* rec;
    CK(hipMalloc(&cur, nb * 8)); CK(hipMalloc(&rec, n_rec * 32));
    for (int mode = 0; mode < 2; ++mode) {
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        for (int rep = 0; rep < 2; ++rep) {
            k_init<<<4096, 256>>>(cur, nb, per);
            CK(hipEventRecord(a));
            if (mode == 0) k<0><<<16384, 256>>>(cur, nb - 1, rec, n_rec, n_ops); else k<1><<<16384, 256>>>(cur, nb - 1, rec, n_rec, n_ops);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        }
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("%-44s %7.1f ms for %.2e records: %5.2f G records/s\n", mode == 0 ? "returning atomic on the cursor + 32-B store" : "base load + held rank + 32-B store", ms, (double)n_ops, n_ops / ms / 1e6);
    }
    return 0;
}
