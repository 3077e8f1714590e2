// tools/check_memset_large.hip -- does hipMemsetAsync zero a buffer larger than 4 GiB completely on this runtime?
// (libdfk zeroes its HBM fallback tables with a kernel of its own; this records why.)
//   hipcc --offload-arch=gfx950 -O2 -o tools/check_memset_large tools/check_memset_large.hip && tools/check_memset_large [GiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void fill(unsigned long long* p, unsigned long long n, unsigned long long v)
{ for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) p[i] = v; }
__global__ void count_nonzero(const unsigned long long* p, unsigned long long n, unsigned long long* out, unsigned long long* first)
{
    unsigned long long c = 0;
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x)
        if (p[i]) { ++c; atomicMin(first, i); }
    if (c) atomicAdd(out, c);
}
int main(int argc, char** argv)
{
    const unsigned long long gib = argc > 1 ? strtoull(argv[1], 0, 10) : 9, bytes = gib << 30, n = bytes / 8;
    unsigned long long *p = 0, *d = 0, h[2] = {0, ~0ull};
    if (hipMalloc(&p, bytes) != hipSuccess || hipMalloc(&d, 16) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, p, n, ~0ull);
    hipMemcpy(d, h, 16, hipMemcpyHostToDevice);
    hipError_t e = hipMemsetAsync(p, 0, bytes, 0);
    hipLaunchKernelGGL(count_nonzero, dim3(4096), dim3(256), 0, 0, p, n, d, d + 1);
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("hipMemsetAsync(%llu GiB): %s; %llu of %llu words left non-zero (first at byte %llu)\n", gib, hipGetErrorString(e), h[0], n, h[0] ? h[1] * 8 : 0ull);
    return h[0] ? 2 : 0;
}
