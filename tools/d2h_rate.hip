// tools/d2h_rate.hip -- device memory -> a tmpfs file the way dfk's writers do it (lanes with two pinned buffers each: a DMA in
// flight while the one before is copied into the file's mapping), and the same with the halves taken apart: the DMA alone, the
// DMA from one stream per lane with hipMemcpyAsync vs a copy KERNEL writing the pinned buffer.
//   hipcc -O2 --offload-arch=gfx950 tools/d2h_rate.hip -o /tmp/d2h_rate -pthread && /tmp/d2h_rate /dev/shm/x 16
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <thread>
#include <unistd.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k_copy(const uint4* __restrict__ s, uint4* __restrict__ d, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
}
int main(int argc, char** argv)
{
    const char* path = argc > 1 ? argv[1] : "/dev/shm/d2h_rate.bin";
    const uint64_t bytes = (uint64_t)(argc > 2 ? atoll(argv[2]) : 16) << 30, piece = argc > 3 ? (uint64_t)atoll(argv[3]) << 20 : 32ull << 20;
    char* dev; CK(hipMalloc(&dev, bytes)); CK(hipMemset(dev, 5, bytes)); CK(hipDeviceSynchronize());
    enum { DMA, KERNEL, DMA_FILE, KERNEL_FILE, N };
    const char* names[N] = {"DMA only", "copy kernel only", "DMA + copy into the file's mapping", "copy kernel + copy into the file's mapping"};
    for (int T : {1, 2, 4, 8, 16})
        for (int mode = 0; mode < N; ++mode) {
            unlink(path);
            int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0666);
            char* map = nullptr;
            if (mode >= DMA_FILE) { if (posix_fallocate(fd, 0, bytes)) return 1; map = (char*)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0); }
            std::vector<std::thread> th;
            std::vector<void*> pins(2 * T);
            for (auto& p : pins) CK(hipHostMalloc(&p, piece));
            const double t0 = now();
            for (int t = 0; t < T; ++t) th.emplace_back([&, t] {
                CK(hipSetDevice(0));
                hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
                hipEvent_t ev[2]; for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
                uint64_t off[2] = {~0ull, ~0ull}; int k = 0;
                auto flush = [&](int j) {
                    if (off[j] == ~0ull) return;
                    CK(hipEventSynchronize(ev[j]));
                    if (map) { madvise(map + off[j], piece, 23); memcpy(map + off[j], pins[2 * t + j], piece); }
                    off[j] = ~0ull;
                };
                for (uint64_t o = (uint64_t)t * piece; o + piece <= bytes; o += (uint64_t)T * piece, k ^= 1) {
                    flush(k);
                    if (mode == DMA || mode == DMA_FILE) CK(hipMemcpyAsync(pins[2 * t + k], dev + o, piece, hipMemcpyDeviceToHost, st));
                    else hipLaunchKernelGGL(k_copy, dim3(64), dim3(256), 0, st, (const uint4*)(dev + o), (uint4*)pins[2 * t + k], piece / 16);
                    CK(hipEventRecord(ev[k], st));
                    off[k] = o;
                    flush(k ^ 1);
                }
                flush(0); flush(1);
                CK(hipStreamDestroy(st));
            });
            for (auto& x : th) x.join();
            const double dt = now() - t0;
            printf("%-44s lanes=%2d  %6.2f GB/s\n", names[mode], T, bytes / dt / 1e9); fflush(stdout);
            if (map) munmap(map, bytes);
            close(fd);
            for (auto& p : pins) CK(hipHostFree(p));
        }
    unlink(path);
    return 0;
}
