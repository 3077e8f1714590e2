// tools/tmpfs_write.cc -- how fast T threads fill one tmpfs file from private buffers, by the ways dfk's file writers could:
//   pwrite into a growing file | pwrite into pages made by fallocate | memcpy into a mapping of such pages, with and without
//   MADV_POPULATE_WRITE first, and the same on a file that was only ftruncate()d to size (no pages yet).  g++ -O2 -pthread tools/tmpfs_write.cc -o /tmp/tmpfs_write && /tmp/tmpfs_write /dev/shm/x 8 4
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <mutex>
#include <sys/mman.h>
#include <thread>
#include <unistd.h>
#include <vector>
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv)
{
    const char* path = argc > 1 ? argv[1] : "/dev/shm/tmpfs_write.bin";
    const uint64_t gb = argc > 2 ? atoll(argv[2]) : 8, bytes = gb << 30, piece = 32ull << 20;
    const int T = argc > 3 ? atoi(argv[3]) : 4;
    std::vector<std::vector<char>> src(T, std::vector<char>(piece, 7));
    enum { GROW, PRE_PWRITE, PRE_PWRITE_LOCKED, MAP, MAP_POP, SPARSE_POP, SPARSE_MAP, N };
    const char* names[N] = {"pwrite, growing file", "pwrite, pages exist", "pwrite, pages exist, one at a time", "mapping, pages exist", "mapping + MADV_POPULATE_WRITE", "sparse file, mapping + MADV_POPULATE_WRITE", "sparse file, mapping"};
    for (int mode = 0; mode < N; ++mode) {
        unlink(path);
        int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0666);
        double t_alloc = 0;
        if (mode >= SPARSE_POP) { if (ftruncate(fd, bytes)) return 1; }
        else if (mode != GROW) { const double t0 = now(); if (posix_fallocate(fd, 0, bytes)) { perror("fallocate"); return 1; } t_alloc = now() - t0; }
        char* map = nullptr;
        if (mode >= MAP) map = (char*)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        std::mutex mu;
        const double t0 = now();
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t) th.emplace_back([&, t] {
            for (uint64_t o = (uint64_t)t * piece; o < bytes; o += (uint64_t)T * piece) {
                if (mode == GROW || mode == PRE_PWRITE_LOCKED) { std::lock_guard<std::mutex> g(mu); if (pwrite(fd, src[t].data(), piece, o) != (ssize_t)piece) abort(); }
                else if (mode == PRE_PWRITE) { if (pwrite(fd, src[t].data(), piece, o) != (ssize_t)piece) abort(); }
                else { if (mode == MAP_POP || mode == SPARSE_POP) madvise(map + o, piece, MADV_POPULATE_WRITE); memcpy(map + o, src[t].data(), piece); }
            }
        });
        for (auto& x : th) x.join();
        const double dt = now() - t0;
        if (map) munmap(map, bytes);
        close(fd);
        printf("%-40s T=%2d  %6.2f GB/s  (fallocate %.2f s = %.1f GB/s)\n", names[mode], T, bytes / dt / 1e9, t_alloc, t_alloc ? bytes / t_alloc / 1e9 : 0.0);
        fflush(stdout);
    }
    {   // fallocate itself from T threads, a range each
        unlink(path);
        int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0666);
        const double t0 = now();
        std::vector<std::thread> th;
        const uint64_t each = bytes / T;
        for (int t = 0; t < T; ++t) th.emplace_back([&, t] { if (posix_fallocate(fd, (off_t)(t * each), (off_t)each)) abort(); });
        for (auto& x : th) x.join();
        printf("%-40s T=%2d  %6.2f GB/s\n", "fallocate, a range per thread", T, bytes / (now() - t0) / 1e9);
        const double t1 = now();
        if (ftruncate(fd, 0)) return 1;
        printf("%-40s       %6.2f GB/s\n", "ftruncate to 0 (pages freed)", bytes / (now() - t1) / 1e9);
        close(fd);
    }
    unlink(path);
    return 0;
}
