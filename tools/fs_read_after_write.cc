// tools/fs_read_after_write.cc -- is the FIRST read of a tmpfs file slower than the second?  (The DF stage reads input files
// another process has just written; dfk_count's upload ran at 16-20 GB/s on them and at 50 GB/s on a second reading.)
//   g++ -O2 -pthread -o /tmp/fsr tools/fs_read_after_write.cc && /tmp/fsr <GiB> <threads> <0 pread|1 mmap|2 mmap + MADV_SEQUENTIAL|3 mmap + MADV_RANDOM|4 mmap, no MADV_DONTNEED behind the copy> [path]
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <thread>
#include <unistd.h>
#include <vector>
int main(int argc, char** argv)
{
    const size_t total = (size_t)atoll(argv[1]) << 30; const int T = atoi(argv[2]); const int mode = atoi(argv[3]);
    const char* path = argc > 4 ? argv[4] : "/dev/shm/rtest.bin";
    const size_t chunk = 4 << 20;
    {   // written by 4 threads, one pwrite at a time each (as bench.py's writer does)
        int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0666);
        if (ftruncate(fd, total) != 0) return 1;
        std::atomic<size_t> next{0};
        std::vector<std::thread> th;
        for (int t = 0; t < 4; ++t) th.emplace_back([&, t] {
            char* buf = (char*)malloc(chunk); memset(buf, t + 1, chunk);
            for (size_t i; (i = next.fetch_add(1)) * chunk < total;) if (pwrite(fd, buf, chunk, i * chunk) < 0) break;
            free(buf);
        });
        for (auto& x : th) x.join();
        close(fd);
    }
    for (int rep = 0; rep < 3; ++rep) {
        int fd = open(path, O_RDONLY);
        char* map = mode ? (char*)mmap(nullptr, total, PROT_READ, MAP_SHARED, fd, 0) : nullptr;
        if (mode == 2) madvise(map, total, MADV_SEQUENTIAL);
        if (mode == 3) madvise(map, total, MADV_RANDOM);
        std::atomic<size_t> next{0};
        std::atomic<unsigned long long> sum{0};
        auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t) th.emplace_back([&] {
            char* buf = (char*)malloc(chunk); unsigned long long s = 0;
            for (size_t i; (i = next.fetch_add(1)) * chunk < total;) {
                if (mode) { memcpy(buf, map + i * chunk, chunk); if (mode != 4) madvise(map + i * chunk, chunk, MADV_DONTNEED); }
                else if (pread(fd, buf, chunk, i * chunk) < 0) break;
                s += (unsigned char)buf[17];
            }
            sum += s; free(buf);
        });
        for (auto& x : th) x.join();
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("read %d (%s, %d threads): %.2f s, %.1f GB/s\n", rep, mode == 0 ? "pread" : mode == 1 ? "mmap" : mode == 2 ? "mmap seq" : mode == 3 ? "mmap random" : "mmap kept", T, s, total / s / 1e9);
        if (map) munmap(map, total);
        close(fd);
    }
    unlink(path);
}
