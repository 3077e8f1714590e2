// tools/fs_write_scaling.cc -- how fast can N threads fill ONE file (pwrite / shared mapping / after fallocate)?
//   g++ -O2 -pthread -o /tmp/fsw tools/fs_write_scaling.cc && /tmp/fsw <threads> <0 pwrite|1 mmap|2 fallocate+pwrite|4 fallocate, then mmap (timed apart)|5 mmap behind a thread that fallocates ahead> <GiB> <chunk MiB> [path]
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
    const int T = atoi(argv[1]); const int mode = atoi(argv[2]); const size_t total = (size_t)atoll(argv[3]) << 30; const size_t chunk = (size_t)atoi(argv[4]) << 20;
    const char* path = argc > 5 ? argv[5] : "/dev/shm/wtest.bin";
    int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0666);
    if (mode != 3) ftruncate(fd, total);
    if (mode == 4) { auto ta = std::chrono::steady_clock::now(); posix_fallocate(fd, 0, total); printf("fallocate %.2f s (%.2f GB/s), not in the time below\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - ta).count(), total / 1e9 / std::chrono::duration<double>(std::chrono::steady_clock::now() - ta).count()); }
    char* map = (mode == 1 || mode == 4 || mode == 5) ? (char*)mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0) : nullptr;
    std::atomic<size_t> next{0};
    std::atomic<size_t> allocated{mode == 5 ? 0 : total};
    auto t0 = std::chrono::steady_clock::now();
    std::thread alloc;
    if (mode == 5) alloc = std::thread([&] { const size_t step = (size_t)256 << 20; for (size_t o = 0; o < total; o += step) { posix_fallocate(fd, o, std::min(step, total - o)); allocated = std::min(total, o + step); } });
    if (mode == 2) { posix_fallocate(fd, 0, total); printf("fallocate %.2f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count()); }
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back([&, t] {
        char* buf = (char*)malloc(chunk); memset(buf, t + 1, chunk);
        for (size_t i; (i = next.fetch_add(1)) * chunk < total;) {
            if (mode == 1 || mode == 4 || mode == 5) { while (allocated.load() < (i + 1) * chunk) std::this_thread::yield(); memcpy(map + i * chunk, buf, chunk); if (mode == 5) madvise(map + i * chunk, chunk, MADV_DONTNEED); }
            else pwrite(fd, buf, chunk, i * chunk);
        }
        free(buf);
    });
    for (auto& x : th) x.join();
    if (alloc.joinable()) alloc.join();
    double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("T=%d mode=%d chunk=%zuMB: %.2f s, %.2f GB/s\n", T, mode, chunk >> 20, s, total / s / 1e9);
    close(fd); unlink(path);
}
