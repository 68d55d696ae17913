// outbench.cpp — how fast can N threads put G GiB into ONE new regular file (what `filtlong ... > out.fastq` does)?
//   pwrite      : threads pwrite() disjoint ranges (buffered writes take the inode lock: serial in the kernel)
//   trunc+pwrite: ftruncate to the final size first, then the same
//   mmap        : ftruncate + mmap(MAP_SHARED) + memcpy by N threads (page faults allocate page-cache pages in parallel)
//   falloc+mmap : fallocate first (no SIGBUS on a full disk later), then mmap + memcpy
// usage: outbench <file> <GiB> <threads>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
    if (argc < 4) return 2;
    const char *path = argv[1];
    const size_t bytes = (size_t)(atof(argv[2]) * (1ull << 30));
    const int nt = atoi(argv[3]);
    std::vector<char> src(64u << 20);
    for (size_t i = 0; i < src.size(); ++i) src[i] = (char)('A' + i % 23);
    const size_t piece = 8u << 20;
    const size_t n_pieces = bytes / piece;
    for (int mode = 0; mode < 4; ++mode) {
        unlink(path);
        const double t0 = now();
        int fd = open(path, O_CREAT | O_RDWR | O_TRUNC, 0644);
        if (fd < 0) { perror("open"); return 1; }
        if (mode >= 1 && ftruncate(fd, (off_t)bytes) != 0) { perror("ftruncate"); return 1; }
        if (mode == 3 && posix_fallocate(fd, 0, (off_t)bytes) != 0) { perror("fallocate"); }
        const double t1 = now();
        char *map = nullptr;
        if (mode >= 2) {
            map = (char *)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            if (map == MAP_FAILED) { perror("mmap"); return 1; }
        }
        std::vector<std::thread> th;
        for (int t = 0; t < nt; ++t)
            th.emplace_back([&, t]() {
                for (size_t p = t; p < n_pieces; p += nt) {
                    const char *s = src.data() + (p % 8) * piece;
                    if (map) memcpy(map + p * piece, s, piece);
                    else if (pwrite(fd, s, piece, (off_t)(p * piece)) != (ssize_t)piece) perror("pwrite");
                }
            });
        for (auto &x : th) x.join();
        const double t2 = now();
        if (map) munmap(map, bytes);
        close(fd);
        const double t3 = now();
        static const char *names[] = {"pwrite", "trunc+pwrite", "mmap", "falloc+mmap"};
        printf("%-13s %2d threads: setup %.3f  copy %.3f  unmap+close %.3f  total %.3f s  = %.1f GB/s\n", names[mode], nt, t1 - t0, t2 - t1,
               t3 - t2, t3 - t0, bytes / (t3 - t0) / 1e9);
    }
    unlink(path);
    return 0;
}
