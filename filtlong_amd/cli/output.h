// output.h — the output pass's engine: n independent pieces produced by several threads, written in order or, into a regular
// file, each at its own offset (src/main.cpp:263-313 writes record by record).  Included by main.cpp only.
#pragma once
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/stat.h>
#include <sys/uio.h>
#include <unistd.h>

#include "fastx.h"

// ---- ordered pieces ----------------------------------------------------------------------------------------------------
// The output is produced as `n` independent pieces by several threads.  produce(j, piece) fills piece j and says whether it
// could.  With `offsets` (n + 1 byte offsets, the sink a regular file that is not in append mode) every thread writes its
// own pieces with pwrite at base + offsets[j] and the file position is moved behind the last one; without, this thread
// writes the pieces in order as they become ready, and no more than 2 x threads of them exist at a time.
static bool g_direct_pieces = false;  // write_pieces: the producers write their pieces themselves (pwrite at known offsets)
static off_t g_direct_base = 0;
static int g_shared_out = -1;         // ranks forked by --gpus N: a duplicate of the job's stdout (the SAME open file in every rank)
// `forced_base` >= 0: the sink is a regular file shared with other processes and this process's pieces start at that offset (the
// file position is then nobody's to move here)
template <class Produce>
static bool write_pieces(size_t n, Produce &&produce, FILE *sink, const std::vector<uint64_t> *offsets, off_t forced_base = -1) {
    fflush(sink);
    const int fd = fileno(sink);
    struct stat st;
    const int fl = fcntl(fd, F_GETFL);
    const off_t base = forced_base >= 0 ? forced_base : lseek(fd, 0, SEEK_CUR);
    const bool direct = offsets && (forced_base >= 0 || !getenv("FLX_CLI_ORDERED_OUTPUT")) && fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && fl >= 0 &&
                        !(fl & O_APPEND) && base >= 0;
    if (forced_base >= 0 && !direct) return false;
    g_direct_pieces = direct;
    g_direct_base = base;
    std::vector<std::string> piece(n);
    std::vector<char> state(n, 0);  // 1: ready (or written), 2: failed
    std::mutex mu;
    std::condition_variable cv;
    std::atomic<size_t> next{0};
    size_t written = 0;
    const unsigned n_workers = (unsigned)std::max<size_t>(1, std::min<size_t>(host_threads(), n));
    const size_t ahead = 2 * (size_t)n_workers;
    auto worker = [&] {
        for (;;) {
            const size_t j = next.fetch_add(1);
            if (j >= n) return;
            if (!direct) {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return j < written + ahead; });
            }
            std::string &buf = piece[j];
            if (offsets && !direct) buf.reserve((size_t)((*offsets)[j + 1] - (*offsets)[j]));
            bool ok = produce(j, buf);
            const bool self_written = direct && ok && buf.empty() && (*offsets)[j + 1] != (*offsets)[j];  // the producer used pwritev itself
            if (offsets && !self_written) ok = ok && buf.size() == (*offsets)[j + 1] - (*offsets)[j];
            if (direct) {
                for (size_t done = 0; ok && done < buf.size();) {
                    const ssize_t w = pwrite(fd, buf.data() + done, buf.size() - done, base + (off_t)((*offsets)[j] + done));
                    if (w <= 0) ok = false;
                    else done += (size_t)w;
                }
                std::string().swap(buf);
            }
            {
                std::unique_lock<std::mutex> lk(mu);
                state[j] = ok ? 1 : 2;
            }
            cv.notify_all();
        }
    };
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < n_workers; ++t) pool.emplace_back(worker);
    bool failed = false;
    for (size_t j = 0; j < n; ++j) {
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return state[j] != 0; });
            failed = failed || state[j] == 2;
        }
        if (!direct) {
            if (!failed && fwrite(piece[j].data(), 1, piece[j].size(), sink) != piece[j].size()) failed = true;  // (the producers run dry below)
            std::string().swap(piece[j]);
            {
                std::unique_lock<std::mutex> lk(mu);
                written = j + 1;
            }
            cv.notify_all();
        }
    }
    for (auto &t : pool) t.join();
    if (direct && forced_base < 0 && !failed && lseek(fd, base + (off_t)(*offsets)[n], SEEK_SET) < 0) failed = true;
    return !failed;
}

