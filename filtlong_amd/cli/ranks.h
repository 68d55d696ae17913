// ranks.h — one process per GPU under `--gpus N`: the forked ranks, their pipes, the private directory of the output parts, the
// watchdog that ends the job when a rank dies early.  Included by main.cpp only.
#pragma once
#include <atomic>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <signal.h>
#include <sys/wait.h>
#include <unistd.h>

// ---- rank 0 of `--gpus N`: the forked ranks, their pipes, the private directory of the output parts -------------------------
// A watchdog thread reaps the children while rank 0 works: a child that dies early (no such device, RCCL missing, ...) would
// otherwise leave rank 0 blocked inside a collective for ever — the job then ends at once with a message, and whichever way
// main() is left no child and no file stays behind.
struct Job {
    std::vector<pid_t> children;
    std::vector<int> id_pipes;
    std::string dir;
    std::thread watchdog;
    std::mutex mu;
    std::atomic<bool> stop{false};
    std::vector<int> exited;  // exit status per child, -1 while it runs
    Job() = default;
    Job &operator=(Job &&o) {  // (a forked child drops its copy; no thread exists at that point)
        children = std::move(o.children); id_pipes = std::move(o.id_pipes); dir = std::move(o.dir);
        exited.clear();
        return *this;
    }
    void remove_dir() {
        if (dir.empty()) return;
        for (size_t r = 0; r <= children.size(); ++r)
            for (const char *kind : {"part", "vblocks", "vtable"}) unlink((dir + "/out." + kind + std::to_string(r)).c_str());
        rmdir(dir.c_str());
        dir.clear();
    }
    void kill_all() {
        std::lock_guard<std::mutex> lk(mu);
        for (size_t i = 0; i < children.size(); ++i)
            if (exited.empty() || exited[i] < 0) kill(children[i], SIGKILL);
    }
    bool poll_once(bool block) {  // returns false when a child ended badly
        bool ok = true;
        std::lock_guard<std::mutex> lk(mu);
        if (exited.size() < children.size()) exited.resize(children.size(), -1);  // (a pipe() / fork() that failed midway: start_watchdog never sized it — advisor, round 5)
        for (size_t i = 0; i < children.size(); ++i) {
            if (exited[i] >= 0) { ok = ok && exited[i] == 0; continue; }
            int st = 0;
            const pid_t r = waitpid(children[i], &st, block ? 0 : WNOHANG);
            if (r == children[i]) exited[i] = (WIFEXITED(st) ? WEXITSTATUS(st) : 128 + (WIFSIGNALED(st) ? WTERMSIG(st) : 0));
            else if (r < 0) exited[i] = 255;
            if (exited[i] > 0) ok = false;
        }
        return ok;
    }
    void start_watchdog() {
        exited.assign(children.size(), -1);
        watchdog = std::thread([this]() {
            while (!stop.load()) {
                if (!poll_once(false)) {
                    int which = 0, status = 0;
                    { std::lock_guard<std::mutex> lk(mu); for (size_t i = 0; i < exited.size(); ++i) if (exited[i] > 0) { which = (int)i + 1; status = exited[i]; } }
                    const std::string msg = "\nError: rank " + std::to_string(which) + " ended early (status " + std::to_string(status) +
                                            "): no GPU for it, or the RCCL library could not be loaded?\n";
                    (void)!write(2, msg.data(), msg.size());
                    kill_all();
                    poll_once(true);
                    remove_dir();
                    _exit(1);
                }
                usleep(20000);
            }
        });
    }
    bool finish() {  // normal end: every child must have left with status 0
        if (children.empty()) return true;
        stop.store(true);
        if (watchdog.joinable()) watchdog.join();
        const bool ok = poll_once(true);
        remove_dir();
        children.clear();
        return ok;
    }
};
static Job g_job;
struct JobGuard {
    ~JobGuard() {  // an early return of rank 0
        if (g_job.children.empty()) return;
        g_job.stop.store(true);
        if (g_job.watchdog.joinable()) g_job.watchdog.join();
        g_job.kill_all();
        g_job.poll_once(true);
        g_job.remove_dir();
    }
};

// Either a launcher set RANK / WORLD_SIZE (/ LOCAL_RANK) + FLX_COMM_ID_FILE, or --gpus N forks N-1 copies of this process here, before
// any GPU state exists.  Returns -1 to go on (g_rank / g_world / g_job are set), or the process's exit code.
static int start_ranks(const Args &args, std::string &id_file, int &id_pipe) {
    // ---- ranks: one process per GPU (north_star / SURVEY §8e) ---------------------------------------------------
    // Either a launcher set RANK / WORLD_SIZE (/ LOCAL_RANK), or --gpus N forks N-1 copies of this process here, before
    // any GPU state exists.  Reads are sharded by count in contiguous blocks of file order; every rank parses the (mapped)
    // input's record index, scores its own block and takes part in the global stage through the library's RCCL
    // communicator; rank 0 owns stderr and stdout.
    // Launcher mode is an explicit opt-in — RANK + WORLD_SIZE + FLX_COMM_ID_FILE (a path unique to the job) all set: a bare
    // WORLD_SIZE inherited from a SLURM / torchrun shell must not turn a plain run into a rank that waits for peers.
    if (getenv("WORLD_SIZE") && getenv("RANK") && getenv("FLX_COMM_ID_FILE")) {
        g_world = std::max(1, atoi(getenv("WORLD_SIZE")));
        g_rank = atoi(getenv("RANK"));
        id_file = getenv("FLX_COMM_ID_FILE");
        g_part_prefix = id_file + ".out";
    } else if (args.gpus > 1) {
        if (!getenv("FLX_DEVICE")) {  // (FLX_DEVICE pins every rank to one device: the one-GPU tests of this path)
            // the HIP runtime does not survive a fork, so the device count comes from a probe child
            const pid_t probe = fork();
            if (probe == 0) _exit(std::max(0, std::min(flx_device_count(), 255)));
            int st = 0;
            if (probe < 0 || waitpid(probe, &st, 0) < 0 || !WIFEXITED(st)) { std::cerr << "Error: cannot probe the GPUs\n"; return 1; }
            if (WEXITSTATUS(st) < args.gpus) {
                std::cerr << "Error: --gpus " << args.gpus << " but only " << WEXITSTATUS(st) << " GPU(s) visible\n";
                return 1;
            }
        }
        g_world = args.gpus;
        // a private directory for the ranks' output parts (mkdtemp: mode 0700, unpredictable name)
        const char *td = getenv("TMPDIR");
        std::string tmpl = std::string(td && *td ? td : "/tmp") + "/flx_XXXXXX";
        if (!mkdtemp(&tmpl[0])) { std::cerr << "Error: cannot create a temporary directory under " << (td && *td ? td : "/tmp") << "\n"; return 1; }
        g_job.dir = tmpl;
        g_part_prefix = tmpl + "/out";
        // the communicator id reaches every rank through a pipe made before the fork
        std::vector<int> wr;
        for (int r = 1; r < g_world; ++r) {
            int fds[2];
            if (pipe(fds) != 0) { std::cerr << "Error: pipe failed\n"; return 1; }
            const pid_t pid = fork();
            if (pid < 0) { std::cerr << "Error: fork failed\n"; return 1; }
            if (pid == 0) {
                g_rank = r;
                g_job = Job();  // a child owns neither children nor the directory
                for (int w : wr) close(w);
                close(fds[1]);
                id_pipe = fds[0];
                break;
            }
            close(fds[0]);
            wr.push_back(fds[1]);
            g_job.children.push_back(pid);
        }
        if (g_rank == 0) {
            g_job.id_pipes = wr;
            g_job.start_watchdog();
        }
    }
    return -1;
}

// The communicator's 128-byte id reaches every rank (pipes made before the fork, or FLX_COMM_ID_FILE under a launcher), then the
// library's communicator is initialised.  Returns -1 to go on, or the process's exit code.
static int exchange_communicator_id(flx_ctx *ctx, const std::string &id_file, int id_pipe) {
    {
        // the communicator's 128-byte id: --gpus hands it to every child through its pipe; under a launcher it travels through
        // FLX_COMM_ID_FILE (rank 0: exclusive create of a temp name, never through a symlink, then rename; the others accept
        // only a file written after they started — a stale one from a crashed earlier job is older)
        unsigned char id[FLX_COMM_ID_BYTES];
        if (g_rank == 0) {
            if (flx_comm_unique_id(ctx, id) != FLX_OK) return fail_flx(ctx, "communicator");
            if (!g_job.children.empty()) {
                for (int w : g_job.id_pipes) {
                    if (write(w, id, sizeof id) != (ssize_t)sizeof id) { std::cerr << "Error: cannot hand the communicator id to a rank\n"; return 1; }
                    close(w);
                }
                g_job.id_pipes.clear();
            } else {
                const std::string tmp = id_file + ".tmp";
                unlink(tmp.c_str());
                const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW, 0600);
                if (fd < 0 || write(fd, id, sizeof id) != (ssize_t)sizeof id) { std::cerr << "Error: cannot write " << tmp << "\n"; return 1; }
                close(fd);
                if (rename(tmp.c_str(), id_file.c_str()) != 0) { std::cerr << "Error: cannot create " << id_file << "\n"; return 1; }
            }
        } else if (id_pipe >= 0) {
            size_t got = 0;
            struct pollfd pfd = {id_pipe, POLLIN, 0};
            while (got < sizeof id && poll(&pfd, 1, 120000) > 0) {  // rank 0 gone: EOF, at once
                const ssize_t k = read(id_pipe, id + got, sizeof id - got);
                if (k <= 0) break;
                got += (size_t)k;
            }
            close(id_pipe);
            if (got != sizeof id) return 1;
        } else {
            const time_t started = time(nullptr);
            bool ok = false;
            for (int tries = 0; tries < 6000 && !ok; ++tries) {  // up to 60 s
                struct stat sb;
                const int fd = open(id_file.c_str(), O_RDONLY | O_NOFOLLOW);
                if (fd >= 0) {
                    if (fstat(fd, &sb) == 0 && sb.st_mtime + 2 >= started) ok = read(fd, id, sizeof id) == (ssize_t)sizeof id;
                    close(fd);
                }
                if (!ok) usleep(10000);
            }
            if (!ok) return 1;
        }
        if (flx_comm_init(ctx, id, g_rank, g_world) != FLX_OK) return fail_flx(ctx, "communicator");
        uint64_t ready = 1;  // everybody has read the id
        if (flx_comm_sum_u64(ctx, &ready, 1) != FLX_OK) return fail_flx(ctx, "communicator");
        if (g_rank == 0 && !id_file.empty()) unlink(id_file.c_str());
    }

    return -1;
}
