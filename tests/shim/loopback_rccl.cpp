// loopback_rccl.cpp — TEST INFRASTRUCTURE, not part of the product.
//
// The library talks to RCCL through eight entry points it resolves with dlopen (filtlong_amd/csrc/comm.hip, FLX_RCCL_LIB).
// RCCL refuses two ranks on one device, and the GPU boxes the tests run on have one GPU, so the world-size > 1 control
// flow of the C++ path (flx_rank_and_cut_comm_dev, the CLI's --gpus N) could never run on hardware.  This file implements
// those eight entry points for several PROCESSES SHARING ONE GPU: every collective is staged through a POSIX shared-memory
// segment on the host (device -> host copy, process barrier, reduce / gather on the host, host -> device copy), in the
// order the ranks call them.  It is slow and only meant for the few-megabyte exchanges of tests/test_gpu_comm2.py; the
// product never loads it unless FLX_RCCL_LIB points here.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

namespace {

constexpr int kMaxRanks = 8;
constexpr size_t kSlot = (size_t)64 << 20;  // staging bytes per rank

struct Shared {
    std::atomic<int> ready;
    std::atomic<int> arrived;
    std::atomic<int> generation;
    int world;
    unsigned char data[1];  // kMaxRanks * kSlot follow
};

struct Comm {
    Shared *sh = nullptr;
    int rank = 0, world = 1;
    std::string name;
};

void barrier(Comm *c) {
    Shared *s = c->sh;
    const int gen = s->generation.load();
    if (s->arrived.fetch_add(1) + 1 == c->world) {
        s->arrived.store(0);
        s->generation.fetch_add(1);
    } else {
        while (s->generation.load() == gen) usleep(50);
    }
}

unsigned char *slot(Comm *c, int r) { return c->sh->data + (size_t)r * kSlot; }

size_t type_size(ncclDataType_t t) {
    switch (t) {
        case ncclUint8: case ncclInt8: return 1;
        case ncclUint64: case ncclInt64: case ncclFloat64: return 8;
        case ncclUint32: case ncclInt32: case ncclFloat32: return 4;
        default: return 0;
    }
}

std::string shm_name(const ncclUniqueId &id) {
    char buf[64];
    unsigned long long v;
    memcpy(&v, id.internal, 8);
    snprintf(buf, sizeof buf, "/flx_loopback_%llx", v);
    return buf;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    memset(id, 0, sizeof *id);
    unsigned long long v = ((unsigned long long)getpid() << 32) ^ (unsigned long long)(uintptr_t)id ^ 0x5bd1e995ull;
    FILE *f = fopen("/dev/urandom", "rb");
    if (f) { unsigned long long r = 0; if (fread(&r, 8, 1, f) == 1) v ^= r; fclose(f); }
    memcpy(id->internal, &v, 8);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *out, int world, ncclUniqueId id, int rank) {
    if (world < 1 || world > kMaxRanks || rank < 0 || rank >= world) return ncclInvalidArgument;
    Comm *c = new Comm();
    c->rank = rank;
    c->world = world;
    c->name = shm_name(id);
    const size_t bytes = sizeof(Shared) + kMaxRanks * kSlot;
    int fd = -1;
    if (rank == 0) {
        fd = shm_open(c->name.c_str(), O_CREAT | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)bytes) != 0) return ncclSystemError;
    } else {
        for (int tries = 0; tries < 6000 && fd < 0; ++tries) {
            fd = shm_open(c->name.c_str(), O_RDWR, 0600);
            struct stat st;
            if (fd >= 0 && (fstat(fd, &st) != 0 || (size_t)st.st_size < bytes)) { close(fd); fd = -1; }
            if (fd < 0) usleep(10000);
        }
        if (fd < 0) return ncclSystemError;
    }
    void *m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return ncclSystemError;
    c->sh = (Shared *)m;
    if (rank == 0) {
        c->sh->arrived.store(0);
        c->sh->generation.store(0);
        c->sh->world = world;
        c->sh->ready.store(1);
    } else {
        while (c->sh->ready.load() != 1) usleep(1000);
    }
    barrier(c);
    if (rank == 0) shm_unlink(c->name.c_str());  // everybody has it mapped
    *out = (ncclComm_t)c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    Comm *c = (Comm *)comm;
    if (c) {
        if (c->sh) munmap(c->sh, sizeof(Shared) + kMaxRanks * kSlot);
        delete c;
    }
    return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t type, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t stream) {
    Comm *c = (Comm *)comm;
    if (type != ncclUint64 || op != ncclSum || count * 8 > kSlot) return ncclInvalidArgument;
    if (hipMemcpyAsync(slot(c, c->rank), send, count * 8, hipMemcpyDeviceToHost, stream) != hipSuccess) return ncclUnhandledCudaError;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    barrier(c);
    uint64_t *acc = new uint64_t[count ? count : 1];
    for (size_t i = 0; i < count; ++i) {
        uint64_t s = 0;
        for (int r = 0; r < c->world; ++r) s += ((const uint64_t *)slot(c, r))[i];
        acc[i] = s;
    }
    barrier(c);  // nobody overwrites a slot before everybody has read it
    const hipError_t e = hipMemcpy(recv, acc, count * 8, hipMemcpyHostToDevice);
    delete[] acc;
    return e == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
}

ncclResult_t ncclBroadcast(const void *send, void *recv, size_t count, ncclDataType_t type, int root, ncclComm_t comm,
                           hipStream_t stream) {
    Comm *c = (Comm *)comm;
    const size_t bytes = count * type_size(type);
    if (!type_size(type) || bytes > kSlot || root < 0 || root >= c->world) return ncclInvalidArgument;
    if (c->rank == root) {
        if (hipMemcpyAsync(slot(c, root), send, bytes, hipMemcpyDeviceToHost, stream) != hipSuccess) return ncclUnhandledCudaError;
    }
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    barrier(c);
    hipError_t e = hipSuccess;
    if (c->rank != root || send != recv) e = hipMemcpy(recv, slot(c, root), bytes, hipMemcpyHostToDevice);
    barrier(c);
    return e == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
}

ncclResult_t ncclAllGather(const void *send, void *recv, size_t sendcount, ncclDataType_t type, ncclComm_t comm,
                           hipStream_t stream) {
    Comm *c = (Comm *)comm;
    const size_t bytes = sendcount * type_size(type);
    if (!type_size(type) || bytes > kSlot) return ncclInvalidArgument;
    if (hipMemcpyAsync(slot(c, c->rank), send, bytes, hipMemcpyDeviceToHost, stream) != hipSuccess) return ncclUnhandledCudaError;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    barrier(c);
    hipError_t e = hipSuccess;
    for (int r = 0; r < c->world && e == hipSuccess; ++r)
        e = hipMemcpy((char *)recv + (size_t)r * bytes, slot(c, r), bytes, hipMemcpyHostToDevice);
    barrier(c);
    return e == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
}

ncclResult_t ncclGroupStart() { return ncclSuccess; }  // every call above is complete when it returns: nothing to defer
ncclResult_t ncclGroupEnd() { return ncclSuccess; }

const char *ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "no error (loopback)";
        case ncclInvalidArgument: return "invalid argument (loopback: only u64 sums and byte moves of <= 64 MiB per rank)";
        case ncclSystemError: return "shared memory error (loopback)";
        default: return "error (loopback)";
    }
}

}  // extern "C"
