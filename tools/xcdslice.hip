// xcdslice.hip — can an exact table larger than one L2 be served at the L2's rate when it is cut into one slice per XCD?
// (round-3 review, item 3 (i): an "XCD-sliced exact table" for k-mer sets without loci.)  MI355X has 8 XCDs with 4 MiB of L2
// each; a table of 16 MiB fits none of them but its eighth does.  The experiment: 2048 workgroups x 256 threads, every lane
// its own random 128-byte line, 16 loads in flight per thread —
//   whole     every workgroup reads anywhere in the 16 MiB table                      (what the cover kernel's far class does)
//   by-block  workgroup b reads only slice b % 8                                       (dispatch order as the affinity)
//   by-xcc    the workgroup reads the slice of the XCD it runs on (s_getreg_b32 HW_REG_XCC_ID)
//   2 MiB     every workgroup reads one 2 MiB table                                    (the L2-resident reference point)
// and, for the queue such a design needs, the streaming cost of writing and reading 8 bytes per question.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE>
__global__ void __launch_bounds__(256) k_sliced(const uint32_t *buf, uint64_t slice_lines, int iters, uint32_t *out, unsigned *xcc_hist) {
    uint32_t slice = 0;
    if (MODE == 1) slice = blockIdx.x & 7u;
    if (MODE == 2) {
        slice = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u;  // HW_REG_XCC_ID, bits 3:0
        if (threadIdx.x == 0) atomicAdd(&xcc_hist[8 * (blockIdx.x & 7u) + slice], 1u);
    }
    const uint64_t lines = MODE == 0 ? slice_lines * 8 : slice_lines;
    const uint32_t *base = buf + (uint64_t)slice * slice_lines * 32;
    uint64_t x = (blockIdx.x * 256ull + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        uint32_t v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            x = x * 6364136223846793005ull + 1442695040888963407ull;
            v[j] = base[((x >> 20) % lines) * 32 + (uint32_t)(x >> 59)];
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) acc ^= v[j];
    }
    if (acc == 0x12345) out[0] = acc;
}

__global__ void __launch_bounds__(256) k_queue_write(uint2 *q, uint64_t n) {
    for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) q[i] = make_uint2((uint32_t)i, (uint32_t)(i >> 7));
}
__global__ void __launch_bounds__(256) k_queue_read(const uint2 *q, uint64_t n, uint32_t *out) {
    uint32_t acc = 0;
    for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) acc ^= q[i].x + q[i].y;
    if (acc == 0x12345) out[0] = acc;
}

int main() {
    const uint64_t slice_bytes = 2ull << 20, total = 8 * slice_bytes;
    uint32_t *buf, *out;
    unsigned *hist;
    CK(hipMalloc(&buf, total));
    CK(hipMalloc(&out, 64));
    CK(hipMalloc(&hist, 64 * 4));
    CK(hipMemset(buf, 1, total));
    CK(hipMemset(hist, 0, 64 * 4));
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int blocks = 2048, iters = 200;
    const double lookups = (double)blocks * 256 * 16 * iters;
    auto run = [&](auto kernel, uint64_t slice_lines, const char *what) {
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, buf, slice_lines, 10, out, hist);
        hipDeviceSynchronize();
        hipEventRecord(a);
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, buf, slice_lines, iters, out, hist);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        printf("%-44s %8.3f ms  %7.1f G lookups/s\n", what, ms, lookups / ms / 1e6);
    };
    const uint64_t sl = slice_bytes / 128;
    run(k_sliced<0>, sl, "whole 16 MiB table, any workgroup anywhere");
    run(k_sliced<1>, sl, "8 slices of 2 MiB, slice = blockIdx % 8");
    CK(hipMemset(hist, 0, 64 * 4));
    run(k_sliced<2>, sl, "8 slices of 2 MiB, slice = XCC_ID of the CU");
    run(k_sliced<1>, sl / 8, "one 2 MiB table in 8 slices of 256 KiB (reference)");
    unsigned h[64];
    CK(hipMemcpy(h, hist, sizeof h, hipMemcpyDeviceToHost));
    printf("# workgroups by (blockIdx %% 8) x XCC_ID (the run above, incl. its warm-up launch):\n");
    for (int i = 0; i < 8; ++i) {
        printf("#  b%%8=%d:", i);
        for (int j = 0; j < 8; ++j) printf(" %5u", h[8 * i + j]);
        printf("\n");
    }
    // the queue: 8 bytes per question written once and read once
    const uint64_t nq = 1ull << 28;
    uint2 *q;
    CK(hipMalloc(&q, nq * 8));
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL(k_queue_write, dim3(4096), dim3(256), 0, 0, q, nq);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float w = 0, r = 0;
        hipEventElapsedTime(&w, a, b);
        hipEventRecord(a);
        hipLaunchKernelGGL(k_queue_read, dim3(4096), dim3(256), 0, 0, q, nq, out);
        hipEventRecord(b);
        hipEventSynchronize(b);
        hipEventElapsedTime(&r, a, b);
        if (rep) printf("queue of 2^28 questions x 8 B: write %7.3f ms (%5.2f TB/s)  read %7.3f ms (%5.2f TB/s)\n", w, nq * 8 / w / 1e9, r, nq * 8 / r / 1e9);
    }
    return 0;
}
