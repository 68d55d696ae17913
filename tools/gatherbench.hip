// gatherbench.hip — cost per wave-instruction of random table gathers on gfx950 (LDS vs vector L1).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

// MODE: 0 lds u8, 1 lds u16, 2 lds b32, 3 lds b64, 4 lds b128, 5 global b32, 6 global b64, 7 mixed: 2 lds b64 + 1 global b64
template <int MODE>
__global__ void k(const double* tab_g, int iters, double* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* q = (double*)smem;
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) q[i] = tab_g[i];
    __syncthreads();
    constexpr int N = 16;
    uint32_t x = threadIdx.x * 2654435761u + 12345u;
    uint32_t idx[N];
#pragma unroll
    for (int i = 0; i < N; ++i) { x = x * 1664525u + 1013904223u; idx[i] = ((x >> 9) & 31) + 33; }
    unsigned long long acc = 0;
    const unsigned tabreg = (unsigned)__double_as_longlong(tab_g[threadIdx.x & 63]);
    const unsigned tabreg2 = (unsigned)(__double_as_longlong(tab_g[threadIdx.x & 63]) >> 32);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            if (MODE == 0) acc ^= ((const unsigned char*)q)[idx[i]];
            else if (MODE == 1) acc ^= ((const unsigned short*)q)[idx[i]];
            else if (MODE == 2) acc ^= ((const uint32_t*)q)[idx[i]];
            else if (MODE == 3) acc ^= __double_as_longlong(q[idx[i]]);
            else if (MODE == 4) { const ulonglong2 v = ((const ulonglong2*)q)[idx[i]]; acc ^= v.x ^ v.y; }
            else if (MODE == 5) acc ^= ((const uint32_t*)tab_g)[idx[i]];
            else if (MODE == 6) acc ^= __double_as_longlong(tab_g[idx[i]]);
            else if (MODE == 8) { acc ^= (unsigned)__builtin_amdgcn_ds_bpermute((int)(idx[i] & 63) << 2, (int)tabreg); }
            else if (MODE == 9) { acc ^= (unsigned)__builtin_amdgcn_ds_bpermute((int)(idx[i] & 63) << 2, (int)tabreg) ^ ((unsigned long long)(unsigned)__builtin_amdgcn_ds_bpermute((int)(idx[i] & 63) << 2, (int)tabreg2) << 32); }
            else { acc ^= __double_as_longlong(q[idx[i]]) ^ __double_as_longlong(q[idx[i] + 264]) ^ __double_as_longlong(tab_g[idx[(i + 1) % N]]); }
        }
#pragma unroll
        for (int i = 0; i < N; ++i) idx[i] = ((idx[i] + ((uint32_t)acc & 1)) & 31) + 33;
    }
    if (acc == 0x1234) out[0] = 1.0;
}

template <int MODE>
void run(const double* tab, int wpb, int bpc, double* out, const char* name, int per_iter) {
    const int iters = 3000;
    size_t lds = 160 * 1024 / bpc - 2048;
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    dim3 grid(256 * bpc), block(wpb * 64);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, grid, block, lds, 0, tab, 10, out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, grid, block, lds, 0, tab, iters, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double n = (double)iters * 16 * wpb * bpc;
    printf("%-28s waves/CU %2d: %7.3f ms  %6.2f cycles per gather-group per CU (%d instr)\n", name, wpb * bpc, ms,
           ms * 1e-3 * 2.1e9 / n, per_iter);
}

int main() {
    static double h[1024]; for (int i = 0; i < 1024; ++i) h[i] = 1.0 - 1.0 / (1 + i);
    double *tab, *out; hipMalloc(&tab, sizeof h); hipMalloc(&out, 64); hipMemcpy(tab, h, sizeof h, hipMemcpyHostToDevice);
    for (int w : {7, 14}) {
        int wpb = 7, bpc = w / 7;
        run<0>(tab, wpb, bpc, out, "lds u8", 1);
        run<1>(tab, wpb, bpc, out, "lds u16", 1);
        run<2>(tab, wpb, bpc, out, "lds b32", 1);
        run<3>(tab, wpb, bpc, out, "lds b64", 1);
        run<4>(tab, wpb, bpc, out, "lds b128", 1);
        run<5>(tab, wpb, bpc, out, "L1 b32", 1);
        run<6>(tab, wpb, bpc, out, "L1 b64", 1);
        run<7>(tab, wpb, bpc, out, "2x lds b64 + 1x L1 b64", 3);
        run<8>(tab, wpb, bpc, out, "ds_bpermute_b32", 1);
        run<9>(tab, wpb, bpc, out, "2x ds_bpermute_b32 (64-bit)", 2);
    }
    return 0;
}
