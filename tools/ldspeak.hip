// ldspeak.hip — peak rate of random ds_read_b64 table lookups vs occupancy on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int NLOADS>
__global__ void k(const double* tab_g, int iters, double* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* q = (double*)smem;
    for (int i = threadIdx.x; i < 264; i += blockDim.x) q[i] = tab_g[i % 257];
    __syncthreads();
    uint32_t x = threadIdx.x * 2654435761u + 12345u;
    uint32_t addr[NLOADS];
#pragma unroll
    for (int i = 0; i < NLOADS; ++i) { x = x * 1664525u + 1013904223u; addr[i] = ((x >> 9) & 31) * 8 + 33 * 8; }
    unsigned long long acc = 0;
    for (int it = 0; it < iters; ++it) {
        double v[NLOADS];
#pragma unroll
        for (int i = 0; i < NLOADS; ++i) v[i] = *(const double*)((const char*)q + addr[i]);
#pragma unroll
        for (int i = 0; i < NLOADS; ++i) acc ^= __double_as_longlong(v[i]);
#pragma unroll
        for (int i = 0; i < NLOADS; ++i) addr[i] = (addr[i] + 8 * ((uint32_t)acc & 1)) & 0x7f8;   // keep loads dependent on data, cheap
    }
    if (acc == 0x1234) out[0] = 1.0;
}

template <int NLOADS>
void run(const double* tab, int waves_per_block, int blocks_per_cu, double* out) {
    const int iters = 4000;
    size_t lds = 160 * 1024 / blocks_per_cu - 2048;
    if (lds < 4096) lds = 4096;
    hipFuncSetAttribute((const void*)k<NLOADS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    dim3 grid(256 * blocks_per_cu), block(waves_per_block * 64);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<NLOADS>, grid, block, lds, 0, tab, 10, out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<NLOADS>, grid, block, lds, 0, tab, iters, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double instr = (double)iters * NLOADS * waves_per_block * blocks_per_cu;  // per CU
    printf("loads/iter %2d  waves/CU %2d: %7.3f ms  %5.2f cycles per ds_read_b64 per CU (2.1 GHz)\n", NLOADS,
           waves_per_block * blocks_per_cu, ms, ms * 1e-3 * 2.1e9 / instr);
}

int main() {
    double h[257]; for (int i = 0; i < 257; ++i) h[i] = 1.0 - 1.0 / (1 + i);
    double *tab, *out; hipMalloc(&tab, sizeof h); hipMalloc(&out, 64); hipMemcpy(tab, h, sizeof h, hipMemcpyHostToDevice);
    int cfg[][2] = {{4, 1}, {7, 1}, {4, 2}, {4, 3}, {7, 2}, {4, 4}, {8, 3}, {8, 4}};
    for (auto& c : cfg) { run<8>(tab, c[0], c[1], out); run<15>(tab, c[0], c[1], out); run<32>(tab, c[0], c[1], out); }
    return 0;
}
