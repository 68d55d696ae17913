// valubench.hip — issue cost of FP64 VALU instructions on gfx950 (cycles per wave-instruction per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>

template <int OP>
__global__ void k(double* out, int iters, double a, double b) {
    double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < iters; ++i) {
#define STEP(x)                                                        \
    if (OP == 0) x = x + a;                                            \
    else if (OP == 1) x = x * a;                                       \
    else if (OP == 2) x = __builtin_fma(x, a, b);                      \
    else if (OP == 3) x = fmin(x, a + x);                              \
    else if (OP == 4) asm volatile("v_mov_b64 %0, %1" : "=v"(x) : "v"(x)); \
    else if (OP == 5) { unsigned long long u = __double_as_longlong(x); u += 12345; x = __longlong_as_double(u); }
        STEP(x0) STEP(x1) STEP(x2) STEP(x3) STEP(x4) STEP(x5) STEP(x6) STEP(x7)
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

template <int OP>
void run(const char* name, int waves_per_simd, double* out, int mult) {
    const int iters = 20000;
    dim3 grid(256), block(256 * waves_per_simd > 1024 ? 1024 : 256 * waves_per_simd);
    if (waves_per_simd > 4) grid = dim3(256 * (waves_per_simd / 4));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<OP>, grid, block, 0, 0, out, 10, 1.000001, 0.5);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<OP>, grid, block, 0, 0, out, iters, 1.000001, 0.5);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double instr_per_simd = (double)iters * 8 * mult * waves_per_simd;
    printf("%-18s waves/SIMD %d: %7.3f ms  %5.2f cycles per instruction per SIMD (2.1 GHz)\n", name, waves_per_simd, ms,
           ms * 1e-3 * 2.1e9 / instr_per_simd);
}

int main() {
    double* out; hipMalloc(&out, 8 * 1024 * 1024);
    for (int w : {1, 2, 4}) {
        run<0>("v_add_f64", w, out, 1);
        run<1>("v_mul_f64", w, out, 1);
        run<2>("v_fma_f64", w, out, 1);
        run<3>("add+min f64", w, out, 2);
        run<4>("v_mov_b64", w, out, 1);
        run<5>("u64 add (2 ops)", w, out, 2);
    }
    return 0;
}
