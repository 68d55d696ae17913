#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(double *out, int a, int b) {
    const int lane = threadIdx.x;
    double A = lane == a ? 2.0 : 0.0, B = lane == b ? 3.0 : 0.0, C = 100.0 + lane, D;
    asm volatile("s_nop 7\n\ts_nop 7\n\tv_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %3\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : "=&v"(D) : "v"(A), "v"(B), "v"(C));
    out[lane] = D;
}
int main() {
    double *d; hipMalloc(&d, 512);
    int pairs[][2] = {{0,0},{0,1},{0,2},{0,3},{0,4},{1,0},{1,1},{1,4},{4,0},{4,1},{4,4},{16,0},{16,16},{16,17},{5,20},{1,16}};
    for (auto &p : pairs) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, p[0], p[1]);
        std::vector<double> h(64); hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost);
        printf("A%2d B%2d:", p[0], p[1]);
        for (int l = 0; l < 64; ++l) if (h[l] != 100.0 + l) printf(" D[%d]=%g", l, h[l] - (100.0 + l));
        printf("\n");
    }
}
