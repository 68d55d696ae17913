// mfma_probe.hip — which lanes of srcA / srcB feed which lane of the result of v_mfma_f64_4x4x4_4b_f64 (one f64 per lane in
// all three operands), found by one-hot inputs; and whether D = C + x through the matrix pipe rounds like v_add_f64.
// usage: mfma_probe          (prints "a b d" triples and the add check)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <vector>
#include <random>

__global__ void k_onehot(int *out) {  // out[a * 64 + b] = lane d whose result is non-zero (or -1)
    const int lane = threadIdx.x;
    for (int a = 0; a < 64; ++a)
        for (int b = 0; b < 64; ++b) {
            double A = lane == a ? 1.0 : 0.0, B = lane == b ? 1.0 : 0.0, C = 0.0, D;
            asm volatile("s_nop 7\n\ts_nop 7\n\tv_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %3\n\ts_nop 15\n\ts_nop 15" : "=&v"(D) : "v"(A), "v"(B), "v"(C));
            const unsigned long long m = __ballot(D != 0.0);
            if (lane == 0) out[a * 64 + b] = m ? (int)__ffsll((long long)m) - 1 : -1;
            if (lane == 0 && m && (m & (m - 1))) out[a * 64 + b] = -2;  // more than one lane
        }
}

// acc += x through the matrix pipe with a per-lane constant `ident` as srcA, and through v_add_f64
__global__ void k_add(const double *c, const double *x, const double *ident, double *via_mfma, double *via_add, int n) {
    const int lane = threadIdx.x;
    const double I = ident[lane];
    for (int i = lane; i < n; i += 64) {
        double C = c[i], X = x[i], D;
        asm volatile("s_nop 7\n\ts_nop 7\n\tv_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %3\n\ts_nop 15\n\ts_nop 15" : "=&v"(D) : "v"(I), "v"(X), "v"(C));
        via_mfma[i] = D;
        double E;
        asm volatile("v_add_f64 %0, %1, %2" : "=v"(E) : "v"(C), "v"(X));
        via_add[i] = E;
    }
}

int main() {
    // layout found with tools/mfma_probe2 (one-hot inputs): A lane = i + 4 blk + 16 k, B lane = j + 4 blk + 16 k, D lane = j + 4 blk + 16 i.
    // With A = identity (lanes where lane % 4 == lane / 16 hold 1.0) every lane gets D = C + its own B.
    std::vector<double> ident(64, 0.0);
    for (int l = 0; l < 64; ++l) ident[l] = (l % 4 == l / 16) ? 1.0 : 0.0;
    // rounding check on hard operands
    const int n = 1 << 20;
    std::vector<double> c(n), x(n);
    std::mt19937_64 rng(7);
    for (int i = 0; i < n; ++i) {
        uint64_t a = rng(), b = rng();
        const int kind = i & 7;
        if (kind < 3) { a = (a & 0x000fffffffffffffull) | ((uint64_t)(1023 + 10 + (int)(rng() % 20)) << 52); b = (b & 0x000fffffffffffffull) | ((uint64_t)(1023 + (int)(rng() % 8)) << 52); }
        else if (kind == 3) { a = (a & 0x000fffffffffffffull) | ((uint64_t)(1023 + 30) << 52); b = ((uint64_t)(1023 + 30 - 53) << 52); }  // exact half-ulp ties
        else if (kind == 4) { a &= 0x000fffffffffffffull; b &= 0x000fffffffffffffull; }  // subnormals
        else if (kind == 5) { a = 0; b = (b & 0x000fffffffffffffull) | ((uint64_t)(1000) << 52); }
        else { a &= 0x7fffffffffffffffull; b &= 0x7fffffffffffffffull; if (((a >> 52) & 0x7ff) == 0x7ff) a = 0; if (((b >> 52) & 0x7ff) == 0x7ff) b = 0; }
        memcpy(&c[i], &a, 8); memcpy(&x[i], &b, 8);
    }
    double *dc, *dx, *di, *dm, *da;
    hipMalloc(&dc, n * 8); hipMalloc(&dx, n * 8); hipMalloc(&di, 64 * 8); hipMalloc(&dm, n * 8); hipMalloc(&da, n * 8);
    hipMemcpy(dc, c.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipMemcpy(di, ident.data(), 64 * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_add, dim3(1), dim3(64), 0, 0, dc, dx, di, dm, da, n);
    std::vector<double> m(n), ad(n);
    hipMemcpy(m.data(), dm, n * 8, hipMemcpyDeviceToHost); hipMemcpy(ad.data(), da, n * 8, hipMemcpyDeviceToHost);
    long bad = 0, bad_host = 0;
    for (int i = 0; i < n; ++i) {
        if (memcmp(&m[i], &ad[i], 8)) { if (bad < 5) printf("mismatch %d (kind %d): c %a x %a mfma %a add %a\n", i, i & 7, c[i], x[i], m[i], ad[i]); ++bad; }
        volatile double h = c[i] + x[i]; double hh = h;
        if (memcmp(&hh, &ad[i], 8)) ++bad_host;
    }
    printf("mfma vs v_add_f64: %ld mismatches of %d; v_add_f64 vs host: %ld\n", bad, n, bad_host);
    return 0;
}
