// sort.hip — device radix sort and prefix scans for the global stage.
//
// Replaces the reference's `std::sort(reads2, by m_final_score desc)` (src/main.cpp:247-248) and turns
// its serial cut walk (main.cpp:251-257) into an exclusive scan.  64-bit keys (order-preserving image of
// the FP64 score), 32-bit payload (read index), LSD, 8 bits per pass, stable — so equal scores keep
// reads2 order on the device; the reference's (unstable) tie order only matters when a tie straddles the
// cut, which rank.hip detects and resolves on the host.
//
// Layout: a "wave tile" is 1024 consecutive keys owned by one wavefront (16 rounds of 64 coalesced
// keys).  Per pass: (1) per-tile digit histogram -> table[digit][tile], (2) exclusive scan of the
// table, (3) stable scatter with wave-level multi-split (8 ballots give each lane its rank among the
// lanes holding the same digit).  Passes whose 8 key bits are identical in every key are skipped
// (scores live in [0,100], so the top byte and often more are constant).
#include "flx_internal.h"
#include "rank_internal.h"

namespace {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

template <typename T>
__device__ __forceinline__ T wave_incl_scan(T v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        T u = __shfl_up(v, o, 64);
        if (lane >= o) v += u;
    }
    return v;
}

// block-wide exclusive scan of one value per thread (256 threads); returns exclusive prefix, total in *total
template <typename T>
__device__ __forceinline__ T block_excl_scan(T v, T *total) {
    __shared__ T wsum[SCAN_THREADS / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const T incl = wave_incl_scan(v, lane);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    T base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SCAN_THREADS / 64; ++w) {
        if (w < wave) base += wsum[w];
        tot += wsum[w];
    }
    __syncthreads();
    *total = tot;
    return base + incl - v;
}

template <typename T>
__global__ void __launch_bounds__(SCAN_THREADS) k_scan_reduce(uint64_t n, const T *in, T *block_sums) {
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE;
    T acc = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        const uint64_t i = base + (uint64_t)k * SCAN_THREADS + threadIdx.x;
        if (i < n) acc += in[i];
    }
    T tot;
    block_excl_scan(acc, &tot);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

// single block: exclusive scan of m values in place (m small)
template <typename T>
__global__ void __launch_bounds__(SCAN_THREADS) k_scan_small(uint64_t m, T *data) {
    T carry = 0;
    for (uint64_t base = 0; base < m; base += SCAN_THREADS) {
        const uint64_t i = base + threadIdx.x;
        const T v = i < m ? data[i] : 0;
        T tot;
        const T ex = block_excl_scan(v, &tot);
        if (i < m) data[i] = carry + ex;
        carry += tot;
    }
}

template <typename T>
__global__ void __launch_bounds__(SCAN_THREADS) k_scan_apply(uint64_t n, const T *in, T *out, const T *block_offsets) {
    // thread t owns SCAN_ITEMS consecutive elements so the scan order is the memory order
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
    T v[SCAN_ITEMS];
    T acc = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        v[k] = (base + k < n) ? in[base + k] : 0;
        acc += v[k];
    }
    T tot;
    T ex = block_excl_scan(acc, &tot) + block_offsets[blockIdx.x];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        if (base + k < n) out[base + k] = ex;
        ex += v[k];
    }
}

template <typename T>
int exclusive_scan(flx_ctx *ctx, uint64_t n, const T *in, T *out, void *workspace, size_t workspace_bytes) {
    if (n == 0) return FLX_OK;
    const uint64_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
    if (nb * sizeof(T) > workspace_bytes) return flx_fail(ctx, FLX_ERR_INVALID, "scan workspace too small");
    T *sums = (T *)workspace;
    hipLaunchKernelGGL(k_scan_reduce<T>, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, ctx->stream, n, in, sums);
    if (nb <= 16384) {
        hipLaunchKernelGGL(k_scan_small<T>, dim3(1), dim3(SCAN_THREADS), 0, ctx->stream, nb, sums);
    } else {
        char *next = (char *)workspace + ((nb * sizeof(T) + 255) & ~(size_t)255);
        const size_t used = (size_t)(next - (char *)workspace);
        FLX_CHECK(exclusive_scan<T>(ctx, nb, sums, sums, next, workspace_bytes - used));
    }
    hipLaunchKernelGGL(k_scan_apply<T>, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, ctx->stream, n, in, out, sums);
    FLX_HIP(ctx, hipGetLastError());
    return FLX_OK;
}

// ---------------------------------------------------------------------------------------------
// radix sort
// ---------------------------------------------------------------------------------------------
constexpr int RS_ROUNDS = 16;
constexpr int RS_TILE = 64 * RS_ROUNDS;  // keys per wave tile
constexpr int RS_WAVES = 4;              // waves per block

__global__ void __launch_bounds__(256) k_key_bits(uint64_t n, const uint64_t *keys, unsigned long long *or_and) {
    unsigned long long o = 0, a = ~0ull;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const unsigned long long k = keys[i];
        o |= k;
        a &= k;
    }
    for (int s = 32; s > 0; s >>= 1) {
        o |= __shfl_xor(o, s, 64);
        a &= __shfl_xor(a, s, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicOr(&or_and[0], o);
        atomicAnd(&or_and[1], a);
    }
}

__global__ void __launch_bounds__(RS_WAVES * 64) k_radix_hist(uint64_t n, const uint64_t *keys, int shift,
                                                              uint32_t *table, uint64_t n_tiles) {
    __shared__ uint32_t hist[RS_WAVES][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t tile = (uint64_t)blockIdx.x * RS_WAVES + wave;
    for (int d = lane; d < 256; d += 64) hist[wave][d] = 0;
    __builtin_amdgcn_wave_barrier();
    if (tile < n_tiles) {
        const uint64_t base = tile * RS_TILE;
#pragma unroll 4
        for (int r = 0; r < RS_ROUNDS; ++r) {
            const uint64_t i = base + (uint64_t)r * 64 + lane;
            if (i < n) atomicAdd(&hist[wave][(keys[i] >> shift) & 0xff], 1u);
        }
    }
    __builtin_amdgcn_wave_barrier();
    __syncthreads();
    if (tile < n_tiles)
        for (int d = lane; d < 256; d += 64) table[(uint64_t)d * n_tiles + tile] = hist[wave][d];
}

__global__ void __launch_bounds__(RS_WAVES * 64) k_radix_scatter(uint64_t n, const uint64_t *keys_in,
                                                                 const uint32_t *vals_in, uint64_t *keys_out,
                                                                 uint32_t *vals_out, int shift,
                                                                 const uint32_t *table_scanned, uint64_t n_tiles) {
    __shared__ uint32_t offs[RS_WAVES][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t tile = (uint64_t)blockIdx.x * RS_WAVES + wave;
    if (tile >= n_tiles) return;
    volatile uint32_t *my = offs[wave];
    for (int d = lane; d < 256; d += 64) my[d] = table_scanned[(uint64_t)d * n_tiles + tile];
    __builtin_amdgcn_wave_barrier();
    const uint64_t base = tile * RS_TILE;
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int r = 0; r < RS_ROUNDS; ++r) {
        const uint64_t i = base + (uint64_t)r * 64 + lane;
        const bool act = i < n;
        uint64_t key = 0;
        uint32_t val = 0;
        if (act) {
            key = keys_in[i];
            val = vals_in[i];
        }
        const uint32_t d = (uint32_t)(key >> shift) & 0xffu;
        unsigned long long mask = __ballot(act);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long bal = __ballot(bit);
            mask &= bit ? bal : ~bal;
        }
        if (act) {
            const uint32_t rank = (uint32_t)__popcll(mask & lt);
            const uint32_t b0 = my[d];
            const uint32_t pos = b0 + rank;
            keys_out[pos] = key;
            vals_out[pos] = val;
            if ((mask >> lane) == 1ull) my[d] = b0 + (uint32_t)__popcll(mask);  // highest lane of the group
        }
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace

size_t flx_radix_sort_workspace(uint64_t n) {
    const uint64_t tiles = (n + RS_TILE - 1) / RS_TILE + 1;
    const uint64_t m = 256 * tiles;
    size_t b = m * 4;                                   // table
    b += (m / SCAN_TILE + 4096) * 8 * 2;                // scan aux for the table
    b += (n / SCAN_TILE + 4096) * 8 * 2;                // scan aux for the i64 cut scan
    return b + 65536;
}

int flx_exclusive_scan_i64(flx_ctx *ctx, uint64_t n, const int64_t *in, int64_t *out, void *workspace,
                           size_t workspace_bytes) {
    return exclusive_scan<int64_t>(ctx, n, in, out, workspace, workspace_bytes);
}

int flx_exclusive_scan_f64_approx(flx_ctx *ctx, uint64_t n, double *data, void *workspace) {
    return exclusive_scan<double>(ctx, n, data, data, workspace, (n / SCAN_TILE + 4096) * 8 * 2);
}

int flx_exclusive_scan_u32(flx_ctx *ctx, uint64_t n, const uint32_t *in, uint32_t *out, void *workspace,
                           size_t workspace_bytes) {
    return exclusive_scan<uint32_t>(ctx, n, in, out, workspace, workspace_bytes);
}

int flx_radix_sort_pairs(flx_ctx *ctx, uint64_t n, uint64_t *keys0, uint64_t *keys1, uint32_t *vals0, uint32_t *vals1,
                         void *workspace, size_t workspace_bytes, uint64_t **sorted_keys, uint32_t **sorted_vals) {
    *sorted_keys = keys0;
    *sorted_vals = vals0;
    if (n <= 1) return FLX_OK;
    if (workspace_bytes < flx_radix_sort_workspace(n)) return flx_fail(ctx, FLX_ERR_INVALID, "sort workspace too small");
    hipStream_t st = ctx->stream;
    const uint64_t tiles = (n + RS_TILE - 1) / RS_TILE;
    const uint64_t m = 256 * tiles;
    uint32_t *table = (uint32_t *)workspace;
    char *aux = (char *)workspace + ((m * 4 + 255) & ~(size_t)255);
    const size_t aux_bytes = workspace_bytes - (size_t)(aux - (char *)workspace);

    // which key bits vary at all?
    unsigned long long *d_bits = (unsigned long long *)aux;
    const unsigned long long init[2] = {0ull, ~0ull};
    FLX_HIP(ctx, hipMemcpyAsync(d_bits, init, 16, hipMemcpyHostToDevice, st));
    flx_time_begin(ctx, "flx_sort_keybits");
    hipLaunchKernelGGL(k_key_bits, dim3(1024), dim3(256), 0, st, n, keys0, d_bits);
    flx_time_end(ctx);
    unsigned long long bits[2];
    FLX_HIP(ctx, hipMemcpyAsync(bits, d_bits, 16, hipMemcpyDeviceToHost, st));
    FLX_HIP(ctx, hipStreamSynchronize(st));
    const unsigned long long varying = bits[0] & ~bits[1];

    uint64_t *kin = keys0, *kout = keys1;
    uint32_t *vin = vals0, *vout = vals1;
    const unsigned blocks = (unsigned)((tiles + RS_WAVES - 1) / RS_WAVES);
    for (int pass = 0; pass < 8; ++pass) {
        const int shift = pass * 8;
        if (((varying >> shift) & 0xffull) == 0) continue;
        flx_time_begin(ctx, "flx_sort_pass");
        hipLaunchKernelGGL(k_radix_hist, dim3(blocks), dim3(RS_WAVES * 64), 0, st, n, kin, shift, table, tiles);
        FLX_CHECK(exclusive_scan<uint32_t>(ctx, m, table, table, aux, aux_bytes));
        hipLaunchKernelGGL(k_radix_scatter, dim3(blocks), dim3(RS_WAVES * 64), 0, st, n, kin, vin, kout, vout, shift,
                           table, tiles);
        flx_time_end(ctx);
        std::swap(kin, kout);
        std::swap(vin, vout);
    }
    FLX_HIP(ctx, hipGetLastError());
    *sorted_keys = kin;
    *sorted_vals = vin;
    return FLX_OK;
}
