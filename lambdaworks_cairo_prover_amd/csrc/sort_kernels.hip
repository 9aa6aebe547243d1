// See sort_kernels.h.
#include "sort_kernels.h"
#include <utility>

namespace sp {

constexpr uint32_t RS_WAVE = 64;            // one wave per work-group: the stable rank of an element is a wave-wide digit match
constexpr uint32_t RS_ITEMS = 32;           // elements per lane and pass
constexpr uint32_t RS_TILE = RS_WAVE * RS_ITEMS;
constexpr uint32_t RS_DIGITS = 256;
constexpr uint32_t SCAN_SEG = 1024;         // entries per work-group of the histogram scan

// lanes of the wave whose 8-bit digit equals this lane's (among the lanes of `valid`)
__device__ __forceinline__ uint64_t match_digit(uint32_t d, uint64_t valid) {
    uint64_t m = valid;
#pragma unroll
    for (int bit = 0; bit < 8; ++bit) {
        const uint64_t b = __ballot((d >> bit) & 1u);
        m &= ((d >> bit) & 1u) ? b : ~b;
    }
    return m;
}

// ghist[digit * nblocks + block] = elements of the block's tile whose digit (key >> shift) & 255 is `digit`
__global__ void __launch_bounds__(RS_WAVE) rs_hist_kernel(const uint64_t* __restrict__ keys, uint64_t n, uint32_t shift, uint32_t* __restrict__ ghist, uint32_t nblocks) {
    __shared__ volatile uint32_t hist[RS_DIGITS];
    const uint32_t lane = threadIdx.x;
    for (uint32_t d = lane; d < RS_DIGITS; d += RS_WAVE) hist[d] = 0;
    __builtin_amdgcn_wave_barrier();
    const uint64_t base = (uint64_t)blockIdx.x * RS_TILE;
    const uint64_t lt = (1ULL << lane) - 1ULL;
    for (uint32_t k = 0; k < RS_ITEMS; ++k) {
        const uint64_t e = base + (uint64_t)k * RS_WAVE + lane;
        const bool ok = e < n;
        const uint32_t d = ok ? (uint32_t)(keys[e] >> shift) & 255u : 0u;
        const uint64_t m = match_digit(d, __ballot(ok));
        if (ok && (m & lt) == 0) hist[d] = hist[d] + (uint32_t)__popcll(m);      // the lowest lane of every digit class adds the class
        __builtin_amdgcn_wave_barrier();
    }
    for (uint32_t d = lane; d < RS_DIGITS; d += RS_WAVE) ghist[(uint64_t)d * nblocks + blockIdx.x] = hist[d];
}

// exclusive prefix sums of `count` uint32 entries in three steps: every work-group scans a segment of 1024 entries in place and
// reports its total; one work-group scans the totals; the totals are added back
__device__ __forceinline__ uint32_t block_scan_256(uint32_t v, uint32_t* sh, uint32_t* total) {
    const uint32_t t = threadIdx.x;
    sh[t] = v;
    __syncthreads();
    for (uint32_t off = 1; off < 256; off <<= 1) {
        const uint32_t o = t >= off ? sh[t - off] : 0u;
        __syncthreads();
        sh[t] += o;
        __syncthreads();
    }
    if (total) *total = sh[255];
    return sh[t] - v;   // exclusive
}
__global__ void __launch_bounds__(256) scan_segments_kernel(uint32_t* data, uint64_t count, uint32_t* totals) {
    __shared__ uint32_t sh[256];
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_SEG + (uint64_t)threadIdx.x * 4;
    uint32_t v[4], s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { v[k] = base + k < count ? data[base + k] : 0u; s += v[k]; }
    uint32_t tot;
    uint32_t run = block_scan_256(s, sh, &tot);
#pragma unroll
    for (int k = 0; k < 4; ++k) { if (base + k < count) data[base + k] = run; run += v[k]; }
    if (threadIdx.x == 0) totals[blockIdx.x] = tot;
}
__global__ void __launch_bounds__(256) scan_totals_kernel(uint32_t* totals, uint64_t count) {
    __shared__ uint32_t sh[256];
    const uint64_t per = (count + 255) / 256, base = (uint64_t)threadIdx.x * per;
    uint32_t s = 0;
    for (uint64_t k = 0; k < per; ++k) if (base + k < count) s += totals[base + k];
    uint32_t run = block_scan_256(s, sh, nullptr);
    for (uint64_t k = 0; k < per; ++k)
        if (base + k < count) { const uint32_t cur = totals[base + k]; totals[base + k] = run; run += cur; }
}
__global__ void __launch_bounds__(256) scan_add_kernel(uint32_t* data, uint64_t count, const uint32_t* totals) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < count) data[i] += totals[i / SCAN_SEG];
}
static int exclusive_scan_u32(hipStream_t st, uint32_t* data, uint64_t count, uint32_t* totals) {
    const uint64_t segs = (count + SCAN_SEG - 1) / SCAN_SEG;
    hipLaunchKernelGGL(scan_segments_kernel, dim3((unsigned)segs), dim3(256), 0, st, data, count, totals);
    hipLaunchKernelGGL(scan_totals_kernel, dim3(1), dim3(256), 0, st, totals, segs);
    hipLaunchKernelGGL(scan_add_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, data, count, totals);
    SP_HIP_CHECK(hipGetLastError());
    return SP_OK;
}

// stable scatter: the elements of a tile are taken 64 at a time in index order; an element goes to
// (start of its digit in this tile) + (elements of that digit before it in the tile)
__global__ void __launch_bounds__(RS_WAVE) rs_scatter_kernel(const uint64_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, uint64_t* __restrict__ keys_out,
                                                            uint32_t* __restrict__ vals_out, uint64_t n, uint32_t shift, const uint32_t* __restrict__ ghist, uint32_t nblocks) {
    __shared__ volatile uint32_t next[RS_DIGITS];
    const uint32_t lane = threadIdx.x;
    for (uint32_t d = lane; d < RS_DIGITS; d += RS_WAVE) next[d] = ghist[(uint64_t)d * nblocks + blockIdx.x];
    __builtin_amdgcn_wave_barrier();
    const uint64_t base = (uint64_t)blockIdx.x * RS_TILE;
    const uint64_t lt = (1ULL << lane) - 1ULL;
    for (uint32_t k = 0; k < RS_ITEMS; ++k) {
        const uint64_t e = base + (uint64_t)k * RS_WAVE + lane;
        const bool ok = e < n;
        const uint64_t key = ok ? keys_in[e] : 0ULL;
        const uint32_t val = ok ? vals_in[e] : 0u;
        const uint32_t d = (uint32_t)(key >> shift) & 255u;
        const uint64_t m = match_digit(d, __ballot(ok));
        const uint32_t start = next[d];                       // (read by the whole class before its lowest lane moves it on)
        __builtin_amdgcn_wave_barrier();
        if (ok) {
            const uint32_t pos = start + (uint32_t)__popcll(m & lt);
            keys_out[pos] = key;
            vals_out[pos] = val;
            if ((m & lt) == 0) next[d] = start + (uint32_t)__popcll(m);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

size_t radix_sort_workspace_bytes(uint64_t n) {
    const uint64_t nblocks = (n + RS_TILE - 1) / RS_TILE, entries = nblocks * RS_DIGITS;
    return (size_t)(entries * sizeof(uint32_t) + ((entries + SCAN_SEG - 1) / SCAN_SEG + 1) * sizeof(uint32_t) + 512);
}

int radix_sort_pairs_u64(hipStream_t st, uint64_t* keys_in, uint64_t* keys_out, uint32_t* vals_in, uint32_t* vals_out, uint64_t n, uint32_t key_bits,
                         void* workspace) {
    if (n == 0) return SP_OK;
    if (n >= (1ULL << 32) || key_bits == 0 || key_bits > 64) return SP_E_INVALID_ARG;
    const uint32_t nblocks = (uint32_t)((n + RS_TILE - 1) / RS_TILE);
    const uint64_t entries = (uint64_t)nblocks * RS_DIGITS;
    uint32_t* ghist = static_cast<uint32_t*>(workspace);
    uint32_t* totals = ghist + ((entries + 63) & ~63ULL);
    const uint32_t passes = (key_bits + 7) / 8;
    uint64_t *ki = keys_in, *ko = keys_out;
    uint32_t *vi = vals_in, *vo = vals_out;
    for (uint32_t p = 0; p < passes; ++p) {
        hipLaunchKernelGGL(rs_hist_kernel, dim3(nblocks), dim3(RS_WAVE), 0, st, ki, n, 8 * p, ghist, nblocks);
        SP_TRY(exclusive_scan_u32(st, ghist, entries, totals));
        hipLaunchKernelGGL(rs_scatter_kernel, dim3(nblocks), dim3(RS_WAVE), 0, st, ki, vi, ko, vo, n, 8 * p, ghist, nblocks);
        SP_HIP_CHECK(hipGetLastError());
        std::swap(ki, ko); std::swap(vi, vo);
    }
    if (ki != keys_out) {   // an even number of passes left the result in the input arrays
        SP_HIP_CHECK(hipMemcpyAsync(keys_out, ki, n * sizeof(uint64_t), hipMemcpyDeviceToDevice, st));
        SP_HIP_CHECK(hipMemcpyAsync(vals_out, vi, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
    }
    return SP_OK;
}

// ---- counting sort of 16-bit keys (a Cairo trace uses a handful of distinct offsets: the wave adds a whole class of equal keys
// with one atomic instead of serialising 3n atomics on a few addresses)
// A wave adds every class of equal keys with one add.  A Cairo trace uses a handful of distinct offsets, so adding straight into the
// global histogram sends every wave's few adds to the same few addresses - 250 000 serialized atomics, 0.73 ms at 2^20 rows (round 3).
// A work-group therefore takes CS_CHUNK keys and collects its classes in a small open-addressing table in LDS (key -> count; a
// class that finds no slot within eight probes goes to the global histogram directly, so any key distribution stays correct),
// and flushes the table once: a few hundred global atomics per launch instead.
constexpr uint32_t CS_CHUNK = 256 * 32, CS_TAB = 1024, CS_EMPTY = 0xffffffffu;
__global__ void __launch_bounds__(256) cs_hist_kernel(const uint16_t* __restrict__ keys, uint64_t n, uint32_t* __restrict__ hist) {
    __shared__ uint32_t tab_key[CS_TAB], tab_cnt[CS_TAB];
    for (uint32_t k = threadIdx.x; k < CS_TAB; k += 256) { tab_key[k] = CS_EMPTY; tab_cnt[k] = 0; }
    __syncthreads();
    const uint64_t base = (uint64_t)blockIdx.x * CS_CHUNK;
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t it = 0; it < CS_CHUNK / 256; ++it) {
        const uint64_t e = base + (uint64_t)it * 256 + threadIdx.x;
        const bool ok = e < n;
        const uint32_t v = ok ? keys[e] : 0u;
        uint64_t m = __ballot(ok);
#pragma unroll
        for (int bit = 0; bit < 16; ++bit) {
            const uint64_t b = __ballot((v >> bit) & 1u);
            m &= ((v >> bit) & 1u) ? b : ~b;
        }
        if (ok && (m & ((1ULL << lane) - 1ULL)) == 0) {      // the first lane of its class
            const uint32_t cnt = (uint32_t)__popcll(m);
            uint32_t h = (v * 2654435761u) >> 22;
            bool placed = false;
            for (int probe = 0; probe < 8 && !placed; ++probe, h = (h + 1) & (CS_TAB - 1)) {
                const uint32_t prev = atomicCAS(&tab_key[h], CS_EMPTY, v);
                if (prev == CS_EMPTY || prev == v) { atomicAdd(&tab_cnt[h], cnt); placed = true; }
            }
            if (!placed) atomicAdd(hist + v, cnt);
        }
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < CS_TAB; k += 256)
        if (tab_key[k] != CS_EMPTY) atomicAdd(hist + tab_key[k], tab_cnt[k]);
}
// single work-group: hist[0 .. 65536] -> exclusive prefix sums, hist[65536] = n
__global__ void __launch_bounds__(256) cs_scan_kernel(uint32_t* hist) {
    __shared__ uint32_t sh[256];
    const uint32_t base = threadIdx.x * 256;
    uint32_t s = 0;
    for (uint32_t k = 0; k < 256; ++k) s += hist[base + k];
    uint32_t tot;
    uint32_t run = block_scan_256(s, sh, &tot);
    for (uint32_t k = 0; k < 256; ++k) { const uint32_t cur = hist[base + k]; hist[base + k] = run; run += cur; }
    if (threadIdx.x == 0) hist[65536] = tot;
}
// out[i] = the largest v with start[v] <= i.  One binary search over the 65 537 prefix sums per OUTPUT is sixteen dependent loads for
// every one of the 3n elements (0.9 ms at 2^20 rows, round 3); the outputs of a work-group are consecutive, so two searches - for
// its first and its last index - bracket the values all of them can take: the same value for the whole group when it lies inside a
// run (nearly always: a trace has a handful of distinct offsets), a short search between the two otherwise.
__device__ __forceinline__ uint32_t cs_search(const uint32_t* __restrict__ start, uint32_t i, uint32_t lo, uint32_t hi) {
    while (lo < hi) {
        const uint32_t mid = (lo + hi + 1) >> 1;
        if (start[mid] <= i) lo = mid; else hi = mid - 1;
    }
    return lo;
}
__global__ void __launch_bounds__(256) cs_expand_kernel(const uint32_t* __restrict__ start, uint64_t n, uint16_t* __restrict__ out) {
    __shared__ uint32_t bounds[2];
    const uint64_t i0 = (uint64_t)blockIdx.x * 256;
    if (threadIdx.x < 2) {
        const uint64_t i = threadIdx.x == 0 ? i0 : (i0 + 255 < n ? i0 + 255 : n - 1);
        bounds[threadIdx.x] = cs_search(start, (uint32_t)i, 0, 65535);
    }
    __syncthreads();
    const uint64_t i = i0 + threadIdx.x;
    if (i >= n) return;
    const uint32_t lo = bounds[0], hi = bounds[1];
    out[i] = (uint16_t)(lo == hi ? lo : cs_search(start, (uint32_t)i, lo, hi));
}

int counting_sort_u16(hipStream_t st, const uint16_t* keys, uint16_t* out, uint64_t n, uint32_t* hist) {
    if (n == 0) return SP_OK;
    if (n >= (1ULL << 32)) return SP_E_INVALID_ARG;
    SP_HIP_CHECK(hipMemsetAsync(hist, 0, 65537 * sizeof(uint32_t), st));
    hipLaunchKernelGGL(cs_hist_kernel, dim3((unsigned)((n + CS_CHUNK - 1) / CS_CHUNK)), dim3(256), 0, st, keys, n, hist);
    hipLaunchKernelGGL(cs_scan_kernel, dim3(1), dim3(256), 0, st, hist);
    hipLaunchKernelGGL(cs_expand_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, hist, n, out);
    SP_HIP_CHECK(hipGetLastError());
    return SP_OK;
}

}  // namespace sp
