// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on the access patterns of the NTT passes (MI355X_MICROARCH.md, section HBM:
// "calibrate on a known byte count in your own access pattern before trusting an absolute").  Every kernel moves a KNOWN number
// of bytes with the loads and stores of ntt.hip (two 16-byte halves per 32-byte element, lanes 32 bytes apart):
//   contig   : consecutive lanes read consecutive elements (the s = 0 pass);
//   rows<G>  : tiles of R rows x G adjacent elements, rows 2^s elements apart (the strided passes: G = 8 -> 256-byte rows, G = 4 -> 128).
// The footprint (1 GiB) is far beyond the 256 MiB infinity cache.  Prints the useful bytes and the achieved rate of every
// kernel; run it under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes) to read the counters per kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct alignas(16) fe { uint32_t v[8]; };
__device__ __forceinline__ fe ld_fe(const fe* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 lo = q[0], hi = q[1];
    fe r; r.v[0] = lo.x; r.v[1] = lo.y; r.v[2] = lo.z; r.v[3] = lo.w; r.v[4] = hi.x; r.v[5] = hi.y; r.v[6] = hi.z; r.v[7] = hi.w;
    return r;
}
__device__ __forceinline__ void st_fe(fe* p, const fe& a) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(a.v[0], a.v[1], a.v[2], a.v[3]);
    q[1] = make_uint4(a.v[4], a.v[5], a.v[6], a.v[7]);
}
// MODE 0: read only (xor-reduced so the loads stay), 1: read + write back in place
template <int MODE>
__global__ void __launch_bounds__(256) contig_kernel(fe* data, uint32_t* sink) {
    const uint64_t base = (uint64_t)blockIdx.x * 1024;
    uint32_t acc = 0;
    for (int q = 0; q < 4; ++q) {
        fe x = ld_fe(data + base + threadIdx.x + q * 256);
        if (MODE) { x.v[0] ^= 1u; st_fe(data + base + threadIdx.x + q * 256, x); }
        else acc ^= x.v[0] ^ x.v[7];
    }
    if (!MODE && acc == 0x12345678u) sink[0] = acc;
}
// tile = 1024 elements: R = 1024 / G rows of G adjacent elements, rows (1 << s) elements apart; tiles cover the array exactly
// like ntt_pass_kernel's strided tiles: position = (hi << (s + r)) + (t << s) + lo0 + gl
template <int MODE, int GLOG>
__global__ void __launch_bounds__(256) rows_kernel(fe* data, uint32_t s, uint32_t* sink) {
    constexpr uint32_t G = 1u << GLOG, r = 10 - GLOG;
    const uint32_t tile = blockIdx.x;
    const uint32_t lo_tiles = 1u << (s - GLOG);
    const uint32_t lo0 = (tile & (lo_tiles - 1)) << GLOG, hi = tile >> (s - GLOG);
    uint32_t acc = 0;
    for (int q = 0; q < 4; ++q) {
        const uint32_t e = threadIdx.x + q * 256, gl = e & (G - 1), t = e >> GLOG;
        const uint64_t pos = ((uint64_t)hi << (s + r)) + ((uint64_t)t << s) + lo0 + gl;
        fe x = ld_fe(data + pos);
        if (MODE) { x.v[0] ^= 1u; st_fe(data + pos, x); }
        else acc ^= x.v[0] ^ x.v[7];
    }
    if (!MODE && acc == 0x12345678u) sink[0] = acc;
}
template <class F> static float timed(F launch, int reps) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) launch();
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}
int main() {
    const uint32_t logn = 25;                      // 2^25 elements x 32 B = 1 GiB
    const uint64_t n = 1ull << logn, bytes = n * 32;
    fe* d; uint32_t* sink;
    CHECK(hipMalloc(&d, bytes)); CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(d, 0x5a, bytes));
    const unsigned tiles = (unsigned)(n >> 10);
    const int reps = 5;
    auto report = [&](const char* name, float ms, double moved) {
        printf("%-44s %8.3f ms  useful bytes per launch %.0f  -> %7.1f GB/s\n", name, ms, moved, moved / ms / 1e6);
    };
    report("contig read", timed([&] { hipLaunchKernelGGL(contig_kernel<0>, dim3(tiles), dim3(256), 0, 0, d, sink); }, reps), (double)bytes);
    report("contig read+write", timed([&] { hipLaunchKernelGGL(contig_kernel<1>, dim3(tiles), dim3(256), 0, 0, d, sink); }, reps), 2.0 * bytes);
    for (uint32_t s : {8u, 15u}) {
        char nm[96];
        snprintf(nm, sizeof nm, "rows 256 B (G=8), stride 2^%u: read", s);
        report(nm, timed([&] { hipLaunchKernelGGL((rows_kernel<0, 3>), dim3(tiles), dim3(256), 0, 0, d, s, sink); }, reps), (double)bytes);
        snprintf(nm, sizeof nm, "rows 256 B (G=8), stride 2^%u: read+write", s);
        report(nm, timed([&] { hipLaunchKernelGGL((rows_kernel<1, 3>), dim3(tiles), dim3(256), 0, 0, d, s, sink); }, reps), 2.0 * bytes);
        snprintf(nm, sizeof nm, "rows 128 B (G=4), stride 2^%u: read", s);
        report(nm, timed([&] { hipLaunchKernelGGL((rows_kernel<0, 2>), dim3(tiles), dim3(256), 0, 0, d, s, sink); }, reps), (double)bytes);
        snprintf(nm, sizeof nm, "rows 128 B (G=4), stride 2^%u: read+write", s);
        report(nm, timed([&] { hipLaunchKernelGGL((rows_kernel<1, 2>), dim3(tiles), dim3(256), 0, 0, d, s, sink); }, reps), 2.0 * bytes);
        snprintf(nm, sizeof nm, "rows 64 B (G=2), stride 2^%u: read", s);
        report(nm, timed([&] { hipLaunchKernelGGL((rows_kernel<0, 1>), dim3(tiles), dim3(256), 0, 0, d, s, sink); }, reps), (double)bytes);
        snprintf(nm, sizeof nm, "rows 32 B (G=1), stride 2^%u: read", s);
        report(nm, timed([&] { hipLaunchKernelGGL((rows_kernel<0, 0>), dim3(tiles), dim3(256), 0, 0, d, s, sink); }, reps), (double)bytes);
    }
    return 0;
}
