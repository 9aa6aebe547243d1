// Host-side microbenchmark behind prover_upload.cpp's group widths: T threads take a row-major table of 2^19 x 34 field elements
// apart, (a) as bitmaps of the 16 flag columns, cw columns per pass, (b) as column-major copies of the 18 other columns, cw per
// pass, (c) one sequential read of the whole table.  g++ -O3 -march=native -pthread gather_bench.cpp && ./a.out [threads]
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void pack(const uint8_t* src, uint64_t n, size_t row_bytes, size_t off, uint32_t cw, uint8_t* dst, uint64_t r0, uint64_t r1) {
    const uint64_t one[4] = {1, 0, 0, 0};
    for (uint64_t i = r0; i < r1; i += 64) {
        uint64_t words[16] = {0};
        const uint8_t* s = src + i * row_bytes + off;
        for (uint64_t k = 0; k < 64; ++k, s += row_bytes) {
            for (uint32_t l = 0; l < cw * 32; l += 64) __builtin_prefetch(s + 24 * row_bytes + l, 0, 0);
            for (uint32_t u = 0; u < cw; ++u) {
                uint64_t v[4];
                __builtin_memcpy(v, s + 32 * (size_t)u, 32);
                words[u] |= (uint64_t)(((v[0] ^ one[0]) | (v[1] ^ one[1]) | (v[2] ^ one[2]) | (v[3] ^ one[3])) == 0) << k;
            }
        }
        for (uint32_t u = 0; u < cw; ++u) __builtin_memcpy(dst + (size_t)u * (n / 8) + (i / 64) * 8, &words[u], 8);
    }
}
static void gather(const uint8_t* src, uint64_t n, size_t row_bytes, size_t off, uint32_t cw, uint8_t* dst, uint64_t r0, uint64_t r1) {
    const uint8_t* s = src + r0 * row_bytes + off;
    for (uint64_t i = r0; i < r1; ++i, s += row_bytes) {
        for (uint32_t l = 0; l < cw * 32; l += 64) __builtin_prefetch(s + 24 * row_bytes + l, 0, 0);
        for (uint32_t u = 0; u < cw; ++u) __builtin_memcpy(dst + ((size_t)u * n + i) * 32, s + 32 * (size_t)u, 32);
    }
}
template <class F> static double run(int T, uint64_t n, uint64_t br, F f) {
    double best = 1e30;
    for (int rep = 0; rep < 4; ++rep) {
        std::atomic<uint64_t> next{0};
        std::atomic<int> go{0};
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t) th.emplace_back([&] { while (!go.load()) {} for (;;) { uint64_t b = next.fetch_add(1); if (b * br >= n) return; f(b * br, std::min(n, (b + 1) * br)); } });
        const double t0 = now_ms();
        go.store(1);
        for (auto& t : th) t.join();
        best = std::min(best, now_ms() - t0);
    }
    return best;
}
int main(int argc, char** argv) {
    const uint64_t n = 1 << 19; const uint32_t cols = 34; const size_t rb = cols * 32;
    const int T = argc > 1 ? atoi(argv[1]) : 8;
    std::vector<uint8_t> tab(n * rb), out((size_t)18 * n * 32);
    for (size_t i = 0; i < tab.size(); i += 8) tab[i] = (uint8_t)(i * 2654435761u >> 13);
    std::memset(out.data(), 1, out.size());
    for (uint32_t cw : {2u, 4u, 8u, 16u}) {
        double ms = 0;
        for (uint32_t c = 0; c < 16; c += cw) ms += run(T, n, 1024, [&](uint64_t r0, uint64_t r1) { pack(tab.data(), n, rb, c * 32, cw, out.data() + (size_t)c * n / 8, r0, r1); });
        printf("threads %2d  bitmaps of 16 columns, %2u per pass: %6.2f ms (%.1f GB/s read)\n", T, cw, ms, n * 512.0 / ms * 1e-6);
    }
    for (uint32_t cw : {2u, 6u, 18u}) {
        double ms = 0;
        for (uint32_t c = 0; c < 18; c += cw) ms += run(T, n, 1024, [&](uint64_t r0, uint64_t r1) { gather(tab.data(), n, rb, (16 + c) * 32, cw, out.data() + (size_t)c * n * 32, r0, r1); });
        printf("threads %2d  copies of 18 columns,  %2u per pass: %6.2f ms (%.1f GB/s read + as much written)\n", T, cw, ms, n * 576.0 / ms * 1e-6);
    }
    std::atomic<uint64_t> sink{0};
    double ms = run(T, n, 1024, [&](uint64_t r0, uint64_t r1) { uint64_t a = 0; const uint64_t* p = (const uint64_t*)(tab.data() + r0 * rb); for (size_t k = 0; k < (r1 - r0) * rb / 8; ++k) a += p[k]; sink += a; });
    printf("threads %2d  sequential read of the table:      %6.2f ms (%.1f GB/s)\n", T, ms, n * (double)rb / ms * 1e-6);
}
