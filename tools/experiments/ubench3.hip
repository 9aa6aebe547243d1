// Issue cost of individual gfx950 VALU instructions in exact (inline-asm) streams: which forms of the 64-bit
// multiply-add, carry chains and 3-operand logic ops are cheap. 8 waves per SIMD, registers only.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench3.hip -o /tmp/ubench3 && /tmp/ubench3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITERS 512

#define REP8(X) X X X X X X X X

template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t* out, const uint32_t* in) {
    uint32_t a = in[threadIdx.x], b = in[threadIdx.x + 256], c = in[threadIdx.x + 512], d = in[threadIdx.x + 768];
    uint32_t e = a ^ 0x1234567u, f = b ^ 0x89abcdu, g = c + 77u, h = d + 99u;
    uint64_t A0 = a, A1 = b, A2 = c, A3 = d, A4 = e, A5 = f, A6 = g, A7 = h;
    uint64_t T0 = a + 1, T1 = b + 2, T2 = c + 3, T3 = d + 4;
    for (int it = 0; it < ITERS; ++it) {
        if (OP == 0) {  // VOP2 xor, 8 independent chains x 8
            REP8(asm volatile("v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %5\n v_xor_b32 %2, %2, %6\n v_xor_b32 %3, %3, %7\n"
                              "v_xor_b32 %4, %4, %0\n v_xor_b32 %5, %5, %1\n v_xor_b32 %6, %6, %2\n v_xor_b32 %7, %7, %3"
                              : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));)
        } else if (OP == 1) {  // mad, accumulate in place, 8 independent accumulators
            REP8(asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %10, %1\n v_mad_u64_u32 %2, vcc, %8, %11, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n"
                              "v_mad_u64_u32 %4, vcc, %8, %10, %4\n v_mad_u64_u32 %5, vcc, %8, %11, %5\n v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %10, %7"
                              : "+v"(A0), "+v"(A1), "+v"(A2), "+v"(A3), "+v"(A4), "+v"(A5), "+v"(A6), "+v"(A7) : "v"(a), "v"(b), "v"(c), "v"(d) : "vcc");)
        } else if (OP == 2) {  // mad, addend from a different pair (D = a*b + T), results overwritten (no chain)
            REP8(asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %12\n v_mad_u64_u32 %1, vcc, %8, %10, %13\n v_mad_u64_u32 %2, vcc, %8, %11, %14\n v_mad_u64_u32 %3, vcc, %8, %9, %15\n"
                              "v_mad_u64_u32 %4, vcc, %8, %10, %12\n v_mad_u64_u32 %5, vcc, %8, %11, %13\n v_mad_u64_u32 %6, vcc, %8, %9, %14\n v_mad_u64_u32 %7, vcc, %8, %10, %15"
                              : "+v"(A0), "+v"(A1), "+v"(A2), "+v"(A3), "+v"(A4), "+v"(A5), "+v"(A6), "+v"(A7) : "v"(a), "v"(b), "v"(c), "v"(d), "v"(T0), "v"(T1), "v"(T2), "v"(T3) : "vcc");)
        } else if (OP == 3) {  // mad with inline-constant addend 0
            REP8(asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, 0\n v_mad_u64_u32 %1, vcc, %8, %10, 0\n v_mad_u64_u32 %2, vcc, %8, %11, 0\n v_mad_u64_u32 %3, vcc, %8, %9, 0\n"
                              "v_mad_u64_u32 %4, vcc, %8, %10, 0\n v_mad_u64_u32 %5, vcc, %8, %11, 0\n v_mad_u64_u32 %6, vcc, %8, %9, 0\n v_mad_u64_u32 %7, vcc, %8, %10, 0"
                              : "+v"(A0), "+v"(A1), "+v"(A2), "+v"(A3), "+v"(A4), "+v"(A5), "+v"(A6), "+v"(A7) : "v"(a), "v"(b), "v"(c), "v"(d) : "vcc");)
        } else if (OP == 4) {  // mad with SGPR-pair carry-out other than vcc
            REP8(asm volatile("v_mad_u64_u32 %0, s[20:21], %8, %9, %0\n v_mad_u64_u32 %1, s[22:23], %8, %10, %1\n v_mad_u64_u32 %2, s[24:25], %8, %11, %2\n v_mad_u64_u32 %3, s[26:27], %8, %9, %3\n"
                              "v_mad_u64_u32 %4, s[20:21], %8, %10, %4\n v_mad_u64_u32 %5, s[22:23], %8, %11, %5\n v_mad_u64_u32 %6, s[24:25], %8, %9, %6\n v_mad_u64_u32 %7, s[26:27], %8, %10, %7"
                              : "+v"(A0), "+v"(A1), "+v"(A2), "+v"(A3), "+v"(A4), "+v"(A5), "+v"(A6), "+v"(A7) : "v"(a), "v"(b), "v"(c), "v"(d)
                              : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");)
        } else if (OP == 5) {  // v_mul_lo_u32
            REP8(asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %5\n v_mul_lo_u32 %2, %2, %6\n v_mul_lo_u32 %3, %3, %7\n"
                              "v_mul_lo_u32 %4, %4, %0\n v_mul_lo_u32 %5, %5, %1\n v_mul_lo_u32 %6, %6, %2\n v_mul_lo_u32 %7, %7, %3"
                              : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));)
        } else if (OP == 6) {  // v_mul_hi_u32
            REP8(asm volatile("v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %5\n v_mul_hi_u32 %2, %2, %6\n v_mul_hi_u32 %3, %3, %7\n"
                              "v_mul_hi_u32 %4, %4, %0\n v_mul_hi_u32 %5, %5, %1\n v_mul_hi_u32 %6, %6, %2\n v_mul_hi_u32 %7, %7, %3"
                              : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));)
        } else if (OP == 7) {  // carry chain: v_add_co_u32 + 7 x v_addc_co_u32 (vcc)
            REP8(asm volatile("v_add_co_u32 %0, vcc, %0, %4\n v_addc_co_u32 %1, vcc, %1, %5, vcc\n v_addc_co_u32 %2, vcc, %2, %6, vcc\n v_addc_co_u32 %3, vcc, %3, %7, vcc\n"
                              "v_addc_co_u32 %4, vcc, %4, %0, vcc\n v_addc_co_u32 %5, vcc, %5, %1, vcc\n v_addc_co_u32 %6, vcc, %6, %2, vcc\n v_addc_co_u32 %7, vcc, %7, %3, vcc"
                              : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : : "vcc");)
        } else if (OP == 8) {  // VOP3-encoded xor (same op as OP 0, 64-bit encoding)
            REP8(asm volatile("v_xor_b32_e64 %0, %0, %4\n v_xor_b32_e64 %1, %1, %5\n v_xor_b32_e64 %2, %2, %6\n v_xor_b32_e64 %3, %3, %7\n"
                              "v_xor_b32_e64 %4, %4, %0\n v_xor_b32_e64 %5, %5, %1\n v_xor_b32_e64 %6, %6, %2\n v_xor_b32_e64 %7, %7, %3"
                              : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));)
        } else if (OP == 9) {  // v_alignbit_b32 with constant shift
            REP8(asm volatile("v_alignbit_b32 %0, %0, %4, 7\n v_alignbit_b32 %1, %1, %5, 7\n v_alignbit_b32 %2, %2, %6, 7\n v_alignbit_b32 %3, %3, %7, 7\n"
                              "v_alignbit_b32 %4, %4, %0, 9\n v_alignbit_b32 %5, %5, %1, 9\n v_alignbit_b32 %6, %6, %2, 9\n v_alignbit_b32 %7, %7, %3, 9"
                              : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));)
        } else if (OP == 10) {  // v_bfi_b32, three VGPR sources
            REP8(asm volatile("v_bfi_b32 %0, %1, %0, %4\n v_bfi_b32 %1, %2, %1, %5\n v_bfi_b32 %2, %3, %2, %6\n v_bfi_b32 %3, %0, %3, %7\n"
                              "v_bfi_b32 %4, %5, %4, %0\n v_bfi_b32 %5, %6, %5, %1\n v_bfi_b32 %6, %7, %6, %2\n v_bfi_b32 %7, %4, %7, %3"
                              : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));)
        } else if (OP == 11) {  // v_add3_u32
            REP8(asm volatile("v_add3_u32 %0, %1, %0, %4\n v_add3_u32 %1, %2, %1, %5\n v_add3_u32 %2, %3, %2, %6\n v_add3_u32 %3, %0, %3, %7\n"
                              "v_add3_u32 %4, %5, %4, %0\n v_add3_u32 %5, %6, %5, %1\n v_add3_u32 %6, %7, %6, %2\n v_add3_u32 %7, %4, %7, %3"
                              : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));)
        } else if (OP == 12) {  // v_mov_b32
            REP8(asm volatile("v_mov_b32 %0, %4\n v_mov_b32 %1, %5\n v_mov_b32 %2, %6\n v_mov_b32 %3, %7\n"
                              "v_mov_b32 %4, %1\n v_mov_b32 %5, %2\n v_mov_b32 %6, %3\n v_mov_b32 %7, %0"
                              : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));)
        } else if (OP == 13) {  // v_lshl_add_u64 (64-bit add, no carry-out)
            REP8(asm volatile("v_lshl_add_u64 %0, %4, 0, %0\n v_lshl_add_u64 %1, %5, 0, %1\n v_lshl_add_u64 %2, %6, 0, %2\n v_lshl_add_u64 %3, %7, 0, %3\n"
                              "v_lshl_add_u64 %4, %0, 0, %4\n v_lshl_add_u64 %5, %1, 0, %5\n v_lshl_add_u64 %6, %2, 0, %6\n v_lshl_add_u64 %7, %3, 0, %7"
                              : "+v"(A0), "+v"(A1), "+v"(A2), "+v"(A3), "+v"(A4), "+v"(A5), "+v"(A6), "+v"(A7));)
        } else if (OP == 14) {  // v_mul_u32_u24 (VOP2)
            REP8(asm volatile("v_mul_u32_u24 %0, %0, %4\n v_mul_u32_u24 %1, %1, %5\n v_mul_u32_u24 %2, %2, %6\n v_mul_u32_u24 %3, %3, %7\n"
                              "v_mul_u32_u24 %4, %4, %0\n v_mul_u32_u24 %5, %5, %1\n v_mul_u32_u24 %6, %6, %2\n v_mul_u32_u24 %7, %7, %3"
                              : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));)
        } else if (OP == 15) {  // mad (accumulating) interleaved 1:1 with VOP2 xor
            REP8(asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_xor_b32 %12, %12, %9\n v_mad_u64_u32 %1, vcc, %8, %10, %1\n v_xor_b32 %13, %13, %10\n"
                              "v_mad_u64_u32 %2, vcc, %8, %11, %2\n v_xor_b32 %14, %14, %11\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n v_xor_b32 %15, %15, %9\n"
                              "v_mad_u64_u32 %4, vcc, %8, %10, %4\n v_xor_b32 %12, %12, %10\n v_mad_u64_u32 %5, vcc, %8, %11, %5\n v_xor_b32 %13, %13, %11\n"
                              "v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_xor_b32 %14, %14, %9\n v_mad_u64_u32 %7, vcc, %8, %10, %7\n v_xor_b32 %15, %15, %10"
                              : "+v"(A0), "+v"(A1), "+v"(A2), "+v"(A3), "+v"(A4), "+v"(A5), "+v"(A6), "+v"(A7) : "v"(a), "v"(b), "v"(c), "v"(d), "v"(e), "v"(f), "v"(g), "v"(h) : "vcc");)
        } else if (OP == 16) {  // v_mad_u32_u24 (VOP3, 32-bit result)
            REP8(asm volatile("v_mad_u32_u24 %0, %1, %4, %0\n v_mad_u32_u24 %1, %2, %5, %1\n v_mad_u32_u24 %2, %3, %6, %2\n v_mad_u32_u24 %3, %0, %7, %3\n"
                              "v_mad_u32_u24 %4, %5, %0, %4\n v_mad_u32_u24 %5, %6, %1, %5\n v_mad_u32_u24 %6, %7, %2, %6\n v_mad_u32_u24 %7, %4, %3, %7"
                              : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));)
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a ^ b ^ c ^ d ^ e ^ f ^ g ^ h ^ (uint32_t)(A0 ^ A1 ^ A2 ^ A3 ^ A4 ^ A5 ^ A6 ^ A7) ^ (uint32_t)((A0 ^ A1 ^ A2 ^ A3 ^ A4 ^ A5 ^ A6 ^ A7) >> 32);
}

template <int OP>
void run(const char* name, uint32_t* d_out, uint32_t* d_in, int blocks_per_cu) {
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    dim3 grid(prop.multiProcessorCount * blocks_per_cu), block(256);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<OP>), grid, block, 0, 0, d_out, d_in); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<OP>), grid, block, 0, 0, d_out, d_in);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions per SIMD: waves per SIMD x ITERS x 64
    double winst = (double)blocks_per_cu * ITERS * 64;  // 4 waves per block over 4 SIMDs -> blocks_per_cu waves per SIMD
    double ns = ms * 1e6 / winst;
    printf("%-52s waves/SIMD %d  %8.3f ms  %6.3f ns per wave-instruction (%.2f cycles at 2.4 GHz)\n", name, blocks_per_cu, ms, ns, ns * 2.4);
}

int main() {
    uint32_t h[1024];
    for (int i = 0; i < 1024; ++i) h[i] = 0x9e3779b9u * (i + 1);
    uint32_t *d_in, *d_out; (void)hipMalloc(&d_in, sizeof(h)); (void)hipMalloc(&d_out, 4u * 256 * 8 * 256);
    (void)hipMemcpy(d_in, h, sizeof(h), hipMemcpyHostToDevice);
    for (int w : {8, 2, 1}) {
        run<0>("v_xor_b32 (VOP2)", d_out, d_in, w);
        run<8>("v_xor_b32_e64 (VOP3 encoding)", d_out, d_in, w);
        run<12>("v_mov_b32", d_out, d_in, w);
        run<1>("v_mad_u64_u32 acc in place, vcc", d_out, d_in, w);
        run<2>("v_mad_u64_u32 addend = other pair", d_out, d_in, w);
        run<3>("v_mad_u64_u32 addend = 0", d_out, d_in, w);
        run<4>("v_mad_u64_u32 carry-out to s[20:27]", d_out, d_in, w);
        run<15>("v_mad_u64_u32 + v_xor interleaved (per pair /2)", d_out, d_in, w);
        run<5>("v_mul_lo_u32", d_out, d_in, w);
        run<6>("v_mul_hi_u32", d_out, d_in, w);
        run<14>("v_mul_u32_u24 (VOP2)", d_out, d_in, w);
        run<16>("v_mad_u32_u24", d_out, d_in, w);
        run<7>("v_add_co/v_addc_co chain (vcc)", d_out, d_in, w);
        run<9>("v_alignbit_b32", d_out, d_in, w);
        run<10>("v_bfi_b32", d_out, d_in, w);
        run<11>("v_add3_u32", d_out, d_in, w);
        run<13>("v_lshl_add_u64", d_out, d_in, w);
    }
    return 0;
}
