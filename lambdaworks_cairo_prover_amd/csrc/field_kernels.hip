// Element-wise field kernels: batch inversion, layout transposes.
#include "field_kernels.h"
#include <cstdlib>

namespace sp {

__device__ __forceinline__ fe fk_ld(const fe* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 lo = q[0], hi = q[1];
    fe r;
    r.v[0] = lo.x; r.v[1] = lo.y; r.v[2] = lo.z; r.v[3] = lo.w;
    r.v[4] = hi.x; r.v[5] = hi.y; r.v[6] = hi.z; r.v[7] = hi.w;
    return r;
}
__device__ __forceinline__ void fk_st(fe* p, const fe& a) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(a.v[0], a.v[1], a.v[2], a.v[3]);
    q[1] = make_uint4(a.v[4], a.v[5], a.v[6], a.v[7]);
}

// FieldElement::inplace_batch_inverse (reference src/starks/constraints/evaluator.rs:69,171; lambdaworks-math):
// Montgomery's trick per thread over the strided chunk {t, t+T, t+2T, ...} (T = total threads), so that every
// global access of a wave is contiguous. scratch holds the running prefix products (n elements).
__global__ void __launch_bounds__(256) batch_inverse_kernel(fe* data, fe* scratch, uint64_t n, int* zero_flag) {
    const uint64_t T = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    fe acc = fe_one();
    for (uint64_t i = t; i < n; i += T) {
        fk_st(scratch + i, acc);
        acc = fe_mul(acc, fk_ld(data + i));
    }
    if (fe_is_zero(acc)) { atomicExch(zero_flag, 1); return; }
    fe inv = fe_inv(acc);
    uint64_t cnt = (n - t + T - 1) / T;
    for (uint64_t m = cnt; m-- > 0;) {
        uint64_t i = t + m * T;
        fe a = fk_ld(data + i);
        fk_st(data + i, fe_mul(inv, fk_ld(scratch + i)));
        inv = fe_mul(inv, a);
    }
}

// Two-level form for large arrays.  One thread per chunk of BI_CHUNK elements {t, t + T, ...} (T = chunks) keeps the running
// prefix products of its chunk and hands its chunk product to a second batch inversion over the T products; the chunk
// inverse then unwinds the chunk.  Compared with one level of 64-element chunks: four times as many threads in flight (the
// kernel is a chain of dependent loads and products per thread) and 1/16 of the Fermat inversions (250 squarings each).
// The prefixes of positions 0 and 1 of a chunk are 1 and the element itself, so scratch[0 .. T) holds the chunk products and
// scratch[T .. 2T) is the scratch of the second level: no memory beyond the caller's n elements.
constexpr uint32_t BI_CHUNK = 16;
constexpr uint32_t BI_TOP = 8;
__global__ void __launch_bounds__(256) batch_inverse_prefix_kernel(const fe* data, fe* scratch, uint64_t T) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    fe acc = fk_ld(data + t);
#pragma unroll 1
    for (uint32_t k = 1; k < BI_CHUNK; ++k) {
        if (k >= 2) fk_st(scratch + t + k * T, acc);
        acc = fe_mul(acc, fk_ld(data + t + k * T));
    }
    fk_st(scratch + t, acc);   // chunk product
}
__global__ void __launch_bounds__(256) batch_inverse_unwind_kernel(fe* data, const fe* scratch, uint64_t T) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    fe inv = fk_ld(scratch + t);   // inverse of the chunk product
#pragma unroll 1
    for (uint32_t k = BI_CHUNK; k-- > 2;) {
        const uint64_t i = t + k * T;
        fe a = fk_ld(data + i);
        fk_st(data + i, fe_mul(inv, fk_ld(scratch + i)));
        inv = fe_mul(inv, a);
    }
    fe a1 = fk_ld(data + t + T), a0 = fk_ld(data + t);
    fk_st(data + t + T, fe_mul(inv, a0));
    fk_st(data + t, fe_mul(inv, a1));
}

int batch_inverse(hipStream_t st, fe* data, fe* scratch, uint64_t n, int* zero_flag_dev) {
    if (n == 0) return SP_OK;
    if (n >= (1ull << 16) && n % BI_CHUNK == 0) {
        const uint64_t T = n / BI_CHUNK;
        const unsigned blocks = (unsigned)((T + 255) / 256);
        hipLaunchKernelGGL(batch_inverse_prefix_kernel, dim3(blocks), dim3(256), 0, st, data, scratch, T);
        SP_HIP_CHECK(hipGetLastError());
        // a zero element makes its chunk product zero: the second level raises the flag.
        // Second level: BI_TOP elements per thread.  With the Fermat chain (261 dependent products) 64 elements per inversion were the
        // balance; the division-step inversion of fp.h costs about 45 products' worth of instructions, so a thread now takes 8 - a chain
        // of 24 products + one inversion instead of 192 + one (config #4: 283 -> 214 us with the new inversion alone, -> ~70 with this)
        uint64_t threads2 = (T + BI_TOP - 1) / BI_TOP;
        unsigned blocks2 = (unsigned)((threads2 + 255) / 256);
        hipLaunchKernelGGL(batch_inverse_kernel, dim3(blocks2), dim3(256), 0, st, scratch, scratch + T, T, zero_flag_dev);
        SP_HIP_CHECK(hipGetLastError());
        hipLaunchKernelGGL(batch_inverse_unwind_kernel, dim3(blocks), dim3(256), 0, st, data, scratch, T);
        SP_HIP_CHECK(hipGetLastError());
        return SP_OK;
    }
    // a few elements per thread: the inversion costs about 45 products' worth of instructions (fp.h), the chain per thread stays short
    uint64_t threads = (n + BI_TOP - 1) / BI_TOP;
    unsigned blocks = (unsigned)((threads + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(batch_inverse_kernel, dim3(blocks), dim3(256), 0, st, data, scratch, n, zero_flag_dev);
    SP_HIP_CHECK(hipGetLastError());
    return SP_OK;
}

// Row-major table (rows x cols, ABI encoding) -> column-major device layout. One thread per element; reads are
// contiguous along the row, writes along the column (32-byte granules), tiled through LDS to coalesce both.
template <int ENC>
__global__ void __launch_bounds__(256) rows_to_columns_kernel(const uint8_t* rows, uint64_t n_rows, uint32_t n_cols, fe* cols, uint64_t col_stride) {
    // tile: 32 rows x 8 cols
    __shared__ fe tile[32][9];
    uint32_t tx = threadIdx.x & 7, ty = threadIdx.x >> 3;
    uint64_t row0 = (uint64_t)blockIdx.x * 32;
    uint32_t col0 = blockIdx.y * 8;
    {
        uint64_t rr = row0 + ty; uint32_t cc = col0 + tx;
        if (rr < n_rows && cc < n_cols) {
            const uint8_t* p = rows + (rr * n_cols + cc) * 32;
            fe x;
            if (ENC == SP_FE_MONT_LIMBS) {
                const uint64_t* l = reinterpret_cast<const uint64_t*>(p);
                uint64_t w[4] = {l[0], l[1], l[2], l[3]};
                x = fe_canonical_lazy(fe_from_lw_limbs(w));   // lambdaworks limbs are < p; anything else is taken mod p
            } else {
                const uint32_t* w = reinterpret_cast<const uint32_t*>(p);
                fe raw;
#pragma unroll
                for (int k = 0; k < 8; ++k) raw.v[k] = sp_bswap32(w[7 - k]);
                x = fe_to_mont(raw);
            }
            tile[ty][tx] = x;
        }
    }
    __syncthreads();
    {
        uint32_t ry = threadIdx.x & 31, cx = threadIdx.x >> 5;
        uint64_t rr = row0 + ry; uint32_t cc = col0 + cx;
        if (rr < n_rows && cc < n_cols) fk_st(cols + (uint64_t)cc * col_stride + rr, tile[ry][cx]);
    }
}

int rows_to_columns(hipStream_t st, int enc, const uint8_t* rows_dev, uint64_t n_rows, uint32_t n_cols, fe* cols, uint64_t col_stride) {
    dim3 grid((unsigned)((n_rows + 31) / 32), (n_cols + 7) / 8);
    if (enc == SP_FE_MONT_LIMBS)
        hipLaunchKernelGGL((rows_to_columns_kernel<SP_FE_MONT_LIMBS>), grid, dim3(256), 0, st, rows_dev, n_rows, n_cols, cols, col_stride);
    else if (enc == SP_FE_CANON_BE)
        hipLaunchKernelGGL((rows_to_columns_kernel<SP_FE_CANON_BE>), grid, dim3(256), 0, st, rows_dev, n_rows, n_cols, cols, col_stride);
    else return SP_E_INVALID_ARG;
    SP_HIP_CHECK(hipGetLastError());
    return SP_OK;
}

// sp_fe_mul: the product and the square the kernels are built from, on operands the caller chooses
__global__ void __launch_bounds__(256) mul_elements_kernel(const fe* a, const fe* b, uint64_t n, fe* out) {
    uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const fe x = fk_ld(a + i);
    fk_st(out + i, b ? fe_mul(x, fk_ld(b + i)) : fe_sqr(x));
}
int mul_elements(hipStream_t st, const fe* a, const fe* b, uint64_t n, fe* out) {
    if (n == 0) return SP_OK;
    hipLaunchKernelGGL(mul_elements_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, b, n, out);
    SP_HIP_CHECK(hipGetLastError());
    return SP_OK;
}

// column-major device layout -> column-major ABI encoding (one thread per element)
template <int ENC>
__global__ void __launch_bounds__(256) encode_kernel(const fe* in, uint64_t n, uint8_t* out) {
    uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    fe x = fk_ld(in + i);
    if (ENC == SP_FE_MONT_LIMBS) {
        uint64_t l[4];
        fe_to_lw_limbs(x, l);
        uint64_t* o = reinterpret_cast<uint64_t*>(out + 32 * i);
        o[0] = l[0]; o[1] = l[1]; o[2] = l[2]; o[3] = l[3];
    } else {
        fe raw = fe_from_mont(x);
        uint32_t* o = reinterpret_cast<uint32_t*>(out + 32 * i);
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = sp_bswap32(raw.v[7 - k]);
    }
}
template <int ENC>
__global__ void __launch_bounds__(256) decode_kernel(const uint8_t* in, uint64_t n, fe* out) {
    uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    fe x;
    if (ENC == SP_FE_MONT_LIMBS) {
        const uint64_t* l = reinterpret_cast<const uint64_t*>(in + 32 * i);
        uint64_t w[4] = {l[0], l[1], l[2], l[3]};
        x = fe_canonical_lazy(fe_from_lw_limbs(w));   // lambdaworks limbs are < p; anything else is taken mod p
    } else {
        const uint32_t* p = reinterpret_cast<const uint32_t*>(in + 32 * i);
        fe raw;
#pragma unroll
        for (int k = 0; k < 8; ++k) raw.v[k] = sp_bswap32(p[7 - k]);
        x = fe_to_mont(raw);
    }
    fk_st(out + i, x);
}

// Host-to-device copy by a kernel that reads page-locked host memory over PCIe itself (52 - 54 GB/s against 56 for the DMA engine,
// tools/experiments/pinned_dma_probe.hip).  The prover's column uploads use it because hipMemcpyAsync from a page-locked buffer
// was sometimes routed through the runtime's staging path - 28 - 37 GB/s and the enqueueing host thread blocked for milliseconds -
// after earlier page-locked buffers of the process had been freed (profiles/r03_pinned_upload.txt).
__global__ void __launch_bounds__(256) pull_copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const uint4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}
int pull_copy(hipStream_t st, const void* src_pinned_host, void* dst_dev, size_t bytes) {
    if (bytes % 16 || (reinterpret_cast<uintptr_t>(src_pinned_host) | reinterpret_cast<uintptr_t>(dst_dev)) % 16) return SP_E_INVALID_ARG;
    static const unsigned blocks = [] { const char* e = std::getenv("SP_UPLOAD_PULL"); int v = e ? std::atoi(e) : 0; return (unsigned)(v > 1 ? v : 64); }();
    hipLaunchKernelGGL(pull_copy_kernel, dim3(blocks), dim3(256), 0, st, static_cast<const uint4*>(src_pinned_host), static_cast<uint4*>(dst_dev), bytes / 16);
    SP_HIP_CHECK(hipGetLastError());
    return SP_OK;
}

int encode_elements(hipStream_t st, int enc, const fe* in, uint64_t n, uint8_t* out_dev) {
    unsigned blocks = (unsigned)((n + 255) / 256);
    if (enc == SP_FE_MONT_LIMBS) hipLaunchKernelGGL((encode_kernel<SP_FE_MONT_LIMBS>), dim3(blocks), dim3(256), 0, st, in, n, out_dev);
    else if (enc == SP_FE_CANON_BE) hipLaunchKernelGGL((encode_kernel<SP_FE_CANON_BE>), dim3(blocks), dim3(256), 0, st, in, n, out_dev);
    else return SP_E_INVALID_ARG;
    SP_HIP_CHECK(hipGetLastError());
    return SP_OK;
}
int decode_elements(hipStream_t st, int enc, const uint8_t* in_dev, uint64_t n, fe* out) {
    unsigned blocks = (unsigned)((n + 255) / 256);
    if (enc == SP_FE_MONT_LIMBS) hipLaunchKernelGGL((decode_kernel<SP_FE_MONT_LIMBS>), dim3(blocks), dim3(256), 0, st, in_dev, n, out);
    else if (enc == SP_FE_CANON_BE) hipLaunchKernelGGL((decode_kernel<SP_FE_CANON_BE>), dim3(blocks), dim3(256), 0, st, in_dev, n, out);
    else return SP_E_INVALID_ARG;
    SP_HIP_CHECK(hipGetLastError());
    return SP_OK;
}

// out[c][i] = bit i of column c's bitmap ? 1 : 0 (Montgomery form): the sixteen flag columns of a Cairo main trace cross PCIe as
// one bit per cell (prover_upload.cpp) and become field elements here
__global__ void __launch_bounds__(256) expand_bit_columns_kernel(const uint64_t* __restrict__ bits, uint64_t n, uint64_t total, fe* out) {
    const uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const uint64_t c = e / n, i = e - c * n;
    const uint64_t w = bits[c * (n >> 6) + (i >> 6)];
    fk_st(out + e, ((w >> (i & 63)) & 1ULL) ? fe_one() : fe_zero());
}
int expand_bit_columns(hipStream_t st, const uint64_t* bits_dev, uint64_t n, uint32_t cols, fe* out) {
    if (!bits_dev || !out || n < 64 || (n & 63)) return SP_E_INVALID_ARG;
    const uint64_t total = n * cols;
    if (!total) return SP_OK;
    hipLaunchKernelGGL(expand_bit_columns_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, bits_dev, n, total, out);
    SP_HIP_CHECK(hipGetLastError());
    return SP_OK;
}

}  // namespace sp
