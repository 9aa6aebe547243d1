// Keccak-256 Merkle kernels for gfx950. See merkle.h.
//
// One lane hashes one leaf: the 1600-bit state lives in 50 VGPRs (all indices compile-time); the row is streamed
// column by column (each column read is 32 B/lane, coalesced across the wave) and staged through a 17-lane
// per-thread LDS block buffer so that arbitrary row widths need no run-time indexing of registers.
#include "merkle.h"
#include "keccak.h"

namespace sp {

constexpr int MK_THREADS = 256;

__device__ __forceinline__ fe mk_ld_fe(const fe* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 lo = q[0], hi = q[1];
    fe r;
    r.v[0] = lo.x; r.v[1] = lo.y; r.v[2] = lo.z; r.v[3] = lo.w;
    r.v[4] = hi.x; r.v[5] = hi.y; r.v[6] = hi.z; r.v[7] = hi.w;
    return r;
}

__device__ __forceinline__ void absorb_block(uint64_t s[25], const uint64_t* blk) {
#pragma unroll
    for (int i = 0; i < 17; ++i) s[i] ^= blk[i];
    sp_keccak_f1600_dev(s);
}

__global__ void __launch_bounds__(MK_THREADS) leaf_hash_kernel(const fe* cols, uint64_t col_stride, uint32_t ncols,
                                                               uint64_t n_leaves, digest32* leaves_out) {
    __shared__ uint64_t blkbuf[MK_THREADS * 17];
    uint64_t i = (uint64_t)blockIdx.x * MK_THREADS + threadIdx.x;
    if (i >= n_leaves) return;
    uint64_t* blk = blkbuf + threadIdx.x * 17;
    uint64_t s[25];
#pragma unroll
    for (int k = 0; k < 25; ++k) s[k] = 0;
    uint32_t pos = 0;
    for (uint32_t j = 0; j < ncols; ++j) {
        fe raw = fe_from_mont(mk_ld_fe(cols + (uint64_t)j * col_stride + i));
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            // big-endian bytes of the element, 8 at a time, as a little-endian Keccak lane
            uint64_t limb = (uint64_t)raw.v[2 * (3 - l)] | ((uint64_t)raw.v[2 * (3 - l) + 1] << 32);
            blk[pos] = sp_bswap64(limb);
            if (++pos == 17) { absorb_block(s, blk); pos = 0; }
        }
    }
    // original Keccak padding 0x01 .. 0x80 over the 136-byte rate
    blk[pos] = 0x01ULL;
    for (uint32_t k = pos + 1; k < 17; ++k) blk[k] = 0;
    blk[16] ^= 0x8000000000000000ULL;
    absorb_block(s, blk);
    digest32 d;
    d.w[0] = s[0]; d.w[1] = s[1]; d.w[2] = s[2]; d.w[3] = s[3];
    leaves_out[i] = d;
}

// nodes[first + i] = Keccak256(nodes[2(first+i)+1] || nodes[2(first+i)+2]) for i < count
__global__ void __launch_bounds__(MK_THREADS) node_hash_kernel(digest32* nodes, uint64_t first, uint64_t count) {
    uint64_t i = (uint64_t)blockIdx.x * MK_THREADS + threadIdx.x;
    if (i >= count) return;
    uint64_t p = first + i;
    digest32 l = nodes[2 * p + 1], r = nodes[2 * p + 2];
    uint64_t s[25];
#pragma unroll
    for (int k = 0; k < 25; ++k) s[k] = 0;
    s[0] = l.w[0]; s[1] = l.w[1]; s[2] = l.w[2]; s[3] = l.w[3];
    s[4] = r.w[0]; s[5] = r.w[1]; s[6] = r.w[2]; s[7] = r.w[3];
    s[8] = 0x01ULL;
    s[16] = 0x8000000000000000ULL;
    sp_keccak_f1600_dev(s);
    digest32 d;
    d.w[0] = s[0]; d.w[1] = s[1]; d.w[2] = s[2]; d.w[3] = s[3];
    nodes[p] = d;
}

__global__ void gather_paths_kernel(const digest32* nodes, uint64_t n_leaves, uint32_t depth, const uint64_t* positions, uint32_t q, digest32* out) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= q * depth) return;
    uint32_t qi = t / depth, lvl = t % depth;
    uint64_t p = positions[qi] + n_leaves - 1;
    for (uint32_t k = 0; k < lvl; ++k) p = (p - 1) >> 1;
    uint64_t sib = (p & 1) ? p + 1 : p - 1;
    out[(uint64_t)qi * depth + lvl] = nodes[sib];
}

int merkle_hash_leaves(hipStream_t st, const fe* cols, uint64_t col_stride, uint32_t ncols, uint64_t n_leaves, digest32* nodes) {
    if (n_leaves == 0 || (n_leaves & (n_leaves - 1)) || ncols == 0) { sp_set_error("merkle: leaf count must be a power of two"); return SP_E_INVALID_ARG; }
    unsigned blocks = (unsigned)((n_leaves + MK_THREADS - 1) / MK_THREADS);
    hipLaunchKernelGGL(leaf_hash_kernel, dim3(blocks), dim3(MK_THREADS), 0, st, cols, col_stride, ncols, n_leaves, nodes + (n_leaves - 1));
    SP_HIP_CHECK(hipGetLastError());
    return SP_OK;
}

int merkle_hash_leaves_flat(hipStream_t st, const fe* cols, uint64_t col_stride, uint32_t ncols, uint64_t n_leaves, digest32* leaves_out) {
    if (n_leaves == 0 || ncols == 0) return SP_E_INVALID_ARG;
    unsigned blocks = (unsigned)((n_leaves + MK_THREADS - 1) / MK_THREADS);
    hipLaunchKernelGGL(leaf_hash_kernel, dim3(blocks), dim3(MK_THREADS), 0, st, cols, col_stride, ncols, n_leaves, leaves_out);
    SP_HIP_CHECK(hipGetLastError());
    return SP_OK;
}

int merkle_reduce(hipStream_t st, digest32* nodes, uint64_t n_leaves) {
    for (uint64_t count = n_leaves >> 1; count >= 1; count >>= 1) {
        uint64_t first = count - 1;
        unsigned blocks = (unsigned)((count + MK_THREADS - 1) / MK_THREADS);
        hipLaunchKernelGGL(node_hash_kernel, dim3(blocks), dim3(MK_THREADS), 0, st, nodes, first, count);
        SP_HIP_CHECK(hipGetLastError());
    }
    return SP_OK;
}

int merkle_gather_paths(hipStream_t st, const digest32* nodes, uint64_t n_leaves, const uint64_t* positions_dev, uint32_t q, digest32* out) {
    int depth = sp_log2_exact(n_leaves);
    if (depth < 0) return SP_E_INVALID_ARG;
    if (depth == 0 || q == 0) return SP_OK;
    unsigned total = q * (unsigned)depth;
    hipLaunchKernelGGL(gather_paths_kernel, dim3((total + 127) / 128), dim3(128), 0, st, nodes, n_leaves, (uint32_t)depth, positions_dev, q, out);
    SP_HIP_CHECK(hipGetLastError());
    return SP_OK;
}

}  // namespace sp
