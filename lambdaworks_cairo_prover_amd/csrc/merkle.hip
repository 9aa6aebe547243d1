// Keccak-256 Merkle kernels for gfx950. See merkle.h.
//
// One lane hashes one leaf: the 1600-bit state lives in 50 VGPRs (all indices compile-time); the row is streamed
// column by column (each column read is 32 B/lane, coalesced across the wave) and staged through a 17-lane
// per-thread LDS block buffer so that arbitrary row widths need no run-time indexing of registers.
#include "merkle.h"
#include "keccak.h"
#include "poseidon.h"
#include <cstdlib>

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
                                                               uint64_t n_leaves, digest32* leaves_out, LdeOrder order) {
    __shared__ uint64_t blkbuf[MK_THREADS * 17];
    uint64_t i = (uint64_t)blockIdx.x * MK_THREADS + threadIdx.x;
    if (i >= n_leaves) return;
    uint64_t* blk = blkbuf + threadIdx.x * 17;
    uint64_t s[25];
#pragma unroll
    for (int k = 0; k < 25; ++k) s[k] = 0;
    uint32_t pos = 0;
    const uint64_t at = order.at(i);   // coset-major columns: 8 lanes x 8 consecutive rows = 256-byte runs per coset
    for (uint32_t j = 0; j < ncols; ++j) {
        fe raw = fe_from_mont(mk_ld_fe(cols + (uint64_t)j * col_stride + at));
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

// The same for the row widths of the Cairo prover (34 / 43 main, 18 auxiliary, 2 composition, 1 FRI): with the width known at
// compile time the loop over the columns unrolls, every sponge position is a constant, and the row is absorbed straight from
// registers - no block buffer in LDS, no position bookkeeping.  Worth 1.5 - 2 % (2^22 leaves x 34: 4.675 -> 4.58 ms, x 18:
// 2.96 -> 2.90 ms, tools/merkle_width_bench.py): the kernel is bound by the rounds of the permutation, the bookkeeping it saves
// was mostly hidden behind them.
// (compile-time recursion instead of a loop: the optimiser does not unroll 34 iterations with a permutation inside, and a
// run-time sponge position would send the state to scratch memory)
template <uint32_t J, uint32_t NCOLS>
__device__ __forceinline__ void absorb_columns(uint64_t (&s)[25], const fe* p, uint64_t col_stride) {
    if constexpr (J < NCOLS) {
        const fe raw = fe_from_mont(mk_ld_fe(p + (uint64_t)J * col_stride));
#pragma unroll
        for (uint32_t l = 0; l < 4; ++l) {
            const uint32_t pos = (4u * J + l) % 17u;
            const uint64_t limb = (uint64_t)raw.v[2 * (3 - l)] | ((uint64_t)raw.v[2 * (3 - l) + 1] << 32);
            s[pos] ^= sp_bswap64(limb);
            if (pos == 16u) sp_keccak_f1600_dev(s);
        }
        absorb_columns<J + 1, NCOLS>(s, p, col_stride);
    }
}
template <uint32_t NCOLS>
__global__ void __launch_bounds__(MK_THREADS) leaf_hash_fixed_kernel(const fe* cols, uint64_t col_stride, uint64_t n_leaves, digest32* leaves_out, LdeOrder order) {
    uint64_t i = (uint64_t)blockIdx.x * MK_THREADS + threadIdx.x;
    if (i >= n_leaves) return;
    uint64_t s[25];
#pragma unroll
    for (int k = 0; k < 25; ++k) s[k] = 0;
    absorb_columns<0, NCOLS>(s, cols + order.at(i), col_stride);
    constexpr uint32_t tail = (4u * NCOLS) % 17u;         // original Keccak padding 0x01 .. 0x80 over the 136-byte rate
    s[tail] ^= 0x01ULL;
    s[16] ^= 0x8000000000000000ULL;
    sp_keccak_f1600_dev(s);
    digest32 d;
    d.w[0] = s[0]; d.w[1] = s[1]; d.w[2] = s[2]; d.w[3] = s[3];
    leaves_out[i] = d;
}

// the two-launch form (merkle.h): 17 columns are exactly four blocks, so the tail starts at sponge position 0 like a fresh row
__global__ void __launch_bounds__(MK_THREADS) leaf_hash_head_kernel(const fe* cols, uint64_t col_stride, uint64_t n_leaves, uint64_t* state, LdeOrder order) {
    uint64_t i = (uint64_t)blockIdx.x * MK_THREADS + threadIdx.x;
    if (i >= n_leaves) return;
    uint64_t s[25];
#pragma unroll
    for (int k = 0; k < 25; ++k) s[k] = 0;
    absorb_columns<0, MK_HEAD_COLS>(s, cols + order.at(i), col_stride);
#pragma unroll
    for (int k = 0; k < 25; ++k) state[(uint64_t)k * n_leaves + i] = s[k];
}
template <uint32_t NREST>
__global__ void __launch_bounds__(MK_THREADS) leaf_hash_tail_kernel(const fe* cols_rest, uint64_t col_stride, uint64_t n_leaves, const uint64_t* state,
                                                                    digest32* leaves_out, LdeOrder order) {
    uint64_t i = (uint64_t)blockIdx.x * MK_THREADS + threadIdx.x;
    if (i >= n_leaves) return;
    uint64_t s[25];
#pragma unroll
    for (int k = 0; k < 25; ++k) s[k] = state[(uint64_t)k * n_leaves + i];
    absorb_columns<0, NREST>(s, cols_rest + order.at(i), col_stride);
    constexpr uint32_t tail = (4u * NREST) % 17u;
    s[tail] ^= 0x01ULL;
    s[16] ^= 0x8000000000000000ULL;
    sp_keccak_f1600_dev(s);
    digest32 d;
    d.w[0] = s[0]; d.w[1] = s[1]; d.w[2] = s[2]; d.w[3] = s[3];
    leaves_out[i] = d;
}

static bool launch_leaf_hash(hipStream_t st, const fe* cols, uint64_t col_stride, uint32_t ncols, uint64_t n_leaves, digest32* out, LdeOrder order) {
    const dim3 grid((unsigned)((n_leaves + MK_THREADS - 1) / MK_THREADS)), block(MK_THREADS);
    switch (ncols) {
        case 1: hipLaunchKernelGGL(leaf_hash_fixed_kernel<1>, grid, block, 0, st, cols, col_stride, n_leaves, out, order); return true;
        case 2: hipLaunchKernelGGL(leaf_hash_fixed_kernel<2>, grid, block, 0, st, cols, col_stride, n_leaves, out, order); return true;
        case 18: hipLaunchKernelGGL(leaf_hash_fixed_kernel<18>, grid, block, 0, st, cols, col_stride, n_leaves, out, order); return true;
        case 34: hipLaunchKernelGGL(leaf_hash_fixed_kernel<34>, grid, block, 0, st, cols, col_stride, n_leaves, out, order); return true;
        case 43: hipLaunchKernelGGL(leaf_hash_fixed_kernel<43>, grid, block, 0, st, cols, col_stride, n_leaves, out, order); return true;
        default: return false;
    }
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

// Two levels per launch for the large levels: thread t owns the subtree under node g = first + t of the UPPER level - it hashes the
// four consecutive digests below its two children (128 contiguous bytes), writes the two children (64 contiguous bytes) and then
// their parent.  The permutations are the same 3 per 4 digests; what goes away is every second launch boundary of the
// throughput-bound part of a tree (the drain of one level and the fill of the next, ~10 - 20 us each) and the round trip of the
// lower level's digests through HBM.
__global__ void __launch_bounds__(MK_THREADS) node_hash2_kernel(digest32* nodes, uint64_t first, uint64_t count) {
    uint64_t i = (uint64_t)blockIdx.x * MK_THREADS + threadIdx.x;
    if (i >= count) return;
    const uint64_t g = first + i, p0 = 2 * g + 1;           // children p0, p0 + 1; their children 2 p0 + 1 .. 2 p0 + 4
    const digest32* c = nodes + 2 * p0 + 1;
    uint64_t h[8];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const digest32 l = c[2 * half], r = c[2 * half + 1];
        uint64_t s[25];
#pragma unroll
        for (int k = 0; k < 25; ++k) s[k] = 0;
        s[0] = l.w[0]; s[1] = l.w[1]; s[2] = l.w[2]; s[3] = l.w[3];
        s[4] = r.w[0]; s[5] = r.w[1]; s[6] = r.w[2]; s[7] = r.w[3];
        s[8] = 0x01ULL;
        s[16] = 0x8000000000000000ULL;
        sp_keccak_f1600_dev(s);
        h[4 * half] = s[0]; h[4 * half + 1] = s[1]; h[4 * half + 2] = s[2]; h[4 * half + 3] = s[3];
        digest32 d;
        d.w[0] = s[0]; d.w[1] = s[1]; d.w[2] = s[2]; d.w[3] = s[3];
        nodes[p0 + half] = d;
    }
    uint64_t s[25];
#pragma unroll
    for (int k = 0; k < 25; ++k) s[k] = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s[k] = h[k];
    s[8] = 0x01ULL;
    s[16] = 0x8000000000000000ULL;
    sp_keccak_f1600_dev(s);
    digest32 d;
    d.w[0] = s[0]; d.w[1] = s[1]; d.w[2] = s[2]; d.w[3] = s[3];
    nodes[g] = d;
}

// ---- lane-parallel permutation for the small tree levels -------------------------------------------------------------
// A level with few nodes is latency-bound in node_hash_kernel: one lane needs 24 x 180 dependent-issue instructions
// (~11 us) however few nodes there are, and a proof has ~300 such levels.  Here 25 lanes share one permutation, lane
// (x, y) = (l % 5, l / 5) holding state word A[x][y]; a round is two exchanges through a 200-byte LDS block:
//   write A; read the columns x-1 and x+1 (ten words) -> D[x];  t = rotl(A ^ D[x], rho[x][y]);  write t at pi(x, y);
//   read B[x][y], B[x+1][y], B[x+2][y] -> chi; iota on lane 0.
// Two permutations per wave (lanes 0-24 and 32-56).  ~2.5x lower latency, ~8x more lane-instructions per permutation:
// used only below MK_LANES_MAX_NODES nodes, several levels per launch (node_hash_lanes_kernel).
#ifndef SP_MK_LANES_MAX_NODES
#define SP_MK_LANES_MAX_NODES 4096
#endif
constexpr uint32_t MK_LANES_MAX_NODES = SP_MK_LANES_MAX_NODES;
#ifndef SP_MK_PAIR_MIN_UPPER
#define SP_MK_PAIR_MIN_UPPER (1u << 17)
#endif
constexpr uint64_t MK_PAIR_MIN_UPPER = SP_MK_PAIR_MIN_UPPER;
__device__ __constant__ const uint8_t SP_KECCAK_RHO[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};

// rotation by a per-lane amount with 32-bit funnel shifts (two 64-bit variable shifts and an or are three slow VALU ops):
// swap = n >= 32 exchanges the halves, sh = (32 - n % 32) % 32 is the v_alignbit amount, keep = n % 32 == 0 leaves them as they are
struct LaneRot { uint32_t sh; bool swap, keep; };
__device__ __forceinline__ uint64_t rotl64_lane(uint64_t v, const LaneRot r) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    const uint32_t a = r.swap ? hi : lo, b = r.swap ? lo : hi;                   // {b, a} = v rotated by 0 or 32
    const uint32_t rh = __builtin_amdgcn_alignbit(b, a, r.sh), rl = __builtin_amdgcn_alignbit(a, b, r.sh);
    lo = r.keep ? a : rl; hi = r.keep ? b : rh;
    return ((uint64_t)hi << 32) | lo;
}

// one permutation slot = 32 consecutive lanes (25 active); buf = 64 words of LDS owned by the slot (A[32], B[32]: the seven
// spare lanes run the same instructions on entries nobody reads, so a round is straight-line code); rc = the 24 round
// constants in LDS (a scalar load per round would sit in the dependent chain; every lane reads the constant together with
// its other operands and lane 0 alone uses it).
// A slot is half a wave and the LDS serves the instructions of one wave in order: the reads of an exchange see the writes
// issued before them without a wait in between (no fence: it would drain the queue, one more LDS round trip per exchange).
__device__ __forceinline__ uint64_t keccak_f_lanes(uint64_t a, uint32_t l, bool active, uint64_t* buf, const uint64_t* rc) {
    const uint32_t x = l % 5u, y = active ? l / 5u : 0u;
    uint64_t* A = buf;
    uint64_t* B = buf + 32;
    const uint32_t xm = (x + 4u) % 5u, xp = (x + 1u) % 5u, xpp = (x + 2u) % 5u;
    const uint32_t rho = active ? SP_KECCAK_RHO[l] : 0u;
    const LaneRot rot{(32u - (rho & 31u)) & 31u, rho >= 32u, (rho & 31u) == 0u};
    const uint32_t dst = active ? y + 5u * ((2u * x + 3u * y) % 5u) : l;   // pi: B[y][2x + 3y] = rotl(A[x][y], rho[x][y])
    const uint64_t iota_mask = l == 0u ? ~0ULL : 0ULL;
    const uint64_t* Am = A + xm;
    const uint64_t* Ap = A + xp;
    const uint64_t* B1 = B + xp + 5u * y;
    const uint64_t* B2 = B + xpp + 5u * y;
#pragma unroll 1
    for (int r = 0; r < 24; ++r) {
        A[l] = a;
        __builtin_amdgcn_wave_barrier();
        const uint64_t rcv = rc[r] & iota_mask;
        uint64_t m[5], q[5];
#pragma unroll
        for (uint32_t k = 0; k < 5; ++k) { m[k] = Am[5u * k]; q[k] = Ap[5u * k]; }
        const uint64_t cm = m[0] ^ m[1] ^ m[2] ^ m[3] ^ m[4], cp = q[0] ^ q[1] ^ q[2] ^ q[3] ^ q[4];
        B[dst] = rotl64_lane(a ^ cm ^ sp_rotl64(cp, 1), rot);
        __builtin_amdgcn_wave_barrier();
        const uint64_t b0 = B[l], b1 = *B1, b2 = *B2;
        a = sp_chi(b0, b1, b2) ^ rcv;
        __builtin_amdgcn_wave_barrier();
    }
    return a;
}

// Block b reduces `levels` consecutive tree levels of its own nodes: nodes 8b .. 8b+7 of the level with `count` nodes, then
// 4b .. 4b+3 of the level above, and so on (8 = slots per block).  A small level is latency-bound - one dependent Keccak-f per
// level plus a launch boundary - so several levels per launch cost their Keccak latencies only; the slots that fall idle on
// the way up were not needed anyway.  The digests of a level reach the next one through LDS (the copy in the node array is
// written on the side: nobody in this launch reads it).  levels = 1: a plain level.  The last launch of a tree runs to the root.
__global__ void __launch_bounds__(256) node_hash_lanes_kernel(digest32* nodes, uint32_t count, uint32_t levels, FriChallenge ch) {
    __shared__ uint64_t lds[8 * 64];
    __shared__ uint64_t hand[2][8 * 4];
    __shared__ uint64_t rc[24];
    const uint32_t slots = blockDim.x >> 5;   // 8
    const uint32_t slot = threadIdx.x >> 5, l = threadIdx.x & 31u;
    const bool lane_active = l < 25u;
    uint64_t* buf = lds + slot * 64;
    uint64_t* words = reinterpret_cast<uint64_t*>(nodes);
    if (threadIdx.x < 24) rc[threadIdx.x] = SP_KECCAK_RC_DEV[threadIdx.x];
    __syncthreads();
    uint32_t mine = slots;                    // slots of this block that still have a node on the current level
    for (uint32_t lev = 0;; ++lev) {
        const uint32_t i = blockIdx.x * mine + slot;
        if (slot < mine && i < count) {   // uniform per slot (32 lanes), slots never straddle a wave
            const uint64_t p = (uint64_t)(count - 1) + i;
            uint64_t a = 0;
            if (l < 8u) a = lev == 0 ? words[(2 * p + 1) * 4 + l]                       // left digest then right digest: 8 consecutive words
                                     : hand[(lev - 1) & 1u][(2 * slot) * 4 + l];         // = slots 2 slot, 2 slot + 1 of the level below
            else if (l == 8u) a = 0x01ULL;                       // original Keccak padding of a 64-byte message
            else if (l == 16u) a = 0x8000000000000000ULL;
            a = keccak_f_lanes(a, l, lane_active, buf, rc);
            if (l < 4u) { words[p * 4 + l] = a; hand[lev & 1u][slot * 4 + l] = a; }
        }
        if (lev + 1 >= levels || count == 1 || mine == 1) {
            // the launch that reaches the root also takes the transcript step of the FRI commit phase (merkle.h, FriChallenge)
            if (ch.state != nullptr && count == 1 && blockIdx.x == 0) {
                __syncthreads();                        // hand[lev & 1][0..3] = the root
                if (slot == 0) {
                    uint64_t a = 0;
                    if (l < 4u) a = ch.state[l];
                    else if (l < 8u) a = hand[lev & 1u][l - 4u];
                    else if (l == 8u) a = 0x01ULL;
                    else if (l == 16u) a = 0x8000000000000000ULL;
                    a = keccak_f_lanes(a, l, lane_active, buf, rc);
                    uint64_t* d = hand[(lev + 1u) & 1u];
                    if (l < 4u) { d[l] = a; ch.root_copy[l] = hand[lev & 1u][l]; }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    if (l < 4u) ch.state[l] = sp_bswap64(d[3u - l]);          // reverse(d) as little-endian words
                    if (l == 0u) {
                        fe z;
#pragma unroll
                        for (int k = 0; k < 4; ++k) { z.v[2 * k] = (uint32_t)d[k]; z.v[2 * k + 1] = (uint32_t)(d[k] >> 32); }
                        z.v[7] &= 0x07ffffffu;                                // 251 bits (transcript.rs:24-43): below p
                        const fe c = fe_mul(fe_to_mont(z), *ch.mul_in);
                        *ch.cst_out = c;
                    }
                }
            }
            break;
        }
        __syncthreads();
        count >>= 1;
        mine >>= 1;
    }
}

// ---- Poseidon backend (poseidon.h) -------------------------------------------------------------------------------------
// One lane per hash: a permutation is ~214 dependent-ish field products (68 k issue slots), so a leaf of 34 columns is 18 of them
// against the 9 Keccak-f (4.5 k slots each) of the reference's hash - the commitment is bound by the multiplier like the transforms.
constexpr int PS_THREADS = 128;
__global__ void __launch_bounds__(PS_THREADS) poseidon_leaf_kernel(const fe* cols, uint64_t col_stride, uint32_t ncols, uint64_t n_leaves,
                                                                   digest32* leaves_out, LdeOrder order, int single) {
    const uint64_t i = (uint64_t)blockIdx.x * PS_THREADS + threadIdx.x;
    if (i >= n_leaves) return;
    const fe* p = cols + order.at(i);
    const fe h = single ? poseidon_hash1(*p) : poseidon_hash_many(p, col_stride, ncols);
    digest32 d;
    poseidon_digest_from_fe(h, d.w);
    leaves_out[i] = d;
}
__global__ void __launch_bounds__(PS_THREADS) poseidon_node_kernel(digest32* nodes, uint64_t first, uint64_t count) {
    const uint64_t i = (uint64_t)blockIdx.x * PS_THREADS + threadIdx.x;
    if (i >= count) return;
    const uint64_t p = first + i;
    const digest32 l = nodes[2 * p + 1], r = nodes[2 * p + 2];
    const fe h = poseidon_hash2(poseidon_fe_from_digest(l.w), poseidon_fe_from_digest(r.w));
    digest32 d;
    poseidon_digest_from_fe(h, d.w);
    nodes[p] = d;
}
// The transcript step of FriChallenge for a tree whose root was not produced by the Keccak lanes kernel: one lane, one Keccak-f.
__global__ void fri_challenge_kernel(const digest32* root, FriChallenge ch) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const digest32 rt = *root;
    uint64_t s[25];
#pragma unroll
    for (int k = 0; k < 25; ++k) s[k] = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { s[k] = ch.state[k]; s[4 + k] = rt.w[k]; ch.root_copy[k] = rt.w[k]; }
    s[8] = 0x01ULL;
    s[16] = 0x8000000000000000ULL;
    sp_keccak_f1600_dev(s);
#pragma unroll
    for (int k = 0; k < 4; ++k) ch.state[k] = sp_bswap64(s[3 - k]);          // reverse(d) as little-endian words
    fe z;
#pragma unroll
    for (int k = 0; k < 4; ++k) { z.v[2 * k] = (uint32_t)s[k]; z.v[2 * k + 1] = (uint32_t)(s[k] >> 32); }
    z.v[7] &= 0x07ffffffu;                                                    // 251 bits (transcript.rs:24-43): below p
    *ch.cst_out = fe_mul(fe_to_mont(z), *ch.mul_in);
}

static int poseidon_leaves(hipStream_t st, const fe* cols, uint64_t col_stride, uint32_t ncols, uint64_t n_leaves, digest32* out, LdeOrder order, MerkleHash mh) {
    if (mh == MerkleHash::POSEIDON_SINGLE && ncols != 1) { sp_set_error("merkle: a single-element Poseidon tree has one column"); return SP_E_INVALID_ARG; }
    const unsigned blocks = (unsigned)((n_leaves + PS_THREADS - 1) / PS_THREADS);
    hipLaunchKernelGGL(poseidon_leaf_kernel, dim3(blocks), dim3(PS_THREADS), 0, st, cols, col_stride, ncols, n_leaves, out, order,
                       mh == MerkleHash::POSEIDON_SINGLE ? 1 : 0);
    SP_HIP_CHECK(hipGetLastError());
    return SP_OK;
}

int merkle_hash_leaves(hipStream_t st, const fe* cols, uint64_t col_stride, uint32_t ncols, uint64_t n_leaves, digest32* nodes, LdeOrder order, MerkleHash mh) {
    if (n_leaves == 0 || (n_leaves & (n_leaves - 1)) || ncols == 0) { sp_set_error("merkle: leaf count must be a power of two"); return SP_E_INVALID_ARG; }
    if (mh != MerkleHash::KECCAK256) return poseidon_leaves(st, cols, col_stride, ncols, n_leaves, nodes + (n_leaves - 1), order, mh);
    unsigned blocks = (unsigned)((n_leaves + MK_THREADS - 1) / MK_THREADS);
    if (!launch_leaf_hash(st, cols, col_stride, ncols, n_leaves, nodes + (n_leaves - 1), order))
        hipLaunchKernelGGL(leaf_hash_kernel, dim3(blocks), dim3(MK_THREADS), 0, st, cols, col_stride, ncols, n_leaves, nodes + (n_leaves - 1), order);
    SP_HIP_CHECK(hipGetLastError());
    return SP_OK;
}

int merkle_hash_leaves_head(hipStream_t st, const fe* cols, uint64_t col_stride, uint64_t n_leaves, uint64_t* state, LdeOrder order) {
    if (n_leaves == 0 || !state) return SP_E_INVALID_ARG;
    hipLaunchKernelGGL(leaf_hash_head_kernel, dim3((unsigned)((n_leaves + MK_THREADS - 1) / MK_THREADS)), dim3(MK_THREADS), 0, st, cols, col_stride, n_leaves, state, order);
    SP_HIP_CHECK(hipGetLastError());
    return SP_OK;
}
int merkle_hash_leaves_tail(hipStream_t st, const fe* cols, uint64_t col_stride, uint32_t ncols, uint64_t n_leaves, const uint64_t* state, digest32* nodes,
                            LdeOrder order) {
    if (n_leaves == 0 || (n_leaves & (n_leaves - 1)) || !state || !merkle_split_supported(ncols)) return SP_E_INVALID_ARG;
    const dim3 grid((unsigned)((n_leaves + MK_THREADS - 1) / MK_THREADS)), block(MK_THREADS);
    const fe* rest = cols + (uint64_t)MK_HEAD_COLS * col_stride;
    digest32* out = nodes + (n_leaves - 1);
    if (ncols == 34) hipLaunchKernelGGL(leaf_hash_tail_kernel<17>, grid, block, 0, st, rest, col_stride, n_leaves, state, out, order);
    else hipLaunchKernelGGL(leaf_hash_tail_kernel<26>, grid, block, 0, st, rest, col_stride, n_leaves, state, out, order);
    SP_HIP_CHECK(hipGetLastError());
    return SP_OK;
}

int merkle_hash_leaves_flat(hipStream_t st, const fe* cols, uint64_t col_stride, uint32_t ncols, uint64_t n_leaves, digest32* leaves_out, LdeOrder order, MerkleHash mh) {
    if (n_leaves == 0 || ncols == 0) return SP_E_INVALID_ARG;
    if (mh != MerkleHash::KECCAK256) return poseidon_leaves(st, cols, col_stride, ncols, n_leaves, leaves_out, order, mh);
    unsigned blocks = (unsigned)((n_leaves + MK_THREADS - 1) / MK_THREADS);
    if (!launch_leaf_hash(st, cols, col_stride, ncols, n_leaves, leaves_out, order))
        hipLaunchKernelGGL(leaf_hash_kernel, dim3(blocks), dim3(MK_THREADS), 0, st, cols, col_stride, ncols, n_leaves, leaves_out, order);
    SP_HIP_CHECK(hipGetLastError());
    return SP_OK;
}

int merkle_reduce(hipStream_t st, digest32* nodes, uint64_t n_leaves, const FriChallenge* ch, MerkleHash mh) {
    if (ch && n_leaves < 2) return SP_E_INVALID_ARG;
    uint64_t count = n_leaves >> 1;
    if (mh != MerkleHash::KECCAK256) {
        for (; count >= 1; count >>= 1) {
            hipLaunchKernelGGL(poseidon_node_kernel, dim3((unsigned)((count + PS_THREADS - 1) / PS_THREADS)), dim3(PS_THREADS), 0, st, nodes, count - 1, count);
            SP_HIP_CHECK(hipGetLastError());
        }
        if (ch) {
            hipLaunchKernelGGL(fri_challenge_kernel, dim3(1), dim3(64), 0, st, nodes, *ch);
            SP_HIP_CHECK(hipGetLastError());
        }
        return SP_OK;
    }
    // throughput-bound levels in pairs: the level of `count` nodes and the one above it, while the upper one still fills the chip
    // (2^17 threads = two waves per SIMD)
    static const bool pairs = std::getenv("SP_MK_NO_PAIRS") == nullptr;   // (A/B switch of tools/merkle_pair_ab.py)
    for (; pairs && (count >> 1) >= MK_PAIR_MIN_UPPER && (count >> 1) > MK_LANES_MAX_NODES; count >>= 2) {
        const uint64_t upper = count >> 1;
        unsigned blocks = (unsigned)((upper + MK_THREADS - 1) / MK_THREADS);
        hipLaunchKernelGGL(node_hash2_kernel, dim3(blocks), dim3(MK_THREADS), 0, st, nodes, upper - 1, upper);
        SP_HIP_CHECK(hipGetLastError());
    }
    for (; count > MK_LANES_MAX_NODES; count >>= 1) {
        unsigned blocks = (unsigned)((count + MK_THREADS - 1) / MK_THREADS);
        hipLaunchKernelGGL(node_hash_kernel, dim3(blocks), dim3(MK_THREADS), 0, st, nodes, count - 1, count);
        SP_HIP_CHECK(hipGetLastError());
    }
    // the upper levels, four per launch (8 -> 4 -> 2 -> 1 nodes per block)
    while (count >= 1) {
        const uint32_t levels = count >= 8 ? 4u : (count >= 4 ? 3u : (count >= 2 ? 2u : 1u));
        const bool last = (count >> levels) == 0;
        hipLaunchKernelGGL(node_hash_lanes_kernel, dim3((unsigned)((count + 7) / 8)), dim3(256), 0, st, nodes, (uint32_t)count, levels,
                           (last && ch) ? *ch : FriChallenge{nullptr, nullptr, nullptr, nullptr});
        SP_HIP_CHECK(hipGetLastError());
        if (last) break;
        count >>= levels;
    }
    return SP_OK;
}

}  // namespace sp
