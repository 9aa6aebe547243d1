// Stark252 NTT engine for gfx950: multi-stage passes through LDS, radix-2 stages inside a pass.
//
// Replaces the lambdaworks FFTPoly calls of the reference: Polynomial::interpolate_fft (src/starks/trace.rs:107),
// interpolate_offset_fft (src/starks/constraints/evaluation_table.rs:32), evaluate_offset_fft
// (src/starks/prover.rs:117, src/starks/fri/fri_commitment.rs:36).
//
// A size-2^k transform is a chain of passes; each pass runs r consecutive radix-2 stages of the plain in-place algorithm
// (Cooley-Tukey DIT: bit-reversed in -> natural out; Gentleman-Sande DIF: natural in -> bit-reversed out) on tiles staged
// in LDS.  The s = 0 pass works on contiguous tiles, the others on tiles of R = 2^r rows at stride 2^s x G adjacent
// elements (G*32 B contiguous per row -> coalesced) together with the twiddles of exactly those butterflies.
// All memory orders are chosen so the prover never needs a bit-reversal pass: iNTT leaves coefficients
// bit-reversed, the LDE consumes them bit-reversed and emits natural order.
#pragma once
#include "common.h"
#include <map>
#include <vector>

namespace sp {

constexpr int NTT_TILE_LOG = 10;                 // elements per workgroup tile (2^10 x 32 B = 32 KiB of LDS)
constexpr int NTT_THREADS = 256;
constexpr int NTT_MAX_CONTIG_LOG = NTT_TILE_LOG; // s = 0 pass: one row of up to 2^10 contiguous elements
constexpr int NTT_STRIDED_G_LOG = 2;             // strided passes: 4 adjacent elements (128 B) per row
constexpr int NTT_MAX_STRIDED_LOG = NTT_TILE_LOG - NTT_STRIDED_G_LOG;

struct NttPassArgs {
    const fe* src;
    fe* dst;
    uint64_t src_vec_stride, dst_vec_stride;  // elements between consecutive vectors of the batch
    const fe* small_tw;   // s = 0 pass: w_R^(+-e), e in [0, R/2)   (direction chosen by the host)
    const fe* big_tw;     // strided passes: w_M^e, e in [0, M/2)  (forward roots; the inverse butterfly uses w^-e = -w^(M/2-e))
    const fe* post_table; // DIF only, nullable: multiply the element stored at position pos by post_table[pos]
    const fe* scalar;     // nullable: multiply every stored element by *scalar (device pointer)
    uint32_t logM;        // transform size (twiddles are powers of w_(2^logM))
    uint32_t logL;        // log2 of the array this pass addresses (per vector, per coset in the coset-major LDE)
    uint32_t s, r, g;     // this pass: LOCAL stride 2^s, R = 2^r rows, G = 2^g adjacent elements per row
    uint32_t log_expand;  // EXPAND load: src index = pos >> s on the first pass (zero-padded/replicated LDE input)
    // Local index -> index of the size-2^logM transform: (local << tw_shift) | tw_low.  tw_shift = 0 for whole transforms;
    // natural-order LDE sharded over 2^shard ranks: tw_shift = shard, tw_low = rank (this rank holds the cosets
    // c = c_loc * 2^shard + rank interleaved); coset-major LDE: tw_shift = log2(blowup), tw_low = the coset's global index.
    uint32_t tw_shift, tw_low;
    // coset-major LDE: launch "vector" = column * coset_count + local coset; arrays at column * vec_stride + coset * coset_stride;
    // the global coset index is (local coset << shard_log) | tw_low.  coset_count = 0: plain vectors.
    uint32_t coset_count, shard_log;
    uint64_t src_coset_stride, dst_coset_stride;
    // 1 on every pass but the last one of a transform: the stored data is only brought below 2p; 0: canonical values.
    uint32_t weak_out;
    uint32_t radix4;      // two stages per LDS round trip (filled in by the launcher)
    uint32_t batch;       // vectors per launch (filled in by the launcher)
    uint32_t xcd_map;     // 1: XCD-aware block -> (tile, vector) mapping (needs tiles % 8 == 0)
};

enum NttLoadMode { NTT_LOAD_INPLACE = 0, NTT_LOAD_GATHER_BITREV = 1, NTT_LOAD_EXPAND = 2 };
enum NttStoreMode { NTT_STORE_INPLACE = 0, NTT_STORE_SCATTER_BITREV = 1 };

class NttEngine {
  public:
    explicit NttEngine(hipStream_t stream) : stream_(stream) {}
    ~NttEngine();
    // device table w_(2^k)^e, e in [0, 2^(k-1)); cached per k
    int roots(int k, const fe** out);
    // device table w_(2^k)^(-e), e in [0, 2^(k-1)); only for small k (pass twiddles)
    int inv_roots_small(int k, const fe** out);

    // bit-reversed input -> natural output, forward roots, in place.  batch vectors at `stride` elements.
    int dit_bitrev_to_natural(fe* data, int k, uint32_t batch, uint64_t stride);
    // natural input -> bit-reversed output, inverse roots, UNSCALED (x 2^k), in place;
    // post_table (nullable, device, 2^k entries indexed by output position) multiplies the result.
    // src (nullable): read the input from another array with the same layout and leave it untouched.
    int dif_natural_to_bitrev_inverse(fe* data, int k, uint32_t batch, uint64_t stride, const fe* post_table, const fe* src = nullptr);
    // natural -> natural forward DFT (evaluate_fft), out of place (src != dst).
    int forward_natural(const fe* src, fe* dst, int k, uint32_t batch, uint64_t src_stride, uint64_t dst_stride, fe* final_dst = nullptr);
    // natural -> natural inverse DFT including the 1/2^k factor (interpolate_fft); result in `data`, tmp = same-size scratch.
    int inverse_natural(fe* data, fe* tmp, int k, uint32_t batch, uint64_t stride);
    // LDE: coeffs = n = 2^k "h-scaled" coefficients (c_j h^j) in bit-reversed order; dst = N = n*2^logb natural-order
    // evaluations p(h w_N^i).  (zero-padded size-N DIT whose first log2(b) stages are replication.)
    // shard_log/shard_rank: compute only the cosets c = c_loc * 2^shard_log + shard_rank (dst holds n * 2^(logb - shard_log)
    // elements per vector, element (q, c_loc) at q * 2^(logb - shard_log) + c_loc).
    int lde_from_bitrev(const fe* coeffs, fe* dst, int k, int logb, uint32_t batch, uint64_t src_stride, uint64_t dst_stride,
                        int shard_log = 0, int shard_rank = 0);
    // The same evaluations in COSET-MAJOR order: dst column v holds 2^(logb - shard_log) arrays of n elements, array c_loc =
    // the coset c = c_loc * 2^shard_log + shard_rank, element m = p(h w_N^(m b + c))  (natural-order index m b + c).
    int lde_coset_major(const fe* coeffs, fe* dst, int k, int logb, uint32_t batch, uint64_t src_stride, uint64_t dst_stride,
                        int shard_log = 0, int shard_rank = 0);

    // dst[i] = src[i] * base^i * c  (natural index), c nullable. Used by the coset variants of sp_ntt.
    int scale_by_powers(fe* data, uint64_t n, uint32_t batch, uint64_t stride, const fe& base, const fe* c);

    hipStream_t stream() const { return stream_; }

  private:
    int launch_pass(bool dif, int load_mode, int store_mode, const NttPassArgs& a, uint32_t batch);
    hipStream_t stream_;
    std::map<int, fe*> roots_;
    std::map<int, fe*> inv_small_;
    fe* d_scalar_ = nullptr;
};

// host helpers
fe host_primitive_root(int k);          // w of order 2^k (lambdaworks get_primitive_root_of_unity)

}  // namespace sp
