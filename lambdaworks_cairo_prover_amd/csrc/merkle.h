// Keccak-256 Merkle commitments on gfx950.
// Replaces MerkleTree::<BatchKeccak256Tree>::build / MerkleTree::<Keccak256Tree>::build of lambdaworks-crypto as
// configured by reference src/starks/config.rs:10-20 and called at src/starks/prover.rs:96-104 and
// src/starks/fri/fri_commitment.rs:39.  Node order is the lambdaworks one: nodes[0] = root, children of i at
// 2i+1 / 2i+2, leaves at nodes[n-1 .. 2n-2]; a digest is 32 bytes.
#pragma once
#include "common.h"

namespace sp {

struct digest32 { uint64_t w[4]; };

// The hash of a tree.  KECCAK256 is the reference's configuration (config.rs:10-20).  The two Poseidon kinds are the optional
// backend of BASELINE.json configs[4] (poseidon.h - no counterpart in the reference): a digest is the canonical big-endian element,
// a node is hash(left, right), a leaf is hash_many(row) in a tree over rows (POSEIDON_BATCH: trace and composition commitments,
// whatever the row width) and hash_single(x) in a tree over single elements (POSEIDON_SINGLE: FRI layers).
enum class MerkleHash : int { KECCAK256 = 0, POSEIDON_BATCH = 1, POSEIDON_SINGLE = 2 };

// Hash `n_leaves` leaves into nodes[n_leaves-1 ..]; leaf i = Keccak256(col_0[i] || col_1[i] || ...) with every
// element as canonical 32-byte big-endian.  Columns are device arrays in the device fe layout:
// column j starts at cols + j*col_stride, element i of a column at index i (natural LDE order).
// `order`: where leaf i's elements sit inside a column (leaves are always in natural order).
int merkle_hash_leaves(hipStream_t st, const fe* cols, uint64_t col_stride, uint32_t ncols, uint64_t n_leaves, digest32* nodes,
                       LdeOrder order = LdeOrder{0, 0, 0}, MerkleHash mh = MerkleHash::KECCAK256);
// Same, but the n_leaves digests go to a plain array (coset-sharded commitment: leaves are exchanged before the tree is built).
int merkle_hash_leaves_flat(hipStream_t st, const fe* cols, uint64_t col_stride, uint32_t ncols, uint64_t n_leaves, digest32* leaves_out,
                            LdeOrder order = LdeOrder{0, 0, 0}, MerkleHash mh = MerkleHash::KECCAK256);
// Keccak leaf hashing in two launches, for commitments whose columns arrive over PCIe: the head absorbs the first MK_HEAD_COLS = 17
// columns - 4 x 136 bytes, whole blocks of the sponge - as soon as they exist and leaves every leaf's 25-word state in `state`
// (word k of leaf i at state[k * n_leaves + i]); the tail continues with the other ncols - 17 columns, pads and writes the leaf
// digests where merkle_hash_leaves would.  Same digests as the one-launch form; supported for the row widths merkle_split_supported says.
constexpr uint32_t MK_HEAD_COLS = 17;
inline bool merkle_split_supported(uint32_t ncols) { return ncols == 34 || ncols == 43; }
int merkle_hash_leaves_head(hipStream_t st, const fe* cols, uint64_t col_stride, uint64_t n_leaves, uint64_t* state, LdeOrder order);
int merkle_hash_leaves_tail(hipStream_t st, const fe* cols, uint64_t col_stride, uint32_t ncols, uint64_t n_leaves, const uint64_t* state, digest32* nodes,
                            LdeOrder order);
// The Fiat-Shamir step that follows a FRI layer's commitment (fri/mod.rs:45-50: append the root, sample zeta), done by the
// launch that produces the root so that the layers of the commit phase follow each other without a host round trip.
// DefaultTranscript after a challenge holds the 32 reversed digest bytes r; append(root) makes it r || root, one 64-byte
// Keccak block - the same shape as a node hash with "left" = r.  With d = Keccak256(r || root):  new state = reverse(d),
// zeta = the 251 low bits of d read little-endian (transcript.rs:13-43).  All pointers are device memory:
//   state[4]: r as four little-endian words (in/out);  mul_in: a Montgomery constant;  cst_out = zeta * mul_in (Montgomery);
//   root_copy[4]: the root, for the host's own transcript afterwards.
struct FriChallenge { uint64_t* state; const fe* mul_in; fe* cst_out; uint64_t* root_copy; };
// Reduce the inner levels: nodes[i] = Keccak256(nodes[2i+1] || nodes[2i+2]) for i = n_leaves-2 .. 0.
// ch (nullable, n_leaves >= 2): see FriChallenge.
int merkle_reduce(hipStream_t st, digest32* nodes, uint64_t n_leaves, const FriChallenge* ch = nullptr, MerkleHash mh = MerkleHash::KECCAK256);

}  // namespace sp
