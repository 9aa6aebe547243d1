// Device kernels of the STARK rounds beyond NTT and Merkle: Cairo constraint composition, composition-polynomial
// split, out-of-domain evaluation, DEEP composition, FRI folding, grinding, query gathers.
// Reference code each one replaces is cited at the declaration.
#pragma once
#include "common.h"
#include "merkle.h"

#include <algorithm>
#include <cstddef>

namespace sp {

constexpr int CAIRO_MAX_TRANSITIONS = 50;
constexpr int CAIRO_MAX_BOUNDARY = 8;
constexpr int CAIRO_MAX_BLOWUP = 128;       // ProofOptions::blowup_factor is a u8 power of two (options.rs:21-26)
// program AIRs (sp_air_desc): up to 64 transition constraints and 16 boundary constraints share the constant block below
constexpr int COMP_MAX_BOUNDARY = 16;
constexpr int COMP_MAX_TERMS = 64 + COMP_MAX_BOUNDARY;

// out[q] = c * base^(bitrev_bits(q)) for q < count (bitrev_bits = 0: natural exponent q)
int gen_power_table(hipStream_t st, fe* out, uint64_t count, uint32_t bitrev_bits, const fe& base, const fe& c);

// den[d*N + i] = h*w_N^i - point[d]   for d < ndist  (then batch-inverted by the caller).
// Replaces the N-long zerofier tables of reference src/starks/constraints/evaluator.rs:58-72 and the Ruffini
// divisions of src/starks/prover.rs:436-473 (evaluation form).
// Coset sharding: shard = {logb, shard_log, shard_rank}; N is then the LOCAL length n * 2^(logb - shard_log) and local index
// q * b_loc + c_loc stands for the global LDE index q * b + c_loc * 2^shard_log + shard_rank.
struct ShardMap { uint32_t logb, shard_log, shard_rank; };
int coset_minus_points(hipStream_t st, fe* den, uint64_t N, uint32_t logN, const fe* roots_N, const fe& h, const fe* points_host, uint32_t ndist,
                       ShardMap shard = ShardMap{0, 0, 0});
// out[q*b + c_loc*G + r] = gathered[r][q*b_loc + c_loc] for 32-byte items (field elements or digests): reassembles the
// natural LDE order from the all-gathered per-rank coset shards.
int interleave_shards(hipStream_t st, const void* gathered, void* out, uint64_t n, ShardMap shard);

// Constants of one composition evaluation (device copy lives in CompositionConsts_dev).
struct CompositionConsts {
    fe h;                       // coset offset
    fe rap[3];                  // alpha_memory, z_memory, z_range_check (reference src/cairo/air.rs:469-473)
    fe g_last;                  // g^(n-1): root of the single transition exemption X - g^(n-1) (traits.rs:49-79)
    fe b16, b32, b48, b15, two; // constants of the instruction-decoding constraints (air.rs:883-912)
    fe bvalue[COMP_MAX_BOUNDARY];                                           // boundary values
    uint32_t bcol[COMP_MAX_BOUNDARY];                                       // boundary columns
    uint32_t bden[COMP_MAX_BOUNDARY];                                       // index of the inverse-denominator array
    uint64_t bstep[COMP_MAX_BOUNDARY];                                      // boundary rows (trace check only)
    uint32_t n_boundary, n_transitions, main_cols, has_rc_builtin;
    // the per-coset tables last: a proof uploads the block only as far as its blowup factor reaches (composition_consts_bytes)
    fe zerofier[CAIRO_MAX_BLOWUP];                                          // 1/(x^n - 1) per coset (evaluator.rs:156-171)
    fe coef[CAIRO_MAX_BLOWUP][COMP_MAX_TERMS];                              // alpha_k * x^(D-D_k) + beta_k per coset (transitions, then boundary)
};
// bytes of a CompositionConsts that a proof with `cosets` LDE cosets reads
inline size_t composition_consts_bytes(uint32_t cosets) {
    return offsetof(CompositionConsts, coef) + (size_t)std::min<uint32_t>(cosets, CAIRO_MAX_BLOWUP) * COMP_MAX_TERMS * sizeof(fe);
}

// ConstraintEvaluator::evaluate (reference src/starks/constraints/evaluator.rs:38-260) with CairoAIR::compute_transition
// (src/cairo/air.rs:743-767, helpers :869-1160) fused per LDE point.  lde: column-major [C][col_len] natural order;
// point i of the `count` points is element i << stride_log of every column; binv: [ndist][count] inverse boundary
// denominators; out: [count].  (stride_log = 0: the whole local LDE domain; stride_log = logb - 1: the 2n points of the
// cosets 0 and b/2.)
int cairo_composition(hipStream_t st, const fe* lde, uint64_t count, uint64_t col_len, uint32_t stride_log, uint32_t logN, uint32_t logb,
                      const fe* roots_N, const CompositionConsts* consts_dev, const fe* binv, fe* out, uint32_t shard_log = 0, uint32_t shard_rank = 0);
// validate_trace (reference src/starks/debug.rs:13-104) on the device: *flag_dev |= 1 unless every transition
// constraint vanishes on every row it is enforced on and every boundary value matches.  trace: [C][n] natural order.
// rows row0 .. row0 + rows - 1 only (rows = 0: to the end): the ranks of a sharded prover check a slice each and combine the flags
int cairo_trace_check(hipStream_t st, const fe* trace, uint64_t n, const CompositionConsts* consts_dev, int* flag_dev, uint64_t row0 = 0, uint64_t rows = 0);

// ---- AIRs given as a constraint program (include/stark252_hip.h sp_air_desc; reference trait src/starks/traits.rs:15-119)
constexpr int AIR_MAX_OPS = 2048, AIR_MAX_LIVE = 64, AIR_MAX_CONSTS = 96, AIR_MAX_OFFSETS = 8, AIR_MAX_EXEMPT_KINDS = 4, AIR_MAX_TRANSITIONS = 64;
// Device form of one op: the host assigns every value a slot of a small per-point value file (liveness analysis in
// composition_air), so a long straight-line program needs AIR_MAX_LIVE values per point, not one per op.
//   0 LOAD(a = row, b = col) -> dst   1 CONST(a = idx) -> dst   2 ADD / 3 SUB / 4 MUL (a, b = slots) -> dst   5 OUT(a = constraint, b = slot)
struct AirOpDev { uint8_t op, pad; uint16_t a, b, dst; };
struct AirProgram {
    uint32_t n_ops, n_offsets, offsets[AIR_MAX_OFFSETS];
    uint32_t ex_kind[AIR_MAX_TRANSITIONS];    // per constraint: 0 = enforced on every row, else 1 + index into ex_count
    uint32_t ex_count[AIR_MAX_EXEMPT_KINDS];  // rows exempted for that kind (the last ex_count rows of the trace)
    uint32_t ex_rows[AIR_MAX_TRANSITIONS];    // per constraint: its own exemption count (trace check)
    AirOpDev ops[AIR_MAX_OPS];
    fe consts[AIR_MAX_CONSTS];                // constants followed by the RAP challenges
};
// ConstraintEvaluator::evaluate (evaluator.rs:38-260) for a program AIR; K carries the per-coset coefficients, zerofier and
// boundary data exactly as for Cairo (its Cairo-only fields are ignored); ex_roots[j] = g^(n-1-j).
int air_composition(hipStream_t st, const fe* lde, uint64_t count, uint64_t col_len, uint32_t stride_log, uint32_t logN, uint32_t logb,
                    const fe* roots_N, const CompositionConsts* consts_dev, const AirProgram* prog_dev, const fe* ex_roots,
                    const fe* binv, fe* out, uint32_t shard_log = 0, uint32_t shard_rank = 0);
int air_trace_check(hipStream_t st, const fe* trace, uint64_t n, const CompositionConsts* consts_dev, const AirProgram* prog_dev, int* flag_dev);

// Split of the composition polynomial (reference src/starks/prover.rs:250-252, evaluation_table.rs:27-33):
// X = unscaled bit-reversed size-N inverse transform of the N evaluations; writes the h-scaled bit-reversed coefficient
// arrays (n each) of H1 (even) and H2 (odd): H1s[q] = X[q*b/2] * t2[q], H2s[q] = X[N/2 + q*b/2] * t2[q] * hinv,
// t2[q] = N^-1 h^(-rev_n(q)).
int split_composition(hipStream_t st, const fe* X, uint64_t n, uint32_t logb, const fe* t2, const fe& hinv, fe* H1s, fe* H2s);

// Degree check and general split for traces that violate their constraints (the reference still emits a proof for them,
// reference src/starks/prover.rs:106-123 subsamples the longer FFT): see stark_kernels.hip.
int high_coeff_check(hipStream_t st, const fe* X, uint64_t N, uint32_t logb, int* flag_dev);
int split_composition_full(hipStream_t st, const fe* X, uint64_t N, const fe* t_half, const fe& hinv, fe* H1f, fe* H2f);

// One level of the out-of-domain evaluation (reference src/starks/prover.rs:301-304, frame.rs:67-83; Horner replaced
// by a bit-reversed-order fold): out[v][p][q'] = sum_t in[v][(p)][q' + t*M/2^l] * yp[p][t]   for q' < M/2^l.
// first level: in has no point dimension (in_points = 1), later levels have in_points = points.
int fold_eval_level(hipStream_t st, const fe* in, uint64_t in_vec_stride, uint32_t in_points, uint64_t M, uint32_t l,
                    const fe* yp /*[points][2^l]*/, uint32_t points, uint32_t vectors, fe* out /*[vectors][points][M>>l]*/);

struct DeepConsts {
    fe gamma_h1, gamma_h2;       // gamma, gamma'
    fe c_h;                      // gamma*H1(z^2) + gamma'*H2(z^2)
    fe c_t[AIR_MAX_OFFSETS];     // sum_j gamma_{j,k} t_j(z g^ofs_k)
    fe gammas[AIR_MAX_OFFSETS][64];  // gamma_{j,k}, k = frame row, j = column (<= 61 columns)
    uint32_t cols, rows;         // rows = number of frame rows (transition offsets): 2 for Cairo
};
// compute_deep_composition_poly (reference src/starks/prover.rs:410-482) in evaluation form:
// p0(x) = sum_k (sum_j g_jk t_j(x) - c_tk) / (x - z g^ofs_k) + (g H1 + g' H2 - c_h) / (x - z^2);
// inv: [rows + 1][count] = 1/(x - z g^ofs_k) for each frame row, then 1/(x - z^2).
// `count` points, point q = element (q << shift) of every column (columns at col_stride).
int deep_composition(hipStream_t st, const fe* lde, const fe* h1, const fe* h2, uint64_t count, uint64_t col_stride, uint32_t shift,
                     const DeepConsts* consts_dev, const fe* inv, fe* out, LdeOrder order, uint32_t frame_rows = 2);

// fold_polynomial + FriLayer::new (reference src/starks/fri/fri_functions.rs:4-27, fri_commitment.rs:30-47) in evaluation
// form: next[i] = (cur[i] + cur[i+M/2]) / 2 + zeta * (cur[i] - cur[i+M/2]) / (2 x_i),  x_i = offset * w_M^i, i < M/2.
// roots_N: half table of w_N; M = N >> layer. c = zeta / (2 * offset).
// Sharded layer: M = the elements this rank holds, local index l = global index (l << shard_log) | shard_rank.
// the same with the leaf digests of the produced layer (Keccak256 trees, one GPU): leaves_out[i] = Keccak256(next[i] as 32-byte BE)
int fri_fold_hash(hipStream_t st, const fe* cur, fe* next, uint64_t M, uint32_t logN, uint32_t layer, const fe* roots_N, const fe& half, const fe& c,
                  const fe* c_dev, digest32* leaves_out);
int fri_fold(hipStream_t st, const fe* cur, fe* next, uint64_t M, uint32_t logN, uint32_t layer, const fe* roots_N, const fe& half, const fe& c,
             uint32_t shard_log = 0, uint32_t shard_rank = 0, const fe* c_dev = nullptr);   // c_dev (device, nullable) replaces c

// generate_nonce_with_grinding (reference src/starks/grinding.rs:17-48): smallest nonce in [start, start+count) whose
// Keccak256(challenge || nonce_le)[0..8] (BE) has >= factor trailing zeros; *result_dev = min(*result_dev, nonce).
int grind_range(hipStream_t st, const uint8_t challenge[32], uint8_t factor, uint64_t start, uint64_t count, unsigned long long* result_dev);


// dst column v (coset-major order, the `len` evaluations this rank holds) = src column v (natural order of the whole domain,
// columns at src_stride); local natural index l is the global index (l << shard_log) | shard_rank
int natural_to_coset_major(hipStream_t st, const fe* src, uint64_t src_stride, fe* dst, uint64_t len, uint32_t ncols, LdeOrder order,
                           uint32_t shard_log = 0, uint32_t shard_rank = 0);
// Every gather of the query phase in ONE launch (fri/mod.rs:74-127, prover.rs:484-529 open ~45 arrays; a launch per array is
// ~0.4 ms of submission latency per proof).  A job copies 32-byte items into the staging block:
//   kind 0: rows   - item (r, j) = base[j * stride + idx[r]]           (count rows x width columns)
//   kind 1: paths  - item (r, l) = sibling on level l of leaf idx[r]    (count leaves x width levels, lambdaworks node order)
struct GatherJob { const void* base; uint64_t stride_or_leaves; uint64_t idx_off; uint64_t out_off; uint32_t count, width, kind, pad; };
int gather_jobs(hipStream_t st, const GatherJob* jobs_dev, uint32_t njobs, uint32_t max_items, const uint64_t* idx_dev, fe* out);

}  // namespace sp
