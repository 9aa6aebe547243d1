/* stark252_hip.h — C ABI of the MI355X (gfx950) STARK proving hot path for the lambdaworks Cairo prover.
 *
 * The reference (lambdaclass/lambdaworks_cairo_prover) has no FFI: `prove` (src/starks/prover.rs:532-766) calls
 * the lambdaworks-math / lambdaworks-crypto traits directly.  This header is what a Rust `extern "C"` block in
 * src/starks/prover.rs (or a replacement lambdaworks backend, like the existing `metal` feature, Cargo.toml:39)
 * binds; INTEGRATION.md shows the binding.  Each entry point cites the reference code it replaces.
 *
 * Conventions
 *   - return 0 (SP_OK) on success, a negative SP_E_* code otherwise; nothing aborts or throws across the ABI
 *     (the reference `unwrap()`s; a shim maps non-zero to ProvingError::WrongParameter, prover.rs:40-43).
 *   - field elements are contiguous 32-byte records in one of two encodings (sp_fe_encoding):
 *       SP_FE_MONT_LIMBS  4 x u64, limb 0 MOST significant, Montgomery form R = 2^256 — the in-memory layout of
 *                         lambdaworks `FieldElement<Stark252PrimeField>` (zero-copy from a Rust `&[FE]`);
 *       SP_FE_CANON_BE    canonical 32-byte big-endian (`to_bytes_be`, the proof wire format).
 *   - the caller owns every input/output buffer; a context owns its device memory; calls on one context are
 *     serial (the round structure is sequential); different contexts may be used from different threads.
 *   - host pointers unless a parameter is named *_dev.
 */
#ifndef STARK252_HIP_H
#define STARK252_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    SP_OK = 0,
    SP_E_INVALID_ARG = -1,   /* null pointer, non power-of-two size, unknown enum */
    SP_E_NO_DEVICE = -2,     /* no gfx950 device visible / HIP runtime unusable */
    SP_E_HIP = -3,           /* a HIP call failed; see sp_last_error() */
    SP_E_ALLOC = -4,
    SP_E_STATE = -5,         /* round called out of order */
    SP_E_ZERO_INVERSE = -6,  /* batch inverse of a zero element (lambdaworks panics) */
    SP_E_UNSUPPORTED = -7,
    SP_E_PROGRAM = -8        /* Cairo front-end: undecodable instruction, missing memory cell, step limit */
};

typedef enum { SP_FE_MONT_LIMBS = 0, SP_FE_CANON_BE = 1 } sp_fe_encoding;

/* ProofOptions — reference src/starks/proof/options.rs:21-26 */
typedef struct {
    uint8_t blowup_factor;            /* a power of two, 2 .. 128 (the reference's u8); n x blowup <= 2^30 */
    uint64_t fri_number_of_queries;
    uint64_t coset_offset;
    uint8_t grinding_factor;
} sp_proof_options;
/* The presets and checked constructors of the same file (host arithmetic, no device): SecurityLevel (options.rs:5-12),
 * ProofOptions::new_secure (:35-75; every level: blowup 4, grinding 20, queries 31 / 41 / 55 conjecturable, 80 / 104 / 140 provable),
 * new_with_checked_security (:78-102) and new_with_checked_provable_security (:107-129) - restated as the reference computes them,
 * including the latter's use of the u8's LEADING zeros - for a field of `field_bits` bits (252 for Stark252; the reference is generic
 * over the field: F::field_bit_size()).  InsecureOptionError (errors.rs) comes back as SP_E_INVALID_ARG with sp_last_error() =
 * "InsecureOptionError::FieldSize" (field_bits <= security_target + 40) or "InsecureOptionError::SecurityBits". */
typedef enum { SP_SEC_CONJECTURABLE_80 = 0, SP_SEC_CONJECTURABLE_100 = 1, SP_SEC_CONJECTURABLE_128 = 2,
               SP_SEC_PROVABLE_80 = 3, SP_SEC_PROVABLE_100 = 4, SP_SEC_PROVABLE_128 = 5 } sp_security_level;
int sp_proof_options_new_secure(int security_level, uint64_t coset_offset, sp_proof_options* out);
int sp_proof_options_checked(uint8_t blowup_factor, uint64_t fri_number_of_queries, uint64_t coset_offset, uint8_t grinding_factor,
                             uint8_t security_target, int provable, uint32_t field_bits, sp_proof_options* out);

typedef struct {
    int device;        /* HIP device ordinal */
    int fe_encoding;   /* sp_fe_encoding of every field-element buffer crossing the ABI on this context */
} sp_config;

typedef struct sp_ctx sp_ctx;

/* Blocking all-gather hook for coset sharding across GPUs: every rank contributes `bytes_per_rank` bytes at send_dev
 * (device memory) and receives world*bytes_per_rank bytes at recv_dev, rank-major. Must have completed when it returns. */
typedef int (*sp_allgather_fn)(void* user, const void* send_dev, void* recv_dev, uint64_t bytes_per_rank);

const char* sp_version(void);
/* Version of this header: bumped whenever a structure (sp_air_desc, sp_openings, sp_cairo_public_inputs, sp_proof_options) changes
 * layout, an entry point or option key is added, or a call changes meaning (4: round 5's sp_comm_measure / sp_comm_time_ms /
 * sp_proof_options_* / sp_proof_file_verify / SP_OPT_HOST_RANKS family and sp_set_collective keeping the prover across re-installs
 * of the same world; 5: sp_fe_mul).  A binding compares it (and sp_air_desc_size against its own idea of the struct) when it loads the library, so
 * a stale build fails at load time with "rebuild the library" instead of with a missing symbol or shifted fields later. */
#define SP_ABI_VERSION 5
int sp_abi_version(void);
uint64_t sp_air_desc_size(void);
const char* sp_last_error(void);          /* thread-local description of the last failure */
int sp_device_count(int* count_out);      /* number of visible HIP devices (0 without a GPU) */
/* CPUs the host side of the library may really use: hardware threads cut down by the affinity mask and the cgroup CPU quota
 * (what sizes the gather threads of sp_cairo_prove and the front-end's trace builder). */
int sp_host_cpus(int* count_out);
/* This process' share of them: sp_host_cpus / the ranks sharing the host (SP_OPT_HOST_RANKS, see sp_set_option), at least 1 - the
 * figure every host-side thread count is derived from.  ranks_out (optional): the rank count in force. */
int sp_host_cpu_budget(int* budget_out, int* ranks_out);
/* NUMA placement on multi-socket hosts: restricts the CALLING THREAD - and every thread it creates afterwards - to the CPUs of the
 * NUMA node the device hangs off (within the affinity mask it already has), so that the tables a prover process builds are
 * first-touched on that node: the page-locked staging of the host-buffer entry points lives there, and a table on the other node
 * crosses the socket link on its way in (sp_cairo_prove from a far table: 37 - 43 GB/s instead of 52 - 56).  The equivalent of
 * `numactl --cpunodebind=<node of the GPU>`; call it first thing in a one-process-per-GPU prover.  *node_out (nullable): the node,
 * -1 when it is unknown (single-node host, no sysfs, no device) - nothing is changed then.  Touches the HIP runtime (the device's PCI
 * address), creates no context. */
int sp_host_bind_to_device(int device, int* node_out);

int sp_ctx_create(sp_ctx** ctx_out, const sp_config* cfg);
void sp_ctx_destroy(sp_ctx* ctx);

/* ---- multi-GPU: one process and one context per GPU; the LDE cosets are sharded over `world` ranks
 * (world a power of two; beyond the blowup factor the surplus ranks are replicas). Every rank calls the same sequence (sp_cairo_prove or the round-level calls)
 * with the same inputs and obtains the same roots / proof bytes. The data-path exchanges are all-gathers of 32-byte leaf
 * digests, composition evaluations and DEEP evaluations; everything else is local. Either install a hook ... */
int sp_set_collective(sp_ctx* ctx, int world, int rank, sp_allgather_fn fn, void* user);
/* (Re-installing with the SAME world - another rank, other hooks - keeps the prover's device arena: the next proof re-carves it for the new
 * rank; a different world releases it.  Every installation resets the optional hooks below: install them again afterwards.) */
/* ... or let the library own an RCCL communicator (ncclAllGather on the context stream over xGMI): rank 0 obtains a 128-byte
 * id with sp_comm_unique_id and distributes it out of band (e.g. torch.distributed broadcast); every rank then calls
 * sp_comm_init_rccl with it. */
int sp_comm_unique_id(uint8_t id_out[128]);
int sp_comm_init_rccl(sp_ctx* ctx, const uint8_t id[128], int world, int rank);
/* Optional stream-ordered all-gather (same layout as sp_allgather_fn): enqueue the exchange on `hip_stream` (a hipStream_t) and
 * return without waiting.  With it the sharded prover splits the coefficient all-gather of a trace segment into column blocks on a
 * stream of its own, so that the exchange of one block runs beside the inverse transforms of the next and the LDE of the one
 * before (prover.rs:161-185 has no such dependency between columns).  sp_comm_init_rccl installs it (ncclAllGather on that
 * stream); the blocking hook stays in use for every other exchange.  Call after sp_set_collective. */
typedef int (*sp_allgather_async_fn)(void* user, const void* send_dev, void* recv_dev, uint64_t bytes_per_rank, void* hip_stream);
int sp_set_collective_async(sp_ctx* ctx, sp_allgather_async_fn fn);
/* Timing-only transport for projections on fewer GPUs than ranks (bench.py --project-ranks): nothing is exchanged - the own block
 * lands where a collective would put it, the other ranks' blocks are zero-filled (the HBM writes a receive costs) - so ONE rank's
 * share of a sharded proof runs at its real kernel sizes.  The proof bytes that come out are meaningless; never use it to prove. */
int sp_comm_init_null(sp_ctx* ctx, int world, int rank);
/* Optional second hook ("Merkle combine"): blocking all-to-all of equal blocks, send_dev =
 * [world][bytes_per_pair] (block d goes to rank d), recv_dev = [world][bytes_per_pair] (block s came from rank s).  With it
 * the 32-byte leaf digests of a commitment travel once (each rank receives only the contiguous 1/world of the leaves whose
 * subtree it reduces); without it the prover falls back to an all-gather of the digests.  sp_comm_init_rccl installs both
 * (grouped ncclSend/ncclRecv on the context stream).  Call after sp_set_collective. */
typedef int (*sp_alltoall_fn)(void* user, const void* send_dev, void* recv_dev, uint64_t bytes_per_pair);
int sp_set_alltoall(sp_ctx* ctx, sp_alltoall_fn fn);
/* Stream-ordered form of the all-to-all (same layout as sp_alltoall_fn), optional: with it AND sp_set_collective_async the digest
 * exchange and the root all-gather of a commitment are enqueued on the prover's compute stream between the kernels that produce and
 * consume them instead of each costing a host round trip, and the FRI commit phase runs through its sharded layers without any - the
 * launch that finishes a layer's top tree takes the transcript step on the device (fri/mod.rs:37-67 is one dependent chain).
 * sp_comm_init_rccl installs it (grouped ncclSend / ncclRecv on the stream it is given).  Call after sp_set_collective. */
typedef int (*sp_alltoall_async_fn)(void* user, const void* send_dev, void* recv_dev, uint64_t bytes_per_pair, void* hip_stream);
int sp_set_alltoall_async(sp_ctx* ctx, sp_alltoall_async_fn fn);
/* Collective traffic of this context since creation: out = {world, all-gather calls, bytes contributed to all-gathers,
 * all-to-all calls, bytes sent in all-to-alls, bytes received in all collectives}. */
int sp_comm_stats(sp_ctx* ctx, uint64_t out[6]);
/* ... and how long they took, in ms since creation: out = {stream-ordered collectives - an event pair around each on the stream it was
 * enqueued on, i.e. what the exchange occupied that stream for, the wait for the slowest peer included -, blocking collectives - wall
 * clock around the hook}.  Synchronizes the device. */
int sp_comm_time_ms(sp_ctx* ctx, double out[2]);
/* Checks the installed transport (RCCL or hooks): one all-gather and, if installed, one all-to-all of rank-stamped blocks of
 * bytes_per_block bytes (a multiple of 8); every rank must call it.  0 = both deliver the layout documented above. */
int sp_comm_selftest(sp_ctx* ctx, uint64_t bytes_per_block);
/* Times the installed transport - every rank must call it: all-gathers and (if installed) all-to-alls of bytes_per_rank bytes per
 * rank, one untimed of each and then three timed ones whose MEDIAN counts.  out = {all-gather ms, its GB/s per link and direction,
 * all-to-all ms, its GB/s per link and direction, bytes per rank, world}: what a rank received from the others / time / (world - 1)
 * links - the unit of SP_OPT_LINK_GBS - as the MINIMUM over the ranks, so every rank holds the same figures.
 * SP_OPT_SHARD_INTERPOLATION = 2 decides from the all-gather's rate DIVIDED BY SP_LINK_MEASURED_MARGIN (a measured figure near the
 * threshold must not flip the mode from run to run) unless the caller stated SP_OPT_LINK_GBS.  sp_comm_init_rccl runs it once
 * (64 MB per rank; environment SP_COMM_MEASURE_MB, 0 = not at all).  bytes_per_rank = 0: only read the stored figures back (zeros
 * when nothing was measured).  The outcome is agreed between the ranks: everything local (payload buffers, events) is prepared
 * before the first collective, a status word per rank goes round on every path, and a rank that could not prepare or could not time
 * its collectives makes EVERY rank drop the figures and return non-zero - the ranks never end up with different rates, hence never
 * with different interpolation modes. */
#define SP_LINK_MEASURED_MARGIN 1.25
int sp_comm_measure(sp_ctx* ctx, uint64_t bytes_per_rank, double out[6]);
/* The decision rule of SP_OPT_SHARD_INTERPOLATION = 2 as a pure function: 1 when interpolating a trace segment by column (+ an
 * all-gather of the coefficients) beats interpolating every column on every rank - 64 x 1.35e11 < (groups - 1) x link x log2(rows):
 * the butterfly rate of one MI355X against what groups - 1 links deliver. */
int sp_model_shard_interpolation(double link_gbs_per_direction, uint32_t groups, uint32_t log2_rows);
/* Tuning knobs of the sharded prover (defaults in parentheses):
 *   SP_OPT_FRI_SHARD_MIN_LOG (16)  FRI layers with at least 2^value leaves keep their evaluations and trees sharded; smaller
 *                                  layers are all-gathered once and continue replicated (fri/mod.rs:20-72 is sequential in the layers);
 *   SP_OPT_SHARD_INTERPOLATION (2) 1: the size-n inverse transforms of a trace segment are split by column over the ranks and the
 *                                  coefficients all-gathered (prover.rs:161-185, trace.rs:104-110); 0: every rank interpolates all columns;
 *                                  2: whichever is faster for the shape on the link model - a rank saves (1 - 1/G) of the inverse transforms
 *                                  and receives (1 - 1/G) of the coefficients, which pays when (G - 1) x link rate x log2 n > 64 x 1.35e11.
 *   SP_OPT_LINK_GBS (46)           GB/s one xGMI link delivers per direction, for that decision.  Unset: the rate sp_comm_measure found
 *                                  (sp_comm_init_rccl measures once per communicator), else 46 (76.8 GB/s x an assumed 0.6).
 *   SP_OPT_UPLOAD_THREADS (24)     host threads that gather the column groups of a row-major host trace into pinned memory
 *                                  (sp_cairo_prove / sp_commit_trace from host buffers above 64 MB).
 *   SP_OPT_HOST_RANKS (env)        how many ranks (processes, one context and GPU each) share this HOST.  Process-wide.  Every host thread
 *                                  count of the library is a share of the CPUs the process may use (affinity mask, cgroup quota) divided by
 *                                  this: the gather pool is min(SP_OPT_UPLOAD_THREADS, 2 x CPUs / ranks), the front-end's loops CPUs / ranks;
 *                                  with more ranks than CPUs the Fiat-Shamir waits block instead of polling.  Default: the environment -
 *                                  SP_HOST_RANKS, else LOCAL_WORLD_SIZE (what torch.distributed.run exports), else 1.  0 = back to that.
 *   SP_OPT_MERKLE_BACKEND (SP_MERKLE_KECCAK256)  the hash of every commitment of the context (trace, composition and FRI trees,
 *                                  sp_merkle_build*).  SP_MERKLE_KECCAK256 is the reference's configuration (src/starks/config.rs:10-20)
 *                                  and the only one whose proofs the reference verifies.  SP_MERKLE_POSEIDON (BASELINE.json configs[4]; NO
 *                                  reference counterpart at the pinned revision) is Starknet's Poseidon over Stark252 as later lambdaworks
 *                                  versions configure it: digest = the canonical 32-byte big-endian element, node = hash(left, right),
 *                                  leaf of a row of columns = hash_many(row), leaf of a single-element tree (FRI layers) =
 *                                  hash_single(x).  Transcript and grinding stay Keccak; the proof layout does not change.
 *                                  Such proofs are checked with sp_cairo_verify_backend / sp_air_verify_backend.
 *   SP_OPT_DEVICE_TRACE (1)        sp_cairo_prove_run builds the main trace ON THE DEVICE from the run's register states and memory
 *                                  (build_main_trace, src/cairo/execution_trace.rs:57-87: 24 B per step + 32 B per memory cell cross
 *                                  PCIe instead of the n x cols table); 0: the run's host table goes up column group by column group.
 *   SP_OPT_MERKLE_ONE_COLUMN_ROWS (0)  which Poseidon tree sp_merkle_build / sp_merkle_build_dev build for fe_per_leaf == 1: 0 the
 *                                  single-element tree of a FRI layer (leaf = hash_single(x), fri_commitment.rs:39), 1 the tree over
 *                                  rows of ONE column (leaf = hash_many over one element) - what the prover commits a one-column trace
 *                                  segment with (prover.rs:96-104 batch_commit).  Keccak256 trees have one leaf form; no effect there. */
enum { SP_OPT_FRI_SHARD_MIN_LOG = 1, SP_OPT_SHARD_INTERPOLATION = 2, SP_OPT_UPLOAD_THREADS = 3, SP_OPT_MERKLE_BACKEND = 4, SP_OPT_MERKLE_ONE_COLUMN_ROWS = 5,
       SP_OPT_DEVICE_TRACE = 6, SP_OPT_LINK_GBS = 7, SP_OPT_HOST_RANKS = 8 };
enum { SP_MERKLE_KECCAK256 = 0, SP_MERKLE_POSEIDON = 1 };
int sp_set_option(sp_ctx* ctx, int key, int64_t value);

/* ---- fine-grained layer: the lambdaworks seam the reference calls ------------------------ */

/* Natural-order DFT of n = 2^k elements, in place.
 *   inverse == 0, coset == NULL : Polynomial::evaluate_fft                           (data = coefficients)
 *   inverse == 0, coset != NULL : evaluate_offset_fft(1, None, coset)  (prover.rs:117, fri_commitment.rs:36)
 *   inverse != 0, coset == NULL : Polynomial::interpolate_fft                         (trace.rs:107)
 *   inverse != 0, coset != NULL : Polynomial::interpolate_offset_fft  (constraints/evaluation_table.rs:32)
 * `coset` is one field element in the context's encoding. */
int sp_ntt(sp_ctx* ctx, uint8_t* data, uint64_t n, int inverse, const uint8_t* coset);

/* Batched variant on device memory: `batch` vectors of n elements each, vector v at data_dev + v*n*32,
 * elements in the DEVICE layout (8 x u32 little-endian Montgomery, see sp_fe_to_device). Used by bench.py so the
 * timed region starts with inputs resident in HBM. Asynchronous on the context stream; sp_sync() to wait. */
int sp_ntt_dev(sp_ctx* ctx, void* data_dev, uint64_t n, uint32_t batch, int inverse, const uint8_t* coset);

/* evaluate_polynomial_on_lde_domain (prover.rs:106-123) for `cols` polynomials of n coefficients each
 * (column-major: column j at coeffs + j*n*32): out column j holds n*blowup evaluations p_j(coset * w_N^i). */
int sp_lde(sp_ctx* ctx, const uint8_t* coeffs, uint64_t n, uint32_t cols, uint32_t blowup, const uint8_t* coset,
           uint8_t* out);

/* MerkleTree::<BatchKeccak256Tree|Keccak256Tree>::build (prover.rs:96-104, fri_commitment.rs:39; backends
 * config.rs:10-20): n_leaves rows of fe_per_leaf elements (row-major); leaf = Keccak256(row as 32-byte BE
 * elements), parent = Keccak256(left || right). nodes_out (nullable) receives all 2n-1 nodes, root first,
 * children of i at 2i+1 and 2i+2 (the lambdaworks node order).
 * With SP_OPT_MERKLE_BACKEND = SP_MERKLE_POSEIDON: fe_per_leaf = 1 builds the single-element tree (leaf = hash_single),
 * fe_per_leaf > 1 the tree over rows (leaf = hash_many); the same holds for sp_merkle_build_dev. */
int sp_merkle_build(sp_ctx* ctx, const uint8_t* leaves, uint64_t n_leaves, uint32_t fe_per_leaf,
                    uint8_t root_out[32], uint8_t* nodes_out);

/* MerkleTree::build (prover.rs:96-104 batch_commit, fri_commitment.rs:39) on device memory, for bench.py and callers that
 * already hold the evaluations in HBM: `fe_per_leaf` columns in the DEVICE layout, column j at cols_dev + j*col_stride*32,
 * leaf i = Keccak-256 of the canonical big-endian encodings of column 0..fe_per_leaf-1 at row i.  nodes_dev receives the
 * 2*n_leaves - 1 digests in lambdaworks order (root first).  Asynchronous on the context stream; sp_sync() to wait. */
int sp_merkle_build_dev(sp_ctx* ctx, const void* cols_dev, uint64_t n_leaves, uint32_t fe_per_leaf, uint64_t col_stride,
                        void* nodes_dev);

/* FieldElement::inplace_batch_inverse (constraints/evaluator.rs:69,171; cairo/air.rs:540,561). */
int sp_batch_inverse(sp_ctx* ctx, uint8_t* data, uint64_t n);
/* FieldElement<Stark252PrimeField> Mul / square on the device (src/lib.rs:12-13; every product of the path, e.g. cairo/air.rs:525-572,
 * constraints/evaluator.rs:142-154): out[i] = a[i] * b[i] for n elements in the context's encoding; b == NULL squares a.  The kernels'
 * own Montgomery product and square (csrc/fp.h) on caller-chosen operands - what the parity tests drive with adversarial limbs (with
 * SP_FE_MONT_LIMBS the limbs go to the multiplier as they are). */
int sp_fe_mul(sp_ctx* ctx, const uint8_t* a, const uint8_t* b, uint64_t n, uint8_t* out);

/* Encoding helpers between the ABI encodings and the device layout (host side, no GPU needed). */
int sp_fe_to_device(int fe_encoding, const uint8_t* in, uint64_t n, uint8_t* out_device_layout);
int sp_fe_from_device(int fe_encoding, const uint8_t* in_device_layout, uint64_t n, uint8_t* out);

int sp_sync(sp_ctx* ctx);
/* Duration in milliseconds of the kernels launched by the last sp_*_dev call, measured with HIP events on the context
 * stream; waits for that call to finish. */
int sp_last_kernel_ms(sp_ctx* ctx, float* ms_out);
/* HIP-event timer on the context stream around any sequence of asynchronous calls: sp_timer_start records an event,
 * sp_timer_stop records a second one, waits for it and returns the elapsed device time in milliseconds. bench.py
 * brackets its timed region with these, so the per-launch figure contains no host round trips. */
int sp_timer_start(sp_ctx* ctx);
int sp_timer_stop(sp_ctx* ctx, float* ms_out);

/* PublicInputs — reference src/cairo/air.rs:163-181 (HashMaps flattened to arrays). Field elements in
 * SP_FE_CANON_BE regardless of the context encoding. */
typedef struct {
    uint8_t pc_init[32], ap_init[32], fp_init[32], pc_final[32], ap_final[32];
    uint16_t range_check_min, range_check_max;
    uint32_t n_segments;
    const uint8_t* segment_types;      /* 0 = RangeCheck, 1 = Output */
    const uint64_t* segment_ranges;    /* (start, end) pairs */
    uint64_t n_public_memory;
    const uint8_t* public_memory;      /* (address, value) pairs, 64 bytes each */
    uint64_t num_steps;
} sp_cairo_public_inputs;

/* ---- round-level layer: one call per prover round; the Fiat-Shamir transcript stays with the
 * caller (it is sequential Keccak over < 10 KB): roots go out, challenges come in. Field elements in the context
 * encoding. Call order: sp_prove_setup, sp_commit_trace(0), [sp_commit_trace(1)], sp_composition, sp_ood,
 * sp_deep_fri_commit_begin, sp_fri_fold_commit x log2(n), sp_grind, sp_open; out-of-order calls return SP_E_STATE. */

/* A::new + Domain::new (reference src/starks/prover.rs:549-551, src/starks/domain.rs:20-56): sizes the device buffers
 * for a trace of n rows (power of two), main_cols + aux_cols columns. has_rc_builtin selects the 61-column Cairo layout. */
int sp_prove_setup(sp_ctx* ctx, uint64_t n, uint32_t main_cols, uint32_t aux_cols, int has_rc_builtin, const sp_proof_options* opt);

/* Pre-warm for the reference's one-proof-per-process shape (src/main.rs:85-108: run the VM, prove once, exit): call it - on a
 * thread of its own, or before the VM starts - while the trace does not exist yet, with the shape the proof will have.  It does
 * sp_prove_setup's work and everything else a first proof would otherwise pay on its critical path:
 *   SP_PREWARM_KERNELS    a small valid Cairo proof (2^13 rows) from a device-built trace and the commitment of the same trace through
 *                         the three other input forms, on this context: the first launch of every kernel family, the side streams, the
 *                         auxiliary-trace workspace;
 *   SP_PREWARM_CLOCKS     round 1's kernels at the REAL shape on the arena's contents, column slice by column slice: the size-specific
 *                         kernel variants, and the device at its clocks when the trace arrives (sp_prewarm_cancel ends it early);
 *   SP_PREWARM_HOST_ROWS  the page-locked ring and the parked gather threads of the row-major entry points (sp_cairo_prove,
 *                         sp_commit_trace from host tables above 64 MB).
 * flags = 0 means all of them.  The proof that follows (any entry point, same n / columns / blowup / coset offset) then costs what a
 * warm one does; its bytes are not affected.  With several ranks every rank calls it (the small proofs are sharded proofs). */
enum { SP_PREWARM_KERNELS = 1, SP_PREWARM_CLOCKS = 2, SP_PREWARM_HOST_ROWS = 4, SP_PREWARM_ALL = 7 };
int sp_prewarm(sp_ctx* ctx, uint64_t n, uint32_t main_cols, uint32_t aux_cols, int has_rc_builtin, const sp_proof_options* opt, uint32_t flags);
/* Callable from ANY thread: asks the sp_prewarm that runs on this context - or the next one, if none does yet - to return as soon as it
 * can.  Everything a first proof needs (arena, plumbing, first launches) is still done; the clock ramp (SP_PREWARM_CLOCKS: round 1 at the
 * real shape, column slice by column slice) stops behind the slice in flight.  The intended use: start sp_prewarm on a thread, run the VM,
 * call sp_prewarm_cancel when the trace exists, join the thread, prove - the ramp then lasts exactly as long as the VM did.  A request
 * that finds no prewarm to stop is dropped by the next proof on the context. */
int sp_prewarm_cancel(sp_ctx* ctx);

/* interpolate_and_commit (prover.rs:126-159): rows = row-major n x cols trace segment (0 main, 1 auxiliary).
 * iNTT + LDE + batched Keccak Merkle tree; keeps polynomials, LDE and tree on the device. */
int sp_commit_trace(sp_ctx* ctx, int segment, const uint8_t* rows, uint64_t n, uint32_t cols, uint8_t root_out[32]);

/* The same from host COLUMNS - what interpolate_and_commit itself starts from (`trace.cols()`, prover.rs:130, trace.rs:23-31):
 * column j of the segment at cols + j*col_stride*32 (col_stride in elements, 0 = n); device_layout != 0: elements in the DEVICE
 * layout, 0: context encoding.  One DMA per column group, no host-side gather (see sp_cairo_prove_columns, sp_host_alloc). */
int sp_commit_trace_columns(sp_ctx* ctx, int segment, const uint8_t* cols, uint64_t n, uint32_t n_cols, uint64_t col_stride,
                            int device_layout, uint8_t root_out[32]);

/* CairoAIR::build_auxiliary_trace + interpolate_and_commit of the auxiliary segment, entirely on the device
 * (reference src/cairo/air.rs:660-729, src/starks/prover.rs:199-213): rap = alpha_memory, z_memory, z_range_check
 * sampled by the caller after the main root. Requires sp_commit_trace(0) of a Cairo main trace (34|43 columns). */
int sp_cairo_commit_aux(sp_ctx* ctx, const uint8_t* rap, const sp_cairo_public_inputs* pub, uint8_t root_out[32]);

/* BoundaryConstraint (reference src/starks/constraints/boundary.rs:13-17) */
typedef struct { uint32_t col; uint64_t step; uint8_t value[32]; } sp_boundary_constraint;

/* round_2_compute_composition_polynomial (prover.rs:226-286) for the Cairo AIR: rap = 3 RAP challenges (alpha_memory,
 * z_memory, z_range_check); coeffs = alpha^B (n_boundary), beta^B (n_boundary), alpha^T (n_transitions),
 * beta^T (n_transitions), in the order prover.rs:597-615 samples them. Returns the [H1],[H2] root. */
int sp_composition(sp_ctx* ctx, const uint8_t* rap, const sp_boundary_constraint* boundary, uint32_t n_boundary,
                   const uint8_t* coeffs, uint32_t n_transitions, uint8_t root_out[32]);

/* round_3 (prover.rs:288-325): out = H1(z^2), H2(z^2), then t_j(z g^k) row-major [k][j], k = 0,1: (2 + 2C) elements. */
int sp_ood(sp_ctx* ctx, const uint8_t z[32], uint8_t* out);

/* compute_deep_composition_poly + first FRI layer (prover.rs:347-378, fri/mod.rs:27-33): gammas = gamma, gamma',
 * then the 2C trace gammas (index j*2 + k). */
int sp_deep_fri_commit_begin(sp_ctx* ctx, const uint8_t* gammas, uint8_t root0_out[32]);

/* One FRI fold + commitment (fri/mod.rs:37-67). While *is_last == 0 the output is the next layer's Merkle root; the
 * log2(n)-th call sets *is_last = 1 and outputs fri_last_value (one field element) instead. */
int sp_fri_fold_commit(sp_ctx* ctx, const uint8_t zeta[32], uint8_t root_or_last_out[32], int* is_last);

/* generate_nonce_with_grinding (reference src/starks/grinding.rs:40-48): the smallest qualifying nonce. */
int sp_grind(sp_ctx* ctx, const uint8_t challenge[32], uint8_t factor, uint64_t* nonce_out);

/* fri_query_phase + open_deep_composition_poly (fri/mod.rs:74-127, prover.rs:484-529). All arrays are host memory
 * owned by the context until the next call on it; digests are 32 bytes; field elements in the context encoding. */
typedef struct {
    uint32_t n_queries, n_layers, n_cols, depth0;  /* tree depth of FRI layer k is depth0 - k */
    const uint8_t* trace_evals;     /* [q][n_cols]  trace LDE row at iota                      */
    const uint8_t* comp_evals;      /* [q][2]       H1, H2 at iota                             */
    const uint8_t* main_paths;      /* [q][depth0]  authentication paths, bottom-up            */
    const uint8_t* aux_paths;       /* [q][depth0]                                              */
    const uint8_t* comp_paths;      /* [q][depth0]                                              */
    const uint8_t* fri_evals;       /* [q][n_layers]      layer value at iota mod |D_k|        */
    const uint8_t* fri_evals_sym;   /* [q][n_layers]      value at the symmetric index         */
    const uint8_t* fri_paths;       /* [q][sum_k (depth0 - k)], layers concatenated            */
    const uint8_t* fri_paths_sym;
} sp_openings;
int sp_open(sp_ctx* ctx, const uint64_t* iotas, uint32_t q, sp_openings* out);

/* ---- whole proof: generate_cairo_proof (reference src/cairo/air.rs:1165-1171) + StarkProof::serialize
 * (src/starks/proof/stark.rs:161-218). main_trace = row-major n x cols (34, or 43 with the range-check builtin) in the
 * context encoding. *proof_out is malloc'd; release it with sp_free. The bytes equal the reference prover's. */
int sp_cairo_prove(sp_ctx* ctx, const uint8_t* main_trace, uint64_t n, uint32_t cols, const sp_cairo_public_inputs* pub,
                   const sp_proof_options* opt, uint8_t** proof_out, uint64_t* proof_len);
/* Same with the main trace already resident in device memory (row-major, context encoding): the timed region of bench.py
 * starts with its input in HBM; sp_cairo_prove additionally pays the PCIe copy of n*cols*32 bytes. */
int sp_cairo_prove_dev(sp_ctx* ctx, const void* main_trace_dev, uint64_t n, uint32_t cols, const sp_cairo_public_inputs* pub,
                       const sp_proof_options* opt, uint8_t** proof_out, uint64_t* proof_len);
/* Same from host COLUMNS: column j of the main trace at main_trace_cols + j*col_stride*32 (col_stride in elements, 0 = n),
 * the layout `TraceTable::cols()` produces (reference src/starks/trace.rs:23-31, the input of compute_trace_polys, :104-110).
 * device_layout != 0: elements in the DEVICE layout (sp_fe_to_device); 0: in the context encoding.  No host-side gather: every
 * column group is one DMA overlapped with the transforms of the group before it.  Allocate the buffer with sp_host_alloc and
 * the DMA runs at PCIe speed; ordinary (pageable) memory works too, through the runtime's staging copy. */
int sp_cairo_prove_columns(sp_ctx* ctx, const uint8_t* main_trace_cols, uint64_t n, uint32_t cols, uint64_t col_stride, int device_layout,
                           const sp_cairo_public_inputs* pub, const sp_proof_options* opt, uint8_t** proof_out, uint64_t* proof_len);
/* Page-locked host memory for trace tables (hipHostMalloc): SP_E_ALLOC when the runtime cannot provide it. */
int sp_host_alloc(uint64_t bytes, void** out);
void sp_host_free(void* p);
void sp_free(void* p);
/* How the main trace of the last proof reached the device: out = {kind (0: one copy / resident, 1: row-major host buffer gathered
 * into column groups by host threads, 2: DMA of page-locked host columns, 3: host columns in pageable memory, 4: register states +
 * memory of a run from page-locked memory, trace built on the device, 5: the same from pageable memory), column groups, bytes, host ms until the last chunk had been gathered and sent (kind 1), bytes over that time in GB/s, DMA ms
 * (sum over the groups), DMA GB/s, exposed ms (how long the compute stream waited for column groups in total), longest wait
 * for one group ms, host wall ms of the upload loop}. */
int sp_last_upload_stats(sp_ctx* ctx, double out[10]);
/* Device time (ms, HIP events on the context stream) of rounds 0..4 of the last sp_cairo_prove. */
/* What the last proof on this context did: out[0] = composition path (1: 2n-point evaluation after a clean trace check,
 * 2: whole LDE domain, deg H < 2n, 3: whole domain, deg H >= 2n - a constraint-violating trace), out[1] = FRI layers kept
 * sharded, out[2] = groups the LDE is sharded over, out[3] = 1 when the trace interpolation was split by column. */
int sp_last_proof_info(sp_ctx* ctx, uint32_t out[4]);
/* Device memory the prover of this context holds (trace, coefficients, LDE, trees, FRI layers, staging), in bytes: linear in
 * the trace length for a given column count, blowup factor and world size. */
int sp_prover_device_bytes(sp_ctx* ctx, uint64_t* bytes_out);
int sp_last_round_ms(sp_ctx* ctx, float out[5]);

/* ---- AIRs other than Cairo ------------------------------------------------------------------ */

/* What an implementor of the reference's `AIR` trait provides (src/starks/traits.rs:15-119, context.rs:4-18), with
 * `compute_transition` as a straight-line program over frame cells:
 *   op 0 LOAD  a = frame row (index into `offsets`), b = column of main||aux      -> value
 *   op 1 CONST a = index into consts; indices >= n_consts are the RAP challenges   -> value
 *   op 2 ADD, 3 SUB, 4 MUL   a, b = indices of earlier ops                          -> value
 *   op 5 OUT   a = constraint index, b = index of the op holding its value
 * The reference's examples (src/starks/example/{simple_fibonacci,fibonacci_2_columns,quadratic_air,fibonacci_rap,
 * dummy_air}.rs) are given in this form by lambdaworks_cairo_prover_amd/air.py. */
typedef struct { uint8_t op; uint8_t pad; uint16_t a; uint16_t b; uint16_t pad2; } sp_air_op;
typedef struct { uint32_t col; uint32_t pad; uint64_t step; uint8_t value[32]; /* canonical BE */ } sp_air_boundary;
/* build_auxiliary_trace(main_trace, rap_challenges) of the caller's AIR (traits.rs:25-29), for aux_kind 2: called once, after
 * the main commitment, with the n_rap challenges (32 bytes each, context encoding); must write the row-major n x aux_cols
 * auxiliary trace (context encoding) to aux_rows_out and return 0. */
typedef int (*sp_aux_trace_fn)(void* user, const uint8_t* rap, uint32_t n_rap, uint8_t* aux_rows_out);
typedef struct {
    uint32_t main_cols, aux_cols;            /* AirContext::trace_columns = main_cols + aux_cols (<= 64) */
    uint32_t n_offsets; uint32_t offsets[8]; /* transition_offsets */
    uint32_t n_transitions; uint32_t degrees[64]; uint32_t exemptions[64]; /* transition_degrees / transition_exemptions */
    uint32_t num_transition_exemptions;      /* AirContext::num_transition_exemptions */
    uint32_t degree_bound_factor;            /* composition_poly_degree_bound() / trace_length (1 or 2) */
    uint32_t n_ops; const sp_air_op* ops;    /* <= 2048 ops, <= 64 values alive at any point of the program */
    uint32_t n_consts; const uint8_t* consts; /* canonical BE */
    uint32_t n_rap;                          /* build_rap_challenges: this many transcript_to_field samples */
    uint32_t aux_kind;                       /* build_auxiliary_trace: 0 none, 1 fibonacci_rap permutation column, 2 aux_fn */
    uint32_t n_boundary; const sp_air_boundary* boundary;   /* <= 16, at most 3 distinct steps */
    sp_aux_trace_fn aux_fn; void* aux_user;  /* aux_kind 2 */
} sp_air_desc;

/* prove::<Stark252PrimeField, A> (reference src/starks/prover.rs:532-766) + Serializable::serialize for the AIR `air`.
 * main_trace: row-major n x air->main_cols, context encoding, host memory.  *proof_out is malloc'd (sp_free). */
int sp_air_prove(sp_ctx* ctx, const sp_air_desc* air, const uint8_t* main_trace, uint64_t n, const sp_proof_options* opt,
                 uint8_t** proof_out, uint64_t* proof_len);

/* verify::<Stark252PrimeField, A> (reference src/starks/verifier.rs:559-657) on the host CPU: 1 accept, 0 reject (also for
 * malformed proofs or descriptors). */
int sp_air_verify(const uint8_t* proof, uint64_t proof_len, const sp_air_desc* air, const sp_proof_options* opt);

/* ---- Cairo front-end on the host --------------------------------------------------- */

typedef struct sp_cairo_run sp_cairo_run;  /* register trace + memory + public inputs + main trace */

/* run_program + PublicInputs::from_regs_and_mem + build_main_trace for a hint-free, builtin-free program given as
 * `n_words` canonical-BE field elements (reference src/cairo/runner/run.rs:242-263). */
int sp_cairo_run_program(const uint8_t* program_words, uint64_t n_words, uint64_t max_steps, sp_cairo_run** out);
/* Same with main's entry point at address `entry_pc` (1-based; the compiled program's "main" identifier pc + 1). */
int sp_cairo_run_program_at(const uint8_t* program_words, uint64_t n_words, uint64_t entry_pc, uint64_t max_steps, sp_cairo_run** out);
/* Same for a hint-free program that declares builtins (reference tests/integration_tests.rs:151-172 `rc_program`,
 * `signed_div_rem`; run.rs:211-222): builtins_mask bit 0 = output, bit 1 = range_check.  The builtin base pointers are main's
 * implicit arguments, main returns the advanced pointers; the used range of each builtin segment becomes a memory segment of
 * the public inputs (the range-check segment switches the AIR to its 43 + 18 column layout, cairo/air.rs:623-629; the output
 * cells join the public memory, air.rs:200-206).  Range-checked values must lie in [0, 2^128). */
int sp_cairo_run_program_builtins(const uint8_t* program_words, uint64_t n_words, uint64_t entry_pc, uint64_t max_steps,
                                  uint32_t builtins_mask, sp_cairo_run** out);
/* The 22-word fibonacci program of benches/proofs/fibonacci_70000.proof with index `fib_index`. */
int sp_cairo_run_fibonacci(uint64_t fib_index, sp_cairo_run** out);
/* From cairo-run's binary dumps: .trace (24 B/row LE, register_states.rs:51-78) and .memory (8+32 B/row LE,
 * cairo_mem.rs:35-61), program occupying addresses 1..program_size. */
int sp_cairo_run_from_dumps(const uint8_t* trace, uint64_t trace_len, const uint8_t* memory, uint64_t memory_len,
                            uint64_t program_size, sp_cairo_run** out);
/* The same from what cairo-vm hands the reference IN MEMORY (run_program, reference src/cairo/runner/run.rs:64-240): regs = steps x
 * (ap, fp, pc) relocated register states, (addrs[k], values[k]) = the n_cells relocated memory cells (values 32 bytes each in
 * `fe_encoding`), and the builtin segments run.rs:211-222 reads back (seg_types: 0 = RangeCheck, 1 = Output; seg_ranges: (start, end)
 * pairs) - generate_prover_args (run.rs:242-263) without the binary dump files.  For a shim that keeps cairo-vm (hints, Cairo 1). */
int sp_cairo_run_from_arrays(const uint64_t* regs, uint64_t steps, const uint64_t* addrs, const uint8_t* values, int fe_encoding, uint64_t n_cells,
                             uint64_t program_size, const uint8_t* seg_types, const uint64_t* seg_ranges, uint32_t n_segments, sp_cairo_run** out);
/* The register states and the memory of a run in that form (cells in increasing address order): call with null buffers for the
 * two counts, then with buffers of steps x 3 words, n_cells words and n_cells x 32 bytes. */
int sp_cairo_run_export(const sp_cairo_run* run, int fe_encoding, uint64_t* steps_out, uint64_t* n_cells_out, uint64_t* regs_out, uint64_t* addrs_out,
                        uint8_t* values_out);
void sp_cairo_run_free(sp_cairo_run* run);
/* Where the front-end's time went, in ms: out = {the VM (0 for runs read from dumps), the shape pass of build_main_trace (addresses,
 * offsets, range-check and memory holes, every validity check), the upload image for the device-side trace builder, the n x cols
 * host table (0 until something asks for it: sp_cairo_run_main_trace / _columns, or a proof with SP_OPT_DEVICE_TRACE off)}. */
int sp_cairo_run_timings(const sp_cairo_run* run, double out[4]);
/* Shape of the main trace: n rows (power of two) x cols (34, or 43 with the range-check builtin). */
int sp_cairo_run_shape(const sp_cairo_run* run, uint64_t* n_rows, uint32_t* n_cols, uint64_t* num_steps);
/* Copies the row-major main trace in `fe_encoding` (n*cols*32 bytes). */
int sp_cairo_run_main_trace(const sp_cairo_run* run, int fe_encoding, uint8_t* out);
/* Fills `pi`; the pointers inside stay valid until sp_cairo_run_free(run). */
int sp_cairo_run_public_inputs(const sp_cairo_run* run, sp_cairo_public_inputs* pi);
/* The main trace as the run keeps it: column-major [cols][n_rows] in the DEVICE layout, page-locked when *pinned_out = 1 (the
 * HIP runtime was usable when the run was built).  The pointer lives as long as the run. */
int sp_cairo_run_columns(const sp_cairo_run* run, const void** cols_out, uint64_t* n_rows, uint32_t* n_cols, int* pinned_out);
/* generate_prover_args + generate_cairo_proof (reference src/cairo/runner/run.rs:242-263, src/cairo/air.rs:1165-1171) for a run
 * of this front-end.  By default the register states and the memory of the run go up and the device writes the main trace itself
 * (SP_OPT_DEVICE_TRACE; the host never builds the table); with the option off - or for a run whose memory is not one flat array -
 * the run's own column-major page-locked table goes up by DMA, column group by column group, behind the transforms of the groups
 * before it (sp_cairo_prove_columns on sp_cairo_run_columns + sp_cairo_run_public_inputs). */
int sp_cairo_prove_run(sp_ctx* ctx, const sp_cairo_run* run, const sp_proof_options* opt, uint8_t** proof_out, uint64_t* proof_len);
/* build_main_trace (reference src/cairo/execution_trace.rs:57-87) on the device: the table sp_cairo_prove_run proves (SP_OPT_DEVICE_TRACE),
 * downloaded as sp_cairo_run_main_trace would return it (row-major n x cols, `fe_encoding`) - for callers that want the table, and
 * for the parity tests of the device builder. */
int sp_cairo_run_main_trace_dev(sp_ctx* ctx, const sp_cairo_run* run, int fe_encoding, uint8_t* out);

/* verify_cairo_proof (reference src/cairo/air.rs:1176-1182, src/starks/verifier.rs:559-657) on the host CPU: returns 1 when the
 * proof is accepted, 0 when it is rejected or malformed. Ships with the library so that proofs of shapes without a golden
 * file can be checked where they are produced.
 * STRICTER THAN THE REFERENCE'S PARSER, on purpose: StarkProof::deserialize (src/starks/proof/stark.rs:225-440) reads each part inside
 * the slice its length prefix announces and the nonce from the last eight bytes, so it tolerates padding inside and behind the parts;
 * this verifier accepts only the bytes StarkProof::serialize writes (every prefix equals its part, element length 32, nothing behind the
 * nonce, Poseidon path digests below p) - a proof the reference accepts can be refused here if someone re-framed it.  After a 0,
 * sp_last_error() says which it was by its first word: "rejected:" a well-formed proof failed a verification step; "malformed:" neither
 * parser could read it; "non-canonical framing:" only this strict parser refuses it (re-serialize the proof and try again). */
int sp_cairo_verify(const uint8_t* proof, uint64_t proof_len, const sp_cairo_public_inputs* pub, const sp_proof_options* opt);
/* The same two verifiers for proofs whose commitments use another hash (SP_OPT_MERKLE_BACKEND): merkle_backend = SP_MERKLE_*.
 * With SP_MERKLE_KECCAK256 they are sp_cairo_verify / sp_air_verify. */
int sp_cairo_verify_backend(const uint8_t* proof, uint64_t proof_len, const sp_cairo_public_inputs* pub, const sp_proof_options* opt,
                            int merkle_backend);
int sp_air_verify_backend(const uint8_t* proof, uint64_t proof_len, const sp_air_desc* air, const sp_proof_options* opt, int merkle_backend);
/* Starknet Poseidon over Stark252 on the host (csrc/poseidon.h), for known-answer tests of the Poseidon backend: in = n elements in
 * the given encoding; mode 0: hash_many(in[0..n)), 1: hash(in[0], in[1]) (n = 2), 2: hash_single(in[0]) (n = 1),
 * 3: the Hades permutation of in[0..3) (n = 3; out receives three elements).  out: 32 bytes (96 for mode 3), same encoding. */
int sp_poseidon_host(int fe_encoding, int mode, const uint8_t* in, uint64_t n, uint8_t* out);
/* The reference CLI's `verify` command (src/main.rs:113-143) on the bytes of a proof file - u64_be(len(proof)) || StarkProof::serialize ||
 * PublicInputs::serialize (src/cairo/air.rs:223-276), read back as PublicInputs::deserialize does (:278-450; bytes behind num_steps are
 * ignored, public-memory addresses must fit 64 bits): 1 accepted, 0 rejected or malformed, sp_last_error() as for sp_cairo_verify.
 * benches/proofs/fibonacci_70000.proof - written by the reference itself - is accepted as it is. */
int sp_proof_file_verify(const uint8_t* file, uint64_t file_len, const sp_proof_options* opt);
int sp_proof_file_verify_backend(const uint8_t* file, uint64_t file_len, const sp_proof_options* opt, int merkle_backend);
/* CLI proof file of the reference (src/main.rs:98-102): u64_be(len(proof)) || proof || PublicInputs::serialize
 * (src/cairo/air.rs:223-276). *out is malloc'd; release with sp_free. */
int sp_proof_file_encode(const uint8_t* proof, uint64_t proof_len, const sp_cairo_run* run, uint8_t** out, uint64_t* out_len);

#ifdef __cplusplus
}
#endif
#endif /* STARK252_HIP_H */
