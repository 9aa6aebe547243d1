// Device-resident STARK prover state: one object per proof, driven round by round (the transcript stays on the
// host; roots go out, challenges come in).  Mirrors the round structure of reference src/starks/prover.rs:532-766.
#pragma once
#include "ctx.h"
#include "stark_kernels.h"
#include "cairo_air_host.h"
#include "aux_kernels.h"
#include "trace_kernels.h"
#include "cairo_host.h"
#include <vector>
#include <memory>
#include <array>
#include <algorithm>

namespace sp {

struct ProofOptionsHost { uint8_t blowup_factor; uint64_t fri_number_of_queries; uint64_t coset_offset; uint8_t grinding_factor; };

// Host form of sp_air_desc (include/stark252_hip.h).
struct AirOpHost { uint8_t op; uint16_t a, b; };   // as the caller wrote it: operands are indices of earlier ops
struct AirDescHost {
    uint32_t main_cols = 0, aux_cols = 0;
    std::vector<uint32_t> offsets, degrees, exemptions;
    uint32_t num_transition_exemptions = 1, degree_bound_factor = 1;
    std::vector<AirOpHost> ops;
    std::vector<fe> consts;
    uint32_t n_rap = 0, aux_kind = 0;
    sp_aux_trace_fn aux_fn = nullptr; void* aux_user = nullptr;   // aux_kind 2: build_auxiliary_trace supplied by the caller
    std::vector<BoundaryConstraint> boundary;
};

struct Openings {
    uint32_t n_queries = 0, n_layers = 0, n_cols = 0, depth0 = 0;
    bool values_canonical_be = false;                            // the four value arrays hold wire-format bytes, not Montgomery limbs
    std::vector<fe> trace_evals, comp_evals;                     // [q][C], [q][2]
    std::vector<digest32> main_paths, aux_paths, comp_paths;     // [q][depth0]
    std::vector<fe> fri_evals, fri_evals_sym;                    // [q][L]
    std::vector<digest32> fri_paths, fri_paths_sym;              // [q][sum_k (depth0-k)]
};

class HostPool;   // prover.cpp: parked host threads for the upload pipeline

class StarkProver : public sp_deletable {
  public:
    StarkProver(sp_ctx* ctx) : c_(ctx) {}
    ~StarkProver() override;

    int setup(uint64_t n, uint32_t main_cols, uint32_t aux_cols, bool has_rc_builtin, const ProofOptionsHost& opt);
    // round 1: interpolate + LDE + Merkle of one trace segment (0 = main, 1 = aux); rows = row-major n x cols, ABI encoding
    // rows_on_device: `rows` is device memory (same row-major ABI encoding) — no PCIe copy inside the round
    // src = TRACE_HOST_COLUMNS: column-major [cols][n] host memory (column j at src + j * col_stride * 32; col_stride 0 = n), in the
    // DEVICE layout (col_enc < 0) or an ABI encoding (col_enc = sp_fe_encoding): every column group is one DMA, no host gather
    // src = TRACE_DEVICE_BUILD: `rows` points at a TraceBuildInput - the register states and the memory of a run go up (24 B per step,
    // 32 B per cell) and the table is written by the device (trace_kernels.h)
    enum TraceSource { TRACE_HOST_ROWS = 0, TRACE_DEVICE_ROWS = 1, TRACE_HOST_COLUMNS = 2, TRACE_DEVICE_BUILD = 3 };
    struct TraceBuildInput { const TracePlan* plan; TraceImage* image; };
    int commit_trace(int segment, const uint8_t* rows, uint32_t cols, uint8_t root_out[32], TraceSource src = TRACE_HOST_ROWS,
                     int col_enc = -1, uint64_t col_stride = 0);
    // round 1, Cairo auxiliary segment built on the device from the resident main trace (reference cairo/air.rs:660-729)
    int commit_aux_cairo(const PublicInputs& pub, const fe rap[3], uint8_t root_out[32]);
    // round 2: constraint composition, H1/H2 split, LDE and commitment
    int composition(const fe rap[3], const std::vector<BoundaryConstraint>& bcs, const std::vector<fe>& b_alpha,
                    const std::vector<fe>& b_beta, const std::vector<fe>& t_alpha, const std::vector<fe>& t_beta,
                    const std::vector<uint32_t>& degrees, const std::vector<uint32_t>& exemptions, uint8_t root_out[32]);
    // Optional, before composition() of the Cairo AIR: the exact constraint check on the trace (what decides between the 2n-point
    // and the whole-domain evaluation) needs the RAP challenges and the boundary constraints but none of the alpha / beta
    // coefficients - queued here it runs while the caller samples those ~110 challenges and composition() builds its tables.
    int composition_precheck(const fe rap[3], const std::vector<BoundaryConstraint>& bcs, uint32_t n_transitions);
    // round 2 for an AIR given as a constraint program (reference traits.rs:15-119 + evaluator.rs:38-260); rap = its RAP
    // challenges (appended to the program's constants); also sets the frame offsets used by rounds 3 and 4.
    int composition_air(const AirDescHost& air, const std::vector<fe>& rap, const std::vector<fe>& b_alpha, const std::vector<fe>& b_beta,
                        const std::vector<fe>& t_alpha, const std::vector<fe>& t_beta, uint8_t root_out[32]);
    // round 3: H1(z^2), H2(z^2), t_j(z g^ofs_k) for every frame row k (row-major [k][j])
    int ood(const fe& z, fe* h1_z2, fe* h2_z2, std::vector<fe>& trace_ood);
    // round 4
    int deep_fri_begin(const fe& gamma, const fe& gamma_p, const std::vector<fe>& trace_gammas /*[j*2+k]*/, uint8_t root0_out[32]);
    int fri_fold_commit(const fe& zeta, uint8_t root_out[32], fe* last_value, int* is_last);
    // The whole commit phase after layer 0 without a host round trip per layer (one GPU): the launch that produces a layer's
    // root also takes the transcript step (append root, sample zeta - merkle.h FriChallenge) and leaves zeta's fold constant
    // in device memory for the next layer.  state32 = the transcript buffer after that zeta was sampled (32 bytes); roots_out =
    // the roots of the layers committed here (those after the last one committed before the call), which the caller feeds to its
    // own transcript afterwards.
    // Several ranks: with a stream-ordered transport the sharded layers are part of the chain (their digest exchange and root
    // all-gather sit on the compute stream); with blocking hooks it is available from the first layer every rank holds whole (the
    // sharded layers in front of it go through fri_fold_commit, one exchange each).  The caller passes the zeta of the layer to fold next.
    bool fri_chain_available() const { return stage_ == 6 && logn_ >= 2 && fri_layer_ >= 1 && (!fri_sharded(fri_layer_ - 1) || comm_async()); }
    int fri_commit_chain(const fe& zeta0, const uint8_t state32[32], std::vector<std::array<uint8_t, 32>>& roots_out, fe* last_value);
    int grind(const uint8_t challenge[32], uint8_t factor, uint64_t* nonce_out);
    // values_canonical_be: the opened field elements come back as their canonical 32-byte big-endian encodings (the proof's wire format,
    // written by the gather's launch-mate on the device) in the same `fe`-sized slots - for the whole-proof drivers, whose serializer
    // then copies bytes instead of converting 7 - 8 thousand elements on one host thread.  The round-level ABI keeps Montgomery values.
    int open(const std::vector<uint64_t>& iotas, Openings& out, bool values_canonical_be = false);

    // sp_prewarm (capi_prove.cpp): host-side plumbing of a first proof, and round 1's kernels at the real shape on arena contents
    int warm_plumbing(bool host_rows);
    int warm_round1();
    // AIRs other than Cairo: frame rows of the transition constraints (reset to {0, 1} by setup)
    void set_frame_offsets(const std::vector<uint32_t>& ofs) { offsets_ = ofs; }
    uint32_t frame_rows() const { return (uint32_t)offsets_.size(); }
    bool ready() const { return ready_; }
    uint64_t n() const { return n_; }
    uint64_t N() const { return N_; }
    uint32_t cols() const { return C_; }
    uint32_t fri_layers() const { return logn_; }
    // per-round device time of the last proof (ms, hipEvent)
    float round_ms[5] = {0, 0, 0, 0, 0};

  private:
    // One commitment tree.  Single GPU: `sub` is the whole tree (lambdaworks node order, 2 leaves - 1 nodes), top == sub.
    // Sharded: `sub` is the subtree over this rank's CONTIGUOUS 1/G of the leaves, `top` the replicated tree over the G
    // subtree roots; root = top[0].
    struct TreeBuf { digest32* sub = nullptr; digest32* top = nullptr; uint64_t sub_leaves = 0; };
    void free_all();
    // small device -> host read-back on the compute stream through a pinned slot, waiting by polling the stream (a blocking
    // synchronisation costs ~20 us of wake-up per Fiat-Shamir round trip; a proof has ~35 of them)
    int readback(void* dst_host, const void* src_dev, size_t bytes);
    int wait_stream();
    int alloc(void** p, size_t bytes);
    void release(void* p, size_t bytes);
    std::vector<uint8_t> h_up_fri_, h_up_open_;   // host sides of small asynchronous uploads (fri_commit_chain, open)
    std::vector<uint64_t> h_idx_open_;
    int alloc_tree(TreeBuf& t, uint64_t leaves_total, bool sharded);
    int setup_impl(uint64_t n, uint32_t main_cols, uint32_t aux_cols, bool has_rc_builtin, const ProofOptionsHost& opt);
    bool ready_ = false;   // setup() completed: every buffer below exists
    // Commitment over `L` leaves this rank holds in local natural order (leaf l = global leaf (l << logG) | rank): hash,
    // exchange the digests so that every rank owns a contiguous range, reduce the subtree, combine the G roots.
    // ch (FRI commit chain): the launch that produces the root also takes the transcript step (merkle.h, FriChallenge); nothing is read
    // back, root_out is not written
    int commit_local(const fe* cols_dev, uint64_t stride, uint32_t ncols, uint64_t L, LdeOrder order, TreeBuf& tree, uint8_t root_out[32],
                     bool single_element_tree = false, const FriChallenge* ch = nullptr);
    // the hash of the commitments (sp_set_option SP_OPT_MERKLE_BACKEND): rows of columns, or the single elements of a FRI layer
    MerkleHash merkle_hash(bool single_element_tree) const {
        return c_->opt_merkle_backend == SP_MERKLE_POSEIDON ? (single_element_tree ? MerkleHash::POSEIDON_SINGLE : MerkleHash::POSEIDON_BATCH)
                                                            : MerkleHash::KECCAK256;
    }
    int commit_columns(const fe* cols_dev, uint64_t stride, uint32_t ncols, TreeBuf& tree, uint8_t root_out[32]) {
        return commit_local(cols_dev, stride, ncols, Nl_, lde_order(), tree, root_out);
    }
    int commit_segment_resident(int segment, uint32_t cols, uint8_t root_out[32]);
    int commit_trace_pipelined(int segment, const uint8_t* rows_host, uint32_t table_cols, uint8_t root_out[32], uint32_t c_begin = 0, uint32_t c_count = 0,
                               bool window_only = false, uint32_t binary_cols = 0);
    // The first `binary_cols` columns of the main segment hold only 0 and 1 (the sixteen instruction flags of a Cairo trace, reference
    // src/cairo/air.rs:29-46: 47 % of the table): the row-major upload sends them as one bit per cell.  A hint, checked cell by cell by
    // the gather threads - a table that breaks it (an invalid trace) is uploaded again in full, so the bytes never depend on it.
    uint32_t binary_cols_hint_ = 0;
    uint64_t* d_flagbits_ = nullptr; uint64_t flagbits_words_ = 0;
    static constexpr int SP_RETRY_RAW_UPLOAD = 1000;   // internal: the packed upload met a cell that is neither 0 nor 1
    // several ranks, row-major host table: every rank uploads the columns of its role only, the trace columns are all-gathered
    int commit_trace_rows_sharded(int segment, const uint8_t* rows_host, uint32_t cols, uint8_t root_out[32], uint32_t binary_cols = 0);
    uint32_t pending_up_groups_ = 0; uint64_t pending_up_bytes_ = 0; double pending_up_gather_ms_ = 0, pending_up_host_ms_ = 0;
    int shard_mode_ = 2;
    bool shard_interp_ = false;    // this shape interpolates by column and all-gathers coefficients (SP_OPT_SHARD_INTERPOLATION, setup())
    int commit_trace_columns(int segment, const uint8_t* cols_host, uint32_t cols, int col_enc, uint64_t col_stride, uint8_t root_out[32]);
    int commit_trace_built(const TraceBuildInput& in, uint8_t root_out[32]);
    // upload pipeline bookkeeping (sp_last_upload_stats): per column group the DMA interval on the copy stream, the moment the
    // group is usable and the moment the compute stream is done with it
    static constexpr int UPLOAD_MAX_GROUPS = 48;
    struct UploadTimers { hipEvent_t dma0 = nullptr, dma1 = nullptr, ready = nullptr, done = nullptr; };
    UploadTimers up_ev_[UPLOAD_MAX_GROUPS];
    hipEvent_t up_start_ = nullptr;
    int ensure_upload(uint32_t groups);
    int ensure_ring_and_pool();   // page-locked ring slots + parked gather threads of the row-major upload
    int finish_upload_stats(uint32_t groups, uint64_t bytes, double gather_ms, double host_ms, int kind);
    hipStream_t copy_stream_ = nullptr;                       // host-buffer uploads (commit_trace_pipelined)
    static constexpr int UPLOAD_SLOTS = 4;                    // ring of chunk slots: the gather may run three chunks ahead of the DMA
    hipEvent_t ev_dma_[UPLOAD_SLOTS] = {};
    // Two-launch leaf hashing of an uploaded segment (merkle.h): the head over the first 17 columns has run (state in d_scratch_)
    bool leaf_head_done_ = false;
    // decides it for a segment of `cols` columns that arrives over PCIe and launches the head once `cols_extended` >= 17 of them exist
    int maybe_leaf_head(uint32_t cols, uint32_t cols_extended, const fe* lde);
    bool pool_bound_ = false;                                 // the gather workers have been placed (prover_upload.cpp)
    void* h_stage_[UPLOAD_SLOTS] = {}; size_t stage_bytes_ = 0;   // pinned staging ring, one chunk (<= 32 MB) each
    HostPool* pool_ = nullptr;
    // elements of d_scratch_: inverse arrays and their scratch (<= 7 local LDE columns), OOD folds (>= 4n and the
    // per-level power tables of up to five points, which dominate for tiny traces)
    // order of the evaluations inside the trace / composition LDE columns this rank holds (coset-major, common.h)
    LdeOrder lde_order() const { return LdeOrder{1u, logb_ - logG_, logn_}; }
    uint64_t scratch_elems() const { return std::max<uint64_t>(std::max<uint64_t>(Nl_ * 7, 4 * n_), 8192); }
    int composition_core(const CompositionConsts& K, const std::vector<fe>& points, const AirProgram* prog_dev, const fe* ex_roots_dev,
                         bool allow_sub_coset, uint8_t root_out[32]);

    sp_ctx* c_;
    ProofOptionsHost opt_{};
    uint64_t n_ = 0, N_ = 0;
    uint32_t logn_ = 0, logb_ = 0, logN_ = 0, Cm_ = 0, Ca_ = 0, C_ = 0;
    // Sharding: G = min(world, blowup) groups; the rank with role r = rank mod G holds the LDE points with
    // global index i = r (mod G) - the cosets c = c_loc * G + r - as Nl_ = N / G local points, local natural index
    // l = i div G.  Ranks beyond the blowup factor are replicas of the role they share (world_ > G_).
    uint32_t world_ = 1, wrank_ = 0;   // communicator size and rank
    uint32_t G_ = 1, rank_ = 0, logG_ = 0;   // groups, this rank's role, log2(G_)
    uint64_t Nl_ = 0;
    fe* d_local_ = nullptr;    // [Nl] 32-byte items: local leaf digests (send side of the exchange)
    fe* d_recv_ = nullptr;     // [Nl] 32-byte items: the digests of this rank's contiguous leaf range, by source rank
    fe* d_gather_ = nullptr; uint64_t gather_cap_ = 0;   // all-gather landing zone, grown on demand
    fe* d_cstage_ = nullptr; uint32_t cpr_max_ = 0;       // [world][cpr_max][n] coefficient all-gather (column-sharded interpolation)
    digest32* d_roots_ = nullptr;                          // [world] subtree roots
    fe* d_small_ = nullptr;                                // out-of-domain values of this rank's columns and their all-gather
    fe* d_fullN_ = nullptr;                                // [N] whole-domain scratch when FRI layer 0 is sharded (exceptional paths)
    int* d_flags_all_ = nullptr;                           // [world] flags of the row-sharded trace check
    // the slice of the trace rows this rank checks in round 2 (the Cairo constraint check is row-local: the trace is replicated)
    uint64_t check_row0() const { return (world_ > 1 && n_ >= 256ull * world_) ? (n_ / world_) * wrank_ : 0; }
    uint64_t check_rows() const { return (world_ > 1 && n_ >= 256ull * world_) ? (wrank_ + 1 == world_ ? n_ - (n_ / world_) * wrank_ : n_ / world_) : n_; }
    int ensure_gather(uint64_t elems);
    int ensure_deep_scratch(uint64_t elems);
    fe* d_deepx_ = nullptr; uint64_t deepx_cap_ = 0;   // DEEP inverses when they outgrow the shared scratch
    int full_domain_buffer(fe** out);
    // stream_ordered: the caller consumes the result on the compute stream only (or waits for that stream itself): with a transport
    // that can enqueue on a stream the exchange takes its place between the kernels instead of costing a host round trip
    int all_gather(const void* send_dev, void* recv_dev, uint64_t bytes_per_rank, bool stream_ordered = false);
    // The same in two halves: begin() starts the exchange behind everything queued on the compute stream so far - on the
    // communication stream when the transport is stream-ordered, by blocking in the hook otherwise - and end() makes the compute
    // stream wait for it.  What is queued on the compute stream between the two runs beside the exchange.
    static constexpr int COMM_BLOCKS = 4;
    hipStream_t comm_stream_ = nullptr;
    hipEvent_t ev_comm_fork_ = nullptr, ev_comm_done_[COMM_BLOCKS] = {};
    bool comm_async() const { return c_->allgather_async != nullptr; }
    int all_gather_begin(const void* send_dev, void* recv_dev, uint64_t bytes_per_rank, int slot);
    int all_gather_end(int slot);
    // recv[s] = the block rank s addressed to this role: send = [G][bytes], recv = [G][bytes]
    int exchange_blocks(const void* send_dev, void* recv_dev, uint64_t bytes_per_block, bool stream_ordered = false);
    ShardMap shard_map() const { return ShardMap{logb_, logG_, rank_}; }
    bool has_rc_ = false;
    fe h_, hinv_, g_;                       // coset offset, its inverse, trace generator
    std::vector<void*> allocs_;             // buffers outside the arena (grown on demand, or when the arena could not be had)
    uint64_t alloc_bytes_ = 0;              // bytes of those
    // One device allocation for everything setup() sizes: a context that proves traces of different shapes re-carves it instead
    // of freeing and allocating ~25 buffers (22 GB at config #3) - hipFree / hipMalloc of that took 35 ms on some boxes of the
    // pool and 0.4 - 1.9 s on others (profiles/r03_pinned_upload.txt).  It only ever grows.
    uint8_t* arena_ = nullptr;
    uint64_t arena_cap_ = 0, arena_off_ = 0;
    bool measuring_ = false; uint64_t measured_ = 0;
    void publish_device_bytes() { c_->prover_device_bytes = arena_cap_ + alloc_bytes_; }
    fe *d_coeffs_ = nullptr, *d_lde_ = nullptr, *d_t1_ = nullptr, *d_t2_ = nullptr;
    fe* d_trace_ = nullptr;  // [C][n] the trace itself, natural order (kept for the constraint check of round 2)
    fe *d_h12s_ = nullptr, *d_h12_ = nullptr, *d_scratch_ = nullptr;  // scratch: 4N elements
    TreeBuf tree_main_, tree_aux_, tree_comp_;
    // FRI layer k: N >> k evaluations; layers below fri_rep_ are sharded (local natural order, N >> k >> logG elements and
    // a sharded tree), the others replicated on every rank
    std::vector<fe*> d_fri_evals_;
    std::vector<TreeBuf> fri_trees_;
    uint32_t fri_rep_ = 0;                  // first replicated layer (0: the whole FRI is replicated)
    bool fri_sharded(uint32_t k) const { return k < fri_rep_; }
    uint32_t fri_layer_ = 0;                // number of committed layers so far
    fe fri_offset_, fri_offset_inv_;        // h^(2^layer) and its inverse
    fe half_, binv_;                        // 1/2, 1/blowup
    CompositionConsts* d_comp_consts_ = nullptr;
    CompositionConsts* d_comp_consts_chk_ = nullptr;   // the constants of an early constraint check (composition_precheck)
    std::unique_ptr<CompositionConsts> h_comp_chk_;    // its host copy (stays put until the upload has happened)
    bool check_pending_ = false;
    AirProgram* d_air_prog_ = nullptr; fe* d_ex_roots_ = nullptr; uint32_t ex_roots_cap_ = 0;
    DeepConsts* d_deep_consts_ = nullptr;
    unsigned long long* d_nonce_ = nullptr;
    uint8_t* d_fri_chain_ = nullptr; uint32_t fri_chain_layers_ = 0;   // [state 32 B][L x constants][L x zeta constants][L x roots]
    void* h_pin_ = nullptr;   // 4 KB of pinned host memory for readback()
    void* h_open_pin_ = nullptr; size_t h_open_cap_ = 0;   // page-locked landing zone of the openings' download (kept across proofs)
    int* h_wide_ = nullptr;   // pinned: the presort's "wide address" flag, copied behind the sorts on the side stream
    // Side stream: latency-bound work that does not wait for the next challenge runs beside the compute stream instead of in
    // its way - a batch inversion is one chain of ~260 dependent field products (~0.3 ms whatever the size).
    //   * boundary denominators 1 / (x - g^step) of round 2 (shape and public inputs only): during round 1;
    //   * DEEP denominators 1 / (x - z g^k), 1 / (x - z^2) of round 4 (known once z is sampled): during round 3;
    //   * the range-check half of the Cairo auxiliary trace beside its memory half.
    hipStream_t side_stream_ = nullptr;
    hipEvent_t ev_side_fork_ = nullptr, ev_side_deep_ = nullptr, ev_side_bnd_ = nullptr, ev_side_aux_ = nullptr;
    int* d_flag_side_ = nullptr;            // [4] flags of the side-stream work: DEEP inverses, boundary inverses, presort
    int ensure_side();
    int prefetch_deep_inverses();           // from ood(): z_ is set
    bool deep_pref_ = false;
    fe* d_bpre_ = nullptr; uint64_t bpre_cap_ = 0;   // [3][2n] boundary inverses + [3][2n] scratch
    std::vector<fe> bpre_points_; bool bpre_valid_ = false;
    // the challenge-free part of the Cairo auxiliary trace (sorts) beside round 1's transforms and hashing
    const PublicInputs* presort_pub_ = nullptr; bool presorted_ = false;
    hipEvent_t ev_side_presort_ = nullptr;
    std::vector<fe> pm_addr_h_, pm_val_h_;  // get_pub_memory_addrs and the matching values (host copies behind the async uploads)
    int public_memory_lists(const PublicInputs& pub);
    int ensure_aux_workspace(uint64_t pm);
    int launch_aux_presort();               // from commit_trace(0, ..) once the main trace columns are queued
  public:
    // round 2's boundary denominators ahead of time (whole-proof entry points call it before round 1; optional)
    int prefetch_boundary_inverses(const std::vector<uint64_t>& steps);
    // optional, before commit_trace(0, ..) of a Cairo proof: sort the memory accesses and the offsets while round 1 runs
    void request_aux_presort(const PublicInputs& pub) { presort_pub_ = &pub; }
    // optional, before commit_trace(0, ..) from a row-major host table: columns [0, count) are 0 / 1 in a valid trace
    void hint_binary_columns(uint32_t count) { binary_cols_hint_ = count; }
  private:
    fe* d_memcols_ = nullptr;               // natural-order main-trace columns 19..29 kept for the auxiliary trace
    void* d_auxws_ = nullptr; size_t auxws_bytes_ = 0; uint64_t auxws_pm_cap_ = 0;
    AuxWorkspace auxws_{};
    fe* d_hfull_ = nullptr; bool h_full_ = false;  // general (degree >= 2n) composition polynomial: N/2 coefficients per half
    fe* d_hnat_ = nullptr;    // natural-order staging of the same (exceptional path)
    fe* d_post_comp_ = nullptr; fe* d_post_deep_ = nullptr;   // shape-only post-factor tables (setup)
    fe* d_post_comp0_ = nullptr;                               // the same for c0 = 0 (one coset per rank)
    fe z_; fe h1_z2_, h2_z2_;
    std::vector<fe> trace_ood_;
    std::vector<uint32_t> offsets_{0, 1};   // transition offsets of the AIR (frame rows); Cairo: {0, 1}
    int stage_ = 0;  // 0 new, 1 setup, 2 main committed, 3 aux committed, 4 composition, 5 ood, 6 fri running, 7 fri done
};

// The prover of a context with the host-side buffers of the round-level ABI: one object whichever entry point created it, so a
// caller can size and warm everything with sp_prove_setup while its trace is still being built and prove with any of them.
struct ProverHolder : public sp_deletable {
    StarkProver prover;
    Openings open;
    std::vector<uint8_t> trace_evals, comp_evals, fri_evals, fri_evals_sym;
    float round_ms[5] = {0, 0, 0, 0, 0};
    explicit ProverHolder(sp_ctx* c) : prover(c) {}
};
ProverHolder* prover_holder(sp_ctx* c, bool create);

// Whole proof on the device: generate_cairo_proof (reference src/cairo/air.rs:1165-1171) + serialize
// (src/starks/proof/stark.rs:161-218). main_trace: row-major n x cols in the context encoding.
int cairo_prove(sp_ctx* ctx, const uint8_t* main_trace, uint64_t n, uint32_t cols, const PublicInputs& pub,
                const ProofOptionsHost& opt, std::vector<uint8_t>& proof_out, float round_ms[5],
                StarkProver::TraceSource src = StarkProver::TRACE_HOST_ROWS, int col_enc = -1, uint64_t col_stride = 0);
// Whole proof for an AIR given as a constraint program: `prove::<F, A>` (reference src/starks/prover.rs:532-766) + serialize.
// main_trace: row-major n x air.main_cols in the context encoding (host memory).
int air_prove(sp_ctx* ctx, const AirDescHost& air, const uint8_t* main_trace, uint64_t n, const ProofOptionsHost& opt,
              std::vector<uint8_t>& proof_out);

}  // namespace sp
