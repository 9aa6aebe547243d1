// Host-buffer uploads of a trace segment (see prover.h): the row-major pipeline (host threads gather column groups into a ring of
// page-locked slots, DMA, rows -> columns, transforms of the groups overlapped) and the column path (one DMA per column group).
#include "prover_internal.h"
#include "cairo_host.h"
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <cctype>
#include <cstdio>
#include <pthread.h>
#include <sched.h>
#include <sys/syscall.h>
#include <unistd.h>

namespace sp {

int StarkProver::ensure_upload(uint32_t groups) {
    if (groups > (uint32_t)UPLOAD_MAX_GROUPS) { sp_set_error("commit_trace: too many column groups"); return SP_E_UNSUPPORTED; }
    if (!copy_stream_) {
        // A stream for the DMAs alone (highest priority; the optional pull copy of SP_UPLOAD_PULL is the only kernel it ever sees).
        // Kernels do NOT belong here: rows -> columns / decode kernels on a highest-priority stream made the transforms running beside
        // them 1.3 - 2.2 x slower for as long as the upload lasted (far more than their own run time; tools/upload_interference.py),
        // and on an ordinary second stream they queued behind the transforms and stalled the ring (24 GB/s).
        int prio_lo = 0, prio_hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
        SP_HIP_CHECK(hipStreamCreateWithPriority(&copy_stream_, hipStreamNonBlocking, prio_hi));
        for (int i = 0; i < UPLOAD_SLOTS; ++i) SP_HIP_CHECK(hipEventCreateWithFlags(&ev_dma_[i], hipEventDisableTiming));
    }
    if (!up_start_) SP_HIP_CHECK(hipEventCreate(&up_start_));
    for (uint32_t g = 0; g < groups; ++g)
        for (hipEvent_t* e : {&up_ev_[g].dma0, &up_ev_[g].dma1, &up_ev_[g].ready, &up_ev_[g].done})
            if (!*e) SP_HIP_CHECK(hipEventCreate(e));
    SP_HIP_CHECK(hipStreamSynchronize(copy_stream_));   // (nothing pending unless an earlier call failed half-way)
    return SP_OK;
}

// Leaf hashing beside an upload that bounds round 1.  When the columns arrive slower than they are transformed (blowup 2 and 4: 0.34
// ms of PCIe per column of 2^19 rows against 0.18 ms of transforms), everything after the last column - its transforms, then ALL of
// the leaf hashing - is exposed.  The first 17 columns are whole blocks of the Keccak sponge: their 4 of the 9 permutations per leaf
// run as soon as those columns are extended, in the gaps the compute stream has anyway, and the commitment continues from the
// saved states (25 words per leaf in the scratch area, free in round 1).  Not when the transforms are the bound (blowup 8 and up):
// the GPU is busy throughout and the 400 bytes per leaf of state traffic would only add to it.
int StarkProver::maybe_leaf_head(uint32_t cols, uint32_t cols_extended, const fe* lde) {
    if (leaf_head_done_ || cols_extended < MK_HEAD_COLS || G_ != 1 || !merkle_split_supported(cols)) return SP_OK;
    if (c_->opt_merkle_backend != SP_MERKLE_KECCAK256) return SP_OK;
    static const bool off = std::getenv("SP_NO_SPLIT_HASH") != nullptr;
    const double transform_ms_per_col = (1.0 + (double)(1u << logb_)) * (double)n_ * logn_ / 2 / 1.35e11 * 1e3;
    const double upload_ms_per_col = (double)n_ * 32 / 50e9 * 1e3;
    if (off || transform_ms_per_col >= 0.9 * upload_ms_per_col) return SP_OK;
    if ((uint64_t)Nl_ * 25 * sizeof(uint64_t) > scratch_elems() * sizeof(fe)) return SP_OK;
    SP_TRY(merkle_hash_leaves_head(c_->stream, lde, Nl_, Nl_, reinterpret_cast<uint64_t*>(d_scratch_), lde_order()));
    leaf_head_done_ = true;
    return SP_OK;
}

// After the commitment's read-back (every event has completed): what the upload cost and how long the compute stream waited
// for it.  kind 1: gathered from a row-major host buffer, 2: DMA of host columns.
int StarkProver::finish_upload_stats(uint32_t groups, uint64_t bytes, double gather_ms, double host_ms, int kind) {
    double dma_ms = 0, exposed = 0, worst = 0;
    for (uint32_t g = 0; g < groups; ++g) {
        float d = 0, w = 0;
        if (hipEventElapsedTime(&d, up_ev_[g].dma0, up_ev_[g].dma1) == hipSuccess) dma_ms += d;
        // the compute stream could have started group g when it was done with group g - 1 (or, for the first one, at the start)
        if (hipEventElapsedTime(&w, g ? up_ev_[g - 1].done : up_start_, up_ev_[g].ready) == hipSuccess && w > 0) { exposed += w; worst = std::max<double>(worst, w); }
    }
    (void)hipGetLastError();
    double* u = c_->upload_stats;
    u[0] = kind; u[1] = groups; u[2] = (double)bytes; u[3] = gather_ms; u[4] = gather_ms > 0 ? bytes / gather_ms * 1e-6 : 0;
    u[5] = dma_ms; u[6] = dma_ms > 0 ? bytes / dma_ms * 1e-6 : 0; u[7] = exposed; u[8] = worst; u[9] = host_ms;
    return SP_OK;
}

// interpolate_and_commit (reference prover.rs:126-159) from host COLUMNS (the layout trace.rs:23-31 `cols()` produces, and what
// sp_cairo_run keeps): a column group is one contiguous DMA straight into the trace area - no gather, no landing slot - and the
// groups stay small (1, 1, 2, 2, ...) so that only the first column's 0.6 ms stay in front of the transforms.
int StarkProver::commit_trace_columns(int segment, const uint8_t* cols_host, uint32_t cols, int col_enc, uint64_t col_stride, uint8_t root_out[32]) {
    const uint32_t col0 = segment == 0 ? 0 : Cm_;
    std::vector<uint32_t> gsize;
    for (uint32_t done = 0; done < cols;) {
        // one, one, then two columns at a time: a group is ready when its DMA is (0.6 ms per column of 2^20 rows) and its
        // transforms take 0.7 ms per column, so with small groups the compute stream waits for the first column only; doubling
        // groups (1, 1, 2, 4, 8, 8, ...) made it wait 4 ms per proof - every group twice the size of the one being transformed
        uint32_t w = done < 2 ? 1u : 2u;
        if (G_ > 1 && shard_interp_) w = cols;        // column-sharded interpolation works on the whole segment
        w = std::min(w, cols - done);
        gsize.push_back(w);
        done += w;
    }
    const uint32_t groups = (uint32_t)gsize.size();
    SP_TRY(ensure_upload(groups));
    // page-locked source (sp_host_alloc, a run built after the context): DMA at PCIe speed; pageable: the runtime's staging copy.
    // SP_UPLOAD_PULL=1 replaces the DMA of a page-locked source by a copy kernel that reads it over PCIe (experiment).
    hipPointerAttribute_t attr{};
    const bool pinned = hipPointerGetAttributes(&attr, cols_host) == hipSuccess && attr.type == hipMemoryTypeHost;
    (void)hipGetLastError();
    static const bool want_pull = std::getenv("SP_UPLOAD_PULL") != nullptr;
    const bool pull = pinned && want_pull && (reinterpret_cast<uintptr_t>(cols_host) % 16 == 0);
    const double t0 = wall_ms();
    fe* coeffs = d_coeffs_ + (uint64_t)col0 * n_;
    fe* trace = d_trace_ + (uint64_t)col0 * n_;
    fe* lde = d_lde_ + (uint64_t)col0 * Nl_;
    SP_HIP_CHECK(hipEventRecord(up_start_, c_->stream));             // the trace area's previous readers are behind this point
    SP_HIP_CHECK(hipStreamWaitEvent(copy_stream_, up_start_, 0));
    const bool sharded_interp = G_ > 1 && shard_interp_ && cols >= G_;
    uint32_t c0 = 0;
    for (uint32_t g = 0; g < groups; c0 += gsize[g], ++g) {
        const uint32_t w = gsize[g];
        fe* dst = trace + (uint64_t)c0 * n_;
        // One copy in flight at a time: the runtime picks the SDMA engine of a copy when it is ENQUEUED, and with the first engine
        // still busy it takes another one - copies of one stream hopping between engines ran at 27 - 37 GB/s instead of 56 for whole
        // proofs (profiles/r03_pinned_upload.txt: config #4 38 ms instead of 27).  The host has nothing else to do here.
        static const bool free_running = std::getenv("SP_UPLOAD_FREE_RUNNING") != nullptr;
        if (g >= 1 && !free_running) SP_HIP_CHECK(hipEventSynchronize(up_ev_[g - 1].dma1));
        SP_HIP_CHECK(hipEventRecord(up_ev_[g].dma0, copy_stream_));
        auto h2d = [&](void* to, const uint8_t* from, size_t bytes) -> int {
            if (pull) return pull_copy(copy_stream_, from, to, bytes);
            SP_HIP_CHECK(hipMemcpyAsync(to, from, bytes, hipMemcpyHostToDevice, copy_stream_));
            return SP_OK;
        };
        if (col_stride == n_) {
            SP_TRY(h2d(dst, cols_host + (size_t)c0 * n_ * 32, (size_t)w * n_ * 32));
        } else {
            for (uint32_t j = 0; j < w; ++j) SP_TRY(h2d(dst + (uint64_t)j * n_, cols_host + (size_t)(c0 + j) * col_stride * 32, (size_t)n_ * 32));
        }
        SP_HIP_CHECK(hipEventRecord(up_ev_[g].dma1, copy_stream_));
        SP_HIP_CHECK(hipEventRecord(up_ev_[g].ready, copy_stream_));
        SP_HIP_CHECK(hipStreamWaitEvent(c_->stream, up_ev_[g].ready, 0));
        // element-wise, in place, on the compute stream in front of the group's transforms (kernels on the highest-priority copy stream
        // slow the transforms running beside them far more than their own run time: tools/upload_interference.py)
        if (col_enc >= 0) SP_TRY(decode_elements(c_->stream, col_enc, reinterpret_cast<const uint8_t*>(dst), (uint64_t)w * n_, dst));
        if (!sharded_interp) {
            // interpolate_fft + evaluate_offset_fft of this group (reference trace.rs:104-110, prover.rs:161-185)
            SP_TRY(c_->ntt->dif_natural_to_bitrev_inverse(coeffs + (uint64_t)c0 * n_, (int)logn_, w, n_, d_t1_, dst));
            SP_TRY(c_->ntt->lde_coset_major(coeffs + (uint64_t)c0 * n_, lde + (uint64_t)c0 * Nl_, (int)logn_, (int)logb_, w, n_, Nl_, (int)logG_, (int)rank_));
            if (segment == 0) SP_TRY(maybe_leaf_head(cols, c0 + w, lde));
        }
        SP_HIP_CHECK(hipEventRecord(up_ev_[g].done, c_->stream));
    }
    const double host_ms = wall_ms() - t0;
    if (segment == 0) SP_TRY(launch_aux_presort());   // every column is behind this point of the compute stream
    int rc;
    if (sharded_interp) rc = commit_segment_resident(segment, cols, root_out);
    else {
        rc = commit_columns(lde, Nl_, cols, segment == 0 ? tree_main_ : tree_aux_, root_out);
        if (rc == SP_OK) stage_ = segment == 0 ? 2 : 3;
    }
    if (rc == SP_OK) {
        SP_TRY(finish_upload_stats(groups, (uint64_t)cols * n_ * 32, 0.0, host_ms, pinned ? 2 : 3));
    }
    return rc;
}

// ---- NUMA placement of the gather (two-socket hosts: the table of the caller lives on the node its threads ran on, the page-locked
// ring always on the GPU's node, so on half of the boxes every byte crosses the socket link inside the gather)
static std::vector<int> numa_node_cpus(int node) {
    std::vector<int> cpus;
    char path[96];
    std::snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
    FILE* f = std::fopen(path, "r");
    if (!f) return cpus;
    char buf[4096] = {0};
    const bool ok = std::fgets(buf, sizeof(buf), f) != nullptr;
    std::fclose(f);
    if (!ok) return cpus;
    for (char* p = buf; *p;) {   // "0-63,128-191"
        char* e;
        const long a = std::strtol(p, &e, 10);
        if (e == p) break;
        long b = a;
        if (*e == '-') { p = e + 1; b = std::strtol(p, &e, 10); }
        for (long c = a; c <= b && c < 4096; ++c) cpus.push_back((int)c);
        p = (*e == ',') ? e + 1 : e;
        if (*e != ',' ) break;
    }
    return cpus;
}
// the node that holds most of a few sampled pages of [p, p + bytes): move_pages(2) with nodes = NULL only queries; -1 when unknown
static int numa_node_of_memory(const void* p, size_t bytes) {
    constexpr int SAMPLES = 16;
    void* pages[SAMPLES];
    int status[SAMPLES];
    const size_t step = std::max<size_t>(4096, (bytes / SAMPLES) & ~(size_t)4095);
    int count = 0;
    for (int i = 0; i < SAMPLES && (size_t)i * step < bytes; ++i) pages[count++] = reinterpret_cast<void*>((reinterpret_cast<uintptr_t>(p) & ~(uintptr_t)4095) + (size_t)i * step);
    if (count == 0 || syscall(SYS_move_pages, 0, (unsigned long)count, pages, nullptr, status, 0) != 0) return -1;
    int votes[8] = {0};
    for (int i = 0; i < count; ++i) if (status[i] >= 0 && status[i] < 8) ++votes[status[i]];
    int best = -1, bv = 0;
    for (int n = 0; n < 8; ++n) if (votes[n] > bv) { bv = votes[n]; best = n; }
    return best;
}
static int numa_node_of_device(int device) {
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, sizeof(bus), device) != hipSuccess) { (void)hipGetLastError(); return -1; }
    for (char* c = bus; *c; ++c) *c = (char)std::tolower((unsigned char)*c);
    char path[160];
    std::snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bus);
    FILE* f = std::fopen(path, "r");
    if (!f) return -1;
    int node = -1;
    if (std::fscanf(f, "%d", &node) != 1) node = -1;
    std::fclose(f);
    return node;
}

// sp_host_bind_to_device (include/stark252_hip.h)
int host_bind_calling_thread_to_device_node(int device, int* node_out) {
    const int node = numa_node_of_device(device);
    if (node_out) *node_out = -1;
    if (node < 0) return SP_OK;
    cpu_set_t allowed, want;
    CPU_ZERO(&allowed); CPU_ZERO(&want);
    if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return SP_OK;
    int count = 0;
    for (int c : numa_node_cpus(node)) if (c < CPU_SETSIZE && CPU_ISSET(c, &allowed)) { CPU_SET(c, &want); ++count; }
    if (count == 0 || sched_setaffinity(0, sizeof(want), &want) != 0) return SP_OK;
    if (node_out) *node_out = node;
    return SP_OK;
}

// A few parked host threads for the column gathers of the upload pipeline (creating them per group would put ~5 ms of
// pthread_create on the critical path of a proof).
class HostPool {
  public:
    explicit HostPool(unsigned workers) {
        for (unsigned w = 0; w < workers; ++w) threads_.emplace_back([this, w] { loop(w); });
    }
    ~HostPool() {
        { std::lock_guard<std::mutex> lk(m_); stop_ = true; gen_.fetch_add(1); }
        cv_.notify_all();
        for (auto& t : threads_) t.join();
    }
    unsigned size() const { return (unsigned)threads_.size() + 1; }
    // restricts the workers to a set of CPUs (the NUMA node of the upload's source or of the GPU); an empty set lifts the restriction
    void bind(const std::vector<int>& cpus) {
        cpu_set_t set;
        CPU_ZERO(&set);
        if (cpus.empty()) { for (int c = 0; c < CPU_SETSIZE; ++c) CPU_SET(c, &set); }
        else for (int c : cpus) if (c >= 0 && c < CPU_SETSIZE) CPU_SET(c, &set);
        for (auto& t : threads_) (void)pthread_setaffinity_np(t.native_handle(), sizeof(set), &set);
    }
    // Between begin_burst() and end_burst() idle workers spin on the generation counter instead of sleeping on the condition
    // variable: the groups of one upload follow each other within a millisecond and a futex wake-up of 30-60 threads costs
    // 50-100 us each time.
    void begin_burst() { { std::lock_guard<std::mutex> lk(m_); burst_.store(true, std::memory_order_release); } cv_.notify_all(); }
    void end_burst() { burst_.store(false, std::memory_order_release); }
    // runs job(part, parts) for part = 0 .. parts-1 (parts = workers + 1; the caller takes part 0) and waits for all of them
    void run(const std::function<void(unsigned, unsigned)>& job) {
        {
            std::lock_guard<std::mutex> lk(m_);
            job_ = &job; pending_.store((unsigned)threads_.size(), std::memory_order_relaxed);
            gen_.fetch_add(1, std::memory_order_release);
        }
        cv_.notify_all();
        job(0, size());
        // the parts are equal: the others finish within microseconds of the caller
        for (int spin = 0; pending_.load(std::memory_order_acquire) != 0; ++spin) {
            if (spin < 20000) { sp_cpu_relax(); continue; }
            std::unique_lock<std::mutex> lk(m_);
            done_.wait(lk, [this] { return pending_.load(std::memory_order_acquire) == 0; });
        }
        job_ = nullptr;
    }
  private:
    static void sp_cpu_relax() { __builtin_ia32_pause(); }
    void loop(unsigned w) {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(unsigned, unsigned)>* job;
            // (bounded: a worker that finds nothing for ~0.3 ms goes back to sleep, so a stalled upload does not burn the cores)
            for (int spin = 0; spin < 100000 && burst_.load(std::memory_order_acquire) && gen_.load(std::memory_order_acquire) == seen; ++spin) sp_cpu_relax();
            {
                std::unique_lock<std::mutex> lk(m_);
                if (gen_.load(std::memory_order_acquire) == seen) {
                    const bool was_burst = burst_.load(std::memory_order_acquire);
                    cv_.wait(lk, [&] { return gen_.load(std::memory_order_acquire) != seen || (!was_burst && burst_.load(std::memory_order_acquire)); });
                    if (gen_.load(std::memory_order_acquire) == seen) continue;   // woken into a burst: go spinning
                }
                seen = gen_.load(std::memory_order_acquire);
                if (stop_) return;
                job = job_;
            }
            (*job)(w + 1, size());
            if (pending_.fetch_sub(1, std::memory_order_acq_rel) == 1) { std::lock_guard<std::mutex> lk(m_); done_.notify_one(); }
        }
    }
    std::vector<std::thread> threads_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    const std::function<void(unsigned, unsigned)>* job_ = nullptr;
    std::atomic<unsigned> pending_{0};
    std::atomic<uint64_t> gen_{0};
    std::atomic<bool> burst_{false};
    bool stop_ = false;
};
void host_pool_delete(HostPool* p) { delete p; }

// One block of a chunk: rows [r0, r1) of the chunk (src points at the chunk's first row), columns [off/32, off/32 + cw) of the table,
// written column-major into the chunk's ring slot (column j at dst + j * n * 32: every column of a chunk is one contiguous DMA
// straight into the trace area).
// The reads are one or two cache lines out of every row (a stride of cols x 32 bytes: beyond what the hardware prefetchers follow),
// so the rows PF_ROWS ahead are requested by hand - 16 -> 31-34 GB/s on eight threads of the build container; two rows at a time,
// so that every store run is one whole 64-byte line (+5-10 %; gather microbenchmark, tools/experiments/README.md).
static void host_gather_block(const uint8_t* src, uint64_t n, size_t row_bytes, size_t off, uint32_t cw, uint8_t* dst, uint64_t r0, uint64_t r1) {
    constexpr size_t PF_ROWS = 24;
    const uint8_t* s = src + r0 * row_bytes + off;
    uint64_t i = r0;
    for (; i + 2 <= r1; i += 2, s += 2 * row_bytes) {
        for (uint32_t l = 0; l < cw * 32; l += 64) {
            __builtin_prefetch(s + PF_ROWS * row_bytes + l, 0, 0);
            __builtin_prefetch(s + (PF_ROWS + 1) * row_bytes + l, 0, 0);
        }
        for (uint32_t u = 0; u < cw; ++u) {
            uint8_t* d = dst + ((size_t)u * n + i) * 32;
            __builtin_memcpy(d, s + 32 * (size_t)u, 32);              // (fixed-size copies inline as vector moves)
            __builtin_memcpy(d + 32, s + row_bytes + 32 * (size_t)u, 32);
        }
    }
    for (; i < r1; ++i, s += row_bytes)
        for (uint32_t u = 0; u < cw; ++u) __builtin_memcpy(dst + ((size_t)u * n + i) * 32, s + 32 * (size_t)u, 32);
}

// The same block for columns that hold only 0 and 1 (the instruction flags of a Cairo trace): one BIT per cell instead of 32 bytes -
// column u's bitmap at dst + u * (n / 8), row i of the chunk at bit i (little-endian 64-bit words; r0 and r1 are multiples of 64).
// A cell that is neither `zero` nor `one` (32-byte images in the table's encoding) sets *dirty: the caller uploads the table again in full.
static void host_pack_bits_block(const uint8_t* src, uint64_t n, size_t row_bytes, size_t off, uint32_t cw, uint8_t* dst, uint64_t r0, uint64_t r1,
                                 const uint64_t zero[4], const uint64_t one[4], std::atomic<bool>& dirty) {
    constexpr size_t PF_ROWS = 24;
    bool bad = false;
    for (uint64_t i = r0; i < r1; i += 64) {
        uint64_t words[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const uint64_t lim = std::min<uint64_t>(64, r1 - i);
        const uint8_t* s = src + i * row_bytes + off;
        for (uint64_t k = 0; k < lim; ++k, s += row_bytes) {
            for (uint32_t l = 0; l < cw * 32; l += 64) __builtin_prefetch(s + PF_ROWS * row_bytes + l, 0, 0);
            for (uint32_t u = 0; u < cw; ++u) {
                uint64_t v[4];
                __builtin_memcpy(v, s + 32 * (size_t)u, 32);
                const bool is_one = ((v[0] ^ one[0]) | (v[1] ^ one[1]) | (v[2] ^ one[2]) | (v[3] ^ one[3])) == 0;
                const bool is_zero = ((v[0] ^ zero[0]) | (v[1] ^ zero[1]) | (v[2] ^ zero[2]) | (v[3] ^ zero[3])) == 0;
                bad |= !(is_one | is_zero);
                words[u] |= (uint64_t)is_one << k;
            }
        }
        for (uint32_t u = 0; u < cw; ++u) __builtin_memcpy(dst + (size_t)u * (n / 8) + (i / 64) * 8, &words[u], 8);
    }
    if (bad) dirty.store(true, std::memory_order_relaxed);
}

static size_t upload_chunk_bytes() {
    static const size_t chunk_mb = [] { const char* e = std::getenv("SP_UPLOAD_CHUNK_MB"); return e ? (size_t)std::min(64, std::max(1, std::atoi(e))) : (size_t)32; }();
    return chunk_mb << 20;
}

int StarkProver::ensure_ring_and_pool() {
    // gather threads: the option, but at most twice the CPUs this process can really have.  A cgroup quota counts CPU time per
    // 100 ms period and the upload is a burst of a quarter of a proof, so twice the quota's CPUs for that long stays inside it
    // (24 threads move 55-80 GB/s where the quota's own 14 move 40); far beyond it the whole process gets throttled - the 130 ms
    // proofs of a 64-thread gather in a 16-CPU container.
    // Per HOST, not per context: with R ranks on the host (SP_OPT_HOST_RANKS / LOCAL_WORLD_SIZE) each pool gets the R-th part of that -
    // eight ranks of a 16-CPU container used to start 8 x 23 gather threads.
    const unsigned want = std::max(2u, std::min(c_->opt_upload_threads, 2 * host_cpu_budget()));
    if (pool_ && pool_->size() != want) { delete pool_; pool_ = nullptr; }      // (the option changed between two proofs)
    if (!pool_) pool_ = new HostPool(want - 1u);
    const size_t chunk = upload_chunk_bytes();
    if (stage_bytes_ < chunk) {
        for (auto& p : h_stage_) { if (p) (void)hipHostFree(p); p = nullptr; }
        stage_bytes_ = 0;
        for (auto& p : h_stage_) if (hipHostMalloc(&p, chunk, hipHostMallocDefault) != hipSuccess) { sp_set_error("commit_trace: pinned staging allocation failed"); return SP_E_ALLOC; }
        stage_bytes_ = chunk;
    }
    return SP_OK;
}

// interpolate_and_commit (reference prover.rs:126-159) from a row-major HOST buffer (the reference's TraceTable, trace.rs:9-13),
// in column groups: while group g is interpolated and extended on the compute stream, the chunks of the groups behind it are
// transposed out of the table into a small ring of page-locked slots by a few host threads (256 KB blocks off a shared counter: a
// thread that is slow - a busy core, a throttled container - takes fewer blocks) and cross PCIe on a second stream, column by column
// straight into the trace area.  A chunk is a block of rows of one group, 32 MB at most: the DMA of one chunk runs beside the gather
// of the next whatever the size of the group, and the ring (4 x 32 MB) takes a sixth of the time to pin that three group-sized slots
// did (56 ms of a first proof at 2^20 rows).
// window_only: several ranks - only the columns [c_begin, c_begin + c_count) of the table (the block this rank's role contributes to
// the all-gather of the trace, commit_trace_rows_sharded) are gathered, uploaded and decoded; no transforms, no commitment.
// binary_cols: columns [0, binary_cols) of the table hold only 0 and 1 in a valid trace (StarkProver::hint_binary_columns): groups
// that lie inside them cross PCIe as bitmaps and are expanded on the device; returns SP_RETRY_RAW_UPLOAD when a cell breaks the hint.
int StarkProver::commit_trace_pipelined(int segment, const uint8_t* rows_host, uint32_t table_cols, uint8_t root_out[32], uint32_t c_begin, uint32_t c_count,
                                        bool window_only, uint32_t binary_cols) {
    const uint32_t col0 = segment == 0 ? 0 : Cm_;
    const uint32_t cols = window_only ? c_count : table_cols;
    if (!window_only) { c_begin = 0; c_count = table_cols; }
    // Column groups.  Group g can be transformed once it has crossed PCIe (~0.6 ms per column of 2^20 rows) and everything behind
    // it still has to be transformed (~0.7 ms per column at blowup 8, 0.2 ms at blowup 4, where the upload is the bound): two single
    // columns start the pipeline, then pairs, always from an even column on (two columns share a 64-byte line of a row).  A group is
    // usable when all of it has landed, so narrow groups keep the compute stream fed; and since the gather writes a chunk column by
    // column (host_gather_block) wider groups no longer stream better either - four-column groups: 30.1 ms at config #4, pairs 29.3.
    static const uint32_t maxw_env = [] { const char* e = std::getenv("SP_UPLOAD_MAXW"); return e ? (uint32_t)std::min(8, std::max(2, std::atoi(e))) : 0u; }();
    const uint32_t maxw = maxw_env ? maxw_env : 2u;
    // Columns that cross as bitmaps go eight at a time: the threads read 256 contiguous bytes of every row per pass instead of one
    // 64-byte line (tools/experiments/gather_bench.cpp on an MI355X host, 24 threads, 16 flag columns of 2^19 rows: 3.5 ms as eight
    // pair-wide passes, 1.8 ms as two 8-wide ones, 1.6 ms as one) and the link carries next to nothing for them either way.  In the
    // proof (tools/rows_tail.py, config #4 / #3): gather 8.0 -> 5.9 ms / 21.9 -> 13.8 ms, host CPU time 256 -> 209 / 656 -> 484 ms,
    // the proof itself within the box spread (24.7 - 25.3 ms either way: the transforms are the bound by then).  Sixteen at a time
    // would put 1.6 ms of gathering in front of the compute stream's first cheap columns.  SP_UPLOAD_PACKW=1..8 overrides.
    static const uint32_t packw = [] { const char* e = std::getenv("SP_UPLOAD_PACKW"); return e ? (uint32_t)std::min(8, std::max(1, std::atoi(e))) : 8u; }();
    struct Group { uint32_t c, w; };
    std::vector<Group> packed_g, wide_g;
    {
        const uint32_t c_end = c_begin + cols, p_end = std::min(c_end, std::max(c_begin, binary_cols));
        for (uint32_t c = c_begin; c < p_end;) { const uint32_t w = std::min(packw, p_end - c); packed_g.push_back(Group{c, w}); c += w; }
        for (uint32_t done = 0, left = c_end - p_end; done < left;) {
            uint32_t w = done < 2 ? 1u : std::min<uint32_t>(maxw, std::max<uint32_t>(2, 2 * ((done + 1) / 2)));
            if (left - done <= w + 1) w = left - done;     // no one-column tail
            wide_g.push_back(Group{p_end + done, w});
            done += w;
        }
    }
    const uint32_t groups = (uint32_t)(packed_g.size() + wide_g.size());
    // Upload order.  A bitmap group costs the gather threads their share of the table's bytes and the link nothing; a group of
    // full-width columns costs the link more than the threads (0.6 ms of DMA against 0.5 ms of gathering per pair of 2^19 rows), and
    // the transforms of a column take less than its DMA at blowup 4.  All bitmaps first leave the link idle while they are gathered and
    // make it the bound afterwards; all last starve the compute stream, which gets a column every 0.3 ms and needs 0.25.  Taken in
    // turns - a wide group, a bitmap group, ... - the DMA of a wide group runs beside the gather of the next bitmap group, and the
    // eight cheap columns of a bitmap group keep the compute stream busy while the next wide ones cross.  SP_UPLOAD_ORDER=seq: table order.
    std::vector<Group> order;
    {
        static const bool seq = [] { const char* e = std::getenv("SP_UPLOAD_ORDER"); return e && e[0] == 's'; }();
        size_t ip = 0, iw = 0;
        if (seq || window_only || packed_g.empty() || wide_g.empty()) { order = packed_g; order.insert(order.end(), wide_g.begin(), wide_g.end()); }
        else
            while (ip < packed_g.size() || iw < wide_g.size()) {
                if (iw < wide_g.size()) order.push_back(wide_g[iw++]);
                if (ip < packed_g.size()) order.push_back(packed_g[ip++]);
            }
    }
    std::vector<uint8_t> extended(cols, 0);      // columns whose transforms are queued (the leaf head wants a whole prefix of them)
    uint32_t extended_prefix = 0;
    const size_t chunk = upload_chunk_bytes();   // bytes per ring slot
    double _tp = wall_ms();
    sp_ctx* ctx = c_;
    SP_TRY(ensure_ring_and_pool());
    if (!pool_bound_) {
        // Where the workers run.  The ring is page-locked memory and lives on the GPU's NUMA node; the caller's table lives where
        // the caller's threads ran.  On the two-socket hosts of the pool half of the boxes have the two on different nodes, and a
        // gather by threads of the table's node (where the scheduler keeps the process) then moves 35 - 40 GB/s - remote stores -
        // where threads of the GPU's node move 51 (remote loads, which the software prefetch covers; local stores): 80 against 76.5 ms at
        // config #3 (tools/rows_path_rounds.py ... far).  With both on one node the binding changes nothing (52 - 56 GB/s).
        // SP_UPLOAD_BIND=none|table|gpu overrides (default gpu).
        pool_bound_ = true;
        const char* bind_env = std::getenv("SP_UPLOAD_BIND");
        const char mode = bind_env ? bind_env[0] : 'g';
        const int node_table = numa_node_of_memory(rows_host, (size_t)n_ * table_cols * 32), node_gpu = numa_node_of_device(c_->device);
        const int node = mode == 't' ? node_table : (mode == 'g' ? node_gpu : -1);
        std::vector<int> cpus;
        if (node >= 0) {
            cpu_set_t allowed;
            CPU_ZERO(&allowed);
            if (sched_getaffinity(0, sizeof(allowed), &allowed) == 0)
                for (int c : numa_node_cpus(node)) if (c < CPU_SETSIZE && CPU_ISSET(c, &allowed)) cpus.push_back(c);
            if (cpus.size() < pool_->size()) cpus.clear();   // (a cpuset that leaves the node too few CPUs: do not squeeze the workers onto them)
        }
        if (timing_enabled()) std::fprintf(stderr, "[sp_timing]   upload: table on NUMA node %d, GPU on node %d, %zu CPUs for the workers%s\n", node_table, node_gpu,
                                           cpus.size(), cpus.empty() ? " (unbound)" : "");
        if (!cpus.empty()) pool_->bind(cpus);
    }
    SP_TRY(ensure_upload(groups));
    fe* coeffs = d_coeffs_ + (uint64_t)col0 * n_;
    fe* trace = d_trace_ + (uint64_t)col0 * n_;
    fe* lde = d_lde_ + (uint64_t)col0 * Nl_;
    SP_TIMEPOINT("  upload: threads, streams, pinned slots");
    // The chunks of the whole segment, in upload order: a block of rows of one column group, one ring slot (32 MB) at most; the
    // gather reads w x 32 contiguous bytes of every row (the wider the group, the better it streams).
    struct Chunk { uint32_t g, c, cw, slot; uint64_t r0, rows, first_block, blocks; bool last_of_group, packed; };
    if (binary_cols) {
        const uint64_t words = (uint64_t)binary_cols * (n_ >> 6);
        if (flagbits_words_ < words) { SP_TRY(alloc((void**)&d_flagbits_, words * 8)); flagbits_words_ = words; }
    }
    uint64_t zero_img[4] = {0, 0, 0, 0}, one_img[4] = {0, 0, 0, 0};      // 0 and 1 as the table encodes them
    {
        const fe one = fe_one();
        if (c_->enc == SP_FE_CANON_BE) { uint8_t b[32]; fe_to_bytes_be(one, b); std::memcpy(one_img, b, 32); }
        else fe_to_lw_limbs(one, one_img);
    }
    std::atomic<bool> dirty{false};
    uint64_t dma_bytes = 0;
    std::vector<Chunk> chunks;
    uint64_t n_blocks = 0;
    {
        for (uint32_t g = 0; g < groups; ++g) {
            const uint32_t c = order[g].c, cw = order[g].w;
            // (the two single columns that start the pipeline in quarter-size chunks: the first DMA leaves after 0.16 ms of gathering
            // instead of 0.65, and the compute stream gets its first column that much earlier)
            const size_t chunk_g = g < 2 && cw == 1 ? chunk / 4 : chunk;
            const uint64_t rows_per_chunk = std::max<uint64_t>(256, (chunk_g / ((size_t)cw * 32)) & ~(uint64_t)255);
            const uint64_t block_rows = std::max<uint64_t>(256, (256u << 10) / ((size_t)cw * 32)) & ~(uint64_t)63;   // ~256 KB written per block
            for (uint64_t r0 = 0; r0 < n_; r0 += rows_per_chunk) {
                const uint64_t rows = std::min<uint64_t>(rows_per_chunk, n_ - r0);
                const uint64_t blocks = (rows + block_rows - 1) / block_rows;
                const bool packed = c + cw <= binary_cols && c >= c_begin;
                dma_bytes += packed ? (uint64_t)cw * rows / 8 : (uint64_t)cw * rows * 32;
                chunks.push_back(Chunk{g, c, cw, (uint32_t)(chunks.size() % UPLOAD_SLOTS), r0, rows, n_blocks, blocks, r0 + rows >= n_, packed});
                n_blocks += blocks;
            }
        }
    }
    const size_t n_chunks = chunks.size();
    std::vector<std::atomic<uint32_t>> blocks_done(n_chunks);
    for (auto& b : blocks_done) b.store(0, std::memory_order_relaxed);
    std::atomic<uint64_t> next_block{0};
    std::atomic<uint64_t> writable{UPLOAD_SLOTS};     // chunks [0, writable) may be gathered: the slot of chunk k is free once chunk k - SLOTS has crossed PCIe
    std::atomic<bool> abort_upload{false};
    const double t0 = wall_ms();
    double gather_end = t0;
    int rc_upload = SP_OK;
    struct Burst { HostPool* p; ~Burst() { p->end_burst(); } } burst{pool_};   // (also on the error paths)
    pool_->begin_burst();
    SP_HIP_CHECK(hipEventRecord(up_start_, c_->stream));             // the trace area's previous users are behind this point
    SP_HIP_CHECK(hipStreamWaitEvent(copy_stream_, up_start_, 0));
    SP_HIP_CHECK(hipEventRecord(up_ev_[0].dma0, copy_stream_));
    // ONE job for the whole upload, no barrier per chunk: the workers take 256 KB blocks off a shared counter across chunk
    // boundaries (up to the ring's four slots ahead of the DMA), so a thread that loses its core for a scheduler quantum delays the
    // one chunk its block belongs to instead of stopping everybody at the end of every chunk - on the shared hosts of the test
    // boxes a barrier every 0.7 ms met a descheduled thread most of the time.  The calling thread does not gather: it enqueues
    // the DMAs of a chunk when its last block is in, the group's transforms behind the last chunk of a group, and hands slots
    // back when their DMA has completed.
    const bool yield_idle = host_cpu_budget() < 2;   // this thread shares its core with a gather worker (or another rank): give it up when idle
    auto orchestrate = [&]() -> int {
        size_t k = 0, d = 0;   // next chunk to send, next chunk whose DMA completion is awaited
        uint64_t idle_spins = 0;
        double idle_since = 0.0;
        while (k < n_chunks) {
            if (dirty.load(std::memory_order_acquire)) return SP_RETRY_RAW_UPLOAD;   // a cell broke the 0 / 1 hint: stop, the caller uploads in full
            bool progress = false;
            if (d < k) {
                const hipError_t q = hipEventQuery(ev_dma_[chunks[d].slot]);
                if (q == hipSuccess) {
                    ++d;
                    writable.store(d + UPLOAD_SLOTS, std::memory_order_release);
                    progress = true;
                } else if (q != hipErrorNotReady) {   // a failed copy must end the loop, not spin in it
                    sp_set_error(std::string("commit_trace: upload DMA failed: ") + hipGetErrorString(q));
                    return SP_E_HIP;
                }
            }
            const Chunk& ck = chunks[k];
            if (blocks_done[k].load(std::memory_order_acquire) == ck.blocks) {
                uint8_t* slot = static_cast<uint8_t*>(h_stage_[ck.slot]);
                if (ck.packed) {
                    for (uint32_t j = 0; j < ck.cw; ++j)   // a column's bits of this chunk: rows / 8 bytes
                        SP_HIP_CHECK(hipMemcpyAsync(d_flagbits_ + ((uint64_t)(ck.c + j) * n_ + ck.r0) / 64, slot + (size_t)j * (ck.rows / 8), (size_t)ck.rows / 8,
                                                    hipMemcpyHostToDevice, copy_stream_));
                } else
                for (uint32_t j = 0; j < ck.cw; ++j)   // column by column, straight into the trace area (host encoding; decoded in place below)
                    SP_HIP_CHECK(hipMemcpyAsync(trace + (uint64_t)(ck.c + j) * n_ + ck.r0, slot + (size_t)j * ck.rows * 32, (size_t)ck.rows * 32, hipMemcpyHostToDevice, copy_stream_));
                SP_HIP_CHECK(hipEventRecord(ev_dma_[ck.slot], copy_stream_));
                if (ck.last_of_group) {
                    const uint32_t g = ck.g, w = ck.cw, gc0 = ck.c;
                    SP_HIP_CHECK(hipEventRecord(up_ev_[g].dma1, copy_stream_));
                    SP_HIP_CHECK(hipEventRecord(up_ev_[g].ready, copy_stream_));
                    if (g + 1 < groups) SP_HIP_CHECK(hipEventRecord(up_ev_[g + 1].dma0, copy_stream_));
                    SP_HIP_CHECK(hipStreamWaitEvent(c_->stream, up_ev_[g].ready, 0));
                    // host encoding -> device layout, in place, in front of the group's transforms on the compute stream (an ordinary
                    // kernel of the proof: the rows -> columns kernels this replaces ran on a highest-priority stream beside the
                    // transforms and made those 1.3 - 2.2 x slower for as long as the upload lasted - tools/upload_interference.py)
                    if (ck.packed) SP_TRY(expand_bit_columns(c_->stream, d_flagbits_ + (uint64_t)gc0 * (n_ >> 6), n_, w, trace + (uint64_t)gc0 * n_));
                    else
                    SP_TRY(decode_elements(c_->stream, c_->enc, reinterpret_cast<const uint8_t*>(trace + (uint64_t)gc0 * n_), (uint64_t)w * n_, trace + (uint64_t)gc0 * n_));
                    if (!window_only) {
                        // interpolate_fft + evaluate_offset_fft of this group (reference trace.rs:104-110, prover.rs:161-185)
                        SP_TRY(c_->ntt->dif_natural_to_bitrev_inverse(coeffs + (uint64_t)gc0 * n_, (int)logn_, w, n_, d_t1_, trace + (uint64_t)gc0 * n_));
                        SP_TRY(c_->ntt->lde_coset_major(coeffs + (uint64_t)gc0 * n_, lde + (uint64_t)gc0 * Nl_, (int)logn_, (int)logb_, w, n_, Nl_, 0, 0));
                        for (uint32_t j = 0; j < w; ++j) extended[gc0 - c_begin + j] = 1;
                        while (extended_prefix < cols && extended[extended_prefix]) ++extended_prefix;
                        if (segment == 0) SP_TRY(maybe_leaf_head(cols, extended_prefix, lde));
                    }
                    SP_HIP_CHECK(hipEventRecord(up_ev_[g].done, c_->stream));
                }
                if (++k == n_chunks) gather_end = wall_ms();
                progress = true;
            }
            if (progress) { idle_since = 0.0; continue; }
            if (yield_idle) std::this_thread::yield(); else __builtin_ia32_pause();
            if ((++idle_spins & 0xfffffu) == 0) {   // (~every few ms) a watchdog: neither a block gathered nor a DMA completed for 60 s
                const double now = wall_ms();
                if (idle_since == 0.0) idle_since = now;
                else if (now - idle_since > 60e3) { sp_set_error("commit_trace: the upload pipeline made no progress for 60 s"); return SP_E_HIP; }
            }
        }
        return SP_OK;
    };
    pool_->run([&](unsigned part, unsigned) {
        if (part == 0) {
            rc_upload = orchestrate();
            if (rc_upload != SP_OK) abort_upload.store(true, std::memory_order_release);
            return;
        }
        size_t k = 0;
        for (;;) {
            const uint64_t b = next_block.fetch_add(1, std::memory_order_relaxed);
            if (b >= n_blocks || dirty.load(std::memory_order_relaxed)) return;
            while (b >= chunks[k].first_block + chunks[k].blocks) ++k;
            const Chunk& ck = chunks[k];
            for (unsigned spin = 0; k >= writable.load(std::memory_order_acquire); ++spin) {   // the slot is still crossing PCIe
                if (abort_upload.load(std::memory_order_acquire)) return;
                if (spin < 2000) __builtin_ia32_pause(); else std::this_thread::yield();
            }
            const uint64_t block_rows = std::max<uint64_t>(256, (256u << 10) / ((size_t)ck.cw * 32)) & ~(uint64_t)63;
            const uint64_t br0 = (b - ck.first_block) * block_rows;
            if (ck.packed)
                host_pack_bits_block(rows_host + ck.r0 * (size_t)table_cols * 32, ck.rows, (size_t)table_cols * 32, (size_t)ck.c * 32, ck.cw,
                                     static_cast<uint8_t*>(h_stage_[ck.slot]), br0, std::min<uint64_t>(ck.rows, br0 + block_rows), zero_img, one_img, dirty);
            else
            host_gather_block(rows_host + ck.r0 * (size_t)table_cols * 32, ck.rows, (size_t)table_cols * 32, (size_t)ck.c * 32, ck.cw, static_cast<uint8_t*>(h_stage_[ck.slot]),
                              br0, std::min<uint64_t>(ck.rows, br0 + block_rows));
            blocks_done[k].fetch_add(1, std::memory_order_acq_rel);
        }
    });
    if (rc_upload != SP_OK) return rc_upload;     // (SP_RETRY_RAW_UPLOAD included: everything queued so far is overwritten by the full upload that follows)
    if (dirty.load(std::memory_order_acquire)) return SP_RETRY_RAW_UPLOAD;
    const double gather_ms = gather_end - t0;
    const double host_ms = wall_ms() - t0;
    SP_TIMEPOINT("  upload + transforms of the groups");
    if (window_only) {   // (the caller finishes the statistics once the compute stream has been waited for)
        pending_up_groups_ = groups; pending_up_bytes_ = dma_bytes; pending_up_gather_ms_ = gather_ms; pending_up_host_ms_ = host_ms;
        return SP_OK;
    }
    if (segment == 0) SP_TRY(launch_aux_presort());   // every group has been turned into columns behind this point of the compute stream
    SP_TIMEPOINT("  aux presort queued (+ its workspace)");
    SP_TRY(commit_columns(lde, Nl_, cols, segment == 0 ? tree_main_ : tree_aux_, root_out));
    SP_TIMEPOINT("  leaf hashing + tree");
    stage_ = segment == 0 ? 2 : 3;
    return finish_upload_stats(groups, dma_bytes, gather_ms, host_ms, 1);
}

}  // namespace sp
