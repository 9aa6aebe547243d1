// See cairo_host.h. Host-only C++ (compiled by hipcc as plain host code).
#include "cairo_host.h"
#include <algorithm>
#include <cstring>
#include <stdexcept>
#include <thread>
#include <atomic>
#include <sched.h>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <exception>
#include <new>
#include <memory>
#include <mutex>

namespace sp {

static inline uint64_t fe_low_u64(const fe& a) {
    fe r = fe_from_mont(a);
    return (uint64_t)r.v[0] | ((uint64_t)r.v[1] << 32);
}
// field element -> signed 64-bit offset (x or -(p - x)); throws when neither fits
static inline int64_t fe_to_i64(const fe& a) {
    fe r = fe_from_mont(a);
    bool small = true;
    for (int i = 2; i < 8; ++i) if (r.v[i]) small = false;
    if (small) return (int64_t)((uint64_t)r.v[0] | ((uint64_t)r.v[1] << 32));
    fe n = fe_from_mont(fe_neg(a));
    for (int i = 2; i < 8; ++i) if (n.v[i]) throw std::runtime_error("offset does not fit in 64 bits");
    return -(int64_t)((uint64_t)n.v[0] | ((uint64_t)n.v[1] << 32));
}

bool parse_trace_le(const uint8_t* b, size_t len, std::vector<RegisterState>& out) {
    if (len % 24 != 0) return false;
    out.resize(len / 24);
    for (size_t i = 0; i < out.size(); ++i) {
        uint64_t v[3];
        std::memcpy(v, b + 24 * i, 24);  // little-endian host
        out[i] = RegisterState{v[0], v[1], v[2]};
    }
    return true;
}
bool parse_memory_le(const uint8_t* b, size_t len, CairoMemory& out) {
    if (len % 40 != 0) return false;
    for (size_t i = 0; i < len / 40; ++i) {
        uint64_t addr;
        std::memcpy(&addr, b + 40 * i, 8);
        uint8_t be[32];
        for (int k = 0; k < 32; ++k) be[k] = b[40 * i + 8 + 31 - k];  // value is 32 bytes little-endian
        out.set(addr, fe_from_bytes_be(be));
    }
    return true;
}

PublicInputs public_inputs_from_regs_and_mem(const std::vector<RegisterState>& regs, const CairoMemory& mem,
                                             size_t program_size, const std::vector<MemorySegment>& segments) {
    PublicInputs p;
    p.memory_segments = segments;
    for (uint64_t i = 1; i <= program_size; ++i) {
        const fe* v = mem.get(i);
        if (!v) throw std::runtime_error("program word missing from memory");
        p.public_memory.push_back({i, *v});
    }
    if (const MemorySegment* out = p.segment(1))
        for (uint64_t a = out->start; a < out->end; ++a) {
            const fe* v = mem.get(a);
            if (!v) throw std::runtime_error("output cell missing from memory");
            p.public_memory.push_back({a, *v});
        }
    const RegisterState& last = regs.back();
    p.pc_init = fe_from_u64(regs[0].pc); p.ap_init = fe_from_u64(regs[0].ap); p.fp_init = fe_from_u64(regs[0].fp);
    p.pc_final = fe_from_u64(last.pc); p.ap_final = fe_from_u64(last.ap);
    p.num_steps = regs.size();
    return p;
}

namespace {
struct Decoded {
    uint32_t flags;  // 15 bits
    uint32_t off_dst, off_op0, off_op1;  // biased 16-bit fields
    int dst_reg, op0_reg, op1_src, res_logic, pc_update, ap_update, opcode;
};
Decoded decode(const fe& word) {
    // an instruction is a 63-bit word (whitepaper section 4.5; the AIR's constraint 15 wants flag 15 = 0 and constraint 16 rebuilds the
    // cell from its 15 flags and three offsets): anything larger cannot be a row of a valid trace
    {
        const fe raw = fe_from_mont(word);
        for (int l = 2; l < 8; ++l) if (raw.v[l]) throw std::runtime_error("instruction cell beyond 64 bits");
    }
    uint64_t w = fe_low_u64(word);
    if (w >> 63) throw std::runtime_error("InstructionNonZeroHighBit");
    Decoded d;
    d.off_dst = (uint32_t)(w & 0xffff); d.off_op0 = (uint32_t)((w >> 16) & 0xffff); d.off_op1 = (uint32_t)((w >> 32) & 0xffff);
    uint32_t f = (uint32_t)(w >> 48);
    d.flags = f & 0x7fff;
    d.dst_reg = f & 1; d.op0_reg = (f >> 1) & 1; d.op1_src = (f >> 2) & 7; d.res_logic = (f >> 5) & 3;
    d.pc_update = (f >> 7) & 7; d.ap_update = (f >> 10) & 3; d.opcode = (f >> 12) & 7;
    auto one_hot_or_zero = [](int v) { return v == 0 || v == 1 || v == 2 || v == 4; };
    if (!one_hot_or_zero(d.op1_src)) throw std::runtime_error("InvalidOp1Src");
    if (d.res_logic == 3) throw std::runtime_error("InvalidResLogic");
    if (!one_hot_or_zero(d.pc_update)) throw std::runtime_error("InvalidPcUpdate");
    if (d.ap_update == 3) throw std::runtime_error("InvalidApUpdate");
    if (!one_hot_or_zero(d.opcode)) throw std::runtime_error("InvalidOpcode");
    // call: the frame goes to [ap] and [ap + 1] and ap moves by two - the encoding cairo-lang asserts (dst = [ap + 0], op0 = [ap + 1],
    // no other ap update); any other leaves the AIR's CALL_1 / CALL_2 / NEXT_AP constraints unsatisfied
    if (d.opcode == 1 && (d.dst_reg != 0 || d.op0_reg != 0 || d.off_dst != 0x8000 || d.off_op0 != 0x8001 || d.ap_update != 0))
        throw std::runtime_error("InvalidCallEncoding");
    return d;
}
inline uint64_t add_signed(uint64_t base, uint32_t biased_off) { return base + (uint64_t)biased_off - 0x8000ULL; }
const fe& mem_at(const CairoMemory& m, uint64_t a) {
    const fe* v = m.get(a);
    if (!v) throw std::runtime_error("memory cell not found");
    return *v;
}
}  // namespace

void TraceColumns::release() {
    if (data) { if (pinned) (void)hipHostFree(data); else std::free(data); }
    if (retired) std::free(retired);
    data = nullptr; retired = nullptr; n_rows = n_cols = 0; pinned = false;
}
static std::atomic<int> g_hip_in_use{0};
void hip_runtime_mark_in_use() { g_hip_in_use.store(1); }
bool hip_runtime_in_use() { return g_hip_in_use.load() != 0; }

static void* alloc_pinned(size_t bytes) {
    void* p = nullptr;
    const char* env = std::getenv("SP_HOST_PINNED");
    if (env && env[0] == '0') return nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) == hipSuccess && p) return p;
    (void)hipGetLastError();
    return nullptr;
}
void TraceColumns::allocate(size_t rows, size_t cols) {
    release();
    const size_t bytes = std::max<size_t>(rows * cols * sizeof(fe), 64);
    // page-locked when this process already drives a device (the upload of a column group is then a plain DMA); a process that
    // has not touched the GPU stays host-only here - try_pin() migrates the table when a prover first needs it
    void* p = hip_runtime_in_use() ? alloc_pinned(bytes) : nullptr;
    if (p) pinned = true;
    else if (posix_memalign(&p, 4096, bytes) != 0 || !p) throw std::bad_alloc();
    data = static_cast<fe*>(p); n_rows = rows; n_cols = cols;
}
bool TraceColumns::try_pin() {
    std::lock_guard<std::mutex> lk(pin_mutex);
    if (pinned || !data) return pinned;
    const size_t bytes = std::max<size_t>(n_rows * n_cols * sizeof(fe), 64);
    uint8_t* p = static_cast<uint8_t*>(alloc_pinned(bytes));
    if (!p) return false;
    const uint8_t* src = reinterpret_cast<const uint8_t*>(data);
    host_parallel_for(bytes, 1 << 22, [&](size_t b, size_t e) { std::memcpy(p + b, src + b, e - b); });
    retired = data;     // NOT freed: the old address was handed out and may be in use (released with the run)
    data = reinterpret_cast<fe*>(p); pinned = true;
    return true;
}

// CPUs this process may really use: the hardware threads, cut down by the affinity mask and by the cgroup CPU quota (a container
// on a 256-thread host is often limited to a few CPUs' worth of time per period; threads beyond the quota only get the whole
// group throttled - the 100 ms stalls of an oversubscribed upload).
unsigned host_effective_cpus() {
    static const unsigned cached = [] {
        unsigned n = std::max(1u, std::thread::hardware_concurrency());
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof(set), &set) == 0) n = std::min<unsigned>(n, (unsigned)std::max(1, CPU_COUNT(&set)));
        auto quota = [](const char* path_max, const char* path_quota, const char* path_period) -> double {
            if (FILE* f = std::fopen(path_max, "r")) {                      // cgroup v2: "max 100000" or "<quota> <period>"
                char q[64] = {0}; long long per = 0;
                const int got = std::fscanf(f, "%63s %lld", q, &per);
                std::fclose(f);
                if (got == 2 && per > 0 && q[0] != 'm') return std::atof(q) / (double)per;
                return 0.0;
            }
            long long qv = -1, pv = 0;                                      // cgroup v1
            if (FILE* f = std::fopen(path_quota, "r")) { if (std::fscanf(f, "%lld", &qv) != 1) qv = -1; std::fclose(f); }
            if (FILE* f = std::fopen(path_period, "r")) { if (std::fscanf(f, "%lld", &pv) != 1) pv = 0; std::fclose(f); }
            return (qv > 0 && pv > 0) ? (double)qv / (double)pv : 0.0;
        };
        const double q = quota("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us");
        if (q > 0) n = std::min<unsigned>(n, (unsigned)std::max(1.0, std::floor(q + 0.5)));
        return n;
    }();
    return cached;
}

static std::atomic<unsigned> g_host_ranks{0};    // 0: not set by the caller - the environment decides
unsigned host_ranks() {
    const unsigned set = g_host_ranks.load(std::memory_order_relaxed);
    if (set) return set;
    static const unsigned from_env = [] {
        for (const char* name : {"SP_HOST_RANKS", "LOCAL_WORLD_SIZE"})
            if (const char* e = std::getenv(name)) { const int v = std::atoi(e); if (v >= 1) return (unsigned)std::min(v, 1024); }
        return 1u;
    }();
    return from_env;
}
void set_host_ranks(unsigned ranks) { g_host_ranks.store(std::min(ranks, 1024u), std::memory_order_relaxed); }
unsigned host_cpu_budget() { return std::max(1u, host_effective_cpus() / host_ranks()); }
bool host_oversubscribed() { return host_ranks() > host_effective_cpus(); }

void host_parallel_for(size_t n, size_t min_chunk, const std::function<void(size_t, size_t)>& fn) {
    static const int env_threads = [] { const char* e = std::getenv("SP_HOST_THREADS"); return e ? std::atoi(e) : 0; }();
    const unsigned max_threads = env_threads >= 1 ? (unsigned)std::min(env_threads, 256) : std::min(64u, host_cpu_budget());
    const size_t parts = std::max<size_t>(1, std::min<size_t>(max_threads, n / std::max<size_t>(min_chunk, 1)));
    if (parts <= 1) { fn(0, n); return; }
    const size_t per = (n + parts - 1) / parts;
    std::vector<std::thread> ts;
    std::vector<std::exception_ptr> errs(parts);
    for (size_t t = 1; t < parts; ++t)
        ts.emplace_back([&, t] { try { const size_t b = std::min(n, t * per); fn(b, std::min(n, b + per)); } catch (...) { errs[t] = std::current_exception(); } });
    try { fn(0, std::min(n, per)); } catch (...) { errs[0] = std::current_exception(); }
    for (auto& t : ts) t.join();
    for (auto& e : errs) if (e) std::rethrow_exception(e);
}

// In-place Montgomery batch inversion of non-zero elements (one inversion per call)
static void host_batch_inverse_nonzero(std::vector<fe>& v) {
    if (v.empty()) return;
    std::vector<fe> pre(v.size());
    fe acc = fe_one();
    for (size_t i = 0; i < v.size(); ++i) { pre[i] = acc; acc = fe_mul(acc, v[i]); }
    fe inv = fe_inv(acc);
    for (size_t i = v.size(); i-- > 0;) { const fe x = v[i]; v[i] = fe_mul(inv, pre[i]); inv = fe_mul(inv, x); }
}

// Pass A of build_main_trace: everything that decides the SHAPE of the trace (and everything that can fail), without touching the
// table - the four memory addresses and three offsets of every step, the unused range-check values and memory addresses that
// become extra rows, the row count.  Validates what the fill (host or device) relies on: every instruction decodes, every
// operand cell exists, jnz rows have the flags the reference demands.
void plan_main_trace(const std::vector<RegisterState>& regs, const CairoMemory& mem, PublicInputs& pub, TracePlan& P) {
    const size_t steps = regs.size();
    if (steps == 0) throw std::runtime_error("empty register trace");
    const MemorySegment* rc_seg = pub.segment(0);
    P = TracePlan();
    P.steps = steps;
    P.cols = rc_seg ? 43 : 34;
    std::vector<uint64_t> addr(4 * steps);    // [step][pc, dst, op0, op1]
    // every part of the pass runs on the host's threads: each chunk of steps keeps its own set of seen offsets and its own address range
    std::mutex merge;
    std::vector<uint64_t> seen_words(1024, 0);               // 65536 bits: the offsets some step uses
    uint64_t lo_addr = ~0ULL, hi_addr = 0;
    host_parallel_for(steps, 4096, [&](size_t b, size_t e) {
        std::vector<uint64_t> seen_local(1024, 0);
        uint64_t lo = ~0ULL, hi = 0;
        for (size_t i = b; i < e; ++i) {
            const RegisterState& r = regs[i];
            const Decoded d = decode(mem_at(mem, r.pc));
            const uint64_t dst_addr = add_signed(d.dst_reg ? r.fp : r.ap, d.off_dst);
            const uint64_t op0_addr = add_signed(d.op0_reg ? r.fp : r.ap, d.off_op0);
            const fe& op0 = mem_at(mem, op0_addr);
            const uint64_t op1_base = d.op1_src == 0 ? fe_low_u64(op0) : d.op1_src == 1 ? r.pc : d.op1_src == 2 ? r.fp : r.ap;
            const uint64_t op1_addr = add_signed(op1_base, d.off_op1);
            (void)mem_at(mem, dst_addr);
            (void)mem_at(mem, op1_addr);
            if (d.pc_update == 4 && !(d.res_logic == 0 && d.opcode == 0 && d.ap_update != 1)) throw std::runtime_error("Undefined Behavior");   // execution_trace.rs:382-440
            addr[4 * i] = r.pc; addr[4 * i + 1] = dst_addr; addr[4 * i + 2] = op0_addr; addr[4 * i + 3] = op1_addr;
            lo = std::min(std::min(lo, r.pc), std::min(dst_addr, std::min(op0_addr, op1_addr)));
            hi = std::max(std::max(hi, r.pc), std::max(dst_addr, std::max(op0_addr, op1_addr)));
            seen_local[d.off_dst >> 6] |= 1ULL << (d.off_dst & 63u);
            seen_local[d.off_op0 >> 6] |= 1ULL << (d.off_op0 & 63u);
            seen_local[d.off_op1 >> 6] |= 1ULL << (d.off_op1 & 63u);
        }
        std::lock_guard<std::mutex> lk(merge);
        for (size_t w = 0; w < 1024; ++w) seen_words[w] |= seen_local[w];
        lo_addr = std::min(lo_addr, lo); hi_addr = std::max(hi_addr, hi);
    });
    // get_rc_holes (execution_trace.rs:136-185): every value strictly between the smallest and the largest offset that no step uses
    std::vector<uint16_t>& missing = P.missing;
    {
        auto seen = [&](uint32_t v) { return (seen_words[v >> 6] >> (v & 63u)) & 1ULL; };
        uint32_t lo = 0, hi = 65535;
        while (!seen(lo)) ++lo;
        while (!seen(hi)) --hi;
        for (uint32_t v = lo + 1; v < hi; ++v) if (!seen(v)) missing.push_back((uint16_t)v);
        const size_t pad = ((missing.size() + 2) / 3) * 3 - missing.size();
        for (size_t i = 0; i < pad; ++i) missing.push_back((uint16_t)hi);
        pub.range_check_min = (uint16_t)lo; pub.range_check_max = (uint16_t)hi;
        pub.has_rc_min = pub.has_rc_max = true;
    }
    // get_memory_holes (execution_trace.rs:195-255): the addresses above the public memory that lie between two accessed
    // addresses and are not accessed themselves, in increasing order
    std::vector<uint64_t>& holes = P.holes;
    uint64_t hi_addr_all = hi_addr;
    {
        const uint64_t codelen = pub.public_memory.size();
        const uint64_t lo = lo_addr, hi = hi_addr;
        // A run's memory is (nearly) continuous, so [lo, hi] is a small multiple of the 4 * steps accesses: presence bitmap.  A sparse
        // range (an odd or hostile dump) takes the sorted formulation, whose cost is O(accesses) before any hole is enumerated; and a
        // dump whose holes could never fit a trace is refused before gigabytes of them are collected.
        constexpr uint64_t MAX_HOLES = 1ULL << 30;
        if (hi - lo <= 64ULL * 4 * steps + 4096) {           // presence bitmap over [lo, hi], filled and scanned by all threads
            const size_t words = (size_t)((hi - lo) >> 6) + 1;
            std::unique_ptr<std::atomic<uint64_t>[]> bits(new std::atomic<uint64_t>[words]);
            host_parallel_for(words, 1 << 16, [&](size_t b, size_t e) { for (size_t w = b; w < e; ++w) bits[w].store(0, std::memory_order_relaxed); });
            host_parallel_for(addr.size(), 1 << 16, [&](size_t b, size_t e) {
                for (size_t k = b; k < e; ++k) {     // (test first: every step's pc hits the same few words from every thread)
                    const uint64_t a = addr[k] - lo, m = 1ULL << (a & 63);
                    if (!(bits[a >> 6].load(std::memory_order_relaxed) & m)) bits[a >> 6].fetch_or(m, std::memory_order_relaxed);
                }
            });
            const uint64_t first = std::max(lo + 1, codelen + 1);
            if (first < hi) {
                const size_t w0 = (size_t)((first - lo) >> 6), w1 = (size_t)((hi - 1 - lo) >> 6) + 1;    // words that hold [first, hi)
                const size_t parts = std::max<size_t>(1, std::min<size_t>(64, (w1 - w0) / 4096));
                std::vector<std::vector<uint64_t>> found(parts);
                const size_t per = (w1 - w0 + parts - 1) / parts;
                host_parallel_for(parts, 1, [&](size_t pb, size_t pe) {
                    for (size_t part = pb; part < pe; ++part)
                        for (size_t w = w0 + part * per; w < std::min(w1, w0 + (part + 1) * per); ++w) {
                            uint64_t zero = ~bits[w].load(std::memory_order_relaxed);
                            while (zero) {
                                const uint64_t h = lo + ((uint64_t)w << 6) + (uint64_t)__builtin_ctzll(zero);
                                zero &= zero - 1;
                                if (h >= first && h < hi) found[part].push_back(h);
                            }
                        }
                });
                for (auto& f : found) holes.insert(holes.end(), f.begin(), f.end());
            }
        } else {                                // scattered addresses: sort (the reference's own formulation)
            std::vector<uint64_t> sorted(addr);
            std::sort(sorted.begin(), sorted.end());
            uint64_t prev = sorted[0], count = 0;
            for (uint64_t a : sorted) {     // count first
                if (a - prev > 1 && a > codelen) count += a - std::max(prev + 1, codelen + 1);
                if (count > MAX_HOLES) throw std::runtime_error("more than 2^30 memory holes: not the memory of one run");
                prev = a;
            }
            holes.reserve(count);
            prev = sorted[0];
            for (uint64_t a : sorted) {
                const uint64_t diff = a - prev;
                if (diff != 1 && diff != 0 && a > codelen)
                    for (uint64_t h = prev + 1; h < a; ++h) if (h > codelen) holes.push_back(h);
                prev = a;
            }
        }
    }
    uint64_t& hi_addr_ref = hi_addr_all;
    if (rc_seg) {  // add_rc_builtin_columns (execution_trace.rs:358-379, :604-624): the first rows carry the range-checked values
        P.rc_start = rc_seg->start;
        P.rc_count = std::min<uint64_t>(rc_seg->end > rc_seg->start ? rc_seg->end - rc_seg->start : 0, steps);
        for (uint64_t k = 0; k < P.rc_count; ++k) (void)mem_at(mem, P.rc_start + k);
        if (P.rc_count) hi_addr_ref = std::max(hi_addr_ref, P.rc_start + P.rc_count - 1);
    }
    P.r_rc = steps; P.r_holes = P.r_rc + missing.size() / 3; P.r_dummy = P.r_holes + (holes.size() + 3) / 4;
    const size_t r_pad = P.r_dummy + (pub.public_memory.size() >> 2) + 1;
    size_t n = 1;
    while (n < r_pad) n <<= 1;
    P.n = n;
    P.mem_cells = hi_addr_all + 1;
    P.dense = mem.sparse.empty() && P.mem_cells <= mem.dense.size();   // every cell a row reads lives in the flat array
}

// Pass B: build_cairo_execution_trace (execution_trace.rs:261-356) and the rows behind the steps, into a host table.
void fill_main_trace(const std::vector<RegisterState>& regs, const CairoMemory& mem, const TracePlan& P, TraceColumns& T) {
    const size_t steps = P.steps, cols = P.cols, n = P.n, r_rc = P.r_rc, r_holes = P.r_holes, r_dummy = P.r_dummy;
    const std::vector<uint16_t>& missing = P.missing;
    const std::vector<uint64_t>& holes = P.holes;
    const fe zero = fe_zero(), one = fe_one();
    T.allocate(n, cols);
    host_parallel_for(steps, 4096, [&](size_t b, size_t e) {
        std::vector<size_t> jnz_rows;
        std::vector<fe> jnz_dst;
        for (size_t i = b; i < e; ++i) {
            const RegisterState& r = regs[i];
            const fe& inst = mem_at(mem, r.pc);
            const Decoded d = decode(inst);
            for (int k = 0; k < 15; ++k) T.at(i, k) = ((d.flags >> k) & 1) ? one : zero;
            T.at(i, 15) = zero;
            const uint64_t dst_addr = add_signed(d.dst_reg ? r.fp : r.ap, d.off_dst);
            const uint64_t op0_addr = add_signed(d.op0_reg ? r.fp : r.ap, d.off_op0);
            fe dst = mem_at(mem, dst_addr);
            fe op0 = mem_at(mem, op0_addr);
            const uint64_t op1_base = d.op1_src == 0 ? fe_low_u64(op0) : d.op1_src == 1 ? r.pc : d.op1_src == 2 ? r.fp : r.ap;
            const uint64_t op1_addr = add_signed(op1_base, d.off_op1);
            const fe op1 = mem_at(mem, op1_addr);
            fe res;
            bool deferred = false;
            if (d.pc_update == 4) {  // jnz: res holds dst^-1 (execution_trace.rs:382-440); inverted in one batch below
                if (!(d.res_logic == 0 && d.opcode == 0 && d.ap_update != 1)) throw std::runtime_error("Undefined Behavior");
                res = dst;
                if (!fe_is_zero(dst)) { deferred = true; jnz_rows.push_back(i); jnz_dst.push_back(dst); }
            } else {
                res = d.res_logic == 0 ? op1 : d.res_logic == 1 ? fe_add(op0, op1) : fe_mul(op0, op1);
            }
            // update_values (execution_trace.rs:572-592)
            if (d.opcode == 1) { op0 = fe_from_u64(r.pc + (d.op1_src == 1 ? 2 : 1)); dst = fe_from_u64(r.fp); }
            else if (d.opcode == 4) { res = dst; }
            T.at(i, 16) = res; T.at(i, 17) = fe_from_u64(r.ap); T.at(i, 18) = fe_from_u64(r.fp); T.at(i, 19) = fe_from_u64(r.pc);
            T.at(i, 20) = fe_from_u64(dst_addr); T.at(i, 21) = fe_from_u64(op0_addr); T.at(i, 22) = fe_from_u64(op1_addr);
            T.at(i, 23) = inst; T.at(i, 24) = dst; T.at(i, 25) = op0; T.at(i, 26) = op1;
            T.at(i, 27) = fe_from_u64(d.off_dst); T.at(i, 28) = fe_from_u64(d.off_op0); T.at(i, 29) = fe_from_u64(d.off_op1);
            const fe t0 = ((d.flags >> 9) & 1) ? dst : zero;
            T.at(i, 30) = t0; T.at(i, 31) = deferred ? zero : fe_mul(t0, res); T.at(i, 32) = fe_mul(op0, op1);
            T.at(i, 33) = (i + 1 == steps) ? zero : one;
            for (size_t c = 34; c < cols; ++c) T.at(i, c) = zero;
        }
        host_batch_inverse_nonzero(jnz_dst);
        for (size_t k = 0; k < jnz_rows.size(); ++k) {
            const size_t i = jnz_rows[k];
            T.at(i, 16) = jnz_dst[k];
            T.at(i, 31) = fe_mul(T.at(i, 30), jnz_dst[k]);
        }
    });
    if (P.rc_count) {  // add_rc_builtin_columns (execution_trace.rs:358-379, :604-624)
        host_parallel_for(P.rc_count, 4096, [&](size_t b, size_t e) {
            for (size_t k = b; k < e; ++k) {
                const fe& v = mem_at(mem, P.rc_start + k);
                const fe raw = fe_from_mont(v);
                for (int c = 0; c < 8; ++c) T.at(k, 34 + c) = fe_from_u64((raw.v[c / 2] >> (16 * (c & 1))) & 0xffff);
                T.at(k, 42) = v;
            }
        });
    }
    auto get_row = [&](size_t r) { std::vector<fe> row(cols); for (size_t c = 0; c < cols; ++c) row[c] = T.at(r, c); return row; };
    // fill_rc_holes (execution_trace.rs:136-185): zero rows that carry three missing offsets each
    for (size_t r = r_rc; r < r_holes; ++r) {
        for (size_t c = 0; c < cols; ++c) T.at(r, c) = zero;
        for (int k = 0; k < 3; ++k) T.at(r, 27 + k) = fe_from_u64(missing[3 * (r - r_rc) + k]);
    }
    // fill_memory_holes (execution_trace.rs:195-255): copies of the last row with four holes each as addresses
    if (r_dummy > r_holes) {
        const std::vector<fe> last = get_row(r_holes - 1);
        host_parallel_for(r_dummy - r_holes, 1024, [&](size_t b, size_t e) {
            for (size_t k = b; k < e; ++k) {
                const size_t r = r_holes + k;
                for (size_t c = 0; c < cols; ++c) T.at(r, c) = last[c];
                for (int c = 0; c < 4; ++c) if (4 * k + c < holes.size()) T.at(r, 19 + c) = fe_from_u64(holes[4 * k + c]);
            }
        });
    }
    // add_pub_memory_dummy_accesses (execution_trace.rs:91-96, :112-127) and pad_with_last_row (:82-84): the last row with
    // zeroed memory columns, up to the next power of two
    {
        std::vector<fe> last = get_row(r_dummy - 1);
        for (int c = 19; c <= 26; ++c) last[c] = zero;
        host_parallel_for(cols, 1, [&](size_t b, size_t e) {
            for (size_t c = b; c < e; ++c) std::fill(&T.at(r_dummy, c), &T.at(0, c) + n, last[c]);
        });
    }
}

void build_main_trace(const std::vector<RegisterState>& regs, const CairoMemory& mem, PublicInputs& pub, TraceColumns& T) {
    TracePlan P;
    plan_main_trace(regs, mem, pub, P);
    fill_main_trace(regs, mem, P, T);
}

// The inputs of the device's trace builder in one buffer (page-locked when the process already drives a device, so that it crosses
// PCIe as one plain DMA): register states, the memory cells the rows read, the two hole lists.
void TraceImage::release() {
    if (base) { if (pinned) (void)hipHostFree(base); else std::free(base); }
    if (retired) std::free(retired);
    base = nullptr; retired = nullptr; bytes = 0; pinned = false;
}
void TraceImage::build(const std::vector<RegisterState>& regs, const CairoMemory& mem, const TracePlan& P) {
    release();
    if (!P.dense) return;
    auto up = [](uint64_t x) { return (x + 255) & ~(uint64_t)255; };
    off_regs = 0;
    off_mem = up(off_regs + 24 * (uint64_t)P.steps);
    off_missing = up(off_mem + 32 * P.mem_cells);
    off_holes = up(off_missing + 2 * (uint64_t)P.missing.size());
    bytes = up(off_holes + 8 * (uint64_t)P.holes.size());
    void* p = hip_runtime_in_use() ? alloc_pinned(bytes) : nullptr;
    if (p) pinned = true;
    else if (posix_memalign(&p, 4096, bytes) != 0 || !p) throw std::bad_alloc();
    base = static_cast<uint8_t*>(p);
    static_assert(sizeof(RegisterState) == 24, "register states are uploaded as they are");
    const uint8_t* rsrc = reinterpret_cast<const uint8_t*>(regs.data());
    host_parallel_for(24 * (size_t)P.steps, 1 << 22, [&](size_t b, size_t e) { std::memcpy(base + off_regs + b, rsrc + b, e - b); });
    const uint8_t* msrc = reinterpret_cast<const uint8_t*>(mem.dense.data());
    host_parallel_for(32 * (size_t)P.mem_cells, 1 << 22, [&](size_t b, size_t e) { std::memcpy(base + off_mem + b, msrc + b, e - b); });
    if (!P.missing.empty()) std::memcpy(base + off_missing, P.missing.data(), 2 * P.missing.size());
    if (!P.holes.empty()) std::memcpy(base + off_holes, P.holes.data(), 8 * P.holes.size());
}
bool TraceImage::try_pin() {
    std::lock_guard<std::mutex> lk(pin_mutex);
    if (pinned || !base) return pinned;
    uint8_t* p = static_cast<uint8_t*>(alloc_pinned(bytes));
    if (!p) return false;
    host_parallel_for(bytes, 1 << 22, [&](size_t b, size_t e) { std::memcpy(p + b, base + b, e - b); });
    retired = base;           // not freed before the run is: another prover of the process may be uploading from it right now
    base = p; pinned = true;
    return true;
}

// cairo-run's non-proof-mode layout (reference src/cairo/runner/run.rs:64-240 drives cairo-vm 0.6.0 that way): program at
// 1..L, execution segment behind it, then one segment per builtin the program declares (layout order: output, range_check),
// then the two empty segments whose address is the return fp / end pc of main.  main's initial stack is
// [builtin bases ..., return fp, end pc]; it returns the advanced builtin pointers on top of the stack, which give the
// used length of every builtin segment (cairo-vm's read_return_values -> stop_ptr; run.rs:211-222 for range_check).
// The bases depend on the length of the execution segment, known only after the run: a first run with far-away
// placeholder bases measures the segments, the second one uses the real addresses (no hint-free program can branch on them).
void run_program_builtins(const std::vector<fe>& program, uint32_t builtins_mask, std::vector<RegisterState>& regs, CairoMemory& mem,
                          uint64_t max_steps, uint64_t entry_pc, std::vector<MemorySegment>& segments_out) {
    const uint64_t L = program.size();
    std::vector<uint8_t> kinds;   // segment types in stack order (air.rs:156-160: 0 RangeCheck, 1 Output)
    if (builtins_mask & 1u) kinds.push_back(1);
    if (builtins_mask & 2u) kinds.push_back(0);
    const size_t nb = kinds.size();
    std::vector<uint64_t> base(nb), used(nb, 0);
    for (size_t b = 0; b < nb; ++b) base[b] = (1ULL << 40) * (b + 1);
    uint64_t end_marker = ~0ULL >> 8;
    for (int pass = 0; pass < 2; ++pass) {
        mem.clear();
        regs.clear();
        for (uint64_t i = 0; i < L; ++i) mem.set(i + 1, program[i]);
        for (size_t b = 0; b < nb; ++b) mem.set(L + 1 + b, fe_from_u64(base[b]));
        mem.set(L + 1 + nb, fe_from_u64(end_marker));  // return fp
        mem.set(L + 2 + nb, fe_from_u64(end_marker));  // return pc
        std::vector<Decoded> dcache; std::vector<uint8_t> dcache_ok;
        uint64_t pc = entry_pc, ap = L + 3 + nb, fp = L + 3 + nb;
        const uint64_t fp0 = fp;
        bool done = false;
        while (!done) {
            if (regs.size() >= max_steps) throw std::runtime_error("step limit exceeded");
            regs.push_back(RegisterState{ap, fp, pc});
            Decoded d;
            if (pc < dcache.size() && dcache_ok[pc]) d = dcache[pc];      // (program cells 1..L are never written: ap starts above them)
            else {
                d = decode(mem_at(mem, pc));
                if (pc <= L && pc < (1u << 24)) {
                    if (pc >= dcache.size()) { dcache.resize(pc + 1024); dcache_ok.resize(pc + 1024, 0); }
                    dcache[pc] = d; dcache_ok[pc] = 1;
                }
            }
            uint64_t size = d.op1_src == 1 ? 2 : 1;
            uint64_t dst_addr = add_signed(d.dst_reg ? fp : ap, d.off_dst);
            uint64_t op0_addr = add_signed(d.op0_reg ? fp : ap, d.off_op0);
            if (d.opcode == 1) {  // call: [ap] = fp, [ap+1] = pc + size
                // (memory is write-once: a cell an earlier instruction has written with another value - `[ap + 4] = ...` ahead of a call -
                // is an inconsistent run, cairo-vm's InconsistentMemory, not something to overwrite and prove)
                auto write_once = [&](uint64_t addr, const fe& v) {
                    const fe* old = mem.get(addr);
                    if (old && !fe_eq(*old, v)) throw std::runtime_error("inconsistent memory: a call frame over a cell that holds another value");
                    if (!old) mem.set(addr, v);
                };
                write_once(ap, fe_from_u64(fp));
                write_once(ap + 1, fe_from_u64(pc + size));
            }
            const fe* op0p = mem.get(op0_addr);
            uint64_t op1_base;
            if (d.op1_src == 0) {
                if (!op0p) throw std::runtime_error("op0 unknown");
                const fe raw = fe_from_mont(*op0p);      // [op0 + off]: op0 has to be an address
                for (int l = 2; l < 8; ++l) if (raw.v[l]) throw std::runtime_error("op0 is not an address");
                op1_base = fe_low_u64(*op0p);
            }
            else op1_base = d.op1_src == 1 ? pc : d.op1_src == 2 ? fp : ap;
            uint64_t op1_addr = add_signed(op1_base, d.off_op1);
            const fe* op1p = mem.get(op1_addr);
            const fe* dstp = mem.get(dst_addr);
            if (d.opcode == 4) {  // assert_eq with operand deduction
                if (!dstp) {
                    if (!op1p || (d.res_logic != 0 && !op0p)) throw std::runtime_error("cannot deduce dst");
                    fe r = d.res_logic == 0 ? *op1p : d.res_logic == 1 ? fe_add(*op0p, *op1p) : fe_mul(*op0p, *op1p);
                    mem.set(dst_addr, r);
                } else if (!op1p) {
                    fe r;
                    if (d.res_logic == 0) r = *dstp;
                    else if (!op0p) throw std::runtime_error("cannot deduce op1");
                    else if (d.res_logic == 1) r = fe_sub(*dstp, *op0p);
                    else r = fe_mul(*dstp, fe_inv(*op0p));
                    mem.set(op1_addr, r);
                } else if (!op0p && d.res_logic != 0) {
                    mem.set(op0_addr, d.res_logic == 1 ? fe_sub(*dstp, *op1p) : fe_mul(*dstp, fe_inv(*op1p)));
                }
                dstp = mem.get(dst_addr); op0p = mem.get(op0_addr); op1p = mem.get(op1_addr);
            }
            if (!dstp || !op1p) throw std::runtime_error("operand unknown");
            fe res = fe_zero();
            if (d.pc_update != 4) {
                if (d.res_logic != 0 && !op0p) throw std::runtime_error("op0 unknown");
                res = d.res_logic == 0 ? *op1p : d.res_logic == 1 ? fe_add(*op0p, *op1p) : fe_mul(*op0p, *op1p);
            }
            if (d.opcode == 4 && !fe_eq(*dstp, res)) throw std::runtime_error("assert_eq failed");
            uint64_t next_pc;
            switch (d.pc_update) {
                case 0: next_pc = pc + size; break;
                case 1: next_pc = (uint64_t)fe_to_i64(res); break;
                case 2: next_pc = pc + (uint64_t)fe_to_i64(res); break;
                default: next_pc = fe_is_zero(*dstp) ? pc + size : pc + (uint64_t)fe_to_i64(*op1p); break;
            }
            uint64_t next_ap = ap, next_fp = fp;
            if (d.ap_update == 1) next_ap = ap + (uint64_t)fe_to_i64(res);
            else if (d.ap_update == 2) next_ap = ap + 1;
            if (d.opcode == 1) { next_ap = ap + 2; next_fp = ap + 2; }
            else if (d.opcode == 2) {
                next_fp = fe_low_u64(*dstp);
                if (fp == fp0) done = true;  // main's own ret
            }
            pc = next_pc; ap = next_ap; fp = next_fp;
        }
        // main leaves the advanced builtin pointers on top of its stack (read_return_values)
        const uint64_t final_ap = regs.back().ap;
        for (size_t b = 0; b < nb; ++b) {
            const fe* stop = mem.get(final_ap - nb + b);
            if (!stop) throw std::runtime_error("builtin stop pointer missing from the final stack");
            const uint64_t sp = fe_low_u64(*stop);
            if (sp < base[b] || sp - base[b] > (1ULL << 32)) throw std::runtime_error("invalid builtin stop pointer");
            used[b] = sp - base[b];
        }
        uint64_t next = final_ap;   // the execution segment ends where ap ends
        for (size_t b = 0; b < nb; ++b) { base[b] = next; next += used[b]; }
        end_marker = next;
    }
    segments_out.clear();
    for (size_t b = 0; b < nb; ++b) {
        // every cell of a builtin segment must have been written; range-checked values lie in [0, 2^128)
        for (uint64_t a = base[b]; a < base[b] + used[b]; ++a) {
            const fe* v = mem.get(a);
            if (!v) throw std::runtime_error("hole in a builtin segment");
            if (kinds[b] == 0) {
                fe raw = fe_from_mont(*v);
                for (int k = 4; k < 8; ++k) if (raw.v[k]) throw std::runtime_error("range-check builtin: value out of [0, 2^128)");
            }
        }
        segments_out.push_back(MemorySegment{kinds[b], base[b], base[b] + used[b]});
    }
}

void run_program_plain(const std::vector<fe>& program, std::vector<RegisterState>& regs, CairoMemory& mem, uint64_t max_steps, uint64_t entry_pc) {
    std::vector<MemorySegment> none;
    run_program_builtins(program, 0, regs, mem, max_steps, entry_pc, none);
}

std::vector<fe> fibonacci_program(uint64_t fib_index) {
    static const char* words[22] = {
        "480680017fff8000", "1", "480680017fff8000", "1", "480680017fff8000", nullptr, "1104800180018000", "3",
        "208b7fff7fff7ffe", "20780017fff7ffd", "5", "480a7ffc7fff8000", "480a7ffc7fff8000", "208b7fff7fff7ffe",
        "482a7ffc7ffb8000", "480a7ffc7fff8000", "48127ffe7fff8000", "482680017ffd8000",
        "800000000000011000000000000000000000000000000000000000000000000",  // -1
        "1104800180018000",
        "800000000000010fffffffffffffffffffffffffffffffffffffffffffffff7",  // -10 (relative call offset)
        "208b7fff7fff7ffe"};
    std::vector<fe> prog(22);
    for (int i = 0; i < 22; ++i) {
        if (!words[i]) { prog[i] = fe_from_u64(fib_index); continue; }
        std::string h(words[i]);
        h = std::string(64 - h.size(), '0') + h;
        uint8_t b[32];
        for (int k = 0; k < 32; ++k) b[k] = (uint8_t)std::stoul(h.substr(2 * k, 2), nullptr, 16);
        prog[i] = fe_from_bytes_be(b);
    }
    return prog;
}

std::vector<uint8_t> serialize_public_inputs(const PublicInputs& p) {
    std::vector<uint8_t> b;
    auto u64 = [&](uint64_t v) { for (int i = 7; i >= 0; --i) b.push_back((uint8_t)(v >> (8 * i))); };
    auto felt = [&](const fe& x) { uint8_t t[32]; fe_to_bytes_be(x, t); b.insert(b.end(), t, t + 32); };
    u64(32);
    felt(p.pc_init); felt(p.ap_init); felt(p.fp_init); felt(p.pc_final); felt(p.ap_final);
    if (p.has_rc_min) { b.push_back(1); b.push_back((uint8_t)(p.range_check_min >> 8)); b.push_back((uint8_t)p.range_check_min); } else b.push_back(0);
    if (p.has_rc_max) { b.push_back(1); b.push_back((uint8_t)(p.range_check_max >> 8)); b.push_back((uint8_t)p.range_check_max); } else b.push_back(0);
    u64(p.memory_segments.size());
    for (auto& s : p.memory_segments) { b.push_back(s.type); u64(s.start); u64(s.end); }
    u64(p.public_memory.size());
    for (auto& kv : p.public_memory) { felt(fe_from_u64(kv.first)); felt(kv.second); }
    u64(p.num_steps);
    return b;
}

// PublicInputs::deserialize (reference src/cairo/air.rs:278-450): the inverse of the writer above, with the reference's tolerance
// (bytes behind num_steps are ignored; a HashMap on their side, so a repeated address keeps its last value and the order is free).
// Throws "malformed: ..." on what the reference answers with a DeserializationError; addresses must fit 64 bits here.
PublicInputs deserialize_public_inputs(const uint8_t* d, size_t len) {
    size_t pos = 0;
    auto need = [&](size_t k) { if (k > len - pos) throw std::runtime_error("malformed: InvalidAmountOfBytes (public inputs)"); };
    auto u64 = [&]() { need(8); uint64_t v = 0; for (int i = 0; i < 8; ++i) v = (v << 8) | d[pos + i]; pos += 8; return v; };
    static const uint8_t P_BE[32] = {0x08, 0, 0, 0, 0, 0, 0, 0x11, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0x01};
    const uint64_t felt_len = u64();
    if (felt_len != 32) throw std::runtime_error("malformed: element length is not 32 (public inputs)");
    auto felt = [&]() {
        need(32);
        if (std::memcmp(d + pos, P_BE, 32) >= 0) throw std::runtime_error("malformed: field element out of range (public inputs)");
        const fe x = fe_from_bytes_be(d + pos); pos += 32; return x;
    };
    PublicInputs p;
    p.pc_init = felt(); p.ap_init = felt(); p.fp_init = felt(); p.pc_final = felt(); p.ap_final = felt();
    auto opt_u16 = [&](bool& has, uint16_t& v) {
        need(1);
        const uint8_t tag = d[pos++];
        if (tag > 1) throw std::runtime_error("malformed: FieldFromBytesError (range-check tag)");
        has = tag == 1;
        if (has) { need(2); v = (uint16_t)((d[pos] << 8) | d[pos + 1]); pos += 2; }
    };
    opt_u16(p.has_rc_min, p.range_check_min);
    opt_u16(p.has_rc_max, p.range_check_max);
    const uint64_t n_seg = u64();
    if (n_seg > (len - pos) / 17) throw std::runtime_error("malformed: InvalidAmountOfBytes (memory segments)");
    for (uint64_t i = 0; i < n_seg; ++i) {
        need(1);
        const uint8_t type = d[pos++];
        if (type > 1) throw std::runtime_error("malformed: FieldFromBytesError (memory segment type)");
        const uint64_t start = u64(), end = u64();
        bool seen = false;
        for (auto& sgm : p.memory_segments) if (sgm.type == type) { sgm.start = start; sgm.end = end; seen = true; }   // (a map: the last entry of a type stays)
        if (!seen) p.memory_segments.push_back(MemorySegment{type, start, end});
    }
    const uint64_t n_pm = u64();
    if (n_pm > (len - pos) / 64) throw std::runtime_error("malformed: InvalidAmountOfBytes (public memory)");
    std::vector<std::pair<uint64_t, fe>> cells;
    cells.reserve(n_pm);
    for (uint64_t i = 0; i < n_pm; ++i) {
        need(64);
        for (int k = 0; k < 24; ++k) if (d[pos + k]) throw std::runtime_error("malformed: a public-memory address beyond 64 bits");
        uint64_t a = 0;
        for (int k = 24; k < 32; ++k) a = (a << 8) | d[pos + k];
        pos += 32;
        cells.emplace_back(a, felt());
    }
    std::stable_sort(cells.begin(), cells.end(), [](const auto& x, const auto& y) { return x.first < y.first; });
    for (size_t i = 0; i < cells.size(); ++i) {
        if (i + 1 < cells.size() && cells[i + 1].first == cells[i].first) continue;      // the last value of a repeated address
        p.public_memory.push_back(cells[i]);
    }
    p.num_steps = u64();
    return p;
}

}  // namespace sp
