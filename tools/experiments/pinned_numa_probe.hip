// Where does hipHostMalloc put its pages, and what does that do to the host-to-device DMA rate?  (round 3: the page-locked trace
// of a run uploaded at 56 GB/s in one process and at 31 GB/s in the next.)  For a few allocations: first-touch by this thread or
// by threads spread over the machine, the NUMA node of the pages (/proc/self/numa_maps), the H2D rate of one large copy.
// build: hipcc -O2 --offload-arch=gfx950 pinned_numa_probe.hip -o pinned_numa_probe -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <thread>
#include <vector>
#include <sched.h>
#include <unistd.h>
#include <sys/syscall.h>
static void run_on_cpu(int cpu) { cpu_set_t s; CPU_ZERO(&s); CPU_SET(cpu, &s); sched_setaffinity(0, sizeof s, &s); }
static void run_anywhere() { cpu_set_t s; CPU_ZERO(&s); for (int i = 0; i < 256; ++i) CPU_SET(i, &s); sched_setaffinity(0, sizeof s, &s); }
static long prefer_node(int node) {   // MPOL_PREFERRED = 1, MPOL_DEFAULT = 0
    unsigned long mask = node >= 0 ? 1ul << node : 0;
    return syscall(SYS_set_mempolicy, node >= 0 ? 1 : 0, node >= 0 ? &mask : nullptr, node >= 0 ? 64 : 0);
}
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static std::string numa_of(void* p) {
    std::ifstream f("/proc/self/numa_maps");
    char key[32];
    snprintf(key, sizeof key, "%lx", (unsigned long)p);
    std::string line;
    while (std::getline(f, line))
        if (line.compare(0, strlen(key), key) == 0) {
            std::string out;
            std::istringstream is(line);
            std::string tok;
            while (is >> tok) if (tok[0] == 'N' && tok.find('=') != std::string::npos) out += tok + " ";
            return out.empty() ? "(no node counts)" : out;
        }
    return "(mapping not found)";
}
int main() {
    const size_t bytes = (size_t)570 << 20;
    void* dev = nullptr;
    hipMalloc(&dev, bytes);
    int node = -1;
    char bus[64] = {0};
    hipDeviceGetPCIBusId(bus, sizeof bus, 0);
    { std::string path = std::string("/sys/bus/pci/devices/") + bus + "/numa_node"; for (auto& ch : path) ch = tolower(ch); std::ifstream f(path); f >> node; }
    printf("GPU 0 at %s, numa_node %d\n", bus, node);
    {   // the allocating thread on the other socket, with and without a memory policy that prefers the GPU's node
        struct C2 { const char* name; int cpu; int prefer; } c2[] = {{"allocated from CPU 70 (node 1), default policy", 70, -1}, {"allocated from CPU 70 (node 1), MPOL_PREFERRED node of the GPU", 70, node},
                                                                     {"allocated from CPU 3 (node 0), default policy", 3, -1}};
        for (auto& c : c2) {
            run_on_cpu(c.cpu);
            if (c.prefer >= 0) prefer_node(c.prefer);
            void* h = nullptr;
            if (hipHostMalloc(&h, bytes, hipHostMallocDefault) != hipSuccess) { printf("%s: allocation failed\n", c.name); continue; }
            prefer_node(-1);
            memset(h, 1, bytes);
            hipMemcpy(dev, h, bytes, hipMemcpyHostToDevice);
            double best = 1e9;
            for (int r = 0; r < 3; ++r) { double a = now(); hipMemcpy(dev, h, bytes, hipMemcpyHostToDevice); best = std::min(best, now() - a); }
            printf("%-64s H2D %5.1f GB/s, pages: %s\n", c.name, bytes / best * 1e-6, numa_of(h).c_str());
            hipHostFree(h);
            run_anywhere();
        }
    }
    struct Case { const char* name; unsigned flags; int touch_threads; } cases[] = {
        {"default flags, touched by this thread", hipHostMallocDefault, 1}, {"default flags, touched by 16 threads", hipHostMallocDefault, 16},
        {"hipHostMallocNumaUser, touched by this thread", hipHostMallocNumaUser, 1}, {"hipHostMallocNumaUser, touched by 16 threads", hipHostMallocNumaUser, 16},
        {"default flags, touched by 16 threads (again)", hipHostMallocDefault, 16}, {"default flags, touched by 64 threads", hipHostMallocDefault, 64}};
    for (auto& c : cases) {
        void* h = nullptr;
        double t0 = now();
        if (hipHostMalloc(&h, bytes, c.flags) != hipSuccess) { printf("%s: allocation failed\n", c.name); continue; }
        double t1 = now();
        std::vector<std::thread> ts;
        const size_t per = bytes / c.touch_threads;
        for (int t = 0; t < c.touch_threads; ++t) ts.emplace_back([=] { memset((char*)h + t * per, t + 1, per); });
        for (auto& t : ts) t.join();
        double t2 = now();
        hipMemcpy(dev, h, bytes, hipMemcpyHostToDevice);
        double best = 1e9;
        for (int r = 0; r < 3; ++r) { double a = now(); hipMemcpy(dev, h, bytes, hipMemcpyHostToDevice); best = std::min(best, now() - a); }
        printf("%-52s alloc %6.1f ms, touch %6.1f ms, H2D %5.1f GB/s, pages: %s\n", c.name, t1 - t0, t2 - t1, bytes / best * 1e-6, numa_of(h).c_str());
        hipHostFree(h);
    }
    return 0;
}
