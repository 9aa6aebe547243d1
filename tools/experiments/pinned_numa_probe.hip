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
