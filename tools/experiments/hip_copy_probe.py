import ctypes, time, numpy as np, torch
torch.cuda.init()
hip = ctypes.CDLL("libamdhip64.so")
n = 134217728
def t(f, reps=5):
    f(); ts=[]
    for _ in range(reps):
        t0=time.perf_counter(); f(); ts.append((time.perf_counter()-t0)*1e3)
    return min(ts)
p = ctypes.c_void_p()
def mf():
    hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(n)); hip.hipFree(p)
print("hipMalloc+hipFree 134 MB: %.3f ms" % t(mf))
hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(n))
x = np.random.default_rng(1).integers(0, 256, size=n, dtype=np.uint8)
xp = torch.from_numpy(x.copy()).pin_memory().numpy()
for name, buf in (("pageable", x), ("pinned", xp)):
    print(name, "H2D %.2f ms" % t(lambda: hip.hipMemcpy(p, buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(n), 1)),
          "D2H %.2f ms" % t(lambda: hip.hipMemcpy(buf.ctypes.data_as(ctypes.c_void_p), p, ctypes.c_size_t(n), 2)))
