import ctypes, time, torch
torch.cuda.init(); torch.zeros(1, device="cuda")
hip = ctypes.CDLL("libamdhip64.so")
for gib in (12, 25, 25, 25, 50):
    p = ctypes.c_void_p()
    t0 = time.perf_counter(); rc = hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(gib << 30)); t1 = time.perf_counter()
    rc2 = hip.hipMemset(p, 0, ctypes.c_size_t(gib << 30)); hip.hipDeviceSynchronize(); t2 = time.perf_counter()
    rc3 = hip.hipFree(p); t3 = time.perf_counter()
    print("hipMalloc %d GiB: %.1f ms (rc %d), memset %.1f ms, hipFree %.1f ms" % (gib, (t1 - t0) * 1e3, rc, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
