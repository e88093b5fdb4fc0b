"""Where the host-pointer call's time outside the kernel goes: pageable -> page-locked memcpy, H2D / D2H DMA, launch + sync latency.
usage: python tools/host_copy_probe.py"""
import ctypes as C
import json
import time

import numpy as np
import torch

dev = torch.device("cuda", 0)
out = {}


def best(f, n=20):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        f()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2] * 1e6


for name, nbytes in (("q_4096", 4096 * 384), ("q_5904", 5904 * 384), ("q_10000", 10000 * 384), ("out_10000", 10000 * 120), ("q_1024", 1024 * 384)):
    src = np.random.default_rng(1).integers(0, 255, nbytes, dtype=np.uint8)
    pin = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
    dst = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    p_pin, p_src = pin.data_ptr(), src.ctypes.data
    r = {"bytes": nbytes}
    r["memcpy_pageable_to_pinned_us"] = best(lambda: C.memmove(p_pin, p_src, nbytes))
    r["memcpy_pinned_to_pageable_us"] = best(lambda: C.memmove(p_src, p_pin, nbytes))
    half = nbytes // 2

    def two_threads():
        import threading
        t = threading.Thread(target=C.memmove, args=(p_pin + half, p_src + half, nbytes - half))
        t.start()
        C.memmove(p_pin, p_src, half)
        t.join()

    r["memcpy_two_python_threads_us"] = best(two_threads)

    def h2d():
        dst.copy_(pin, non_blocking=True)
        torch.cuda.synchronize()

    def d2h():
        pin.copy_(dst, non_blocking=True)
        torch.cuda.synchronize()

    r["h2d_pinned_us"] = best(h2d)
    r["d2h_pinned_us"] = best(d2h)
    srct = torch.from_numpy(src)

    def h2d_pageable():
        dst.copy_(srct)
        torch.cuda.synchronize()

    r["h2d_pageable_us"] = best(h2d_pageable)
    out[name] = {k: round(v, 1) if isinstance(v, float) else v for k, v in r.items()}
one = torch.zeros(64, device=dev)


def launch_sync():
    one.add_(1.0)
    torch.cuda.synchronize()


out["launch_plus_sync_us"] = round(best(launch_sync, 50), 1)
print(json.dumps(out, indent=1))
