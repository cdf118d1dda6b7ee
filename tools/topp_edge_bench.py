import sys, ctypes as C, numpy as np
sys.path.insert(0, '.')
import rama_amd
from rama_amd._lib import check
dev = rama_amd.Hip(0)
n = 32000
cases = {"flat": (np.random.default_rng(0).standard_normal(n) * 0.05).astype(np.float32),
         "spike": np.concatenate(([14.0], np.zeros(n - 1))).astype(np.float32),
         "equal": np.zeros(n, np.float32),
         "std1": (np.random.default_rng(1).standard_normal(n)).astype(np.float32)}
for name, x in cases.items():
    d_x = dev.allocate(x); d_r = dev.allocate(np.zeros(1, dtype=np.float32))
    for topp in (0.9, 1.0):
        for dist in (1, 0):
            check(dev.lib.rama_set_tuning(dev.ctx, b"topp_dist", dist))
            for _ in range(3): check(dev.lib.rama_sample_topp_dev(dev.ctx, d_x.ptr, n, 1.0, topp, 0.999, d_r.ptr))
            check(dev.lib.rama_timer_start(dev.ctx))
            for _ in range(50): check(dev.lib.rama_sample_topp_dev(dev.ctx, d_x.ptr, n, 1.0, topp, 0.999, d_r.ptr))
            ms = C.c_float(); check(dev.lib.rama_timer_stop(dev.ctx, C.byref(ms)))
            tok = int(dev.download(d_r).view(np.int32)[0])
            print(name, "topp", topp, "dist", dist, round(ms.value * 1000 / 50, 1), "us, token", tok, flush=True)
    check(dev.lib.rama_set_tuning(dev.ctx, b"topp_dist", 1))
