"""Time seqsum_fast.hpp's sum of 4096 squares on 1 / 2 / 4 waves (in-kernel 100 MHz stamps) and report the items walked."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import rama_amd
from rama_amd._lib import check

dev = rama_amd.Hip(0)
f = dev.lib.rama_internal_seqsum_fast
f.restype = C.c_int
f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
to = rama_amd.MutView(dev.allocate(np.zeros(4, np.float32)))
for n in (768, 4096, 11008):
    for seed in range(3):
        x = (np.random.default_rng(seed).standard_normal(n) * 0.7).astype(np.float32)
        a = (x * x).astype(np.float32)
        want = np.add.accumulate(a, dtype=np.float32)[-1]
        ta = rama_amd.MutView(dev.allocate(a))
        for nw in (1, 2, 4):
            if n > 4096 * nw:
                continue
            ticks = []
            for _ in range(20):
                check(f(dev.ctx, ta.ptr, n, nw, to.ptr))
                got = dev.download(to)
                ticks.append(got[3])
            print(f"n={n} seed={seed} waves={nw}: sum ok={got[0] == want} held={got[1]} items={int(got[2])} us={np.median(ticks) / 100:.2f} (min {min(ticks) / 100:.2f})")
