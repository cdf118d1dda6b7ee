#!/usr/bin/env python3
"""Write the synthetic weight set of a named shape as a llama2.c v0 .bin (needs the GPU: the
weights are generated in HBM by the fill kernel and written out by rama_model_save, so the file
is what the bench / parity runs compute on).  Upstream Rama's `-m` loads it unchanged.

    python tools/make_synth_checkpoint.py stories15M /tmp/synth15M.bin --seed 0
"""
import argparse
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import rama_amd
from bench import SHAPES

ap = argparse.ArgumentParser()
ap.add_argument("shape", choices=sorted(SHAPES))
ap.add_argument("out")
ap.add_argument("--seed", type=int, default=0)
a = ap.parse_args()
d, h, L, H, V, seq, shared = SHAPES[a.shape]
cfg = rama_amd.Config(d, h, L, H, H, V, seq, shared)
dev = rama_amd.Hip(0)
m = rama_amd.Model.synth(dev, cfg, seed=a.seed)
m.save(a.out)
print(f"{a.out}: {Path(a.out).stat().st_size} bytes ({a.shape}, seed {a.seed})")
