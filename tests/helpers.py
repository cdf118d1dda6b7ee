"""Shared test helpers: golden-fixture loading and oracle construction."""
from __future__ import annotations

from pathlib import Path

import numpy as np

from oracle import oracle as O
from oracle import synth as S

GOLDEN = Path(__file__).resolve().parent / "golden"

CKPT_CASES = ["ckpt_tied", "ckpt_untied"]
SYNTH_CASES = ["synth_d64_h4", "synth_d288_h6", "synth_d768_h12", "synth_d128_h1"]
BIG_SYNTH_CASES = ["synth_7bshape_l1"]

# Tolerances vs the reference's own PyTorch model (different summation order from
# the Rust loop: torch BLAS / SDPA).  north_star bar: 1e-4 absolute on logits.
LOGIT_ATOL = 1e-4


def cfg_from_array(a) -> O.Config:
    a = [int(v) for v in a]
    return O.Config(a[0], a[1], a[2], a[3], a[4], a[5], a[6], bool(a[7]))


def load_case(name):
    """-> (cfg, weights dict of numpy arrays, golden npz)"""
    g = np.load(GOLDEN / f"{name}.npz")
    if name.startswith("ckpt_"):
        cfg, w = O.read_checkpoint(GOLDEN / f"{name}.bin")
        gc = cfg_from_array(g["cfg"])
        assert cfg == gc, (cfg, gc)
    else:
        cfg = cfg_from_array(g["cfg"])
        w = S.synth_weights(cfg, int(g["seed"]), rope=(g["freq_cis_real"], g["freq_cis_imag"]))
    return cfg, w, g
