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


# ------------------------------------------------------------------ GPU-side helpers

def to_rama_cfg(cfg: O.Config):
    import rama_amd
    return rama_amd.Config(cfg.dim, cfg.hidden_dim, cfg.n_layers, cfg.n_heads, cfg.n_kv_heads,
                           cfg.vocab_size, cfg.seq_len, cfg.shared_weight)


def gpu_views(dev, cfg: O.Config, w: dict):
    """Upload numpy weights tensor by tensor (hbm.rs:55-90) and allocate a zeroed RunState."""
    import rama_amd
    rcfg = to_rama_cfg(cfg)
    ws = rama_amd.TransformerWeights.from_numpy(rcfg, w, dev)
    rs = rama_amd.RunState.from_config(rcfg, dev)
    return rcfg, ws, rama_amd.TransformerWeightsView.from_gpu_ws(ws), rs, rama_amd.RunStateView.from_rs(rs)


def synth_at(indices, seed, tag, scale, bias=0.0):
    """oracle synthetic generator evaluated at arbitrary flat indices (numpy restatement)."""
    idx = np.asarray(indices, dtype=np.uint64)
    M = (1 << 64) - 1
    base = np.uint64((tag * 0x9E3779B97F4A7C15 + seed * 0xD1B54A32D192ED03) & M)
    with np.errstate(over="ignore"):
        z = idx + base
        z ^= z >> np.uint64(30); z *= np.uint64(0xBF58476D1CE4E5B9)
        z ^= z >> np.uint64(27); z *= np.uint64(0x94D049BB133111EB)
        z ^= z >> np.uint64(31)
    m = np.uint64(0xFFFF)
    s = ((z & m) + ((z >> np.uint64(16)) & m) + ((z >> np.uint64(32)) & m) + (z >> np.uint64(48))).astype(np.int64)
    return (np.float32(bias) + (s - 131070).astype(np.float32) * np.float32(scale)).astype(np.float32)
