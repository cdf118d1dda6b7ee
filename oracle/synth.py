"""Synthetic v0-layout weights for the oracle side (TEST INFRASTRUCTURE ONLY).

Real checkpoints (stories15M/110M, llama2-7B) are not available offline, so every
size is exercised on synthetic weights.  The generator is integer-only up to one
int->float convert, one multiply and one add (oracle_fill_synth), so this module
(C via ctypes), a numpy restatement and the product's HIP fill kernel all produce
the same bits; tests check that.

Spec (tag, std, bias) -- stds follow the reference's init, engine/export/model.py:
232-247: matrices N(0, 0.02), wo / w3 N(0, 0.02/sqrt(2L)); norm gains are 1 + noise
so a mixed-up gain cannot hide behind all-ones.
"""
from __future__ import annotations

import math

import numpy as np

from . import oracle as O

IH4_STD = math.sqrt(4.0 * (65536.0 ** 2 - 1.0) / 12.0)   # std of the 4 x u16 sum

TAGS = dict(token_embedding_table=1, rms_att_weight=2, wq=3, wk=4, wv=5, wo=6,
            rms_ffn_weight=7, w1=8, w2=9, w3=10, rms_final_weight=11, wcls=12)


def synth_spec(cfg: O.Config):
    """name -> (tag, scale(float32), bias(float32))"""
    res = 0.02 / math.sqrt(2.0 * cfg.n_layers)
    std = dict(token_embedding_table=0.02, wq=0.02, wk=0.02, wv=0.02, wo=res,
               w1=0.02, w2=0.02, w3=res, wcls=0.02,
               rms_att_weight=0.05, rms_ffn_weight=0.05, rms_final_weight=0.05)
    out = {}
    for name, tag in TAGS.items():
        bias = 1.0 if name.startswith("rms_") else 0.0
        out[name] = (tag, np.float32(std[name] / IH4_STD), np.float32(bias))
    return out


def rope_tables(seq_len: int, head_size: int):
    """cos/sin(t * 10000^(-2i/hs)), the table of engine/export/model.py:41-47."""
    i = np.arange(0, head_size, 2, dtype=np.float64)[: head_size // 2]
    freqs = 1.0 / (10000.0 ** (i / head_size))
    ang = np.outer(np.arange(seq_len, dtype=np.float64), freqs)
    return np.cos(ang).astype(np.float32), np.sin(ang).astype(np.float32)


def fill_numpy(n: int, seed: int, tag: int, scale, bias=0.0, offset: int = 0) -> np.ndarray:
    """numpy restatement of oracle_fill_synth (uint64 wrap-around arithmetic)."""
    M = (1 << 64) - 1
    base = np.uint64((offset + tag * 0x9E3779B97F4A7C15 + seed * 0xD1B54A32D192ED03) & M)
    with np.errstate(over="ignore"):
        z = np.arange(n, dtype=np.uint64) + base
        z ^= z >> np.uint64(30); z *= np.uint64(0xBF58476D1CE4E5B9)
        z ^= z >> np.uint64(27); z *= np.uint64(0x94D049BB133111EB)
        z ^= z >> np.uint64(31)
    m = np.uint64(0xFFFF)
    s = ((z & m) + ((z >> np.uint64(16)) & m) + ((z >> np.uint64(32)) & m) + (z >> np.uint64(48))).astype(np.int64)
    v = (s - 131070).astype(np.float32) * np.float32(scale)
    return (np.float32(bias) + v).astype(np.float32)


def synth_weights(cfg: O.Config, seed: int, rope=None) -> dict:
    spec = synth_spec(cfg)
    w = {}
    for name, shp in O.weight_shapes(cfg):
        if name.startswith("freq_cis"):
            continue
        tag, scale, bias = spec[name]
        w[name] = O.fill_synth(int(np.prod(shp)), seed, tag, scale, bias).reshape(shp)
    fr, fi = rope if rope is not None else rope_tables(cfg.seq_len, cfg.head_size)
    w["freq_cis_real"], w["freq_cis_imag"] = fr, fi
    if cfg.shared_weight:
        w["wcls"] = w["token_embedding_table"]
    return w
