#!/usr/bin/env python3
"""Generate tests/golden/* by running the REFERENCE's own PyTorch model definition.

Runs only in the build container (it imports /root/reference/engine/export/model.py
and export.py, which do not exist on the GPU box).  Only data is written: configs,
seeds, token ids, expected logits / intermediates, and two tiny v0 ``.bin`` files
produced by the reference's own ``export.legacy_export``.  No reference source is
copied.

Two kinds of fixture:

* ``ckpt_*.bin`` + ``ckpt_*.npz`` -- a tiny torch-initialised model
  (``model.py:226-247`` init) written by the reference exporter (``export.py:75-127``),
  with the logits the reference model gives for every prefix of a token sequence.
  Pins the checkpoint reader + forward on a file the reference itself produced.
* ``synth_*.npz`` -- larger shapes (stories15M widths, head sizes 48/64/128, one
  llama2-7B-shaped layer) whose weights are NOT stored: they are regenerated
  bit-exactly from (seed, tag, index) by the integer hash of
  ``oracle_fill_synth`` and assigned into the reference model before it is run.

Usage:  python tools/make_goldens.py            (rewrites tests/golden/)
"""
from __future__ import annotations

import os
import sys
import tempfile
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parent.parent
REF_EXPORT = Path("/root/reference/engine/export")
sys.path.insert(0, str(REF_EXPORT))
sys.path.insert(0, str(REPO))

import model as ref_model      # noqa: E402  (reference, container-only)
import export as ref_export    # noqa: E402

from oracle import oracle as O  # noqa: E402

OUT = REPO / "tests" / "golden"

# synthetic tensor tags / scales: the single definition lives in oracle.synth_spec
from oracle.synth import synth_weights  # noqa: E402


def build_ref_model(cfg: O.Config, multiple_of: int):
    args = ref_model.ModelArgs(dim=cfg.dim, n_layers=cfg.n_layers, n_heads=cfg.n_heads,
                               n_kv_heads=None, vocab_size=cfg.vocab_size,
                               hidden_dim=cfg.hidden_dim, multiple_of=multiple_of,
                               max_seq_len=cfg.seq_len, dropout=0.0)
    m = ref_model.Transformer(args)
    m.eval()
    return m


def assign_weights(m, cfg: O.Config, w: dict):
    """Copy v0-layout numpy tensors into the reference nn.Module."""
    t = lambda a: torch.from_numpy(np.array(a, dtype=np.float32, copy=True))
    with torch.no_grad():
        if not cfg.shared_weight:  # untie (model.py:213 ties them by default)
            m.output.weight = torch.nn.Parameter(t(w["wcls"]))
            m.tok_embeddings.weight = torch.nn.Parameter(t(w["token_embedding_table"]))
        else:
            m.tok_embeddings.weight.copy_(t(w["token_embedding_table"]))
        for l, layer in enumerate(m.layers):
            layer.attention_norm.weight.copy_(t(w["rms_att_weight"][l]))
            layer.attention.wq.weight.copy_(t(w["wq"][l]))
            layer.attention.wk.weight.copy_(t(w["wk"][l]))
            layer.attention.wv.weight.copy_(t(w["wv"][l]))
            layer.attention.wo.weight.copy_(t(w["wo"][l]))
            layer.ffn_norm.weight.copy_(t(w["rms_ffn_weight"][l]))
            layer.feed_forward.w1.weight.copy_(t(w["w1"][l]))
            layer.feed_forward.w2.weight.copy_(t(w["w2"][l]))
            layer.feed_forward.w3.weight.copy_(t(w["w3"][l]))
        m.norm.weight.copy_(t(w["rms_final_weight"]))


def run_prefixes(m, tokens):
    """logits[t] = reference model on tokens[:t+1] (model.py:266 returns last position)."""
    outs = []
    with torch.no_grad():
        for t in range(len(tokens)):
            x = torch.tensor([tokens[: t + 1]], dtype=torch.long)
            outs.append(m(x)[0, -1].float().numpy().copy())
    return np.stack(outs)


def capture_intermediates(m, tokens):
    """Layer-0 intermediates at the LAST position of the full sequence, by hooks."""
    cap = {}
    l0 = m.layers[0]
    hs = []
    hs.append(l0.attention_norm.register_forward_hook(lambda mod, i, o: cap.__setitem__("l0_xb_attnorm", o[0, -1])))
    hs.append(l0.attention.wq.register_forward_hook(lambda mod, i, o: cap.__setitem__("l0_q_prerope", o[0, -1])))
    hs.append(l0.attention.wk.register_forward_hook(lambda mod, i, o: cap.__setitem__("l0_k_prerope", o[0, -1])))
    hs.append(l0.attention.wv.register_forward_hook(lambda mod, i, o: cap.__setitem__("l0_v", o[0, -1])))
    hs.append(l0.attention.wo.register_forward_pre_hook(lambda mod, i: cap.__setitem__("l0_att_out", i[0][0, -1])))
    # TransformerBlock calls attention.forward() directly (model.py:199), so hook wo itself
    hs.append(l0.attention.wo.register_forward_hook(lambda mod, i, o: cap.__setitem__("l0_xb2", o[0, -1])))
    hs.append(l0.ffn_norm.register_forward_hook(lambda mod, i, o: cap.__setitem__("l0_xb_ffnnorm", o[0, -1])))
    hs.append(l0.feed_forward.w2.register_forward_pre_hook(lambda mod, i: cap.__setitem__("l0_hb", i[0][0, -1])))
    hs.append(l0.register_forward_hook(lambda mod, i, o: cap.__setitem__("l0_x_out", o[0, -1])))
    hs.append(m.norm.register_forward_hook(lambda mod, i, o: cap.__setitem__("final_x", o[0, -1])))
    with torch.no_grad():
        m(torch.tensor([tokens], dtype=torch.long))
    for h in hs:
        h.remove()
    return {k: v.float().numpy().copy() for k, v in cap.items()}


def cfg_array(cfg: O.Config):
    return np.array([cfg.dim, cfg.hidden_dim, cfg.n_layers, cfg.n_heads, cfg.n_kv_heads,
                     cfg.vocab_size, cfg.seq_len, int(cfg.shared_weight)], dtype=np.int32)


def make_ckpt_case(name, seed, shared):
    torch.manual_seed(seed)
    cfg = O.Config(dim=32, hidden_dim=96, n_layers=2, n_heads=2, n_kv_heads=2,
                   vocab_size=64, seq_len=16, shared_weight=shared)
    m = build_ref_model(cfg, multiple_of=32)
    if not shared:
        with torch.no_grad():
            m.output.weight = torch.nn.Parameter(torch.randn(cfg.vocab_size, cfg.dim) * 0.02)
            m.tok_embeddings.weight = torch.nn.Parameter(m.tok_embeddings.weight.detach().clone())
        # non-trivial norm gains so a norm-weight mix-up cannot hide
    with torch.no_grad():
        for layer in m.layers:
            layer.attention_norm.weight.add_(torch.randn(cfg.dim) * 0.1)
            layer.ffn_norm.weight.add_(torch.randn(cfg.dim) * 0.1)
        m.norm.weight.add_(torch.randn(cfg.dim) * 0.1)
    rng = np.random.default_rng(seed)
    tokens = [1] + rng.integers(2, cfg.vocab_size, size=cfg.seq_len - 1).tolist()
    logits = run_prefixes(m, tokens)          # forward BEFORE export: legacy_export negates
    inter = capture_intermediates(m, tokens)  # p.vocab_size in place for untied models
    path = OUT / f"{name}.bin"
    ref_export.legacy_export(m, str(path))
    np.savez_compressed(OUT / f"{name}.npz", cfg=cfg_array(cfg), tokens=np.array(tokens, np.int32),
                        logits=logits, **inter)
    print(name, "bin bytes", path.stat().st_size, "logits", logits.shape)


def make_v1_case(name, seed):
    """the same tiny model written by the reference's version1_export (export.py:132-180): a 256-byte
    header that starts with the magic "ak42".  The engine cannot read it (mod.rs:141-166 reads seven
    ints); the loader must reject it with an explanation instead of mis-parsing the magic as `dim`."""
    torch.manual_seed(seed)
    cfg = O.Config(dim=32, hidden_dim=96, n_layers=2, n_heads=2, n_kv_heads=2,
                   vocab_size=64, seq_len=16, shared_weight=True)
    m = build_ref_model(cfg, multiple_of=32)
    path = OUT / f"{name}.bin"
    ref_export.version1_export(m, str(path))
    print(name, "bin bytes", path.stat().st_size, "magic", path.read_bytes()[:4])


def make_synth_case(name, cfg: O.Config, seed, n_tokens, multiple_of=32):
    w = synth_weights(cfg, seed)
    m = build_ref_model(cfg, multiple_of=multiple_of)
    assign_weights(m, cfg, w)
    # RoPE tables: the reference model evaluates model.py:41-47 in torch fp32; record its
    # tables in the fixture so the oracle / GPU read the very same floats (a numpy
    # re-evaluation may differ in the last bit).
    fr = m.freqs_cos.numpy().copy()
    fi = m.freqs_sin.numpy().copy()
    dr = float(np.abs(fr - w["freq_cis_real"]).max())
    assert dr < 1e-5, dr   # oracle.synth.rope_tables agrees with the reference formula
    rng = np.random.default_rng(seed + 1000)
    tokens = [1] + rng.integers(2, cfg.vocab_size, size=n_tokens - 1).tolist()
    logits = run_prefixes(m, tokens)
    inter = capture_intermediates(m, tokens)
    np.savez_compressed(OUT / f"{name}.npz", cfg=cfg_array(cfg), seed=np.int64(seed),
                        tokens=np.array(tokens, np.int32), logits=logits,
                        freq_cis_real=fr, freq_cis_imag=fi, **inter)
    print(name, "logits", logits.shape, "max|logit|", float(np.abs(logits).max()))


def main():
    OUT.mkdir(parents=True, exist_ok=True)
    torch.set_num_threads(8)
    if len(sys.argv) > 1 and sys.argv[1] == "v1":      # only the v1 fixture (added in round 2)
        make_v1_case("ckpt_v1_ak42", seed=0)
        return
    make_v1_case("ckpt_v1_ak42", seed=0)
    make_ckpt_case("ckpt_tied", seed=0, shared=True)
    make_ckpt_case("ckpt_untied", seed=1, shared=False)
    # head_size 16, generic small
    make_synth_case("synth_d64_h4", O.Config(64, 192, 2, 4, 4, 128, 32, True), seed=0, n_tokens=32)
    # stories15M widths: dim 288 (72 float4: ragged vs a 64-lane wave), head_size 48, hidden 768
    make_synth_case("synth_d288_h6", O.Config(288, 768, 2, 6, 6, 512, 64, True), seed=1, n_tokens=24)
    # stories110M widths: dim 768, head 64, hidden 2048, untied classifier
    make_synth_case("synth_d768_h12", O.Config(768, 2048, 2, 12, 12, 320, 48, False), seed=2, n_tokens=16)
    # single head of size 128
    make_synth_case("synth_d128_h1", O.Config(128, 352, 1, 1, 1, 64, 16, False), seed=3, n_tokens=16)
    # one llama2-7B-shaped layer: dim 4096, 32 heads x 128, hidden 11008
    make_synth_case("synth_7bshape_l1", O.Config(4096, 11008, 1, 32, 32, 256, 16, False), seed=4,
                    n_tokens=8, multiple_of=256)


if __name__ == "__main__":
    main()
