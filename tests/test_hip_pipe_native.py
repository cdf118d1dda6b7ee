"""-m gpu: the layer pipeline under the C ABI (csrc/pipe.hip: RCCL by dlopen, the native tick loop,
stage-wise checkpoint loading).  One GPU is all this box has, so the communicator has ONE rank: RCCL's
grouped send-to-self / receive-from-self still runs every line of rama_pipe_exchange (symbol
resolution, signatures, stream order), and the tick loop is checked against the oracle's generate().
The multi-rank schedule itself is covered on CPU (tests/test_pipeline_gloo.py)."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle as O
from oracle import synth as S

from .helpers import GOLDEN, load_case, to_rama_cfg

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    # PyTorch bundles its own librccl.so.1; /opt/rocm has another build with the same soname.  Whichever is
    # loaded first serves both users of the process, so torch goes first (as in the driver's whole-suite
    # order and in bench.py): torch running on the other build crashes at interpreter exit.
    import torch  # noqa: F401
    import rama_amd
    d = rama_amd.Hip(0)
    yield d
    d.close()


@pytest.fixture(scope="module")
def pipe(dev):
    from rama_amd._lib import check
    ident = (C.c_char * 128)()
    check(dev.lib.rama_pipe_unique_id(ident), "rama_pipe_unique_id")
    h = C.c_void_p()
    check(dev.lib.rama_pipe_create(dev.ctx, ident, 0, 1, C.byref(h)), "rama_pipe_create")
    yield h
    dev.lib.rama_pipe_destroy(h)


def test_exchange_with_self(dev, pipe):
    """x[dim] and a token id sent to rank 0 and received from rank 0 in one group"""
    import rama_amd
    from rama_amd._lib import check
    x = np.random.default_rng(0).standard_normal(4096).astype(np.float32)
    sx = dev.allocate(x); rx = dev.alloc(4096)
    tok = dev.allocate(np.array([12345], np.int32).view(np.float32)); rtok = dev.alloc(1)
    check(dev.lib.rama_pipe_exchange(pipe, sx.ptr, 4096, 0, rx.ptr, 4096, 0, tok.ptr, 0, rtok.ptr, 0), "rama_pipe_exchange")
    assert np.array_equal(dev.download(rx), x)
    assert dev.download(rtok).view(np.int32)[0] == 12345
    with pytest.raises(rama_amd.RamaError):
        check(dev.lib.rama_pipe_exchange(pipe, sx.ptr, 4096, 1, None, 0, 0, None, 0, None, 0))      # peer outside the communicator


@pytest.mark.parametrize("n_seq,temperature,graph,parity", [(1, 0.0, 0, 0), (3, 0.0, 0, 0), (2, 1.0, 0, 0), (3, 0.0, 1, 0), (2, 1.0, 1, 0),
                                                            (2, 0.0, 0, 1), (3, 0.0, 1, 1)])
def test_native_tick_loop_equals_generate(dev, pipe, n_seq, temperature, graph, parity):
    """rama_pipe_run_ticks on a one-rank pipe = generate() (mod.rs:169-206) for every sequence in flight:
    BOS, the forced prompt tokens, then the sampled ones; the history lands in out_tokens_dev.
    graph = 1: the stage passes are replayed from hipGraphs (one per sequence state); parity = 1: the stage passes run
    in the reference's rounding order (the mode of bench.py's headline, also for --gpus N)"""
    import rama_amd
    from rama_amd._lib import check, rama_pipe_plan, rama_run_state, rama_stage
    check(dev.lib.rama_set_tuning(dev.ctx, b"ref_order", parity))
    cfg, w, g = load_case("synth_d288_h6")
    prompt = g["tokens"].tolist()[1:4]
    n_pos = 24
    u = 0.2721174359321594
    orc = O.Oracle(cfg, w)
    want = []
    token = 1
    for pos in range(n_pos):
        lo = orc.forward(token, pos)
        nxt = int(O.sample(lo.copy(), temperature, 0.9, u))
        want.append(nxt)
        token = prompt[pos] if pos < len(prompt) else nxt
    model = rama_amd.Model.synth(dev, to_rama_cfg(cfg), int(g["seed"]), rope=(g["freq_cis_real"], g["freq_cis_imag"]))
    engines = [rama_amd.Engine(dev, model) for _ in range(n_seq)]
    states = (rama_run_state * n_seq)(*[e.state for e in engines])
    toks = [dev.alloc(1) for _ in range(n_seq)]
    tok_ptrs = (C.c_void_p * n_seq)(*[t.ptr for t in toks])
    out = dev.alloc(n_seq * n_pos)
    pr = (C.c_int32 * len(prompt))(*prompt)
    plan = rama_pipe_plan(n_seq, n_pos, 0, pr, len(prompt), temperature, 0.9, u, out.ptr)
    stage = rama_stage(0, cfg.n_layers, 1, 1)
    total = dev.lib.rama_pipe_total_ticks(pipe, C.byref(plan))
    assert total == n_seq * n_pos
    dev.lib.rama_set_graph_mode(dev.ctx, graph)      # 1: every stage pass of a sequence is a hipGraph replay (rama_forward_stage* in graph mode)
    if temperature != 0.0:      # the device sampler's scratch is sized outside the loop
        check(dev.lib.rama_sample_topp_dev(dev.ctx, engines[0].state.logits, cfg.vocab_size, temperature, 0.9, u, toks[0].ptr))
    check(dev.lib.rama_pipe_run_ticks(pipe, C.byref(model.ccfg), C.byref(model.weights), states, tok_ptrs, C.byref(stage),
                                      C.byref(plan), 0, total // 2), "rama_pipe_run_ticks")
    check(dev.lib.rama_pipe_run_ticks(pipe, C.byref(model.ccfg), C.byref(model.weights), states, tok_ptrs, C.byref(stage),
                                      C.byref(plan), total // 2, total), "rama_pipe_run_ticks")
    dev.lib.rama_set_graph_mode(dev.ctx, 0)
    check(dev.lib.rama_set_tuning(dev.ctx, b"ref_order", 0))
    hist = dev.download(out).view(np.int32).reshape(n_seq, n_pos)
    for s in range(n_seq):
        assert hist[s].tolist() == want, (s, hist[s].tolist(), want)
    for e in engines:
        e.free()
    model.free()


@pytest.mark.parametrize("name,parity", [("ckpt_tied", 0), ("ckpt_untied", 0), ("ckpt_untied", 1)])
def test_load_stage_two_stages_on_one_gpu(dev, name, parity):
    """rama_model_load_stage: the checkpoint split into two stages (each holds only its tensors) gives
    the whole model's logits when x is handed from stage 0 to stage 1 (parity = 1: bit for bit, each stage
    streaming the chain-order copy of its own layers)"""
    import rama_amd
    from rama_amd._lib import check, rama_config, rama_stage
    check(dev.lib.rama_set_tuning(dev.ctx, b"ref_order", parity))
    cfg, w, g = load_case(name)
    path = str(GOLDEN / f"{name}.bin").encode()
    mid = cfg.n_layers // 2
    models, engines = [], []
    for st in (rama_stage(0, mid, 1, 0), rama_stage(mid, cfg.n_layers, 0, 1)):
        h = C.c_void_p()
        check(dev.lib.rama_model_load_stage(dev.ctx, path, C.byref(st), C.byref(h)), "rama_model_load_stage")
        m = rama_amd.Model(dev, h, st)
        models.append(m); engines.append(rama_amd.Engine(dev, m))
    whole = rama_amd.Model.load(dev, GOLDEN / f"{name}.bin")
    assert models[0].bytes < whole.bytes and models[1].bytes < whole.bytes
    assert not models[1].weights.token_embedding_table or cfg.shared_weight
    orc = O.Oracle(cfg, w)
    toks = g["tokens"].tolist()[:6]
    for pos, t in enumerate(toks):
        lo = orc.forward(t, pos)
        engines[0].forward(t, pos)
        engines[1].set_buffer("x", engines[0].buffer("x", cfg.dim))
        engines[1].forward(t, pos)
        assert np.abs(engines[1].logits() - lo).max() <= 1e-4
        if parity:
            assert np.array_equal(engines[1].logits().view(np.uint32), lo.view(np.uint32))
    check(dev.lib.rama_set_tuning(dev.ctx, b"ref_order", 0))
    for e in engines:
        e.free()
    for m in models + [whole]:
        m.free()


def test_n1_pipeline_rehearsal_agrees_with_the_plain_n1_line():
    """`RAMA_FORCE_PIPELINE=1 python bench.py --gpus 1` runs every line of the N > 1 path on a one-rank communicator.
    Its `value` must agree with the plain N = 1 line's (the driver divides the per-N values of a SCALE run by the N = 1
    one: they have to be like for like), and it must carry the single-stream figure SURVEY 8e asks for beside the
    aggregate.  Two child processes (the bench owns its GPU context); llama2-7B, parity mode, 48 steps."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path
    repo = Path(__file__).resolve().parent.parent
    cmd = [sys.executable, str(repo / "bench.py"), "--gpus", "1", "--mode", "parity", "--steps", "48", "--warmup", "8",
           "--no-other-configs", "--no-cpu-baseline", "--no-prefill", "--no-kprof", "--no-sampled", "--no-by-position", "--no-trait-ops"]

    def run(extra_env):
        env = dict(os.environ, **extra_env)
        r = subprocess.run(cmd, cwd=repo, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        return json.loads(r.stdout.strip().splitlines()[-1])

    plain = run({})
    pipe_line = run({"RAMA_FORCE_PIPELINE": "1"})
    assert plain["config"]["mode"].startswith("parity") and pipe_line["config"]["mode"].startswith("parity")
    assert abs(pipe_line["value"] / plain["value"] - 1.0) <= 0.02, (pipe_line["value"], plain["value"])
    assert pipe_line["rccl_ranks"] == 1
    # one sequence in flight on one stage IS the whole job: the single-stream figure equals the aggregate
    assert abs(pipe_line["single_stream_tok_s"] / pipe_line["value"] - 1.0) <= 0.03, (pipe_line["single_stream_tok_s"], pipe_line["value"])
