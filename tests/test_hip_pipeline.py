"""-m gpu: the pipeline's HIP backend (rama_amd.pipeline.HipStage) on ONE device: two and
three stages mapped onto the same GPU, hand-offs done by device-to-device copies where RCCL
would send/recv.  Tokens of every in-flight sequence must equal the oracle's generate()."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.helpers import load_case, to_rama_cfg

pytestmark = pytest.mark.gpu


def drive(stages, sched, prompts):
    """run_ticks for all ranks of one process; exchanges become tensor copies"""
    from rama_amd.pipeline import BOS
    world = sched.world
    produced = [[] for _ in range(sched.n_seq)]
    for tick in range(sched.ticks):
        for r in range(world):
            it = sched.item(r, tick)
            if it is None:
                continue
            token = None
            if r == 0:
                p = prompts[it.seq]
                token = BOS if it.pos == 0 else (p[it.pos - 1] if it.pos <= len(p) else None)
            stages[r].compute(it.seq, it.pos, token)
            if r == world - 1:
                produced[it.seq].append(int(stages[r].tok_buffers[it.seq].item()))
        for r in range(world):
            for kind, seq, peer in sched.sends(r, tick):
                src = stages[r].x_buffers[seq] if kind == "x" else stages[r].tok_buffers[seq]
                dst = stages[peer].x_buffers[seq] if kind == "x" else stages[peer].tok_buffers[seq]
                dst.copy_(src)
    return produced


@pytest.mark.parametrize("name,world,n_seq", [("synth_d64_h4", 1, 1), ("synth_d64_h4", 2, 2),
                                               ("synth_d288_h6", 2, 3), ("synth_d768_h12", 2, 2)])
def test_hipstage_pipeline_tokens(name, world, n_seq):
    import torch
    from rama_amd.pipeline import HipStage, Schedule
    cfg, w, g = load_case(name)
    rcfg = to_rama_cfg(cfg)
    rope = (g["freq_cis_real"], g["freq_cis_imag"])
    n_pos = min(cfg.seq_len, 12)
    toks = g["tokens"].tolist()
    prompts = [toks[1 + s:1 + s + (s % 3)] for s in range(n_seq)]
    stream = torch.cuda.Stream()      # all stages of this one-process rehearsal share a stream
    stages = [HipStage(rcfg, r, world, 0, n_seq, seed=int(g["seed"]), rope=rope, torch_stream=stream)
              for r in range(world)]
    produced = drive(stages, Schedule(world, n_seq, n_pos), prompts)
    torch.cuda.synchronize()
    for s in range(n_seq):
        want = O.Oracle(cfg, w).generate_greedy(prompts[s], n_pos)
        assert produced[s][len(prompts[s]):] == want[len(prompts[s]):], (s, produced[s], want)
    for st in stages:
        st.free()
