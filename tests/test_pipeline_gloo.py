"""CPU (gloo, world_size 2 and 3): the layer-pipeline schedule + point-to-point exchanges of
rama_amd/pipeline.py, with the CPU oracle standing in for the per-stage compute.  The tokens
every in-flight sequence produces must equal a single-process generate() of the same model."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import oracle as O
from rama_amd.pipeline import BOS, Schedule, run_ticks, split_layers
from tests.helpers import load_case

CASE = "synth_d64_h4"
N_POS = 9


def prompts_for(n_seq, g):
    toks = g["tokens"].tolist()
    return [toks[1 + s:1 + s + (s % 3)] for s in range(n_seq)]   # lengths 0, 1, 2, 0, ...


class OracleStage:
    """pipeline backend whose stage compute is oracle_forward_range on CPU tensors"""

    def __init__(self, cfg, w, rank, world, n_seq):
        self.rank, self.world, self.cfg = rank, world, cfg
        self.lo, self.hi = split_layers(cfg.n_layers, world, rank)
        self.x_buffers = [torch.zeros(cfg.dim, dtype=torch.float32) for _ in range(n_seq)]
        self.tok_buffers = [torch.zeros(1, dtype=torch.int32) for _ in range(n_seq)]
        self.orcs = []
        for s in range(n_seq):
            o = O.Oracle(cfg, w)
            o.s["x"] = self.x_buffers[s].numpy()                  # share memory with the hand-off buffer
            o._cs = O.OracleState(*[O._p(o.s[n]) for n in O._S_FIELDS])
            self.orcs.append(o)
        self.produced = [[] for _ in range(n_seq)]

    def compute(self, seq, pos, token):
        if self.rank == 0 and token is None:
            token = int(self.tok_buffers[seq][0])
        self.orcs[seq].forward_range(int(token or 0), pos, self.lo, self.hi, self.rank == 0, self.rank == self.world - 1)
        if self.rank == self.world - 1:
            nxt = O.argmax(self.orcs[seq].s["logits"])
            self.tok_buffers[seq][0] = nxt
            self.produced[seq].append(nxt)


def _worker(rank, world, port, n_seq, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    O.lib().oracle_set_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg, w, g = load_case(CASE)
        be = OracleStage(cfg, w, rank, world, n_seq)
        sched = Schedule(world, n_seq, N_POS)
        run_ticks(sched, rank, be, 0, sched.ticks, prompts_for(n_seq, g), dist)
        dist.barrier()
        if rank == world - 1:
            q.put(be.produced)
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,n_seq", [(2, 2), (2, 3), (3, 3)])
def test_pipeline_tokens_equal_single_process(world, n_seq):
    cfg, w, g = load_case(CASE)
    if world > cfg.n_layers + 1:
        pytest.skip("more stages than layers")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_seq, q)) for r in range(world)]
    for p in procs:
        p.start()
    produced = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    prompts = prompts_for(n_seq, g)
    for s in range(n_seq):
        want = O.Oracle(cfg, w).generate_greedy(prompts[s], N_POS)
        got = produced[s]
        assert len(got) == N_POS
        # the last stage always records the argmax; generate() reports the forced token while
        # pos < len(prompt), so compare from the first free position on
        assert got[len(prompts[s]):] == want[len(prompts[s]):], (s, got, want)


def test_schedule_needs_enough_sequences():
    with pytest.raises(AssertionError):
        Schedule(4, 2, 3)      # fewer sequences than stages: a token would be needed before it exists


def test_schedule_bookkeeping():
    for world, n_seq, n_pos in [(1, 1, 4), (2, 2, 3), (4, 4, 2), (3, 5, 2), (8, 8, 3)]:
        assert n_seq >= world
        sc = Schedule(world, n_seq, n_pos)
        seen = {r: [] for r in range(world)}
        for tick in range(sc.ticks):
            sent, recvd = {}, {}
            for r in range(world):
                it = sc.item(r, tick)
                if it is not None:
                    seen[r].append((it.seq, it.pos))
                for kind, seq, peer in sc.sends(r, tick):
                    sent[(r, peer, kind, seq)] = 1
                for kind, seq, peer in sc.recvs(r, tick):
                    recvd[(peer, r, kind, seq)] = 1
            assert sent == recvd, (world, tick, sent, recvd)          # every send has its receive in the same tick
        for r in range(world):
            assert seen[r] == [(j % n_seq, j // n_seq) for j in range(n_seq * n_pos)]
        # a sequence's position p+1 is computed on rank 0 strictly after position p left the last rank
        for s in range(n_seq):
            for p in range(n_pos - 1):
                t_done = (p * n_seq + s) + (world - 1)
                t_next = (p + 1) * n_seq + s
                assert t_next > t_done or world == 1 or n_seq >= world


def test_split_layers_covers_everything():
    for L, N in [(32, 1), (32, 2), (32, 8), (12, 5), (6, 4), (2, 2)]:
        parts = [split_layers(L, N, r) for r in range(N)]
        assert parts[0][0] == 0 and parts[-1][1] == L
        assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
        assert max(h - l for l, h in parts) - min(h - l for l, h in parts) <= 1


def test_run_ticks_wraps_at_seq_len():
    """a run longer than seq_len: every slot starts a NEW generation (BOS, then its prompt) when
    its position reaches `wrap`, and positions handed to the stage stay inside [0, wrap)"""
    class Rec:
        def __init__(self):
            self.calls = []

        def compute(self, seq, pos, token):
            self.calls.append((seq, pos, token))
    be = Rec()
    sched = Schedule(1, 2, 7)
    prompts = [[11, 12], []]
    run_ticks(sched, 0, be, 0, sched.ticks, prompts, None, wrap=3)
    seq0 = [(p, t) for s, p, t in be.calls if s == 0]
    seq1 = [(p, t) for s, p, t in be.calls if s == 1]
    assert seq0 == [(0, BOS), (1, 11), (2, 12), (0, BOS), (1, 11), (2, 12), (0, BOS)]
    assert seq1 == [(0, BOS), (1, None), (2, None), (0, BOS), (1, None), (2, None), (0, BOS)]


def test_native_schedule_matches_python_schedule():
    """csrc/pipe.hip's tick -> item arithmetic (rama_pipe_item, no GPU involved) against Schedule, for
    every rank and tick; with fewer sequences than stages the native plan idles the extra slots"""
    import ctypes as C
    import rama_amd
    from rama_amd._lib import rama_pipe_plan
    from rama_amd.pipeline import Schedule
    lib = rama_amd.load()
    for world, n_seq, n_pos in [(1, 1, 5), (2, 2, 4), (3, 5, 3), (4, 4, 6), (8, 8, 3)]:
        sched = Schedule(world, n_seq, n_pos)
        plan = rama_pipe_plan(n_seq, n_pos, 0, None, 0, 0.0, 0.9, 0.0, None)
        for rank in range(world):
            for tick in range(sched.ticks + 2):
                seq, pos = C.c_int(), C.c_int()
                on = lib.rama_pipe_item(C.byref(plan), world, rank, tick, C.byref(seq), C.byref(pos))
                it = sched.item(rank, tick)
                assert on == (1 if it is not None else 0), (world, n_seq, rank, tick)
                if it is not None:
                    assert (seq.value, pos.value) == (it.seq, it.pos)
    # one sequence through four stages: a slot round is 4 ticks, position p enters rank r at tick 4 p + r
    plan = rama_pipe_plan(1, 3, 0, None, 0, 0.0, 0.9, 0.0, None)
    for rank in range(4):
        busy = []
        for tick in range(4 * 3 + 3):
            seq, pos = C.c_int(), C.c_int()
            if lib.rama_pipe_item(C.byref(plan), 4, rank, tick, C.byref(seq), C.byref(pos)) == 1:
                busy.append((tick, seq.value, pos.value))
        assert busy == [(4 * p + rank, 0, p) for p in range(3)]
