"""Layer pipeline over N ranks (one process per GPU, torch.distributed; backend "nccl" is RCCL
over xGMI on ROCm, "gloo" on CPU for tests).  New functionality: the reference is single-device
(SURVEY.md section 8e).

Partition: contiguous layer ranges, as even as possible (the first L mod N ranks hold one layer more:
split_layers), each with its weights and KV slabs; rank 0 also owns
the embedding table, the last rank the final norm + classifier.  The only exchanges are
point-to-point: the residual x[dim] (16 KiB at llama2-7B) from rank r to r+1 and the sampled token
id (4 bytes) from the last rank back to rank 0 -- no collective.

Batch-1 decode is sequential in the layers, so ONE sequence gains nothing from a pipeline.  The
schedule therefore keeps N sequences in flight: work item j = (sequence j % N, position j // N);
rank r processes item tau - r at tick tau, so in steady state every rank is busy every tick and one
token leaves the pipe per tick.  After computing, a rank posts ONE grouped exchange
(batch_isend_irecv: its output to the next rank + the receive of its next input), which is
deadlock-free for any N including N = 2, where both directions share one peer.

`Schedule` is pure bookkeeping and `run_pipeline` only needs a backend object with
x_buffers / tok_buffers / compute(); tests drive it with gloo and the CPU oracle as the backend.

Two data paths run the same schedule on GPUs:
  * native (default): csrc/pipe.hip under the C ABI -- RCCL ncclSend/ncclRecv issued by the library on the
    context's stream, the tick loop in C++ (rama_pipe_run_ticks), no host synchronisation per tick.
    torch.distributed (gloo) only carries the 128-byte RCCL unique id and the barriers of the bench.
  * torch (RAMA_TORCH_PIPE=1, or when the native end cannot be created): the exchanges as
    torch.distributed P2P ops (backend "nccl" = RCCL), the tick loop in Python.
"""
from __future__ import annotations

import ctypes as C
import os
import time
from dataclasses import dataclass
from typing import List, Optional

BOS = 1


@dataclass
class Item:
    seq: int
    pos: int


class Schedule:
    """Which item a rank computes at a tick, and what it sends / receives afterwards."""

    def __init__(self, world: int, n_seq: int, n_pos: int):
        assert n_seq >= 1 and world >= 1 and n_pos >= 1
        # item j leaves the last rank at tick j + world - 1 and its successor j + n_seq enters
        # rank 0 at tick j + n_seq: the sampled token exists in time iff n_seq >= world
        assert n_seq >= world, "a pipeline of `world` stages needs at least `world` sequences in flight"
        self.world, self.n_seq, self.n_pos = world, n_seq, n_pos
        self.total = n_seq * n_pos
        self.ticks = self.total + world - 1

    def item(self, rank: int, tick: int) -> Optional[Item]:
        j = tick - rank
        if 0 <= j < self.total:
            return Item(j % self.n_seq, j // self.n_seq)
        return None

    def sends(self, rank: int, tick: int):
        """-> list of (kind, seq, peer): what `rank` sends after computing at `tick`."""
        it = self.item(rank, tick)
        if it is None or self.world == 1:
            return []
        if rank < self.world - 1:
            return [("x", it.seq, rank + 1)]
        return [("tok", it.seq, 0)]

    def recvs(self, rank: int, tick: int):
        """-> list of (kind, seq, peer): what `rank` receives in the exchange of `tick`."""
        if self.world == 1:
            return []
        src = rank - 1 if rank > 0 else self.world - 1
        it = self.item(src, tick)
        if it is None:
            return []
        return [("x" if rank > 0 else "tok", it.seq, src)]


def split_layers(n_layers: int, world: int, rank: int):
    """contiguous, as even as possible; earlier ranks get the remainder"""
    base, rem = divmod(n_layers, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def run_ticks(sched: Schedule, rank: int, backend, tick_from: int, tick_to: int, prompts, dist=None, wrap: int = 0):
    """Run ticks [tick_from, tick_to).  backend.compute(seq, pos, token) runs this rank's stage:
    token is an int (BOS / forced prompt token, rank 0 only) or None (= use tok_buffers[seq] on
    rank 0; ignored elsewhere).  After the last rank's compute tok_buffers[seq] holds the argmax.
    wrap > 0: a sequence that reaches position `wrap` (the model's seq_len) starts over at position 0
    with BOS and its prompt -- a new generation in the same slot, so a run may be longer than seq_len."""
    for tick in range(tick_from, tick_to):
        it = sched.item(rank, tick)
        if it is not None:
            pos = it.pos % wrap if wrap else it.pos
            token = None
            if rank == 0:
                p = prompts[it.seq]
                if pos == 0:
                    token = BOS                       # mod.rs:182
                elif pos <= len(p):
                    token = p[pos - 1]                # mod.rs:190-191 forced prompt token
            backend.compute(it.seq, pos, token)
        if dist is None or sched.world == 1:
            if sched.world == 1 and it is not None:
                pass   # single rank: tok_buffers[seq] already holds the next token
            continue
        ops = []
        for kind, seq, peer in sched.sends(rank, tick):
            buf = backend.x_buffers[seq] if kind == "x" else backend.tok_buffers[seq]
            ops.append(dist.P2POp(dist.isend, buf, peer))
        for kind, seq, peer in sched.recvs(rank, tick):
            buf = backend.x_buffers[seq] if kind == "x" else backend.tok_buffers[seq]
            ops.append(dist.P2POp(dist.irecv, buf, peer))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()


# ------------------------------------------------------------------ the HIP backend

class HipStage:
    """This rank's layer range on its GPU, one run state per in-flight sequence."""

    def __init__(self, cfg, rank: int, world: int, local_rank: int, n_seq: int, seed: int = 0, rope=None,
                 torch_stream=None):
        import torch
        import rama_amd
        from rama_amd._lib import check, rama_run_state, rama_stage
        self.torch, self.check = torch, check
        self.cfg, self.rank, self.world = cfg, rank, world
        torch.cuda.set_device(local_rank)
        # Kernels, tensor copies and the RCCL ops torch issues must share ONE stream order.  The
        # legacy default stream has handle 0, which rama_ctx_create reads as "make your own", so
        # run on an explicit torch stream, make it current (ProcessGroupNCCL orders its p2p ops
        # against the current stream) and let the context adopt its handle.
        self.stream = torch_stream if torch_stream is not None else torch.cuda.Stream(torch.device("cuda", local_rank))
        torch.cuda.set_stream(self.stream)
        assert self.stream.cuda_stream != 0
        self.dev = rama_amd.Hip(local_rank, stream=self.stream.cuda_stream)
        lo, hi = split_layers(cfg.n_layers, world, rank)
        self.stage = rama_stage(lo, hi, int(rank == 0), int(rank == world - 1))
        self.model = rama_amd.Model.synth(self.dev, cfg, seed, self.stage, rope)
        device = torch.device("cuda", local_rank)
        self.x_buffers = [torch.zeros(cfg.dim, dtype=torch.float32, device=device) for _ in range(n_seq)]
        self.tok_buffers = [torch.zeros(1, dtype=torch.int32, device=device) for _ in range(n_seq)]
        self.states, self._blob = [], []
        for s in range(n_seq):
            st = rama_run_state()
            check(self.dev.lib.rama_state_create(self.dev.ctx, C.byref(self.model.ccfg), hi - lo, C.byref(st)))
            self._blob.append(st.x)                       # blob base, needed to free
            st.x = self.x_buffers[s].data_ptr()           # the hand-off buffer IS the stage's x
            self.states.append(st)
        self.dev.sync()

    def compute(self, seq: int, pos: int, token):
        L, st = self.dev.lib, self.states[seq]
        if token is not None:
            self.check(L.rama_forward_stage(self.dev.ctx, C.byref(self.model.ccfg), C.byref(self.model.weights),
                                            C.byref(st), int(token), pos, C.byref(self.stage)), "rama_forward_stage")
        else:
            tok_ptr = self.tok_buffers[seq].data_ptr() if self.rank == 0 else None
            self.check(L.rama_forward_stage_devtok(self.dev.ctx, C.byref(self.model.ccfg), C.byref(self.model.weights),
                                                   C.byref(st), tok_ptr, pos, C.byref(self.stage)),
                       "rama_forward_stage_devtok")
        if self.rank == self.world - 1:
            self.check(L.rama_argmax_dev(self.dev.ctx, st.logits, self.cfg.vocab_size,
                                         self.tok_buffers[seq].data_ptr()), "rama_argmax_dev")

    def free(self):
        for st, blob in zip(self.states, self._blob):
            st.x = blob
            self.dev.lib.rama_state_free(self.dev.ctx, C.byref(st))
        self.model.free()
        self.dev.close()


class _stdout_to_stderr:
    """route file descriptor 1 to stderr for a while (native libraries that print banners on stdout)"""

    def __enter__(self):
        import sys
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


class NativeStage:
    """This rank's end of the native pipe: its stage of the model, one run state and one device token
    word per in-flight sequence, the RCCL communicator (csrc/pipe.hip)."""

    def __init__(self, cfg, rank: int, world: int, local_rank: int, n_seq: int, ident: bytes, seed: int = 0, rope=None):
        import rama_amd
        from rama_amd._lib import check, rama_run_state, rama_stage
        self.check = check
        self.cfg, self.rank, self.world, self.n_seq = cfg, rank, world, n_seq
        self.dev = rama_amd.Hip(local_rank)
        lo, hi = split_layers(cfg.n_layers, world, rank)
        self.stage = rama_stage(lo, hi, int(rank == 0), int(rank == world - 1))
        self.model = rama_amd.Model.synth(self.dev, cfg, seed, self.stage, rope)
        self.states = (rama_run_state * n_seq)()
        for s in range(n_seq):
            check(self.dev.lib.rama_state_create(self.dev.ctx, C.byref(self.model.ccfg), hi - lo, C.byref(self.states[s])))
        self.toks = [self.dev.alloc(1) for _ in range(n_seq)]
        self.tok_ptrs = (C.c_void_p * n_seq)(*[t.ptr for t in self.toks])
        self.pipe = C.c_void_p()
        buf = (C.c_char * 128).from_buffer_copy(ident)
        check(self.dev.lib.rama_pipe_create(self.dev.ctx, buf, rank, world, C.byref(self.pipe)), "rama_pipe_create")
        if not os.environ.get("RAMA_PIPE_EAGER"):      # every stage pass of a sequence is one hipGraph replay (rama_forward_stage* in graph mode)
            check(self.dev.lib.rama_set_graph_mode(self.dev.ctx, 1), "rama_set_graph_mode")
        self.dev.sync()

    @staticmethod
    def unique_id(lib) -> bytes:
        from rama_amd._lib import check
        buf = (C.c_char * 128)()
        check(lib.rama_pipe_unique_id(buf), "rama_pipe_unique_id")
        return bytes(buf.raw)

    def plan(self, n_pos: int, prompt, wrap: int = 0, temperature: float = 0.0, topp: float = 0.9, u: float = 0.0, out=None, n_seq=None):
        """n_seq: sequences in flight (default: all this stage holds states for); 1 = a single stream, N - 1 of N ticks idle per stage"""
        from rama_amd._lib import rama_pipe_plan
        self._prompt = (C.c_int32 * max(len(prompt), 1))(*prompt)
        return rama_pipe_plan(self.n_seq if n_seq is None else n_seq, n_pos, wrap, self._prompt, len(prompt), temperature, topp, u, out)

    def total_ticks(self, plan) -> int:
        return self.dev.lib.rama_pipe_total_ticks(self.pipe, C.byref(plan))

    def run_ticks(self, plan, tick_from: int, tick_to: int):
        self.check(self.dev.lib.rama_pipe_run_ticks(self.pipe, C.byref(self.model.ccfg), C.byref(self.model.weights), self.states,
                                                    self.tok_ptrs, C.byref(self.stage), C.byref(plan), tick_from, tick_to),
                   "rama_pipe_run_ticks")

    def free(self):
        self.dev.lib.rama_pipe_destroy(self.pipe)
        for s in range(self.n_seq):
            self.dev.lib.rama_state_free(self.dev.ctx, C.byref(self.states[s]))
        self.model.free()
        self.dev.close()


def _bench_line(args, cfg, world, n_seq, tok_s, dt, roofline, path, hipgraph, rccl_ranks=None, mode="fast"):
    import rama_amd
    from bench import HBM_PEAK_GBPS
    bytes_ = rama_amd.algorithmic_bytes(cfg)
    return {
        "metric": f"tokens/sec decode + matvec achieved-HBM-GB/s vs roofline, {args.config} fp32 layer pipeline over {world}xMI355X"
                  if world > 1 else "tokens/sec decode + matvec achieved-HBM-GB/s vs roofline, llama2-7B fp32 1xMI355X",
        "value": round(tok_s, 3), "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "settle_s": getattr(args, "settle_s", 0.0),
        "ms_per_step": round(dt * 1e3 / args.steps, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.config} fp32 decode, layer pipeline over {world} GPUs, {n_seq} sequences in flight, greedy",
                   "mode": {"tol": "tolerance (experiment): every stage runs the chain-order matvecs in the reference CPU path's rounding order with tree-summed norms folded in and the fast attention",
                            "parity": "parity: every op of every stage in the reference CPU path's rounding order (chain-order weight copies), as the N = 1 line",
                            "fast": "fast: fused multiply-adds, tree-shaped sums (the pipeline stages run the default kernels)"}[mode],
                   "dim": cfg.dim, "hidden_dim": cfg.hidden_dim, "n_layers": cfg.n_layers, "n_heads": cfg.n_heads,
                   "vocab_size": cfg.vocab_size, "seq_len": cfg.seq_len, "sequences_in_flight": n_seq,
                   "parallelism": f"pp{world} (RCCL send/recv of x[dim] and the token id; {path})", "hipgraph": bool(hipgraph)},
        "rccl_ranks": rccl_ranks,
        "token_level": {"algorithmic_bytes_per_token": bytes_["token"],
                        "achieved_GBps_aggregate": round(bytes_["token"] * tok_s / 1e9, 1),
                        "frac_of_aggregate_8TBps": round(bytes_["token"] * tok_s / 1e9 / (HBM_PEAK_GBPS * world), 4)},
        "roofline": roofline, "cpu_baseline": None,   # cpu_baseline: rank 0 at N = 1 only (bench.py)
    }


def _stage_roofline(check, lib, ctx, compute, n_local, cfg, mode="fast"):
    """dominant kernel (W1|W3 SwiGLU matvec) of this rank's stage, event-bracketed per launch; traffic from
    the committed PMC pass of the same kernel (bench.pmc_traffic)"""
    import rama_amd
    from bench import HBM_PEAK_GBPS, pmc_traffic
    bytes_ = rama_amd.algorithmic_bytes(cfg)
    reps = 8
    check(lib.rama_kprof_enable(ctx, 3, reps * n_local))
    for _ in range(reps):
        compute()
    n, tot = C.c_int(), C.c_double()
    check(lib.rama_kprof_read(ctx, C.byref(n), C.byref(tot)))
    if not n.value:
        return None
    avg_ms = tot.value / n.value
    a = bytes_["w13"] / (avg_ms * 1e-3) / 1e9
    kname = {"tol": "gemv_chain_kernel<1, 16, 4, 3, 2>", "parity": "gemv_chain_kernel<1, 16, 4, 3, 0>", "fast": "gemv_rows<4, 2, 8, true, 5>"}[mode]
    traffic, src = pmc_traffic(kname) if cfg.dim == 4096 else (None, None)
    return {"bound": "hbm", "kernel": {"tol": "tree-summed rmsnorm + chain-order W1|W3 matvec in the reference's rounding order + SiLU*gate",
                                       "parity": "chain-order W1|W3 matvec in the reference's rounding order + SiLU*gate",
                                       "fast": "rmsnorm + W1|W3 matvec + SiLU*gate"}[mode] + ", rank 0's stage",
            "achieved": round(a, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(a / HBM_PEAK_GBPS, 4),
            "traffic": traffic, "traffic_source": src,
            "algorithmic_bytes_per_launch": bytes_["w13"], "avg_launch_us": round(avg_ms * 1e3, 2)}


def _rendezvous_env():
    """MASTER_ADDR / MASTER_PORT come from the launcher (torch.distributed.run, or bench.py's own spawn_ranks, which
    picks a free port); only a single rank rehearsing the path (RAMA_FORCE_PIPELINE) has to pick them itself.
    NCCL_DEBUG: the user's value stands unless it is one of the two levels that print a banner on stdout."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:
        import socket
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s_:
            s_.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(s_.getsockname()[1])
    cur = os.environ.get("NCCL_DEBUG", "")
    if "RAMA_NCCL_DEBUG" in os.environ:
        os.environ["NCCL_DEBUG"] = os.environ["RAMA_NCCL_DEBUG"]
    elif cur.upper() in ("", "VERSION", "WARN"):
        os.environ["NCCL_DEBUG"] = "NONE"


def run_pipeline_bench_native(args, cfg, rank: int, world: int, local_rank: int) -> dict:
    """bench.py --gpus N > 1 on the native path: N sequences in flight, a step = N ticks."""
    import torch
    import torch.distributed as dist
    import rama_amd
    from bench import PROMPT

    _rendezvous_env()
    torch.cuda.set_device(local_rank)  # torch.cuda.synchronize() below must mean THIS rank's GPU, not device 0 for everybody
    if not dist.is_initialized():      # control plane only: the id, barriers, the max over ranks
        with _stdout_to_stderr():      # gloo announces its connections on stdout; the bench prints ONE JSON line there
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    box = [NativeStage.unique_id(rama_amd.load()) if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    n_seq = world
    n_pos = args.warmup + args.steps
    st = NativeStage(cfg, rank, world, local_rank, n_seq, box[0], seed=0)
    # the headline mode is the N = 1 line's: parity (every stage in the reference's rounding order) unless --mode says otherwise;
    # with --mode both / all the other modes are timed too and reported beside it
    from bench import REF_ORDER
    want = getattr(args, "mode", "both")
    modes = {"all": ["parity", "fast", "tol"], "both": ["parity", "fast"]}.get(want, [want])
    t_warm, t_end = args.warmup * n_seq, (args.warmup + args.steps) * n_seq
    timed = {}
    for mode in modes:
        st.check(st.dev.lib.rama_set_tuning(st.dev.ctx, b"ref_order", REF_ORDER[mode]), "rama_set_tuning")
        if mode == modes[0] and getattr(args, "settle_s", 0.0) > 0:
            # the parts' clocks first (bench.py --settle-s): whole untimed generations over the same positions, as many on every rank (the first one's
            # duration, the maximum over the ranks, says how many more)
            import math
            n_more = 0
            for i in range(21):
                plan0 = st.plan(n_pos, PROMPT, wrap=cfg.seq_len)
                ts = time.perf_counter()
                st.run_ticks(plan0, 0, st.total_ticks(plan0))
                st.dev.sync(); torch.cuda.synchronize()
                if i == 0:
                    d0 = torch.tensor([time.perf_counter() - ts], dtype=torch.float64)
                    dist.all_reduce(d0, op=dist.ReduceOp.MAX)
                    n_more = min(20, max(0, math.ceil(args.settle_s / max(float(d0.item()), 1e-3)) - 1))
                if i >= n_more:
                    break
        plan = st.plan(n_pos, PROMPT, wrap=cfg.seq_len)
        total = st.total_ticks(plan)
        st.run_ticks(plan, 0, t_warm)
        st.dev.sync(); torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        st.run_ticks(plan, t_warm, t_end)
        st.dev.sync(); torch.cuda.synchronize()
        dist.barrier()
        dt = time.perf_counter() - t0
        st.run_ticks(plan, t_end, total)      # drain, untimed
        st.dev.sync()
        tmax = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        timed[mode] = float(tmax.item())
    head = modes[0]
    dt = timed[head]
    st.check(st.dev.lib.rama_set_tuning(st.dev.ctx, b"ref_order", REF_ORDER[head]), "rama_set_tuning")
    # SURVEY 8e asks for both figures: the aggregate over the N sequences in flight (`value`) and what ONE sequence gets
    # from the pipeline -- its stages take turns, N - 1 of N ticks idle each, plus a hop per boundary: no faster than N = 1
    single_dt = None
    if world > 1 or os.environ.get("RAMA_FORCE_PIPELINE"):
        ss_steps = max(8, min(args.steps, 32))
        S1 = max(1, world)                                    # ticks per position with one sequence
        plan1 = st.plan(args.warmup + ss_steps, PROMPT, wrap=cfg.seq_len, n_seq=1)
        total1 = st.total_ticks(plan1)
        st.run_ticks(plan1, 0, args.warmup * S1)
        st.dev.sync(); torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        st.run_ticks(plan1, args.warmup * S1, (args.warmup + ss_steps) * S1)
        st.dev.sync(); torch.cuda.synchronize()
        dist.barrier()
        d1 = time.perf_counter() - t0
        st.run_ticks(plan1, (args.warmup + ss_steps) * S1, total1)
        st.dev.sync()
        t1 = torch.tensor([d1], dtype=torch.float64)
        dist.all_reduce(t1, op=dist.ReduceOp.MAX)
        single_dt = (float(t1.item()), ss_steps)
    # ... and the aggregate with 200 / 1 000 / 1 900 positions of context in front (bench.py by_position, the same points as the N = 1 line): ONE
    # generation of the N sequences to position 1 932, three timed windows of 32 positions on the way (barrier + synchronize on either side of each)
    bypos = None
    if not getattr(args, "no_by_position", False) and cfg.seq_len >= 512:
        points, win = [p for p in (200, 1000, 1900) if p + 32 <= cfg.seq_len], 32
        planp = st.plan(points[-1] + win, PROMPT, wrap=cfg.seq_len)
        totalp = st.total_ticks(planp)
        bypos, at = {}, 0
        for p in points:
            st.run_ticks(planp, at, p * n_seq)
            st.dev.sync(); torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            st.run_ticks(planp, p * n_seq, (p + win) * n_seq)
            st.dev.sync(); torch.cuda.synchronize()
            dist.barrier()
            tp = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
            dist.all_reduce(tp, op=dist.ReduceOp.MAX)
            at = (p + win) * n_seq
            bypos[str(p)] = {"tok_s": round(win * n_seq / float(tp.item()), 2), "ms_per_step": round(float(tp.item()) * 1e3 / win, 4), "positions": f"{p}..{p + win - 1}"}
        st.run_ticks(planp, at, totalp)       # drain, untimed
        st.dev.sync()
    roofline = None
    n_local = st.stage.layer_end - st.stage.layer_begin
    if n_local > 0 and not args.no_kprof:
        from rama_amd._lib import rama_run_state
        L = st.dev.lib

        def one():
            if rank == 0:
                st.check(L.rama_forward_stage(st.dev.ctx, C.byref(st.model.ccfg), C.byref(st.model.weights), C.byref(st.states[0]), BOS, 5, C.byref(st.stage)))
            else:
                st.check(L.rama_forward_stage_devtok(st.dev.ctx, C.byref(st.model.ccfg), C.byref(st.model.weights), C.byref(st.states[0]), None, 5, C.byref(st.stage)))
        roofline = _stage_roofline(st.check, L, st.dev.ctx, one, n_local, cfg, head)
    nr, rk = C.c_int(), C.c_int()
    st.check(st.dev.lib.rama_pipe_comm_info(st.pipe, C.byref(nr), C.byref(rk)), "rama_pipe_comm_info")
    assert nr.value == world and rk.value == rank, (nr.value, rk.value, world, rank)
    graphs = not os.environ.get("RAMA_PIPE_EAGER")
    line = _bench_line(args, cfg, world, n_seq, args.steps * n_seq / dt, dt, roofline,
                       "native: csrc/pipe.hip, tick loop in C++, one hipGraph per (sequence, stage)" if graphs else "native: csrc/pipe.hip, tick loop in C++, eager launches",
                       graphs, rccl_ranks=nr.value, mode=head)
    for m_ in modes:
        line[("tolerance" if m_ == "tol" else m_) + "_mode"] = {"tok_s": round(args.steps * n_seq / timed[m_], 3), "ms_per_step": round(timed[m_] * 1e3 / args.steps, 4)}
    if bypos:
        line["by_position"] = bypos
    if single_dt:
        line["single_stream_tok_s"] = round(single_dt[1] / single_dt[0], 3)       # one sequence in flight through the N stages (same mode as `value`)
        line["single_stream_ms_per_token"] = round(single_dt[0] * 1e3 / single_dt[1], 4)
        line["single_stream_steps"] = single_dt[1]
    st.check(st.dev.lib.rama_set_tuning(st.dev.ctx, b"ref_order", 0), "rama_set_tuning")
    st.free()
    dist.barrier()
    dist.destroy_process_group()
    return line


def run_pipeline_bench_torch(args, cfg, rank: int, world: int, local_rank: int) -> dict:
    """bench.py --gpus N > 1 on the torch.distributed path: N sequences in flight, a step = N ticks (every sequence advances
    one token, every rank does one full-stage pass per sequence: per-GPU work is fixed -> weak)."""
    import torch
    import torch.distributed as dist
    import rama_amd
    from bench import HBM_PEAK_GBPS, PROMPT

    _rendezvous_env()
    if not dist.is_initialized():
        dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
    n_seq = world
    n_pos = args.warmup + args.steps        # longer than seq_len: the slots start new generations (run_ticks wrap)
    backend = HipStage(cfg, rank, world, local_rank, n_seq, seed=0)
    sched = Schedule(world, n_seq, n_pos)
    prompts = [PROMPT for _ in range(n_seq)]
    t_warm = args.warmup * n_seq                     # ticks; >= world - 1 fills the pipe
    t_end = t_warm + args.steps * n_seq
    # t_warm < world - 1 (e.g. --warmup 0): the timed region then includes the pipeline fill
    run_ticks(sched, rank, backend, 0, t_warm, prompts, dist, wrap=cfg.seq_len)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_ticks(sched, rank, backend, t_warm, t_end, prompts, dist, wrap=cfg.seq_len)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    run_ticks(sched, rank, backend, t_end, sched.ticks, prompts, dist, wrap=cfg.seq_len)   # drain, untimed
    torch.cuda.synchronize()
    tmax = torch.tensor([dt], dtype=torch.float64, device=torch.device("cuda", local_rank))
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    roofline = None
    n_local = backend.stage.layer_end - backend.stage.layer_begin
    if n_local > 0 and not args.no_kprof:
        roofline = _stage_roofline(backend.check, backend.dev.lib, backend.dev.ctx,
                                   lambda: backend.compute(0, (n_pos - 1) % cfg.seq_len, BOS if rank == 0 else None), n_local, cfg)
    line = _bench_line(args, cfg, world, n_seq, args.steps * n_seq / dt, dt, roofline, "torch.distributed P2P, tick loop in Python",
                       False, rccl_ranks=dist.get_world_size())
    backend.free()
    dist.barrier()
    dist.destroy_process_group()
    return line


def run_pipeline_bench(args, cfg, rank: int, world: int, local_rank: int) -> dict:
    """bench.py --gpus N > 1: the native path unless RAMA_TORCH_PIPE=1."""
    if os.environ.get("RAMA_TORCH_PIPE"):
        return run_pipeline_bench_torch(args, cfg, rank, world, local_rank)
    return run_pipeline_bench_native(args, cfg, rank, world, local_rank)
