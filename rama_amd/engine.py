"""Resident-model fast path over the C ABI: weights as one blob in HBM (rama_model_load /
rama_model_synth), a run state (rama_state_create) and the fused, device-chained decode
loop.  This is what bench.py times.  No CPU fallback anywhere."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

from ._lib import check, rama_config, rama_run_state, rama_stage, rama_weights
from .transformer import Config, Hip

KERNEL_IDS = dict(qkv=0, attn=1, wo=2, w13=3, w2=4, cls=5, norm=6, sample=7)


def algorithmic_bytes(cfg: Config) -> dict:
    """fp32 weight bytes each kernel class streams per launch, and per token:
    B_w = 4 * [L * (4 dim^2 + 3 dim hidden) + vocab dim]."""
    d, h, V, L = cfg.dim, cfg.hidden_dim, cfg.vocab_size, cfg.n_layers
    per = dict(qkv=4 * 3 * d * d, wo=4 * d * d, w13=4 * 2 * h * d, w2=4 * d * h, cls=4 * V * d)
    per["token"] = L * (per["qkv"] + per["wo"] + per["w13"] + per["w2"]) + per["cls"]
    return per


class Model:
    def __init__(self, device: Hip, handle, stage: rama_stage):
        self.device, self.handle, self.stage = device, handle, stage
        c = rama_config()
        check(device.lib.rama_model_config(handle, C.byref(c)))
        self.cfg = Config(c.dim, c.hidden_dim, c.n_layers, c.n_heads, c.n_kv_heads, c.vocab_size,
                          c.seq_len, bool(c.shared_weight))
        self.ccfg = c
        self.weights = rama_weights()
        check(device.lib.rama_model_weights(handle, C.byref(self.weights)))

    @staticmethod
    def load(device: Hip, path) -> "Model":
        """llama2.c v0 .bin -> HBM (mmap + one staged copy)."""
        h = C.c_void_p()
        check(device.lib.rama_model_load(device.ctx, str(path).encode(), C.byref(h)), "rama_model_load")
        c = rama_config()
        check(device.lib.rama_model_config(h, C.byref(c)))
        return Model(device, h, rama_stage(0, c.n_layers, 1, 1))

    @staticmethod
    def synth(device: Hip, cfg: Config, seed: int, stage: Optional[rama_stage] = None,
              rope: Optional[Sequence[np.ndarray]] = None) -> "Model":
        """Synthetic weights generated in HBM (bit-identical to the oracle's generator)."""
        st = stage or rama_stage(0, cfg.n_layers, 1, 1)
        h = C.c_void_p()
        fr = fi = None
        if rope is not None:
            fr = np.ascontiguousarray(rope[0], dtype=np.float32)
            fi = np.ascontiguousarray(rope[1], dtype=np.float32)
        check(device.lib.rama_model_synth(device.ctx, C.byref(cfg.c()), seed, C.byref(st),
                                          fr.ctypes.data if fr is not None else None,
                                          fi.ctypes.data if fi is not None else None, C.byref(h)),
              "rama_model_synth")
        return Model(device, h, st)

    def save(self, path):
        """write the model as a llama2.c v0 .bin (export.py legacy layout; upstream Rama loads it)"""
        check(self.device.lib.rama_model_save(self.device.ctx, self.handle, str(path).encode()), "rama_model_save")

    @property
    def bytes(self) -> int:
        return self.device.lib.rama_model_bytes(self.handle)

    def tensor(self, name: str, n: int, offset: int = 0) -> np.ndarray:
        """download n floats of a weight tensor (tests)"""
        out = np.empty(n, dtype=np.float32)
        check(self.device.lib.rama_download_f32(self.device.ctx, getattr(self.weights, name) + 4 * offset, n, out.ctypes.data))
        return out

    def release_copies(self, mask: int = 3):
        """give the chain-order (1) / tile-order (2) weight copies back; they are made again on demand"""
        check(self.device.lib.rama_model_release_copies(self.device.ctx, self.handle, mask), "rama_model_release_copies")

    def free(self):
        if self.handle:
            check(self.device.lib.rama_model_free(self.device.ctx, self.handle))
            self.handle = None


class Engine:
    """model + run state + decode cursor on one device/stream."""

    def __init__(self, device: Hip, model: Model):
        self.device, self.model, self.cfg = device, model, model.cfg
        self.stage = model.stage
        self.state = rama_run_state()
        n_local = self.stage.layer_end - self.stage.layer_begin
        check(device.lib.rama_state_create(device.ctx, C.byref(model.ccfg), n_local, C.byref(self.state)),
              "rama_state_create")

    # -- single step (infer.rs:8-53 contract)
    def forward(self, token: int, pos: int):
        L = self.device.lib
        check(L.rama_forward_stage(self.device.ctx, C.byref(self.model.ccfg), C.byref(self.model.weights),
                                   C.byref(self.state), token, pos, C.byref(self.stage)), "rama_forward_stage")

    def buffer(self, name: str, n: int, offset: int = 0) -> np.ndarray:
        out = np.empty(n, dtype=np.float32)
        check(self.device.lib.rama_download_f32(self.device.ctx, getattr(self.state, name) + 4 * offset, n, out.ctypes.data))
        return out

    def set_buffer(self, name: str, data: np.ndarray, offset: int = 0):
        a = np.ascontiguousarray(data, dtype=np.float32).reshape(-1)
        check(self.device.lib.rama_copy_h2d_f32(self.device.ctx, getattr(self.state, name) + 4 * offset, a.ctypes.data, a.size))

    def logits(self) -> np.ndarray:
        return self.buffer("logits", self.cfg.vocab_size)

    # -- generate() at T == 0, chained on the device
    def generate_greedy(self, prompt_tokens, steps: int):
        pt = (C.c_int32 * max(len(prompt_tokens), 1))(*prompt_tokens)
        out = (C.c_int32 * max(steps, 1))()
        check(self.device.lib.rama_generate_greedy(self.device.ctx, C.byref(self.model.ccfg), C.byref(self.model.weights),
                                                   C.byref(self.state), pt, len(prompt_tokens), steps, out),
              "rama_generate_greedy")
        return [int(v) for v in out[:steps]]

    def generate(self, prompt_tokens, steps: int, temperature: float = 0.0, topp: float = 0.9, u: float = 0.0):
        """generate() for any temperature, sampled on the device (rama_generate)"""
        pt = (C.c_int32 * max(len(prompt_tokens), 1))(*prompt_tokens)
        out = (C.c_int32 * max(steps, 1))()
        check(self.device.lib.rama_generate(self.device.ctx, C.byref(self.model.ccfg), C.byref(self.model.weights),
                                            C.byref(self.state), pt, len(prompt_tokens), steps,
                                            temperature, topp, u, out), "rama_generate")
        return [int(v) for v in out[:steps]]

    def generate_stream(self, prompt_tokens, steps: int, on_token, temperature: float = 0.0, topp: float = 0.9, u: float = 0.0):
        """generate_stream (mod.rs:209-248): on_token(index, token) is called for every token as soon as the device has
        produced it, while the loop runs on, chained on the device (rama_generate_stream); returns the whole list"""
        pt = (C.c_int32 * max(len(prompt_tokens), 1))(*prompt_tokens)
        out = (C.c_int32 * max(steps, 1))()
        cb_t = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int32)
        cb = cb_t(lambda user, index, token: on_token(int(index), int(token)))
        check(self.device.lib.rama_generate_stream(self.device.ctx, C.byref(self.model.ccfg), C.byref(self.model.weights),
                                                   C.byref(self.state), pt, len(prompt_tokens), steps, temperature, topp, u,
                                                   cb, None, out), "rama_generate_stream")
        return [int(v) for v in out[:steps]]

    def decode_stream_poll(self, start: int, max_tokens: int = 64):
        """the tokens `start`.. the chained loop has produced so far (host-visible ring; never blocks, may be empty)"""
        buf = (C.c_int32 * max_tokens)()
        n = C.c_int()
        check(self.device.lib.rama_decode_stream_poll(self.device.ctx, start, buf, max_tokens, C.byref(n)), "rama_decode_stream_poll")
        return [int(v) for v in buf[:n.value]]

    def decode_sampler(self, temperature: float, topp: float = 0.9, u: float = 0.0):
        check(self.device.lib.rama_decode_sampler(self.device.ctx, temperature, topp, u), "rama_decode_sampler")

    def decode_begin(self, token: int, pos: int, forced=()):
        ft = (C.c_int32 * max(len(forced), 1))(*forced)
        check(self.device.lib.rama_decode_begin(self.device.ctx, token, pos, ft, len(forced)), "rama_decode_begin")

    def decode_steps(self, n: int):
        check(self.device.lib.rama_decode_steps(self.device.ctx, C.byref(self.model.ccfg), C.byref(self.model.weights),
                                                C.byref(self.state), n), "rama_decode_steps")

    def decode_tokens(self, max_tokens: int = 65536):
        out = (C.c_int32 * max_tokens)()
        n = C.c_int()
        check(self.device.lib.rama_decode_tokens(self.device.ctx, out, max_tokens, C.byref(n)))
        return [int(v) for v in out[:n.value]]

    def set_graph_mode(self, on: bool):
        check(self.device.lib.rama_set_graph_mode(self.device.ctx, int(on)))

    def set_tuning(self, key: str, value: int):
        check(self.device.lib.rama_set_tuning(self.device.ctx, key.encode(), value), "rama_set_tuning")

    # -- measurement (HIP events on the launch stream)
    def timer_start(self):
        check(self.device.lib.rama_timer_start(self.device.ctx))

    def timer_stop(self) -> float:
        ms = C.c_float()
        check(self.device.lib.rama_timer_stop(self.device.ctx, C.byref(ms)))
        return ms.value

    def kprof(self, kernel: str, n_steps: int):
        """average device time (ms) of one launch of a kernel class over n_steps decode steps"""
        L = self.device.lib
        check(L.rama_kprof_enable(self.device.ctx, KERNEL_IDS[kernel], n_steps * (2 * max(self.cfg.n_layers, 1) + 1)))
        self.decode_steps(n_steps)
        n, tot = C.c_int(), C.c_double()
        check(L.rama_kprof_read(self.device.ctx, C.byref(n), C.byref(tot)))
        return (tot.value / n.value if n.value else float("nan")), n.value

    def free(self):
        check(self.device.lib.rama_state_free(self.device.ctx, C.byref(self.state)))


def decode_batch(engines: Sequence["Engine"], tokens: Sequence[int], positions: Sequence[int]):
    """One decode step for up to 64 independent sequences over the same model (rama_decode_batch):
    engines[i] advances by forward(tokens[i], positions[i]); its logits()/caches are updated."""
    assert 1 <= len(engines) == len(tokens) == len(positions) <= 128
    e0 = engines[0]
    states = (rama_run_state * len(engines))(*[e.state for e in engines])
    toks = (C.c_int32 * len(engines))(*tokens)
    poss = (C.c_int32 * len(engines))(*positions)
    check(e0.device.lib.rama_decode_batch(e0.device.ctx, C.byref(e0.model.ccfg), C.byref(e0.model.weights),
                                          states, toks, poss, len(engines)), "rama_decode_batch")


def decode_batch_chained(engines: Sequence["Engine"], tokens: Sequence[int], positions: Sequence[int], n_steps: int, on_token=None):
    """n_steps greedy decode steps of up to 128 independent sequences chained on the device (rama_decode_batch_begin /
    _steps / _tokens): -> per sequence the n_steps tokens it produced.  engines[i]'s caches are advanced.
    on_token(sequence, index, token), when given, is called for every token as it appears in the host-visible rings
    (rama_decode_batch_stream_poll) while the steps run."""
    assert 1 <= len(engines) == len(tokens) == len(positions) <= 128
    e0 = engines[0]
    L = e0.device.lib
    states = (rama_run_state * len(engines))(*[e.state for e in engines])
    toks = (C.c_int32 * len(engines))(*tokens)
    poss = (C.c_int32 * len(engines))(*positions)
    check(L.rama_decode_batch_begin(e0.device.ctx, C.byref(e0.model.ccfg), C.byref(e0.model.weights), states, toks, poss,
                                    len(engines), max(n_steps, 1)), "rama_decode_batch_begin")
    check(L.rama_decode_batch_steps(e0.device.ctx, n_steps), "rama_decode_batch_steps")
    if on_token is not None:
        import time
        seen = [0] * len(engines)
        buf = (C.c_int32 * 64)()
        k = C.c_int()
        idle_after_done = 0
        while min(seen) < n_steps:
            progress = 0
            for s_ in range(len(engines)):
                check(L.rama_decode_batch_stream_poll(e0.device.ctx, s_, seen[s_], buf, 64, C.byref(k)), "rama_decode_batch_stream_poll")
                for i in range(k.value):
                    on_token(s_, seen[s_] + i, int(buf[i]))
                seen[s_] += k.value
                progress += k.value
            if progress:
                continue
            # nothing new: is the stream still working on it?  (1 = yes; 0 = everything has run -- one more sweep collects what is there;
            # anything else is a failed stream: the check raises instead of spinning for ever)
            q = L.rama_stream_query(e0.device.ctx)
            if q == 1:
                time.sleep(0.0002)
                continue
            check(q, "rama_stream_query")
            idle_after_done += 1
            if idle_after_done > 1:
                raise RuntimeError(f"decode_batch_chained: the steps have run but only {min(seen)} of {n_steps} tokens appeared")
    out = (C.c_int32 * (len(engines) * max(n_steps, 1)))()
    n = C.c_int()
    check(L.rama_decode_batch_tokens(e0.device.ctx, out, max(n_steps, 1), C.byref(n)), "rama_decode_batch_tokens")
    return [[int(out[s * max(n_steps, 1) + k]) for k in range(n.value)] for s in range(len(engines))]
