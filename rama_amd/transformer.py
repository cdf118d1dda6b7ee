"""Host-side mirror of the reference engine's interface for the decode path, over the
C ABI (include/rama_hip.h).  Names, argument meaning and error behaviour follow

  engine/src/transformer/mod.rs    Storage / View / MutView / range_from / Config / generate
  engine/src/transformer/state.rs  RunState(+View), TransformerWeights(+View)
  engine/src/transformer/hbm.rs    device allocation / upload / into_state
  engine/src/device/device.rs      trait Device<T>         (class Hip below)
  engine/src/transformer/infer.rs  forward()

so a test written against the reference reads the same here.  The reference trait has no
Result: every driver error panics; here every non-zero return raises RamaError.

There is no CPU fallback in this module: every op runs a HIP kernel from librama_hip.so.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import _lib
from ._lib import RamaError, S_FIELDS, W_FIELDS, check, rama_config, rama_run_state, rama_stage, rama_weights
from .sampler_const import TOPP_U_CPU      # the reference's constant draw (seed 100, cpu.rs:161-162), derived in sampler_const.py


# ------------------------------------------------------------------ Storage / View / MutView

class HipSlice:
    """Storage bound to device memory (reference: `impl Storage for CudaSlice<f32>`, hbm.rs:6-10)."""

    def __init__(self, device: "Hip", ptr: int, n: int, owner: bool = True):
        self.device, self.ptr, self.n, self._owner = device, ptr, n, owner

    def length(self) -> int:
        return self.n

    def free(self):
        if self._owner and self.ptr:
            check(self.device.lib.rama_free(self.device.ctx, self.ptr), "rama_free")
            self.ptr = 0


def range_from(start: Optional[int], end: Optional[int], max_len: int) -> range:
    """mod.rs:26-41: an open end means the STORAGE length, not the parent view's end."""
    return range(0 if start is None else start, max_len if end is None else end)


class View:
    """mod.rs:16-19,43-59.  `range` is ABSOLUTE in the backing storage (mod.rs:44-51)."""

    def __init__(self, data: HipSlice, rng: Optional[range] = None):
        self.data = data
        self.range = rng if rng is not None else range(0, data.length())

    def slice(self, start: Optional[int] = None, end: Optional[int] = None) -> "View":
        return View(self.data, range_from(start, end, self.data.length()))

    @property
    def ptr(self) -> int:   # reference: cudaview() = storage.slice(range), gpu.rs:51-69
        return self.data.ptr + 4 * self.range.start

    def __len__(self):
        return len(self.range)


class MutView(View):
    """mod.rs:21-24,65-96"""

    def as_view(self) -> View:
        return View(self.data, self.range)

    def mut_slice(self, start: Optional[int] = None, end: Optional[int] = None) -> "MutView":
        return MutView(self.data, range_from(start, end, self.data.length()))


# ------------------------------------------------------------------ Config

@dataclass
class Config:
    """mod.rs:128-138"""
    dim: int
    hidden_dim: int
    n_layers: int
    n_heads: int
    n_kv_heads: int
    vocab_size: int
    seq_len: int
    shared_weight: bool

    @staticmethod
    def from_file(path) -> "Config":
        """mod.rs:141-166: 7 x i32; vocab_size > 0 means the classifier is shared."""
        h = np.fromfile(path, dtype="<i4", count=7)
        if h.size != 7:
            raise RamaError("error reading file")   # utils/read.rs:27
        v = int(h[5])
        return Config(int(h[0]), int(h[1]), int(h[2]), int(h[3]), int(h[4]), abs(v), int(h[6]), v > 0)

    @property
    def head_size(self) -> int:
        return self.dim // self.n_heads

    def c(self) -> rama_config:
        return rama_config(self.dim, self.hidden_dim, self.n_layers, self.n_heads, self.n_kv_heads,
                           self.vocab_size, self.seq_len, int(self.shared_weight))


def weight_shapes(cfg: Config):
    """file order = struct-literal order of ram.rs:31-49"""
    hs, L, d, h, V, S = cfg.head_size, cfg.n_layers, cfg.dim, cfg.hidden_dim, cfg.vocab_size, cfg.seq_len
    out = [("token_embedding_table", (V, d)), ("rms_att_weight", (L, d)),
           ("wq", (L, d, d)), ("wk", (L, d, d)), ("wv", (L, d, d)), ("wo", (L, d, d)),
           ("rms_ffn_weight", (L, d)), ("w1", (L, h, d)), ("w2", (L, d, h)), ("w3", (L, h, d)),
           ("rms_final_weight", (d,)), ("freq_cis_real", (S, hs // 2)), ("freq_cis_imag", (S, hs // 2))]
    if not cfg.shared_weight:
        out.append(("wcls", (V, d)))
    return out


# ------------------------------------------------------------------ the Device trait

class Hip:
    """`impl Device<HipSlice> for Hip` (trait: device.rs:3-24; replaces gpu.rs's GPU).

    One context = one HIP stream.  `stream` may be an existing hipStream_t handle (int)."""

    def __init__(self, device: int = 0, stream: int = 0):
        self.lib = _lib.load()
        ctx = C.c_void_p()
        check(self.lib.rama_ctx_create(device, stream or None, C.byref(ctx)), "rama_ctx_create")
        self.ctx = ctx
        self.device_index = device

    def close(self):
        if self.ctx:
            self.lib.rama_ctx_destroy(self.ctx)
            self.ctx = None

    def sync(self):
        check(self.lib.rama_sync(self.ctx), "rama_sync")

    def info(self):
        name = C.create_string_buffer(64)
        cus, mem = C.c_int(), C.c_size_t()
        check(self.lib.rama_device_info(self.ctx, name, C.byref(cus), C.byref(mem)))
        return name.value.decode(), cus.value, mem.value

    # ---- allocation (hbm.rs:14-16 allocate = htod_sync_copy; ram.rs zero-init)
    def alloc(self, n: int) -> HipSlice:
        p = C.c_void_p()
        check(self.lib.rama_alloc_f32(self.ctx, n, C.byref(p)), "rama_alloc_f32")
        return HipSlice(self, p.value, n)

    def allocate(self, data: np.ndarray) -> HipSlice:
        a = np.ascontiguousarray(data, dtype=np.float32).reshape(-1)
        p = C.c_void_p()
        check(self.lib.rama_upload_f32(self.ctx, a.ctypes.data, a.size, C.byref(p)), "rama_upload_f32")
        return HipSlice(self, p.value, a.size)

    def download(self, v) -> np.ndarray:
        """dtoh_sync_copy of a storage or a view"""
        ptr, n = (v.ptr, len(v)) if isinstance(v, View) else (v.ptr, v.length())
        out = np.empty(n, dtype=np.float32)
        check(self.lib.rama_download_f32(self.ctx, ptr, n, out.ctypes.data), "rama_download_f32")
        return out

    def upload_into(self, v, data: np.ndarray):
        a = np.ascontiguousarray(data, dtype=np.float32).reshape(-1)
        ptr = v.ptr
        check(self.lib.rama_copy_h2d_f32(self.ctx, ptr, a.ctypes.data, a.size), "rama_copy_h2d_f32")

    # ---- trait methods, same names and argument order as device.rs:4-21
    def array_add(self, target: MutView, source: View, n: int):
        check(self.lib.rama_array_add(self.ctx, target.ptr, source.ptr, n), "array_add")

    def array_mult(self, target: MutView, source: View, n: int):
        check(self.lib.rama_array_mult(self.ctx, target.ptr, source.ptr, n), "array_mult")

    def sinu(self, o: MutView, n: int):
        check(self.lib.rama_sinu(self.ctx, o.ptr, n), "sinu")

    def multi_head_attention(self, rsv: "RunStateView", cfg: Config, layer: int, pos: int):
        check(self.lib.rama_multi_head_attention(
            self.ctx, rsv.xb.ptr, rsv.att.ptr, rsv.q.ptr, rsv.key_cache.ptr, rsv.value_cache.ptr,
            layer, cfg.dim, pos, cfg.head_size, cfg.seq_len, cfg.n_heads), "multi_head_attention")

    def copy_from_slice(self, target: MutView, source: View, n: int):
        check(self.lib.rama_copy_from_slice(self.ctx, target.ptr, source.ptr, n), "copy_from_slice")

    def rmsnorm(self, o: MutView, x: View, weight: View, n: int):
        check(self.lib.rama_rmsnorm(self.ctx, o.ptr, x.ptr, weight.ptr, n), "rmsnorm")

    def apply_position(self, q: MutView, k: MutView, pos_real: View, pos_img: View, head_size: int):
        check(self.lib.rama_apply_position(self.ctx, q.ptr, k.ptr, pos_real.ptr, pos_img.ptr, head_size), "apply_position")

    def matmul(self, o: MutView, a: View, b: View, width: int, o_rows: int, o_cols: int):
        check(self.lib.rama_matmul(self.ctx, o.ptr, a.ptr, b.ptr, width, o_rows, o_cols), "matmul")

    def softmax(self, x: MutView, n: int):
        check(self.lib.rama_softmax(self.ctx, x.ptr, n), "softmax")

    def sample(self, cfg: Config, rsv: "RunStateView", temperature: float, topp: float, u: float = TOPP_U_CPU) -> int:
        """device.rs:16.  `u`: the reference's draw is one constant because ChaCha20 is re-seeded
        on every call (cpu.rs:161-162); the default is the value derived for seed 100 (SURVEY 8c,
        provisional -- no Rust toolchain here to confirm it)."""
        nxt = C.c_int32()
        if temperature == 0.0:
            check(self.lib.rama_sample_argmax(self.ctx, rsv.logits.ptr, cfg.vocab_size, C.byref(nxt)), "sample")
        else:
            check(self.lib.rama_sample_topp(self.ctx, rsv.logits.ptr, cfg.vocab_size, temperature, topp, u, C.byref(nxt)), "sample")
        return nxt.value

    def to_cpu(self, state: "RunStateView", cpu_state: dict):
        """device.rs:21 / gpu.rs:196-209: dump every run-state buffer into host arrays."""
        for name in S_FIELDS:
            cpu_state[name] = self.download(getattr(state, name).data)


# ------------------------------------------------------------------ RunState / TransformerWeights

class RunState:
    """state.rs:3-17; sizes ram.rs:7-23; device copy hbm.rs:19-34 (from_state)."""

    def __init__(self, **bufs):
        for k in S_FIELDS:
            setattr(self, k, bufs[k])

    @staticmethod
    def from_config(cfg: Config, device: Hip) -> "RunState":
        kv_dim = cfg.dim * cfg.n_kv_heads // cfg.n_heads
        sizes = dict(x=cfg.dim, xb=cfg.dim, xb2=cfg.dim, hb=cfg.hidden_dim, hb2=cfg.hidden_dim,
                     q=cfg.dim, k=cfg.dim, v=cfg.dim, att=cfg.n_heads * cfg.seq_len,
                     logits=cfg.vocab_size, key_cache=cfg.n_layers * cfg.seq_len * kv_dim,
                     value_cache=cfg.n_layers * cfg.seq_len * kv_dim)
        return RunState(**{k: device.alloc(sizes[k]) for k in S_FIELDS})

    def into_state(self, device: Hip) -> dict:
        """hbm.rs:38-51"""
        return {k: device.download(getattr(self, k)) for k in S_FIELDS}

    def c(self) -> rama_run_state:
        return rama_run_state(*[getattr(self, k).ptr for k in S_FIELDS])

    def free(self):
        for k in S_FIELDS:
            getattr(self, k).free()


class RunStateView:
    """state.rs:19-51"""

    def __init__(self, rs: RunState):
        for k in S_FIELDS:
            setattr(self, k, MutView(getattr(rs, k)))

    @staticmethod
    def from_rs(rs: RunState) -> "RunStateView":
        return RunStateView(rs)


class TransformerWeights:
    """state.rs:53-74; upload hbm.rs:55-90 (one allocation per tensor, like from_weight)."""

    def __init__(self, cfg: Config, tensors: dict, wcls_exists: bool):
        self.cfg, self.wcls_exists = cfg, wcls_exists
        for k in W_FIELDS:
            setattr(self, k, tensors[k])

    @staticmethod
    def from_numpy(cfg: Config, w: dict, device: Hip) -> "TransformerWeights":
        t = {}
        for name, _shape in weight_shapes(cfg):
            t[name] = device.allocate(w[name])
        if cfg.shared_weight:
            t["wcls"] = device.allocate(np.array([1.0], dtype=np.float32))   # ram.rs:46 placeholder
        return TransformerWeights(cfg, t, not cfg.shared_weight)

    @staticmethod
    def synth(cfg: Config, seed: int, device: Hip) -> "TransformerWeights":
        """the synthetic weights of rama_model_synth (csrc/model.hip: same tags, scales and RoPE tables, so the same bits), but ONE ALLOCATION
        PER TENSOR like hbm.rs:55-90 and filled on the device -- what bench.py's `trait_ops_path` runs at llama2-7B without 27 GB of host memory"""
        ih4_std = np.sqrt(4.0 * (65536.0 * 65536.0 - 1.0) / 12.0)
        res = 0.02 / np.sqrt(2.0 * cfg.n_layers)
        spec = {"token_embedding_table": (1, 0.02, 0.0), "rms_att_weight": (2, 0.05, 1.0), "wq": (3, 0.02, 0.0), "wk": (4, 0.02, 0.0),
                "wv": (5, 0.02, 0.0), "wo": (6, res, 0.0), "rms_ffn_weight": (7, 0.05, 1.0), "w1": (8, 0.02, 0.0), "w2": (9, 0.02, 0.0),
                "w3": (10, res, 0.0), "rms_final_weight": (11, 0.05, 1.0), "wcls": (12, 0.02, 0.0)}
        hs = cfg.head_size
        f = 1.0 / np.power(10000.0, (2.0 * np.arange(hs // 2)) / hs)
        ang = np.arange(cfg.seq_len, dtype=np.float64)[:, None] * f[None, :]
        t = {}
        for name, shp in weight_shapes(cfg):
            n = int(np.prod(shp))
            if name.startswith("freq_cis"):
                t[name] = device.allocate((np.cos(ang) if name.endswith("real") else np.sin(ang)).astype(np.float32))
                continue
            p = C.c_void_p()
            check(device.lib.rama_alloc_f32(device.ctx, n, C.byref(p)), "rama_alloc_f32")
            tag, std, bias = spec[name]
            check(device.lib.rama_fill_synth(device.ctx, p, n, seed, tag, 0, float(std / ih4_std), bias), "rama_fill_synth")
            t[name] = HipSlice(device, p.value, n)
        if cfg.shared_weight:
            t["wcls"] = device.allocate(np.array([1.0], dtype=np.float32))   # ram.rs:46 placeholder
        return TransformerWeights(cfg, t, not cfg.shared_weight)

    @staticmethod
    def from_file(path, cfg: Config, device: Hip) -> "TransformerWeights":
        """ram.rs:28-51 order; tensors read from a memory map, then uploaded."""
        body = np.memmap(path, dtype="<f4", mode="r", offset=28)
        w, off = {}, 0
        for name, shp in weight_shapes(cfg):
            n = int(np.prod(shp))
            w[name] = body[off:off + n]
            off += n
        if off != body.size:
            raise RamaError("error reading file")
        return TransformerWeights.from_numpy(cfg, w, device)

    def free(self):
        for k in W_FIELDS:
            getattr(self, k).free()


class TransformerWeightsView:
    """state.rs:76-122 / hbm.rs:95-120: wcls aliases the embedding table when shared."""

    def __init__(self, ws: TransformerWeights):
        for k in W_FIELDS:
            setattr(self, k, View(getattr(ws, k)))
        self.wcls_exists = ws.wcls_exists
        if not ws.wcls_exists:
            self.wcls = View(ws.token_embedding_table)

    @staticmethod
    def from_gpu_ws(ws: TransformerWeights) -> "TransformerWeightsView":
        return TransformerWeightsView(ws)

    def c(self) -> rama_weights:
        return rama_weights(*[getattr(self, k).ptr for k in W_FIELDS])


# ------------------------------------------------------------------ forward / generate

def forward(cfg: Config, wv: TransformerWeightsView, rsv: RunStateView, token: int, pos: int, device: Hip):
    """infer.rs:8-53, op for op through the Device trait (the Wq product of :20-21 is issued
    once: it is idempotent).  This is the drop-in path; rama_forward is the fused one."""
    dim, hidden_dim = cfg.dim, cfg.hidden_dim
    head_size = dim // cfg.n_heads
    device.copy_from_slice(rsv.x, wv.token_embedding_table.slice(token * dim, (token + 1) * dim), dim)
    pos_real = wv.freq_cis_real.slice(pos * (head_size // 2))
    pos_img = wv.freq_cis_imag.slice(pos * (head_size // 2))
    for layer in range(cfg.n_layers):
        device.rmsnorm(rsv.xb, rsv.x.as_view(), wv.rms_att_weight.slice(layer * dim), dim)
        device.matmul(rsv.q, wv.wq.slice(layer * dim * dim), rsv.xb.as_view(), dim, dim, 1)
        device.matmul(rsv.k, wv.wk.slice(layer * dim * dim), rsv.xb.as_view(), dim, dim, 1)
        device.matmul(rsv.v, wv.wv.slice(layer * dim * dim), rsv.xb.as_view(), dim, dim, 1)
        for h in range(cfg.n_heads):
            q = rsv.q.mut_slice(h * head_size)
            k = rsv.k.mut_slice(h * head_size)
            device.apply_position(q, k, pos_real, pos_img, head_size)
        lo = layer * cfg.seq_len * dim
        device.copy_from_slice(rsv.key_cache.mut_slice(lo + pos * dim, lo + (pos + 1) * dim), rsv.k.as_view(), dim)
        device.copy_from_slice(rsv.value_cache.mut_slice(lo + pos * dim, lo + (pos + 1) * dim), rsv.v.as_view(), dim)
        device.multi_head_attention(rsv, cfg, layer, pos)
        device.matmul(rsv.xb2, wv.wo.slice(layer * dim * dim), rsv.xb.as_view(), dim, dim, 1)
        device.array_add(rsv.x, rsv.xb2.as_view(), dim)
        device.rmsnorm(rsv.xb, rsv.x.as_view(), wv.rms_ffn_weight.slice(layer * dim), dim)
        device.matmul(rsv.hb, wv.w1.slice(layer * hidden_dim * dim), rsv.xb.as_view(), dim, hidden_dim, 1)
        device.matmul(rsv.hb2, wv.w3.slice(layer * hidden_dim * dim), rsv.xb.as_view(), dim, hidden_dim, 1)
        device.sinu(rsv.hb, hidden_dim)
        device.array_mult(rsv.hb, rsv.hb2.as_view(), hidden_dim)
        device.matmul(rsv.xb, wv.w2.slice(layer * dim * hidden_dim), rsv.hb.as_view(), hidden_dim, dim, 1)
        device.array_add(rsv.x, rsv.xb.as_view(), dim)
    device.copy_from_slice(rsv.xb, rsv.x.as_view(), dim)
    device.rmsnorm(rsv.x, rsv.xb.as_view(), wv.rms_final_weight, dim)
    device.matmul(rsv.logits, wv.wcls, rsv.x.as_view(), dim, cfg.vocab_size, 1)


def forward_fused(cfg: Config, wv: TransformerWeightsView, rsv: RunStateView, token: int, pos: int, device: Hip):
    """Same contract as forward() (logits, caches, residual x), five fused launches per layer."""
    rs = rama_run_state(*[getattr(rsv, k).ptr for k in S_FIELDS])
    check(device.lib.rama_forward(device.ctx, C.byref(cfg.c()), C.byref(wv.c()), C.byref(rs), token, pos), "rama_forward")


def generate(cfg: Config, prompt_tokens, temperature: float, steps: int, topp: float,
             wv: TransformerWeightsView, rsv: RunStateView, device: Hip, fused: bool = True):
    """mod.rs:169-206 minus the tokenizer/printing: BOS (1) at pos 0, forced prompt tokens,
    then device.sample; never stops on EOS; exactly `steps` forwards.  Returns every `next`."""
    token, out = 1, []
    fwd = forward_fused if fused else forward
    for pos in range(steps):
        fwd(cfg, wv, rsv, token, pos, device)
        nxt = prompt_tokens[pos] if pos < len(prompt_tokens) else device.sample(cfg, rsv, temperature, topp)
        out.append(int(nxt))
        token = nxt
    return out


def generate_device(cfg: Config, prompt_tokens, temperature: float, steps: int, topp: float,
                    wv: TransformerWeightsView, rsv: RunStateView, device: Hip, u: float = TOPP_U_CPU):
    """generate() chained on the device for any temperature (rama_generate): argmax or top-p sampling
    without the per-token logits download; one D2H of the token list at the end."""
    rs = rama_run_state(*[getattr(rsv, k).ptr for k in S_FIELDS])
    pt = (C.c_int32 * max(len(prompt_tokens), 1))(*prompt_tokens)
    out = (C.c_int32 * max(steps, 1))()
    check(device.lib.rama_generate(device.ctx, C.byref(cfg.c()), C.byref(wv.c()), C.byref(rs),
                                   pt, len(prompt_tokens), steps, temperature, topp, u, out), "rama_generate")
    return [int(v) for v in out[:steps]]


def generate_greedy_device(cfg: Config, prompt_tokens, steps: int, wv: TransformerWeightsView,
                           rsv: RunStateView, device: Hip):
    """The T == 0 loop chained on the device (rama_generate_greedy): one D2H at the end."""
    rs = rama_run_state(*[getattr(rsv, k).ptr for k in S_FIELDS])
    pt = (C.c_int32 * max(len(prompt_tokens), 1))(*prompt_tokens)
    out = (C.c_int32 * max(steps, 1))()
    check(device.lib.rama_generate_greedy(device.ctx, C.byref(cfg.c()), C.byref(wv.c()), C.byref(rs),
                                          pt, len(prompt_tokens), steps, out), "rama_generate_greedy")
    return [int(v) for v in out[:steps]]
