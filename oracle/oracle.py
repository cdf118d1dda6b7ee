"""ctypes binding of the CPU oracle (oracle/rama_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; nothing under rama_amd/ may import this module.

Besides the binding it restates, in numpy, the reference's checkpoint layout
(engine/src/transformer/mod.rs:141-166 header, ram.rs:28-51 tensor order) so the
oracle can be fed from a llama2.c v0 ``.bin`` independently of the product's loader.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB_PATH = _HERE / "librama_oracle.so"

_f32p = C.POINTER(C.c_float)


class OracleConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "dim", "hidden_dim", "n_layers", "n_heads", "n_kv_heads",
        "vocab_size", "seq_len", "shared_weight")]


_W_FIELDS = ("token_embedding_table", "rms_att_weight", "rms_ffn_weight",
             "wq", "wk", "wv", "wo", "w1", "w2", "w3",
             "rms_final_weight", "freq_cis_real", "freq_cis_imag", "wcls")
_S_FIELDS = ("x", "xb", "xb2", "hb", "hb2", "q", "k", "v", "att", "logits",
             "key_cache", "value_cache")


class OracleWeights(C.Structure):
    _fields_ = [(n, _f32p) for n in _W_FIELDS]


class OracleState(C.Structure):
    _fields_ = [(n, _f32p) for n in _S_FIELDS]


def build(force: bool = False) -> Path:
    """Compile the oracle with gcc (recipe: oracle/Makefile)."""
    if force or not _LIB_PATH.exists():
        subprocess.run(["make", "-C", str(_HERE)] + (["-B"] if force else []),
                       check=True, capture_output=True)
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    build()
    try:
        L = C.CDLL(str(_LIB_PATH))
    except OSError:
        build(force=True)
        L = C.CDLL(str(_LIB_PATH))
    sz = C.c_size_t
    L.oracle_set_threads.argtypes = [C.c_int]
    L.oracle_get_threads.restype = C.c_int
    L.oracle_set_lane_reduce.argtypes = [C.c_int]
    L.oracle_get_lane_reduce.restype = C.c_int
    L.oracle_set_softmax_split.argtypes = [C.c_int]
    L.oracle_get_softmax_split.restype = C.c_int
    L.oracle_array_add.argtypes = [_f32p, _f32p, sz]
    L.oracle_array_mult.argtypes = [_f32p, _f32p, sz]
    L.oracle_sinu.argtypes = [_f32p, sz]
    L.oracle_expf_array.argtypes = [_f32p, _f32p, sz]
    L.oracle_copy_from_slice.argtypes = [_f32p, _f32p, sz]
    L.oracle_rmsnorm.argtypes = [_f32p, _f32p, _f32p, sz]
    L.oracle_apply_position.argtypes = [_f32p, _f32p, _f32p, _f32p, sz]
    L.oracle_matmul.argtypes = [_f32p, _f32p, _f32p, sz, sz, sz]
    L.oracle_matmul.restype = C.c_int
    L.oracle_softmax.argtypes = [_f32p, sz]
    L.oracle_multi_head_attention.argtypes = [C.POINTER(OracleConfig), C.POINTER(OracleState), C.c_int, C.c_int]
    L.oracle_argmax.argtypes = [_f32p, sz]
    L.oracle_argmax.restype = C.c_int
    L.oracle_sample.argtypes = [_f32p, sz, C.c_float, C.c_float, C.c_float]
    L.oracle_sample.restype = C.c_int
    L.oracle_sample_top_q.argtypes = [_f32p, sz, C.c_float, C.c_float]
    L.oracle_sample_top_q.restype = C.c_int
    L.oracle_forward.argtypes = [C.POINTER(OracleConfig), C.POINTER(OracleWeights), C.POINTER(OracleState), C.c_int, C.c_int]
    L.oracle_forward_range.argtypes = L.oracle_forward.argtypes + [C.c_int] * 4
    L.oracle_forward_f64.argtypes = L.oracle_forward.argtypes
    L.oracle_fill_synth.argtypes = [_f32p, sz, C.c_uint64, C.c_uint64, C.c_uint64, C.c_float, C.c_float]
    # The GPU box reports 256 CPUs but a 1-GPU job owns a 16-CPU share: an OpenMP team of
    # 256 on that share makes every tiny parallel region crawl.  Callers may override.
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    L.oracle_set_threads(max(1, min(16, ncpu)))
    _lib = L
    return L


def _p(a: np.ndarray):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"], (a.dtype, a.flags)
    return a.ctypes.data_as(_f32p)


# ------------------------------------------------------------------ the two orders the reference leaves to its crates (rama_oracle.h)
LANES = {"pairwise": 0, "strided": 1, "sequential": 2}      # wide::f32x4::reduce_add, cpu.rs:148


class orders:
    """with orders(lane_reduce="strided", softmax_split=3): ...  -- process-wide switches of the C library, restored on exit"""

    def __init__(self, lane_reduce=None, softmax_split=None):
        self.lr = LANES[lane_reduce] if isinstance(lane_reduce, str) else lane_reduce
        self.ss = softmax_split

    def __enter__(self):
        L = lib()
        self.old = (L.oracle_get_lane_reduce(), L.oracle_get_softmax_split())
        if self.lr is not None:
            L.oracle_set_lane_reduce(int(self.lr))
        if self.ss is not None:
            L.oracle_set_softmax_split(int(self.ss))
        return self

    def __exit__(self, *exc):
        L = lib()
        L.oracle_set_lane_reduce(self.old[0])
        L.oracle_set_softmax_split(self.old[1])
        return False


# ------------------------------------------------------------------ ops (1:1 with Device)

def array_add(target, source, n): lib().oracle_array_add(_p(target), _p(source), n)
def array_mult(target, source, n): lib().oracle_array_mult(_p(target), _p(source), n)
def sinu(o, n): lib().oracle_sinu(_p(o), n)


def expf(x: np.ndarray) -> np.ndarray:
    out = np.empty_like(x)
    lib().oracle_expf_array(_p(out), _p(x), x.size)
    return out


def copy_from_slice(target, source, n): lib().oracle_copy_from_slice(_p(target), _p(source), n)
def rmsnorm(o, x, weight, n): lib().oracle_rmsnorm(_p(o), _p(x), _p(weight), n)
def softmax(x, n): lib().oracle_softmax(_p(x), n)
def argmax(logits) -> int: return lib().oracle_argmax(_p(logits), logits.size)


def apply_position(q, k, pos_real, pos_img, head_size):
    lib().oracle_apply_position(_p(q), _p(k), _p(pos_real), _p(pos_img), head_size)


def matmul(o, a, b, width, o_rows, o_cols=1):
    rc = lib().oracle_matmul(_p(o), _p(a), _p(b), width, o_rows, o_cols)
    if rc != 0:
        raise ValueError("reference would panic: width %% 4 != 0 (cpu.rs:142-143), width=%d" % width)


def sample(logits, temperature, topp, u) -> int:
    return lib().oracle_sample(_p(logits), logits.size, temperature, topp, u)


def sample_top_q(probabilities, num, topp, u) -> int:
    p = np.ascontiguousarray(probabilities, np.float32)
    return lib().oracle_sample_top_q(_p(p), num, topp, u)


def fill_synth(n: int, seed: int, tag: int, scale: float, bias: float = 0.0, offset: int = 0) -> np.ndarray:
    out = np.empty(n, dtype=np.float32)
    lib().oracle_fill_synth(_p(out), n, seed, tag, offset, np.float32(scale), np.float32(bias))
    return out


# ------------------------------------------------------------------ checkpoint (v0)

@dataclass
class Config:
    """engine/src/transformer/mod.rs:128-138"""
    dim: int
    hidden_dim: int
    n_layers: int
    n_heads: int
    n_kv_heads: int
    vocab_size: int
    seq_len: int
    shared_weight: bool

    @property
    def head_size(self) -> int:
        return self.dim // self.n_heads

    def c(self) -> OracleConfig:
        return OracleConfig(self.dim, self.hidden_dim, self.n_layers, self.n_heads,
                            self.n_kv_heads, self.vocab_size, self.seq_len, int(self.shared_weight))


def weight_shapes(cfg: Config):
    """Tensor order in the file = struct-literal order in ram.rs:31-49."""
    hs = cfg.head_size
    L, d, h, V, S = cfg.n_layers, cfg.dim, cfg.hidden_dim, cfg.vocab_size, cfg.seq_len
    shapes = [
        ("token_embedding_table", (V, d)),
        ("rms_att_weight", (L, d)),
        ("wq", (L, d, d)), ("wk", (L, d, d)), ("wv", (L, d, d)), ("wo", (L, d, d)),
        ("rms_ffn_weight", (L, d)),
        ("w1", (L, h, d)), ("w2", (L, d, h)), ("w3", (L, h, d)),
        ("rms_final_weight", (d,)),
        ("freq_cis_real", (S, hs // 2)), ("freq_cis_imag", (S, hs // 2)),
    ]
    if not cfg.shared_weight:
        shapes.append(("wcls", (V, d)))
    return shapes


def read_checkpoint(path):
    """Parse a llama2.c v0 .bin: 7 x i32 header (mod.rs:141-166; sign of vocab_size is
    the shared-classifier flag) then fp32 LE tensors (ram.rs:28-51)."""
    raw = np.memmap(path, dtype=np.uint8, mode="r")
    hdr = np.frombuffer(raw[:28].tobytes(), dtype="<i4")
    if int(hdr[0]) == 0x616b3432:      # "ak42": a v1 / v2 file (export.py:132-260), which the engine does not read
        raise ValueError("llama2.c v1/v2 checkpoint (ak42 header): the reference engine reads the v0 legacy format only")
    vocab = int(hdr[5])
    cfg = Config(int(hdr[0]), int(hdr[1]), int(hdr[2]), int(hdr[3]), int(hdr[4]),
                 abs(vocab), int(hdr[6]), vocab > 0)
    body = np.frombuffer(raw, dtype="<f4", offset=28)
    w, off = {}, 0
    for name, shp in weight_shapes(cfg):
        n = int(np.prod(shp))
        w[name] = np.ascontiguousarray(body[off:off + n]).reshape(shp)
        off += n
    assert off == body.size, (off, body.size)
    if cfg.shared_weight:
        w["wcls"] = w["token_embedding_table"]   # state.rs:111-117
    return cfg, w


class Oracle:
    """Holds host weights + a RunState (ram.rs:7-23, zero-initialised) and steps forward()."""

    def __init__(self, cfg: Config, weights: dict, threads: int = 0):
        self.cfg = cfg
        self.w = {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in weights.items()}
        if "wcls" not in self.w:
            self.w["wcls"] = self.w["token_embedding_table"]
        self._cw = OracleWeights(*[_p(self.w[n]) for n in _W_FIELDS])
        self._cc = cfg.c()
        if threads:
            lib().oracle_set_threads(threads)
        self.reset()

    def reset(self):
        c = self.cfg
        kv = c.n_layers * c.seq_len * c.dim
        sizes = dict(x=c.dim, xb=c.dim, xb2=c.dim, hb=c.hidden_dim, hb2=c.hidden_dim,
                     q=c.dim, k=c.dim, v=c.dim, att=c.n_heads * c.seq_len,
                     logits=c.vocab_size, key_cache=kv, value_cache=kv)
        self.s = {n: np.zeros(sizes[n], dtype=np.float32) for n in _S_FIELDS}
        self._cs = OracleState(*[_p(self.s[n]) for n in _S_FIELDS])

    def forward(self, token: int, pos: int) -> np.ndarray:
        lib().oracle_forward(C.byref(self._cc), C.byref(self._cw), C.byref(self._cs), token, pos)
        return self.s["logits"]

    def forward_range(self, token, pos, layer_begin, layer_end, do_embed, do_cls):
        lib().oracle_forward_range(C.byref(self._cc), C.byref(self._cw), C.byref(self._cs),
                                   token, pos, layer_begin, layer_end, int(do_embed), int(do_cls))

    def forward_f64(self, token: int, pos: int) -> np.ndarray:
        lib().oracle_forward_f64(C.byref(self._cc), C.byref(self._cw), C.byref(self._cs), token, pos)
        return self.s["logits"]

    def multi_head_attention(self, layer: int, pos: int):
        lib().oracle_multi_head_attention(C.byref(self._cc), C.byref(self._cs), layer, pos)

    def generate_greedy(self, prompt_tokens, steps):
        """generate() loop of mod.rs:169-206 at temperature 0: BOS=1 at pos 0, forced
        prompt tokens, argmax afterwards; returns the `next` token of every step."""
        token, out = 1, []
        for pos in range(steps):
            logits = self.forward(token, pos)
            nxt = prompt_tokens[pos] if pos < len(prompt_tokens) else argmax(logits)
            out.append(int(nxt))
            token = nxt
        return out
