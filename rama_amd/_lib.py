"""ctypes binding of librama_hip.so (C ABI declared in include/rama_hip.h).

The product path has no fallback: if the shared library is missing or a call fails,
this raises.  Nothing here imports the CPU oracle.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_HERE = Path(__file__).resolve().parent
# (RAMA_HIP_LIB: another build of the same library -- A/B runs of compile-time variants, tools/ab_lib.sh; never a fallback)
LIB_PATH = Path(os.environ["RAMA_HIP_LIB"]) if os.environ.get("RAMA_HIP_LIB") else _HERE / "librama_hip.so"

f32p = C.POINTER(C.c_float)
i32p = C.POINTER(C.c_int32)


class RamaError(RuntimeError):
    pass


class rama_config(C.Structure):
    """engine/src/transformer/mod.rs:128-138"""
    _fields_ = [(n, C.c_int32) for n in (
        "dim", "hidden_dim", "n_layers", "n_heads", "n_kv_heads", "vocab_size", "seq_len", "shared_weight")]


W_FIELDS = ("token_embedding_table", "rms_att_weight", "rms_ffn_weight",
            "wq", "wk", "wv", "wo", "w1", "w2", "w3",
            "rms_final_weight", "freq_cis_real", "freq_cis_imag", "wcls")
S_FIELDS = ("x", "xb", "xb2", "hb", "hb2", "q", "k", "v", "att", "logits", "key_cache", "value_cache")


class rama_weights(C.Structure):
    """engine/src/transformer/state.rs:53-74 (device pointers)"""
    _fields_ = [(n, C.c_void_p) for n in W_FIELDS]


class rama_run_state(C.Structure):
    """engine/src/transformer/state.rs:3-17 (device pointers)"""
    _fields_ = [(n, C.c_void_p) for n in S_FIELDS]


class rama_stage(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("layer_begin", "layer_end", "do_embed", "do_cls")]


class rama_pipe_plan(C.Structure):
    _fields_ = [("n_seq", C.c_int32), ("n_pos", C.c_int32), ("wrap", C.c_int32), ("prompt", C.POINTER(C.c_int32)), ("n_prompt", C.c_int32),
                ("temperature", C.c_float), ("topp", C.c_float), ("u", C.c_float), ("out_tokens_dev", C.c_void_p)]


class rama_pipe_tick(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("on", "seq", "pos", "pos_wrapped", "token_kind", "token", "samples", "send_kind", "send_seq", "send_peer",
                                         "recv_kind", "recv_seq", "recv_peer", "recv_pos")]


# name -> (restype, argtypes); every symbol include/rama_hip.h declares
_vp, _sz, _int = C.c_void_p, C.c_size_t, C.c_int
_cfgp, _wp, _sp, _stp = C.POINTER(rama_config), C.POINTER(rama_weights), C.POINTER(rama_run_state), C.POINTER(rama_stage)
SIGNATURES = {
    "rama_ctx_create": (_int, [_int, _vp, C.POINTER(_vp)]),
    "rama_ctx_destroy": (_int, [_vp]),
    "rama_sync": (_int, [_vp]),
    "rama_stream_query": (_int, [_vp]),
    "rama_last_error": (C.c_char_p, []),
    "rama_device_info": (_int, [_vp, C.c_char_p, C.POINTER(_int), C.POINTER(_sz)]),
    "rama_alloc_f32": (_int, [_vp, _sz, C.POINTER(_vp)]),
    "rama_upload_f32": (_int, [_vp, _vp, _sz, C.POINTER(_vp)]),
    "rama_copy_h2d_f32": (_int, [_vp, _vp, _vp, _sz]),
    "rama_download_f32": (_int, [_vp, _vp, _sz, _vp]),
    "rama_free": (_int, [_vp, _vp]),
    "rama_array_add": (_int, [_vp, _vp, _vp, _sz]),
    "rama_array_mult": (_int, [_vp, _vp, _vp, _sz]),
    "rama_sinu": (_int, [_vp, _vp, _sz]),
    "rama_copy_from_slice": (_int, [_vp, _vp, _vp, _sz]),
    "rama_rmsnorm": (_int, [_vp, _vp, _vp, _vp, _sz]),
    "rama_apply_position": (_int, [_vp, _vp, _vp, _vp, _vp, _sz]),
    "rama_matmul": (_int, [_vp, _vp, _vp, _vp, _sz, _sz, _sz]),
    "rama_softmax": (_int, [_vp, _vp, _sz]),
    "rama_multi_head_attention": (_int, [_vp, _vp, _vp, _vp, _vp, _vp] + [_int] * 6),
    "rama_sample_argmax": (_int, [_vp, _vp, _sz, i32p]),
    "rama_sample_topp": (_int, [_vp, _vp, _sz, C.c_float, C.c_float, C.c_float, i32p]),
    "rama_model_load": (_int, [_vp, C.c_char_p, C.POINTER(_vp)]),
    "rama_model_load_stage": (_int, [_vp, C.c_char_p, _stp, C.POINTER(_vp)]),
    "rama_model_synth": (_int, [_vp, _cfgp, C.c_uint64, _stp, _vp, _vp, C.POINTER(_vp)]),
    "rama_model_save": (_int, [_vp, _vp, C.c_char_p]),
    "rama_model_config": (_int, [_vp, _cfgp]),
    "rama_model_weights": (_int, [_vp, _wp]),
    "rama_model_bytes": (_sz, [_vp]),
    "rama_model_free": (_int, [_vp, _vp]),
    "rama_model_release_copies": (_int, [_vp, _vp, _int]),
    "rama_state_create": (_int, [_vp, _cfgp, _int, _sp]),
    "rama_state_free": (_int, [_vp, _sp]),
    "rama_fill_synth": (_int, [_vp, _vp, _sz, C.c_uint64, C.c_uint64, C.c_uint64, C.c_float, C.c_float]),
    "rama_forward": (_int, [_vp, _cfgp, _wp, _sp, _int, _int]),
    "rama_forward_stage": (_int, [_vp, _cfgp, _wp, _sp, _int, _int, _stp]),
    "rama_prefill": (_int, [_vp, _cfgp, _wp, _sp, i32p, _int, _int]),
    "rama_decode_batch": (_int, [_vp, _cfgp, _wp, _sp, i32p, i32p, _int]),
    "rama_decode_batch_begin": (_int, [_vp, _cfgp, _wp, _sp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _int, _int]),
    "rama_decode_batch_steps": (_int, [_vp, _int]),
    "rama_decode_batch_tokens": (_int, [_vp, C.POINTER(C.c_int32), _int, C.POINTER(_int)]),
    "rama_forward_stage_devtok": (_int, [_vp, _cfgp, _wp, _sp, _vp, _int, _stp]),
    "rama_argmax_dev": (_int, [_vp, _vp, _sz, _vp]),
    "rama_generate_greedy": (_int, [_vp, _cfgp, _wp, _sp, i32p, _int, _int, i32p]),
    "rama_generate": (_int, [_vp, _cfgp, _wp, _sp, i32p, _int, _int, C.c_float, C.c_float, C.c_float, i32p]),
    "rama_sample_topp_dev": (_int, [_vp, _vp, _sz, C.c_float, C.c_float, C.c_float, _vp]),
    "rama_decode_sampler": (_int, [_vp, C.c_float, C.c_float, C.c_float]),
    "rama_decode_begin": (_int, [_vp, _int, _int, i32p, _int]),
    "rama_decode_steps": (_int, [_vp, _cfgp, _wp, _sp, _int]),
    "rama_decode_tokens": (_int, [_vp, i32p, _int, C.POINTER(_int)]),
    "rama_decode_stream_poll": (_int, [_vp, _int, i32p, _int, C.POINTER(_int)]),
    "rama_decode_batch_stream_poll": (_int, [_vp, _int, _int, i32p, _int, C.POINTER(_int)]),
    "rama_generate_stream": (_int, [_vp, _cfgp, _wp, _sp, i32p, _int, _int, C.c_float, C.c_float, C.c_float,
                                    C.CFUNCTYPE(None, C.c_void_p, _int, C.c_int32), C.c_void_p, i32p]),
    "rama_set_graph_mode": (_int, [_vp, _int]),
    "rama_set_tuning": (_int, [_vp, C.c_char_p, _int]),
    "rama_ref_expf": (_int, [_vp, _vp, _vp, _sz]),
    "rama_pipe_unique_id": (_int, [_vp]),
    "rama_pipe_create": (_int, [_vp, _vp, _int, _int, C.POINTER(_vp)]),
    "rama_pipe_destroy": (_int, [_vp]),
    "rama_pipe_exchange": (_int, [_vp, _vp, _sz, _int, _vp, _sz, _int, _vp, _int, _vp, _int]),
    "rama_pipe_item": (_int, [_vp, _int, _int, _int, C.POINTER(_int), C.POINTER(_int)]),
    "rama_pipe_total_ticks": (_int, [_vp, _vp]),
    "rama_pipe_plan_ticks": (_int, [_vp, _int]),
    "rama_pipe_tick_plan": (_int, [_vp, _int, _int, _int, C.POINTER(rama_pipe_tick)]),
    "rama_pipe_comm_info": (_int, [_vp, C.POINTER(_int), C.POINTER(_int)]),
    "rama_pipe_run_ticks": (_int, [_vp, _cfgp, _wp, _sp, C.POINTER(_vp), _stp, _vp, _int, _int]),
    "rama_pipe_last_error": (C.c_char_p, []),
    "rama_timer_start": (_int, [_vp]),
    "rama_timer_stop": (_int, [_vp, C.POINTER(C.c_float)]),
    "rama_kprof_enable": (_int, [_vp, _int, _int]),
    "rama_kprof_read": (_int, [_vp, C.POINTER(_int), C.POINTER(C.c_double)]),
}

_lib = None


def load() -> C.CDLL:
    """dlopen the in-tree library.  Raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise RamaError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                        f"or `make -C rama_amd/csrc`")
    L = C.CDLL(str(LIB_PATH))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)          # AttributeError if the library lacks a declared symbol
        fn.restype, fn.argtypes = res, args
    _lib = L
    return L


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().rama_last_error()
        raise RamaError(f"{what or 'rama call'} failed ({rc}): {msg.decode() if msg else ''}")
