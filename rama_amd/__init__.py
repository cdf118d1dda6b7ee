"""rama_amd -- MI355X-native (gfx950) backend for oliverhu/rama's fp32 Llama-2 decode path.

Layout: csrc/ (HIP kernels + the C ABI of include/rama_hip.h, built into librama_hip.so),
transformer.py (host mirror of the reference's Device/forward/generate interface),
engine.py (resident-model fused decode path).  There is no CPU fallback: importing the
ops without the built library raises.
"""
from ._lib import RamaError, load  # noqa: F401
from .transformer import (Config, Hip, HipSlice, MutView, RunState, RunStateView,  # noqa: F401
                          TransformerWeights, TransformerWeightsView, View, forward,
                          forward_fused, generate, generate_device, generate_greedy_device)
from .engine import Engine, Model, algorithmic_bytes, decode_batch, decode_batch_chained  # noqa: F401
