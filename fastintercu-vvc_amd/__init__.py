"""MI355X-native MLT-CNN inter-CU split predictor (hot path of smu-ivpl/FastInterCU-VVC).

The directory name carries a hyphen (it mirrors the reference repo's name), so import it
through `mltcnn_pkg.load()` at the repo root, which registers it as `fastintercu_vvc_amd`.
"""
from . import build, capi, shard, synth, weights  # noqa: F401
from .capi import MltCnn, MltError  # noqa: F401

__all__ = ["build", "capi", "shard", "synth", "weights", "MltCnn", "MltError"]
