"""adaptigraph_amd — MI355X-native engine for AdaptiGraph's message-passing rollout (hot path only).

Submodules are imported lazily so that `adaptigraph_amd.synth` (pure numpy) stays usable
without the HIP library; anything that computes loads `libadaptigraph_hip.so` and fails loudly
if it is missing (there is no CPU fallback in the product path).
"""
__all__ = ["synth"]
