"""sailor_amd -- MI355X-native Forward+ lighting path for the Sailor engine (light cull, PBR shade, CSM, ECS sweep).

The arithmetic lives in hand-written HIP kernels behind the C-ABI of include/sailor_hip.h (sailor_amd/csrc); this
package binds that ABI (`_lib`), exposes the host-side set-up (`host`), the synthetic frame generator (`synth`) and the
torch-facing path objects (`forward_plus`, `dist`).  There is no CPU fallback.
"""
from . import _lib, host  # noqa: F401

__all__ = ["_lib", "host"]
