"""mistral.rs_amd -- MI355X-native (gfx950) quantized-inference hot path for mistral.rs.

Product = the C-ABI shared libraries under lib/ (hand-written HIP, see csrc/) that export the
reference's own `extern "C"` kernel-launcher symbols, plus a host-side mirror of the reference's
operator interface.  PyTorch is used only as plumbing (device buffers, streams, torch.distributed).
Nothing in this package imports the CPU oracle; every op raises if its HIP library is missing.
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
