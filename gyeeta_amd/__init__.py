"""gyeeta_amd: MI355X-native streaming-sketch aggregation engine for Gyeeta's madhava/shyama roll-up hot path.

The product is the C-ABI shared library gyeeta_amd/lib/libgysketch.so (include/gysketch.h; HIP kernels in gyeeta_amd/csrc).
This package is the thin host-side plumbing used by tests and bench.py: ctypes binding (capi), the engine wrapper with the
multi-GPU window reduce over torch.distributed/RCCL (engine), and numpy builders for the reference's wire records (wire)."""
from . import capi  # noqa: F401

__all__ = ["capi", "engine", "wire", "build"]
