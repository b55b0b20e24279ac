"""quandary_amd — MI355X-native forward/adjoint propagator for Quandary's hot path.

The product is ``csrc/libquandary_amd.so`` (hand-written HIP kernels for gfx950
behind the C ABI of ``include/quandary_amd.h``).  The Python modules here are
plumbing only: ``capi`` binds the C ABI with ctypes (no CPU fallback — calls
raise if the library or a GPU is missing), ``config`` turns a reference config
file into the C-ABI structs.
"""
__version__ = "0.1.0"
