"""doda_amd — MI355X-native sparse-conv training path behind DODA's operator API.

Sub-modules: spconv (spconv-v1.2-shaped modules), pg_op / pointops2_cuda (extension-shaped
functions), pointgroup_ops / pointops2 (host-side wrapper mirrors), ops (tensor-level C-ABI calls),
model (SparseConv U-Net counterpart), scene (synthetic ScanNet-shaped batches).
The native library is loaded lazily by doda_amd._lib.lib(); there is no CPU fallback.
"""
__version__ = "0.1.0"
