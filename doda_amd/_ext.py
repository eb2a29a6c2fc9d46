"""Optional compiled glue (doda_amd/_doda_torch.so, built by doda_amd.build.build_torch_ext): C++
autograd functions over the same C ABI.  `ext` is None when it is not built or DODA_NO_EXT=1; the
Python/ctypes glue in doda_amd.ops is then used — both routes launch the same HIP kernels."""
import os

ext = None
if os.environ.get("DODA_NO_EXT", "0") != "1":
    try:
        import torch  # noqa: F401  (libtorch must be loaded first)
        from . import _lib
        _lib.lib()     # loads libdoda_hip.so (fails loudly if missing)
        from . import _doda_torch as ext  # type: ignore
        if ext.abi_version() != _lib.ABI_VERSION or getattr(ext, "built_for_abi", lambda: -1)() != _lib.ABI_VERSION:
            ext = None
    except ImportError:
        ext = None
