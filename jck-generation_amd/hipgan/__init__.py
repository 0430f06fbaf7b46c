"""hipgan - Python binding of libjckgan_hip.so (the MI355X C-ABI library, include/jckgan.h).

The product path has NO CPU fallback: every op raises if the HIP library is missing or no GPU is
visible.  PyTorch is used for device memory, streams and torch.distributed only.
"""
from ._lib import lib, JckError, PREC_BF16, PREC_F32, load_library  # noqa: F401
