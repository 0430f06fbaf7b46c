"""Device selection (reference utils.py:4-8).  On ROCm `torch.cuda.is_available()` is the HIP check."""
import torch


def get_default_device() -> torch.device:
    return torch.device("cuda" if torch.cuda.is_available() else "cpu")


def require_gpu(what: str) -> torch.device:
    """The training hot path has no CPU fallback: fail loudly instead of silently running ATen on the host."""
    dev = get_default_device()
    if dev.type != "cuda":
        from hipgan import JckError
        raise JckError(f"{what} needs an MI355X (HIP) device: the gfx950 kernels are the only implementation of the hot path")
    return dev
