def time_to_str(time_diff: float) -> str:
    """'<h>h <m>m <s>' exactly as the reference formats the total training time (logger/utils.py:1-2)."""
    hours, rest = time_diff // 3600, time_diff % 3600
    return f"{hours}h {rest // 60}m {rest % 60}"
