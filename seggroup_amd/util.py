"""`IOStream` logger: drop-in for the reference's `seggroup/util.py:41-51` (print + append + flush).
The training losses of that file are out of scope."""
from .infer import IOStream  # noqa: F401
