"""Minimal stand-in for torchtext 0.2.3 so that the reference package can be
IMPORTED in the build container (oracle pinning only; never shipped, never used
by the product path).  Only the names the reference touches at import time and
in the fake-batch path are provided."""
from . import data, vocab  # noqa: F401
