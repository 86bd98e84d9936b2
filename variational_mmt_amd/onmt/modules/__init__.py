from .Dists import Normal  # noqa: F401
from . import SRU  # noqa: F401,E402
