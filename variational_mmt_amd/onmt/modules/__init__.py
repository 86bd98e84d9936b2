from .Dists import Normal  # noqa: F401
