from .vitlens import ViTLens  # noqa: F401
