from .ffmlp import FFMLP, ffmlp_forward, convert_activation  # noqa: F401
