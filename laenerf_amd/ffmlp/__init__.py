from .ffmlp import FFMLP, ffmlp_forward, convert_activation  # noqa: F401
from .head import nerf_head, nerf_density  # noqa: F401
