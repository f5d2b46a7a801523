from .style_encoder import LAENeRF, palette_recompose  # noqa: F401
from .editgrid import EditGrid  # noqa: F401
