from .style_encoder import LAENeRF, palette_recompose  # noqa: F401
from .editgrid import EditGrid  # noqa: F401
from .edit_dataset import extract_view, extract_views, select_edit_pixels  # noqa: F401
