from .style_encoder import LAENeRF, palette_recompose  # noqa: F401
