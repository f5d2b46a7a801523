"""get_encoder (reference encoding.py:45-82) on the HIP encoders."""
from .freqencoder import FreqEncoder
from .gridencoder import GridEncoder
from .shencoder import SHEncoder


def get_encoder(encoding, input_dim=3, multires=6, degree=4, num_levels=16, level_dim=2, base_resolution=16,
                log2_hashmap_size=19, desired_resolution=2048, align_corners=False, **kwargs):
    if encoding == "None":
        return (lambda x, **kw: x), input_dim
    if encoding == "frequency":
        encoder = FreqEncoder(input_dim=input_dim, degree=multires)           # encoding.py:59-62
    elif encoding == "sphere_harmonics":
        encoder = SHEncoder(input_dim=input_dim, degree=degree)
    elif encoding in ("hashgrid", "tiledgrid"):
        encoder = GridEncoder(input_dim=input_dim, num_levels=num_levels, level_dim=level_dim,
                              base_resolution=base_resolution, log2_hashmap_size=log2_hashmap_size,
                              desired_resolution=desired_resolution,
                              gridtype="hash" if encoding == "hashgrid" else "tiled", align_corners=align_corners)
    else:
        # 'ash' is out of scope for the hot path
        raise NotImplementedError(f"encoding {encoding!r} is not part of the MI355X hot path "
                                  "[None, frequency, sphere_harmonics, hashgrid, tiledgrid]")
    return encoder, encoder.output_dim
