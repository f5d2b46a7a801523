"""NeRFNetwork on the HIP operators: the fused-MLP variant of the reference (nerf/network_ff.py:12-81).

hash grid (L=16, F=2, T=2^19, finest resolution 2048*bound) -> sigma FFMLP (32->64->64->16) -> trunc_exp;
SH(4) of the view direction + 15 geometry features + 1 zero pad -> colour FFMLP (32->64->64->64->16) -> sigmoid.
Parameter names match the reference (`encoder.embeddings`, `sigma_net.weights`, `color_net.weights`).
"""
import torch
import torch.nn as nn

from .activation import trunc_exp
from .encoding import get_encoder
from .ffmlp import FFMLP, nerf_head


class NeRFNetwork(nn.Module):
    def __init__(self, encoding="hashgrid", encoding_dir="sphere_harmonics", num_layers=2, hidden_dim=64, geo_feat_dim=15,
                 num_layers_color=3, hidden_dim_color=64, bound=1, num_levels=16, log2_hashmap_size=19):
        super().__init__()
        self.bound = bound
        self.num_layers = num_layers
        self.hidden_dim = hidden_dim
        self.geo_feat_dim = geo_feat_dim
        self.encoder, self.in_dim = get_encoder(encoding, desired_resolution=2048 * bound, num_levels=num_levels,
                                                log2_hashmap_size=log2_hashmap_size)
        self.sigma_net = FFMLP(input_dim=self.in_dim, output_dim=1 + self.geo_feat_dim, hidden_dim=self.hidden_dim,
                               num_layers=self.num_layers)
        self.num_layers_color = num_layers_color
        self.hidden_dim_color = hidden_dim_color
        self.encoder_dir, self.in_dim_color = get_encoder(encoding_dir)
        self.in_dim_color += self.geo_feat_dim + 1          # padded to 32 (network_ff.py:42)
        self.color_net = FFMLP(input_dim=self.in_dim_color, output_dim=3, hidden_dim=self.hidden_dim_color,
                               num_layers=self.num_layers_color)
        # MI355X: the default architecture runs everything after the encoder as ONE kernel (ffmlp/head.py);
        # set fused_head = False for the operator-by-operator path of the reference.
        self.fused_head = (encoding_dir == "sphere_harmonics" and self.in_dim == 32 and self.in_dim_color == 32
                           and num_layers == 2 and num_layers_color == 3 and hidden_dim == 64 and hidden_dim_color == 64
                           and geo_feat_dim == 15 and getattr(self.encoder_dir, "degree", 0) == 4)

    def forward(self, x, d):
        """x [N,3] in [-bound,bound], d [N,3] unit -> sigma [N] fp32, rgb [N,3]   (network_ff.py:51-81)"""
        x = self.encoder(x, bound=self.bound)
        if self.fused_head and x.is_cuda and x.shape[0] % 16 == 0 and x.dtype == torch.half and not d.requires_grad:
            return nerf_head(x, d, self.sigma_net.weights, self.color_net.weights)
        h = self.sigma_net(x)
        sigma = trunc_exp(h[..., 0])
        geo_feat = h[..., 1:]
        d = self.encoder_dir(d)
        p = torch.zeros_like(geo_feat[..., :1])
        h = torch.cat([d.to(geo_feat.dtype), geo_feat, p], dim=-1)
        h = self.color_net(h)
        rgb = torch.sigmoid(h)
        return sigma, rgb

    def density(self, x):
        """network_ff.py:83-96"""
        x = self.encoder(x, bound=self.bound)
        h = self.sigma_net(x)
        return {"sigma": trunc_exp(h[..., 0]), "geo_feat": h[..., 1:]}

    def get_params(self, lr):
        return [{"params": self.encoder.parameters(), "lr": lr}, {"params": self.sigma_net.parameters(), "lr": lr},
                {"params": self.color_net.parameters(), "lr": lr}]
