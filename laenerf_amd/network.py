"""NeRFNetwork on the HIP operators: the fused-MLP variant of the reference (nerf/network_ff.py:12-139) and, as
`NeRFNetworkLinear`, the default `nn.Linear` variant the shipped scripts run (nerf/network.py:33-165).

hash grid (L=16, F=2, T=2^19, finest resolution 2048*bound) -> sigma FFMLP (32->64->64->16) -> trunc_exp;
SH(4) of the view direction + 15 geometry features + 1 zero pad -> colour FFMLP (32->64->64->64->16) -> sigmoid.
Parameter names match the reference (`encoder.embeddings`, `sigma_net.weights`, `color_net.weights`).
"""
import torch
import torch.nn as nn

from .activation import trunc_exp
from .encoding import get_encoder
from .ffmlp import FFMLP, nerf_density, nerf_head
from .field import field_backward_plan, field_density, field_supported, nerf_field


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
        self.fused_field = self.fused_head and encoding == "hashgrid" and field_supported(self.encoder, self.sigma_net, self.color_net)

    def _field_ok(self, x, d):
        return (self.fused_field and self.fused_head and x.is_cuda and x.shape[0] % 16 == 0 and torch.is_autocast_enabled("cuda")
                and not x.requires_grad and not d.requires_grad)

    def plan_backward(self, x):
        """MI355X-native: the position-only half of the table-gradient pass for the samples x (None when the fused field
        op does not apply); hand the result to forward(x, d, plan=...)"""
        if not (self.fused_field and self.fused_head and x.is_cuda and x.shape[0] % 16 == 0):
            return None
        return field_backward_plan(x.view(-1, 3), self.encoder, self.bound)

    def forward(self, x, d, plan=None):
        """x [N,3] in [-bound,bound], d [N,3] unit -> sigma [N] fp32, rgb [N,3]   (network_ff.py:51-81)"""
        if self._field_ok(x, d):
            # encoder + head as one op: features stay level-major between the grid and MLP kernels (field.py)
            return nerf_field(x.view(-1, 3), d, self.encoder, self.sigma_net, self.color_net, self.bound, plan=plan)
        x = self.encoder(x, bound=self.bound)
        if self.fused_head and x.is_cuda and x.shape[0] % 16 == 0 and x.dtype == torch.half and not d.requires_grad:
            return nerf_head(x, d, self.sigma_net.weights, self.color_net.weights, 1.0, self.sigma_net.shadow, self.color_net.shadow)
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
        if self.fused_head and x.is_cuda and x.dtype == torch.half and not torch.is_grad_enabled():
            sh = self.sigma_net.shadow
            sigma, h = nerf_density(x, self.sigma_net.weights if sh is None else sh.table_half(self.sigma_net.weights))
            return {"sigma": sigma, "geo_feat": h[..., 1:]}
        h = self.sigma_net(x)
        return {"sigma": trunc_exp(h[..., 0]), "geo_feat": h[..., 1:]}

    @torch.no_grad()
    def density_sigma(self, x):
        """MI355X-native: `density(x)["sigma"]` alone (what update_extra_state reads, nerf/renderer.py:594, 625) -- same bits,
        without the feature transpose and without storing geo_feat"""
        if self.fused_field and x.is_cuda and torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.half:
            return field_density(x.view(-1, 3), self.encoder, self.sigma_net, self.bound)
        return self.density(x)["sigma"]

    def color(self, x, d, mask=None, geo_feat=None, **kwargs):
        """masked colour query of the `run` path (network_ff.py:98-139): rows outside `mask` stay zero"""
        return _masked_color(self, d, mask, geo_feat)

    def _color_rows(self, d, geo_feat):
        d = self.encoder_dir(d)
        h = torch.cat([d.to(geo_feat.dtype), geo_feat, torch.zeros_like(geo_feat[..., :1])], dim=-1)
        return torch.sigmoid(self.color_net(h))

    def get_params(self, lr):
        """network_ff.py:142-154: four groups, the third one (direction encoder) without parameters -- kept so that group
        indices line up with a checkpoint of the reference's optimizer"""
        return [{"params": list(self.encoder.parameters()), "lr": lr}, {"params": list(self.sigma_net.parameters()), "lr": lr},
                {"params": list(self.encoder_dir.parameters()), "lr": lr}, {"params": list(self.color_net.parameters()), "lr": lr}]


def _masked_color(net, d, mask, geo_feat):
    if mask is None:
        return net._color_rows(d, geo_feat)
    rgbs = torch.zeros(mask.shape[0], 3, dtype=torch.float32, device=d.device)
    if not mask.any():
        return rgbs
    rgbs[mask] = net._color_rows(d[mask], geo_feat[mask]).to(rgbs.dtype)
    return rgbs


class NeRFNetworkLinear(nn.Module):
    """The reference's default network (nerf/network.py:33-165): bias-free `nn.Linear` chains (hipBLASLt GEMMs under
    autocast) -- sigma: in -> 64 -> 1+15 (num_layers 2), colour: 16+15 -> 64 -> 64 -> 3 (num_layers_color 3).  It is
    the arithmetic the fused MLP replaces and what `ffmlp` is checked against."""

    def __init__(self, encoding="hashgrid", encoding_dir="sphere_harmonics", num_layers=2, hidden_dim=64, geo_feat_dim=15,
                 num_layers_color=3, hidden_dim_color=64, bound=1, num_levels=16, log2_hashmap_size=19):
        super().__init__()
        self.bound = bound
        self.geo_feat_dim = geo_feat_dim
        self.encoder, self.in_dim = get_encoder(encoding, desired_resolution=2048 * bound, num_levels=num_levels,
                                                log2_hashmap_size=log2_hashmap_size)
        dims = [self.in_dim] + [hidden_dim] * (num_layers - 1) + [1 + geo_feat_dim]
        self.sigma_net = nn.ModuleList([nn.Linear(a, b, bias=False) for a, b in zip(dims[:-1], dims[1:])])
        self.encoder_dir, self.in_dim_dir = get_encoder(encoding_dir)
        dims = [self.in_dim_dir + geo_feat_dim] + [hidden_dim_color] * (num_layers_color - 1) + [3]
        self.color_net = nn.ModuleList([nn.Linear(a, b, bias=False) for a, b in zip(dims[:-1], dims[1:])])

    @staticmethod
    def _chain(layers, h):
        for i, layer in enumerate(layers):
            h = layer(h)
            if i != len(layers) - 1:
                h = torch.relu(h)
        return h

    def forward(self, x, d):
        """network.py:95-124"""
        out = self.density(x)
        return out["sigma"], self._color_rows(d, out["geo_feat"])

    def density(self, x):
        """network.py:126-143"""
        h = self._chain(self.sigma_net, self.encoder(x, bound=self.bound))
        return {"sigma": trunc_exp(h[..., 0]), "geo_feat": h[..., 1:]}

    def _color_rows(self, d, geo_feat):
        d = self.encoder_dir(d)
        return torch.sigmoid(self._chain(self.color_net, torch.cat([d.to(geo_feat.dtype), geo_feat], dim=-1)))

    def color(self, x, d, mask=None, geo_feat=None, **kwargs):
        """network.py:145-180"""
        return _masked_color(self, d, mask, geo_feat)

    def get_params(self, lr):
        """network.py:182-195 (same four groups as network_ff.py)"""
        return [{"params": list(self.encoder.parameters()), "lr": lr}, {"params": list(self.sigma_net.parameters()), "lr": lr},
                {"params": list(self.encoder_dir.parameters()), "lr": lr}, {"params": list(self.color_net.parameters()), "lr": lr}]
