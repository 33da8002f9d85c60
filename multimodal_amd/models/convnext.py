"""ConvNeXt-1D backbone on the MI355X hot path.  Same classes, constructor arguments, forward signatures
and state_dict keys as reference models/convnext.py:5-141; the arithmetic runs in libmmhip through
multimodal_amd.functional (channels-last inside, one fused node per block).  nn.Conv1d / nn.Linear /
nn.LayerNorm members are PARAMETER CONTAINERS only (their forward is never called): they give the
reference's key names, shapes and default initialisation."""
import torch
import torch.nn as nn

from .. import functional as FN
from .. import ops


class DropPath(nn.Module):
    """reference models/convnext.py:5-18.  `scale(batch)` is the per-sample factor floor(keep+U)/keep that the
    fused block applies; it draws the same torch.rand the reference draws (same shape, dtype, device, order)."""

    def __init__(self, drop_prob=None):
        super().__init__()
        self.drop_prob = drop_prob

    def scale(self, batch, dtype, device):
        if self.drop_prob == 0. or not self.training:
            return None
        keep = 1 - self.drop_prob
        u = torch.rand((batch, 1, 1), dtype=dtype, device=device)
        if u.is_cuda and dtype == torch.float32:
            return ops.droppath_scale(u, keep)             # one launch for add, floor, div
        r = keep + u
        r.floor_()
        return (r / keep).float().view(batch).contiguous()

    def forward(self, x):
        s = self.scale(x.shape[0], x.dtype, x.device)
        return x if s is None else x * s.to(x.dtype).view((x.shape[0],) + (1,) * (x.ndim - 1))


class Permute(nn.Module):
    def forward(self, x):
        return x.permute(0, 2, 1)


class ConvNeXtBlock1D(nn.Module):
    def __init__(self, dim, drop_path=0., layer_scale_init_value=1e-6):
        super().__init__()
        self.dwconv = nn.Conv1d(dim, dim, kernel_size=7, padding=3, groups=dim)
        self.norm = nn.LayerNorm(dim)
        self.pwconv1 = nn.Linear(dim, 4 * dim)
        self.act = nn.GELU()
        self.pwconv2 = nn.Linear(4 * dim, dim)
        self.gamma = nn.Parameter(layer_scale_init_value * torch.ones((dim)),
                                  requires_grad=True) if layer_scale_init_value > 0 else None
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()

    def forward_blc(self, x):
        """x: [B, L, C] channels-last, compute dtype."""
        s = None
        if isinstance(self.drop_path, DropPath):
            # the reference draws rand in the activation dtype (fp32 there); keep that stream identical
            s = self.drop_path.scale(x.shape[0], torch.float32, x.device)
        return FN.ConvNeXtBlock.apply(x, self.dwconv.weight, self.dwconv.bias, self.norm.weight, self.norm.bias,
                                      self.pwconv1.weight, self.pwconv1.bias, self.pwconv2.weight, self.pwconv2.bias,
                                      self.gamma, s)

    def forward(self, x):
        """x: [B, C, L] as in the reference -> [B, C, L]."""
        dt = FN.get_compute_dtype()
        y = self.forward_blc(x.to(dt).transpose(1, 2).contiguous())
        return y.transpose(1, 2).to(x.dtype)


class ConvNeXt1D(nn.Module):
    def __init__(self, in_chans=1, num_classes=1000, depths=[3, 3, 9, 3], dims=[96, 192, 384, 768], drop_path_rate=0.,
                 layer_scale_init_value=1e-6, head_init_scale=1., regression=False, regression_dim=2):
        super().__init__()
        if in_chans != 1:
            raise NotImplementedError("libmmhip stem kernel supports in_chans=1 (the IR spectrum) only")
        self.regression = regression
        self.downsample_layers = nn.ModuleList()
        self.downsample_layers.append(nn.Sequential(
            nn.Conv1d(in_chans, dims[0], kernel_size=4, stride=4), Permute(), nn.LayerNorm(dims[0]), Permute()))
        for i in range(3):
            self.downsample_layers.append(nn.Sequential(
                Permute(), nn.LayerNorm(dims[i]), Permute(), nn.Conv1d(dims[i], dims[i + 1], kernel_size=2, stride=2)))
        self.stages = nn.ModuleList()
        dp_rates = [x.item() for x in torch.linspace(0, drop_path_rate, sum(depths))]
        cur = 0
        for i in range(4):
            self.stages.append(nn.Sequential(*[
                ConvNeXtBlock1D(dim=dims[i], drop_path=dp_rates[cur + j], layer_scale_init_value=layer_scale_init_value)
                for j in range(depths[i])]))
            cur += depths[i]
        self.norm = nn.LayerNorm(dims[-1])
        self.head = nn.Linear(dims[-1], regression_dim if regression else num_classes)
        self.apply(self._init_weights)
        self.head.weight.data.mul_(head_init_scale)
        self.head.bias.data.mul_(head_init_scale)

    def _init_weights(self, m):
        if isinstance(m, (nn.Conv1d, nn.Linear)):
            nn.init.trunc_normal_(m.weight, std=.02)
            nn.init.constant_(m.bias, 0)

    def _features_blc(self, x):
        """[B,1,Lin] or [B,Lin] f32 -> [B, L_last, C_last] channels-last, compute dtype."""
        dt = FN.get_compute_dtype()
        if x.dim() == 3:
            x = x[:, 0]
        stem = self.downsample_layers[0]
        y = FN.StemLN.apply(x.float().contiguous(), stem[0].weight, stem[0].bias, stem[2].weight, stem[2].bias, dt)
        for i in range(4):
            if i > 0:
                ds = self.downsample_layers[i]
                cw = ds[3].weight                                   # (Cout, Cin, 2) -> tap-major (Cout, 2*Cin)
                w2 = cw.permute(0, 2, 1).reshape(cw.shape[0], -1)
                y = FN.LNLinear.apply(y, ds[1].weight, ds[1].bias, w2, ds[3].bias, 2)
            for blk in self.stages[i]:
                y = blk.forward_blc(y)
        return y

    def forward_features(self, x):
        y = self._features_blc(x).float().mean(1)
        return FN.LayerNorm.apply(y.to(FN.get_compute_dtype()), self.norm.weight, self.norm.bias)

    def forward(self, x, keep_sequence=False):
        y = self._features_blc(x)
        if keep_sequence:
            if self.regression:
                return FN.LNLinear.apply(y, self.norm.weight, self.norm.bias, self.head.weight, self.head.bias, 1)
            return FN.LayerNorm.apply(y, self.norm.weight, self.norm.bias)
        z = y.float().mean(1, keepdim=True).to(y.dtype)             # global pooling branch (unused by the model)
        if self.regression:
            return FN.LNLinear.apply(z, self.norm.weight, self.norm.bias, self.head.weight, self.head.bias, 1)[:, 0]
        return FN.LayerNorm.apply(z, self.norm.weight, self.norm.bias)[:, 0]
