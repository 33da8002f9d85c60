"""IR spectral encoder wrapper: same surface as reference models/spectral_encoder.py:7-50."""
import torch.nn as nn

from .convnext import ConvNeXt1D


class MultimodalSpectralEncoder(nn.Module):
    def __init__(self, embed_dim=768, verbose=True):
        super().__init__()
        self.verbose = verbose
        self.ir_backbone = ConvNeXt1D(in_chans=1, depths=[3, 3, 6, 3], dims=[64, 128, 256, embed_dim], drop_path_rate=0.1,
                                      layer_scale_init_value=1e-6, regression=True, regression_dim=embed_dim)

    def forward(self, nmr_data, ir_data, c_nmr_data):
        """Only ir_data is used (reference :31-50): [B,1000] or [B,1,1000] -> [B,31,E] in the compute dtype."""
        if ir_data is None:
            return None
        if ir_data.dim() == 2:
            ir_data = ir_data.unsqueeze(1)
        emb = self.ir_backbone(ir_data, keep_sequence=True)
        if self.verbose:
            print(f"IR embedding: {emb.shape}")
        return emb
