"""Top-level model: same surface as reference models/multimodal_to_smiles.py:9-112, plus the KV-cached
`generate` (new API; legality: SURVEY.md A9)."""
import torch as th
import torch.nn as nn

from .spectral_encoder import MultimodalSpectralEncoder
from .transformer_decoder import SMILESDecoder


class MultiModalToSMILESModel(nn.Module):
    def __init__(self, smiles_vocab_size, nmr_vocab_size, max_seq_length=512, max_nmr_length=128, max_memory_length=128,
                 embed_dim=768, num_heads=8, num_layers=6, dropout=0.1, verbose=False, use_stablemax=False):
        super().__init__()
        self.verbose = verbose
        self.encoder = MultimodalSpectralEncoder(embed_dim=embed_dim, verbose=verbose)
        self.decoder = SMILESDecoder(
            smiles_vocab_size=smiles_vocab_size, nmr_vocab_size=nmr_vocab_size, max_seq_length=max_seq_length,
            max_nmr_length=max_nmr_length, max_memory_length=max_memory_length, memory_dim=embed_dim, embed_dim=embed_dim,
            num_heads=num_heads, num_layers=num_layers, dropout=dropout, verbose=verbose, use_stablemax=use_stablemax)

    def forward(self, nmr_tokens, ir_data, target_seq=None, target_mask=None):
        """target_mask is accepted and ignored, exactly as in the reference (:74)."""
        memory = self.encoder(None, ir_data, None)
        return self.decoder(target_seq, memory, nmr_tokens)

    @th.no_grad()
    def generate(self, nmr_tokens, ir_data, max_len=128, bos_id=12, eos_id=13, early_stop=True, graph=None):
        """Greedy decode with a KV cache; returns LongTensor [B, <=max_len+1] starting with BOS.  Token ids equal the
        reference loop's (train_autoregressive.py:80-116); rows keep emitting argmax ids after their EOS, as the
        reference's `current_token` does -- cut with `multimodal_amd.generate.ids_to_strings`."""
        from ..generate import greedy_generate
        return greedy_generate(self, nmr_tokens, ir_data, max_len=max_len, bos_id=bos_id, eos_id=eos_id,
                               early_stop=early_stop, graph=graph)
