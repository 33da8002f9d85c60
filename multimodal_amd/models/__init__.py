from .convnext import ConvNeXt1D, ConvNeXtBlock1D, DropPath, Permute
from .cross_attention import CrossAttentionBlock
from .multimodal_to_smiles import MultiModalToSMILESModel
from .smiles_tokenizer import BasicSmilesTokenizer, SmilesTokenizer, load_vocab
from .spectral_encoder import MultimodalSpectralEncoder
from .transformer_decoder import (DecoderPromptLayer, RotaryEmbedding, SMILESDecoder, apply_rotary_pos_emb, rotate_half,
                                  stable_s, stablemax)

__all__ = ["ConvNeXt1D", "ConvNeXtBlock1D", "DropPath", "Permute", "CrossAttentionBlock", "MultiModalToSMILESModel",
           "BasicSmilesTokenizer", "SmilesTokenizer", "load_vocab", "MultimodalSpectralEncoder", "DecoderPromptLayer",
           "RotaryEmbedding", "SMILESDecoder", "apply_rotary_pos_emb", "rotate_half", "stable_s", "stablemax"]
