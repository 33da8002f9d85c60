"""MI355X-native hot path of tomoqt/multimodal (spectra -> SMILES): ConvNeXt-1D encoder, prompt-decoder attention,
KV-cached greedy decode, data-parallel training.  The path's own kernels -- depthwise conv + LayerNorm, the encoder's
GEMMs with fused epilogues, prefix-LM / decode attention, LayerNorm, RoPE, relu^2, the fused optimizer, and the decoder
products routed to the large-tile GEMM -- are hand-written HIP for gfx950 in libmmhip behind the C ABI of
include/mmhip.h.  By default the remaining plain decoder GEMMs run on hipBLASLt through torch.mm where it measured faster
(set_gemm_backend("own") routes every product to libmmhip), embedding gathers and the cross-entropy on ATen (DESIGN.md 1
lists exactly which).  There is no CPU or eager fallback for the libmmhip part."""
import os as _os

# hipBLASLt's GEMMs are stream-K kernels whose workgroups wait for each other's partial tiles; two of them launched from
# two HIP streams (backward: dX chain on the main stream, weight gradients on the side stream) were seen to deadlock at
# large row counts.  Data-parallel tile assignment has no inter-workgroup waits (and measured no slower here: 27.4 vs
# 27.6 ms per step at batch 32, +12 % at batch 256).  The library reads the variable on every call; a user setting wins.
_os.environ.setdefault("TENSILE_STREAMK_DATA_PARALLEL", "1")

from .functional import compute_dtype, get_compute_dtype, set_compute_dtype  # noqa: F401
from .functional import get_gemm_backend, set_gemm_backend  # noqa: F401

__version__ = "0.1.0"
