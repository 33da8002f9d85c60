"""MI355X-native hot path of tomoqt/multimodal (spectra -> SMILES): ConvNeXt-1D encoder, prompt-decoder attention,
KV-cached greedy decode, data-parallel training.  All arithmetic of the hot path runs in libmmhip (hand-written HIP for
gfx950) behind the C ABI in include/mmhip.h; there is no CPU or eager fallback."""
import os as _os

# hipBLASLt's GEMMs are stream-K kernels whose workgroups wait for each other's partial tiles; two of them launched from
# two HIP streams (backward: dX chain on the main stream, weight gradients on the side stream) were seen to deadlock at
# large row counts.  Data-parallel tile assignment has no inter-workgroup waits (and measured no slower here: 27.4 vs
# 27.6 ms per step at batch 32, +12 % at batch 256).  The library reads the variable on every call; a user setting wins.
_os.environ.setdefault("TENSILE_STREAMK_DATA_PARALLEL", "1")

from .functional import compute_dtype, get_compute_dtype, set_compute_dtype  # noqa: F401

__version__ = "0.1.0"
