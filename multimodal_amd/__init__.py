"""MI355X-native hot path of tomoqt/multimodal (spectra -> SMILES): ConvNeXt-1D encoder, prompt-decoder attention,
KV-cached greedy decode, data-parallel training.  All arithmetic of the hot path runs in libmmhip (hand-written HIP for
gfx950) behind the C ABI in include/mmhip.h; there is no CPU or eager fallback."""
from .functional import compute_dtype, get_compute_dtype, set_compute_dtype  # noqa: F401

__version__ = "0.1.0"
