"""CPU-side checks of the product's host logic: the C-ABI library loads and exports every symbol include/mmhip.h
declares, the module tree has the reference's state_dict contract, the tokenizer is bit-identical, and the hot path
refuses to run without a GPU (no fallback)."""
import os
import re

import pytest
import torch

from conftest import GOLDEN, ROOT, cfg_of
from oracle.shapes import state_shapes


def test_library_exports_every_declared_symbol():
    import ctypes
    from multimodal_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "mmhip.h")).read()
    declared = set(re.findall(r"\b(mm_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"mm_gemm_args"}
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/mmhip.h but not exported"
    assert declared == set(_lib.exported_symbols())
    assert _lib.lib().mm_version() >= 100


def test_state_dict_contract_matches_reference_layout(golden_meta):
    from multimodal_amd.models import MultiModalToSMILESModel
    for name in ("tiny_sm", "d128_sdpa"):
        case = golden_meta["cases"][name]
        m = MultiModalToSMILESModel(591, 1000, case["max_seq"], case["max_nmr"], case["max_mem"], embed_dim=case["E"],
                                    num_heads=case["H"], num_layers=case["L"], use_stablemax=case["sm"])
        got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
        assert got == state_shapes(cfg_of(case))


def test_no_cpu_fallback():
    from multimodal_amd.models import MultiModalToSMILESModel
    m = MultiModalToSMILESModel(591, 1000, 95, 190, 32, embed_dim=128, num_heads=4, num_layers=1, use_stablemax=True)
    with pytest.raises(RuntimeError, match="GPU"):
        m(torch.zeros(1, 4, dtype=torch.long), torch.zeros(1, 1000), torch.zeros(1, 3, dtype=torch.long))


def test_product_does_not_import_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "multimodal_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f


def test_tokenizer_bit_identical(golden_meta):
    from multimodal_amd.models import SmilesTokenizer
    tok = SmilesTokenizer(os.path.join(GOLDEN, "vocab.txt"))
    tk = golden_meta["tokenizer"]
    assert len(tok) == 591
    assert (tok.pad_token_id, tok.unk_token_id, tok.cls_token_id, tok.sep_token_id) == (0, 11, 12, 13)
    for s, ids in tk["encodings"].items():
        assert tok.encode(s) == ids
    assert tok.decode([15, 16, 17]) == "c C ("
