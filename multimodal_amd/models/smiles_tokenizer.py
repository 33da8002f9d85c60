"""SMILES tokenizer / indexing, bit-identical to reference models/smiles_tokenizer.py (regex :10, vocab :13-21,
encode :69-75,:84-85, decode :77-82) without the HuggingFace BertTokenizer base class (host-side, no GPU work)."""
import collections
import os
import re
from typing import Dict, List

SMI_REGEX_PATTERN = r"""(\[[^\]]+]|Br?|Cl?|N|O|S|P|F|I|b|c|n|o|s|p|\(|\)|\.|=|#|-|\+|\\|\/|:|~|@|\?|>>?|\*|\$|\%[0-9]{2}|[0-9])"""


def load_vocab(vocab_file: str) -> "collections.OrderedDict[str, int]":
    vocab = collections.OrderedDict()
    with open(vocab_file, "r", encoding="utf-8") as reader:
        for index, token in enumerate(reader.readlines()):
            vocab[token.rstrip("\n")] = index
    return vocab


class BasicSmilesTokenizer:
    def __init__(self, regex_pattern: str = SMI_REGEX_PATTERN):
        self.regex_pattern = regex_pattern
        self.regex = re.compile(regex_pattern)

    def tokenize(self, text: str) -> List[str]:
        return list(self.regex.findall(text))


class SmilesTokenizer:
    unk_token, cls_token, sep_token, pad_token, mask_token = "[UNK]", "[CLS]", "[SEP]", "[PAD]", "[MASK]"

    def __init__(self, vocab_file: str = "", **kwargs):
        if not os.path.isfile(vocab_file):
            raise ValueError(f"Can't find a vocab file at path '{vocab_file}'.")
        self.vocab = load_vocab(vocab_file)
        self.ids_to_tokens: Dict[int, str] = collections.OrderedDict((i, t) for t, i in self.vocab.items())
        self.basic_tokenizer = BasicSmilesTokenizer()
        self.unk_token_id = self.vocab[self.unk_token]
        self.cls_token_id = self.vocab[self.cls_token]
        self.sep_token_id = self.vocab[self.sep_token]
        self.pad_token_id = self.vocab[self.pad_token]

    def __len__(self):
        return len(self.vocab)

    @property
    def vocab_size(self):
        return len(self.vocab)

    def tokenize(self, text: str, max_seq_length: int = 512) -> List[str]:
        return self.basic_tokenizer.tokenize(text)[: max_seq_length - 2]

    def convert_tokens_to_ids(self, tokens: List[str]) -> List[int]:
        return [self.vocab.get(t, self.unk_token_id) for t in tokens]

    def encode(self, text: str, max_seq_length: int = 512, add_special_tokens: bool = True, max_length=None,
               truncation: bool = False) -> List[int]:
        """The HuggingFace call the reference's dataset makes (train_autoregressive.py:332-337:
        encode(s, add_special_tokens=True, max_length=L, truncation=True)): _tokenize keeps max_seq_length-2 tokens
        (:69-72), truncation then cuts the token list so that ids + [CLS]/[SEP] fit max_length."""
        ids = self.convert_tokens_to_ids(self.tokenize(text, max_seq_length))
        n_special = 2 if add_special_tokens else 0
        if truncation and max_length is not None:
            ids = ids[: max(0, int(max_length) - n_special)]
        return [self.cls_token_id] + ids + [self.sep_token_id] if add_special_tokens else ids

    def convert_tokens_to_string(self, tokens: List[str]) -> str:
        return " ".join(tokens).replace(" ##", "").strip()

    def decode(self, ids: List[int]) -> str:
        return self.convert_tokens_to_string([self.ids_to_tokens.get(int(i), self.unk_token) for i in ids])
