"""CLIP byte-pair-encoding tokenizer (host side, integer work: bit-exact by construction).

Restates the published algorithm of OpenAI CLIP's ``SimpleTokenizer`` / ``clip.tokenize`` (third-party package
``clip``, ``pyproject.toml:21`` of the reference; call sites ``otter/policy/otter.py:73-75,357`` and
``otter/policy/otter_interface.py:164,285``):

* vocabulary = 256 byte symbols + the same 256 with ``</w>`` + one symbol per merge + ``<|startoftext|>`` +
  ``<|endoftext|>``; merges are lines 1..48894 of ``bpe_simple_vocab_16e6.txt.gz``;
* text is cleaned (html-unescaped twice, whitespace collapsed, stripped), lower-cased, split with the CLIP
  regular expression, each piece mapped to byte symbols and merged greedily by merge rank;
* ``tokenize`` wraps every text in start/end tokens, zero-pads to ``context_length`` and refuses (or truncates)
  longer inputs.

The merges file is not redistributed here and the reference does not vendor it either; pass its path or set
``OTTER_CLIP_BPE`` (plain text or ``.gz``).  ``ftfy.fix_text`` (the original's first clean-up step: mojibake repair, NFC
normalisation, curly quotes -> straight quotes, ...) is applied when ``ftfy`` is installed.  It leaves plain ASCII text
unchanged, so ASCII prompts get identical ids either way; for text with any non-ASCII character the ids may differ from
``clip.tokenize`` when ``ftfy`` is missing, and the tokenizer says so instead of diverging silently: a
:class:`UnrepairedTextWarning` once per process, or a ``RuntimeError`` with ``OTTER_TOKENIZER_STRICT=1``.
"""
from __future__ import annotations

import gzip
import html
import os
import warnings
from functools import lru_cache
from typing import Dict, Iterable, List, Optional, Sequence, Tuple, Union

import regex as re
import torch

SOT_TOKEN = 49406     # encoder["<|startoftext|>"] for the 49408-entry CLIP vocabulary
EOT_TOKEN = 49407     # encoder["<|endoftext|>"]
N_MERGES = 49152 - 256 - 2

_PATTERN = re.compile(
    r"""<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+""", re.IGNORECASE)


@lru_cache()
def bytes_to_unicode() -> Dict[int, str]:
    """Reversible byte -> printable unicode character table of GPT-2 / CLIP BPE."""
    keep = list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + list(range(ord("®"), ord("ÿ") + 1))
    table = {b: chr(b) for b in keep}
    extra = 0
    for b in range(256):
        if b not in table:
            table[b] = chr(256 + extra)
            extra += 1
    # the original builds the table in the order `keep` then the remaining bytes ascending
    ordered = {b: table[b] for b in keep}
    for b in range(256):
        if b not in ordered:
            ordered[b] = table[b]
    return ordered


class UnrepairedTextWarning(UserWarning):
    """A prompt with non-ASCII characters was tokenised without ``ftfy.fix_text``: its ids may differ from ``clip.tokenize``."""


_warned_unrepaired = False


def _clean(text: str) -> str:
    global _warned_unrepaired
    try:
        import ftfy  # type: ignore
        text = ftfy.fix_text(text)
    except ImportError:
        if not text.isascii():
            msg = ("prompt contains non-ASCII characters and the `ftfy` package is not installed: clip.tokenize repairs such text with "
                   "ftfy.fix_text before encoding it, so the token ids produced here may differ from the reference's "
                   f"(first such prompt: {text[:60]!r}); install ftfy for bit-exact ids")
            if os.environ.get("OTTER_TOKENIZER_STRICT", "0") == "1":
                raise RuntimeError(msg)
            if not _warned_unrepaired:
                _warned_unrepaired = True
                warnings.warn(msg, UnrepairedTextWarning, stacklevel=3)
    text = html.unescape(html.unescape(text)).strip()
    return re.sub(r"\s+", " ", text).strip()


def read_merges(path: str, n_merges: int = N_MERGES) -> List[Tuple[str, str]]:
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "rb") as f:
        lines = f.read().decode("utf-8").split("\n")
    lines = lines[1: n_merges + 1]           # line 0 is a version header
    return [tuple(l.split()) for l in lines]  # type: ignore[misc]


class SimpleTokenizer:
    def __init__(self, bpe_path: Optional[str] = None, merges: Optional[Sequence[Tuple[str, str]]] = None):
        if merges is None:
            bpe_path = bpe_path or os.environ.get("OTTER_CLIP_BPE")
            if not bpe_path or not os.path.exists(bpe_path):
                raise FileNotFoundError(
                    "CLIP BPE merges file not found: pass bpe_path or set OTTER_CLIP_BPE to bpe_simple_vocab_16e6.txt.gz "
                    "(or pass pre-tokenised int32 ids of shape (B, 77) instead of strings)")
            merges = read_merges(bpe_path)
        self.byte_encoder = bytes_to_unicode()
        self.byte_decoder = {v: k for k, v in self.byte_encoder.items()}
        vocab = list(self.byte_encoder.values())
        vocab = vocab + [v + "</w>" for v in vocab]
        vocab += ["".join(m) for m in merges]
        vocab += ["<|startoftext|>", "<|endoftext|>"]
        self.encoder = {tok: i for i, tok in enumerate(vocab)}
        self.decoder = {i: tok for tok, i in self.encoder.items()}
        self.bpe_ranks = {tuple(m): i for i, m in enumerate(merges)}
        self._cache = {"<|startoftext|>": "<|startoftext|>", "<|endoftext|>": "<|endoftext|>"}

    def bpe(self, token: str) -> str:
        if token in self._cache:
            return self._cache[token]
        word = tuple(token[:-1]) + (token[-1] + "</w>",)
        while len(word) > 1:
            pairs = set(zip(word[:-1], word[1:]))
            best = min(pairs, key=lambda p: self.bpe_ranks.get(p, float("inf")))
            if best not in self.bpe_ranks:
                break
            first, second = best
            merged: List[str] = []
            i = 0
            while i < len(word):
                if i < len(word) - 1 and word[i] == first and word[i + 1] == second:
                    merged.append(first + second)
                    i += 2
                else:
                    merged.append(word[i])
                    i += 1
            word = tuple(merged)
        out = " ".join(word)
        self._cache[token] = out
        return out

    def encode(self, text: str) -> List[int]:
        ids: List[int] = []
        for piece in re.findall(_PATTERN, _clean(text).lower()):
            piece = "".join(self.byte_encoder[b] for b in piece.encode("utf-8"))
            ids.extend(self.encoder[t] for t in self.bpe(piece).split(" "))
        return ids

    def decode(self, ids: Iterable[int]) -> str:
        text = "".join(self.decoder[int(i)] for i in ids)
        return bytearray(self.byte_decoder[c] for c in text).decode("utf-8", errors="replace").replace("</w>", " ")


_default: Optional[SimpleTokenizer] = None


def default_tokenizer() -> SimpleTokenizer:
    global _default
    if _default is None:
        _default = SimpleTokenizer()
    return _default


def tokenize(texts: Union[str, List[str]], context_length: int = 77, truncate: bool = False,
             tokenizer: Optional[SimpleTokenizer] = None) -> torch.Tensor:
    """``clip.tokenize``: (B, context_length) int32, ``[SOT] ids [EOT]`` then zeros."""
    if isinstance(texts, str):
        texts = [texts]
    tk = tokenizer or default_tokenizer()
    sot, eot = tk.encoder["<|startoftext|>"], tk.encoder["<|endoftext|>"]
    out = torch.zeros(len(texts), context_length, dtype=torch.int32)
    for i, text in enumerate(texts):
        ids = [sot] + tk.encode(text) + [eot]
        if len(ids) > context_length:
            if not truncate:
                raise RuntimeError(f"Input {text} is too long for context length {context_length}")
            ids = ids[:context_length]
            ids[-1] = eot
        out[i, : len(ids)] = torch.tensor(ids, dtype=torch.int32)
    return out
