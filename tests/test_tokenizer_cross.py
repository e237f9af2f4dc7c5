"""The BPE tokenizer against an independent implementation: ``transformers.CLIPTokenizer`` (the Rust ``tokenizers``
BPE model with CLIP's normaliser and split pattern), both built from the SAME vocabulary and merges table.

The real CLIP vocabulary file is not available offline and the reference does not vendor it, so the table is learned here,
deterministically, from English text that ships with Python (a few thousand merges: words really do merge several levels
deep, unlike a made-up table).  What this pins: the byte-to-symbol table, the split pattern, the greedy merge order, the
``</w>`` handling, start / end tokens, padding and overflow behaviour.  What it cannot pin: the contents of the real table
(a known-answer test runs when ``OTTER_CLIP_BPE`` points at the file).  Known, deliberate differences between the two
implementations are tested as differences at the end instead of being hidden.
"""
import collections
import html
import random

import pytest
import torch

from otter_amd import tokenizer as tk


def _corpus():
    import argparse, collections as c, json, os, random as r, re, string, textwrap, this  # noqa: E401,F401  (docstrings = English text)
    import codecs
    zen = codecs.decode(this.s, "rot13")
    docs = [zen]
    for mod in (argparse, c, json, os, r, re, string, textwrap):
        docs.append(mod.__doc__ or "")
        for name in sorted(dir(mod)):
            d = getattr(getattr(mod, name), "__doc__", None)
            if isinstance(d, str):
                docs.append(d)
    docs.append("pick up the blue cube and place it on the red plate. open the top drawer, then close it. push the green block to the left "
                "of the yellow bowl. it's the robot's turn; don't move! we'll see what they've done: i'm sure you'd agree they're fast. " * 20)
    return "\n".join(docs)


def _learn_merges(text, n_merges):
    """Plain BPE training (most frequent adjacent pair first, ties by first occurrence) over CLIP's word pieces."""
    b2u = tk.bytes_to_unicode()
    words = collections.Counter()
    for piece in tk.re.findall(tk._PATTERN, tk._clean(text).lower()):
        sym = [b2u[b] for b in piece.encode("utf-8")]
        sym[-1] += "</w>"
        words[tuple(sym)] += 1
    merges = []
    words = dict(words)
    for _ in range(n_merges):
        pairs = collections.Counter()
        for w, f in words.items():
            for a, b in zip(w[:-1], w[1:]):
                pairs[(a, b)] += f
        if not pairs:
            break
        best = max(pairs.items(), key=lambda kv: (kv[1], -len(kv[0][0] + kv[0][1]), kv[0]))[0]
        merges.append(best)
        new = {}
        for w, f in words.items():
            out, i = [], 0
            while i < len(w):
                if i + 1 < len(w) and (w[i], w[i + 1]) == best:
                    out.append(w[i] + w[i + 1])
                    i += 2
                else:
                    out.append(w[i])
                    i += 1
            new[tuple(out)] = new.get(tuple(out), 0) + f
        words = new
    return merges


@pytest.fixture(scope="module")
def pair():
    transformers = pytest.importorskip("transformers")
    merges = _learn_merges(_corpus(), 3000)
    assert len(merges) > 2000 and len(set(merges)) == len(merges)
    ours = tk.SimpleTokenizer(merges=merges)
    theirs = transformers.CLIPTokenizer(vocab=dict(ours.encoder), merges=[tuple(m) for m in merges])
    return ours, theirs


def _prompts(n, seed=0):
    rng = random.Random(seed)
    verbs = ["pick up", "Place", "PUSH", "open", "close", "slide", "don't touch", "it's", "we'll move", "they've stacked", "I'm holding", "you'd take"]
    things = ["the blue cube", "the RED ball", "drawer #2", "3 blocks", "block 42", "the robot's arm", "a cup (small)", "the left-most bowl", "item_7",
              "50% of it", "the café table", "naïve über-gripper", "the box...", "that one?!", "A/B side", "x=1.5,y=-2", "the \"top\" shelf", "thing's"]
    glue = [" ", "  ", " and ", ", then ", "; ", " -- ", "\t", " \n ", "...", "!!! ", " & ", " @ "]
    out = []
    for _ in range(n):
        k = rng.randint(1, 5)
        s = ""
        for j in range(k):
            s += rng.choice(verbs) + rng.choice([" ", "  "]) + rng.choice(things) + (rng.choice(glue) if j + 1 < k else rng.choice(["", ".", " ", "!"]))
        out.append(s)
    out += ["", " ", "a", "'s", "''''", "12345 67", "!!!???", "tab\tseparated\twords", "trailing space ", "  leading", "MiXeD CaSe 'S 'T 'RE",
            "emoji \U0001F916 robot", "日本語 のテキスト", "don't can't won't it'll they'd I've we're", "<|startoftext|> literal <|endoftext|>"]
    return out


def test_same_ids_as_the_independent_implementation(pair):
    ours, theirs = pair
    prompts = _prompts(400)
    n_multi = 0
    for text in prompts:
        want = theirs(text, add_special_tokens=False)["input_ids"]
        got = ours.encode(text)
        assert got == want, (text, got, want)
        n_multi += any(len(ours.decoder[i].replace("</w>", "")) > 3 for i in got)
    assert n_multi > 300                       # the learned table really merges: most prompts contain multi-character symbols
    # whole tokenize(): start / end tokens, zero padding, int32, shape
    batch = [p for p in prompts if len(ours.encode(p)) <= 75][:64]
    ids = tk.tokenize(batch, tokenizer=ours)
    assert ids.shape == (len(batch), 77) and ids.dtype == torch.int32
    for row, text in zip(ids, batch):
        want = theirs(text)["input_ids"]                              # [SOT] ids [EOT]
        assert want[0] == tk.SOT_TOKEN or want[0] == ours.encoder["<|startoftext|>"]
        assert row[: len(want)].tolist() == want and int(row[len(want):].abs().sum()) == 0
    # round trip
    assert ours.decode(ours.encode("pick up the blue cube")).strip() == "pick up the blue cube"


def test_overflow_raises_or_truncates_like_the_independent_implementation(pair):
    ours, theirs = pair
    long_text = " ".join(["pick up the blue cube and place it"] * 30)
    assert len(ours.encode(long_text)) > 75
    with pytest.raises(RuntimeError):
        tk.tokenize([long_text], tokenizer=ours)
    row = tk.tokenize([long_text], tokenizer=ours, truncate=True)[0].tolist()
    want = theirs(long_text, truncation=True, max_length=77)["input_ids"]
    assert row == want and row[-1] == ours.encoder["<|endoftext|>"] and row[0] == ours.encoder["<|startoftext|>"]


def test_documented_differences_between_the_two_cleaners(pair):
    """OpenAI's cleaner (followed here) html-unescapes twice and strips; the other implementation does neither but its
    pre-tokeniser drops whitespace anyway, so only the entity handling shows."""
    ours, theirs = pair
    text = "salt &amp; pepper &amp;lt;3"
    assert ours.encode(text) != theirs(text, add_special_tokens=False)["input_ids"]
    cleaned = html.unescape(html.unescape(text)).strip()
    assert cleaned == "salt & pepper <3"
    assert ours.encode(text) == theirs(cleaned, add_special_tokens=False)["input_ids"]
