"""Synthetic finance-chunk texts for BASELINE.json configs[3].
TEST / BENCH INFRASTRUCTURE ONLY (same rules as the rest of oracle/).

The reference's corpus is 16 chunks (FinRag_knowledge_graph/chunks.json, committed as
tests/golden/chunks_golden.json); config 4 asks for 10 000.  They are made by re-templating
those 16 texts with perturbed figures (every digit redrawn from a seeded generator), which keeps
the length distribution (156-623 characters, ~40-250 WordPiece tokens) and the vocabulary of
the real chunks.  all-MiniLM-L6-v2's vocab.txt does not exist offline, so the vocabulary is
built from the words of the chunk texts and padded with [unusedN] up to the model's 30 522."""
from __future__ import annotations

import json
import os
import re

import numpy as np

_GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                     "chunks_golden.json")


def base_texts() -> list[str]:
    with open(_GOLD) as f:
        return [c["text"] for c in json.load(f)]


def retemplated_texts(n: int, seed: int) -> list[str]:
    base = base_texts()
    rng = np.random.default_rng(seed)
    return [re.sub(r"\d", lambda m: str(int(rng.integers(0, 10))), base[i % len(base)]) for i in range(n)]


def vocab_for(texts=None, size: int = 30522) -> list[str]:
    words = set()
    for t in (texts if texts is not None else base_texts()):
        words.update(re.findall(r"[a-z]+|[0-9]|[^\sa-z0-9]", t.lower()))
    vocab = (["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + sorted(words) +
             ["##" + w for w in sorted(words) if w.isalpha()] + ["##%d" % i for i in range(10)])
    vocab = list(dict.fromkeys(vocab))
    vocab += ["[unused%d]" % i for i in range(max(0, size - len(vocab)))]
    return vocab[:size]
