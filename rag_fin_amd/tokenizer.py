"""BERT WordPiece tokenizer (uncased), host-side Python.

Stands in for the tokenizer SentenceTransformer('all-MiniLM-L6-v2') loads by name
(vector_rag_mcp/main.py:41); the vocabulary file must be supplied from a local
path -- nothing is fetched.  Algorithm: BERT "basic" tokenisation (clean, lower,
strip accents, split punctuation, space CJK) followed by greedy longest-match
WordPiece with the "##" continuation prefix, wrapped in [CLS] ... [SEP] and
truncated to max_seq_length (256 for all-MiniLM-L6-v2).
"""
from __future__ import annotations

import unicodedata
from typing import Iterable


def _is_whitespace(ch: str) -> bool:
    return ch in " \t\n\r" or unicodedata.category(ch) == "Zs"


def _is_control(ch: str) -> bool:
    if ch in "\t\n\r":
        return False
    return unicodedata.category(ch).startswith("C")


def _is_punctuation(ch: str) -> bool:
    cp = ord(ch)
    if 33 <= cp <= 47 or 58 <= cp <= 64 or 91 <= cp <= 96 or 123 <= cp <= 126:
        return True
    return unicodedata.category(ch).startswith("P")


def _is_cjk(cp: int) -> bool:
    return (0x4E00 <= cp <= 0x9FFF or 0x3400 <= cp <= 0x4DBF or 0x20000 <= cp <= 0x2A6DF
            or 0x2A700 <= cp <= 0x2B73F or 0x2B740 <= cp <= 0x2B81F or 0x2B820 <= cp <= 0x2CEAF
            or 0xF900 <= cp <= 0xFAFF or 0x2F800 <= cp <= 0x2FA1F)


class _LazyTable(dict):
    """str.translate table that classifies a code point the first time it is seen."""

    def __init__(self, fn):
        super().__init__()
        self._fn = fn

    def __missing__(self, cp):
        v = self._fn(cp)
        self[cp] = v
        return v


def _clean_map(cp: int):
    """_clean() for one code point (ASCII is left to the C++ side)."""
    if cp < 128:
        return cp
    ch = chr(cp)
    if cp == 0xFFFD or _is_control(ch):
        return None
    if _is_cjk(cp):
        return " " + ch + " "
    if _is_whitespace(ch) or ch.isspace():   # Zl / Zp are not "whitespace" to _clean, but str.split() splits on them
        return " "
    return cp


_SIMPLE = None


def _simple_tables():
    """(regex that finds a character needing the Python pre-pass, sorted list of 'simple'
    non-ASCII punctuation code points).  A non-ASCII character is SIMPLE when the basic
    tokenizer leaves it as it is: not a control / format / unassigned character, space or mark,
    not a CJK ideograph or conjoining Hangul jamo, caseless, without a decomposition and with
    combining class 0 -- so clean, NFC, lower, NFD and the Mn strip are all identities on text
    made of ASCII and simple characters, and only punctuation splitting is left (done natively)."""
    global _SIMPLE
    if _SIMPLE is None:
        import re
        ranges, punct = [], []
        start = prev = None
        for cp in range(128, 0x110000):
            ch = chr(cp)
            cat = unicodedata.category(ch)
            ok = (cp != 0xFFFD and cat[0] not in "CZM" and not _is_cjk(cp) and not unicodedata.combining(ch)
                  and not unicodedata.decomposition(ch) and ch.lower() == ch
                  and not (0x1100 <= cp <= 0x11FF or 0xA960 <= cp <= 0xA97F or 0xD7B0 <= cp <= 0xD7FF))
            if ok:
                if cat[0] == "P":
                    punct.append(cp)
                if prev is not None and cp == prev + 1:
                    prev = cp
                else:
                    if start is not None:
                        ranges.append((start, prev))
                    start = prev = cp
        if start is not None:
            ranges.append((start, prev))
        cls = "".join(re.escape(chr(a)) if a == b else re.escape(chr(a)) + "-" + re.escape(chr(b))
                      for a, b in ranges)
        _SIMPLE = (re.compile("[^\\x00-\\x7f" + cls + "]"), punct)
    return _SIMPLE


class WordPieceTokenizer:
    def __init__(self, vocab: dict[str, int] | Iterable[str], do_lower_case: bool = True,
                 unk_token: str = "[UNK]", cls_token: str = "[CLS]", sep_token: str = "[SEP]",
                 pad_token: str = "[PAD]", max_input_chars_per_word: int = 100):
        if not isinstance(vocab, dict):
            vocab = {tok: i for i, tok in enumerate(vocab)}
        self.vocab = vocab
        self.do_lower_case = do_lower_case
        self.unk, self.cls, self.sep, self.pad = unk_token, cls_token, sep_token, pad_token
        for t in (unk_token, cls_token, sep_token, pad_token):
            if t not in vocab:
                raise ValueError(f"vocabulary lacks the special token {t}")
        self.unk_id, self.cls_id = vocab[unk_token], vocab[cls_token]
        self.sep_id, self.pad_id = vocab[sep_token], vocab[pad_token]
        self.special = {unk_token, cls_token, sep_token, pad_token, "[MASK]"}
        self.max_chars = max_input_chars_per_word

    @classmethod
    def from_vocab_file(cls, path: str, **kw) -> "WordPieceTokenizer":
        with open(path, encoding="utf-8") as f:
            toks = [line.rstrip("\n") for line in f]
        return cls({t: i for i, t in enumerate(toks)}, **kw)

    # -- basic tokenisation ---------------------------------------------------------
    def _clean(self, text: str) -> str:
        out = []
        for ch in text:
            cp = ord(ch)
            if cp == 0 or cp == 0xFFFD or _is_control(ch):
                continue
            if _is_cjk(cp):
                out.append(" " + ch + " ")
            elif _is_whitespace(ch):
                out.append(" ")
            else:
                out.append(ch)
        return "".join(out)

    def _split_punct(self, word: str) -> list[str]:
        pieces, cur = [], []
        for ch in word:
            if _is_punctuation(ch):
                if cur:
                    pieces.append("".join(cur))
                    cur = []
                pieces.append(ch)
            else:
                cur.append(ch)
        if cur:
            pieces.append("".join(cur))
        return pieces

    def basic_tokens(self, text: str) -> list[str]:
        text = unicodedata.normalize("NFC", self._clean(text))
        out = []
        for word in text.strip().split():
            if word in self.special:
                out.append(word)
                continue
            if self.do_lower_case:
                word = word.lower()
                word = "".join(c for c in unicodedata.normalize("NFD", word)
                               if unicodedata.category(c) != "Mn")
            out.extend(self._split_punct(word))
        return out

    # -- wordpiece ----------------------------------------------------------------------
    def wordpiece(self, word: str) -> list[int]:
        if len(word) > self.max_chars:
            return [self.unk_id]
        ids, start = [], 0
        while start < len(word):
            end = len(word)
            cur = None
            while start < end:
                sub = word[start:end]
                if start > 0:
                    sub = "##" + sub
                if sub in self.vocab:
                    cur = self.vocab[sub]
                    break
                end -= 1
            if cur is None:
                return [self.unk_id]
            ids.append(cur)
            start = end
        return ids

    def encode(self, text: str, max_length: int = 256) -> list[int]:
        ids = []
        for w in self.basic_tokens(text):
            ids.extend(self.wordpiece(w) if w not in self.special else [self.vocab[w]])
        ids = ids[:max(0, max_length - 2)]
        return [self.cls_id] + ids + [self.sep_id]

    # -- native path (libragfin_hip.so: csrc/tokenizer.cpp, multi-threaded) ---------------
    # The C++ tokenizer applies the ASCII rules; text with non-ASCII characters is first
    # brought to a form in which those rules are all that is left to apply, using only
    # C-speed string primitives (translate with lazily built tables, normalize, lower).
    def _native(self):
        if getattr(self, "_nat", None) is None:
            from ctypes import byref, c_void_p
            from . import _lib
            lib = _lib.load_library()
            toks = [None] * (max(self.vocab.values()) + 1)
            for t, i in self.vocab.items():
                toks[i] = t
            blob = "\n".join("" if t is None else t for t in toks).encode("utf-8")
            h = c_void_p()
            _lib.check(lib.rf_tokenizer_create(byref(h), blob, len(blob), 1 if self.do_lower_case else 0,
                                               self.max_chars))
            complex_re, punct = _simple_tables()
            import numpy as np
            arr = np.asarray(punct, dtype=np.int32)
            _lib.check(lib.rf_tokenizer_set_punctuation(h, c_void_p(arr.ctypes.data), len(punct)))
            self._complex_re = complex_re
            self._nat = (lib, h)
            self._clean_tab = _LazyTable(_clean_map)
            self._mn_tab = _LazyTable(lambda cp: None if unicodedata.category(chr(cp)) == "Mn" else cp)
            self._mn_punct_tab = _LazyTable(lambda cp: None if unicodedata.category(chr(cp)) == "Mn" else
                                            (" " + chr(cp) + " " if cp >= 128 and
                                             unicodedata.category(chr(cp)).startswith("P") else cp))
            self._punct_tab = _LazyTable(lambda cp: " " + chr(cp) + " " if cp >= 128 and
                                         unicodedata.category(chr(cp)).startswith("P") else cp)
        return self._nat

    def _prenormalise(self, text: str):
        """Non-ASCII text -> a string on which the ASCII rules of the C++ tokenizer give the
        tokens of basic_tokens(); None when the text must take the Python path (it contains a
        literal special token, which whole-string lower-casing would destroy)."""
        s = text.translate(self._clean_tab)
        if not unicodedata.is_normalized("NFC", s):
            s = unicodedata.normalize("NFC", s)
        if "[" in s and any(sp in s for sp in self.special):
            return None
        if self.do_lower_case:
            s = s.lower()
            if not unicodedata.is_normalized("NFD", s):
                s = unicodedata.normalize("NFD", s)
            return s.translate(self._mn_punct_tab)   # strip Mn and pad punctuation in one pass
        return s.translate(self._punct_tab)

    def _only_simple_non_ascii(self, joined: str) -> bool:
        """True if every non-ASCII character of the batch is one the native rules handle as they are (no
        pre-normalisation): decided on the batch's DISTINCT non-ASCII code units -- a handful -- instead of running
        the character-class regex over megabytes of text (30 ms per 4 M characters; this: a UTF-16 view, one
        comparison and one `unique` in numpy)."""
        import numpy as np
        try:
            u16 = np.frombuffer(joined.encode("utf-16-le"), dtype=np.uint16)
        except UnicodeEncodeError:                   # lone surrogates
            return False
        hi = np.unique(u16[u16 >= 128])
        if hi.size == 0:
            return True
        if hi.size > 4096 or bool(((hi >= 0xD800) & (hi <= 0xDFFF)).any()):   # beyond the BMP: let the per-text path look
            return False
        return self._complex_re.search("".join(map(chr, hi.tolist()))) is None

    def batch_native(self, texts: list[str], max_length: int = 256, n_threads: int = 0):
        """Same result as batch(), through rf_tokenize_batch.  -> (ids int32 [B, T], lens int32 [B])."""
        import os
        import numpy as np
        if n_threads <= 0:
            n_threads = int(os.environ.get("RAGFIN_TOKENIZER_THREADS", "0") or 0)
        if n_threads <= 0 and len(texts) < 64:
            n_threads = 1        # a query-sized batch: no pool
        if n_threads <= 0:   # the CPUs the process may BURN (CFS quota), not the ones it may run on: hostcpu.py
            from .hostcpu import cpu_budget
            n_threads = min(cpu_budget(), 64)
        from ctypes import c_void_p
        from . import _lib
        import time
        lib, h = self._native()
        n = len(texts)
        if n == 0:
            return np.full((0, 1), self.pad_id, dtype=np.int32), np.zeros((0,), dtype=np.int32)
        t_start = time.perf_counter()
        python_rows = {}
        offsets = np.zeros(n + 1, dtype=np.int64)
        joined = "".join(texts)
        blob = None
        if joined.isascii() or self._only_simple_non_ascii(joined):
            # The usual ingest batch -- ASCII, or with "simple" non-ASCII characters only (a currency sign) that the
            # native rules handle: ONE join, ONE test and ONE UTF-8 encode of the whole batch in C, the per-text
            # character counts turned into byte offsets by rf_utf8_offsets -- instead of a Python loop that encoded and
            # measured every text (1.6 us per text: most of the host's share of the text ingest).
            try:
                blob = joined.encode("utf-8")
            except UnicodeEncodeError:               # a lone surrogate somewhere: the per-text path below sorts it out
                blob = None
        if blob is not None:
            np.cumsum(np.fromiter(map(len, texts), dtype=np.int64, count=n), out=offsets[1:])
            if len(blob) != int(offsets[n]):         # non-ASCII present: characters -> bytes
                chars = offsets.copy()
                _lib.check(lib.rf_utf8_offsets(blob, len(blob), c_void_p(chars.ctypes.data), n, c_void_p(offsets.ctypes.data)))
        else:
            enc = []
            for i, t in enumerate(texts):
                if not t.isascii() and self._complex_re.search(t) is not None:
                    t = self._prenormalise(t)
                    if t is None:
                        python_rows[i] = self.encode(texts[i], max_length)
                        t = ""
                try:
                    enc.append(t.encode("utf-8"))
                except UnicodeEncodeError:           # lone surrogates: Python path
                    python_rows[i] = self.encode(texts[i], max_length)
                    enc.append(b"")
            np.cumsum([len(b) for b in enc], out=offsets[1:])
            blob = b"".join(enc)
        del joined
        ids = np.empty((n, max_length), dtype=np.int32)
        lens = np.empty((n,), dtype=np.int32)
        t_native = time.perf_counter()
        _lib.check(lib.rf_tokenize_batch(h, blob, c_void_p(offsets.ctypes.data), n, max_length,
                                         c_void_p(ids.ctypes.data), c_void_p(lens.ctypes.data), n_threads))
        t_done = time.perf_counter()
        # host seconds of the last call: Python pre-pass (ASCII test, UTF-8 bytes, join) / the native call
        self.last_timing = {"prepass_s": t_native - t_start, "native_s": t_done - t_native}
        for i, row in python_rows.items():
            ids[i, :] = self.pad_id
            ids[i, :len(row)] = row
            lens[i] = len(row)
        T = max(int(lens.max()), 1)
        return np.ascontiguousarray(ids[:, :T]), lens

    def batch(self, texts: list[str], max_length: int = 256):
        """-> (ids int32 [B, T] padded with [PAD], lens int32 [B]), T = longest row."""
        import numpy as np
        rows = [self.encode(t, max_length) for t in texts]
        T = max((len(r) for r in rows), default=1)
        ids = np.full((len(rows), T), self.pad_id, dtype=np.int32)
        for i, r in enumerate(rows):
            ids[i, :len(r)] = r
        return ids, np.asarray([len(r) for r in rows], dtype=np.int32)
