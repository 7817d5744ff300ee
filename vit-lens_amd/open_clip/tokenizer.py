"""CLIP byte-level BPE tokenizer (host side, integer-exact).

Behavioural twin of the reference's `open_clip.tokenize` (open_clip/tokenizer.py:177-208):
lower-case + whitespace clean + regex split + byte-BPE, `[SOT] ids [EOT]`, zero padded to
`context_length`, truncated with EOT forced into the last slot.  The merge table is the CLIP vocabulary
(data asset clip_bpe_merges.txt.xz, see tools/make_bpe_asset.py); the algorithm below is written from
the published BPE definition: repeatedly fuse the adjacent symbol pair of lowest merge rank.
Pinned bit-exactly by tests/golden/tokenizer_kat.json (52 strings, produced by the reference).
"""
import html
import lzma
import os
from typing import List, Union

import regex
import torch

try:  # the reference cleans text with ftfy; without it ASCII input is unaffected
    import ftfy

    def _fix(s):
        return ftfy.fix_text(s)
except ImportError:  # pragma: no cover
    def _fix(s):
        return s

_ASSET = os.path.join(os.path.dirname(os.path.abspath(__file__)), "clip_bpe_merges.txt.xz")
_SPECIALS = ("<start_of_text>", "<end_of_text>")
_SPLIT = regex.compile(
    "|".join(regex.escape(s) for s in _SPECIALS)
    + r"""|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+""", regex.IGNORECASE)


def _byte_alphabet():
    """byte value -> printable stand-in character (GPT-2 convention): printable latin-1 bytes map to
    themselves, every other byte to U+0100 + running index."""
    keep = [b for b in range(256) if (33 <= b <= 126) or (161 <= b <= 172) or (174 <= b <= 255)]
    table, extra = {}, 0
    for b in keep:
        table[b] = chr(b)
    for b in range(256):
        if b not in table:
            table[b] = chr(256 + extra)
            extra += 1
    order = keep + [b for b in range(256) if b not in set(keep)]
    return table, [table[b] for b in order]


class ClipBPE:
    def __init__(self, merges_path: str = _ASSET):
        with lzma.open(merges_path, "rt", encoding="utf-8") as f:
            pairs = [tuple(line.split()) for line in f.read().split("\n") if line]
        self.byte_sym, base = _byte_alphabet()
        symbols = base + [s + "</w>" for s in base] + [a + b for a, b in pairs] + list(_SPECIALS)
        self.token_id = {s: i for i, s in enumerate(symbols)}
        self.id_token = {i: s for s, i in self.token_id.items()}
        self.rank = {p: i for i, p in enumerate(pairs)}
        self.sot, self.eot = self.token_id[_SPECIALS[0]], self.token_id[_SPECIALS[1]]
        self.vocab_size = len(symbols)
        self._memo = {}

    def _merge_word(self, word: str) -> List[str]:
        hit = self._memo.get(word)
        if hit is not None:
            return hit
        parts = list(word[:-1]) + [word[-1] + "</w>"]
        inf = len(self.rank)
        while len(parts) > 1:
            best, best_rank = None, inf
            for a, b in zip(parts, parts[1:]):
                r = self.rank.get((a, b), inf)
                if r < best_rank:
                    best, best_rank = (a, b), r
            if best is None:
                break
            fused, i = [], 0
            while i < len(parts):
                if i + 1 < len(parts) and parts[i] == best[0] and parts[i + 1] == best[1]:
                    fused.append(parts[i] + parts[i + 1]); i += 2
                else:
                    fused.append(parts[i]); i += 1
            parts = fused
        self._memo[word] = parts
        return parts

    def encode(self, text: str) -> List[int]:
        text = html.unescape(html.unescape(_fix(text))).strip()
        text = regex.sub(r"\s+", " ", text).strip().lower()
        ids = []
        for piece in _SPLIT.findall(text):
            if piece in _SPECIALS:
                ids.append(self.token_id[piece]); continue
            word = "".join(self.byte_sym[b] for b in piece.encode("utf-8"))
            ids.extend(self.token_id[s] for s in self._merge_word(word))
        return ids

    def decode(self, ids) -> str:
        inv = {c: b for b, c in self.byte_sym.items()}
        s = "".join(self.id_token[int(i)] for i in ids)
        return bytearray(inv[c] for c in s.replace("</w>", " ") if c in inv).decode("utf-8", errors="replace")


_TOK = None


def _tokenizer() -> ClipBPE:
    global _TOK
    if _TOK is None:
        _TOK = ClipBPE()
    return _TOK


SimpleTokenizer = ClipBPE


def tokenize(texts: Union[str, List[str]], context_length: int = 77) -> torch.LongTensor:
    """-> int64 [len(texts), context_length]; same contract as open_clip.tokenize."""
    if isinstance(texts, str):
        texts = [texts]
    t = _tokenizer()
    out = torch.zeros(len(texts), context_length, dtype=torch.long)
    for i, s in enumerate(texts):
        ids = [t.sot] + t.encode(s) + [t.eot]
        if len(ids) > context_length:
            ids = ids[:context_length]
            ids[-1] = t.eot
        out[i, :len(ids)] = torch.tensor(ids, dtype=torch.long)
    return out


def decode(output_ids: torch.Tensor) -> str:
    return _tokenizer().decode(output_ids.cpu().tolist())
