#!/usr/bin/env python
"""Build vit-lens_amd/open_clip/clip_bpe_merges.txt.xz from the CLIP BPE vocabulary table that the
reference vendors (open_clip/bpe_simple_vocab_16e6.txt.gz).  DATA ONLY: the ordered merge list
(48,894 pairs) that defines token ids; no code is copied.  Run in the build container."""
import gzip
import lzma
import os
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/vitlens/src/open_clip/bpe_simple_vocab_16e6.txt.gz"
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vit-lens_amd", "open_clip",
                   "clip_bpe_merges.txt.xz")
lines = gzip.open(src).read().decode("utf-8").split("\n")
merges = lines[1:49152 - 256 - 2 + 1]
with lzma.open(dst, "wt", encoding="utf-8", preset=9) as f:
    f.write("\n".join(merges))
print(len(merges), "merges ->", dst, os.path.getsize(dst), "bytes")
