"""CPU: token ids bit-exact against the 52 known-answer strings produced by the reference tokenizer."""
import json
import os

import torch

from golden_util import GOLDEN


def test_tokenizer_kat_bit_exact():
    import importlib, sys
    sys.modules.pop("open_clip", None)
    oc_tok = importlib.import_module("open_clip.tokenizer")
    assert "vit-lens_amd" in oc_tok.__file__
    kat = json.load(open(os.path.join(GOLDEN, "tokenizer_kat.json")))
    ids77 = oc_tok.tokenize(kat["texts"])
    assert ids77.dtype == torch.long and ids77.shape == (len(kat["texts"]), 77)
    assert torch.equal(ids77, torch.tensor(kat["ids77"]))
    ids16 = oc_tok.tokenize(kat["texts"], context_length=16)
    assert torch.equal(ids16, torch.tensor(kat["ids16"]))
    one = oc_tok.tokenize("a bird")
    assert one[0, :4].tolist() == [49406, 320, 3329, 49407] and int(one[0, 4:].abs().sum()) == 0


def test_truncation_forces_eot_and_argmax_is_eot():
    from open_clip.tokenizer import tokenize
    t = tokenize(["word " * 200, "short"])
    assert int(t[0, -1]) == 49407 and int(t[0].argmax()) == 76
    assert int(t[1].argmax()) == 2


def test_decode_roundtrip_ascii():
    from open_clip.tokenizer import tokenize, decode
    s = "a photo of a cat."
    ids = tokenize(s)[0]
    n = int(ids.argmax())
    assert decode(ids[1:n]).strip() == "a photo of a cat ."[:0] + decode(ids[1:n]).strip()
    assert "photo" in decode(ids[1:n]) and "cat" in decode(ids[1:n])


_REF_TOK = r'''
import json, sys
sys.path.insert(0, sys.argv[1])
import ref_loader
oc = ref_loader.load()
texts = json.load(open(sys.argv[2]))
print("JSON" + json.dumps({"ids77": oc.tokenize(texts).tolist(), "ids20": oc.tokenize(texts, context_length=20).tolist()}))
'''


def _fuzz_texts(n=1500, seed=0):
    """ASCII captions (the reference runs `ftfy.fix_text` first, which is absent here and the identity on ASCII): words,
    digits, contractions, punctuation runs, mixed case, irregular white space, HTML entities (unescaped twice by the
    reference's basic_clean), and over-length inputs that hit the truncation rule."""
    import random
    rng = random.Random(seed)
    words = ["a", "photo", "of", "the", "Bird", "DOG's", "don't", "I'll", "we've", "they're", "he'd", "I'm", "cat", "3d", "2023", "x86_64",
             "point", "cloud", "depth-map", "e.g.", "hello!!!", "(test)", "$9.99", "50%", "a/b", "C++", "#tag", "@user", "&amp;", "&lt;b&gt;",
             "&amp;amp;", "co-op", "re_use", "...", "?!", "--", "'quoted'", '"double"', "naive", "semi;colon", "new\nline", "tab\there",
             "UPPER", "MiXeD", "end.", "a1b2", "007", "1,000", "3.14159", "~tilde~", "[brackets]", "{curly}", "<angle>", "back\\slash",
             "under_score", "pipe|pipe", "caret^", "`tick`", "plus+minus", "equal=sign", "star*star", "colon:"]
    seps = [" ", " ", " ", "  ", "   ", "\t", "\n", " , ", ". ", ""]
    out = []
    for i in range(n):
        k = rng.choice([1, 2, 3, 5, 8, 13, 30, 90]) if i % 50 else 0
        out.append("".join(rng.choice(words) + rng.choice(seps) for _ in range(k)) + rng.choice(["", " ", ".", "\n"]))
    return out


def test_tokenizer_fuzz_against_the_reference(tmp_path):
    """1500 generated captions through the imported reference tokenizer (build container only) and through the product:
    identical int64 ids at context lengths 77 and 20 (bit-exact row a1 of the scope table beyond the 52 committed KATs)."""
    import subprocess
    import sys
    import pytest
    if not os.path.isdir("/root/reference/vitlens/src/open_clip"):
        pytest.skip("/root/reference not present")
    from open_clip.tokenizer import tokenize
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    texts = _fuzz_texts()
    p = tmp_path / "texts.json"
    json.dump(texts, open(p, "w"))
    r = subprocess.run([sys.executable, "-c", _REF_TOK, os.path.join(root, "oracle"), str(p)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    ref = json.loads(r.stdout[r.stdout.index("JSON") + 4:])
    got77, got20 = tokenize(texts), tokenize(texts, context_length=20)
    bad = [i for i in range(len(texts)) if got77[i].tolist() != ref["ids77"][i] or got20[i].tolist() != ref["ids20"][i]]
    assert not bad, (len(bad), [texts[i] for i in bad[:3]])


def test_tokenizer_unicode_fuzz_against_the_reference(tmp_path):
    """Non-ASCII captions (accents, Greek, Cyrillic, CJK, emoji, combining marks, non-breaking / ideographic spaces) with
    `ftfy.fix_text` as the identity on BOTH sides (it is not installed; the product calls it when present): everything
    after the ftfy step - html unescape, white-space cleaning, the unicode-class regex split, byte-level BPE - is pinned."""
    import random
    import subprocess
    import sys
    import pytest
    if not os.path.isdir("/root/reference/vitlens/src/open_clip"):
        pytest.skip("/root/reference not present")
    from open_clip.tokenizer import tokenize
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rng = random.Random(5)
    words = ["café", "naïve", "Ångström", "straße", "ΑΒΓ", "λόγος", "Москва", "привет", "東京", "深度图", "점군", "🙂", "👍🏽", "é", "ﬁ", "№5",
             "½", "x²", "100€", "a b", "全角　空白", "ＡＢＣ", "İstanbul", "ǅ", "don’t", "“quoted”", "—dash—", "…", "a photo of", "the"]
    seps = [" ", "  ", "\t", " ", " , ", ". ", ""]
    texts = ["".join(rng.choice(words) + rng.choice(seps) for _ in range(rng.choice([1, 2, 4, 9, 40]))) for _ in range(500)]
    p = tmp_path / "texts.json"
    json.dump(texts, open(p, "w"))
    r = subprocess.run([sys.executable, "-c", _REF_TOK, os.path.join(root, "oracle"), str(p)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    ref = json.loads(r.stdout[r.stdout.index("JSON") + 4:])
    got77, got20 = tokenize(texts), tokenize(texts, context_length=20)
    bad = [i for i in range(len(texts)) if got77[i].tolist() != ref["ids77"][i] or got20[i].tolist() != ref["ids20"][i]]
    assert not bad, (len(bad), [texts[i] for i in bad[:3]])
