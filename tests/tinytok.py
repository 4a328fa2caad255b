"""A tiny CLIP-style BPE vocabulary written by the tests themselves (no vocabulary ships with the reference, and there is no
network): 29 symbols, their end-of-word forms, four merges, BOS / EOS.  ``write(dir, pad="!")`` gives the SDXL second
tokenizer's convention (padding with id 0, the id of "!")."""
import json
import os

LETTERS = list("!,.abcdefghijklmnopqrstuvwxyz")
MERGES = [("c", "a"), ("ca", "t</w>"), ("d", "o"), ("do", "g</w>")]


def vocab():
    v = {}
    for ch in LETTERS:
        v[ch] = len(v)
    v["</w>"] = len(v)
    for ch in LETTERS:
        v[ch + "</w>"] = len(v)
    for a, b in MERGES:
        v[a + b] = len(v)
    v["<|startoftext|>"] = len(v)
    v["<|endoftext|>"] = len(v)
    return v


def write(d, pad="<|endoftext|>"):
    os.makedirs(d, exist_ok=True)
    json.dump(vocab(), open(os.path.join(d, "vocab.json"), "w"))
    open(os.path.join(d, "merges.txt"), "w").write("#version: 0.2\n" + "\n".join(f"{a} {b}" for a, b in MERGES) + "\n")
    sp = {"bos_token": "<|startoftext|>", "eos_token": "<|endoftext|>", "unk_token": "<|endoftext|>", "pad_token": pad}
    json.dump(sp, open(os.path.join(d, "special_tokens_map.json"), "w"))
    json.dump(dict(sp, model_max_length=77, tokenizer_class="CLIPTokenizer", do_lower_case=True),
              open(os.path.join(d, "tokenizer_config.json"), "w"))
    return d
