#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/ (run in the build container, which
has `transformers` + torch CPU; the GPU box only reads the committed outputs).

1. hf_tiny_f32.bin + hf_tiny_golden.json
   An INDEPENDENT float implementation (HuggingFace `BertModel`, random init, fixed seed) of the
   architecture bert.cpp evaluates, exported in bert.cpp's file format the way the reference's
   models/convert-to-ggml.py:84-108 does (state-dict order, reversed dims, 1-D tensors f32), and
   its mean-pooled, L2-normalised sentence embeddings for a few variable-length id sequences.
   HF config is set to the numerics the reference's ggml ops use: tanh-approximation GELU
   ("gelu_new") and LayerNorm eps 1e-5.  The CPU oracle (plain mode) and
   the HIP path must both reproduce these.

2. tokenizer_golden.json
   The four known-answer vectors of reference examples/test_tokenizer.cpp:70-73 together with a
   SPARSE vocab reconstructed from them (the real 30522-entry vocab file is not available
   offline): every id that the expected vectors mention gets the word piece that the reference
   algorithm (bert.cpp:252-325) must have matched at that position; all other ids are
   "[unusedN]".  Multi-piece words are split by hand below (PIECES) following the real
   bert-base-uncased vocabulary.
"""
import json
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from bert_cpp_amd import ggml_file as gf  # noqa: E402


def make_hf_tiny():
    import torch
    from transformers import BertConfig, BertModel

    torch.manual_seed(1234)
    cfg = BertConfig(vocab_size=200, hidden_size=64, num_hidden_layers=2, num_attention_heads=2,
                     intermediate_size=128, max_position_embeddings=48, hidden_act="gelu_new",
                     layer_norm_eps=1e-5, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    model = BertModel(cfg, add_pooling_layer=False).eval()
    # HF init is N(0, 0.02): scale matrices up so attention / GELU are exercised non-trivially
    with torch.no_grad():
        g = torch.Generator().manual_seed(99)
        for name, p in model.named_parameters():
            if p.ndim == 2 and "embeddings" not in name:
                p.mul_(6.0)
            elif p.ndim == 2:
                p.mul_(30.0)
            elif "LayerNorm.weight" in name:
                p.add_(0.1 * torch.randn(p.shape, generator=g))
            else:
                p.add_(0.1 * torch.randn(p.shape, generator=g))
    sd = {k: v.detach().numpy().astype(np.float32) for k, v in model.state_dict().items()
          if k not in ("embeddings.position_ids",)}
    hp = gf.BertHParams(200, 48, 64, 128, 2, 2)
    assert set(gf.tensor_names(2)) <= set(sd.keys()), set(gf.tensor_names(2)) - set(sd.keys())
    gf.write_model(os.path.join(HERE, "hf_tiny_f32.bin"), hp, sd, gf.FTYPE_F32)

    rng = np.random.default_rng(7)
    sents = []
    for n in (1, 2, 5, 17, 32, 33, 48):
        ids = rng.integers(0, 200, size=n).tolist()
        sents.append(ids)
    embs = []
    hiddens = []
    with torch.no_grad():
        for ids in sents:
            out = model(input_ids=torch.tensor([ids]), output_hidden_states=True)
            h = out.last_hidden_state[0]            # [N, H]
            e = h.mean(dim=0)
            e = e / e.norm()
            embs.append(e.numpy().astype(np.float64).tolist())
            hiddens.append([float(x) for x in out.hidden_states[1][0, 0, :8]])   # layer-1, token 0, first 8
    with open(os.path.join(HERE, "hf_tiny_golden.json"), "w") as f:
        json.dump({"model": "hf_tiny_f32.bin", "sentences": sents, "embeddings": embs,
                   "layer1_tok0_first8": hiddens,
                   "note": "HF BertModel(gelu_new, eps=1e-5), mean-pool + L2; see make_golden.py"}, f)
    print("wrote hf_tiny_f32.bin, hf_tiny_golden.json")


# ------------------------------------------------------------------------------------------------
# tokenizer goldens from reference examples/test_tokenizer.cpp:70-73
# ------------------------------------------------------------------------------------------------
TOK_TESTS = [
    ("Québec", [101, 5447, 102]),
    ("syömme \t  täällä    tänään", [101, 25353, 5358, 4168, 11937, 25425, 9092, 14634, 102]),
    ("I'm going to the store to buy 3 apples and a banana! You're welcome to come along if you'd like. The time is 2:30 p.m. and it's partly cloudy outside. I'll be back soon, so don't go anywhere.",
     [101, 1045, 1005, 1049, 2183, 2000, 1996, 3573, 2000, 4965, 1017, 18108, 1998, 1037, 15212, 999, 2017, 1005, 2128, 6160, 2000, 2272, 2247, 2065, 2017, 1005, 1040, 2066, 1012, 1996, 2051, 2003, 1016, 1024, 2382, 1052, 1012, 1049, 1012, 1998, 2009, 1005, 1055, 6576, 24706, 2648, 1012, 1045, 1005, 2222, 2022, 2067, 2574, 1010, 2061, 2123, 1005, 1056, 2175, 5973, 1012, 102]),
    ("\"5 2 + 3 * 4 -\"; int stack[1000], top = -1; int calculate(int a, int b, char operator) { return operator == '+' ? a + b : operator == '-' ? a - b : operator == '*' ? a * b : a / b; } void push(int x) { stack[++top] = x; } int pop() { return stack[top--]; } int evaluatePostfix(char* expression) { for (int i = 0; expression[i]; i++) { if (isdigit(expression[i])) push(expression[i] - '0'); else { int a = pop(), b = pop(); push(calculate(b, a, expression[i])); } } return pop(); } int result = evaluatePostfix(input);",
     [101, 1000, 1019, 1016, 1009, 1017, 1008, 1018, 1011, 1000, 1025, 20014, 9991, 1031, 6694, 1033, 1010, 2327, 1027, 1011, 1015, 1025, 20014, 18422, 1006, 20014, 1037, 1010, 20014, 1038, 1010, 25869, 6872, 1007, 1063, 2709, 6872, 1027, 1027, 1005, 1009, 1005, 1029, 1037, 1009, 1038, 1024, 6872, 1027, 1027, 1005, 1011, 1005, 1029, 1037, 1011, 1038, 1024, 6872, 1027, 1027, 1005, 1008, 1005, 1029, 1037, 1008, 1038, 1024, 1037, 1013, 1038, 1025, 1065, 11675, 5245, 1006, 20014, 1060, 1007, 1063, 9991, 1031, 1009, 1009, 2327, 1033, 1027, 1060, 1025, 1065, 20014, 3769, 1006, 1007, 1063, 2709, 9991, 1031, 2327, 1011, 1011, 1033, 1025, 1065, 20014, 16157, 19894, 8873, 2595, 1006, 25869, 1008, 3670, 1007, 1063, 2005, 1006, 20014, 1045, 1027, 1014, 1025, 3670, 1031, 1045, 1033, 1025, 1045, 1009, 1009, 1007, 1063, 2065, 1006, 2003, 4305, 23806, 1006, 3670, 1031, 1045, 1033, 1007, 1007, 5245, 1006, 3670, 1031, 1045, 1033, 1011, 1005, 1014, 1005, 1007, 1025, 2842, 1063, 20014, 1037, 1027, 3769, 1006, 1007, 1010, 1038, 1027, 3769, 1006, 1007, 1025, 5245, 1006, 18422, 1006, 1038, 1010, 1037, 1010, 3670, 1031, 1045, 1033, 1007, 1007, 1025, 1065, 1065, 2709, 3769, 1006, 1007, 1025, 1065, 20014, 2765, 1027, 16157, 19894, 8873, 2595, 1006, 7953, 1007, 1025, 102]),
]

# words that the expected vectors split into several pieces (first piece, then "##" pieces)
PIECES = {
    "syomme": ["sy", "##om", "##me"],
    "taalla": ["ta", "##alla"],
    "tanaan": ["tan", "##aan"],
    "partly": ["partly"],
    "cloudy": ["cloudy"],
    "evaluatepostfix": ["evaluate", "##post", "##fi", "##x"],
    "isdigit": ["is", "##di", "##git"],
}

ACCENTS = {}
for grp, r in [("ÀÁÂÃÄÅ", "A"), ("àáâãäå", "a"), ("ÈÉÊË", "E"), ("èéêë", "e"), ("ÌÍÎÏ", "I"), ("ìíîï", "i"),
               ("ÒÓÔÕÖ", "O"), ("òóôõö", "o"), ("ÙÚÛÜ", "U"), ("ùúûü", "u"), ("Ý", "Y"), ("ý", "y"),
               ("Ç", "C"), ("ç", "c"), ("Ñ", "N"), ("ñ", "n")]:
    for ch in grp:
        ACCENTS[ch] = r


def ref_words(text: str):
    """Word split of the reference for well-formed UTF-8 input (bert.cpp:206-282)."""
    t = "".join(ACCENTS.get(ch, ch) for ch in text)
    t = "".join(ch.lower() if "A" <= ch <= "Z" else ch for ch in t)
    b = t.encode("utf-8")
    return [m.group().decode() for m in re.finditer(rb"[!-/:-@\[-`{-~]|[A-Za-z]+|[0-9]+", b)]


def make_tokenizer_golden():
    id2piece = {}
    for text, ids in TOK_TESTS:
        words = ref_words(text)
        body = ids[1:-1]
        k = 0
        for w in words:
            pcs = PIECES.get(w, [w])
            assert "".join(p[2:] if p.startswith("##") else p for p in pcs) == w, w
            for p in pcs:
                i = body[k]; k += 1
                assert id2piece.get(i, p) == p, (i, id2piece.get(i), p)
                id2piece[i] = p
        assert k == len(body), (text[:20], k, len(body))
    # consistency: one id per piece string
    inv = {}
    for i, p in id2piece.items():
        assert inv.get(p, i) == i, (p, inv.get(p), i)
        inv[p] = i
    n_vocab = 30522
    vocab = [f"[unused{i}]" for i in range(n_vocab)]
    vocab[0], vocab[100], vocab[101], vocab[102], vocab[103] = "[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"
    for i, p in id2piece.items():
        vocab[i] = p
    with open(os.path.join(HERE, "tokenizer_golden.json"), "w") as f:
        json.dump({"source": "reference examples/test_tokenizer.cpp:70-73",
                   "n_vocab": n_vocab,
                   "sparse_vocab": {str(i): p for i, p in sorted(id2piece.items())},
                   "tests": [{"text": t, "ids": ids} for t, ids in TOK_TESTS]}, f, ensure_ascii=False, indent=0)
    print(f"wrote tokenizer_golden.json ({len(id2piece)} pieces)")


if __name__ == "__main__":
    make_tokenizer_golden()
    make_hf_tiny()
