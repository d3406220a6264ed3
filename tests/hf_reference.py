"""The HuggingFace side of tests/test_gpu_parity.py::test_baseline_dims_match_huggingface_live, in a process of its own
(torch's wheel brings its own ROCm libraries; the test process keeps to the system's).  Test infrastructure only.

usage: python hf_reference.py <dims> <ftype> <out_dir>  ->  <out_dir>/hf_<dims>_<ftype>.bin (the model file) and
<out_dir>/hf_reference.npz (ids<i>, want<i>: sentences and their mean-pooled, normalised embeddings in f32 arithmetic).
exit code 77: torch / transformers are not importable."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main(dims, ftype, out_dir):
    try:
        import torch
        import transformers
    except Exception:
        return 77
    from bert_cpp_amd import ggml_file as gf
    gf.MODEL_DIMS.setdefault("bert-base-l2", gf.BertHParams(30522, 512, 768, 3072, 12, 2))
    hp = gf.MODEL_DIMS[dims]
    torch.manual_seed(4321)
    cfg = transformers.BertConfig(vocab_size=hp.n_vocab, hidden_size=hp.n_embd, num_hidden_layers=hp.n_layer, num_attention_heads=hp.n_head,
                                  intermediate_size=hp.n_intermediate, max_position_embeddings=hp.n_max_tokens, hidden_act="gelu_new",
                                  layer_norm_eps=1e-5, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    model = transformers.BertModel(cfg, add_pooling_layer=False).eval()
    g = torch.Generator().manual_seed(99)
    with torch.no_grad():
        for name, p in model.named_parameters():       # (HF's N(0, 0.02) init scaled up: attention and GELU away from their linear range)
            if p.ndim == 2 and "embeddings" not in name:
                p.mul_(2.5)
            elif p.ndim == 2:
                p.mul_(20.0)
            else:
                p.add_(0.1 * torch.randn(p.shape, generator=g))
    sd = {k: v.detach().numpy().astype(np.float32) for k, v in model.state_dict().items() if k != "embeddings.position_ids"}
    path = os.path.join(out_dir, f"hf_{dims}_{ftype}.bin")
    gf.write_model(path, hp, sd, gf.FTYPE_BY_NAME[ftype])
    # the HF model gets the matrices the file holds
    stored = gf.read_model(path)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if p.ndim == 2:
                p.copy_(torch.from_numpy(np.ascontiguousarray(stored.dequantized(name), dtype=np.float32)))
    rng = np.random.default_rng(11)
    lens = [128, 77, 128, 5, 1, 64] if hp.n_max_tokens < 512 or dims == "minilm-l6" else [512, 300, 128, 17]
    sents = [rng.integers(1000, hp.n_vocab, size=n).astype(np.int32) for n in lens]
    out = {"n": np.int64(len(sents))}
    with torch.no_grad():
        for i, ids in enumerate(sents):
            h = model(input_ids=torch.tensor(ids[None].astype(np.int64))).last_hidden_state[0]
            e = h.mean(dim=0)
            out[f"ids{i}"] = ids
            out[f"want{i}"] = (e / e.norm()).numpy().astype(np.float64)
    np.savez(os.path.join(out_dir, "hf_reference.npz"), **out)
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1], sys.argv[2], sys.argv[3]))
