"""CPU baseline of the embedder for bench.py's cpu_baseline leg.  TEST / BENCH INFRASTRUCTURE
ONLY (same rules as oracle/encoder.py).

The reference embeds on the host with SentenceTransformer('all-MiniLM-L6-v2').encode(texts)
("chunking_storing (1).py":379-380, vector_rag_mcp/main.py:50), i.e. transformers' BertModel in
fp32 on torch-CPU + mean-pool + L2-normalise.  sentence-transformers is not installed and the
checkpoint does not exist offline, so the baseline is the SAME model class (transformers.BertModel
from a local BertConfig), the same seeded random weights as the GPU side, fp32, all host cores:
what the reference's ingest costs per token on this host (`kind: "port"` in the bench line)."""
from __future__ import annotations

import numpy as np


def build_bert(cfg: dict, w: dict):
    """transformers.BertModel carrying the stacked weights `w` (layout of rf_encoder_weights)."""
    import torch
    import transformers
    bc = transformers.BertConfig(vocab_size=cfg["vocab_size"], hidden_size=cfg["hidden"],
                                 num_hidden_layers=cfg["layers"], num_attention_heads=cfg["heads"],
                                 intermediate_size=cfg["intermediate"], max_position_embeddings=cfg["max_position"],
                                 type_vocab_size=cfg["type_vocab"], layer_norm_eps=cfg["ln_eps"], hidden_act="gelu",
                                 hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    model = transformers.BertModel(bc, add_pooling_layer=False).eval()
    H = cfg["hidden"]
    sd = {"embeddings.word_embeddings.weight": w["word_emb"], "embeddings.position_embeddings.weight": w["pos_emb"],
          "embeddings.token_type_embeddings.weight": w["type_emb"], "embeddings.LayerNorm.weight": w["emb_ln_g"],
          "embeddings.LayerNorm.bias": w["emb_ln_b"]}
    for l in range(cfg["layers"]):
        p = f"encoder.layer.{l}."
        qw, kw, vw = w["qkv_w"][l][:H], w["qkv_w"][l][H:2 * H], w["qkv_w"][l][2 * H:]
        qb, kb, vb = w["qkv_b"][l][:H], w["qkv_b"][l][H:2 * H], w["qkv_b"][l][2 * H:]
        sd.update({p + "attention.self.query.weight": qw, p + "attention.self.query.bias": qb,
                   p + "attention.self.key.weight": kw, p + "attention.self.key.bias": kb,
                   p + "attention.self.value.weight": vw, p + "attention.self.value.bias": vb,
                   p + "attention.output.dense.weight": w["ao_w"][l], p + "attention.output.dense.bias": w["ao_b"][l],
                   p + "attention.output.LayerNorm.weight": w["ln1_g"][l], p + "attention.output.LayerNorm.bias": w["ln1_b"][l],
                   p + "intermediate.dense.weight": w["ff1_w"][l], p + "intermediate.dense.bias": w["ff1_b"][l],
                   p + "output.dense.weight": w["ff2_w"][l], p + "output.dense.bias": w["ff2_b"][l],
                   p + "output.LayerNorm.weight": w["ln2_g"][l], p + "output.LayerNorm.bias": w["ln2_b"][l]})
    missing, unexpected = model.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()},
                                                strict=False)
    assert not unexpected and all("position_ids" in m or "token_type_ids" in m for m in missing), (missing, unexpected)
    return model


def encode(model, ids: np.ndarray, lens: np.ndarray, batch_size: int = 32) -> np.ndarray:
    """sentence-transformers' encode() on token ids: batches of 32 (the library default),
    attention mask from lens, mean-pool (clamp 1e-9), F.normalize.  -> float32 [n, H]."""
    import torch
    out = []
    with torch.no_grad():
        for s in range(0, ids.shape[0], batch_size):
            ln = torch.from_numpy(lens[s:s + batch_size].astype(np.int64))
            T = int(ln.max())
            x = torch.from_numpy(ids[s:s + batch_size, :T].astype(np.int64))
            mask = (torch.arange(T)[None, :] < ln[:, None]).to(torch.int64)
            hid = model(input_ids=x, attention_mask=mask).last_hidden_state
            m = mask.unsqueeze(-1).float()
            pooled = (hid * m).sum(1) / torch.clamp(m.sum(1), min=1e-9)
            out.append(torch.nn.functional.normalize(pooled, p=2, dim=1).numpy())
    return np.concatenate(out)
