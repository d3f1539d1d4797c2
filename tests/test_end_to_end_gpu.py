"""Config 1 of BASELINE.json on the GPU: the reference's real corpus (16 ICICI
chunks) through chunker -> WordPiece -> rf_encode -> CorpusStore -> VectorRAG /
MCP tools, plus the CorpusStore API the reference's other consumers use.

all-MiniLM-L6-v2 weights and vocab are not available offline, so the encoder has
seeded random weights and the vocabulary is built from the corpus itself; what is
checked is the plumbing and that every stage agrees with the CPU oracles."""
import os

import numpy as np
import pytest

from conftest import record_measurement
from oracle import c_oracle, encoder as oenc, search as osearch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")

QUESTIONS = ["What was ICICI's Q1 net profit and profitability?",      # "chunking_storing (1).py":431
             "How did retail banking perform in Q2?",                   # :432
             "net profit Q1",                                           # test_vector.py:98
             "What is the deposit to funding ratio in Q3?",
             "Basic EPS and diluted EPS for the March quarter"]


@pytest.fixture(scope="module")
def rig(gpu_device):
    from rag_fin_amd import chunker
    from rag_fin_amd.embedder import Embedder
    from rag_fin_amd.rag import VectorRAG
    from rag_fin_amd.service import ingest
    from rag_fin_amd.store import CorpusStore
    from rag_fin_amd.tokenizer import WordPieceTokenizer
    chunks = chunker.build_all_chunks(os.path.join(GOLD, "extract_data"))
    probe = WordPieceTokenizer(["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"])
    words = sorted({w for t in [c["text"] for c in chunks] + QUESTIONS for w in probe.basic_tokens(t)})
    vocab = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + words[::2] + \
            [ch for ch in "abcdefghijklmnopqrstuvwxyz0123456789"] + ["##" + ch for ch in "abcdefghijklmnopqrstuvwxyz0123456789"]
    tok = WordPieceTokenizer(list(dict.fromkeys(vocab)))
    cfg = dict(oenc.MINILM_L6, vocab_size=len(tok.vocab))
    w = oenc.random_weights(cfg, 42)
    emb = Embedder(w, cfg, tokenizer=tok, device=gpu_device)
    store = CorpusStore("fin_chunks", dim=384, capacity=8, device=gpu_device)   # forces growth
    assert ingest(store, emb, chunks) == 16
    rag = VectorRAG("no-key", "fin_chunks", embedder=emb, store=store)
    return dict(chunks=chunks, tok=tok, cfg=cfg, w=w, emb=emb, store=store, rag=rag)


def test_ingest_matches_oracle_embeddings(rig):
    ids, lens = rig["tok"].batch([c["text"] for c in rig["chunks"]], 256)
    assert lens.max() <= 256 and lens.min() > 20
    want = oenc.encode(oenc.round_weights_fp16(rig["w"]), rig["cfg"], ids, lens)
    got = rig["store"].index.get_rows(np.arange(16)).float().cpu().numpy()
    err = np.abs(got - want).max()
    record_measurement("config1_ingest_rows_vs_oracle", max_abs=err, l2_max=np.linalg.norm(got - want, axis=1).max())
    assert err < 5e-4          # fp16 rows: 3 x the 1.6e-4 measured at T = 256 (profiles/r02a_encoder_error.json)
    assert rig["store"].num_entities == 16


def test_search_top5_equals_oracle_on_stored_vectors(rig):
    import torch
    rag, store = rig["rag"], rig["store"]
    c16 = store.index.get_rows(np.arange(16)).cpu().numpy()
    for qtext in QUESTIONS:
        ctx = rag.search(qtext, 5)
        assert [c["rank"] for c in ctx] == [1, 2, 3, 4, 5]
        q16 = store._prepare_queries(rig["emb"].encode([qtext])).cpu().numpy()
        os_, oi = c_oracle.search(q16, c16, 5)
        assert [c["text"] for c in ctx] == [rig["chunks"][i]["text"] for i in oi[0]]
        assert np.allclose([c["score"] for c in ctx], os_[0], atol=1e-6)
        assert all(-1.001 <= c["score"] <= 1.001 for c in ctx)


NORTH_STAR_TOL = 1e-3     # BASELINE.json north_star: "scores within 1e-3 fp16"
MEASURED_TOL = 4.5e-4     # <= 3 x the 1.5e-4 .. 1.7e-4 measured over all pairs (profiles/r02a_encoder_error.json, r02zz_encoder_error.json)


def test_north_star_gpu_pipeline_vs_all_cpu_pipeline(rig):
    """GPU encode -> GPU search against oracle encode -> oracle search on the reference's own
    corpus (vector_rag_mcp/main.py:50-70 end to end): every returned score within 1e-3 of the
    CPU pipeline's score for the same chunk (asserted at 3x the measured error), and the GPU
    ranking is a valid ranking of the CPU scores up to that error (two chunks may swap only if
    their CPU scores are closer than twice the error)."""
    rag = rig["rag"]
    texts = [c["text"] for c in rig["chunks"]]
    wq = oenc.round_weights_fp16(rig["w"])
    ids, lens = rig["tok"].batch(texts, 256)
    c_or = oenc.encode(wq, rig["cfg"], ids, lens).astype(np.float32).astype(np.float16)
    row_of = {t: i for i, t in enumerate(texts)}
    worst = 0.0
    for qtext in QUESTIONS:
        ids, lens = rig["tok"].batch([qtext], 256)
        q_or = oenc.encode(wq, rig["cfg"], ids, lens).astype(np.float32).astype(np.float16)
        full_or = osearch.exact_scores(q_or, c_or)[0]
        os_, oi = c_oracle.search(q_or, c_or, 5)
        ctx = rag.search(qtext, 5)
        rows = [row_of[c["text"]] for c in ctx]
        got = np.array([c["score"] for c in ctx])
        diff = np.abs(got - full_or[rows]).max()
        worst = max(worst, diff)
        assert diff <= MEASURED_TOL <= NORTH_STAR_TOL, diff
        assert np.all(full_or[rows][:-1] >= full_or[rows][1:] - 2 * MEASURED_TOL)      # order valid up to the error
        assert np.all(full_or[rows] >= os_[0][-1] - 2 * MEASURED_TOL)                  # nothing outside the CPU top-5 band
        top6 = np.sort(full_or)[::-1][:6]
        if np.min(top6[:-1] - top6[1:]) > 2 * MEASURED_TOL:                           # well separated: identical ranking
            assert rows == list(oi[0])
    record_measurement("config1_north_star_score_diff", max_abs=worst)


def test_top_k_larger_than_corpus_returns_all_rows_once(rig):
    ctx = rig["rag"].search(QUESTIONS[0], 20)
    assert len(ctx) == 16 and len({c["text"] for c in ctx}) == 16
    scores = [c["score"] for c in ctx]
    assert scores == sorted(scores, reverse=True)


def test_batch_search_equals_single_queries(rig):
    rag = rig["rag"]
    batch = rag.search_batch(QUESTIONS, 3)
    for qtext, ctx in zip(QUESTIONS, batch):
        single = rag.search(qtext, 3)
        assert [c["text"] for c in single] == [c["text"] for c in ctx]
        assert np.allclose([c["score"] for c in single], [c["score"] for c in ctx], atol=2e-3)


def test_mcp_tools_on_the_real_stack(rig):
    from rag_fin_amd import mcp_server
    mcp_server.set_rag(rig["rag"])
    try:
        r = mcp_server.search_vectors("net profit Q1", 3)
        assert r["status"] == "success" and r["result_count"] == 3
        assert mcp_server.get_collection_stats()["total_chunks"] == 16
        assert mcp_server.health_check()["total_chunks"] == 16
        big = mcp_server.search_vectors("net profit Q1", 100)    # > RF_MAX_K: paged path, 16 rows exist
        assert big["status"] == "success" and big["result_count"] == 16
    finally:
        mcp_server.set_rag(None)


def test_store_query_and_pymilvus_shaped_hits(rig):
    store = rig["store"]
    rows = store.query(expr="", limit=3, output_fields=["id", "period", "chunk_type"])   # test_vector.py:35-39
    assert len(rows) == 3 and rows[0]["id"] == "icici_q1_fy2024_profitability_analysis"
    got = store.query(expr='id in ["icici_q2_fy2024_key_ratios", "missing", "icici_q1_fy2024_key_ratios"]',
                      output_fields=["id", "text", "period", "embedding"])               # graph_cons.py:308-311
    assert [g["id"] for g in got] == ["icici_q2_fy2024_key_ratios", "icici_q1_fy2024_key_ratios"]
    assert len(got[0]["embedding"]) == 384 and abs(np.linalg.norm(got[0]["embedding"]) - 1) < 2e-3
    hits = store.search(rig["emb"].encode(["deposits"]), "embedding", {"metric_type": "COSINE"}, 2,
                        output_fields=["id", "text"])[0]
    assert hits[0].entity.get("id") == hits[0].id and hits[0].score >= hits[1].score
    with pytest.raises(ValueError):
        store.search(rig["emb"].encode(["x"]), "embedding", {"metric_type": "L2"}, 2)
    with pytest.raises(ValueError, match="duplicate"):
        store.add(["icici_q1_fy2024_key_ratios"], ["t"], np.zeros((1, 384), np.float32), ["p"], ["c"], ["s"], [0])


def test_float32_insert_path_and_drop(gpu_device):
    from rag_fin_amd.store import CorpusStore
    rng = np.random.default_rng(0)
    vec = rng.standard_normal((100, 384)).astype(np.float32) * 3.0       # un-normalised, like raw encode()
    st = CorpusStore("t", dim=384, capacity=128, device=gpu_device)
    st.insert([[f"k{i}" for i in range(100)], ["x"] * 100, vec.tolist(), ["p"] * 100, ["c"] * 100,
               ["s"] * 100, [0.0] * 100])
    got = st.index.get_rows(np.arange(100)).float().cpu().numpy()
    want = osearch.l2_normalize_f32(vec)
    assert np.abs(got - want).max() < 1e-3                                # one fp16 ulp at |x| < 1
    hits = st.search(vec[7:8], "embedding", {"metric_type": "COSINE"}, 1, output_fields=["id"])[0]
    assert hits[0].id == "k7" and abs(hits[0].score - 1) < 2e-3
    st.drop()
    assert st.num_entities == 0 and st.search(vec[:1], limit=3) == [[]]


def test_save_and_load_round_trip(rig, tmp_path, gpu_device):
    """On-disk corpus format: vectors + scalar columns survive
    a save/load cycle bit for bit and search results are unchanged."""
    from rag_fin_amd.store import CorpusStore
    store = rig["store"]
    store.save(str(tmp_path / "corpus"))
    assert (tmp_path / "corpus" / "vectors.f16").stat().st_size == 16 * 384 * 2
    again = CorpusStore.load_from(str(tmp_path / "corpus"), device=gpu_device)
    assert again.num_entities == 16 and again.columns == store.columns
    a = store.index.get_rows(np.arange(16)).cpu().numpy().view(np.uint16)
    b = again.index.get_rows(np.arange(16)).cpu().numpy().view(np.uint16)
    assert np.array_equal(a, b)
    q = rig["emb"].encode([QUESTIONS[0]])
    h1 = store.search(q, "embedding", {"metric_type": "COSINE"}, 5, output_fields=["id"])[0]
    h2 = again.search(q, "embedding", {"metric_type": "COSINE"}, 5, output_fields=["id"])[0]
    assert [h.id for h in h1] == [h.id for h in h2] and [h.score for h in h1] == [h.score for h in h2]
    # appending after a load keeps working (resume)
    again.add(["extra"], ["t"], np.ones((1, 384), np.float32), ["p"], ["c"], ["s"], [1.0])
    assert again.num_entities == 17 and again.query(expr='id in ["extra"]')[0]["id"] == "extra"
    with pytest.raises(ValueError):
        (tmp_path / "corpus" / "vectors.f16").write_bytes(b"xx")
        CorpusStore.load_from(str(tmp_path / "corpus"), device=gpu_device)
