"""The chunker restatement is pinned byte for byte by the reference's own golden
file (FinRag_knowledge_graph/chunks.json -> tests/golden/chunks_golden.json) on the
reference's own inputs (extract_data/ -> tests/golden/extract_data/)."""
import json
import os
import random

from rag_fin_amd import chunker

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def golden():
    with open(os.path.join(GOLD, "chunks_golden.json"), encoding="utf-8") as f:
        return {r["id"]: r for r in json.load(f)}


def test_all_16_chunks_byte_exact():
    chunks = chunker.build_all_chunks(os.path.join(GOLD, "extract_data"))
    gold = golden()
    assert len(chunks) == 16 == len(gold)
    assert [c["id"] for c in chunks][:4] == [
        "icici_q1_fy2024_profitability_analysis", "icici_q1_fy2024_balance_sheet_health",
        "icici_q1_fy2024_key_ratios", "icici_q1_fy2024_segment_performance"]
    for c in chunks:
        g = gold[c["id"]]
        assert c["text"] == g["text"], c["id"]
        assert len(c["text"]) == g["size"]
        assert c["chunk_type"] == g["type"] and c["period"] == g["period"]
        assert c["statement_type"] == "consolidated"


def test_primary_values_of_q1():
    by = {c["id"]: c for c in chunker.build_all_chunks(os.path.join(GOLD, "extract_data"))}
    # SURVEY.md 8c sample values
    assert by["icici_q1_fy2024_profitability_analysis"]["primary_value"] == 10636.12
    assert by["icici_q1_fy2024_balance_sheet_health"]["primary_value"] == 2039897.4
    assert by["icici_q1_fy2024_key_ratios"]["primary_value"] == 15.22
    assert abs(by["icici_q1_fy2024_segment_performance"]["primary_value"] - 87473.33) < 1e-6


def test_file_order_does_not_matter():
    folder = os.path.join(GOLD, "extract_data", "icici_q2_2023")
    docs = chunker.load_quarter(folder)
    base = chunker.create_complete_context_chunks(docs, "Q2_FY2024")
    rng = random.Random(0)
    for _ in range(5):
        rng.shuffle(docs)
        assert chunker.create_complete_context_chunks(docs, "Q2_FY2024") == base


def test_missing_documents_and_period_keys():
    assert chunker.create_complete_context_chunks([], "Q1_FY2024") == []
    assert chunker.create_complete_context_chunks([{"reportType": "OTHER"}], "Q1_FY2024") == []
    assert chunker.period_keys("Q1_FY2024") == ("june2023", "june2022")
    assert chunker.period_keys("Q4_FY2024") == ("march2024", "march2023")
    assert chunker.period_keys("Q3_FY2023") == ("december2022", "december2021")
    assert chunker.period_keys("H1") == (None, None)
    folder = os.path.join(GOLD, "extract_data", "icici_q1_2023")
    docs = [d for d in chunker.load_quarter(folder) if d.get("reportType") == "CONSOLIDATED FINANCIAL RESULTS"]
    only_fin = chunker.create_complete_context_chunks(docs, "Q1_FY2024")
    assert [c["chunk_type"] for c in only_fin] == ["profitability_analysis", "financial_ratios"]


def test_insert_columns_order():
    chunks = chunker.build_all_chunks(os.path.join(GOLD, "extract_data"))[:2]
    cols = chunker.insert_columns(chunks, [[0.0], [1.0]])
    assert len(cols) == 7 and cols[0] == [c["id"] for c in chunks] and cols[2] == [[0.0], [1.0]]
    assert cols[6] == [c["primary_value"] for c in chunks]
