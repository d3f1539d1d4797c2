"""Host-side stages of the text ingest (Embedder.encode_to_device) on the GPU box: which stage the host spends its
time in, and how the native tokenizer's thread count changes it (a cgroup CPU quota below the affinity mask
throttles the whole process after a burst of threads)."""
import sys, time, json, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import encoder as oenc, synth_text
from rag_fin_amd.embedder import Embedder
from rag_fin_amd.tokenizer import WordPieceTokenizer
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/sys/fs/cgroup/cpu.stat"):
    try:
        print(f, open(f).read().strip().replace("\n", " | "))
    except OSError as e:
        print(f, "absent")
dev = torch.device("cuda:0")
cfg = dict(oenc.MINILM_L6)
tok = WordPieceTokenizer(synth_text.vocab_for(size=cfg["vocab_size"]))
emb = Embedder(oenc.random_weights(cfg, 0), cfg, tokenizer=tok, device=dev)
texts = synth_text.retemplated_texts(10000, 11)
emb.encode_to_device(texts[:256]); torch.cuda.synchronize()
print("affinity", len(os.sched_getaffinity(0)))
import torch as _t
from rag_fin_amd.hostcpu import cpu_budget
print("cpu budget", cpu_budget(), "torch threads", _t.get_num_threads())
for nt in (0, 8, 16, 64):
    os.environ["RAGFIN_TOKENIZER_THREADS"] = str(nt)
    tot = []
    for r in range(5):
        t = time.perf_counter(); v = emb.encode_to_device(texts); t_enq = time.perf_counter() - t
        torch.cuda.synchronize(); tot.append(time.perf_counter() - t)
    st = emb.ingest_stats
    print("threads=%d: total_s %s ; last run stages %s" % (nt, [round(x, 3) for x in tot],
          {k: (round(x, 4) if isinstance(x, float) else x) for k, x in st.items()}))
os.environ["RAGFIN_TOKENIZER_THREADS"] = "0"
for ct in (1024, 2048, 4096):
    os.environ["RAGFIN_INGEST_CHUNK_TEXTS"] = str(ct)
    tot = []
    for r in range(6):
        t = time.perf_counter(); v = emb.encode_to_device(texts); torch.cuda.synchronize(); tot.append(time.perf_counter() - t)
    print("chunk_texts=%d: total_s %s chunks %d buckets %d" % (ct, [round(x, 4) for x in tot], emb.ingest_stats["chunks"], emb.ingest_stats["buckets"]))
os.environ.pop("RAGFIN_INGEST_CHUNK_TEXTS")
try:
    print("/sys/fs/cgroup/cpu.stat", open("/sys/fs/cgroup/cpu.stat").read().strip().replace("\n", " | "))
except OSError:
    pass
