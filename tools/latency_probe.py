import sys, os, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
from oracle import search as osearch
from rag_fin_amd.store import GpuIndex
dev = torch.device("cuda:0")
n, d = 100_000, 384
c = osearch.synth_unit_rows(n, d, 1)
ix = GpuIndex(d, n, dev); ix.add(torch.from_numpy(c).to(dev))
for B in (1, 8, 64):
    q = torch.from_numpy(osearch.synth_unit_rows(B, d, 2)).to(dev)
    for _ in range(5): ix.search_profile(q, 5)
    st = [ix.search_profile(q, 5) for _ in range(50)]
    print("B", B, {k: round(float(np.median([s[k] for s in st])) * 1e3, 1) for k in st[0]}, "us")
    torch.cuda.synchronize()
    ts = []
    for _ in range(200):
        t = time.perf_counter(); r = ix.search_raw(q, 5); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    print("   search_raw + sync p50 %.1f us" % (np.median(ts) * 1e6))
