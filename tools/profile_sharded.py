#!/usr/bin/env python3
"""Where the sharded front-end (omni-recall-rag_amd/sharded.py) spends host time per step, rehearsed on one GPU
(world size 1, no collective): cProfile over 200 steps of B queries against a 1M x 3072 shard."""
import cProfile
import importlib
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import __graft_entry__ as graft  # noqa: E402

P = graft.load_package()
syn = importlib.import_module(graft.PKG_NAME + ".synthetic")
sharded = importlib.import_module(graft.PKG_NAME + ".sharded")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rows, dim = 1_000_000, 3072
dev = torch.device("cuda", 0)
idx = P.RecallIndex(dim=dim, device=0, capacity_rows=rows)
for r0 in range(0, rows, 32768):
    m = min(32768, rows - r0)
    pool, off = syn.contents(r0, m, dev)
    idx.append(syn.embeddings(r0, m, dim, dev), syn.created_ticks(r0, m, rows, dev), pool, off)
idx.seal()
front = sharded.ShardedRecallSearch(idx, dim, dev)
steps = 220
qs = [syn.query_vectors(s * B, B, dim, rows, dev) for s in range(steps)]
ts = [P.PackedTerms(P.pack_terms([P.text.query_terms(t) for t in syn.query_texts(s * B, B, rows)])) for s in range(steps)]
torch.cuda.synchronize()
for s in range(20):
    front.search(qs[s], ts[s], syn.NOW_TICKS, 10, rows)
t0 = time.perf_counter()
for s in range(20, 120):
    front.search(qs[s], ts[s], syn.NOW_TICKS, 10, rows)
print("front ms/step", 1e3 * (time.perf_counter() - t0) / 100)
t0 = time.perf_counter()
for s in range(20, 120):
    idx.search(qs[s], ts[s], syn.NOW_TICKS, 10, candidate_limit=rows)
print("direct ms/step", 1e3 * (time.perf_counter() - t0) / 100)
pr = cProfile.Profile()
pr.enable()
for s in range(120, 220):
    front.search(qs[s], ts[s], syn.NOW_TICKS, 10, rows)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
