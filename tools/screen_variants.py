#!/usr/bin/env python3
"""Times the screening GEMM (dots-to-S form) alone: python tools/screen_variants.py [rows] [B].
ORR_SCREEN_MODE=1/2/3 removes the MFMAs / the LDS-DMA requests / the stores (diagnostic)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as graft
P = graft.load_package()
syn = importlib.import_module(graft.PKG_NAME + ".synthetic")
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dim = 3072
dev = torch.device("cuda", 0)
idx = P.RecallIndex(dim=dim, capacity_rows=rows)
for r0 in range(0, rows, 32768):
    m = min(32768, rows - r0)
    pool, off = syn.contents(r0, m, dev)
    idx.append(syn.embeddings(r0, m, dim, dev), syn.created_ticks(r0, m, rows, dev), pool, off)
idx.seal()
q = syn.query_vectors(0, B, dim, rows, dev).cpu().numpy()
out = torch.empty((B, rows), dtype=torch.float32, device=dev)
import ctypes as C
N = P.native
idx.set_profiling(True)
for _ in range(6):
    N.check(N.hip.orr_index_screen_dots(idx._h, B, dim, q.ctypes.data, out.data_ptr()))
st = idx.kernel_stats()["screen_bf16"]
ms = st["total_ms"] / st["launches"]
print(f"mode={os.environ.get('ORR_SCREEN_MODE','0')} rows={rows} B={B} screen_bf16 {ms:.3f} ms  {2.0*rows*dim*B/ms/1e9:.0f} TFLOP/s  E-read {2.0*rows*dim/ms/1e6:.0f} GB/s")
