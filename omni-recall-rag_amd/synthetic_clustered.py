"""A CLUSTERED synthetic corpus: the unfriendly case for a screening pass.

`synthetic.py` draws every component independently, so cosines against a query sit at 0 +- 0.018 and one planted
row stands out; real embedding corpora are nothing like that.  Here every row is one of N_CENTROIDS centroids plus
0.1 x noise, and a query is a row plus 0.05 x noise: the ~n/64 rows of the query's cluster all have a cosine of
about 0.99 with it, within 0.002 of each other -- closer than the int8 screen's per-pair bound can tell apart, so
thousands of (query,row) pairs survive it and must be re-scored exactly.  Same interface as `synthetic`
(timestamps, contents and query texts are the same functions of the row index).
"""
from __future__ import annotations

import torch

from .synthetic import (CHUNKS_PER_DOC, NOW_TICKS, SEED, TOKENS_PER_ROW, VOCAB, _lsr, _unit_fixed, contents,  # noqa: F401
                        created_ticks, planted_rows, query_texts, splitmix64, token_ids)

N_CENTROIDS = 64
NOISE = 0.1
QUERY_NOISE = 0.05
PLANTED_WINS = False          # a newer row of the same cluster can outrank the planted one (recency 0.1 vs a cosine gap of 0.009)


def cluster_of(rows: torch.Tensor, seed: int = SEED) -> torch.Tensor:
    return _lsr(splitmix64(rows ^ (seed + 6)), 3) % N_CENTROIDS


def _centroids(c: torch.Tensor, dim: int, device, seed: int) -> torch.Tensor:
    col = torch.arange(dim, dtype=torch.int64, device=device).unsqueeze(0)
    return _unit_fixed(splitmix64((c.unsqueeze(1) * dim + col) ^ (seed + 7)))


def _rows(r: torch.Tensor, dim: int, device, seed: int) -> torch.Tensor:
    col = torch.arange(dim, dtype=torch.int64, device=device).unsqueeze(0)
    noise = _unit_fixed(splitmix64((r.unsqueeze(1) * dim + col) ^ seed))
    return (_centroids(cluster_of(r, seed), dim, device, seed) + NOISE * noise).contiguous()


def embeddings(row0: int, n: int, dim: int, device="cpu", seed: int = SEED) -> torch.Tensor:
    return _rows(torch.arange(row0, row0 + n, dtype=torch.int64, device=device), dim, device, seed)


def query_vectors(b0: int, B: int, dim: int, n_total: int, device="cpu", seed: int = SEED) -> torch.Tensor:
    r = torch.tensor(planted_rows(b0, B, n_total, seed), dtype=torch.int64, device=device)
    col = torch.arange(dim, dtype=torch.int64, device=device).unsqueeze(0)
    b = torch.arange(b0, b0 + B, dtype=torch.int64, device=device).unsqueeze(1)
    noise = _unit_fixed(splitmix64((b * dim + col) ^ (seed + 1)))
    return (_rows(r, dim, device, seed) + QUERY_NOISE * noise).contiguous()
