"""A keyword-heavy synthetic corpus: the unfriendly case for the keyword chain.

`synthetic.py` draws 120 six-letter tokens per row from a vocabulary of 4096: every query term equals
exactly one vocabulary token and the vocabulary scan is over in microseconds.  A real 1M-chunk corpus has 10^5..10^6
distinct tokens of mixed length, a Zipf-like frequency profile, and query terms that are substrings of many tokens.
Here: 2^18 token ids drawn log-uniformly (rank r with probability ~ 1/r, in integer arithmetic so that CPU and GPU
agree bit for bit), token lengths 3..12 bytes with one id in 32 stretched to 17..24 (the wave-per-token path of the
vocabulary scan), and queries of two whole tokens of the planted row plus a 3-letter piece of a third, which is a
substring of dozens to hundreds of vocabulary tokens.  Embeddings, timestamps and query vectors are `synthetic`'s.
"""
from __future__ import annotations

from typing import List, Tuple

import torch

from .synthetic import (NOW_TICKS, SEED, TOKENS_PER_ROW, _lsr, created_ticks, embeddings, planted_rows,  # noqa: F401
                        query_vectors, splitmix64)

VOCAB_BITS = 18
MAX_LEN = 24
PLANTED_WINS = True


def token_ids(rows: torch.Tensor, seed: int = SEED) -> torch.Tensor:
    """[n, TOKENS_PER_ROW] ids in [1, 2^VOCAB_BITS): bit length uniform, mantissa uniform (a staircase of 1/r)."""
    j = torch.arange(TOKENS_PER_ROW, dtype=torch.int64, device=rows.device).unsqueeze(0)
    h = splitmix64((rows.unsqueeze(1) * TOKENS_PER_ROW + j) ^ (seed + 11))
    bits = _lsr(h, 8) % VOCAB_BITS                                    # 0 .. VOCAB_BITS-1
    mant = _lsr(h, 24) & ((torch.ones_like(bits) << bits) - 1)
    return (torch.ones_like(bits) << bits) | mant


def _token_lengths(ids: torch.Tensor, seed: int) -> torch.Tensor:
    h = splitmix64(ids ^ (seed + 12))
    short = 3 + _lsr(h, 4) % 10                                       # 3 .. 12
    long_ = 17 + _lsr(h, 20) % 8                                      # 17 .. 24
    return torch.where((_lsr(h, 40) & 31) == 0, long_, short)


def _token_chars(ids: torch.Tensor, seed: int) -> torch.Tensor:
    """[..., MAX_LEN] lowercase letters of the tokens (positions past the length are garbage)."""
    k = torch.arange(MAX_LEN, dtype=torch.int64, device=ids.device)
    h = splitmix64((ids.unsqueeze(-1) * MAX_LEN + k) ^ (seed + 13))
    return (_lsr(h, 16) % 26 + ord("a")).to(torch.uint8)


def contents(row0: int, n: int, device="cpu", seed: int = SEED) -> Tuple[torch.Tensor, torch.Tensor]:
    """Lowercase ASCII content: (pool uint8, offsets int64 [n+1]); tokens joined by single spaces."""
    rows = torch.arange(row0, row0 + n, dtype=torch.int64, device=device)
    ids = token_ids(rows, seed)                                       # [n, T]
    lens = _token_lengths(ids, seed)                                  # [n, T]
    chars = _token_chars(ids, seed)                                   # [n, T, MAX_LEN]
    k = torch.arange(MAX_LEN + 1, dtype=torch.int64, device=device)
    space = torch.full(chars.shape[:-1] + (1,), ord(" "), dtype=torch.uint8, device=device)
    cells = torch.cat([chars, space], dim=-1)                         # [n, T, MAX_LEN+1]: the token, then its separator
    keep = k < lens.unsqueeze(-1)                                     # the token's letters ...
    sep = (k == MAX_LEN) & (torch.arange(TOKENS_PER_ROW, device=device) < TOKENS_PER_ROW - 1).view(1, -1, 1)
    mask = keep | sep                                                 # ... and a space behind every token but the row's last
    pool = cells[mask]
    row_bytes = lens.sum(dim=1) + (TOKENS_PER_ROW - 1)
    off = torch.zeros(n + 1, dtype=torch.int64, device=device)
    off[1:] = torch.cumsum(row_bytes, dim=0)
    return pool.contiguous(), off


def token_text(t: int, seed: int = SEED) -> bytes:
    ids = torch.tensor([t], dtype=torch.int64)
    ln = int(_token_lengths(ids, seed)[0])
    return bytes(_token_chars(ids, seed)[0, :ln].tolist())


def query_texts(b0: int, B: int, n_total: int, seed: int = SEED) -> List[str]:
    """Two whole tokens of the planted row, a 3-letter piece of a third, and a stop word."""
    rows = torch.tensor(planted_rows(b0, B, n_total, seed), dtype=torch.int64)
    ids = token_ids(rows, seed)
    texts = []
    for i in range(B):
        b = b0 + i
        pick = [int(ids[i, (7 * b + 11 * kk) % TOKENS_PER_ROW]) for kk in range(3)]
        w = [token_text(t, seed).decode() for t in pick]
        piece = w[2][(b % max(1, len(w[2]) - 2)):][:3]
        texts.append(f"{w[0]} the {w[1].upper()} {piece}")
    return texts
