"""Deterministic synthetic corpus and queries, written with torch
integer ops so the same code produces the data on the GPU (bench, large tests)
and on the CPU (the oracle's bounded sample) from row indices alone.

Every row is a pure function of its GLOBAL candidate position, so any shard or
sub-range can be regenerated anywhere.  Rows are generated already in candidate
order: CreatedAt is non-increasing in the row index.
"""
from __future__ import annotations

from typing import List, Tuple

import torch

SEED = 20260515
NOW_TICKS = 639144000000000000          # 2026-05-15T00:00:00Z in DateTime ticks
TICKS_PER_DAY = 864000000000
CHUNKS_PER_DOC = 8                      # chunks of one document share CreatedAtUtc (DocumentIngestionService.cs:102)
TOKENS_PER_ROW = 120                    # IngestionOptions chunk size in words
VOCAB = 4096
WORD_LEN = 6
ROW_BYTES = TOKENS_PER_ROW * (WORD_LEN + 1) - 1   # single spaces between tokens, none trailing
STOP_FILLERS = ["the", "of", "what", "is", "for"]


def _i64(v: int) -> int:
    v &= (1 << 64) - 1
    return v - (1 << 64) if v >= (1 << 63) else v


_C0, _C1, _C2 = _i64(0x9E3779B97F4A7C15), _i64(0xBF58476D1CE4E5B9), _i64(0x94D049BB133111EB)


def _lsr(x: torch.Tensor, s: int) -> torch.Tensor:
    return (x >> s) & ((1 << (64 - s)) - 1)


def splitmix64(x: torch.Tensor) -> torch.Tensor:
    """splitmix64 finaliser on int64 tensors (two's-complement wrap-around)."""
    z = x + _C0
    z = (z ^ _lsr(z, 30)) * _C1
    z = (z ^ _lsr(z, 27)) * _C2
    return z ^ _lsr(z, 31)


def _unit_fixed(h: torch.Tensor) -> torch.Tensor:
    """top 24 bits -> k * 2^-23 with k in [-2^23, 2^23): exactly representable in fp32."""
    k = _lsr(h, 40) - (1 << 23)
    return k.to(torch.float32) * (1.0 / (1 << 23))


def embeddings(row0: int, n: int, dim: int, device="cpu", seed: int = SEED) -> torch.Tensor:
    r = torch.arange(row0, row0 + n, dtype=torch.int64, device=device).unsqueeze(1)
    c = torch.arange(dim, dtype=torch.int64, device=device).unsqueeze(0)
    return _unit_fixed(splitmix64((r * dim + c) ^ seed)).contiguous()


def created_ticks(row0: int, n: int, n_total: int, device="cpu", seed: int = SEED, now: int = NOW_TICKS) -> torch.Tensor:
    """Spread over one year, newest first; every CHUNKS_PER_DOC rows share a timestamp."""
    r = torch.arange(row0, row0 + n, dtype=torch.int64, device=device)
    doc = r // CHUNKS_PER_DOC
    n_docs = (n_total + CHUNKS_PER_DOC - 1) // CHUNKS_PER_DOC
    step = max(1, (365 * TICKS_PER_DAY) // max(1, n_docs))
    jitter = _lsr(splitmix64(doc ^ (seed + 2)), 1) % step
    return now - (doc * step + jitter)


def _vocab_table(device) -> torch.Tensor:
    t = torch.arange(VOCAB, dtype=torch.int64, device=device)
    h = _lsr(splitmix64(t ^ (SEED + 5)), 8)
    cols = [(t // 676) % 26, (t // 26) % 26, t % 26, h % 26, (h // 26) % 26, (h // 676) % 26]
    letters = torch.stack(cols, dim=1) + ord("a")
    space = torch.full((VOCAB, 1), ord(" "), dtype=torch.int64, device=device)
    return torch.cat([letters, space], dim=1).to(torch.uint8)      # [VOCAB, 7]


def token_ids(row0: int, n: int, device="cpu", seed: int = SEED) -> torch.Tensor:
    r = torch.arange(row0, row0 + n, dtype=torch.int64, device=device).unsqueeze(1)
    j = torch.arange(TOKENS_PER_ROW, dtype=torch.int64, device=device).unsqueeze(0)
    return splitmix64((r * TOKENS_PER_ROW + j) ^ (seed + 3)) & (VOCAB - 1)


def contents(row0: int, n: int, device="cpu", seed: int = SEED) -> Tuple[torch.Tensor, torch.Tensor]:
    """Lowercase ASCII content: (pool uint8 [n*ROW_BYTES], offsets uint64-as-int64 [n+1])."""
    tab = _vocab_table(device)
    ids = token_ids(row0, n, device, seed)
    rows = tab[ids].reshape(n, TOKENS_PER_ROW * (WORD_LEN + 1))[:, :ROW_BYTES].contiguous()
    off = torch.arange(n + 1, dtype=torch.int64, device=device) * ROW_BYTES
    return rows.reshape(-1), off


def vocab_word(t: int) -> bytes:
    return bytes(_vocab_table("cpu")[t, :WORD_LEN].tolist())


def planted_rows(b0: int, B: int, n_total: int, seed: int = SEED) -> List[int]:
    b = torch.arange(b0, b0 + B, dtype=torch.int64)
    return (_lsr(splitmix64(b ^ (seed + 4)), 1) % n_total).tolist()


def query_vectors(b0: int, B: int, dim: int, n_total: int, device="cpu", seed: int = SEED) -> torch.Tensor:
    """q_b = e[r*_b] + 0.25 * noise_b: the planted row r*_b is the cosine winner."""
    r = torch.tensor(planted_rows(b0, B, n_total, seed), dtype=torch.int64, device=device).unsqueeze(1)
    c = torch.arange(dim, dtype=torch.int64, device=device).unsqueeze(0)
    b = torch.arange(b0, b0 + B, dtype=torch.int64, device=device).unsqueeze(1)
    e = _unit_fixed(splitmix64((r * dim + c) ^ seed))                      # = embeddings(r, 1, dim) row by row
    noise = _unit_fixed(splitmix64((b * dim + c) ^ (seed + 1)))
    return (e + 0.25 * noise).contiguous()


def query_texts(b0: int, B: int, n_total: int, seed: int = SEED) -> List[str]:
    """Three tokens of the planted row plus stop-word filler in mixed case."""
    rows = torch.tensor(planted_rows(b0, B, n_total, seed), dtype=torch.int64).unsqueeze(1)
    b = torch.arange(b0, b0 + B, dtype=torch.int64).unsqueeze(1)
    j = (7 * b + 11 * torch.arange(3, dtype=torch.int64).unsqueeze(0)) % TOKENS_PER_ROW        # [B, 3] token slots
    picks = (splitmix64((rows * TOKENS_PER_ROW + j) ^ (seed + 3)) & (VOCAB - 1)).tolist()        # = token_ids(row)[slot]
    tab = _vocab_table("cpu")[:, :WORD_LEN].numpy()
    f = STOP_FILLERS
    texts = []
    for i in range(B):
        w = [bytes(tab[t]).decode() for t in picks[i]]
        bb = b0 + i
        texts.append(f"{f[bb % 5].capitalize()} {w[0]} {f[(bb + 1) % 5]} {w[1].upper()} {w[2]}")
    return texts
