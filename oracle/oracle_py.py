"""ctypes front-end for oracle/liborr_oracle.so plus an independent numpy
restatement used to cross-check the C oracle on small cases.

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package never imports this module.

Reference followed (read as text): RecallSearchService.cs:13-119,
InMemoryIngestionStore.cs:57-65, TextSnippetHelper.cs:5-11.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import subprocess
from typing import List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liborr_oracle.so")


def build(force: bool = False) -> str:
    """Compile the C oracle (gcc, a second or two)."""
    src = os.path.join(_HERE, "recall_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "liborr_oracle.so"])
    return _LIB_PATH


class _Corpus(C.Structure):
    _fields_ = [
        ("n_chunks", C.c_int64),
        ("emb", C.c_void_p),
        ("emb_off", C.c_void_p),
        ("emb_len", C.c_void_p),
        ("created_ticks", C.c_void_p),
        ("content", C.c_void_p),
        ("content_off", C.c_void_p),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.orc_cosine.restype = C.c_double
        L.orc_cosine.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]
        L.orc_dot.restype = C.c_double
        L.orc_dot.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.orc_keyword_score.restype = C.c_double
        L.orc_keyword_score.argtypes = [C.c_char_p, C.c_int64, C.c_char_p, C.c_int64]
        L.orc_recency.restype = C.c_double
        L.orc_recency.argtypes = [C.c_int64, C.c_int64]
        L.orc_round4.restype = C.c_double
        L.orc_round4.argtypes = [C.c_double]
        L.orc_query_terms.restype = C.c_int32
        L.orc_query_terms.argtypes = [C.c_char_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32]
        L.orc_is_blank.restype = C.c_int32
        L.orc_is_blank.argtypes = [C.c_char_p, C.c_int64]
        L.orc_lower_invariant.restype = C.c_int64
        L.orc_lower_invariant.argtypes = [C.c_char_p, C.c_int64, C.c_void_p, C.c_int64]
        L.orc_snippet.restype = C.c_int64
        L.orc_snippet.argtypes = [C.c_char_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int64]
        L.orc_recent_chunks.restype = C.c_int64
        L.orc_recent_chunks.argtypes = [C.POINTER(_Corpus), C.c_int64, C.c_void_p]
        L.orc_search.restype = C.c_int64
        L.orc_search.argtypes = [C.POINTER(_Corpus), C.c_int64, C.c_void_p, C.c_int64, C.c_char_p, C.c_int64,
                                 C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_score_all.restype = C.c_int64
        L.orc_score_all.argtypes = [C.POINTER(_Corpus), C.c_int64, C.c_void_p, C.c_int64, C.c_char_p, C.c_int64,
                                    C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class OracleCorpus:
    """Chunk table in store enumeration order (CosmosIngestionRecords.cs:19-30).

    embeddings: a 2-D float32 array [n, dim], or a list whose entries are
    1-D float32 sequences / None (ragged: null, empty or odd-dimension rows).
    contents:   list of bytes/str, or (pool uint8 array, offsets int64[n+1]).
    """

    def __init__(self, embeddings, created_ticks: Sequence[int], contents):
        created = np.ascontiguousarray(np.asarray(created_ticks, dtype=np.int64))
        n = int(created.shape[0])
        if isinstance(embeddings, np.ndarray) and embeddings.ndim == 2:
            emb = np.ascontiguousarray(embeddings, dtype=np.float32)
            assert emb.shape[0] == n
            dim = emb.shape[1]
            emb_off = np.arange(n, dtype=np.int64) * dim
            emb_len = np.full(n, dim, dtype=np.int32)
            emb_flat = emb.reshape(-1)
        else:
            rows = [np.zeros(0, np.float32) if e is None else np.asarray(e, dtype=np.float32).reshape(-1)
                    for e in embeddings]
            assert len(rows) == n
            emb_len = np.array([r.shape[0] for r in rows], dtype=np.int32)
            emb_off = np.zeros(n, dtype=np.int64)
            if n:
                emb_off[1:] = np.cumsum(emb_len.astype(np.int64))[:-1]
            emb_flat = np.concatenate(rows) if n and int(emb_len.sum()) else np.zeros(1, np.float32)
            emb_flat = np.ascontiguousarray(emb_flat, dtype=np.float32)
        if isinstance(contents, tuple):
            pool, off = contents
            pool = np.ascontiguousarray(pool, dtype=np.uint8)
            off = np.ascontiguousarray(off, dtype=np.int64)
        else:
            bs = [c.encode("utf-8") if isinstance(c, str) else bytes(c) for c in contents]
            assert len(bs) == n
            off = np.zeros(n + 1, dtype=np.int64)
            if n:
                off[1:] = np.cumsum([len(b) for b in bs])
            pool = np.frombuffer(b"".join(bs) + b"\0", dtype=np.uint8).copy()
        self.n = n
        self.emb, self.emb_off, self.emb_len = emb_flat, emb_off, emb_len
        self.created, self.pool, self.content_off = created, pool, off
        self._c = _Corpus(n, _ptr(emb_flat), _ptr(emb_off), _ptr(emb_len), _ptr(created), _ptr(pool), _ptr(off))

    def content_bytes(self, r: int) -> bytes:
        return bytes(self.pool[self.content_off[r]:self.content_off[r + 1]])

    def recent_chunks(self, max_count: int) -> np.ndarray:
        out = np.zeros(max(self.n, 1), dtype=np.int64)
        k = lib().orc_recent_chunks(C.byref(self._c), max_count, _ptr(out))
        return out[:k].copy()

    def search(self, qvec, query, now_ticks: int, topk: int, candidate_limit: int = 300, threads: int = 1):
        """RecallSearchService.SearchAsync up to the ranked list:
        returns (rows int64[k], scores float64[k], rounded float64[k])."""
        q = np.ascontiguousarray(np.asarray(qvec, dtype=np.float32).reshape(-1))
        qb = query.encode("utf-8") if isinstance(query, str) else bytes(query)
        kk = max(1, int(topk))
        rows = np.zeros(kk, dtype=np.int64)
        sc = np.zeros(kk, dtype=np.float64)
        rd = np.zeros(kk, dtype=np.float64)
        qp = _ptr(q) if q.shape[0] else None
        k = lib().orc_search(C.byref(self._c), candidate_limit, qp, q.shape[0], qb, len(qb),
                             now_ticks, topk, threads, _ptr(rows), _ptr(sc), _ptr(rd))
        assert k >= 0
        return rows[:k].copy(), sc[:k].copy(), rd[:k].copy()

    def score_all(self, qvec, query, now_ticks: int, candidate_limit: int, threads: int = 1):
        q = np.ascontiguousarray(np.asarray(qvec, dtype=np.float32).reshape(-1))
        qb = query.encode("utf-8") if isinstance(query, str) else bytes(query)
        order = np.zeros(max(self.n, 1), dtype=np.int64)
        sc = np.zeros(max(self.n, 1), dtype=np.float64)
        qp = _ptr(q) if q.shape[0] else None
        k = lib().orc_score_all(C.byref(self._c), candidate_limit, qp, q.shape[0], qb, len(qb),
                                now_ticks, threads, _ptr(order), _ptr(sc))
        return order[:k].copy(), sc[:k].copy()


def cosine(a, b) -> float:
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float32).reshape(-1))
    bb = None if b is None else np.ascontiguousarray(np.asarray(b, dtype=np.float32).reshape(-1))
    return lib().orc_cosine(_ptr(a) if a.shape[0] else None, a.shape[0],
                            None if bb is None or bb.shape[0] == 0 else _ptr(bb),
                            0 if bb is None else bb.shape[0])


def dot(a, b) -> float:
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float32).reshape(-1))
    b = np.ascontiguousarray(np.asarray(b, dtype=np.float32).reshape(-1))
    assert a.shape == b.shape
    return lib().orc_dot(_ptr(a), _ptr(b), a.shape[0])


def keyword_score(query, content) -> float:
    qb = query.encode("utf-8") if isinstance(query, str) else bytes(query)
    cb = content.encode("utf-8") if isinstance(content, str) else bytes(content)
    return lib().orc_keyword_score(qb, len(qb), cb, len(cb))


def recency(created_ticks: int, now_ticks: int) -> float:
    return lib().orc_recency(created_ticks, now_ticks)


def round4(x: float) -> float:
    return lib().orc_round4(x)


def query_terms(query) -> List[bytes]:
    qb = query.encode("utf-8") if isinstance(query, str) else bytes(query)
    cap = 4 * len(qb) + 16
    buf = np.zeros(cap, dtype=np.uint8)
    off = np.zeros(len(qb) + 2, dtype=np.int32)
    t = lib().orc_query_terms(qb, len(qb), _ptr(buf), cap, _ptr(off), off.shape[0])
    assert t >= 0
    return [bytes(buf[off[i]:off[i + 1]]) for i in range(t)]


def is_blank(s) -> bool:
    b = s.encode("utf-8") if isinstance(s, str) else bytes(s)
    return bool(lib().orc_is_blank(b, len(b)))


def lower_invariant(s) -> bytes:
    b = s.encode("utf-8") if isinstance(s, str) else bytes(s)
    out = np.zeros(4 * len(b) + 8, dtype=np.uint8)
    m = lib().orc_lower_invariant(b, len(b), _ptr(out), out.shape[0])
    assert m >= 0
    return bytes(out[:m])


def snippet(content, max_chars: int = 180) -> bytes:
    b = content.encode("utf-8") if isinstance(content, str) else bytes(content)
    out = np.zeros(4 * len(b) + 16, dtype=np.uint8)
    m = lib().orc_snippet(b, len(b), max_chars, _ptr(out), out.shape[0])
    assert m >= 0
    return bytes(out[:m])


# ---------------------------------------------------------------------------
# Independent restatement in numpy / plain Python (small cases only).  Written
# separately from the C so that two readings of the C# have to agree.
# ---------------------------------------------------------------------------

_WS = set([0x09, 0x0A, 0x0B, 0x0C, 0x0D, 0x20, 0x85, 0xA0, 0x1680, 0x2028, 0x2029, 0x202F, 0x205F, 0x3000]
          + list(range(0x2000, 0x200B)))
_STOP = {"a", "an", "and", "are", "as", "at", "be", "by", "for", "from", "how", "in", "is", "it", "of",
         "on", "or", "that", "the", "to", "was", "what", "when", "where", "which", "who", "why", "with"}


def _py_lower(s: str) -> str:
    out = []
    for ch in s:
        if ch == "İ":
            out.append(ch)
            continue
        lo = ch.lower()
        out.append(lo if len(lo) == 1 else ch)
    return "".join(out)


def _py_split(s: str) -> List[str]:
    toks, cur = [], []
    for ch in s:
        if ord(ch) in _WS:
            if cur:
                toks.append("".join(cur))
                cur = []
        else:
            cur.append(ch)
    if cur:
        toks.append("".join(cur))
    return toks


def py_query_terms(query: str) -> List[str]:
    raw = []
    for t in _py_split(query):                    # RecallSearchService.cs:95
        t = _py_lower(t)                          # :96
        if t not in raw:                          # :97
            raw.append(t)
    kept = [t for t in raw if t not in _STOP]     # :103-105
    return kept if kept else raw                  # :107-108


def py_keyword_score(query: str, content: str) -> float:
    if all(ord(c) in _WS for c in query) or all(ord(c) in _WS for c in content):   # :92-93
        return 0.0
    terms = py_query_terms(query)
    if not terms:
        return 0.0
    low = _py_lower(content)                      # :110
    return sum(1 for t in terms if t in low) / len(terms)    # :111-112


def py_cosine(a, b) -> float:
    a = np.asarray(a, dtype=np.float32).reshape(-1)
    if b is None:
        return 0.0
    b = np.asarray(b, dtype=np.float32).reshape(-1)
    if a.shape[0] == 0 or b.shape[0] == 0 or a.shape[0] != b.shape[0]:            # :71-72
        return 0.0
    # float32 array products are rounded to binary32; cumsum over float64 is a
    # left-to-right sequential sum (:77-82)
    dot = float(np.cumsum((a * b).astype(np.float64))[-1])
    na = float(np.cumsum((a * a).astype(np.float64))[-1])
    nb = float(np.cumsum((b * b).astype(np.float64))[-1])
    if na <= 0.0 or nb <= 0.0:                    # :84-85
        return 0.0
    return dot / (math.sqrt(na) * math.sqrt(nb))  # :87


def py_recency(created_ticks: int, now_ticks: int) -> float:
    age_days = max(0.0, float(now_ticks - created_ticks) / 864000000000.0)        # :117
    return math.exp(-age_days / 30.0)             # :118


def py_search(embeddings, created, contents, qvec, query: str, now_ticks: int, topk: int,
              candidate_limit: int = 300):
    n = len(created)
    order = sorted(range(n), key=lambda r: -created[r])[:max(1, candidate_limit)]   # stable; IMStore:59-63
    scored = []
    for r in order:
        s = (py_cosine(qvec, embeddings[r]) * 0.7) + (py_keyword_score(query, contents[r]) * 0.2) \
            + (py_recency(created[r], now_ticks) * 0.1)                              # :66
        scored.append((r, s))

    def key(item):
        r, s = item
        nan = math.isnan(s)
        return (0 if not nan else 1, -s if not nan else 0.0, -created[r])           # :34-35, NaN last
    scored.sort(key=key)                          # Python's sort is stable
    top = scored[:max(1, topk)]                   # :36
    return [r for r, _ in top], [s for _, s in top]
