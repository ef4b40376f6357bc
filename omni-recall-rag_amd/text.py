"""Host string semantics (libomnirecall_host.so): what the C# host does with BCL
calls around the native search -- RecallSearchService.cs:22,92-110,
TextSnippetHelper.cs:5-11, Math.Round at :51."""
from __future__ import annotations

import ctypes as C
from typing import List, Union

from . import _native as N

Str = Union[str, bytes]


def _b(s: Str) -> bytes:
    return s.encode("utf-8") if isinstance(s, str) else bytes(s)


def is_blank(s: Str) -> bool:
    b = _b(s)
    return bool(N.host.orrh_is_blank(b, len(b)))


def lower_invariant(s: Str) -> bytes:
    b = _b(s)
    cap = 4 * len(b) + 4
    buf = C.create_string_buffer(cap)
    m = N.host.orrh_lower_invariant(b, len(b), C.cast(buf, C.c_void_p), cap)
    if m < 0:
        raise ValueError("orrh_lower_invariant: buffer too small")
    return buf.raw[:m]


def query_terms(query: Str) -> List[bytes]:
    """queryTerms of RecallSearchService.cs:95-108 as lowercased UTF-8."""
    b = _b(query)
    cap = 4 * len(b) + 16
    buf = C.create_string_buffer(cap)
    off = (C.c_uint32 * (len(b) + 2))()
    t = N.host.orrh_query_terms(b, len(b), C.cast(buf, C.c_void_p), cap, C.cast(off, C.c_void_p), len(off))
    if t < 0:
        raise ValueError("orrh_query_terms: buffer too small")
    return [buf.raw[off[i]:off[i + 1]] for i in range(t)]


def build_snippet(content: Str, max_chars: int = 180) -> bytes:
    b = _b(content)
    cap = 4 * len(b) + 16
    buf = C.create_string_buffer(cap)
    m = N.host.orrh_build_snippet(b, len(b), max_chars, C.cast(buf, C.c_void_p), cap)
    if m < 0:
        raise ValueError("orrh_build_snippet: buffer too small")
    return buf.raw[:m]


def round4(x: float) -> float:
    return N.host.orrh_round4(x)


def has_sufficient_evidence(citation_scores, minimum_citation_count: int, minimum_strong_citation_score: float) -> bool:
    """ChatOrchestrationService.HasSufficientEvidence (ChatOrchestrationService.cs:58-65) over the citations'
    4-decimal scores."""
    import numpy as np
    sc = np.ascontiguousarray(citation_scores, dtype=np.float64)
    return bool(N.host.orrh_has_sufficient_evidence(sc.ctypes.data if sc.size else None, int(sc.size),
                                                    int(minimum_citation_count), float(minimum_strong_citation_score)))


def format_score_f4(rounded_score: float) -> str:
    """The `score={c.Score:F4}` text of the grounded prompt (ChatOrchestrationService.cs:85)."""
    import ctypes as C
    buf = C.create_string_buffer(64)
    n = N.host.orrh_format_score_f4(float(rounded_score), C.cast(buf, C.c_void_p), 64)
    if n < 0:
        raise ValueError("orrh_format_score_f4 failed")
    return buf.raw[:n].decode("utf-8")
