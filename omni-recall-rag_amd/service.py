"""Python faces of the C++ store/service mirrors in libomnirecall_host.so, named after the
reference types they stand in for (InMemoryIngestionStore, RecallSearchService,
CosmosDocumentRecord, CosmosChunkRecord) so that tests read like the reference's own."""
from __future__ import annotations

import ctypes as C
import json
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

from . import _native as N


class HostError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(message)
        self.code = code


def _check(status: int) -> None:
    if status != 0:
        raise HostError(status, (N.host.orrh_last_error() or b"").decode("utf-8", "replace"))


@dataclass
class CosmosDocumentRecord:
    Id: str
    FileName: str
    CreatedAtTicks: int = 0


@dataclass
class CosmosChunkRecord:
    Id: str
    DocumentId: str
    ChunkIndex: int
    Content: str
    Embedding: Optional[Sequence[float]] = None
    CreatedAtTicks: int = 0


class InMemoryIngestionStore:
    def __init__(self):
        self._h = C.c_void_p(N.host.orrh_store_create())

    def UpsertDocument(self, d: CosmosDocumentRecord) -> None:
        _check(N.host.orrh_store_upsert_document(self._h, d.Id.encode(), d.FileName.encode(), d.CreatedAtTicks))

    def UpsertChunks(self, chunks: List[CosmosChunkRecord]) -> None:
        if not chunks:
            return
        n = len(chunks)
        ids = (C.c_char_p * n)(*[c.Id.encode() for c in chunks])
        contents = (C.c_char_p * n)(*[c.Content.encode() for c in chunks])
        idxs = np.asarray([c.ChunkIndex for c in chunks], dtype=np.int32)
        created = np.asarray([c.CreatedAtTicks for c in chunks], dtype=np.int64)
        embs = [np.zeros(0, np.float32) if c.Embedding is None else np.asarray(c.Embedding, np.float32) for c in chunks]
        emb_len = np.asarray([e.shape[0] for e in embs], dtype=np.int32)
        flat = np.ascontiguousarray(np.concatenate(embs) if int(emb_len.sum()) else np.zeros(1, np.float32))
        _check(N.host.orrh_store_upsert_chunks(self._h, chunks[0].DocumentId.encode(), n, C.cast(ids, C.c_void_p),
                                               idxs.ctypes.data, C.cast(contents, C.c_void_p), flat.ctypes.data,
                                               emb_len.ctypes.data, created.ctypes.data))

    def DeleteDocument(self, document_id: str) -> None:
        _check(N.host.orrh_store_delete_document(self._h, document_id.encode()))

    def ChunkCount(self) -> int:
        return int(N.host.orrh_store_chunk_count(self._h))

    def ImportCosmosJson(self, data) -> tuple:
        """orrh_store_import_cosmos_json: Cosmos-style camelCase items (array, JSON lines or query pages).
        Returns (documents, chunks) upserted; HostError(ORR_EINVAL) leaves the store untouched."""
        raw = data.encode("utf-8") if isinstance(data, str) else bytes(data)
        nd, nc = C.c_int64(0), C.c_int64(0)
        _check(N.host.orrh_store_import_cosmos_json(self._h, raw, len(raw), C.cast(C.byref(nd), C.c_void_p),
                                                    C.cast(C.byref(nc), C.c_void_p)))
        return int(nd.value), int(nc.value)

    def ExportCosmosJson(self) -> bytes:
        """orrh_store_export_cosmos_json: the store as a JSON array of Cosmos-style items."""
        out, ln = C.c_void_p(), C.c_int64()
        _check(N.host.orrh_store_export_cosmos_json(self._h, C.byref(out), C.byref(ln)))
        try:
            return C.string_at(out, ln.value)
        finally:
            N.host.orrh_free(out)

    def close(self):
        if self._h:
            N.host.orrh_store_destroy(self._h)
            self._h = None


class StubQueryEmbeddingClient:
    """IEmbeddingClient returning a fixed vector (RecallSearchServiceTests.cs:120-127)."""

    def __init__(self, vector: Sequence[float]):
        self.vector = np.asarray(vector, dtype=np.float32)

    def Embed(self, text: str) -> np.ndarray:
        return self.vector


class RecallSearchService:
    """IRecallSearchService over the HIP scorer.  candidate_limit=300 reproduces the reference."""

    def __init__(self, store: InMemoryIngestionStore, embedding_client, device: int = 0, candidate_limit: int = 300,
                 now_ticks: Optional[int] = None):
        self.store, self.embedding_client, self.now_ticks = store, embedding_client, now_ticks
        self._h = C.c_void_p(N.host.orrh_service_create(store._h, device, candidate_limit))

    def Search(self, query: str, topK: int, now_ticks: Optional[int] = None) -> dict:
        """SearchAsync: returns the response body as a dict (RecallSearchResponseDto, camelCase)."""
        vec = np.ascontiguousarray(self.embedding_client.Embed(query), dtype=np.float32) \
            if query is not None and query.strip() else np.zeros(0, np.float32)
        out, ln = C.c_void_p(), C.c_int64()
        now = now_ticks if now_ticks is not None else self.now_ticks
        _check(N.host.orrh_service_search_json(self._h, (query or "").encode(), vec.ctypes.data if vec.size else None,
                                               int(vec.size), topK, now, C.byref(out), C.byref(ln)))
        try:
            return json.loads(C.string_at(out, ln.value).decode("utf-8"))
        finally:
            N.host.orrh_free(out)

    def Stats(self) -> dict:
        n, full, delta = C.c_int32(0), C.c_int64(0), C.c_int64(0)
        N.host.orrh_service_stats(self._h, C.cast(C.byref(n), C.c_void_p), C.cast(C.byref(full), C.c_void_p),
                                  C.cast(C.byref(delta), C.c_void_p))
        return {"shards": n.value, "full_rebuilds": full.value, "delta_builds": delta.value,
                "tombstoned_rows": int(N.host.orrh_service_tombstoned_rows(self._h)),
                "compactions": int(N.host.orrh_service_compactions(self._h)),
                "delta_merges": int(N.host.orrh_service_delta_merges(self._h))}

    def close(self):
        if self._h:
            N.host.orrh_service_destroy(self._h)
            self._h = None
