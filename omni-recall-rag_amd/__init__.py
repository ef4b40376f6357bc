"""omni-recall-rag_amd: MI355X-native drop-in for the reference's hybrid
recall-search hot path (RecallSearchService.cs:26-37 behind
IRecallSearchService.SearchAsync).

The product is the C-ABI library libomnirecall_hip.so (include/omnirecall_hip.h);
this Python package is the harness that tests, benches and shards it.  The
directory name is not a valid Python identifier, so load it through
`__graft_entry__.load_package()` (registered in sys.modules as
`omni_recall_rag_amd`).
"""
from . import _native as native
from ._native import OrrError
from .index import CAND_DTYPE, MicroBatcher, PackedTerms, RecallCluster, RecallIndex, merge_candidates, pack_contents, pack_terms
from . import text
from . import service

__all__ = ["native", "OrrError", "RecallIndex", "RecallCluster", "MicroBatcher", "merge_candidates", "pack_contents", "pack_terms", "PackedTerms", "text", "service",
           "CAND_DTYPE"]
