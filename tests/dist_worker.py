"""Worker for the world_size-2 gloo rehearsal of the row-sharded search (CPU).

Each rank owns a contiguous range of the candidate order.  The shard-local search
is a STUB built from the oracle's exact pieces (test infrastructure) because there
is no GPU here; everything after it -- query exchange, the single all-gather of
candidate records, the host merge with certification and k' escalation -- is the
product code path (omni_recall_rag_amd.sharded + orr_merge_candidates)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def make_corpus(n, dim, seed):
    from helpers import random_corpus
    rng = np.random.default_rng(seed)
    c = random_corpus(rng, n, dim)
    order = np.argsort(-c["created"], kind="stable")
    return {"emb": [c["emb"][i] for i in order], "created": c["created"][order],
            "contents": [c["contents"][i] for i in order], "dim": dim}


def oracle_shard_search(P, orc, c, lo, hi):
    """Returns shard_search(q, terms, now, kprime, limit, out=...) over rows [lo, hi)."""
    import torch

    def search(q_all, terms_all, now, kprime, limit, out=None, mode=0, topk=0):
        B = len(terms_all)
        n_part = max(0, min(max(1, limit) - lo, hi - lo))
        rec = np.zeros((B, kprime + 1), dtype=P.CAND_DTYPE)
        rec["row_id"] = -1
        rec["order_key"] = -1
        qn = None if q_all is None else q_all.cpu().numpy()
        for b in range(B):
            scored = []
            for r in range(lo, lo + n_part):
                e = c["emb"][r]
                use_cos = qn is not None and qn.shape[1] == c["dim"] and c["dim"] > 0
                dot = orc.dot(qn[b], e) if (use_cos and e is not None) else 0.0
                nb = orc.dot(e, e) if (e is not None and c["dim"] > 0) else 0.0
                low = P.text.lower_invariant(c["contents"][r])
                m = sum(1 for t in terms_all[b] if t in low)
                cosv = 0.0
                if use_cos:
                    na = orc.dot(qn[b], qn[b])
                    cosv = 0.0 if (na <= 0 or nb <= 0) else dot / (np.sqrt(na) * np.sqrt(nb))
                kw = m / len(terms_all[b]) if terms_all[b] else 0.0
                s = (cosv * 0.7) + (kw * 0.2) + (orc.recency(int(c["created"][r]), now) * 0.1)
                scored.append((s, r, dot, nb, m))
            scored.sort(key=lambda t: (-(t[0]) if t[0] == t[0] else float("inf"), t[1]))
            keep = scored[:kprime]
            for i, (s, r, dot, nb, m) in enumerate(keep):
                rec[b, i] = (s, dot, nb, c["created"][r], r, r, m, P.native.ORR_CAND_DOT_EXACT)
            cut = -np.inf if n_part <= kprime or not keep else keep[-1][0]
            rec[b, kprime] = (cut, 0, 0, 0, -1, n_part, len(keep), P.native.ORR_CAND_TRAILER)
        raw = torch.from_numpy(rec.view(np.uint8).reshape(-1).copy())
        if out is not None:
            out.copy_(raw)
            return out
        return raw
    return search


def run(rank, world, port, n, dim, seed, result_path, use_gpu=False, long_queries=False, nccl=False):
    import torch
    import torch.distributed as dist
    import importlib
    import __graft_entry__ as graft
    from oracle import oracle_py as orc
    P = graft.load_package()
    sharded = importlib.import_module(graft.PKG_NAME + ".sharded")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if nccl:        # RCCL itself (one rank per card: on a one-GPU box that is a world of one), collectives on device buffers
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    c = make_corpus(n, dim, seed)
    bounds = [n * r // world for r in range(world + 1)]
    lo, hi = bounds[rank], bounds[rank + 1]
    if use_gpu:      # real shards: every rank keeps its rows on cuda:0 (<= 6 processes may share the card)
        from helpers import build_index
        sub = {"emb": c["emb"][lo:hi], "created": c["created"][lo:hi], "contents": c["contents"][lo:hi], "dim": dim}
        front = sharded.ShardedRecallSearch(build_index(sub, device=rank if nccl else 0, row_base=lo), dim,
                                            torch.device("cuda", rank) if nccl else "cpu")
    else:
        front = sharded.ShardedRecallSearch(None, dim, "cpu", shard_search=oracle_shard_search(P, orc, c, lo, hi))
    front.always_collect = bool(nccl)          # (a world of one issues its collectives too: RCCL on the box's one card)
    rng = np.random.default_rng(100 + rank)
    B_local = 2
    results = []
    texts_all = ["alpha the helm", "kubernetes", "what is the", "GAMMA zzz"]
    for trial, (topk, limit, kprime) in enumerate([(5, n, 8), (10, 300, 4), (3, n, 2), (12, n, 3)]):
        q = torch.from_numpy(rng.standard_normal((B_local, dim)).astype(np.float32))
        if nccl:
            q = q.to(front.device)
        texts = [texts_all[(rank * B_local + i + trial) % 4] for i in range(B_local)]
        terms = [P.text.query_terms(t) for t in texts]
        rows, scores, counts = front.search(q, terms, 639144000000000000, topk, limit, kprime=kprime)
        corpus = orc.OracleCorpus(c["emb"], c["created"], c["contents"])
        for i in range(B_local):
            orow, osc, _ = corpus.search(q[i].cpu().numpy(), texts[i], 639144000000000000, topk, candidate_limit=limit)
            ok = list(rows[i, :counts[i]]) == list(orow) and np.array_equal(scores[i, :counts[i]], osc)
            results.append(bool(ok))
    expected = 8
    if nccl:
        results.append(bool(front.rccl_ranks_seen() == world))
        expected += 1
    # ---- queries that cannot be certified in a larger batch: a zero vector without terms scores 0.1 * recency only, so its whole
    # top-3 are the newest rows, all on shard 0, whose k' = 3 cut-off then EQUALS the third best score (and the rows of a
    # document share their timestamp: exact ties across the cut) -> not certified; queries whose top-3 spread over shards are.
    # ONLY the tied queries may be repeated (as a compacted sub-batch), not the whole batch.
    B4 = 4
    q4 = rng.standard_normal((B4, dim)).astype(np.float32)
    q4[1] = 0.0
    texts4 = ["alpha the helm", "", "kubernetes", "GAMMA zzz"]
    esc0, escq0 = front.escalations, front.escalated_queries
    q4t = torch.from_numpy(q4).to(front.device) if nccl else torch.from_numpy(q4)
    rows, scores, counts = front.search(q4t, [P.text.query_terms(t) if t else [] for t in texts4],
                                        639144000000000000, 3, n, kprime=3)
    corpus4 = orc.OracleCorpus(c["emb"], c["created"], c["contents"])
    for i in range(B4):
        orow, osc, _ = corpus4.search(q4[i], texts4[i], 639144000000000000, 3, candidate_limit=n)
        results.append(bool(list(rows[i, :counts[i]]) == list(orow) and np.array_equal(scores[i, :counts[i]], osc)))
    esc, escq = front.escalations - esc0, front.escalated_queries - escq0
    print("tied-query trial: escalations", esc, "escalated queries", escq, flush=True)
    results.append(bool(esc >= 1 and world <= escq < esc * world * B4))      # repeats happened, and never for the whole batch
    expected += B4 + 1
    if long_queries:
        # rank 1 originates a query whose terms alone exceed the first collective's budget; rank 0 one with 300 terms
        long_text = " ".join(["kubernetes"] + ["w%04dxyzxyzxyzxyz" % i for i in range(60)]) if rank == 1 else \
                    " ".join(["helm"] + ["t%d" % i for i in range(300)])
        texts = [long_text, "alpha"]
        terms = [P.text.query_terms(t) for t in texts]
        assert sum(len(t) for t in terms[0]) > 255
        q = torch.from_numpy(rng.standard_normal((B_local, dim)).astype(np.float32))
        before = front.collectives
        rows, scores, counts = front.search(q, terms, 639144000000000000, 5, n, kprime=8)
        assert front.collectives - before >= 3, "the long terms must have taken the second query collective"
        corpus = orc.OracleCorpus(c["emb"], c["created"], c["contents"])
        for i in range(B_local):
            orow, osc, _ = corpus.search(q[i].numpy(), texts[i], 639144000000000000, 5, candidate_limit=n)
            results.append(bool(list(rows[i, :counts[i]]) == list(orow) and np.array_equal(scores[i, :counts[i]], osc)))
        # the one-origin form: rank 0 holds the whole batch, the others pass nothing
        qb = torch.from_numpy(np.random.default_rng(5).standard_normal((3, dim)).astype(np.float32))
        tb = ["alpha the helm", long_text if rank == 0 else "", "GAMMA zzz"]
        rows, scores, counts = front.search_from(0, qb if rank == 0 else None,
                                                 [P.text.query_terms(t) for t in tb] if rank == 0 else None,
                                                 639144000000000000, 4, n, kprime=6)
        tb0 = ["alpha the helm", " ".join(["helm"] + ["t%d" % i for i in range(300)]), "GAMMA zzz"]
        for i in range(3):
            orow, osc, _ = corpus.search(qb[i].numpy(), tb0[i], 639144000000000000, 4, candidate_limit=n)
            results.append(bool(list(rows[i, :counts[i]]) == list(orow) and np.array_equal(scores[i, :counts[i]], osc)))
        expected += B_local + 3
    dist.barrier()
    dist.destroy_process_group()
    with open(result_path + ".%d" % rank, "w") as f:
        f.write("ok" if all(results) and len(results) == expected else "FAIL %r" % results)


if __name__ == "__main__":
    run(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), sys.argv[7],
        use_gpu=len(sys.argv) > 8 and sys.argv[8] in ("gpu", "nccl"), long_queries=len(sys.argv) > 9 and sys.argv[9] == "long",
        nccl=len(sys.argv) > 8 and sys.argv[8] == "nccl")
