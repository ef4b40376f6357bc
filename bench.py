#!/usr/bin/env python3
"""bench.py -- queries/sec of the hybrid recall-search hot path on MI355X.

Workload (BASELINE.json configs[1], "C2"): 1M chunks x 3072-d fp32 PER GPU, top-k=10,
full hybrid score (cosine + keyword + recency fused 0.7/0.2/0.1), candidate_limit =
whole corpus.  Results are exact (reference arithmetic); by default the library screens the
corpus through its bf16 shadow and re-scores the survivors from the fp32 master,
so the dominant kernel streams 2*N*D bytes; `--set-option two_stage=0` restores the exact kernel
over all 4*N*D bytes.  A step is one pass of the hot path over one batch: every rank
originates ONE query; with N ranks the corpus is N x 1M rows, row-sharded, and each
query is scored against ALL shards (queries all-gathered, per-shard top-k' records
all-gathered over RCCL, exact host finish).  Weak scaling: per-GPU rows are fixed,
corpus and query count grow with N.  value = queries answered per second, whole job.

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Inputs are resident in HBM before the timed region; results land in host memory
inside it.  The oracle is used here ONLY for the `cpu_baseline` leg.
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16 MFMA peak (no sparsity)
MFMA_I8_PEAK_TOPS = 5000.0       # dense int8 MFMA peak (no sparsity)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rows-per-gpu", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=3072)
    ap.add_argument("--batch", type=int, default=1, help="queries originated per rank per step")
    ap.add_argument("--topk", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-rows", type=int, default=262144)
    ap.add_argument("--cpu-sample-queries", type=int, default=32)
    ap.add_argument("--inflight", type=int, default=1,
                    help="single-GPU runs: steps are issued from this many threads, each on its own search lane "
                         "(orr_index_view), so the host finish and the small kernels of one step overlap the "
                         "dominant kernel of another")
    ap.add_argument("--no-overlap-leg", action="store_true", help="skip the extra two-steps-in-flight measurement")
    ap.add_argument("--no-terms", action="store_true", help="diagnostic: queries without keyword terms")
    ap.add_argument("--set-option", action="append", default=[], metavar="NAME=VALUE",
                    help="orr_index_set_option on the shard before the run (e.g. two_stage=1)")
    return ap.parse_args()


def build_shard(P, syn, torch, rank, rows, dim, n_total, dev, options=()):
    idx = P.RecallIndex(dim=dim, device=dev.index or 0, capacity_rows=rows, row_base=rank * rows)
    step = 32768
    for r0 in range(0, rows, step):
        m = min(step, rows - r0)
        g0 = rank * rows + r0
        pool, off = syn.contents(g0, m, dev)
        idx.append(syn.embeddings(g0, m, dim, dev), syn.created_ticks(g0, m, n_total, dev), pool, off)
    torch.cuda.synchronize()
    idx.seal()
    for opt in options:
        name, _, value = opt.partition("=")
        idx.set_option(name, int(value or 1))
    return idx


def cpu_baseline(P, syn, args, n_total, torch, dev):
    """The reference-faithful oracle (kind "port") timed on the host cores over a bounded sample."""
    import numpy as np
    from oracle import oracle_py as orc
    m = min(args.cpu_sample_rows, args.rows_per_gpu)
    nq = args.cpu_sample_queries
    cores = max(1, min(os.cpu_count() or 1, 64))
    # the sample is generated on the GPU (same deterministic generator) and copied to the host
    emb = torch.cat([syn.embeddings(r0, min(32768, m - r0), args.dim, dev).cpu() for r0 in range(0, m, 32768)]).numpy()
    created = syn.created_ticks(0, m, n_total, dev).cpu().numpy()
    pool, off = syn.contents(0, m, dev)
    corpus = orc.OracleCorpus(emb, created, (pool.cpu().numpy(), off.cpu().numpy()))
    qs = syn.query_vectors(0, nq, args.dim, n_total, dev).cpu().numpy()
    texts = syn.query_texts(0, nq, n_total)
    t0 = time.perf_counter()
    for b in range(nq):
        corpus.search(qs[b], texts[b], syn.NOW_TICKS, args.topk, candidate_limit=m, threads=cores)
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    corpus.search(qs[0], texts[0], syn.NOW_TICKS, args.topk, candidate_limit=m, threads=1)
    dt1 = time.perf_counter() - t1
    rows_per_s = m * nq / dt
    # an optimised CPU variant beside the reference-faithful one, labelled as such -- cosine part only,
    # fp32 BLAS GEMM over the same sample with precomputed norms (NOT the reference arithmetic), top-k by partition
    norms = np.concatenate([np.sqrt(np.square(emb[i:i + 8192], dtype=np.float64).sum(axis=1)) for i in range(0, m, 8192)]).astype(np.float32)
    qn = np.sqrt(np.square(qs, dtype=np.float64).sum(axis=1)).astype(np.float32)
    t2 = time.perf_counter()
    cos = (emb @ qs.T) / (norms[:, None] * qn[None, :] + np.float32(1e-30))
    top = np.argpartition(-cos, args.topk, axis=0)[: args.topk]
    dt2 = time.perf_counter() - t2
    del cos, top
    return {
        "value": rows_per_s / args.rows_per_gpu, "unit": "queries/s", "cores": cores, "kind": "port",
        "sample": f"{nq} queries x {m} of {args.rows_per_gpu} rows x {args.dim}-d scored by the C oracle "
                  f"(reference arithmetic, full hybrid) on {cores} threads in {dt:.2f}s; "
                  f"value = row-rate / rows per query (linear extrapolation)",
        "single_thread_value": (m / dt1) / args.rows_per_gpu,
        "optimised_variant": {"value": (m * nq / dt2) / args.rows_per_gpu, "unit": "queries/s",
                              "what": "cosine part only: fp32 BLAS GEMM with precomputed norms + argpartition (numpy), "
                                      "not the reference arithmetic, no keyword or recency term"},
    }


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist
    import __graft_entry__ as graft
    P = graft.load_package()
    syn = importlib.import_module(graft.PKG_NAME + ".synthetic")
    sharded = importlib.import_module(graft.PKG_NAME + ".sharded")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    rows, dim, B_local, k = args.rows_per_gpu, args.dim, args.batch, args.topk
    n_total = rows * world
    idx = build_shard(P, syn, torch, rank, rows, dim, n_total, dev, args.set_option)
    # ORR_BENCH_FORCE_SHARDED=1 drives the sharded front-end (device tensors in, records out, host
    # merge) even on one GPU: a rehearsal of the N>1 code path without RCCL
    use_front = world > 1 or os.environ.get("ORR_BENCH_FORCE_SHARDED") == "1"
    front = sharded.ShardedRecallSearch(idx, dim, dev) if use_front else None

    n_steps_total = args.warmup + args.steps
    # queries for every step, generated up front and resident in HBM (rank r originates queries
    # r*B_local .. of each step's global batch)
    q_steps, term_steps = [], []
    for s in range(n_steps_total):
        b0 = (s * world + rank) * B_local
        q_steps.append(syn.query_vectors(b0, B_local, dim, n_total, dev))
        # tokenised and packed into the ABI's term arrays up front, like the query vectors (host-side input preparation)
        term_steps.append(P.PackedTerms(P.pack_terms([[] if args.no_terms else P.text.query_terms(t) for t in syn.query_texts(b0, B_local, n_total)])))
    torch.cuda.synchronize()

    n_lanes = max(1, args.inflight) if front is None else 1
    lanes = [idx] + [idx.view() for _ in range(n_lanes - 1)]

    def step(s, lane=0):
        if front is not None:
            return front.search(q_steps[s], term_steps[s], syn.NOW_TICKS, k, n_total)
        return lanes[lane].search(q_steps[s], term_steps[s], syn.NOW_TICKS, k, candidate_limit=n_total)

    for s in range(args.warmup):
        step(s, s % n_lanes)
    for ln in lanes:
        ln.set_profiling(True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    results = {}
    t0 = time.perf_counter()
    if n_lanes == 1:
        for s in range(args.warmup, n_steps_total):
            results[s] = step(s)
    else:
        import threading

        def run_lane(lane):
            for s in range(args.warmup + lane, n_steps_total, n_lanes):
                results[s] = step(s, lane)

        threads = [threading.Thread(target=run_lane, args=(ln,)) for ln in range(n_lanes)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    last = results[n_steps_total - 1]

    stats = {}
    for ln in lanes:
        for name, v in ln.kernel_stats().items():
            acc = stats.setdefault(name, {"launches": 0, "total_ms": 0.0, "algo_bytes": 0.0})
            for key in acc:
                acc[key] += v[key]
        ln.set_profiling(False)

    # Reported beside the headline, not as it: the same steps again with two of them in flight (two threads,
    # the second on a view of the shard), which is how concurrent requests reach the service.
    overlap = None
    if front is None and n_lanes == 1 and not args.no_overlap_leg:
        import threading
        lane2 = None
        try:
            lane2 = idx.view()
            both = [idx, lane2]
            errors = []

            def run2(lane):
                try:
                    for s in range(args.warmup + lane, n_steps_total, 2):
                        both[lane].search(q_steps[s], term_steps[s], syn.NOW_TICKS, k, candidate_limit=n_total)
                except Exception as exc:         # e.g. no room for a second set of workspaces
                    errors.append(repr(exc))

            for _ in range(2):                   # the first round warms the view's workspaces
                threads = [threading.Thread(target=run2, args=(ln,)) for ln in range(2)]
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for t in threads:
                    t.start()
                for t in threads:
                    t.join()
                torch.cuda.synchronize()
                dt2 = time.perf_counter() - t1
            overlap = {"steps_in_flight": 2, "value": args.steps * B_local / dt2, "unit": "queries/s",
                       "ms_per_step": 1e3 * dt2 / args.steps} if not errors else {"steps_in_flight": 2, "error": errors[0]}
        except Exception as exc:
            overlap = {"steps_in_flight": 2, "error": repr(exc)}
        finally:
            if lane2 is not None:
                lane2.close()

    # sanity inside the bench: the planted row of the last step must be rank 1
    planted = syn.planted_rows(((n_steps_total - 1) * world + rank) * B_local, B_local, n_total)
    ok = [int(r) for r in last[0][:, 0]] == planted

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        okt = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        ok = bool(okt.item())

    if rank == 0:
        queries = args.steps * world * B_local
        dom = stats.get("dot_exact", {"launches": 0, "total_ms": 0.0, "algo_bytes": 0.0})
        roofline = None
        # HBM bytes per dot_exact launch from the committed PMC passes (FETCH_SIZE x2 + WRITE_SIZE,
        # corrected as MI355X_MICROARCH.md prescribes); only valid for the default workload shape.
        traffic = None
        traffic_gemv = None
        traffic_i8 = None
        pmc_path = os.path.join(ROOT, "profiles", "r01_pmc_hbm_traffic.json")
        if os.path.exists(pmc_path) and (rows, dim, B_local) == (1_000_000, 3072, 1):
            with open(pmc_path) as f:
                for kname, kv in json.load(f)["kernels"].items():
                    if kname == "orr::dot_exact_tiled<1, false, true, true>" and "hbm_bytes_per_launch_corrected" in kv:
                        traffic = kv["hbm_bytes_per_launch_corrected"]
                    if kname.startswith("orr::screen_gemv_bf16_kernel<1>") and "hbm_bytes_per_launch_corrected" in kv:
                        traffic_gemv = kv["hbm_bytes_per_launch_corrected"]
                    if kname.startswith("orr::screen_gemv_i8_kernel<1, false>") and "hbm_bytes_per_launch_corrected" in kv:
                        traffic_i8 = kv["hbm_bytes_per_launch_corrected"]
        stream_name = "screen_gemv_i8" if stats.get("screen_gemv_i8", {}).get("launches") else "screen_gemv_bf16"
        if stream_name in stats and stats[stream_name]["launches"]:
            # 1..8 queries per step: the dominant kernel streams a shadow of the rows -- int8 (N*D + 12*N bytes, 1..4
            # queries) or bf16 (2*N*D bytes); survivors are re-scored from the fp32 master
            sg = stats[stream_name]
            avg_ms = sg["total_ms"] / sg["launches"]
            bytes_per_launch = sg["algo_bytes"] / sg["launches"]
            achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
            roofline = {"bound": "hbm", "kernel": stream_name, "achieved": achieved, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic_gemv if stream_name == "screen_gemv_bf16" else traffic_i8,
                        "traffic_source": "profiles/r01_pmc_hbm_traffic.json (rocprofv3 --pmc, separate passes)",
                        "avg_launch_ms": avg_ms, "algo_bytes_per_launch": bytes_per_launch}
            if roofline["traffic"] is None:
                roofline["traffic_source"] = None
        elif dom["launches"]:
            avg_ms = dom["total_ms"] / dom["launches"]
            bytes_per_launch = dom["algo_bytes"] / dom["launches"]
            achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
            roofline = {"bound": "hbm", "kernel": "dot_exact", "achieved": achieved, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                        "traffic_source": "profiles/r01_pmc_hbm_traffic.json (rocprofv3 --pmc, separate passes)" if traffic else None,
                        "avg_launch_ms": avg_ms, "algo_bytes_per_launch": bytes_per_launch}
        elif any(stats.get(k, {}).get("launches") for k in ("screen_i8_fused", "screen_bf16_fused")):
            # batched runs (--batch > 8): the dominant kernel is the screening GEMM of the two-stage pass, 2*B*N*D
            # multiply-adds per launch against the dense MFMA peak of its type (MI355X_MICROARCH.md): int8 on the
            # int8 shadow (5 POP/s), bf16 otherwise (2.5 PFLOP/s)
            gname = "screen_i8_fused" if stats.get("screen_i8_fused", {}).get("launches") else "screen_bf16_fused"
            sc = stats[gname]
            avg_ms = sc["total_ms"] / sc["launches"]
            if world * B_local <= 128:
                # up to 128 queries fill at most half a query tile: the kernel multiplies only the live query tiles
                # and is bound by streaming the shadow once (N*D bytes of int8 rows, 2*N*D of bf16 ones)
                bytes_per_launch = sc["algo_bytes"] / sc["launches"]
                achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
                traffic_live = None
                if os.path.exists(pmc_path) and (rows, dim) == (1_000_000, 3072) and world * B_local <= 32 and gname == "screen_i8_fused":
                    with open(pmc_path) as f:     # measured with --batch 32 (one live query tile), committed with the profiles
                        traffic_live = json.load(f)["kernels"].get("orr::screen_bf16_kernel<true, 0, true, 1> (bench.py --batch 32)", {}) \
                            .get("hbm_bytes_per_launch_corrected")
                roofline = {"bound": "hbm", "kernel": gname, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic_live,
                            "traffic_source": "profiles/r01_pmc_hbm_traffic.json (rocprofv3 --pmc, separate passes)" if traffic_live else None,
                            "avg_launch_ms": avg_ms, "algo_bytes_per_launch": bytes_per_launch}
            else:
                flops = 2.0 * (world * B_local) * rows * dim
                achieved = flops / (avg_ms * 1e-3) / 1e12
                peak = MFMA_I8_PEAK_TOPS if gname == "screen_i8_fused" else MFMA_BF16_PEAK_TFLOPS
                roofline = {"bound": "mfma", "kernel": gname, "achieved": achieved, "peak": peak,
                            "unit": "TOP/s" if gname == "screen_i8_fused" else "TFLOP/s", "frac": achieved / peak, "traffic": None,
                            "avg_launch_ms": avg_ms, "algo_flops_per_launch": flops}
        out = {
            "metric": "queries/sec at top-k=10 over N x 3072-d chunks",
            "value": queries / elapsed, "unit": "queries/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32 products summed in f64 (reference arithmetic)",
            "data": "synthetic",
            "config": {"workload": f"C2: {rows} chunks x {dim}-d fp32 per GPU, {B_local} query per GPU per step, "
                                   f"top-k={k}, full hybrid (cosine+keyword+recency), candidate_limit=corpus",
                       "corpus_rows": n_total, "queries_per_step": world * B_local, "options": args.set_option,
                       "steps_in_flight": n_lanes,
                       "parallelism": f"row-sharded x{world}, all-gather of per-shard top-k'" if world > 1 else "single GPU"},
            "row_scores_per_sec": queries * n_total / elapsed,
            "two_steps_in_flight": overlap,
            "rank1_is_planted_row": ok,
            "roofline": roofline,
            "kernels": {n: {"launches": v["launches"], "avg_ms": v["total_ms"] / max(1, v["launches"])}
                        for n, v in stats.items()},
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(P, syn, args, n_total, torch, dev)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    for ln in lanes[1:]:
        ln.close()
    idx.close()


if __name__ == "__main__":
    main()
