"""N>1 path on CPU: two gloo ranks, each owning half of the candidate order; query
exchange, one all-gather of per-shard candidate records, host merge with k'
escalation.  Results must equal the oracle over the whole corpus."""
import os
import socket
import subprocess
import sys

import pytest

from helpers import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_world(tmp_path, world, n, dim, extra=()):
    port = _free_port()
    res = str(tmp_path / "res")
    env = dict(os.environ, OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), str(r), str(world),
                               str(port), str(n), str(dim), "77", res, *extra], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(world)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out.decode(errors="replace"))
    for r, p in enumerate(procs):
        assert p.returncode == 0, outs[r][-3000:]
        assert open(res + ".%d" % r).read() == "ok", outs[r][-3000:]


@pytest.mark.parametrize("world", [2, 3])
def test_row_sharded_search_over_gloo(tmp_path, world):
    _run_world(tmp_path, world, 240, 16)


@pytest.mark.gpu
def test_row_sharded_search_real_shards_on_one_gpu(tmp_path):
    """Three ranks, each with a REAL HIP shard on cuda:0 (gloo carries the exchange here; the
    bench uses RCCL): identical to the oracle over the whole corpus, including k' escalation."""
    _run_world(tmp_path, 3, 3000, 128, extra=("gpu",))


@pytest.mark.gpu
def test_row_sharded_search_over_rccl_with_the_ranks_this_box_has(tmp_path):
    """The product backend itself: torch.distributed "nccl" (= RCCL), one rank per visible GPU (one on this pool's boxes),
    queries and the records' all-gather on device buffers -- the branch the gloo rehearsals cannot take.  Results equal the
    oracle's, the escalation of uncertified queries runs, every rank answers the all-gather."""
    import torch
    world = max(1, min(torch.cuda.device_count(), 4))
    _run_world(tmp_path, world, 3000, 128, extra=("nccl",))


def test_term_sections_round_trip():
    """The query exchange carries the ABI's packed term arrays length-prefixed: no per-query or per-term size limit."""
    import importlib
    import numpy as np
    from helpers import pkg
    P = pkg()
    sh = importlib.import_module("omni_recall_rag_amd.sharded")
    batches = [[[b"alpha", b"beta"], [], [b"x" * 254], [b"\xc3\xa9t\xc3\xa9", b"k8s", b"a"], [b"q"] * 100],
               [[b"y" * 300, b"z" * 1000]], [[]], [[b"t%d" % i for i in range(400)], [b"w" * 70000]]]
    parts = []
    for batch in batches:
        packed = P.pack_terms(batch)
        sec = sh._term_section(packed)
        back = sh._parse_term_section(sec, len(batch))
        one = sh._concat_packed([back])
        assert list(one) == batch
        assert np.array_equal(one.arrays[1], packed[1]) and np.array_equal(one.arrays[2], packed[2])
        parts.append(back)
    allq = sh._concat_packed(parts)                       # several ranks' sections -> the whole batch, in rank order
    assert list(allq) == [q for batch in batches for q in batch]
    # a section cut out of the middle of a larger packed batch (qoff[0] > 0) is rebased
    whole = P.pack_terms(batches[0])
    sub = (whole[0], whole[1], whole[2][2:5])
    assert list(sh._concat_packed([sh._parse_term_section(sh._term_section(sub), 2)])) == batches[0][2:4]


def test_long_queries_over_gloo(tmp_path):
    """A query with more than 255 bytes of terms (the round-1 exchange refused it) and one with 300 terms go through
    the second, variable-length collective and still equal the oracle."""
    _run_world(tmp_path, 2, 120, 8, extra=("cpu", "long"))
