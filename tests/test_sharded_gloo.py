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


def test_term_slot_roundtrip():
    import importlib
    import __graft_entry__ as graft
    graft.load_package()
    sh = importlib.import_module(graft.PKG_NAME + ".sharded")
    for terms in ([], [b"a"], [b"kubernetes", "naïve".encode(), b"x" * 100]):
        assert sh._unpack_terms_fixed(sh._pack_terms_fixed(terms)) == terms
    with pytest.raises(ValueError):
        sh._pack_terms_fixed([b"y" * 200, b"z" * 100])


def test_term_slots_round_trip_for_a_whole_batch():
    """The exchange slots are built from and parsed back into the ABI's packed arrays with numpy."""
    import importlib
    import numpy as np
    from helpers import pkg
    P = pkg()
    sh = importlib.import_module("omni_recall_rag_amd.sharded")
    batch = [[b"alpha", b"beta"], [], [b"x" * 254], [b"\xc3\xa9t\xc3\xa9", b"k8s", b"a"], [b"q"] * 100]
    packed = P.pack_terms(batch)
    slots = sh._slots_from_packed(*packed)
    assert slots.shape == (len(batch), sh.TERM_SLOT) and slots.dtype == np.uint8
    back = sh._packed_from_slots(slots)
    assert len(back) == len(batch)
    assert all(np.array_equal(a, b) for a, b in zip(packed, back.arrays))
    assert [back[b] for b in range(len(batch))] == batch and list(back) == batch
    assert all(np.array_equal(a, b) for a, b in zip(P.pack_terms(back), packed))      # PackedTerms passes through
    # a few queries take plain-Python paths, more the vectorised ones: every size round-trips to the same arrays
    rng = np.random.default_rng(3)
    for size in (1, 2, 4, 5, 16, 17, 40):
        many = [[bytes(rng.integers(97, 123, int(rng.integers(1, 12))).astype(np.uint8)) for _ in range(int(rng.integers(0, 6)))]
                for _ in range(size)]
        pk = P.pack_terms(many)
        sl = sh._slots_from_packed(*pk)
        bk = sh._packed_from_slots(sl)
        assert list(bk) == many and all(np.array_equal(a, b) for a, b in zip(pk, bk.arrays)), size
        assert np.array_equal(sl, np.stack([sh._slots_from_packed(*P.pack_terms([q]))[0] for q in many])), size
    for bad in ([[b"y" * 256]], [[b"y" * 200, b"z" * 100]], [[b"a"], [b"y" * 256], [], [], []], [[b"q"] * 256]):
        with pytest.raises(ValueError):
            sh._slots_from_packed(*P.pack_terms(bad))
