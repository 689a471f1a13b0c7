import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a GPU selection: torch (which carries its own HIP runtime) must meet the device before libknhip.so's runtime does, or
    # torch.cuda reports no device for the rest of the process -- whatever subset of the test files was asked for
    expr = config.getoption("markexpr", "") or ""
    if "gpu" in expr and "not gpu" not in expr:
        try:
            import torch
            torch.cuda.is_available()
        except ImportError:
            pass


@pytest.fixture(scope="session")
def port():
    from oracle import binding as ob
    return ob.Port()


@pytest.fixture(scope="session")
def ref():
    from oracle import binding as ob
    if not ob.Ref.available():
        pytest.skip("oracle/_ref/libknowhere_ref.so not built (needs /root/reference)")
    return ob.Ref()


@pytest.fixture(scope="session")
def kref():
    from oracle import binding as ob
    if not ob.KRef.available():
        pytest.skip("oracle/_ref/libknowhere_kref.so not built (needs /root/reference)")
    return ob.KRef()


def gen_data(n, d, seed, lo=0.0, hi=100.0):
    """reference fixture: tests/ut/utils.h:41-50 GenDataSet = uniform_real(0, 100), seeded"""
    r = np.random.default_rng(seed)
    return (r.random((n, d), dtype=np.float32) * (hi - lo) + lo).astype(np.float32)


# how often the one licensed deviation (ties at the k-th distance: first-scanned vs canonical-first) was actually used
PARITY_STATS = {"comparisons": 0, "ids_compared": 0, "licensed_id_mismatches": 0, "comparisons_with_licensed_mismatches": 0}


def pytest_terminal_summary(terminalreporter):
    if PARITY_STATS["comparisons"]:
        terminalreporter.write_line(
            f"assert_parity: {PARITY_STATS['comparisons']} comparisons, {PARITY_STATS['ids_compared']} ids; licensed k-th-"
            f"boundary tie mismatches: {PARITY_STATS['licensed_id_mismatches']} ids in "
            f"{PARITY_STATS['comparisons_with_licensed_mismatches']} comparisons")


def assert_parity(Do, Io, Dg, Ig, metric, what="", licensed_ties=False):
    """Parity bar (BASELINE.json north_star): distances bit-equal (tolerance 0 -- far inside the
    1e-4 relative bound), ids EQUAL -- including which of several candidates tied at the k-th distance
    is returned: the library applies the reference's first-come admission rule there (knhip_api.hip,
    search_batch_ties; refine.hip).  licensed_ties=True admits, and counts, a difference confined to entries
    whose distance equals the query's k-th distance bit for bit; only the cases the library documents as not
    covered may pass it: results merged from several shards, brute force with k >= 100 (the reference's
    reservoir), k = 1024."""
    Do, Dg = np.asarray(Do, np.float32), np.asarray(Dg, np.float32)
    Io, Ig = np.asarray(Io, np.int64), np.asarray(Ig, np.int64)
    assert Do.shape == Dg.shape and Io.shape == Ig.shape, what
    db = Do.view(np.uint32) != Dg.view(np.uint32)
    assert not db.any(), (f"{what}: {db.sum()} distances differ bitwise; first at {np.argwhere(db)[0]}: "
                          f"oracle {Do[db][0]!r} gpu {Dg[db][0]!r}")
    bad = Io != Ig
    PARITY_STATS["comparisons"] += 1
    PARITY_STATS["ids_compared"] += int(Io.size)
    if bad.any():
        assert licensed_ties, (f"{what}: {int(bad.sum())} ids differ (first at {np.argwhere(bad)[0]}: oracle "
                               f"{Io[bad][0]} gpu {Ig[bad][0]}); ties at the k-th boundary are not licensed here")
        kth = Do[:, -1:]
        licensed = bad & (Do == kth)
        PARITY_STATS["licensed_id_mismatches"] += int(licensed.sum())
        PARITY_STATS["comparisons_with_licensed_mismatches"] += 1
        # within a run of equal distances the same id multiset must appear unless it touches the k-th
        assert (bad == licensed).all(), (f"{what}: {int((bad & ~licensed).sum())} id mismatches that are not "
                                         f"k-th-boundary ties; first at {np.argwhere(bad & ~licensed)[0]}")


def recall(I_true, I, k=None):
    k = k or I.shape[1]
    hit = 0
    for a, b in zip(I_true[:, :k], I[:, :k]):
        hit += len(set(a.tolist()) & set(b.tolist()))
    return hit / (I.shape[0] * k)
