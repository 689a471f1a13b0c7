"""Range search: knhip_range_search against the oracle's restatement of
IndexIVF::range_search_preassigned / IndexFlat::range_search (itself pinned to the reference's FAISS in
tests/test_oracle.py), through the C ABI.  Bar: lims equal, ids equal IN THE REFERENCE'S EMISSION ORDER (lists
in coarse order, storage order inside a list), distances bit-equal -- for every early-stop setting, with and
without a bitset."""
import numpy as np
import pytest

from conftest import gen_data
from helpers import finish_ivfpq
from oracle import binding as ob

pytestmark = pytest.mark.gpu


def _gpu(ix):
    from knowhere_amd import GpuIndex
    return GpuIndex.from_data(ix, device=0)


def _bitset(n, frac, seed):
    filt = np.random.default_rng(seed).random(n) < frac
    bs = np.zeros((n + 7) // 8, np.uint8)
    for i in np.nonzero(filt)[0]:
        bs[i >> 3] |= 1 << (i & 7)
    return bs


def _same(a, b, what):
    assert np.array_equal(a[0], b[0]), f"{what}: lims differ"
    assert np.array_equal(a[1], b[1]), f"{what}: ids differ (or are in a different order)"
    assert np.array_equal(a[2].view(np.uint32), b[2].view(np.uint32)), f"{what}: distances differ bitwise"


@pytest.mark.parametrize("path", [p for p in __import__("helpers").golden_files() if "ivfpq" not in p or "ivfpq32" in p],
                         ids=lambda p: __import__("os").path.basename(p)[:-4])
def test_golden_range(path):
    """the committed range results of the reference's own FAISS (tests/golden/make_golden.py); the small golden IVF-PQ
    index has m = 8, outside the range path -- the d = 128, m = 32 fixtures (h128_ivfpq32_*) are inside it"""
    from helpers import load_golden, load_golden_range
    ix, xq, _ = load_golden(path)
    g = _gpu(ix)
    radius, cases = load_golden_range(path)
    for c in cases:
        got = g.range_search(xq, np.float32(radius), c["max_empty"], c["bitset"], c["nbits"])
        _same((c["lims"], c["ids"], c["dis"]), got, f"golden range max_empty={c['max_empty']}")


def _radii(port, ix, xq, metric, nprobe):
    """radii that give empty, sparse and dense results"""
    D, _ = port.search(ix, xq, 40, nprobe)
    col = np.sort(D[:, [0, 5, 39]].reshape(-1))
    if metric == ob.L2:
        return [float(col[0]) * 0.5, float(np.median(D[:, 5])), float(np.median(D[:, 39]))]
    return [float(col[-1]) * 2.0 + 1.0, float(np.median(D[:, 5])), float(np.median(D[:, 39]))]


@pytest.mark.parametrize("metric", [ob.L2, ob.IP], ids=["l2", "ip"])
def test_brute_force_range(port, metric):
    nb, nq, d = 20000, 33, 24  # > 2 segments of 8192 rows, ragged last one
    xb, xq = gen_data(nb, d, 42), gen_data(nq, d, 44)
    ix = ob.IndexData(ob.FLAT, metric, d)
    ix.base = xb
    g = _gpu(ix)
    bs = _bitset(nb, 0.4, 7)
    for radius in _radii(port, ix, xq, metric, 1):
        for bitset in (None, bs):
            exp = port.range_search(ix, xq, radius, 0, bitset, nb if bitset is not None else 0)
            got = g.range_search(xq, radius, 0, bitset, nb if bitset is not None else 0)
            _same(exp, got, f"flat radius={radius}")


@pytest.mark.parametrize("kind,M,d", [(ob.IVF_FLAT, 0, 20), (ob.IVF_SQ8, 0, 40), (ob.IVF_PQ, 32, 64),
                                      (ob.IVF_PQ, 32, 128)],
                         ids=["ivfflat", "ivfsq8", "ivfpq32_d64", "ivfpq32_d128"])
@pytest.mark.parametrize("metric", [ob.L2, ob.IP], ids=["l2", "ip"])
def test_ivf_range(port, kind, M, d, metric):
    nb, nq, nlist = 12000, 40, 48
    xb, xq = gen_data(nb, d, 42), gen_data(nq, d, 44)
    ix = finish_ivfpq(port, ob.make_index(port, kind, metric, xb, nlist=nlist, M=max(M, 1), nbits=8))
    g = _gpu(ix)
    bs = _bitset(nb, 0.4, 7)
    total = 0
    for radius in _radii(port, ix, xq, metric, nlist):
        for max_empty in (0, 1, 2, 5):
            for bitset in (None, bs):
                exp = port.range_search(ix, xq, radius, max_empty, bitset, nb if bitset is not None else 0)
                got = g.range_search(xq, radius, max_empty, bitset, nb if bitset is not None else 0)
                _same(exp, got, f"kind={kind} radius={radius} max_empty={max_empty} bitset={bitset is not None}")
                total += int(exp[0][-1])
    assert total > 0


def test_ivf_range_early_stop_changes_the_result(port):
    """the heuristic is real: stopping after one empty list must lose hits that scanning every list finds"""
    nb, nq, d, nlist = 12000, 40, 20, 48
    xb, xq = gen_data(nb, d, 42), gen_data(nq, d, 44)
    ix = ob.make_index(port, ob.IVF_FLAT, ob.L2, xb, nlist=nlist)
    g = _gpu(ix)
    D, _ = port.search(ix, xq, 40, nlist)
    radius = float(np.median(D[:, 39]))
    full = g.range_search(xq, radius, 0)
    one = g.range_search(xq, radius, 1)
    assert one[0][-1] < full[0][-1]
    for q in range(nq):
        assert set(one[1][one[0][q]:one[0][q + 1]].tolist()) <= set(full[1][full[0][q]:full[0][q + 1]].tolist())


@pytest.mark.parametrize("kind,M,d", [(ob.IVF_FLAT, 0, 24), (ob.IVF_SQ8, 0, 40), (ob.IVF_PQ, 32, 128), (ob.IVF_PQ, 8, 32)],
                         ids=["ivfflat", "ivfsq8", "ivfpq32", "ivfpq8"])
@pytest.mark.parametrize("metric", [ob.L2, ob.IP], ids=["l2", "ip"])
def test_ivf_range_rank_waves(port, kind, M, d, metric, monkeypatch):
    """nlist > 128: the probes go in waves of coarse ranks (64, 128, ...) and the scan stops once every query has met
    the early stop.  Same lims / ids / order / bits as the oracle (which walks all nlist ranks), the same as the
    one-wave path, and -- for a small max_empty -- far fewer ranks scanned than nlist."""
    nb, nq, nlist = 30000, 37, 600
    xb, xq = gen_data(nb, d, 42), gen_data(nq, d, 44)
    ix = finish_ivfpq(port, ob.make_index(port, kind, metric, xb, nlist=nlist, M=max(M, 1), nbits=8))
    g = _gpu(ix)
    bs = _bitset(nb, 0.4, 7)
    radii = _radii(port, ix, xq, metric, nlist)
    total = 0
    for radius in radii[1:]:
        for max_empty in (1, 2, 40, 0):
            for bitset in (None, bs):
                nbits = nb if bitset is not None else 0
                exp = port.range_search(ix, xq, radius, max_empty, bitset, nbits)
                got = g.range_search(xq, radius, max_empty, bitset, nbits)
                ranks = g.last_range_ranks()
                _same(exp, got, f"kind={kind} radius={radius} max_empty={max_empty} bitset={bitset is not None}")
                total += int(exp[0][-1])
                if max_empty == 0:
                    assert ranks == nlist  # (no early stop: one pass over every list)
                elif max_empty <= 2:
                    assert ranks < nlist, f"max_empty={max_empty}: all {ranks} ranks were scanned"
                monkeypatch.setenv("KNHIP_RANGE_NO_WAVES", "1")
                one = g.range_search(xq, radius, max_empty, bitset, nbits)
                monkeypatch.delenv("KNHIP_RANGE_NO_WAVES")
                assert g.last_range_ranks() == nlist or max_empty == 0
                _same(one, got, "waves vs one pass")
    assert total > 0


def test_range_unsupported_and_edge_cases(port):
    nb, d = 3000, 16
    xb, xq = gen_data(nb, d, 1), gen_data(5, d, 2)
    # (IVF_PQ with m != 32 was refused until the plain ADC dump kernel of range.hip had run on hardware: round 3)
    pq8 = finish_ivfpq(port, ob.make_index(port, ob.IVF_PQ, ob.L2, xb, nlist=8, M=8))
    D8, _ = port.search(pq8, xq, 20, 8)
    r8 = float(np.median(D8[:, 10]))
    exp = port.range_search(pq8, xq, r8, 2)
    got = _gpu(pq8).range_search(xq, np.float32(r8), 2)
    assert np.array_equal(exp[0], got[0]) and np.array_equal(exp[1], got[1])
    assert np.array_equal(exp[2].view(np.uint32), got[2].view(np.uint32))
    fl = ob.make_index(port, ob.IVF_FLAT, ob.L2, xb, nlist=8)
    g = _gpu(fl)
    lims, ids, dis = g.range_search(xq[:0], 1.0)  # no queries
    assert lims.tolist() == [0] and ids.size == 0
    lims, ids, dis = g.range_search(xq, -1.0)  # nothing is inside a negative L2 radius
    assert lims.tolist() == [0] * 6 and ids.size == 0
    lims, ids, dis = g.range_search(xq, 3.0e38, 0)  # everything is
    assert lims[-1] == 5 * nb


@pytest.mark.parametrize("kind,M,d", [(ob.IVF_FLAT, 0, 32), (ob.IVF_SQ8, 0, 32), (ob.IVF_PQ, 32, 128), (ob.IVF_PQ, 12, 48)],
                         ids=["ivfflat", "ivfsq8", "ivfpq32", "ivfpq12"])
@pytest.mark.parametrize("metric", [ob.L2, ob.IP], ids=["l2", "ip"])
def test_ranked_range_search_counts_rebuild_every_early_stop(port, kind, M, d, metric):
    """knhip_range_search_ranked (what a list-sharded deployment merges by): every list visited, hits per coarse rank.
    The counts add up to the per-query totals, and walking them with the reference's rule reproduces the oracle's result
    for every max_empty_result_buckets -- here with the lists dealt to two "shards" whose results are merged the way
    the node does (knowhere_amd/host/hip_index_node.cc, RangeSearch)"""
    from knowhere_amd import GpuIndex
    nb, nq, nlist = 6000, 24, 48
    xb, xq = gen_data(nb, d, 42), gen_data(nq, d, 44)
    ix = ob.make_index(port, kind, metric, xb, nlist=nlist, M=M or 8)
    if kind == ob.IVF_PQ:
        finish_ivfpq(port, ix)
    g = GpuIndex.from_data(ix, device=0)
    D40, _ = port.search(ix, xq, 40, nlist)
    radius = float(np.median(D40[:, 12]))
    lims, ids, dis, cnt = g.range_search_ranked(xq, np.float32(radius))
    assert cnt.shape == (nq, nlist) and np.array_equal(cnt.sum(1), np.diff(lims))
    full = g.range_search(xq, np.float32(radius), 0)
    assert np.array_equal(full[0], lims) and np.array_equal(full[1], ids)
    # two shards: the same centroids, the lists dealt alternately
    import copy
    parts = []
    for s in range(2):
        sub = copy.copy(ix)
        sub.list_codes = [c if l % 2 == s else c[:0] for l, c in enumerate(ix.list_codes)]
        sub.list_ids = [i if l % 2 == s else i[:0] for l, i in enumerate(ix.list_ids)]
        gs = GpuIndex.from_data(sub, device=0)
        parts.append(gs.range_search_ranked(xq, np.float32(radius)))
        gs.close()
    assert np.array_equal(parts[0][3] + parts[1][3], cnt), "every shard ranks the lists alike"
    for max_empty in (0, 1, 2, 5):
        exp = port.range_search(ix, xq, radius, max_empty)
        got_i, got_d = [], []
        for q in range(nq):
            ptr = [p[0][q] for p in parts]
            nempty = 0
            for r in range(nlist):
                hits = 0
                for s, p in enumerate(parts):
                    c = int(p[3][q, r])
                    got_i.append(p[1][ptr[s]:ptr[s] + c])
                    got_d.append(p[2][ptr[s]:ptr[s] + c])
                    ptr[s] += c
                    hits += c
                if max_empty > 0:
                    nempty = nempty + 1 if hits == 0 else 0
                    if nempty >= max_empty:
                        break
        ii = np.concatenate(got_i) if got_i else np.empty(0, np.int64)
        dd = np.concatenate(got_d) if got_d else np.empty(0, np.float32)
        assert np.array_equal(ii, exp[1]), (max_empty,)
        assert np.array_equal(dd.view(np.uint32), exp[2].view(np.uint32)), (max_empty,)
    g.close()


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("KNHIP_FUZZ_SEEDS", "10"))))
def test_range_search_equals_the_oracle_on_random_shapes(port, seed):
    """random kind (brute force, IVF-Flat, IVF-SQ8, IVF-PQ with m = 32 / 8 / 12 and 8 / 5-bit codes), metric, nlist above and
    below the rank-wave limit of 128, early stop 0 .. 3, filter, radius taken from a k-th distance: lims, ids in the
    reference's emission order and distance bits equal to the oracle's"""
    r = np.random.default_rng(6000 + seed)
    kind, M, d, nbits = [(ob.FLAT, 0, 40, 8), (ob.IVF_FLAT, 0, 40, 8), (ob.IVF_SQ8, 0, 72, 8), (ob.IVF_PQ, 32, 128, 8),
                         (ob.IVF_PQ, 8, 64, 8), (ob.IVF_PQ, 12, 48, 5)][int(r.integers(0, 6))]
    metric = int(r.integers(0, 2))
    nb = int(r.choice([1500, 8000]))
    nlist = int(r.choice([6, 40, 150]))
    xb = gen_data(nb, d, seed, -2.0, 2.0)
    ix = ob.make_index(port, kind, metric, xb, nlist=nlist, M=max(M, 1), nbits=nbits, seed=seed) if kind != ob.FLAT else \
        ob.make_index(port, ob.FLAT, metric, xb)
    ix = finish_ivfpq(port, ix)
    g = _gpu(ix)
    for case in range(3):
        nq = int(r.choice([1, 9, 60]))
        xq = gen_data(nq, d, 50 + case, -2.0, 2.0)
        kk = int(r.choice([1, 20, 200]))
        np_ = max(1, nlist // 3)
        Dk, _ = port.search(ix, xq, kk, np_) if kind != ob.FLAT else port.search(ix, xq, kk)
        v = Dk[:, -1]
        v = v[np.isfinite(v) & (np.abs(v) < 1e30)]
        if v.size == 0:
            continue
        radius = np.float32(np.median(v))
        max_empty = int(r.integers(0, 4))
        frac = float(r.choice([0.0, 0.5, 0.97]))
        bs = _bitset(nb, frac, seed + case) if frac > 0 else None
        exp = port.range_search(ix, xq, radius, max_empty, bs, nb if bs is not None else 0)
        got = g.range_search(xq, radius, max_empty, bs, nb if bs is not None else 0)
        _same(exp, got, f"seed={seed} kind={kind} m={M} nbits={nbits} metric={metric} nb={nb} nlist={nlist} nq={nq} "
                        f"radius={radius} max_empty={max_empty} filter={frac}")
    g.close()
