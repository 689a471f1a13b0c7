"""The reference's admission rule at the k-th boundary, and the closed form the GPU uses for it (CPU).

HeapResultHandler::add_result (thirdparty/faiss/faiss/impl/ResultHandler.h:258-279) admits a candidate only if it strictly
improves on the heap's top; heap_replace_top (utils/Heap.h:113-151) orders equal distances by id (cmp2,
utils/ordered_key_value.h:51, 74), so the top among ties is the largest id for L2 (CMax) and the smallest for the inner
product (CMin); heap_reorder sorts the survivors.  knowhere_amd/csrc/knhip_api.hip::search_batch_ties and
refine.hip use the equivalent closed form -- with v the final k-th distance,

    a candidate tied with v is eligible iff it is among the first k arrivals with distance <= v (>= v for IP);
    result = canonical top-k of {candidates better than v} U {eligible ties}

-- which needs the arrival order only for the (rare) queries whose canonical (k + 1)-th result ties with the k-th.  Here
the heap is replayed step by step (a literal transcription of heap_replace_top) against the closed form on arrival
sequences dense with ties, and against oracle.c's heap (the restatement pinned to the reference)."""
import numpy as np
import pytest


def heap_search(dis, ids, k, is_l2):
    """faiss: heap_heapify (neutral values, id -1), add_result with strict admission, heap_replace_top with cmp2, reorder"""
    neutral = np.float32(np.finfo(np.float32).max) if is_l2 else -np.float32(np.finfo(np.float32).max)
    val = [neutral] * (k + 1)  # 1-based
    idx = [-1] * (k + 1)

    def cmp(a, b):  # C::cmp(a, b): CMax a > b, CMin a < b
        return a > b if is_l2 else a < b

    def cmp2(a1, b1, a2, b2):
        return (a1 > b1 or (a1 == b1 and a2 > b2)) if is_l2 else (a1 < b1 or (a1 == b1 and a2 < b2))

    for d, i in zip(dis, ids):
        if not cmp(val[1], d):
            continue
        p = 1
        while True:
            i1, i2 = 2 * p, 2 * p + 1
            if i1 > k:
                break
            if i2 == k + 1 or cmp2(val[i1], val[i2], idx[i1], idx[i2]):
                if cmp2(d, val[i1], i, idx[i1]):
                    break
                val[p], idx[p] = val[i1], idx[i1]
                p = i1
            else:
                if cmp2(d, val[i2], i, idx[i2]):
                    break
                val[p], idx[p] = val[i2], idx[i2]
                p = i2
        val[p], idx[p] = d, i
    got = [(val[j], idx[j]) for j in range(1, k + 1) if idx[j] >= 0]
    got.sort(key=lambda t: (t[0], t[1]) if is_l2 else (-t[0], -t[1]))  # heap_reorder: cmp2 order, best first
    return got


def closed_form(dis, ids, k, is_l2):
    order = sorted(range(len(dis)), key=lambda j: (dis[j], ids[j]) if is_l2 else (-dis[j], -ids[j]))
    if len(order) <= k:
        return [(dis[j], ids[j]) for j in order]
    v = dis[order[k - 1]]
    better = [(dis[j], ids[j]) for j in range(len(dis)) if (dis[j] < v if is_l2 else dis[j] > v)]
    arrivals = [j for j in range(len(dis)) if (dis[j] <= v if is_l2 else dis[j] >= v)]
    eligible = [(dis[j], ids[j]) for j in arrivals[:k] if dis[j] == v]
    pool = better + eligible
    pool.sort(key=lambda t: (t[0], t[1]) if is_l2 else (-t[0], -t[1]))
    return pool[:k]


@pytest.mark.parametrize("is_l2", [True, False], ids=["l2", "ip"])
def test_closed_form_equals_the_heap(is_l2):
    rng = np.random.default_rng(5 if is_l2 else 6)
    checked_ambiguous = 0
    for trial in range(4000):
        n = int(rng.integers(1, 60))
        k = int(rng.integers(1, 12))
        levels = int(rng.integers(1, 6))  # few distinct distances: ties everywhere
        dis = [np.float32(x) for x in rng.integers(0, levels, n)]
        ids = [int(x) for x in rng.permutation(200)[:n]]
        a, b = heap_search(dis, ids, k, is_l2), closed_form(dis, ids, k, is_l2)
        assert a == b, (trial, k, list(zip(dis, ids)), a, b)
        canon = sorted(zip(dis, ids), key=lambda t: (t[0], t[1]) if is_l2 else (-t[0], -t[1]))[:k]
        checked_ambiguous += canon != a
    assert checked_ambiguous > 300  # the canonical answer really differs often on such data: the rule is exercised


@pytest.mark.parametrize("is_l2", [True, False], ids=["l2", "ip"])
def test_canonical_k_plus_one_detects_every_ambiguous_query(is_l2):
    """the GPU only resolves a query whose canonical (k + 1)-th result ties with its k-th: whenever the heap's answer differs
    from the canonical top-k that tie is there"""
    rng = np.random.default_rng(9 if is_l2 else 10)
    for trial in range(3000):
        n = int(rng.integers(2, 50))
        k = int(rng.integers(1, 10))
        dis = [np.float32(x) for x in rng.integers(0, 4, n)]
        ids = [int(x) for x in rng.permutation(100)[:n]]
        canon = sorted(zip(dis, ids), key=lambda t: (t[0], t[1]) if is_l2 else (-t[0], -t[1]))
        if heap_search(dis, ids, k, is_l2) != canon[:k]:
            assert len(canon) > k and canon[k][0] == canon[k - 1][0]


@pytest.mark.parametrize("metric", [0, 1], ids=["l2", "ip"])
def test_oracle_flat_search_follows_the_closed_form(port, metric):
    """oracle.c (pinned to the reference) on rows full of duplicates: its heap gives the closed form's answer"""
    from oracle import binding as ob
    rng = np.random.default_rng(21 + metric)
    d, nb, nq, k = 8, 300, 40, 7
    proto = rng.integers(0, 3, (12, d)).astype(np.float32)   # 12 distinct rows, each many times
    xb = proto[rng.integers(0, 12, nb)]
    xq = proto[rng.integers(0, 12, nq)] + rng.integers(0, 2, (nq, d)).astype(np.float32)
    ix = ob.IndexData(ob.FLAT, metric, d)
    ix.base = xb
    D, I = port.search(ix, xq, k, 1)
    for q in range(nq):
        dis = [np.float32(((xq[q] - xb[j]) ** 2).sum() if metric == 0 else (xq[q] * xb[j]).sum()) for j in range(nb)]
        want = closed_form(dis, list(range(nb)), k, metric == 0)
        assert [int(i) for i in I[q]] == [i for _, i in want], q
