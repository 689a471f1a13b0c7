"""knhip_search / knhip_range_search are documented as thread-safe for concurrent calls on one index
(include/knhip.h; the IndexNode's Search is const and is called from a thread pool,
reference include/knowhere/index/index_factory.h:157-165): hammer one index from several host threads
and require every result to be bit-identical to the single-threaded one."""
import threading

import numpy as np
import pytest

from conftest import gen_data
from helpers import finish_ivfpq
from oracle import binding as ob

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind,M", [(ob.IVF_PQ, 32), (ob.IVF_FLAT, 0)], ids=["ivfpq32", "ivfflat"])
def test_concurrent_searches_on_one_index(port, kind, M):
    from knowhere_amd import GpuIndex
    nb, d, nlist = 30000, 64, 64
    xb = gen_data(nb, d, 42)
    ix = finish_ivfpq(port, ob.make_index(port, kind, ob.L2, xb, nlist=nlist, M=max(M, 1), nbits=8))
    g = GpuIndex.from_data(ix, device=0)
    nthreads, rounds = 6, 5
    queries = [gen_data(50 + 7 * t, d, 100 + t) for t in range(nthreads)]
    ks = [10, 100, 1, 37, 64, 128]
    expect = [g.search(queries[t], ks[t], 16) for t in range(nthreads)]
    D0, _ = expect[0]
    radius = float(np.median(D0[:, -1]))
    expect_r = g.range_search(queries[0], radius, 2)
    errors = []

    def work(t):
        try:
            for _ in range(rounds):
                D, I = g.search(queries[t], ks[t], 16)
                if not (np.array_equal(D.view(np.uint32), expect[t][0].view(np.uint32)) and np.array_equal(I, expect[t][1])):
                    errors.append(f"thread {t}: top-k result changed under concurrency")
                if t == 0:
                    r = g.range_search(queries[0], radius, 2)
                    if not all(np.array_equal(a, b) for a, b in zip(r, expect_r)):
                        errors.append("range result changed under concurrency")
        except Exception as e:  # noqa: BLE001
            errors.append(f"thread {t}: {e!r}")

    th = [threading.Thread(target=work, args=(t,)) for t in range(nthreads)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errors, errors[:3]


def _free_hbm():
    import ctypes
    from knowhere_amd._lib import load
    free, total = ctypes.c_int64(0), ctypes.c_int64(0)
    assert load().knhip_device_memory(0, ctypes.byref(free), ctypes.byref(total)) == 0
    return free.value


def test_scratch_of_concurrent_searches_is_pooled_and_bounded(port):
    """Every concurrent host-boundary search takes its own scratch (include/knhip.h: knhip_search); the scratch goes back to
    the index's pool when the call ends.  The pool therefore never holds more than (largest number of calls in flight)
    scratches: a second wave of the same calls must run entirely on what the first wave left -- free HBM does not drop again
    -- and the whole pool stays within threads x (a generous per-call bound for these shapes)."""
    from knowhere_amd import GpuIndex
    nb, d, nlist = 30000, 64, 64
    xb = gen_data(nb, d, 43)
    ix = finish_ivfpq(port, ob.make_index(port, ob.IVF_PQ, ob.L2, xb, nlist=nlist, M=32, nbits=8))
    g = GpuIndex.from_data(ix, device=0)
    nthreads = 8
    queries = [gen_data(200, d, 300 + t) for t in range(nthreads)]
    g.search(queries[0], 10, 16)  # (lazy layouts of the index itself are built by the first call)
    free0 = _free_hbm()

    def wave():
        bar = threading.Barrier(nthreads)
        errs = []

        def work(t):
            try:
                bar.wait()
                for _ in range(3):
                    g.search(queries[t], 10, 16)
            except Exception as e:  # noqa: BLE001
                errs.append(repr(e))

        th = [threading.Thread(target=work, args=(t,)) for t in range(nthreads)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        assert not errs, errs[:3]

    wave()
    free1 = _free_hbm()
    wave()
    wave()
    free2 = _free_hbm()
    slack = 64 << 20  # (the allocator's own granularity; other processes do not share the box)
    assert free2 >= free1 - slack, f"scratch grew between identical waves: {free1 - free2} bytes"
    assert free0 - free1 <= nthreads * (256 << 20), f"{nthreads} scratches hold {(free0 - free1) >> 20} MiB"
