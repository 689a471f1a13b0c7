"""GPU tests (-m gpu): IVF-PQ with ANY number of 8-bit sub-quantizers (pq_scan_any.hip).

The reference's IVF_PQ takes every m that divides dim (src/index/ivf/ivf_config.h:118, :138-147); the fast kernels of this
backend exist for m in {8, 16, 32, 64}, every other width up to 128 runs on the plain exact kernel.  Same bar as everywhere:
distances bit-equal, ids equal, boundary ties as the reference admits them -- for the tables the reference builds
(precomputed, residual, inner product), with a bitset, through RangeSearch, after a device-side Train / Add, through the
node, and against the reference reading the node's bytes."""
import ctypes as C

import numpy as np
import pytest

from conftest import assert_parity, gen_data
from helpers import finish_ivfpq
from oracle import binding as ob

pytestmark = pytest.mark.gpu

# (m, dim): word loads (m % 4 == 0) and byte loads, one-dimensional sub-vectors, the LDS limit (m = 128)
SHAPES = [(1, 16), (2, 32), (3, 24), (4, 32), (6, 48), (12, 48), (24, 96), (48, 96), (96, 96), (128, 128), (20, 100)]


def _gpu(ix, **kw):
    from knowhere_amd import GpuIndex
    return GpuIndex.from_data(ix, device=0, **kw)


@pytest.mark.parametrize("M,d", SHAPES, ids=[f"m{m}_d{d}" for m, d in SHAPES])
@pytest.mark.parametrize("mode", ["l2_precomputed", "l2_residual", "ip"])
def test_any_m_equals_the_oracle(port, M, d, mode):
    nb, nq, nlist = 6000, 48, 24
    metric = ob.IP if mode == "ip" else ob.L2
    xb, xq = gen_data(nb, d, 42 + M), gen_data(nq, d, 44)
    ix = ob.make_index(port, ob.IVF_PQ, metric, xb, nlist=nlist, M=M)
    kw = {}
    if mode == "l2_residual":
        ix.use_precomputed_table = 0
        ix.precomputed_table = None
        kw["precomputed_table_max_bytes"] = 1024
    else:
        finish_ivfpq(port, ix)
    g = _gpu(ix, **kw)
    bs = np.packbits(np.random.default_rng(5).random(nb) < 0.35, bitorder="little")
    for k, nprobe in ((10, 8), (1, 1), (100, nlist), (600, 5)):
        for bitset, nbits in ((None, 0), (bs, nb)):
            Do, Io = port.search(ix, xq, k, nprobe, bitset, nbits)
            D, I = g.search(xq, k, nprobe, bitset, nbits)
            assert_parity(Do, Io, D, I, metric, f"m={M} d={d} {mode} k={k} nprobe={nprobe} bitset={bitset is not None}")
    g.close()


@pytest.mark.parametrize("M,d", [(12, 48), (3, 24), (96, 96)], ids=["m12", "m3", "m96"])
@pytest.mark.parametrize("metric", [ob.L2, ob.IP], ids=["l2", "ip"])
def test_any_m_boundary_ties_and_range_search(port, M, d, metric):
    from test_gpu_ties import _dup_data
    xb, xq = _dup_data(6000, d, 40, 17 + M)
    ix = finish_ivfpq(port, ob.make_index(port, ob.IVF_PQ, metric, xb, nlist=24, M=M))
    g = _gpu(ix)
    g.profile_enable(True)
    g.profile_reset()
    for k in (1, 7, 40):
        Do, Io = port.search(ix, xq, k, 9)
        D, I = g.search(xq, k, 9)
        assert_parity(Do, Io, D, I, metric, f"m={M} ties k={k}")
    assert g.profile_get()["tie_queries"] > 0
    g.close()
    # range search on ordinary data (every list a candidate, early stop)
    xb, xq = gen_data(5000, d, 42), gen_data(16, d, 44)
    ix = finish_ivfpq(port, ob.make_index(port, ob.IVF_PQ, metric, xb, nlist=24, M=M))
    g = _gpu(ix)
    D40, _ = port.search(ix, xq, 40, 24)
    radius = float(np.median(D40[:, 20]))
    for max_empty in (0, 2):
        exp = port.range_search(ix, xq, radius, max_empty)
        got = g.range_search(xq, np.float32(radius), max_empty)
        assert np.array_equal(exp[0], got[0]) and np.array_equal(exp[1], got[1])
        assert np.array_equal(exp[2].view(np.uint32), got[2].view(np.uint32))
    g.close()


@pytest.mark.parametrize("metric", [ob.L2, ob.IP], ids=["l2", "ip"])
def test_any_m_long_lists_and_large_k(port, metric):
    """k > 64 takes the block selection (pq_scan_any_block_kernel): lists longer than its 4096-row tile (the carried list),
    k up to 1024, a bitset, and duplicated rows that tie at the boundary of the tile selection"""
    from test_gpu_ties import _dup_data
    M, d = 12, 48
    for xb, xq, what in ((gen_data(14000, d, 42), gen_data(24, d, 44), "random"), (*_dup_data(14000, d, 60, 5), "duplicates")):
        nb = len(xb)
        ix = finish_ivfpq(port, ob.make_index(port, ob.IVF_PQ, metric, xb, nlist=3, M=M))
        assert max(len(i) for i in ix.list_ids) > 4096
        g = _gpu(ix)
        bs = np.packbits(np.random.default_rng(5).random(nb) < 0.5, bitorder="little")
        for k, nprobe in ((65, 3), (100, 2), (1000, 3), (1024, 1)):
            for bitset, nbits in ((None, 0), (bs, nb)):
                Do, Io = port.search(ix, xq, k, nprobe, bitset, nbits)
                D, I = g.search(xq, k, nprobe, bitset, nbits)
                assert_parity(Do, Io, D, I, metric, f"{what} k={k} nprobe={nprobe} bitset={bitset is not None}",
                              licensed_ties=(k == 1024))  # (k + 1 results are not available at k = 1024: canonical ties)
        g.close()


def test_unsupported_shapes_are_refused():
    from knowhere_amd import GpuIndex, KnhipError
    for M, d in ((129, 258), (256, 256), (5, 32), (1, 200)):  # above 128; does not divide; sub-vectors above 144 dims
        with pytest.raises(KnhipError):
            GpuIndex(ob.IVF_PQ, ob.L2, d, nlist=8, pq_m=M)


@pytest.mark.parametrize("metric", ["L2", "IP"])
@pytest.mark.parametrize("m", [12, 4, 0], ids=["m12", "m4", "auto"])
def test_node_builds_any_m_and_the_reference_reads_it(ref, metric, m):
    """IndexFactory-level: Build with m = 12 / 4 (and m = 0 on dim 36: no fast width divides it, the node picks 18),
    Search; the reference reads the node's bytes and returns the node's results; the refine store works on top"""
    from test_faiss_io import GPU_NAME, NODE_SO, _search, _u8
    node = C.CDLL(NODE_SO)
    node.knhip_node_create.restype = C.c_void_p
    node.knhip_node_serialize.restype = C.c_int64
    node.knhip_node_last_error.restype = C.c_char_p
    nb, nq, d, k, nprobe = 4000, 32, 36 if m == 0 else 48, 10, 8
    xb, xq = gen_data(nb, d, 42), gen_data(nq, d, 44)
    h = node.knhip_node_create(GPU_NAME[ob.IVF_PQ].encode())
    try:
        cfg = f"metric_type={metric};nlist=32;nbits=8;refine=true;refine_type=fp16" + (f";m={m}" if m else "")
        rc = node.knhip_node_build(C.c_void_p(h), xb.ctypes.data_as(C.POINTER(C.c_float)), C.c_int64(nb), C.c_int64(d),
                                   cfg.encode())
        assert rc == 0, node.knhip_node_last_error().decode()
        n = node.knhip_node_serialize(C.c_void_p(h), None, C.c_int64(0))
        blob = np.empty(n, np.uint8)
        assert node.knhip_node_serialize(C.c_void_p(h), _u8(blob), C.c_int64(n)) == n
        mm = ob.L2 if metric == "L2" else ob.IP
        for kf in (1, 5):
            D, I = _search(node, h, xq, f"k={k};nprobe={nprobe}" + (f";refine_k={kf}" if kf != 1 else ""), k)
            Dr, Ir = ref.blob_search_refine(blob, xq, k, float(kf), nprobe)
            assert_parity(Dr, Ir, D, I, mm, f"node m={m} -> reference, k_factor {kf}")
        h2, _ = ref.deserialize(blob, d)
        got_m = ref.lib.ref_code_size(h2)
        ref.destroy(h2)
        assert got_m == (18 if m == 0 else m)
    finally:
        node.knhip_node_destroy(C.c_void_p(h))
