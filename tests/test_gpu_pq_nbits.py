"""GPU parity tests (-m gpu) of IVF-PQ with codes narrower than 8 bits (nbits 4 .. 7: the widths the reference's GPU node
accepts, src/index/gpu_cuvs/gpu_cuvs_ivf_pq_config.h:55-58; 1 .. 3 besides: its CPU node takes 1 .. 24, ivf_config.h:118-120).
On the device every width is one byte per sub-quantizer indexing 256-entry tables of which 2^nbits are in use (the rest repeat
entry 0, which no code refers to); on the host side of the C ABI the list codes are the reference's bit strings
(ProductQuantizer.cpp:69, PQEncoderGeneric).  Oracle: oracle.c's generic-width scan, pinned against the reference build in
tests/test_oracle.py::test_oracle_matches_reference_pq_code_widths."""
import numpy as np
import pytest

from conftest import assert_parity, gen_data
from helpers import finish_ivfpq
from oracle import binding as ob

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    return torch


def _bitset(n, frac, seed):
    return np.packbits(np.random.default_rng(seed).random(n) < frac, bitorder="little")


@pytest.mark.parametrize("nbits", [4, 5, 6, 7, 1, 3])
@pytest.mark.parametrize("metric", [ob.L2, ob.IP], ids=["l2", "ip"])
def test_search_equals_the_oracle(torch_cuda, port, metric, nbits):
    """every kernel family the width can reach: m = 32 / d = 128 (prefilter forms + exact 4-query kernel), m = 8 and 16
    (systolic exact kernel), m = 12 (pq_scan_any), with and without a bitset, residual tables as well as the precomputed one"""
    from knowhere_amd import GpuIndex
    for (d, M, nb, nlist, nq) in ((128, 32, 30000, 32, 200), (64, 8, 8000, 16, 40), (64, 16, 8000, 16, 40), (48, 12, 6000, 12, 30)):
        xb, xq = gen_data(nb, d, 42), gen_data(nq, d, 44)
        ix = finish_ivfpq(port, ob.make_index(port, ob.IVF_PQ, metric, xb, nlist=nlist, M=M, nbits=nbits))
        assert ix.list_codes[0].shape[1] == (M * nbits + 7) // 8
        g = GpuIndex.from_data(ix, device=0)
        assert np.array_equal(g.get_pq(), ix.pq_centroids.reshape(M, 1 << nbits, d // M))
        sizes, codes, ids = g.get_lists()  # (the reference's bytes come back)
        assert np.array_equal(codes, np.concatenate([c for c in ix.list_codes if len(c)]))
        bs = _bitset(nb, 0.4, 3)
        for k, nprobe in ((10, 8), (1, 1), (100, nlist)):
            for b, nbs in ((None, 0), (bs, nb)):
                Do, Io = port.search(ix, xq, k, nprobe, b, nbs)
                D, I = g.search(xq, k, nprobe, b, nbs)
                assert_parity(Do, Io, D, I, metric, f"nbits={nbits} m={M} d={d} k={k} nprobe={nprobe} bitset={b is not None}")
        g.close()
        if metric == ob.L2 and M == 32:  # residual tables (no precomputed table)
            g = GpuIndex.from_data(ix, device=0, precomputed_table_max_bytes=1)
            assert g.uses_precomputed_table == 0
            ix2 = ob.make_index(port, ob.IVF_PQ, metric, xb, nlist=nlist, M=M, nbits=nbits)
            ix2.use_precomputed_table, ix2.precomputed_table = 0, None
            Do, Io = port.search(ix2, xq, 10, 8)
            D, I = g.search(xq, 10, 8)
            assert_parity(Do, Io, D, I, metric, f"nbits={nbits} residual tables")
            g.close()


@pytest.mark.parametrize("nbits", [4, 6])
def test_train_add_equals_the_reference_restatement(torch_cuda, port, nbits):
    """knhip_index_train / add with narrow codes: coarse centroids, codebooks (2^nbits entries, trained on at most 256 x 2^nbits
    residuals: IndexIVFPQ.cpp:97-99) and list contents equal the restated IndexIVF::train + add (pinned against the reference
    build for these widths in tests/test_oracle.py), then the search equals the oracle's on that index"""
    from knowhere_amd import GpuIndex
    nb, d, nlist, M = 9000, 64, 24, 8
    xb, xq = gen_data(nb, d, 42), gen_data(50, d, 44)
    g = GpuIndex(2, ob.L2, d, nlist, M, nbits, device=0)
    g.train(xb)
    g.add(xb)
    cen, pq, _ = port.train_ivf(ob.IVF_PQ, ob.L2, xb, nlist, M=M, nbits=nbits)
    assert g.get_coarse().tobytes() == cen.tobytes(), "coarse centroids"
    assert g.get_pq().tobytes() == pq.tobytes(), "codebooks"
    assign = port.assign(ob.L2, cen, xb)
    codes = port.pq_encode(d, M, nbits, pq, np.ascontiguousarray(xb - cen[assign]))
    sizes, gc, gi = g.get_lists()
    pos = 0
    ix = ob.IndexData(ob.IVF_PQ, ob.L2, d, nlist, M, nbits)
    ix.centroids, ix.pq_centroids = cen, pq
    ix.use_precomputed_table = 1  # (the table fits: the index builds it, IndexIVFPQ.cpp:428-456)
    for l in range(nlist):
        sel = np.nonzero(assign == l)[0]
        n = int(sizes[l])
        assert np.array_equal(gi[pos:pos + n], sel) and np.array_equal(gc[pos:pos + n], codes[sel]), f"list {l}"
        ix.list_codes.append(codes[sel])
        ix.list_ids.append(sel.astype(np.int64))
        pos += n
    ix = finish_ivfpq(port, ix)
    Do, Io = port.search(ix, xq, 10, 8)
    D, I = g.search(xq, 10, 8)
    assert_parity(Do, Io, D, I, ob.L2, f"trained on the device, nbits={nbits}")
    g.close()


@pytest.mark.parametrize("nbits", [4, 5, 6, 7])
def test_node_builds_narrow_codes_and_both_directions_of_the_wire_format(ref, port, nbits):
    """IndexFactory-level (GPU_HIP_IVF_PQ with nbits in the cuVS node's range, gpu_cuvs_ivf_pq_config.h:55-58): Build + Search
    through the plugin; the REFERENCE reads the node's bytes (code_size (m nbits + 7) / 8) and returns the node's results;
    the node reads an index the reference trained and wrote and returns the reference's results; a sharded node ("0,0")
    returns the single device's."""
    import ctypes as C
    from test_faiss_io import CPU_NAME, GPU_NAME, NODE_SO, _search, _u8
    node = C.CDLL(NODE_SO)
    node.knhip_node_create.restype = C.c_void_p
    node.knhip_node_serialize.restype = C.c_int64
    node.knhip_node_last_error.restype = C.c_char_p
    nb, nq, d, m, k, nprobe, nlist = 6000, 32, 64, 16, 10, 8, 24
    xb, xq = gen_data(nb, d, 42), gen_data(nq, d, 44)
    h = node.knhip_node_create(GPU_NAME[ob.IVF_PQ].encode())
    hs = node.knhip_node_create(GPU_NAME[ob.IVF_PQ].encode())
    h3 = node.knhip_node_create(GPU_NAME[ob.IVF_PQ].encode())
    href = ref.create(ob.IVF_PQ, ob.L2, d, nlist, m, nbits)
    try:
        cfg = f"metric_type=L2;nlist={nlist};nbits={nbits};m={m}"
        px = xb.ctypes.data_as(C.POINTER(C.c_float))
        assert node.knhip_node_build(C.c_void_p(h), px, C.c_int64(nb), C.c_int64(d), cfg.encode()) == 0, \
            node.knhip_node_last_error().decode()
        n = node.knhip_node_serialize(C.c_void_p(h), None, C.c_int64(0))
        blob = np.empty(n, np.uint8)
        assert node.knhip_node_serialize(C.c_void_p(h), _u8(blob), C.c_int64(n)) == n
        D, I = _search(node, h, xq, f"k={k};nprobe={nprobe}", k)
        h2, _ = ref.deserialize(blob, d)
        assert ref.lib.ref_code_size(h2) == (m * nbits + 7) // 8
        Dr, Ir = ref.search(h2, xq, k, nprobe)
        ref.destroy(h2)
        assert_parity(Dr, Ir, D, I, ob.L2, f"node nbits={nbits} -> reference")
        # two shards on one device: the single device's answer
        assert node.knhip_node_build(C.c_void_p(hs), px, C.c_int64(nb), C.c_int64(d), (cfg + ";gpu_ids=0,0").encode()) == 0, \
            node.knhip_node_last_error().decode()
        Ds, Is = _search(node, hs, xq, f"k={k};nprobe={nprobe}", k)
        assert_parity(D, I, Ds, Is, ob.L2, f"sharded node nbits={nbits}")
        # reference -> node
        ref.train_add(href, xb)
        blob_r = ref.serialize(href)
        rc = node.knhip_node_deserialize(C.c_void_p(h3), CPU_NAME[ob.IVF_PQ].encode(), _u8(blob_r), C.c_int64(blob_r.size), b"")
        assert rc == 0, node.knhip_node_last_error().decode()
        D3, I3 = _search(node, h3, xq, f"k={k};nprobe={nprobe}", k)
        Dr, Ir = ref.search(href, xq, k, nprobe)
        assert_parity(Dr, Ir, D3, I3, ob.L2, f"reference nbits={nbits} -> node")
    finally:
        ref.destroy(href)
        for x in (h, hs, h3):
            node.knhip_node_destroy(C.c_void_p(x))
