"""Wire format: the HIP node reads and writes the FAISS index bytes the CPU nodes
put into the BinarySet (reference src/index/ivf/ivf.cc:1717-1834 -> faiss::write_index / read_index).

CPU: knowhere_amd/host/faiss_io.cc parses the reference's own bytes (committed fixtures
tests/golden/blob_*.npz, and live oracle/_ref output where present) and re-emits them BYTE-IDENTICALLY;
malformed blobs are rejected.  GPU: a CPU-built index loads into the node and returns the
reference's results bit-exactly; a node-built index is read back by the reference's faiss::read_index
and returns the node's results."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import ROOT, assert_parity, gen_data
from helpers import golden_blobs
from oracle import binding as ob

NODE_SO = os.path.join(ROOT, "knowhere_amd", "libknowhere_hip_node.so")
CPU_NAME = {ob.FLAT: "FLAT", ob.IVF_FLAT: "IVF_FLAT", ob.IVF_PQ: "IVF_PQ", ob.IVF_SQ8: "IVF_SQ8"}
GPU_NAME = {ob.FLAT: "GPU_HIP_BRUTE_FORCE", ob.IVF_FLAT: "GPU_HIP_IVF_FLAT", ob.IVF_PQ: "GPU_HIP_IVF_PQ",
            ob.IVF_SQ8: "GPU_HIP_IVF_SQ8"}


@pytest.fixture(scope="module")
def node():
    assert os.path.exists(NODE_SO), "build with __graft_entry__.build()"
    L = C.CDLL(NODE_SO)
    L.knhip_host_faiss_roundtrip.restype = C.c_int64
    L.knhip_node_create.restype = C.c_void_p
    L.knhip_node_serialize.restype = C.c_int64
    L.knhip_node_last_error.restype = C.c_char_p
    L.knhip_node_count.restype = C.c_int64
    return L


def _u8(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def roundtrip(L, blob):
    blob = np.ascontiguousarray(blob, np.uint8)
    err = C.create_string_buffer(256)
    out = np.empty(blob.size + 64, np.uint8)
    n = L.knhip_host_faiss_roundtrip(_u8(blob), C.c_int64(blob.size), _u8(out), C.c_int64(out.size), err, C.c_int64(256))
    if n < 0:
        raise ValueError(err.value.decode())
    return out[:n]


def info(L, blob):
    blob = np.ascontiguousarray(blob, np.uint8)
    err = C.create_string_buffer(256)
    v = np.zeros(10, np.int64)
    rc = L.knhip_host_faiss_info(_u8(blob), C.c_int64(blob.size), v.ctypes.data_as(C.POINTER(C.c_int64)), err,
                                 C.c_int64(256))
    if rc != 0:
        raise ValueError(err.value.decode())
    return dict(fourcc=int(v[0]).to_bytes(4, "little").decode(), d=int(v[1]), ntotal=int(v[2]), metric=int(v[3]),
                nlist=int(v[4]), code_size=int(v[5]), pq_M=int(v[6]), has_refine=bool(v[7]), is_cosine=bool(v[8]),
                list_total=int(v[9]))


@pytest.mark.parametrize("path", golden_blobs(), ids=lambda p: os.path.basename(p)[5:-4])
def test_reference_bytes_roundtrip_identically(node, path):
    z = np.load(path)
    blob = z["blob"]
    out = roundtrip(node, blob)
    assert out.size == blob.size and np.array_equal(out, blob)
    i = info(node, blob)
    kind, metric = int(z["kind"]), int(z["metric"])
    assert i["d"] == int(z["d"]) and i["ntotal"] == int(z["nb"]) and i["metric"] == (1 if metric == ob.L2 else 0)
    assert i["has_refine"] == path.endswith("_refine.npz") and not i["is_cosine"]
    if kind == ob.FLAT:
        assert i["fourcc"] == ("IxF2" if metric == ob.L2 else "IxFI")
    else:
        assert i["fourcc"] == {ob.IVF_FLAT: "IwFl", ob.IVF_PQ: "IwPQ", ob.IVF_SQ8: "IwSq"}[kind]
        assert i["nlist"] == int(z["nlist"]) and i["list_total"] == int(z["nb"])
        assert i["code_size"] == {ob.IVF_FLAT: 4 * int(z["d"]), ob.IVF_PQ: int(z["M"]), ob.IVF_SQ8: int(z["d"])}[kind]


def test_live_reference_bytes_roundtrip(node, ref):
    """fresh indexes from the reference (incl. mostly-empty lists -> the sparse size table, custom ids)"""
    d = 8
    xb = gen_data(400, d, 5)
    for kind in (ob.IVF_FLAT, ob.IVF_PQ, ob.IVF_SQ8):
        for nlist, nadd in ((4, 400), (10, 3)):  # 3 rows over 10 lists: <= nlist/2 non-empty -> "sprs"
            h = ref.create(kind, ob.L2, d, nlist, 4, 8)
            ref._chk(ref.lib.ref_train(h, C.c_int64(400), xb.ctypes.data_as(C.POINTER(C.c_float))))
            ids = np.arange(nadd, dtype=np.int64) * 7 + 100
            ref._chk(ref.lib.ref_add(h, C.c_int64(nadd), xb.ctypes.data_as(C.POINTER(C.c_float)),
                                     ids.ctypes.data_as(C.POINTER(C.c_int64))))
            blob = ref.serialize(h)
            assert np.array_equal(roundtrip(node, blob), blob)
            assert info(node, blob)["list_total"] == nadd
            ref.destroy(h)


@pytest.mark.parametrize("rt", [1, 2, 3, 4, 5, 6], ids=["fp16", "bf16", "sq8", "sq6", "int8", "sq4u"])
def test_live_reference_quantised_refine_bytes_roundtrip(node, ref, rt):
    """IndexRefine(base, IndexScalarQuantizer) as the reference writes it ("IxRF" ... "IxSQ" ... k_factor): parsed and
    re-emitted byte-identically; truncations and a wrong quantizer type are rejected"""
    d = 8
    xb = gen_data(400, d, 5)
    for kind in (ob.IVF_PQ, ob.IVF_SQ8):
        h = ref.create(kind, ob.IP, d, 4, 4, 8)
        ref.train_add(h, xb)
        blob = ref.serialize_sq(h, rt, xb)
        ref.destroy(h)
        assert bytes(blob[:4]) == b"IxRF"
        assert np.array_equal(roundtrip(node, blob), blob)
        i = info(node, blob)
        assert i["has_refine"] and i["ntotal"] == 400
        for bad in (blob[:-1], blob[:-5], np.concatenate([blob, np.zeros(1, np.uint8)])):
            with pytest.raises(ValueError):
                roundtrip(node, bad)
        pos = blob.tobytes().rindex(b"IxSQ")
        hdr = 4 + 4 + 8 + 16 + 1 + 4  # fourcc, d, ntotal, 2 reserved int64, is_trained, metric
        assert int(np.frombuffer(blob[pos + hdr:pos + hdr + 4].tobytes(), np.int32)[0]) == {1: 4, 2: 7, 3: 0, 4: 6, 5: 8, 6: 3}[rt]
        wrong = blob.copy()
        wrong[pos + hdr] = 1  # QT_4bit (per-dimension 4-bit ranges): not a refine type of Knowhere, not a store this backend reads
        with pytest.raises(ValueError):
            roundtrip(node, wrong)


def test_malformed_blobs_are_rejected(node):
    blob = np.load(golden_blobs()[0])["blob"]
    pq = np.load([p for p in golden_blobs() if "ivfpq_l2.npz" in p][0])["blob"]
    for bad in (pq[:100], pq[:-1], np.concatenate([pq, np.zeros(3, np.uint8)]), blob[:3]):
        with pytest.raises(ValueError):
            roundtrip(node, bad)
    unknown = pq.copy()
    unknown[:4] = np.frombuffer(b"IHNf", np.uint8)  # an index type that is not on the path
    with pytest.raises(ValueError, match="not on the HIP"):
        roundtrip(node, unknown)
    huge = pq.copy()
    huge[4 + 4 + 8 + 16 + 1 + 4: 4 + 4 + 8 + 16 + 1 + 4 + 8] = 255  # nlist = 2^64-1
    with pytest.raises(ValueError):
        roundtrip(node, huge)


def test_corrupted_blobs_never_crash_the_parser(node):
    """random byte / length-field corruption: the parser either round-trips or reports an error -- no crash,
    no runaway allocation (every length is checked against the bytes that are left)"""
    rng = np.random.default_rng(11)
    blobs = [np.load(p)["blob"] for p in golden_blobs() if "ivfpq_l2" in p or "ivfflat_l2" in p or "ivfsq8_l2_refine" in p]
    outcomes = {"ok": 0, "rejected": 0}
    for blob in blobs:
        for _ in range(150):
            b = blob.copy()
            for _ in range(int(rng.integers(1, 4))):
                pos = int(rng.integers(0, min(b.size, 400)))  # headers and size tables live at the front
                if rng.random() < 0.5:
                    b[pos] = rng.integers(0, 256)
                else:
                    b[pos:pos + 8] = 255
            try:
                roundtrip(node, b)
                outcomes["ok"] += 1
            except ValueError:
                outcomes["rejected"] += 1
    assert outcomes["rejected"] > 50 and outcomes["ok"] + outcomes["rejected"] == 150 * len(blobs)


# ------------------------------------------------------------------------------------------ GPU
def _search(L, h, xq, cfg, k):
    nq, d = xq.shape
    I = np.empty((nq, k), np.int64)
    D = np.empty((nq, k), np.float32)
    rc = L.knhip_node_search(C.c_void_p(h), xq.ctypes.data_as(C.POINTER(C.c_float)), C.c_int64(nq), C.c_int64(d),
                             cfg.encode(), None, C.c_int64(0), C.c_int64(k), I.ctypes.data_as(C.POINTER(C.c_int64)),
                             D.ctypes.data_as(C.POINTER(C.c_float)))
    assert rc == 0, L.knhip_node_last_error().decode()
    return D, I


@pytest.mark.gpu
@pytest.mark.parametrize("path", golden_blobs(), ids=lambda p: os.path.basename(p)[5:-4])
def test_cpu_built_index_loads_into_the_hip_node(node, path):
    """the bytes of a CPU IVF_PQ / IVF_FLAT / IVF_SQ8 / FLAT index, under the CPU node's BinarySet key"""
    z = np.load(path)
    kind, metric, k, nprobe = int(z["kind"]), int(z["metric"]), int(z["k"]), int(z["nprobe"])
    blob, xq = np.ascontiguousarray(z["blob"]), np.ascontiguousarray(z["xq"])
    h = node.knhip_node_create(GPU_NAME[kind].encode())
    assert h
    try:
        rc = node.knhip_node_deserialize(C.c_void_p(h), CPU_NAME[kind].encode(), _u8(blob), C.c_int64(blob.size), b"")
        assert rc == 0
        assert node.knhip_node_count(C.c_void_p(h)) == int(z["nb"])
        D, I = _search(node, h, xq, f"k={k};nprobe={nprobe}", k)
        if "Dr" in z.files:
            # an index that carries a refine index is searched THROUGH it, also with the default refine_k = 1 (ivf.cc:
            # 1076-1098: refine_k always has a value): the k results of the base index re-scored against the raw rows and
            # re-sorted (IndexRefine::search with k_base == k).  The raw rows are the tail of the blob (IxRF: ... "IxF2" /
            # "IxFI", header, vector<float>, float k_factor)
            nb, d = int(z["nb"]), xq.shape[1]
            raw = np.frombuffer(blob[-(4 + nb * d * 4):-4].tobytes(), np.float32).reshape(nb, d)
            D1, I1 = ob.Port().refine(metric, raw, xq, z["I"], k)
            assert_parity(D1, I1, D, I, metric, "cpu blob -> hip node, refine index with the default refine_k")
            D, I = _search(node, h, xq, f"k={k};nprobe={nprobe};refine_k=4", k)  # k_factor 4
            assert_parity(z["Dr"], z["Ir"], D, I, metric, "cpu blob -> hip node, refine")
        else:
            assert_parity(z["D"], z["I"], D, I, metric, "cpu blob -> hip node")
        # and the node writes the same kind of bytes back: identical up to the 16 reserved header bytes
        n = node.knhip_node_serialize(C.c_void_p(h), None, C.c_int64(0))
        out = np.empty(n, np.uint8)
        assert node.knhip_node_serialize(C.c_void_p(h), _u8(out), C.c_int64(n)) == n
        assert n == blob.size
        diff = np.nonzero(out != blob)[0]
        assert bytes(out[:4]) == bytes(blob[:4])
        # baseline faiss writes 1 << 20 into the two reserved int64 of every header, Knowhere zeros
        assert len(diff) <= 2 * (3 if "Dr" in z.files else 1) * 2 and all(blob[i] == 0x10 for i in diff)
    finally:
        node.knhip_node_destroy(C.c_void_p(h))


@pytest.mark.gpu
@pytest.mark.parametrize("kind", [ob.FLAT, ob.IVF_FLAT, ob.IVF_PQ, ob.IVF_SQ8], ids=["flat", "ivfflat", "ivfpq", "ivfsq8"])
@pytest.mark.parametrize("metric", ["L2", "IP"])
def test_hip_built_index_is_read_by_the_reference(node, ref, kind, metric):
    """Build through the plugin API on the GPU, Serialize, faiss::read_index the bytes with the
    reference, search there: same results."""
    nb, nq, d, k, nprobe = 4000, 32, 32, 10, 8
    xb, xq = gen_data(nb, d, 42), gen_data(nq, d, 44)
    h = node.knhip_node_create(GPU_NAME[kind].encode())
    try:
        refine = kind in (ob.IVF_PQ, ob.IVF_SQ8)
        cfg = f"metric_type={metric};nlist=32;m=8;nbits=8" + (";refine=true;refine_type=fp32" if refine else "")
        rc = node.knhip_node_build(C.c_void_p(h), xb.ctypes.data_as(C.POINTER(C.c_float)), C.c_int64(nb), C.c_int64(d),
                                   cfg.encode())
        assert rc == 0
        D, I = _search(node, h, xq, f"k={k};nprobe={nprobe}", k)
        n = node.knhip_node_serialize(C.c_void_p(h), None, C.c_int64(0))
        blob = np.empty(n, np.uint8)
        assert node.knhip_node_serialize(C.c_void_p(h), _u8(blob), C.c_int64(n)) == n
        assert bytes(blob[:4]) == (b"IxRF" if refine else {ob.FLAT: b"IxF2" if metric == "L2" else b"IxFI",
                                                           ob.IVF_FLAT: b"IwFl"}[kind])
        h2, raw = ref.deserialize(blob, d)
        m = ob.L2 if metric == "L2" else ob.IP
        if refine:  # (searched through the refine index, k_factor = the default refine_k = 1)
            Dr, Ir = ref.search_refine(h2, raw, xq, k, 1.0, nprobe)
        else:
            Dr, Ir = ref.search(h2, xq, k, nprobe)
        assert_parity(Dr, Ir, D, I, m, "hip blob -> reference")
        if refine:
            assert np.array_equal(raw, xb)
            D2, I2 = _search(node, h, xq, f"k={k};nprobe={nprobe};refine_k=4", k)
            Dr2, Ir2 = ref.search_refine(h2, raw, xq, k, 4.0, nprobe)
            assert_parity(Dr2, Ir2, D2, I2, m, "hip blob -> reference, refine")
        ref.destroy(h2)
    finally:
        node.knhip_node_destroy(C.c_void_p(h))


@pytest.mark.gpu
@pytest.mark.parametrize("kind", [ob.FLAT, ob.IVF_FLAT, ob.IVF_PQ], ids=["flat", "ivfflat", "ivfpq32"])
def test_node_range_search_equals_the_reference_on_the_same_bytes(node, ref, kind):
    """IndexNode::RangeSearch on a HIP-built index == faiss range_search of the reference on the index read back
    from the node's own bytes: same lims, same ids in the same order, bit-equal distances; range_filter applied."""
    nb, nq, d, k = 4000, 32, 64, 10
    xb, xq = gen_data(nb, d, 42), gen_data(nq, d, 44)
    node.knhip_node_range_search.restype = C.c_int
    h = node.knhip_node_create(GPU_NAME[kind].encode())
    try:
        cfg = "metric_type=L2;nlist=32;m=32;nbits=8"
        assert node.knhip_node_build(C.c_void_p(h), xb.ctypes.data_as(C.POINTER(C.c_float)), C.c_int64(nb), C.c_int64(d),
                                     cfg.encode()) == 0
        D, _ = _search(node, h, xq, f"k={k};nprobe=32", k)
        radius = float(np.median(D[:, k - 1]))
        n = node.knhip_node_serialize(C.c_void_p(h), None, C.c_int64(0))
        blob = np.empty(n, np.uint8)
        assert node.knhip_node_serialize(C.c_void_p(h), _u8(blob), C.c_int64(n)) == n
        h2, _ = ref.deserialize(blob, d)
        for max_empty, range_filter in ((2, None), (0, None), (2, radius * 0.5)):
            lims = np.zeros(nq + 1, np.int64)
            pi, pd = C.POINTER(C.c_int64)(), C.POINTER(C.c_float)()
            c = f"radius={radius!r};max_empty_result_buckets={max_empty}" + (
                f";range_filter={range_filter!r}" if range_filter is not None else "")
            rc = node.knhip_node_range_search(C.c_void_p(h), xq.ctypes.data_as(C.POINTER(C.c_float)), C.c_int64(nq),
                                              C.c_int64(d), c.encode(), None, C.c_int64(0),
                                              lims.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(pi), C.byref(pd))
            assert rc == 0, node.knhip_node_last_error().decode()
            tot = int(lims[-1])
            ids = np.ctypeslib.as_array(pi, shape=(max(tot, 1),))[:tot].copy()
            dis = np.ctypeslib.as_array(pd, shape=(max(tot, 1),))[:tot].copy()
            el, ei, ed = ref.range_search(h2, xq, np.float32(radius), max_empty)
            if range_filter is not None:  # [range_filter, radius), reference src/common/range_util.cc:27-48
                keep = ed >= np.float32(range_filter)
                cnt = np.array([keep[el[i]:el[i + 1]].sum() for i in range(nq)])
                el, ei, ed = np.concatenate([[0], np.cumsum(cnt)]), ei[keep], ed[keep]
            assert tot > 0 and np.array_equal(lims, el) and np.array_equal(ids, ei)
            assert np.array_equal(dis.view(np.uint32), ed.view(np.uint32))
        ref.destroy(h2)
    finally:
        node.knhip_node_destroy(C.c_void_p(h))


# ------------------------------------------------------------------------------------------ COSINE (a12)
def test_node_normalize_is_the_references_normalizevec(node):
    """hip_index_node.cc::NormalizeRow == knowhere::NormalizeVecs (src/common/utils.cc:60-93; norm^2 from the scalar hook,
    float products summed in a double) on the fixture the reference itself normalised: rows and returned norms, bit for
    bit, incl. the zero row and the already-unit row that are left alone"""
    from helpers import load_cosine_golden
    zf, _, _ = load_cosine_golden()
    x = np.array(zf["xb"], np.float32, order="C", copy=True)
    norms = np.empty(x.shape[0], np.float32)
    node.knhip_node_normalize_rows(x.ctypes.data_as(C.POINTER(C.c_float)), C.c_int64(x.shape[0]), C.c_int64(x.shape[1]),
                                   norms.ctypes.data_as(C.POINTER(C.c_float)))
    assert x.tobytes() == zf["normalized"].tobytes() and norms.tobytes() == zf["norms"].tobytes()


@pytest.mark.gpu
def test_cpu_built_cosine_blobs_load_into_the_hip_node(node):
    """the "IxF9" (IndexFlatCosine: raw rows + L2 norms) and stored-norm "IwFl" (IndexIVFFlatCosine) blobs written by the
    reference's own classes (tests/golden/make_cosine_golden.py): Deserialize under the CPU node's key, Search with
    metric COSINE (the node normalises the query) == the reference's results bit for bit; Serialize writes the bytes back"""
    from helpers import load_cosine_golden
    zf, zi, _ = load_cosine_golden()
    xq = np.ascontiguousarray(zf["xq"])
    for kind, blob, cases in ((ob.FLAT, zf["flat_blob"], [(k, 1, zf[f"flat_D_{k}"], zf[f"flat_I_{k}"]) for k in (1, 10, 120)]),
                              (ob.IVF_FLAT, zi["blob"], [(k, p, zi[f"D_{k}_{p}"], zi[f"I_{k}_{p}"])
                                                         for k, p in ((1, 1), (10, 4), (10, 16), (120, 16))])):
        blob = np.ascontiguousarray(blob)
        h = node.knhip_node_create(GPU_NAME[kind].encode())
        assert h
        try:
            rc = node.knhip_node_deserialize(C.c_void_p(h), CPU_NAME[kind].encode(), _u8(blob), C.c_int64(blob.size),
                                             b"metric_type=COSINE")
            assert rc == 0, node.knhip_node_last_error().decode()
            for k, nprobe, De, Ie in cases:
                D, I = _search(node, h, xq, f"metric_type=COSINE;k={k};nprobe={nprobe}", k)
                assert_parity(De, Ie, D, I, ob.IP, f"cosine blob kind={kind} k={k} nprobe={nprobe}")
            n = node.knhip_node_serialize(C.c_void_p(h), None, C.c_int64(0))
            out = np.empty(n, np.uint8)
            assert node.knhip_node_serialize(C.c_void_p(h), _u8(out), C.c_int64(n)) == n
            assert n == blob.size and bytes(out[:4]) == bytes(blob[:4])
            diff = np.nonzero(out != blob)[0]
            assert len(diff) <= 8, (kind, len(diff))  # (reserved header bytes only)
        finally:
            node.knhip_node_destroy(C.c_void_p(h))


@pytest.mark.gpu
def test_hip_built_cosine_index_equals_the_reference(node, kref):
    """Build FLAT / IVF_FLAT with COSINE through the plugin API on the GPU: FLAT == IndexFlatCosine on the same rows; the
    IVF_FLAT blob the node writes is read back by the reference's IndexIVFFlatCosine and searched there: same results"""
    r = np.random.default_rng(5)
    d, nb, nq, k = 32, 3000, 12, 10
    xb = (r.random((nb, d), dtype=np.float32) * 10 - 3).astype(np.float32)
    xq = (r.random((nq, d), dtype=np.float32) * 2 - 1).astype(np.float32)
    h = node.knhip_node_create(GPU_NAME[ob.FLAT].encode())
    try:
        assert node.knhip_node_build(C.c_void_p(h), xb.ctypes.data_as(C.POINTER(C.c_float)), C.c_int64(nb), C.c_int64(d),
                                     b"metric_type=COSINE") == 0
        D, I = _search(node, h, xq, f"metric_type=COSINE;k={k}", k)
        De, Ie, _ = kref.flat_cosine_search(xb, xq, k)
        assert_parity(De, Ie, D, I, ob.IP, "hip-built FLAT cosine vs IndexFlatCosine")
    finally:
        node.knhip_node_destroy(C.c_void_p(h))


ROW_TYPES = [("fp16", 1), ("bf16", 2), ("sq8", 3), ("sq6", 4), ("int8", 5), ("sq4u", 6)]


def _node_blob(node, h):
    n = node.knhip_node_serialize(C.c_void_p(h), None, C.c_int64(0))
    blob = np.empty(n, np.uint8)
    assert node.knhip_node_serialize(C.c_void_p(h), _u8(blob), C.c_int64(n)) == n
    return blob


@pytest.mark.gpu
@pytest.mark.parametrize("rt_name,rt", ROW_TYPES, ids=[r[0] for r in ROW_TYPES])
@pytest.mark.parametrize("kind", [ob.IVF_PQ, ob.IVF_SQ8], ids=["ivfpq", "ivfsq8"])
@pytest.mark.parametrize("metric", ["L2", "IP"])
def test_quantised_refine_store_round_trips_with_the_reference(node, ref, port, kind, metric, rt_name, rt):
    """refine_type = fp16 / bf16 / sq8 / sq6 / int8 / sq4u (IndexRefine over faiss::IndexScalarQuantizer, refine_utils.cc:150-185):
    node-built -> the reference reads the bytes ("IxRF" ... "IxSQ") and searches through ITS IndexRefine: the node's
    results; the refine store's code bytes and sq8 ranges are the reference's for the same rows; and a blob the
    reference wrote loads into the node and answers like the reference."""
    nb, nq, d, k, nprobe = 4000, 32, 32, 10, 8
    xb, xq = gen_data(nb, d, 42), gen_data(nq, d, 44)
    m = ob.L2 if metric == "L2" else ob.IP
    h = node.knhip_node_create(GPU_NAME[kind].encode())
    h3 = node.knhip_node_create(GPU_NAME[kind].encode())
    try:
        cfg = f"metric_type={metric};nlist=32;m=8;nbits=8;refine=true;refine_type={rt_name}"
        rc = node.knhip_node_build(C.c_void_p(h), xb.ctypes.data_as(C.POINTER(C.c_float)), C.c_int64(nb), C.c_int64(d),
                                   cfg.encode())
        assert rc == 0, node.knhip_node_last_error().decode()
        blob = _node_blob(node, h)
        assert bytes(blob[:4]) == b"IxRF" and b"IxSQ" in blob.tobytes()
        for kf in (1, 4):
            D, I = _search(node, h, xq, f"k={k};nprobe={nprobe}" + (f";refine_k={kf}" if kf != 1 else ""), k)
            Dr, Ir = ref.blob_search_refine(blob, xq, k, float(kf), nprobe)
            assert_parity(Dr, Ir, D, I, m, f"hip blob ({rt_name}) -> reference, k_factor {kf}")
        # the store inside the blob = the reference's IndexScalarQuantizer over the same rows: same tail bytes
        codes_r, tr_r = ref.sq_rows(rt, m, xb)
        tail = codes_r.tobytes() + np.float32(1.0).tobytes()
        assert blob.tobytes().endswith(tail), "code bytes of the refine store"
        if rt in (3, 4, 6):
            assert tr_r.tobytes() in blob.tobytes(), "trained ranges"
        # reference-written bytes -> node
        h2, _ = ref.deserialize(blob, d)
        blob_r = ref.serialize_sq(h2, rt, xb)
        ref.destroy(h2)
        rc = node.knhip_node_deserialize(C.c_void_p(h3), CPU_NAME[kind].encode(), _u8(blob_r), C.c_int64(blob_r.size), b"")
        assert rc == 0, node.knhip_node_last_error().decode()
        D3, I3 = _search(node, h3, xq, f"k={k};nprobe={nprobe};refine_k=4", k)
        Dr, Ir = ref.blob_search_refine(blob_r, xq, k, 4.0, nprobe)
        assert_parity(Dr, Ir, D3, I3, m, f"reference blob ({rt_name}) -> hip node")
        out = _node_blob(node, h3)
        assert out.size == blob_r.size
        diff = np.nonzero(out != blob_r)[0]
        assert len(diff) <= 12 and all(blob_r[i] == 0x10 for i in diff)  # (the reserved header bytes, as above)
    finally:
        node.knhip_node_destroy(C.c_void_p(h))
        node.knhip_node_destroy(C.c_void_p(h3))


@pytest.mark.gpu
def test_refine_needs_refine_type_and_one_device_for_quantised_stores(node):
    """`refine = true` without `refine_type` builds NO refine index (ivf_wrapper.cc:170: both are needed); an unknown type is refused;
    the type name is case-insensitive (str_to_lower, refine_utils.cc:28)"""
    nb, d = 3000, 32
    xb = gen_data(nb, d, 42)
    name = GPU_NAME[ob.IVF_PQ].encode()

    def build(cfg):
        h = node.knhip_node_create(name)
        rc = node.knhip_node_build(C.c_void_p(h), xb.ctypes.data_as(C.POINTER(C.c_float)), C.c_int64(nb), C.c_int64(d),
                                   ("metric_type=L2;nlist=16;m=8;nbits=8;" + cfg).encode())
        return h, rc

    h, rc = build("refine=true")
    assert rc == 0 and bytes(_node_blob(node, h)[:4]) == b"IwPQ"
    node.knhip_node_destroy(C.c_void_p(h))
    h, rc = build("refine=true;refine_type=sq4")
    assert rc != 0
    node.knhip_node_destroy(C.c_void_p(h))
    h, rc = build("refine=true;refine_type=FP16")
    assert rc == 0 and bytes(_node_blob(node, h)[:4]) == b"IxRF"
    node.knhip_node_destroy(C.c_void_p(h))
