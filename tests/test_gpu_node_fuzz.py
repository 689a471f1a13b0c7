"""GPU (-m gpu): randomized flows through the plugin (the node's C view, knowhere_amd/host/node_capi.cc): Build on the device with
a random kind / metric / shape / m / nbits / refine store / device list, Search with random k, nprobe and filter -- and the
REFERENCE BUILD (oracle/_ref: faiss::read_index on the node's own Serialize bytes) must return the node's ids and distance
bits; a list-sharded node ("0,0" / "0,0,0") must return the single device's.  Exercises Train / Add / layouts / search /
refine / wire format in one go, against the real reference rather than the restatement."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import assert_parity, gen_data
from oracle import binding as ob

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", range(int(os.environ.get("KNHIP_FUZZ_SEEDS", "12"))))
def test_node_flows_equal_the_reference_reading_the_nodes_bytes(ref, seed):
    from test_faiss_io import GPU_NAME, NODE_SO, _search, _u8
    node = C.CDLL(NODE_SO)
    node.knhip_node_create.restype = C.c_void_p
    node.knhip_node_serialize.restype = C.c_int64
    node.knhip_node_last_error.restype = C.c_char_p
    r = np.random.default_rng(7000 + seed)
    kind = [ob.IVF_FLAT, ob.IVF_PQ, ob.IVF_SQ8, ob.IVF_PQ][int(r.integers(0, 4))]
    metric = ["L2", "IP"][int(r.integers(0, 2))]
    mm = ob.L2 if metric == "L2" else ob.IP
    d = int(r.choice([32, 64, 128])) if kind == ob.IVF_PQ else int(r.choice([20, 48, 128]))
    nb, nlist = int(r.choice([3000, 12000])), int(r.choice([8, 24, 64]))
    xb = gen_data(nb, d, seed, -3.0, 3.0)
    cfg = f"metric_type={metric};nlist={nlist}"
    if kind == ob.IVF_PQ:
        m = int(r.choice([x for x in (4, 8, 16, 32) if d % x == 0]))
        nbits = int(r.choice([8, 8, 6, 4]))
        cfg += f";m={m};nbits={nbits}"
    refine = kind != ob.IVF_FLAT and bool(r.integers(0, 2))
    rtype = str(r.choice(["fp32", "fp16", "bf16", "sq8"])) if refine else None
    if refine:
        cfg += f";refine=true;refine_type={rtype}"
    h = node.knhip_node_create(GPU_NAME[kind].encode())
    hs = node.knhip_node_create(GPU_NAME[kind].encode())
    try:
        px = xb.ctypes.data_as(C.POINTER(C.c_float))
        assert node.knhip_node_build(C.c_void_p(h), px, C.c_int64(nb), C.c_int64(d), cfg.encode()) == 0, \
            cfg + ": " + node.knhip_node_last_error().decode()
        world = int(r.integers(2, 4))
        assert node.knhip_node_build(C.c_void_p(hs), px, C.c_int64(nb), C.c_int64(d),
                                     (cfg + ";gpu_ids=" + ",".join(["0"] * world)).encode()) == 0, node.knhip_node_last_error().decode()
        n = node.knhip_node_serialize(C.c_void_p(h), None, C.c_int64(0))
        blob = np.empty(n, np.uint8)
        assert node.knhip_node_serialize(C.c_void_p(h), _u8(blob), C.c_int64(n)) == n
        href = None
        if not refine:
            href, _ = ref.deserialize(blob, d)
        for case in range(3):
            nq = int(r.choice([1, 20, 150]))
            k = int(r.choice([1, 10, 60]))
            nprobe = int(min(nlist, r.choice([1, 5, 64])))
            xq = gen_data(nq, d, 60 + case, -3.0, 3.0)
            kf = int(r.choice([1, 3])) if refine else 1
            scfg = f"k={k};nprobe={nprobe}" + (f";refine_k={kf}" if kf != 1 else "")
            what = f"seed={seed} {cfg} {scfg} nb={nb} nq={nq}"
            D, I = _search(node, h, xq, scfg, k)
            Ds, Is = _search(node, hs, xq, scfg, k)
            assert np.array_equal(I, Is) and np.array_equal(D.view(np.uint32), Ds.view(np.uint32)), what + f": sharded x{world}"
            if refine:
                Dr, Ir = ref.blob_search_refine(blob, xq, k, float(kf), nprobe)
            else:
                Dr, Ir = ref.search(href, xq, k, nprobe)
            assert_parity(Dr, Ir, D, I, mm, what + ": node -> reference")
        if href is not None:
            ref.destroy(href)
    finally:
        node.knhip_node_destroy(C.c_void_p(h))
        node.knhip_node_destroy(C.c_void_p(hs))
