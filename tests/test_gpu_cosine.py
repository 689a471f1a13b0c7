"""a12 COSINE on the GPU (-m gpu): the HIP path with the stored-norm semantics of the CPU nodes against the known
answers produced by the classes the reference's nodes instantiate (oracle/_ref/libknowhere_kref.so: IndexFlatCosine,
IndexIVFFlatCosine, knowhere::NormalizeVecs; fixtures tests/golden/cosine/*.npz) and against the oracle's restatement.
Bar: ids equal, distances bit-equal.
  FLAT      dis = clamp(<q_n, y> * inverse_norm_y, -1, 1)     (cppcontrib/knowhere/utils/distances.cpp:367-409)
  IVF_FLAT  dis = <q_n, y> / norm_y, raw rows + norms stored   (cppcontrib/knowhere/IndexIVFFlat.cpp:199-210)
The C ABI level is tested here; the node (NormalizeRow, blobs with stored norms) in tests/test_faiss_io.py."""
import numpy as np
import pytest

from conftest import assert_parity
from helpers import load_cosine_golden
from oracle import binding as ob

pytestmark = pytest.mark.gpu


def _gpu_flat(zf):
    from knowhere_amd import GpuIndex
    from knowhere_amd.index import BRUTE_FORCE
    g = GpuIndex(BRUTE_FORCE, ob.IP, zf["xb"].shape[1])
    g.add_vectors(np.ascontiguousarray(zf["xb"]))
    g.set_row_scale(zf["inv_norms"], 2)
    return g


@pytest.mark.parametrize("k", [1, 10, 120])
def test_flat_cosine_matches_the_reference(port, k):
    zf, _, _ = load_cosine_golden()
    g = _gpu_flat(zf)
    qn, _ = port.normalize(zf["xq"])  # (the node normalises the query: CopyAndNormalizeVecs, flat.cc:112-115)
    D, I = g.search(qn, k, 1)
    assert_parity(zf[f"flat_D_{k}"], zf[f"flat_I_{k}"], D, I, ob.IP, f"flat cosine k={k}")
    g.close()


def test_flat_cosine_bitset_and_range(port):
    zf, _, _ = load_cosine_golden()
    g = _gpu_flat(zf)
    qn, _ = port.normalize(zf["xq"])
    nb = zf["xb"].shape[0]
    D, I = g.search(qn, 10, 1, zf["bitset"], nb)
    assert_parity(zf["flat_D_bs"], zf["flat_I_bs"], D, I, ob.IP, "flat cosine + bitset")
    # range search: every row with similarity above the radius, the oracle's flat scan as the reference
    Do, Io = port.flat_cosine_search(zf["xb"], zf["xq"], nb)
    radius = np.float32(np.median(Do[:, 20]))
    lims, ids, dis = g.range_search(qn, radius, 0)
    for q in range(qn.shape[0]):
        exp = {(int(i), float(d)) for d, i in zip(Do[q], Io[q]) if d > radius}
        got = {(int(i), float(d)) for d, i in zip(dis[lims[q]:lims[q + 1]], ids[lims[q]:lims[q + 1]])}
        assert exp == got, q
    g.close()


@pytest.mark.parametrize("k,nprobe", [(1, 1), (10, 4), (10, 16), (120, 16)])
def test_ivfflat_cosine_matches_the_reference(port, k, nprobe):
    from knowhere_amd import GpuIndex
    zf, zi, ix = load_cosine_golden()
    g = GpuIndex.from_data(ix, device=0)  # raw rows + their norms (list_norms -> knhip_index_set_row_scale mode 1)
    qn, _ = port.normalize(zf["xq"])
    D, I = g.search(qn, k, nprobe)
    assert_parity(zi[f"D_{k}_{nprobe}"], zi[f"I_{k}_{nprobe}"], D, I, ob.IP, f"ivfflat cosine k={k} nprobe={nprobe}")
    Do, Io = port.search(ix, qn, k, nprobe)
    assert_parity(Do, Io, D, I, ob.IP, "ivfflat cosine vs oracle")
    g.close()


def test_ivfflat_cosine_bitset_and_the_norms_matter(port):
    from knowhere_amd import GpuIndex
    zf, zi, ix = load_cosine_golden()
    g = GpuIndex.from_data(ix, device=0)
    qn, _ = port.normalize(zf["xq"])
    D, I = g.search(qn, 10, 8, zf["bitset"], zf["xb"].shape[0])
    assert_parity(zi["D_bs"], zi["I_bs"], D, I, ob.IP, "ivfflat cosine + bitset")
    # without the stored norms the same index answers a plain inner-product search: different numbers
    g.set_row_scale(None, 0)
    D2, _ = g.search(qn, 10, 8)
    assert not np.array_equal(D2, zi["D_10_4"])
    g.close()
