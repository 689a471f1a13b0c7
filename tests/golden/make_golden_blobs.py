"""tests/golden/make_golden_blobs.py -- index BYTES written by the reference's own faiss::write_index
(what IvfIndexNode::SerializeImpl puts in the BinarySet, reference src/index/ivf/ivf.cc:1717-1744),
plus the results the reference returns on the index read back from those bytes.

Run in the dev container (needs oracle/_ref, i.e. /root/reference):
    python tests/golden/make_golden_blobs.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import binding as ob  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def gen(n, d, seed):
    return (np.random.default_rng(seed).random((n, d), dtype=np.float32) * 100).astype(np.float32)


def main():
    ref = ob.Ref()
    nb, nq, d, nlist, M, k, nprobe = 1500, 16, 16, 8, 8, 10, 4
    xb, xq = gen(nb, d, 42), gen(nq, d, 44)
    for kind, name in ((ob.FLAT, "flat"), (ob.IVF_FLAT, "ivfflat"), (ob.IVF_PQ, "ivfpq"), (ob.IVF_SQ8, "ivfsq8")):
        for metric, mname in ((ob.L2, "l2"), (ob.IP, "ip")):
            h = ref.create(kind, metric, d, nlist, M, 8)
            ref.train_add(h, xb)
            variants = [("", None)]
            if kind in (ob.IVF_PQ, ob.IVF_SQ8) and metric == ob.L2:
                variants.append(("_refine", xb))
            for suffix, raw in variants:
                blob = ref.serialize(h, raw)
                h2, raw2 = ref.deserialize(blob, d)
                D, I = ref.search(h2, xq, k, nprobe)
                arrs = dict(blob=blob, kind=kind, metric=metric, d=d, nb=nb, nlist=nlist, M=M, k=k, nprobe=nprobe,
                            xq=xq, D=D, I=I)
                if raw is not None:
                    assert np.array_equal(raw2, xb)
                    arrs["Dr"], arrs["Ir"] = ref.search_refine(h2, raw2, xq, k, 4.0, nprobe)
                np.savez_compressed(os.path.join(OUT, f"blob_{name}_{mname}{suffix}.npz"), **arrs)
                print(name, mname, suffix, "bytes", blob.size, bytes(blob[:4]))
                ref.destroy(h2)
            ref.destroy(h)


if __name__ == "__main__":
    main()
