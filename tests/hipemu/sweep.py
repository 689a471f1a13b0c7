"""tests/hipemu/sweep.py -- exploratory sweep of boundary shapes through the emulated library (not part of the test suite:
minutes per case).  Every case builds an index from the oracle's data, searches through the product's ctypes harness
on the emulated library and compares with the oracle bit for bit.

usage:  KNHIP_LIB=tests/hipemu/_build/libknhip_emu.so KNHIP_COARSE=exact KNHIP_MSCAN=0 python tests/hipemu/sweep.py FIRST LAST
(build the library first: python -c "import sys; sys.path.insert(0, 'tests/hipemu'); import emu_build; emu_build.build_api()")
Round 2: all ten cases below returned the oracle's ids and distances (exact kernels: k = 1024, d = 1, one list, more
queries than rows, bitsets with k above a wave, brute force with ragged chunks)."""
import os
import sys
import time
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
from conftest import gen_data
from oracle import binding as ob
from knowhere_amd import GpuIndex
port = ob.Port()
def run(kind, metric, nb, d, nlist, nq, k, nprobe, bitset_frac=0.0, M=8, tag=""):
    xb, xq = gen_data(nb, d, 42), gen_data(nq, d, 44)
    if kind == ob.FLAT:
        ix = ob.IndexData(ob.FLAT, metric, d); ix.base = xb
    else:
        ix = ob.make_index(port, kind, metric, xb, nlist=nlist, M=M)
    bs, nbits = None, 0
    if bitset_frac > 0:
        bs = np.packbits(np.random.default_rng(1).random(nb) < bitset_frac, bitorder="little"); nbits = nb
    t0 = time.time()
    g = GpuIndex.from_data(ix, device=0)
    Do, Io = port.search(ix, xq, k, nprobe, bs, nbits)
    D, I = g.search(xq, k, nprobe, bs, nbits)
    g.close()
    ok = np.array_equal(I, Io) and np.array_equal(D.view(np.uint32), Do.view(np.uint32))
    print(f"{'OK ' if ok else 'BAD'} {tag} kind={kind} metric={metric} nb={nb} d={d} nlist={nlist} nq={nq} k={k} nprobe={nprobe} bs={bitset_frac} ({time.time()-t0:.1f}s)", flush=True)
cases = [
 (ob.IVF_FLAT, ob.L2, 700, 7, 9, 3, 1024, 9),
 (ob.IVF_FLAT, ob.IP, 700, 1, 9, 3, 600, 4),
 (ob.IVF_FLAT, ob.L2, 300, 33, 1, 2, 10, 1),
 (ob.IVF_FLAT, ob.L2, 65, 4, 5, 70, 3, 5),
 (ob.IVF_SQ8, ob.L2, 700, 17, 9, 3, 1024, 9),
 (ob.IVF_SQ8, ob.IP, 500, 3, 6, 5, 64, 2),
 (ob.FLAT, ob.L2, 5000, 5, 0, 3, 1024, 1),
 (ob.FLAT, ob.IP, 900, 130, 0, 70, 65, 1),
 (ob.IVF_FLAT, ob.L2, 900, 16, 12, 9, 129, 12, 0.9),
 (ob.IVF_SQ8, ob.L2, 900, 16, 12, 9, 257, 7, 0.5),
]
for c in cases[int(sys.argv[1]):int(sys.argv[2])]:
    run(*c)
