"""oracle/binding.py -- ctypes bindings for the CHECKERS (test infrastructure only).

  * ``Port``  : oracle/liboracle.so, the plain-C restatement (oracle.c).
  * ``Ref``   : oracle/_ref/libknowhere_ref.so, the reference's own FAISS sources compiled in
                place (only present when /root/reference was available at build time).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
The product package (knowhere_amd/) must never import it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

FLAT, IVF_FLAT, IVF_PQ, IVF_SQ8 = 0, 1, 2, 3
L2, IP = 0, 1

_f32p = C.POINTER(C.c_float)
_i64p = C.POINTER(C.c_int64)
_u8p = C.POINTER(C.c_uint8)


def rows_code_size(row_type, d):
    """bytes per row of a quantised refine store: 1 fp16, 2 bf16 (2 d); 3 sq8, 5 int8 (d); 4 sq6 (four values per 3 bytes);
    6 sq4u (two values per byte)"""
    if row_type == 6:
        return (d * 4 + 7) // 8
    return (d * 6 + 7) // 8 if row_type == 4 else (d if row_type in (3, 5) else 2 * d)


def _p(a, t):
    if a is None:
        return None
    return a.ctypes.data_as(t)


def build(ref=True):
    """(Re)build liboracle.so and, when the reference tree is present, _ref."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "port"])
    if ref and os.path.isdir("/root/reference/thirdparty/faiss/faiss"):
        subprocess.check_call(["make", "-s", "-j8", "-C", _HERE, "ref", "kref"])


class _OrcIndex(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("metric", C.c_int32), ("d", C.c_int32), ("M", C.c_int32),
        ("nbits", C.c_int32), ("use_precomputed_table", C.c_int32),
        ("nlist", C.c_int64), ("code_size", C.c_int64),
        ("centroids", _f32p), ("pq_centroids", _f32p), ("precomputed_table", _f32p),
        ("sq_trained", _f32p), ("list_sizes", _i64p),
        ("list_codes", C.POINTER(_u8p)), ("list_ids", C.POINTER(_i64p)), ("list_norms", C.POINTER(_f32p)),
    ]


class IndexData:
    """A trained + populated index as plain numpy arrays: the bytes shared by oracle and GPU."""

    def __init__(self, kind, metric, d, nlist=0, M=0, nbits=8):
        self.kind, self.metric, self.d, self.nlist, self.M, self.nbits = kind, metric, d, nlist, M, nbits
        self.centroids = None        # [nlist, d] f32
        self.pq_centroids = None     # [M, ksub, dsub] f32
        self.sq_trained = None       # [2d] f32
        self.precomputed_table = None
        self.use_precomputed_table = 0
        self.list_codes = []         # per list uint8 [len, code_size]
        self.list_ids = []           # per list int64 [len]
        self.base = None             # FLAT: [n, d] f32
        self.list_norms = None       # IVF_FLAT + COSINE: per list float32 [len] (rows stay raw)

    @property
    def code_size(self):
        # (IVF_PQ: the reference's code bytes -- M indices of nbits bits as a little-endian bit string, ProductQuantizer.cpp:69)
        return {FLAT: self.d * 4, IVF_FLAT: self.d * 4, IVF_PQ: (self.M * self.nbits + 7) // 8, IVF_SQ8: self.d}[self.kind]

    @property
    def ntotal(self):
        if self.kind == FLAT:
            return 0 if self.base is None else self.base.shape[0]
        return int(sum(len(i) for i in self.list_ids))


class _SimdTable:
    """The src/simd hook table (reference src/simd/hook.h:33-123) as exposed by either checker: Port = the restatement
    in oracle.c, Ref = the reference's own distances_ref.cc through oracle/ref_simd.cpp.  Same call shapes for both so
    tests can compare them entry by entry.  Typed operands: numpy float16, uint16 bit patterns (bf16) or int8."""
    _PORT_PLAIN = ("fvec_inner_product", "fvec_L2sqr", "fvec_L2sqr_ny", "fvec_inner_products_ny", "fvec_madd")

    def _sfn(self, name, restype=None):
        if self._simd_prefix == "orc_simd_" and name in self._PORT_PLAIN:
            fn = getattr(self.lib, "orc_" + name)
        else:
            fn = getattr(self.lib, self._simd_prefix + name)
        fn.restype = restype
        return fn

    @staticmethod
    def _typed(a):
        a = np.ascontiguousarray(a)
        t = {np.dtype(np.float16): 0, np.dtype(np.uint16): 1, np.dtype(np.int8): 2}[a.dtype]
        return t, a, C.c_void_p(a.ctypes.data)

    def simd_scalar(self, name, x, y=None):
        """fvec_inner_product | fvec_L2sqr | fvec_L1 | fvec_Linf | fvec_norm_L2sqr"""
        x = np.ascontiguousarray(x, np.float32)
        fn = self._sfn(name, C.c_float)
        if name == "fvec_norm_L2sqr":
            return np.float32(fn(_p(x, _f32p), C.c_int64(x.size)))
        y = np.ascontiguousarray(y, np.float32)
        return np.float32(fn(_p(x, _f32p), _p(y, _f32p), C.c_int64(x.size)))

    def simd_ny(self, name, x, y):
        """fvec_L2sqr_ny | fvec_inner_products_ny : x [d], y [ny, d]"""
        ny, d = y.shape
        out = np.empty(ny, np.float32)
        self._sfn(name)(_p(out, _f32p), _p(x, _f32p), _p(y, _f32p), C.c_int64(d), C.c_int64(ny))
        return out

    def simd_ny_transposed(self, x, yt, y_sqlen, ny, nearest=False):
        """fvec_L2sqr_ny_transposed (+ _nearest_y_transposed): yt [d, d_offset] with vector i in column i"""
        d, d_offset = yt.shape
        out = np.empty(ny, np.float32)
        if nearest:
            idx = self._sfn("fvec_L2sqr_ny_nearest_y_transposed", C.c_int64)(
                _p(out, _f32p), _p(x, _f32p), _p(yt, _f32p), _p(y_sqlen, _f32p), C.c_int64(d), C.c_int64(d_offset),
                C.c_int64(ny))
            return int(idx), out
        self._sfn("fvec_L2sqr_ny_transposed")(_p(out, _f32p), _p(x, _f32p), _p(yt, _f32p), _p(y_sqlen, _f32p),
                                              C.c_int64(d), C.c_int64(d_offset), C.c_int64(ny))
        return out

    def simd_ny_nearest(self, x, y):
        ny, d = y.shape
        out = np.empty(ny, np.float32)
        idx = self._sfn("fvec_L2sqr_ny_nearest", C.c_int64)(_p(out, _f32p), _p(x, _f32p), _p(y, _f32p), C.c_int64(d),
                                                            C.c_int64(ny))
        return int(idx), out

    def simd_madd(self, a, bf, b, argmin=False):
        c = np.empty_like(a)
        if argmin:
            i = self._sfn("fvec_madd_and_argmin", C.c_int)(C.c_int64(a.size), _p(a, _f32p), C.c_float(bf), _p(b, _f32p),
                                                           _p(c, _f32p))
            return int(i), c
        self._sfn("fvec_madd")(C.c_int64(a.size), _p(a, _f32p), C.c_float(bf), _p(b, _f32p), _p(c, _f32p))
        return c

    def simd_batch_4(self, is_l2, x, ys):
        """fvec_/fp16_vec_/bf16_vec_/int8_vec_ {L2sqr, inner_product}_batch_4: ys = four rows of x's dtype"""
        out = np.empty(4, np.float32)
        if x.dtype == np.float32:
            rows = [np.ascontiguousarray(r, np.float32) for r in ys]
            self._sfn("fvec_batch_4")(C.c_int(int(is_l2)), _p(x, _f32p), *[_p(r, _f32p) for r in rows],
                                      C.c_int64(x.size), _p(out, _f32p))
            return out
        t, xa, xp = self._typed(x)
        rows = [self._typed(r) for r in ys]
        self._sfn("typed_batch_4")(C.c_int(t), C.c_int(int(is_l2)), xp, *[r[2] for r in rows], C.c_int64(xa.size),
                                   _p(out, _f32p))
        return out

    def simd_typed(self, op, x, y=None):
        """{fp16,bf16,int8}_vec_{L2sqr (op 0), inner_product (1), norm_L2sqr (2)}"""
        t, xa, xp = self._typed(x)
        yp = self._typed(y)[2] if y is not None else None
        if y is not None:
            keep = np.ascontiguousarray(y)  # keep the buffer alive across the call
            yp = C.c_void_p(keep.ctypes.data)
        return np.float32(self._sfn("typed", C.c_float)(C.c_int(t), C.c_int(op), xp, yp, C.c_int64(xa.size)))

    def simd_ivec(self, is_l2, x, y):
        i8p = C.POINTER(C.c_int8)
        if self._simd_prefix == "orc_simd_":
            return int(self._sfn("ivec", C.c_int32)(C.c_int(int(is_l2)), _p(x, i8p), _p(y, i8p), C.c_int64(x.size)))
        name = "ivec_L2sqr" if is_l2 else "ivec_inner_product"
        return int(self._sfn(name, C.c_int32)(_p(x, i8p), _p(y, i8p), C.c_int64(x.size)))


class Port(_SimdTable):
    """oracle.c through ctypes."""
    _simd_prefix = "orc_simd_"

    def __init__(self, path=None):
        path = path or os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build(ref=False)
        self.lib = L = C.CDLL(path)
        L.orc_fvec_L2sqr.restype = C.c_float
        L.orc_fvec_inner_product.restype = C.c_float
        L.orc_fvec_norm_L2sqr.restype = C.c_float
        L.orc_ivfpq_list_table.restype = C.c_float
        L.orc_int8_vec_inner_product.restype = C.c_int32
        L.orc_int8_vec_L2sqr.restype = C.c_int32
        L.orc_heap_reorder.restype = C.c_size_t

    # -- primitives --
    def fvec_L2sqr(self, x, y):
        return float(self.lib.orc_fvec_L2sqr(_p(x, _f32p), _p(y, _f32p), C.c_size_t(x.size)))

    def fvec_inner_product(self, x, y):
        return float(self.lib.orc_fvec_inner_product(_p(x, _f32p), _p(y, _f32p), C.c_size_t(x.size)))

    def fvec_norm_L2sqr(self, x):
        return float(self.lib.orc_fvec_norm_L2sqr(_p(x, _f32p), C.c_size_t(x.size)))

    def fvec_L2sqr_ny(self, x, y):
        ny, d = y.shape
        out = np.empty(ny, np.float32)
        self.lib.orc_fvec_L2sqr_ny(_p(out, _f32p), _p(x, _f32p), _p(y, _f32p), C.c_size_t(d), C.c_size_t(ny))
        return out

    def fvec_inner_products_ny(self, x, y):
        ny, d = y.shape
        out = np.empty(ny, np.float32)
        self.lib.orc_fvec_inner_products_ny(_p(out, _f32p), _p(x, _f32p), _p(y, _f32p), C.c_size_t(d), C.c_size_t(ny))
        return out

    def fvec_madd(self, a, bf, b):
        c = np.empty_like(a)
        self.lib.orc_fvec_madd(C.c_size_t(a.size), _p(a, _f32p), C.c_float(bf), _p(b, _f32p), _p(c, _f32p))
        return c

    def int8_ny(self, x, y, metric):
        fn = self.lib.orc_int8_vec_L2sqr if metric == L2 else self.lib.orc_int8_vec_inner_product
        i8p = C.POINTER(C.c_int8)
        return np.array([float(fn(_p(x, i8p), _p(np.ascontiguousarray(r), i8p), C.c_size_t(x.size))) for r in y],
                        np.float32)

    # -- index marshalling --
    def _marshal(self, ix):
        s = _OrcIndex()
        keep = []
        s.kind, s.metric, s.d, s.M, s.nbits = ix.kind, ix.metric, ix.d, ix.M, ix.nbits
        s.use_precomputed_table = ix.use_precomputed_table
        s.nlist, s.code_size = ix.nlist, ix.code_size
        s.centroids = _p(ix.centroids, _f32p)
        s.pq_centroids = _p(ix.pq_centroids, _f32p)
        s.precomputed_table = _p(ix.precomputed_table, _f32p)
        s.sq_trained = _p(ix.sq_trained, _f32p)
        sizes = np.array([len(i) for i in ix.list_ids], np.int64)
        codes = (_u8p * ix.nlist)()
        ids = (_i64p * ix.nlist)()
        for l in range(ix.nlist):
            c = np.ascontiguousarray(ix.list_codes[l], np.uint8)
            i = np.ascontiguousarray(ix.list_ids[l], np.int64)
            keep += [c, i]
            codes[l] = _p(c, _u8p) if c.size else None
            ids[l] = _p(i, _i64p) if i.size else None
        s.list_sizes = _p(sizes, _i64p)
        s.list_codes = codes
        s.list_ids = ids
        keep += [sizes, codes, ids]
        if getattr(ix, "list_norms", None) is not None:
            norms = (_f32p * ix.nlist)()
            for l in range(ix.nlist):
                v = np.ascontiguousarray(ix.list_norms[l], np.float32)
                keep.append(v)
                norms[l] = _p(v, _f32p) if v.size else None
            s.list_norms = norms
            keep.append(norms)
        return s, keep

    # -- cosine --
    def normalize(self, x):
        """knowhere::NormalizeVecs on a copy -> (normalised rows, norms)"""
        x = np.array(x, np.float32, order="C", copy=True)
        norms = np.empty(x.shape[0], np.float32)
        self.lib.orc_normalize_vecs(_p(x, _f32p), C.c_int64(x.shape[0]), C.c_int(x.shape[1]), _p(norms, _f32p))
        return x, norms

    def inverse_l2_norms(self, x):
        x = np.ascontiguousarray(x, np.float32)
        inv = np.empty(x.shape[0], np.float32)
        self.lib.orc_inverse_l2_norms(_p(x, _f32p), C.c_int64(x.shape[0]), C.c_int(x.shape[1]), _p(inv, _f32p))
        return inv

    def flat_cosine_search(self, xb, xq, k, bitset=None):
        """FLAT + COSINE as FlatIndexNode: raw queries in (normalised here, as the node does)"""
        xb = np.ascontiguousarray(xb, np.float32)
        qn, _ = self.normalize(xq)
        inv = self.inverse_l2_norms(xb)
        nb, d = xb.shape
        nq = qn.shape[0]
        D = np.empty((nq, k), np.float32)
        I = np.empty((nq, k), np.int64)
        self.lib.orc_flat_cosine_search(C.c_int(d), C.c_int64(nb), _p(xb, _f32p), _p(inv, _f32p), C.c_int64(nq),
                                        _p(qn, _f32p), C.c_int64(k), _p(bitset, _u8p),
                                        C.c_int64(0 if bitset is None else nb), _p(D, _f32p), _p(I, _i64p))
        return D, I

    # -- searches --
    def flat_search(self, metric, xb, xq, k, bitset=None):
        xb = np.ascontiguousarray(xb, np.float32)
        xq = np.ascontiguousarray(xq, np.float32)
        nb, d = xb.shape
        nq = xq.shape[0]
        D = np.empty((nq, k), np.float32)
        I = np.empty((nq, k), np.int64)
        self.lib.orc_flat_search(C.c_int(metric), C.c_int(d), C.c_int64(nb), _p(xb, _f32p), C.c_int64(nq),
                                 _p(xq, _f32p), C.c_int64(k), _p(bitset, _u8p),
                                 C.c_int64(0 if bitset is None else nb), _p(D, _f32p), _p(I, _i64p))
        return D, I

    def coarse_search(self, ix, xq, nprobe):
        xq = np.ascontiguousarray(xq, np.float32)
        s, keep = self._marshal(ix)
        nq = xq.shape[0]
        D = np.empty((nq, nprobe), np.float32)
        I = np.empty((nq, nprobe), np.int64)
        self.lib.orc_coarse_search(C.byref(s), C.c_int64(nq), _p(xq, _f32p), C.c_int64(nprobe), _p(D, _f32p), _p(I, _i64p))
        return D, I

    def ivf_search(self, ix, xq, k, nprobe, bitset=None, nbits=0):
        xq = np.ascontiguousarray(xq, np.float32)
        s, keep = self._marshal(ix)
        nq = xq.shape[0]
        D = np.empty((nq, k), np.float32)
        I = np.empty((nq, k), np.int64)
        rc = self.lib.orc_ivf_search(C.byref(s), C.c_int64(nq), _p(xq, _f32p), C.c_int64(k), C.c_int64(nprobe),
                                     _p(bitset, _u8p), C.c_int64(nbits), _p(D, _f32p), _p(I, _i64p))
        assert rc == 0
        return D, I

    def ivf_search_preassigned(self, ix, xq, k, keys, cdis, bitset=None, nbits=0):
        """IndexIVF::search_preassigned: the scan half of ivf_search with a given coarse assignment"""
        xq = np.ascontiguousarray(xq, np.float32)
        keys = np.ascontiguousarray(keys, np.int64)
        cdis = np.ascontiguousarray(cdis, np.float32)
        nq, nprobe = keys.shape
        m, keep = self._marshal(ix)
        D = np.empty((nq, k), np.float32)
        I = np.empty((nq, k), np.int64)
        rc = self.lib.orc_ivf_search_preassigned(C.byref(m), C.c_int64(nq), _p(xq, _f32p), C.c_int64(k),
                                                 C.c_int64(nprobe), _p(keys, _i64p), _p(cdis, _f32p),
                                                 _p(bitset, _u8p), C.c_int64(nbits), _p(D, _f32p), _p(I, _i64p))
        assert rc == 0
        return D, I

    def search(self, ix, xq, k, nprobe=1, bitset=None, nbits=0):
        if ix.kind == FLAT:
            return self.flat_search(ix.metric, ix.base, xq, k, bitset)
        return self.ivf_search(ix, xq, k, nprobe, bitset, nbits)

    def _collect(self, lims, pi, pd, free):
        n = int(lims[-1])
        ids = np.ctypeslib.as_array(pi, shape=(max(n, 1),))[:n].copy() if n else np.empty(0, np.int64)
        dis = np.ctypeslib.as_array(pd, shape=(max(n, 1),))[:n].copy() if n else np.empty(0, np.float32)
        free(pi)
        free(pd)
        return lims, ids, dis

    def range_search(self, ix, xq, radius, max_empty=2, bitset=None, nbits=0):
        """-> (lims[nq+1], ids, distances): FLAT or IVF range search, results in the reference's emission order"""
        xq = np.ascontiguousarray(xq, np.float32)
        nq = xq.shape[0]
        lims = np.zeros(nq + 1, np.int64)
        pi, pd = _i64p(), _f32p()
        L = self.lib
        if ix.kind == FLAT:
            base = np.ascontiguousarray(ix.base, np.float32)
            rc = L.orc_flat_range_search(C.c_int(ix.metric), C.c_int(ix.d), C.c_int64(base.shape[0]), _p(base, _f32p),
                                         C.c_int64(nq), _p(xq, _f32p), C.c_float(radius), _p(bitset, _u8p),
                                         C.c_int64(nbits), _p(lims, _i64p), C.byref(pi), C.byref(pd))
        else:
            m, keep = self._marshal(ix)
            rc = L.orc_ivf_range_search(C.byref(m), C.c_int64(nq), _p(xq, _f32p), C.c_float(radius),
                                        C.c_int64(max_empty), _p(bitset, _u8p), C.c_int64(nbits), _p(lims, _i64p),
                                        C.byref(pi), C.byref(pd))
        assert rc == 0
        return self._collect(lims, pi, pd, L.orc_free)

    def merge_topk(self, metric, D_parts, I_parts):
        nshard, nq, k = D_parts.shape
        D = np.empty((nq, k), np.float32)
        I = np.empty((nq, k), np.int64)
        Dp = np.ascontiguousarray(D_parts, np.float32)
        Ip = np.ascontiguousarray(I_parts, np.int64)
        self.lib.orc_merge_topk(C.c_int(metric), C.c_int64(nq), C.c_int64(k), C.c_int(nshard), _p(Dp, _f32p),
                                _p(Ip, _i64p), _p(D, _f32p), _p(I, _i64p))
        return D, I

    def refine(self, metric, base, xq, cand_ids, k, id_base=0):
        base = np.ascontiguousarray(base, np.float32)
        xq = np.ascontiguousarray(xq, np.float32)
        cand = np.ascontiguousarray(cand_ids, np.int64)
        nq, kb = cand.shape
        D = np.empty((nq, k), np.float32)
        I = np.empty((nq, k), np.int64)
        self.lib.orc_refine(C.c_int(metric), C.c_int(base.shape[1]), _p(base, _f32p), C.c_int64(base.shape[0]),
                            C.c_int64(id_base), C.c_int64(nq), _p(xq, _f32p), C.c_int64(kb), _p(cand, _i64p),
                            C.c_int64(k), _p(D, _f32p), _p(I, _i64p))
        return D, I

    # ---- quantised refine store (row_type 1 fp16, 2 bf16, 3 sq8, 4 sq6, 5 int8 = QT_8bit_direct_signed)
    def rows_train(self, x):
        x = np.ascontiguousarray(x, np.float32)
        tr = np.empty(2 * x.shape[1], np.float32)
        self.lib.orc_rows_train(C.c_int(x.shape[1]), C.c_int64(x.shape[0]), _p(x, _f32p), _p(tr, _f32p))
        return tr

    def rows_train_uniform(self, metric, x):
        """the one range of a QT_4bit_uniform refine store as Knowhere trains it: quantiles 1 % / 99 % for L2, min / max else"""
        x = np.ascontiguousarray(x, np.float32)
        tr = np.empty(2, np.float32)
        self.lib.orc_rows_train_uniform(C.c_int(2 if metric == L2 else 0), C.c_float(0.01 if metric == L2 else 0.0),
                                        C.c_int64(x.size), _p(x, _f32p), _p(tr, _f32p))
        return tr

    def rows_encode(self, row_type, x, trained=None):
        x = np.ascontiguousarray(x, np.float32)
        n, d = x.shape
        cs = rows_code_size(row_type, d)
        codes = np.empty((n, cs), np.uint8)
        self.lib.orc_rows_encode(C.c_int(row_type), C.c_int(d), C.c_int64(n), _p(x, _f32p), _p(trained, _f32p),
                                 _p(codes, _u8p))
        return codes

    def rows_decode(self, row_type, d, codes, trained=None):
        codes = np.ascontiguousarray(codes, np.uint8)
        x = np.empty((codes.shape[0], d), np.float32)
        self.lib.orc_rows_decode(C.c_int(row_type), C.c_int(d), C.c_int64(codes.shape[0]), _p(codes, _u8p),
                                 _p(trained, _f32p), _p(x, _f32p))
        return x

    def refine_rows(self, metric, row_type, d, codes, trained, xq, cand_ids, k):
        codes = np.ascontiguousarray(codes, np.uint8)
        xq = np.ascontiguousarray(xq, np.float32)
        cand = np.ascontiguousarray(cand_ids, np.int64)
        nq, kb = cand.shape
        D = np.empty((nq, k), np.float32)
        I = np.empty((nq, k), np.int64)
        self.lib.orc_refine_rows(C.c_int(metric), C.c_int(d), C.c_int(row_type), _p(codes, _u8p), _p(trained, _f32p),
                                 C.c_int64(codes.shape[0]), C.c_int64(nq), _p(xq, _f32p), C.c_int64(kb), _p(cand, _i64p),
                                 C.c_int64(k), _p(D, _f32p), _p(I, _i64p))
        return D, I

    def pq_precompute_table(self, d, M, nbits, centroids, cb):
        nlist = centroids.shape[0]
        out = np.empty((nlist, M * (1 << nbits)), np.float32)
        self.lib.orc_pq_precompute_table(C.c_int(d), C.c_int(M), C.c_int(nbits), C.c_int64(nlist),
                                         _p(centroids, _f32p), _p(cb, _f32p), _p(out, _f32p))
        return out

    # -- training restatements (Clustering / IndexIVF::train) --
    def rand_perm(self, n, seed):
        out = np.empty(n, np.int64)
        self.lib.orc_rand_perm(_p(out, _i64p), C.c_int64(n), C.c_int64(seed))
        return out

    def kmeans(self, metric, x, k, niter=25, max_points=256, seed=1234, ld=None, off=0, d=None, spherical=False):
        x = np.ascontiguousarray(x, np.float32)
        ld = x.shape[1] if ld is None else ld
        d = x.shape[1] if d is None else d
        cen = np.empty((k, d), np.float32)
        self.lib.orc_kmeans(C.c_int(metric), C.c_int(d), C.c_int64(x.shape[0]), _p(x, _f32p), C.c_int64(ld), C.c_int(off),
                            C.c_int64(k), C.c_int(niter), C.c_int(max_points), C.c_int64(seed),
                            C.c_int(1 if spherical else 0), _p(cen, _f32p))
        return cen

    def train_ivf(self, kind, metric, x, nlist, M=0, niter=0, max_points=256, seed=1234, centroids=None, nbits=8):
        """-> (centroids, pq_centroids or None, sq_trained or None); niter 0 = the level-1 quantizer's default (10)"""
        x = np.ascontiguousarray(x, np.float32)
        n, d = x.shape
        given = centroids is not None
        cen = np.ascontiguousarray(centroids, np.float32).copy() if given else np.empty((nlist, d), np.float32)
        pq = np.empty((max(M, 1), 1 << nbits, d // max(M, 1)), np.float32) if kind == IVF_PQ else None
        sq = np.empty(2 * d, np.float32) if kind == IVF_SQ8 else None
        self.lib.orc_train_ivf_nbits(C.c_int(kind), C.c_int(metric), C.c_int(d), C.c_int64(nlist), C.c_int(max(M, 1)),
                                     C.c_int(nbits), C.c_int64(n), _p(x, _f32p), C.c_int(niter), C.c_int(max_points),
                                     C.c_int64(seed), C.c_int(1 if given else 0), _p(cen, _f32p), _p(pq, _f32p),
                                     _p(sq, _f32p))
        return cen, pq, sq

    # -- build helpers (restated add path) --
    def assign(self, metric, centroids, x):
        out = np.empty(x.shape[0], np.int64)
        self.lib.orc_assign(C.c_int(metric), C.c_int(x.shape[1]), C.c_int64(centroids.shape[0]),
                            _p(centroids, _f32p), C.c_int64(x.shape[0]), _p(x, _f32p), _p(out, _i64p))
        return out

    def pq_encode(self, d, M, nbits, cb, x):
        """-> the reference's code bytes [n, (M nbits + 7) / 8]"""
        cs = (M * nbits + 7) // 8
        codes = np.empty((x.shape[0], cs), np.uint8)
        buf = np.empty(max(M, cs), np.uint8)
        for i in range(x.shape[0]):
            self.lib.orc_pq_compute_code(C.c_int(d), C.c_int(M), C.c_int(nbits), _p(cb, _f32p),
                                         _p(np.ascontiguousarray(x[i]), _f32p),
                                         buf.ctypes.data_as(_u8p))
            codes[i] = buf[:cs]
        return codes

    def pq_pack(self, M, nbits, idx):
        """[n, M] byte-wide indices -> the reference's code bytes"""
        idx = np.ascontiguousarray(idx, np.uint8)
        out = np.empty((idx.shape[0], (M * nbits + 7) // 8), np.uint8)
        for i in range(idx.shape[0]):
            self.lib.orc_pq_pack(C.c_int(M), C.c_int(nbits), idx[i].ctypes.data_as(_u8p), out[i].ctypes.data_as(_u8p))
        return out

    def pq_unpack(self, M, nbits, codes):
        codes = np.ascontiguousarray(codes, np.uint8)
        out = np.empty((codes.shape[0], M), np.uint8)
        for i in range(codes.shape[0]):
            self.lib.orc_pq_unpack(C.c_int(M), C.c_int(nbits), codes[i].ctypes.data_as(_u8p), out[i].ctypes.data_as(_u8p))
        return out

    def sq8_encode(self, trained, x):
        d = x.shape[1]
        codes = np.empty((x.shape[0], d), np.uint8)
        for i in range(x.shape[0]):
            self.lib.orc_sq8_encode(C.c_int(d), _p(trained, _f32p), _p(np.ascontiguousarray(x[i]), _f32p),
                                    codes[i].ctypes.data_as(_u8p))
        return codes


def make_index(port, kind, metric, xb, nlist=16, M=8, nbits=8, seed=123, ids=None):
    """Build a VALID (not good) index without any training library: centroids / codebooks are
    samples of the data, SQ ranges are min/max; assignment and encoding use the restated add
    path.  Parity of Search() only needs both sides to hold the same index bytes."""
    xb = np.ascontiguousarray(xb, np.float32)
    n, d = xb.shape
    rng = np.random.default_rng(seed)
    ix = IndexData(kind, metric, d, nlist if kind != FLAT else 0, M if kind == IVF_PQ else 0, nbits)
    if kind == FLAT:
        ix.base = xb
        return ix
    ids = np.arange(n, dtype=np.int64) if ids is None else np.asarray(ids, np.int64)
    ix.centroids = np.ascontiguousarray(xb[rng.choice(n, nlist, replace=False)])
    assign = port.assign(metric, ix.centroids, xb)
    resid = xb - ix.centroids[assign]
    if kind == IVF_FLAT:
        codes = xb.view(np.uint8).reshape(n, d * 4)
    elif kind == IVF_PQ:
        ksub, dsub = 1 << nbits, d // M
        pick = rng.choice(n, ksub, replace=n < ksub)
        ix.pq_centroids = np.ascontiguousarray(
            resid[pick].reshape(ksub, M, dsub).transpose(1, 0, 2)).astype(np.float32)
        codes = port.pq_encode(d, M, nbits, ix.pq_centroids, np.ascontiguousarray(resid))
        if metric == L2:
            ix.precomputed_table = port.pq_precompute_table(d, M, nbits, ix.centroids, ix.pq_centroids)
            ix.use_precomputed_table = 1
    elif kind == IVF_SQ8:
        vmin = resid.min(0)
        vdiff = resid.max(0) - vmin
        ix.sq_trained = np.concatenate([vmin, vdiff]).astype(np.float32)
        codes = port.sq8_encode(ix.sq_trained, np.ascontiguousarray(resid))
    else:
        raise ValueError(kind)
    for l in range(nlist):
        sel = np.nonzero(assign == l)[0]
        ix.list_codes.append(np.ascontiguousarray(codes[sel]))
        ix.list_ids.append(np.ascontiguousarray(ids[sel]))
    return ix


class Ref(_SimdTable):
    _simd_prefix = "ref_simd_"

    """oracle/_ref/libknowhere_ref*.so -- the reference's own FAISS, driven Knowhere-style.
    simd = "scalar": the SIMDLevel::NONE build every parity pin uses; "avx2": the dynamic-dispatch AVX2 build
    (oracle/Makefile ref_avx2), used ONLY as the timed cpu_baseline of bench.py."""

    @staticmethod
    def path(simd="scalar"):
        name = "libknowhere_ref.so" if simd == "scalar" else f"libknowhere_ref_{simd}.so"
        return os.path.join(_HERE, "_ref", name)

    @staticmethod
    def available(simd="scalar"):
        if not os.path.exists(Ref.path(simd)):
            return False
        try:
            Ref(simd)
            return True
        except OSError:
            return False

    def __init__(self, simd="scalar"):
        # libmkl_rt picks its threading layer at load time
        os.environ.setdefault("MKL_THREADING_LAYER", "GNU")
        # the .so was linked against MKL by full path (oracle/Makefile); preload it the same way so
        # no LD_LIBRARY_PATH is needed (adding /opt/conda/lib to the search path would also drag in
        # conda's older libstdc++)
        for mkl in ("/opt/conda/lib/libmkl_rt.so.1", "/opt/conda/lib/libmkl_rt.so"):
            if os.path.exists(mkl):
                C.CDLL(mkl, mode=C.RTLD_GLOBAL)
                break
        # local + deep binding: the scalar and the AVX2 build define the same faiss symbols and may live in one
        # process; each must call its own
        self.simd = simd
        self.lib = L = C.CDLL(Ref.path(simd), mode=os.RTLD_LOCAL | os.RTLD_DEEPBIND | os.RTLD_NOW)
        L.ref_create.restype = C.c_void_p
        L.ref_deserialize.restype = C.c_void_p
        L.ref_serialize.restype = C.c_int64
        if hasattr(L, "ref_serialize_sq"):
            L.ref_serialize_sq.restype = C.c_int64
        L.ref_last_error.restype = C.c_char_p
        for f in ("ref_ntotal", "ref_nlist", "ref_code_size", "ref_list_size", "ref_get_precomputed_table"):
            getattr(L, f).restype = C.c_int64
        L.ref_fvec_L2sqr.restype = C.c_float
        L.ref_fvec_inner_product.restype = C.c_float
        L.ref_fvec_norm_L2sqr.restype = C.c_float

    def free(self, h):
        self.destroy(h)

    def kmeans(self, metric, x, k, niter=25, max_points=256, seed=1234, spherical=False):
        x = np.ascontiguousarray(x, np.float32)
        cen = np.empty((k, x.shape[1]), np.float32)
        self._chk(self.lib.ref_kmeans(C.c_int(metric), C.c_int(x.shape[1]), C.c_int64(x.shape[0]), _p(x, _f32p),
                                      C.c_int64(k), C.c_int(niter), C.c_int(max_points), C.c_int64(seed),
                                      C.c_int(1 if spherical else 0), _p(cen, _f32p)))
        return cen

    def _chk(self, rc):
        if rc != 0:
            raise RuntimeError(self.lib.ref_last_error().decode())

    def create(self, kind, metric, d, nlist=0, M=0, nbits=8):
        h = self.lib.ref_create(kind, metric, d, nlist, M, nbits)
        if not h:
            raise RuntimeError(self.lib.ref_last_error().decode())
        return C.c_void_p(h)

    def destroy(self, h):
        self.lib.ref_destroy(h)

    def train_add(self, h, xb, ids=None):
        xb = np.ascontiguousarray(xb, np.float32)
        self._chk(self.lib.ref_train(h, C.c_int64(xb.shape[0]), _p(xb, _f32p)))
        self._chk(self.lib.ref_add(h, C.c_int64(xb.shape[0]), _p(xb, _f32p), _p(ids, _i64p)))

    def export(self, h, kind, metric, d, nlist=0, M=0, nbits=8, base=None):
        """reference-trained index -> IndexData (plain arrays)"""
        ix = IndexData(kind, metric, d, nlist, M, nbits)
        if kind == FLAT:
            n = self.lib.ref_ntotal(h)
            ix.base = np.empty((n, d), np.float32)
            self._chk(self.lib.ref_get_flat_vectors(h, _p(ix.base, _f32p)))
            return ix
        ix.centroids = np.empty((nlist, d), np.float32)
        self._chk(self.lib.ref_get_centroids(h, _p(ix.centroids, _f32p)))
        cs = self.lib.ref_code_size(h)
        if kind == IVF_PQ:
            ksub = 1 << nbits
            ix.pq_centroids = np.empty((M, ksub, d // M), np.float32)
            self._chk(self.lib.ref_get_pq_centroids(h, _p(ix.pq_centroids, _f32p)))
            ix.use_precomputed_table = self.lib.ref_use_precomputed_table(h)
            n = self.lib.ref_get_precomputed_table(h, None)
            if n > 0:
                ix.precomputed_table = np.empty(n, np.float32)
                self.lib.ref_get_precomputed_table(h, _p(ix.precomputed_table, _f32p))
        if kind == IVF_SQ8:
            ix.sq_trained = np.empty(2 * d, np.float32)
            self._chk(self.lib.ref_get_sq_trained(h, _p(ix.sq_trained, _f32p)))
        for l in range(nlist):
            n = self.lib.ref_list_size(h, C.c_int64(l))
            c = np.empty((n, cs), np.uint8)
            i = np.empty(n, np.int64)
            if n:
                self._chk(self.lib.ref_get_list(h, C.c_int64(l), _p(c, _u8p), _p(i, _i64p)))
            ix.list_codes.append(c)
            ix.list_ids.append(i)
        return ix

    def from_data(self, ix):
        """IndexData -> reference index object (no training)"""
        h = self.create(ix.kind, ix.metric, ix.d, ix.nlist, ix.M, ix.nbits)
        if ix.kind == FLAT:
            self._chk(self.lib.ref_add(h, C.c_int64(ix.base.shape[0]), _p(ix.base, _f32p), None))
            return h
        self._chk(self.lib.ref_set_centroids(h, C.c_int64(ix.nlist), _p(ix.centroids, _f32p)))
        if ix.kind == IVF_PQ:
            self._chk(self.lib.ref_set_pq_centroids(h, _p(ix.pq_centroids, _f32p)))
        if ix.kind == IVF_SQ8:
            self._chk(self.lib.ref_set_sq_trained(h, _p(ix.sq_trained, _f32p)))
        for l in range(ix.nlist):
            c = np.ascontiguousarray(ix.list_codes[l], np.uint8)
            i = np.ascontiguousarray(ix.list_ids[l], np.int64)
            if i.size:
                self._chk(self.lib.ref_add_list_entries(h, C.c_int64(l), C.c_int64(i.size), _p(c, _u8p), _p(i, _i64p)))
        self._chk(self.lib.ref_finalize(h))
        return h

    def search(self, h, xq, k, nprobe=1, bitset=None, nbits=0, nthreads=1):
        xq = np.ascontiguousarray(xq, np.float32)
        nq = xq.shape[0]
        D = np.empty((nq, k), np.float32)
        I = np.empty((nq, k), np.int64)
        self._chk(self.lib.ref_search(h, C.c_int64(nq), _p(xq, _f32p), C.c_int64(k), C.c_int64(nprobe),
                                      _p(bitset, _u8p), C.c_int64(nbits), _p(D, _f32p), _p(I, _i64p),
                                      C.c_int(nthreads)))
        return D, I

    def serialize(self, h, xb_refine=None):
        """faiss::write_index bytes of the index (wrapped in IndexRefine(base, IndexFlat(xb_refine)) if given)"""
        nbr, pr = 0, None
        if xb_refine is not None:
            xb_refine = np.ascontiguousarray(xb_refine, np.float32)
            nbr, pr = xb_refine.shape[0], _p(xb_refine, _f32p)
        n = self.lib.ref_serialize(h, C.c_int64(nbr), pr, None, C.c_int64(0))
        if n < 0:
            raise RuntimeError(self.lib.ref_last_error().decode())
        out = np.empty(n, np.uint8)
        n2 = self.lib.ref_serialize(h, C.c_int64(nbr), pr, _p(out, _u8p), C.c_int64(n))
        assert n2 == n
        return out

    def deserialize(self, blob, d=None):
        """faiss::read_index; returns (handle of the base index, refine vectors or None)"""
        blob = np.ascontiguousarray(blob, np.uint8)
        nbr = C.c_int64(0)
        h = self.lib.ref_deserialize(_p(blob, _u8p), C.c_int64(blob.size), C.byref(nbr), None)
        if not h:
            raise RuntimeError(self.lib.ref_last_error().decode())
        h = C.c_void_p(h)
        if nbr.value == 0:
            return h, None
        self.lib.ref_destroy(h)
        assert d is not None, "pass d to receive the refine vectors"
        xb = np.empty((nbr.value, d), np.float32)
        h = C.c_void_p(self.lib.ref_deserialize(_p(blob, _u8p), C.c_int64(blob.size), C.byref(nbr), _p(xb, _f32p)))
        return h, xb

    def range_search(self, h, xq, radius, max_empty=2, bitset=None, nbits=0):
        xq = np.ascontiguousarray(xq, np.float32)
        nq = xq.shape[0]
        lims = np.zeros(nq + 1, np.int64)
        pi, pd = _i64p(), _f32p()
        self._chk(self.lib.ref_range_search(h, C.c_int64(nq), _p(xq, _f32p), C.c_float(radius), C.c_int64(max_empty),
                                            _p(bitset, _u8p), C.c_int64(nbits), _p(lims, _i64p), C.byref(pi),
                                            C.byref(pd)))
        n = int(lims[-1])
        ids = np.ctypeslib.as_array(pi, shape=(max(n, 1),))[:n].copy() if n else np.empty(0, np.int64)
        dis = np.ctypeslib.as_array(pd, shape=(max(n, 1),))[:n].copy() if n else np.empty(0, np.float32)
        self.lib.ref_free(pi)
        self.lib.ref_free(pd)
        return lims, ids, dis

    def search_refine(self, h, xb, xq, k, k_factor, nprobe):
        """IndexRefine(base = h, refine = IndexFlat(xb)).search, one query per call"""
        xb = np.ascontiguousarray(xb, np.float32)
        xq = np.ascontiguousarray(xq, np.float32)
        nq = xq.shape[0]
        D = np.empty((nq, k), np.float32)
        I = np.empty((nq, k), np.int64)
        self._chk(self.lib.ref_search_refine(h, C.c_int64(xb.shape[0]), _p(xb, _f32p), C.c_int64(nq), _p(xq, _f32p),
                                             C.c_int64(k), C.c_float(k_factor), C.c_int64(nprobe), _p(D, _f32p),
                                             _p(I, _i64p)))
        return D, I

    def sq_rows(self, row_type, metric, xb):
        """(codes, trained) of the IndexScalarQuantizer Knowhere builds as the refine index for fp16 / bf16 / sq8"""
        xb = np.ascontiguousarray(xb, np.float32)
        n, d = xb.shape
        codes = np.empty((n, rows_code_size(row_type, d)), np.uint8)
        tr = np.zeros(2 * d if row_type != 6 else 2, np.float32)
        self._chk(self.lib.ref_sq_rows(C.c_int(row_type), C.c_int(metric), C.c_int(d), C.c_int64(n), _p(xb, _f32p),
                                       _p(codes, _u8p), _p(tr, _f32p)))
        return codes, tr

    def search_refine_sq(self, h, row_type, xb, xq, k, k_factor, nprobe):
        """IndexRefine(base = h, refine = IndexScalarQuantizer(xb)).search, one query per call"""
        xb = np.ascontiguousarray(xb, np.float32)
        xq = np.ascontiguousarray(xq, np.float32)
        nq = xq.shape[0]
        D = np.empty((nq, k), np.float32)
        I = np.empty((nq, k), np.int64)
        self._chk(self.lib.ref_search_refine_sq(h, C.c_int(row_type), C.c_int64(xb.shape[0]), _p(xb, _f32p), C.c_int64(nq),
                                                _p(xq, _f32p), C.c_int64(k), C.c_float(k_factor), C.c_int64(nprobe),
                                                _p(D, _f32p), _p(I, _i64p)))
        return D, I

    def serialize_sq(self, h, row_type, xb):
        xb = np.ascontiguousarray(xb, np.float32)
        a = (h, C.c_int(row_type), C.c_int64(xb.shape[0]), _p(xb, _f32p))
        n = self.lib.ref_serialize_sq(*a, None, C.c_int64(0))
        if n < 0:
            raise RuntimeError(self.lib.ref_last_error().decode())
        out = np.empty(n, np.uint8)
        assert self.lib.ref_serialize_sq(*a, _p(out, _u8p), C.c_int64(n)) == n
        return out

    def blob_search_refine(self, blob, xq, k, k_factor, nprobe):
        """read_index(blob) must be an IndexRefine; search through it, one query per call"""
        blob = np.ascontiguousarray(blob, np.uint8)
        xq = np.ascontiguousarray(xq, np.float32)
        nq = xq.shape[0]
        D = np.empty((nq, k), np.float32)
        I = np.empty((nq, k), np.int64)
        self._chk(self.lib.ref_blob_search_refine(_p(blob, _u8p), C.c_int64(blob.size), C.c_int64(nq), _p(xq, _f32p),
                                                  C.c_int64(k), C.c_float(k_factor), C.c_int64(nprobe), _p(D, _f32p),
                                                  _p(I, _i64p)))
        return D, I

    def coarse(self, h, xq, nprobe):
        xq = np.ascontiguousarray(xq, np.float32)
        nq = xq.shape[0]
        D = np.empty((nq, nprobe), np.float32)
        I = np.empty((nq, nprobe), np.int64)
        self._chk(self.lib.ref_coarse(h, C.c_int64(nq), _p(xq, _f32p), C.c_int64(nprobe), _p(D, _f32p), _p(I, _i64p)))
        return D, I

    def fvec_L2sqr(self, x, y):
        return float(self.lib.ref_fvec_L2sqr(_p(x, _f32p), _p(y, _f32p), C.c_int64(x.size)))

    def fvec_inner_product(self, x, y):
        return float(self.lib.ref_fvec_inner_product(_p(x, _f32p), _p(y, _f32p), C.c_int64(x.size)))

    def fvec_norm_L2sqr(self, x):
        return float(self.lib.ref_fvec_norm_L2sqr(_p(x, _f32p), C.c_int64(x.size)))


class KRef:
    """oracle/_ref/libknowhere_kref.so -- the index classes Knowhere's nodes instantiate for COSINE (the reference's FAISS
    fork + src/common/utils.cc + the scalar hook table), driven as the nodes drive them (oracle/kref_driver.cpp)."""

    @staticmethod
    def path():
        return os.path.join(_HERE, "_ref", "libknowhere_kref.so")

    @staticmethod
    def available():
        if not os.path.exists(KRef.path()):
            return False
        try:
            KRef()
            return True
        except OSError:
            return False

    def __init__(self):
        os.environ.setdefault("MKL_THREADING_LAYER", "GNU")
        for mkl in ("/opt/conda/lib/libmkl_rt.so.1", "/opt/conda/lib/libmkl_rt.so"):
            if os.path.exists(mkl):
                C.CDLL(mkl, mode=C.RTLD_GLOBAL)
        # the fork and the baseline FAISS objects of libknowhere_ref*.so share symbol names: keep this copy private
        self.lib = L = C.CDLL(KRef.path(), mode=C.RTLD_LOCAL | getattr(os, "RTLD_DEEPBIND", 0))
        L.kref_last_error.restype = C.c_char_p
        L.kref_ivfflat_cosine_create.restype = C.c_void_p
        L.kref_ivfflat_cosine_list_size.restype = C.c_int64
        L.kref_ivfflat_cosine_serialize.restype = C.c_int64
        L.kref_flat_cosine_serialize.restype = C.c_int64

    def _chk(self, rc):
        if rc != 0:
            raise RuntimeError(self.lib.kref_last_error().decode())

    def normalize(self, x):
        x = np.array(x, np.float32, order="C", copy=True)
        norms = np.empty(x.shape[0], np.float32)
        self._chk(self.lib.kref_normalize(_p(x, _f32p), C.c_int64(x.shape[0]), C.c_int(x.shape[1]), _p(norms, _f32p)))
        return x, norms

    def flat_cosine_search(self, xb, xq, k, bitset=None):
        xb = np.ascontiguousarray(xb, np.float32)
        xq = np.ascontiguousarray(xq, np.float32)
        nb, d = xb.shape
        nq = xq.shape[0]
        D = np.empty((nq, k), np.float32)
        I = np.empty((nq, k), np.int64)
        inv = np.empty(nb, np.float32)
        self._chk(self.lib.kref_flat_cosine_search(C.c_int(d), C.c_int64(nb), _p(xb, _f32p), C.c_int64(nq), _p(xq, _f32p),
                                                   C.c_int64(k), _p(bitset, _u8p),
                                                   C.c_int64(0 if bitset is None else nb), _p(D, _f32p), _p(I, _i64p),
                                                   _p(inv, _f32p)))
        return D, I, inv

    def flat_cosine_blob(self, xb):
        xb = np.ascontiguousarray(xb, np.float32)
        nb, d = xb.shape
        cap = nb * d * 4 + nb * 4 + 4096
        out = np.empty(cap, np.uint8)
        n = self.lib.kref_flat_cosine_serialize(C.c_int(d), C.c_int64(nb), _p(xb, _f32p), _p(out, _u8p), C.c_int64(cap))
        assert 0 < n <= cap, self.lib.kref_last_error().decode()
        return out[:n].copy()

    # IVF_FLAT + COSINE
    def ivfflat_create(self, d, nlist):
        h = self.lib.kref_ivfflat_cosine_create(C.c_int(d), C.c_int64(nlist))
        if not h:
            raise RuntimeError(self.lib.kref_last_error().decode())
        return C.c_void_p(h)

    def ivfflat_destroy(self, h):
        self.lib.kref_ivfflat_cosine_destroy(h)

    def ivfflat_train(self, h, x, niter=0, seed=-1):
        x = np.ascontiguousarray(x, np.float32)
        self._chk(self.lib.kref_ivfflat_cosine_train(h, C.c_int64(x.shape[0]), _p(x, _f32p), C.c_int(niter), C.c_int(seed)))

    def ivfflat_centroids(self, h, d, nlist):
        out = np.empty((nlist, d), np.float32)
        self._chk(self.lib.kref_ivfflat_cosine_get_centroids(h, _p(out, _f32p)))
        return out

    def ivfflat_add(self, h, x):
        x = np.ascontiguousarray(x, np.float32)
        self._chk(self.lib.kref_ivfflat_cosine_add(h, C.c_int64(x.shape[0]), _p(x, _f32p)))

    def ivfflat_lists(self, h, d, nlist):
        codes, ids, norms = [], [], []
        for l in range(nlist):
            n = int(self.lib.kref_ivfflat_cosine_list_size(h, C.c_int64(l)))
            c = np.empty((n, d * 4), np.uint8)
            i = np.empty(n, np.int64)
            v = np.empty(n, np.float32)
            if n:
                self._chk(self.lib.kref_ivfflat_cosine_get_list(h, C.c_int64(l), _p(c, _u8p), _p(i, _i64p), _p(v, _f32p)))
            codes.append(c)
            ids.append(i)
            norms.append(v)
        return codes, ids, norms

    def ivfflat_search(self, h, xq, k, nprobe, bitset=None, nbits=0):
        xq = np.ascontiguousarray(xq, np.float32)
        nq = xq.shape[0]
        D = np.empty((nq, k), np.float32)
        I = np.empty((nq, k), np.int64)
        self._chk(self.lib.kref_ivfflat_cosine_search(h, C.c_int64(nq), _p(xq, _f32p), C.c_int64(k), C.c_int64(nprobe),
                                                      _p(bitset, _u8p), C.c_int64(nbits), _p(D, _f32p), _p(I, _i64p)))
        return D, I

    def ivfflat_blob(self, h, cap):
        out = np.empty(cap, np.uint8)
        n = self.lib.kref_ivfflat_cosine_serialize(h, _p(out, _u8p), C.c_int64(cap))
        assert 0 < n <= cap, self.lib.kref_last_error().decode()
        return out[:n].copy()
