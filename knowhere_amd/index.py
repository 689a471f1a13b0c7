"""knowhere_amd/index.py -- Python handle over a ``knhip_index`` (one device, one process).

Mirrors the C ABI one-to-one; numpy arrays cross the host boundary (``search``), torch tensors
the device boundary (``search_device``).  No arithmetic happens here.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import KnhipError, check  # noqa: F401

BRUTE_FORCE, IVF_FLAT, IVF_PQ, IVF_SQ8 = _lib.BRUTE_FORCE, _lib.IVF_FLAT, _lib.IVF_PQ, _lib.IVF_SQ8
L2, IP = _lib.L2, _lib.IP

KIND_NAMES = {"GPU_HIP_BRUTE_FORCE": BRUTE_FORCE, "GPU_HIP_IVF_FLAT": IVF_FLAT, "GPU_HIP_IVF_PQ": IVF_PQ,
              "GPU_HIP_IVF_SQ8": IVF_SQ8}


def _np_ptr(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


def _t_ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _bitset_nbits(bitset, nbits):
    """a bitset without a bit count covers all its bytes (the C side ignores a bitset whose nbits <= 0, which
    would silently return filtered ids)"""
    if bitset is None:
        return 0
    n = int(bitset.numel() if hasattr(bitset, "numel") else bitset.size)
    if nbits is None or nbits <= 0:
        return 8 * n
    if nbits > 8 * n:
        raise ValueError(f"bitset of {n} bytes cannot hold {nbits} bits")
    return int(nbits)


class GpuIndex:
    def __init__(self, kind, metric, dim, nlist=0, pq_m=0, pq_nbits=8, device=0,
                 precomputed_table_max_bytes=0):
        self.L = _lib.load()
        self.kind, self.metric, self.dim, self.nlist, self.pq_m = kind, metric, dim, nlist, pq_m
        self.pq_nbits = pq_nbits
        self.device = device
        d = _lib.Desc(kind, metric, dim, device, nlist, pq_m, pq_nbits, precomputed_table_max_bytes)
        h = C.c_void_p()
        check(self.L.knhip_index_create(C.byref(d), C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.knhip_index_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- contents (host) ----
    def set_coarse(self, centroids):
        c = np.ascontiguousarray(centroids, np.float32)
        assert c.shape == (self.nlist, self.dim)
        check(self.L.knhip_index_set_coarse(self.h, _np_ptr(c)))

    def set_pq(self, codebooks):
        c = np.ascontiguousarray(codebooks, np.float32)
        assert c.size == (1 << self.pq_nbits) * self.dim  # [pq_m][2^nbits][dim / pq_m]
        check(self.L.knhip_index_set_pq(self.h, _np_ptr(c)))

    def set_sq(self, vmin, vdiff):
        a = np.ascontiguousarray(vmin, np.float32)
        b = np.ascontiguousarray(vdiff, np.float32)
        check(self.L.knhip_index_set_sq(self.h, _np_ptr(a), _np_ptr(b)))

    def set_row_scale(self, scale, mode):
        """COSINE with stored norms (knhip_index_set_row_scale): one float per entry in canonical order; mode 1 = ip / norm
        (IVF_FLAT), 2 = clamp(ip * inverse norm) (FLAT); None switches it off"""
        if scale is None:
            check(self.L.knhip_index_set_row_scale(self.h, None, 0))
            return
        a = np.ascontiguousarray(scale, np.float32)
        check(self.L.knhip_index_set_row_scale(self.h, _np_ptr(a), mode))

    def add_lists(self, list_codes, list_ids):
        nlist = self.nlist
        sizes = np.array([len(i) for i in list_ids], np.int64)
        keep = []
        cp = (C.c_void_p * nlist)()
        ip = (C.c_void_p * nlist)()
        for l in range(nlist):
            c = np.ascontiguousarray(list_codes[l], np.uint8)
            i = np.ascontiguousarray(list_ids[l], np.int64)
            keep += [c, i]
            cp[l] = c.ctypes.data if i.size else None
            ip[l] = i.ctypes.data if i.size else None
        check(self.L.knhip_index_add_lists(self.h, _np_ptr(sizes), C.cast(cp, C.c_void_p), C.cast(ip, C.c_void_p)))

    def add_vectors(self, x, id_offset=0):
        x = np.ascontiguousarray(x, np.float32)
        check(self.L.knhip_index_add_vectors(self.h, x.shape[0], _np_ptr(x), None, id_offset))

    @classmethod
    def from_data(cls, ix, device=0, precomputed_table_max_bytes=0):
        """ix: any object with the fields of oracle.binding.IndexData (duck typed; the product does
        not import the oracle)."""
        kind = {0: BRUTE_FORCE, 1: IVF_FLAT, 2: IVF_PQ, 3: IVF_SQ8}[ix.kind]
        g = cls(kind, ix.metric, ix.d, ix.nlist, ix.M, ix.nbits, device, precomputed_table_max_bytes)
        if kind == BRUTE_FORCE:
            g.add_vectors(ix.base)
            return g
        g.set_coarse(ix.centroids)
        if kind == IVF_PQ:
            g.set_pq(ix.pq_centroids)
        if kind == IVF_SQ8:
            g.set_sq(ix.sq_trained[:ix.d], ix.sq_trained[ix.d:])
        g.add_lists(ix.list_codes, ix.list_ids)
        norms = getattr(ix, "list_norms", None)
        if kind == IVF_FLAT and norms:  # COSINE with stored norms (IndexIVFFlatCosine): dis = ip / norm
            g.set_row_scale(np.concatenate([np.asarray(n, np.float32) for n in norms]), 1)
        return g

    # ---- contents (device tensors, GPU build path) ----
    def set_coarse_device(self, centroids_t):
        assert centroids_t.is_cuda and centroids_t.is_contiguous()
        check(self.L.knhip_index_set_coarse_device(self.h, _t_ptr(centroids_t)))

    def set_lists_device(self, list_offsets, codes_t, ids_t):
        off = np.ascontiguousarray(list_offsets, np.int64)
        assert off.size == self.nlist + 1
        check(self.L.knhip_index_set_lists_device(self.h, _np_ptr(off), _t_ptr(codes_t), _t_ptr(ids_t)))

    def add_vectors_device(self, x_t, id_offset=0):
        assert x_t.is_cuda and x_t.is_contiguous()
        check(self.L.knhip_index_add_vectors_device(self.h, x_t.shape[0], _t_ptr(x_t), None, id_offset))

    # ---- GPU build: Train / Add on the device (knhip_index_train* / knhip_index_add*) ----
    @staticmethod
    def _tp(niter, max_points, seed):
        if niter is None and max_points is None and seed is None:
            return None
        return C.byref(_lib.TrainParams(niter or 0, max_points or 0, seed or 0))

    def train(self, x, niter=None, max_points=None, seed=None):
        """x: host numpy [n, dim] or a device tensor"""
        if hasattr(x, "data_ptr"):
            assert x.is_cuda and x.is_contiguous()
            check(self.L.knhip_index_train_device(self.h, x.shape[0], _t_ptr(x), self._tp(niter, max_points, seed)))
        else:
            x = np.ascontiguousarray(x, np.float32)
            check(self.L.knhip_index_train(self.h, x.shape[0], _np_ptr(x), self._tp(niter, max_points, seed)))

    def add(self, x, ids=None):
        if hasattr(x, "data_ptr"):
            assert x.is_cuda and x.is_contiguous()
            check(self.L.knhip_index_add_device(self.h, x.shape[0], _t_ptr(x), _t_ptr(ids)))
        else:
            x = np.ascontiguousarray(x, np.float32)
            i = None if ids is None else np.ascontiguousarray(ids, np.int64)
            check(self.L.knhip_index_add(self.h, x.shape[0], _np_ptr(x), _np_ptr(i)))

    def encode_device(self, x_t):
        """-> (assign int64 [n], codes uint8 [n, code_size]) device tensors"""
        import torch
        n = x_t.shape[0]
        cs = {IVF_FLAT: 4 * self.dim, IVF_PQ: self.pq_m, IVF_SQ8: self.dim}[self.kind]
        a = torch.empty(n, dtype=torch.int64, device=x_t.device)
        c = torch.empty((n, cs), dtype=torch.uint8, device=x_t.device)
        s = torch.cuda.current_stream(x_t.device).cuda_stream
        check(self.L.knhip_index_encode_device(self.h, n, _t_ptr(x_t), _t_ptr(a), _t_ptr(c), C.c_void_p(s)))
        return a, c

    def get_coarse(self):
        out = np.empty((self.nlist, self.dim), np.float32)
        check(self.L.knhip_index_get_coarse(self.h, _np_ptr(out)))
        return out

    def get_pq(self):
        out = np.empty((self.pq_m, 1 << self.pq_nbits, self.dim // self.pq_m), np.float32)
        check(self.L.knhip_index_get_pq(self.h, _np_ptr(out)))
        return out

    def get_sq(self):
        out = np.empty(2 * self.dim, np.float32)
        check(self.L.knhip_index_get_sq(self.h, _np_ptr(out), C.c_void_p(out.ctypes.data + 4 * self.dim)))
        return out

    def get_lists(self):
        """-> (sizes [nlist], codes [count, code_size] uint8, ids [count]) list after list"""
        sizes = np.zeros(self.nlist, np.int64)
        check(self.L.knhip_index_get_list_sizes(self.h, _np_ptr(sizes)))
        n = int(sizes.sum())
        # (IVF_PQ: the reference's code bytes -- pq_m indices of nbits bits as a little-endian bit string; the DEVICE side
        # entry points -- encode_device, set_lists_device -- speak one byte per sub-quantizer)
        cs = {IVF_FLAT: 4 * self.dim, IVF_PQ: (self.pq_m * self.pq_nbits + 7) // 8, IVF_SQ8: self.dim}[self.kind]
        codes = np.empty((n, cs), np.uint8)
        ids = np.empty(n, np.int64)
        check(self.L.knhip_index_get_lists(self.h, _np_ptr(codes), _np_ptr(ids)))
        return sizes, codes, ids

    # ---- info ----
    @property
    def count(self):
        return int(self.L.knhip_index_count(self.h))

    @property
    def device_bytes(self):
        return int(self.L.knhip_index_device_bytes(self.h))

    @property
    def uses_precomputed_table(self):
        return int(self.L.knhip_index_uses_precomputed_table(self.h))

    # ---- search ----
    def search(self, xq, k, nprobe=1, bitset=None, nbits=0):
        xq = np.ascontiguousarray(xq, np.float32)
        nq = xq.shape[0]
        D = np.empty((nq, k), np.float32)
        I = np.empty((nq, k), np.int64)
        bs = None if bitset is None else np.ascontiguousarray(bitset, np.uint8)
        nbits = _bitset_nbits(bs, nbits)
        check(self.L.knhip_search(self.h, _np_ptr(xq), nq, k, nprobe, _np_ptr(bs), nbits, _np_ptr(I), _np_ptr(D)))
        return D, I

    def search_refine(self, raw, xq, k, k_base, nprobe=1, bitset=None, nbits=0):
        """knhip_search_refine across the HOST boundary: k_base candidates from this index, re-ranked exactly against the
        rows of `raw` (a BRUTE_FORCE GpuIndex on the same device), k best returned -- what the node's Search() calls"""
        xq = np.ascontiguousarray(xq, np.float32)
        nq = xq.shape[0]
        D = np.empty((nq, k), np.float32)
        I = np.empty((nq, k), np.int64)
        bs = None if bitset is None else np.ascontiguousarray(bitset, np.uint8)
        nbits = _bitset_nbits(bs, nbits)
        check(self.L.knhip_search_refine(self.h, raw.h, _np_ptr(xq), nq, k, k_base, nprobe, _np_ptr(bs), nbits,
                                         _np_ptr(I), _np_ptr(D)))
        return D, I

    def search_refine_rows(self, rows, xq, k, k_base, nprobe=1, bitset=None, nbits=0):
        """knhip_search_refine_rows: as search_refine with the second stage reading a quantised RowStore (refine_type =
        fp16 / bf16 / sq8)"""
        xq = np.ascontiguousarray(xq, np.float32)
        nq = xq.shape[0]
        D = np.empty((nq, k), np.float32)
        I = np.empty((nq, k), np.int64)
        bs = None if bitset is None else np.ascontiguousarray(bitset, np.uint8)
        nbits = _bitset_nbits(bs, nbits)
        check(self.L.knhip_search_refine_rows(self.h, rows.h, _np_ptr(xq), nq, k, k_base, nprobe, _np_ptr(bs), nbits,
                                              _np_ptr(I), _np_ptr(D)))
        return D, I

    def get_vectors(self, ids):
        """stored fp32 rows by id (knhip_index_get_vectors: BRUTE_FORCE, or IVF_FLAT through its direct map)"""
        ids = np.ascontiguousarray(ids, np.int64)
        out = np.empty((len(ids), self.dim), np.float32)
        check(self.L.knhip_index_get_vectors(self.h, len(ids), _np_ptr(ids), _np_ptr(out)))
        return out

    def last_range_ranks(self):
        """coarse ranks per query the last range_search scanned (rank waves: knhip_index_last_range_ranks)"""
        return int(self.L.knhip_index_last_range_ranks(self.h))

    def range_search(self, xq, radius, max_empty_result_buckets=2, bitset=None, nbits=0):
        """-> (lims[nq + 1], ids, distances) in the reference's emission order (include/knhip.h)"""
        xq = np.ascontiguousarray(xq, np.float32)
        nq = xq.shape[0]
        lims = np.zeros(nq + 1, np.int64)
        pi, pd = C.POINTER(C.c_int64)(), C.POINTER(C.c_float)()
        bs = None if bitset is None else np.ascontiguousarray(bitset, np.uint8)
        nbits = _bitset_nbits(bs, nbits)
        check(self.L.knhip_range_search(self.h, _np_ptr(xq), nq, float(radius), int(max_empty_result_buckets),
                                        _np_ptr(bs), nbits, _np_ptr(lims), C.byref(pi), C.byref(pd)))
        n = int(lims[-1])
        ids = np.ctypeslib.as_array(pi, shape=(n,)).copy() if n else np.empty(0, np.int64)
        dis = np.ctypeslib.as_array(pd, shape=(n,)).copy() if n else np.empty(0, np.float32)
        self.L.knhip_free(pi)  # (NULL when there were no queries)
        self.L.knhip_free(pd)
        return lims, ids, dis

    def range_search_ranked(self, xq, radius, bitset=None, nbits=0):
        """knhip_range_search_ranked -> (lims, ids, distances, counts[nq][nlist]): every list visited, hits per coarse rank"""
        xq = np.ascontiguousarray(xq, np.float32)
        nq = xq.shape[0]
        lims = np.zeros(nq + 1, np.int64)
        pi, pd, pc = C.POINTER(C.c_int64)(), C.POINTER(C.c_float)(), C.POINTER(C.c_int32)()
        bs = None if bitset is None else np.ascontiguousarray(bitset, np.uint8)
        nbits = _bitset_nbits(bs, nbits)
        check(self.L.knhip_range_search_ranked(self.h, _np_ptr(xq), nq, float(radius), _np_ptr(bs), nbits, _np_ptr(lims),
                                               C.byref(pi), C.byref(pd), C.byref(pc)))
        n = int(lims[-1])
        ids = np.ctypeslib.as_array(pi, shape=(n,)).copy() if n else np.empty(0, np.int64)
        dis = np.ctypeslib.as_array(pd, shape=(n,)).copy() if n else np.empty(0, np.float32)
        cnt = np.ctypeslib.as_array(pc, shape=(nq * self.nlist,)).copy().reshape(nq, self.nlist) if nq else \
            np.empty((0, self.nlist), np.int32)
        for p in (pi, pd, pc):
            self.L.knhip_free(p)
        return lims, ids, dis, cnt

    def search_device(self, xq_t, k, nprobe=1, bitset_t=None, nbits=0, out=None, stream=None):
        import torch
        nq = xq_t.shape[0]
        if out is None:
            D = torch.empty((nq, k), dtype=torch.float32, device=xq_t.device)
            I = torch.empty((nq, k), dtype=torch.int64, device=xq_t.device)
        else:
            D, I = out
        s = torch.cuda.current_stream(xq_t.device).cuda_stream if stream is None else stream
        nbits = _bitset_nbits(bitset_t, nbits)
        check(self.L.knhip_search_device(self.h, _t_ptr(xq_t), nq, k, nprobe, _t_ptr(bitset_t), nbits,
                                         _t_ptr(I), _t_ptr(D), C.c_void_p(s)))
        return D, I

    def search_preassigned_device(self, xq_t, k, keys_t, cdis_t, bitset_t=None, nbits=0, stream=None):
        """search with a given coarse assignment (keys_t / cdis_t: [nq][nprobe] from coarse_search_device)"""
        import torch
        nq, nprobe = keys_t.shape
        assert xq_t.shape[0] == nq and keys_t.is_contiguous() and cdis_t.is_contiguous()
        D = torch.empty((nq, k), dtype=torch.float32, device=xq_t.device)
        I = torch.empty((nq, k), dtype=torch.int64, device=xq_t.device)
        s = torch.cuda.current_stream(xq_t.device).cuda_stream if stream is None else stream
        nbits = _bitset_nbits(bitset_t, nbits)
        check(self.L.knhip_search_preassigned_device(self.h, _t_ptr(xq_t), nq, k, nprobe, _t_ptr(keys_t), _t_ptr(cdis_t),
                                                     _t_ptr(bitset_t), nbits, _t_ptr(I), _t_ptr(D), C.c_void_p(s)))
        return D, I

    def search_canonical_device(self, xq_t, k, nprobe, keys_t=None, cdis_t=None, bitset_t=None, nbits=0, stream=None):
        """canonical top-k, no tie rule (knhip_search_canonical_device): what a shard contributes to a list-sharded
        search; keys_t / cdis_t: the coarse assignment [nq][nprobe] (None: assigned inside / BRUTE_FORCE)"""
        import torch
        nq = xq_t.shape[0]
        D = torch.empty((nq, k), dtype=torch.float32, device=xq_t.device)
        I = torch.empty((nq, k), dtype=torch.int64, device=xq_t.device)
        s = torch.cuda.current_stream(xq_t.device).cuda_stream if stream is None else stream
        nbits = _bitset_nbits(bitset_t, nbits)
        check(self.L.knhip_search_canonical_device(self.h, _t_ptr(xq_t), nq, k, nprobe, _t_ptr(keys_t), _t_ptr(cdis_t),
                                                   _t_ptr(bitset_t), nbits, _t_ptr(I), _t_ptr(D), C.c_void_p(s)))
        return D, I

    def tie_arrivals_device(self, xq_t, flagged_t, can_d_t, k, nprobe, keys_t=None, cdis_t=None, bitset_t=None, nbits=0,
                            key_base=0, stream=None):
        """this index's first k arrivals at or below the k-th distance of the flagged queries (knhip_tie_arrivals_device):
        -> (arr_d [nflag, k], arr_i, arr_key, arr_n [nflag])"""
        import torch
        nflag = int(flagged_t.numel())
        dev = xq_t.device
        arr_d = torch.zeros((nflag, k), dtype=torch.float32, device=dev)
        arr_i = torch.full((nflag, k), -1, dtype=torch.int64, device=dev)
        arr_key = torch.zeros((nflag, k), dtype=torch.int64, device=dev)
        arr_n = torch.zeros((nflag,), dtype=torch.int64, device=dev)
        s = torch.cuda.current_stream(dev).cuda_stream if stream is None else stream
        nbits = _bitset_nbits(bitset_t, nbits)
        check(self.L.knhip_tie_arrivals_device(self.h, _t_ptr(xq_t), _t_ptr(flagged_t), nflag, _t_ptr(can_d_t), k, nprobe,
                                               _t_ptr(keys_t), _t_ptr(cdis_t), _t_ptr(bitset_t), nbits, C.c_int64(key_base),
                                               _t_ptr(arr_d), _t_ptr(arr_i), _t_ptr(arr_key), _t_ptr(arr_n), C.c_void_p(s)))
        return arr_d, arr_i, arr_key, arr_n

    def coarse_search_device(self, xq_t, nprobe, stream=None):
        import torch
        nq = xq_t.shape[0]
        D = torch.empty((nq, nprobe), dtype=torch.float32, device=xq_t.device)
        I = torch.empty((nq, nprobe), dtype=torch.int64, device=xq_t.device)
        s = torch.cuda.current_stream(xq_t.device).cuda_stream if stream is None else stream
        check(self.L.knhip_coarse_search_device(self.h, _t_ptr(xq_t), nq, nprobe, _t_ptr(I), _t_ptr(D), C.c_void_p(s)))
        return D, I

    # ---- profiling ----
    def profile_enable(self, on=True):
        check(self.L.knhip_profile_enable(self.h, 1 if on else 0))

    def profile_reset(self):
        check(self.L.knhip_profile_reset(self.h))

    def profile_get(self):
        st = _lib.StageTimes()
        check(self.L.knhip_profile_get(self.h, C.byref(st)))
        return {"ms": list(st.ms), "launches": list(st.launches), "scan_bytes": st.scan_bytes,
                "coarse_flops": st.coarse_flops, "scan_items": st.scan_items,
                "coarse_fallback_queries": st.coarse_fallback_queries, "scan_bytes_rank0": st.scan_bytes_rank0,
                "mscan_queries": st.mscan_queries, "mscan_overflow_queries": st.mscan_overflow_queries,
                "mscan_candidates": st.mscan_candidates, "mscan_stream_bytes": st.mscan_stream_bytes,
                "mscan_recomputed": st.mscan_recomputed, "pq_filter_form": st.pq_filter_form,
                "tie_queries": st.tie_queries, "tie_anomalies": st.tie_anomalies}


def kmeans_device(metric, x_t, k, niter=None, max_points=None, seed=None, spherical=False):
    """faiss Clustering restated on the device (knhip_kmeans_device): x_t [n, d] device tensor -> centroids [k, d]"""
    import torch
    L = _lib.load()
    cen = torch.empty((k, x_t.shape[1]), dtype=torch.float32, device=x_t.device)
    tp = None if (niter is None and max_points is None and seed is None and not spherical) else C.byref(
        _lib.TrainParams(niter or 0, max_points or 0, seed or 0, 1 if spherical else 0, 0))
    check(L.knhip_kmeans_device(metric, x_t.shape[1], x_t.shape[0], _t_ptr(x_t), k, tp, _t_ptr(cen),
                                x_t.device.index or 0))
    return cen


def merge_topk_host(metric, D_parts, I_parts):
    """[nshard, nq, k] partial results -> [nq, k] (host path of the shard merge)."""
    L = _lib.load()
    Dp = np.ascontiguousarray(D_parts, np.float32)
    Ip = np.ascontiguousarray(I_parts, np.int64)
    nshard, nq, k = Dp.shape
    D = np.empty((nq, k), np.float32)
    I = np.empty((nq, k), np.int64)
    check(L.knhip_merge_topk_host(metric, nq, k, nshard, _np_ptr(Dp), _np_ptr(Ip), _np_ptr(D), _np_ptr(I)))
    return D, I


def merge_topk_device(metric, D_parts_t, I_parts_t, stream=None):
    import torch
    L = _lib.load()
    nshard, nq, k = D_parts_t.shape
    D = torch.empty((nq, k), dtype=torch.float32, device=D_parts_t.device)
    I = torch.empty((nq, k), dtype=torch.int64, device=D_parts_t.device)
    s = torch.cuda.current_stream(D_parts_t.device).cuda_stream if stream is None else stream
    check(L.knhip_merge_topk_device(metric, nq, k, nshard, _t_ptr(D_parts_t), _t_ptr(I_parts_t), _t_ptr(D),
                                    _t_ptr(I), C.c_void_p(s)))
    return D, I


def tie_flag(can_d_t, can_i_t, k):
    """rows of k + 1 canonical results -> (D [nq, k], I [nq, k], flagged: ascending int32 query numbers whose (k + 1)-th
    entry ties with the k-th).  Device tensors: knhip_tie_flag_device (one 4-byte read-back); CPU tensors: the host form."""
    import torch
    L = _lib.load()
    nq = can_d_t.shape[0]
    assert can_d_t.shape[1] == k + 1 and can_d_t.is_contiguous() and can_i_t.is_contiguous()
    D = torch.empty((nq, k), dtype=torch.float32, device=can_d_t.device)
    I = torch.empty((nq, k), dtype=torch.int64, device=can_d_t.device)
    if can_d_t.is_cuda:
        fl = torch.empty((2 * nq + 1,), dtype=torch.int32, device=can_d_t.device)
        n = C.c_int32(0)
        s = torch.cuda.current_stream(can_d_t.device).cuda_stream
        check(L.knhip_tie_flag_device(_t_ptr(can_d_t), _t_ptr(can_i_t), nq, k, _t_ptr(D), _t_ptr(I), _t_ptr(fl), C.byref(n),
                                      C.c_void_p(s)))
        return D, I, fl[:n.value].contiguous()
    fl = torch.zeros((nq,), dtype=torch.uint8)
    check(L.knhip_tie_flag_host(_t_ptr(can_d_t), _t_ptr(can_i_t), nq, k, _t_ptr(D), _t_ptr(I), _t_ptr(fl)))
    return D, I, torch.nonzero(fl)[:, 0].to(torch.int32).contiguous()


def tie_resolve(metric, flagged_t, k, can_d_t, can_i_t, arr_d_t, arr_i_t, arr_key_t, arr_n_t, D_t, I_t):
    """the reference's admission rule over ALL shards' arrivals ([nshards, nflag, k] / [nshards, nflag]) written over the
    flagged rows of (D_t, I_t) in place"""
    import torch
    L = _lib.load()
    nsh, nflag = arr_n_t.shape
    if nflag == 0:
        return D_t, I_t
    if can_d_t.is_cuda:
        s = torch.cuda.current_stream(can_d_t.device).cuda_stream
        check(L.knhip_tie_resolve_device(metric, nsh, _t_ptr(flagged_t), nflag, k, _t_ptr(can_d_t), _t_ptr(can_i_t),
                                         _t_ptr(arr_d_t.contiguous()), _t_ptr(arr_i_t.contiguous()),
                                         _t_ptr(arr_key_t.contiguous()), _t_ptr(arr_n_t.contiguous()), _t_ptr(D_t), _t_ptr(I_t),
                                         C.c_void_p(s)))
        return D_t, I_t
    # host form: arrival arrays indexed by the query
    nq = can_d_t.shape[0]
    fl = torch.zeros((nq,), dtype=torch.uint8)
    idx = flagged_t.long()
    fl[idx] = 1
    fd = torch.zeros((nsh, nq, k), dtype=torch.float32)
    fi = torch.full((nsh, nq, k), -1, dtype=torch.int64)
    fk = torch.zeros((nsh, nq, k), dtype=torch.int64)
    fn = torch.zeros((nsh, nq), dtype=torch.int64)
    fd[:, idx], fi[:, idx], fk[:, idx], fn[:, idx] = arr_d_t, arr_i_t, arr_key_t, arr_n_t
    check(L.knhip_tie_resolve_host(metric, nsh, nq, k, _t_ptr(fl), _t_ptr(can_d_t), _t_ptr(can_i_t), _t_ptr(fd), _t_ptr(fi),
                                   _t_ptr(fk), _t_ptr(fn), _t_ptr(D_t), _t_ptr(I_t)))
    return D_t, I_t


def refine_distances_device(metric, base_t, xq_t, cand_ids_t, id_base=0, stream=None):
    """distances of the candidates whose rows live in base_t (row r = id id_base + r); the all-ones pattern elsewhere
    (knhip_refine_distances_device): a shard's share of a sharded refine"""
    import torch
    L = _lib.load()
    nq, kbase = cand_ids_t.shape
    D = torch.empty((nq, kbase), dtype=torch.float32, device=xq_t.device)
    s = torch.cuda.current_stream(xq_t.device).cuda_stream if stream is None else stream
    check(L.knhip_refine_distances_device(metric, xq_t.shape[1], _t_ptr(base_t), base_t.shape[0], id_base, _t_ptr(xq_t), nq,
                                          _t_ptr(cand_ids_t), kbase, _t_ptr(D), C.c_void_p(s)))
    return D


def refine_select_device(metric, cand_ids_t, dist_parts_t, k, stream=None):
    """the shards' distance arrays [nshards, nq, k_base] combined (every candidate is held once) and the single index's
    selection -- tie rule in candidate order included -- run on them"""
    import torch
    L = _lib.load()
    nsh, nq, kbase = dist_parts_t.shape
    dev = cand_ids_t.device
    if not cand_ids_t.is_cuda:  # results combined on the CPU (gloo): the host form of the same selection
        bits = dist_parts_t.contiguous().view(torch.int32)
        held = bits != -1
        first = held.to(torch.int8).argmax(dim=0, keepdim=True)
        dist = torch.where(held.any(dim=0), torch.gather(bits, 0, first)[0], torch.full((nq, kbase), -1, dtype=torch.int32))
        dist = dist.contiguous().view(torch.float32)
        D = torch.empty((nq, k), dtype=torch.float32)
        I = torch.empty((nq, k), dtype=torch.int64)
        check(L.knhip_refine_select_host(metric, nq, _t_ptr(cand_ids_t), _t_ptr(dist), kbase, k, _t_ptr(D), _t_ptr(I)))
        return D, I
    dist = torch.empty((nq, kbase), dtype=torch.float32, device=dev)
    D = torch.empty((nq, k), dtype=torch.float32, device=dev)
    I = torch.empty((nq, k), dtype=torch.int64, device=dev)
    s = torch.cuda.current_stream(dev).cuda_stream if stream is None else stream
    check(L.knhip_refine_combine_device(nsh, nq * kbase, _t_ptr(dist_parts_t.contiguous()), _t_ptr(dist), C.c_void_p(s)))
    check(L.knhip_refine_select_device(metric, nq, _t_ptr(cand_ids_t), _t_ptr(dist), kbase, k, _t_ptr(D), _t_ptr(I),
                                       C.c_void_p(s)))
    return D, I


def refine_device(metric, base_t, xq_t, cand_ids_t, k, id_base=0, stream=None):
    """exact re-rank (IndexRefine second stage): base_t [nbase, d] fp32 device, cand_ids_t [nq, k_base]"""
    import torch
    L = _lib.load()
    nq, kbase = cand_ids_t.shape
    D = torch.empty((nq, k), dtype=torch.float32, device=xq_t.device)
    I = torch.empty((nq, k), dtype=torch.int64, device=xq_t.device)
    s = torch.cuda.current_stream(xq_t.device).cuda_stream if stream is None else stream
    check(L.knhip_refine_device(metric, base_t.shape[1], _t_ptr(base_t), base_t.shape[0], id_base, _t_ptr(xq_t), nq,
                                _t_ptr(cand_ids_t), kbase, k, _t_ptr(D), _t_ptr(I), C.c_void_p(s)))
    return D, I


ROWS_FP16, ROWS_BF16, ROWS_SQ8, ROWS_SQ6, ROWS_INT8, ROWS_SQ4U = 1, 2, 3, 4, 5, 6


class RowStore:
    """knhip_rows: the encoded raw vectors a quantised refine reads (include/knhip.h; the reference's
    faiss::IndexScalarQuantizer refine index, src/index/refine/refine_utils.cc:150-185).  Row r = vector id r."""

    def __init__(self, row_type, dim, device=0):
        self.L = _lib.load()
        self.row_type, self.dim = row_type, dim
        h = C.c_void_p()
        check(self.L.knhip_rows_create(device, dim, row_type, C.byref(h)))
        self.h = h

    def close(self):
        if self.h:
            self.L.knhip_rows_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def train(self, x):
        x = np.ascontiguousarray(x, np.float32)
        check(self.L.knhip_rows_train(self.h, x.shape[0], _np_ptr(x)))

    def train_uniform(self, x, rangestat=0, rangestat_arg=0.0):
        """sq4u: the one range of all dimensions; rangestat 0 = min / max, 2 = quantiles (Knowhere: 2 with 0.01 for L2)"""
        x = np.ascontiguousarray(x, np.float32)
        check(self.L.knhip_rows_train_uniform(self.h, x.shape[0], _np_ptr(x), rangestat, C.c_float(rangestat_arg)))

    def _nrange(self):
        return 1 if self.row_type == ROWS_SQ4U else self.dim

    def set_trained(self, trained):
        t = np.ascontiguousarray(trained, np.float32)
        n = self._nrange()
        check(self.L.knhip_rows_set_trained(self.h, _np_ptr(np.ascontiguousarray(t[:n])), _np_ptr(np.ascontiguousarray(t[n:]))))

    def trained(self):
        n = self._nrange()
        lo, hi = np.empty(n, np.float32), np.empty(n, np.float32)
        check(self.L.knhip_rows_get_trained(self.h, _np_ptr(lo), _np_ptr(hi)))
        return np.concatenate([lo, hi])

    def add(self, x):
        x = np.ascontiguousarray(x, np.float32)
        check(self.L.knhip_rows_add(self.h, x.shape[0], _np_ptr(x)))

    def add_codes(self, codes):
        c = np.ascontiguousarray(codes, np.uint8)
        check(self.L.knhip_rows_add_codes(self.h, c.shape[0], _np_ptr(c)))

    def count(self):
        return int(self.L.knhip_rows_count(self.h))

    def codes(self):
        out = np.empty((self.count(), int(self.L.knhip_rows_code_size(self.h))), np.uint8)
        check(self.L.knhip_rows_get_codes(self.h, _np_ptr(out)))
        return out
