"""knowhere_amd/build.py -- GPU index builder used by bench.py and the scale tests.

Build-side counterpart of the reference's Train/Add (reference src/index/ivf/ivf.cc:547-844,
thirdparty/faiss/faiss/IndexIVF.cpp:55-121 train_q1, 212-287 add_core; Clustering.h:24-77).  The
arithmetic -- k-means, PQ / SQ8 training, nearest-centroid assignment, residual encoding -- runs in the product's
HIP kernels through the C ABI (knhip_index_train_device / knhip_index_encode_device, csrc/build.hip: the
reference's Clustering / IndexIVF::train / add_core restated, bit-equal to the scalar reference).  What stays here is
plumbing: the synthetic data generator, chunking, the owner filter of the list-sharded build and the final grouping
of the encoded rows by list (torch.sort).  It produces exactly the objects the C ABI ingests, so the oracle can be
handed the same index BYTES.

Synthetic data: a counter-style generator keyed by (seed, chunk) so any rank can
regenerate any slice without storing the 51 GB of raw vectors.
  "mixture": ncenter Gaussian centres, points = centre + sigma * N(0, I)  (recall-meaningful)
  "uniform": uniform [0, 100), the reference's own test fixture (tests/ut/utils.h:41-50)
"""
import math

import numpy as np
import torch

CHUNK = 1 << 20  # rows per generation chunk (the unit of the counter-based generator)


class DataSpec:
    def __init__(self, n, d, kind="mixture", seed=42, ncenter=4096, sigma=0.35, center_seed=7, latent=0,
                 noise=0.02):
        """latent > 0: within-component variation lives in a `latent`-dimensional subspace (shared
        random basis) plus `noise` * sigma isotropic jitter -- low intrinsic dimension, like real
        descriptor data (SIFT ~ 12-16); latent = 0: isotropic d-dimensional Gaussians."""
        self.n, self.d, self.kind, self.seed = n, d, kind, seed
        self.ncenter, self.sigma, self.center_seed = ncenter, sigma, center_seed
        self.latent, self.noise = latent, noise
        self._centers = {}
        self._basis = {}

    def basis(self, device):
        key = str(device)
        if key not in self._basis:
            g = torch.Generator(device="cpu").manual_seed(self.center_seed + 1)
            a = torch.randn((self.latent, self.d), generator=g)
            q, _ = torch.linalg.qr(a.t())  # orthonormal columns [d, latent]
            self._basis[key] = (q.t() * math.sqrt(self.d / self.latent)).contiguous().to(device)
        return self._basis[key]

    def _component(self, m, device, g):
        if self.latent > 0:
            z = torch.randn((m, self.latent), device=device, generator=g) * self.sigma
            x = z @ self.basis(device)
            x += torch.randn((m, self.d), device=device, generator=g) * (self.sigma * self.noise)
            return x
        return torch.randn((m, self.d), device=device, generator=g) * self.sigma

    def centers(self, device):
        key = str(device)
        if key not in self._centers:
            g = torch.Generator(device="cpu").manual_seed(self.center_seed)
            self._centers[key] = torch.randn((self.ncenter, self.d), generator=g).to(device)
        return self._centers[key]

    def chunk(self, c, device):
        """rows [c*CHUNK, min((c+1)*CHUNK, n)) as an fp32 device tensor; pure function of (seed, c)"""
        lo = c * CHUNK
        m = min(CHUNK, self.n - lo)
        g = torch.Generator(device=device).manual_seed(self.seed * 1000003 + c)
        if self.kind == "uniform":
            return torch.rand((m, self.d), device=device, generator=g) * 100.0
        cen = self.centers(device)
        which = torch.randint(0, self.ncenter, (m,), device=device, generator=g)
        x = self._component(m, device, g)
        x += cen[which]
        if self.kind == "int8":
            x = self._to_int8_valued(x)
        return x

    @staticmethod
    def _to_int8_valued(x):
        """"int8" inputs in Knowhere are int8 vectors converted to fp32 before training / SQ8 encoding
        (reference include/knowhere/index/index_factory.h:144-145): integer-valued fp32 in [-128, 127]"""
        return torch.round(x * 32.0).clamp_(-128.0, 127.0)

    def nchunks(self):
        return (self.n + CHUNK - 1) // CHUNK

    def rows(self, lo, hi, device):
        """arbitrary slice (concatenates chunk pieces)"""
        out = []
        c = lo // CHUNK
        while c * CHUNK < hi:
            x = self.chunk(c, device)
            a = max(lo - c * CHUNK, 0)
            b = min(hi - c * CHUNK, x.shape[0])
            out.append(x[a:b])
            c += 1
        return torch.cat(out) if len(out) > 1 else out[0]

    def sample(self, m, device, seed=1234):
        """m rows spread over the data set (training subsample)"""
        m = min(m, self.n)
        nch = self.nchunks()
        per = int(math.ceil(m / nch))
        out = []
        for c in range(nch):
            x = self.chunk(c, device)
            g = torch.Generator(device=device).manual_seed(seed + c)
            sel = torch.randperm(x.shape[0], device=device, generator=g)[:per]
            out.append(x[sel])
            if sum(o.shape[0] for o in out) >= m:
                break
        return torch.cat(out)[:m].contiguous()


def queries(spec, nq, device, seed=44):
    """queries from the same distribution, different stream (reference uses seed and seed+2,
    tests/ut/test_gpu_search.cc:64-65)"""
    g = torch.Generator(device=device).manual_seed(seed * 7919 + 13)
    if spec.kind == "uniform":
        return torch.rand((nq, spec.d), device=device, generator=g) * 100.0
    cen = spec.centers(device)
    which = torch.randint(0, spec.ncenter, (nq,), device=device, generator=g)
    x = spec._component(nq, device, g) + cen[which]
    if spec.kind == "int8":
        x = spec._to_int8_valued(x)
    return x.contiguous()


class BuiltIndex:
    """device-resident build result + export to the plain arrays the oracle understands"""

    def __init__(self):
        self.kind = self.metric = self.d = self.nlist = self.M = None
        self.centroids = self.codebooks = self.sq_trained = None
        self.codes = self.ids = None       # list-sorted, ids ascending inside a list
        self.list_offsets = None           # numpy int64 [nlist+1]
        self.gpu = None                    # the GpuIndex trained by build_ivf (lists are attached by to_gpu_index)
        self.timings = {}

    def to_gpu_index(self, device=0, owned_lists=None):
        """the GpuIndex that was trained on the device, with the inverted lists attached.
        owned_lists: optional boolean numpy mask [nlist]; lists not owned are left empty."""
        from .index import GpuIndex, IVF_PQ, IVF_SQ8
        g = self.gpu if owned_lists is None else None
        self.gpu = None  # handed out at most once (the caller owns and closes it); further calls build a new handle
        if g is None or g.h is None or g.device != device:
            g = GpuIndex(self.kind, self.metric, self.d, nlist=self.nlist, pq_m=self.M or 0, device=device)
            g.set_coarse_device(self.centroids)
            if self.kind == IVF_PQ:
                g.set_pq(self.codebooks.cpu().numpy())
            if self.kind == IVF_SQ8:
                t = self.sq_trained.cpu().numpy()
                g.set_sq(t[:self.d], t[self.d:])
        if owned_lists is None:
            g.set_lists_device(self.list_offsets, self.codes, self.ids)
        else:
            off = self.list_offsets
            sizes = np.where(owned_lists, off[1:] - off[:-1], 0)
            new_off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
            keep = torch.from_numpy(np.repeat(owned_lists, off[1:] - off[:-1])).to(self.codes.device)
            g.set_lists_device(new_off, self.codes[keep].contiguous(), self.ids[keep].contiguous())
        return g

    def export(self, IndexData, lists=None):
        """-> oracle IndexData (host numpy).  Used ONLY by tests / the cpu_baseline leg.
        lists: optional list numbers -- only those lists are filled (the others stay empty): the SUB-INDEX a set of queries
        probes, for an index whose codes do not fit the host (C5: 76.8 GB); a search of those queries with the same nprobe
        returns what the whole index returns, since the coarse quantizer is complete and nothing else is visited."""
        ix = IndexData({1: 1, 2: 2, 3: 3}[self.kind], self.metric, self.d, self.nlist, self.M or 0, 8)
        ix.centroids = self.centroids.cpu().numpy()
        if self.codebooks is not None:
            ix.pq_centroids = self.codebooks.cpu().numpy()
        if self.sq_trained is not None:
            ix.sq_trained = self.sq_trained.cpu().numpy()
        off = self.list_offsets
        if lists is None:
            codes = self.codes.cpu().numpy()
            ids = self.ids.cpu().numpy()
            ix.list_codes = [codes[off[l]:off[l + 1]] for l in range(self.nlist)]
            ix.list_ids = [ids[off[l]:off[l + 1]] for l in range(self.nlist)]
            return ix
        cs = self.codes.shape[1]
        want = np.zeros(self.nlist, bool)
        want[np.asarray(lists, np.int64)] = True
        e_c, e_i = np.empty((0, cs), np.uint8), np.empty(0, np.int64)
        ix.list_codes, ix.list_ids = [], []
        for l in range(self.nlist):
            if want[l] and off[l + 1] > off[l]:
                ix.list_codes.append(self.codes[off[l]:off[l + 1]].cpu().numpy())
                ix.list_ids.append(self.ids[off[l]:off[l + 1]].cpu().numpy())
            else:
                ix.list_codes.append(e_c)
                ix.list_ids.append(e_i)
        return ix


def build_ivf(spec, kind, metric, nlist, M=32, device="cuda:0", train_per_centroid=256, niter=None,
              pq_train=1 << 20, centroids=None, codebooks=None, sq_trained=None, row_range=None, verbose=False,
              keep_vectors=False, train_only=False, owned_lists=None):
    """Train (unless centroids/codebooks are given, e.g. broadcast from rank 0) and encode
    rows [row_range) of the synthetic data set.  Returns a BuiltIndex on `device`.
    Clustering defaults are the reference's: 25 iterations, at most 256 training points per centroid
    (thirdparty/faiss/faiss/Clustering.h:24-77); they also give well balanced lists (measured at
    100M / 16384 lists: size-biased mean list length 1.43x the mean vs 1.83x with 10 iterations on
    64 points per centroid -- i.e. 22 % fewer bytes scanned at the same recall)."""
    import time
    from .index import GpuIndex, IVF_FLAT, IVF_PQ, IVF_SQ8
    dev = torch.device(device)
    d = spec.d
    out = BuiltIndex()
    out.kind, out.metric, out.d, out.nlist, out.M = kind, metric, d, nlist, (M if kind == IVF_PQ else 0)
    g = GpuIndex(kind, metric, d, nlist=nlist, pq_m=out.M, device=dev.index or 0)
    out.gpu = g
    t0 = time.time()
    lo, hi = row_range if row_range is not None else (0, spec.n)
    c0, c1 = lo // CHUNK, (hi + CHUNK - 1) // CHUNK
    own_t = torch.from_numpy(np.asarray(owned_lists, bool)).to(dev) if owned_lists is not None else None
    # the raw rows when they are kept whole (single-GPU refine): generated once, used for training AND encoding
    vectors = None
    if keep_vectors and own_t is None:  # (also for train_only: rank 0 of a sharded build trains on what a single GPU trains on)
        vectors = torch.empty((hi - lo, d), device=dev)
        vpos = 0
        for c in range(c0, c1):
            x = spec.chunk(c, dev)
            x = x[max(lo - c * CHUNK, 0):min(hi - c * CHUNK, x.shape[0])]
            vectors[vpos:vpos + x.shape[0]] = x
            vpos += x.shape[0]
    # ---- training on the device (knhip_index_train_device): the reference trains on the data set it is given and
    # sub-samples internally (<= train_per_centroid rows per centroid); where the rows are not resident a sample is
    if centroids is not None:
        g.set_coarse_device(centroids.contiguous())
    need_codec = (kind == IVF_PQ and codebooks is None) or (kind == IVF_SQ8 and sq_trained is None)
    if centroids is None or need_codec:
        if vectors is not None and vectors.shape[0] < (1 << 31):
            xt = vectors
        else:
            xt = spec.sample(min(spec.n, max(train_per_centroid * nlist, pq_train)), dev)
        if kind == IVF_PQ and codebooks is not None:
            g.set_pq(codebooks.cpu().numpy())
        if kind == IVF_SQ8 and sq_trained is not None:
            t = sq_trained.cpu().numpy()
            g.set_sq(t[:d], t[d:])
        if centroids is None or need_codec:
            g.train(xt, niter=niter, max_points=train_per_centroid)
        del xt
    else:
        if kind == IVF_PQ:
            g.set_pq(codebooks.cpu().numpy())
        if kind == IVF_SQ8:
            t = sq_trained.cpu().numpy()
            g.set_sq(t[:d], t[d:])
    out.centroids = torch.from_numpy(g.get_coarse()).to(dev)
    if kind == IVF_PQ:
        out.codebooks = torch.from_numpy(g.get_pq()).to(dev)
    if kind == IVF_SQ8:
        out.sq_trained = torch.from_numpy(g.get_sq()).to(dev)
    torch.cuda.synchronize(dev)
    out.timings["train_s"] = time.time() - t0
    if train_only:
        return out
    # ---- assignment + encoding on the device (knhip_index_encode_device), chunk by chunk
    t0 = time.time()
    assign_parts, code_parts, id_parts, vec_parts = [], [], [], []
    vpos = 0
    for c in range(c0, c1):
        a_lo = max(lo - c * CHUNK, 0)
        if vectors is not None:
            n_c = min(hi - c * CHUNK, CHUNK) - a_lo
            x = vectors[vpos:vpos + n_c]
            vpos += n_c
        else:
            x = spec.chunk(c, dev)
            x = x[a_lo:min(hi - c * CHUNK, x.shape[0])].contiguous()
        a, codes_c = g.encode_device(x)
        rid = torch.arange(c * CHUNK + a_lo, c * CHUNK + a_lo + x.shape[0], device=dev, dtype=torch.int64)
        if own_t is not None:
            keep = own_t[a]
            a, codes_c, rid = a[keep], codes_c[keep], rid[keep]
            if keep_vectors:
                vec_parts.append(x[keep].contiguous())
        assign_parts.append(a.to(torch.int32))
        code_parts.append(codes_c)
        id_parts.append(rid)
        if verbose and (c - c0) % 16 == 0:
            print(f"  encoded chunk {c - c0 + 1}/{c1 - c0}", flush=True)
    assign = torch.cat(assign_parts).to(torch.int64)
    del assign_parts
    codes = torch.cat(code_parts)
    del code_parts
    row_ids = torch.cat(id_parts)
    del id_parts
    torch.cuda.synchronize(dev)
    out.timings["encode_s"] = time.time() - t0
    t0 = time.time()
    order = torch.sort(assign, stable=True).indices  # stable: ids stay ascending inside a list
    out.codes = codes[order].contiguous()
    del codes
    out.ids = row_ids[order].contiguous()
    counts = torch.bincount(assign, minlength=nlist).cpu().numpy().astype(np.int64)
    out.list_offsets = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    if keep_vectors and own_t is not None:
        out.vectors = torch.cat(vec_parts) if vec_parts else torch.empty((0, d), device=dev)
        out.vector_ids = row_ids  # ascending: rows were visited in id order
    elif keep_vectors:
        out.vectors = vectors  # row r <-> id lo + r
        out.vector_ids = None if lo == 0 else row_ids
    torch.cuda.synchronize(dev)
    out.timings["sort_s"] = time.time() - t0
    del order, assign
    torch.cuda.empty_cache()  # the index allocates with hipMalloc: hand torch's cached blocks (C5: ~150 GB) back
    return out


def ground_truth(spec, xq, k, metric=0, device="cuda:0", row_range=None):
    """exact k-NN of xq over the synthetic base by streaming chunks through GEMM + topk.
    Recall bookkeeping only (validated against the oracle at small sizes in tests)."""
    dev = torch.device(device)
    nq = xq.shape[0]
    best_d = torch.full((nq, k), float("inf"), device=dev)
    best_i = torch.full((nq, k), -1, dtype=torch.int64, device=dev)
    q_sq = (xq * xq).sum(1, keepdim=True)
    lo, hi = row_range if row_range is not None else (0, spec.n)
    for c in range(lo // CHUNK, (hi + CHUNK - 1) // CHUNK):
        x = spec.chunk(c, dev)
        base = c * CHUNK
        if metric == 0:
            dist = q_sq + (x * x).sum(1).unsqueeze(0) - 2.0 * (xq @ x.t())
        else:
            dist = -(xq @ x.t())
        dd, ii = torch.topk(dist, min(k, x.shape[0]), dim=1, largest=False)
        cat_d = torch.cat([best_d, dd], 1)
        cat_i = torch.cat([best_i, ii + base], 1)
        sel = torch.topk(cat_d, k, dim=1, largest=False).indices
        best_d = torch.gather(cat_d, 1, sel)
        best_i = torch.gather(cat_i, 1, sel)
    return best_d, best_i
