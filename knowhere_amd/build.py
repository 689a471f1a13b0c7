"""knowhere_amd/build.py -- GPU index builder used by bench.py and the scale tests.

Build-side counterpart of the reference's Train/Add (reference src/index/ivf/ivf.cc:547-844,
thirdparty/faiss/faiss/IndexIVF.cpp:55-121 train_q1, 212-287 add_core; Clustering.h:24-77).  The
hot path of this project is Search(); building is "next" scope, so this
module is plain PyTorch (rocBLAS GEMMs) -- plumbing, not product kernels.  It produces exactly the
objects the C ABI ingests: coarse centroids, PQ codebooks / SQ ranges, and list-sorted codes + ids.
Search parity is defined on the index BYTES, so the oracle is handed the same arrays.

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


# ---- nearest centroid (L2) in row blocks --------------------------------------------------------
def _assign_l2(x, cen, cen_sq, block=1 << 17, metric=0):
    """nearest centroid: L2 (metric 0) or largest inner product (metric 1: the coarse quantizer of an IP index is
    an IndexFlatIP, reference src/index/ivf/ivf.cc:592, 613, 641)"""
    out = torch.empty(x.shape[0], dtype=torch.int64, device=x.device)
    for lo in range(0, x.shape[0], block):
        xb = x[lo:lo + block]
        if metric == 1:
            out[lo:lo + block] = (xb @ cen.t()).argmax(dim=1)
        else:
            dist = torch.addmm(cen_sq.unsqueeze(0), xb, cen.t(), beta=1.0, alpha=-2.0)  # ||c||^2 - 2 x.c
            out[lo:lo + block] = dist.argmin(dim=1)
    return out


def kmeans(x, k, niter=10, seed=1234, verbose=False):
    """Lloyd k-means (L2), random-sample init, empty clusters re-seeded from the largest ones
    (spirit of faiss Clustering::train, thirdparty/faiss/faiss/Clustering.cpp)."""
    n, d = x.shape
    g = torch.Generator(device=x.device).manual_seed(seed)
    cen = x[torch.randperm(n, device=x.device, generator=g)[:k]].clone()
    for it in range(niter):
        cen_sq = (cen * cen).sum(1)
        a = _assign_l2(x, cen, cen_sq)
        cnt = torch.bincount(a, minlength=k).to(x.dtype)
        s = torch.zeros_like(cen).index_add_(0, a, x)
        nz = cnt > 0
        cen[nz] = s[nz] / cnt[nz].unsqueeze(1)
        nempty = int((~nz).sum().item())
        if nempty:
            big = torch.argsort(cnt, descending=True)[:nempty]
            cen[~nz] = cen[big] * (1 + 1e-4)
        if verbose:
            print(f"  kmeans it {it}: empty {nempty}", flush=True)
    return cen.contiguous()


def train_pq(resid, M, niter=10, seed=1234):
    """per-sub-space k-means with 256 codewords, batched over the M sub-spaces"""
    n, d = resid.shape
    dsub = d // M
    xs = resid.view(n, M, dsub).permute(1, 0, 2).contiguous()  # [M, n, dsub]
    g = torch.Generator(device=resid.device).manual_seed(seed)
    sel = torch.randperm(n, device=resid.device, generator=g)[:256]
    cb = xs[:, sel, :].clone()  # [M, 256, dsub]
    for _ in range(niter):
        a = _pq_assign(xs, cb)  # [M, n]
        for m in range(M):
            cnt = torch.bincount(a[m], minlength=256).to(xs.dtype)
            s = torch.zeros((256, dsub), device=xs.device, dtype=xs.dtype).index_add_(0, a[m], xs[m])
            nz = cnt > 0
            cb[m][nz] = s[nz] / cnt[nz].unsqueeze(1)
    return cb.contiguous()


def _pq_assign(xs, cb, block=1 << 18):
    """xs [M, n, dsub], cb [M, 256, dsub] -> nearest codeword [M, n]"""
    M, n, _ = xs.shape
    out = torch.empty((M, n), dtype=torch.int64, device=xs.device)
    cb_sq = (cb * cb).sum(2)  # [M, 256]
    for lo in range(0, n, block):
        xb = xs[:, lo:lo + block, :]
        dist = cb_sq.unsqueeze(1) - 2.0 * torch.bmm(xb, cb.transpose(1, 2))  # [M, b, 256]
        out[:, lo:lo + block] = dist.argmin(dim=2)
    return out


def pq_encode(resid, cb):
    n, d = resid.shape
    M = cb.shape[0]
    xs = resid.view(n, M, d // M).permute(1, 0, 2).contiguous()
    return _pq_assign(xs, cb).t().contiguous().to(torch.uint8)  # [n, M]


class BuiltIndex:
    """device-resident build result + export to the plain arrays the oracle understands"""

    def __init__(self):
        self.kind = self.metric = self.d = self.nlist = self.M = None
        self.centroids = self.codebooks = self.sq_trained = None
        self.codes = self.ids = None       # list-sorted, ids ascending inside a list
        self.list_offsets = None           # numpy int64 [nlist+1]
        self.timings = {}

    def to_gpu_index(self, device=0, owned_lists=None):
        """owned_lists: optional boolean numpy mask [nlist]; lists not owned are left empty
        (multi-GPU list sharding: each rank holds only its own lists)."""
        from .index import GpuIndex, IVF_PQ, IVF_SQ8
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

    def export(self, IndexData, list_limit=None):
        """-> oracle IndexData (host numpy).  Used ONLY by tests / the cpu_baseline leg."""
        ix = IndexData({1: 1, 2: 2, 3: 3}[self.kind], self.metric, self.d, self.nlist, self.M or 0, 8)
        ix.centroids = self.centroids.cpu().numpy()
        if self.codebooks is not None:
            ix.pq_centroids = self.codebooks.cpu().numpy()
        if self.sq_trained is not None:
            ix.sq_trained = self.sq_trained.cpu().numpy()
        codes = self.codes.cpu().numpy()
        ids = self.ids.cpu().numpy()
        off = self.list_offsets
        ix.list_codes = [codes[off[l]:off[l + 1]] for l in range(self.nlist)]
        ix.list_ids = [ids[off[l]:off[l + 1]] for l in range(self.nlist)]
        return ix


def build_ivf(spec, kind, metric, nlist, M=32, device="cuda:0", train_per_centroid=256, niter=25,
              pq_train=1 << 20, centroids=None, codebooks=None, sq_trained=None, row_range=None, verbose=False,
              keep_vectors=False, train_only=False, owned_lists=None):
    """Train (unless centroids/codebooks are given, e.g. broadcast from rank 0) and encode
    rows [row_range) of the synthetic data set.  Returns a BuiltIndex on `device`.
    Clustering defaults are the reference's: 25 iterations, at most 256 training points per centroid
    (thirdparty/faiss/faiss/Clustering.h:24-77); they also give well balanced lists (measured at
    100M / 16384 lists: size-biased mean list length 1.43x the mean vs 1.83x with 10 iterations on
    64 points per centroid -- i.e. 22 % fewer bytes scanned at the same recall)."""
    import time
    from .index import IVF_FLAT, IVF_PQ, IVF_SQ8
    dev = torch.device(device)
    d = spec.d
    out = BuiltIndex()
    out.kind, out.metric, out.d, out.nlist, out.M = kind, metric, d, nlist, (M if kind == IVF_PQ else 0)
    t0 = time.time()
    if centroids is None:
        xt = spec.sample(min(spec.n, train_per_centroid * nlist), dev)
        centroids = kmeans(xt, nlist, niter=niter, verbose=verbose)
        del xt
    out.centroids = centroids.contiguous()
    cen_sq = (centroids * centroids).sum(1)
    torch.cuda.synchronize(dev)
    out.timings["train_coarse_s"] = time.time() - t0
    t0 = time.time()
    if kind == IVF_PQ and codebooks is None:
        xt = spec.sample(min(spec.n, pq_train), dev, seed=4321)
        a = _assign_l2(xt, centroids, cen_sq, metric=metric)
        codebooks = train_pq(xt - centroids[a], M, niter=niter)
        del xt, a
    out.codebooks = codebooks
    if kind == IVF_SQ8 and sq_trained is not None:
        out.sq_trained = sq_trained
    elif kind == IVF_SQ8:
        xt = spec.sample(min(spec.n, pq_train), dev, seed=4321)
        a = _assign_l2(xt, centroids, cen_sq, metric=metric)
        r = xt - centroids[a]
        vmin = r.min(0).values
        vdiff = r.max(0).values - vmin  # RS_minmax, rangestat_arg 0 (ScalarQuantizer.h:67-74)
        out.sq_trained = torch.cat([vmin, vdiff]).contiguous()
        del xt, a, r
    torch.cuda.synchronize(dev)
    out.timings["train_codec_s"] = time.time() - t0
    if train_only:
        return out
    t0 = time.time()
    lo, hi = row_range if row_range is not None else (0, spec.n)
    assign_parts, code_parts, id_parts, vec_parts = [], [], [], []
    own_t = torch.from_numpy(np.asarray(owned_lists, bool)).to(dev) if owned_lists is not None else None
    vectors = torch.empty((hi - lo, d), device=dev) if (keep_vectors and own_t is None) else None
    vpos = 0
    c0, c1 = lo // CHUNK, (hi + CHUNK - 1) // CHUNK
    for c in range(c0, c1):
        x = spec.chunk(c, dev)
        a_lo = max(lo - c * CHUNK, 0)
        a_hi = min(hi - c * CHUNK, x.shape[0])
        x = x[a_lo:a_hi]
        a = _assign_l2(x, centroids, cen_sq, metric=metric)
        rid = torch.arange(c * CHUNK + a_lo, c * CHUNK + a_hi, device=dev, dtype=torch.int64)
        if own_t is not None:
            keep = own_t[a]
            x, a, rid = x[keep], a[keep], rid[keep]
        assign_parts.append(a.to(torch.int32))
        id_parts.append(rid)
        if kind == IVF_PQ:
            code_parts.append(pq_encode(x - centroids[a], codebooks))
        elif kind == IVF_SQ8:
            r = x - centroids[a]
            vmin, vdiff = out.sq_trained[:d], out.sq_trained[d:]
            xi = torch.where(vdiff != 0, (r - vmin) / vdiff, torch.zeros_like(r)).clamp_(0, 1)
            code_parts.append((255 * xi).to(torch.int32).clamp_(0, 255).to(torch.uint8))
        else:
            code_parts.append(x.contiguous().view(torch.uint8).reshape(x.shape[0], d * 4))
        if keep_vectors:
            if own_t is None:
                vectors[vpos:vpos + x.shape[0]] = x
                vpos += x.shape[0]
            else:
                vec_parts.append(x.contiguous())
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
