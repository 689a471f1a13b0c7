"""Executable model of the half-precision ADC prefilter (knowhere_amd/csrc/pq_filter.hip + the KIND 2 finish of
mfma_scan.hip), checked against the oracle on the CPU.

The kernels cannot run here (no GPU); what CAN be checked is that the data layouts and index formulas they are built on
agree with each other and that the pipeline they implement returns the oracle's bits:

  * the rotated token stream (pq_stream16r_kernel): token = code << 8 | m << 3 with m = (T + phase(lane)) & 31;
  * the per-query half table in its permuted layout (pqf_query_table_kernel) and the register transposition that turns
    8 such tables into LUT[c][m][8 queries] (v_perm selectors of pqf_kernel);
  * the scan loop's addressing (token << 1 = LDS byte address of the 16-byte entry) and half-precision accumulation;
  * bound -> filter -> candidates -> exact ADC in the reference's operation order -> canonical top-k.

Each function below restates the corresponding kernel line by line (same index arithmetic, numpy instead of lanes).
The end-to-end test runs the model on a small IVF-PQ index and requires the oracle's ids and distances bit for bit,
with every true result inside the filter's candidate set."""
import numpy as np
import pytest

from conftest import gen_data
from oracle import binding as ob

f32 = np.float32
f16 = np.float16
M, KSUB, DSUB = 32, 256, 4
U = f32(2.0 ** -24)
UH = f32(2.0 ** -11)


def phase(lane):  # kernels.h::pq_stream_phase
    l = lane & 31
    return l if l < 4 else l + 4 if l < 12 else l - 8 if l < 16 else l - 16 if l < 20 else l - 12 if l < 28 else l - 24


def stream16r(codes):
    """pq_stream16r_kernel: uint16 tokens [nblk][64 lanes][8 steps]; nblk = pq_stream16r_blocks(len)"""
    n = codes.shape[0]
    nblk = ((n + 63) // 64) * 4 + 8
    out = np.zeros((nblk, 64, 8), np.uint16)
    for blk in range(nblk):
        for L in range(64):
            v = (blk >> 2) * 64 + L
            for s in range(8):
                T = (blk & 3) * 8 + s
                m = (T + phase(L)) & 31
                code = int(codes[v, m]) if v < n else 0
                out[blk, L, s] = (code << 8) | (m << 3)
    return out


def ip_tables(q, cb):
    """<q_m, cb[m][c]> accumulated in dimension order from 0, one rounding per operation -> [M][KSUB] fp32"""
    T = np.zeros((M, KSUB), f32)
    for m in range(M):
        t = np.zeros(KSUB, f32)
        for e in range(DSUB):
            t = (t + (q[m * DSUB + e] * cb[m, :, e]).astype(f32)).astype(f32)
        T[m] = t
    return T


def query_table(q, cb, is_l2, pabs_max):
    """pqf_query_table_kernel -> (halves in the permuted layout [c >> 2][m & 15][c & 3][m >> 4], sc, 1/sc, eps_base)"""
    T = ip_tables(q, cb)
    Qf = (f32(-2.0) * T).astype(f32) if is_l2 else T
    A = f32(0)
    for m in range(M):
        A = f32(A + np.abs(Qf[m]).max())
    sc = f32(f32(2032.0) / A) if A > 0 else f32(1.0)
    eps = f32(f32(16.5) / sc + f32(64.0) * U * f32(pabs_max + A))
    qh = np.zeros((KSUB // 4, 16, 4, 2), f16)
    for c in range(KSUB):
        for l16 in range(16):
            qh[c >> 2, l16, c & 3, 0] = f16(np.rint(f32(Qf[l16, c] * sc)))
            qh[c >> 2, l16, c & 3, 1] = f16(np.rint(f32(Qf[l16 + 16, c] * sc)))
    return qh, sc, f32(1.0) / sc, eps, T


def perm(s0, s1, sel):
    """v_perm_b32 D = perm(S0, S1, sel): selector byte k picks byte k of {S0 (bytes 4..7), S1 (bytes 0..3)}"""
    src = [(s1 >> (8 * i)) & 0xff for i in range(4)] + [(s0 >> (8 * i)) & 0xff for i in range(4)]
    out = 0
    for i in range(4):
        out |= src[(sel >> (8 * i)) & 0xff] << (8 * i)
    return out


def build_lut(tables):
    """pqf_kernel's LUT build: thread t holds, per query j, the 16-byte piece t of the query's table (4 words = cells
    (c = 4 (t >> 4) + cc, m = (t & 15) + 16 h), word cc, half h); 8 stores of 16 bytes: entry (c, m) = 8 halves"""
    lds = np.zeros(KSUB * M * 8, np.uint16)  # LDS as halves; byte address / 2
    words = [tb.reshape(-1).view(np.uint32).reshape(1024, 4) for tb in tables]  # [t][cc]
    for t in range(1024):
        c4, l16 = t >> 4, t & 15
        for e in range(8):
            cc, h = e >> 1, e & 1
            sel = 0x07060302 if h else 0x05040100
            o = [perm(int(words[2 * w + 1][t, cc]), int(words[2 * w][t, cc]), sel) for w in range(4)]
            entry = (c4 * 4 + cc) * M + l16 + 16 * h            # uint4 index of lut[]
            lds[entry * 8:entry * 8 + 8] = np.array(o, np.uint32).view(np.uint16)
    return lds


def scan_list(tokens, lds, n):
    """the window loop: per group of 64 vectors 32 steps of {address = token << 1, 16-byte read, 8 half additions}"""
    ngroups = (n + 63) // 64
    out = np.zeros((ngroups * 64, 8), f16)
    lut = lds.view(f16)
    for g in range(ngroups):
        for L in range(64):
            acc = np.zeros(8, f16)
            for T in range(32):
                tok = int(tokens[4 * g + T // 8, L, T % 8])
                addr = tok << 1                                   # LDS byte address (v_lshlrev_b32_sdwa by 1)
                acc = (acc + lut[addr // 2:addr // 2 + 8]).astype(f16)
            out[g * 64 + L] = acc
    return out[:n]


def model_search(port, ix, xq, k, nprobe):
    """sample-free form of the pipeline: tau_q = the exact k-th distance over the probed lists would be the tightest
    bound; the model takes the pessimistic k-th of the closest list (as the sample pass does), widened by eps"""
    is_l2 = ix.metric == ob.L2
    nq = xq.shape[0]
    cdis, keys = port.coarse_search(ix, xq, nprobe)
    cb = ix.pq_centroids.reshape(M, KSUB, DSUB)
    P = ix.precomputed_table.reshape(ix.nlist, M, KSUB) if is_l2 else None
    ar = np.arange(M)
    pabs_max = f32(0)
    psum = {}
    if is_l2:
        for l in range(ix.nlist):
            codes = ix.list_codes[l].astype(np.int64)
            if len(codes):
                t2 = P[l][ar[None, :], codes]
                ps = np.zeros(len(codes), f32)
                for m in range(M):
                    ps = (ps + t2[:, m]).astype(f32)
                psum[l] = ps
                pabs_max = max(pabs_max, f32(np.abs(t2).astype(f32).sum(1, dtype=f32).max()))
    tok = {l: stream16r(ix.list_codes[l]) for l in range(ix.nlist) if len(ix.list_codes[l])}
    D = np.full((nq, k), np.finfo(f32).max if is_l2 else -np.finfo(f32).max, f32)
    I = np.full((nq, k), -1, np.int64)
    ncand = []
    for q in range(nq):
        qh, sc, isc, eps_base, T = query_table(xq[q], cb, is_l2, pabs_max)
        lds = build_lut([qh] * 8)  # (the 8 slots of a unit: here the same query, as in a one-query unit)
        approx = {}
        for s in range(nprobe):
            l = int(keys[q, s])
            if l < 0 or l not in tok:
                continue
            h = scan_list(tok[l], lds, len(ix.list_codes[l]))
            for j in range(1, 8):
                assert np.array_equal(h[:, j].view(np.uint16), h[:, 0].view(np.uint16))  # every slot: the same sums
            dis0 = f32(cdis[q, s])
            eps = f32(eps_base + f32(64.0) * U * abs(dis0))
            if is_l2:
                pess = (h[:, 0].astype(f32) * isc + (f32(dis0 + eps) + psum[l]).astype(f32)).astype(f32)
            else:
                pess = (h[:, 0].astype(f32) * isc + f32(dis0 - eps)).astype(f32)
            approx[s] = (h[:, 0].astype(f32), pess, dis0)
        if not approx:
            continue
        allp = np.concatenate([a[1] for a in approx.values()])
        if len(allp) < k:
            tau = None  # no bound: the exact kernels take the query
        else:
            tau = f32(np.sort(allp)[k - 1] if is_l2 else -np.sort(-allp)[k - 1])
        cand = []
        for s, (hf, pess, dis0) in approx.items():
            l = int(keys[q, s])
            if tau is None:
                hit = np.ones(len(hf), bool)
            else:
                eps = f32(eps_base + f32(64.0) * U * f32(abs(dis0) + abs(tau)))
                if is_l2:
                    thr = f32(f32(f32(tau + eps) - dis0) * sc)
                    hit = (psum[l] * sc + hf).astype(f32) <= thr   # (the kernel fuses this multiply-add: eps covers both)
                else:
                    thr = f32(f32(f32(tau - eps) - dis0) * sc)
                    hit = hf >= thr
            cand += [(s, int(p)) for p in np.nonzero(hit)[0]]
        ncand.append(len(cand))
        # ---- exact finish (mscan_finish_kernel KIND 2) ----
        res = []
        for s, pos in cand:
            l = int(keys[q, s])
            code = ix.list_codes[l][pos].astype(np.int64)
            acc = f32(0)
            for m in range(M):
                t = T[m, code[m]]
                if is_l2:
                    t = f32(P[l][m, code[m]] + f32(f32(-2.0) * t))
                acc = f32(acc + t)
            dis = f32(f32(cdis[q, s]) + acc)
            res.append((dis, int(ix.list_ids[l][pos])))
        res.sort(key=(lambda r: (r[0], r[1])) if is_l2 else (lambda r: (-r[0], -r[1])))
        for e, (dd, ii) in enumerate(res[:k]):
            D[q, e] = dd
            I[q, e] = ii
    return D, I, ncand


@pytest.mark.parametrize("metric", [ob.L2, ob.IP], ids=["l2", "ip"])
def test_model_pipeline_returns_the_oracles_bits(port, metric):
    nb, d, nlist, nq, k, nprobe = 2500, 128, 6, 5, 10, 3
    xb, xq = gen_data(nb, d, 42), gen_data(nq, d, 44)
    ix = ob.make_index(port, ob.IVF_PQ, metric, xb, nlist=nlist, M=32)
    Do, Io = port.search(ix, xq, k, nprobe)
    D, I, ncand = model_search(port, ix, xq, k, nprobe)
    assert np.array_equal(I, Io)
    assert np.array_equal(D.view(np.uint32), Do.view(np.uint32))
    scanned = sum(len(c) for c in ix.list_codes) * nprobe / nlist
    assert max(ncand) < 0.5 * scanned, (ncand, scanned)  # the filter filters


def test_lut_transposition_matches_the_tables():
    """8 different per-query tables through build_lut: LUT entry (c, m) holds query j's half at slot j"""
    rng = np.random.default_rng(1)
    cb = rng.standard_normal((M, KSUB, DSUB)).astype(f32)
    tabs, plain = [], []
    for j in range(8):
        q = rng.standard_normal(M * DSUB).astype(f32)
        qh, sc, _, _, T = query_table(q, cb, True, f32(0))
        tabs.append(qh)
        plain.append(np.rint(((f32(-2.0) * T).astype(f32) * sc).astype(f32)).astype(f16))  # [m][c]
    lds = build_lut(tabs).view(f16).reshape(KSUB, M, 8)
    for j in range(8):
        assert np.array_equal(lds[:, :, j].view(np.uint16), plain[j].T.view(np.uint16))
