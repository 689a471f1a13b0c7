"""tests/simd_cases.py -- the src/simd hook table (reference src/simd/hook.h:33-123) as a list of seeded cases, evaluated
through any object with the _SimdTable call shapes (oracle.binding.Port / Ref, or the GPU adapter of
tests/test_gpu_simd.py).  Dimensions follow the reference's own test (tests/ut/test_simd.cc:259-568: odd and even dims,
1 .. ~1k) plus the shapes the index path uses (4 = PQ sub-vector, 128, 768)."""
import numpy as np

DIMS = (1, 2, 3, 4, 7, 8, 15, 16, 17, 31, 32, 33, 64, 100, 128, 257, 768, 1000)
NY = 77


def bf16_bits(a):
    """float32 -> bfloat16 bit patterns by truncation (operands.h:133-138: bits >> 16)"""
    return (np.ascontiguousarray(a, np.float32).view(np.uint32) >> 16).astype(np.uint16)


def inputs(d, seed=0):
    r = np.random.default_rng(1000 * d + seed)
    x = (r.random(d, dtype=np.float32) * 200 - 100).astype(np.float32)
    y = (r.random((NY, d), dtype=np.float32) * 200 - 100).astype(np.float32)
    y[5] = y[3]        # a tie: the first minimum must win
    y[11] = x          # an exact hit (distance 0)
    a = (r.random(513, dtype=np.float32) * 50).astype(np.float32)
    b = (r.random(513, dtype=np.float32) * 50).astype(np.float32)
    a[100] = a[7] = 0.0
    b[100] = b[7] = 0.0  # tie in madd_and_argmin
    xi = r.integers(-128, 128, d, dtype=np.int8)
    yi = r.integers(-128, 128, (NY, d), dtype=np.int8)
    xh = (r.random(d) * 8 - 4).astype(np.float16)
    yh = (r.random((NY, d)) * 8 - 4).astype(np.float16)
    xh[0] = np.float16(6e-8)   # fp16 subnormal
    xb = bf16_bits(r.random(d, dtype=np.float32) * 8 - 4)
    yb = bf16_bits(r.random((NY, d), dtype=np.float32) * 8 - 4)
    return dict(x=x, y=y, a=a, b=b, xi=xi, yi=yi, xh=xh, yh=yh, xb=xb, yb=yb)


def evaluate(T, d, seed=0):
    """every entry of the table on the inputs of dimension d -> {name: ndarray}"""
    z = inputs(d, seed)
    x, y = z["x"], z["y"]
    out = {}
    for name in ("fvec_inner_product", "fvec_L2sqr", "fvec_L1", "fvec_Linf"):
        out[name] = np.array([T.simd_scalar(name, x, r) for r in y[:8]], np.float32)
    out["fvec_norm_L2sqr"] = np.array([T.simd_scalar("fvec_norm_L2sqr", r) for r in y[:8]], np.float32)
    out["fvec_L2sqr_ny"] = T.simd_ny("fvec_L2sqr_ny", x, y)
    out["fvec_inner_products_ny"] = T.simd_ny("fvec_inner_products_ny", x, y)
    # transposed block: vector i in column i, row pitch d_offset > ny
    d_offset = NY + 3
    yt = np.zeros((d, d_offset), np.float32)
    yt[:, :NY] = y.T
    ysq = np.array([T.simd_scalar("fvec_inner_product", r, r) for r in y], np.float32)
    out["fvec_L2sqr_ny_transposed"] = T.simd_ny_transposed(x, yt, ysq, NY)
    i, dis = T.simd_ny_transposed(x, yt, ysq, NY, nearest=True)
    out["fvec_L2sqr_ny_nearest_y_transposed"] = np.array([i], np.int64)
    out["fvec_L2sqr_ny_nearest_y_transposed.dis"] = dis
    i, dis = T.simd_ny_nearest(x, y)
    out["fvec_L2sqr_ny_nearest"] = np.array([i], np.int64)
    i3, _ = T.simd_ny_nearest(y[5], y)   # y[3] == y[5]: index 3 wins
    out["fvec_L2sqr_ny_nearest.tie"] = np.array([i3], np.int64)
    out["fvec_madd"] = T.simd_madd(z["a"], -2.5, z["b"])
    i, c = T.simd_madd(z["a"], 3.0, z["b"], argmin=True)
    out["fvec_madd_and_argmin"] = np.array([i], np.int64)
    out["fvec_madd_and_argmin.c"] = c
    big = (z["a"] + np.float32(2e20)).astype(np.float32)
    i, _ = T.simd_madd(big, 1.0, z["b"], argmin=True)   # nothing below 1e20 -> -1
    out["fvec_madd_and_argmin.none"] = np.array([i], np.int64)
    for is_l2 in (1, 0):
        tag = "L2sqr" if is_l2 else "inner_product"
        out[f"fvec_{tag}_batch_4"] = T.simd_batch_4(is_l2, x, y[20:24])
        out[f"fp16_vec_{tag}_batch_4"] = T.simd_batch_4(is_l2, z["xh"], z["yh"][20:24])
        out[f"bf16_vec_{tag}_batch_4"] = T.simd_batch_4(is_l2, z["xb"], z["yb"][20:24])
        out[f"int8_vec_{tag}_batch_4"] = T.simd_batch_4(is_l2, z["xi"], z["yi"][20:24])
        out[f"ivec_{tag}"] = np.array([T.simd_ivec(is_l2, z["xi"], r) for r in z["yi"][:8]], np.int64)
    for pre, xx, yy in (("fp16", z["xh"], z["yh"]), ("bf16", z["xb"], z["yb"]), ("int8", z["xi"], z["yi"])):
        out[f"{pre}_vec_L2sqr"] = np.array([T.simd_typed(0, xx, r) for r in yy[:8]], np.float32)
        out[f"{pre}_vec_inner_product"] = np.array([T.simd_typed(1, xx, r) for r in yy[:8]], np.float32)
        out[f"{pre}_vec_norm_L2sqr"] = np.array([T.simd_typed(2, r) for r in yy[:8]], np.float32)
    return out


def same(a, b):
    return a.dtype == b.dtype and a.shape == b.shape and a.tobytes() == b.tobytes()
