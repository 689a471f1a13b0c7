"""a10: the src/simd hook table (reference src/simd/hook.h:33-123).  The oracle's restatement (oracle.c orc_simd_*) is
pinned (1) against known answers produced by the reference's own scalar definitions (src/simd/distances_ref.cc; fixture
tests/golden/simd/table.npz, generator tests/golden/make_simd_golden.py) and (2) live against the same functions where
oracle/_ref is built.  Bar: bit-equal, every entry, every dimension.  The GPU side is tests/test_gpu_simd.py."""
import os

import numpy as np
import pytest

import simd_cases as sc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "simd", "table.npz")


@pytest.mark.parametrize("d", sc.DIMS)
def test_port_equals_reference_known_answers(port, d):
    z = np.load(GOLD)
    got = sc.evaluate(port, d)
    assert len(got) == 35
    for name, v in got.items():
        assert sc.same(v, z[f"d{d}/{name}"]), f"{name} d={d}: {v[:4]} vs {z[f'd{d}/{name}'][:4]}"


@pytest.mark.parametrize("d", (5, 24, 96, 130, 512))
def test_port_equals_reference_live(port, ref, d):
    """dimensions and seeds outside the fixture"""
    for seed in (1, 2):
        a, b = sc.evaluate(port, d, seed), sc.evaluate(ref, d, seed)
        for name in a:
            assert sc.same(a[name], b[name]), f"{name} d={d} seed={seed}"


def test_known_semantics(port):
    """the corner semantics the table pins: first minimum wins, exact self hit, -1 when nothing is below 1e20,
    the double-accumulated norm differs from the float-accumulated one the IVF-PQ path uses"""
    got = sc.evaluate(port, 128)
    assert got["fvec_L2sqr_ny_nearest"][0] == 11 and got["fvec_L2sqr_ny"][11] == 0.0
    assert got["fvec_L2sqr_ny_nearest.tie"][0] == 3
    assert got["fvec_madd_and_argmin"][0] == 7
    assert got["fvec_madd_and_argmin.none"][0] == -1
    z = sc.inputs(1000)
    n_ref = port.simd_scalar("fvec_norm_L2sqr", z["y"][0])
    n_faiss = np.float32(port.fvec_norm_L2sqr(z["y"][0]))
    exact = np.float32(np.sum(z["y"][0].astype(np.float64) ** 2))
    assert abs(float(n_ref) - float(exact)) <= abs(float(n_faiss) - float(exact))
