"""CPU checks of the arithmetic that compute mode f16x3 rests on (include/scfgp_hip.h: SCFGP_F16X3; scfgp_amd/csrc/apply_f16.hip,
gram_f16.hip), in numpy's float16 / float32: what the split represents, what the three terms lose, and the bound that sets V's scale
before V exists.  No reference counterpart (the reference is float64 throughout, SCFGP/SCFGP.py:95-96)."""
import numpy as np

from scfgp_amd import synth
from tests.cpu_f16x3_emulation import split16


def test_the_split_represents_an_fp32_value_to_two_to_the_minus_23_or_to_half_a_subnormal_step_of_the_low_part():
    rng = np.random.default_rng(0x5CF6)
    x = (rng.standard_normal(200000) * np.exp(rng.uniform(-9, 0, 200000))).astype(np.float32).astype(np.float64)
    h, l, e = split16(x)
    back = (h.astype(np.float64) + l.astype(np.float64)) * 2.0 ** -e
    assert 2.0 ** 14 <= np.abs(x).max() * 2.0 ** e < 2.0 ** 15         # one power-of-two scale: the largest entry lands in [2^14, 2^15)
    # h rounds to 11 bits, l rounds what is left to 11 more -- or, where that remainder falls below fp16's normal range (entries
    # under about 2^-15 of the largest), to the subnormal step 2^-24: an absolute floor of 2^-40 of the largest entry
    assert np.all(np.abs(back - x) <= 2.0 ** -22.9 * np.abs(x) + 2.0 ** -25 * 2.0 ** -e)
    big = np.abs(x) >= np.abs(x).max() * 2.0 ** -15
    assert np.all(np.abs(back - x)[big] <= 2.0 ** -22.9 * np.abs(x)[big])
    assert np.all(np.abs(back - x) <= 2.0 ** -23.9 * np.abs(x).max())
    assert np.all(np.isfinite(h)) and np.all(np.isfinite(l))


def test_three_terms_in_fp32_differ_from_the_exact_product_by_the_dropped_low_times_low_term():
    rng = np.random.default_rng(0x5CF7)
    a = rng.standard_normal(4096).astype(np.float32).astype(np.float64)
    b = rng.standard_normal(4096).astype(np.float32).astype(np.float64)
    ah, al, ea = split16(a); bh, bl, eb = split16(b)
    # products of two fp16 values are exact in fp32 (11 + 11 bits)
    for u, v in ((ah, bh), (al, bh), (ah, bl)):
        assert np.array_equal((u.astype(np.float32) * v.astype(np.float32)).astype(np.float64), u.astype(np.float64) * v.astype(np.float64))
    three = (ah.astype(np.float64) * bh + al.astype(np.float64) * bh + ah.astype(np.float64) * bl) * 2.0 ** -(ea + eb)
    dropped = al.astype(np.float64) * bl * 2.0 ** -(ea + eb)
    ra = (ah.astype(np.float64) + al) * 2.0 ** -ea; rb = (bh.astype(np.float64) + bl) * 2.0 ** -eb
    assert np.allclose(three + dropped, ra * rb, rtol=0, atol=1e-18)
    assert np.all(np.abs(dropped) <= 2.0 ** -22 * np.abs(a).max() * np.abs(b).max())
    # on a Gram's diagonal the dropped term is a sum of squares: the split's G is short by it, never long
    sq = np.sum((ah.astype(np.float64) ** 2 + 2 * ah.astype(np.float64) * al) * 4.0 ** -ea)
    assert sq <= np.sum(ra * ra) and np.sum(ra * ra) - sq <= 2.0 ** -22 * np.sum(ra * ra)


def test_the_bound_that_scales_V_before_V_exists():
    """|V| = |Phi B| <= s sqrt(M) max_j |B_j| (gram_f16.hip: v_bound_kernel): every row of Phi has norm s sqrt(M) because
    cos^2 + sin^2 = 1 per feature (SCFGP/SCFGP.py:98,102), the rest is Cauchy-Schwarz."""
    from oracle import scfgp_oracle as O
    N, D, S, M = 3000, 6, 5, 40
    seed = 0x5CF8
    X = synth.make_X(seed, N, D)
    p = synth.make_params(seed + 2, D, S, M, abc=(-1.0, 0.3, -1.0))
    Phi = O.feature_map(X, p, D, S, M); K = Phi.shape[1]
    s = np.exp(p[1]) * np.sqrt(2.0 / M)
    assert np.allclose(np.linalg.norm(Phi, axis=1), s * np.sqrt(S + M), rtol=1e-12)      # K / 2 = S + M features in all
    B = np.linalg.inv(Phi.T @ Phi + (np.exp(2 * p[0]) + 1e-6) * np.eye(K))
    bound = s * np.sqrt(S + M) * np.linalg.norm(B, axis=0).max()
    assert np.abs(Phi @ B).max() <= bound
