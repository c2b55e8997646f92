r"""Analysis helpers built on the HIP kernels against values the
reference produced (tests/golden/make_golden_metrics.py)."""
from pathlib import Path

import numpy as np
import pytest

from bblean_amd.fingerprints import make_fake_fingerprints, unpack_fingerprints

pytestmark = pytest.mark.gpu
GOLD = np.load(Path(__file__).parent / "golden" / "metrics.npz")


@pytest.mark.parametrize("c", [0, 1])
def test_metrics_match_reference(c):
    from bblean_amd.metrics import jt_dbi, jt_isim_chi, jt_isim_dunn
    from bblean_amd.similarity import estimate_jt_std, jt_sim_matrix_packed

    seed, n, take = (int(x) for x in GOLD[f"c{c}_case"])
    fps = make_fake_fingerprints(n, seed=seed, pack=True)
    sizes = GOLD[f"c{c}_sizes"]
    members = np.split(GOLD[f"c{c}_members"], np.cumsum(sizes)[:-1])
    assert len(members) == take
    clusters = [fps[m] for m in members]
    unpacked = [unpack_fingerprints(x) for x in clusters]
    got = [
        jt_isim_chi(clusters), jt_isim_chi(unpacked, input_is_packed=False),
        jt_dbi(clusters), jt_dbi(clusters, centrals="medoid"), jt_dbi(unpacked, input_is_packed=False),
        jt_isim_dunn(clusters), jt_isim_dunn(unpacked, input_is_packed=False),
        estimate_jt_std(fps, n_samples=40),
    ]
    want = GOLD[f"c{c}_values"]
    # same kernels results (bit-identical similarities) and the reference's float64 operation
    # order: identical up to the last bit; 1e-12 relative is the stated tolerance
    np.testing.assert_allclose(np.array(got, dtype=np.float64), want, rtol=1e-12, atol=0)
    assert np.array_equal(jt_sim_matrix_packed(fps[:37]), GOLD[f"c{c}_simmat"])


def test_dunn_pair_kernel_many_clusters():
    r"""`jt_isim_dunn`'s pair loop (reference metrics.py:186-199) is one launch here (a wave per pair): 90 clusters = 4 005 pairs
    against the reference's loop restated in NumPy (exact uint64 sums, float64 in the order of similarity.cpp:297-300)."""
    from bblean_amd.metrics import jt_isim_dunn
    from bblean_amd.similarity import jt_isim_packed

    rng = np.random.default_rng(5)
    fps = make_fake_fingerprints(3000, seed=77, pack=True)
    cuts = np.sort(rng.choice(np.arange(2, 3000, 2), 89, replace=False))  # (even cuts: no cluster of one row - its iSIM is NaN + a warning)
    clusters = np.split(fps, cuts)
    got = jt_isim_dunn(clusters)
    sums = [unpack_fingerprints(c).astype(np.uint64).sum(axis=0) for c in clusters]
    best = 1.0
    for i in range(len(clusters) - 1):
        for j in range(i + 1, len(clusters)):
            x = sums[i] + sums[j]
            n = len(clusters[i]) + len(clusters[j])
            s1, s2 = int(x.sum()), int((x * x).sum())
            a = np.float64(s2 - s1) / 2.0
            isim = 1.0 if s1 == 0 else a / ((a + np.float64(n * s1)) - np.float64(s2))
            best = min(best, 1 - isim)
    want = best / max(jt_isim_packed(c) for c in clusters)
    assert got == want
    assert jt_isim_dunn(clusters[:1]) == 1.0 / jt_isim_packed(clusters[0])
