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
