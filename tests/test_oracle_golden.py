r"""Pins the CPU oracle (oracle/bb_oracle.c) and the host logic of bblean_amd.BitBirch
against fixtures produced by the reference itself (tests/golden/make_golden.py) and
against the known answers in the reference's own tests.  CPU only."""
from __future__ import annotations

import ctypes as C
import json
import hashlib
from pathlib import Path

import numpy as np
import pytest

from cases import TREE_CASES
from oracle_engine import OracleEngine, oracle_lib
from tree_cases import run_case

from bblean_amd import make_fake_fingerprints
from bblean_amd._merges import CRITERION_CODES

GOLD = Path(__file__).resolve().parent / "golden"
SIM = dict(np.load(GOLD / "similarity.npz"))
MRG = dict(np.load(GOLD / "merges.npz"))
SHAPES = [(10, 256), (1, 256), (7, 128), (51, 256), (255, 256), (33, 253), (64, 4), (5, 64)]


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_fake_generator_matches_reference():
    man = json.loads((GOLD / "manifest.json").read_text())
    for key, digest in man["fake_digests"].items():
        n, seed, nf = (int(x) for x in key.split("_"))
        assert _sha(make_fake_fingerprints(n, n_features=nf, seed=seed)) == digest
    # reference tests/test_fake_fps.py:4-30 (first row of the 20 x 32-bit case)
    fps = make_fake_fingerprints(20, n_features=32, seed=12620509540149709235, pack=False)
    assert fps.shape == (20, 32) and set(np.unique(fps)) <= {0, 1}


@pytest.mark.parametrize("k", range(len(SHAPES)))
def test_oracle_kernels_vs_reference(k):
    lib = oracle_lib()
    pre = f"s{k}_"
    arr, vec = SIM[pre + "arr"], SIM[pre + "vec"]
    n, nb = arr.shape
    pc = np.empty(n, dtype=np.uint32)
    lib.bbo_popcount_rows(arr.ctypes.data, n, nb, pc.ctypes.data)
    assert (pc == SIM[pre + "popcount_cpp"]).all() and (pc == SIM[pre + "popcount_py"]).all()
    sims = np.empty(n)
    lib.bbo_jt_arr_vec(arr.ctypes.data, n, nb, vec.ctypes.data, None, sims.ctypes.data, None, None)
    assert (sims == SIM[pre + "sims_cpp"]).all()  # bit-exact f64
    assert (sims == SIM[pre + "sims_py"]).all()
    zero = np.zeros(nb, dtype=np.uint8)
    lib.bbo_jt_arr_vec(arr.ctypes.data, n, nb, zero.ctypes.data, None, sims.ctypes.data, None, None)
    assert (sims == SIM[pre + "sims_zero_cpp"]).all()
    i1, i2 = C.c_int64(), C.c_int64()
    s1, s2 = np.empty(n), np.empty(n)
    lib.bbo_most_dissimilar(arr.ctypes.data, n, nb, nb * 8, C.byref(i1), C.byref(i2), s1.ctypes.data, s2.ctypes.data)
    assert [i1.value, i2.value] == SIM[pre + "md_idx"].tolist()
    assert (s1 == SIM[pre + "md_s1"]).all() and (s2 == SIM[pre + "md_s2"]).all()
    un = np.empty((n, nb * 8), dtype=np.uint8)
    lib.bbo_unpack(arr.ctypes.data, n, nb, nb * 8, un.ctypes.data)
    assert (un == np.unpackbits(arr, axis=-1)).all()
    back = np.empty((n, nb), dtype=np.uint8)
    lib.bbo_pack(un.ctypes.data, n, nb * 8, back.ctypes.data)
    assert (back == arr).all()
    ls = np.empty(nb * 8, dtype=np.uint64)
    lib.bbo_add_rows(un.ctypes.data, n, nb * 8, ls.ctypes.data)
    assert (ls == SIM[pre + "add_rows"]).all()
    isim = lib.bbo_isim_from_sum(ls.ctypes.data, nb * 8, n)
    exp = SIM[pre + "isim_cpp"][0]
    assert (np.isnan(isim) and np.isnan(exp)) or isim == exp
    assert (np.isnan(exp) and np.isnan(SIM[pre + "isim_py"][0])) or exp == SIM[pre + "isim_py"][0]
    cen = np.empty(nb, dtype=np.uint8)
    lib.bbo_centroid_from_sum(ls.ctypes.data, nb * 8, n, 1, cen.ctypes.data)
    assert (cen == SIM[pre + "centroid_py"]).all() and (cen == SIM[pre + "centroid_cpp"]).all()
    cu = np.empty(nb * 8, dtype=np.uint8)
    lib.bbo_centroid_from_sum(ls.ctypes.data, nb * 8, n, 0, cu.ctypes.data)
    assert (cu == SIM[pre + "centroid_unpacked_py"]).all()
    if n >= 2:
        assert lib.bbo_isim_radius_compl_from_sum(ls.ctypes.data, nb * 8, n) == SIM[pre + "radius_compl"][0]


def test_oracle_reference_known_answers():
    r"""Values asserted by the reference's tests/test_similarity.py."""
    lib = oracle_lib()
    fps = make_fake_fingerprints(10, seed=17408390758220920002)
    assert (fps == SIM["ka_fps10"]).all()
    pc = np.empty(10, dtype=np.uint32)
    lib.bbo_popcount_rows(fps.ctypes.data, 10, 256, pc.ctypes.data)
    assert pc.tolist() == [1137, 124, 558, 1159, 281, 323, 1264, 1252, 879, 631]  # :80-94
    sims = np.empty(10)
    lib.bbo_jt_arr_vec(fps.ctypes.data, 10, 256, fps[0].ctypes.data, None, sims.ctypes.data, None, None)
    expect = [1.0, 0.050833333333333, 0.234522942461763, 0.400854179377669, 0.128980891719745,
              0.130030959752322, 0.411522633744856, 0.411104548139398, 0.309090909090909,
              0.246826516220028]  # :137-170
    assert np.isclose(sims, expect).all() and (sims == SIM["ka_sims_first"]).all()
    i1, i2 = C.c_int64(), C.c_int64()
    s1, s2 = np.empty(10), np.empty(10)
    lib.bbo_most_dissimilar(fps.ctypes.data, 10, 256, 2048, C.byref(i1), C.byref(i2), s1.ctypes.data, s2.ctypes.data)
    assert (i1.value, i2.value) == (1, 2)  # :24-76
    assert (s1 == SIM["ka_md_s1"]).all() and (s2 == SIM["ka_md_s2"]).all()
    un = make_fake_fingerprints(100, seed=17408390758220920002, pack=False)
    ls = un.sum(0).astype(np.uint64)
    assert lib.bbo_isim_from_sum(ls.ctypes.data, 2048, 100) == 0.21824334501491158  # :173-204
    # disjoint -> 0, homogeneous -> 1, single -> NaN (:207-249)
    one = make_fake_fingerprints(1, seed=17408390758220920002, pack=False)
    both = np.concatenate((one, 1 - one)).sum(0).astype(np.uint64)
    assert lib.bbo_isim_from_sum(both.ctypes.data, 2048, 2) == 0
    eye = np.ones(2048, dtype=np.uint64)
    assert lib.bbo_isim_from_sum(eye.ctypes.data, 2048, 2048) == 0
    z = np.zeros(2048, dtype=np.uint64)
    assert lib.bbo_isim_from_sum(z.ctypes.data, 2048, 100) == 1.0
    h = np.full(2048, 100, dtype=np.uint64)
    assert lib.bbo_isim_from_sum(h.ctypes.data, 2048, 100) == 1.0
    assert np.isnan(lib.bbo_isim_from_sum(one.sum(0).astype(np.uint64).ctypes.data, 2048, 1))
    # unpack with n_features=2024 (:114-134)
    f = SIM["ka_fps2024"]
    un = np.empty((10, 2024), dtype=np.uint8)
    lib.bbo_unpack(f.ctypes.data, 10, 253, 2024, un.ctypes.data)
    assert (un == SIM["ka_unpack2024"]).all()


@pytest.mark.parametrize("crit", list(CRITERION_CODES))
def test_oracle_merge_truth_table(crit):
    lib = oracle_lib()
    tab = MRG["tol_table"]
    exp = MRG["accept_" + crit]
    for i in range(len(exp)):
        old = MRG["old_ls"][i].astype(np.uint64)
        new = (MRG["old_ls"][i] + MRG["nom_ls"][i]).astype(np.uint64)
        on, nn = int(MRG["old_n"][i]), int(MRG["nom_n"][i])
        got = lib.bbo_merge_accept(CRITERION_CODES[crit], float(MRG["thr"][i]), 0.05, tab.ctypes.data,
                                   tab.size, new.ctypes.data, on + nn, old.ctypes.data, on, nn, old.size)
        assert got == exp[i], (crit, i)


def test_tolerance_table_matches_reference_formula():
    from bblean_amd._merges import get_merge_accept_fn

    assert (get_merge_accept_fn("tolerance-diameter", 0.05).tolerance_table() == MRG["tol_table"]).all()


@pytest.mark.parametrize("case", TREE_CASES, ids=[c["name"] for c in TREE_CASES])
def test_oracle_tree_vs_reference(case):
    run_case(case, OracleEngine)


def test_refine_reference_known_answer():
    r"""tests/test_refine.py of the reference: first/last labels before and after."""
    from tree_cases import trees

    g = trees()
    assert g["refine_100_assign"][:12].tolist() == [1, 5, 6, 1, 1, 7, 8, 9, 1, 10, 1, 2]
    assert g["refine_100_refine_assign"][:12].tolist() == [2, 1, 1, 3, 3, 1, 4, 1, 3, 5, 3, 1]


def test_consistency_reference_known_answer():
    r"""tests/test_bb_consistency.py:20-38 top clusters."""
    from tree_cases import trees

    g = trees()
    sizes, mem = g["diam065_3000_sizes"], g["diam065_3000_members"]
    assert mem[: sizes[0]].tolist() == [2195, 2196, 2378, 2440, 2443, 2454, 2463, 2464, 2465, 2467, 2527, 2544]
