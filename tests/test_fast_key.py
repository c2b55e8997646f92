r"""The steady-state tree kernel (bblean_amd/csrc/bb_tree_fast.inc) orders the similarities of a node compare by the
bit pattern of the correctly rounded float32 quotient inter / union instead of comparing exact fractions pairwise.
This checks the claim that makes that exact for 2048-bit rows (unions <= 4096): over ALL fractions the key is
strictly monotone in the fraction and equal for equal fractions, so np.argmax's result and its first-index
tie-break (reference bitbirch.py:320 on similarity.cpp:326-331's float64 values) are reproduced."""
import numpy as np


def test_float32_quotient_orders_all_fractions_exactly():
    un = np.arange(1, 4097, dtype=np.int64)
    parts = []
    for u in un:  # inter <= min(union, 2048): the intersection of two 2048-bit rows
        i = np.arange(0, min(u, 2048) + 1, dtype=np.int64)
        parts.append(np.stack([i, np.full_like(i, u)], axis=1))
    fr = np.concatenate(parts)
    inter, union = fr[:, 0], fr[:, 1]
    key = (inter.astype(np.float32) / union.astype(np.float32)).view(np.uint32).astype(np.int64)
    # exact order: sort by the float64 quotient the reference computes; ties (equal fractions) are exact there
    f64 = inter.astype(np.float64) / union.astype(np.float64)
    order = np.argsort(f64, kind="stable")
    k, q, a, b = key[order], f64[order], inter[order], union[order]
    same_fraction = a[1:] * b[:-1] == a[:-1] * b[1:]  # exact cross-multiplication
    assert ((q[1:] == q[:-1]) == same_fraction).all()  # float64 is exact on ties as well
    assert (k[1:][same_fraction] == k[:-1][same_fraction]).all()
    assert (k[1:][~same_fraction] > k[:-1][~same_fraction]).all()
    assert fr.shape[0] > 6_000_000
