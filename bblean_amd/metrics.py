r"""Clustering metrics on Tanimoto similarity (mirror of reference `bblean/metrics.py`),
composed from the HIP kernels behind `bblean_amd.similarity`.

The reference evaluates these with O(k) / O(k^2) sequences of small NumPy calls
(`metrics.py:47-214`).  Here each cluster costs one arr-vec Tanimoto launch, the k x k centroid
distances of the Davies-Bouldin index come from ONE batched all-pairs launch
(`bbh_jt_best_match` with the full matrix), and the pair sums of the Dunn index are formed from
per-cluster linear sums (exact u64) instead of re-summing both clusters for every pair.  The
float64 arithmetic that follows is performed in the reference's order, so results are
bit-identical (tests/test_hip_metrics.py against reference-generated goldens).
"""
from __future__ import annotations

import numpy as np
from numpy.typing import NDArray

from bblean_amd.fingerprints import pack_fingerprints
from bblean_amd.similarity import (
    _sum_rows_u64,
    centroid as centroid_from_fps,
    centroid_from_sum,
    jt_best_match_packed,
    jt_isim_from_sum,
    jt_isim_medoid,
    jt_isim_packed,
    jt_isim_unpacked,
    jt_sim_packed,
)

__all__ = ["jt_isim_chi", "jt_isim_dunn", "jt_dbi"]


def _calc_centrals(
    cluster_fps: list[NDArray[np.uint8]],
    kind: str,
    input_is_packed: bool = True,
    n_features: int | None = None,
    pack: bool = True,
) -> list[NDArray[np.uint8]]:
    r"""(metrics.py:23-44)"""
    if kind == "medoid":
        return [jt_isim_medoid(c, input_is_packed=input_is_packed, n_features=n_features, pack=pack)[1]
                for c in cluster_fps]
    if kind == "centroid":
        return [centroid_from_fps(c, input_is_packed=input_is_packed, n_features=n_features, pack=pack)
                for c in cluster_fps]
    raise ValueError(f"Unknown arg {kind} use 'medoids|centroids'")


def jt_isim_chi(
    cluster_fps: list[NDArray[np.uint8]],
    all_fps_central: NDArray[np.uint8] | str = "centroid",
    centrals: list[NDArray[np.uint8]] | str = "centroid",
    input_is_packed: bool = True,
    n_features: int | None = None,
    verbose: bool = False,
) -> float:
    r"""Calinski-Harabasz index on the Tanimoto iSIM, higher is better (metrics.py:47-105)."""
    all_fps_num = sum(len(c) for c in cluster_fps)
    if isinstance(all_fps_central, str):
        if not all_fps_central == "centroid":
            raise NotImplementedError("Currently only 'centroid' implemented for CHI")
        total_linear_sum = sum(_sum_rows_u64(c, input_is_packed, n_features) for c in cluster_fps)
        all_fps_central = centroid_from_sum(total_linear_sum, all_fps_num)
    if isinstance(centrals, str):
        if not centrals == "centroid":
            raise NotImplementedError("Currently only 'centroid' implemented for CHI")
        centrals = _calc_centrals(cluster_fps, centrals, input_is_packed, n_features)
    elif not input_is_packed:
        centrals = [pack_fingerprints(c) for c in centrals]
    clusters_num = len(cluster_fps)
    if not input_is_packed:
        cluster_fps = [pack_fingerprints(c) for c in cluster_fps]
    if clusters_num <= 1:
        return 0
    # similarities of every central to the global central: one launch for all clusters
    to_global = jt_sim_packed(np.stack(centrals), all_fps_central)
    wcss = 0.0
    bcss = 0.0
    for i, (central, clust) in enumerate(zip(centrals, cluster_fps)):
        bcss += len(clust) * (1 - to_global[i].item()) ** 2
        d = 1 - jt_sim_packed(clust, central)
        wcss += np.dot(d, d)
    return bcss * (all_fps_num - clusters_num) / (wcss * (clusters_num - 1))


def jt_dbi(
    cluster_fps: list[NDArray[np.uint8]],
    centrals: list[NDArray[np.uint8]] | str = "centroid",
    input_is_packed: bool = True,
    n_features: int | None = None,
    verbose: bool = False,
) -> float:
    r"""Davies-Bouldin index on the Tanimoto distance, lower is better (metrics.py:108-159)."""
    if isinstance(centrals, str):
        centrals = _calc_centrals(cluster_fps, centrals, input_is_packed, n_features)
    elif not input_is_packed:
        centrals = [pack_fingerprints(c) for c in centrals]
    if not input_is_packed:
        cluster_fps = [pack_fingerprints(c) for c in cluster_fps]
    fps_num = 0
    S: list[float] = []
    for central, clust_fps in zip(centrals, cluster_fps):
        size = len(clust_fps)
        S.append(np.sum(1 - jt_sim_packed(clust_fps, central)) / size)
        fps_num += size
    if fps_num == 0:
        return 0
    # all central-to-central similarities in one batched launch (the reference loops k^2 calls)
    cmat = np.stack(centrals)
    _, _, _, sims = jt_best_match_packed(cmat, cmat, return_sims=True)
    assert sims is not None
    numerator = 0.0
    for i in range(len(centrals)):
        max_d = 0.0
        for j in range(len(centrals)):
            if i == j:
                continue
            Mij = 1 - sims[i, j].item()
            max_d = max(max_d, (S[i] + S[j]) / Mij)
        numerator += max_d
    return numerator / fps_num


def jt_isim_dunn(
    cluster_fps: list[NDArray[np.uint8]],
    input_is_packed: bool = True,
    n_features: int | None = None,
    verbose: bool = False,
) -> float:
    r"""Dunn index variant of the BitBIRCH article, higher is better (metrics.py:163-214)."""
    if input_is_packed:
        D = [jt_isim_packed(clust) for clust in cluster_fps]
    else:
        D = [jt_isim_unpacked(clust) for clust in cluster_fps]
    max_d = max(D)
    if max_d == 0:
        return 1
    sums = [_sum_rows_u64(c, input_is_packed, n_features) for c in cluster_fps]  # exact column sums, once
    sizes = [len(c) for c in cluster_fps]
    min_d = 1.00
    for i in range(len(cluster_fps) - 1):
        for j in range(i + 1, len(cluster_fps)):
            dij = 1 - jt_isim_from_sum(sums[i] + sums[j], sizes[i] + sizes[j])
            min_d = min(dij, min_d)
    return min_d / max(D)
