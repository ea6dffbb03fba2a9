"""Query selection on the device embeddings (reference: ActiveLearning.py:467-617, 798-850).

What the reference computes with sklearn on a host float64 ``fvecs_matrix`` after every evaluation pass:

  influence / diversity   row sums of the all-pairs cosine-distance matrix        -> ``vatl_cosine_rowsum``
  core-set                k-center greedy over Euclidean distances (+ uncertainty) -> ``vatl_kcenter_update`` / ``_pick``
  K-Means / weighted      sklearn ``KMeans(n_clusters=query_size, random_state=318)``  (stays on the host like in the
                          reference: it is a seeded third-party algorithm; sklearn must be importable)

The embeddings never leave the GPU for the first two; the greedy loop chains pick -> update launches on the stream
without host round trips and copies the selected indices back once.
"""
from __future__ import annotations

import numpy as np
import torch

import vatl_hip as vh


def minmax(v: np.ndarray) -> np.ndarray:
    """(v - min) / (max - min) exactly as the reference writes it (nan when all values are equal, like numpy)."""
    v = np.asarray(v, np.float64)
    with np.errstate(all="ignore"):
        return (v - np.min(v)) / (np.max(v) - np.min(v))


def cosine_distance_sums(emb: torch.Tensor) -> np.ndarray:
    """(n, D) device embeddings -> (n,) float64: np.sum(KNeighborsTransformer(mode='distance', metric='cosine',
    n_neighbors=n-1).fit_transform(emb), axis=1)   (ActiveLearning.py:471-473, 585-587)."""
    return vh.cosine_rowsum(emb.float().contiguous()).cpu().numpy()


def influence_scores(emb: torch.Tensor) -> np.ndarray:
    """Normalised influence score of the unlabeled items (ActiveLearning.py:467-476)."""
    n = emb.shape[0]
    if n in (0, 1):
        return np.zeros(n)
    return minmax(cosine_distance_sums(emb))


def diversity_queries(emb: torch.Tensor, candidate_list, query_size: int):
    """filter == 'Diversity' (ActiveLearning.py:583-592): candidates with the SMALLEST distance sums first."""
    score = cosine_distance_sums(emb)
    order = np.argsort(score, kind="stable")                     # sorted(dict.items(), key=score) is stable too
    return [int(candidate_list[i]) for i in order[:query_size]]


def coreset_selection(emb: torch.Tensor, labeled_idx, uncertainty: np.ndarray, query_size: int, mode: str = "moks", moks: float = 0.0,
                      unc_lambda: float = 1.0, rng=np.random):
    """k-center greedy (ActiveLearning.py:798-850).  ``emb`` (N, D) covers the whole pool, ``uncertainty`` (N,) is 0 on
    labeled items.  mode: 'kcenter' (no uncertainty term), 'fixed' (min_dist + lambda*unc), 'moks'
    ((1-moks)*min_dist + lambda*moks*unc).  Returns the selected pool indices in selection order."""
    emb = emb.float().contiguous()
    n = emb.shape[0]
    dev = emb.device
    labeled = np.asarray(labeled_idx, np.int64).reshape(-1)
    a, b = {"kcenter": (1.0, 0.0), "fixed": (1.0, float(unc_lambda)), "moks": (1.0 - float(moks), float(unc_lambda) * float(moks))}[mode]
    unc = torch.as_tensor(np.asarray(uncertainty, np.float64), device=dev).contiguous()
    min_dist = torch.empty(n, device=dev, dtype=torch.float64)
    sel = torch.zeros(max(query_size, 1), device=dev, dtype=torch.int32)
    have = labeled.size > 0
    if have:
        vh.kcenter_update(emb, torch.as_tensor(labeled, dtype=torch.int32, device=dev), min_dist, first=True)
    for step in range(query_size):
        if not have:                                             # no labeled item yet (reference: len(labeled_idx) == 0)
            if mode == "kcenter":
                sel[step] = int(rng.choice(np.arange(n)))
                unc[int(sel[step])] = 0.0
            else:
                vh.kcenter_pick(None, unc, 0.0, 1.0, sel, step, n)       # np.argmax(uncertainty)
        else:
            vh.kcenter_pick(min_dist, unc, a, b, sel, step, n)
        vh.kcenter_update(emb, sel[step:step + 1], min_dist, first=not have)
        have = True
    return [int(i) for i in sel[:query_size].cpu().numpy()]


def kmeans_queries(emb_np: np.ndarray, candidate_list, query_size: int, weight=None):
    """filters 'K-Means' / 'weighted' (ActiveLearning.py:553-582, 595-611): cluster, then the member closest to its
    centre represents each cluster.  Runs sklearn on the host exactly as the reference does."""
    try:
        from sklearn.cluster import KMeans
    except ImportError as e:                                     # pragma: no cover
        raise ValueError("Filter type is not supported (K-Means needs scikit-learn on the host)") from e
    emb_np = np.asarray(emb_np, np.float64)
    learner = KMeans(n_clusters=query_size, random_state=318)
    cluster_idxs = learner.fit_predict(emb_np, sample_weight=weight) if weight is not None else learner.fit_predict(emb_np)
    cluster_num = len(np.unique(cluster_idxs))
    centers = learner.cluster_centers_[cluster_idxs]
    dis = ((emb_np - centers) ** 2).sum(axis=1)
    picks = [np.arange(emb_np.shape[0])[cluster_idxs == i][dis[cluster_idxs == i].argmin()] for i in range(cluster_num)]
    return [int(candidate_list[i]) for i in picks], picks
