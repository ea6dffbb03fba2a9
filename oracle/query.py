"""CPU restatement of the reference's query-selection arithmetic (test infrastructure only: imported by tests/).

Follows active_learning/ActiveLearning.py:467-476 (influence), :583-592 (diversity), :798-850 (core-set) with the
same scikit-learn calls the reference makes (scikit-learn is present in this image, so these ARE the pinned third-party
functions: sklearn.neighbors.KNeighborsTransformer, sklearn.metrics.pairwise_distances)."""
import numpy as np


def cosine_distance_sums(fvecs: np.ndarray) -> np.ndarray:
    from sklearn.neighbors import KNeighborsTransformer
    fvecs = np.asarray(fvecs, np.float64)
    knn = KNeighborsTransformer(mode="distance", metric="cosine", n_neighbors=len(fvecs) - 1)
    dist_mat = knn.fit_transform(fvecs)
    return np.asarray(np.sum(dist_mat, axis=1)).flatten()


def influence_scores(fvecs: np.ndarray) -> np.ndarray:
    s = cosine_distance_sums(fvecs)
    return (s - np.min(s)) / (np.max(s) - np.min(s))


def coreset_selection(embeddings, labeled_idx, uncertainty, query_size, mode="moks", moks=0.0, unc_lambda=1.0, rng=np.random):
    """ActiveLearning.py:798-850 with `self.*` turned into arguments."""
    from sklearn.metrics import pairwise_distances
    embeddings = np.asarray(embeddings, np.float64)
    uncertainty = np.array(uncertainty, np.float64)
    labeled_idx = np.asarray(labeled_idx, np.int64)
    query_list = []

    def update_distances(cluster_centers, encoding, min_distances=None):
        if len(cluster_centers) != 0:
            dist = pairwise_distances(encoding, encoding[cluster_centers], metric="euclidean")
            if min_distances is None:
                min_distances = np.min(dist, axis=1).reshape(-1, 1)
            else:
                min_distances = np.minimum(min_distances, dist)
        return min_distances

    def pick(min_distances):
        if len(labeled_idx) == 0:
            return rng.choice(np.arange(embeddings.shape[0])) if mode == "kcenter" else np.argmax(uncertainty)
        if mode == "kcenter":
            return np.argmax(min_distances.reshape(-1))
        if mode == "fixed":
            return np.argmax(min_distances.reshape(-1) + unc_lambda * uncertainty)
        return np.argmax((1 - moks) * min_distances.reshape(-1) + unc_lambda * moks * uncertainty)

    min_distances = update_distances(labeled_idx, embeddings, None)
    for _ in range(query_size):
        ind = pick(min_distances)
        min_distances = update_distances([ind], embeddings, min_distances)
        labeled_idx = np.concatenate([labeled_idx, [ind]], axis=0).astype(np.int32)
        uncertainty[ind] = 0
        query_list.append(int(ind))
    return query_list
