"""End-to-end active-learning rounds on MI355X with the seeded SyntheticVideo dataset: the reference's driver
sequence (Run_active_learning.py:165-173: eval_and_query -> outcome -> ... until the pool is empty)."""
import types

import numpy as np
import pytest
import torch

from oracle import scorers
from tests.gpu_util import dev

pytestmark = pytest.mark.gpu


def _cfg():
    from alphapose.utils.config import edict
    return edict({
        "DATASET": {"TRAIN": {"TYPE": "SyntheticVideo", "NUM_ITEMS": 24, "TRACKS": 2}, "EVAL": {"TYPE": "SyntheticVideo", "NUM_ITEMS": 24, "TRACKS": 2}},
        "DATA_PRESET": {"TYPE": "simple", "SIGMA": 2, "NUM_JOINTS": 17, "IMAGE_SIZE": [256, 192], "HEATMAP_SIZE": [64, 48]},
        "MODEL": {"TYPE": "SimplePose", "PRETRAINED": "", "TRY_LOAD": "", "NUM_DECONV_FILTERS": [256, 256, 256], "NUM_LAYERS": 50},
        "LOSS": {"TYPE": "MSELoss"},
        "AE": {"Z_DIM": 4, "INPUT_DIM": 42, "PRETRAINED": "", "EPOCH": 2, "LR": 1e-3},
        "RETRAIN": {"BATCH_SIZE": 8, "BASE": 1, "OPTIMIZER": "AdamW", "LR": 2.5e-4, "ALPHA": 2, "WEIGHT_DECAY": 0.7, "LR_GAMMA": 0.99},
        "VAL": {"BATCH_SIZE": 10, "W_UNC": 0.01, "UNC_LAMBDA": 0.01, "QUERY_RATIO": [0.25, 0.5, 1.0]},
    })


@pytest.mark.parametrize("unc", ["THC+WPU", "TPC", "HP"])
def test_active_learning_rounds(unc, tmp_path):
    from active_learning import ActiveLearning
    opt = types.SimpleNamespace(work_dir=str(tmp_path), uncertainty=unc, representativeness="None", filter="None", strategy=unc, video_id="syn", get_prenext=True,
                                from_scratch=True, continual=True, num_gpu=1, onebyone=False, retrain_thresh=1, THCvsWPU="const")
    torch.manual_seed(0)
    al = ActiveLearning(_cfg(), opt)
    assert al.dedup
    al.eval_and_query()
    # first evaluation: the stream scorer agrees with the per-item oracle on the produced heat-maps
    ds = al.eval_dataset
    al.model.eval()
    with torch.no_grad():
        hm = al.model(torch.stack([ds[i][1][0] for i in range(4)]).to(dev())).cpu().numpy()
    for i in range(4):
        d = scorers.decode_heatmaps(hm[i], ds.bbox[i])
        np.testing.assert_allclose(al.keypoints[i].reshape(17, 3)[:, :2], d["coords"], rtol=1e-4, atol=1e-4)
    assert len(al.labeled_id) == 6 and len(al.unlabeled_id) == 18
    # result records (ActiveLearning.py:310-327): COCO-style dicts, written where the driver expects them
    import json, os
    al.flush_records()                           # written by a host thread: complete at the next entry point / outcome() / flush_records()
    recs = json.load(open(os.path.join(opt.work_dir, "predicted_kpt.json")))
    assert len(recs) == 24 and set(recs[0]) == {"bbox", "image_id", "id", "score", "category_id", "keypoints", "GT_keypoints", "OKS"}
    np.testing.assert_allclose(recs[3]["keypoints"], al.keypoints[3], rtol=1e-6)
    sc = np.asarray(recs[3]["keypoints"])[2::3]
    np.testing.assert_allclose(recs[3]["score"], sc.mean() + 1.25 * sc.max(), rtol=1e-5)
    assert recs[5]["id"] == int(ds[5][6]) and recs[5]["image_id"] == int(ds[5][5])
    assert os.path.exists(os.path.join(opt.work_dir, "predicted_kpt_ann.json"))
    gt = json.load(open(os.path.join(opt.work_dir, "GT_kpt.json")))                            # COCO layout (save_GT_dict :693-705)
    assert set(gt) == {"images", "categories", "annotations"} and len(gt["annotations"]) == 24 and gt["categories"][0]["name"] == "person"
    assert gt["annotations"][3]["keypoints"] == recs[3]["GT_keypoints"] and {im["id"] for im in gt["images"]} == {r["image_id"] for r in recs}
    assert al.performance[0]["AP"] is None and 0.0 <= al.performance[0]["mOKS"] <= 1.0         # no COCO API in this image: device mOKS only
    if unc != "HP":
        u = al.uncertainty_dict["Round0"]
        picked = al.query_list_list["Round0"]
        score = al._total_score(np.array([[*(u[i] if isinstance(u[i], list) else [u[i], 0])] for i in range(24)]))
        assert set(picked) == set(np.argsort(-score, kind="stable")[:6].tolist())              # top-k by normalised uncertainty
    rounds = 0
    result = None
    while result is None and rounds < 6:
        result = al.outcome()
        rounds += 1
        if result is None:
            assert np.isfinite(al.last_train_loss)
            if "WPU" in unc:                                   # the AE was re-initialised and fine-tuned on the labeled poses
                assert np.isfinite(al.last_ae_loss) and al.last_ae_loss > 0
            al.eval_and_query()
    assert result is not None and len(result) == 20                                            # the reference's 20-tuple
    assert len(al.unlabeled_id) == 0 and sorted(al.labeled_id) == list(range(24))
    assert result[0][-1] == 100.0                                                              # label percentage reaches 100


def test_unsupported_strategies_raise():
    from active_learning import ActiveLearning
    base = dict(uncertainty="THC_L1", representativeness="None", filter="None", from_scratch=True)
    for bad in (dict(uncertainty="Nope"), dict(filter="Spectral"), dict(representativeness="Density")):
        with pytest.raises(ValueError):
            ActiveLearning(_cfg(), types.SimpleNamespace(**{**base, **bad}))


def test_query_selection_kernels_match_sklearn_restatement():
    """influence / diversity (cosine-distance row sums) and k-center-greedy core-set on the
    device embeddings against the reference's own sklearn calls (oracle/query.py)."""
    import numpy as np
    import torch
    from active_learning import query as Q
    from oracle import query as OQ
    r = np.random.RandomState(7)
    n, d = 300, 2048
    base = np.abs(r.standard_normal((12, d))).astype(np.float32)                       # clustered, non-negative like ReLU + GAP features
    emb = (base[r.randint(0, 12, n)] + 0.3 * np.abs(r.standard_normal((n, d)))).astype(np.float32)
    dev = torch.device("cuda:0")
    e = torch.from_numpy(emb).to(dev)
    want = OQ.cosine_distance_sums(emb)
    got = Q.cosine_distance_sums(e)
    np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(Q.influence_scores(e), OQ.influence_scores(emb), rtol=1e-7, atol=1e-9)
    cand = list(range(100, 300))
    div = Q.diversity_queries(e[100:], cand, 15)
    order = np.argsort(OQ.cosine_distance_sums(emb[100:]), kind="stable")[:15]
    assert div == [cand[i] for i in order]
    unc = np.zeros(n); unlabeled = np.arange(40, n); unc[unlabeled] = r.random_sample(len(unlabeled))
    labeled = np.arange(40)
    for mode, moks, lam in (("moks", 0.6, 1.0), ("fixed", 0.0, 0.5), ("kcenter", 0.0, 1.0)):
        want_q = OQ.coreset_selection(emb, labeled, unc.copy(), 25, mode, moks, lam)
        got_q = Q.coreset_selection(e, labeled, unc.copy(), 25, mode, moks, lam)
        assert got_q == want_q, (mode, got_q, want_q)
    # empty labeled pool: the first pick is arg-max of the uncertainty
    assert Q.coreset_selection(e, [], unc.copy(), 5, "moks", 0.5, 1.0) == OQ.coreset_selection(emb, [], unc.copy(), 5, "moks", 0.5, 1.0)


@pytest.mark.parametrize("rep,flt", [("Influence", "Coreset"), ("Influence", "None"), ("None", "Diversity"), ("Random", "Random"),
                                     ("None", "K-Means"), ("Influence", "weighted")])
def test_active_learning_representativeness_and_filters(rep, flt):
    """Two rounds with every representativeness / filter combination family of ActiveLearning.py:465-617; for the
    device-side ones (Influence, Coreset, Diversity) the queried set is re-derived from the embeddings with the
    sklearn restatement."""
    from active_learning import ActiveLearning
    from oracle import query as OQ
    opt = types.SimpleNamespace(uncertainty="THC_L1", representativeness=rep, filter=flt, strategy="THC_L1", video_id="syn", get_prenext=True,
                                from_scratch=True, continual=True, num_gpu=1, onebyone=False, retrain_thresh=1, THCvsWPU="const", fixed_lambda=False)
    torch.manual_seed(0); np.random.seed(0)
    al = ActiveLearning(_cfg(), opt)
    al.eval_and_query()
    q0 = al.query_list_list["Round0"]
    assert len(q0) == 6 and len(set(q0)) == 6 and len(al.labeled_id) == 6
    assert sorted(al.retrain_id) == sorted(q0)
    if rep == "Influence":
        ds = al.eval_dataset
        al.model.eval()
        with torch.no_grad():
            emb = al.model.get_embedding(torch.stack([ds[i][1][0] for i in range(24)]).to(dev())).cpu().numpy()
        inf = al.influence_dict["Round0"]
        np.testing.assert_allclose([inf[i] for i in range(24)], OQ.influence_scores(emb), rtol=1e-6, atol=1e-9)
    assert al.outcome() is None                                   # fine-tune on the queried items
    al.eval_and_query()
    q1 = al.query_list_list["Round1"]
    assert len(q1) == 6 and not (set(q1) & set(q0)) and len(al.labeled_id) == 12
    if flt == "Coreset":                                          # second round has labeled centres: re-derive with sklearn
        ds = al.eval_dataset
        al.model.eval()
        with torch.no_grad():
            emb = al.model.get_embedding(torch.stack([ds[i][1][0] for i in range(24)]).to(dev())).cpu().numpy()
        assert set(q1).isdisjoint(q0)
        assert len(set(q1)) == 6
