"""Host side of the result records (reference ActiveLearning.py:310-327, 438-447, 693-705): the three json files of a round."""
import json
import types

import numpy as np


def test_record_files_are_json_dumps_of_the_record_lists():
    """`_records_json` encodes every record's shared fields once and assembles the three files from the pieces; the text must be
    `json.dumps` of the record lists character for character (NaN key-points, labeled items carrying their ground truth included),
    and the vectorised record rows must equal the per-item arithmetic of the reference (float32 mean + 1.25 * max)."""
    from active_learning.ActiveLearning import ActiveLearning
    from alphapose.utils.config import edict
    al = ActiveLearning.__new__(ActiveLearning)
    r = np.random.RandomState(1)
    n = 257
    al.labeled_id = list(range(0, n, 3))
    al.opt = types.SimpleNamespace(work_dir=None)
    al.cfg = edict({"DATASET": {"EVAL": {"TYPE": "x"}}})
    kp = (r.standard_normal((n, 51)) * 100).astype(np.float32)
    kp[5, 7] = np.nan
    side = np.concatenate([np.arange(n)[:, None] // 4 * 1.0, np.arange(n)[:, None] + 1000.0, r.uniform(0, 500, (n, 4)), r.uniform(0, 500, (n, 51))], 1)
    oks = r.uniform(0, 1, n)
    al._write_records(kp, oks, side)
    pred, ann, gt = al._records_json()
    assert pred == json.dumps(al.kpt_json) and ann == json.dumps(al.kpt_json_ann) and gt == json.dumps(al._gt_dict())
    for i in (0, 3, 4, 256):
        rec, k = al.kpt_json[i], kp[i]
        assert rec["score"] == float(np.mean(k[2::3]) + 1.25 * np.max(k[2::3])) and rec["keypoints"] == [float(x) for x in k] or i == 5
        assert rec["image_id"] == i // 4 and rec["id"] == 1000 + i and rec["bbox"] == side[i, 2:6].tolist() and rec["OKS"] == float(oks[i])
        assert al.GT_json[i]["keypoints"] == side[i, 6:].tolist()
        assert al.kpt_json_ann[i]["keypoints"] == (side[i, 6:].tolist() if i % 3 == 0 else rec["keypoints"])
    assert json.loads(gt)["annotations"][7]["GT_keypoints"] == side[7, 6:].tolist()


def test_record_files_reuse_the_data_set_fields_between_rounds():
    """The fields that come from the data set alone (box, ids, ground truth) are encoded once per run and reused while unchanged
    (`_records_fixed`); new predictions, a changed labeled set, a changed ground truth and a NaN in it must all still give json.dumps'
    text; an empty record list too."""
    from active_learning.ActiveLearning import ActiveLearning
    from alphapose.utils.config import edict
    al = ActiveLearning.__new__(ActiveLearning)
    r = np.random.RandomState(2)
    n = 64
    al.opt = types.SimpleNamespace(work_dir=None)
    al.cfg = edict({"DATASET": {"EVAL": {"TYPE": "x"}}})
    side = np.concatenate([np.arange(n)[:, None] * 1.0, np.arange(n)[:, None] + 7.0, r.uniform(0, 500, (n, 4)), r.uniform(0, 500, (n, 51))], 1)

    def check(labeled, side):
        al.labeled_id = labeled
        kp = (r.standard_normal((len(side), 51)) * 50).astype(np.float32)
        al._write_records(kp, r.uniform(0, 1, len(side)), side)
        pred, ann, gt = al._records_json()
        assert pred == json.dumps(al.kpt_json) and ann == json.dumps(al.kpt_json_ann) and gt == json.dumps(al._gt_dict())
    check([], side)
    first = al._records_fixed[1][4]
    check([1, 2, 40], side)
    assert al._records_fixed[1][4] is first                  # reused
    side2 = side.copy(); side2[3, 10] += 1.0
    check([1, 2, 40], side2)
    assert al._records_fixed[1][4] is not first and al._records_fixed[1][0] is al._records_fixed[1][0]
    side2[5, 8] = np.nan
    check([5], side2)
    check([5], side2)
    check([0], side[:1])
    check([], side[:0])
