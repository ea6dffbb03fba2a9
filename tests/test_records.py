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
        assert rec["score"] == float(np.float64(np.mean(k[2::3])) + 1.25 * np.float64(np.max(k[2::3]))) and rec["keypoints"] == [float(x) for x in k] or i == 5
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


def test_records_are_written_by_a_gated_host_thread(tmp_path):
    """`_write_records` hands the round's records to a host thread; while the main thread is host-critical (gate closed) the thread only
    advances at its 0.25 s check-point time-outs, `flush_records()` opens the gate and waits; the files are `json.dumps` of the lists,
    the next round's call waits for the previous one (files in round order), and an error in the thread is raised by the flush."""
    import os
    import time
    from active_learning.ActiveLearning import ActiveLearning
    from alphapose.utils.config import edict
    al = ActiveLearning.__new__(ActiveLearning)
    r = np.random.RandomState(3)
    n = 300
    al.labeled_id = [1, 5, 9]
    al.opt = types.SimpleNamespace(work_dir=str(tmp_path))
    al.cfg = edict({"DATASET": {"EVAL": {"TYPE": "x"}}})
    side = np.concatenate([np.arange(n)[:, None] * 1.0, np.arange(n)[:, None] + 7.0, r.uniform(0, 500, (n, 4)), r.uniform(0, 500, (n, 51))], 1)
    kp1, kp2 = (r.standard_normal((n, 51)) * 50).astype(np.float32), (r.standard_normal((n, 51)) * 50).astype(np.float32)
    al._host_critical(True)
    t0 = time.perf_counter()
    al._write_records(kp1, r.uniform(0, 1, n), side)
    assert time.perf_counter() - t0 < 0.2                        # the call itself only snapshots its arguments
    kp1[:] = 0                                                   # ... so the caller may reuse its buffers
    al.flush_records()
    assert al._records_job is None
    pred = open(os.path.join(tmp_path, "predicted_kpt.json")).read()
    assert pred == json.dumps(al.kpt_json) and al.kpt_json[3]["keypoints"][0] != 0.0
    assert open(os.path.join(tmp_path, "predicted_kpt_ann.json")).read() == json.dumps(al.kpt_json_ann)
    assert open(os.path.join(tmp_path, "GT_kpt.json")).read() == json.dumps(al._gt_dict())
    al._host_critical(False)
    al._write_records(kp2, r.uniform(0, 1, n), side)             # round 2 ...
    al._host_critical(True)                                      # (round 3's thread pauses at its first check-point: 0.25 s to look at the files)
    al._write_records(kp1, r.uniform(0, 1, n), side)             # ... is complete before round 3 starts
    assert json.loads(open(os.path.join(tmp_path, "predicted_kpt.json")).read())[0]["keypoints"] == kp2[0].astype(np.float64).tolist()
    al.flush_records()
    assert json.loads(open(os.path.join(tmp_path, "predicted_kpt.json")).read())[0]["keypoints"] == [0.0] * 51
    al.opt.work_dir = os.path.join(str(tmp_path), "predicted_kpt.json", "not-a-directory")
    al._write_records(kp2, r.uniform(0, 1, n), side)
    try:
        al.flush_records()
        raise AssertionError("the thread's error must surface")
    except (NotADirectoryError, FileExistsError, OSError):
        pass
    al.flush_records()                                           # raised once
