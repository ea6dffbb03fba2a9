"""Host-side mirror of the reference's plugin surface, CPU only."""
import numpy as np
import pytest
import torch

from oracle import synth


def _cfgs():
    from alphapose.utils.config import edict
    cfg = edict({"TYPE": "SimplePose", "PRETRAINED": "", "TRY_LOAD": "", "NUM_DECONV_FILTERS": [256, 256, 256], "NUM_LAYERS": 50})
    preset = edict({"TYPE": "simple", "SIGMA": 2, "NUM_JOINTS": 17, "IMAGE_SIZE": [256, 192], "HEATMAP_SIZE": [64, 48]})
    return cfg, preset


def test_registry_behaviour():
    from alphapose.models import builder
    from alphapose.utils import Registry, build_from_cfg
    assert builder.SPPE.get("SimplePose") is not None
    assert builder.LOSS.get("MSELoss") is torch.nn.MSELoss
    with pytest.raises(KeyError):
        build_from_cfg({"TYPE": "Nope"}, builder.SPPE)
    r = Registry("x")

    @r.register_module
    class A:
        def __init__(self, P=1, Q=2):
            self.P, self.Q = P, Q
    with pytest.raises(KeyError):
        r.register_module(A)
    with pytest.raises(TypeError):
        r.register_module(3)
    a = build_from_cfg({"TYPE": "A", "P": 5}, r, {"Q": 7, "P": 9})
    assert (a.P, a.Q) == (5, 7)                     # explicit keys win over defaults
    assert isinstance(builder.build_loss({"TYPE": "MSELoss"}), torch.nn.MSELoss)


def test_simplepose_state_dict_is_checkpoint_compatible(golden_simplepose):
    from alphapose.models import builder
    cfg, preset = _cfgs()
    m = builder.build_sppe(cfg, preset_cfg=preset)
    sd = m.state_dict()
    assert list(sd.keys()) == list(golden_simplepose["keys"])                     # the reference module's keys, in order
    assert [str(tuple(v.shape)) for v in sd.values()] == list(golden_simplepose["shapes"])
    assert len(sd) == 338
    m.load_state_dict(synth.state_dict_for(m), strict=True)
    for attr in ("preact", "deconv_layers", "final_layer", "_initialize", "get_embedding"):
        assert hasattr(m, attr)
    groups = [list(m.final_layer.parameters()), list(m.preact.parameters()), list(m.deconv_layers.parameters())]
    assert sum(len(g) for g in groups) == len(list(m.parameters()))                 # optimizer param groups cover the model
    m._initialize()
    assert float(m.final_layer.bias.abs().sum()) == 0.0


def test_forward_without_gpu_fails_loudly():
    from alphapose.models import builder
    cfg, preset = _cfgs()
    m = builder.build_sppe(cfg, preset_cfg=preset).eval()
    with pytest.raises(Exception) as e:
        m(torch.zeros(1, 3, 256, 192))
    assert "fallback" in str(e.value) or "HIP" in str(e.value)


def test_config_loader(tmp_path):
    from alphapose.utils.config import update_config
    p = tmp_path / "c.yaml"
    p.write_text("MODEL:\n  TYPE: 'SimplePose'\n  NUM_LAYERS: 50\nVAL:\n  QUERY_RATIO: [0.05, 0.1]\n")
    cfg = update_config(str(p))
    assert cfg.MODEL.TYPE == "SimplePose" and cfg["MODEL"]["NUM_LAYERS"] == 50 and cfg.VAL.QUERY_RATIO[1] == 0.1


def test_driver_imports_resolve():
    """Every name scripts/Run_active_learning.py:38-46 imports from `alphapose` / `active_learning` exists with our
    packages first on the path (third-party imports of the driver are the environment's business)."""
    from alphapose.models import builder                                                   # noqa: F401
    from alphapose.utils.config import update_config                                       # noqa: F401
    from alphapose.utils.metrics import evaluate_mAP, calc_accuracy, DataLogger            # noqa: F401
    from alphapose.utils.transforms import flip, flip_heatmap, get_func_heatmap_to_coord   # noqa: F401
    from alphapose.utils.vis import vis_frame_fast, vis_frame
    from active_learning import ActiveLearning                                             # noqa: F401
    from active_learning.al_metric import plot_learning_curves, compute_alc                # noqa: F401
    with pytest.raises(NotImplementedError):
        vis_frame(None, None)
    with pytest.raises(NotImplementedError):
        vis_frame_fast(None, None)
    for name in ("SimplePose", "FastPose", "PoseHighResolutionNet"):
        assert builder.SPPE.get(name) is not None
    for name in ("MSELoss", "L1JointRegression"):
        assert builder.LOSS.get(name) is not None


def test_engine_chunking_respects_the_32bit_offset_limit():
    """Batches are cut so that no tensor of a launch reaches 2^30 elements, in balanced chunks."""
    from alphapose.models import hip_engine as h
    assert h._chunk_limit((256, 192)) == h.MAX_CHUNK == 1024
    lim = h._chunk_limit((384, 288))
    assert lim * 192 * 144 * 64 < 2 ** 30 <= (lim + 1) * 192 * 144 * 64 and lim == 606
    assert h._chunks(1080, (256, 192)) == [(0, 540), (540, 1080)]
    assert h._chunks(1024, (384, 288)) == [(0, 512), (512, 1024)]
    assert h._chunks(5, (256, 192)) == [(0, 5)]
    cuts = h._chunks(2000, (384, 288))
    assert cuts[0][0] == 0 and cuts[-1][1] == 2000 and all(b - a <= lim for a, b in cuts) and all(x[1] == y[0] for x, y in zip(cuts, cuts[1:]))


def test_crop_geometry_host_mirror_matches_reference_fixture():
    """The host side of the crop producer (alphapose.utils.bbox / transforms of this package) against the outputs of the
    reference's own functions in tests/golden/crop.npz — per item and batched."""
    import os
    import numpy as np
    from alphapose.utils import bbox as B, transforms as T
    from oracle import synth
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "crop.npz"))
    box, rot = synth.crop_cases(24)
    for tag, (h, w) in (("a", (256, 192)), ("b", (384, 288))):
        cb, sb = B.box_to_center_scale_batch(box, float(w) / h)
        assert cb.dtype == np.float32 and np.array_equal(cb, g[f"{tag}_center"]) and np.array_equal(sb, g[f"{tag}_scale"])
        tb = T.get_affine_transform_batch(cb, sb, rot, [w, h])
        np.testing.assert_allclose(tb, g[f"{tag}_trans"], rtol=1e-13, atol=1e-11)
        np.testing.assert_allclose(np.float32(B.center_scale_to_box_batch(cb, sb)), np.float32(g[f"{tag}_box"]), rtol=2e-7)
        for i, (xmin, ymin, xmax, ymax) in enumerate(box.tolist()):
            c, s = B._box_to_center_scale(xmin, ymin, xmax - xmin, ymax - ymin, float(w) / h)
            assert np.array_equal(c, cb[i]) and np.array_equal(s, sb[i])
            t = T.get_affine_transform(c, s, rot[i], [w, h])
            assert np.array_equal(t, tb[i])
            np.testing.assert_allclose(T.affine_transform(np.float32([xmin, ymax]), t), g[f"{tag}_pt"][i], rtol=1e-12, atol=1e-9)
            np.testing.assert_allclose(np.float32(B._center_scale_to_box(c, s)), np.float32(g[f"{tag}_box"][i]), rtol=2e-7)
        # inverse map: forward o inverse = identity
        inv = T.invert_affine_batch(tb)
        for f, i2 in zip(tb, inv):
            full = np.vstack([f, [0, 0, 1]]) @ np.vstack([i2, [0, 0, 1]])
            np.testing.assert_allclose(full, np.eye(3), atol=1e-9)
    frame = synth.u8_frame(40, 56)
    assert np.array_equal(T.im_to_torch(frame).numpy(), g["tensor_bright"])
    assert np.array_equal(T.im_to_torch((frame > 250).astype(np.uint8)).numpy(), g["tensor_dark"])


def test_simple_transform_without_gpu_fails_loudly():
    import numpy as np
    import pytest
    import torch
    import vatl_hip as vh
    from alphapose.utils.presets import SimpleTransform
    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    ds = type("D", (), {"joint_pairs": [[1, 2]]})()
    st = SimpleTransform(ds, scale_factor=0, add_dpg=False, input_size=[256, 192], output_size=[64, 48], rot=0, sigma=2, train=False)
    with pytest.raises(vh.VatlError):
        st.test_transform(np.zeros((48, 64, 3), np.uint8), [4, 4, 30, 40])


@pytest.mark.parametrize("name,count", [("fastpose", 348), ("hrnet", 1754)])
def test_fastpose_hrnet_state_dicts_are_checkpoint_compatible(name, count):
    """Keys (in order) and shapes of the package's modules against those of the reference's own modules
    (tests/golden/fastpose_hrnet.npz): a strict load_state_dict of a reference checkpoint must work."""
    import os
    from alphapose.models import builder
    from alphapose.utils.config import edict
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "fastpose_hrnet.npz"))
    hr = {"TYPE": "PoseHighResolutionNet", "PRETRAINED": "", "TRY_LOAD": "", "NUM_LAYERS": 50, "FINAL_CONV_KERNEL": 1, "PRETRAINED_LAYERS": ["*"],
          "STAGE2": {"NUM_MODULES": 1, "NUM_BRANCHES": 2, "NUM_BLOCKS": [4, 4], "NUM_CHANNELS": [32, 64], "BLOCK": "BASIC", "FUSE_METHOD": "SUM"},
          "STAGE3": {"NUM_MODULES": 4, "NUM_BRANCHES": 3, "NUM_BLOCKS": [4, 4, 4], "NUM_CHANNELS": [32, 64, 128], "BLOCK": "BASIC", "FUSE_METHOD": "SUM"},
          "STAGE4": {"NUM_MODULES": 3, "NUM_BRANCHES": 4, "NUM_BLOCKS": [4, 4, 4, 4], "NUM_CHANNELS": [32, 64, 128, 256], "BLOCK": "BASIC", "FUSE_METHOD": "SUM"}}
    cfg = edict({"TYPE": "FastPose", "PRETRAINED": "", "TRY_LOAD": "", "NUM_LAYERS": 50} if name == "fastpose" else hr)
    m = builder.build_sppe(cfg, preset_cfg=_cfgs()[1])
    sd = m.state_dict()
    assert len(sd) == count
    assert list(sd.keys()) == [str(k) for k in g[f"{name}_keys"]]
    assert [str(tuple(v.shape)) for v in sd.values()] == [str(s) for s in g[f"{name}_shapes"]]
    m.load_state_dict(synth.state_dict_for(m), strict=True)
    if name == "fastpose":
        for attr in ("conv_out", "preact", "suffle1", "duc1", "duc2", "get_embedding", "_initialize"):
            assert hasattr(m, attr)


@pytest.mark.parametrize("name", ["Posetrack21", "JRDB2022"])
def test_coco_video_datasets_follow_the_reference_annotation_rules(tmp_path, name):
    """builder.build_dataset on a COCO-format json (posetrack21.py:40-129 / jrdb2022.py): invalid persons dropped, boxes
    xywh -> xyxy (-1) and clipped, one item per person, items sorted by the reference's composite id, track ids as built there."""
    from alphapose.models import builder
    from alphapose.utils.config import edict
    ann, frames, kept = synth.write_coco_video(str(tmp_path), n_frames=3, tracks=2, fmt="posetrack" if name == "Posetrack21" else "jrdb")
    cfg = edict({"TYPE": name, "ROOT": str(tmp_path), "IMG_PREFIX": "", "ANN": ann})
    ds = builder.build_dataset(cfg, preset_cfg=_cfgs()[1], train=False, get_prenext=True)
    if name == "JRDB2022":                                               # jrdb2022.py keeps a degenerate box (only xmax < xmin is dropped): person 50
        assert len(ds) == 7 and ds._labels[-1]["track_id"] == 50
        ds._labels.pop(); ds._items.pop()
    assert len(ds) == 6 and sorted(a["ann_id"] for a in ds._labels) == kept
    digits = 2 if name == "Posetrack21" else 3
    ids = [a["id"] for a in ds._labels]
    assert ids == sorted(ids) and all(a["id"] == int(str(a["ann_id"])[-digits:] + str(a["img_id"])) for a in ds._labels)
    first = ds._labels[0]
    assert first["track_id"] == ("70" if name == "Posetrack21" else 0)
    x, y, w, h = 10.0, 8.0, 40.0, 70.0                                   # track 0, frame 0 in write_coco_video
    assert first["bbox"] == (x, y, x + w - 1, y + h - 1) and first["width"] == 160 and first["height"] == 120
    assert first["joints_3d"].shape == (17, 3, 2) and first["joints_3d"][0, 0, 1] == 1.0
    assert ds._items[0]["path"].endswith("000000.png") and ds._items[0]["keypoint"] == first["keypoint"]
    # the same person in consecutive frames is id-adjacent: prev / next flags follow the track
    assert [ds._neighbour(i, -1) for i in range(6)] == [False, True, True, False, True, True]
    arena, where = ds._frames_for([ds._labels[0]["frame"], ds._labels[1]["frame"], ds._labels[0]["frame"]]) if torch.cuda.is_available() else (None, None)
    with pytest.raises(AssertionError):
        import json, os
        bad = json.load(open(os.path.join(str(tmp_path), ann)))
        bad["categories"][0]["name"] = "cat"
        json.dump(bad, open(os.path.join(str(tmp_path), "annotations", "bad.json"), "w"))
        builder.build_dataset(edict({"TYPE": name, "ROOT": str(tmp_path), "IMG_PREFIX": "", "ANN": "annotations/bad.json"}), preset_cfg=_cfgs()[1], train=False)


def test_image_datasets_registered_and_parse_coco_json(tmp_path):
    """Mscoco / Mpii: registry names, the reference's person filters (crowd, empty key-points, zero area) and the 30-image
    SHORTEN switch of mscoco.py:53-55."""
    import json, os
    from PIL import Image
    from alphapose.models import builder
    from alphapose.utils.config import edict
    root = str(tmp_path)
    os.makedirs(os.path.join(root, "val2017")); os.makedirs(os.path.join(root, "annotations"))
    images, anns = [], []
    for i in range(34):
        name = f"{i + 1:012d}.png"
        Image.fromarray(synth.u8_frame(48, 64, 90 + i)).save(os.path.join(root, "val2017", name))
        images.append({"id": i + 1, "coco_url": f"http://images.cocodataset.org/val2017/{name}", "file_name": name, "width": 64, "height": 48})
        kp = [10.0, 12.0, 2] * 17
        anns.append({"id": 100 + i, "image_id": i + 1, "category_id": 1, "iscrowd": 0, "bbox": [4.0, 5.0, 30.0, 35.0], "area": 900.0, "num_keypoints": 17, "keypoints": kp})
    anns.append({"id": 900, "image_id": 1, "category_id": 1, "iscrowd": 1, "bbox": [4.0, 5.0, 30.0, 35.0], "area": 900.0, "num_keypoints": 17, "keypoints": [10.0, 12.0, 2] * 17})
    anns.append({"id": 901, "image_id": 1, "category_id": 1, "iscrowd": 0, "bbox": [4.0, 5.0, 30.0, 35.0], "area": 0.0, "num_keypoints": 17, "keypoints": [10.0, 12.0, 2] * 17})
    anns.append({"id": 902, "image_id": 2, "category_id": 1, "iscrowd": 0, "bbox": [4.0, 5.0, 30.0, 35.0], "area": 9.0, "num_keypoints": 0, "keypoints": [0, 0, 0] * 17})
    json.dump({"images": images, "annotations": anns, "categories": [{"id": 1, "name": "person"}]}, open(os.path.join(root, "annotations", "k.json"), "w"))
    ds = builder.build_dataset(edict({"TYPE": "Mscoco", "ROOT": root, "IMG_PREFIX": "val2017", "ANN": "annotations/k.json"}), preset_cfg=_cfgs()[1], train=False)
    assert len(ds) == 30 and ds._labels[0]["bbox"] == (4.0, 5.0, 33.0, 39.0) and not ds.ID_SORTED_STREAM
    assert ds._items[3]["path"].endswith(os.path.join("val2017", "000000000004.png"))
    assert builder.DATASET.get("Mpii").num_joints == 16 and len(builder.DATASET.get("Mpii").joint_pairs) == 6


def test_uncertainty_option_strings_follow_the_reference_dispatch():
    """Run_active_learning.py:57 documents `HP, TPC, THC_L1, WPU_hybrid`; the reference matches by substring
    (ActiveLearning.py:345, 364, 371) and pairs THC with WPU only under the exact name (:403, :494)."""
    from active_learning.ActiveLearning import uncertainty_kind
    want = {"None": "None", "HP": "HP", "TPC": "TPC", "MPE": "MPE", "Margin": "Margin", "Entropy": "Entropy",
            "THC": "THC", "THC_L1": "THC", "THC_L2": "THC", "WPU": "WPU", "WPU_hybrid": "WPU", "WPU_raw": "WPU",
            "THC+WPU": "THC+WPU", "THC_L1+WPU_hybrid": "THC"}
    for name, kind in want.items():
        assert uncertainty_kind(name) == kind, name
    for bad in ("Nope", "VL4Pose", ""):
        with pytest.raises(ValueError):
            uncertainty_kind(bad)


def _bare_al(query_ratio, n=40):
    """An ActiveLearning object with the round state only (no GPU, no datasets): what outcome() works on."""
    import types
    from active_learning.ActiveLearning import ActiveLearning
    from alphapose.utils.config import edict
    al = object.__new__(ActiveLearning)
    al._depth = 0
    al.cfg = edict({"RETRAIN": {"BASE": 1, "ALPHA": 2}, "VAL": {"QUERY_RATIO": query_ratio}})
    al.opt = types.SimpleNamespace()
    al.query_ratio = list(query_ratio)
    al.eval_len = n
    al.query_sizes = [int(n * x) for x in query_ratio]
    al.query_size = al.query_sizes[0]
    al.round_cnt, al.is_early_stop, al.one_by_one, al.continual = 0, False, False, True
    al.unlabeled_id, al.labeled_id, al.moks_queried = list(range(n)), [], 0.0
    al.percentage, al.performance, al.performance_ann = [], [], []
    al.ospa_list, al.ospa_list_ann, al.combine_weight, al.uncertainty_mean, al.moksQ_list = [], [], [], [], []
    al.query_list_list, al.uncertainty_dict, al.influence_dict = {}, {}, {}
    al.spearmanr_list, al.corr_list = [], []
    al.true_labeled_dict, al.false_labeled_dict, al.true_unlabeled_dict, al.false_unlabeled_dict = {}, {}, {}, {}
    al.actual_finish = al.finished_minerror = al.finished_oursc = 100
    return al


def test_outcome_pads_an_early_stopped_video_like_the_reference():
    """ActiveLearning.py:168-178 + Run_active_learning.py:165-173, 211-244: replay do_al()'s loop with a stubbed evaluation that
    stops early in round 1; every per-round list of the 20-tuple must have len(query_ratio)+1 entries, the percentage axis
    must continue along the query schedule, and the un-clamped query sizes (:199-202) must be the reference's."""
    ratio = [0.05, 0.1, 0.2, 0.3, 0.4, 1]
    al = _bare_al(ratio)
    trained, evals = [], []

    def fake_eval():
        k = len(evals)
        evals.append(al.query_size)
        q = al.unlabeled_id[:max(al.query_size, 0)]
        al.percentage.append(len(al.labeled_id) / al.eval_len * 100)
        al.performance.append({"AP": 0.1 * k}); al.performance_ann.append({"AP": 0.2 * k})
        al.ospa_list.append(k); al.ospa_list_ann.append(10 + k); al.uncertainty_mean.append(1.0 / (k + 1))
        al.combine_weight.append(0.5 + k); al.moksQ_list.append(0.9)
        al.query_list_list[f"Round{al.round_cnt}"] = q
        al.labeled_id = sorted(set(al.labeled_id) | set(q))
        al.unlabeled_id = [i for i in al.unlabeled_id if i not in set(q)]
        if k == 1:
            al.is_early_stop, al.actual_finish = True, al.percentage[-1]
    al.eval_and_query = fake_eval
    al.retrain_model = lambda: trained.append(al.round_cnt)
    result = None
    for _ in range(20):                                              # do_al(): eval_and_query -> outcome until a result comes back
        al.eval_and_query()
        result = al.outcome()
        if result is not None:
            break
    assert result is not None and len(result) == 20
    assert trained == [0] and len(evals) == 2                        # one fine-tune, then the early stop ends the video
    assert evals == [2, 4 - 2]                                       # int(40*0.05), then query_sizes[1] - len(labeled) (un-clamped)
    full = len(ratio) + 1
    for idx in (0, 1, 2, 5, 7, 17, 18, 19):                          # percentages, performances(_ann), mean uncertainty, combine weight, ospa(_ann), moks
        assert len(result[idx]) == full, (idx, len(result[idx]))
    assert result[0] == [0.0, 5.0] + [r * 100 for r in ratio[1:]]    # the evaluated points, then query_ratio[round_cnt - 1] * 100
    assert result[1][2:] == [result[1][1]] * (full - 2) and result[18][2:] == [11] * (full - 2)
    assert result[14] == 5.0 and al.round_cnt == len(ratio)          # actual_finish; the padding loop advances round_cnt
    # save_result reads exactly these 20 positions (Run_active_learning.py:214-237)
    keys = dict(percentages=0, performances=1, performances_ann=2, query_list=3, uncertaity=4, mean_uncertaity=5, influence=6, combine_weight=7,
                spearmanr=8, corrcoef=9, true_labeled=10, true_unlabeled=11, false_labeled=12, false_unlabeled=13, actual_finish=14,
                finished_minerror=15, finished_oursc=16, ospa=17, ospa_ann=18, moks_queried=19)
    import json
    json.dumps({k: result[i] for k, i in keys.items()})              # serialisable, like result.json


def test_outcome_round_schedule_without_early_stop():
    ratio = [0.25, 0.5, 1.0]
    al = _bare_al(ratio, n=24)
    sizes = []

    def fake_eval():
        sizes.append(al.query_size)
        q = al.unlabeled_id[:al.query_size]
        al.labeled_id = sorted(set(al.labeled_id) | set(q))
        al.unlabeled_id = [i for i in al.unlabeled_id if i not in set(q)]
        for lst in (al.percentage, al.performance, al.performance_ann, al.ospa_list, al.ospa_list_ann, al.uncertainty_mean, al.combine_weight, al.moksQ_list):
            lst.append(len(al.labeled_id))
    al.eval_and_query = fake_eval
    al.retrain_model = lambda: None
    result = None
    while result is None:
        al.eval_and_query()
        result = al.outcome()
    assert sizes == [6, 6, 12, 12]                                   # query_sizes[k] - labeled; then the final evaluation of the fully labeled video
    assert len(result[1]) == len(ratio) + 1 and result[1][-1] == 24

    one = _bare_al(ratio, n=24)
    one.one_by_one = True
    one.eval_and_query = lambda: [lst.append(0) for lst in (one.percentage, one.performance, one.performance_ann, one.ospa_list, one.ospa_list_ann,
                                                            one.uncertainty_mean, one.combine_weight, one.moksQ_list)]
    one.eval_and_query()
    res = one.outcome()                                              # --onebyone: a single evaluation, padded to the full axis
    assert res is not None and len(res[1]) == len(ratio) + 1 and res[0] == [0, 25.0, 50.0, 100.0]


def test_host_augmentation_arithmetic_matches_reference_fixture():
    """tests/golden/hostaug.npz = the reference's SimpleTransform.half_body_transform / _integral_target_generator on seeded
    joints: same boxes, same number of random draws, same targets and weights (bit for bit: it is float32 numpy on the host)."""
    import os
    from tests.conftest import GOLDEN
    from alphapose.utils.presets.simple_transform import SimpleTransform

    class DS:
        joint_pairs = [[1, 2]]
        num_joints_half_body, prob_half_body = 8, 0.3
        upper_body_ids, lower_body_ids = (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10), (11, 12, 13, 14, 15, 16)
    g = np.load(os.path.join(GOLDEN, "hostaug.npz"))
    st = SimpleTransform(DS, scale_factor=0.3, add_dpg=False, input_size=[256, 192], output_size=[64, 48], rot=40, sigma=2, train=True)
    st.num_joints = 17
    assert g["hb_ok"].sum() > 20 and (~g["hb_ok"]).sum() > 3
    for i in range(len(g["hb_ok"])):
        np.random.seed(1000 + i)
        c, s = st.half_body_transform(g["hb_joints"][i], g["hb_vis"][i])
        assert np.random.rand() == g["hb_next_draw"][i]                          # exactly one draw consumed
        if not g["hb_ok"][i]:
            assert c is None and s is None
        else:
            assert np.array_equal(c, g["hb_center"][i]) and np.array_equal(s, g["hb_scale"][i]) and s.dtype == np.float32
    for nj in (17, 136, 133, 68):
        t, w = st._integral_target_generator(g[f"int{nj}_joints"], nj, 256, 192)
        assert np.array_equal(t, g[f"int{nj}_target"]) and np.array_equal(w, g[f"int{nj}_weight"]) and t.dtype == w.dtype == np.float32


def test_plan_key_sees_every_way_a_model_can_change():
    """`hip_engine._version_key` (what decides whether the packed weights of a plan are still the model's) re-validates a cached walk of
    the module tree instead of walking it on every call: it must still change for an in-place update, a re-assigned parameter, a replaced
    layer at any depth, an added and a removed sub-module, a buffer update — and must not change when nothing did."""
    from alphapose.models import builder, hip_engine
    cfg, preset = _cfgs()
    m = builder.build_sppe(cfg, preset_cfg=preset)
    ref = []
    for t in list(m.parameters()) + list(m.buffers()):
        ref += [t.data_ptr(), t._version]
    k0 = hip_engine._version_key(m, "cpu")
    assert sorted(k0[1:]) == sorted(ref) and hip_engine._version_key(m, "cpu") == k0
    keys = [k0]

    def changed():
        k = hip_engine._version_key(m, "cpu")
        assert k not in keys and hip_engine._version_key(m, "cpu") == k
        keys.append(k)
    with torch.no_grad():
        m.final_layer.weight.add_(1.0)
    changed()
    m.preact.bn1.running_mean.add_(1.0)                                  # a buffer
    changed()
    m.final_layer.bias = torch.nn.Parameter(torch.zeros(17))             # re-assigned parameter
    changed()
    m.final_layer = torch.nn.Conv2d(256, 17, 1)                          # replaced layer
    changed()
    m.preact.layer1[0].conv1 = torch.nn.Conv2d(64, 64, 1, bias=False)    # ... three levels down
    changed()
    before = keys[-1]
    m.extra = torch.nn.Linear(2, 2)                                      # added
    changed()
    del m.extra                                                          # removed: the previous state again
    assert hip_engine._version_key(m, "cpu") == before
    m.double()                                                           # new storage for every tensor
    changed()
    fl = m.final_layer                                                   # a module's parameter DICT swapped for another of the same length (round-5 advisor): the walk reads
    fl._parameters = {k: torch.nn.Parameter(v.detach().clone()) for k, v in fl._parameters.items()}   # the dict from the module at check time, not a remembered object
    changed()
    bn = m.preact.bn1
    bn._buffers = {k: (None if v is None else v.clone()) for k, v in bn._buffers.items()}
    changed()
    import copy
    r = copy.copy(m)                                                     # what DataParallel's replicate() does: the original's attribute dict, copied ...
    r.__dict__ = dict(m.__dict__)
    r._parameters, r._buffers, r._modules = dict(m._parameters), dict(m._buffers), {k: copy.deepcopy(v) for k, v in m._modules.items()}   # ... its own tensors
    kr = hip_engine._version_key(r, "cpu")
    assert kr != keys[-1] and hip_engine._version_key(m, "cpu") == keys[-1]       # the original's cached walk is not taken for the replica's


def test_parameter_guard_notices_writes_that_bump_no_version_counter(monkeypatch):
    """`p.data.add_(1)` / `bn.running_mean.data.zero_()` bump no torch version counter, so `_version_key` cannot see them (asserted: that
    is the hole) — the checksum guard of a plan must.  Driven here without a GPU: `vh.ChecksumTable` is replaced by a host table that
    evaluates the published formula (oracle/checksum.py, the one the GPU test pins the kernel to), `_ParamGuard` itself is the product's."""
    import weakref
    from alphapose.models import builder, hip_engine
    from oracle import checksum as oc
    import vatl_hip as vh

    class HostTable:
        def __init__(self, tensors):
            self.ts, self.n = list(tensors), len(tensors)

        def launch(self, out=None):
            return torch.tensor([np.int64(np.uint64(oc.checksum_tensor(t)).view(np.int64)) for t in self.ts], dtype=torch.int64)
    monkeypatch.setattr(vh, "ChecksumTable", HostTable)
    cfg, preset = _cfgs()
    m = builder.build_sppe(cfg, preset_cfg=preset)
    dev = torch.device("cpu")
    k0 = hip_engine._version_key(m, dev)
    g = hip_engine._ParamGuard(m, dev)
    assert len(g.names) == len(list(m.parameters())) + len(list(m.buffers())) and "preact.bn1.running_mean" in g.names
    m.__dict__["_vatl_plan"] = (k0, object(), g)
    g.launch(); g.check(wait=True)                                        # untouched: fine
    assert g.unguarded == []
    m.register_buffer("flags", torch.zeros(3, dtype=torch.bool))          # 3 bytes: not made of 32-bit words — left out (and named), the plan does not fail
    g2 = hip_engine._ParamGuard(m, dev)
    assert g2.unguarded == ["flags"] and len(g2.names) == len(g.names)
    del m._buffers["flags"]
    m.final_layer.weight.data.add_(1.0)
    assert hip_engine._version_key(m, dev) == k0                          # the version key is blind to it ...
    g.launch()
    with pytest.raises(hip_engine.StalePlanError, match="final_layer.weight"):
        g.check(wait=True)                                                # ... the guard is not, and names the tensor
    assert "_vatl_plan" not in m.__dict__ and "_vatl_walk" not in m.__dict__        # the stale plan is gone
    g = hip_engine._ParamGuard(m, dev)                                    # what the next call builds
    m.preact.bn1.running_mean.data.add_(0.25)                           # (default-initialised running means ARE zero: .zero_() would change nothing)
    m.preact.layer1[0].bn2.running_var.data.fill_(2.0)
    g.launch()
    with pytest.raises(hip_engine.StalePlanError) as ei:
        g.check()
    assert "preact.bn1.running_mean" in str(ei.value) and "preact.layer1.0.bn2.running_var" in str(ei.value) and "2 parameter" in str(ei.value)
    # a single changed bit anywhere, and two swapped elements, change the sum
    t = torch.arange(1000, dtype=torch.float32)
    base = oc.checksum_tensor(t)
    u = t.clone(); u[[10, 20]] = u[[20, 10]]
    v = t.clone(); v.view(torch.int32)[999] ^= 1
    assert len({base, oc.checksum_tensor(u), oc.checksum_tensor(v)}) == 3
    # invalidate(): the public way out after a `.data` write — drops plan and walk of the model and of every sub-module
    m.__dict__["_vatl_plan"] = (k0, object(), None)
    m.preact.__dict__["_vatl_plan"] = (k0, object(), None)
    hip_engine._version_key(m, dev)
    hip_engine.invalidate(m)
    assert "_vatl_plan" not in m.__dict__ and "_vatl_walk" not in m.__dict__ and "_vatl_plan" not in m.preact.__dict__
    # the cached walk holds no strong reference to its root: a model is freed by reference counting alone once the caller lets go
    hip_engine._version_key(m, dev)
    import gc
    gc.collect(); gc.disable()
    try:
        r = weakref.ref(m)
        del m, g, ei
        assert r() is None, "the model is kept alive by a reference cycle through its cached walk"
    finally:
        gc.enable()
