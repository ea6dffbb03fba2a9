"""Crop producer on the GPU against the oracle's restatement of cv2.warpAffine + im_to_torch.

Integer work: the u8 warp and therefore the fp32 tensor (an exact function of the u8 value) must be BIT-EXACT.
The oracle's warp is a restatement of OpenCV 4.8's published algorithm (cv2 is absent here: parity unpinned, see
oracle/crop.py); its geometry pieces are pinned by tests/golden/crop.npz.
"""
import random

import numpy as np
import pytest
import torch

from gpu_util import dev, record

pytestmark = pytest.mark.gpu


class _DS:
    joint_pairs = [[1, 2], [3, 4], [5, 6], [7, 8], [9, 10], [11, 12], [13, 14], [15, 16]]
    num_joints_half_body = 8
    prob_half_body = 0.3
    upper_body_ids = (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10)
    lower_body_ids = (11, 12, 13, 14, 15, 16)


def _st(train=False, size=(256, 192), hm=(64, 48), sf=0.25, rot=30):
    from alphapose.utils.presets import SimpleTransform
    return SimpleTransform(_DS(), scale_factor=sf, add_dpg=False, input_size=list(size), output_size=list(hm), rot=rot, sigma=2, train=train)


@pytest.mark.parametrize("size", [(256, 192), (384, 288)])
def test_batched_warp_is_bit_exact(size):
    """24 boxes (wide, tall, tiny, partly outside the frame), half of them rotated, some mirrored, over three frames of
    different sizes packed in one arena."""
    from alphapose.utils.bbox import box_to_center_scale_batch
    from alphapose.utils.presets.simple_transform import FrameArena
    from oracle import crop, synth
    frames = [synth.u8_frame(480, 640, 58), synth.u8_frame(360, 500, 59), synth.u8_frame(97, 131, 60)]
    box, rot = synth.crop_cases(24)
    fi = np.arange(24) % 3
    mirror = (np.arange(24) % 5 == 0)
    st = _st(size=size, hm=(size[0] // 4, size[1] // 4))
    c, s = box_to_center_scale_batch(box, st._aspect_ratio)
    got, trans = st.crop_batch(FrameArena(frames), fi, c, s, rot, mirror=mirror)
    got = got.cpu().numpy()
    bad = 0
    for i in range(24):
        src = frames[fi[i]][:, ::-1] if mirror[i] else frames[fi[i]]
        t = crop.affine_transform_matrix(c[i], s[i], rot[i], [size[1], size[0]])
        assert np.array_equal(t, trans[i])
        ref = crop.image_to_tensor(crop.warp_affine_u8(src, t, (size[1], size[0])))
        bad += int((got[i] != ref).sum())
    record(f"crop_warp_{size[0]}x{size[1]}", mismatching_values=bad, crops=24)
    assert bad == 0


def test_dark_crop_is_not_divided_and_every_u8_value_divides_exactly():
    """im_to_torch divides by 255 only when the crop's maximum exceeds 1; the division itself must be IEEE for all 256 values."""
    from alphapose.utils.presets.simple_transform import FrameArena
    from oracle import crop
    ramp = np.arange(256, dtype=np.uint8).reshape(16, 16)
    bright = np.stack([ramp, ramp.T, ramp[::-1]], 2)
    bright = np.kron(bright, np.ones((4, 4, 1), np.uint8))
    dark = (bright > 128).astype(np.uint8)
    black = np.zeros_like(bright)
    st = _st(size=(64, 64), hm=(16, 16))
    ident_c, ident_s = np.float32([[32, 32]] * 3), np.float32([[64, 64]] * 3)       # scale 64 -> 64 px: identity up to the -0.5 px centre
    got, trans = st.crop_batch(FrameArena([bright, dark, black]), [0, 1, 2], ident_c, ident_s, 0.0)
    got = got.cpu().numpy()
    for i, f in enumerate((bright, dark, black)):
        u8 = crop.warp_affine_u8(f, trans[i], (64, 64))
        assert np.array_equal(got[i], crop.image_to_tensor(u8)), i
    assert set(np.unique(crop.warp_affine_u8(bright, trans[0], (64, 64)))) == set(range(256))
    mean = np.float32(crop.MEAN).reshape(3, 1, 1)
    assert np.array_equal(np.unique(got[1] + mean), np.float32([0, 1]))             # undivided 0/1
    assert (got[2] == -mean).all()


def test_per_item_interface_matches_oracle():
    from oracle import crop, scorers, synth
    frame = synth.u8_frame(480, 640, 58)
    box, _ = synth.crop_cases(6)
    st = _st()
    for b in box.tolist():
        img, bb = st.test_transform(frame, b)
        ref_img, ref_bb = crop.test_transform(frame, b, (256, 192))
        assert img.is_cuda and img.shape == (3, 256, 192) and bb.dtype == torch.float32
        assert np.array_equal(img.cpu().numpy(), ref_img)
        assert np.array_equal(bb.numpy(), np.float32(ref_bb))
    # __call__ in eval mode: crop + Gaussian targets of the transformed joints + the corrected box
    r = np.random.RandomState(3)
    for b in box[:3].tolist():
        j3 = np.zeros((17, 3, 2), np.float32)
        j3[:, 0, 0] = r.uniform(b[0], b[2], 17); j3[:, 1, 0] = r.uniform(b[1], b[3], 17)
        j3[:, :2, 1] = (r.random_sample(17) > 0.2).astype(np.float32)[:, None]
        label = {"bbox": tuple(b), "width": 640, "height": 480, "joints_3d": j3.copy()}
        img, target, weight, bb = st(frame, label)
        c, s = crop.box_to_center_scale(b[0], b[1], b[2] - b[0], b[3] - b[1], 0.75)
        t = crop.affine_transform_matrix(c, s * 1.0, 0, [192, 256])
        jt = j3.copy()
        for i in range(17):
            if jt[i, 0, 1] > 0:
                jt[i, 0:2, 0] = crop.transform_point(jt[i, 0:2, 0], t)
        rt, rw = scorers.target_generator(jt[:, 0:2, 0], jt[:, 0, 1])
        assert np.array_equal(img.cpu().numpy(), crop.test_transform(frame, b)[0])
        np.testing.assert_allclose(target.cpu().numpy(), rt, rtol=1e-6, atol=1e-7)
        assert np.array_equal(weight.cpu().numpy().reshape(-1), rw.reshape(-1))
        assert np.array_equal(bb.numpy(), np.float32(crop.center_scale_to_box(c, s)))


def test_train_mode_augmentations_follow_the_reference_draw_order():
    """Seeded np.random / random: replay the reference's draw sequence (half-body, scale, rotation, flip) in the test and
    compare the crop with the oracle warp of the (mirrored) frame under that matrix."""
    from alphapose.utils.transforms import flip_joints_3d
    from oracle import crop, synth
    frame = synth.u8_frame(360, 500, 59)
    st = _st(train=True)
    seen = set()
    for seed in range(12):
        b = [120.0 + seed, 40.0, 330.0, 340.0 - seed]
        j3 = np.zeros((17, 3, 2), np.float32)
        rr = np.random.RandomState(seed)
        j3[:, 0, 0] = rr.uniform(b[0], b[2], 17); j3[:, 1, 0] = rr.uniform(b[1], b[3], 17)
        j3[:, :2, 1] = 1.0
        label = {"bbox": tuple(b), "width": 500, "height": 360, "joints_3d": j3.copy()}
        np.random.seed(seed); random.seed(seed)
        img, target, weight, bb = st(frame, label)
        # --- replay ---
        np.random.seed(seed); random.seed(seed)
        c, s = crop.box_to_center_scale(b[0], b[1], b[2] - b[0], b[3] - b[1], 0.75)
        half = np.random.rand() < _DS.prob_half_body
        if half:
            st.num_joints = 17
            c, s = st.half_body_transform(j3[:, :, 0], j3[:, :1, 1])
        s = (s * np.float32(np.clip(np.random.randn() * 0.25 + 1, 0.75, 1.25))).astype(np.float32)
        r = np.clip(np.random.randn() * 30, -60, 60) if random.random() <= 0.6 else 0
        flipped = random.random() > 0.5
        src = frame[:, ::-1] if flipped else frame
        if flipped:
            c[0] = 500 - c[0] - 1
        seen.add((bool(half), bool(flipped), r != 0))
        t = crop.affine_transform_matrix(c, s, r, [192, 256])
        assert np.array_equal(img.cpu().numpy(), crop.image_to_tensor(crop.warp_affine_u8(src, t, (192, 256)))), seed
        assert np.array_equal(bb.numpy(), np.float32(crop.center_scale_to_box(c, s)))
        jf = flip_joints_3d(j3, 500, _DS.joint_pairs) if flipped else j3
        assert target.shape == (17, 64, 48) and float(weight.sum()) <= 17
    assert len(seen) >= 4                                   # the seeds cover flips, rotations and half-body crops


def test_full_batch_properties_and_errors():
    """1024 crops of 16 frames in one launch: equal to the same crops made in chunks (no cross-crop state), pure integer
    shifts are exact copies, and malformed input is refused."""
    import vatl_hip as vh
    from alphapose.utils.bbox import box_to_center_scale_batch
    from alphapose.utils.presets.simple_transform import FrameArena
    from oracle import synth
    frames = [synth.u8_frame(480, 640, 100 + i) for i in range(16)]
    arena = FrameArena(frames)
    r = np.random.RandomState(5)
    n = 1024
    x0, y0 = r.uniform(0, 400, n), r.uniform(0, 250, n)
    box = np.stack([x0, y0, x0 + r.uniform(30, 220, n), y0 + r.uniform(40, 230, n)], 1)
    fi = r.randint(0, 16, n)
    st = _st()
    c, s = box_to_center_scale_batch(box, 0.75)
    full, _ = st.crop_batch(arena, fi, c, s, 0.0)
    for lo in (0, 700):
        part, _ = st.crop_batch(arena, fi[lo:lo + 100], c[lo:lo + 100], s[lo:lo + 100], 0.0)
        assert torch.equal(part, full[lo:lo + 100])
    # window of exactly 192x256 px whose centre puts output pixel (0,0) on source pixel (40,30): an exact copy
    cc, ss = np.float32([[40 + 96, 30 + 128]]), np.float32([[192, 256]])
    cp, _ = st.crop_batch(arena, [3], cc, ss, 0.0)
    ref = torch.from_numpy(frames[3][30:286, 40:232].transpose(2, 0, 1).copy()).float() / 255 - torch.tensor([0.406, 0.457, 0.480]).view(3, 1, 1)
    assert torch.equal(cp[0].cpu(), ref)
    with pytest.raises(ValueError):
        FrameArena([np.zeros((4, 4, 3), np.float32)])
    empty, _ = st.crop_batch(arena, np.zeros(0, np.int64), np.zeros((0, 2), np.float32), np.zeros((0, 2), np.float32), 0.0)
    assert empty.shape == (0, 3, 256, 192)
    with pytest.raises(vh.VatlError):
        vh.crop_warp_affine(arena.data, torch.zeros(1, dtype=torch.int64, device=dev()), torch.zeros((1, 3), dtype=torch.int32, device=dev()),
                            torch.zeros((1, 2, 3), dtype=torch.float64, device=dev()), (256, 8192))


def _video_cfg():
    from alphapose.utils.config import edict
    return edict({
        "DATASET": {"TRAIN": {"TYPE": "FrameVideo"}, "EVAL": {"TYPE": "FrameVideo"}},
        "DATA_PRESET": {"TYPE": "simple", "SIGMA": 2, "NUM_JOINTS": 17, "IMAGE_SIZE": [256, 192], "HEATMAP_SIZE": [64, 48]},
        "MODEL": {"TYPE": "SimplePose", "PRETRAINED": "", "TRY_LOAD": "", "NUM_DECONV_FILTERS": [256, 256, 256], "NUM_LAYERS": 50},
        "LOSS": {"TYPE": "MSELoss"},
        "AE": {"Z_DIM": 4, "INPUT_DIM": 42, "PRETRAINED": "", "EPOCH": 1, "LR": 1e-3},
        "RETRAIN": {"BATCH_SIZE": 8, "BASE": 1, "OPTIMIZER": "AdamW", "LR": 2.5e-4, "ALPHA": 2, "WEIGHT_DECAY": 0.7, "LR_GAMMA": 0.99},
        "VAL": {"BATCH_SIZE": 6, "W_UNC": 0.01, "UNC_LAMBDA": 0.01, "QUERY_RATIO": [0.25, 1.0]},
    })


def test_frame_video_items_follow_the_reference_item_contract():
    """FrameVideo over decoded u8 frames: every field of the 11-tuple against the oracle composition, prev/next crops
    = test_transform of the id-adjacent item of the same track, and a DataLoader batch = one launch = the same items."""
    from torch.utils.data import DataLoader
    from alphapose.datasets import FrameVideo
    from oracle import crop, scorers, synth
    frames, anns = synth.frame_video(6, 2)
    preset = {"IMAGE_SIZE": [256, 192], "HEATMAP_SIZE": [64, 48], "SIGMA": 2}
    ds = FrameVideo(frames, list(reversed(anns)), train=False, get_prenext=True, PRESET=preset)      # sorts by id itself
    assert len(ds) == 12 and ds.ID_SORTED_STREAM
    by_id = sorted(anns, key=lambda a: a["id"])
    for i in (0, 3, 5, 6, 11):
        a = by_id[i]
        idx, inp, label, mask, gt, img_id, ann_id, bb_crop, bb_ann, is_prev, is_next = ds[i]
        assert idx == i and img_id == a["img_id"] and ann_id == a["ann_id"]
        assert (is_prev, is_next) == (i not in (0, 6), i not in (5, 11))                          # two tracks of six frames
        ref_img, ref_bb = crop.test_transform(frames[a["frame"]], a["bbox"])
        assert inp.shape == (3, 3, 256, 192) and np.array_equal(inp[0].cpu().numpy(), ref_img)
        for slot, flag, step in ((1, is_prev, -1), (2, is_next, 1)):
            if flag:
                nb = by_id[i + step]
                assert np.array_equal(inp[slot].cpu().numpy(), crop.test_transform(frames[nb["frame"]], nb["bbox"])[0])
            else:
                assert float(inp[slot].abs().sum()) == 0.0
        assert np.array_equal(bb_crop.numpy(), np.float32(ref_bb)) and np.array_equal(bb_ann.numpy(), np.float32(a["bbox"]))
        assert np.array_equal(gt.numpy(), np.float32(a["keypoint"]))
        c, s = crop.box_to_center_scale(a["bbox"][0], a["bbox"][1], a["bbox"][2] - a["bbox"][0], a["bbox"][3] - a["bbox"][1], 0.75)
        t = crop.affine_transform_matrix(c, s, 0, [192, 256])
        jt = a["joints_3d"].copy()
        for j in range(17):
            if jt[j, 0, 1] > 0:
                jt[j, 0:2, 0] = crop.transform_point(jt[j, 0:2, 0], t)
        rt, rw = scorers.target_generator(jt[:, 0:2, 0], jt[:, 0, 1])
        np.testing.assert_allclose(label.cpu().numpy(), rt, rtol=1e-6, atol=1e-7)
        assert np.array_equal(mask.cpu().numpy().reshape(-1), rw.reshape(-1))
    batch = next(iter(DataLoader(ds, batch_size=5, shuffle=False, collate_fn=ds.my_collate_fn)))
    assert batch[1].shape == (5, 3, 3, 256, 192) and batch[1].is_cuda
    for k in range(5):
        assert torch.equal(batch[1][k], ds[k][1]) and torch.equal(batch[2][k], ds[k][2])
    assert batch[9] == [False, True, True, True, True]
    direct = ds.collated(range(5))                                   # what ActiveLearning's scoring loop takes: the same 11 columns, no per-item objects
    assert len(direct) == 11
    for a, b in zip(batch, direct):
        assert (torch.equal(a, b) and a.dtype == b.dtype) if torch.is_tensor(a) else list(a) == list(b)
    ds.emit_neighbour_crops = False                                  # (the id-sorted stream mode: current crops only)
    lean = ds.collated([3, 1, 4])
    assert lean[1].shape == (3, 1, 3, 256, 192) and torch.equal(lean[1][:, 0], torch.stack([ds[3][1][0], ds[1][1][0], ds[4][1][0]]))
    assert lean[0] == [3, 1, 4] and torch.equal(lean[4][1], ds[1][4]) and lean[5] == [ds[3][5], ds[1][5], ds[4][5]]
    ds.emit_neighbour_crops = True


def test_active_learning_round_on_decoded_frames(tmp_path):
    """u8 frames -> device crops -> backbone -> scorers -> query -> fine-tune on augmented device crops -> next evaluation:
    the whole loop with no host-side pixel work."""
    import types
    from active_learning import ActiveLearning
    from alphapose.datasets import FrameVideo
    from oracle import synth
    frames, anns = synth.frame_video(8, 2)
    preset = {"IMAGE_SIZE": [256, 192], "HEATMAP_SIZE": [64, 48], "SIGMA": 2}
    aug = {"SCALE_FACTOR": 0.25, "ROT_FACTOR": 30, "NUM_JOINTS_HALF_BODY": 8, "PROB_HALF_BODY": 0.3}
    ev = FrameVideo(frames, anns, train=False, get_prenext=True, PRESET=preset)
    tr = FrameVideo(frames, anns, train=True, get_prenext=False, PRESET=preset, AUG=aug)
    assert not tr.ID_SORTED_STREAM
    opt = types.SimpleNamespace(work_dir=str(tmp_path), uncertainty="THC+WPU", representativeness="None", filter="None", strategy="THC+WPU", video_id="syn",
                                get_prenext=True, from_scratch=True, continual=True, num_gpu=1, onebyone=False, retrain_thresh=1, THCvsWPU="const")
    torch.manual_seed(0); np.random.seed(0); random.seed(0)
    al = ActiveLearning(_video_cfg(), opt, eval_dataset=ev, train_dataset=tr)
    assert al.dedup
    al.eval_and_query()
    assert len(al.labeled_id) == 4 and len(al.unlabeled_id) == 12
    kp0 = np.array(al.keypoints[0])
    # the scored key-points are the decode of the model's heat-map for the device-made crop
    al.model.eval()
    with torch.no_grad():
        hm = al.model(ev[0][1][0][None]).cpu().numpy()
    from oracle import scorers
    d = scorers.decode_heatmaps(hm[0], ev[0][7].numpy())
    np.testing.assert_allclose(kp0.reshape(17, 3)[:, :2], d["coords"], rtol=1e-4, atol=1e-4)
    assert al.outcome() is None and np.isfinite(al.last_train_loss)
    al.eval_and_query()
    assert len(al.unlabeled_id) == 0


def test_posetrack_json_dataset_items_and_evaluation(tmp_path):
    """Posetrack21 from annotation json + image files: items equal the oracle composition on the decoded frames (PNG: lossless),
    prev / next follow the tracks, and an active-learning evaluation runs on it through builder.build_dataset."""
    import types
    from active_learning import ActiveLearning
    from alphapose.models import builder
    from alphapose.utils.config import edict
    from oracle import crop, synth
    ann, frames, kept = synth.write_coco_video(str(tmp_path), n_frames=4, tracks=2)
    cfg = _video_cfg()
    for split in ("TRAIN", "EVAL"):
        cfg.DATASET[split] = edict({"TYPE": "Posetrack21", "ROOT": str(tmp_path), "IMG_PREFIX": "", "ANN": ann,
                                    "AUG": {"SCALE_FACTOR": 0.25, "ROT_FACTOR": 30, "NUM_JOINTS_HALF_BODY": 8, "PROB_HALF_BODY": 0.3}})
    ds = builder.build_dataset(cfg.DATASET.EVAL, preset_cfg=cfg.DATA_PRESET, train=False, get_prenext=True)
    assert len(ds) == 8 and ds.ID_SORTED_STREAM
    for i in (0, 3, 4, 7):
        a = ds._labels[i]
        f = int(a["frame"][-10:-4])
        idx, inp, label, mask, gt, img_id, ann_id, bb_crop, bb_ann, is_prev, is_next = ds[i]
        assert (is_prev, is_next) == (i % 4 != 0, i % 4 != 3) and ann_id in kept and img_id == 1000200 + f
        assert np.array_equal(inp[0].cpu().numpy(), crop.test_transform(frames[f], a["bbox"])[0])
        if is_next:
            nb = ds._labels[i + 1]
            assert np.array_equal(inp[2].cpu().numpy(), crop.test_transform(frames[f + 1], nb["bbox"])[0])
    opt = types.SimpleNamespace(work_dir=str(tmp_path / "work"), uncertainty="THC+WPU", representativeness="None", filter="None", strategy="THC+WPU",
                                video_id="vid0", get_prenext=True, from_scratch=True, continual=True, num_gpu=1, onebyone=False, retrain_thresh=1,
                                THCvsWPU="const")
    os_mod = __import__("os"); os_mod.makedirs(opt.work_dir, exist_ok=True)
    torch.manual_seed(0); np.random.seed(0); random.seed(0)
    al = ActiveLearning(cfg, opt)
    assert type(al.eval_dataset).__name__ == "Posetrack21" and al.dedup
    al.eval_and_query()
    assert len(al.labeled_id) == 2 and len(al.unlabeled_id) == 6
    import json
    al.flush_records()                                                                          # the record files are written by a host thread (complete at the next entry point / outcome / flush)
    gt = json.load(open(os_mod.path.join(opt.work_dir, "GT_kpt.json")))                        # images / categories come from the annotation file
    assert len(gt["images"]) == 4 and gt["images"][0]["file_name"].endswith("000000.png") and len(gt["annotations"]) == 8
    assert al.outcome() is None and np.isfinite(al.last_train_loss)


def test_image_dataset_batch_on_device(tmp_path):
    """Mpii-style image dataset (16 joints): a DataLoader batch of CustomDataset's 5-tuples made by one warp + one target launch,
    equal to the oracle composition on the decoded images."""
    import json, os
    from PIL import Image
    from torch.utils.data import DataLoader
    from alphapose.models import builder
    from alphapose.utils.config import edict
    from oracle import crop, scorers, synth
    root = str(tmp_path)
    os.makedirs(os.path.join(root, "images")); os.makedirs(os.path.join(root, "annot"))
    images, anns, frames = [], [], []
    r = np.random.RandomState(4)
    for i in range(5):
        name = f"{i + 1:09d}.png"
        f = synth.u8_frame(96, 128, 40 + i); frames.append(f)
        Image.fromarray(f).save(os.path.join(root, "images", name))
        images.append({"id": i + 1, "file_name": name, "width": 128, "height": 96})
        kp = []
        for j in range(16):
            kp += [float(r.uniform(20, 80)), float(r.uniform(10, 80)), int(r.random_sample() > 0.2)]
        kp[2] = 1
        anns.append({"id": 10 + i, "image_id": i + 1, "category_id": 1, "bbox": [15.0 + i, 8.0, 70.0, 80.0], "num_keypoints": 16, "keypoints": kp})
    json.dump({"images": images, "annotations": anns, "categories": [{"id": 1, "name": "person"}]}, open(os.path.join(root, "annot", "v.json"), "w"))
    preset = edict({"TYPE": "simple", "SIGMA": 2, "NUM_JOINTS": 16, "IMAGE_SIZE": [256, 192], "HEATMAP_SIZE": [64, 48]})
    ds = builder.build_dataset(edict({"TYPE": "Mpii", "ROOT": root, "IMG_PREFIX": "images", "ANN": "annot/v.json"}), preset_cfg=preset, train=False)
    img, label, mask, ids, bbox = next(iter(DataLoader(ds, batch_size=5, shuffle=False, collate_fn=ds.my_collate_fn)))
    assert img.shape == (5, 3, 256, 192) and label.shape == (5, 16, 64, 48) and mask.shape == (5, 16, 1, 1) and ids == [1, 2, 3, 4, 5]
    for i in range(5):
        a = ds._labels[i]
        ref_img, ref_bb = crop.test_transform(frames[i], a["bbox"])
        assert np.array_equal(img[i].cpu().numpy(), ref_img) and np.array_equal(bbox[i].numpy(), np.float32(ref_bb))
        c, s = crop.box_to_center_scale(a["bbox"][0], a["bbox"][1], a["bbox"][2] - a["bbox"][0], a["bbox"][3] - a["bbox"][1], 0.75)
        t = crop.affine_transform_matrix(c, s, 0, [192, 256])
        jt = a["joints_3d"].copy()
        for j in range(16):
            if jt[j, 0, 1] > 0:
                jt[j, 0:2, 0] = crop.transform_point(jt[j, 0:2, 0], t)
        rt, rw = scorers.target_generator(jt[:, 0:2, 0], jt[:, 0, 1])
        np.testing.assert_allclose(label[i].cpu().numpy(), rt, rtol=1e-6, atol=1e-7)
        assert np.array_equal(mask[i].cpu().numpy().reshape(-1), rw.reshape(-1))
