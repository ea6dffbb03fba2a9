#!/usr/bin/env python3
"""End-to-end evaluation pass of the active-learning loop on decoded frames (host + device): u8 frames -> device crops ->
SimpleBaseline-R50 -> decode / THC / WPU / local-peak -> query, through ActiveLearning.eval_and_query.

    python tools/al_eval_bench.py [--items 1024] [--batch 256] [--rounds 3]
Prints items/s of eval_and_query (wall clock, everything included) next to bench.py's device-resident figure.
"""
import argparse
import json
import os
import sys
import tempfile
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vatl4pose-wacv2024_amd")):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--items", type=int, default=1024)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--uncertainty", default="THC+WPU")
    ap.add_argument("--retrain", action="store_true", help="also time one fine-tune epoch over all items")
    ap.add_argument("--representativeness", default="None")
    ap.add_argument("--filter", default="None")
    ap.add_argument("--loader", action="store_true", help="evaluation batches through torch's DataLoader + the data set's collate function instead of FrameVideo.collated")
    ap.add_argument("--trace", action="store_true", help="print the wall-clock check-points of every eval_and_query round (ms since its start)")
    ap.add_argument("--b2b", action="store_true", help="bench.py's loop: rounds back to back with NO synchronize between them (the host runs ahead), one flush at the end")
    ap.add_argument("--gc-freeze", action="store_true", help="opt.gc_freeze = True (the constructor freezes what it built; round 5's default)")
    ap.add_argument("--gc-log", action="store_true", help="print every pass of the cyclic collector (generation, ms, collected) with the round it fell into")
    ap.add_argument("--cprofile", action="store_true", help="cProfile the rounds after the first and print the 30 most expensive functions (own time)")
    a = ap.parse_args()
    from active_learning import ActiveLearning
    from alphapose.datasets import FrameVideo
    from alphapose.utils.config import edict
    import bench
    tracks = 16
    frames, anns = bench._synthetic_frame_video(a.items // tracks, tracks, hw=(480, 640))      # (the bench line's own generator: tools do not use oracle/)
    preset = {"IMAGE_SIZE": [256, 192], "HEATMAP_SIZE": [64, 48], "SIGMA": 2}
    ev = FrameVideo(frames, anns, train=False, get_prenext=True, PRESET=preset)
    tr = FrameVideo(frames, anns, train=True, get_prenext=False, PRESET=preset, AUG={"SCALE_FACTOR": 0.25, "ROT_FACTOR": 30, "NUM_JOINTS_HALF_BODY": 8, "PROB_HALF_BODY": 0.3})
    cfg = edict({
        "DATASET": {"TRAIN": {"TYPE": "FrameVideo"}, "EVAL": {"TYPE": "FrameVideo"}},
        "DATA_PRESET": {"TYPE": "simple", "SIGMA": 2, "NUM_JOINTS": 17, "IMAGE_SIZE": [256, 192], "HEATMAP_SIZE": [64, 48]},
        "MODEL": {"TYPE": "SimplePose", "PRETRAINED": "", "TRY_LOAD": "", "NUM_DECONV_FILTERS": [256, 256, 256], "NUM_LAYERS": 50},
        "LOSS": {"TYPE": "MSELoss"}, "AE": {"Z_DIM": 4, "INPUT_DIM": 42, "PRETRAINED": "", "EPOCH": 1, "LR": 1e-3},
        "RETRAIN": {"BATCH_SIZE": 120, "BASE": 1, "OPTIMIZER": "AdamW", "LR": 2.5e-4, "ALPHA": 2, "WEIGHT_DECAY": 0.7, "LR_GAMMA": 0.99},
        "VAL": {"BATCH_SIZE": a.batch, "W_UNC": 0.01, "UNC_LAMBDA": 0.01, "QUERY_RATIO": [0.05, 0.1, 1.0]}})
    with tempfile.TemporaryDirectory() as wd:
        opt = types.SimpleNamespace(work_dir=wd, uncertainty=a.uncertainty, representativeness=a.representativeness, filter=a.filter, strategy=a.uncertainty, video_id="syn",
                                    get_prenext=True, from_scratch=True, continual=True, num_gpu=1, onebyone=False, retrain_thresh=1, THCvsWPU="const",
                                    device_batches=not a.loader, gc_freeze=a.gc_freeze)
        torch.manual_seed(0); np.random.seed(0)
        al = ActiveLearning(cfg, opt, eval_dataset=ev, train_dataset=tr)
        times = []
        gc_log, cur = [], [None]
        if a.gc_log:
            import gc

            def on_gc(phase, info):
                if phase == "start":
                    cur[0] = time.perf_counter()
                else:
                    gc_log.append((len(times), info["generation"], 1e3 * (time.perf_counter() - cur[0]), info["collected"]))
            gc.callbacks.append(on_gc)
        prof = None
        for r in range(a.rounds):
            al.unlabeled_id = list(range(len(ev))); al.labeled_id = []
            if a.cprofile and r == 1:
                import cProfile
                prof = cProfile.Profile(); prof.enable()
            if not a.b2b or r == 0:
                torch.cuda.synchronize()
            t0 = time.perf_counter()
            if a.trace:
                al._trace = []
            al.eval_and_query()
            if r == a.rounds - 1:
                al.flush_records()                               # the last round's record files inside its time; earlier rounds' were written during the next round's device waits
            if not a.b2b or r == a.rounds - 1:
                torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
            if a.trace:
                print(f"round {r}: " + " | ".join(f"{lb} {1e3 * (t - t0):.1f}" for lb, t in list(al._trace)) + f" | end {1e3 * times[-1]:.1f}", flush=True)
        if a.gc_log:
            gc.callbacks.remove(on_gc)
            print("collector passes (round, generation, ms, collected): " + ", ".join(f"({r}, {g}, {ms:.1f}, {c})" for r, g, ms, c in gc_log if g > 0 or ms > 1.0)
                  + f"; {sum(1 for x in gc_log if x[1] == 0)} gen-0 passes, {sum(x[2] for x in gc_log if x[1] == 0):.1f} ms in total", flush=True)
        if prof is not None:
            import pstats
            prof.disable()
            pstats.Stats(prof).sort_stats("tottime").print_stats(30)
        if a.retrain:
            al.retrain_id = list(range(len(ev))); al.labeled_id = list(range(len(ev))); al.retrain_epoch = 1
            rt = []
            for _ in range(a.rounds):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                al.retrain_model()
                torch.cuda.synchronize(); rt.append(time.perf_counter() - t0)
            steps = -(-len(ev) // cfg.RETRAIN.BATCH_SIZE)
            print(json.dumps({"metric": "ActiveLearning.retrain_model, one epoch over all items (wall clock; incl. the WPU auto-encoder refit)", "items": len(ev),
                              "batch": cfg.RETRAIN.BATCH_SIZE, "steps": steps, "seconds": [round(t, 3) for t in rt], "ms_per_step": round(min(rt) / steps * 1e3, 1)}))
        print(json.dumps({"metric": "ActiveLearning.eval_and_query on decoded frames (wall clock)", "items": len(ev), "batch": a.batch,
                          "seconds": [round(t, 3) for t in times], "items_per_s": round(len(ev) / min(times), 1),
                          "sustained_items_per_s": round(len(ev) * (len(times) - 1) / sum(times[1:]), 1) if len(times) > 1 else None}))


if __name__ == "__main__":
    main()
