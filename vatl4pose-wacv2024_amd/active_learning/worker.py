"""Rank > 0 of the one-process-per-GPU group that replaces the reference's ``nn.DataParallel`` replicas
(ActiveLearning.py:233) when the unchanged driver runs as a single process:

    python -m active_learning.worker            (started by distributed.ensure_workers, never by hand)

The worker joins the rendezvous from its environment, then executes whatever rank 0 broadcasts:
``("new", cfg, opt, payload_path)`` builds an ``ActiveLearning`` replica (datasets from the config, or from the payload
file when the driver handed dataset objects to the constructor), ``("call", name)`` runs one of its public methods
(``eval_and_query`` / ``outcome`` / ``retrain_model``) in lock-step with rank 0, ``("exit",)`` leaves.  A watchdog ends
the process when the driver disappears without saying goodbye.
"""
from __future__ import annotations

import os
import pickle
import sys
import threading
import time


def _watch_parent(pid: int):
    def loop():
        while True:
            time.sleep(2.0)
            try:
                os.kill(pid, 0)
            except OSError:
                os._exit(3)
    threading.Thread(target=loop, daemon=True).start()


def serve(make, recv):
    """The command loop, separated from the process plumbing so that it can be exercised on CPU: ``recv()`` returns the
    next message, ``make(cfg, opt, payload)`` builds the object the calls go to."""
    obj = None
    while True:
        msg = recv()
        if msg[0] == "exit":
            return
        if msg[0] == "new":
            payload = None
            if msg[3]:
                with open(msg[3], "rb") as f:
                    payload = pickle.load(f)
            obj = make(msg[1], msg[2], payload)
        elif msg[0] == "call":
            getattr(obj, msg[1])()
        else:
            raise ValueError(f"unknown worker command {msg[0]!r}")


def main():
    parent = int(os.environ.get("VATL_WORKER_PARENT", "0"))
    if parent:
        _watch_parent(parent)
    from active_learning import distributed as D
    if not D.init_from_env():
        sys.exit("active_learning.worker needs the WORLD_SIZE / RANK / MASTER_* environment of its driver")
    # the class whose calls are mirrored: the package's ActiveLearning unless the driver runs a subclass of its own
    mod, _, cls = os.environ.get("VATL_WORKER_CLASS", "active_learning.ActiveLearning:ActiveLearning").partition(":")
    import importlib
    klass = getattr(importlib.import_module(mod), cls)

    def make(cfg, opt, payload):
        kw = payload or {}
        return klass(cfg, opt, **kw)

    serve(make, lambda: D.broadcast_object(None, src=0))
    import torch.distributed as dist
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
