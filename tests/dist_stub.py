"""A CPU stand-in with ActiveLearning's bring-up lines, for the worker-process plumbing tests (tests/test_distributed_cpu.py)."""
import json
import os

import torch
import torch.distributed as dist

from active_learning import distributed as D
from active_learning.ActiveLearning import _collective


class StubAL:
    def __init__(self, cfg, opt, **_):
        self._depth = 0
        self.world = D.ensure_workers(opt["num_gpu"])
        if D.have_workers() and D.is_main():
            D.command("new", cfg, opt, "")
        self.cfg, self.opt, self.calls = cfg, opt, []

    def _log(self, what, value):
        _, rank = D.world_rank()
        with open(os.path.join(self.opt["dir"], f"rank{rank}.jsonl"), "a") as f:
            f.write(json.dumps({"call": what, "value": value, "world": self.world}) + "\n")

    @_collective
    def eval_and_query(self):
        t = torch.tensor([float(D.world_rank()[1] + 1)])
        D.allreduce_sum_(t)
        self._log("eval_and_query", float(t))

    @_collective
    def retrain_model(self):
        self._log("retrain_model", D.broadcast_object(self.cfg["token"] if D.is_main() else None))

    @_collective
    def outcome(self):
        self.retrain_model()                               # nested: announced once, by the outermost call
        self._log("outcome", 0)
