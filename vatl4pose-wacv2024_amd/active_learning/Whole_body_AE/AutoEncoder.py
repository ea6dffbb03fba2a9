"""Whole-body pose auto-encoder (reference: Whole_body_AE/AutoEncoder.py:5-39).

``encoder`` / ``decoder`` are ``nn.Sequential`` parameter containers with the reference's
state-dict keys (``encoder.{0,2,4,6}.*``, ``decoder.{0,2,4,6}.*``); ``forward`` runs the
one-wave-per-item HIP kernel.  ``input_dim`` is a constructor argument because the
released code feeds 42 values to a module declared with 38.
"""
from __future__ import annotations

import torch
import torch.nn as nn

import vatl_hip as vh


class WholeBodyAE(nn.Module):
    def __init__(self, z_dim=2, kp_direct=False, input_dim=None):
        super().__init__()
        self.z_dim = z_dim
        self.input_dim = input_dim if input_dim is not None else (51 if kp_direct else 38)
        d = self.input_dim
        self.encoder = nn.Sequential(nn.Linear(d, 24), nn.ReLU(True), nn.Linear(24, 12), nn.ReLU(True),
                                     nn.Linear(12, 7), nn.ReLU(True), nn.Linear(7, z_dim))
        self.decoder = nn.Sequential(nn.Linear(z_dim, 7), nn.ReLU(True), nn.Linear(7, 12), nn.ReLU(True),
                                     nn.Linear(12, 24), nn.ReLU(True), nn.Linear(24, d), nn.Sigmoid())

    def packed(self) -> torch.Tensor:
        """Weights flattened in state-dict order for the C ABI (cached per parameter version)."""
        key = tuple(p._version for p in self.parameters()) + (str(next(self.parameters()).device),)
        c = self.__dict__.get("_vatl_packed")
        if c is None or c[0] != key:
            c = (key, vh.pack_ae(self.state_dict(), next(self.parameters()).device))
            self.__dict__["_vatl_packed"] = c
        return c[1]

    def forward(self, x):
        if not x.is_cuda:
            raise vh.VatlError("WholeBodyAE runs on MI355X only (no CPU fallback)")
        flat = x.detach().float().reshape(-1, self.input_dim).contiguous()
        recon, _ = vh.ae_forward(flat, self.packed(), self.input_dim, self.z_dim)
        return recon.reshape(x.shape)


def fit_autoencoder(ae: WholeBodyAE, features: torch.Tensor, epochs: int, lr: float, batch_size: int = 10, generator=None):
    """retrain_AE (ActiveLearning.py:905-925): ``epochs`` passes of shuffled mini-batches (batch 10 in the reference),
    AE forward + MSELoss(output, input) + backward + torch.optim.Adam(lr) — every mini-batch is ONE
    ``vatl_ae_train_step`` launch on the packed parameters.  Returns the mean loss over all steps."""
    if not features.is_cuda:
        raise vh.VatlError("the auto-encoder trains on MI355X only (no CPU fallback)")
    feats = features.detach().float().reshape(-1, ae.input_dim).contiguous()
    n = feats.shape[0]
    if n == 0 or epochs <= 0:
        return 0.0
    flat = vh.pack_ae(ae.state_dict(), feats.device).clone()
    m, v = torch.zeros_like(flat), torch.zeros_like(flat)
    losses, step = [], 0
    for _ in range(int(epochs)):
        perm = torch.randperm(n, generator=generator, device="cpu").to(feats.device)
        for i in range(0, n, batch_size):
            step += 1
            losses.append(vh.ae_train_step(flat, m, v, feats[perm[i:i + batch_size]].contiguous(), ae.input_dim, ae.z_dim, step, lr))
    vh.unpack_ae(flat, ae)
    for p in ae.parameters():                                   # in-place update through the C ABI: bump the version counters
        torch.autograd.graph.increment_version(p)
    return float(torch.stack(losses).mean())
