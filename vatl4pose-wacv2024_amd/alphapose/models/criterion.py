"""Loss registry entries (reference: alphapose/models/criterion.py:97 registers torch.nn.MSELoss).

``MSELoss`` stays torch's class so ``build_loss(cfg.LOSS)`` is call-compatible;
the fine-tune step itself uses the fused masked-MSE kernel
(``vatl_hip.masked_mse_fwd_bwd``), see active_learning/scoring.py.
"""
import torch

from .builder import LOSS

LOSS.register_module(torch.nn.MSELoss)
