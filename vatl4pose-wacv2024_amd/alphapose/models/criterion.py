"""Loss registry entries (reference: alphapose/models/criterion.py).

``MSELoss`` stays torch's class so ``build_loss(cfg.LOSS)`` is call-compatible (criterion.py:97); the fine-tune step
itself uses the fused masked-MSE kernel (``vatl_hip.masked_mse_fwd_bwd``).

``L1JointRegression`` (criterion.py:46-94) is the integral-regression loss of the ``LOSS.TYPE: L1JointRegression``
configs: soft-arg-max coordinates with the reference's symmetric +-2 "IngetralCoordinate" gradient and a weighted L1;
forward and backward are one ``vatl_l1_joint_regression_fwd_bwd`` launch behind a ``torch.autograd.Function``.
"""
import torch
import torch.nn as nn

import vatl_hip as vh

from .builder import LOSS


class _L1JointRegressionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, preds, gt_joints, gt_joints_vis, norm_type, size_average):
        loss, grad, _ = vh.l1_joint_regression_fwd_bwd(preds.detach().float().contiguous(), gt_joints.float().contiguous(),
                                                       gt_joints_vis.float().contiguous(), norm_type, size_average)
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None, None, None


@LOSS.register_module
class L1JointRegression(nn.Module):
    """L1 Joint Regression Loss (criterion.py:46-80): ``forward(preds (B,J,H,W), gt_joints (B,2J), gt_joints_vis (B,2J))``."""

    def __init__(self, OUTPUT_3D=False, size_average=True, reduce=True, NORM_TYPE="softmax"):
        super().__init__()
        if OUTPUT_3D:
            raise NotImplementedError("3-D heat-maps are not used by any pose config of the reference")
        if NORM_TYPE not in vh.NORM_TYPES:
            raise NotImplementedError(NORM_TYPE)
        self.size_average, self.reduce, self.output_3d, self.norm_type = size_average, reduce, OUTPUT_3D, NORM_TYPE

    def forward(self, preds, *args):
        gt_joints, gt_joints_vis = args[0], args[1]
        assert not gt_joints.requires_grad and not gt_joints_vis.requires_grad, \
            "nn criterions don't compute the gradient w.r.t. targets - please mark these tensors as not requiring gradients"
        if not preds.is_cuda:
            raise vh.VatlError("L1JointRegression runs on MI355X only (there is deliberately no CPU fallback)")
        return _L1JointRegressionFn.apply(preds, gt_joints, gt_joints_vis, self.norm_type, self.size_average)


LOSS.register_module(torch.nn.MSELoss)
