"""FastPose behind the reference's builder surface (reference: alphapose/models/fastpose.py:15-73).

SE-ResNet trunk -> PixelShuffle(2) -> DUC(512->1024) -> DUC(256->512) -> conv3x3 128->17.
Same kwargs (``PRESET``, ``NUM_LAYERS``, optional ``CONV_DIM``), attributes (``preact``,
``suffle1``, ``duc1``, ``duc2``, ``conv_out``) and state-dict keys; forward = HIP plan.
"""
import torch.nn as nn

from .builder import SPPE
from .layers.DUC import DUC
from .layers.SE_Resnet import SEResnet


@SPPE.register_module
class FastPose(nn.Module):
    def __init__(self, norm_layer=nn.BatchNorm2d, **cfg):
        super().__init__()
        self._preset_cfg = cfg["PRESET"]
        self.conv_dim = cfg.get("CONV_DIM", 128)
        if "DCN" in cfg:
            raise NotImplementedError("DCN is outside the hot path (no shipped config uses it; SURVEY.md §2.1 row 3)")
        assert cfg["NUM_LAYERS"] in [18, 34, 50, 101, 152]
        self.preact = SEResnet(f"resnet{cfg['NUM_LAYERS']}")
        self._imagenet_init(cfg["NUM_LAYERS"])
        self.suffle1 = nn.PixelShuffle(2)
        self.duc1 = DUC(512, 1024, upscale_factor=2, norm_layer=norm_layer)
        self.duc2 = DUC(256, 1024 if self.conv_dim == 256 else 512, upscale_factor=2, norm_layer=norm_layer)
        self.conv_out = nn.Conv2d(self.conv_dim, self._preset_cfg["NUM_JOINTS"], kernel_size=3, stride=1, padding=1)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))

    def _imagenet_init(self, depth):
        # same policy as SimplePose: only when torchvision and its weight cache exist (never the network)
        try:
            import torchvision.models as tm
            tv = getattr(tm, f"resnet{depth}")(weights="IMAGENET1K_V2")
        except Exception:
            return
        own = self.preact.state_dict()
        own.update({k: v for k, v in tv.state_dict().items() if k in own and v.size() == own[k].size()})
        self.preact.load_state_dict(own)

    def _initialize(self):
        nn.init.normal_(self.conv_out.weight, std=0.001)
        nn.init.constant_(self.conv_out.bias, 0)

    def forward(self, x):
        from . import hip_engine
        return hip_engine.run_module_nchw(self, x)

    def get_embedding(self, x):
        from . import hip_engine
        return hip_engine.embedding(self, x)
