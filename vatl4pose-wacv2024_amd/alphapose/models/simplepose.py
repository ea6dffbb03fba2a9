"""SimpleBaseline pose network behind the reference's builder surface.

Reference: alphapose/models/simplepose.py:13-91.  Same constructor kwargs
(``PRESET``, ``NUM_LAYERS``, ``NUM_DECONV_FILTERS``), attribute names
(``preact``, ``deconv_layers``, ``final_layer``) and state-dict keys; the forward
pass is a plan of hand-written gfx950 kernels (``hip_engine``).
"""
import torch.nn as nn

from .builder import SPPE
from .layers.Resnet import ResNet


@SPPE.register_module
class SimplePose(nn.Module):
    def __init__(self, norm_layer=nn.BatchNorm2d, **cfg):
        super().__init__()
        self._preset_cfg = cfg["PRESET"]
        self.deconv_dim = cfg["NUM_DECONV_FILTERS"]
        self._norm_layer = norm_layer
        assert cfg["NUM_LAYERS"] in [18, 34, 50, 101, 152]
        self.preact = ResNet(f"resnet{cfg['NUM_LAYERS']}")
        self._imagenet_init(cfg["NUM_LAYERS"])
        self.deconv_layers = self._make_deconv_layer()
        self.final_layer = nn.Conv2d(self.deconv_dim[2], self._preset_cfg["NUM_JOINTS"], kernel_size=1, stride=1, padding=0)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))

    def _imagenet_init(self, depth):
        # The reference pulls torchvision's IMAGENET1K_V2 weights here on every
        # construction (simplepose.py:23-31) and then overwrites them with the
        # pose checkpoint.  Keep that only when torchvision AND its weight cache
        # are available; never touch the network (SURVEY.md §7 hard part 7).
        try:
            import torchvision.models as tm
            tv = getattr(tm, f"resnet{depth}")(weights="IMAGENET1K_V2")
        except Exception:
            return
        own = self.preact.state_dict()
        own.update({k: v for k, v in tv.state_dict().items() if k in own and v.size() == own[k].size()})
        self.preact.load_state_dict(own)

    def _make_deconv_layer(self):
        mods, cin = [], 2048
        for cout in self.deconv_dim:
            mods += [nn.ConvTranspose2d(cin, cout, kernel_size=4, stride=2, padding=1, bias=False),
                     self._norm_layer(cout), nn.ReLU(inplace=True)]
            cin = cout
        return nn.Sequential(*mods)

    def _initialize(self):
        for m in self.deconv_layers.modules():
            if isinstance(m, nn.ConvTranspose2d):
                nn.init.normal_(m.weight, std=0.001)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        nn.init.normal_(self.final_layer.weight, std=0.001)
        nn.init.constant_(self.final_layer.bias, 0)

    def forward(self, x):
        from . import hip_engine
        return hip_engine.run_module_nchw(self, x)

    def get_embedding(self, x):
        from . import hip_engine
        return hip_engine.embedding(self, x)
