"""Dense upsampling convolution container: conv3x3 - BN - ReLU - PixelShuffle (reference: layers/DUC.py:9-29)."""
import torch.nn as nn


class DUC(nn.Module):
    def __init__(self, inplanes, planes, upscale_factor=2, norm_layer=nn.BatchNorm2d):
        super().__init__()
        if upscale_factor != 2:
            raise NotImplementedError("the HIP plan implements PixelShuffle(2) only (every FastPose config uses 2)")
        self.conv = nn.Conv2d(inplanes, planes, kernel_size=3, padding=1, bias=False)
        self.bn = norm_layer(planes, momentum=0.1)
        self.relu = nn.ReLU(inplace=True)
        self.pixel_shuffle = nn.PixelShuffle(upscale_factor)
