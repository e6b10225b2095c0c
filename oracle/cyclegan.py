"""ORACLE (test infrastructure): the Image Motion-Extractor generator (CycleGAN ResNet-9, forward only).

Follows mmseg/models/cyclegan/cyclegan_model.py define_G :119-160, ResnetGenerator :316-374, ResnetBlock :377-434 with
the arguments DACS uses (dacs.py:96-103: define_G() -> 1->1 channels, ngf 64, InstanceNorm, reflect padding, 9 blocks).
Parameter names (`model.<idx>...`) match the reference's nn.Sequential indices.  Pinned by tests/golden.
"""
import torch.nn as nn


class ResnetBlock(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.conv_block = nn.Sequential(
            nn.ReflectionPad2d(1), nn.Conv2d(dim, dim, 3, bias=True), nn.InstanceNorm2d(dim), nn.ReLU(True),
            nn.ReflectionPad2d(1), nn.Conv2d(dim, dim, 3, bias=True), nn.InstanceNorm2d(dim))

    def forward(self, x):
        return x + self.conv_block(x)


class ResnetGenerator(nn.Module):
    def __init__(self, input_nc=1, output_nc=1, ngf=64, n_blocks=9):
        super().__init__()
        m = [nn.ReflectionPad2d(3), nn.Conv2d(input_nc, ngf, 7, bias=True), nn.InstanceNorm2d(ngf), nn.ReLU(True)]
        for i in range(2):
            c = ngf * 2 ** i
            m += [nn.Conv2d(c, c * 2, 3, stride=2, padding=1, bias=True), nn.InstanceNorm2d(c * 2), nn.ReLU(True)]
        m += [ResnetBlock(ngf * 4) for _ in range(n_blocks)]
        for i in range(2):
            c = ngf * 2 ** (2 - i)
            m += [nn.ConvTranspose2d(c, c // 2, 3, stride=2, padding=1, output_padding=1, bias=True),
                  nn.InstanceNorm2d(c // 2), nn.ReLU(True)]
        m += [nn.ReflectionPad2d(3), nn.Conv2d(ngf, output_nc, 7), nn.Tanh()]
        self.model = nn.Sequential(*m)

    def forward(self, x):
        return self.model(x)
