"""Siamese attention U-Net change detector: the downstream consumer of the generator's synthetic pairs.

Drop-in for `SiameseUNet(n_channels, n_classes)` (reference models.py:47-145; `double_conv` :7-15, `AttentionGate`
:18-44), imported by the reference's train.py:16 and evaluate.py:12 from the same `models` module as the GAN classes.
It is outside the accelerated path: plain PyTorch modules, registered under the
reference's attribute names and in its construction order, so checkpoints load strictly in either direction and
`torch.manual_seed(s)` yields the reference's initial weights bit for bit.
"""
import torch
import torch.nn as nn


def double_conv(in_channels, out_channels):
    """[3x3 conv (no bias) -> BatchNorm -> ReLU] twice; Sequential indices 0..5 as in the reference."""
    layers, c = [], in_channels
    for _ in range(2):
        layers += [nn.Conv2d(c, out_channels, kernel_size=3, padding=1, bias=False), nn.BatchNorm2d(out_channels),
                   nn.ReLU(inplace=True)]
        c = out_channels
    return nn.Sequential(*layers)


def _project(cin, cout):
    return nn.Sequential(nn.Conv2d(cin, cout, kernel_size=1, stride=1, padding=0, bias=True), nn.BatchNorm2d(cout))


class AttentionGate(nn.Module):
    """x * sigmoid(BN(conv1x1(relu(W_g g + W_x x)))): additive attention of a skip tensor x by the gating signal g."""

    def __init__(self, F_g, F_l, F_int):
        super().__init__()
        self.W_g = _project(F_g, F_int)
        self.W_x = _project(F_l, F_int)
        self.psi = nn.Sequential(*_project(F_int, 1), nn.Sigmoid())
        self.relu = nn.ReLU(inplace=True)

    def forward(self, g, x):
        return x * self.psi(self.relu(self.W_g(g) + self.W_x(x)))


class SiameseUNet(nn.Module):
    """Shared-weight encoder applied to both dates; the two feature pyramids are concatenated level by level,
    attention-gated by the upsampled decoder state and decoded to `n_classes` logits (no sigmoid)."""

    _WIDTHS = (64, 128, 256, 512)
    _GATES = ("att_last", "att1", "att2", "att3")                      # gate of level 0..3
    _DECODERS = ("dconv_last", "dconv_up1", "dconv_up2", "dconv_up3")  # decoder block of level 0..3

    def __init__(self, n_channels, n_classes):
        super().__init__()
        self.n_channels, self.n_classes = n_channels, n_classes
        c = n_channels
        for i, wd in enumerate(self._WIDTHS, 1):
            setattr(self, f"dconv_down{i}", double_conv(c, wd))
            c = wd
        self.maxpool = nn.MaxPool2d(2)
        self.bottleneck = double_conv(c, 2 * c)
        self.upsample = nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True)
        # decoder state entering level k (both branches concatenated at the bottom, then the decoder widths)
        state = {3: 4 * c, 2: 512, 1: 256, 0: 128}
        for k in (3, 2, 1, 0):                                          # gates first, deepest first (reference order)
            skip = 2 * self._WIDTHS[k]
            setattr(self, self._GATES[k], AttentionGate(F_g=state[k], F_l=skip, F_int=skip // 2))
        for k in (3, 2, 1, 0):
            setattr(self, self._DECODERS[k], double_conv(state[k] + 2 * self._WIDTHS[k], self._WIDTHS[k]))
        self.conv_last = nn.Conv2d(self._WIDTHS[0], n_classes, 1)

    def forward_encoder(self, x):
        feats = []
        for i in range(1, 5):
            x = getattr(self, f"dconv_down{i}")(x)
            feats.append(x)
            x = self.maxpool(x)
        return (*feats, self.bottleneck(x))

    def forward(self, x1, x2):
        f1, f2 = self.forward_encoder(x1), self.forward_encoder(x2)
        y = torch.cat([f1[4], f2[4]], dim=1)
        for k in (3, 2, 1, 0):
            g = self.upsample(y)
            skip = torch.cat([f1[k], f2[k]], dim=1)
            y = getattr(self, self._DECODERS[k])(torch.cat([g, getattr(self, self._GATES[k])(g, skip)], dim=1))
        return self.conv_last(y)
