"""Oracle (test infrastructure): plain-PyTorch restatement of the two networks.

Module attribute names and construction order equal the reference's so that
(i) ``state_dict()`` keys/shapes are identical and (ii) seeding the global RNG
and constructing reproduces the reference's initial weights bit-for-bit
(SURVEY.md §8c "seeded-init trick").

Reference:
  models_twomodalinputs/netblocks.py:9-19   up path (bilinear+conv3x3 | ConvT 2x2)
  models_twomodalinputs/netblocks.py:21-33  basic_block (conv-bn-relu x2)
  models_twomodalinputs/netblocks.py:128-147 down / up blocks
  models_twomodalinputs/fuseunet.py:6-91    fuseunet
  models_singlemodalinput/UNet.py:110-165   UNet (pool inside the down block)
  models_twomodalinputs/netblocks.py:68-89  Spatial_Attention (dup UNet.py:85-107)
  models_twomodalinputs/fuseunet.py:93-221  fuseunetsa ; models_singlemodalinput/UNet.py:168-208 UNetsa
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class _DoubleConv(nn.Module):
    # netblocks.py:21-33 ; registration order conv1, bn1, conv2, bn2 matters for seeding
    def __init__(self, cin, cout):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.bn1 = nn.BatchNorm2d(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.bn2 = nn.BatchNorm2d(cout)

    def forward(self, x):
        x = F.relu(self.bn1(self.conv1(x)))
        return F.relu(self.bn2(self.conv2(x)))


class _Down(nn.Module):
    # netblocks.py:128-135 (pool=False) ; UNet.py:110-121 (pool inside when down_size)
    def __init__(self, cin, cout, pool=False):
        super().__init__()
        self.block = _DoubleConv(cin, cout)
        self.pool = pool

    def forward(self, x):
        if self.pool:
            x = F.max_pool2d(x, 2, 2)
        return self.block(x)


def _up_path(cin, cout, learned):
    # netblocks.py:9-19 — Sequential indices are part of the state_dict keys
    if learned:
        return nn.Sequential(nn.ConvTranspose2d(cin, cout, kernel_size=2, stride=2),
                             nn.BatchNorm2d(cout), nn.ReLU())
    return nn.Sequential(nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True),
                         nn.Conv2d(cin, cout, kernel_size=3, padding=1),
                         nn.BatchNorm2d(cout), nn.ReLU())


class _Up(nn.Module):
    # netblocks.py:137-147 — cat order is (upsampled, skip)
    def __init__(self, cin, cprev, cout, learned=False):
        super().__init__()
        self.bilinear_up = _up_path(cin, cprev, learned)
        self.block = _DoubleConv(cprev * 2, cout)

    def forward(self, skip, x):
        return self.block(torch.cat((self.bilinear_up(x), skip), dim=1))


class Spatial_Attention(nn.Module):
    # netblocks.py:68-89: conv1 1x1 (C -> C/r), conv2 / conv3 3x3 dilated (padding = dilation), conv4 1x1 -> 1,
    # BatchNorm2d(1), sigmoid; the caller multiplies the gate onto its input
    def __init__(self, input_channel, reduction=16, dilation=4):
        super().__init__()
        r = input_channel // reduction
        self.conv1 = nn.Conv2d(input_channel, r, kernel_size=1, stride=1, padding=0)
        self.conv2 = nn.Conv2d(r, r, kernel_size=3, dilation=dilation, stride=1, padding=dilation)
        self.conv3 = nn.Conv2d(r, r, kernel_size=3, dilation=dilation, stride=1, padding=dilation)
        self.conv4 = nn.Conv2d(r, 1, kernel_size=1, stride=1, padding=0)
        self.bn = nn.BatchNorm2d(1)
        self.sigmoid = nn.Sigmoid()

    def forward(self, x):
        return self.sigmoid(self.bn(self.conv4(self.conv3(self.conv2(self.conv1(x))))))


class fuseunet(nn.Module):
    """fuseunet.py:6-91. ``reduction``/``dilation`` accepted and ignored (fuseunet.py:7)."""
    ATTENTION = False
    M1 = [(3, 32), (64, 64), (128, 128), (256, 256), (512, 512)]     # fuseunet.py:12-20
    M2 = [(3, 32), (32, 64), (64, 128), (128, 256), (256, 512)]      # fuseunet.py:24-32
    UP = [(1024, 512, 512), (512, 256, 256), (256, 128, 128), (128, 64, 64)]  # :36-39

    def __init__(self, num_classes=2, reduction=16, dilation=4, learned_bilinear=False):
        super().__init__()
        for m, widths in (('modal1', self.M1), ('modal2', self.M2)):
            for i, (a, b) in enumerate(widths, 1):
                setattr(self, '%s_downblock%d' % (m, i), _Down(a, b))
                if self.ATTENTION:                                    # fuseunet.py:99-127 registration order
                    setattr(self, '%s_sa%d' % (m, i), Spatial_Attention(b, reduction=reduction, dilation=dilation))
        for i, (a, p, o) in enumerate(self.UP, 1):
            setattr(self, 'up_block%d' % i, _Up(a, p, o, learned_bilinear))
        self.last_conv1 = nn.Conv2d(64, num_classes, 1, padding=0)

    SEPARATE = False

    def forward(self, modal1_inputs, modal2_inputs):
        y, x = modal1_inputs, modal2_inputs
        skips = []
        for i in range(1, 6):                                         # fuseunet.py:45-81
            if i > 1:
                # modal-1 consumes the fused tensor; fuseunetsaseparate pools its own stream (fuseunet.py:270-271)
                y = F.max_pool2d(y if self.SEPARATE else skips[-1], 2, 2)
                x = F.max_pool2d(x, 2, 2)
            y = getattr(self, 'modal1_downblock%d' % i)(y)
            x = getattr(self, 'modal2_downblock%d' % i)(x)
            if self.ATTENTION:                                        # fuseunet.py:139-145: y = sa(y) * y
                y = getattr(self, 'modal1_sa%d' % i)(y) * y
                x = getattr(self, 'modal2_sa%d' % i)(x) * x
            skips.append(torch.cat((y, x), dim=1))  # (modal1, modal2)
        y = skips[4]
        for i in range(1, 5):                                         # fuseunet.py:85-88
            y = getattr(self, 'up_block%d' % i)(skips[4 - i], y)
        return self.last_conv1(y)                                     # fuseunet.py:89


class fuseunetsa(fuseunet):
    """fuseunet.py:93-221."""
    ATTENTION = True


class fuseunetsaseparate(fuseunet):
    """fuseunet.py:210-322: two independent attention encoders (3-32-64-128-256-512 each) whose gated outputs are
    concatenated per level for the decoder only."""
    ATTENTION = True
    SEPARATE = True
    M1 = fuseunet.M2                                                  # fuseunet.py:216-229


class UNet(nn.Module):
    """UNet.py:135-165; the width variants UNet128 ... UNet2 (UNet.py:210-400) differ only in BASE."""
    ATTENTION = False
    BASE = 64

    def __init__(self, num_classes=2, learned_bilinear=False):
        super().__init__()
        b0 = self.BASE
        enc = [(3, b0)] + [(b0 << k, b0 << (k + 1)) for k in range(4)]       # UNet.py:139-143
        for i, (a, b) in enumerate(enc, 1):
            setattr(self, 'down_block%d' % i, _Down(a, b, pool=(i > 1)))
            if self.ATTENTION:                                        # UNet.py:172-181
                setattr(self, 'sa%d' % i, Spatial_Attention(b, reduction=16, dilation=4))
        for i in range(1, 5):                                         # UNet.py:145-148
            w = b0 << (4 - i)
            setattr(self, 'up_block%d' % i, _Up(2 * w, w, w, learned_bilinear))
        self.last_conv1 = nn.Conv2d(b0, num_classes, 1, padding=0)

    def forward(self, x):
        feats = []
        for i in range(1, 6):
            x = getattr(self, 'down_block%d' % i)(x)
            if self.ATTENTION:                                        # UNet.py:191-200
                x = getattr(self, 'sa%d' % i)(x) * x
            feats.append(x)
        for i in range(1, 5):
            x = getattr(self, 'up_block%d' % i)(feats[4 - i], x)
        return self.last_conv1(x)


class UNetsa(UNet):
    """UNet.py:168-208."""
    ATTENTION = True


class UNet128(UNet):
    """UNet.py:210-240."""
    BASE = 128


class UNet32(UNet):
    """UNet.py:242-272."""
    BASE = 32


class UNet16(UNet):
    """UNet.py:274-304."""
    BASE = 16


class UNet8(UNet):
    """UNet.py:306-336."""
    BASE = 8


class UNet4(UNet):
    """UNet.py:338-368."""
    BASE = 4


class UNet2(UNet):
    """UNet.py:370-400."""
    BASE = 2
