"""MI355X-native drop-in for the reference's single-modality `UNet`
(models_singlemodalinput/UNet.py:135-165): identical names, signatures and state_dict keys;
forward/backward run as HIP kernels through aide_amd.engine."""
import torch.nn as nn

from ..engine import Engine, Graph
from ..models_twomodalinputs.netblocks import (UNet_basic_down_block, UNet_basic_up_block, Spatial_Attention,
                                                add_decoder)


class UNet(nn.Module):
    _ATTENTION = False
    _BASE = 64      # first-level width; the variants UNet128 ... UNet2 (UNet.py:210-400) differ only here

    def __init__(self, num_classes=2, learned_bilinear=False):
        nn.Module.__init__(self)
        b0 = self._BASE
        self._ENC = ((3, b0),) + tuple((b0 << k, b0 << (k + 1)) for k in range(4))      # UNet.py:139-143
        self._UP = tuple((b0 << (5 - i), b0 << (4 - i), b0 << (4 - i)) for i in range(1, 5))   # UNet.py:145-148
        for i, (a, b) in enumerate(self._ENC, 1):
            blk = UNet_basic_down_block(a, b, i > 1)
            blk.max_pool = nn.MaxPool2d(2, 2)          # parameter-free; kept for module-tree parity
            setattr(self, 'down_block%d' % i, blk)
            if self._ATTENTION:                        # UNetsa: UNet.py:172-181
                setattr(self, 'sa%d' % i, Spatial_Attention(b, reduction=16, dilation=4))
        for i, (a, p, o) in enumerate(self._UP, 1):
            setattr(self, 'up_block%d' % i, UNet_basic_up_block(a, p, o, learned_bilinear))
        self.last_conv1 = nn.Conv2d(b0, num_classes, 1, padding=0)
        self._engine = [Engine(self, self._build_graph, num_classes)]

    @property
    def engine(self):
        return self._engine[0]

    def _build_graph(self):
        """UNet.py:152-165; the pool sits at the start of down blocks 2-5 (UNet.py:117-121)."""
        g = Graph()
        x = g.input('image', 3)
        prev = [p for _, p, _ in self._UP]
        enc_c = [b for _, b in self._ENC]
        cats = [g.tensor('cat_s%d' % s, 2 * prev[4 - s], s - 1) for s in range(1, 5)]
        x5 = g.tensor('x5', enc_c[4], 4)
        src = x
        for s in range(1, 6):
            dst = cats[s - 1].slice(prev[4 - s], enc_c[s - 1], 'x%d' % s) if s <= 4 else x5
            blk = getattr(self, 'down_block%d' % s).block
            t = g.tensor('enc_s%d_mid' % s, enc_c[s - 1], s - 1)
            g.conv_bn_relu(src, t, blk.conv1, blk.bn1)
            if self._ATTENTION:                        # x_s = sa_s(x_s) * x_s  (UNet.py:191-200)
                pre = g.tensor('enc_s%d_pre' % s, enc_c[s - 1], s - 1)
                g.conv_bn_relu(t, pre, blk.conv2, blk.bn2)
                g.spatial_attention(pre, dst, getattr(self, 'sa%d' % s))
            else:
                g.conv_bn_relu(t, dst, blk.conv2, blk.bn2)
            if s < 5:
                p = g.tensor('pool_s%d' % s, dst.C, s)
                g.pool(dst, p)
                src = p
        skips = [(cats[4 - k], prev[k - 1]) for k in range(1, 5)]
        add_decoder(g, self, skips, x5, [o for _, _, o in self._UP])
        return g

    def forward(self, x):
        return self.engine.run(x)

    def forward_groups(self, input_groups):
        """[self(*inputs).detach() for inputs in input_groups] in one pass over the stacked batch (BatchNorm statistics
        per group, in order); see Engine.run_groups.  Extension for the co-teaching loop's augmentation forwards."""
        return self.engine.run_groups([tuple(g) if isinstance(g, (tuple, list)) else (g,) for g in input_groups])


class UNetsa(UNet):
    """models_singlemodalinput/UNet.py:168-208: UNet with a Spatial_Attention gate after every down block."""
    _ATTENTION = True


class UNet128(UNet):
    """models_singlemodalinput/UNet.py:210-240 (widths 128 ... 2048)."""
    _BASE = 128


class UNet32(UNet):
    """models_singlemodalinput/UNet.py:242-272 (widths 32 ... 512)."""
    _BASE = 32


class UNet16(UNet):
    """models_singlemodalinput/UNet.py:274-304 (widths 16 ... 256; the direct kernels mask the partial channel tile)."""
    _BASE = 16


class UNet8(UNet):
    """models_singlemodalinput/UNet.py:306-336 (widths 8 ... 128)."""
    _BASE = 8


class UNet4(UNet):
    """models_singlemodalinput/UNet.py:338-368 (widths 4 ... 64)."""
    _BASE = 4


class UNet2(UNet):
    """models_singlemodalinput/UNet.py:370-400 (widths 2 ... 32)."""
    _BASE = 2
