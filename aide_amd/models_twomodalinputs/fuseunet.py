"""MI355X-native drop-in for the reference's dual-modality `fuseunet`
(models_twomodalinputs/fuseunet.py:6-91): same class name, constructor and forward signature,
parameter registration order and state_dict keys; forward/backward run as hand-written HIP kernels
through aide_amd.engine."""
import torch.nn as nn

from ..engine import Engine, Graph
from .netblocks import UNet_basic_down_block, UNet_basic_up_block, Spatial_Attention, add_decoder


class fuseunet(nn.Module):
    _ATTENTION = False
    _SEPARATE = False       # fuseunetsaseparate: modal-1 pools its own stream instead of the fused tensor
    _M1 = ((3, 32), (64, 64), (128, 128), (256, 256), (512, 512))      # fuseunet.py:12-20
    _M2 = ((3, 32), (32, 64), (64, 128), (128, 256), (256, 512))       # fuseunet.py:24-32
    _UP = ((1024, 512, 512), (512, 256, 256), (256, 128, 128), (128, 64, 64))   # fuseunet.py:36-39

    def __init__(self, num_classes=2, reduction=16, dilation=4, learned_bilinear=False):
        nn.Module.__init__(self)
        # fuseunet: `reduction` / `dilation` are accepted and ignored, as in the reference (fuseunet.py:7);
        # fuseunetsa (fuseunet.py:93-136) registers a Spatial_Attention after every down block
        for m, widths in (('modal1', self._M1), ('modal2', self._M2)):
            for i, (a, b) in enumerate(widths, 1):
                setattr(self, '%s_downblock%d' % (m, i), UNet_basic_down_block(a, b))
                if self._ATTENTION:
                    setattr(self, '%s_sa%d' % (m, i), Spatial_Attention(input_channel=b, reduction=reduction,
                                                                         dilation=dilation))
                if i < 5:
                    setattr(self, '%s_maxpool%d' % (m, i), nn.MaxPool2d(kernel_size=2, stride=2))
        for i, (a, p, o) in enumerate(self._UP, 1):
            setattr(self, 'up_block%d' % i, UNet_basic_up_block(a, p, o, learned_bilinear))
        self.last_conv1 = nn.Conv2d(64, num_classes, 1, padding=0)
        self._engine = [Engine(self, self._build_graph, num_classes)]   # list: not a sub-module/buffer

    @property
    def engine(self):
        return self._engine[0]

    def _build_graph(self):
        """fuseunet.py:43-91 as a static graph over concatenation buffers.

        Stage-s fused skip y_s = cat(modal1_s, modal2_s) lives in the second half of the decoder's
        cat buffer [up | modal1_s | modal2_s]; pool(y_s) feeds modal1 (all channels) and modal2 (its
        own slice — max-pool is per-channel, so pool(x) == pool(y_s)[:, c1:])."""
        g = Graph()
        x1, x2 = g.input('modal1', 3), g.input('modal2', 3)
        c1 = [b for _, b in self._M1]
        c2 = [b for _, b in self._M2]
        prev = [p for _, p, _ in self._UP]
        cats = []
        for s in range(1, 5):                      # decoder k = 5 - s consumes skip s
            cats.append(g.tensor('cat_s%d' % s, 2 * prev[4 - s], s - 1))
        y5 = g.tensor('y5', c1[4] + c2[4], 4)
        src1, src2 = x1, x2
        for s in range(1, 6):
            if s <= 4:
                up_c = prev[4 - s]
                skip = cats[s - 1].slice(up_c, c1[s - 1] + c2[s - 1], 'y%d' % s)
            else:
                skip = y5
            d1 = skip.slice(0, c1[s - 1])
            d2 = skip.slice(c1[s - 1], c2[s - 1])
            lvl = s - 1
            # modal-2 first: in the backward schedule modal-1 (which reads all channels of the pooled
            # tensor) then writes the gradient first and modal-2 accumulates into its slice
            for m, src, d, cw in (('modal2', src2, d2, c2[s - 1]), ('modal1', src1, d1, c1[s - 1])):
                blk = getattr(self, '%s_downblock%d' % (m, s)).block
                t = g.tensor('%s_s%d_mid' % (m, s), cw, lvl)
                # the two encoders of a level are independent chains: modal-2's may run beside modal-1's (second stream)
                lane = 1 if (m == 'modal2' and not self._ATTENTION) else 0
                g.conv_bn_relu(src, t, blk.conv1, blk.bn1, lane=lane)
                if self._ATTENTION:            # y = sa(y) * y  (fuseunet.py:139-141): the gated tensor is the skip
                    pre = g.tensor('%s_s%d_pre' % (m, s), cw, lvl)
                    g.conv_bn_relu(t, pre, blk.conv2, blk.bn2)
                    g.spatial_attention(pre, d, getattr(self, '%s_sa%d' % (m, s)))
                else:
                    g.conv_bn_relu(t, d, blk.conv2, blk.bn2, lane=lane)
            if s < 5:
                p = g.tensor('pool_s%d' % s, skip.C, s)
                # (channels [0, c1) of the skip come from modal-1's chain, the rest from modal-2's: see Graph.pool)
                g.pool(skip, p, lane_split=None if self._ATTENTION else c1[s - 1])
                # max-pool is per channel: both encoders of fuseunetsaseparate read their own slice of pool(cat(y, x))
                src1 = p.slice(0, c1[s - 1]) if self._SEPARATE else p
                src2 = p.slice(c1[s - 1], c2[s - 1])
        skips = [(cats[4 - k], prev[k - 1]) for k in range(1, 5)]
        add_decoder(g, self, skips, y5, [o for _, _, o in self._UP])
        return g

    def forward(self, modal1_inputs, modal2_inputs):
        return self.engine.run(modal1_inputs, modal2_inputs)

    def forward_groups(self, input_groups):
        """[self(*inputs).detach() for inputs in input_groups] in one pass over the stacked batch (BatchNorm statistics
        per group, in order); see Engine.run_groups.  Extension for the co-teaching loop's augmentation forwards."""
        return self.engine.run_groups([tuple(g) if isinstance(g, (tuple, list)) else (g,) for g in input_groups])


class fuseunetsa(fuseunet):
    """models_twomodalinputs/fuseunet.py:93-221: fuseunet with a Spatial_Attention gate after every down block of both
    modalities (same constructor, forward signature, registration order and state_dict keys)."""
    _ATTENTION = True


class fuseunetsaseparate(fuseunetsa):
    """models_twomodalinputs/fuseunet.py:210-322: two independent attention encoders (3-32-64-128-256-512 each); the
    gated level outputs meet only in the decoder's skip concatenations."""
    _SEPARATE = True
    _M1 = fuseunet._M2                                                   # fuseunet.py:216-229
