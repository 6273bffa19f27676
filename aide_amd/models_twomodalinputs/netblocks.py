"""Parameter containers with the reference's block names, constructor signatures and registration
order (models_twomodalinputs/netblocks.py:9-33,128-147), so that state_dict keys and seeded
initialisation are identical.  They hold parameters only: the arithmetic of a whole network runs
through aide_amd.engine (HIP kernels), never through these sub-modules' own forward."""
import torch.nn as nn


def _no_forward(self, *a, **k):
    raise RuntimeError('aide_amd blocks are parameter containers; call the whole model '
                       '(fuseunet / UNet), whose forward runs the HIP engine')


def UNet_up_conv_bn_relu(input_channel, output_channel, learned_bilinear=False):
    # netblocks.py:9-19 — Sequential slot numbers are part of the checkpoint keys
    if learned_bilinear:
        layers = [nn.ConvTranspose2d(input_channel, output_channel, kernel_size=2, stride=2),
                  nn.BatchNorm2d(output_channel), nn.ReLU()]
    else:
        layers = [nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True),
                  nn.Conv2d(input_channel, output_channel, kernel_size=3, padding=1),
                  nn.BatchNorm2d(output_channel), nn.ReLU()]
    seq = nn.Sequential(*layers)
    seq.forward = lambda *a, **k: _no_forward(seq)
    return seq


class basic_block(nn.Module):
    # netblocks.py:21-33 : conv1, bn1, conv2, bn2 (+ a parameter-free ReLU)
    forward = _no_forward

    def __init__(self, input_channel, output_channel):
        super(basic_block, self).__init__()
        self.conv1 = nn.Conv2d(input_channel, output_channel, 3, padding=1)
        self.bn1 = nn.BatchNorm2d(output_channel)
        self.conv2 = nn.Conv2d(output_channel, output_channel, 3, padding=1)
        self.bn2 = nn.BatchNorm2d(output_channel)
        self.relu = nn.ReLU()


class Spatial_Attention(nn.Module):
    # netblocks.py:68-89 (dup UNet.py:85-107): conv1 (1x1, C -> C/r), conv2, conv3 (3x3, dilation = padding),
    # conv4 (1x1 -> 1), bn(1), sigmoid.  Runs as aide_amd/csrc/attention.hip through the engine's `sa` op.
    forward = _no_forward

    def __init__(self, input_channel, reduction=16, dilation=4):
        super(Spatial_Attention, self).__init__()
        r = input_channel // reduction
        self.conv1 = nn.Conv2d(input_channel, r, kernel_size=1, stride=1, padding=0)
        self.conv2 = nn.Conv2d(r, r, kernel_size=3, dilation=dilation, stride=1, padding=dilation)
        self.conv3 = nn.Conv2d(r, r, kernel_size=3, dilation=dilation, stride=1, padding=dilation)
        self.conv4 = nn.Conv2d(r, 1, kernel_size=1, stride=1, padding=0)
        self.bn = nn.BatchNorm2d(1)
        self.sigmoid = nn.Sigmoid()


class UNet_basic_down_block(nn.Module):
    # netblocks.py:128-135 ; the single-modal flavour (UNet.py:110-121) adds `down_size`
    forward = _no_forward

    def __init__(self, input_channel, output_channel, down_size=False):
        super(UNet_basic_down_block, self).__init__()
        self.block = basic_block(input_channel, output_channel)
        self.down_size = down_size


class UNet_basic_up_block(nn.Module):
    # netblocks.py:137-147
    forward = _no_forward

    def __init__(self, input_channel, prev_channel, output_channel, learned_bilinear=False):
        super(UNet_basic_up_block, self).__init__()
        self.bilinear_up = UNet_up_conv_bn_relu(input_channel, prev_channel, learned_bilinear)
        self.block = basic_block(prev_channel * 2, output_channel)
        self.learned_bilinear = learned_bilinear


def add_decoder(g, module, skips, bottom, widths):
    """Shared decoder wiring (netblocks.py:143-147 inside fuseunet.py:85-89 / UNet.py:159-164).
    skips[k] / cat buffers are laid out [upsampled | skip] so that torch.cat((x, pre_feature_map), 1)
    is free.  `skips` = list of (cat_buffer GTensor, prev_channels) for up_block1..4."""
    x = bottom
    for k in range(1, 5):
        blk = getattr(module, 'up_block%d' % k)
        cat, prev = skips[k - 1]
        dst = cat.slice(0, prev, 'up%d' % k)
        if blk.learned_bilinear:
            g.convT_bn_relu(x, dst, blk.bilinear_up[0], blk.bilinear_up[1])
        else:
            u = g.tensor('upsampled%d' % k, x.C, cat.level)
            g.upsample(x, u)
            g.conv_bn_relu(u, dst, blk.bilinear_up[1], blk.bilinear_up[2])
        out_c = widths[k - 1]
        t = g.tensor('up%d_mid' % k, out_c, cat.level)
        e = g.tensor('up%d_out' % k, out_c, cat.level)
        g.conv_bn_relu(cat, t, blk.block.conv1, blk.block.bn1)
        g.conv_bn_relu(t, e, blk.block.conv2, blk.block.bn2)
        x = e
    g.head(x, module.last_conv1)
