"""Whole-network executor for the U-Net family on MI355X.

A model describes itself once as a small static graph (conv+BN+ReLU, ConvT+BN+ReLU, max-pool,
bilinear x2, 1x1 head) over *channel-slice views* of pre-planned NCHW buffers.  Because every
producer writes straight into its slice of the consumer's concatenation buffer, torch.cat
(models_twomodalinputs/fuseunet.py:49-81, netblocks.py:145) never materialises.  For each
(N, H, W, mode) the graph is compiled into a Plan: buffers, packed filters, split-K workspaces and
the backward schedule with statically resolved overwrite/accumulate decisions for tensors that have
several consumers (skip connections).  The forward and the backward of the whole network are then two
flat launch sequences on the current HIP stream, wrapped in ONE torch.autograd.Function so the
reference training loops (`net(...)`, `loss.backward()`, `optimizer.step()`) work unchanged.

All arithmetic is in libaide_hip.so (see include/aide_hip.h); torch only owns memory and streams.
"""
import ctypes

import torch

from . import ops
from ._lib import lib, check as _check
from .tape import Tape

# bumped by aide_amd.optim.Adam (which updates parameters through raw pointers, invisible to
# tensor._version) so that cached packed filters are refreshed
PARAM_EPOCH = [0]
# bumped by every train-mode forward (its BatchNorm kernels update the running statistics through raw pointers): the cached
# eval-mode coefficients of every plan are stale afterwards
STATS_EPOCH = [0]


class GTensor(object):
    """A [N, C, H>>level, W>>level] activation: either a root buffer or a channel slice of one."""

    def __init__(self, name, channels, level, root=None, c0=0, is_input=False):
        self.name, self.C, self.level = name, channels, level
        self.root = root if root is not None else self
        self.c0 = c0
        self.is_input = is_input

    def slice(self, c0, channels, name=None):
        assert 0 <= c0 and c0 + channels <= self.C
        return GTensor(name or '%s[%d:%d]' % (self.name, c0, c0 + channels), channels, self.level,
                       self.root, self.c0 + c0)


class Graph(object):
    def __init__(self):
        self.roots, self.inputs, self.ops = [], [], []
        self.output = None

    def tensor(self, name, channels, level):
        t = GTensor(name, channels, level)
        self.roots.append(t)
        return t

    def input(self, name, channels):
        t = GTensor(name, channels, 0, is_input=True)
        self.inputs.append(t)
        return t

    def conv_bn_relu(self, src, dst, conv, bn, lane=0):
        """lane = 1 marks a chain that is independent of the lane-0 ops next to it (the second encoder of a two-modality
        net): the forward pass may run it on a second stream (Plan._forward_impl)."""
        self.ops.append(dict(kind='conv', src=src, dst=dst, conv=conv, bn=bn, lane=lane))

    def convT_bn_relu(self, src, dst, conv, bn):
        self.ops.append(dict(kind='convT', src=src, dst=dst, conv=conv, bn=bn))

    def pool(self, src, dst, lane_split=None):
        """lane_split = c: channels [0, c) of src were written by lane-0 ops, [c, C) by the lane-1 chain.  Max-pooling is per
        channel, so the forward pass may pool each part on the stream that produced it: lane 1 then runs from level to level
        without waiting for lane 0 (Plan._forward_impl); the backward pass treats the op as one."""
        self.ops.append(dict(kind='pool', src=src, dst=dst, lane_split=lane_split))

    def upsample(self, src, dst):
        self.ops.append(dict(kind='up', src=src, dst=dst))

    def head(self, src, conv):
        self.ops.append(dict(kind='head', src=src, conv=conv))

    def spatial_attention(self, src, dst, mod):
        """dst = mod(src) * src  (Spatial_Attention, netblocks.py:68-89 + fuseunet.py:139-141)"""
        self.ops.append(dict(kind='sa', src=src, dst=dst, mod=mod))


# multiplies a launch executes per algorithmic (direct-convolution) multiply, by conv mode
WINO_EXEC = 16.0 / 36.0
BF16 = 16                      # conv mode id of the bf16-MFMA kernels (conv3x3_bf16.hip)
EXEC_FRAC = {0: 1.0, 2: 16.0 / 36.0, 4: 36.0 / 144.0, BF16: 1.0}
# profiler tags = the kernel that does the work of one conv operator call (its split reduce rides along)
FWD_TAG = {0: 'conv3x3_mfma_kernel', 2: 'conv3x3_wino_kernel', 4: 'conv3x3_wino4_kernel', BF16: 'conv3x3_bf16_kernel'}
WGRAD_TAG = {0: 'conv3x3_wgrad_kernel', 2: 'conv3x3_wgrad_wino_kernel', 4: 'conv3x3_wgrad4_kernel',
             BF16: 'conv3x3_wgrad_bf16_kernel'}
PRECISIONS = ('fp32', 'bf16')
import os as _os


class EngineConfig(object):
    """The in-process switches of ONE Engine (`net.engine.config`).  Every alternative is the same computation on another
    schedule, storage form or kernel family: the tests that prove it flip a switch on their own network
    (tests/test_gpu_models.py, test_gpu_steps.py, test_gpu_bf16.py, test_gpu_fullsize.py).  Assigning a switch drops the
    engine's plans, launch tapes and shared filter packs, so the next forward is built under the new setting -- nothing
    replays an old launch sequence, and no other engine is affected.  The only ENVIRONMENT switches of the package are
    AIDE_HIP_LIB (library path), AIDE_DIST_BACKEND (dry-run backend), AIDE_PICK_STREAMS, AIDE_REPLAY, AIDE_DIRECT_GRADS and the
    measurement aid AIDE_ABI_COVERAGE; every A-B switch whose alternative was measured and lost is gone with its code path."""
    DEFAULTS = dict(
        use_winograd=True,          # Winograd kernels at all (False: direct implicit GEMM everywhere)
        use_winograd_dgrad=True,    # ... for the dgrad direction
        use_winograd4=True,         # F(4x4) where the layer rule picks it (False: F(2x2) / direct)
        store_bf16=True,            # precision='bf16': z / dz stored as bf16
        store_a_bf16=True,          # precision='bf16': activations stored as bf16 where every reader / writer allows it
        store_g_bf16=True,          # ... and the gradients of those activations
        replay=_os.environ.get('AIDE_REPLAY', '1') != '0',     # launch tapes (aide_amd/tape.py)
        shared_packs=True,          # plans of an engine share packed filters; forward-only plans pack no dgrad direction
        fold_eval_bn=True,          # eval-mode BatchNorm + ReLU in the conv epilogue (no pass over the conv output)
        tail_wgrad_main=True,       # the weight gradient of the LAST op of the backward pass on the main stream
        dual_fwd=True,              # lane-1 chains of the forward pass on a second stream
        free_lane=True,             # lane 1 pools its own channels and runs ahead (no fork / join per level)
        handover_on_kernel=True,    # the dz hand-over event of a layer rides on its BatchNorm backward's last dispatch
        w4_half_tile=True,          # F(4x4) weight gradient also for Co = 32 layers (trailing half tile computed and dropped)
        grouped_bn=True,            # stacked plans: BatchNorm of all groups in one launch sequence
        lazy_bn=True,               # forward-only stacked plans: BatchNorm + ReLU of a layer in its reader's loader
        direct_grads=_os.environ.get('AIDE_DIRECT_GRADS', '1') != '0',   # parameter gradients assigned by the engine
        fuse_pool_bwd=True,         # max-pooling backward inside the BatchNorm backward of the layer(s) it pooled
        fuse_head_bwd=True,         # the head's data gradient formed inside the BatchNorm backward of the layer under it
        lazy_head=True,             # BatchNorm + ReLU of the layer under the head applied in the head's loaders (no pass, no activation)
        fuse_pool_fwd=True,         # max-pooling forward inside the BatchNorm forward of the layer(s) it pools
    )

    def __init__(self, on_change=None):
        object.__setattr__(self, '_on_change', on_change)
        for k, v in self.DEFAULTS.items():
            object.__setattr__(self, k, v)

    def __setattr__(self, k, v):
        if k not in self.DEFAULTS:
            raise AttributeError('aide_amd: the engine has no switch %r (switches: %s)' % (k, ', '.join(sorted(self.DEFAULTS))))
        v = bool(v)
        if getattr(self, k) != v:
            object.__setattr__(self, k, v)
            if self._on_change is not None:
                self._on_change()

    def snapshot(self):
        return tuple(getattr(self, k) for k in sorted(self.DEFAULTS))


FLUSH_EVERY = 4                # layers per batched slab reduce (a constant: 6 -> 4 C2 +0.5 / +0.8 % on two boxes, C4 +0.2 %, C3 +0.1 %; 2, 3, 5, 8, 12 and one flush at the end all behind 4)


def conv_mode(cfg, n, cin, h, w, cout):
    """0 = direct implicit GEMM, 2 = Winograd F(2x2,3x3), 4 = Winograd F(4x4,3x3).  Static rule from the layer
    sweep (tools/bench_conv.py wino): F(4x4) wins on every layer of >= 8 GFLOP and on the >= 128x128-channel
    layers (1.3-1.5x over F(2x2) on the 19 / 39 GFLOP decoder layers); F(2x2) elsewhere; direct where neither
    is supported."""
    if cfg.use_winograd and cfg.use_winograd4 and lib.aide_conv3x3_wino4_supported(cin, h, w, cout) and w != 16:
        # (w == 16: the kernel's image-pair tile is 11 % faster than F(2x2) there, the step 0.6 % slower -- 4x larger
        # filter pack for the 8 M bottleneck parameters, twice the slabs for BatchNorm to sum; round 6, with the one-pass
        # BatchNorm kernels: 622.0 -> 619.6 images/s, same process, tools/r6_ab_config.py)
        flops = 2.0 * n * h * w * cin * cout * 9
        if cout % 64:                       # trailing half block computed and dropped: 42 vs 52 us on 32->32 @256x256 alone,
            return 0                        # level in the step (round 4 again: C2 620.2 / 620.4, C3 155.2 / 155.4)
        if flops >= 8e9 or cin * cout >= 128 * 128:
            return 4
    return 2 if use_winograd(cfg, n, cin, h, w, cout) else 0


# layer -> weight-gradient algorithm, for measurement only (tools/r6_wgrad_choice.py): {(cout, cin, h, w): 2 | 4}.  Empty in
# the product: the static rule in Plan.__init__ decides.
WGRAD_OVERRIDE = {}


def use_winograd(cfg, n, cin, h, w, cout):
    """Static (deterministic) choice between the Winograd and the direct conv kernel: the layer sweep
    (tools/bench_conv.py all) has Winograd ahead on every layer shape it supports (1.2x-1.9x; the one
    exception, 64->128 @64x64 forward, loses 4 us), so it is used wherever it is supported."""
    return bool(cfg.use_winograd and lib.aide_conv3x3_wino_supported(cin, h, w, cout))


def _preferred(dev):
    """the per-device streams every plan uses instead of creating its own: a data-parallel rank's measured choice
    (aide_amd/streams.py), else ONE weight-gradient stream and ONE lane / pack stream per device shared by every plan of
    every engine (the HIP runtime maps a process' streams onto 4 hardware queues: HISTORY §7a)"""
    from . import streams as _streams
    d = torch.device(dev)
    idx = d.index if d.index is not None else torch.cuda.current_device()
    pref = _streams.PREFERRED.get(idx)
    if pref is None:
        pref = _streams.PREFERRED[idx] = dict(side=torch.cuda.Stream(device=d), lane=torch.cuda.Stream(device=d))
    return pref


def _side_stream(dev):
    """Stream of the weight-gradient kernels (shared per device, see _preferred)."""
    return _preferred(dev)['side']


class _Cover(object):
    """Tracks which channel ranges of a gradient buffer have been written during the backward
    schedule: first writer overwrites, later writers accumulate, gaps are zero-filled once."""

    def __init__(self):
        self.iv = []

    def write(self, c0, c1):
        """-> (accumulate, [uncovered sub-intervals to zero-fill first])"""
        covered = [(max(a, c0), min(b, c1)) for a, b in self.iv if a < c1 and b > c0]
        if not covered:
            self._add(c0, c1)
            return False, []
        gaps, cur = [], c0
        for a, b in sorted(covered):
            if a > cur:
                gaps.append((cur, a))
            cur = max(cur, b)
        if cur < c1:
            gaps.append((cur, c1))
        self._add(c0, c1)
        return True, gaps

    def _add(self, c0, c1):
        iv = sorted(self.iv + [(c0, c1)])
        out = [iv[0]]
        for a, b in iv[1:]:
            if a <= out[-1][1]:
                out[-1] = (out[-1][0], max(out[-1][1], b))
            else:
                out.append((a, b))
        self.iv = out


class Plan(object):
    def __init__(self, graph, params, n, h, w, device, training, precision='fp32', groups=1, shared=None, cfg=None):
        self.g, self.N, self.H, self.W, self.dev, self.training = graph, n, h, w, device, training
        cfg = self.cfg = cfg if cfg is not None else EngineConfig()
        self.precision = precision
        # Packed filters are a function of (layer, direction, kernel family) only: the plans of one engine (the stacked
        # augmentation pass, the training batch, an evaluation batch) share them, and a plan that finds a pack already
        # refreshed for the current weights by another plan skips it (`shared['fresh']`: id(tensor) -> pack key).
        self._shared = shared if shared is not None else dict(buf={}, fresh={}, streams=None)
        # groups > 1: the batch is `groups` independent forwards of n / groups images stacked along N (forward only):
        # convolutions / pooling / up-sampling run once over all of them, every BatchNorm takes its batch statistics
        # and updates its running statistics per group, in order -- the semantics of `groups` sequential forwards
        self.groups = groups
        assert n % groups == 0
        bf16 = precision == 'bf16'
        self.pindex = {id(p): i for i, p in enumerate(params)}
        f32 = dict(device=device, dtype=torch.float32)
        forward_only = (groups > 1 or not training) and cfg.shared_packs   # no backward pass will ever run on this plan: no dgrad-direction packs

        def pack_buf(conv, direction, mode, make):
            key = (id(conv), direction, mode, device.index)
            t = self._shared['buf'].get(key)
            if t is None:
                t = self._shared['buf'][key] = make()
            return t
        self.act = {}
        self.grad = {}
        self.steps = []
        max_dz = max_wg = max_bnc = max_sk = 0
        for op in graph.ops:
            st = dict(op)
            if op['kind'] in ('conv', 'convT'):
                conv, src, dst = op['conv'], op['src'], op['dst']
                hh, ww = h >> dst.level, w >> dst.level
                cout = dst.C
                st['z'] = torch.empty(n, cout, hh, ww, **f32)
                for k in ('mean', 'rstd', 'scale', 'shift', 'fbias'):
                    st[k] = torch.empty(cout, **f32)
                max_dz = max(max_dz, n * cout * hh * ww)
                max_bnc = max(max_bnc, cout)
                if op['kind'] == 'conv':
                    cin = src.C
                    need_dg = not src.root.is_input and not forward_only
                    # Winograd F(2x2,3x3) where it is supported and measured faster (16 MFMA-multiplies
                    # per output instead of 36); the direct implicit GEMM otherwise
                    st['wino_f'] = conv_mode(cfg, n, cin, hh, ww, cout)
                    st['wino_d'] = conv_mode(cfg, n, cout, hh, ww, cin) if (need_dg and cfg.use_winograd_dgrad) else 0
                    # precision='bf16': bf16 operands / fp32 accumulation wherever the bf16 kernels cover the
                    # layer shape, the fp32 kernels elsewhere (narrow deep levels of small inputs)
                    if bf16 and lib.aide_conv3x3_bf16_supported(cin, hh, ww, cout):
                        st['wino_f'] = BF16
                    if bf16 and need_dg and lib.aide_conv3x3_bf16_supported(cout, hh, ww, cin):
                        st['wino_d'] = BF16
                    st['wf'] = st['wd'] = st['uf'] = st['ud'] = None
                    st['plan_f'] = st['plan_d'] = 0
                    if st['wino_f'] == BF16:
                        st['uf'] = pack_buf(conv, 'f', BF16, lambda: ops.bf16_pack_alloc(cout, cin, device))
                        st['plan_f'] = lib.aide_conv3x3_bf16_splitk(n, cin, hh, ww, cout) << 8
                    elif st['wino_f'] == 4:
                        st['uf'] = pack_buf(conv, 'f', 4, lambda: torch.empty(cin, 36, cout, **f32))
                        st['plan_f'] = lib.aide_conv3x3_wino4_splitk(n, cin, hh, ww, cout) << 8
                    elif st['wino_f']:
                        st['uf'] = pack_buf(conv, 'f', 2, lambda: torch.empty(ops.pad_to(cin, 8), 16, cout, **f32))
                        st['plan_f'] = lib.aide_conv3x3_wino_splitk(n, cin, hh, ww, cout) << 8
                    else:
                        st['wf'] = pack_buf(conv, 'f', 0, lambda: torch.empty(ops.pad_to(cin, ops.conv_chunk(cin)), 9, cout, **f32))
                        st['plan_f'] = lib.aide_conv3x3_plan(n, cin, hh, ww, cout)
                    if need_dg and st['wino_d'] == BF16:
                        st['ud'] = pack_buf(conv, 'd', BF16, lambda: ops.bf16_pack_alloc(cin, cout, device))
                        st['plan_d'] = lib.aide_conv3x3_bf16_splitk(n, cout, hh, ww, cin) << 8
                    elif need_dg and st['wino_d'] == 4:
                        st['ud'] = pack_buf(conv, 'd', 4, lambda: torch.empty(cout, 36, cin, **f32))
                        st['plan_d'] = lib.aide_conv3x3_wino4_splitk(n, cout, hh, ww, cin) << 8
                    elif need_dg and st['wino_d']:
                        st['ud'] = pack_buf(conv, 'd', 2, lambda: torch.empty(ops.pad_to(cout, 8), 16, cin, **f32))
                        st['plan_d'] = lib.aide_conv3x3_wino_splitk(n, cout, hh, ww, cin) << 8
                    elif need_dg:
                        st['wd'] = pack_buf(conv, 'd', 0, lambda: torch.empty(ops.pad_to(cout, ops.conv_chunk(cout)), 9, cin, **f32))
                        st['plan_d'] = lib.aide_conv3x3_plan(n, cout, hh, ww, cin)
                    max_sk = max(max_sk, lib.aide_conv3x3_ws_bytes(n, hh, ww, cout, st['plan_f'] >> 8),
                                 lib.aide_conv3x3_ws_bytes(n, hh, ww, cin, st['plan_d'] >> 8) if need_dg else 0)
                    # weight gradient: transposed F(4x4,3x3) wherever supported (ahead on every layer of the sweep),
                    # else transposed F(2x2,3x3), else the direct kernel
                    if bf16 and lib.aide_conv3x3_wgrad_bf16_supported(cout, cin, hh, ww):
                        st['wino_w'] = BF16
                        st['wg_bytes'] = lib.aide_conv3x3_wgrad_bf16_ws_bytes(n, cout, cin, hh, ww, 0)
                    elif cfg.use_winograd and cfg.use_winograd4 and (cout % 64 == 0 or cfg.w4_half_tile) and \
                            WGRAD_OVERRIDE.get((cout, cin, hh, ww), 4) == 4 and \
                            lib.aide_conv3x3_wgrad_wino4_supported(cout, cin, hh, ww):
                        # (a trailing half tile -- 32->32 @256x256 -- is 71 -> 56 us alone.  In round 2 the step lost 0.5 % with it:
                        # the 144 KB workgroups kept the main stream's kernels off the CUs; with the backward pass as it is now
                        # these layers are the last thing the weight-gradient stream does and the step gains: C2 621.9 / 620.9,
                        # 621.8 / 620.7, 621.2 / 619.6 same box, C3 level -- config.w4_half_tile)
                        st['wino_w'] = 4
                        # the LAST such launch of a backward pass whose dependent chain ends with it (every op before it in
                        # the graph is a stem conv without a data gradient -- the single-encoder U-Nets): nothing is left to
                        # share the chip with, so it takes all of it (the default leaves half to the dependent chain)
                        if all(s0['kind'] == 'conv' and s0['src'].root.is_input for s0 in self.steps):
                            st['wg_target'] = 256
                        st['wg_bytes'] = lib.aide_conv3x3_wgrad_wino4_ws_bytes_t(n, cout, cin, hh, ww, st.get('wg_target', 0))
                    elif cfg.use_winograd and lib.aide_conv3x3_wgrad_wino_supported(cout, cin, hh, ww):
                        st['wino_w'] = 2
                        st['wg_bytes'] = lib.aide_conv3x3_wgrad_wino_ws_bytes(n, cout, cin, hh, ww)
                    else:
                        st['wino_w'] = 0
                        st['wg_bytes'] = lib.aide_conv3x3_wgrad_ws_bytes(n, cout, cin, hh, ww)
                    max_wg = max(max_wg, st['wg_bytes'])
                    # BatchNorm statistics from the conv epilogue (big planes, non-split F(4x4) forward, ungrouped training)
                    st['stats'] = None
                    if training and (st['plan_f'] >> 8) <= 1 \
                            and (hh * ww) % 4 == 0 and lib.aide_bn_two_pass(n // max(groups, 1), cout, hh, ww):
                        # F(4x4) forward only.  The same epilogue in the direct and F(2x2) kernels was built and measured: C2
                        # 575 -> 572 images/s (their epilogues are short and the butterflies cost more than the saved pass) and
                        # the fp32 partial sums of the 3->64 stem at 320x320 (|mean| >> std) moved a gradient norm by 3e-3.
                        parts = lib.aide_conv3x3_wino4_stats_parts(n, hh, ww) if st['wino_f'] == 4 else 0
                        # (the same epilogue in the bf16 forward kernel -- sums of the stored bf16 z, one wave butterfly per
                        # channel tile -- was built in round 3 and measured: C5 444 -> 434 images/s, dropped)
                        if parts > 0:
                            st['stats_parts'] = parts
                            st['stats'] = torch.empty(cout * parts * 2, **f32)
                    # eval mode (the per-case inference loop): BatchNorm of the running statistics + ReLU as the epilogue of the
                    # F(4x4) forward kernel -- the conv writes the activation, z is never materialised
                    st['fold'] = not training and cfg.fold_eval_bn and (
                        (st['wino_f'] == 4 and ww != 16) or (st['wino_f'] == 0 and (st['plan_f'] >> 8) <= 1))
                    st['pack_key'] = None
                    st['flops'] = 2.0 * n * hh * ww * cout * cin * 9      # algorithmic, per launch
                    # bf16 mode keeps the conv output z (read only by BatchNorm) and its gradient dz (read only by the
                    # bf16 dgrad / wgrad kernels, which round it to bf16 anyway -- storing it narrow changes nothing
                    # numerically) in HBM as bf16: half the bytes of the conv-output write, of four BatchNorm reads and
                    # of the dz write + two reads
                    st['dz_bf16'] = (st['wino_w'] == BF16 and (not need_dg or st['wino_d'] == BF16) and cfg.store_bf16)
                    if st['wino_f'] == BF16 and cfg.store_bf16:
                        st['z'] = torch.empty(n, cout, hh, ww, device=device, dtype=torch.bfloat16)
                else:
                    cin = src.C
                    st['wg_bytes'] = lib.aide_convT2x2_wgrad_ws_bytes(n, cin, cout, h >> src.level, w >> src.level)
                    max_wg = max(max_wg, st['wg_bytes'])
            elif op['kind'] == 'head':
                src = op['src']
                k = op['conv'].out_channels
                max_wg = max(max_wg, lib.aide_head1x1_ws_bytes(src.C, k))
            elif op['kind'] == 'sa':
                src, mod = op['src'], op['mod']
                hh, ww = h >> src.level, w >> src.level
                r = mod.conv1.out_channels
                for k in ('t1', 't2', 't3'):
                    st[k] = torch.empty(n, r, hh, ww, **f32)
                st['t4'] = torch.empty(n, 1, hh, ww, **f32)
                st['gate'] = torch.empty(n, hh, ww, **f32)
                st['stat'] = torch.empty(2, **f32)
            self.steps.append(st)
        self._plan_lazy_bn()
        self._plan_lazy_head()
        # activation buffers.  bf16 mode stores an activation buffer as bf16 when everything that touches it can: written by
        # BatchNorm-apply / pooling / up-sampling, read by bf16 convolutions (forward AND weight gradient), pooling,
        # up-sampling or the head.  For a conv operand that is numerically free (the kernels round it anyway), max-pooling
        # commutes with the rounding; up-sampling and the head then see rounded inputs (what torch.autocast gives them).
        # Gradients of activations stay fp32.
        narrow = {id(t): bf16 and cfg.store_a_bf16 for t in graph.roots}
        for st in self.steps:
            kind = st['kind']
            if kind == 'conv':
                if not (st['wino_f'] == BF16 and st['wino_w'] == BF16):
                    narrow[id(st['src'].root)] = False
            elif kind in ('convT', 'sa'):
                narrow[id(st['src'].root)] = False
            if kind == 'sa':
                narrow[id(st['dst'].root)] = False
        for t in graph.roots:
            dt = torch.bfloat16 if narrow.get(id(t)) else torch.float32
            self.act[id(t)] = torch.empty(n, t.C, h >> t.level, w >> t.level, device=device, dtype=dt)
        self._plan_pool_fwd()
        # the gradient of a bf16-stored activation buffer is stored as bf16 too when every conv that reads the buffer also
        # runs its dgrad on the bf16 kernel (the other writers / readers -- pooling, up-sampling, head, BatchNorm backward --
        # are storage-generic).  Writers after the first accumulate in fp32 and round once per write (torch.autocast's
        # bf16 activation gradients accumulate the same way).
        self.narrow_grad = {k: v and cfg.store_g_bf16 for k, v in narrow.items()}
        for st in self.steps:
            if st['kind'] == 'conv' and not st['src'].root.is_input and st['wino_d'] != BF16:
                self.narrow_grad[id(st['src'].root)] = False
        self.bn_ws = ops.bn_ws(max(max_bnc, 1), device)
        self.sk_ws = torch.empty(max(max_sk // 4, 1), **f32)
        self.lane_b = None               # second forward stream + its own BatchNorm / split-K workspaces (lane-1 chains)
        if any(st.get('lane') for st in self.steps):
            pref = _preferred(device)
            self.lane_b = pref['lane'] if (pref is not None and pref.get('lane') is not None) else torch.cuda.Stream(device=device)
            self.bn_ws_b = ops.bn_ws(max(max_bnc, 1), device)
            self.sk_ws_b = torch.empty(max(max_sk // 4, 1), **f32)
            self.ev_lane_fork, self.ev_lane_join = ops.new_event(), ops.new_event()
            for st in self.steps:        # one event per split pooling op: "lane 1's half of this level is pooled"
                if st['kind'] == 'pool' and st.get('lane_split'):
                    st['ev_b'] = ops.new_event()
        self._max_dz, self._max_wg = max_dz, max_wg
        self.wg_ws = self.wg_queue = self._wq_active = None
        self._bwd_ready = False
        self.profiler = None             # set by Engine (bench.py's per-kernel HIP-event timing)
        self._pack_key, self._pack_tabs, self._pack_ids, self.side_fwd = None, {}, None, None
        # the weight-gradient stream of a plan that will run backward passes; it also carries the filter re-layout at the start
        # of the forward pass: the HIP runtime maps a process' streams onto 4 hardware queues, and the
        # data-parallel path needs one of them for RCCL's own stream (aide_amd/distributed.py)
        self.side = _side_stream(device) if (training and groups == 1) else None
        self._convs = self._conv_wslots = None
        self._gate_conv = None           # the first conv that needs the side-stream filter packs
        self._tape_f = self._tape_b = None
        self._coef_key, self._coef_tensors, self._coef_ok, self._coef_next, self._coef_bns = None, None, False, None, []
        self._fp = None
        self._fp_slots = self._fp_bns = None
        self.overlap = True              # weight gradients on a side stream (see backward)
        self.trace = None                # tools/phase_trace.py: callable(direction, step) before every op
        self.serial = 0                  # forwards run on this plan; _NetFunction.backward checks it still owns the buffers
        self.key = None

    def _plan_lazy_bn(self):
        """Forward-only stacked plans (the no-grad augmentation passes of the co-teaching step,
        trainchaos_proposed_30cases1labeled.py:263-281): nothing is kept for a backward pass, so the BatchNorm + ReLU of a
        layer whose ONLY reader is an F(4x4) convolution is applied in that convolution's loader (SURVEY 8b
        in_prologue{bn_relu}) -- the producer writes its raw output z straight into the reader's input slot with the
        statistics from its epilogue, a one-wave-per-channel launch turns them into the reader's per-group (scale, shift)
        table, and the normalising pass over the tensor (one read + one write) never runs.  Channels of the reader's input
        that are materialised activations (a skip half of a concat buffer) get (1, 0): ReLU of a non-negative value."""
        self.lazy_bn = 0
        if not (self.training and self.groups > 1) or not self.cfg.lazy_bn:
            return
        for st in self.steps:
            if st['kind'] != 'conv' or st.get('stats') is None:
                continue
            dst = st['dst']
            readers = [o for o in self.steps if o.get('src') is not None and o['src'].root is dst.root
                       and o['src'].c0 < dst.c0 + dst.C and dst.c0 < o['src'].c0 + o['src'].C]
            if len(readers) != 1:
                continue
            rd = readers[0]
            src = rd['src']
            if rd['kind'] != 'conv' or rd['wino_f'] != 4 or src.C > 1024 or not (src.c0 <= dst.c0 and dst.c0 + dst.C <= src.c0 + src.C):
                continue
            if rd.get('in_tab') is None:
                tab = torch.zeros(self.groups, src.C, 2, device=self.dev, dtype=torch.float32)
                tab[:, :, 0] = 1.0
                rd['in_tab'] = tab
            st['lazy_to'], st['tab_c0'] = rd, dst.c0 - src.c0
            st['z'] = None                 # the raw output lives in the reader's input slot
            self.lazy_bn += 1

    def _plan_lazy_head(self):
        """Training plans: the layer under the 1x1 head (its only reader) with statistics from its conv epilogue never runs its
        normalising pass -- a one-wave-per-channel launch turns the statistics into (scale, shift), the head's forward and weight
        gradient apply BatchNorm + ReLU while they read z (ops.head1x1_fwd_bn / head1x1_wgrad_bn), and the head's data gradient is
        formed inside that layer's BatchNorm backward (fuse_head_bwd): its activation is never stored."""
        if not (self.training and self.groups == 1 and self.cfg.lazy_head and self.cfg.fuse_head_bwd):
            return
        for hst in self.steps:
            if hst['kind'] != 'head':
                continue
            src = hst['src']
            readers = [o for o in self.steps if o.get('src') is not None and o['src'].root is src.root
                       and o['src'].c0 < src.c0 + src.C and src.c0 < o['src'].c0 + o['src'].C]
            prods = [o for o in self.steps if o['kind'] == 'conv' and o['dst'].root is src.root
                     and (o['dst'].c0, o['dst'].C) == (src.c0, src.C)]
            if len(readers) != 1 or len(prods) != 1:
                continue
            o = prods[0]
            zn, zc, zh, zw = o['z'].shape
            if o.get('stats') is None or o['z'].dtype != torch.float32 or o.get('dz_bf16') or o.get('lazy_to') is not None or \
                    not lib.aide_bn_one_pass(zn, zc, zh, zw):
                continue
            o['head_lazy'] = True
            o['head_tab'] = torch.zeros(1, zc, 2, device=self.dev, dtype=torch.float32)
            hst['lazy_prod'] = o

    def _plan_pool_fwd(self):
        """Training plans: a max-pooling whose source is exactly the output of one or several conv + BN + ReLU layers that run the
        one-pass BatchNorm on z as it is (no split-K slabs, no conv-epilogue statistics) is written by those layers' BatchNorm
        kernels themselves (ops.bn_train_fwd_pool): the pooling pass and its launch disappear (the first two levels of the U-Nets)."""
        if not (self.training and self.cfg.fuse_pool_fwd) or (self.groups > 1 and not self.cfg.grouped_bn):
            return
        m = self.N // max(self.groups, 1)
        for pst in self.steps:
            if pst['kind'] != 'pool':
                continue
            src, dst = pst['src'], pst['dst']
            if src.root.is_input or self.act[id(src.root)].dtype != torch.float32 or self.act[id(dst.root)].dtype != torch.float32:
                continue
            prods = [o for o in self.steps if o['kind'] == 'conv' and o['dst'].root is src.root
                     and src.c0 <= o['dst'].c0 and o['dst'].c0 + o['dst'].C <= src.c0 + src.C]
            cover = sorted((o['dst'].c0, o['dst'].c0 + o['dst'].C) for o in prods)
            if not cover or cover[0][0] != src.c0 or cover[-1][1] != src.c0 + src.C or any(a[1] != b[0] for a, b in zip(cover, cover[1:])):
                continue
            ok = True
            for o in prods:
                if o.get('z') is None or o['z'].dtype != torch.float32:
                    ok = False
                    break
                zn, zc, zh, zw = o['z'].shape
                if o.get('stats') is not None or (o['plan_f'] >> 8) > 1 or o.get('lazy_to') is not None or o.get('head_lazy') or \
                        not lib.aide_bn_relu_bwd_pool_supported(m, zc, zh, zw):
                    ok = False
            if not ok:
                continue
            for o in prods:
                o['pool_out'] = dst.slice(o['dst'].c0 - src.c0, o['dst'].C)
            pst['fwd_fused'] = True

    # ------------------------------------------------------------------ helpers
    def view(self, t, inputs=None):
        if t.root.is_input:
            x = inputs[self.g.inputs.index(t.root)]
            return x if (t.c0 == 0 and t.C == t.root.C) else x[:, t.c0:t.c0 + t.C]
        buf = self.act[id(t.root)]
        return buf if (t.c0 == 0 and t.C == t.root.C) else buf[:, t.c0:t.c0 + t.C]

    def gview(self, t):
        buf = self.grad[id(t.root)]
        return buf if (t.c0 == 0 and t.C == t.root.C) else buf[:, t.c0:t.c0 + t.C]

    def _prepare_backward(self):
        """Allocate gradient buffers and resolve overwrite/accumulate per backward write."""
        f32 = dict(device=self.dev, dtype=torch.float32)
        n, h, w = self.N, self.H, self.W
        for t in self.g.roots:
            dt = torch.bfloat16 if self.narrow_grad.get(id(t)) else torch.float32
            self.grad[id(t)] = torch.empty(n, t.C, h >> t.level, w >> t.level, device=self.dev, dtype=dt)
        # one dz buffer per conv: the weight-gradient kernels run on a side stream and may still be
        # reading dz of layer L while the main stream already produces dz of layer L-1
        for st in self.steps:
            if st['kind'] in ('conv', 'convT'):
                st['dz'] = torch.empty(st['z'].shape, device=self.dev,
                                       dtype=torch.bfloat16 if st.get('dz_bf16') else torch.float32)
        # weight-gradient slab workspaces: one region per layer, so that the per-split partial results of every layer
        # survive until the ONE batched reduce launch at the end of the backward pass (aide_wgrad_reduce_flush)
        off = 0
        for st in self.steps:
            if st['kind'] in ('conv', 'convT'):
                st['wg_off'] = off
                off += (st['wg_bytes'] // 4 + 63) // 64 * 64
        self.wg_ws = torch.empty(max(off, 1), **f32)
        for st in self.steps:
            if st['kind'] in ('conv', 'convT'):
                st['wg_ws'] = self.wg_ws[st['wg_off']:st['wg_off'] + max(st['wg_bytes'] // 4, 1)]
        self.head_ws = torch.empty(max(self._max_wg // 4, 1), **f32)
        self.wg_queue = ops.new_wgrad_queue()     # pending slab reduces of a backward pass (batched launches)
        sa = [st for st in self.steps if st['kind'] == 'sa']
        if sa:                      # small-channel gradient ping-pong buffers + the gate-backward workspace
            big = max(st['t1'].numel() for st in sa)
            self.sa_da, self.sa_db = torch.empty(big, **f32), torch.empty(big, **f32)
            self.sa_ws = torch.empty(max(st['gate'].numel() for st in sa) * 2 + 8, **f32)
        if self.side is None:
            self.side = _side_stream(self.dev)
        self.ev_fork, self.ev_join = ops.new_event(), ops.new_event()
        for st in self.steps:
            if st['kind'] in ('conv', 'convT'):
                st['ev'] = ops.new_event()     # main -> side fork of this layer's weight gradient
        cover = {id(t): _Cover() for t in self.g.roots}
        for st in reversed(self.steps):
            src = st['src']
            if src.root.is_input:
                st['src_grad'] = None
                continue
            acc, gaps = cover[id(src.root)].write(src.c0, src.c0 + src.C)
            st['src_grad'] = dict(accumulate=acc, gaps=[src.root.slice(a, b - a) for a, b in gaps])
            if st['kind'] == 'convT' and acc:
                raise NotImplementedError('ConvTranspose input with several consumers')
        # A split-K data gradient whose ONLY reader is the BatchNorm backward of the conv right before it (conv1 -> conv2 of
        # a block) stays in its slabs: that kernel sums them (ops.bn_relu_bwd_slabs) -- one launch and one pass less per pair.
        for i, st in enumerate(self.steps):
            st['fold_dgrad'] = False
            sg = st.get('src_grad')
            if i == 0 or st['kind'] != 'conv' or sg is None or sg['accumulate'] or sg['gaps']:
                continue
            prod = self.steps[i - 1]
            src, dst = st['src'], prod.get('dst')
            if prod['kind'] != 'conv' or dst is None or dst.root is not src.root or (dst.c0, dst.C) != (src.c0, src.C):
                continue
            if sum(1 for o in self.steps if o.get('src') is not None and o['src'].root is src.root
                   and o['src'].c0 < src.c0 + src.C and src.c0 < o['src'].c0 + o['src'].C) != 1:
                continue                                   # another reader of (part of) the tensor: its gradient accumulates
            if st['wino_d'] not in (2, 4) or (st['plan_d'] >> 8) <= 1 or prod.get('dz_bf16') or st.get('dz_bf16'):
                continue
            if self.grad[id(src.root)].dtype != torch.float32 or prod['z'].dtype != torch.float32:
                continue
            pn, pc, ph, pw = prod['z'].shape
            if not lib.aide_bn_one_pass(pn, pc, ph, pw) or (ph * pw) % 4:
                continue
            st['fold_dgrad'] = True
        # A max-pooling whose source is exactly the output of one or several conv + BN + ReLU layers (the level's skip tensor:
        # both encoders' second convolutions in the FuseUNet) and whose gradient ACCUMULATES into a skip gradient the decoder has
        # written: its backward pass disappears -- every such layer's BatchNorm backward routes the pooled gradient to the window
        # arg-max itself (the activation recomputed from z, bit-identical) while it reads dA (ops.bn_relu_bwd_pool).
        for st in self.steps:
            st.pop('pool_fuse', None)
            st.pop('bwd_fused', None)
            st.pop('head_fuse', None)
            st.pop('dgrad_fused', None)
        if self.cfg.fuse_pool_bwd:
            for pst in self.steps:
                sg = pst.get('src_grad')
                if pst['kind'] != 'pool' or sg is None or not sg['accumulate'] or sg['gaps']:
                    continue
                src, dst = pst['src'], pst['dst']
                if self.grad[id(src.root)].dtype != torch.float32 or self.grad[id(dst.root)].dtype != torch.float32:
                    continue
                prods = [o for o in self.steps if o['kind'] == 'conv' and o['dst'].root is src.root
                         and src.c0 <= o['dst'].c0 and o['dst'].c0 + o['dst'].C <= src.c0 + src.C]
                cover = sorted((o['dst'].c0, o['dst'].c0 + o['dst'].C) for o in prods)
                if not cover or cover[0][0] != src.c0 or cover[-1][1] != src.c0 + src.C or \
                        any(a[1] != b[0] for a, b in zip(cover, cover[1:])):
                    continue                                   # the producers do not tile the pooled channels exactly
                ok = True
                for o in prods:
                    zn, zc, zh, zw = o['z'].shape
                    if o['z'].dtype != torch.float32 or o.get('dz_bf16') or not lib.aide_bn_relu_bwd_pool_supported(zn, zc, zh, zw):
                        ok = False
                if not ok:
                    continue
                for o in prods:
                    o['pool_fuse'] = dst.slice(o['dst'].c0 - src.c0, o['dst'].C)
                pst['bwd_fused'] = True
        # The 1x1 head's data gradient (a write and a read of the widest feature map) is formed inside the BatchNorm backward of the
        # layer under the head when that layer's activation has no other reader (ops.bn_relu_bwd_head).
        if self.cfg.fuse_head_bwd:
            for hst in self.steps:
                sg = hst.get('src_grad')
                if hst['kind'] != 'head' or sg is None or sg['accumulate'] or sg['gaps']:
                    continue
                src = hst['src']
                readers = [o for o in self.steps if o.get('src') is not None and o['src'].root is src.root
                           and o['src'].c0 < src.c0 + src.C and src.c0 < o['src'].c0 + o['src'].C]
                prods = [o for o in self.steps if o['kind'] == 'conv' and o['dst'].root is src.root
                         and (o['dst'].c0, o['dst'].C) == (src.c0, src.C)]
                if len(readers) != 1 or len(prods) != 1:
                    continue
                o = prods[0]
                zn, zc, zh, zw = o['z'].shape
                if o['z'].dtype != torch.float32 or o.get('dz_bf16') or o.get('pool_fuse') is not None or \
                        not lib.aide_bn_one_pass(zn, zc, zh, zw) or (zh * zw) % 4:
                    continue
                o['head_fuse'] = hst
                hst['dgrad_fused'] = True
        self._bwd_ready = True

    # ------------------------------------------------------------------ forward
    def _pack_filters(self):
        """Refresh the packed forward/dgrad filters when any master weight changed (tensor._version for
        torch optimizers, PARAM_EPOCH for the fused Adam).  Two launches: the first few (tiny, stage-1)
        filters on the main stream, all the others on the side stream so that the 0.25 ms re-layout runs
        under the first convolutions; returns the index of the first conv that must wait for it.
        Packs another plan of the same engine has already refreshed for these weights (shared buffers) are skipped."""
        convs = self._convs
        if convs is None:
            convs = self._convs = [st for st in self.steps if st['kind'] == 'conv']
            self._conv_wslots = [st['conv']._parameters for st in convs]          # (every forward: no module __getattr__)
        ws = [d['weight'] for d in self._conv_wslots]
        key = (PARAM_EPOCH[0],) + tuple((w.data_ptr(), w._version) for w in ws)
        if key == self._pack_key:
            return None
        ptrs = tuple(k[0] for k in key[1:])
        fresh_map = self._shared['fresh']
        mine = self._pack_ids
        if mine is None:
            mine = self._pack_ids = tuple(id(st[k]) for st in convs for k in ('wf', 'wd', 'uf', 'ud') if st[k] is not None)
        fresh = frozenset(i for i in mine if fresh_map.get(i) == key)
        cached = self._pack_tabs.get((ptrs, fresh))
        if cached is None:
            split = min(4, len(convs))

            def tables(group, fwd=True, dgrad=True):
                """pack tables of a group of convs; fwd / dgrad select which direction's packs they write"""
                def pick(st, k, on):
                    t = st[k] if on else None
                    return None if (t is None or id(t) in fresh) else t
                F = lambda st, k: pick(st, k, fwd)
                D = lambda st, k: pick(st, k, dgrad)
                direct = [(st['conv'].weight, F(st, 'wf'), D(st, 'wd')) for st in group
                          if F(st, 'wf') is not None or D(st, 'wd') is not None]
                # a conv may use different modes forward and backward: each table gets only its own packs
                def by_mode(mode):
                    out = []
                    for st in group:
                        uf = F(st, 'uf') if st['wino_f'] == mode else None
                        ud = D(st, 'ud') if st['wino_d'] == mode else None
                        if uf is not None or ud is not None:
                            out.append((st['conv'].weight, uf, ud))
                    return out
                wino, wino4, b16 = by_mode(2), by_mode(4), by_mode(BF16)
                tabs = (ops.pack_table(direct, self.dev) if direct else None,
                        ops.wino_pack_table(wino, self.dev) if wino else None,
                        ops.wino4_pack_table(wino4, self.dev) if wino4 else None,
                        ops.bf16_pack_table(b16, self.dev) if b16 else None)
                return tabs if any(t is not None for t in tabs) else None
            rest = convs[split:]
            # (the dgrad-direction packs launched later, under the decoder forward, measured +-0 twice: one launch.  Round 6: the
            # whole re-layout of the level >= 2 filters started behind the first level instead of beside it -- the stems and the
            # 33 MB BatchNorm passes of level 0 run 3-8x slower next to it than alone -- 633.8 -> 629.1 images/s (C2), 398.6 ->
            # 397.1 (C4): the contention only moves to level 1.  Not kept.)
            rest_tabs = tables(rest, True, True) if rest else None
            cached = (tables(convs[:split]), rest_tabs, convs[split] if rest_tabs is not None else None)
            if len(self._pack_tabs) > 8:
                self._pack_tabs.clear()
            self._pack_tabs[(ptrs, fresh)] = cached
        first, rest, gate = cached
        self._gate_conv = gate

        def launch(tabs):
            d, wn, w4, b16 = tabs
            if b16 is not None:
                ops.check(lib.aide_conv3x3_bf16_pack_multi(ops.ptr(b16[0]), b16[1], b16[2], ops.stream_ptr()),
                          'conv3x3_bf16_pack_multi')
            if w4 is not None:
                ops.check(lib.aide_conv3x3_wino4_pack_multi(ops.ptr(w4[0]), w4[1], w4[2], ops.stream_ptr()),
                          'conv3x3_wino4_pack_multi')
            if d is not None:
                ops.check(lib.aide_conv3x3_pack_weights_multi(ops.ptr(d[0]), d[1], d[2], ops.stream_ptr()),
                          'conv3x3_pack_weights_multi')
            if wn is not None:
                ops.check(lib.aide_conv3x3_wino_pack_multi(ops.ptr(wn[0]), wn[1], wn[2], ops.stream_ptr()),
                          'conv3x3_wino_pack_multi')
        self._launch_pack = launch
        if fresh:
            # packs taken over from another plan: whatever stream(s) wrote them must be done before this forward reads them
            # (the same main stream in every flow of this package; a plan driven from another stream pays two waits)
            cur = torch.cuda.current_stream()
            for s_ in self._shared['streams'] or ():
                if s_ is not None and s_ != cur:
                    cur.wait_stream(s_)
        if first is not None:
            launch(first)
        self._pack_key = key
        for i in mine:
            fresh_map[i] = key
        if rest is None:
            self._shared['streams'] = (torch.cuda.current_stream(), None)
            return None
        if self.side_fwd is None:
            pref = _preferred(self.dev)
            self.side_fwd = (self.side if self.side is not None else
                             pref['lane'] if (pref is not None and pref.get('lane') is not None) else
                             torch.cuda.Stream(device=self.dev))
            self._side_fwd_ptr = ctypes.c_void_p(self.side_fwd.cuda_stream)
            self.ev_pack_fork = ops.new_event()
        ops.order(self.ev_pack_fork, ops.stream_ptr(), self._side_fwd_ptr)
        with ops.use_stream(self._side_fwd_ptr):
            launch(rest)
        self._shared['streams'] = (torch.cuda.current_stream(), self.side_fwd)
        return gate

    def _fingerprint(self):
        """addresses of every parameter and buffer the launch sequence bakes in (a tape is only valid for these)"""
        # ... and everything else a recorded sequence depends on: the runtime schedule switches and the BatchNorm scalars
        fp = [self.cfg.snapshot(), FLUSH_EVERY, self.overlap]
        # (this runs every forward: the tensors are looked up through the modules' own _parameters / _buffers dicts -- a
        # replaced parameter or buffer is seen -- without walking the module tree)
        slots = self._fp_slots
        if slots is None:
            slots, bns = [], []
            for st in self.steps:
                if st.get('bn') is not None:
                    bns.append(st['bn'])
                for key in ('conv', 'bn', 'mod'):
                    m = st.get(key)
                    if m is None:
                        continue
                    for sub in (m.modules() if key == 'mod' else (m,)):
                        slots += [(sub._parameters, k) for k, v in sub._parameters.items() if v is not None]
                        slots += [(sub._buffers, k) for k, v in sub._buffers.items() if v is not None]
            self._fp_slots, self._fp_bns = slots, bns
        for bn in self._fp_bns:
            fp.append(bn.eps)
            fp.append(bn.momentum)
        fp += [d[k].data_ptr() for d, k in slots]
        return tuple(fp)

    def _tapeable(self):
        return self.cfg.replay and self.profiler is None and self.trace is None

    def _coef_fresh(self):
        """eval mode: are the cached BatchNorm coefficients (scale, shift, folded bias per layer) those of the current
        parameters and running statistics?  Marks them fresh for the forward that is about to (re)compute them."""
        if self.training:
            STATS_EPOCH[0] += 1
            return True
        tens = self._coef_tensors
        if tens is None:
            tens = []
            for st in self.steps:
                if st['kind'] in ('conv', 'convT'):
                    bn = st['bn']
                    tens += [st['conv'].bias, bn.weight, bn.bias, bn.running_mean, bn.running_var]
            tens = self._coef_tensors = [t for t in tens if t is not None]
            self._coef_bns = [st['bn'] for st in self.steps if st['kind'] in ('conv', 'convT')]
        self._coef_eps = [bn.eps for bn in self._coef_bns]
        key = (PARAM_EPOCH[0], STATS_EPOCH[0]) + tuple(self._coef_eps) + tuple((t.data_ptr(), t._version) for t in tens)
        self._coef_next = key            # becomes _coef_key once the fold launches of this forward have been issued
        return key == self._coef_key

    def forward(self, inputs, out):
        self.serial += 1                 # the saved activations of this plan now belong to THIS forward (any kind)
        gate = self._pack_filters()
        self._coef_ok = self._coef_fresh()
        self._coef_key = None            # (a forward that dies between here and its fold launches leaves nothing marked fresh)
        self._forward_any(inputs, out, gate)
        if not self.training:
            self._coef_key = self._coef_next
        return out

    def _forward_any(self, inputs, out, gate):
        if not self._tapeable():
            self._tape_f = self._tape_b = None
            return self._forward_impl(inputs, out, gate, None)
        self._fp = self._fingerprint()
        dyn = list(inputs) + [out]
        key = (torch.cuda.current_stream().cuda_stream, self._fp, tuple((tuple(t.shape), t.stride()) for t in dyn))
        tp = self._tape_f
        if tp is not None and tp.key == key:
            tp.replay(dyn, skip_tags=(() if gate is not None else ('gate',)) + (('aide_bn_eval_fold',) if self._coef_ok else ()))
            return out
        tp = Tape(key)
        with tp:
            self._forward_impl(inputs, out, gate, tp)
        self._tape_f = tp.finish(dyn)
        return out

    def _forward_impl(self, inputs, out, gate, tape):
        import ctypes
        n = self.N
        if not self.training and (tape is not None or not self._coef_ok):
            # eval-mode coefficients of every BatchNorm, once per change of the parameters / running statistics (a replayed
            # tape skips these entries while they are fresh)
            for st in self.steps:
                if st['kind'] in ('conv', 'convT'):
                    ops.bn_eval_fold(st['bn'], st['conv'].bias, st['scale'], st['shift'], st['fbias'])
        # Two lanes: the ops a graph marks lane = 1 (the second encoder of a two-modality net: an independent conv -> BN ->
        # conv -> BN chain per level) run on a second stream, beside the lane-0 chain of the same level -- one chain's
        # HBM-bound BatchNorm and launch boundaries under the other's convolutions.  Fork: lane 1 waits for everything
        # enqueued on the main stream so far; join: before the first main-stream op that touches channels lane 1 wrote.
        dual = self.lane_b is not None and self.cfg.dual_fwd and self.profiler is None and self.trace is None
        mp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        bp = ctypes.c_void_p(self.lane_b.cuda_stream) if dual else None
        # Free-running lane 1 (config.free_lane): a pooling op whose source holds lane-0 channels [0, c) and lane-1 channels [c, C)
        # runs as two launches, each on the stream that produced its channels.  Lane 1 then never waits for lane 0 after the
        # first fork (its next level reads only what it pooled itself), and lane 0 waits for lane 1 exactly where it reads
        # lane 1's pooled channels -- on an event recorded right behind that pooling launch, not on lane 1's tail.  (With one
        # pooling launch per level on the main stream each level cost a join + a fork: ~35 us of idle time per lane and level.)
        free = dual and self.cfg.free_lane
        on_b, pend = False, []
        forked = False        # free: lane 1 has been ordered behind the main stream (since the last event that needs it again)
        touched = []          # free: ranges lane-0 ops read or wrote since the last fork
        b_reads = []          # free: ranges lane-1 ops read since the last join
        pooled = []           # free: (root, c0, C, event) of lane-1 pooling halves the main stream has not waited for yet

        def hit(ranges, t):
            return any(r[0] is t.root and r[1] < t.c0 + t.C and t.c0 < r[1] + r[2] for r in ranges)

        def overlaps(t):
            return hit(pend, t)

        def fork_if_needed(srcs, dsts):
            nonlocal forked
            if not forked or any(hit(touched, t) for t in srcs + dsts):
                ops.order(self.ev_lane_fork, mp, bp)
                forked = True
                del touched[:]

        def main_deps(srcs, dsts):
            """order the main stream behind whatever lane 1 work the op depends on"""
            if any(hit(pend, t) for t in srcs + dsts) or any(hit(b_reads, t) for t in dsts):
                ops.order(self.ev_lane_join, bp, mp)
                del pend[:], b_reads[:], pooled[:]
            for e in [e for e in pooled if any(hit([e], t) for t in srcs + dsts)]:
                ops.wait(mp, e[3])
                pooled.remove(e)
        for st in self.steps:
            kind = st['kind']
            if self.trace is not None:
                self.trace('f', st)
            lane = st.get('lane', 0) if dual else 0
            if kind == 'conv' and st is self._gate_conv:      # (when) the remaining filters were re-packed on the side stream
                wait = lambda: torch.cuda.current_stream().wait_stream(self.side_fwd)
                if tape is not None:
                    tape.py(wait, 'gate')
                if gate is not None:
                    wait()
                on_b = forked = False         # a gate conv on lane 1 re-forks: its lane must see the wait as well
            if free:
                srcs = [st['src']]
                dsts = [st['dst']] if st.get('dst') is not None else []
                c = st.get('lane_split') if kind == 'pool' else None
                if c and forked:
                    src, dst = st['src'], st['dst']
                    sa, da, sb, db = src.slice(0, c), dst.slice(0, c), src.slice(c, src.C - c), dst.slice(c, dst.C - c)
                    fork_if_needed([sb], [db])
                    fused = bool(st.get('fwd_fused'))      # both halves were written by their producers' BatchNorm kernels
                    if not fused:
                        with ops.use_stream(bp):
                            ops.maxpool2x2_fwd(self.view(sb, inputs), self.view(db))
                    ops.record(st['ev_b'], bp)
                    pooled.append((db.root, db.c0, db.C, st['ev_b']))
                    b_reads.append((sb.root, sb.c0, sb.C))
                    main_deps([sa], [da])
                    if not fused:
                        ops.maxpool2x2_fwd(self.view(sa, inputs), self.view(da))
                    touched.extend((t.root, t.c0, t.C) for t in (sa, da))
                elif lane:
                    fork_if_needed(srcs, dsts)
                    pend.extend((t.root, t.c0, t.C) for t in dsts)
                    b_reads.extend((t.root, t.c0, t.C) for t in srcs)
                    with ops.use_stream(bp):
                        self._forward_op(st, inputs, out, self.bn_ws_b, self.sk_ws_b)
                else:
                    main_deps(srcs, dsts)
                    touched.extend((t.root, t.c0, t.C) for t in srcs + dsts)
                    self._forward_op(st, inputs, out, self.bn_ws, self.sk_ws)
                continue
            if lane and not on_b:
                ops.order(self.ev_lane_fork, mp, bp)
            on_b = bool(lane)
            if not lane and pend and (overlaps(st['src']) or (st.get('dst') is not None and overlaps(st['dst']))):
                ops.order(self.ev_lane_join, bp, mp)
                del pend[:]
            if lane:
                pend.append((st['dst'].root, st['dst'].c0, st['dst'].C))
                with ops.use_stream(bp):
                    self._forward_op(st, inputs, out, self.bn_ws_b, self.sk_ws_b)
            else:
                self._forward_op(st, inputs, out, self.bn_ws, self.sk_ws)
        if pend or pooled:
            ops.order(self.ev_lane_join, bp, mp)
        return out

    def _forward_op(self, st, inputs, out, bn_ws, sk_ws):
        kind = st['kind']
        if True:
            if kind == 'conv':
                conv, bn = st['conv'], st['bn']
                x = self.view(st['src'], inputs)
                prof = self.profiler
                if prof is not None:
                    prof.begin(FWD_TAG[st['wino_f']], st['flops'], st['flops'] * EXEC_FRAC[st['wino_f']])
                # training forward of a split-K layer: the conv leaves its slabs (accumulate = 2) and the BatchNorm that
                # follows sums them itself -- no split-reduce launch, one pass over z less
                slabs = self.training and (st['plan_f'] >> 8) > 1 and \
                    (st['z'].shape[2] * st['z'].shape[3]) % 4 == 0          # (the slab loader reads 16 bytes)
                acc = 2 if slabs else 0
                if st.get('fold'):                 # eval: y = relu(acc * scale + folded bias) straight into the activation
                    if st['wino_f'] == 4:
                        ops.conv3x3_wino4(x, st['uf'], st['fbias'], self.view(st['dst']), accumulate=0,
                                          splitk=st['plan_f'] >> 8, ws=sk_ws, epi_scale=st['scale'], epi_relu=True)
                    else:                          # direct kernel (the 32-channel first level, the stems), non-split
                        ops.conv3x3_igemm(x, st['wf'], st['fbias'], self.view(st['dst']), accumulate=0, plan=st['plan_f'],
                                          ws=sk_ws, epi_scale=st['scale'], epi_relu=True)
                    if prof is not None:
                        prof.end()
                    return
                lazy = st.get('lazy_to')
                in_tab = st.get('in_tab')               # this conv applies the BatchNorm + ReLU of its input's producer(s)
                if st['wino_f'] == BF16:
                    ops.conv3x3_bf16(x, st['uf'], conv.bias, st['z'], accumulate=acc, splitk=st['plan_f'] >> 8, ws=sk_ws)
                elif st['wino_f'] == 4:
                    # (st['stats']: this launch also writes the BatchNorm statistics partials of its output)
                    ops.conv3x3_wino4(x, st['uf'], conv.bias, self.view(st['dst']) if lazy is not None else st['z'],
                                      accumulate=acc, splitk=st['plan_f'] >> 8, ws=sk_ws, stats=st['stats'],
                                      in_tab=in_tab, in_group_images=(self.N // self.groups) if in_tab is not None else 0)
                elif st['wino_f']:
                    ops.conv3x3_wino(x, st['uf'], conv.bias, st['z'], accumulate=acc, splitk=st['plan_f'] >> 8, ws=sk_ws)
                else:
                    ops.conv3x3_igemm(x, st['wf'], conv.bias, st['z'], accumulate=acc, plan=st['plan_f'], ws=sk_ws)
                if prof is not None:
                    prof.end()
                if st.get('head_lazy'):            # statistics -> (scale, shift); the head applies them while it reads z
                    zn, zc, zh, zw = st['z'].shape
                    ops.bn_finalize_groups(zn, 1, zc, zh, zw, bn, st['stats'], st['stats_parts'], st['stats_parts'], conv.bias,
                                           st['mean'], st['rstd'], st['scale'], st['shift'], st['head_tab'], 0)
                elif lazy is not None:               # statistics -> the reader's (scale, shift) table; no pass over the tensor
                    zz = self.view(st['dst'])
                    ops.bn_finalize_groups(self.N // self.groups, self.groups, zz.shape[1], zz.shape[2], zz.shape[3], bn,
                                           st['stats'], st['stats_parts'] // self.groups, st['stats_parts'], conv.bias,
                                           st['mean'], st['rstd'], st['scale'], st['shift'], lazy['in_tab'], st['tab_c0'])
                else:
                    self._bn_apply(st, bn, conv.bias if slabs else None, (st['plan_f'] >> 8) if slabs else 0, bn_ws, sk_ws)
            elif kind == 'convT':
                conv, bn = st['conv'], st['bn']
                ops.convT2x2_fwd(self.view(st['src'], inputs), conv.weight, conv.bias, st['z'])
                self._bn_apply(st, bn, None, 0, bn_ws, sk_ws)
            elif kind == 'pool':
                if not (st.get('fwd_fused') and self.training):
                    ops.maxpool2x2_fwd(self.view(st['src'], inputs), self.view(st['dst']))
            elif kind == 'up':
                ops.upsample2x_fwd(self.view(st['src'], inputs), self.view(st['dst']))
            elif kind == 'head':
                conv = st['conv']
                lp = st.get('lazy_prod')
                if lp is not None:
                    ops.head1x1_fwd_bn(lp['z'], lp['scale'], lp['shift'], conv.weight.view(conv.out_channels, -1), conv.bias, out)
                else:
                    ops.head1x1_fwd(self.view(st['src'], inputs), conv.weight.view(conv.out_channels, -1),
                                    conv.bias, out)
            elif kind == 'sa':
                m = st['mod']
                y = self.view(st['src'], inputs)
                dil = m.conv2.dilation[0]
                ops.pwconv_fwd(y, m.conv1.weight, m.conv1.bias, st['t1'])
                ops.dconv_small(st['t1'], m.conv2.weight, m.conv2.bias, st['t2'], dil)
                ops.dconv_small(st['t2'], m.conv3.weight, m.conv3.bias, st['t3'], dil)
                ops.pwconv_fwd(st['t3'], m.conv4.weight, m.conv4.bias, st['t4'])
                if self.training and self.groups > 1:
                    mg = self.N // self.groups
                    for gi in range(self.groups):
                        ops.sa_gate_fwd(st['t4'][gi * mg:(gi + 1) * mg], m.bn, True, st['stat'],
                                        st['gate'][gi * mg:(gi + 1) * mg])
                else:
                    ops.sa_gate_fwd(st['t4'], m.bn, self.training, st['stat'], st['gate'])
                ops.sa_mul(st['gate'], y, self.view(st['dst']))

    def _bn_apply(self, st, bn, slab_bias=None, splitk=0, bn_ws=None, sk_ws=None):
        """BatchNorm(+ReLU) of one conv output.  splitk > 0: z is still in the split-K slabs of self.sk_ws
        ([splitk][N][C][H][W]) -- the BatchNorm kernels sum them, add `slab_bias` and write z themselves."""
        z, a = st['z'], self.view(st['dst'])
        bn_ws = self.bn_ws if bn_ws is None else bn_ws
        sk_ws = self.sk_ws if sk_ws is None else sk_ws
        ngroups = self.groups if (self.training and self.groups > 1) else 1
        m = self.N // ngroups
        if self.training and st.get('pool_out') is not None and splitk == 0:
            # (its activation is max-pooled: the pooled tensor is written here as well, no pooling pass)
            ops.bn_train_fwd_pool(z, a, self.view(st['pool_out']), ngroups, bn, st['mean'], st['rstd'], st['scale'], st['shift'], bn_ws)
        elif self.training and ngroups > 1 and self.cfg.grouped_bn:
            # the stacked augmentation pass: every group's statistics / running-statistics update in ONE launch sequence
            # (one launch sequence per group cost 3 x 2 x 32 small launches per co-teaching step)
            if st.get('stats') is not None:
                ops.bn_train_fwd_groups(z, a, ngroups, bn, st['mean'], st['rstd'], st['scale'], st['shift'], bn_ws,
                                        parts=st['stats'], nparts=st['stats_parts'] // ngroups,
                                        parts_stride=st['stats_parts'], conv_bias=st['conv'].bias)
            elif splitk > 0:
                import ctypes
                ops.bn_train_fwd_groups(z, a, ngroups, bn, st['mean'], st['rstd'], st['scale'], st['shift'], bn_ws,
                                        slabs=ctypes.c_void_p(sk_ws.data_ptr()), splitk=splitk, split_stride=z.numel(),
                                        slab_bias=slab_bias)
            else:
                ops.bn_train_fwd_groups(z, a, ngroups, bn, st['mean'], st['rstd'], st['scale'], st['shift'], bn_ws)
        elif self.training:
            stride = z.numel()                              # elements of one slab
            per_img = stride // self.N
            for gi in range(ngroups):
                zg, ag = (z, a) if ngroups == 1 else (z[gi * m:(gi + 1) * m], a[gi * m:(gi + 1) * m])
                if st.get('stats') is not None:
                    # (a group of a stacked batch owns a contiguous run of the per-image entries of every channel)
                    gparts = st['stats_parts'] // ngroups
                    ops.bn_train_fwd_parts(zg, ag, st['stats'], gparts, st['conv'].bias, bn.weight, bn.bias, bn.eps,
                                           bn.momentum, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                                           st['mean'], st['rstd'], st['scale'], st['shift'], True,
                                           first=gi * gparts, stride=st['stats_parts'])
                elif splitk > 0:
                    import ctypes
                    sl = ctypes.c_void_p(sk_ws.data_ptr() + 4 * gi * m * per_img)
                    ops.bn_train_fwd_slabs(sl, splitk, stride, slab_bias, zg, ag, bn.weight, bn.bias, bn.eps, bn.momentum,
                                           bn.running_mean, bn.running_var, bn.num_batches_tracked, st['mean'],
                                           st['rstd'], st['scale'], st['shift'], bn_ws, True)
                else:
                    ops.bn_train_fwd(zg, ag, bn.weight, bn.bias, bn.eps, bn.momentum, bn.running_mean, bn.running_var,
                                     bn.num_batches_tracked, st['mean'], st['rstd'], st['scale'], st['shift'],
                                     bn_ws, True)
        else:                                # (scale / shift: computed at the head of the forward pass, _forward_impl)
            ops.bn_relu_apply(z, a, st['scale'], st['shift'], True)

    # ------------------------------------------------------------------ backward
    def backward(self, inputs, dlogits, flat, offsets, after_op=None):
        """flat: 1-D fp32 buffer receiving every parameter gradient at offsets[param index]."""
        if self.groups > 1:
            raise RuntimeError('aide_amd: a grouped forward is forward-only')
        if not self._bwd_ready:
            self._prepare_backward()

        def gslot(p):
            i = self.pindex[id(p)]
            return flat[offsets[i]:offsets[i] + p.numel()].view(p.shape)

        # Two streams: the main stream carries the dependent chain (BN backward -> dgrad -> pool /
        # up-sampling backward); every weight-gradient kernel (MFMA-bound, off the critical path) goes to
        # a side stream as soon as its dz exists, so the HBM-bound kernels of the next layer run in
        # its shadow instead of between two MFMA kernels.
        main = torch.cuda.current_stream()
        # (with a profiler attached the step runs on ONE stream: an event pair then brackets exactly one kernel's own time,
        # not its time under contention with the side stream -- bench.py instruments a single step for that reason)
        side = self.side if (self.overlap and self.profiler is None) else None
        if self._tapeable() and self._fp is not None and self._tape_f is not None:
            # re-issue the recorded launch sequence (aide_amd/tape.py); Python callbacks of a gradient all-reduce hook are
            # tape entries.  Keyed on everything the sequence bakes in.
            dyn = list(inputs) + [dlogits, flat]
            key = (main.cuda_stream, side.cuda_stream if side is not None else 0, self._fp, after_op,
                   tuple((tuple(t.shape), t.stride()) for t in dyn))
            tp = self._tape_b
            if tp is not None and tp.key == key:
                try:
                    tp.replay(dyn)
                except BaseException:
                    # a replay that stops between the recorded defer(1) and defer(0) must not leave the library collecting
                    # slab reduces for ever (nor keep descriptors of this step)
                    if self.wg_queue is not None:
                        self.wg_queue.discard()
                    self._tape_b = None
                    raise
                return
            tp = Tape(key)
            cb = None
            if after_op is not None:
                def cb(w_st):
                    tp.py(lambda: after_op(w_st))
                    after_op(w_st)
            with tp:
                self._backward_streams(inputs, dlogits, gslot, main, side, cb)
            self._tape_b = tp.finish(dyn)
            return
        self._backward_streams(inputs, dlogits, gslot, main, side, after_op)

    def _backward_streams(self, inputs, dlogits, gslot, main, side, after_op):
        # raw stream handles: every fork / join below is a C-ABI call (ops.order) and every side-stream launch takes the
        # stream explicitly (ops.use_stream) -- nothing here depends on torch's stream context, so the whole sequence can be
        # recorded and re-issued by a launch tape
        import ctypes
        mp = ctypes.c_void_p(main.cuda_stream)
        sp = ctypes.c_void_p(side.cuda_stream) if side is not None else None
        if side is not None:
            ops.order(self.ev_fork, mp, sp)
        # The slab reduces of the weight gradients run batched, one launch per FLUSH_EVERY layers: fewer latency-bound
        # launches on the weight-gradient stream, and only the last (small, shallow-encoder) batch sits after the last kernel
        # (bf16 mode: its weight-gradient slabs are several times larger and the batched reduce measured 1 % slower -> per
        # layer).  The queue of pending reduces is this plan's own object (ops.WgradQueue): the library keeps no state.
        queue = self.wg_queue if (self.profiler is None and self.precision != 'bf16') else None
        waiting = []                          # ops whose after_op callback waits for the flush of their weight gradient

        def flush():
            # (pending(): a host-state query decides WHERE the flushes go while recording; a replayed tape re-issues them)
            if queue.pending():
                queue.flush(sp if side is not None else mp)
            if after_op is not None:
                for w_st in waiting:          # their weight gradients are now enqueued in full
                    after_op(w_st)
            del waiting[:]

        hook = after_op
        if queue is not None:
            queue.discard()                   # (entries of a pass that died between its launches and its flush)

            def hook(w_st):
                waiting.append(w_st)
                if queue.pending() >= FLUSH_EVERY:
                    flush()
        self._wq_active = queue
        try:
            with ops.use_stream(mp):
                self._backward_ops(inputs, dlogits, gslot, mp, sp, hook)
            if queue is not None:
                flush()
        except BaseException:
            if queue is not None:             # the queued descriptors point into this step's arena and workspaces
                queue.discard()
            raise
        if side is not None:
            ops.order(self.ev_join, sp, mp)

    def _backward_ops(self, inputs, dlogits, gslot, main, side, after_op):
        """The backward launch sequence: the dependent chain on `main`, every weight gradient on `side`.  (The second
        encoder's chains on a stream of their own, as in the forward pass, measured +-0 beside the weight-gradient stream
        -- HISTORY §3c -- and is not built in.)"""
        fold = [0]                       # split count of the data gradient the NEXT BatchNorm backward reads from sk_ws
        for st in reversed(self.steps):
            if self.trace is not None:
                self.trace('b', st)
            self._backward_op(st, inputs, dlogits, gslot, main, side, self.bn_ws, self.sk_ws, fold)
            if after_op is not None:
                after_op(st)

    def _backward_op(self, st, inputs, dlogits, gslot, main, side, bn_ws, sk_ws, folded):
        """one op of the backward sequence on stream `main` (its lane's stream); folded: [split count] cell of the lane"""
        if True:
            kind = st['kind']
            sg = st.get('src_grad')
            if sg is not None:
                for gap in sg['gaps']:
                    ops.fill_zero(self.gview(gap))
            if kind == 'head':
                conv = st['conv']
                k = conv.out_channels
                assert sg is None or not sg['accumulate']
                fused = bool(st.get('dgrad_fused'))     # the data gradient is formed by the BatchNorm backward of the layer below
                dsrc = self.gview(st['src']) if (sg is not None and not fused) else None
                lp = st.get('lazy_prod')              # the activation under the head was never stored: recomputed from z

                def wgrad_only():
                    if lp is not None:
                        ops.head1x1_wgrad_bn(dlogits, lp['z'], lp['scale'], lp['shift'], gslot(conv.weight).view(k, -1),
                                             gslot(conv.bias), ws=self.head_ws)
                    else:
                        ops.head1x1_bwd(dlogits, self.view(st['src'], inputs), conv.weight.view(k, -1), None,
                                        gslot(conv.weight).view(k, -1), gslot(conv.bias), ws=self.head_ws)
                if side is not None and sg is not None:
                    # the head's weight gradient (one pass over the widest feature map) has no consumer until the
                    # optimizer: side stream, so that the dependent chain starts with the data gradient alone
                    with ops.use_stream(side):
                        wgrad_only()
                    if dsrc is not None:
                        ops.head1x1_bwd(dlogits, self.view(st['src'], inputs), conv.weight.view(k, -1), dsrc, None, None,
                                        ws=self.head_ws)
                elif lp is not None:
                    wgrad_only()
                    if dsrc is not None:
                        ops.head1x1_bwd(dlogits, lp['z'], conv.weight.view(k, -1), dsrc, None, None, ws=self.head_ws)
                else:
                    ops.head1x1_bwd(dlogits, self.view(st['src'], inputs), conv.weight.view(k, -1), dsrc,
                                    gslot(conv.weight).view(k, -1), gslot(conv.bias), ws=self.head_ws)
            elif kind in ('conv', 'convT'):
                conv, bn = st['conv'], st['bn']
                z = st['z']
                dz = st['dz']
                # a layer that hands its dz over to the weight-gradient stream right away: the event rides on the
                # BatchNorm backward's last dispatch (done=) and the other stream only waits for it -- no record packet
                # between this launch and the data-gradient convolution on this queue
                tail_ = self.cfg.tail_wgrad_main and st is self.steps[0] and sg is None and self.profiler is None
                done = st['ev'] if (side is not None and not tail_ and self.cfg.handover_on_kernel) else None
                if st.get('head_fuse') is not None and not folded[0]:
                    hconv = st['head_fuse']['conv']
                    ops.bn_relu_bwd_head(dlogits, hconv.weight.view(hconv.out_channels, -1), z, dz, st['mean'], st['rstd'], st['scale'],
                                         st['shift'], gslot(bn.weight), gslot(bn.bias), gslot(conv.bias), bn_ws, True, done=done)
                elif st.get('pool_fuse') is not None and not folded[0]:
                    # (its activation was max-pooled: the pooled gradient joins dA inside this kernel, no pooling backward pass)
                    ops.bn_relu_bwd_pool(self.gview(st['dst']), self.gview(st['pool_fuse']), z, dz, st['mean'], st['rstd'],
                                         st['scale'], st['shift'], gslot(bn.weight), gslot(bn.bias), gslot(conv.bias), bn_ws,
                                         True, done=done)
                elif folded[0]:                  # dA is still in the split-K slabs of the conv after this one
                    ops.bn_relu_bwd_slabs(sk_ws, folded[0], z, dz, st['mean'], st['rstd'], st['scale'], st['shift'],
                                          gslot(bn.weight), gslot(bn.bias), gslot(conv.bias), bn_ws, True, done=done)
                    folded[0] = 0
                else:
                    ops.bn_relu_bwd(self.gview(st['dst']), z, dz, st['mean'], st['rstd'], st['scale'],
                                    st['shift'], gslot(bn.weight), gslot(bn.bias), gslot(conv.bias),
                                    bn_ws, True, done=done)
                x = self.view(st['src'], inputs)
                if kind == 'conv':
                    prof = self.profiler
                    wq = self._wq_active
                    wfn = (ops.conv3x3_wgrad_bf16 if st['wino_w'] == BF16 else
                           (lambda d_, x_, w_, ws=None, queue=None: ops.conv3x3_wgrad_wino4(d_, x_, w_, ws=ws, target_wgs=256,
                                                                                            queue=queue))
                           if st.get('wg_target') else
                           ops.conv3x3_wgrad_wino4 if st['wino_w'] == 4 else
                           ops.conv3x3_wgrad_wino if st['wino_w'] else ops.conv3x3_wgrad)
                    wgrad = lambda d_, x_, w_, ws=None: wfn(d_, x_, w_, ws=ws, queue=wq)
                    # the last op of the pass, when it has no data gradient (a stem conv): the dependent chain ends with its
                    # BatchNorm backward, so its weight gradient runs on that stream beside whatever the weight-gradient
                    # stream still has queued instead of behind it
                    tail = self.cfg.tail_wgrad_main and st is self.steps[0] and sg is None and prof is None
                    if side is not None and not tail:
                        if done is not None:
                            ops.wait(side, done)
                        else:
                            ops.order(st['ev'], main, side)
                        with ops.use_stream(side):
                            if prof is not None:
                                prof.begin(WGRAD_TAG[st['wino_w']], st['flops'], st['flops'] * EXEC_FRAC[st['wino_w']])
                            wgrad(dz, x, gslot(conv.weight), ws=st['wg_ws'])
                            if prof is not None:
                                prof.end()
                    else:
                        if prof is not None:
                            prof.begin(WGRAD_TAG[st['wino_w']], st['flops'], st['flops'] * EXEC_FRAC[st['wino_w']])
                        if tail and side is not None:
                            # (tail) its slabs are reduced right behind it on this stream (queue = None): handing them to the
                            # final batched launch of the weight-gradient stream would put two stream hand-overs and that
                            # launch between this kernel and the optimizer (C2 +0.3 %, C4 +0.25 %, C5 +0.15 % same box)
                            wfn(dz, x, gslot(conv.weight), ws=st['wg_ws'], queue=None)
                        else:
                            wgrad(dz, x, gslot(conv.weight), ws=st['wg_ws'])
                        if prof is not None:
                            prof.end()
                    if sg is not None:
                        if prof is not None:
                            prof.begin(FWD_TAG[st['wino_d']], st['flops'], st['flops'] * EXEC_FRAC[st['wino_d']])
                        if st['wino_d'] == BF16:
                            ops.conv3x3_bf16(dz, st['ud'], None, self.gview(st['src']),
                                             accumulate=sg['accumulate'], splitk=st['plan_d'] >> 8, ws=sk_ws)
                        elif st['wino_d'] == 4:
                            ops.conv3x3_wino4(dz, st['ud'], None, self.gview(st['src']),
                                              accumulate=2 if st['fold_dgrad'] else sg['accumulate'],
                                              splitk=st['plan_d'] >> 8, ws=sk_ws)
                        elif st['wino_d']:
                            ops.conv3x3_wino(dz, st['ud'], None, self.gview(st['src']),
                                             accumulate=2 if st['fold_dgrad'] else sg['accumulate'],
                                             splitk=st['plan_d'] >> 8, ws=sk_ws)
                        else:
                            ops.conv3x3_igemm(dz, st['wd'], None, self.gview(st['src']),
                                              accumulate=sg['accumulate'], plan=st['plan_d'], ws=sk_ws)
                        if prof is not None:
                            prof.end()
                        if st['fold_dgrad']:
                            folded[0] = st['plan_d'] >> 8
                else:
                    # the last op of the pass, when it has no data gradient (a stem conv): the dependent chain ends with its
                    # BatchNorm backward, so its weight gradient runs on that stream beside whatever the weight-gradient
                    # stream still has queued instead of behind it
                    tail = self.cfg.tail_wgrad_main and st is self.steps[0] and sg is None and prof is None
                    if side is not None and not tail:
                        if done is not None:
                            ops.wait(side, done)
                        else:
                            ops.order(st['ev'], main, side)
                        with ops.use_stream(side):
                            ops.convT2x2_wgrad(x, dz, gslot(conv.weight), ws=st['wg_ws'])
                    else:
                        ops.convT2x2_wgrad(x, dz, gslot(conv.weight), ws=st['wg_ws'])
                    if sg is not None:
                        ops.convT2x2_dgrad(dz, conv.weight, self.gview(st['src']))
            elif kind == 'sa':
                m = st['mod']
                y, dout = self.view(st['src'], inputs), self.gview(st['dst'])
                dil = m.conv2.dilation[0]
                shp = st['t1'].shape
                da, db = self.sa_da[:st['t1'].numel()].view(shp), self.sa_db[:st['t1'].numel()].view(shp)
                mnum = st['gate'].numel()
                dt4 = self.sa_ws[mnum + 8:mnum + 8 + mnum].view(st['t4'].shape)
                ops.sa_gate_bwd(dout, y, st['gate'], st['t4'], st['stat'], m.bn.weight, gslot(m.bn.weight),
                                gslot(m.bn.bias), dt4, self.sa_ws[:mnum + 4])
                ops.pwconv_wgrad(dt4, st['t3'], gslot(m.conv4.weight), gslot(m.conv4.bias))
                ops.pwconv_dgrad(dt4, m.conv4.weight, da)                                  # da = d t3
                ops.dconv_small_wgrad(da, st['t2'], gslot(m.conv3.weight), gslot(m.conv3.bias), dil)
                ops.dconv_small(da, m.conv3.weight, None, db, dil, transposed=True)       # db = d t2
                ops.dconv_small_wgrad(db, st['t1'], gslot(m.conv2.weight), gslot(m.conv2.bias), dil)
                ops.dconv_small(db, m.conv2.weight, None, da, dil, transposed=True)       # da = d t1
                ops.pwconv_wgrad(da, y, gslot(m.conv1.weight), gslot(m.conv1.bias))
                # d y = gate * dout (the multiply) + conv1^T d t1, in one pass over the C channels
                ops.pwconv_dgrad(da, m.conv1.weight, self.gview(st['src']), gate=st['gate'], dout=dout,
                                 accumulate=sg['accumulate'])
            elif kind == 'pool':
                if sg is not None and not st.get('bwd_fused'):
                    ops.maxpool2x2_bwd(self.view(st['src'], inputs), self.gview(st['dst']),
                                       self.gview(st['src']), accumulate=sg['accumulate'])
            elif kind == 'up':
                if sg is not None:
                    ops.upsample2x_bwd(self.gview(st['dst']), self.gview(st['src']),
                                       accumulate=sg['accumulate'])




def _has_param_hooks(params):
    """tensor hooks on a parameter (register_hook, register_post_accumulate_grad_hook) only fire when the parameter is
    an autograd input of the node: the engine-assigned gradient path would bypass them silently"""
    for p in params:
        if p._backward_hooks or getattr(p, '_post_accumulate_grad_hooks', None):
            return True
    return False


import sys as _sys


def _max_refs(views):
    m = 0
    for v in views:
        m = max(m, _sys.getrefcount(v))
    return m


_REF_BASE = _max_refs([object()])               # what a list element nobody else holds shows inside that loop


_STORAGE_USES = getattr(torch._C, '_storage_Use_Count', None)


def _storage_uses(t):
    """tensors sharing t's storage (+1 for the temporary storage wrapper), or -1 where torch does not expose the count"""
    return _STORAGE_USES(t.untyped_storage()._cdata) if _STORAGE_USES is not None else -1


def _views_held_elsewhere(views, arena=None, base_uses=-1):
    """does anybody besides the engine still hold a gradient of the last pass across zero_grad() (a snapshot, manual
    accumulation, gradient statistics)?  Either one of the arena's view OBJECTS (Python reference counts), or any other
    tensor on the arena's storage -- `p.grad.detach()`, `.data`, `.view(-1)`, a slice (the storage's use count against what
    the engine's own tensors account for; torch builds without that counter see only the first kind: keep a .clone())."""
    if _max_refs(views) > _REF_BASE:
        return True
    return arena is not None and base_uses >= 0 and _storage_uses(arena) > base_uses


class _NetFunction(torch.autograd.Function):
    """forward/backward of the whole network as one autograd node.

    The parameters are NOT inputs of the node (130 autograd inputs / outputs and as many AccumulateGrad nodes cost more host
    time per step than the whole launch sequence): the node hangs on one anchor tensor of the engine and its backward
    assigns `p.grad` itself -- views of the engine's persistent flat gradient arena when no gradient exists yet (the
    state after `optimizer.zero_grad()`), an accumulation into the existing gradients otherwise (what AccumulateGrad
    does).  AIDE_DIRECT_GRADS=0 is the plain form (every parameter an autograd input) for torch.autograd.grad on
    parameters; a module whose parameters carry tensor hooks (register_hook / register_post_accumulate_grad_hook) takes it
    by itself.  Not visible from here, so still bypassed: reducers that hook the AccumulateGrad nodes of a WRAPPING module
    (torch DistributedDataParallel / FSDP) -- use aide_amd.distributed.attach() instead.

    Arena aliasing: `p.grad` after such a pass are views of ONE persistent buffer which the next pass overwrites in place;
    a view the caller still references at that point is left alone (the pass takes a new arena), as `p.grad = None`
    leaves a held tensor alone in torch.  The caller's stream is ordered behind the whole pass through the anchor's
    AccumulateGrad node (see the end of backward)."""

    @staticmethod
    def forward(ctx, engine, n_inputs, direct, *tensors):
        inputs = tensors[:n_inputs]
        plan = engine.plan_for(inputs)
        plan.profiler = engine.profiler
        k = engine.num_classes
        n, _, h, w = inputs[0].shape
        out = torch.empty(n, k, h, w, device=inputs[0].device, dtype=torch.float32)
        plan.forward(inputs, out)
        # The backward pass reads the network inputs again (stem weight gradients) -- on THIS stream and on the weight-gradient
        # stream, possibly after the caller has dropped them: a pipelined co-teaching step leaves network 2's backward on its own
        # stream past the end of the call, and the caching allocator would hand a dropped batch to the next main-stream allocation
        # while that stream still reads it (found in round 6: both stems' weight gradients of the pipelined form differed in 1 of 3
        # runs of tools/r6_flaky_c3.py).  record_stream defers the reuse behind what this stream has enqueued when the tensor dies.
        cur = torch.cuda.current_stream(inputs[0].device)
        for x in inputs:
            x.record_stream(cur)
        ctx.engine, ctx.plan, ctx.serial, ctx.inputs, ctx.direct = engine, plan, plan.serial, inputs, direct
        ctx.ntensors = len(tensors)
        return out

    @staticmethod
    def backward(ctx, dlogits):
        eng, plan = ctx.engine, ctx.plan
        if ctx.serial != plan.serial or eng.plans.get(plan.key) is not plan:
            raise RuntimeError('aide_amd: backward() after a newer forward() of the same module at the same input shape '
                               '— its activations were overwritten (the engine keeps one set per module and shape)')
        if not plan.training:
            raise RuntimeError('aide_amd: backward through an eval-mode forward is not supported')
        dlogits = dlogits.contiguous()
        dlogits.record_stream(torch.cuda.current_stream(dlogits.device))      # (as the inputs in forward: read on this pass' streams)
        params = eng.params
        mode = 0                         # 0: plain autograd outputs; 1: fresh gradients = arena views; 2: arena += ; 3: per parameter
        if ctx.direct:
            arena, views = eng.grad_arena(dlogits.device)
            have = [p.grad for p in params]
            if all(g is None for g in have):
                mode = 1
                if _views_held_elsewhere(views, arena, eng._arena_uses):
                    # torch semantics: `p.grad = None` leaves a gradient tensor the caller still holds intact.  The arena
                    # is about to be overwritten in place, so those tensors keep the OLD arena and this pass gets a new one.
                    arena, views = eng.grad_arena(dlogits.device, fresh=True)
            else:
                base = arena.data_ptr()
                if all(g is not None and g.data_ptr() == base + 4 * o and g.dtype == torch.float32
                       for g, o in zip(have, eng.offsets)):
                    mode = 2             # every gradient still lives in the arena (zero_grad(set_to_none=False), or a
                else:                    # second backward before the optimizer): one add over the whole arena
                    mode = 3
        flat = arena if mode == 1 else torch.empty(eng.flat_numel, device=dlogits.device, dtype=torch.float32)
        eng.side_stream = plan.side if (plan._bwd_ready and plan.overlap) else None
        if not plan._bwd_ready:
            plan._prepare_backward()
            eng.side_stream = plan.side if plan.overlap else None
        if eng.before_backward is not None:
            eng.before_backward(flat)
        plan.backward(ctx.inputs, dlogits, flat, eng.offsets, eng.after_backward_op)
        if eng.grad_hook is not None:
            eng.grad_hook(flat)
        if mode == 0:
            grads = tuple(flat[o:o + p.numel()].view(p.shape) for o, p in zip(eng.offsets, params))
            return (None, None, None) + (None,) * len(ctx.inputs) + grads
        if mode == 1:
            for p, v in zip(params, views):
                if p.requires_grad:
                    p.grad = v
        elif mode == 2:
            arena.add_(flat)
        else:
            for p, o, g in zip(params, eng.offsets, have):
                if not p.requires_grad:
                    continue
                v = flat[o:o + p.numel()].view(p.shape)
                if g is None:
                    p.grad = v
                else:
                    g.add_(v)
        # The anchor gets a (never read) gradient so that autograd runs its AccumulateGrad node: that node lives on the
        # stream the anchor was created on, autograd orders it behind this node's stream and -- at the end of the pass --
        # the caller's current stream behind it.  Without it a forward issued on a non-current stream (network 2 of the
        # co-teaching step) leaves `optimizer.step()` on the caller's stream racing this backward: the engine assigns the
        # parameter gradients itself, so autograd's leaf-stream synchronisation never saw them.
        # (a cached zero, not torch.empty: uninitialised bits may be a NaN pattern -- detect_anomaly would flag it -- and a second
        # backward before anchor.grad is cleared accumulates it)
        z = eng._anchor_zero
        if z is None or z.device != dlogits.device:
            z = eng._anchor_zero = torch.zeros(1, device=dlogits.device)
        return (None,) * (3 + ctx.ntensors - 1) + (z,)


class Engine(object):
    """Owned by a model (fuseunet / UNet); compiles and caches plans, runs forward/backward."""
    MAX_PLANS = 6

    def __init__(self, module, build_graph, num_classes):
        if not 1 <= int(num_classes) <= 8:       # the head kernels' template range (csrc/head_adam.hip MAXK)
            raise NotImplementedError('aide_amd: num_classes must be 1 .. 8, got %r' % (num_classes,))
        self.module, self.build_graph, self.num_classes = module, build_graph, num_classes
        self.plans = {}
        self._shared_packs = dict(buf={}, fresh={}, streams=None)      # packed filters shared by this engine's plans
        self.params = None
        self.grad_hook = None            # callable(flat_grad) e.g. DDP all-reduce of the whole arena
        self.after_backward_op = None    # callable(step) e.g. bucketed all-reduce overlap
        self.profiler = None             # object with begin(tag, flops) / end(): per-kernel HIP events
        self.before_backward = None      # callable(flat_grad) at the start of every backward
        self.side_stream = None          # stream carrying the weight-gradient kernels of the running backward
        self.graph = None
        self._precision = 'fp32'
        self.config = EngineConfig(self._config_changed)
        self.pending_stream = None       # a stream still carrying this module's last backward / optimizer step (set by a caller
                                         # that leaves them there: coteach_step(pipeline=True)); see _join_pending
        self._arena = self._views = self._anchor = self._anchor_zero = None
        self._arena_uses = -1
        self._pslots = None

    def _join_pending(self):
        """a forward issued on ANOTHER stream than the one that still carries this module's last backward pass / optimizer
        step (evaluation, a plain forward after a pipelined co-teaching loop) is ordered behind it; a forward on that same
        stream -- the next pipelined step -- needs no wait and keeps the overlap"""
        s = self.pending_stream
        if s is not None:
            cur = torch.cuda.current_stream(s.device)
            if cur != s:
                cur.wait_stream(s)
                self.pending_stream = None

    def _config_changed(self):
        """a switch of self.config was assigned: plans (with their launch tapes) and shared filter packs were built under the old
        setting -- drop them; the next forward compiles under the new one"""
        self.plans = {}
        self._shared_packs = dict(buf={}, fresh={}, streams=None)

    def grad_arena(self, device, fresh=False):
        """the persistent flat gradient arena of this module and the per-parameter views into it (offsets: _refresh_params).
        `p.grad` of a backward pass that found no gradients ARE these views: the next such pass overwrites them in place
        (fresh=True: a new arena -- the pass found one of the old views still referenced by the caller)."""
        a = self._arena
        if fresh or a is None or a.numel() != self.flat_numel or a.device != device:
            a = self._arena = torch.zeros(self.flat_numel, device=device, dtype=torch.float32)
            self._views = [a[o:o + p.numel()].view(p.shape) for o, p in zip(self.offsets, self.params)]
            self._arena_uses = _storage_uses(a)      # the arena and its views: anything above this is a caller's alias
        return a, self._views

    @property
    def precision(self):
        """'fp32' (default: exact-fp32 MFMA / Winograd kernels, the reference's arithmetic) or 'bf16' (BASELINE
        config 5: bf16 conv operands, fp32 accumulation, fp32 everything else)."""
        return self._precision

    @precision.setter
    def precision(self, value):
        if value not in PRECISIONS:
            raise ValueError('aide_amd: precision must be one of %s (got %r)' % (PRECISIONS, value))
        if value != self._precision:
            self._precision = value
            self.plans = {}
            self._shared_packs = dict(buf={}, fresh={}, streams=None)

    def _refresh_params(self):
        # every forward: the known parameters are checked where they live (their owner's _parameters dict) -- walking the
        # module tree costs ~0.4 ms for 130 parameters.  A replaced parameter fails the check and triggers the full walk.
        slots = self._pslots
        if slots is not None and self.params is not None and all(d.get(k) is p for d, k, p in slots):
            return
        params = list(self.module.parameters())
        self._pslots = [(m._parameters, k, p) for m in self.module.modules() for k, p in m._parameters.items()
                        if p is not None]
        if self.params is None or len(params) != len(self.params) or \
                any(a is not b for a, b in zip(params, self.params)):
            self.params = params
            self.graph = self.build_graph()
            # Gradient arena in BACKWARD-COMPLETION order (reverse graph order): contiguous ranges of it become complete
            # progressively, so the data-parallel buckets (distributed.py) go out while the rest of the backward still
            # runs and only the small shallow-encoder tail is reduced after the last kernel.  (In registration order the
            # two ~30 MB buckets holding the stem convolutions completed at the very end.)
            index = {id(p): i for i, p in enumerate(params)}
            order = []
            for op in reversed(self.graph.ops):
                for key in ('conv', 'bn', 'mod'):
                    m = op.get(key)
                    if m is not None:
                        order += [index[id(p)] for p in m.parameters()]
            seen = set(order)
            assert len(seen) == len(order), 'a parameter is owned by two graph ops'
            order += [i for i in range(len(params)) if i not in seen]
            self.offsets, off = [0] * len(params), 0
            for i in order:
                self.offsets[i] = off
                off += (params[i].numel() + 3) // 4 * 4  # 16-byte aligned slots
            self.flat_numel = off
            self.plans = {}
            self._shared_packs = dict(buf={}, fresh={}, streams=None)
            self._arena = self._views = None

    @staticmethod
    def invalidate_coefficients():
        """Call after editing BatchNorm running statistics or affine parameters through `.data` / raw pointers (no tensor
        version changes): the folded eval-mode coefficients of every plan are recomputed by its next forward."""
        STATS_EPOCH[0] += 1

    def plan_for(self, inputs, groups=1):
        self._refresh_params()
        x = inputs[0]
        n, _, h, w = x.shape
        if h % 16 or w % 16:
            raise RuntimeError('aide_amd: H and W must be multiples of 16 (got %dx%d)' % (h, w))
        key = (n, h, w, x.device.index, bool(self.module.training), self._precision, groups)
        plan = self.plans.get(key)
        if plan is None:
            # the kernels address one image of one operand with 32-bit byte offsets below 2^31 (buffer descriptors)
            for op in self.graph.ops:
                for t in (op.get('src'), op.get('dst')):
                    if t is not None and t.C * (h >> t.level) * (w >> t.level) * 4 >= 2 ** 31:
                        raise RuntimeError('aide_amd: %dx%d is too large: %d channels of %s exceed the 2 GiB per-image '
                                           'operand limit of the kernels' % (h, w, t.C, t.name))
            plan = Plan(self.graph, self.params, n, h, w, x.device, bool(self.module.training), self._precision,
                        groups, shared=self._shared_packs if self.config.shared_packs else None, cfg=self.config)
            plan.key = key
            self.plans[key] = plan
            # a plan owns full activation (+ gradient) buffers: ragged last batches / many input shapes must not pile
            # them up -- least recently used plans beyond MAX_PLANS are dropped (a backward of a dropped plan raises)
            while len(self.plans) > self.MAX_PLANS:
                self.plans.pop(next(iter(self.plans)))
        else:
            self.plans[key] = self.plans.pop(key)        # most recently used last
        return plan

    def run_groups(self, input_groups):
        """[net(*inputs).detach() for inputs in input_groups] in ONE pass: the groups are stacked along N, every
        convolution / pooling / up-sampling kernel runs once over all of them (4x the pixels per launch: the deep
        layers need no split-K and fill the chip), every BatchNorm normalises and updates its running statistics per
        group, in order -- exactly what the sequential forwards of the co-teaching loop's augmentation passes do
        (trainchaos_proposed_30cases1labeled.py:265-269).  Forward only (no autograd graph)."""
        self._join_pending()
        groups = len(input_groups)
        nin = len(input_groups[0])
        ins = []
        for k in range(nin):
            xs = [grp[k] for grp in input_groups]
            for x in xs:
                if not isinstance(x, torch.Tensor) or not x.is_cuda or x.dtype != torch.float32 or x.dim() != 4:
                    raise RuntimeError('aide_amd: inputs must be fp32 NCHW tensors on a HIP device')
                if x.shape != xs[0].shape:
                    raise RuntimeError('aide_amd: grouped forward needs equal input shapes')
            ins.append(torch.cat([x.detach() for x in xs], 0))
        with torch.no_grad():
            plan = self.plan_for(ins, groups)
            plan.profiler = self.profiler
            n, _, h, w = ins[0].shape
            out = torch.empty(n, self.num_classes, h, w, device=ins[0].device, dtype=torch.float32)
            plan.forward(ins, out)
        m = n // groups
        return [out[g * m:(g + 1) * m] for g in range(groups)]

    def run(self, *inputs):
        self._join_pending()
        ins = []
        for x in inputs:
            if not isinstance(x, torch.Tensor) or not x.is_cuda:
                raise RuntimeError('aide_amd models run on a HIP device only (input on %s); there is no '
                                   'CPU fallback — use the reference for CPU runs'
                                   % (x.device if isinstance(x, torch.Tensor) else type(x)))
            if x.dtype != torch.float32 or x.dim() != 4:
                raise RuntimeError('aide_amd: inputs must be fp32 NCHW tensors')
            ins.append(x.contiguous())
        self._refresh_params()
        for p in self.params:
            if not p.is_cuda:
                raise RuntimeError('aide_amd: module parameters are on %s; call .to(device) first' % p.device)
        if self.config.direct_grads and torch.is_grad_enabled() and any(p.requires_grad for p in self.params) and \
                not _has_param_hooks(self.params):
            if self._anchor is None or self._anchor.device != ins[0].device:
                self._anchor = torch.zeros(1, device=ins[0].device, requires_grad=True)
            self._anchor.grad = None         # (its AccumulateGrad then takes the returned tensor as it is: no kernel)
            return _NetFunction.apply(self, len(ins), True, *(tuple(ins) + (self._anchor,)))
        return _NetFunction.apply(self, len(ins), False, *(tuple(ins) + tuple(self.params)))
