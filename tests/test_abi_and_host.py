"""CPU: the C-ABI library loads and exports every symbol include/aide_hip.h declares (no compute
calls without a GPU), plus the pure host logic (planner heuristics, backward write analysis, graph
construction, DDP bucket scheduling, synthetic data contract, loud failure on CPU tensors)."""
import os
import subprocess

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def built():
    from aide_amd.build import build
    return build(verbose=False)


def test_header_symbols_exported(built):
    from aide_amd._lib import lib, parse_header
    protos = parse_header()
    assert len(protos) >= 35
    out = subprocess.run(['nm', '-D', '--defined-only', built], capture_output=True, text=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if l.strip()}
    missing = [n for n in protos if n not in exported]
    assert not missing, 'declared in include/aide_hip.h but not exported: %s' % missing
    extra = [n for n in exported if n.startswith('aide_') and n not in protos]
    assert not extra, 'exported but not declared in include/aide_hip.h: %s' % extra
    lib.load()          # dlopen + prototype binding of every entry point


def test_host_side_planners(built):
    """Pure host functions of the ABI (no device needed)."""
    from aide_amd._lib import lib
    assert lib.aide_conv3x3_chunk(3) == 4 and lib.aide_conv3x3_chunk(64) == 8
    for (n, ci, h, co) in [(4, 3, 256, 32), (4, 128, 256, 64), (4, 1024, 32, 512), (4, 512, 16, 512),
                           (4, 512, 20, 1024), (1, 64, 16, 64), (8, 128, 512, 64)]:
        plan = lib.aide_conv3x3_plan(n, ci, h, h, co)
        variant, splitk = plan & 255, plan >> 8
        assert 0 <= variant <= 5 and splitk >= 1
        tco = {0: 32, 1: 64, 2: 128, 3: 64, 4: 64, 5: 128}[variant]
        assert co % tco == 0
        assert lib.aide_conv3x3_ws_bytes(n, h, h, co, splitk) == (0 if splitk == 1 else splitk * n * co * h * h * 4)
        s = lib.aide_conv3x3_wgrad_splits(n, co, ci, h, h)
        assert 1 <= s <= n * ((h + 3) // 4) * ((h + 15) // 16)
        assert lib.aide_conv3x3_wgrad_ws_bytes(n, co, ci, h, h) == s * 9 * co * ci * 4
    assert lib.aide_seg_loss_blocks(256 * 256) >= 1
    assert lib.aide_bn_ws_bytes(64) > 0 and lib.aide_head1x1_ws_bytes(64, 2) > 0


def test_cover_analysis():
    from aide_amd.engine import _Cover
    c = _Cover()
    assert c.write(0, 128) == (False, [])
    assert c.write(64, 128) == (True, [])
    c = _Cover()
    assert c.write(32, 64) == (False, [])
    acc, gaps = c.write(0, 64)
    assert acc and gaps == [(0, 32)]
    acc, gaps = c.write(0, 96)
    assert acc and gaps == [(64, 96)]


@pytest.mark.parametrize('learned', [False, True])
def test_graphs(learned):
    from aide_amd.models_twomodalinputs import fuseunet
    from aide_amd.models_singlemodalinput import UNet
    for ctor, nconv, npool in ((fuseunet, 32, 4), (UNet, 22, 4)):
        net = ctor(2, learned_bilinear=learned)
        net.engine._refresh_params()
        g = net.engine.graph
        kinds = [op['kind'] for op in g.ops]
        assert kinds.count('head') == 1 and kinds[-1] == 'head'
        assert kinds.count('pool') == npool
        assert kinds.count('up') == (0 if learned else 4)
        assert kinds.count('convT') == (4 if learned else 0)
        assert kinds.count('conv') == nconv - (4 if learned else 0)
        # every parameter is owned by exactly one op
        owned = []
        for op in g.ops:
            for key in ('conv', 'bn'):
                if op.get(key) is not None:
                    owned += [id(p) for p in op[key].parameters()]
        assert sorted(owned) == sorted(id(p) for p in net.parameters())
        # cat elimination: the decoder reads [up | skip] buffers that producers wrote in place
        cat_reads = [op for op in g.ops if op['kind'] == 'conv' and op['src'].root.name.startswith('cat_')
                     and op['src'].C == op['src'].root.C]
        assert len(cat_reads) == 4


def test_state_dict_and_init_match_oracle():
    import oracle
    from aide_amd.models_twomodalinputs import fuseunet
    torch.manual_seed(2)
    a = fuseunet(2)
    torch.manual_seed(2)
    b = oracle.fuseunet(2)
    sa, sb = a.state_dict(), b.state_dict()
    assert list(sa) == list(sb) and len(sa) == 226
    assert all(torch.equal(sa[k], sb[k]) for k in sa)
    a.load_state_dict(sb)          # reference-format checkpoints load


def test_variant_state_dicts_match_oracle():
    """fuseunetsaseparate (fuseunet.py:210-322) and the UNet width variants (UNet.py:210-400): same keys and seeded
    initialisation as the oracle (== reference)."""
    import oracle
    from aide_amd.models_twomodalinputs import fuseunetsaseparate
    from aide_amd.models_singlemodalinput import UNet32, UNet128, UNet16, UNet8, UNet4, UNet2
    for ours, ref in ((fuseunetsaseparate, oracle.fuseunetsaseparate), (UNet32, oracle.UNet32),
                      (UNet128, oracle.UNet128)):
        torch.manual_seed(2)
        a = ours(2)
        torch.manual_seed(2)
        b = ref(2)
        sa, sb = a.state_dict(), b.state_dict()
        assert list(sa) == list(sb)
        assert all(torch.equal(sa[k], sb[k]) for k in sa)
        a.engine._refresh_params()
        owned = [id(p) for op in a.engine.graph.ops for key in ('conv', 'bn', 'mod') if op.get(key) is not None
                 for p in op[key].parameters()]
        assert sorted(owned) == sorted(id(p) for p in a.parameters())
    for small, ref in ((UNet16, oracle.UNet16), (UNet8, oracle.UNet8), (UNet4, oracle.UNet4), (UNet2, oracle.UNet2)):
        assert list(small(2).state_dict()) == list(ref(2).state_dict())


def test_no_cpu_fallback():
    from aide_amd.models_twomodalinputs import fuseunet
    from aide_amd import utils as U
    net = fuseunet(2)
    with pytest.raises(RuntimeError, match='HIP device only'):
        net(torch.zeros(1, 3, 32, 32), torch.zeros(1, 3, 32, 32))
    with pytest.raises(RuntimeError, match='HIP device only'):
        U.CEMDiceLoss()(torch.zeros(1, 2, 8, 8), torch.zeros(1, 8, 8, dtype=torch.int64))
    with pytest.raises(RuntimeError):
        net.modal1_downblock1(torch.zeros(1, 3, 32, 32))      # blocks are parameter containers
    with pytest.raises(IndexError):
        U.Coteachingloss_dropimage()                           # reference: default 'mean' cannot work
    with pytest.raises(ValueError):
        from aide_amd.train_files.trainchaos_comparison_1case import build_model
        build_model('nope', 2)


def test_bucket_scheduler():
    from aide_amd.distributed import make_buckets, BucketScheduler
    offsets, numels, off = [], [10, 1000, 5, 5000, 20, 3000], 0
    for n in numels:
        offsets.append(off)
        off += (n + 3) // 4 * 4
    buckets = make_buckets(offsets, numels, 2000)
    assert [b[2] for b in buckets] == [[0, 1, 2, 3], [4, 5]]
    sch = BucketScheduler(buckets, len(numels))
    assert sch.mark([5]) == [] and sch.mark([4]) == [1]
    assert sch.mark([3, 2, 1]) == [] and sch.mark([0, 0]) == [0]


def test_synthetic_contract():
    from aide_amd.synthetic import chaos_batch
    a, b, t = chaos_batch(8, 64, seed=3)
    assert a.shape == (8, 3, 64, 64) and a.dtype == torch.float32 and t.dtype == torch.int64
    assert torch.equal(a[:, 0], a[:, 1]) and torch.equal(a[:, 1], a[:, 2])      # grey replicated x3
    assert abs(a[0, 0].mean().item()) < 1e-5 and abs(a[0, 0].std().item() - 1) < 1e-4
    assert set(np.unique(t.numpy()).tolist()) <= {0, 1}
    a2, _, t2 = chaos_batch(8, 64, seed=3)
    assert torch.equal(a, a2) and torch.equal(t, t2)


def test_polylr_matches_reference_sequence():
    """PolyLR (utils/poly_lr_scheduler.py:27-47) against learning-rate sequences recorded from the reference
    (tests/golden/g11_polylr.npz, oracle/gen_golden.py g11); --lr_policy dispatch of the train mirrors."""
    import os
    import numpy as np
    from aide_amd.utils import PolyLR
    from aide_amd.utils.poly_lr_scheduler import make_scheduler
    fx = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'g11_polylr.npz'))
    for tag in 'ab':
        lr, max_epoch, power = fx['cfg_' + tag]
        opt = torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))], lr=float(lr))
        sch = PolyLR(opt, max_epoch=int(max_epoch), power=float(power))
        seq = [opt.param_groups[0]['lr']]
        for _ in range(24):
            opt.step()
            sch.step()
            seq.append(opt.param_groups[0]['lr'])
        assert np.allclose(seq, fx['lr_' + tag], rtol=1e-12, atol=0)
    opt = torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))], lr=1e-3)
    assert type(make_scheduler('StepLR', opt, 10)).__name__ == 'StepLR'
    assert isinstance(make_scheduler('PolyLR', opt, 10), PolyLR)
    assert make_scheduler('None', opt, 10) is None
    with pytest.raises(ValueError):
        make_scheduler('cosine', opt, 10)


def test_per_image_operand_limit():
    """Planes whose channel slices exceed the 32-bit (2 GiB) per-image offsets of the kernels raise instead of
    wrapping: FuseUNet's 128-channel skip buffer at 2048x2048 is exactly 2 GiB."""
    from aide_amd.models_twomodalinputs import fuseunet
    net = fuseunet(2)
    x = torch.empty(1, 3, 2048, 2048, device='meta')
    with pytest.raises(RuntimeError, match='too large'):
        net.engine.plan_for((x, x))


def test_bench_switch_guard_needs_no_gpu():
    """bench.py refuses unknown AIDE_* switches and non-default library builds before it touches a device"""
    import subprocess
    import sys
    for extra, word in ((dict(AIDE_PROBE_SKIP_WGRAD='1'), 'AIDE_PROBE_SKIP_WGRAD'), (dict(AIDE_HIP_LIB='/tmp/x.so'), 'AIDE_HIP_LIB')):
        env = dict(os.environ, **extra)
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '1'], env=env, capture_output=True,
                           text=True, timeout=300)
        assert r.returncode != 0 and word in r.stderr and not r.stdout.strip()


def test_runtime_switch_inventory():
    """the package reads exactly the environment switches bench.py knows (<= 12 runtime switches)"""
    import re
    import sys
    sys.path.insert(0, ROOT)
    import bench
    found = set()
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'aide_amd')):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                src = open(os.path.join(dirpath, f)).read()
                found |= set(re.findall(r"environ(?:\.get)?\(\s*['\"](AIDE_\w+)", src))
                found |= set(re.findall(r"environ\[\s*['\"](AIDE_\w+)", src))
                found |= set(re.findall(r'getenv\(\s*"(AIDE_\w+)"', src))
                assert 'AIDE_PROBE_' not in src or f.endswith('.py'), 'probe macro in shipped source %s' % f
    assert found <= set(bench.KNOWN_SWITCHES), sorted(found - set(bench.KNOWN_SWITCHES))
    assert len(bench.KNOWN_SWITCHES) <= 12


def test_wgrad_queue_is_caller_owned_host_state(built):
    """the batched slab reduce works on a caller-owned queue handle (no launches here: host state only); the library has
    no deferred-mode global any more"""
    import ctypes
    from aide_amd._lib import lib
    dll = lib.load()
    q1, q2 = ctypes.c_void_p(), ctypes.c_void_p()
    assert dll.aide_wgrad_queue_create(ctypes.byref(q1)) == 0 and q1.value
    assert dll.aide_wgrad_queue_create(ctypes.byref(q2)) == 0 and q2.value and q2.value != q1.value
    assert dll.aide_wgrad_queue_pending(q1) == 0 and dll.aide_wgrad_queue_pending(q2) == 0
    assert dll.aide_wgrad_queue_discard(q1) == 0          # nothing pending
    assert dll.aide_wgrad_queue_flush(q1, None) == 0      # an empty flush launches nothing
    assert dll.aide_wgrad_queue_pending(None) < 0 and dll.aide_wgrad_queue_flush(None, None) < 0
    assert dll.aide_wgrad_queue_destroy(q1) == 0 and dll.aide_wgrad_queue_destroy(q2) == 0
    for gone in ('aide_wgrad_reduce_defer', 'aide_conv_stats_sink', 'aide_conv_epilogue_affine', 'aide_stream_create_cumask'):
        assert not hasattr(dll, gone), '%s is still exported' % gone


def test_keep_largest_connected_components():
    """CPU post-processing of the per-case evaluation (trainchaos_comparison_1case.py:68-77): face connectivity, the
    largest blob only, the FIRST of equally large blobs (np.argmax over regionprops order = raster label order)."""
    from aide_amd.inference import keep_largest_connected_components as keep
    m = np.zeros((6, 6, 3), dtype=np.int64)
    m[0:2, 0:2, 0] = 1                      # 4 voxels
    m[4:6, 3:6, 1:3] = 1                    # 12 voxels: the largest
    m[3, 0, 0] = 1                          # touches nothing by a face (diagonal to the first blob's corner)
    out = keep(m)
    assert out.dtype == np.uint8 and out.sum() == 12 and out[4:6, 3:6, 1:3].all()
    tie = np.zeros((1, 7, 1), dtype=np.int64)
    tie[0, 0:2, 0] = 1
    tie[0, 4:6, 0] = 1                      # two blobs of two: the first one in raster order is kept
    out = keep(tie)
    assert out[0, 0:2, 0].all() and out.sum() == 2
    assert keep(np.zeros((4, 4, 2), dtype=np.int64)).sum() == 0


def test_class_weight_host_logic():
    """utils/_seg.py: class-weight arguments of the loss modules -- two classes stay the (w0, w1) pair of the specialised
    kernels, more travel as a tuple; the host array handed to the *_mc entry points has exactly C entries."""
    from aide_amd.utils import _seg
    assert _seg.class_weights(None) == (1.0, 1.0)
    assert _seg.class_weights(torch.tensor([1.0, 3.0])) == (1.0, 3.0)
    w0, w1 = _seg.class_weights([1.0, 2.0, 0.5])
    assert w1 is None and w0 == (1.0, 2.0, 0.5)
    arr = _seg.class_w_array(w0, w1, 3)
    assert len(arr) == 3 and [float(v) for v in arr] == [1.0, 2.0, 0.5]
    assert [float(v) for v in _seg.class_w_array(1.0, 1.0, 5)] == [1.0] * 5         # unweighted: any class count
    with pytest.raises(RuntimeError):
        _seg.class_w_array(w0, w1, 4)                                               # 3 weights, 4 classes
    with pytest.raises(RuntimeError):
        _seg.class_w_array(1.0, 3.0, 3)                                             # a 2-class pair on 3 classes
    with pytest.raises(NotImplementedError):
        _seg.class_weights([1.0] * 9)


def test_engine_config_is_per_engine_and_drops_plans():
    """EngineConfig (net.engine.config): unknown switches raise, assigning a NEW value calls the owner's invalidation hook once,
    re-assigning the same value does not, the snapshot (part of every launch tape's key) follows, two configs are independent."""
    from aide_amd.engine import EngineConfig
    calls = []
    a, b = EngineConfig(lambda: calls.append('a')), EngineConfig(lambda: calls.append('b'))
    s0 = a.snapshot()
    assert s0 == b.snapshot() and len(s0) == len(EngineConfig.DEFAULTS)
    a.use_winograd = False
    a.use_winograd = False
    assert calls == ['a'] and a.snapshot() != s0 and b.snapshot() == s0 and b.use_winograd is True
    a.use_winograd = True
    assert calls == ['a', 'a'] and a.snapshot() == s0
    with pytest.raises(AttributeError, match='no switch'):
        a.use_winogard = False


def test_cli_defaults_match_reference_g21():
    """The mirrored train scripts' parse_args([]) against the reference's own `add_argument` tables (fixture g21, written by
    oracle/gen_golden.py::g21_cli_defaults from the scripts' syntax trees): every reference flag exists with the same default
    and type; the only flag the reference does not have is --steps_per_epoch (size of the synthetic epoch)."""
    import importlib
    import json
    import os
    import numpy as np
    fx = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'g21_cli_defaults.npz'))
    scripts = sorted(set(k.split('/')[0] for k in fx.files))
    assert len(scripts) == 5
    for name in scripts:
        mod = importlib.import_module('aide_amd.train_files.' + name)
        args = vars(mod.parse_args([]))
        flags = [str(f) for f in fx[name + '/flags']]
        for flag, dflt, ty in zip(flags, fx[name + '/defaults'], fx[name + '/types']):
            key = flag.lstrip('-')
            assert key in args, '%s: missing flag %s' % (name, flag)
            want = json.loads(str(dflt))
            assert args[key] == want, '%s %s: default %r, reference %r' % (name, flag, args[key], want)
            if not isinstance(want, list) and 'data_' not in key:            # the flag's type=: what a command-line value becomes
                got = vars(mod.parse_args([flag, '1']))[key]
                assert type(got).__name__ == str(ty), (name, flag, type(got).__name__, str(ty))
        assert sorted(set(args) - set(f.lstrip('-') for f in flags)) == ['steps_per_epoch'], name


def test_resumefile_initialises_both_networks(tmp_path):
    """trainkidney_proposed_mask1.py:180-182: torch.load(args.resumefile)['net'] goes into BOTH networks before the loop; a
    missing file warns and leaves the seeded initialisation (the reference would fail)."""
    import oracle
    from aide_amd.models_singlemodalinput import UNet
    from aide_amd.train_files.trainchaos_proposed_30cases1labeled import load_resumefile, VARIANTS
    assert VARIANTS['kidney']['resume'] and not VARIANTS['breast']['resume'] and not VARIANTS['prostate']['resume']
    torch.manual_seed(5)
    ref = oracle.UNet(2)
    path = str(tmp_path / 'UNet_besttraindice_Task1Mask1.pkl')
    torch.save({'net': ref.state_dict(), 'loss': 0.1, 'epoch': 7}, path)
    torch.manual_seed(6)
    n1, n2 = UNet(2), UNet(2)
    assert not torch.equal(n1.state_dict()['last_conv1.weight'], ref.state_dict()['last_conv1.weight'])
    assert load_resumefile(path, (n1, n2)) is True
    for net in (n1, n2):
        sd = net.state_dict()
        assert list(sd) == list(ref.state_dict()) and all(torch.equal(sd[k], v) for k, v in ref.state_dict().items())
    assert load_resumefile(str(tmp_path / 'nothing.pkl'), (n1, n2)) is False
    assert load_resumefile(None, (n1, n2)) is False
