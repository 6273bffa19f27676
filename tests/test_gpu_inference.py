"""-m gpu: the per-case inference path: eval-mode batched forward + label map + Dice3d."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def test_label_map_matches_argmax_softmax(dev):
    """aide_label_map == torch.argmax(F.softmax(x, 1), 1) on CPU, bit-exact: random logits, exact ties
    (-> class 0) and margins around the softmax rounding threshold (2^-25 .. 2^-22)."""
    from aide_amd.inference import label_map
    g = torch.Generator().manual_seed(5)
    x = torch.randn(3, 2, 40, 24, generator=g) * 3
    x[0, 1, :8] = x[0, 0, :8]                                   # exact ties
    for k, e in enumerate((2.0 ** -22, 2.0 ** -20, 1e-5, -1e-5, -2.0 ** -22)):
        x[1, 1, k] = x[1, 0, k] * 0 + 1.0 + e                   # z0 = 1, z1 = 1 + e (representable steps)
        x[1, 0, k] = 1.0
    ref = torch.argmax(F.softmax(x, dim=1), dim=1)
    out = label_map(x.to(dev)).cpu()
    assert out.dtype == torch.int64 and torch.equal(out, ref)
    # sub-threshold margins: both sides of the rounding boundary are implementation-defined in aten
    # itself; our rule (z1 > z0 and exp(z0 - z1) < 1) is checked on its own terms
    z0 = torch.zeros(1, 1, 1, 4)
    z1 = torch.tensor([0.0, 2.0 ** -30, 2.0 ** -23, -2.0 ** -23]).view(1, 1, 1, 4)
    got = label_map(torch.cat([z0, z1], 1).to(dev)).cpu().view(-1).tolist()
    assert got == [0, 0, 1, 0]
    with pytest.raises(RuntimeError):
        label_map(x)                                            # CPU tensor: no fallback


@pytest.mark.parametrize('name', ['fuseunet', 'unet'])
def test_predict_case_vs_reference_volume(dev, name):
    """trainchaos_comparison_1case.py:233-273 vs the real reference's label volume (g6_inference.npz):
    identical wherever the reference's logit margin exceeds fp32 noise; Dice3d within 1e-3."""
    from aide_amd import utils as U
    from aide_amd.inference import predict_case, predict_labels, Dice3d_fn
    from aide_amd.models_singlemodalinput import UNet
    from aide_amd.models_twomodalinputs import fuseunet
    from aide_amd.optim import Adam
    fx = np.load(os.path.join(GOLD, 'g6_inference.npz'))
    two = name == 'fuseunet'
    g = torch.Generator().manual_seed(1234)
    xs = [torch.randn(2, 3, 32, 32, generator=g).to(dev) for _ in range(2 if two else 1)]
    t = (torch.rand(2, 32, 32, generator=g) > 0.7).long().to(dev)
    torch.manual_seed(2)
    net = (fuseunet(2) if two else UNet(2)).to(dev)
    net.train()
    w = torch.tensor([1.0, 1.0])
    crit = U.CEMDiceLoss(cediceweight=w, ceclassweight=w, diceclassweight=w)
    opt = Adam(net.parameters(), lr=1e-4, amsgrad=True)
    for _ in range(2):
        opt.zero_grad()
        crit(net(*xs), t).backward()
        opt.step()
    sl = [torch.from_numpy(fx['%s/slices%d' % (name, i)]) for i in range(2 if two else 1)]
    with pytest.raises(RuntimeError):
        predict_labels(net, *sl)                                # train mode is not the reference loop
    net.eval()
    with torch.no_grad():
        net.last_conv1.bias[1] -= float(fx[name + '/head_bias1_shift'])
    vol = predict_case(net, *sl, batch_size=4)                  # 6 slices -> batches of 4 + 2
    assert vol.shape == (48, 32, 6) and vol.dtype == np.int64
    margin = np.transpose(fx[name + '/margin'], (1, 2, 0))
    bad = vol != fx[name + '/labels']
    # two fp32 training steps + eval forward: logits agree to ~1e-5, so labels may differ only there
    assert not np.any(bad & (np.abs(margin) > 2e-4)), np.abs(margin)[bad].max()
    assert bad.mean() < 2e-3
    tgt = fx[name + '/targets'].astype(np.int64)
    assert abs(Dice3d_fn(vol, tgt) - float(fx[name + '/dice3d'])) < 1e-3
    # batching is exact: bs=1 slices give the same labels as one batch (eval BN, per-sample kernels)
    one = np.stack([predict_case(net, *[s[i:i + 1] for s in sl])[..., 0] for i in range(6)], -1)
    assert (one != vol).mean() < 1e-3
    # device-side Dice3d
    lab = predict_labels(net, *sl)
    assert abs(Dice3d_fn(lab.permute(1, 2, 0), torch.from_numpy(tgt)) - Dice3d_fn(vol, tgt)) < 1e-12
