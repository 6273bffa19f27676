"""Quick end-to-end parity probe on the GPU (not a test): model fwd/bwd/Adam vs the oracle on CPU."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import oracle
from aide_amd.models_twomodalinputs import fuseunet
from aide_amd.models_singlemodalinput import UNet
from aide_amd import utils as U
from aide_amd.optim import Adam

dev = torch.device('cuda:0')
size = int(sys.argv[1]) if len(sys.argv) > 1 else 32
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 2


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


for name, ours_c, ref_c, kw, two in (('fuseunet', fuseunet, oracle.fuseunet, {}, True),
                                     ('fuseunet-learned', fuseunet, oracle.fuseunet, dict(learned_bilinear=True), True),
                                     ('UNet', UNet, oracle.UNet, {}, False),
                                     ('UNet-learned', UNet, oracle.UNet, dict(learned_bilinear=True), False)):
    torch.manual_seed(2); ref = ref_c(2, **kw)
    torch.manual_seed(2); net = ours_c(2, **kw).to(dev)
    g = torch.Generator().manual_seed(1234)
    xs = [torch.randn(nb, 3, size, size, generator=g) for _ in range(2 if two else 1)]
    t = (torch.rand(nb, size, size, generator=g) > 0.7).long()
    w = torch.tensor([1.0, 1.0])
    ref.train(); net.train()
    out_r = ref(*xs)
    loss_r = oracle.CEMDiceLoss(w, w, w)(out_r, t)
    loss_r.backward()
    out = net(*[x.to(dev) for x in xs])
    crit = U.CEMDiceLoss(cediceweight=w, ceclassweight=w, diceclassweight=w)
    loss = crit(out, t.to(dev))
    loss.backward()
    print(name, 'logits rel err %.2e' % rel(out, out_r), 'loss', loss.item(), loss_r.item())
    worst = 0
    for (k, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        dead = k.endswith('bias') and ('conv' in k or 'bilinear_up' in k) and 'last_conv1' not in k and 'bn' not in k
        e = (p.grad.cpu().double() - q.grad.double()).abs().max().item()
        s = q.grad.double().abs().max().item()
        if dead:
            assert e < 1e-5, (k, e)
        else:
            worst = max(worst, e / (s + 1e-12))
            if e / (s + 1e-12) > 1e-3:
                print('   BAD grad', k, e, s)
    print('   worst live-grad rel err %.2e' % worst)
    opt_r = torch.optim.Adam(ref.parameters(), lr=1e-4, amsgrad=True); opt_r.step()
    opt = Adam(net.parameters(), lr=1e-4, amsgrad=True); opt.step()
    worst = 0
    for (k, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        dead = k.endswith('bias') and 'last_conv1' not in k and 'bn' not in k.split('.')[-2] and not k.split('.')[-2].isdigit() or (k.endswith('bias') and k.split('.')[-2] in ('conv1', 'conv2', '0', '1') and 'last' not in k and 'bn' not in k)
        if dead:
            continue
        worst = max(worst, (p.detach().cpu().double() - q.detach().double()).abs().max().item())
    print('   post-Adam max abs param diff %.2e' % worst)
    for (k, b), (_, c) in zip(net.named_buffers(), ref.named_buffers()):
        if 'running' in k:
            assert rel(b, c) < 1e-3, (k, rel(b, c))
    # second step sanity + eval forward
    out2 = net(*[x.to(dev) for x in xs]); out2_r = ref(*xs)
    print('   step-2 logits rel err %.2e' % rel(out2, out2_r))
    net.eval(); ref.eval()
    with torch.no_grad():
        print('   eval logits rel err %.2e' % rel(net(*[x.to(dev) for x in xs]), ref(*xs)))
print('done')
