"""Three comparison steps on the g1 fixture under different kernel-selection rules (noise vs bug triage)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aide_amd import engine, utils as U
from aide_amd.models_twomodalinputs import fuseunet
from aide_amd.optim import Adam
from aide_amd._lib import lib
GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
fx, g1 = np.load(os.path.join(GOLD, 'g5_adam.npz')), np.load(os.path.join(GOLD, 'g1_fuseunet.npz'))
dev = 'cuda'
x1, x2 = torch.from_numpy(g1['x0']).to(dev), torch.from_numpy(g1['x1']).to(dev)
t = torch.from_numpy(g1['targets']).to(dev)
print('input', tuple(x1.shape), 'golden', fx['losses'])
def run(tag):
    w = torch.tensor([1.0, 1.0]); torch.manual_seed(2)
    net = fuseunet(2).to(dev); net.train()
    crit = U.CEMDiceLoss(cediceweight=w, ceclassweight=w, diceclassweight=w)
    opt = Adam(net.parameters(), lr=1e-4, amsgrad=True)
    ls = []
    for _ in range(3):
        opt.zero_grad(); loss = crit(net(x1, x2), t); loss.backward(); opt.step(); ls.append(loss.item())
    print('%-28s' % tag, ['%.7f' % v for v in ls], 'rel', ['%.1e' % abs(a / b - 1) for a, b in zip(ls, fx['losses'])])
run('winograd everywhere')
old = engine.use_winograd
engine.use_winograd = lambda n, cin, h, w, cout: bool(lib.aide_conv3x3_wino_supported(cin, h, w, cout)) and (cin >= cout or h * w >= 128 * 128)
run('previous rule')
engine.use_winograd = old
engine.USE_WINOGRAD[0] = False
run('direct only')
