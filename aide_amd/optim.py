"""Fused multi-tensor Adam(amsgrad) — one HIP launch for all parameter tensors of a group.

Replaces torch.optim.Adam(net.parameters(), lr=args.lr, amsgrad=True)
(train_files/trainchaos_comparison_1case.py:170; two instances in
trainchaos_proposed_30cases1labeled.py:231-232). Same constructor arguments, param_groups layout and
state keys ('step', 'exp_avg', 'exp_avg_sq', 'max_exp_avg_sq') as torch.optim.Adam, so LR schedulers
(StepLR(30, 0.5), :173-176) and state_dict round-trips keep working.  HBM-bound: 20 B/param read,
16 B/param written (amsgrad)."""
import ctypes

import torch

from . import engine
from ._lib import lib, check
from .ops import stream_ptr, ptr


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1:
            raise ValueError('invalid Adam hyper-parameters')
        super(Adam, self).__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                                                amsgrad=amsgrad))
        self._tables = {}
        self._gtabs = {}
        self._fast = {}

    def load_state_dict(self, state_dict):
        super(Adam, self).load_state_dict(state_dict)
        self._tables = {}                          # raw pointers to the old moment tensors
        self._fast = {}

    def add_param_group(self, param_group):
        super(Adam, self).add_param_group(param_group)
        self._fast = {}

    def zero_grad(self, set_to_none=True):
        """set_to_none=True (torch's default): drop the gradients -- the next backward of an aide_amd model then hands out
        views of its gradient arena without any accumulation pass."""
        if not set_to_none:
            return super(Adam, self).zero_grad(set_to_none=False)
        for group in self.param_groups:
            for p in group['params']:
                p.grad = None

    def _launch(self, tab, gtab, n, group, step):
        b1, b2 = group['betas']
        check(lib.aide_adam_amsgrad_multi(ptr(tab['p']), ptr(gtab), ptr(tab['m']), ptr(tab['v']),
                                          ptr(tab['vmax']), ptr(tab['sizes']), ptr(tab['starts']),
                                          n, tab['total_blocks'], float(group['lr']), float(b1),
                                          float(b2), float(group['eps']), float(group['weight_decay']),
                                          int(bool(group['amsgrad'])), step, stream_ptr()), 'adam')

    def _static_table(self, gi, plist):
        # the table holds raw pointers to the parameters AND their moment tensors: load_state_dict (or any replacement
        # of a state tensor) must invalidate it
        key = tuple(p.data_ptr() for p in plist) + tuple(
            t.data_ptr() for p in plist for t in (self.state[p]['exp_avg'], self.state[p]['exp_avg_sq'],
                                                  self.state[p].get('max_exp_avg_sq')) if t is not None)
        tab = self._tables.get(gi)
        if tab is not None and tab['key'] == key:
            return tab
        dev = plist[0].device
        sizes = [p.numel() for p in plist]
        starts, acc = [], 0
        for s in sizes:
            starts.append(acc)
            acc += (s + 1023) // 1024
        st = [self.state[p] for p in plist]

        def table(vals):
            return torch.tensor(vals, dtype=torch.int64).to(dev)
        tab = dict(key=key, total_blocks=acc,
                   p=table([p.data_ptr() for p in plist]),
                   m=table([s['exp_avg'].data_ptr() for s in st]),
                   v=table([s['exp_avg_sq'].data_ptr() for s in st]),
                   vmax=table([s['max_exp_avg_sq'].data_ptr() if 'max_exp_avg_sq' in s else 0 for s in st]),
                   sizes=table(sizes), starts=table(starts))
        self._tables[gi] = tab
        return tab

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            # steady state: same parameters, same moment tensors, every gradient present and where it was last step (the
            # engine's arena views) -- nothing to validate or rebuild, one launch
            fast = self._fast.get(gi)
            if fast is not None:
                params = group['params']
                grads = [p.grad for p in params]
                if fast['params'] is params and len(params) == fast['n'] and all(g is not None for g in grads) and \
                        tuple(g.data_ptr() for g in grads) == fast['gkey'] and \
                        tuple(p.data_ptr() for p in params) == fast['pkey'] and \
                        bool(group['amsgrad']) == fast['amsgrad']:
                    step = fast['step'] = fast['step'] + 1
                    for st in fast['states']:
                        st['step'] = step
                    self._launch(fast['tab'], fast['gtab'], fast['n'], group, step)
                    self._keep = (fast['gtab'], None)
                    continue
                self._fast.pop(gi, None)
            plist = [p for p in group['params'] if p.grad is not None]
            if not plist:
                continue
            for p in plist:
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError('aide_amd.optim.Adam: parameters must be contiguous fp32 HIP tensors')
                st = self.state[p]
                if len(st) and group['amsgrad'] and 'max_exp_avg_sq' not in st:
                    st['max_exp_avg_sq'] = st['exp_avg_sq'].clone()       # state loaded from a non-amsgrad run
                if len(st) == 0:
                    st['step'] = 0
                    st['exp_avg'] = torch.zeros_like(p)
                    st['exp_avg_sq'] = torch.zeros_like(p)
                    if group['amsgrad']:
                        st['max_exp_avg_sq'] = torch.zeros_like(p)
            if len(plist) != len(group['params']):
                self._tables.pop(gi, None)        # membership changed: rebuild
            tab = self._static_table(gi, plist)
            grads = []
            for p in plist:
                g = p.grad
                if not g.is_contiguous():
                    g = g.contiguous()
                grads.append(g)
            # the gradient pointer table only changes when the gradients move (the engine's arena usually comes back at
            # the same address every step): rebuild + upload it only then
            gkey = tuple(g.data_ptr() for g in grads)
            cached = self._gtabs.get(gi)
            if cached is not None and cached[0] == gkey:
                gtab = cached[1]
            else:
                gtab = torch.tensor(gkey, dtype=torch.int64).to(plist[0].device, non_blocking=True)
                self._gtabs[gi] = (gkey, gtab)
            # a torch.optim.Adam checkpoint stores 'step' as a (float) tensor: coerce
            step = int(self.state[plist[0]]['step']) + 1
            for p in plist:
                self.state[p]['step'] = step
            self._launch(tab, gtab, len(plist), group, step)
            # keep alive until the next step (async launch): the pointer table and the contiguous COPIES made above -- not
            # the parameters' own gradient tensors: a reference held here makes the engine's next backward pass take them
            # for gradients the caller still wants (it then leaves that arena alone and fills a new one, every step)
            self._keep = (gtab, [g for g, p in zip(grads, plist) if g is not p.grad])
            if len(plist) == len(group['params']) and all(g is p.grad for g, p in zip(grads, plist)):
                self._fast[gi] = dict(params=group['params'], n=len(plist), gkey=gkey,
                                      pkey=tuple(p.data_ptr() for p in plist), amsgrad=bool(group['amsgrad']),
                                      step=step, states=[self.state[p] for p in plist], tab=tab, gtab=gtab)
        engine.PARAM_EPOCH[0] += 1                # parameters changed behind tensor._version's back
        return loss
