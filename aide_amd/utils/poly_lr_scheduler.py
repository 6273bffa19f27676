"""PolyLR (reference: utils/poly_lr_scheduler.py:27-47): lr_e = lr_0 * (1 - (e mod max_epoch) / max_epoch) ** power,
stepped once per epoch, with the reference's one-step lag (see _factor; pinned by tests/golden/g11_polylr.npz).  Host logic only: the fused Adam (aide_amd/optim.py) reads `param_group['lr']` at every step."""
from torch.optim.lr_scheduler import LambdaLR


class PolyLR(LambdaLR):
    def __init__(self, optimizer, max_epoch, power=0.9, last_epoch=-1):
        self.max_epoch, self.power = max_epoch, power
        LambdaLR.__init__(self, optimizer, self._factor, last_epoch)

    def _factor(self, epoch):
        # the reference's constructor steps to epoch 0 and then resets its counter to -1 (poly_lr_scheduler.py:19-20), so
        # the first scheduler.step() of a run repeats epoch 0: after k >= 1 steps the rate is that of epoch k - 1
        e = max(epoch - 1, 0)
        return (1.0 - float(e % self.max_epoch) / float(self.max_epoch)) ** self.power


def make_scheduler(policy, optimizer, end_epoch):
    """--lr_policy of the train scripts (trainchaos_comparison_1case.py:172-176, :325-326): 'StepLR' (step 30, gamma 0.5),
    'PolyLR' (power 0.9 over the run) or 'None' (constant)."""
    from torch.optim.lr_scheduler import StepLR
    if policy == 'StepLR':
        return StepLR(optimizer, step_size=30, gamma=0.5)
    if policy == 'PolyLR':
        return PolyLR(optimizer, max_epoch=end_epoch, power=0.9)
    if policy == 'None':
        return None
    raise ValueError('unknown --lr_policy %r' % (policy,))
