"""CPU oracle for the AIDE FuseUNet/UNet training hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in ``aide_amd/`` (the product path) may
import this package; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` use it, and only as the checker.

The oracle is a plain-PyTorch (stock aten, CPU, fp32) restatement of the
reference algorithm.  Each function cites the reference file:line it follows.
It is pinned by ``tests/golden/*.npz`` which were produced by importing the
real reference (``oracle/gen_golden.py``, run in the build container where
``/root/reference`` exists); see DESIGN.md §2.
"""
from .nets import (fuseunet, UNet, fuseunetsa, UNetsa, Spatial_Attention, fuseunetsaseparate,  # noqa: F401
                   UNet128, UNet32, UNet16, UNet8, UNet4, UNet2)
from .losses import (  # noqa: F401
    CrossEntropyLoss2d, DiceLoss, MulticlassDiceLoss, MulticlassMSELoss,
    CEMDiceLoss, CEMDiceLossImage, Coteachingloss_dropimage,
    Coteachingloss_weightimage, Coteachingloss_dropregionce, Coteachingloss_dropimagedroppixel,
    KLbidirection, Pixelcoreg_Focalloss, Pixelcoreg_Focalloss_twomodel, Dice_fn, sharpen,
    Dice_fn_Nozero, TP_TN_FP_FN, IoU_fn, Dice_Loss, CEDiceLoss,
)
from .steps import comparison_step, proposed_step  # noqa: F401
