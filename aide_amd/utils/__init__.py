# mirrors the reference's utils/__init__.py:1-10 for the hot-path symbols
from .loss2d import (CrossEntropyLoss2d, DiceLoss, CEDiceLoss, CEMDiceLoss, MulticlassDiceLoss, Dice_Loss,  # noqa: F401
                     MulticlassMSELoss, CEMDiceLossImage)
from .metrics2d import Dice_fn, Dice_fn_Nozero, TP_TN_FP_FN, IoU_fn  # noqa: F401
from .coteach_loss import (Coteachingloss_dropimage, Coteachingloss_weightimage, Coteachingloss_dropregionce,  # noqa: F401
                           Coteachingloss_dropimagedroppixel, KLbidirection, CoTeachingProposedLoss,
                           pseudo_label_ensemble)
from .augment import reverseaug, reverse_aug_tensor  # noqa: F401
from .reg_loss import Pixelcoreg_Focalloss, Pixelcoreg_Focalloss_twomodel  # noqa: F401
from .poly_lr_scheduler import PolyLR  # noqa: F401
