from .UNet import UNet, UNetsa, UNet128, UNet32, UNet16, UNet8, UNet4, UNet2  # noqa: F401  (reference: models_singlemodalinput/__init__.py:1)
