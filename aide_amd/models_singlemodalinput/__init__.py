from .UNet import UNet, UNetsa  # noqa: F401  (reference: models_singlemodalinput/__init__.py:1)
