from .fuseunet import fuseunet  # noqa: F401  (reference: models_twomodalinputs/__init__.py:1)
