from .fuseunet import fuseunet, fuseunetsa  # noqa: F401  (reference: models_twomodalinputs/__init__.py:1)
