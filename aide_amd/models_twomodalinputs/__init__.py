from .fuseunet import fuseunet, fuseunetsa, fuseunetsaseparate  # noqa: F401  (reference: models_twomodalinputs/__init__.py:1)
