def build_model(config):
    """Empty stub, as in the reference (model/__init__.py:1-5)."""
