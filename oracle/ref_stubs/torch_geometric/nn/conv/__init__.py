from . import gcn_conv  # noqa: F401
