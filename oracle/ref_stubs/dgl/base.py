"""TEST INFRASTRUCTURE ONLY."""


class DGLError(Exception):
    pass
