class NormalizeFeatures:
    pass
