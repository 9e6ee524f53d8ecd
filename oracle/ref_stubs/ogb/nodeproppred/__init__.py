class PygNodePropPredDataset:
    pass


class Evaluator:
    def __init__(self, *a, **k):
        pass
