class _Absent:
    def __init__(self, *a, **k):
        raise RuntimeError('torch_geometric datasets are not available (stub)')


Planetoid = Coauthor = WebKB = Actor = Amazon = WikipediaNetwork = _Absent
