"""Mirror of the reference's GNN_model package (TeacherGNN body) on the HIP path."""
from .GCN import GCNConv, TricksComb  # noqa: F401
from .GNN_normalizations import GNN_norm, TeacherGNN  # noqa: F401
