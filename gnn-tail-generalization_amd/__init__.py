"""MI355X-native Cold Brew TeacherGNN hot path (package directory `gnn-tail-generalization_amd`,
importable as `gnn_tail_generalization_amd` through the shim at the repo root).

Layout: csrc/ (hand-written gfx950 HIP kernels + the C ABI of include/coldbrew_hip.h),
_lib.py (ctypes binding), graph.py / ops.py (device graph + autograd operators), and the
host-side mirror of the reference's interface for this path: GNN_model/, utils.py,
base_options.py, trainer_node_classification.py.
"""
__version__ = '0.1.0'
