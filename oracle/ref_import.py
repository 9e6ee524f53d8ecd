"""TEST INFRASTRUCTURE ONLY — imports the *unmodified* reference from /root/reference.

Used in the build container (never on the GPU box, never by the product path) to
  (a) generate the golden vectors under tests/golden/ (tests/golden/make_golden.py), and
  (b) validate oracle/coldbrew_oracle.py against the real reference modules.

Recipe: put oracle/ref_stubs (stand-in `dgl`, stubs for
torch_scatter / torch_geometric / ogb / cv2) and /root/reference on sys.path, never
write bytecode into the read-only reference tree, chdir to a scratch dir because the
trainer writes saved_models/, wIns/ relative to cwd (trainer_node_classification.py:290-292).
"""
import contextlib
import os
import sys
import tempfile

REFERENCE_ROOT = os.environ.get('COLDBREW_REFERENCE_ROOT', '/root/reference')
_STUBS = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'ref_stubs')


def reference_available():
    return os.path.isfile(os.path.join(REFERENCE_ROOT, 'GNN_model', 'GCN.py'))


_loaded = {}


def load_reference():
    """Returns a namespace of reference modules: GCN, norm_tricks, res_tricks,
    GNN_normalizations, utils, trainer_node_classification."""
    if _loaded:
        return _loaded['ns']
    if not reference_available():
        raise RuntimeError(f'reference tree not found at {REFERENCE_ROOT}')
    sys.dont_write_bytecode = True
    os.environ.setdefault('MPLBACKEND', 'Agg')
    for p in (REFERENCE_ROOT, _STUBS):
        if p not in sys.path:
            sys.path.insert(0, p)
    # the product package mirrors some reference module names (GNN_model, utils, ...)
    # under its own package namespace, so there is no clash with these top-level imports.
    scratch = tempfile.mkdtemp(prefix='coldbrew_ref_')
    cwd = os.getcwd()
    os.chdir(scratch)
    try:
        import GNN_model.GCN as GCN
        import GNN_model.norm_tricks as norm_tricks
        import GNN_model.res_tricks as res_tricks
        import GNN_model.drop_tricks as drop_tricks
        import GNN_model.GNN_normalizations as GNN_normalizations
        import utils as ref_utils
        import trainer_node_classification as ref_trainer
    finally:
        os.chdir(cwd)

    class NS:
        pass

    ns = NS()
    ns.GCN, ns.norm_tricks, ns.res_tricks, ns.drop_tricks = GCN, norm_tricks, res_tricks, drop_tricks
    ns.GNN_normalizations, ns.utils, ns.trainer = GNN_normalizations, ref_utils, ref_trainer
    ns.scratch = scratch
    _loaded['ns'] = ns
    return ns


@contextlib.contextmanager
def in_scratch():
    ns = load_reference()
    cwd = os.getcwd()
    os.chdir(ns.scratch)
    try:
        yield ns
    finally:
        os.chdir(cwd)
