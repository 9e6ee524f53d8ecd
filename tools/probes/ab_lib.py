"""A/B of two builds of the library in one process-per-variant run: CB_EXP_LIB=<path to .so> python tools/probes/ab_lib.py <script> [args]"""
import os, runpy, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gnn_tail_generalization_amd import _lib
if os.environ.get('CB_EXP_LIB'):
    import ctypes
    _lib.LIB_PATH = os.path.abspath(os.environ['CB_EXP_LIB'])
    probe = ctypes.CDLL(_lib.LIB_PATH)          # an older build may lack the newest entry points: do not bind those
    for name in [n for n in _lib.SIGNATURES if not hasattr(probe, n)]:
        del _lib.SIGNATURES[name]
sys.argv = sys.argv[1:]
runpy.run_path(sys.argv[0], run_name='__main__')
