"""A/B of two builds of the library in one process-per-variant run: CB_EXP_LIB=<path to .so> python tools/probes/ab_lib.py <script> [args]"""
import os, runpy, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gnn_tail_generalization_amd import _lib
if os.environ.get('CB_EXP_LIB'):
    _lib.LIB_PATH = os.path.abspath(os.environ['CB_EXP_LIB'])
sys.argv = sys.argv[1:]
runpy.run_path(sys.argv[0], run_name='__main__')
