"""CU-partitioned streams for the backward of the residual trunk (new relative to the reference, which is a single-stream PyTorch
program: trainer_node_classification.py:303-320).  MEASURED SLOWER — off unless CB_BWD_OVERLAP=1 (profiles/r03_cu_partition_overlap.md).

The idea: the weight gradients dW_l = X_l^T (a * dZ_l) (autograd of GNN_model/GCN.py:225) are off the critical path of the backward
(nothing needs them before the optimiser step) and MFMA-bound at the power-limited clock, while the chain they hang off (reverse
aggregation -> trunk backward -> next reverse aggregation) moves bytes.  So run the chain on a stream confined to one CU set and the
weight-gradient GEMMs beside it on a stream confined to the rest (hipExtStreamCreateWithCUMask through cb_stream_create_cu_mask).

What the measurements say (tools/overlap_probe.py, bench.py with CB_BWD_OVERLAP=1):
  * the stand-alone aggregation keeps its speed on 192 of 256 CUs (15.5 vs 15.2 ms) and a copy-like pass hides a GEMM completely
    (GEMM on 64 CUs || 6 copies on 192: 27.8 ms against 33.8 one after the other) — but
  * the aggregation + GEMM kernel that the chain actually consists of is bound per CU (gathers and matrix-core time add up on a SIMD):
    on 192 CUs it takes 23.5 - 25.1 ms instead of 18.4, the elementwise passes 5.2 / 14.0 instead of 3.8 / 11.5, and the GEMM on 64 CUs
    23 - 25 ms instead of 6.5: the step gets 13 - 60 ms SLOWER (213 - 262 ms, erratic run to run);
  * a small CU set for the elementwise passes does not work either: a CU mask of 32 CUs is one XCD, whose path to memory carries
    1.5 TB/s (copy 13.0 ms instead of 4.4), and the GEMM on the other seven XCDs loses its 8-way XCD tile mapping (10.1 vs 7.3 ms);
  * plain (unmasked) streams: the kernels do not overlap at all (52.4 ms for GEMM || 3 aggregations, 52.8 one after the other).
Results are bit-identical in every form (tests/test_gpu_agg_gemm.py runs a training step both ways)."""
import ctypes
import os

import torch

from . import _lib

_PAIRS = {}


def side_cus_default(n_cus):
    v = os.environ.get('CB_BWD_OVERLAP_CUS')
    return int(v) if v else n_cus // 4


def enabled():
    return os.environ.get('CB_BWD_OVERLAP', '0') == '1'


def _masked(lib, bits, n_cus):
    words = (n_cus + 31) // 32
    arr = (ctypes.c_uint32 * words)(*[(bits >> (32 * i)) & 0xffffffff for i in range(words)])
    st = ctypes.c_void_p()
    _lib.check(lib.cb_stream_create_cu_mask(arr, words, ctypes.byref(st)), 'cb_stream_create_cu_mask')
    return torch.cuda.ExternalStream(st.value)


def partition(device):
    """(main_stream, side_stream, main_cus) for `device`, created once: side = the first CB_BWD_OVERLAP_CUS CUs (default a quarter), main =
    the others."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), os.environ.get('CB_BWD_OVERLAP_CUS'))
    pair = _PAIRS.get(key)
    if pair is None:
        lib = _lib.load()
        n_cus = torch.cuda.get_device_properties(device).multi_processor_count
        k = max(1, min(n_cus - 1, side_cus_default(n_cus)))
        with torch.cuda.device(device):
            side = _masked(lib, (1 << k) - 1, n_cus)
            main = _masked(lib, ((1 << n_cus) - 1) & ~((1 << k) - 1), n_cus)
        pair = _PAIRS[key] = (main, side, n_cus - k)
    return pair
