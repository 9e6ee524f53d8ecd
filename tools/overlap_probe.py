"""Does an MFMA-bound weight-gradient GEMM overlap with the HBM-bound aggregation when the two run on two streams — plain streams
(round 2: no) and streams confined to disjoint CU sets (hipExtStreamCreateWithCUMask)?
usage: python tools/overlap_probe.py [--gemm-cus 64]     prints times alone, together, and the mask layouts tried"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_tail_generalization_amd import gemm  # noqa: E402
from gnn_tail_generalization_amd.data import synthetic_data  # noqa: E402
from gnn_tail_generalization_amd.graph import CSRGraph  # noqa: E402

hip = ctypes.CDLL('libamdhip64.so')


def masked_stream(bits):
    """torch ExternalStream confined to the CUs whose bit is set (256 bits = 8 words)."""
    words = (ctypes.c_uint32 * 8)(*[(bits >> (32 * i)) & 0xffffffff for i in range(8)])
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


def mask_first(n):           # CUs 0 .. n-1
    return (1 << n) - 1


def mask_strided(n, total=256):      # every (total / n)-th CU
    step = total // n
    m = 0
    for i in range(0, total, step):
        m |= 1 << i
    return m


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gemm-cus', type=int, default=64)
    a = ap.parse_args()
    dev = 'cuda:0'
    data = synthetic_data('S-pl10M', seed=0, device=dev)
    G = CSRGraph(data.edge_index, data.x.shape[0])
    n = G.N
    h = torch.rand(n, 256, device=dev)
    out = torch.empty_like(h)
    x = torch.rand(n, 256, device=dev)
    g = torch.rand(n, 256, device=dev)
    rs = torch.rand(n, device=dev)
    b2 = torch.empty_like(h)

    def t(fn, it=3):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(it):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / it

    def spmm():
        G.spmm(h, out=out)

    def tn():
        gemm.mm_tn(x, g, rowscale=rs)

    def copy():
        b2.copy_(h)

    print('alone, whole chip: spmm %.2f ms   tn %.2f ms   copy(10 GB -> 10 GB) %.2f ms' % (t(spmm), t(tn), t(copy)), flush=True)
    full = (1 << 256) - 1
    k = a.gemm_cus
    for name, mg in (('first %d CUs' % k, mask_first(k)), ('every %d-th CU' % (256 // k), mask_strided(k))):
        sg, sm = masked_stream(mg), masked_stream(full & ~mg)
        with torch.cuda.stream(sg):
            t_tn = t(tn)
        with torch.cuda.stream(sm):
            t_sp, t_cp = t(spmm), t(copy)
        print('[%s] alone on their CU sets: tn (%d CUs) %.2f ms   spmm (%d CUs) %.2f ms   copy %.2f ms' % (name, k, t_tn, 256 - k, t_sp, t_cp), flush=True)
        for label, fn, reps in (('spmm', spmm, 3), ('copy', copy, 6)):
            def both():
                ev = torch.cuda.Event()
                ev.record()
                with torch.cuda.stream(sg):
                    sg.wait_event(ev)
                    tn()
                    e2 = torch.cuda.Event()
                    e2.record()
                with torch.cuda.stream(sm):
                    sm.wait_event(ev)
                    for _ in range(reps):
                        fn()
                    e3 = torch.cuda.Event()
                    e3.record()
                torch.cuda.current_stream().wait_event(e2)
                torch.cuda.current_stream().wait_event(e3)
            print('   tn || %d x %s on disjoint CU sets: %.2f ms' % (reps, label, t(both)), flush=True)
    # elementwise pass on a SMALL CU set beside the GEMM on the rest (the chain's trunk backward || the weight gradient of the same layer)
    for k in (16, 32, 48):
        se, sg = masked_stream(mask_first(k)), masked_stream(full & ~mask_first(k))
        with torch.cuda.stream(se):
            t_cp = t(copy)
        with torch.cuda.stream(sg):
            t_tn = t(tn)

        def pair():
            ev = torch.cuda.Event()
            ev.record()
            with torch.cuda.stream(sg):
                sg.wait_event(ev)
                tn()
                e2 = torch.cuda.Event()
                e2.record()
            with torch.cuda.stream(se):
                se.wait_event(ev)
                copy()
                e3 = torch.cuda.Event()
                e3.record()
            torch.cuda.current_stream().wait_event(e2)
            torch.cuda.current_stream().wait_event(e3)
        print('[copy on the first %d CUs, tn on the other %d] alone: copy %.2f ms  tn %.2f ms;  together %.2f ms  (one after the other on the whole chip: see first line)'
              % (k, 256 - k, t_cp, t_tn, t(pair)), flush=True)
    # plain streams, no masks
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def both_plain():
        ev = torch.cuda.Event()
        ev.record()
        with torch.cuda.stream(s1):
            s1.wait_event(ev)
            tn()
            e2 = torch.cuda.Event()
            e2.record()
        with torch.cuda.stream(s2):
            s2.wait_event(ev)
            for _ in range(3):
                spmm()
            e3 = torch.cuda.Event()
            e3.record()
        torch.cuda.current_stream().wait_event(e2)
        torch.cuda.current_stream().wait_event(e3)
    print('plain streams: tn || 3 x spmm %.2f ms' % t(both_plain), flush=True)


if __name__ == '__main__':
    main()
