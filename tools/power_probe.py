"""Board power and shader clock while ONE kernel family runs back to back for a few seconds (rocm-smi sampled from a side
thread): is the three-limb GEMM at the power limit?  usage: python tools/power_probe.py [--seconds 4]
Workloads: idle, aggregation (S-pl10M, d = 256), GEMM NN / TN (10M x 256 x 256, three-limb), fp32-input MFMA GEMM, HBM copy."""
import argparse
import os
import re
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def sample(stop, out):
    while not stop.is_set():
        try:
            txt = subprocess.run(['rocm-smi', '-d', '0', '--showpower', '--showclocks'], capture_output=True, text=True, timeout=5).stdout
        except Exception as e:          # noqa: BLE001
            out.append(('err', str(e)))
            return
        pw = re.search(r'Power \(W\):\s*([\d.]+)', txt)
        sclk = re.search(r'sclk clock level:.*\((\d+)Mhz\)', txt)
        mclk = re.search(r'mclk clock level:.*\((\d+)Mhz\)', txt)
        out.append((float(pw.group(1)) if pw else None, int(sclk.group(1)) if sclk else None, int(mclk.group(1)) if mclk else None))
        time.sleep(0.05)


def run(name, fn, seconds, iters_flops=None):
    fn()
    torch.cuda.synchronize()
    stop, out = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, out))
    th.start()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        n += 10
    dt = time.perf_counter() - t0
    stop.set()
    th.join()
    good = [s for s in out if s[0] not in (None, 'err')]
    tail = good[len(good) // 3:] or good       # skip the ramp
    pw = [s[0] for s in tail]
    sc = [s[1] for s in tail if s[1]]
    print(f'{name:34s} {dt / max(n, 1) * 1e3:8.3f} ms/launch   power avg {sum(pw) / max(len(pw), 1):7.1f} W  max {max(pw, default=0):7.1f} W   '
          f'sclk avg {sum(sc) / max(len(sc), 1):6.0f} MHz  min {min(sc, default=0)}   ({len(good)} samples)', flush=True)
    if not good and out:
        print('   sampler:', out[:2])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seconds', type=float, default=4.0)
    a = ap.parse_args()
    from gnn_tail_generalization_amd import gemm
    from gnn_tail_generalization_amd.data import synthetic_data
    from gnn_tail_generalization_amd.graph import CSRGraph
    dev = torch.device('cuda:0')
    print(subprocess.run(['rocm-smi', '-d', '0', '--showmaxpower'], capture_output=True, text=True).stdout.strip().splitlines()[-3:])
    run('idle (sync only)', lambda: None, 1.5)
    M = 10_000_000
    x = torch.randn(M, 256, device=dev)
    w = torch.randn(256, 256, device=dev) * 0.05
    y = torch.randn(M, 256, device=dev)
    run('GEMM NN three-limb 10Mx256x256', lambda: gemm.mm_nn(x, w), a.seconds)
    run('GEMM TN three-limb 256x10Mx256', lambda: gemm.mm_tn(x, y), a.seconds)
    os.environ['CB_GEMM_PLAIN_F32'] = '1'
    # the switch is read once per process: time the fp32-input kernel in a child
    code = ("import sys,os,torch; sys.path.insert(0, %r); sys.argv=['x'];\n"
            "import tools.power_probe as pp\n"
            "from gnn_tail_generalization_amd import gemm\n"
            "x=torch.randn(10_000_000,256,device='cuda:0'); w=torch.randn(256,256,device='cuda:0')*0.05\n"
            "pp.run('GEMM NN fp32-input MFMA', lambda: gemm.mm_nn(x, w), %f)\n") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), a.seconds)
    del y
    torch.cuda.empty_cache()
    z = torch.empty_like(x)
    run('HBM copy 10 GB -> 10 GB', lambda: z.copy_(x), a.seconds)
    del z
    data = synthetic_data('S-pl10M', seed=0, device=dev)
    G = CSRGraph(data.edge_index, data.x.shape[0])
    del data
    out = torch.empty_like(x)
    bias = torch.rand(256, device=dev)
    run('aggregation S-pl10M d=256', lambda: G.spmm(x, row_scale=G.norm_in, bias=bias, relu=True, out=out), a.seconds)
    del G, out, x
    torch.cuda.empty_cache()
    print(subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, env=dict(os.environ)).stdout.strip())


if __name__ == '__main__':
    main()
