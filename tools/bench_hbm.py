"""HBM streaming rates of this GPU with plain device ops on 10 GB buffers (read-only, write-only, copy): the ceilings the
HBM-bound kernels of the step are measured against.  usage: python tools/bench_hbm.py"""
import torch


def t(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    ev[0].record()
    for i in range(iters):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    return min(ev[i].elapsed_time(ev[i + 1]) for i in range(iters))


def main():
    n = 10_000_000 * 256
    a = torch.rand(n, device='cuda:0')
    b = torch.empty_like(a)
    gb = n * 4 / 1e9
    ms = t(lambda: a.sum())
    print(f'read  {gb:.1f} GB (sum)      : {ms:7.3f} ms  {gb / ms:6.2f} TB/s')
    ms = t(lambda: b.fill_(1.0))
    print(f'write {gb:.1f} GB (fill)     : {ms:7.3f} ms  {gb / ms:6.2f} TB/s')
    ms = t(lambda: b.copy_(a))
    print(f'copy  {gb:.1f} + {gb:.1f} GB       : {ms:7.3f} ms  {2 * gb / ms:6.2f} TB/s')
    ms = t(lambda: torch.add(a, b, out=b))
    print(f'2 reads + 1 write ({3 * gb:.1f} GB): {ms:7.3f} ms  {3 * gb / ms:6.2f} TB/s')


if __name__ == '__main__':
    main()
