"""Per-dispatch kernel sequence of the LAST training step in a rocprofv3 (rocpd sqlite) result: name, duration, gap to the
previous kernel, grid.  usage: python tools/step_trace.py <results.db> [n_last_kernels]"""
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r'\(.*', '', n).replace('void ', '').replace('cb::', '')
    return n[:64]


def main():
    con = sqlite3.connect(sys.argv[1])
    rows = list(con.execute('select name,start,end,grid_x,grid_y from kernels order by start'))
    # the last step = everything after the second-to-last optimizer launch
    adam = [i for i, r in enumerate(rows) if 'k_adam' in r[0]]
    n_last = int(sys.argv[2]) if len(sys.argv) > 2 else None
    if n_last:
        seq = rows[-n_last:]
    else:
        ends = [i for j, i in enumerate(adam) if j + 1 == len(adam) or adam[j + 1] != i + 1]   # last launch of each optimizer burst
        seq = rows[ends[-2] + 1: ends[-1] + 1] if len(ends) >= 2 else rows
    prev, total, busy = None, 0.0, 0.0
    for n, s, e, gx, gy in seq:
        gap = (s - prev) / 1000 if prev else 0.0
        print(f'{short(n):66s} {(e - s) / 1000:9.1f} us   gap {gap:7.1f}   grid {gx}x{gy}')
        busy += (e - s) / 1000
        prev = e
    total = (seq[-1][2] - seq[0][1]) / 1000
    print(f'-- {len(seq)} kernels, span {total / 1000:.2f} ms, kernel time {busy / 1000:.2f} ms, idle {100 * (1 - busy / total):.1f} %')


if __name__ == '__main__':
    main()
