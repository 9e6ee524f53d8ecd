"""Turns a rocprofv3 (rocpd sqlite) result into a compact markdown kernel table.
usage: python tools/prof_summary.py <results.db> <out.md> "<title / command>" """
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(.*', '', name)                      # drop the argument list
    name = re.sub(r'^void ', '', name)
    return name[:96]


def main():
    db, out, title = sys.argv[1], sys.argv[2], sys.argv[3]
    con = sqlite3.connect(db)
    rows = list(con.execute('select name,total_calls,total_duration,average,percentage from top_kernels'))
    unit_div = 1000.0 if rows and rows[0][2] > 1e7 else 1.0   # rocpd reports ns; keep the table in us
    with open(out, 'w') as f:
        f.write(f'# rocprofv3 --kernel-trace --stats\n\n{title}\n\n| kernel | calls | total_us | avg_us | % |\n|---|---|---|---|---|\n')
        for n, c, t, a, p in rows[:22]:
            f.write(f'| `{short(n)}` | {c} | {t / unit_div:.0f} | {a / unit_div:.1f} | {p:.2f} |\n')
    print(open(out).read())


if __name__ == '__main__':
    main()
