"""Per-kernel sums of a rocprofv3 --pmc counter_collection.csv (values averaged over the dispatches of each kernel).
usage: python tools/pmc_summary.py <counter_collection.csv> [name-substring ...]"""
import csv
import re
import sys
from collections import defaultdict


def main():
    path, filt = sys.argv[1], sys.argv[2:]
    acc = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(set)
    for r in csv.DictReader(open(path)):
        name = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '')[:70]
        if filt and not any(f in name for f in filt):
            continue
        acc[name][r['Counter_Name']] += float(r['Counter_Value'])
        disp[name].add(r['Dispatch_Id'])
    for name, cs in acc.items():
        n = len(disp[name])
        print(f'{name}  ({n} dispatches; per-dispatch means)')
        for c, v in sorted(cs.items()):
            print(f'    {c:28s} {v / n:16.0f}')


if __name__ == '__main__':
    main()
