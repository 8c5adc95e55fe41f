#!/usr/bin/env python
"""Average PMC counter values per dispatch of the kernels matching a substring, from rocprofv3 --pmc result DBs.
    python tools/pmc_kernel.py <substring> <db> [<db> ...]"""
import sqlite3
import sys
from collections import defaultdict


def main():
    sub = sys.argv[1]
    for path in sys.argv[2:]:
        db = sqlite3.connect(path)
        agg = defaultdict(lambda: [0, 0.0])
        for name, ctr, val in db.execute("select kernel_name, counter_name, value from counters_collection"):
            if sub in name:
                agg[ctr][0] += 1
                agg[ctr][1] += float(val)
        for ctr, (n, tot) in sorted(agg.items()):
            print(f"{ctr:32s} dispatches {n:4d}  avg {tot / n:16.1f}")


if __name__ == "__main__":
    main()
