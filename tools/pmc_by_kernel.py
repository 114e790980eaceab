#!/usr/bin/env python3
"""Sum a rocprofv3 --pmc counter per kernel name:  pmc_by_kernel.py <output dir> <COUNTER> [steps divisor] [name regex] [max MB per launch: drops set-up copies]

FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE is doubled (gfx950 tallies 128-B read requests at 64 B:
/opt/skills/guides/MI355X_MICROARCH.md "HBM")."""
import collections
import csv
import glob
import os
import re
import sys


def main():
    d, counter = sys.argv[1], sys.argv[2]
    steps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
    keep = re.compile(sys.argv[4]) if len(sys.argv) > 4 else None
    max_mb = float(sys.argv[5]) if len(sys.argv) > 5 else None
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {d}")
    tot, cnt = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(files[0])):
        if r["Counter_Name"] != counter:
            continue
        full = r["Kernel_Name"]
        if keep is not None and not keep.search(full):
            continue
        m = re.search(r"(\w+_kernel)\b", full.replace("(anonymous namespace)::", ""))
        name = (m.group(1) + re.sub(r".*?_kernel", "", full.replace("(anonymous namespace)::", ""), count=1).split("(")[0])[:70] if m else full.split("(")[0][-70:]
        if max_mb is not None and float(r["Counter_Value"]) * 1024.0 / 1e6 > max_mb:
            continue
        tot[name] += float(r["Counter_Value"])
        cnt[name] += 1
    scale = 1024.0 * (2.0 if counter == "FETCH_SIZE" else 1.0)
    grand = 0.0
    for name, v in tot.most_common():
        b = v * scale / steps
        grand += b
        print(f"{counter:10s} {b / 1e6:10.2f} MB/step  {cnt[name] / steps:7.1f} launches/step  {name}")
    print(f"{counter:10s} {grand / 1e6:10.2f} MB/step  TOTAL" + ("  (raw x2: gfx950 correction)" if counter == "FETCH_SIZE" else ""))


if __name__ == "__main__":
    main()
