"""Summarise rocprofv3 --pmc CSV output (counter_collection.csv): per kernel name, mean of each counter."""
import csv
import re
import sys
from collections import defaultdict
from pathlib import Path

rows = defaultdict(lambda: defaultdict(list))
for path in Path(sys.argv[1]).rglob("*counter_collection.csv"):
    with open(path) as fh:
        for r in csv.DictReader(fh):
            full = r["Kernel_Name"]
            m = re.search(r"(?:\)::|\s)(\w+)(<[^(]*>)?\(", full)
            name = (m.group(1) + (m.group(2) or "")) if m else full[:80]
            name = re.sub(r"\(anonymous namespace\)::|pe::", "", name)[:90]
            rows[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
want = sys.argv[2:] 
for name, cs in sorted(rows.items()):
    if want and not any(w in name for w in want):
        continue
    print(name)
    for c, v in sorted(cs.items()):
        print(f"   {c:32s} n={len(v):3d} mean={sum(v) / len(v):.5g}")
