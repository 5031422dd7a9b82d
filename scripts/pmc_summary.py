"""Summarise rocprofv3 --pmc counter_collection CSVs: per (kernel, counter) mean value per dispatch."""
import csv
import json
import re
import sys
from collections import defaultdict

out = {}
for path in sys.argv[1:]:
    acc = defaultdict(lambda: [0, 0.0, 0.0])
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", name)  # a kernel rocprofv3 left mangled (templates with _Float16 arguments: the kNN filter)
        if m:
            k0 = m.end()
            name = "(anonymous namespace)::" + name[k0:k0 + int(m.group(1))] + "("
        if "at::native" in name or "rocprim" in name or "anonymous namespace)::" not in name:
            continue
        short = re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0].replace("void ", "")
        a = acc[(short, r["Counter_Name"])]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
        a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    for (k, c), (n, v, ms) in sorted(acc.items()):
        out.setdefault(k, {})[c] = {"dispatches": n, "mean": v / n, "mean_ms": ms / n}
print(json.dumps(out, indent=1))
