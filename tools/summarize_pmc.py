"""Aggregate rocprofv3 CSV output (counter_collection.csv of --pmc passes, kernel_stats.csv of --kernel-trace --stats) per kernel.
usage: summarize_pmc.py <dir with pmc_* / *_trace sub-dirs> [name filter regex]"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

root = sys.argv[1]
flt = re.compile(sys.argv[2] if len(sys.argv) > 2 else r"gemm_bf16|attn_fwd|ascore|cscore|layernorm|splitk|ln_stats|conv3x3_halo|groupnorm")


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)[:60]


def counters(d):
    out = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(root, d, "**", "*counter_collection.csv"), recursive=True):
        disp = defaultdict(dict)
        for r in csv.DictReader(open(f)):
            if not flt.search(r["Kernel_Name"]):
                continue
            key = (r["Dispatch_Id"], short(r["Kernel_Name"]), r["Grid_Size"])
            disp[key][r["Counter_Name"]] = disp[key].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            disp[key]["_dur"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        for (_, k, g), c in disp.items():
            for n, v in c.items():
                out[(k, g)][n].append(v)
    return out


def stats(d):
    rows = []
    for f in glob.glob(os.path.join(root, d, "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((short(r["Name"]), int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
    return sorted(rows, key=lambda x: -x[2])


for d in sorted(os.listdir(root)):
    p = os.path.join(root, d)
    if not os.path.isdir(p):
        continue
    st = stats(d)
    if st:
        print(f"\n## {d}: kernel stats\n| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|")
        for n, c, t, a, pc in st[:16]:
            print(f"| `{n}` | {c} | {t:.2f} | {a:.1f} | {pc:.1f} |")
    cs = counters(d)
    if cs:
        names = sorted({n for v in cs.values() for n in v if n != "_dur"})
        print(f"\n## {d}: counters (mean per launch)\n| kernel | grid | launches | us | " + " | ".join(names) + " |\n|---|---|---|---|" + "---|" * len(names))
        for (k, g), v in sorted(cs.items()):
            mean = lambda xs: sum(xs) / len(xs)
            print(f"| `{k}` | {g} | {len(v['_dur'])} | {mean(v['_dur']):.1f} | " + " | ".join(f"{mean(v[n]):.4g}" if n in v else "-" for n in names) + " |")
