#!/usr/bin/env python
"""Summarise a rocprofv3 run (rocpd sqlite .db or kernel_trace.csv) into a per-kernel table (markdown).

    python scripts/rocprof_summary.py <results.db | kernel_trace.csv> [--top N] [--match mrs::]
"""
import argparse, collections, csv, re, sqlite3, sys


def short(name, n=110):
    name = re.sub(r"\s+", " ", name)
    return name if len(name) <= n else name[: n - 3] + "..."


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("path")
    ap.add_argument("--top", type=int, default=25)
    ap.add_argument("--match", default="")
    a = ap.parse_args()
    rows = []  # (name, dur_ns, grid, wg, lds, vgpr)
    if a.path.endswith(".db"):
        c = sqlite3.connect(a.path)
        for r in c.execute("select name, duration, grid_x*grid_y*grid_z, workgroup_x*workgroup_y*workgroup_z, lds_size, vgpr_count from kernels"):
            rows.append(r)
    else:
        for r in csv.DictReader(open(a.path)):
            def dim(prefix):
                if prefix in r:
                    return int(r[prefix] or 0)
                return int(r.get(prefix + "_X", 0) or 0) * max(1, int(r.get(prefix + "_Y", 1) or 1)) * max(1, int(r.get(prefix + "_Z", 1) or 1))
            rows.append((r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), dim("Grid_Size"), dim("Workgroup_Size"),
                         int(r.get("LDS_Block_Size", 0) or 0), int(r.get("VGPR_Count", 0) or 0)))
    agg = collections.OrderedDict()
    for name, dur, grid, wg, lds, vgpr in rows:
        if a.match and a.match not in name:
            continue
        e = agg.setdefault(name, [0, 0, 10 ** 18, 0, grid, wg, lds, vgpr])
        e[0] += 1; e[1] += dur; e[2] = min(e[2], dur); e[3] = max(e[3], dur)
    total = sum(e[1] for e in agg.values()) or 1
    print("| kernel | calls | total ms | avg us | min us | max us | % | grid (threads) | wg | LDS B | VGPR |")
    print("|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|")
    for name, e in sorted(agg.items(), key=lambda kv: -kv[1][1])[: a.top]:
        print(f"| `{short(name)}` | {e[0]} | {e[1]/1e6:.3f} | {e[1]/e[0]/1e3:.2f} | {e[2]/1e3:.2f} | {e[3]/1e3:.2f} | {100*e[1]/total:.1f} | {e[4]} | {e[5]} | {e[6]} | {e[7]} |")


if __name__ == "__main__":
    main()
