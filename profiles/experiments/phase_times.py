"""Per-phase durations of the persistent step kernel launched phase by phase (MRS_DEC_PERSIST=2) from a rocprofv3 kernel trace CSV."""
import csv, sys, glob, collections
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = [r for r in csv.DictReader(open(f)) if "dec_step_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
L = int(sys.argv[2]) if len(sys.argv) > 2 else 32
np_ = 2 + 6 * L
steps = len(rows) // np_
rows = rows[-(steps - 1) * np_:] if steps > 1 else rows
names = ["qkv", "attn", "merge", "o", "gate_up", "down"]
acc, gaps = collections.defaultdict(list), []
for i, r in enumerate(rows):
    p = i % np_
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    key = "embed" if p == 0 else ("lm_head" if p == np_ - 1 else names[(p - 1) % 6])
    acc[key].append(d)
    if i: gaps.append((int(r["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"])) / 1e3)
tot = 0
for k in ["embed"] + names + ["lm_head"]:
    v = acc[k]
    per_step = sum(v) / max(1, len(rows) // np_)
    tot += per_step
    print(f"{k:8s} n={len(v):5d} avg {sum(v)/len(v):7.2f} us  min {min(v):7.2f}  max {max(v):7.2f}  per step {per_step:8.1f} us")
print(f"sum of kernel time per step {tot:.1f} us; mean gap between launches {sum(gaps)/len(gaps):.2f} us")
