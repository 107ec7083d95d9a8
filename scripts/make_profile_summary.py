#!/usr/bin/env python
"""Turn the rocprofv3 CSVs of scripts/profile_round.sh (gpurun_out/<tag>/{kt,fetch,write}) into the tracked evidence under profiles/:
  profiles/<round>_kernel_stats.md     per-kernel table of the --kernel-trace --stats pass (bench.py defaults)
  profiles/<round>_hbm_traffic.json    FETCH_SIZE / WRITE_SIZE per launch of every mrs:: kernel (separate --pmc passes);
                                       read side doubled as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950
    python scripts/make_profile_summary.py gpurun_out/final round1
"""
import collections, csv, glob, json, os, subprocess, sys


def counters(d, name):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    out = collections.defaultdict(list)
    if not f:
        return out
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == name:
            out[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return out


def build_identity(src, root):
    """source digest of the PROFILED tree (written on the GPU box by scripts/profile_round.sh: bench.source_digest()) and, when the tree here still has that digest, the
    commit it belongs to -- bench.py quotes a committed profile only when the digest matches the tree it runs from."""
    dig = None
    try:
        dig = open(os.path.join(src, "source_digest.txt")).read().strip() or None
    except OSError:
        pass
    sys.path.insert(0, root)
    import bench
    here = bench.source_digest()
    commit = None
    if dig is None:
        dig = here
    if dig == here:
        try:
            commit = subprocess.run(["git", "-C", root, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip() or None
            dirty = subprocess.run(["git", "-C", root, "status", "--porcelain", "--", "mistral.rs_amd/csrc", "include"], capture_output=True, text=True).stdout.strip()
            if dirty:
                commit = (commit or "") + "+uncommitted"
        except OSError:
            pass
    return dig, commit


def main():
    src, tag = sys.argv[1], sys.argv[2]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    digest, commit = build_identity(src, root)
    kt = glob.glob(os.path.join(src, "kt", "**", "*kernel_trace.csv"), recursive=True)[0]
    table = subprocess.run([sys.executable, os.path.join(root, "scripts", "rocprof_summary.py"), kt, "--top", "40", "--match", "mrs::"],
                           capture_output=True, text=True, check=True).stdout
    with open(os.path.join(root, "profiles", f"{tag}_kernel_stats.md"), "w") as f:
        f.write(f"# {tag}: `rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-extra` (MI355X, per-kernel, mrs:: kernels)\n\n")
        f.write(table)
    # per-kernel average durations of the same pass as JSON (bench.py reads the dominant kernel's IN-GRAPH time from it: inside a captured graph the start / end stamps
    # of consecutive kernels touch, so these durations include the launch boundary that the HIP-event figure of bench.py's `roofline` does not)
    stats = glob.glob(os.path.join(src, "kt", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        ks = {r["Name"]: {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "min_us": float(r["MinNs"]) / 1e3, "max_us": float(r["MaxNs"]) / 1e3}
              for r in csv.DictReader(open(stats[0])) if "mrs::" in r["Name"]}
        with open(os.path.join(root, "profiles", f"{tag}_kernel_stats.json"), "w") as f:
            json.dump({"source": "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-extra (decode graph replays + the prompt)",
                       "source_digest": digest, "commit": commit, "kernels": ks}, f, indent=1, sort_keys=True)
    fetch, write = counters(os.path.join(src, "fetch"), "FETCH_SIZE"), counters(os.path.join(src, "write"), "WRITE_SIZE")
    out = {}
    for k, v in fetch.items():
        if "mrs::" not in k:
            continue
        w = write.get(k, [0.0])
        out[k] = {"launches": len(v), "fetch_size_kb_avg": sum(v) / len(v), "read_bytes_per_launch": 2.0 * 1024.0 * sum(v) / len(v),
                  "write_size_kb_avg": sum(w) / len(w), "write_bytes_per_launch": 1024.0 * sum(w) / len(w)}
    meta = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- python bench.py --no-cpu-baseline --no-dropin --steps 32",
            "correction": "FETCH_SIZE is in KB and counts 64 B per 128-B request on gfx950 for wide coalesced reads: read bytes = 2 * 1024 * FETCH_SIZE",
            "source_digest": digest, "commit": commit, "kernels": out}
    with open(os.path.join(root, "profiles", f"{tag}_hbm_traffic.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["read_bytes_per_launch"] * kv[1]["launches"])[:12]:
        print(f"{v['launches']:6d}  read {v['read_bytes_per_launch'] / 1e6:9.2f} MB  write {v['write_bytes_per_launch'] / 1e6:8.2f} MB  {k[:100]}")


if __name__ == "__main__":
    main()
