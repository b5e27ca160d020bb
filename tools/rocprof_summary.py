"""Summarise rocprofv3 (rocpd sqlite) outputs under gpurun_out/ into small tracked files in profiles/.

usage: python tools/rocprof_summary.py <round-tag> [--config NAME] <stats_db> [<pmc_db> ...]
Writes profiles/<tag>_kernel_stats[_NAME].md and merges the per-kernel counter averages per launch into
profiles/<tag>_pmc.json under the key NAME (the bench config the profiled command ran; default "metric").
HBM bytes follow MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are in KiB and on gfx950
FETCH_SIZE reports half of a wide coalesced read stream (doubled here; the raw value is kept too).
"""
import json
import os
import sqlite3
import sys


def main():
    argv = sys.argv[1:]
    config = "metric"
    if "--config" in argv:
        k = argv.index("--config")
        config = argv[k + 1]
        del argv[k:k + 2]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out_dir = os.path.join(root, "profiles")
    if "--out" in argv:  # (on the GPU box: summarise next to the databases, only the summaries travel back)
        k = argv.index("--out")
        out_dir = argv[k + 1]
        del argv[k:k + 2]
    tag, stats_db, pmc_dbs = argv[0], argv[1], argv[2:]
    os.makedirs(out_dir, exist_ok=True)
    con = sqlite3.connect(stats_db)
    rows = list(con.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    lines = [f"# rocprofv3 --kernel-trace --stats ({tag})", "",
             f"command: `rocprofv3 --kernel-trace --stats -- python bench.py --config {config} --steps 60 --warmup 10 "
             "--no-cpu-baseline --no-final-reward --no-extras --repeats 2` (durations in microseconds)", "",
             "| kernel | calls | total_us | avg_us | % |", "|---|---:|---:|---:|---:|"]
    for name, calls, tot, avg, pct in rows:
        lines.append(f"| `{name[:110]}` | {calls} | {tot:.1f} | {avg:.2f} | {pct:.2f} |")
    kinfo = list(con.execute("select name, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, scratch_size, "
                             "grid_x, workgroup_x, count(*) from kernels group by name"))
    lines += ["", "| kernel | vgpr | agpr | sgpr | lds | scratch | grid_x | wg_x | dispatches |",
              "|---|---:|---:|---:|---:|---:|---:|---:|---:|"]
    for r in kinfo:
        lines.append("| `" + str(r[0])[:80] + "` | " + " | ".join(str(x) for x in r[1:]) + " |")
    suffix = "" if config == "metric" else f"_{config}"
    with open(os.path.join(out_dir, f"{tag}_kernel_stats{suffix}.md"), "w") as f:
        f.write("\n".join(lines) + "\n")
    pmc = {}
    for db in pmc_dbs:
        c = sqlite3.connect(db)
        for kname, cname, n, avg in c.execute("select kernel_name, counter_name, count(*), avg(value) from "
                                              "counters_collection group by kernel_name, counter_name"):
            pmc.setdefault(kname, {})[cname] = {"launches": n, "avg_per_launch": avg}
    for kname, d in pmc.items():
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            f_kib, w_kib = d["FETCH_SIZE"]["avg_per_launch"], d["WRITE_SIZE"]["avg_per_launch"]
            d["hbm_bytes_per_launch_corrected"] = (2.0 * f_kib + w_kib) * 1024.0
            d["hbm_bytes_per_launch_raw"] = (f_kib + w_kib) * 1024.0
    path = os.path.join(out_dir, f"{tag}_pmc.json")
    merged = {}
    if os.path.exists(path):
        with open(path) as f:
            merged = json.load(f)
    if pmc:
        merged[config] = pmc
    with open(path, "w") as f:
        json.dump(merged, f, indent=1, sort_keys=True)
    print("\n".join(lines[:14]))
    for k, d in pmc.items():
        if "rollout" in k:
            print(json.dumps({a: (b["avg_per_launch"] if isinstance(b, dict) else b) for a, b in d.items()}))


if __name__ == "__main__":
    main()
