"""The per-config table of profiles/README.md: kernel, average duration, algorithmic bytes, PMC bytes and their ratio for the
two kernels of a diffusion step, from profiles/<tag>_pmc.json and <tag>_kernel_stats[_config].md.
usage: python tools/profile_table.py r04"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFGS = [("metric", 1024, 17, 50, False, 1), ("hopper512", 512, 3, 50, False, 1), ("halfcheetah1024", 1024, 6, 50, False, 1),
        ("humanoidrun4096", 4096, 17, 50, False, 1), ("humanoidtrack2048demo", 2048, 17, 50, True, 1),
        ("humanoidrun8192", 8192, 17, 50, False, 1), ("sweep8", 1024, 17, 50, False, 8)]


def main():
    tag = sys.argv[1]
    pm = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_pmc.json")))
    print("| config | rollout kernel | avg µs | B_alg MB | PMC MB | ratio | score kernel | avg µs | alg. MB | PMC MB | ratio |")
    print("|---|---|---:|---:|---:|---:|---|---:|---:|---:|---:|")
    for c, N, Nu, H, demo, P in CFGS:
        if c not in pm:
            continue
        balg = P * (4 * (2 * N * H * Nu + 2 * N + 2 * H * Nu) + (8 * N if demo else 0))
        salg = P * 4 * (N * H * Nu + 2 * N + 2 * H * Nu)                                   # one read of the normals + vectors
        d = pm[c]
        rk = [k for k in d if "rollout_" in k and "hbm_bytes_per_launch_corrected" in d[k]][0]
        sk = [k for k in d if "score_wmean" in k][0]
        st = open(os.path.join(ROOT, "profiles", f"{tag}_kernel_stats{'' if c == 'metric' else '_' + c}.md")).read()

        def us(pat):
            for line in st.split("\n"):
                q = [x.strip() for x in line.split("|")]
                if pat in line and len(q) > 5 and q[2].isdigit():
                    return float(q[4])
            return float("nan")
        rb, sb = d[rk]["hbm_bytes_per_launch_corrected"], d[sk]["hbm_bytes_per_launch_corrected"]
        print(f"| `{c}` | `{rk.split('(')[0].replace('void mbd::', '')[:48]}` | {us('rollout_'):.1f} | {balg / 1e6:.2f} | "
              f"{rb / 1e6:.2f} | {rb / balg:.2f} | `{sk.split('(')[0].replace('mbd::', '')}` | {us('score_wmean'):.1f} | "
              f"{salg / 1e6:.2f} | {sb / 1e6:.2f} | {sb / salg:.2f} |")


if __name__ == "__main__":
    main()
