"""Build a variant of libmbd_hip.so with extra compiler flags into model-based-diffusion_amd/lib/variants/ (the same build
path as the library: both translation units through the assembly pass) — for same-box A/B runs with tools/gpu_ab2.sh.
usage: python tools/build_variant.py NAME [--default-sched UNIT.hip ...] [-DFLAG ...]
  --default-sched UNIT.hip   build that translation unit without its own scheduler flags (A/B of the per-unit strategies)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

name, flags = sys.argv[1], sys.argv[2:]
while "--sched" in flags:  # --sched UNIT.hip=STRATEGY: that unit with -mllvm -amdgpu-sched-strategy=STRATEGY
    k = flags.index("--sched")
    unit, strat = flags[k + 1].split("=")
    del flags[k:k + 2]
    g.TUS = [(n, ["-mllvm", f"-amdgpu-sched-strategy={strat}"] if n == unit else f) for n, f in g.TUS]
while "--default-sched" in flags:
    k = flags.index("--default-sched")
    unit = flags[k + 1]
    del flags[k:k + 2]
    g.TUS = [(n, [] if n == unit else f) for n, f in g.TUS]
g.HIPCC_FLAGS = g.HIPCC_FLAGS + flags
out = os.path.join(g.PKG, "lib", "variants", f"libmbd_hip_{name}.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
sys.path.insert(0, os.path.join(ROOT, "tools"))
mode = g._build_lib(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), os.path.join(g.PKG, "csrc"), out)
print(f"{out}: {mode} build with {flags}")
