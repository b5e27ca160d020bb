"""The build-time assembly pass (tools/fix_straddles.py) on a small kernel: needs hipcc / the LLVM tools, no GPU."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

SRC = r"""
#include <hip/hip_runtime.h>
namespace mbd {
struct RolloutParams { float* x; int n; };
__global__ void rollout_kernel(RolloutParams P) {
  float a = P.x[threadIdx.x], b = a * 1.5f, c = a - 2.0f;
  for (int i = 0; i < P.n; ++i) {
    float2 u = make_float2(a, b), v = make_float2(b, c);
    u.x = __builtin_fmaf(u.x, v.x, c); u.y = __builtin_fmaf(u.y, v.y, a);
    a = u.x * 0.999f + b; b = u.y - c * 0.5f; c = __builtin_fmaf(a, b, c) * 0.25f;
    a = a > 1e3f ? a * 0.001f : a; b = __builtin_fabsf(b) + 0.125f;
  }
  P.x[threadIdx.x] = a + b + c;
}
__global__ void other_kernel(float* x) { x[threadIdx.x] *= 2.0f; }
}  // namespace mbd
"""


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_fix_straddles_changes_encodings_only(tmp_path):
    import fix_straddles
    src, asm = tmp_path / "k.hip", tmp_path / "k.s"
    src.write_text(SRC)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "--cuda-device-only", "-S",
                    str(src), "-o", str(asm)], check=True, capture_output=True)
    text = asm.read_text()
    fixed, stats = fix_straddles.fix(text)
    assert len(stats) == 1 and "rollout_kernel" in next(iter(stats))  # only the rollout kernels are touched
    st = next(iter(stats.values()))
    assert st["straddles_after"] <= st["straddles_before"] and st["straddles_after"] == st["left"]
    a, b = text.split("\n"), fixed.split("\n")
    assert len(a) == len(b)
    changed = [(x, y) for x, y in zip(a, b) if x != y]
    assert len(changed) == st["promoted"]
    for x, y in changed:  # the same instruction, operands untouched, VOP1/2/C re-encoded as VOP3
        assert re.sub(r"_e32\b", "_e64", x, count=1) == y and x.strip().startswith("v_")
    # and it still assembles (fix() has already checked that the sizes are the predicted ones)
    out = tmp_path / "k.o"
    (tmp_path / "f.s").write_text(fixed)
    subprocess.run(["/opt/rocm/lib/llvm/bin/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c",
                    str(tmp_path / "f.s"), "-o", str(out)], check=True, capture_output=True)


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_measurement_tools_find_the_substep_loop_of_a_product_kernel():
    """tools/count_flops.py (the `issued` side of the bench's valu block, profiles/r02_static_flops.json) and
    tools/tune_phase.py (code placement) both locate the substep loop in the emitted ISA of a real instantiation — the
    cartpole's planar kernel, the smallest one: same loop, no scratch, no memory access inside it."""
    import count_flops
    import tune_phase
    targs = count_flops.INSTANCES["cartpole_planar"]
    res, hist = count_flops.count(targs)
    assert res["scratch_bytes"] == 0 and 200 < res["instructions_per_substep"] < 400
    assert res["fp32_flops_per_lane_substep"] > res["valu_per_substep"]  # (packed and fused ops count 2 and 4)
    assert not any(k.startswith(("global_", "flat_", "scratch_", "buffer_")) for k in hist)
    kind, t, _ = next(x for x in tune_phase.TARGETS if x[0] == "planar" and "planar:" + x[1] == targs)
    ph = tune_phase.analyse(kind, t)
    unroll = int(t.split(",")[6]) // 2
    assert abs(ph["instructions"] - unroll * res["instructions_per_substep"]) <= unroll + 8  # (+ the loop's own bookkeeping)
    assert set(ph["straddles"]) == set(range(0, 32, 4)) and ph["straddles"][ph["shift"]] == min(ph["straddles"].values())
