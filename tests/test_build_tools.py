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
