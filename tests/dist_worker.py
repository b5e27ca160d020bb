"""Worker of tests/test_gpu_parity.py::test_two_real_ranks_*: launched by torch.distributed.run with
--nproc-per-node 2 on a ONE-GPU box (backend gloo, both ranks on device 0).  Every rank runs the PRODUCT's sharded
path — run_diffusion -> reverse_distributed: mbd_plan_sample_rollout on its shard, the per-step all-gather,
mbd_plan_score_update — with real HIP kernels, then the unsharded plan (mbd_plan_run) in the same process, and
writes what it got; the test asserts every rank's mu_0ts / mean rewards / final reward equal the unsharded ones
bit for bit.  usage: dist_worker.py OUT_DIR ENV N H ND TEMP DEMO"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "model-based-diffusion_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    out, env_name, N, H, Nd, temp, demo = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), \
        int(sys.argv[5]), float(sys.argv[6]), bool(int(sys.argv[7]))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    import __graft_entry__
    __graft_entry__.build()
    from mbd_hip.planners.mbd_planner import Args, run_diffusion
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    assert world >= 2
    a = Args(seed=3, env_name=env_name, Nsample=N, Hsample=H, Ndiffuse=Nd, temp_sample=temp, enable_demo=demo,
             disable_recommended_params=True, not_render=True)
    rew, det = run_diffusion(Args(**vars(a)), device=0, return_details=True)   # sharded over the ranks
    assert det["sharded"] and det["world"] == world
    rew1, det1 = run_diffusion(Args(**vars(a)), device=0, return_details=True, force_single=True)  # unsharded
    ok = bool(np.array_equal(det["mu_0ts"], det1["mu_0ts"]) and np.array_equal(det["rew_means"], det1["rew_means"])
              and np.float32(rew) == np.float32(rew1))
    # round-2 advice: an UNSHARDED plan stepped through the Python loop (progress callback) under an initialised
    # multi-rank group must not touch the group — with a different seed per rank a stray all-gather would score rank
    # 0's rewards on every rank (or hang)
    b = Args(**{**vars(a), "seed": 11 + rank, "Ndiffuse": min(Nd, 6)})
    seen = []
    rew2, det2 = run_diffusion(Args(**vars(b)), device=0, return_details=True, force_single=True,
                               progress=lambda i, r: seen.append(r))
    rew3, det3 = run_diffusion(Args(**vars(b)), device=0, return_details=True, force_single=True)
    ok_single = bool(np.array_equal(det2["mu_0ts"], det3["mu_0ts"]) and np.float32(rew2) == np.float32(rew3)
                     and len(seen) == b.Ndiffuse - 1 and np.array_equal(np.float32(seen), det3["rew_means"]))
    np.save(os.path.join(out, f"mu_rank{rank}.npy"), det["mu_0ts"])
    with open(os.path.join(out, f"rank{rank}.json"), "w") as f:
        json.dump({"rank": rank, "world": world, "equal_to_unsharded": ok, "force_single_progress_ok": ok_single, "rew": float(rew), "rew_unsharded": float(rew1),
                   "steps_per_sec_sharded": det["steps_per_sec"], "steps_per_sec_unsharded": det1["steps_per_sec"],
                   "phase_ms": det.get("phase_ms")}, f)
    dist.barrier()
    dist.destroy_process_group()
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
