"""Worker of tests/test_gpu_parity.py::test_exchange_reports_a_peer_that_never_arrives: two processes on one GPU set up the
in-library exchange; rank 1 never pushes.  Rank 0's waits must run into their time limit ONCE (the flag is sticky: later
steps return at once), mbd_exchange_status must say so, and nothing hangs.  usage: exchange_timeout_worker.py OUT_DIR"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "model-based-diffusion_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    out = sys.argv[1]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    from mbd_hip import _capi
    from mbd_hip.planners.mbd_planner import P2PExchange
    dist.init_process_group("gloo")
    rank = dist.get_rank()
    x = P2PExchange(0, 1, 64)
    res = {"rank": rank}
    if rank == 0:
        local = torch.ones((1, 64), dtype=torch.float32, device="cuda")
        stream = torch.cuda.current_stream().cuda_stream
        t0 = time.time()
        for _ in range(4):
            x.all_gather(local, stream)
        try:
            x.status()
            res["error"] = None
        except _capi.MbdError as e:
            res["error"] = str(e)
        res["seconds"] = time.time() - t0
    dist.barrier()  # (rank 1 never pushed; it only keeps its window alive until rank 0 is done)
    x.close()
    with open(os.path.join(out, f"xt_rank{rank}.json"), "w") as f:
        json.dump(res, f)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
