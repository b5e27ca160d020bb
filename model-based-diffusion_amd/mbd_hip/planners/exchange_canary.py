"""Canary for the in-library exchange (include/mbd_hip.h "in-library exchange") on a multi-GPU node.

    python -m mbd_hip.planners.exchange_canary RANK WORLD DEVICE RENDEZVOUS_DIR

One short-lived process per rank, started by the caller BESIDE its real process (bench.py does): the canaries exchange
their window handles through files in RENDEZVOUS_DIR, map each other's windows, run a few all-gathers with known values
and check them.  What this buys: mapping peer memory and storing into it from a kernel is exactly the kind of thing that
ends in a GPU memory fault — which kills the process, uncatchably — when a node's peer access is not what the code
assumes.  A fault here kills the canary; the caller sees a non-zero exit status and keeps its process group's
all-gather.  Exit status 0: every step delivered every rank's values."""
import os
import sys
import time


def main():
    rank, world, device, rdv = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import ctypes as C
    import torch
    from mbd_hip import _capi
    lib = _capi.load()
    torch.cuda.set_device(device)
    rows, shard = 2, 96
    h = C.c_void_p()
    _capi.check(lib.mbd_exchange_create(device, rank, world, rows, shard, C.byref(h)))
    mine = (C.c_ubyte * 64)()
    _capi.check(lib.mbd_exchange_local_handle(h, mine))
    os.makedirs(rdv, exist_ok=True)
    tmp = os.path.join(rdv, f"h{rank}.tmp")
    with open(tmp, "wb") as f:
        f.write(bytes(mine))
    os.replace(tmp, os.path.join(rdv, f"h{rank}.bin"))
    t0 = time.time()
    handles = []
    for r in range(world):
        p = os.path.join(rdv, f"h{r}.bin")
        while not os.path.exists(p):
            if time.time() - t0 > 90:
                print(f"canary {rank}: rank {r} never published its handle", file=sys.stderr)
                sys.exit(3)
            time.sleep(0.02)
        with open(p, "rb") as f:
            handles.append(f.read())
    blob = (C.c_ubyte * (64 * world)).from_buffer_copy(b"".join(handles))
    _capi.check(lib.mbd_exchange_connect(h, blob))
    # everybody has mapped everybody before anybody pushes (a second round of files)
    open(os.path.join(rdv, f"m{rank}.ok"), "w").close()
    for r in range(world):
        while not os.path.exists(os.path.join(rdv, f"m{r}.ok")):
            if time.time() - t0 > 120:
                sys.exit(4)
            time.sleep(0.02)
    stream = torch.cuda.current_stream().cuda_stream
    out = C.c_void_p()
    ok = True
    for step in range(4):
        local = (torch.arange(rows * shard, dtype=torch.float32, device="cuda").reshape(rows, shard)
                 + 1000.0 * rank + 100000.0 * step).contiguous()
        _capi.check(lib.mbd_exchange_all_gather(h, local.data_ptr(), C.byref(out), stream))
        got = torch.empty(rows * world * shard, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        C.cdll.LoadLibrary("libamdhip64.so").hipMemcpy(C.c_void_p(got.data_ptr()), out, 4 * rows * world * shard, 3)
        got = got.reshape(rows, world, shard).cpu()
        for r in range(world):
            want = (torch.arange(rows * shard, dtype=torch.float32).reshape(rows, shard) + 1000.0 * r + 100000.0 * step)
            ok = ok and bool(torch.equal(got[:, r, :], want))
    try:
        _capi.check(lib.mbd_exchange_status(h))
    except _capi.MbdError as e:
        print(f"canary {rank}: {e}", file=sys.stderr)
        ok = False
    # keep the window alive until every rank has finished reading it
    open(os.path.join(rdv, f"d{rank}.ok"), "w").close()
    for r in range(world):
        while not os.path.exists(os.path.join(rdv, f"d{r}.ok")):
            if time.time() - t0 > 150:
                break
            time.sleep(0.02)
    lib.mbd_exchange_destroy(h)
    sys.exit(0 if ok else 5)


if __name__ == "__main__":
    main()
