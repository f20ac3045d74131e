"""Two ranks on GPU 0; rank 1 never joins the second exchange.  Rank 0's bounded wait must time out (no hang), the
error word must be raised, and a fused solve afterwards must fail loudly with MI_ERR_COMM."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from optimization_amd import capi
import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
c = capi.Context(0)
assert c.enable_peer_memory(world, rank, dist, force=True)
x = c.upload(np.ones(1000))
out = {"rank": rank, "first": x.dot(x)}          # both ranks: fine
dist.barrier()
if rank == 0:
    t0 = time.time()
    out["second"] = x.dot(x)                     # rank 1 is absent: bounded wait
    out["waited_s"] = time.time() - t0
    out["err"] = c.comm_ipc_error()
    try:
        c.stpcg(x, c.op_diag(c.upload(np.full(1000, 2.0))), Delta=1.0, max_iterations=3)
        out["stpcg"] = "returned"
    except capi.MiError as e:
        out["stpcg"] = "MiError: " + str(e)[:80]
else:
    time.sleep(3.0)
with open(os.path.join(os.environ["IPC_WORKER_OUT"], f"rank{rank}.json"), "w") as f:
    json.dump(out, f)
dist.barrier()
os._exit(0)    # skip collective teardown: the ranks are deliberately out of step
