"""tools/rccl_one_gpu_probe.py -- can two ranks of one RCCL communicator share ONE GPU?  (The N > 1 shard protocol has only
run over gloo and with every rank in one process: a one-GPU box would be the place to see ncclSend / ncclRecv between ranks --
if RCCL allowed it.)  Run as: python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/rccl_one_gpu_probe.py
Prints what happened; exits 0 either way."""
import os, sys, datetime
import torch, torch.distributed as dist
rank = int(os.environ["RANK"])
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", timeout=datetime.timedelta(seconds=60))
    t = torch.ones(4, device="cuda") * (rank + 1)
    dist.all_reduce(t)
    torch.cuda.synchronize()
    print("rank", rank, "RCCL with two ranks on cuda:0: all_reduce ->", t.tolist(), flush=True)
    dist.destroy_process_group()
except Exception as e:                                       # noqa: BLE001
    print("rank", rank, "RCCL with two ranks on cuda:0 refused:", type(e).__name__, str(e).replace("\n", " | ")[:600], flush=True)
os._exit(0)
