"""API-level check of the RCCL calls bench.py makes for several GPUs, with a world of ONE rank on the one GPU of the
box (RCCL refuses two ranks on one device, so the collectives' semantics over ranks cannot be exercised here):
init_process_group("nccl", device_id=...), all_gather_into_tensor on the uint8 record buffer and on float64, all_reduce
MAX / SUM, barrier, destroy_process_group."""
import datetime
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cyclic-gps_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29577")
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from cyclic_gps import sharded  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(seconds=120))
n = 1 << 16
Rs, Os, b, O_left, mahal_true, logdet_true = sharded.make_sharded_system(n, 4, torch.float64, dev, 0, 1)
plan = sharded.ShardedMahalLogdet(Rs, Os, b, O_left, n, 0, 1)
plan.reduce_to_send()
plan.gather(plan.send, plan.recv[:plan.send.numel()])          # the ONE collective of a step, on the uint8 buffer
out = torch.zeros(2, dtype=torch.float64, device=dev)
plan.ops.finish(plan.recv, 1, plan.rec_bytes, plan.msg_bytes, n, n, out)
dist.barrier()
t = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
torch.cuda.synchronize()
res = out.cpu()
print("nccl world-1 check:", "mahal rel err %.1e logdet rel err %.1e" % (abs(float(res[0]) - mahal_true) / abs(mahal_true),
                                                                        abs(float(res[1]) - logdet_true) / abs(logdet_true)), float(t))
dist.destroy_process_group()
print("ok")
