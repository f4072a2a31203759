"""RCCL on a one-GPU box (run by tests/test_parallel_gpu.py in a fresh process): a world of ONE rank on the "nccl" backend (= RCCL on
ROCm).  It cannot show a transfer over xGMI, but it runs what the multi-GPU path calls — communicator creation with a device id, the
barrier and MAX all-reduce of bench.py's timing, and FeatureExchange's all_gather_into_tensor on the feature slabs of a real extraction —
through librccl on the GPU, and checks the gathered slabs byte for byte."""
import os
import sys
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import datetime
import numpy as np
import torch
import torch.distributed as dist
from morb_slam_amd import parallel, synth
from morb_slam_amd.extractor import ORBextractor

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", sys.argv[1])
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, timeout=datetime.timedelta(seconds=120))
assert dist.get_backend() == "nccl"
dist.barrier()
t = torch.tensor([3.25], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t.item()) == 3.25
l, r = synth.make_stereo_pair(640, 480, seed=3)
ex = ORBextractor(600, 1.2, 8, 20, 7)
kps, desc, cnt, _ = ex.extract_batch(torch.from_numpy(np.stack([l, r])).to(dev))
torch.cuda.synchronize()
fx = parallel.FeatureExchange(always_collective=True)
node = torch.arange(kps.shape[0] * kps.shape[1], dtype=torch.int32, device=dev).reshape(kps.shape[0], kps.shape[1])
pk, pd, pc, pn = fx.exchange(kps, desc, cnt, node)
torch.cuda.synchronize()
for a, b in ((pk, kps), (pd, desc), (pc, cnt), (pn, node)):
    assert a.data_ptr() != b.data_ptr()          # the collective wrote its own buffer
    assert torch.equal(a, b)
assert int(cnt.sum()) > 500
dist.barrier()
dist.destroy_process_group()
print("rccl world-1 ok", int(cnt.sum()))
