"""What ONE GPU can say about the RCCL side of `bench.py --gpus N`: a one-rank "nccl" process group on the GPU box -- the library
loads, `init_process_group(device_id=...)` is accepted, an in-place `all_gather_into_tensor`, a second `wait()` on a finished work
and an asynchronous broadcast issued from a side stream all behave as `sharding.FrameBroadcaster` assumes.  (Two ranks on one
GPU are refused by RCCL; the schedule itself runs over gloo in tests/test_sharding_cpu.py.)  Run under gpurun."""
import os, datetime, torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29611", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(seconds=60))
t = torch.ones(4, device=dev); dist.all_reduce(t); torch.cuda.synchronize(); print("all_reduce", t.tolist())
buf = torch.arange(8., device=dev).reshape(4, 2); mine = buf[0:4]
w = dist.all_gather_into_tensor(buf, mine, async_op=True); w.wait(); w.wait(); torch.cuda.synchronize(); print("in-place all_gather ok", buf.flatten().tolist())
w = dist.broadcast(buf, src=0, async_op=True); w.wait(); torch.cuda.synchronize(); print("broadcast ok")
ops = []  # (no peers at world 1)
s = torch.cuda.Stream(device=dev)
with torch.cuda.stream(s):
    w = dist.broadcast(buf, src=0, async_op=True)
w.wait(); torch.cuda.synchronize(); print("broadcast on side stream ok; backend", dist.get_backend())
dist.destroy_process_group(); print("done")
