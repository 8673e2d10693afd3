import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev)
t = torch.arange(12, dtype=torch.float64, device=dev).reshape(2, 2, 3)
bufs = [torch.empty_like(t)]
dist.gather(t, gather_list=bufs, dst=0)
assert torch.equal(bufs[0], t)
x = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(x, op=dist.ReduceOp.MAX); dist.all_reduce(x, op=dist.ReduceOp.SUM)
dist.barrier(); torch.cuda.synchronize()
print("nccl one-rank ok", dist.get_backend(), float(x))
dist.destroy_process_group()
