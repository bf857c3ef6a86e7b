import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
x = torch.arange(1 << 20, dtype=torch.int64, device=dev).to(torch.uint8)
out = torch.empty(1 << 20, dtype=torch.uint8, device=dev)
dist.all_gather_into_tensor(out, x)
dist.barrier()
t = torch.tensor([1.5], dtype=torch.float64, device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX)
torch.cuda.synchronize()
assert torch.equal(out, x) and float(t) == 1.5
dist.destroy_process_group()
print("rccl single-rank smoke ok")
