import os, torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
t = torch.ones(1024, device="cuda")
dist.all_reduce(t); dist.barrier(); torch.cuda.synchronize()
print("rccl ok", float(t.sum()), dist.get_world_size(), flush=True)
dist.destroy_process_group()
