import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29511")
dev=torch.device("cuda",0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
t=torch.tensor([1.5],dtype=torch.float64,device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX); dist.barrier()
print("nccl ok", float(t.item()))
dist.destroy_process_group()
