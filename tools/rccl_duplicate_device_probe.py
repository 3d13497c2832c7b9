import os, sys, torch, torch.distributed as dist
rank=int(os.environ["RANK"]); world=int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda",0))
    t=torch.ones(4,device="cuda")*(rank+1); dist.all_reduce(t); torch.cuda.synchronize()
    print("RCCL_DUP_OK", rank, t.tolist(), flush=True)
except Exception as e:
    print("RCCL_DUP_FAIL", rank, repr(e)[:400], flush=True)
