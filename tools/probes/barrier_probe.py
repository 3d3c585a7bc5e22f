import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda",0))
t=torch.zeros(1,device="cuda")
for _ in range(5): dist.barrier(); dist.all_reduce(t); torch.cuda.synchronize()
def tm(f,n=20):
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter()-t0)/n*1e6
print("dist.barrier(): %.0f us"%tm(dist.barrier))
print("all_reduce + synchronize: %.0f us"%tm(lambda:(dist.all_reduce(t), torch.cuda.synchronize())))
print("synchronize only: %.0f us"%tm(torch.cuda.synchronize))
dist.destroy_process_group()
