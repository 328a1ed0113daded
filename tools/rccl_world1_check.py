"""Experiment script (GPU): the RCCL calls of bench.py and multi_gpu.py with a one-rank process group (all a one-GPU box allows). Not a test."""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda",0))
t=torch.tensor([1.5],dtype=torch.float64,device="cuda"); dist.all_reduce(t,op=dist.ReduceOp.MAX); dist.barrier()
r=torch.tensor([2.0],dtype=torch.float64,device="cuda"); dist.all_reduce(r,op=dist.ReduceOp.SUM)
import sys; sys.path[:0]=[os.getcwd()]
from ray_tracing_amd import multi_gpu
g=multi_gpu.TileGather(128,64,64,1,0,4,"cuda:0"); f=torch.ones(128*64*4,device="cuda"); g.combine(f)
s=multi_gpu.SampleReduce(128*64*4,1,0,"cuda:0",gather_to=0); out=s.combine(f.clone(),2)
torch.cuda.synchronize(); print("nccl world=1 ok", float(t), float(r), float(out.sum()))
dist.destroy_process_group()
