"""Experiment: can two ranks share ONE GPU under RCCL send/recv?  (dev tool, not product)"""
import os
import sys
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def w(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    t = torch.full((1024,), float(rank), device="cuda")
    r = torch.empty_like(t)
    ops = [dist.P2POp(dist.isend, t, 1 - rank), dist.P2POp(dist.irecv, r, 1 - rank)]
    for x in dist.batch_isend_irecv(ops):
        x.wait()
    torch.cuda.synchronize()
    print(rank, "got", r[0].item(), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    mp.start_processes(w, args=(2, 29533), nprocs=2, join=True, start_method="spawn")
