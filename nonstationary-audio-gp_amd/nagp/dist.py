"""Multi-GPU layer: one process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on ROCm,
"gloo" on CPU for tests).  The path shards by PROBLEM (audio segments / hyper-parameter replicas,
SURVEY 8e): rank r owns problems r, r+W, r+2W, ... ; there is no data-path collective -- the only
exchange is the all-reduce of the per-sweep negative log marginal likelihood
nlZ[itt] = -sum_k lZ_k (gf_ep_modulator_nmf.m:187, 277, 525) over all problems.
"""
import os

import numpy as np


def env_world():
    return int(os.environ.get('RANK', 0)), int(os.environ.get('LOCAL_RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))


def shard(n_problems, rank, world):
    """Round-robin problem -> rank map (problem i -> rank i mod W)."""
    return list(range(rank, n_problems, world))


def init(backend=None):
    import torch.distributed as dist
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        method = os.environ.get('NAGP_DIST_INIT_METHOD')       # e.g. file:///tmp/x/rdv -- no TCP port to race for
        if method:
            dist.init_process_group(backend=backend or 'nccl', init_method=method, rank=rank, world_size=world)
        else:
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', '29500')
            dist.init_process_group(backend=backend or 'nccl', rank=rank, world_size=world)
    return rank, local_rank, world


def file_rendezvous_env(dirpath, world):
    """Environment entries for ranks started by hand (tests, `bench.py --gpus N` without a launcher): a file store in a
    directory only these ranks know instead of a TCP port picked by bind-then-close, which two children can lose to
    another process between the close and their own bind."""
    path = os.path.join(os.path.abspath(dirpath), 'nagp_rendezvous')
    if os.path.exists(path):
        os.remove(path)
    return {'NAGP_DIST_INIT_METHOD': 'file://' + path, 'WORLD_SIZE': str(world)}


def allreduce_nlz(nlz_local, device=None):
    """Sum the (n_local_problems x ep_itts) partial nlZ over problems and ranks -> (ep_itts,) total."""
    import torch
    import torch.distributed as dist
    part = torch.as_tensor(np.asarray(nlz_local, dtype=np.float64).reshape(-1, np.shape(nlz_local)[-1]).sum(axis=0))
    if device is not None:
        part = part.to(device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(part, op=dist.ReduceOp.SUM)
    return part.cpu().numpy()


def allgather_sums(nlz_local, device=None):
    """(world, ep_itts): every rank's own sum over its problems, gathered (an independent path to the total allreduce_nlz returns)."""
    import torch
    import torch.distributed as dist
    part = torch.as_tensor(np.asarray(nlz_local, dtype=np.float64).reshape(-1, np.shape(nlz_local)[-1]).sum(axis=0))
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return part.numpy()[None, :]
    if device is not None:
        part = part.to(device)
    outs = [torch.empty_like(part) for _ in range(dist.get_world_size())]
    dist.all_gather(outs, part)
    return np.stack([o.cpu().numpy() for o in outs])


def allreduce_max(x, device=None):
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(x)], dtype=torch.float64)
    if device is not None:
        t = t.to(device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.cpu()[0])


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def finalize():
    """Leave the process group in step: a barrier, then destroy it.  A rank that simply exits while a peer's gloo / RCCL
    threads still hold its sockets makes the peer abort at interpreter exit ("terminate called ... connection reset")."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        if dist.get_world_size() > 1:
            dist.barrier()
        dist.destroy_process_group()


def world_size():
    """World size the initialised process group reports (1 without one)."""
    import torch.distributed as dist
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def backend_id():
    """'nccl' (RCCL on ROCm), 'gloo', ... as torch.distributed reports it; 'none' without a process group."""
    import torch.distributed as dist
    return str(dist.get_backend()) if (dist.is_available() and dist.is_initialized()) else 'none'


def gather_device_ids(local_rank):
    """['rank r: <PCI bus id> <device name>'] of every rank's GPU (all-gathered): N ranks on N physical devices show N different bus ids."""
    import torch
    import torch.distributed as dist
    me = 'no GPU'
    if torch.cuda.is_available():
        pr = torch.cuda.get_device_properties(local_rank)
        bus = getattr(pr, 'pci_bus_id', None)
        pci = ('%04x:%02x:%02x' % (getattr(pr, 'pci_domain_id', 0), bus, getattr(pr, 'pci_device_id', 0))) if bus is not None else str(getattr(pr, 'uuid', 'unknown'))
        me = '%s %s' % (pci, pr.name)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        out = [None] * dist.get_world_size()
        dist.all_gather_object(out, me)
        return ['rank %d: %s' % (r, v) for r, v in enumerate(out)]
    return ['rank 0: %s' % me]


def backend_name():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        b = dist.get_backend()
        return 'RCCL (torch.distributed backend "nccl")' if b == 'nccl' else str(b)
    return 'single process, no collective'
