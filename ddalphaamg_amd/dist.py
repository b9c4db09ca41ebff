"""One process per GPU on a Cartesian process grid: helpers around the halo transports of
include/ddamg_hip.h (reference: the MPI_Cart setup of src/init.c:455-520 and data_layout.c:23-60).

The host application of the reference is an MPI code; here the launcher is torch.distributed
(torchrun), which only bootstraps: it broadcasts the RCCL id, or moves the staged host buffers
(gloo) when the host transport is chosen.  The exchange itself runs inside libddamg_hip.so.
"""
import numpy as np
from . import api


def process_grid_for(nranks):
    """split T first, then Z, Y, X (powers of two), e.g. 8 -> [2,2,2,1]"""
    P = [1, 1, 1, 1]
    mu = 0
    n = int(nranks)
    while n > 1:
        if n % 2:
            raise api.DDAMGError("number of processes must be a power of two")
        P[mu % 4] *= 2
        n //= 2
        mu += 1
    return P


def coords_of(rank, P):
    """inverse of rank = ((pt*Pz+pz)*Py+py)*Px+px"""
    c = [0, 0, 0, 0]
    for mu in (3, 2, 1, 0):
        c[mu] = rank % P[mu]
        rank //= P[mu]
    return c


def local_part(global_lex, global_lattice, P, coords):
    """cut the part of process `coords` out of a lexicographic global field [V_global, ...]"""
    G = list(global_lattice)
    L = [G[mu] // P[mu] for mu in range(4)]
    a = np.asarray(global_lex).reshape(G + [-1])
    sl = tuple(slice(coords[mu] * L[mu], (coords[mu] + 1) * L[mu]) for mu in range(4))
    return np.ascontiguousarray(a[sl]).reshape(int(np.prod(L)), -1)


def attach_rccl(ctx, rank, group=None):
    """RCCL transport: rank 0 creates the id, torch.distributed broadcasts it"""
    import torch.distributed as dist
    obj = [api.rccl_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(obj, src=0, group=group)
    ctx.comm_init_rccl(obj[0])


def attach_host(ctx, group=None):
    """host transport over a torch.distributed group that accepts CPU tensors (gloo)"""
    import torch
    import torch.distributed as dist

    def exchange(msgs):
        reqs = []
        for send_peer, recv_peer, tag, snd, rcv in msgs:
            reqs.append(dist.irecv(torch.from_numpy(rcv), src=recv_peer, group=group, tag=tag))
        for send_peer, recv_peer, tag, snd, rcv in msgs:
            reqs.append(dist.isend(torch.from_numpy(snd), dst=send_peer, group=group, tag=tag))
        for r in reqs:
            r.wait()

    def allreduce(buf):
        dist.all_reduce(torch.from_numpy(buf), op=dist.ReduceOp.SUM, group=group)

    ctx.comm_init_host(exchange, allreduce)
