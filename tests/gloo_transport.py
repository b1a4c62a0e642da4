"""Test harness only: a sacamd_transport (include/sac_amd.h) over torch.distributed, so that the library's record gather
(sacamd_gather_records_via -- the same gather_core the RCCL communicator drives) runs between CPU processes under gloo."""
import ctypes

import numpy as np
import torch
import torch.distributed as dist

import sac_amd.api as api


def make_transport(rank: int, world: int) -> api.TransportC:
    def allgather(_self, send, recv, count):
        t = torch.from_numpy(np.ctypeslib.as_array(send, shape=(count,)).copy())
        outs = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(outs, t)
        np.ctypeslib.as_array(recv, shape=(count * world,))[:] = torch.cat(outs).numpy()
        return 0

    def group_begin(_self):
        return 0

    def send(_self, peer, buf, nbytes):
        a = np.ctypeslib.as_array(ctypes.cast(buf, ctypes.POINTER(ctypes.c_ubyte)), shape=(nbytes,))
        dist.send(torch.from_numpy(a.copy()), dst=peer)
        return 0

    def recv(_self, peer, buf, nbytes):
        t = torch.zeros(nbytes, dtype=torch.uint8)
        dist.recv(t, src=peer)
        ctypes.memmove(buf, t.numpy().ctypes.data, nbytes)
        return 0

    def group_end(_self):
        return 0

    cbs = (api._AG(allgather), api._GB(group_begin), api._SR(send), api._SR(recv), api._GB(group_end))
    tr = api.TransportC(None, rank, world, *cbs)
    tr._keep = cbs     # the callbacks must outlive the struct
    return tr
