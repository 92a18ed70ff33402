"""Data-parallel plumbing over torch.distributed (backend "nccl" = RCCL over xGMI on ROCm; "gloo"
on CPU for tests).  One process per GPU; splat parameters are replicated, the passes of an
iteration are sharded by CAMERA (camera c -> rank c % world, both its white and its black pass, so the
rank keeps the shared projection / tile lists / fused two-pass backward of the camera), and ONE sum
all-reduce of the averaged-gradient buffer [(12+3M) planes x plane stride fp32] runs between accumulate
and apply.  S (the divisor of accumulateGradients, src/Trainer.cu:419) stays the GLOBAL pass
count, so local sums are already correctly scaled and the reduction is a plain sum.

The reference has no multi-GPU path; this module is the build-side addition north_star asks for.
"""
import ctypes as C
import os

import numpy as np


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_views(total_views, rank, world):
    """Passes owned by `rank`.  The iteration has C = total_views / 2 cameras, pass c = camera c on white, pass C + c =
    camera c on black (src/Trainer.cu:311-314).  Camera c goes to rank c % world WITH BOTH PASSES, so every rank keeps
    the per-camera sharing (projection, lists, forward blend once; fused two-pass backward).  Only when there are
    fewer cameras than ranks are the passes dealt one by one (v % world) — twins then split, and ranks beyond the pass
    count own nothing: they still join the collective with a zero gradient (gs_trainer_set_views with n_views = 0)."""
    C = total_views // 2
    if total_views % 2 == 0 and C >= world:
        return [v for v in range(total_views) if (v % C) % world == rank]
    return [v for v in range(total_views) if v % world == rank]


def allreduce_sum_numpy(arr):
    """In-place sum all-reduce of a numpy array through the default process group (gloo on CPU)."""
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(arr)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return arr


def broadcast_bytes(data, src=0, max_len=1 << 16):
    """`data` (bytes on `src`, ignored elsewhere) to every rank, as a fixed-size CPU tensor: with a "cpu:gloo,cuda:nccl" group a CPU
    tensor travels over gloo, whatever device the object collectives would have picked."""
    import torch
    import torch.distributed as dist
    buf = torch.zeros(max_len + 8, dtype=torch.uint8)
    if dist.get_rank() == src:
        raw = bytes(data)
        if len(raw) > max_len:
            raise ValueError(f"broadcast_bytes: {len(raw)} bytes exceed {max_len}")
        buf[:8] = torch.frombuffer(bytearray(len(raw).to_bytes(8, "little")), dtype=torch.uint8)
        buf[8:8 + len(raw)] = torch.frombuffer(bytearray(raw), dtype=torch.uint8)
    dist.broadcast(buf, src=src)
    n = int.from_bytes(bytes(buf[:8].tolist()), "little")
    return bytes(buf[8:8 + n].tolist())


def all_gather_bytes(data, width=64):
    """One fixed-width byte string per rank (e.g. a digest), gathered over CPU tensors."""
    import torch
    import torch.distributed as dist
    raw = bytes(data)[:width].ljust(width, b"\0")
    mine = torch.frombuffer(bytearray(raw), dtype=torch.uint8)
    out = [torch.zeros(width, dtype=torch.uint8) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [bytes(t.tolist()) for t in out]


class _DevPtr:
    """Minimal __cuda_array_interface__ carrier so torch can alias a raw device pointer (no copy)."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (ptr, False), "version": 2}


class TorchAllReduce:
    """gs_allreduce_fn implemented with torch.distributed.all_reduce on a tensor that aliases the
    trainer's gradient buffer, enqueued behind the trainer's HIP stream."""

    def __init__(self, trainer):
        import torch
        import torch.distributed as dist
        from . import capi
        self.torch, self.dist = torch, dist
        st = C.c_void_p()
        capi.check(capi.lib().gs_trainer_get_stream(trainer.handle, C.byref(st)))
        self.stream = torch.cuda.ExternalStream(st.value)
        self._alias = {}

        def hook(buf, n, hip_stream, user):
            try:
                key = (buf, n)
                t = self._alias.get(key)
                if t is None:
                    t = torch.as_tensor(_DevPtr(buf, n), device=torch.device("cuda", torch.cuda.current_device()))
                    self._alias = {key: t}
                with torch.cuda.stream(self.stream):
                    dist.all_reduce(t, op=dist.ReduceOp.SUM)
                return 0
            except Exception as e:  # never let an exception cross the C boundary
                print("all-reduce hook failed:", repr(e), flush=True)
                return 1

        self._cb = capi.ALLREDUCE_FN(hook)
        capi.check(capi.lib().gs_trainer_set_allreduce(trainer.handle, C.cast(self._cb, C.c_void_p), None))


class TorchShardedUpdate:
    """gs_trainer_set_sharded_update with torch.distributed: reduce_scatter_tensor / all_gather_into_tensor IN PLACE on
    tensors that alias the library's plane-major buffers (the rank's chunk is a view of the whole buffer), enqueued
    behind the trainer's HIP stream.  Each rank then updates 1/world of the parameters and keeps 1/world of the Adam
    moments current; the bytes on the wire equal an all-reduce's.  The reduce-scatter's form is chosen ONCE, from the
    backend that serves device tensors: nccl (= RCCL) runs reduce_scatter_tensor; gloo (the two-processes-on-one-GPU
    test) has none and runs an all-reduce, which leaves the same sums in the rank's chunk.  A failing collective returns
    non-zero to the library on this rank — it is never retried in another form, which would pair a different collective
    with the peers' and hang them."""

    def __init__(self, trainer, rank, world):
        import torch
        import torch.distributed as dist
        from . import capi
        self.rank, self.world = int(rank), int(world)
        st = C.c_void_p()
        capi.check(capi.lib().gs_trainer_get_stream(trainer.handle, C.byref(st)))
        self.stream = torch.cuda.ExternalStream(st.value)
        self._alias = {}
        self.calls = {"reduce_scatter": 0, "all_gather": 0}
        try:   # the backend object that will serve device tensors ("cpu:gloo,cuda:nccl" groups carry two)
            backend = dist.distributed_c10d._get_default_group()._get_backend(torch.device("cuda")).__class__.__name__
        except Exception:
            backend = str(dist.get_backend())
        self.native_reduce_scatter = "nccl" in backend.lower()

        def alias(buf, n):
            t = self._alias.get((buf, n))
            if t is None:
                if len(self._alias) > 8:
                    self._alias.clear()
                t = torch.as_tensor(_DevPtr(buf, n), device=torch.device("cuda", torch.cuda.current_device()))
                self._alias[(buf, n)] = t
            return t

        def rs(buf, n, hip_stream, user):
            try:
                t = alias(buf, n)
                c = n // self.world
                with torch.cuda.stream(self.stream):
                    if self.native_reduce_scatter:
                        dist.reduce_scatter_tensor(t[self.rank * c:(self.rank + 1) * c], t, op=dist.ReduceOp.SUM)
                    else:
                        # gloo stand-in: the sums everywhere, then everything outside the rank's own chunk POISONED — a real
                        # reduce-scatter leaves this rank's partial sums there, and nothing downstream may read them
                        dist.all_reduce(t, op=dist.ReduceOp.SUM)
                        t[:self.rank * c] = float("nan")
                        t[(self.rank + 1) * c:] = float("nan")
                self.calls["reduce_scatter"] += 1
                return 0
            except Exception as e:  # never let an exception cross the C boundary
                print("reduce-scatter hook failed:", repr(e), flush=True)
                return 1

        def ag(buf, n, hip_stream, user):
            try:
                t = alias(buf, n)
                c = n // self.world
                with torch.cuda.stream(self.stream):
                    dist.all_gather_into_tensor(t, t[self.rank * c:(self.rank + 1) * c])
                self.calls["all_gather"] += 1
                return 0
            except Exception as e:
                print("all-gather hook failed:", repr(e), flush=True)
                return 1

        self._rs, self._ag = capi.ALLREDUCE_FN(rs), capi.ALLREDUCE_FN(ag)
        capi.check(capi.lib().gs_trainer_set_sharded_update(trainer.handle, C.cast(self._rs, C.c_void_p), C.cast(self._ag, C.c_void_p),
                                                            None, self.rank, self.world))


def exchange_wire_bytes(form, n_cameras, world, P, M):
    """Bytes a rank RECEIVES per step under each data-parallel form (the figure that loads its xGMI links; ring algorithms send
    as much).  P splats (plane stride ~ P), M SH coefficients, n_cameras cameras of the iteration, `world` ranks.
      allreduce : ring all-reduce of the (12 + 3M) P fp32 gradient buffer: 2 (G-1)/G of it
      sharded   : reduce-scatter of the gradients + all-gather of the parameters: (G-1)/G of each
      compact   : all-gather of one dL_dRGB record (3 P fp32) per camera of the other ranks + ring all-reduce of 12 planes"""
    G = world
    if G <= 1:
        return 0
    grad = (12 + 3 * M) * P * 4
    if form == "allreduce":
        return int(2 * (G - 1) / G * grad)
    if form == "sharded":
        return int((G - 1) / G * (grad + (11 + 3 * M) * P * 4))
    if form == "compact":
        slots = -(-n_cameras // G)
        return int((G - 1) * slots * 3 * P * 4 + 2 * (G - 1) / G * 12 * P * 4)
    raise ValueError(form)


def choose_exchange(n_cameras, world, M):
    """--collective auto: the compact exchange where it moves fewer bytes than the all-reduce and its layout contract holds
    (whole cameras per rank, at least one each), i.e. while the iteration has fewer cameras than about 2 M; else all-reduce."""
    if world > 1 and n_cameras >= world and exchange_wire_bytes("compact", n_cameras, world, 1 << 16, M) < exchange_wire_bytes("allreduce", n_cameras, world, 1 << 16, M):
        return "compact"
    return "allreduce"


class TorchCompactExchange:
    """gs_trainer_set_compact_exchange with torch.distributed: all_gather_into_tensor IN PLACE on the dL_dRGB record buffer and
    all_reduce on the twelve geometry planes, each enqueued behind the HIP stream the library names (the all-reduce comes on the
    trainer's second stream, beside the all-gather).  With the nccl backend the all-reduce runs on a process group of its own
    (`reduce_group`): collectives of one communicator execute in issue order, so sharing the default group would serialise
    the two.  cameras: every camera of the iteration in the reference's order (camera c -> rank c % world, dist.shard_views)."""

    def __init__(self, trainer, rank, world, cameras, reduce_group=None):
        import torch
        import torch.distributed as dist
        from . import capi
        self.rank, self.world = int(rank), int(world)
        self.calls = {"all_gather": 0, "all_reduce": 0}
        self._alias, self._streams = {}, {}
        self.reduce_group = reduce_group
        dev = torch.device("cuda", torch.cuda.current_device())

        def alias(buf, n):
            t = self._alias.get((buf, n))
            if t is None:
                if len(self._alias) > 8:
                    self._alias.clear()
                t = torch.as_tensor(_DevPtr(buf, n), device=dev)
                self._alias[(buf, n)] = t
            return t

        def stream_of(ptr):
            st = self._streams.get(ptr)
            if st is None:
                st = self._streams[ptr] = torch.cuda.ExternalStream(ptr)
            return st

        def ag(buf, n, hip_stream, user):
            try:
                t = alias(buf, n)
                c = n // self.world
                with torch.cuda.stream(stream_of(hip_stream)):
                    dist.all_gather_into_tensor(t, t[self.rank * c:(self.rank + 1) * c])
                self.calls["all_gather"] += 1
                return 0
            except Exception as e:  # never let an exception cross the C boundary
                print("all-gather hook failed:", repr(e), flush=True)
                return 1

        def ar(buf, n, hip_stream, user):
            try:
                t = alias(buf, n)
                with torch.cuda.stream(stream_of(hip_stream)):
                    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.reduce_group)
                self.calls["all_reduce"] += 1
                return 0
            except Exception as e:
                print("all-reduce hook failed:", repr(e), flush=True)
                return 1

        self._ag, self._ar = capi.ALLREDUCE_FN(ag), capi.ALLREDUCE_FN(ar)
        campos = np.ascontiguousarray([c.location for c in cameras], np.float32).reshape(-1, 3)
        capi.check(capi.lib().gs_trainer_set_compact_exchange(trainer.handle, C.cast(self._ag, C.c_void_p), C.cast(self._ar, C.c_void_p), None,
                                                              self.rank, self.world, campos.shape[0], campos.ctypes.data_as(C.c_void_p)))


class NativeRcclComm:
    """The library's own RCCL communicator (gs_comm_*): the 128-byte unique id is created on rank 0
    and broadcast through torch.distributed's store; the collective itself never touches Python.
    sharded=True installs ncclReduceScatter / ncclAllGather as the sharded update's hooks instead of the all-reduce."""

    def __init__(self, trainer, rank, world, sharded=False, compact_cameras=None):
        from . import capi
        L = capi.lib()
        self.handle = self._communicator(rank, world)
        if compact_cameras is not None:   # compact exchange: a second communicator carries the all-reduce beside the all-gather
            self.handle2 = self._communicator(rank, world)
            campos = np.ascontiguousarray([c.location for c in compact_cameras], np.float32).reshape(-1, 3)
            capi.check(L.gs_trainer_attach_comm_compact(trainer.handle, self.handle, self.handle2, campos.shape[0], campos.ctypes.data_as(C.c_void_p)))
        else:
            capi.check((L.gs_trainer_attach_comm_sharded if sharded else L.gs_trainer_attach_comm)(trainer.handle, self.handle))

    @staticmethod
    def _communicator(rank, world):
        import torch.distributed as dist
        from . import capi
        L = capi.lib()
        ident = (C.c_char * capi.GS_COMM_ID_BYTES)()
        if rank == 0:
            capi.check(L.gs_comm_unique_id(ident))
        if world > 1:
            got = broadcast_bytes(bytes(ident.raw) if rank == 0 else b"", src=0, max_len=capi.GS_COMM_ID_BYTES)
            C.memmove(ident, got, capi.GS_COMM_ID_BYTES)
        handle = C.c_void_p()
        capi.check(L.gs_comm_create(ident, rank, world, C.byref(handle)))
        return handle
