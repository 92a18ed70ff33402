"""Single-rank plumbing test of the data-parallel path on the GPU: bench.py with --force-dist initialises
torch.distributed (nccl = RCCL) with world_size 1 and installs the collective hook, once through
torch.distributed.all_reduce on the aliased gradient buffer and once through the library's own RCCL
communicator (gs_comm_*).  The step results must not change (a 1-rank sum is the identity) and the hook
must actually have run (stage "collective" shows launches)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--config", "2", "--steps", "3", "--warmup", "1",
                          "--no-cpu-baseline", "--prewarm-seconds", "0", "--long-steps", "0"] + extra, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


@pytest.mark.parametrize("collective", ["auto", "torch", "rccl", "torch-sharded", "rccl-sharded", "torch-compact", "rccl-compact"])
def test_collective_hook_runs_with_one_rank(collective):
    base = _run([])
    got = _run(["--force-dist", "--collective", collective])
    assert "collective" in got["stages_ms_per_launch"], got["stages_ms_per_launch"]
    if collective == "auto":   # the default of an N-GPU run: the library's own RCCL communicators wherever librccl loads (no Python inside the step)
        assert got["config"]["collective_note"].startswith("auto -> rccl") and got["config"]["collective"].startswith("rccl"), got["config"]["collective_note"]
    assert got["config"]["mean_num_rendered_per_view"] == base["config"]["mean_num_rendered_per_view"]
    assert got["value"] > 0


def test_failing_collective_hooks_end_the_step_cleanly():
    """A collective that fails on this rank (gs_allreduce_fn / gs_collective_fn returning non-zero): gs_trainer_step returns
    GS_ERR_COLLECTIVE (-11), gs_last_error names the collective, a second step fails the same way instead of hanging or
    applying a half-reduced gradient, the trainer works again once the hook is removed, and destroying the library's own
    communicator after a failure is safe (the hooks abort it on the spot)."""
    import ctypes as C

    import numpy as np

    import gsplat_amd as gs
    from gsplat_amd import capi
    L = capi.lib()
    P, M, W, H = 400, 1, 64, 64
    s = gs.synth.random_splats(P, M, 3)
    cams = gs.camera.get_cameras(1)
    tr = gs.Trainer(W, H)
    tr.model = gs.ModelSplatsDevice(gs.ModelSplatsHost.fromVectors(s["loc"], s["sh"], s["scale"], s["opac"], s["rot"]))
    blank = np.full(W * H, 0xFF808080, np.uint32)
    tr.captureTruths(cams, [blank], [blank])
    proj = gs.Project()
    tr.train(proj)
    before = gs.ModelSplatsHost.fromDevice(tr.model).locations[:3 * P].copy()
    calls = []

    def failing(buf, n, stream, user):
        calls.append(n)
        return 7
    cb = capi.ALLREDUCE_FN(failing)
    capi.check(L.gs_trainer_set_allreduce(tr.handle, C.cast(cb, C.c_void_p), None))
    for _ in range(2):
        with pytest.raises(capi.GsError) as e:
            tr.train(proj)
        assert e.value.status == capi.GS_ERR_COLLECTIVE == -11 and "all-reduce hook failed with 7" in str(e.value)
    assert len(calls) == 2 and calls[0] == (12 + 3 * M) * 448
    assert np.array_equal(gs.ModelSplatsHost.fromDevice(tr.model).locations[:3 * P], before)   # no update was applied
    capi.check(L.gs_trainer_set_allreduce(tr.handle, None, None))
    tr.train(proj)
    assert not np.array_equal(gs.ModelSplatsHost.fromDevice(tr.model).locations[:3 * P], before)
    # the sharded form: reduce-scatter fails first; with a working reduce-scatter the all-gather's failure is reported
    ok = capi.ALLREDUCE_FN(lambda buf, n, stream, user: 0)
    for rs, ag, what in ((cb, ok, "reduce-scatter hook failed"), (ok, cb, "all-gather (parameters) hook failed")):
        capi.check(L.gs_trainer_set_sharded_update(tr.handle, C.cast(rs, C.c_void_p), C.cast(ag, C.c_void_p), None, 0, 1))
        with pytest.raises(capi.GsError) as e:
            tr.train(proj)
        assert e.value.status == capi.GS_ERR_COLLECTIVE and what in str(e.value), str(e.value)
    capi.check(L.gs_trainer_set_sharded_update(tr.handle, None, None, None, 0, 1))
    tr.train(proj)
    # the compact exchange: whichever of its two collectives fails is named; with the overlap the all-reduce is issued first
    campos = np.ascontiguousarray([c.location for c in cams], np.float32)
    for overlap in (1, 0):
        tr.set_option("exchange_overlap", overlap)
        for ag, ar, what in ((cb, ok, "all-gather (dL_dRGB records) hook failed"), (ok, cb, "all-reduce (geometry planes) hook failed")):
            capi.check(L.gs_trainer_set_compact_exchange(tr.handle, C.cast(ag, C.c_void_p), C.cast(ar, C.c_void_p), None, 0, 1, 1,
                                                         campos.ctypes.data_as(C.c_void_p)))
            with pytest.raises(capi.GsError) as e:
                tr.train(proj)
            assert e.value.status == capi.GS_ERR_COLLECTIVE and what in str(e.value), str(e.value)
    capi.check(L.gs_trainer_set_compact_exchange(tr.handle, C.cast(ok, C.c_void_p), C.cast(ok, C.c_void_p), None, 0, 1, 1, campos.ctypes.data_as(C.c_void_p)))
    tr.train(proj)          # one rank: the exchange is the identity and the step goes through
    with pytest.raises(capi.GsError) as e:   # the layout contract is checked: two cameras announced, one camera's passes set
        capi.check(L.gs_trainer_set_compact_exchange(tr.handle, C.cast(ok, C.c_void_p), C.cast(ok, C.c_void_p), None, 0, 1, 2,
                                                     np.zeros(6, np.float32).ctypes.data_as(C.c_void_p)))
        tr.train(proj)
    assert e.value.status == capi.GS_ERR_INVALID_ARGUMENT and "compact exchange" in str(e.value)
    capi.check(L.gs_trainer_set_compact_exchange(tr.handle, None, None, None, 0, 1, 0, None))
    tr.train(proj)
    # the library's own communicator with one rank: attach, step, then destroy — and destroy again after detaching
    ident = (C.c_char * capi.GS_COMM_ID_BYTES)()
    capi.check(L.gs_comm_unique_id(ident))
    comm = C.c_void_p()
    capi.check(L.gs_comm_create(ident, 0, 1, C.byref(comm)))
    capi.check(L.gs_trainer_attach_comm(tr.handle, comm))
    tr.train(proj)
    tr.synchronize()
    capi.check(L.gs_trainer_attach_comm(tr.handle, None))
    capi.check(L.gs_comm_destroy(comm))
    tr.train(proj)
    tr.close()
