"""tests/util.py::step_budget on the CPU: its fp32 numpy restatement of accumulateGradients (src/Trainer.cu:47-77) equals
the oracle's orc_train_views bit for bit (the GPU parity tests take the oracle's averaged gradients from it, one oracle
pass per view instead of two), and the blocks of the chain's matrix it skips are exactly zero."""
import numpy as np

import gsplat_amd as gs
from util import oracle_forward, step_budget, view_parts


def test_step_budget_restates_accumulate_gradients(orc):
    P, M, W, H, n = 400, 4, 64, 48, 2
    s = gs.synth.random_splats(P, M, 3)
    views = gs.camera.train_views(gs.camera.get_cameras(n), W, H)
    truths = np.random.default_rng(0).integers(0, 2 ** 32, (2 * n, W * H), dtype=np.uint32)
    o = orc.train_views(P, 1, M, W, H, s["loc"], s["sh"], s["scale"], s["opac"], s["rot"], views, truths, 2.0 * n)
    b = step_budget(orc, s, 1, M, W, H, views, truths, 2.0 * n)
    for k in ("loc", "sh", "scale", "opac", "rot", "var"):
        assert np.array_equal(b[k]["want"].view(np.uint32), o[k].view(np.uint32)), k
        assert b[k]["budget"].shape == o[k].shape and (b[k]["budget"] >= 1e-4 * b[k]["sumabs"] * (1 - 1e-6)).all()
    assert np.array_equal(b["num_rendered"], o["num_rendered"])
    assert np.abs(o["loc"]).max() > 0 and b["loc"]["sumabs"].max() > 0


def test_chain_matrix_zero_blocks(orc):
    """The sums that do not reach an output: step_budget skips those blocks of the chain's matrix."""
    P, M, W, H = 300, 16, 64, 64
    s = gs.synth.random_splats(P, M, 8)
    views = gs.camera.train_views(gs.camera.get_cameras(1), W, H)
    r, _, _ = oracle_forward(orc, s, 3, M, view_parts(views[0]), W, H)
    for q in range(9):
        unit = np.zeros((P, 9), np.float32); unit[:, q] = 1.0
        col = orc.chain(r, unit)
        if q >= 3:
            assert not col["dL_dsh"].any()
        else:   # colour sum q reaches channel q of every SH coefficient only
            sh = col["dL_dsh"].reshape(P, M, 3)
            assert not np.delete(sh, q, axis=2).any() and sh[:, :, q].any()
        if q not in (5, 6, 7):
            assert not col["dL_dscale"].any() and not col["dL_drot"].any()
        if q == 8:
            assert not col["dL_dmean3D"].any()
