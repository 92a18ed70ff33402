"""GPU: the reference's canonical start state (initFieldGrid) driven by the auto-train loop against truth
images rendered from a target splat set — capture / densify intervals, model re-indexing, .gobj export."""
import numpy as np
import pytest

import gsplat_amd as gs
from gsplat_amd import capi

pytestmark = pytest.mark.gpu


def test_auto_train_on_grid_field(tmp_path):
    W = H = 128
    target = gs.synth.random_splats(600, 4, 77)
    thost = gs.ModelSplatsHost.fromVectors(target["loc"], target["sh"], target["scale"], target["opac"], target["rot"])
    painter = gs.Trainer(W, H)                       # stands in for the reference's OptiX truth renderer
    painter.model = gs.ModelSplatsDevice(thost)

    def capture(cameras):
        fw = [painter.render(W, H, 1.0, c, background=(1.0, 1.0, 1.0)).reshape(-1) for c in cameras]
        fb = [painter.render(W, H, 1.0, c, background=(0.0, 0.0, 0.0)).reshape(-1) for c in cameras]
        return fw, fb

    tr = gs.Trainer(W, H)
    tr.model = gs.ModelSplatsDevice(gs.fields.initFieldGrid())
    assert tr.model.count == 4913
    p = gs.Project.initProject()
    p.sphere1.count, p.intervalCapture, p.intervalDensify = 3, 3, 4
    p.paramCullOpacity, p.lrOpacity = 0.999, 0.0     # iteration 0 densifies: opacity-1 grid splats survive (1.0 > 0.999) ...
    drv = gs.driver.AutoTrainer(tr, p, capture)
    flags = [drv.step() for _ in range(6)]
    assert [f[0] for f in flags] == [True, False, False, True, False, False]
    assert [f[1] for f in flags] == [True, False, False, False, True, False]
    assert p.iterations == 6 and len(tr.truthCameras) == 3 and len(tr.truthFrameBuffersW) == 3
    n = tr.model.count
    assert 0 < n <= 1000000 and n != 4913   # densify at iterations 0 and 4 re-indexed the model (clones, splits, prunes)
    host = gs.ModelSplatsHost.fromDevice(tr.model)
    assert np.isfinite(host.locations[:3 * n]).all() and np.isfinite(host.shs[:12 * n]).all()
    gs.io.saveSplats(tmp_path / "out.gobj", host)
    back = gs.io.loadSplats(tmp_path / "out.gobj")
    assert back.count == n and np.allclose(back.opacities[:n], host.opacities[:n], rtol=1e-5)
    tr.close(); painter.close()
