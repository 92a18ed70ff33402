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
                          "--no-cpu-baseline"] + extra, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


@pytest.mark.parametrize("collective", ["torch", "rccl", "torch-sharded", "rccl-sharded"])
def test_collective_hook_runs_with_one_rank(collective):
    base = _run([])
    got = _run(["--force-dist", "--collective", collective])
    assert "collective" in got["stages_ms_per_launch"], got["stages_ms_per_launch"]
    assert got["config"]["mean_num_rendered_per_view"] == base["config"]["mean_num_rendered_per_view"]
    assert got["value"] > 0
