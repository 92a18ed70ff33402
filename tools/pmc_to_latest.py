#!/usr/bin/env python3
"""profiles/pmc_latest.json from a tools/pmc_pass.sh summary (FETCH_SIZE / WRITE_SIZE / SQ passes).
usage: tools/pmc_to_latest.py gpurun_out/<dir>/summary.json profiles/pmc_latest.json

The file is STAMPED with the hash of the kernel sources it was collected from (tools/source_stamp.py: csrc/*.hip,
csrc/*.h, csrc/Makefile, include/gsplat.h).  bench.py reports `roofline.traffic` / `roofline.valu` only when the stamp
equals the stamp of the sources it runs from; after any kernel change the numbers are null until the counters are
collected again.  Counter units: FETCH_SIZE / WRITE_SIZE in KiB; FETCH_SIZE is doubled per the gfx950 correction of
MI355X_MICROARCH.md (an upper bound for non-streaming reads)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from source_stamp import source_stamp  # noqa: E402

src, dst = sys.argv[1], sys.argv[2]
s = json.load(open(src))
stage_of = {"gs::k_render_bwd_pair": "render_backward", "gs::k_render_bwd1": "render_backward_per_pass", "gs::k_render_fwd": "render_forward",
            "gs::k_tile_build_sort": "tile_sort", "void gs::k_preprocess<3, false>": "preprocess", "void gs::k_splat_bwd_view<3, false>": "splat_backward", "void gs::k_splat_bwd_reduce<3>": "splat_backward_reduce",
            "void gs::k_tile_scatter<true, 256>": "tile_scatter", "void gs::k_tile_count<256>": "tile_count",
            "gs::k_coarse_scatter": "scatter", "gs::k_update": "update"}
out = {"_stamp": {"source_sha256": source_stamp(), "command": "tools/pmc_pass.sh (rocprofv3 --pmc, one pass per counter group) around `python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline`; "
                  "per kernel the LARGEST dispatch (the 16-view launch)"}}
for k, st in stage_of.items():
    if k not in s:
        continue
    e = {"kernel": k}
    if "FETCH_SIZE" in s[k]:
        e["fetch_bytes_x2"] = s[k]["FETCH_SIZE"] * 1024.0 * 2.0
    if "WRITE_SIZE" in s[k]:
        e["write_bytes"] = s[k]["WRITE_SIZE"] * 1024.0
    for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_MFMA", "SQ_BUSY_CYCLES", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU",
              "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_VALU_MFMA_BUSY_CYCLES", "TCC_HIT_sum", "TCC_MISS_sum", "grid"):
        if c in s[k]:
            e[c] = s[k][c]
    out[st] = e
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1)[:800])
