#!/usr/bin/env python3
"""profiles/pmc_latest.json from a tools/pmc_pass.sh summary (FETCH_SIZE / WRITE_SIZE passes).
usage: tools/pmc_to_latest.py gpurun_out/pmc_c/summary.json profiles/pmc_latest.json
FETCH_SIZE / WRITE_SIZE are reported in KiB-like units of 1024 bytes... the guide's gfx950 note: FETCH_SIZE under-counts
non-streaming reads by up to 2x, so `fetch_bytes_x2` is the corrected upper bound bench.py reports as traffic."""
import json, sys
src, dst = sys.argv[1], sys.argv[2]
s = json.load(open(src))
stage_of = {"gs::k_render_bwd2": "render_backward", "gs::k_render_fwd": "render_forward", "gs::k_tile_build_sort": "tile_sort",
            "void gs::k_preprocess<3>": "preprocess", "void gs::k_splat_bwd_view<3>": "splat_backward", "gs::k_coarse_scatter": "scatter",
            "gs::k_update": "update"}
out = {}
for k, st in stage_of.items():
    if k in s and "FETCH_SIZE" in s[k] and "WRITE_SIZE" in s[k]:
        out[st] = {"kernel": k, "fetch_bytes_x2": s[k]["FETCH_SIZE"] * 1024.0 * 2.0, "write_bytes": s[k]["WRITE_SIZE"] * 1024.0,
                   "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/pmc_pass.sh) of `python3 bench.py --steps 2 --warmup 1`, "
                           "largest dispatch (16 views); counter unit 1 KiB; FETCH_SIZE doubled per the gfx950 correction (upper bound for non-streaming reads)"}
        for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_BUSY_CYCLES", "SQ_WAVES", "SQ_WAVE_CYCLES"):
            if c in s[k]:
                out[st][c] = s[k][c]
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1)[:600])
