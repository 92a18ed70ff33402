#!/usr/bin/env python3
"""profiles/rNN/parity_table.md from the log of `python -m pytest tests -m gpu -s` on the GPU box: the per-configuration
accounting the parity tests print (unexplained entries, worst error / budget, entries outside the legacy array-scale bar,
pixels per admissible blend, pixel-stage outliers by flip margin) — so that these numbers live in a tracked file and not only
in a scratch log.   usage: tools/parity_table.py gpurun_out/<log> profiles/r04/parity_table.md"""
import re
import sys

src, dst = sys.argv[1], sys.argv[2]
lines = [l.lstrip(".") for l in open(src, errors="replace").read().splitlines()]   # pytest's progress dots share the line with the first print of a test
groups = [
    ("Whole step against the oracle (tests/test_gpu_trainer.py::test_step_sgd_matches_oracle, tests/test_gpu_fullsize.py::test_cfg5_per_rank_load_with_fp16_sh): "
     "every averaged-gradient entry — unexplained = outside `1e-4 of sum|term| carried through the chain + decision-flip allowance`; "
     "the legacy array-scale bar (1e-4 of max(|want|, 1e-3 max|array|)) is evaluated and COUNTED, not asserted",
     re.compile(r"^\[(\d+ splats, \d+ passes|cfg5 per-rank load)")),
    ("Rasterizer seam, forward (tests/test_gpu_raster.py::_check_forward): every pixel against the admissible blends of that pixel",
     re.compile(r"^pixels \d+x\d+:")),
    ("Rasterizer seam, backward: the nine pixel-stage sums of every splat, out-of-budget splats by flip margin",
     re.compile(r"^(pixel-stage outliers|\s+flip allowance / sum)")),
    ("Rasterizer seam, backward: per-splat chain outputs outside the accounted budget (asserted zero)",
     re.compile(r"^\[seam, ")),
    ("Scene sweep (tests/test_gpu_sweep.py): 105 seeded scenes outside the fixed cases' family; per scene forward state / lists bit-exact, every "
     "pixel an admissible blend, nine sums inside their budget, chain and pass average bit for bit, densify bit-exact — all asserted; the line "
     "reports the averaged gradients against the oracle (budget incl. the three conditioning terms of DESIGN.md 5)",
     re.compile(r"^\[sweep \d+:")),
]
out = ["# Parity accounting as printed by the GPU test run", "",
       f"Source: `{src}` (`python -m pytest tests -m gpu -s` on the MI355X box).  Every line below is followed in its test by an assert; "
       "the oracle is a CPU restatement of the reference semantics — parity unpinned (DESIGN.md section 3).", ""]
for title, rx in groups:
    hit = [l.rstrip() for l in lines if rx.match(l)]
    out += [f"## {title}", ""]
    out += ["```"] + (hit if hit else ["(no such line in this log)"]) + ["```", ""]
tail = [l for l in lines if re.search(r"\d+ passed", l)]
out += ["## Test run", "", "```"] + tail[-1:] + ["```", ""]
open(dst, "w").write("\n".join(out))
print(f"wrote {dst}: " + ", ".join(str(sum(1 for l in lines if rx.match(l))) for _, rx in groups) + " lines per group")
