"""Shared helpers for the parity tests: scenes, and a wrapper that drives the C-ABI rasterizer seam
(gs_rasterize_forward / gs_rasterize_backward) the way src/Trainer.cu:334-412 drives the reference's."""
import ctypes as C
import math

import numpy as np

import gsplat_amd as gs
from gsplat_amd import capi

REC_DTYPE = np.dtype([("x", "f4"), ("y", "f4"), ("conA", "f4"), ("conB", "f4"), ("conC", "f4"), ("opacity", "f4"),
                      ("r", "f4"), ("g", "f4"), ("b", "f4"), ("hx", "f4"), ("hy", "f4"), ("depth", "f4"),
                      ("radius", "i4"), ("flags", "u4"), ("rect_min", "u4"), ("rect_max", "u4")])


def make_scene(P, M, seed, W, H, n_cams=1, fov=60.0, distance=10.0):
    s = gs.synth.random_splats(P, M, seed)
    cams = gs.camera.get_cameras(n_cams, distance, fov)
    views = gs.camera.train_views(cams, W, H)
    return s, cams, views


def view_parts(block):
    return dict(view=block[0:16].copy(), proj=block[16:32].copy(), campos=block[32:35].copy(), tanx=float(block[35]),
                tany=float(block[36]), bg=block[37:40].copy())


class SeamRaster:
    """Device-side rasterizer pair through the C-ABI seam, reference argument order."""

    def __init__(self):
        self.L = capi.lib()
        self.chunks = {}
        self._cbs = []

    def _alloc(self, name):
        def cb(nbytes, user):
            buf = capi.DeviceBuffer(nbytes)
            self.chunks[name] = buf
            return buf.ptr.value
        f = capi.ALLOC_FN(cb)
        self._cbs.append(f)
        return f

    def forward(self, s, D, M, vp, W, H, mod=1.0):
        self.P, self.D, self.M, self.W, self.H, self.mod = s["opac"].size, D, M, W, H, mod
        P = self.P
        dev = lambda a: capi.DeviceBuffer.from_numpy(np.ascontiguousarray(a, np.float32))
        self.d = dict(loc=dev(s["loc"]), sh=dev(s["sh"]), scale=dev(s["scale"]), opac=dev(s["opac"]), rot=dev(s["rot"]),
                      view=dev(vp["view"]), proj=dev(vp["proj"]), campos=dev(vp["campos"]), bg=dev(vp["bg"]))
        self.vp = vp
        self.out = capi.DeviceBuffer(3 * W * H * 4)
        R = C.c_int(-1)
        d = self.d
        capi.check(self.L.gs_rasterize_forward(
            self._alloc("geometry"), None, self._alloc("binning"), None, self._alloc("image"), None, P, D, M, d["bg"].ptr, W,
            H, d["loc"].ptr, d["sh"].ptr, None, d["opac"].ptr, d["scale"].ptr, C.c_float(mod), d["rot"].ptr, None,
            d["view"].ptr, d["proj"].ptr, d["campos"].ptr, C.c_float(vp["tanx"]), C.c_float(vp["tany"]), 0, self.out.ptr,
            None, 1, C.byref(R)))
        self.R = R.value
        return self.out.to_numpy(np.float32).reshape(3, H, W), self.R

    def field(self, chunk, name, dtype):
        off, nb = C.c_size_t(), C.c_size_t()
        capi.check(self.L.gs_raster_chunk_field(chunk.encode(), name.encode(), self.P, self.W, self.H, self.R,
                                                C.byref(off), C.byref(nb)))
        dtype = np.dtype(dtype)
        return self.chunks[chunk].to_numpy(dtype, nb.value // dtype.itemsize, off.value)

    def backward(self, dL_dpix, prefill=None):
        P, M = self.P, self.M
        sizes = dict(dL_dmean2D=3 * P, dL_dconic=4 * P, dL_dopacity=P, dL_dcolor=3 * P, dL_dmean3D=3 * P,
                     dL_dcov3D=6 * P, dL_dsh=3 * M * P, dL_dscale=3 * P, dL_drot=4 * P)
        g = {}
        for k, n in sizes.items():
            init = np.zeros(n, np.float32) if prefill is None else np.full(n, prefill.get(k, 0.0), np.float32)
            g[k] = capi.DeviceBuffer.from_numpy(init) if n else capi.DeviceBuffer(4)
        dpix = capi.DeviceBuffer.from_numpy(np.ascontiguousarray(dL_dpix, np.float32).reshape(-1))
        d, vp = self.d, self.vp
        capi.check(self.L.gs_rasterize_backward(
            P, self.D, M, self.R, d["bg"].ptr, self.W, self.H, d["loc"].ptr, d["sh"].ptr, None, d["scale"].ptr,
            C.c_float(self.mod), d["rot"].ptr, None, d["view"].ptr, d["proj"].ptr, d["campos"].ptr, C.c_float(vp["tanx"]),
            C.c_float(vp["tany"]), None, self.chunks["geometry"].ptr, self.chunks["binning"].ptr, self.chunks["image"].ptr,
            dpix.ptr, g["dL_dmean2D"].ptr, g["dL_dconic"].ptr, g["dL_dopacity"].ptr, g["dL_dcolor"].ptr,
            g["dL_dmean3D"].ptr, g["dL_dcov3D"].ptr, g["dL_dsh"].ptr, g["dL_dscale"].ptr, g["dL_drot"].ptr, 1))
        return {k: g[k].to_numpy(np.float32, sizes[k]) for k in sizes}


def oracle_forward(orc, s, D, M, vp, W, H, mod=1.0):
    r = orc.Rasterizer(np.float32)
    out, R = r.forward(D, M, vp["bg"], W, H, s["loc"], s["sh"], s["opac"], s["scale"], mod, s["rot"], vp["view"], vp["proj"],
                       vp["campos"], vp["tanx"], vp["tany"])
    return r, out, R


def assert_close_rel(name, got, want, rtol=1e-4, floor=None, max_bad_frac=0.0):
    """|got - want| <= rtol * max(|want|, floor); floor defaults to 1e-3 * max|want| (sums of many
    fp32 terms are only meaningful relative to the scale of the array)."""
    got = np.asarray(got, np.float64).reshape(-1)
    want = np.asarray(want, np.float64).reshape(-1)
    assert got.shape == want.shape, (name, got.shape, want.shape)
    if want.size == 0:
        return
    scale = np.abs(want).max()
    fl = (1e-3 * scale if floor is None else floor)
    tol = rtol * np.maximum(np.abs(want), fl) + 1e-30
    bad = np.abs(got - want) > tol
    frac = bad.mean()
    if frac > max_bad_frac:
        i = int(np.argmax(np.abs(got - want) / tol))
        raise AssertionError(f"{name}: {bad.sum()}/{bad.size} outside rtol={rtol} (worst idx {i}: got {got[i]!r} want {want[i]!r}, scale {scale!r})")


def step_budget(orc, s, D, M, W, H, views40, truths, samples, flip_margin=1e-4, chain_noise_trials=0, cond_kappa=0.0, images=None,
                atomic_seeds=(), atomic_mode=0):
    """What the averaged gradients of one iteration may differ from the oracle's by, per entry, with every part of it
    accounted for — and the oracle's averaged gradients themselves (one forward + backward per pass serves both).
    For every pass the oracle reports, per splat and pixel-stage sum q, sum|term| of the fp32 summation
    (abs9) and the decision-flip allowance (flip9: |term change| of every blend decision within `flip_margin` of its
    threshold, see tests/test_gpu_raster.py::check_pixel_stage).  The per-splat chain is linear in the nine sums and the
    HIP chain repeats the oracle's fp32 operations, so an output's budget is the sums' budget carried through the chain,
        |d out_k| <= sum_q |A_kq| * (1e-4 * abs9_q + flip9_q),    A = the chain evaluated on the nine unit inputs,
    and accumulateGradients (src/Trainer.cu:47-77) adds the passes' gradients divided by S: the budgets add the same way
    (`var` = sum of |g_loc| / S: | |a| - |b| | <= |a - b|).
    chain_noise_trials > 0 (tests/test_gpu_sweep.py): the per-splat chain is the reference's fp32 op sequence and is ILL-CONDITIONED
    for needle-shaped splats — with one scale axis 40-100 x another its dL_dscale is a difference of products 1e4-1e5 times its own
    size, so the fp32 value carries rounding noise far above 1e-4 of the sums it is made of (sweep scene 4, splat 23: the oracle's own
    dL_dscale.y = 72.415 where the exact linear map of the same sums gives 72.449).  That noise is a property of the op sequence, the same
    for every implementation of it, and it is re-drawn whenever the inputs change by more than a few ulps.  It is MEASURED on the oracle:
    the sums are perturbed by 2^-17 relative (far inside their own 1e-4 budget), the chain's exactly linear response chain(delta) is
    taken out, and what remains — chain(sums + delta) - chain(sums) - chain(delta), largest of that many trials — is the noise; the
    budget then also holds 4 x it ("noise" in the result).  The fixed-size parity cases (near-isotropic splats) run WITHOUT this term.
    cond_kappa > 0 (same sweep): the nine sums themselves are ill-conditioned where big splats are seen far along their long axis — the
    exponent, dG/dmean and dL/dalpha subtract products orders of magnitude above their difference — and the budget of the sums gains
    cond_kappa x 2^-24 x cond9, the oracle's first-order bound on what that costs any fp32 evaluation order (gs_oracle.cpp, pixel_cond);
    the flip allowance then also covers the `power > 0: skip` decision of pairs whose power lies within cond_kappa units of its own
    conditioning of zero (a pixel on the ridge line of a needle: the sign of the computed power belongs to the evaluation order).
    images (same sweep): per pass, the image the loss is formed from instead of the oracle's own — the implementation's.  The loss
    truth / 255 - render is a difference whose relative sensitivity to the render has no bound where the two are close, so where the
    forward images differ by more than rounding (needle splats: pixel_run's exp_cond) the backward is compared on the SAME dL/dpixel
    and the images are compared by the pixel check; the fixed cases leave this None and compare end to end.
    atomic_seeds (tests/test_reference_noise.py, the GPU envelope tests): besides the oracle's double-summed gradients, one run of the
    reference's OWN arithmetic per seed — the nine sums added in fp32 in a seeded arbitrary order, as upstream's atomicAdd does
    (gs_oracle.cpp, atomic_prepare / atomic_sums; atomic_mode picks the family of orders), then the unchanged chain and
    accumulateGradients: result["runs"][array] = float32 [K, entries], result["flip"][array] = the decision-flip part of the budget alone.
    Returns {array: {"budget": with flips, "sumabs": sum|term| carried through the chain, without the 1e-4 and flips,
    "want": the oracle's averaged gradient — accumulateGradients restated in fp32 numpy, bit-identical to orc.train_views
    (tests/test_gpu_trainer.py::test_step_budget_restates_accumulate_gradients)}, "num_rendered": [per pass]}."""
    P = s["opac"].size
    V = views40.shape[0]
    N = W * H
    f32 = np.float32
    S = f32(samples)
    chain_names = {"loc": ("dL_dmean3D", 3), "sh": ("dL_dsh", 3 * M), "scale": ("dL_dscale", 3), "rot": ("dL_drot", 4)}
    strides = dict(loc=3, sh=3 * M, scale=3, rot=4, opac=1, var=1)
    out = {k: {"budget": np.zeros((P, st), f32), "sumabs": np.zeros((P, st), f32), "want": np.zeros((P, st), f32), "noise": np.zeros((P, st), np.float64),
               "flip": np.zeros((P, st), f32)}
           for k, st in strides.items()}
    K = len(atomic_seeds)
    runs = {k: np.zeros((K, P, st), f32) for k, st in strides.items()}
    num_rendered = []
    truths = np.asarray(truths, np.uint32).reshape(V, N)
    for v in range(V):
        vp = view_parts(views40[v])
        r, img, R = oracle_forward(orc, s, D, M, vp, W, H)
        num_rendered.append(R)
        dpix = orc.image_int_to_loss(truths[v], img if images is None else np.asarray(images[v], np.float32).reshape(-1), W, H)
        og = r.backward(dpix, want_abs=True, flip_margin=flip_margin, want_cond=cond_kappa > 0, power_ulps=cond_kappa)
        if K:
            r.atomic_prepare(dpix)
            for j, seed in enumerate(atomic_seeds):
                ag = r.atomic_backward(seed, atomic_mode)
                am = ag["dL_dmean3D"].reshape(P, 3)
                runs["var"][j, :, 0] += np.sqrt((am[:, 0] * am[:, 0] + am[:, 1] * am[:, 1]) + am[:, 2] * am[:, 2]) / S
                for k, (n, st) in chain_names.items():
                    runs[k][j] += ag[n].reshape(P, st) / S
                runs["opac"][j, :, 0] += ag["dL_dopacity"] / S
            r.atomic_release()
        # accumulateGradients, src/Trainer.cu:47-77 (same fp32 operations in the same order as oracle/gs_oracle.cpp)
        gm = og["dL_dmean3D"].reshape(P, 3)
        out["var"]["want"][:, 0] += np.sqrt((gm[:, 0] * gm[:, 0] + gm[:, 1] * gm[:, 1]) + gm[:, 2] * gm[:, 2]) / S
        for k, (n, st) in chain_names.items():
            out[k]["want"] += og[n].reshape(P, st) / S
        out["opac"]["want"][:, 0] += og["dL_dopacity"] / S
        abs9 = og["abs9"].astype(f32)
        # (the flipped terms are fp32 sums like the others: 1e-4 of THEM too — a splat that is blended only through a flipped decision
        #  has sum|term| = 0 and must not be held to the oracle's flipped value exactly; found on a 100 000-splat scene of the sweep.
        #  Their conditioning is part of cond9 for the same reason: gs_oracle.cpp, render_backward)
        tol9 = (1e-4 * (og["abs9"] + og["flip9"]) + og["flip9"] + (cond_kappa * 2.0 ** -24 * og["cond9"] if cond_kappa > 0 else 0.0)).astype(f32)
        flp9 = og["flip9"].astype(f32)
        loc_b, loc_a, loc_f = np.zeros((P, 3), f32), np.zeros((P, 3), f32), np.zeros((P, 3), f32)
        for q in range(8):   # (sum 8, dL_dopacity, does not enter the chain)
            unit = np.zeros((P, 9), f32); unit[:, q] = 1.0
            col = orc.chain(r, unit)
            # A itself is an fp32 evaluation of the chain: where the chain cancels internally (needle splats) its entries are good to
            # a few per cent only (finite differences -0.0567 against -0.0597 on unit input), and a budget |A| x tolerance inherits that.
            # With the chain noise requested, A's own noise — same measurement, on the unit input — is added to |A| (4 x, as below).
            a_noise = None
            if chain_noise_trials:
                arng = np.random.default_rng(0xA000 + 16 * v + q)
                a_noise = {n: np.zeros(col[n].shape) for n, _ in chain_names.values()}
                for _ in range(min(chain_noise_trials, 4)):
                    shifted = (unit.astype(np.float64) * (1.0 + arng.uniform(-1.0, 1.0, (P, 1)) * 2.0 ** -17)).astype(f32)
                    pert, lin = orc.chain(r, shifted), orc.chain(r, shifted - unit)
                    for n in a_noise:
                        np.maximum(a_noise[n], np.abs(pert[n].astype(np.float64) - col[n] - lin[n]), out=a_noise[n])
            coef = lambda n: np.abs(col[n]) if a_noise is None else np.abs(col[n]) + 4.0 * a_noise[n]
            for k, (n, st) in chain_names.items():
                # which sums reach which output (tests/test_step_budget.py checks the zero blocks): the SH gradient is
                # basis x dL_dcolour (sums 0-2); scale and rotation come from the conic sums (5-7) alone
                if (k == "sh" and q >= 3) or (k in ("scale", "rot") and q not in (5, 6, 7)):
                    continue
                if k == "sh":   # colour sum q reaches channel q of every coefficient only (same test as the other zero blocks)
                    A = coef(n).reshape(P, M, 3)[:, :, q]
                    out[k]["budget"].reshape(P, M, 3)[:, :, q] += A * (tol9[:, q, None] / S)
                    out[k]["sumabs"].reshape(P, M, 3)[:, :, q] += A * (abs9[:, q, None] / S)
                    out[k]["flip"].reshape(P, M, 3)[:, :, q] += A * (flp9[:, q, None] / S)
                    continue
                A = coef(n).reshape(P, st)
                out[k]["budget"] += A * (tol9[:, q, None] / S)
                out[k]["sumabs"] += A * (abs9[:, q, None] / S)
                out[k]["flip"] += A * (flp9[:, q, None] / S)
                if k == "loc":
                    loc_b += A * tol9[:, q, None]; loc_a += A * abs9[:, q, None]; loc_f += A * flp9[:, q, None]
        if chain_noise_trials:
            sums9 = np.zeros((P, 9), f32)
            sums9[:, 0:3] = og["dL_dcolor"].reshape(P, 3)
            sums9[:, 3:5] = og["dL_dmean2D"].reshape(P, 3)[:, :2]
            sums9[:, 5:8] = og["dL_dconic"].reshape(P, 4)[:, [0, 1, 3]]
            base = orc.chain(r, sums9)
            assert all(np.array_equal(base[n].view(np.uint32), og[n].view(np.uint32)) for n, _ in chain_names.values())   # the same chain
            moved = {k: np.zeros((P, st)) for k, (n, st) in chain_names.items()}
            rng = np.random.default_rng(0xC4A1 + v)
            for _ in range(chain_noise_trials):
                shifted = (sums9.astype(np.float64) * (1.0 + rng.uniform(-1.0, 1.0, sums9.shape) * 2.0 ** -17)).astype(f32)
                pert, lin = orc.chain(r, shifted), orc.chain(r, shifted - sums9)
                for k, (n, st) in chain_names.items():
                    resid = pert[n].reshape(P, st).astype(np.float64) - base[n].reshape(P, st) - lin[n].reshape(P, st)
                    np.maximum(moved[k], np.abs(resid), out=moved[k])
            for k in chain_names:
                out[k]["noise"] += moved[k] / float(S)
            out["var"]["noise"] += np.linalg.norm(moved["loc"], axis=1, keepdims=True) / float(S)
        out["opac"]["budget"] += tol9[:, 8:9] / S
        out["opac"]["sumabs"] += abs9[:, 8:9] / S
        out["opac"]["flip"] += flp9[:, 8:9] / S
        out["var"]["flip"] += np.linalg.norm(loc_f, axis=1, keepdims=True) / S
        out["var"]["budget"] += np.linalg.norm(loc_b, axis=1, keepdims=True) / S
        out["var"]["sumabs"] += np.linalg.norm(loc_a, axis=1, keepdims=True) / S
    res = {k: {a: (b.reshape(-1) if a == "want" else b.reshape(-1).astype(np.float64)) for a, b in d.items()} for k, d in out.items()}
    if chain_noise_trials:
        for k in res:
            res[k]["budget"] = res[k]["budget"] + 4.0 * res[k]["noise"]
    res["num_rendered"] = np.asarray(num_rendered, np.int64)
    if K:
        res["runs"] = {k: a.reshape(K, -1) for k, a in runs.items()}
    return res


def envelope(runs):
    """[lo, hi] per entry over K runs of the reference's own arithmetic (float64 views of fp32 values)."""
    r = np.asarray(runs, np.float64)
    return r.min(0), r.max(0)


def envelope_distance(got, lo, hi):
    """How far outside [lo, hi] each entry of `got` lies (0 inside)."""
    got = np.asarray(got, np.float64).reshape(-1)
    return np.maximum(np.maximum(lo - got, got - hi), 0.0)


def unexplained(name, got, want, budget, stride=1, eps_rel=4e-6):
    """Entries of `got` outside  budget + eps_rel * max|want of the same splat|.  The second term is the fp32 rounding of the
    per-splat chain itself and of the division / accumulation over the passes: the components of one splat's gradient
    (`stride` per splat) come out of shared intermediates, so a component that nearly cancels — e.g. one scale axis four
    orders of magnitude below its siblings — carries rounding noise of the siblings' size, not of its own.
    Returns (count, worst ratio |error| / tolerance)."""
    got = np.asarray(got, np.float64).reshape(-1); want = np.asarray(want, np.float64).reshape(-1)
    per_splat = np.repeat(np.abs(want).reshape(-1, stride).max(1), stride)
    tol = budget + eps_rel * per_splat + 1e-37
    ratio = np.abs(got - want) / tol
    return int((ratio > 1.0).sum()), float(ratio.max()) if ratio.size else 0.0


ENVELOPE_WIDEN = 16.0     # the K = 8 run envelope, widened about its middle
ENVELOPE_ULPS = 256.0     # x 2^-24 x sum|term|: 1.5e-5, a sixth of the 1e-4 budget


def envelope_verdict(got, want, runs, sumabs, flip, budget, stride=1, widen=ENVELOPE_WIDEN, ulps=ENVELOPE_ULPS):
    """Every entry of an implementation's result against the REFERENCE'S OWN run-to-run envelope (`runs`: K runs of the reference's
    fp32 atomicAdd arithmetic in seeded orders, step_budget(atomic_seeds=...) / Rasterizer.atomic_backward) — north_star's "<= 1e-4
    relative vs the reference" restated as "indistinguishable from a second run of the reference".  Class of each entry, first that holds:
      0  within 1e-4 of the VALUE itself: |got - want| <= 1e-4 |want| (want: the double-summed, correctly rounded sum of the same terms);
      1  inside the envelope widened `widen` x about its middle (a held-out reference run is inside the 4 x widened one 99.7 % of the time);
      2  within `ulps` x 2^-24 of sum|term| carried through the chain (`sumabs`): the terms themselves are fp32 values whose exp and
         operation order differ from the oracle's in the last bits — what is left when all K runs agree bit for bit (a splat with one
         or two terms has NO order noise) or nearly so; this replaces the former 1e-4 of sum|term|, at a sixth of its size (callers scale
         it with the number of factors in the transmittance product where pixels blend thousands of entries: one ulp per factor);
      3  a NAMED decision flip: the oracle's flip analysis moved this entry (`flip` > 0) and the entry lies inside the accounted
         budget (`budget`, util.step_budget) — the one place the old budget survives;
      4  unexplained.
    No conditioning, chain-noise or A-noise term takes part.  Returns (classes int8[n], dict of rates)."""
    got = np.asarray(got, np.float64).reshape(-1)
    want = np.asarray(want, np.float64).reshape(-1)
    runs = np.asarray(runs, np.float64).reshape(len(runs), -1)
    lo, hi = runs.min(0), runs.max(0)
    mid, hw = (lo + hi) / 2, (hi - lo) / 2
    err = np.abs(got - want)
    per_splat = np.repeat(np.abs(want).reshape(-1, stride).max(1), stride)
    cls = np.full(got.size, 4, np.int8)
    named = (np.asarray(flip).reshape(-1) > 0) & (err <= np.asarray(budget).reshape(-1) + 4e-6 * per_splat + 1e-37)
    cls[named] = 3
    cls[err <= ulps * 2.0 ** -24 * np.asarray(sumabs).reshape(-1)] = 2
    cls[np.abs(got - mid) <= widen * hw] = 1
    cls[err <= 1e-4 * np.abs(want)] = 0
    # the yardstick: each reference run against the envelope of the other K - 1
    K = runs.shape[0]
    loo_in, loo_4 = [], []
    for j in range(K):
        o = np.delete(runs, j, axis=0)
        l2, h2 = o.min(0), o.max(0)
        m2, w2 = (l2 + h2) / 2, (h2 - l2) / 2
        loo_in.append(float((np.abs(runs[j] - m2) <= w2).mean()))
        loo_4.append(float((np.abs(runs[j] - m2) <= 4 * w2).mean()))
    rates = dict(n=int(got.size), strict=float((err <= 1e-4 * np.abs(want)).mean()), inside=float((np.abs(got - mid) <= hw).mean()),
                 within4=float((np.abs(got - mid) <= 4 * hw).mean()), within16=float((np.abs(got - mid) <= widen * hw).mean()),
                 ref_inside=float(np.mean(loo_in)), ref_within4=float(np.mean(loo_4)),
                 by_ulps=int((cls == 2).sum()), by_flip=int((cls == 3).sum()), unexplained=int((cls == 4).sum()))
    bad = cls == 4
    if bad.any():     # what an unexplained entry looks like, for the assertion message
        sa = np.asarray(sumabs, np.float64).reshape(-1)
        i = int(np.flatnonzero(bad)[np.argmax((err / np.maximum(sa, 1e-300))[bad])])
        rates["worst_unexplained"] = dict(index=i, got=float(got[i]), want=float(want[i]), lo=float(lo[i]), hi=float(hi[i]),
                                          ulps_of_sumabs=float(err[i] / max(2.0 ** -24 * sa[i], 1e-300)), flip=float(np.asarray(flip).reshape(-1)[i]),
                                          budget=float(np.asarray(budget).reshape(-1)[i]))
    return cls, rates


def unexplained_bytes(frame, want_img, w, h, rtol=1e-4, floor=1e-3, details=None):
    """An RGBA8 frame (uint32[h * w], imageFloatToInt layout: R in the low byte, src/Trainer.cu:19-29) against the oracle's FLOAT image
    [3][h][w] of the same render.  A byte may differ from imageFloatToInt(oracle float) only by ONE step and only where the oracle's
    float v lies within the forward pixel tolerance — rtol * max(|v|, floor), the bar every float pixel is held to — of the boundary
    k / 256 between the two byte values: (int)(v * 256) steps exactly there, so a float inside the tolerance may land on either side.
    Returns (bytes that differ, bytes that differ WITHOUT such an explanation); the alpha byte must be 0xFF everywhere."""
    frame = np.asarray(frame, np.uint32).reshape(-1)
    v = np.asarray(want_img, np.float32).reshape(3, -1).astype(np.float64)
    assert frame.size == w * h and v.shape[1] == w * h
    assert np.all((frame >> 24) == 0xFF)
    n_diff = n_bad = 0
    for c in range(3):
        got = ((frame >> (8 * c)) & 0xFF).astype(np.int64)
        want = np.clip((v[c].astype(np.float32) * np.float32(256.0)).astype(np.int64), 0, 255)     # (int) truncates towards zero, as the kernel's cast
        d = got != want
        if not d.any():
            continue
        n_diff += int(d.sum())
        boundary = np.maximum(got, want)[d] / 256.0
        tol = rtol * np.maximum(np.abs(v[c][d]), floor)
        ok = (np.abs(got - want)[d] == 1) & (np.abs(v[c][d] - boundary) <= tol)
        n_bad += int((~ok).sum())
        if details is not None:     # (channel, pixel, oracle float, byte got, byte wanted, distance to the boundary / max(|v|, floor))
            idx = np.flatnonzero(d)
            for j in np.flatnonzero(~ok):
                details.append((c, int(idx[j]), float(v[c][idx[j]]), int(got[idx[j]]), int(want[idx[j]]),
                                float(abs(v[c][idx[j]] - boundary[j]) / max(abs(v[c][idx[j]]), floor))))
    return n_diff, n_bad
