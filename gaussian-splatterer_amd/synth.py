"""Deterministic synthetic inputs: PCG32, seed = 0x5EED0000 + cfg, one stream per array.

Splat arrays use the reference's ModelSplatsHost layout (src/ModelSplatsHost.h:16-20): flat fp32,
locations[3P], shs[3*M*P] (per splat coefficient-major, channel-minor), scales[3P], opacities[P],
rotations[4P] with element 0 the quaternion's real part (what the rasterizer reads).
"""
import numpy as np

_MULT = np.uint64(6364136223846793005)


def pcg32(seed, stream, n):
    """n outputs of PCG32 (XSH-RR 64/32) seeded as pcg32_srandom(seed, stream); vectorised LCG jump."""
    if n == 0:
        return np.zeros(0, np.uint32)
    with np.errstate(over="ignore"):
        inc = (np.uint64(stream) << np.uint64(1)) | np.uint64(1)
        s = np.uint64(0) * _MULT + inc
        s = s + np.uint64(seed)
        s = s * _MULT + inc          # state before the first output
        pw = np.empty(n, np.uint64)
        pw[0] = 1
        pw[1:] = _MULT
        A = np.cumprod(pw, dtype=np.uint64)            # a^k, k = 0..n-1
        G = np.concatenate([[np.uint64(0)], np.cumsum(A[:-1], dtype=np.uint64)])  # sum_{j<k} a^j
        old = A * s + inc * G                           # state used by output k
        xorshifted = (((old >> np.uint64(18)) ^ old) >> np.uint64(27)).astype(np.uint32)
        rot = (old >> np.uint64(59)).astype(np.uint32)
        return (xorshifted >> rot) | (xorshifted << ((np.uint32(32) - rot) & np.uint32(31)))


def uniform(seed, stream, n, lo=0.0, hi=1.0):
    u = (pcg32(seed, stream, n) >> np.uint32(8)).astype(np.float64) * (1.0 / 16777216.0)
    return (lo + (hi - lo) * u).astype(np.float32)


def random_splats(P, M, seed):
    """'random-init' splats: loc U[-4,4]^3, scale U[0.01,0.06]^3, uniform unit quaternion [w,x,y,z],
    opacity U[0.1,1], SH DC U[-1.5,1.5], higher SH U[-0.2,0.2]."""
    loc = uniform(seed, 1, 3 * P, -4.0, 4.0)
    scale = uniform(seed, 2, 3 * P, 0.01, 0.06)
    u = uniform(seed, 3, 3 * P).astype(np.float64).reshape(P, 3)
    a, b = np.sqrt(1.0 - u[:, 0]), np.sqrt(u[:, 0])
    rot = np.stack([a * np.sin(2 * np.pi * u[:, 1]), a * np.cos(2 * np.pi * u[:, 1]),
                    b * np.sin(2 * np.pi * u[:, 2]), b * np.cos(2 * np.pi * u[:, 2])], axis=1).astype(np.float32).reshape(-1)
    opac = uniform(seed, 4, P, 0.1, 1.0)
    sh = uniform(seed, 5, 3 * M * P, -0.2, 0.2).reshape(P, M, 3)
    sh[:, 0, :] = uniform(seed, 6, 3 * P, -1.5, 1.5).reshape(P, 3)
    return dict(loc=loc, sh=np.ascontiguousarray(sh).reshape(-1), scale=scale, opac=opac, rot=rot,
                count=P, M=M, D=sh_degree_for(M))


def sh_degree_for(M):
    """Active SH degree for M coefficients.  The reference's own formula is (M-1)/3 (src/ModelSplatsHost.cpp:36),
    which yields 0,1,2,5 for M = 1,4,9,16; the rasterizer treats every degree > 2 as 3."""
    return {1: 0, 4: 1, 9: 2, 16: 3}[M]


CONFIGS = {
    # cfg: (P, M, views, W, H)   -- BASELINE.json configs / SURVEY §8
    1: (1000, 4, 1, 256, 256),
    2: (10000, 1, 8, 512, 512),
    3: (100000, 16, 16, 1024, 1024),
    4: (100000, 16, 32, 1024, 1024),
    5: (1000000, 16, 64, 2048, 2048),
}


def seed_for(cfg):
    return 0x5EED0000 + cfg
