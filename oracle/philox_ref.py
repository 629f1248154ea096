"""Pure-Python restatement of the throughput-mode random stream.  TEST INFRASTRUCTURE ONLY.

The reference draws from numpy's global MT19937 (``MRRL.PY:189-191,215-217,232,246,267-272,351,
355,412-413``); a 2.5 KB sequential generator per environment is the wrong shape for a GPU, so the
throughput mode of the build uses a counter-based Philox4x32-10 stream (Salmon et al., SC'11)
addressed by ``(seed; env, episode, step, purpose, index)`` and its own deterministic
``log/exp/sin`` (plain IEEE mul/add only, no fused multiply-add, so host and device agree bit for
bit).  It is distribution-equivalent to the reference, not stream-equivalent; stream parity with
the reference is what the tape mode is for.

This file is the slow, obviously-correct form used to pin ``oracle/edge_oracle.c`` and, through it,
the HIP kernels.  Known-answer vectors for Philox4x32-10 are checked in tests/test_philox.py.
"""
from __future__ import annotations

import math

M0, M1 = 0xD2511F53, 0xCD9E8D57
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = 0xFFFFFFFF

# draw purposes (byte 3 of counter word 3)
P_DT, P_POIS, P_TASK_A, P_TASK_B, P_EXPL, P_EXPL_RES, P_NET, P_ACTION, P_NOISE = range(9)


def philox4x32_10(ctr, key):
    c0, c1, c2, c3 = ctr
    k0, k1 = key
    for _ in range(10):
        p0 = M0 * c0
        p1 = M1 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & MASK, p1 & MASK, ((p0 >> 32) ^ c3 ^ k1) & MASK, p0 & MASK
        k0 = (k0 + W0) & MASK
        k1 = (k1 + W1) & MASK
    return (c0, c1, c2, c3)


def block(seed, env, episode, step, purpose, idx):
    ctr = (env & MASK, episode & MASK, step & MASK, ((purpose & 0xFF) << 24) | (idx & 0xFFFFFF))
    return philox4x32_10(ctr, (seed & MASK, (seed >> 32) & MASK))


def _dbl(a, b):
    return ((a >> 5) * 67108864.0 + (b >> 6)) / 9007199254740992.0


def d0(blk):
    return _dbl(blk[0], blk[1])


def d1(blk):
    return _dbl(blk[2], blk[3])


def bounded(x, n):
    return (x * n) >> 32


def uniform(u, lo, hi):
    return lo + (hi - lo) * u


LN2 = 0.6931471805599453
INV_LN2 = 1.4426950408889634
SQRT_HALF = 0.7071067811865476


def det_log(x):
    m, e = math.frexp(x)
    if m < SQRT_HALF:
        m = m * 2.0
        e -= 1
    s = (m - 1.0) / (m + 1.0)
    z = s * s
    p = 1.0 / 25.0
    for den in (23.0, 21.0, 19.0, 17.0, 15.0, 13.0, 11.0, 9.0, 7.0, 5.0, 3.0):
        p = p * z + 1.0 / den
    p = p * z + 1.0
    return e * LN2 + 2.0 * s * p


_INV_FACT = [1.0 / math.factorial(i) for i in range(18)]


def det_exp(x):
    k = math.floor(x * INV_LN2 + 0.5)
    r = x - k * LN2
    p = _INV_FACT[14]
    for i in range(13, -1, -1):
        p = p * r + _INV_FACT[i]
    return math.ldexp(p, int(k))


TWO_PI = 6.283185307179586


def det_sin2pi(phase):
    j = math.floor(phase * 4.0 + 0.5)
    x = TWO_PI * (phase - j * 0.25)
    z = x * x
    q = int(j) & 3
    if q & 1:   # cosine series
        p = _INV_FACT[16] * 1.0
        for i, sgn in ((14, -1.0), (12, 1.0), (10, -1.0), (8, 1.0), (6, -1.0), (4, 1.0), (2, -1.0), (0, 1.0)):
            p = p * z + sgn * _INV_FACT[i]
        v = p
    else:       # sine series
        p = _INV_FACT[15] * -1.0
        for i, sgn in ((13, 1.0), (11, -1.0), (9, 1.0), (7, -1.0), (5, 1.0), (3, -1.0), (1, 1.0)):
            p = p * z + sgn * _INV_FACT[i]
        v = p * x
    return -v if q & 2 else v


POISSON_CAP = 4096


def poisson_mult(block_of, lam):
    """Knuth's multiplication method on the U stream of purpose P_POIS (two uniforms per block)."""
    limit = det_exp(-lam)
    prod = 1.0
    count = 0
    i = 0
    while True:
        blk = block_of(i >> 1)
        u = d1(blk) if (i & 1) else d0(blk)
        prod = prod * u
        if prod > limit and count < POISSON_CAP:
            count += 1
            i += 1
        else:
            return count


# ---------------------------------------------------------------------------------------------------------------------
# Policy noise of the throughput mode: the counterpart of Normal(mu, sigma).rsample() in MAPPO.get_actions
# (mappo_algorithm.py:206-213) when em_policy_step / em_gmlp_policy run with noise == NULL.  Component o of the action of
# (global env, episode, step, agent) is a Box-Muller normal from Philox block
#     (env, episode, step, (P_NOISE << 24) | (agent << agent_shift) | (o // 4))          key = seed
# (agent_shift 4 in mappo_fwd.hip: out_dim <= 16; 8 in gmlp.hip: out_dim <= 80): words (0, 1) of the block give
# components 4*(o//4) + {0, 1} (cosine, sine branch), words (2, 3) give + {2, 3}.  float32 arithmetic as in the kernels;
# the device's logf/sincosf are not bit-reproducible on the host, so this restatement pins the stream to ~1e-6, the
# float32 rounding of the transcendental functions.
# ---------------------------------------------------------------------------------------------------------------------
def _unit24(word):
    return ((word >> 8) + 0.5) / 16777216.0


def box_muller(a, b):
    """Two standard normals from two 32-bit words."""
    u1, u2 = _unit24(a), _unit24(b)
    rad = math.sqrt(-2.0 * math.log(u1))
    return rad * math.cos(TWO_PI * u2), rad * math.sin(TWO_PI * u2)


def policy_noise(seed, env, episode, step, agent, out_dim, agent_shift=4):
    """eps[out_dim] of one (env, episode, step, agent); slow scalar form."""
    out = []
    for blk_i in range((out_dim + 3) // 4):
        blk = block(seed, env, episode, step, P_NOISE, (agent << agent_shift) | blk_i)
        out.extend(box_muller(blk[0], blk[1]) + box_muller(blk[2], blk[3]))
    return out[:out_dim]


def policy_noise_np(seed, env, episode, step, agent, out_dim, agent_shift=4):
    """Vectorised form over arrays env/step/agent (numpy, uint64 lanes): returns float32 [len(env), out_dim]."""
    import numpy as np
    env = np.asarray(env, np.uint64) & np.uint64(MASK)
    step = np.asarray(step, np.uint64) & np.uint64(MASK)
    agent = np.asarray(agent, np.uint64)
    n = env.shape[0]
    out = np.empty((n, 4 * ((out_dim + 3) // 4)), np.float32)
    m0, m1, msk = np.uint64(M0), np.uint64(M1), np.uint64(MASK)
    for blk_i in range((out_dim + 3) // 4):
        c0 = env.copy()
        c1 = np.full(n, episode & MASK, np.uint64)
        c2 = step.copy()
        c3 = np.uint64(P_NOISE << 24) | (agent << np.uint64(agent_shift)) | np.uint64(blk_i)
        k0, k1 = seed & MASK, (seed >> 32) & MASK
        for _ in range(10):
            p0, p1 = m0 * c0, m1 * c2
            c0, c1, c2, c3 = ((p1 >> np.uint64(32)) ^ c1 ^ np.uint64(k0)) & msk, p1 & msk, \
                ((p0 >> np.uint64(32)) ^ c3 ^ np.uint64(k1)) & msk, p0 & msk
            k0, k1 = (k0 + W0) & MASK, (k1 + W1) & MASK
        f32 = np.float32
        for pair, (a, b) in enumerate(((c0, c1), (c2, c3))):
            u1 = ((a >> np.uint64(8)).astype(f32) + f32(0.5)) * f32(1.0 / 16777216.0)
            u2 = ((b >> np.uint64(8)).astype(f32) + f32(0.5)) * f32(1.0 / 16777216.0)
            rad = np.sqrt(f32(-2.0) * np.log(u1))
            ang = f32(6.2831853071795865) * u2
            out[:, 4 * blk_i + 2 * pair] = rad * np.cos(ang)
            out[:, 4 * blk_i + 2 * pair + 1] = rad * np.sin(ang)
    return out[:, :out_dim]
