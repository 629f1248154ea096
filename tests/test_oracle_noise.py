"""CPU checks of the policy-noise restatement (oracle/philox_ref.py): the vectorised numpy form equals the scalar
pure-Python form built on the known-answer-tested Philox4x32-10 (tests/test_oracle_env.py), and the stream is N(0, 1)."""
import numpy as np

from oracle import philox_ref as p


def test_vectorised_noise_equals_scalar_form():
    env = np.array([0, 5, 2**32 - 1, 77, 12345678])
    step = np.array([0, 3, 999, 12, 500])
    agent = np.array([0, 9, 3, 255, 17])
    for shift, width in ((4, 11), (8, 65), (8, 9)):
        v = p.policy_noise_np(0x0123456789ABCDEF, env, 7, step, agent, width, shift)
        assert v.shape == (5, width) and v.dtype == np.float32
        for i in range(len(env)):
            s = p.policy_noise(0x0123456789ABCDEF, int(env[i]), 7, int(step[i]), int(agent[i]), width, shift)
            np.testing.assert_allclose(v[i], s, rtol=3e-6, atol=3e-6)


def test_noise_blocks_are_addressed_by_every_coordinate():
    base = p.policy_noise(1, 2, 3, 4, 5, 11)
    assert base != p.policy_noise(2, 2, 3, 4, 5, 11) and base != p.policy_noise(1, 3, 3, 4, 5, 11)
    assert base != p.policy_noise(1, 2, 4, 4, 5, 11) and base != p.policy_noise(1, 2, 3, 5, 5, 11)
    assert base != p.policy_noise(1, 2, 3, 4, 6, 11)
    assert base[:8] == p.policy_noise(1, 2, 3, 4, 5, 8)            # components come in blocks of four


def test_noise_is_standard_normal():
    n = 60_000
    rs = np.random.RandomState(0)
    z = p.policy_noise_np(42, rs.randint(0, 1 << 20, n), 1, rs.randint(0, 1000, n), rs.randint(0, 10, n), 11).ravel().astype(np.float64)
    m = z.size
    assert abs(z.mean()) < 5 / np.sqrt(m) and abs(z.var() - 1) < 5 * np.sqrt(2 / m) and abs((z**4).mean() - 3) < 5 * np.sqrt(96 / m)
