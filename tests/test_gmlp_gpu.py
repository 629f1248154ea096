"""GPU parity of the grouped MLP / per-agent PPO kernels (csrc/gmlp.hip) through the C ABI against the plain
PyTorch-CPU float32 restatement ``oracle/ippo_oracle.py`` (tolerance: 1e-5 relative on forward values, log-probs and
loss sums, as north_star states for fp32; gradients 2e-4 relative to the tensor's largest entry because their
summation order over the batch differs; bit-exact for the integer placement decisions).

Parity with the REFERENCE is unpinned for independent PPO (RLlib is not vendored); with one group these kernels compute
the reference's shared-actor arithmetic that tests/test_mappo_gpu.py pins to the recorded reference run."""
import numpy as np
import pytest
import torch

from oracle import ippo_oracle as io

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _close(got, want, rtol=1e-5, scale=None):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    s = np.abs(want).max() if scale is None else scale
    np.testing.assert_allclose(got, want, rtol=rtol, atol=rtol * max(s, 1e-30))


def _setup(G, in_dim, out_dim, rows, with_ls, seed=0):
    from marl_amd import gmlp
    net = gmlp.GroupedMLP(G, in_dim, out_dim, with_ls, DEV, seed=seed)
    gen = torch.Generator().manual_seed(seed + 1)
    if with_ls:   # non-trivial log_std
        o, n = net.offsets["log_std"]
        net.flat[:, o:o + n] = (torch.rand((G, n), generator=gen) * 0.6 - 0.4).to(DEV)
    x = torch.randn((rows, in_dim), generator=gen)
    agent = torch.randint(0, G, (rows,), generator=gen)
    if G > 2:
        agent[agent == 1] = 0       # an empty group
    return net, x, agent


@pytest.mark.parametrize("G,in_dim,out_dim,rows,split", [(7, 68, 65, 1000, 1), (1, 68, 65, 777, 5), (5, 68, 1, 333, 2),
                                                         (3, 14, 11, 50, 1), (4, 80, 80, 260, 3)])
def test_grouped_forward(G, in_dim, out_dim, rows, split):
    from marl_amd import gmlp
    net, x, agent = _setup(G, in_dim, out_dim, rows, False)
    off, perm = gmlp.group_index(agent.to(DEV), G)
    y = gmlp.forward(net, x.to(DEV).contiguous(), off if G > 1 else None, perm if G > 1 else None, split)
    want = io.forward(net.flat.cpu(), in_dim, out_dim, False, x, agent)
    _close(y.cpu(), want)


@pytest.mark.parametrize("G,explore", [(9, True), (1, True), (6, False)])
def test_policy_epilogue(G, explore):
    from marl_amd import gmlp
    S, rows = 64, 700
    net, x, agent = _setup(G, S + 4, S + 1, rows, True, seed=3)
    gen = torch.Generator().manual_seed(9)
    noise = torch.randn((rows, S + 1), generator=gen)
    x[5] = 0.0
    off, perm = gmlp.group_index(agent.to(DEV), G)
    ent_env = torch.arange(rows, dtype=torch.int32, device=DEV)
    dst = torch.randperm(rows, generator=gen).to(torch.int32).to(DEV)
    placement = torch.full((rows,), -7, dtype=torch.int32, device=DEV)
    resource = torch.zeros(rows, dtype=torch.float32, device=DEV)
    act = torch.zeros((rows, S + 1), dtype=torch.float32, device=DEV)
    logp = torch.zeros(rows, dtype=torch.float32, device=DEV)
    gmlp.policy(net, x.to(DEV).contiguous(), ent_env, agent.to(torch.int32).to(DEV), dst, placement, resource, act, logp,
                off if G > 1 else None, perm if G > 1 else None, split=2 if G == 1 else 1, noise=noise.to(DEV).contiguous(),
                explore=explore)
    flat = net.flat.cpu()
    mu = io.forward(flat, S + 4, S + 1, True, x, agent)
    ls_rows = torch.stack([io.unpack(flat[g], S + 4, S + 1, True)["log_std"] for g in agent.tolist()])
    if explore:
        a, lp = io.sample(mu, ls_rows, noise)
    else:
        a, lp = mu, torch.zeros(rows)
    _close(act.cpu(), a)
    _close(logp.cpu(), lp, rtol=2e-5)
    # decisions from the kernel's own actions (a 1-ulp difference in mu must not flip an argmax in this test)
    place, res = io.format_actions(act.cpu(), S)
    d = dst.cpu().long()
    assert torch.equal(placement.cpu()[d], place)
    assert torch.equal(resource.cpu()[d], res)


def test_value_epilogue():
    from marl_amd import gmlp
    G, rows = 11, 500
    net, x, agent = _setup(G, 68, 1, rows, False, seed=5)
    off, perm = gmlp.group_index(agent.to(DEV), G)
    out = torch.zeros(rows, dtype=torch.float32, device=DEV)
    gmlp.value(net, x.to(DEV).contiguous(), out, off, perm, norm_mean=0.25, norm_var=4.0)
    want = io.forward(net.flat.cpu(), 68, 1, False, x, agent).squeeze(-1) * np.sqrt(4.0 + 1e-8) + 0.25
    _close(out.cpu(), want)


@pytest.mark.parametrize("G,rows,split,out_dim", [(6, 900, 1, 65), (1, 1500, 7, 65), (3, 200, 2, 11)])
def test_actor_gradients(G, rows, split, out_dim):
    from marl_amd import gmlp
    in_dim = 68 if out_dim == 65 else 14
    net, x, agent = _setup(G, in_dim, out_dim, rows, True, seed=7)
    if G > 2:
        agent[0] = 2
        agent[1:][agent[1:] == 2] = 3 % G     # group 2 holds exactly one transition
    gen = torch.Generator().manual_seed(21)
    flat = net.flat.cpu()
    mu = io.forward(flat, in_dim, out_dim, True, x, agent)
    ls_rows = torch.stack([io.unpack(flat[g], in_dim, out_dim, True)["log_std"] for g in agent.tolist()])
    action, lp = io.sample(mu, ls_rows, torch.randn((rows, out_dim), generator=gen))
    old_logp = lp + 0.3 * torch.randn(rows, generator=gen)      # ratios on both sides of the clip range
    adv = torch.randn(rows, generator=gen) * 3 + 1
    off, perm = gmlp.group_index(agent.to(DEV), G)
    moments = torch.zeros((G, 3), dtype=torch.float64, device=DEV)
    gmlp.group_moments(adv.to(DEV), off, perm, G, moments)
    for g in range(G):
        a = adv[agent == g].double()
        np.testing.assert_allclose(moments[g].cpu().numpy(), [a.numel(), a.sum().item(), (a * a).sum().item()], rtol=1e-12)
    sf = net.slab_floats()
    slab = torch.empty((G, 4 * split, sf), dtype=torch.float32, device=DEV)
    grads = torch.zeros((G, sf - 3), dtype=torch.float32, device=DEV)
    loss3 = torch.zeros((G, 3), dtype=torch.float32, device=DEV)
    gmlp.actor_grad(net, x.to(DEV).contiguous(), action.to(DEV).contiguous(), old_logp.to(DEV), adv.to(DEV), moments, grads,
                    loss3, slab, off if G > 1 else None, perm if G > 1 else None, split)
    want_g, want_l = io.actor_grads(flat, in_dim, out_dim, x, action, old_logp, adv, agent)
    _close(loss3.cpu(), want_l, rtol=2e-5)
    for g in range(G):
        for name, (o, n) in net.offsets.items():
            _close(grads[g, o:o + n].cpu(), want_g[g, o:o + n], rtol=2e-4, scale=float(want_g[g].abs().max()))


def test_critic_gradients_and_adam():
    from marl_amd import gmlp
    G, rows = 5, 800
    net, x, agent = _setup(G, 68, 1, rows, False, seed=11)
    gen = torch.Generator().manual_seed(2)
    ret = torch.randn(rows, generator=gen) * 5
    off, perm = gmlp.group_index(agent.to(DEV), G)
    moments = torch.zeros((G, 3), dtype=torch.float64, device=DEV)
    gmlp.group_moments(ret.to(DEV), off, perm, G, moments)       # only the counts are used by the critic
    sf = net.slab_floats()
    slab = torch.empty((G, 4, sf), dtype=torch.float32, device=DEV)
    grads = torch.zeros((G, sf - 3), dtype=torch.float32, device=DEV)
    loss3 = torch.zeros((G, 3), dtype=torch.float32, device=DEV)
    flat0 = net.flat.cpu().clone()
    gmlp.critic_grad(net, x.to(DEV).contiguous(), ret.to(DEV), moments, grads, loss3, slab, off, perm, 1, value_loss_coeff=0.5)
    want_g, want_l = io.critic_grads(flat0, 68, x, ret, agent)
    _close(loss3.cpu(), want_l, rtol=2e-5)
    for g in range(G):
        _close(grads[g].cpu(), want_g[g], rtol=2e-4)
    # clip + Adam on the kernel's own gradients, three steps
    m = torch.zeros_like(net.flat)
    v = torch.zeros_like(net.flat)
    rm, rv, rflat = torch.zeros_like(flat0), torch.zeros_like(flat0), flat0.clone()
    norms = torch.zeros(G, dtype=torch.float32, device=DEV)
    gcpu = grads.cpu()
    for step in (1, 2, 3):
        gmlp.group_adam(net.flat, grads, m, v, step, norms_out=norms)
        rn = io.clip_adam(rflat, gcpu, rm, rv, step)
        _close(norms.cpu(), rn)
        _close(net.flat.cpu(), rflat, rtol=1e-6)
        _close(m.cpu(), rm, rtol=1e-6)
        _close(v.cpu(), rv, rtol=1e-6)


@pytest.mark.parametrize("n_groups,n_per", [(3, 50_001), (1, 1_118_465), (256, 12_866)])
def test_clip_adam_with_several_workgroups_per_group(n_groups, n_per):
    """em_group_adam splits a group over several workgroups from 8,192 parameters (the scaled topology's central critic
    is one group of 1.1 M): every workgroup clips by the same norm, every parameter is updated exactly once."""
    from marl_amd import gmlp
    gen = torch.Generator().manual_seed(7)
    flat = torch.randn((n_groups, n_per), generator=gen) * 0.1
    grads = torch.randn((n_groups, n_per), generator=gen) * torch.logspace(-3, 0, n_groups).view(-1, 1)   # clipped and unclipped groups
    if n_per > 1_000_000:
        grads *= 0.1        # unclipped: the torch restatement's float32 norm of a million elements is 1e-5 off, and a clip
                            # coefficient built on it would carry that into every parameter
    p, g = flat.to(DEV), grads.to(DEV)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    rm, rv, rflat = torch.zeros_like(flat), torch.zeros_like(flat), flat.clone()
    norms = torch.zeros(n_groups, dtype=torch.float32, device=DEV)
    for step in (1, 2):
        gmlp.group_adam(p, g, m, v, step, norms_out=norms)
        rn = io.clip_adam(rflat, grads, rm, rv, step)
        # the kernel sums the squares in float64; torch's float32 norm of a million elements is 1e-5 off that
        _close(norms.cpu(), grads.double().norm(dim=1), rtol=1e-6)
        _close(norms.cpu(), rn, rtol=5e-5)
        _close(p.cpu(), rflat, rtol=1e-6)
        _close(m.cpu(), rm, rtol=1e-6)
        _close(v.cpu(), rv, rtol=1e-6)


def test_gae_of_independent_learners():
    from marl_amd import _lib
    import ctypes as C
    rs = np.random.RandomState(4)
    T, B, D = 60, 37, 12
    k_tb = rs.randint(0, 4, size=(T, B)).astype(np.uint8)
    first_tb = np.zeros((T, B), np.int32)
    E = int(k_tb.sum())
    first_tb.reshape(-1)[:] = np.concatenate([[0], np.cumsum(k_tb.reshape(-1))[:-1]])
    ent_agent = np.zeros(E, np.int32)
    for t in range(T):
        for b in range(B):
            k, f = int(k_tb[t, b]), int(first_tb[t, b])
            ent_agent[f:f + k] = rs.choice(D, size=k, replace=False)
    ent_value = rs.randn(E).astype(np.float32)
    reward = rs.randn(T, B)
    done = np.zeros((T, B), np.uint8)
    done[T - 1] = 1
    done[T // 2, ::5] = 1
    want_adv, want_ret = io.gae_agents(reward, done, k_tb, first_tb, ent_agent, ent_value)
    d = lambda a: torch.as_tensor(a).to(DEV).contiguous()  # noqa: E731
    t_r, t_d, t_k, t_f, t_a, t_v = d(reward), d(done), d(k_tb), d(first_tb), d(ent_agent), d(ent_value)
    state = torch.zeros((B, D, 2), dtype=torch.float64, device=DEV)
    adv = torch.zeros(E, dtype=torch.float32, device=DEV)
    ret = torch.zeros(E, dtype=torch.float32, device=DEV)
    _lib.check(_lib.lib().em_gae_agents(B, T, D, t_r.data_ptr(), t_d.data_ptr(), t_k.data_ptr(), t_f.data_ptr(), t_a.data_ptr(),
                                        t_v.data_ptr(), 0.99, 0.95, state.data_ptr(), adv.data_ptr(), ret.data_ptr(), None))
    torch.cuda.synchronize()
    assert np.array_equal(adv.cpu().numpy(), want_adv) and np.array_equal(ret.cpu().numpy(), want_ret)


@pytest.mark.parametrize("rows,groups", [(0, 4), (1, 1), (63, 5), (2048, 256), (2049, 256), (100_003, 7), (1_500_000, 256),
                                         (300_000, 1000), (5_000_000, 64)])
def test_group_index_is_the_stable_sort(rows, groups):
    """em_group_index (device counting sort: group offsets + position -> row permutation for the grouped kernels, one PPO
    learner per agent as in MRRL.PY:611-638) against torch's stable sort and bincount: identical, i.e. rows of a group keep
    their order; ragged tails, a single group, more groups than a wavefront has lanes, empty input."""
    from marl_amd import gmlp
    gen = torch.Generator().manual_seed(rows + groups)
    key = torch.randint(0, groups, (rows,), generator=gen, dtype=torch.int32)
    if rows > 10:      # an empty group and a run of equal keys across a 64-row pass boundary
        key[key == groups // 2] = 0
        key[40:90] = groups - 1
    key = key.to(DEV)
    off, perm = gmlp.group_index(key, groups)
    want_perm = torch.sort(key.to(torch.int64), stable=True).indices.to(torch.int32)
    want_off = torch.zeros(groups + 1, dtype=torch.int64, device=DEV)
    want_off[1:] = torch.cumsum(torch.bincount(key.to(torch.int64), minlength=groups), 0)
    assert torch.equal(off, want_off)
    assert torch.equal(perm, want_perm)
    off8, perm8 = gmlp.group_index(key.to(torch.int8) if groups <= 127 else key.to(torch.int64), groups)    # other key dtypes
    assert torch.equal(off8, want_off) and torch.equal(perm8, want_perm)
