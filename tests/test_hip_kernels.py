"""Numerics of every gfx950 kernel against a plain PyTorch fp32 reference of the same op.
All tests need a real MI355X and the in-tree libdrl_hip.so (no fallback: a missing library fails loudly)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from distributed_rl_amd.ops import oracle  # noqa: E402


@pytest.fixture(scope="module")
def hip():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from distributed_rl_amd.ops import hip as H
    from distributed_rl_amd import native
    native.hip()          # raises if the extension was not built
    return H


DEV = "cuda"


def bf(x):
    return x.to(torch.bfloat16).contiguous()


# ---- sum-tree ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cap", [100, 5000, 300_000])
def test_sumtree_ingest_total_sample_update(hip, cap):
    H = hip
    lay = H.TreeLayout(cap)
    tree = torch.zeros(lay.total, device=DEV)
    gen = torch.zeros(cap, dtype=torch.int32, device=DEV)
    scratch = torch.zeros(cap, dtype=torch.int32, device=DEV)
    g = torch.Generator(device="cpu").manual_seed(cap)
    n = cap * 3 // 4
    prio = torch.rand(n, generator=g) + 0.01
    prio[::7] = 0.0
    H.sumtree_ingest(tree, lay, 0, prio.to(DEV), gen, scratch)
    st = H.new_step_state(DEV, 0.4, 1000)
    n_items = torch.tensor([n], dtype=torch.int64, device=DEV)
    B = 4096
    idx = torch.zeros(B, dtype=torch.int32, device=DEV)
    p = torch.zeros(B, device=DEV)
    counts = np.zeros(cap, dtype=np.int64)
    rounds = 40
    for r in range(rounds):
        H.step_begin(st, n_items)
        H.sumtree_sample(tree, lay, st, 1234, idx, p)
        i = idx.cpu().numpy()
        assert np.allclose(p.cpu().numpy(), prio.numpy()[i])          # returned priority is the leaf's
        counts += np.bincount(i, minlength=cap)
    s = H.read_step_state(st)
    assert s["total_prio"] == pytest.approx(float(prio.double().sum()), rel=1e-5)
    assert s["rng_ctr"] == rounds and s["n_items"] == n
    assert counts[n:].sum() == 0 and counts[:n][prio.numpy() == 0].sum() == 0   # never an empty / zero leaf
    # distribution: compare aggregated over 64 equal-count buckets (chi-square, 63 dof)
    order = np.argsort(prio.numpy())
    buckets = np.array_split(order, 64)
    obs = np.array([counts[b].sum() for b in buckets], dtype=np.float64)
    exp = np.array([prio.numpy()[b].sum() for b in buckets], dtype=np.float64)
    exp = exp / exp.sum() * obs.sum()
    keep = exp > 5
    chi2 = ((obs[keep] - exp[keep]) ** 2 / exp[keep]).sum()
    assert chi2 < 140, chi2
    # update with duplicates + stale generation
    ui = torch.tensor([3, 3, 10, 11], dtype=torch.int32, device=DEV)
    uv = torch.tensor([5.0, 5.0, 7.0, 9.0], device=DEV)
    ge = gen[ui.long()].clone()
    ge[3] += 1                                                          # slot 11 "was overwritten since sampling"
    H.sumtree_update(tree, lay, ui, uv, gen, ge)
    ref = prio.clone()
    ref[3], ref[10] = 5.0, 7.0
    leaves = tree[:cap].cpu()
    assert torch.equal(leaves[:n], ref)
    H.step_begin(st, n_items)
    H.sumtree_sample(tree, lay, st, 1, idx, p)
    assert H.read_step_state(st)["total_prio"] == pytest.approx(float(ref.double().sum()), rel=1e-5)
    # every level is the exact fp32 sum of its 64 children (no drift)
    t = tree.cpu()
    for lvl in range(1, lay.n_levels):
        child = t[lay.offsets[lvl - 1]:lay.offsets[lvl - 1] + lay.sizes[lvl - 1]].view(-1, 64)
        par = t[lay.offsets[lvl]:lay.offsets[lvl] + child.shape[0]]
        assert torch.allclose(par, child.sum(1), rtol=1e-6, atol=0)
    # ring wrap-around ingest bumps generations and overwrites the oldest slots
    H.sumtree_ingest(tree, lay, cap - 2, torch.tensor([1.5, 2.5, 3.5, 4.5], device=DEV), gen, scratch)
    leaves = tree[:cap].cpu()
    assert leaves[cap - 2] == 1.5 and leaves[cap - 1] == 2.5 and leaves[0] == 3.5 and leaves[1] == 4.5
    assert gen[0].item() == 2 and gen[cap - 1].item() == 1 + (1 if cap - 1 < n else 0)


@pytest.mark.parametrize("cap", [50, 4000, 300_000, 2_000_000])
def test_sumtree_fused_update_and_step(hip, cap):
    """2-launch update (leaves + level 1, then all upper levels from level 1) with clustered duplicates, and the
    fused step_begin / step_end variants."""
    H = hip
    lay = H.TreeLayout(cap)
    tree = torch.zeros(lay.total, device=DEV)
    scratch = torch.zeros(cap, dtype=torch.int32, device=DEV)
    g = torch.Generator().manual_seed(cap + 1)
    prio = torch.rand(cap, generator=g) + 0.01
    H.sumtree_ingest(tree, lay, 0, prio.to(DEV), None, scratch)
    st = H.new_step_state(DEV, 0.4, 100)
    n_items = torch.tensor([cap], dtype=torch.int64, device=DEV)
    B = 512
    # clustered indices: many hits inside the same level-1 node, exact duplicates with different values
    base = torch.randint(0, cap, (B,), generator=g)
    base[:64] = (base[0] // 64 * 64 + torch.randint(0, 64, (64,), generator=g)).clamp(max=cap - 1)
    base[100:110] = base[100]
    ui = base.int().to(DEV)
    uv = (torch.rand(B, generator=g) + 0.5)
    ref = prio.clone()
    for j in range(B):
        ref[base[j]] = uv[j]                       # batch order, last entry wins
    H.sumtree_update(tree, lay, ui, uv.to(DEV), end_state=st)
    t = tree.cpu()
    assert torch.equal(t[:cap], ref)
    for lvl in range(1, lay.n_levels):
        child = t[lay.offsets[lvl - 1]:lay.offsets[lvl - 1] + lay.sizes[lvl - 1]].view(-1, 64)
        par = t[lay.offsets[lvl]:lay.offsets[lvl] + child.shape[0]]
        torch.testing.assert_close(par, child.sum(1), rtol=2e-6, atol=0)
    assert H.read_step_state(st)["step"] == 1     # step_end rode along
    idx, p = torch.zeros(B, dtype=torch.int32, device=DEV), torch.zeros(B, device=DEV)
    H.sumtree_sample(tree, lay, st, 7, idx, p, begin_n_items=n_items)
    s1 = H.read_step_state(st)
    assert s1["n_items"] == cap and s1["beta"] == pytest.approx(0.4 + 0.6 / 100)
    assert s1["total_prio"] == pytest.approx(float(ref.double().sum()), rel=1e-5)
    first = idx.clone()
    H.sumtree_update(tree, lay, idx, p, end_state=st)          # same values: tree unchanged, step -> 2
    H.sumtree_sample(tree, lay, st, 7, idx, p, begin_n_items=n_items)
    assert not torch.equal(first, idx)            # fresh Philox counter per step without step_begin


def test_gather_rows(hip):
    src = torch.randint(0, 256, (50, 28224), dtype=torch.uint8, device=DEV)
    off = (torch.tensor([4, 0, 49, 4, 17], dtype=torch.int64, device=DEV) * 28224)
    dst = torch.zeros(5, 28224, dtype=torch.uint8, device=DEV)
    hip.gather_rows(src, off, dst, 28224)
    assert torch.equal(dst, src[[4, 0, 49, 4, 17]])


# ---- TD kernels ----------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,A", [(1, 4), (50, 6), (512, 4), (512, 18), (1500, 9)])
def test_apex_td(hip, B, A):
    H = hip
    g = torch.Generator().manual_seed(B * 31 + A)
    cap = 4 * B + 7
    q_on, q_tg = torch.randn(2 * B, A, generator=g), torch.randn(B, A, generator=g)
    idx = torch.randint(0, cap, (B,), generator=g).int()
    act, rew = torch.randint(0, A, (cap,), generator=g).int(), torch.randn(cap, generator=g) * 2
    done = (torch.rand(cap, generator=g) < 0.2).float()
    p = torch.rand(B, generator=g) + 0.05
    st = H.new_step_state(DEV, 0.4, 100)
    n_items = torch.tensor([cap], dtype=torch.int64, device=DEV)
    # put a known total into the state through a real tree
    lay = H.TreeLayout(cap)
    tree = torch.zeros(lay.total, device=DEV)
    allp = torch.rand(cap, generator=g) + 0.05
    allp[idx.long()] = p
    p = allp[idx.long()]
    H.sumtree_ingest(tree, lay, 0, allp.to(DEV), None, torch.zeros(cap, dtype=torch.int32, device=DEV))
    for _ in range(31):
        H.step_end(st)                              # step = 31 -> beta = 0.4 + 0.6*31/100
    H.step_begin(st, n_items)
    H.sumtree_sample(tree, lay, st, 5, torch.zeros(4, dtype=torch.int32, device=DEV), torch.zeros(4, device=DEV))
    dq, newp = torch.zeros(B, A, device=DEV), torch.zeros(B, device=DEV)
    delta, w = torch.zeros(B, device=DEV), torch.zeros(B, device=DEV)
    H.apex_td(q_on.to(DEV), q_tg.to(DEV), idx.to(DEV), act.to(DEV), rew.to(DEV), done.to(DEV), p.to(DEV), st, 0.97, 0.6, 1e-5,
              dq, newp, delta, w)
    beta = oracle.beta_schedule(31, 0.4, 100)
    w_ref = oracle.is_weights(p, float(allp.double().sum()), cap, beta)
    il = idx.long()
    td = oracle.apex_td(q_on[:B], q_on[B:], q_tg, act[il], rew[il], done[il], w_ref, 0.97, 0.6)
    torch.testing.assert_close(w.cpu(), w_ref, rtol=2e-4, atol=1e-6)
    torch.testing.assert_close(delta.cpu(), td.delta, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(newp.cpu(), td.prio, rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(dq.cpu(), td.dq, rtol=2e-4, atol=1e-8)
    s = H.read_step_state(st)
    assert s["loss_cur"] == pytest.approx(float(td.loss), rel=2e-4) and s["beta"] == pytest.approx(beta, rel=1e-6)


@pytest.mark.parametrize("B,T,Hd", [(64, 120, 512), (7, 12, 512), (32, 24, 64)])
def test_seq_sample_one_launch_equals_the_five_it_replaces(hip, B, T, Hd):
    """k_seq_sample (draw + window offsets + stored state into both networks' initial rows + two bias sums) against
    sumtree_sample, seq_win_offsets, gather_state2 and add2 on a copy of the step state: same draws, same everything."""
    H = hip
    cap = 1000
    g = torch.Generator().manual_seed(B + T)
    lay = H.TreeLayout(cap)
    tree = torch.zeros(lay.total, device=DEV)
    H.sumtree_ingest(tree, lay, 0, (torch.rand(cap, generator=g) ** 2 + 1e-3).to(DEV), None, torch.zeros(cap, dtype=torch.int32, device=DEV))
    win = torch.randint(0, 1 << 40, (cap, T), generator=g, dtype=torch.int64).to(DEV)
    h0, c0 = torch.randn(cap, Hd, generator=g).to(DEV), torch.randn(cap, Hd, generator=g).to(DEV)
    gen = torch.randint(0, 9, (cap,), generator=g, dtype=torch.int32).to(DEV)
    n_items = torch.tensor([cap], dtype=torch.int64, device=DEV)
    a = [torch.randn(4 * Hd, generator=g).to(DEV) for _ in range(4)]
    st1 = H.new_step_state(DEV, 0.4, 100)
    for _ in range(3):
        H.step_end(st1)
    st2 = st1.clone()

    def outs():
        return dict(idx=torch.zeros(B, dtype=torch.int32, device=DEV), prio=torch.zeros(B, device=DEV),
                    gen_s=torch.zeros(B, dtype=torch.int32, device=DEV), off=torch.zeros(T * B, dtype=torch.int64, device=DEV),
                    hs_a=torch.zeros(B, Hd, dtype=torch.bfloat16, device=DEV), cs_a=torch.zeros(B, Hd, device=DEV),
                    hs_b=torch.zeros(B, Hd, dtype=torch.bfloat16, device=DEV), cs_b=torch.zeros(B, Hd, device=DEV),
                    s0=torch.zeros(4 * Hd, device=DEV), s1=torch.zeros(4 * Hd, device=DEV))
    o1, o2 = outs(), outs()
    H.sumtree_sample(tree, lay, st1, 17, o1["idx"], o1["prio"], gen, o1["gen_s"], begin_n_items=n_items)
    H.seq_win_offsets(o1["idx"], win, T, o1["off"])
    H.gather_state2(h0, c0, o1["idx"], o1["hs_a"], o1["cs_a"], o1["hs_b"], o1["cs_b"])
    H.add2(a[0], a[1], o1["s0"]); H.add2(a[2], a[3], o1["s1"])
    H.seq_sample(tree, lay, st2, 17, o2["idx"], o2["prio"], gen, o2["gen_s"], n_items, win, o2["off"], T, h0, c0,
                 o2["hs_a"], o2["cs_a"], o2["hs_b"], o2["cs_b"], sums=((a[0], a[1], o2["s0"]), (a[2], a[3], o2["s1"])))
    torch.cuda.synchronize()
    assert len(set(o1["idx"].tolist())) > 1
    for k in o1:
        assert torch.equal(o1[k], o2[k]), k
    assert torch.equal(st1, st2)                                     # beta, n_items, total priority: the same step_begin


def test_is_weights_with_the_ranks_maxima_from_a_mailbox(hip):
    """R2D2 data-parallel: k_is_wmax_put writes this rank's batch maximum into its mailbox slot (zeros elsewhere); after the
    SUM all-reduce (here: a peer's slot filled by hand) k_is_weights normalises by the maximum over all ranks - against
    oracle.is_weights(max_weight=...)."""
    H = hip
    B, cap = 64, 3000
    g = torch.Generator().manual_seed(9)
    lay = H.TreeLayout(cap)
    tree = torch.zeros(lay.total, device=DEV)
    allp = torch.rand(cap, generator=g) ** 2 + 0.01
    H.sumtree_ingest(tree, lay, 0, allp.to(DEV), None, torch.zeros(cap, dtype=torch.int32, device=DEV))
    st = H.new_step_state(DEV, 0.4, 100)
    n_items = torch.tensor([cap], dtype=torch.int64, device=DEV)
    idx, p = torch.zeros(B, dtype=torch.int32, device=DEV), torch.zeros(B, device=DEV)
    H.sumtree_sample(tree, lay, st, 3, idx, p, begin_n_items=n_items)
    beta, total = oracle.beta_schedule(0, 0.4, 100), float(allp.double().sum())
    raw = (cap * p.cpu().double() / total) ** (-beta)
    mailbox = torch.full((32,), 7.0, device=DEV)
    H.is_wmax_put(p, st, mailbox, 1, 4)
    mb = mailbox.cpu()
    assert mb[1].item() == pytest.approx(float(raw.max()), rel=1e-4) and float(mb.abs().sum()) == pytest.approx(mb[1].item())
    w = torch.zeros(B, device=DEV)
    for peer in (0.5 * float(raw.max()), 3.0 * float(raw.max())):    # the peer's batch maximum below / above ours
        mailbox[2] = peer
        H.is_weights(p, st, w, mailbox=mailbox, world=4)
        gmax = torch.tensor(max(peer, float(raw.max())))
        torch.testing.assert_close(w.cpu(), oracle.is_weights(p.cpu(), total, cap, beta, max_weight=gmax), rtol=2e-4, atol=1e-7)
        assert H.read_step_state(st)["max_w"] == pytest.approx(float(gmax), rel=1e-4)
    H.is_weights(p, st, w)                                            # no mailbox: the batch's own maximum
    torch.testing.assert_close(w.cpu(), oracle.is_weights(p.cpu(), total, cap, beta), rtol=2e-4, atol=1e-7)


@pytest.mark.parametrize("B,A", [(512, 4), (50, 18), (33, 6)])
def test_apex_td_dh_fused_matches_td_plus_head_bwd(hip, B, A):
    """TD + head input gradient in one launch, max IS weight from the sampling kernel, against the single-block TD
    kernel followed by head_bwd(parts=1)."""
    H = hip
    cap = 5000
    lay = H.TreeLayout(cap)
    tree = torch.zeros(lay.total, device=DEV)
    g = torch.Generator().manual_seed(B * 7 + A)
    H.sumtree_ingest(tree, lay, 0, (torch.rand(cap, generator=g) ** 3 + 1e-3).to(DEV), None, torch.zeros(cap, dtype=torch.int32, device=DEV))
    st = H.new_step_state(DEV, 0.4, 100)
    n_items = torch.tensor([cap], dtype=torch.int64, device=DEV)
    idx, p = torch.zeros(B, dtype=torch.int32, device=DEV), torch.zeros(B, device=DEV)
    act = torch.randint(0, A, (cap,), generator=g).int().to(DEV)
    rew, done = torch.randn(cap, generator=g).to(DEV), (torch.rand(cap, generator=g) < 0.1).float().to(DEV)
    q_on, q_tg = (torch.randn(2 * B, A, generator=g) * 2).to(DEV), (torch.randn(B, A, generator=g) * 2).to(DEV)
    q_on[B:B + 5] = 1.25                                            # ties in the argmax: the first action wins
    hq = bf(torch.relu(torch.randn(B, 1024, generator=g))).to(DEV)
    w_adv, w_val = bf(torch.randn(A, 512, generator=g) * 0.1).to(DEV), bf(torch.randn(1, 512, generator=g) * 0.1).to(DEV)
    for step in range(2):                                           # both parities of the max-weight slot
        H.sumtree_sample(tree, lay, st, 11, idx, p, begin_n_items=n_items, is_n_scale=1.0)
        dq1, pr1 = torch.full((B, A), 9.0, device=DEV), torch.zeros(B, device=DEV)
        dh1 = torch.full((B, 1024), 9.0, dtype=torch.bfloat16, device=DEV)
        H.apex_td_dh(q_on, q_tg, idx, act, rew, done, p, st, 0.97, 0.6, 1e-5, dq1, pr1, hq, w_adv, w_val, dh1)
        s1 = H.read_step_state(st)
        dq0, pr0 = torch.zeros(B, A, device=DEV), torch.zeros(B, device=DEV)
        H.apex_td(q_on, q_tg, idx, act, rew, done, p, st, 0.97, 0.6, 1e-5, dq0, pr0)
        s0 = H.read_step_state(st)
        dh0 = torch.zeros(B, 1024, dtype=torch.bfloat16, device=DEV)
        z = [torch.zeros(A, 512, device=DEV), torch.zeros(A, device=DEV), torch.zeros(1, 512, device=DEV), torch.zeros(1, device=DEV)]
        H.head_bwd(dq0, hq, w_adv, w_val, dh0, *z, parts=1)
        assert s1["max_w"] == pytest.approx(s0["max_w"], rel=1e-5) and s1["loss_cur"] == pytest.approx(s0["loss_cur"], rel=1e-4)
        torch.testing.assert_close(dq1, dq0, rtol=1e-5, atol=1e-9)
        torch.testing.assert_close(pr1, pr0, rtol=1e-6, atol=0)
        torch.testing.assert_close(dh1.float(), dh0.float(), rtol=2e-2, atol=1e-7)
        H.step_end(st)


@pytest.mark.parametrize("done_offset", [0, 5])
def test_r2d2_td(hip, done_offset):
    H = hip
    T, B, A, n, burn, cap = 120, 64, 6, 5, 40, 200
    g = torch.Generator().manual_seed(7 + done_offset)
    qo, qt = torch.randn(T, B, A, generator=g) * 3, torch.randn(T, B, A, generator=g) * 3
    idx = torch.randint(0, cap, (B,), generator=g).int()
    act = torch.randint(0, A, (cap, T), generator=g).int()
    rew = torch.randn(cap, T, generator=g)
    done = torch.zeros(cap, T)
    lengths = torch.full((cap,), T, dtype=torch.int32)
    short = torch.randperm(cap, generator=g)[:cap // 3]
    lengths[short] = torch.randint(burn + 10, T, (len(short),), generator=g).int()
    for s_ in short.tolist():
        L = int(lengths[s_])
        done[s_, L - n + 1:L] = 1.0
        done[s_, L - 1] = 1.0
    w = torch.rand(B, generator=g)
    st = H.new_step_state(DEV, 0.6, 100)
    dq, newp, sl = torch.full((T, B, A), 9.0, device=DEV), torch.zeros(B, device=DEV), torch.zeros(B, device=DEV)
    st.zero_()
    H.r2d2_td(qo.to(DEV), qt.to(DEV), idx.to(DEV), act.to(DEV), rew.to(DEV), done.to(DEV), lengths.to(DEV), w.to(DEV), st,
              burn, n, done_offset, 0.985, 0.9, 1e-3, dq, newp, sl)
    il = idx.long()
    a_tb, r_tb, d_tb = act[il].t().long(), rew[il].t(), done[il].t()
    mask = oracle.r2d2_learn_mask(lengths[il].long(), d_tb, T, burn, n, done_offset)
    td = oracle.r2d2_td(qo, qt, a_tb, r_tb, d_tb, mask, w, 0.985, n, 0.9, 1e-3, done_offset)
    torch.testing.assert_close(sl.cpu(), td.seq_loss, rtol=2e-4, atol=1e-5)
    torch.testing.assert_close(newp.cpu(), td.prio, rtol=2e-4, atol=1e-5)
    torch.testing.assert_close(dq.cpu(), td.dq, rtol=2e-3, atol=1e-5)
    assert H.read_step_state(st)["loss_cur"] == pytest.approx(float(td.loss), rel=5e-4)


# ---- optimizer -----------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind", ["rmsprop", "adam"])
def test_fused_optimizer(hip, kind):
    H = hip
    n = 3_292_864
    g_ = torch.Generator().manual_seed(0)
    p0 = torch.randn(n, generator=g_)
    p, m1, m2 = p0.clone().to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    pb = torch.zeros(n, dtype=torch.bfloat16, device=DEV)
    partial = torch.zeros(1024, device=DEV)
    q, s1, s2 = p0.clone(), torch.zeros(n), torch.zeros(n)
    st = H.new_step_state(DEV, 0.4, 100)
    for step in range(1, 4):
        g = torch.randn(n, generator=g_) * (0.05 if step == 2 else 0.001)    # step 2 clips, others do not
        gd = g.to(DEV)
        npart = H.grad_sumsq(gd, partial)
        if kind == "rmsprop":
            H.rmsprop_step(p, gd, m1, pb, partial, npart, 0.5, 40.0, 6.25e-5, 0.95, 1.5e-7, st)
        else:
            H.adam_step(p, gd, m1, m2, pb, partial, npart, 0.5, 40.0, 1e-4, 0.9, 0.999, 1e-3, st)
        H.step_end(st)
        gs = g * 0.5                                                          # grad_scale = 1/world
        coef = oracle.clip_coef(gs, 40.0)
        if step == 2:
            assert coef < 1.0
        if kind == "rmsprop":
            oracle.rmsprop_step(q, gs, s1, 6.25e-5, 0.95, 1.5e-7, coef)
        else:
            oracle.adam_step(q, gs, s1, s2, step, 1e-4, 0.9, 0.999, 1e-3, coef)
        assert H.read_step_state(st)["grad_norm"] == pytest.approx(float(gs.norm()), rel=1e-4)
        torch.testing.assert_close(p.cpu(), q, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(pb.float().cpu(), q.to(torch.bfloat16).float(), rtol=1e-2, atol=1e-6)


# ---- head ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,A,bias", [(1, 4, True), (50, 6, True), (513, 18, True), (256, 4, False)])
def test_head_fwd_bwd(hip, M, A, bias):
    H = hip
    g = torch.Generator().manual_seed(M + A)
    h = bf(torch.relu(torch.randn(M, 1024, generator=g)))
    w_adv, w_val = bf(torch.randn(A, 512, generator=g) * 0.05), bf(torch.randn(1, 512, generator=g) * 0.05)
    b_adv = torch.randn(A, generator=g) if bias else None
    b_val = torch.randn(1, generator=g)
    q = torch.zeros(M, A, device=DEV)
    act = torch.zeros(M, dtype=torch.int32, device=DEV)
    H.head_fwd(h.to(DEV), w_adv.to(DEV), b_adv.to(DEV) if bias else None, w_val.to(DEV), b_val.to(DEV), q, act)
    hf = h.float().requires_grad_(True)
    wa, wv = w_adv.float().requires_grad_(True), w_val.float().requires_grad_(True)
    ba = b_adv.clone().requires_grad_(True) if bias else None
    bv = b_val.clone().requires_grad_(True)
    adv = hf[:, :512] @ wa.t() + (ba if bias else 0)
    val = hf[:, 512:] @ wv.t() + bv
    q_ref = oracle.dueling_combine(val, adv)
    torch.testing.assert_close(q.cpu(), q_ref.detach(), rtol=1e-3, atol=1e-3)
    assert (act.cpu().long() == q.cpu().argmax(1)).all()
    dq = torch.randn(M, A, generator=g)
    q_ref.backward(dq)
    dh = torch.zeros(M, 1024, dtype=torch.bfloat16, device=DEV)
    gwa, gwv = torch.zeros(A, 512, device=DEV), torch.zeros(1, 512, device=DEV)
    gba, gbv = (torch.zeros(A, device=DEV) if bias else None), torch.zeros(1, device=DEV)
    H.head_bwd(dq.to(DEV), h.to(DEV), w_adv.to(DEV), w_val.to(DEV), dh, gwa, gba, gwv, gbv)
    dh_ref = hf.grad * (h.float() > 0)
    torch.testing.assert_close(dh.float().cpu(), dh_ref, rtol=2e-2, atol=2e-3)
    torch.testing.assert_close(gwa.cpu(), wa.grad, rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(gwv.cpu(), wv.grad, rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(gbv.cpu(), bv.grad, rtol=1e-3, atol=1e-3)
    if bias:
        torch.testing.assert_close(gba.cpu(), ba.grad, rtol=1e-3, atol=1e-3)


def test_head_eps_greedy(hip):
    H = hip
    M, A = 4096, 6
    h = bf(torch.relu(torch.randn(M, 1024)))
    w_adv, w_val = bf(torch.randn(A, 512) * 0.05), bf(torch.randn(1, 512) * 0.05)
    q = torch.zeros(M, A, device=DEV)
    act = torch.zeros(M, dtype=torch.int32, device=DEV)
    eps = torch.tensor([0.0, 1.0, 0.5, 0.25], device=DEV)
    ctr = torch.tensor([3], dtype=torch.int64, device=DEV)
    H.head_fwd(h.to(DEV), w_adv.to(DEV), torch.zeros(A, device=DEV), w_val.to(DEV), torch.zeros(1, device=DEV), q, act, eps, 4, 99, ctr)
    greedy = q.argmax(1).cpu()
    a = act.cpu().long()
    rows = torch.arange(M)
    assert (a[rows % 4 == 0] == greedy[rows % 4 == 0]).all()                     # eps 0: always greedy
    r1 = a[rows % 4 == 1]
    assert np.bincount(r1.numpy(), minlength=A).min() > 0.6 * len(r1) / A        # eps 1: uniform over actions
    frac = (a[rows % 4 == 2] != greedy[rows % 4 == 2]).float().mean().item()
    assert abs(frac - 0.5 * (A - 1) / A) < 0.06
    act2 = torch.zeros_like(act)
    H.head_fwd(h.to(DEV), w_adv.to(DEV), torch.zeros(A, device=DEV), w_val.to(DEV), torch.zeros(1, device=DEV), q, act2, eps, 4, 99, ctr)
    assert torch.equal(act, act2)                                                # counter-based RNG: reproducible


# ---- GEMM ----------------------------------------------------------------------------------------------
def _rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return bf(torch.randn(*shape, generator=g) * scale)


@pytest.mark.parametrize("M,N,K,big", [(64, 64, 64, False), (100, 128, 192, False), (512, 1024, 3136, False),
                                       (1000, 3136, 1024, False), (2500, 2048, 512, True), (130, 256, 128, True)])
def test_gemm_nt_bias_relu(hip, M, N, K, big):
    H = hip
    a, w = _rnd(M, K, seed=1), _rnd(N, K, seed=2, scale=0.05)
    bias = torch.randn(N)
    c = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    H.gemm(H.NT, a.to(DEV), w.to(DEV), c, M, N, K, bias=bias.to(DEV), flags=H.EPI_RELU, big=big)
    ref = torch.relu(a.float() @ w.float().t() + bias)
    torch.testing.assert_close(c.float().cpu(), ref, rtol=2e-2, atol=2e-2)
    c32 = torch.zeros(M, N, device=DEV)
    H.gemm(H.NT, a.to(DEV), w.to(DEV), c32, M, N, K, bias=bias.to(DEV), flags=H.EPI_OUT_F32, big=big)
    torch.testing.assert_close(c32.cpu(), a.float() @ w.float().t() + bias, rtol=1e-3, atol=2e-3)


@pytest.mark.parametrize("M,N,K,big", [(64, 64, 64, False), (512, 3136, 1024, False), (777, 576, 64, False), (300, 256, 128, True)])
def test_gemm_nn_mask(hip, M, N, K, big):
    H = hip
    a, b = _rnd(M, K, seed=3), _rnd(K, N, seed=4, scale=0.05)          # asymmetric operands (guide §3: A=I checks)
    mask = _rnd(M, N, seed=5)
    c = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    H.gemm(H.NN, a.to(DEV), b.to(DEV), c, M, N, K, mask=mask.to(DEV), flags=H.EPI_MASK, big=big)
    ref = (a.float() @ b.float()) * (mask.float() > 0)
    torch.testing.assert_close(c.float().cpu(), ref, rtol=2e-2, atol=2e-2)
    eye = bf(torch.eye(64))
    b2 = _rnd(64, 64, seed=6)
    c2 = torch.zeros(64, 64, dtype=torch.bfloat16, device=DEV)
    H.gemm(H.NN, eye.to(DEV), b2.to(DEV), c2, 64, 64, 64, big=False)
    assert torch.equal(c2.cpu(), b2)


@pytest.mark.parametrize("M,N,K,big,splits", [(64, 64, 64, False, 1), (1024, 3136, 512, False, 1), (64, 576, 25088, False, 28),
                                              (128, 256, 1000, True, 1), (64, 128, 4807, False, 5)])
def test_gemm_tn_splitk(hip, M, N, K, big, splits):
    H = hip
    a, b = _rnd(K, M, seed=7, scale=0.1), _rnd(K, N, seed=8, scale=0.1)
    c = torch.zeros(M, N, device=DEV)
    flags = H.EPI_OUT_F32 | (H.EPI_ATOMIC if splits > 1 else 0)
    H.gemm(H.TN, a.to(DEV), b.to(DEV), c, M, N, K, flags=flags, splits=splits, big=big)
    ref = a.float().t() @ b.float()
    torch.testing.assert_close(c.cpu(), ref, rtol=2e-3, atol=2e-3 * (K ** 0.5) * 0.01 + 1e-3)


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 10, 11, 12, 13, 14])
@pytest.mark.parametrize("mode", ["NT", "NN", "TN"])
def test_gemm_tile_variants(hip, variant, mode):
    """Every tile variant (BMxBNxBK / ring depth) on a ragged-K, ragged-M problem incl. split-K atomics."""
    H = hip
    M, N, K = (256, 384, 1000) if mode == "TN" else (300, 384, 1000 if mode == "NN" else 1096)
    if mode == "NT":
        a, b = _rnd(M, K, seed=1), _rnd(N, K, seed=2, scale=0.05)
        ref = a.float() @ b.float().t()
    elif mode == "NN":
        a, b = _rnd(M, K, seed=3), _rnd(K, N, seed=4, scale=0.05)
        ref = a.float() @ b.float()
    else:
        a, b = _rnd(K, M, seed=5, scale=0.2), _rnd(K, N, seed=6, scale=0.2)
        ref = a.float().t() @ b.float()
    c = torch.zeros(M, N, device=DEV)
    H.gemm(getattr(H, mode), a.to(DEV), b.to(DEV), c, M, N, K, flags=H.EPI_OUT_F32, big=variant)
    torch.testing.assert_close(c.cpu(), ref, rtol=3e-3, atol=3e-2)
    c.zero_()
    H.gemm(getattr(H, mode), a.to(DEV), b.to(DEV), c, M, N, K, flags=H.EPI_OUT_F32 | H.EPI_ATOMIC, splits=3, big=variant)
    torch.testing.assert_close(c.cpu(), ref, rtol=3e-3, atol=3e-2)


# ---- conv ----------------------------------------------------------------------------------------------
LAYERS = {1: (4, 32, 8, 4, 84), 2: (32, 64, 4, 2, 20), 3: (64, 64, 3, 1, 9)}


def _conv_setup(layer, frames, seed):
    cin, cout, ks, s, hw = LAYERS[layer]
    g = torch.Generator().manual_seed(seed)
    if layer == 1:
        x = torch.randint(0, 256, (frames, hw, hw, cin), dtype=torch.uint8, generator=g)
        xf = x.float() / 255
    else:
        x = bf(torch.relu(torch.randn(frames, hw, hw, cin, generator=g)))
        xf = x.float()
    w = bf(torch.randn(cout, ks, ks, cin, generator=g) * (1.0 / (ks * ks * cin) ** 0.5))
    bias = torch.randn(cout, generator=g) * 0.1
    return x, xf, w, bias, (cin, cout, ks, s, hw)


def _plane_store(stacks):
    """HWC stacks u8 [n,84,84,4] -> a plane store [4n, 7056] (every stack as 4 consecutive planes) + the byte offset of
    every stack's window; the replay keeps shared planes once, for the kernels only the offsets matter."""
    n = stacks.shape[0]
    return stacks.permute(0, 3, 1, 2).reshape(n * 4, 7056).contiguous(), torch.arange(n) * (4 * 7056)


def _w1_planes(w):
    """conv1 weights [32,kh,kw,c] (dense HWC input) -> [32,c,kh,kw] (plane store input; PyTorch's own order)."""
    return w.permute(0, 3, 1, 2).contiguous()


@pytest.mark.parametrize("layer", [1, 2, 3])
@pytest.mark.parametrize("frames", [1, 7, 96, 700])
@pytest.mark.parametrize("tiled", [None, 0, 1, 2, 3])     # None: the default kernel (conv1 direct, conv2/3 LDS-halo); 0 = tiled
def test_conv_fwd(hip, layer, frames, tiled):               # register ring, 1..3 = tiled with the LDS-DMA mainloops
    H = hip
    x, xf, w, bias, (cin, cout, ks, s, hw) = _conv_setup(layer, frames, layer * 100 + frames)
    oh = (hw - ks) // s + 1
    out = torch.zeros(frames, oh, oh, cout, dtype=torch.bfloat16, device=DEV)
    H.conv_fwd(layer, x.to(DEV), w.to(DEV), bias.to(DEV), out, frames, tiled=tiled)
    ref = F.relu(F.conv2d(xf.permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), bias, stride=s)).permute(0, 2, 3, 1)
    torch.testing.assert_close(out.float().cpu(), ref, rtol=2e-2, atol=2e-2)


def test_conv1_frame_offsets(hip):
    """conv1 over the replay's plane store (frame = 4 consecutive planes from a byte offset, weights [32,c,kh,kw]) ==
    conv1 over the same frames as dense HWC stacks (weights [32,kh,kw,c]), for the direct and the tiled kernel; windows
    that OVERLAP (a stack shifted by one plane, as consecutive observations are stored) included."""
    H = hip
    stacks = torch.randint(0, 256, (20, 84, 84, 4), dtype=torch.uint8)
    store, off = _plane_store(stacks)
    pick = torch.tensor([7, 3, 19, 7, 0])
    _, _, w, bias, _ = _conv_setup(1, 1, 5)
    out = torch.zeros(5, 20, 20, 32, dtype=torch.bfloat16, device=DEV)
    for tiled in (0, None):
        out.zero_()
        H.conv_fwd(1, store.to(DEV), _w1_planes(w).to(DEV), bias.to(DEV), out, 5, frame_off=off[pick].to(DEV), tiled=tiled)
        ref = torch.zeros_like(out)
        H.conv_fwd(1, stacks[pick].contiguous().to(DEV), w.to(DEV), bias.to(DEV), ref, 5, tiled=tiled)
        # same products, another summation order (k = (c, kh, kw) against (kh, kw, c)): equal up to one bf16 rounding
        torch.testing.assert_close(out.float(), ref.float(), rtol=8e-3, atol=4e-3)
    # windows shifted by 1..3 planes: planes (4 i + k .. 4 i + k + 3) = the stack [c_k.., next stack's c_0..]
    shift = torch.tensor([1, 2, 3, 5, 9])
    dense = store.view(-1, 84, 84)[(shift[:, None] + torch.arange(4)[None, :])].permute(0, 2, 3, 1).contiguous()
    H.conv_fwd(1, store.to(DEV), _w1_planes(w).to(DEV), bias.to(DEV), out, 5, frame_off=(shift * 7056).to(DEV))
    ref = torch.zeros_like(out)
    H.conv_fwd(1, dense.to(DEV), w.to(DEV), bias.to(DEV), ref, 5)
    torch.testing.assert_close(out.float(), ref.float(), rtol=8e-3, atol=4e-3)


@pytest.mark.parametrize("layer", [1, 2, 3])
@pytest.mark.parametrize("f1,f2", [(5, 3), (64, 32), (130, 65)])
def test_conv_fwd_dual(hip, layer, f1, f2):
    """Two networks per launch (online + target) against the fp32 convolution of each."""
    H = hip
    x1, xf1, w1, b1, (cin, cout, ks, s, hw) = _conv_setup(layer, f1, layer * 31 + f1)
    x2, xf2, w2, b2, _ = _conv_setup(layer, f2, layer * 37 + f2)
    oh = (hw - ks) // s + 1
    o1 = torch.zeros(f1, oh, oh, cout, dtype=torch.bfloat16, device=DEV)
    o2 = torch.zeros(f2, oh, oh, cout, dtype=torch.bfloat16, device=DEV)
    if layer == 1:
        store, off = _plane_store(torch.cat([x1, x2]))
        H.conv_fwd_dual(1, store.to(DEV), f1, _w1_planes(w1).to(DEV), b1.to(DEV), o1, None, f2, _w1_planes(w2).to(DEV), b2.to(DEV),
                        o2, frame_off1=off[:f1].to(DEV), frame_off2=off[f1:].to(DEV))
    else:
        H.conv_fwd_dual(layer, x1.to(DEV), f1, w1.to(DEV), b1.to(DEV), o1, x2.to(DEV), f2, w2.to(DEV), b2.to(DEV), o2)
    for o, xf, w, b in ((o1, xf1, w1, b1), (o2, xf2, w2, b2)):
        ref = F.relu(F.conv2d(xf.permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), b, stride=s)).permute(0, 2, 3, 1)
        torch.testing.assert_close(o.float().cpu(), ref, rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("layer,variant", [(2, 0), (2, 1), (3, 0), (3, 1)])
@pytest.mark.parametrize("f1,f2", [(1, 0), (7, 0), (5, 3), (700, 0), (1024, 512), (1537, 700)])
def test_conv_halo_fwd(hip, layer, variant, f1, f2):
    """LDS-halo conv2 / conv3 (conv_halo.hip): one and two networks per launch, frame counts that are not multiples of
    the frame group, more groups than persistent workgroups; a poisoned guard frame behind each output stays untouched."""
    H = hip
    nets = []
    for f, seed in ((f1, 11), (f2, 13)):
        if f == 0:
            nets.append(None)
            continue
        x, xf, w, b, (cin, cout, ks, s, hw) = _conv_setup(layer, f, layer * 41 + f + seed)
        oh = (hw - ks) // s + 1
        o = torch.full((f + 1, oh, oh, cout), 7.0, dtype=torch.bfloat16, device=DEV)
        nets.append((x.to(DEV), xf, w, b, o, s))
    (x1, _, w1, b1, o1, _), n2 = nets[0], nets[1]
    if n2 is None:
        H.conv_halo_fwd(layer, x1, f1, w1.to(DEV), b1.to(DEV), o1, variant=variant)
    else:
        H.conv_halo_fwd(layer, x1, f1, w1.to(DEV), b1.to(DEV), o1, n2[0], f2, n2[2].to(DEV), n2[3].to(DEV), n2[4], variant=variant)
    for net, f in zip(nets, (f1, f2)):
        if net is None:
            continue
        x, xf, w, b, o, s = net
        ref = F.relu(F.conv2d(xf.permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), b, stride=s)).permute(0, 2, 3, 1)
        torch.testing.assert_close(o[:f].float().cpu(), ref, rtol=2e-2, atol=2e-2)
        assert bool((o[f] == 7.0).all())
    # beside a persistent kernel of another stream the grid shrinks to what is resident (H.halo_busy): other shares of the
    # frames per workgroup, bit-identical outputs
    outs = [net[4].clone() for net in nets if net is not None]
    for busy in (32, 64):
        for net in nets:
            if net is not None:
                net[4].fill_(7.0)
        with H.halo_busy(busy):
            if n2 is None:
                H.conv_halo_fwd(layer, x1, f1, w1.to(DEV), b1.to(DEV), o1, variant=variant)
            else:
                H.conv_halo_fwd(layer, x1, f1, w1.to(DEV), b1.to(DEV), o1, n2[0], f2, n2[2].to(DEV), n2[3].to(DEV), n2[4], variant=variant)
        assert all(torch.equal(net[4], o) for net, o in zip([n for n in nets if n is not None], outs))


@pytest.mark.parametrize("M1,M2", [(7, 3), (128, 64), (1024, 512)])
def test_gemm_nt_and_head_dual(hip, M1, M2):
    H = hip
    g = torch.Generator().manual_seed(M1)
    A = 6
    outs = []
    nets = []
    for M in (M1, M2):
        a = bf(torch.relu(torch.randn(M, 3136, generator=g)))
        w = bf(torch.randn(1024, 3136, generator=g) * 0.02)
        b = torch.randn(1024, generator=g)
        wa, wv = bf(torch.randn(A, 512, generator=g) * 0.05), bf(torch.randn(1, 512, generator=g) * 0.05)
        ba, bv = torch.randn(A, generator=g), torch.randn(1, generator=g)
        nets.append((a, w, b, wa, ba, wv, bv))
        outs.append((torch.zeros(M, 1024, dtype=torch.bfloat16, device=DEV), torch.zeros(M, A, device=DEV)))
    d = [[t.to(DEV) for t in n] for n in nets]
    H.gemm_nt_dual(d[0][0], d[0][1], d[0][2], outs[0][0], d[1][0], d[1][1], d[1][2], outs[1][0], 1024, 3136, flags=H.EPI_RELU)
    H.head_fwd_dual(outs[0][0], d[0][3], d[0][4], d[0][5], d[0][6], outs[0][1],
                    outs[1][0], d[1][3], d[1][4], d[1][5], d[1][6], outs[1][1])
    for (a, w, b, wa, ba, wv, bv), (h, q) in zip(nets, outs):
        h_ref = torch.relu(a.float() @ w.float().t() + b)
        torch.testing.assert_close(h.float().cpu(), h_ref, rtol=2e-2, atol=2e-2)
        hf = h.float().cpu()
        q_ref = oracle.dueling_combine(hf[:, 512:] @ wv.float().t() + bv, hf[:, :512] @ wa.float().t() + ba)
        torch.testing.assert_close(q.cpu(), q_ref, rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("layer", [1, 2, 3])
@pytest.mark.parametrize("frames,splits", [(3, 1), (64, 16)])
def test_conv_wgrad_and_bias(hip, layer, frames, splits):
    H = hip
    x, xf, w, bias, (cin, cout, ks, s, hw) = _conv_setup(layer, frames, layer * 7 + frames)
    oh = (hw - ks) // s + 1
    g = torch.Generator().manual_seed(11)
    dy = bf(torch.randn(frames, oh, oh, cout, generator=g) * 0.1)
    wf = w.float().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    bf_ = bias.clone().requires_grad_(True)
    y = F.conv2d(xf.permute(0, 3, 1, 2), wf, bf_, stride=s)
    y.backward(dy.float().permute(0, 3, 1, 2))
    dw = torch.zeros(cout, ks, ks, cin, device=DEV)
    H.conv_wgrad(layer, x.to(DEV), dy.to(DEV), dw, frames, splits=splits)
    ref = wf.grad.permute(0, 2, 3, 1)
    torch.testing.assert_close(dw.cpu(), ref, rtol=5e-3, atol=5e-3 * float(ref.abs().max()))
    db = torch.zeros(cout, device=DEV)
    H.colsum(dy.to(DEV), db, frames * oh * oh, cout)
    torch.testing.assert_close(db.cpu(), bf_.grad, rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("M,N", [(5, 32), (204800, 32), (25088, 64), (512, 1024), (333, 2048), (1, 8)])
def test_colsum_shapes(hip, M, N):
    H = hip
    g = torch.Generator().manual_seed(M + N)
    dy = bf(torch.randn(M, N, generator=g))
    db = torch.full((N,), 0.5, device=DEV)              # accumulates on top of what is there
    H.colsum(dy.to(DEV), db, M, N)
    ref = dy.double().sum(0).float() + 0.5
    torch.testing.assert_close(db.cpu(), ref, rtol=1e-3, atol=1e-3 * max(1.0, M ** 0.5))


@pytest.mark.parametrize("layer", [2, 3])
@pytest.mark.parametrize("frames", [2, 33, 300])
def test_conv_dgrad(hip, layer, frames):
    H = hip
    x, xf, w, bias, (cin, cout, ks, s, hw) = _conv_setup(layer, frames, layer * 13 + frames)
    oh = (hw - ks) // s + 1
    g = torch.Generator().manual_seed(12)
    dy = bf(torch.randn(frames, oh, oh, cout, generator=g) * 0.1)
    xin = xf.permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    F.conv2d(xin, w.float().permute(0, 3, 1, 2), None, stride=s).backward(dy.float().permute(0, 3, 1, 2))
    ref = xin.grad.permute(0, 2, 3, 1) * (xf > 0)
    # implicit one-launch form (parity-class GEMM, fused mask) against fp32 autograd
    dx2 = torch.full((frames, hw, hw, cin), 7.0, dtype=torch.bfloat16, device=DEV)
    H.conv_dgrad(layer, dy.to(DEV), w.to(DEV), x.to(DEV), dx2, frames)
    torch.testing.assert_close(dx2.float().cpu(), ref, rtol=3e-2, atol=3e-2 * float(ref.abs().max()))
    dx3 = torch.full_like(dx2, 7.0)
    H.conv_dgrad(layer, dy.to(DEV), w.to(DEV), None, dx3, frames)
    ref_nomask = xin.grad.permute(0, 2, 3, 1)
    torch.testing.assert_close(dx3.float().cpu(), ref_nomask, rtol=3e-2, atol=3e-2 * float(ref.abs().max()))


@pytest.mark.parametrize("layer", [2, 3])
@pytest.mark.parametrize("frames", [1, 2, 33, 512, 1700])
def test_conv_halo_dgrad(hip, layer, frames):
    """Input gradient of conv2 / conv3 through the LDS-halo kernel (zero-padded dy, pre-transformed weights, ReLU mask in
    the epilogue) against fp32 autograd; the frame behind the last one stays untouched."""
    H = hip
    x, xf, w, bias, (cin, cout, ks, s, hw) = _conv_setup(layer, frames, layer * 17 + frames)
    oh = (hw - ks) // s + 1
    g = torch.Generator().manual_seed(12)
    dy = bf(torch.randn(frames, oh, oh, cout, generator=g) * 0.1)
    xin = xf.permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    F.conv2d(xin, w.float().permute(0, 3, 1, 2), None, stride=s).backward(dy.float().permute(0, 3, 1, 2))
    ref = xin.grad.permute(0, 2, 3, 1) * (xf > 0)
    _, _, w2, _, _ = _conv_setup(2, 1, 5)
    _, _, w3, _, _ = _conv_setup(3, 1, 6)
    ws = {2: w2, 3: w3}
    ws[layer] = w
    wt2 = torch.zeros(4, 32, 2, 2, 64, dtype=torch.bfloat16, device=DEV)
    wt3 = torch.zeros(64, 3, 3, 64, dtype=torch.bfloat16, device=DEV)
    H.conv_halo_wprep(ws[2].to(DEV), ws[3].to(DEV), wt2, wt3)
    dx = torch.full((frames + 1, hw, hw, cin), 7.0, dtype=torch.bfloat16, device=DEV)
    H.conv_halo_dgrad(layer, dy.to(DEV), wt2 if layer == 2 else wt3, x.to(DEV), dx, frames)
    torch.testing.assert_close(dx[:frames].float().cpu(), ref, rtol=3e-2, atol=3e-2 * float(ref.abs().max()))
    assert bool((dx[frames] == 7.0).all())


@pytest.mark.parametrize("layer", [1, 2, 3])
@pytest.mark.parametrize("frames,splits,variant", [(3, 1, None), (33, 8, 0), (512, 32, None), (40, 4, 1), (40, 4, 3), (40, 4, 5)])
def test_conv_bwd_fused(hip, layer, frames, splits, variant):
    """dX + dW + db of a conv layer as ONE launch (k_multi) against autograd of the fp32 convolution."""
    H = hip
    x, xf, w, bias, (cin, cout, ks, s, hw) = _conv_setup(layer, frames, layer * 17 + frames)
    oh = (hw - ks) // s + 1
    g = torch.Generator().manual_seed(13)
    dy = bf(torch.randn(frames, oh, oh, cout, generator=g) * 0.1)
    xin = xf.permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    wf = w.float().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    bf_ = bias.clone().requires_grad_(True)
    F.conv2d(xin, wf, bf_, stride=s).backward(dy.float().permute(0, 3, 1, 2))
    dw = torch.zeros(cout, ks, ks, cin, device=DEV)
    db = torch.zeros(cout, device=DEV)
    dx = torch.full((frames, hw, hw, cin), 7.0, dtype=torch.bfloat16, device=DEV) if layer != 1 else None
    H.conv_bwd(layer, x.to(DEV), dy.to(DEV), w.to(DEV) if layer != 1 else None, dw, db, dx, frames, splits=splits,
               variant=variant if layer != 1 else None)
    ref = wf.grad.permute(0, 2, 3, 1)
    torch.testing.assert_close(dw.cpu(), ref, rtol=5e-3, atol=5e-3 * float(ref.abs().max()))
    torch.testing.assert_close(db.cpu(), bf_.grad, rtol=1e-3, atol=1e-3 * max(1.0, (frames * oh * oh) ** 0.5))
    if layer != 1:
        rdx = xin.grad.permute(0, 2, 3, 1) * (xf > 0)
        torch.testing.assert_close(dx.float().cpu(), rdx, rtol=3e-2, atol=3e-2 * float(rdx.abs().max()))


@pytest.mark.parametrize("cap", [50, 4000, 2_000_000])
def test_conv_bwd_fused_tree_stages(hip, cap):
    """The PER priority write-back as extra bodies of two consecutive fused conv-backward launches must leave the tree
    exactly as the stand-alone two-launch update does (and the step counter advanced)."""
    H = hip
    lay = H.TreeLayout(cap)
    g = torch.Generator().manual_seed(cap + 3)
    prio = (torch.rand(cap, generator=g) + 0.01).to(DEV)
    trees = []
    B = 96
    base = torch.randint(0, cap, (B,), generator=g)
    base[:32] = (base[0] // 64 * 64 + torch.randint(0, 64, (32,), generator=g)).clamp(max=cap - 1)
    base[40:44] = base[40]
    ui, uv = base.int().to(DEV), (torch.rand(B, generator=g) + 0.5).to(DEV)
    x3, _, w3, _, _ = _conv_setup(3, 4, 1)
    x2, _, w2, _, _ = _conv_setup(2, 4, 2)
    dy3 = bf(torch.randn(4, 7, 7, 64, generator=g) * 0.1).to(DEV)
    dy2 = bf(torch.randn(4, 9, 9, 64, generator=g) * 0.1).to(DEV)
    for fused in (False, True):
        tree = torch.zeros(lay.total, device=DEV)
        scratch = torch.zeros(cap, dtype=torch.int32, device=DEV)
        H.sumtree_ingest(tree, lay, 0, prio, None, scratch)
        st = H.new_step_state(DEV, 0.4, 100)
        if fused:
            dw3, dw2, db = torch.zeros(64, 3, 3, 64, device=DEV), torch.zeros(64, 4, 4, 32, device=DEV), torch.zeros(64, device=DEV)
            dx3 = torch.zeros(4, 9, 9, 64, dtype=torch.bfloat16, device=DEV)
            dx2 = torch.zeros(4, 20, 20, 32, dtype=torch.bfloat16, device=DEV)
            H.conv_bwd(3, x3.to(DEV), dy3, w3.to(DEV), dw3, db, dx3, 4, splits=2, extra=H.tree_extra(1, tree, lay, ui, uv))
            H.conv_bwd(2, x2.to(DEV), dy2, w2.to(DEV), dw2, db, dx2, 4, splits=2, extra=H.tree_extra(2, tree, lay, ui, uv, end_state=st))
            assert float(dw3.abs().sum()) > 0 and float(dx2.float().abs().sum()) > 0
        else:
            H.sumtree_update(tree, lay, ui, uv, end_state=st)
        assert H.read_step_state(st)["step"] == 1
        trees.append(tree.cpu())
    assert torch.equal(trees[0], trees[1])


def test_conv1_bwd_fused_frame_offsets(hip):
    """conv1 weight + bias gradient over the plane store ([32,c,kh,kw]) == the dense HWC form ([32,kh,kw,c]) permuted."""
    H = hip
    stacks = torch.randint(0, 256, (20, 84, 84, 4), dtype=torch.uint8)
    store, off = _plane_store(stacks)
    pick = torch.tensor([7, 3, 19, 7, 0])
    g = torch.Generator().manual_seed(14)
    dy = bf(torch.randn(5, 20, 20, 32, generator=g) * 0.1).to(DEV)
    dw, db = torch.zeros(32, 4, 8, 8, device=DEV), torch.zeros(32, device=DEV)
    H.conv_bwd(1, store.to(DEV), dy, None, dw, db, None, 5, frame_off=off[pick].to(DEV), splits=4)
    dw2, db2 = torch.zeros(32, 8, 8, 4, device=DEV), torch.zeros_like(db)
    H.conv_wgrad(1, stacks[pick].contiguous().to(DEV), dy, dw2, 5, splits=1)
    H.colsum(dy, db2, 5 * 400, 32)
    torch.testing.assert_close(dw, _w1_planes(dw2), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(db, db2, rtol=1e-4, atol=1e-5)
    dw3 = torch.zeros_like(dw)
    H.conv_wgrad(1, store.to(DEV), dy, dw3, 5, frame_off=off[pick].to(DEV), splits=2)
    torch.testing.assert_close(dw3, _w1_planes(dw2), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("M,A,with_head,variant", [(512, 4, True, None), (64, 18, True, 0), (100, 6, False, 2), (70, 4, True, 5)])
def test_fc_bwd_fused(hip, M, A, with_head, variant):
    """fc1 backward (dX masked, dW, db) + dueling-head weight gradients in one launch vs fp32 matmuls / the stand-alone
    head kernel."""
    H = hip
    N, K = 1024, 3136
    g = torch.Generator().manual_seed(M + A)
    dy = bf(torch.randn(M, N, generator=g) * 0.1)
    x = bf(torch.relu(torch.randn(M, K, generator=g)))
    w = bf(torch.randn(N, K, generator=g) * 0.02)
    dw = torch.full((N, K), 3.0, device=DEV)                   # overwritten
    db = torch.zeros(N, device=DEV)
    dx = torch.full((M, K), 7.0, dtype=torch.bfloat16, device=DEV)
    head = None
    if with_head:
        dq = (torch.randn(M, A, generator=g) * 0.1).to(DEV)
        hq = bf(torch.relu(torch.randn(M, 1024, generator=g))).to(DEV)
        gw_adv, gb_adv = torch.zeros(A, 512, device=DEV), torch.zeros(A, device=DEV)
        gw_val, gb_val = torch.zeros(1, 512, device=DEV), torch.zeros(1, device=DEV)
        head = (dq, hq, gw_adv, gb_adv, gw_val, gb_val)
    H.fc_bwd(dy.to(DEV), x.to(DEV), w.to(DEV), dw, db, dx, M, N, K, head=head, variant=variant)
    torch.testing.assert_close(dw.cpu(), dy.float().t() @ x.float(), rtol=3e-3, atol=3e-2)
    torch.testing.assert_close(db.cpu(), dy.float().sum(0), rtol=1e-3, atol=1e-3 * M ** 0.5)
    rdx = (dy.float() @ w.float()) * (x.float() > 0)
    torch.testing.assert_close(dx.float().cpu(), rdx, rtol=2e-2, atol=2e-2 * float(rdx.abs().max()))
    if with_head:
        r = [torch.zeros_like(t) for t in (gw_adv, gb_adv, gw_val, gb_val)]
        wz = torch.zeros(A, 512, dtype=torch.bfloat16, device=DEV)
        dh = torch.zeros(M, 1024, dtype=torch.bfloat16, device=DEV)
        H.head_bwd(dq, hq, wz, wz[:1], dh, *r, parts=2)
        for got, want in zip((gw_adv, gb_adv, gw_val, gb_val), r):
            torch.testing.assert_close(got, want, rtol=1e-4, atol=1e-5)


# ---- LSTM recurrence -----------------------------------------------------------------------------------
def _lstm_case(T, M, Hd, seed):
    from distributed_rl_amd.models.packing import _gate_interleave
    g = torch.Generator().manual_seed(seed)
    w_hh = bf(torch.randn(4 * Hd, Hd, generator=g) * 0.04)          # PyTorch row order [i; f; g; o]
    xg = torch.randn(T, M, 4 * Hd, generator=g) * 0.5
    h0 = bf(torch.randn(M, Hd, generator=g) * 0.3)
    c0 = torch.randn(M, Hd, generator=g) * 0.3
    hs_ref, cs_ref = oracle.lstm_unroll(xg, w_hh.float(), h0.float(), c0)
    w_k = _gate_interleave(w_hh, Hd).to(DEV)
    xg_k = _gate_interleave(xg.permute(2, 0, 1).contiguous(), Hd).permute(1, 2, 0).contiguous().to(DEV)
    return w_k, xg_k, h0, c0, hs_ref, cs_ref


@pytest.mark.parametrize("M", [16, 48, 64])
def test_lstm_step_and_persistent_sequence(hip, M):
    """T single-step launches (EPI_LSTM GEMM) and the persistent one-launch recurrence against the fp32 oracle."""
    H = hip
    T, Hd = 12, 512
    w_k, xg_k, h0, c0, hs_ref, cs_ref = _lstm_case(T, M, Hd, 100 + M)
    outs = []
    for mode in ("step", "xchg", "xchgx2", "xchg-g2", "xchgx2-g4"):
        hs = torch.zeros(T + 1, M, Hd, dtype=torch.bfloat16, device=DEV)
        cs = torch.zeros(T + 1, M, Hd, device=DEV)
        gates = torch.zeros(T, M, 4 * Hd, dtype=torch.bfloat16, device=DEV)
        hs[0], cs[0] = h0.to(DEV), c0.to(DEV)
        if mode == "step":
            for t in range(T):
                H.lstm_step(hs[t], w_k, xg_k[t], cs[t], hs[t + 1], cs[t + 1], gates[t])
        else:                                                       # persistent recurrence, one launch for all T steps
            err = torch.zeros(1, dtype=torch.int32, device=DEV)
            nets = [(xg_k, w_k, hs, cs, gates, H.lstm_hx(DEV))]
            if "x2" in mode:                                        # second network in the same launch, no gates
                hs2, cs2 = hs.clone(), cs.clone()
                nets.append((xg_k, w_k, hs2, cs2, None, H.lstm_hx(DEV)))
            groups = int(mode[-1]) if "-g" in mode else 1           # independent row groups (when they divide M / 16)
            if (M // 16) % groups:
                groups = 1
            for rep in range(3):                                    # again: the scratch is reused across launches,
                H.lstm_seq_fwd_x(nets, T, M, Hd, err, groups=groups if rep != 1 else 1)      # ... with changing splits
            assert int(err.item()) == 0
            if "x2" in mode:
                assert torch.equal(hs2, hs) and torch.equal(cs2, cs)
            if "-g" in mode:                                        # same arithmetic in the same order as one group
                assert torch.equal(hs.float().cpu(), outs[1][0]) and torch.equal(cs.cpu(), outs[1][1])
        torch.testing.assert_close(hs[1:].float().cpu(), hs_ref, rtol=3e-2, atol=3e-2)
        torch.testing.assert_close(cs[1:].cpu(), cs_ref, rtol=3e-2, atol=3e-2)
        outs.append((hs.float().cpu(), cs.cpu(), gates.float().cpu()))
    # both kernels see identical bf16 operands: they agree far tighter than either agrees with fp32
    torch.testing.assert_close(outs[1][0], outs[0][0], rtol=2e-2, atol=1e-2)
    torch.testing.assert_close(outs[1][2], outs[0][2], rtol=2e-2, atol=1e-2)


@pytest.mark.parametrize("M", [8, 16, 64])
def test_lstm_bwd_step_fused_matches_two_launches(hip, M):
    """One-launch BPTT step (pointwise + recurrent GEMM, rotating buffers) against lstm_bwd_point + split-K GEMM over
    several consecutive steps."""
    H = hip
    Hd, steps = 512, 4
    g = torch.Generator().manual_seed(M)
    w_hh = bf(torch.randn(4 * Hd, Hd, generator=g) * 0.05).to(DEV)
    gates = bf(torch.rand(steps, M, 4 * Hd, generator=g) * 0.9 + 0.05).to(DEV)
    cs = (torch.randn(steps + 1, M, Hd, generator=g) * 0.5).to(DEV)
    dhs = bf(torch.randn(steps, M, Hd, generator=g) * 0.1).to(DEV)
    # reference: two launches per step, in-place buffers
    dh_rec, dc = torch.zeros(M, Hd, device=DEV), torch.zeros(M, Hd, device=DEV)
    dG0 = torch.zeros(steps, M, 4 * Hd, dtype=torch.bfloat16, device=DEV)
    for s_ in range(steps):
        t = steps - 1 - s_
        H.lstm_bwd_point(dhs[t], dh_rec, dc, gates[t], cs[t + 1], cs[t], dG0[t])
        H.gemm(H.NN, dG0[t], w_hh, dh_rec, M, Hd, 4 * Hd, flags=H.EPI_OUT_F32 | H.EPI_ATOMIC, splits=8, big=5)
    # fused
    R, DC = torch.zeros(3, M, Hd, device=DEV), torch.zeros(2, M, Hd, device=DEV)
    R[2].fill_(7.0)                                                 # must be zeroed by the first step
    dG1 = torch.zeros_like(dG0)
    for s_ in range(steps):
        t = steps - 1 - s_
        H.lstm_bwd_step(dhs[t], R[s_ % 3], R[(s_ + 1) % 3], R[(s_ + 2) % 3], DC[s_ % 2], DC[(s_ + 1) % 2], gates[t], cs[t + 1],
                        cs[t], dG1[t], w_hh)
    torch.testing.assert_close(dG1.float(), dG0.float(), rtol=2e-2, atol=1e-5)
    torch.testing.assert_close(R[steps % 3], dh_rec, rtol=2e-3, atol=2e-4)
    torch.testing.assert_close(DC[steps % 2], dc, rtol=2e-3, atol=1e-4)          # fed by dh_rec: float-atomics order


@pytest.mark.parametrize("M", [16, 64])
def test_lstm_bptt_persistent_matches_step_launches(hip, M):
    """Persistent BPTT (one launch per chunk, two chained launches here) against the one-launch-per-step kernel."""
    H = hip
    Hd, L = 512, 7
    g = torch.Generator().manual_seed(900 + M)
    w = bf(torch.randn(4 * Hd, Hd, generator=g) * 0.05).to(DEV)
    dhs = bf(torch.randn(L, M, Hd, generator=g) * 0.1).to(DEV)
    gates = bf(torch.rand(L, M, 4 * Hd, generator=g) * 0.9 + 0.05).to(DEV)
    cs = (torch.randn(L + 1, M, Hd, generator=g) * 0.5).to(DEV)
    # reference: the per-step kernel with its rotating buffers
    R, DC = torch.zeros(3, M, Hd, device=DEV), torch.zeros(2, M, Hd, device=DEV)
    dG_ref = torch.zeros(L, M, 4 * Hd, dtype=torch.bfloat16, device=DEV)
    for s_, k in enumerate(range(L - 1, -1, -1)):
        if k > 0:
            H.lstm_bwd_step(dhs[k], R[s_ % 3], R[(s_ + 1) % 3], R[(s_ + 2) % 3], DC[s_ % 2], DC[(s_ + 1) % 2], gates[k], cs[k + 1],
                            cs[k], dG_ref[k], w)
        else:
            H.lstm_bwd_point(dhs[k], R[s_ % 3], DC[s_ % 2], gates[k], cs[k + 1], cs[k], dG_ref[k])
    dG = torch.zeros_like(dG_ref)
    rec, dc = torch.zeros(M, Hd, device=DEV), torch.zeros(M, Hd, device=DEV)
    bx, err = H.lstm_bx(DEV), torch.zeros(1, dtype=torch.int32, device=DEV)
    for rep in range(2):                                            # twice: the scratch is reused across launches (epochs)
        H.lstm_bptt(dhs, gates, cs, dG, w, rec, dc, bx, err, k_hi=L - 1, steps=4, first=True, carry=True)
        H.lstm_bptt(dhs, gates, cs, dG, w, rec, dc, bx, err, k_hi=L - 5, steps=3, first=False, carry=False)
    torch.cuda.synchronize()
    assert int(err.item()) == 0
    torch.testing.assert_close(dG.float().cpu(), dG_ref.float().cpu(), rtol=3e-2, atol=2e-4)
    # row groups (independent sets of 32 workgroups, M/16/groups row blocks each): the same arithmetic in the same order,
    # whatever the split - and the scratch may see different splits from launch to launch
    for g1, g2 in ((1, 1), (2, 4), (4, 2), (4, 1)):
        if M // 16 % g1 or M // 16 % g2:
            continue
        dG2 = torch.zeros_like(dG_ref)
        H.lstm_bptt(dhs, gates, cs, dG2, w, rec, dc, bx, err, k_hi=L - 1, steps=4, first=True, carry=True, groups=g1)
        H.lstm_bptt(dhs, gates, cs, dG2, w, rec, dc, bx, err, k_hi=L - 5, steps=3, first=False, carry=False, groups=g2)
        torch.cuda.synchronize()
        assert int(err.item()) == 0 and torch.equal(dG2, dG), (g1, g2)


@pytest.mark.parametrize("frames", [5, 130])
def test_conv1_fwd_pair_matches_single_network_launches(hip, frames):
    """Two weight sets per tile over the same frames: bit-identical to two single-network launches."""
    H = hip
    store = torch.randint(0, 256, (4 * (frames + 7), 7056), dtype=torch.uint8).to(DEV)         # a plane store
    off = (torch.randperm(4 * (frames + 7) - 3)[:frames] * 7056).to(DEV)                       # any 4 consecutive planes
    outs = []
    nets = [(_w1_planes(w), b) for w, b in (_conv_setup(1, 1, 40 + i)[2:4] for i in range(2))]
    for w, b in nets:
        o = torch.zeros(frames, 20, 20, 32, dtype=torch.bfloat16, device=DEV)
        H.conv_fwd(1, store, w.to(DEV), b.to(DEV), o, frames, frame_off=off)
        outs.append(o)
    p1, p2 = torch.zeros_like(outs[0]), torch.zeros_like(outs[0])
    H.conv1_fwd_pair(store, off, frames, nets[0][0].to(DEV), nets[0][1].to(DEV), p1, nets[1][0].to(DEV), nets[1][1].to(DEV), p2)
    assert torch.equal(p1, outs[0]) and torch.equal(p2, outs[1])


@pytest.mark.parametrize("layer", [2, 3])
@pytest.mark.parametrize("frames", [1, 2, 3, 37, 511, 1300])
def test_conv_wgrad_halo(hip, layer, frames):
    """Weight + bias gradient of conv2 / conv3 through the LDS-halo kernel (transposed LDS reads of planar frame images, one
    accumulation pass per workgroup) against fp32 autograd; both outputs accumulate on top of what is there."""
    H = hip
    x, xf, w, bias, (cin, cout, ks, s, hw) = _conv_setup(layer, frames, layer * 19 + frames)
    oh = (hw - ks) // s + 1
    g = torch.Generator().manual_seed(frames)
    dy = bf(torch.randn(frames, oh, oh, cout, generator=g) * 0.1)
    wf = w.float().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    bf_ = bias.clone().requires_grad_(True)
    F.conv2d(xf.permute(0, 3, 1, 2), wf, bf_, stride=s).backward(dy.float().permute(0, 3, 1, 2))
    ref_w = wf.grad.permute(0, 2, 3, 1).reshape(cout, -1)
    dw = torch.full((cout, ks * ks * cin), 0.25, device=DEV)
    db = torch.full((cout,), -0.5, device=DEV)
    H.conv_wgrad_halo(layer, x.to(DEV), dy.to(DEV), dw, db, frames, H.conv_wgrad_ws(layer, DEV))
    scale = float(ref_w.abs().max())
    torch.testing.assert_close(dw.cpu() - 0.25, ref_w, rtol=2e-3, atol=2e-3 * scale)
    torch.testing.assert_close(db.cpu() + 0.5, bf_.grad, rtol=2e-3, atol=2e-3 * float(bf_.grad.abs().max()))
    # and against the split-K implicit GEMM it replaces
    dw2, db2 = torch.zeros_like(dw), torch.zeros_like(db)
    H.conv_bwd(layer, x.to(DEV), dy.to(DEV), None, dw2, db2, None, frames)
    torch.testing.assert_close(dw.cpu() - 0.25, dw2.cpu(), rtol=2e-3, atol=2e-3 * scale)
    # fewer workgroups (a neighbour holds CUs: max_wgs): other shares per workgroup, same sums up to fp32 summation order
    dw3, db3 = torch.zeros_like(dw), torch.zeros_like(db)
    H.conv_wgrad_halo(layer, x.to(DEV), dy.to(DEV), dw3, db3, frames, H.conv_wgrad_ws(layer, DEV), max_wgs=5)
    torch.testing.assert_close(dw3.cpu(), ref_w, rtol=2e-3, atol=2e-3 * scale)
    torch.testing.assert_close(db3.cpu(), bf_.grad, rtol=2e-3, atol=2e-3 * float(bf_.grad.abs().max()))


@pytest.mark.parametrize("frames", [1, 2, 5, 130, 700])
def test_conv1_wgrad_halo(hip, frames):
    """conv1 weight + bias gradient over a plane store (overlapping windows, as consecutive observations are stored) through
    the space-to-depth LDS-halo kernel against fp32 autograd and against the split-K implicit GEMM it replaces."""
    H = hip
    g = torch.Generator().manual_seed(frames)
    nplanes = frames + 9
    store = torch.randint(0, 256, (nplanes, 7056), dtype=torch.uint8, generator=g)
    first = torch.randint(0, nplanes - 3, (frames,), generator=g)                      # any 4 consecutive planes
    off = first * 7056
    stacks = store.view(nplanes, 84, 84)[first[:, None] + torch.arange(4)[None, :]]   # [frames, 4, 84, 84] = NCHW
    dy = bf(torch.randn(frames, 20, 20, 32, generator=g) * 0.1)
    w = torch.zeros(32, 4, 8, 8, requires_grad=True)
    b = torch.zeros(32, requires_grad=True)
    F.conv2d(stacks.float() / 255, w, b, stride=4).backward(dy.float().permute(0, 3, 1, 2))
    dw = torch.full((32, 256), 0.25, device=DEV)
    db = torch.full((32,), -0.5, device=DEV)
    H.conv1_wgrad_halo(store.to(DEV), off.to(DEV), dy.to(DEV), dw, db, frames, H.conv_wgrad_ws(1, DEV))
    ref = w.grad.reshape(32, 256)
    torch.testing.assert_close(dw.cpu() - 0.25, ref, rtol=2e-3, atol=2e-3 * float(ref.abs().max()))
    torch.testing.assert_close(db.cpu() + 0.5, b.grad, rtol=2e-3, atol=2e-3 * float(b.grad.abs().max()))
    dw2, db2 = torch.zeros_like(dw), torch.zeros_like(db)
    H.conv_bwd(1, store.to(DEV), dy.to(DEV), None, dw2, db2, None, frames, frame_off=off.to(DEV), splits=16)
    torch.testing.assert_close(dw.cpu() - 0.25, dw2.cpu(), rtol=2e-3, atol=2e-3 * float(ref.abs().max()))


# ---- large-tile GEMM (csrc/hip/gemm_big.hip) ---------------------------------------------------------------------------
@pytest.mark.parametrize("mode", ["NT", "NN", "TN"])
@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 5])
def test_gemm_big_matches_fp32_product(mode, tile):
    """Every tile shape x operand orientation on sizes that are NOT multiples of the tile (edge rows / columns / k come
    from the zero page), with the fused bias (NT) and ReLU-mask (NN) epilogues, fp32 and bf16 outputs."""
    from distributed_rl_amd.ops import hip as H
    torch.manual_seed(3)
    M, N, K = 328, 456, 200
    dev = "cuda"
    rnd = lambda *s: (torch.rand(*s, device=dev) * 2 - 1).to(torch.bfloat16)   # noqa: E731
    if mode == "NT":
        a, b = rnd(M, K), rnd(N, K)
        ref = a.float() @ b.float().t()
    elif mode == "NN":
        a, b = rnd(M, K), rnd(K, N)
        ref = a.float() @ b.float()
    else:
        a, b = rnd(K, M), rnd(K, N)
        ref = a.float().t() @ b.float()
    bias = torch.randn(N, device=dev) if mode == "NT" else None
    mask = rnd(M, N) if mode == "NN" else None
    if bias is not None:
        ref = ref + bias
    if mask is not None:
        ref = ref * (mask.float() > 0)
    c = torch.full((M, N), 7.0, dtype=torch.bfloat16 if mode == "NN" else torch.float32, device=dev)
    H.gemm_big({"NT": H.NT, "NN": H.NN, "TN": H.TN}[mode], a, b, c, bias=bias, mask=mask, tile=tile)
    torch.cuda.synchronize()
    tol = dict(rtol=2e-2, atol=6e-2) if mode == "NN" else dict(rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(c.float(), ref, **tol)


def test_gemm_big_two_problems_per_launch():
    from distributed_rl_amd.ops import hip as H
    torch.manual_seed(4)
    M, N, K = 520, 384, 320
    rnd = lambda *s: (torch.rand(*s, device="cuda") * 2 - 1).to(torch.bfloat16)   # noqa: E731
    a, a2, b, b2 = rnd(M, K), rnd(M, K), rnd(N, K), rnd(N, K)
    bias, bias2 = torch.randn(N, device="cuda"), torch.randn(N, device="cuda")
    c, c2 = torch.zeros(M, N, device="cuda"), torch.zeros(M, N, device="cuda")
    H.gemm_big(H.NT, a, b, c, bias=bias, second=(a2, b2, c2, bias2, None))
    torch.cuda.synchronize()
    torch.testing.assert_close(c, a.float() @ b.float().t() + bias, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(c2, a2.float() @ b2.float().t() + bias2, rtol=1e-4, atol=1e-4)


# ---- fused torso forward (conv_torso.hip) ------------------------------------------------------------------------------
def _torso_weights(seed):
    """(w1 planes-order, w1 HWC-order, w2, w3), (b1, b2, b3) of one network."""
    ws, bs = [], []
    for layer in (1, 2, 3):
        _, _, w, b, _ = _conv_setup(layer, 1, seed + layer)
        ws.append(w); bs.append(b)
    return (_w1_planes(ws[0]), ws[0], ws[1], ws[2]), bs


def _three_launch_torso(H, x, off, frames, w1, w2, w3, bs):
    a1 = torch.zeros(frames, 20, 20, 32, dtype=torch.bfloat16, device=DEV)
    a2 = torch.zeros(frames, 9, 9, 64, dtype=torch.bfloat16, device=DEV)
    a3 = torch.zeros(frames, 7, 7, 64, dtype=torch.bfloat16, device=DEV)
    H.conv_fwd(1, x, w1, bs[0], a1, frames, frame_off=off)
    H.conv_fwd(2, a1, w2, bs[1], a2, frames)
    H.conv_fwd(3, a2, w3, bs[2], a3, frames)
    return a1, a2, a3


@pytest.mark.parametrize("f1,f2,save1", [(1, 0, (0, 1)), (5, 3, (1, 4)), (64, 0, (0, 0)), (300, 150, (0, 150)), (1024, 512, (0, 512)),
                                         (1537, 700, (100, 1000))])
def test_torso_fwd_fused_matches_three_launches(hip, f1, f2, save1):
    """conv1 -> conv2 -> conv3 in one launch over a plane store, one and two networks, fewer / more frames than workgroups:
    act3 is bit-identical to the three per-layer launches (same products in the same order), act1 / act2 are written
    exactly for the save range (bit-identical there, untouched elsewhere), and the guard row behind act3 stays untouched."""
    H = hip
    n = f1 + f2
    store = torch.randint(0, 256, (4 * (n + 7), 7056), dtype=torch.uint8).to(DEV)              # a plane store
    off = (torch.randperm(4 * (n + 7) - 3)[:n] * 7056).to(DEV)                                 # any 4 consecutive planes
    nets, refs = [], []
    for k, (f, o, save) in enumerate(((f1, off[:f1], save1), (f2, off[f1:], (0, 0)))):
        if f == 0:
            nets.append(None)
            continue
        (w1p, _, w2, w3), bs = _torso_weights(50 + 10 * k)
        w1p, w2, w3, bs = w1p.to(DEV), w2.to(DEV), w3.to(DEV), [b.to(DEV) for b in bs]
        o = o.contiguous()
        refs.append(_three_launch_torso(H, store, o, f, w1p, w2, w3, bs))
        a1 = torch.full((f, 20, 20, 32), 7.0, dtype=torch.bfloat16, device=DEV)
        a2 = torch.full((f, 9, 9, 64), 7.0, dtype=torch.bfloat16, device=DEV)
        a3 = torch.full((f + 1, 7, 7, 64), 7.0, dtype=torch.bfloat16, device=DEV)
        nets.append((H.TorsoNet(o, f, (w1p, w2, w3), bs, a3, a1, a2, save), a1, a2, a3, save))
    H.torso_fwd(store, nets[0][0], nets[1][0] if nets[1] is not None else None)
    torch.cuda.synchronize()
    for net, ref in zip([x for x in nets if x is not None], refs):
        _, a1, a2, a3, (lo, hi) = net
        f = a1.shape[0]
        assert torch.equal(a3[:f], ref[2]) and bool((a3[f] == 7.0).all())
        for a, rf in ((a1, ref[0]), (a2, ref[1])):
            assert torch.equal(a[lo:hi], rf[lo:hi])
            assert bool((a[:lo] == 7.0).all()) and bool((a[hi:] == 7.0).all())
    # next to a persistent kernel of another stream the grid leaves CUs out: other shares per workgroup, same bits
    H.torso_set_busy(64)
    try:
        for net in nets:
            if net is not None:
                net[3][:-1].zero_()
        H.torso_fwd(store, nets[0][0], nets[1][0] if nets[1] is not None else None)
        torch.cuda.synchronize()
    finally:
        H.torso_set_busy(0)
    for net, ref in zip([x for x in nets if x is not None], refs):
        assert torch.equal(net[3][:-1], ref[2])


def test_torso_fwd_fused_dense_stacks_and_weight_prepass(hip):
    """Dense HWC u8 stacks (the actors' input form) through the fused kernel == the three launches; the extra workgroups
    of the launch write the same flipped / transposed weights as conv_halo_wprep."""
    H = hip
    f = 70
    x = torch.randint(0, 256, (f, 84, 84, 4), dtype=torch.uint8).to(DEV)
    (_, w1, w2, w3), bs = _torso_weights(80)
    w1, w2, w3, bs = w1.to(DEV), w2.to(DEV), w3.to(DEV), [b.to(DEV) for b in bs]
    ref = _three_launch_torso(H, x, None, f, w1, w2, w3, bs)
    a3 = torch.zeros(f, 7, 7, 64, dtype=torch.bfloat16, device=DEV)
    wt2, wt3 = torch.zeros(4, 32, 2, 2, 64, dtype=torch.bfloat16, device=DEV), torch.zeros(64, 3, 3, 64, dtype=torch.bfloat16, device=DEV)
    H.torso_fwd(x, H.TorsoNet(None, f, (w1, w2, w3), bs, a3), prep=(w2, w3, wt2, wt3))
    r2, r3 = torch.zeros_like(wt2), torch.zeros_like(wt3)
    H.conv_halo_wprep(w2, w3, r2, r3)
    torch.cuda.synchronize()
    assert torch.equal(a3, ref[2])
    assert torch.equal(wt2, r2) and torch.equal(wt3, r3)
