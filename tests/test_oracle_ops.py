"""Golden semantics of the fused ops' oracles against straightforward autograd re-derivations
(models.py:41-48, 142-162; learner.py:66-68,72)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from distributed_rl_amd.ops import oracle
from distributed_rl_amd.utils import inv_rescale, rescale


def test_dueling_combine_and_backward():
    v, a = torch.randn(7, 1, requires_grad=True), torch.randn(7, 5, requires_grad=True)
    q = oracle.dueling_combine(v, a)
    assert torch.allclose(q.mean(1, keepdim=True), v, atol=1e-6)
    g = torch.randn_like(q)
    q.backward(g)
    dv, da = oracle.dueling_combine_backward(g)
    torch.testing.assert_close(dv, v.grad)
    torch.testing.assert_close(da, a.grad)


def test_is_weights_and_beta():
    assert oracle.beta_schedule(0, 0.4, 1000) == 0.4
    assert oracle.beta_schedule(10 ** 9, 0.4, 1000) == 1.0
    p = torch.tensor([1.0, 2.0, 4.0])
    w = oracle.is_weights(p, 7.0, 3, 0.5)
    ref = (3 * p.numpy() / 7.0) ** -0.5
    ref /= ref.max()
    assert np.allclose(w.numpy(), ref, atol=1e-6)


@pytest.mark.parametrize("B,A", [(1, 4), (50, 6), (512, 18)])
def test_apex_td_matches_autograd(B, A):
    g = torch.Generator().manual_seed(B)
    q_s = torch.randn(B, A, generator=g, requires_grad=True)
    q_s2, q_t2 = torch.randn(B, A, generator=g), torch.randn(B, A, generator=g)
    a = torch.randint(0, A, (B,), generator=g)
    r = torch.randn(B, generator=g) * 2
    d = (torch.rand(B, generator=g) < 0.3).float()
    w = torch.rand(B, generator=g)
    td = oracle.apex_td(q_s.detach(), q_s2, q_t2, a, r, d, w, 0.97, 0.6)
    # reference formulation (models.py:41-47, learner.py:72)
    sav = q_s.gather(1, a.view(-1, 1))
    na = q_s2.argmax(1, keepdim=True)
    nsv = q_t2.gather(1, na)
    y = nsv * 0.97 * (1 - d.view(-1, 1)) + r.view(-1, 1)
    delta = F.smooth_l1_loss(sav, y, reduction="none")
    loss = (delta * w.view(-1, 1)).mean()
    loss.backward()
    torch.testing.assert_close(td.loss, loss.detach())
    torch.testing.assert_close(td.dq, q_s.grad)
    torch.testing.assert_close(td.prio, (delta.detach().abs() + 1e-5).pow(0.6).squeeze(1))


def _r2d2_reference_loop(qo, qt, a, r, d, n, burn, gamma_n, eta, done_offset):
    """Literal time loop of models.py:143-162 on precomputed Q tensors (full-length sequences)."""
    T, B, A = qo.shape
    deltas = []
    for t in range(burn, T - n):
        sav = qo[t].gather(1, a[t].view(-1, 1))
        na = qo[t + n].argmax(1, keepdim=True)
        nsv = qt[t + n].gather(1, na)
        dd = d[t + done_offset].view(-1, 1)
        y = rescale(inv_rescale(nsv) * gamma_n * (1 - dd) + r[t].view(-1, 1))
        deltas.append((sav - y).abs())
    delta = torch.stack(deltas)
    return delta.pow(2).sum(0).squeeze(1), (eta * delta.max(0)[0] + (1 - eta) * delta.mean(0)).squeeze(1)


@pytest.mark.parametrize("done_offset", [0, 5])
def test_r2d2_td_matches_reference_loop(done_offset):
    T, B, A, n, burn = 30, 6, 4, 5, 8
    g = torch.Generator().manual_seed(3)
    qo = torch.randn(T, B, A, generator=g, requires_grad=True)
    qt = torch.randn(T, B, A, generator=g)
    a = torch.randint(0, A, (T, B), generator=g)
    r = torch.randn(T, B, generator=g)
    d = torch.zeros(T, B)
    d[T - 1, 0] = 1.0
    w = torch.rand(B, generator=g)
    lengths = torch.full((B,), T)
    mask = oracle.r2d2_learn_mask(lengths, d, T, burn, n, done_offset)
    if done_offset == 0:
        mask[T - 1, 0] = False       # compare on the reference window only
    assert mask[:, 1].nonzero().flatten().tolist() == list(range(burn, T - n))
    td = oracle.r2d2_td(qo.detach(), qt, a, r, d, mask, w, 0.98, n, 0.9, done_offset=done_offset)
    sl, pr = _r2d2_reference_loop(qo, qt, a, r, d, n, burn, 0.98, 0.9, done_offset)
    torch.testing.assert_close(td.seq_loss, sl.detach())
    torch.testing.assert_close(td.prio, pr.detach())
    (sl * w).mean().backward()
    torch.testing.assert_close(td.dq, qo.grad)


def test_r2d2_mask_variable_length_and_terminal():
    T, n, burn = 20, 3, 4
    lengths = torch.tensor([20, 12])
    d = torch.zeros(T, 2)
    d[9:12, 1] = 1.0                 # flushed tail windows of a terminated episode
    m = oracle.r2d2_learn_mask(lengths, d, T, burn, n, 0)
    assert m[:, 0].nonzero().flatten().tolist() == list(range(4, 17))
    assert m[:, 1].nonzero().flatten().tolist() == list(range(4, 12))   # 9..11 learn without bootstrap


def test_lstm_unroll_matches_lstmcell():
    torch.manual_seed(0)
    cell = torch.nn.LSTMCell(12, 8)
    x = torch.randn(5, 3, 12)
    h, c = torch.zeros(3, 8), torch.zeros(3, 8)
    hs_ref = []
    for t in range(5):
        h, c = cell(x[t], (h, c))
        hs_ref.append(h)
    xg = x @ cell.weight_ih.t() + cell.bias_ih + cell.bias_hh
    hs, cs = oracle.lstm_unroll(xg, cell.weight_hh, torch.zeros(3, 8), torch.zeros(3, 8))
    torch.testing.assert_close(hs, torch.stack(hs_ref))


def test_optimizer_oracles_match_torch():
    torch.manual_seed(0)
    p0 = torch.randn(100)
    for kind in ("rms", "adam"):
        p = torch.nn.Parameter(p0.clone())
        opt = torch.optim.RMSprop([p], lr=6.25e-5, alpha=0.95, eps=1.5e-7) if kind == "rms" else \
            torch.optim.Adam([p], lr=1e-4, eps=1e-3)
        q, s1, s2 = p0.clone(), torch.zeros(100), torch.zeros(100)
        for step in range(1, 4):
            g = torch.randn(100) * 30
            p.grad = g.clone()
            torch.nn.utils.clip_grad_norm_([p], 40.0)
            opt.step()
            coef = oracle.clip_coef(g, 40.0)
            if kind == "rms":
                oracle.rmsprop_step(q, g, s1, 6.25e-5, 0.95, 1.5e-7, coef)
            else:
                oracle.adam_step(q, g, s1, s2, step, 1e-4, 0.9, 0.999, 1e-3, coef)
            torch.testing.assert_close(q, p.detach(), rtol=1e-5, atol=1e-7)


def test_eps_greedy_batched():
    q = torch.tensor([[0.0, 2.0, 1.0]] * 4)
    eps = torch.tensor([0.0, 1.0, 0.5, 0.5])
    ue = torch.tensor([0.3, 0.3, 0.6, 0.4])
    ua = torch.tensor([0.9, 0.9, 0.9, 0.1])
    assert oracle.eps_greedy(q, eps, ue, ua).tolist() == [1, 2, 1, 0]
