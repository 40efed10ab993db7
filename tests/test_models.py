"""Model parity: state_dict keys/shapes, checkpoint interchange with a reference-shaped
nn.Module, kernel-layout packing round trip, stateful vs time-batched LSTM paths."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from distributed_rl_amd.models import DuelingDQN, DuelingLSTMDQN, ParamPack
from distributed_rl_amd.utils import Sequence, Transition


class RefDQN(nn.Module):
    """Module shaped exactly like the reference's DuelingDQN (libs/models.py:7-29), written from SURVEY §2.6."""

    def __init__(self, A):
        super().__init__()
        self.conv1, self.conv2, self.conv3 = nn.Conv2d(4, 32, 8, 4), nn.Conv2d(32, 64, 4, 2), nn.Conv2d(64, 64, 3, 1)
        self.adv1, self.adv2, self.val1, self.val2 = nn.Linear(3136, 512), nn.Linear(512, A), nn.Linear(3136, 512), nn.Linear(512, 1)

    def forward(self, x):
        x = F.relu(self.conv3(F.relu(self.conv2(F.relu(self.conv1(x)))))).view(x.size(0), -1)
        adv, val = self.adv2(F.relu(self.adv1(x))), self.val2(F.relu(self.val1(x)))
        return val + adv - adv.mean(1, keepdim=True)


APEX_KEYS = {"conv1.weight": (32, 4, 8, 8), "conv1.bias": (32,), "conv2.weight": (64, 32, 4, 4), "conv2.bias": (64,),
             "conv3.weight": (64, 64, 3, 3), "conv3.bias": (64,), "adv1.weight": (512, 3136), "adv1.bias": (512,),
             "adv2.weight": (4, 512), "adv2.bias": (4,), "val1.weight": (512, 3136), "val1.bias": (512,),
             "val2.weight": (1, 512), "val2.bias": (1,)}


def test_apex_state_dict_and_param_count(tmp_path):
    m = DuelingDQN(4)
    sd = m.state_dict()
    assert {k: tuple(v.shape) for k, v in sd.items()} == APEX_KEYS
    assert sum(p.numel() for p in m.parameters()) == 3_292_837
    ref = RefDQN(4)
    torch.save(sd, tmp_path / "model_0.pth")
    ref.load_state_dict(torch.load(tmp_path / "model_0.pth", weights_only=True))    # ours → reference
    x = torch.rand(3, 4, 84, 84)
    torch.testing.assert_close(m(x), ref(x))
    m2 = DuelingDQN(4)
    m2.load_state_dict(ref.state_dict())                                            # reference → ours
    torch.testing.assert_close(m2(x), ref(x))


def test_uint8_hwc_input_equals_float_chw():
    m = DuelingDQN(6)
    u8 = torch.randint(0, 256, (2, 84, 84, 4), dtype=torch.uint8)
    fl = u8.permute(0, 3, 1, 2).float() / 255
    torch.testing.assert_close(m(u8), m(fl))


def test_r2d2_state_dict():
    m = DuelingLSTMDQN(4, 2)
    sd = m.state_dict()
    assert "hx" not in sd and "cx" not in sd                                        # models.py:70-71
    assert tuple(sd["lstm.weight_ih"].shape) == (2048, 3136) and tuple(sd["lstm.weight_hh"].shape) == (2048, 512)
    assert "adv2.bias" not in sd and tuple(sd["adv2.weight"].shape) == (4, 512)     # models.py:67
    assert sum(p.numel() for p in m.parameters()) == 8_081_057


def test_packing_round_trip():
    for m in (DuelingDQN(5), DuelingLSTMDQN(5, 1)):
        pack = ParamPack.from_module(m)
        sd2 = pack.state_dict()
        for k, v in m.state_dict().items():
            assert torch.equal(v, sd2[k]), k
        assert pack.flat.numel() >= sum(p.numel() for p in m.parameters())


def test_packed_fc1_is_hwc_permutation():
    m = DuelingDQN(4)
    pack = ParamPack.from_module(m)
    x = torch.rand(2, 4, 84, 84)
    feat_chw = m.features(x)                                                         # NCHW flatten (models.py:24)
    feat_hwc = feat_chw.view(2, 64, 7, 7).permute(0, 2, 3, 1).reshape(2, -1)
    h_ref = torch.cat([m.adv1(feat_chw), m.val1(feat_chw)], 1)
    h_pk = feat_hwc @ pack.view("fc1.w").t() + pack.view("fc1.b")
    torch.testing.assert_close(h_pk, h_ref, rtol=1e-4, atol=1e-4)


def test_lstm_stateful_equals_unroll():
    torch.manual_seed(0)
    m = DuelingLSTMDQN(3, 2, n_burn_in=2, nstep_return=2)
    frames = torch.rand(5, 2, 4, 84, 84)
    h0, c0 = torch.randn(2, 512) * 0.1, torch.randn(2, 512) * 0.1
    m.set_state((h0, c0))
    qs = torch.stack([m(frames[t]) for t in range(5)])
    q2, (h, c) = m.unroll(frames, h0, c0, grad_from=2)
    torch.testing.assert_close(qs, q2, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(h, m.hx, rtol=1e-4, atol=1e-5)
    m.reset(True)
    assert float(m.hx.abs().sum()) == 0.0
    hx, cx = m.get_state()
    assert hx.device.type == "cpu"


def test_calc_priorities_reference_signatures():
    torch.manual_seed(1)
    m, t = DuelingDQN(4), DuelingDQN(4)
    trans = [Transition(torch.rand(4, 84, 84), torch.tensor([i % 4]), torch.tensor([0.5]), torch.rand(4, 84, 84), torch.tensor([0.0]))
             for i in range(5)]
    delta, prio = m.calc_priorities(t, trans, gamma=0.999 ** 3)
    assert delta.shape == (5, 1) and prio.shape == (5, 1) and delta.requires_grad and not prio.requires_grad
    r, tg = DuelingLSTMDQN(4, 2, n_burn_in=2, nstep_return=2), DuelingLSTMDQN(4, 2, n_burn_in=2, nstep_return=2)
    seqs = [Sequence([Transition(torch.rand(4, 84, 84), torch.tensor([1]), torch.tensor([0.1]), done=torch.tensor([0.0]))
                      for _ in range(8 if b == 0 else 6)], (torch.zeros(1, 512), torch.zeros(1, 512))) for b in range(2)]
    loss, prio = r.calc_priorities(tg, seqs, gamma=0.997 ** 2)
    assert loss.shape == (2, 1) and prio.shape == (2, 1)
    loss.sum().backward()
    assert r.lstm.weight_hh.grad is not None
