"""utils parity (reference tests/libs/test_utils.py:9-35) + ε ladder and n-step helpers."""
import numpy as np
import torch

from distributed_rl_amd import utils


def test_preprocess():
    src = np.random.randint(256, size=(10, 10, 3), dtype=np.uint8)
    dst = utils.preprocess(src)
    assert src.shape[:2] == dst.shape[1:] and src.shape[2] == dst.shape[0]
    assert dst.dtype == np.float32 and dst.max() <= 1.0
    assert utils.preprocess(src, shape=(5, 5)).shape[1:] == (5, 5)
    assert utils.preprocess(src, gray=True).shape == src.shape[:2]
    assert utils.preprocess(src, gray=True, dtype=np.uint8).dtype == np.uint8


def test_dumps_loads():
    src = torch.rand(10)
    torch.testing.assert_close(src, utils.loads(utils.dumps(src)))


def test_rescale():
    src = 10 * (torch.rand(10) * 2.0 - 1.0)
    for eps in (0.1, 1e-3, 0):
        torch.testing.assert_close(src, utils.inv_rescale(utils.rescale(src, eps), eps), rtol=1e-3, atol=1e-3)
        torch.testing.assert_close(src, utils.rescale(utils.inv_rescale(src, eps), eps), rtol=1e-3, atol=1e-3)


def test_epsilon_ladder():
    # ape_x/actor.py:66-69
    assert utils.actor_epsilon(1, 1) == 0.4
    assert abs(utils.actor_epsilon(1, 4) - 0.4) < 1e-12
    assert abs(utils.actor_epsilon(4, 4) - 0.4 ** 8) < 1e-12
    lad = utils.epsilon_ladder(4)
    assert np.all(np.diff(lad) < 0)


def test_nstep_return_orders():
    r = [1.0, 2.0, 3.0]
    g = 0.9
    assert abs(utils.nstep_return(r, g) - (1 + 0.9 * 2 + 0.81 * 3)) < 1e-9
    assert abs(utils.nstep_return(r, g, reference_order=True) - (0.81 * 1 + 0.9 * 2 + 3)) < 1e-9


def test_epsilon_greedy_single():
    class Net:
        n_action = 3

        def __call__(self, x):
            return torch.tensor([[0.0, 5.0, 1.0]])
    assert utils.epsilon_greedy(torch.zeros(1), Net(), eps=0.0).item() == 1
    acts = {utils.epsilon_greedy(torch.zeros(1), Net(), eps=1.1).item() for _ in range(100)}
    assert acts == {0, 1, 2}


def test_ctypes_mirrors_of_device_structs():
    """ops/hip.py mirrors two C structs by hand; the C side pins their sizes with static_asserts (gemm.hip TreeExtra = 168
    bytes, common.h StepState = 64 bytes), this pins the Python side."""
    import ctypes
    from distributed_rl_amd.ops import hip as H
    assert ctypes.sizeof(H.TreeExtra) == 168
    assert H.TreeExtra.off.offset == 16 and H.TreeExtra.cap.offset == 64 and H.TreeExtra.st_end.offset == 112
    assert H.TreeExtra.sq_g.offset == 120 and H.TreeExtra.zero_n.offset == 152 and H.TreeExtra.sq_blocks.offset == 160
    assert H.STEP_STATE_BYTES == 64
    lay = H.TreeLayout(2_000_000)
    assert lay.n_levels == 4 and lay.sizes[0] >= 2_000_000 and lay.sizes[-1] == 64 and lay.total == sum(lay.sizes)
