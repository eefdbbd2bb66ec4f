"""az_tower_h2.hip ("f16x2"): the 64-channel GomokuModel-family tower at f32-level accuracy on the f16 MFMA pipe -- every f32
operand split into two f16 halves, three MFMAs per product block, f32 accumulation.  EXPLORATORY / SECONDARY: not the reference's
arithmetic (nn_models/base.py:17-19 is plain f32) and not bit-equal to the f32 tower; held to the f32 tower's own accuracy bars
(tests/test_gpu_tower.py): logits within 2e-5 of the logit scale of an f64 evaluation of the same network, values within 2e-5,
softmax priors within 1e-5 relative -- and deterministic per board."""
import numpy as np
import pytest
import torch

from .test_gpu_tower import _randomised

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("board,blocks,batch", [(15, 2, 64), (15, 2, 1100), (15, 3, 33), (9, 2, 100), (9, 1, 7), (15, 0, 5), (9, 0, 4),
                                                (15, 10, 600), (9, 6, 50), (9, 2, 4096)])
def test_f16x2_tower_meets_the_f32_towers_accuracy_bars(board, blocks, batch):
    from alphazero_amd.fused import FusedTowerEvaluator
    from alphazero_amd.nets import GomokuNet

    torch.manual_seed(0)
    net = _randomised(GomokuNet(board, 64, blocks)).eval().cuda()
    ev = FusedTowerEvaluator(net, "f16x2")
    ev32 = FusedTowerEvaluator(net, "f32")
    x = torch.randint(-1, 2, (batch, 9, board, board), device="cuda").float()
    x[:, 8] = torch.where(torch.rand(batch, 1, 1, device="cuda") < 0.5, -1.0, 1.0)
    torch.backends.cudnn.allow_tf32 = False
    with torch.inference_mode():
        ref64_p, ref64_v = net.double()(x.double())
    net.float()
    p, v = ev(x.contiguous())
    p, v = p.clone(), v.clone()
    p32, _ = ev32(x.contiguous())
    assert not ev.overflowed()
    scale = float(ref64_p.abs().max())
    err = float((p - ref64_p).abs().max())
    err32 = float((p32 - ref64_p).abs().max())
    print(f"f16x2 max |logit error| {err:.3e} (f32 tower {err32:.3e}) at logit scale {scale:.3f}")
    assert err <= 2e-5 * scale + 2e-6, (err, scale)
    assert float((v - ref64_v).abs().max()) <= 2e-5
    pr, pr64 = torch.softmax(p.double(), 1), torch.softmax(ref64_p, 1)
    assert float(((pr - pr64).abs() / pr64).max()) <= 1e-5 * max(1.0, scale)
    # deterministic per board: a second launch and a permuted batch give the same rows bit for bit
    perm = torch.randperm(batch, device="cuda")
    p2, v2 = ev(x[perm].contiguous())
    assert torch.equal(p2, p[perm]) and torch.equal(v2, v[perm])


@pytest.mark.parametrize("n,blocks,G", [(15, 2, 2100), (9, 2, 5300)])
def test_f16x2_tower_packed_flags_equals_dense(n, blocks, G):
    """the engine's packed leaf batch + needs-the-network flags == the dense planes of the same positions, bit for bit, for every
    flag pattern (rows that are off keep their old contents)"""
    from alphazero_amd import Engine
    from alphazero_amd.fused import FusedTowerEvaluator
    from alphazero_amd.nets import GomokuNet

    torch.manual_seed(3)
    net = _randomised(GomokuNet(n, 64, blocks)).eval().cuda()
    ev = FusedTowerEvaluator(net, "f16x2")
    eng = Engine(2, G, 8, board_size=n, nn_dtype=torch.float32, device="cuda:0")
    A = eng.dims.n_action
    nn, lg, vl = eng.new_nn_input(), eng.new(G, A), eng.new(G)
    it = torch.ones(G, device="cuda")
    for _ in range(6):  # a few plies so that the boards differ
        eng.begin_search()
        for _ in range(3):
            eng.select_gather(nn)
            eng.eval_hash(nn, lg, vl, 6.0)
            eng.expand_backup(lg, vl, None)
        eng.play_moves(it, torch.empty(G, A, device="cuda").exponential_(1))
    eng.begin_search()
    eng.select_gather(nn)  # dense planes + packed ring of the same leaves
    pd, vd = ev(nn.contiguous())
    pd, vd = pd.clone(), vd.view(-1).clone()
    ring, leaf, _, _, flags = eng.leaf_buffers()
    gen = torch.Generator().manual_seed(1)
    for name, fl in {"all": torch.ones(G, dtype=torch.uint8), "half": (torch.rand(G, generator=gen) < 0.5).to(torch.uint8),
                     "sparse": (torch.rand(G, generator=gen) < 0.03).to(torch.uint8), "none": torch.zeros(G, dtype=torch.uint8)}.items():
        flg = fl.cuda()
        logits = torch.full((G, A), 77.0, device="cuda")
        _, v = ev(None, logits_out=logits, packed=(ring, leaf, None, None, flg.data_ptr()), n_boards=G)
        torch.cuda.synchronize()
        on = flg.bool()
        assert torch.equal(logits[on], pd[on]), name
        assert torch.equal(v.view(-1)[on], vd[on]), name
    eng.close()


def test_f16x2_tower_flags_an_activation_it_cannot_represent():
    from alphazero_amd.engine import EngineError
    from alphazero_amd.fused import FusedTowerEvaluator
    from alphazero_amd.nets import GomokuNet, QuickNet

    torch.manual_seed(1)
    net = _randomised(GomokuNet(9, 64, 1)).eval().cuda()
    with torch.no_grad():
        net._shared_layers[1].weight.mul_(3.0e4)  # blow the stem's outputs far past 4,094
    ev = FusedTowerEvaluator(net, "f16x2")
    x = torch.randint(-1, 2, (8, 9, 9, 9), device="cuda").float()
    ev(x.contiguous())
    assert ev.overflowed()
    with pytest.raises(EngineError):
        FusedTowerEvaluator(GomokuNet(9, 128, 1).eval().cuda(), "f16x2")
    with pytest.raises(EngineError):
        FusedTowerEvaluator(QuickNet(9, 9, 64, 1, 2).eval().cuda(), "f16x2")
