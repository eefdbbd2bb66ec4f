"""az_tower_w.hip ("f32w"): the 64-channel GomokuModel-family tower (15x15, and 9x9 with two boards per workgroup) in f32 arithmetic with the 3x3 convolutions computed
by the Winograd transform F(2x2, 3x3) (2.1x fewer MFMAs than the direct convolution of az_tower_f32.hip).  Plain f32 operands,
products and sums, as the reference evaluates its network (nn_models/base.py:17-19); another summation order than the direct
tower, so held -- like it -- to the f32 bars of tests/test_gpu_tower.py against an f64 evaluation of the same network: logits
within 2e-5 of the logit scale, values within 2e-5, softmax priors within 1e-5 relative; deterministic per board."""
import numpy as np
import pytest
import torch

from .test_gpu_tower import _randomised

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("board,blocks,batch", [(15, 2, 64), (15, 2, 1100), (15, 3, 33), (15, 0, 5), (15, 1, 1), (15, 10, 600), (15, 2, 16384),
                                                (9, 2, 100), (9, 1, 7), (9, 0, 4), (9, 2, 1), (9, 6, 51), (9, 2, 4096)])
def test_winograd_tower_meets_the_f32_towers_accuracy_bars(board, blocks, batch):
    from alphazero_amd.fused import FusedTowerEvaluator
    from alphazero_amd.nets import GomokuNet

    torch.manual_seed(0)
    net = _randomised(GomokuNet(board, 64, blocks)).eval().cuda()
    ev = FusedTowerEvaluator(net, "f32w")
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
    scale = float(ref64_p.abs().max())
    err = float((p - ref64_p).abs().max())
    err32 = float((p32 - ref64_p).abs().max())
    print(f"f32w max |logit error| {err:.3e} (direct f32 tower {err32:.3e}) at logit scale {scale:.3f}")
    assert err <= 2e-5 * scale + 2e-6, (err, scale)
    assert float((v - ref64_v).abs().max()) <= 2e-5
    pr, pr64 = torch.softmax(p.double(), 1), torch.softmax(ref64_p, 1)
    assert float(((pr - pr64).abs() / pr64).max()) <= 1e-5 * max(1.0, scale)
    # deterministic per board: a second launch and a permuted batch give the same rows bit for bit
    perm = torch.randperm(batch, device="cuda")
    p2, v2 = ev(x[perm].contiguous())
    assert torch.equal(p2, p[perm]) and torch.equal(v2, v[perm])


@pytest.mark.parametrize("n,G", [(15, 2100), (9, 5301)])
def test_winograd_tower_packed_flags_equals_dense(n, G):
    """the engine's packed leaf batch + needs-the-network flags == the dense planes of the same positions, bit for bit, for every
    flag pattern (9x9: an odd number of boards, so the last ticket is half a pair; pairs with one board flagged off)"""
    from alphazero_amd import Engine
    from alphazero_amd.fused import FusedTowerEvaluator
    from alphazero_amd.nets import GomokuNet

    blocks = 2
    torch.manual_seed(3)
    net = _randomised(GomokuNet(n, 64, blocks)).eval().cuda()
    ev = FusedTowerEvaluator(net, "f32w")
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


def test_winograd_tower_refuses_what_it_does_not_cover():
    from alphazero_amd.engine import EngineError
    from alphazero_amd.fused import FusedTowerEvaluator
    from alphazero_amd.nets import GomokuNet, QuickNet

    for net in (GomokuNet(9, 128, 1), GomokuNet(15, 128, 1), QuickNet(15, 15, 64, 1, 2)):
        with pytest.raises(EngineError):
            FusedTowerEvaluator(net.eval().cuda(), "f32w")


def test_winograd_entry_point_refuses_misaligned_pointers_and_oversized_batches():
    """the C-ABI's host-side checks of what the kernel's addressing assumes (no launch happens): flag bytes on a 4-byte boundary
    (the look-ahead reads the aligned word that holds them), 16-byte aligned leaf records and weights, at most 2^22 boards"""
    from alphazero_amd import Engine
    from alphazero_amd.fused import FusedTowerEvaluator
    from alphazero_amd.nets import GomokuNet

    G = 64
    ev = FusedTowerEvaluator(_randomised(GomokuNet(15, 64, 1)).eval().cuda(), "f32w")
    eng = Engine(2, G, 8, board_size=15, nn_dtype=torch.float32, device="cuda:0")
    nn = eng.new_nn_input()
    eng.begin_search()
    eng.select_gather(nn)
    ring, leaf, _, _, flags = eng.leaf_buffers()
    L, fs = ev.lib, ev.feat_stride
    feat = torch.zeros(G, fs, device="cuda")
    st = torch.cuda.current_stream().cuda_stream

    def call(n=G, ring=ring, leaf=leaf, flags=flags, u0=ev.w0.data_ptr(), feat=feat.data_ptr()):
        return L.az_tower_forward_f32w(15, 64, n, ev.n_layers, None, ring, leaf, flags, u0, ev.wl.data_ptr(), ev.bias.data_ptr(),
                                       ev.headw.data_ptr(), ev.headb.data_ptr(), feat, fs, st)

    assert call() == 0
    torch.cuda.synchronize()
    for bad in (dict(flags=flags + 1), dict(ring=ring + 2), dict(leaf=leaf + 8), dict(u0=ev.w0.data_ptr() + 4), dict(feat=feat.data_ptr() + 2),
                dict(n=(1 << 22) + 1), dict(n=0)):
        assert call(**bad) == -1, bad  # AZ_EINVAL
        assert L.az_tower_last_error()
    eng.close()

