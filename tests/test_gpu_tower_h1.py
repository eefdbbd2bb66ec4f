"""az_tower_h1.hip (precisions "bf16r" and "f16"): the 15x15 / 64-channel GomokuModel-family tower with one board row per cell
tile.  Floating point -- a throughput-mode evaluator, not the reference's f32 arithmetic (nn_models/base.py:17-19): "bf16r" is
held to the bf16 tower's own bars (tests/test_gpu_tower.py: max |dlogit| <= 3 % of the logit scale + 0.02, mean <= 0.4 % + 0.002
against the fp32 forward); "f16" (11 significant bits instead of 8, exact power-of-two scalings) to an eighth of them.  Both are
deterministic per board, and the engine's packed leaf batch + flags gives the dense planes' rows bit for bit."""
import numpy as np
import pytest
import torch

from .test_gpu_tower import _randomised

pytestmark = pytest.mark.gpu

BARS = {"bf16r": (0.03, 0.02, 0.004, 0.002, 0.03), "f16": (0.004, 0.0025, 0.0005, 0.00025, 0.004)}


@pytest.mark.parametrize("precision", ["bf16r", "f16"])
@pytest.mark.parametrize("blocks,batch", [(2, 64), (2, 1100), (3, 33), (0, 5), (10, 600), (1, 4096)])
def test_row_tile_tower_matches_fp32_forward(precision, blocks, batch):
    from alphazero_amd.fused import FusedTowerEvaluator
    from alphazero_amd.nets import GomokuNet

    torch.manual_seed(0)
    net = _randomised(GomokuNet(15, 64, blocks)).eval().cuda()
    ev = FusedTowerEvaluator(net, precision)
    x = torch.randint(-1, 2, (batch, 9, 15, 15), device="cuda").float()
    x[:, 8] = torch.where(torch.rand(batch, 1, 1, device="cuda") < 0.5, -1.0, 1.0)
    with torch.inference_mode():
        ref_p, ref_v = net(x)
    p, v = ev(x.contiguous())
    p, v = p.clone(), v.clone()
    assert not ev.overflowed()
    scale = float(ref_p.abs().max())
    a_max, b_max, a_mean, b_mean, v_bar = BARS[precision]
    err, mean = float((p - ref_p).abs().max()), float((p - ref_p).abs().mean())
    print(f"{precision}: max |dlogit| {err:.3e}, mean {mean:.3e} at logit scale {scale:.3f}")
    assert err <= a_max * scale + b_max, (err, scale)
    assert mean <= a_mean * scale + b_mean, (mean, scale)
    assert float((v - ref_v).abs().max()) <= v_bar
    # deterministic per board: a permuted batch gives the same rows bit for bit
    perm = torch.randperm(batch, device="cuda")
    p2, v2 = ev(x[perm].contiguous())
    assert torch.equal(p2, p[perm]) and torch.equal(v2, v[perm])


@pytest.mark.parametrize("precision", ["f32", "f32w", "bf16", "bf16r", "f16", "f16x2"])
def test_every_tower_at_the_bench_batch_is_a_function_of_the_board_alone(precision):
    """BASELINE's full size (16,384 boards per launch, the reference GomokuModel): size-independent properties of every
    hand-written tower -- a board's row does not depend on where in the batch it stands or on what else is in the batch
    (a reversed batch and a 300-board slice give the same rows bit for bit), and stays within its precision's bar of the fp32
    forward on a sample of the rows"""
    from alphazero_amd.fused import FusedTowerEvaluator
    from alphazero_amd.nets import GomokuNet

    torch.manual_seed(5)
    G = 16384
    net = _randomised(GomokuNet(15, 64, 2)).eval().cuda()
    ev = FusedTowerEvaluator(net, precision)
    dt = ev.in_dtype
    x = torch.randint(-1, 2, (G, 9, 15, 15), device="cuda").float()
    x[:, 8] = torch.where(torch.rand(G, 1, 1, device="cuda") < 0.5, -1.0, 1.0)
    p, v = ev(x.to(dt).contiguous())
    p, v = p.clone(), v.clone()
    assert not ev.overflowed() and bool(torch.isfinite(p).all()) and bool((v.abs() <= 1).all())
    p2, v2 = ev(x.flip(0).to(dt).contiguous())
    assert torch.equal(p2.flip(0), p) and torch.equal(v2.flip(0), v)
    lo = 7001
    p3, v3 = ev(x[lo:lo + 300].to(dt).contiguous())
    assert torch.equal(p3, p[lo:lo + 300]) and torch.equal(v3, v[lo:lo + 300])
    with torch.inference_mode():
        ref_p, ref_v = net(x[:512])
    scale = float(ref_p.abs().max())
    bar = {"f32": 2e-4, "f32w": 2e-4, "f16x2": 2e-4, "f16": 0.004, "bf16": 0.03, "bf16r": 0.03}[precision]
    assert float((p[:512] - ref_p).abs().max()) <= bar * scale + bar
    assert float((v[:512] - ref_v).abs().max()) <= bar + 1e-4


def test_f16_is_several_times_closer_to_f32_than_bf16():
    from alphazero_amd.fused import FusedTowerEvaluator
    from alphazero_amd.nets import GomokuNet

    torch.manual_seed(1)
    net = _randomised(GomokuNet(15, 64, 2)).eval().cuda()
    x = torch.randint(-1, 2, (512, 9, 15, 15), device="cuda").float().contiguous()
    with torch.inference_mode():
        ref_p, _ = net.double()(x.double())
    net.float()
    errs = {}
    for prec in ("f16", "bf16r"):
        p, _ = FusedTowerEvaluator(net, prec)(x)
        errs[prec] = float((p.double() - ref_p).abs().mean())
    assert errs["f16"] * 4.0 <= errs["bf16r"], errs


def test_f16_layers_with_tiny_weights_take_the_scaled_path():
    """a layer whose folded weights are all below 2^-6 is packed x a power of two (fused.py) and its epilogue, and the residual
    selector of the block it closes, undo that scale: same bars, and the scales really are not 1"""
    from alphazero_amd.fused import FusedTowerEvaluator
    from alphazero_amd.nets import GomokuNet

    torch.manual_seed(4)
    net = _randomised(GomokuNet(15, 64, 3)).eval().cuda()
    with torch.no_grad():
        blocks = list(net._shared_layers)[3:]
        blocks[0]._conv2.weight.mul_(2.0 ** -9)   # the closing conv of a block: its scale rides on the residual selector
        blocks[1]._conv1.weight.mul_(2.0 ** -8)
        blocks[2]._conv1.weight.mul_(2.0 ** 3)
    ev = FusedTowerEvaluator(net, "f16")
    assert sorted(set(ev.sw.cpu().tolist())) != [1.0]
    x = torch.randint(-1, 2, (300, 9, 15, 15), device="cuda").float().contiguous()
    with torch.inference_mode():
        ref_p, ref_v = net(x)
    p, v = ev(x)
    assert not ev.overflowed()
    scale = float(ref_p.abs().max())
    a_max, b_max, a_mean, b_mean, v_bar = BARS["f16"]
    assert float((p - ref_p).abs().max()) <= a_max * scale + b_max
    assert float((p - ref_p).abs().mean()) <= a_mean * scale + b_mean
    assert float((v - ref_v).abs().max()) <= v_bar


@pytest.mark.parametrize("precision", ["bf16r", "f16"])
@pytest.mark.parametrize("blocks,G", [(2, 2100), (0, 700)])
def test_row_tile_tower_packed_flags_equals_dense(precision, blocks, G):
    """the engine's packed leaf batch + needs-the-network flags == the dense planes of the same positions, bit for bit, for every
    flag pattern (rows that are off are not compared: the heads kernel computes them from stale feature rows); G > 512 workgroups x 2: every workgroup draws several tickets, so
    the one-board-ahead hand-over (flag skip, parked packed input) is exercised"""
    from alphazero_amd import Engine
    from alphazero_amd.fused import FusedTowerEvaluator
    from alphazero_amd.nets import GomokuNet

    torch.manual_seed(3)
    net = _randomised(GomokuNet(15, 64, blocks)).eval().cuda()
    ev = FusedTowerEvaluator(net, precision)
    eng = Engine(2, G, 8, board_size=15, nn_dtype=torch.float32, device="cuda:0")
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
                     "sparse": (torch.rand(G, generator=gen) < 0.03).to(torch.uint8), "none": torch.zeros(G, dtype=torch.uint8),
                     "runs": ((torch.arange(G) // 37) % 2).to(torch.uint8)}.items():
        flg = fl.cuda()
        logits = torch.full((G, A), 77.0, device="cuda")
        _, v = ev(None, logits_out=logits, packed=(ring, leaf, None, None, flg.data_ptr()), n_boards=G)
        v = v.view(-1).clone()
        torch.cuda.synchronize()
        on = flg.bool()
        assert torch.equal(logits[on], pd[on]), name
        assert torch.equal(v[on], vd[on]), name
    eng.close()


def test_row_tile_tower_limits_and_overflow_flag():
    from alphazero_amd.engine import EngineError
    from alphazero_amd.fused import FusedTowerEvaluator
    from alphazero_amd.nets import GomokuNet, QuickNet

    torch.manual_seed(1)
    net = _randomised(GomokuNet(15, 64, 1)).eval().cuda()
    with torch.no_grad():
        net._shared_layers[1].weight.mul_(3.0e4)  # blow the stem's outputs far past 4,094
    ev = FusedTowerEvaluator(net, "f16")
    x = torch.randint(-1, 2, (8, 9, 15, 15), device="cuda").float()
    ev(x.contiguous())
    assert ev.overflowed()
    # ... and a self-play run on such a network fails loudly instead of playing on garbage (SelfPlayRunner.check)
    from alphazero_amd import Engine
    from alphazero_amd.arena import SelfPlayRunner

    eng = Engine(2, 16, 4, board_size=15, device="cuda:0")
    run = SelfPlayRunner(eng, net, 4, fused="f16", seed=1)
    run.move_round(torch.ones(16, device="cuda"))
    with pytest.raises(EngineError, match="overflowed"):
        run.check()
    ok = SelfPlayRunner(eng, net, 4, fused="f32", seed=1)  # the f32 tower takes the same network
    eng.reset_games()
    ok.move_round(torch.ones(16, device="cuda"))
    ok.check()
    eng.close()
    for prec in ("f16", "bf16r"):
        with pytest.raises(EngineError):
            FusedTowerEvaluator(GomokuNet(9, 64, 1).eval().cuda(), prec)  # 15x15 only
        with pytest.raises(EngineError):
            FusedTowerEvaluator(GomokuNet(15, 128, 1).eval().cuda(), prec)
        with pytest.raises(EngineError):
            FusedTowerEvaluator(QuickNet(15, 9, 64, 1, 2).eval().cuda(), prec)
