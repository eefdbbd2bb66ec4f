"""GPU test of the fused conv-tower evaluator (csrc/az_tower.hip) against the plain PyTorch fp32 forward of
the same network.  Floating point: bf16 operands, f32 accumulation, bf16 activations between layers, so
the bar is a bf16-level tolerance, stated below; the reference's model contract (nn_models/base.py:17-19)
fixes shapes and ranges only."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _randomised(net):
    g = torch.Generator().manual_seed(1)
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.2)
            m.running_var.copy_(torch.rand(m.num_features, generator=g) + 0.5)
            m.weight.data.copy_(torch.rand(m.num_features, generator=g) + 0.5)
            m.bias.data.copy_(torch.randn(m.num_features, generator=g) * 0.2)
    return net


@pytest.mark.parametrize("board,blocks,batch,ch", [(15, 2, 64, 64), (15, 2, 1000, 64), (15, 3, 33, 64), (9, 2, 100, 64), (9, 1, 7, 64),
                                                   (15, 0, 5, 64), (15, 10, 600, 64), (9, 6, 50, 64),
                                                   (15, 2, 300, 128), (15, 10, 520, 128), (9, 3, 100, 128), (15, 0, 9, 128)])
def test_fused_tower_matches_fp32_forward(board, blocks, batch, ch):
    from alphazero_amd.fused import FusedTowerEvaluator
    from alphazero_amd.nets import GomokuNet

    torch.manual_seed(0)
    net = _randomised(GomokuNet(board, ch, blocks)).eval().cuda()
    assert FusedTowerEvaluator.supports(net)
    ev = FusedTowerEvaluator(net)
    x = torch.randint(-1, 2, (batch, 9, board, board), device="cuda").float()
    x[:, 8] = torch.where(torch.rand(batch, 1, 1, device="cuda") < 0.5, -1.0, 1.0)  # player plane
    with torch.inference_mode():
        ref_p, ref_v = net(x)
    p, v = ev(x.bfloat16().contiguous())
    assert p.shape == ref_p.shape and v.shape == ref_v.shape and p.dtype == torch.float32
    scale = float(ref_p.abs().max())
    # bf16 has 8 significant bits; 2 + 2*blocks layers of rounding -> a few 1e-2 of the logit scale at worst
    assert float((p - ref_p).abs().max()) <= 0.03 * scale + 0.02, float((p - ref_p).abs().max())
    assert float((v - ref_v).abs().max()) <= 0.03
    assert float((p - ref_p).abs().mean()) <= 0.004 * scale + 0.002
    # and it must beat / match the plain bf16 PyTorch forward's own error
    netb = GomokuNet(board, ch, blocks)
    netb.load_state_dict(net.state_dict())
    netb = netb.eval().cuda().bfloat16()
    with torch.inference_mode():
        bp, _ = netb(x.bfloat16())
    assert float((p - ref_p).abs().mean()) <= 1.5 * float((bp.float() - ref_p).abs().mean()) + 1e-4


@pytest.mark.parametrize("board,blocks,batch,ch", [(15, 2, 64, 64), (15, 2, 1100, 64), (15, 3, 33, 64), (9, 2, 100, 64), (9, 1, 7, 64),
                                                   (15, 0, 5, 64), (9, 0, 4, 64), (15, 10, 600, 64), (9, 6, 50, 64),
                                                   (15, 2, 300, 128), (15, 10, 270, 128), (15, 0, 9, 128), (9, 3, 700, 128), (9, 0, 5, 128),
                                                   (9, 2, 4096, 64), (9, 21, 40, 64), (15, 21, 5, 64)])  # (42 conv layers: the 32-cell-tile kernel)
def test_f32_tower_matches_fp32_forward(board, blocks, batch, ch):
    """az_tower_f32.hip: f32 operands, f32 accumulation -- the reference's own arithmetic (nn_models/base.py:17-19).  Bar
    (north_star: priors within 1e-5 relative): logits within 2e-5 of the logit scale of PyTorch's f32 forward (which differs
    from any other f32 implementation by summation order and by BatchNorm applied after the conv rather than folded)."""
    from alphazero_amd.fused import FusedTowerEvaluator
    from alphazero_amd.nets import GomokuNet

    torch.manual_seed(0)
    net = _randomised(GomokuNet(board, ch, blocks)).eval().cuda()
    ev = FusedTowerEvaluator(net, "f32")
    x = torch.randint(-1, 2, (batch, 9, board, board), device="cuda").float()
    x[:, 8] = torch.where(torch.rand(batch, 1, 1, device="cuda") < 0.5, -1.0, 1.0)
    torch.backends.cudnn.allow_tf32 = False
    with torch.inference_mode():
        ref_p, ref_v = net(x)
        ref64_p, ref64_v = net.double()(x.double())
    net.float()
    p, v = ev(x.contiguous())
    assert p.shape == ref_p.shape and v.shape == ref_v.shape and p.dtype == torch.float32
    scale = float(ref_p.abs().max())
    err = float((p - ref64_p).abs().max())
    err_torch = float((ref_p - ref64_p).abs().max())
    assert err <= 2e-5 * scale + 2e-6, (err, scale)
    assert float((v - ref64_v).abs().max()) <= 2e-5
    assert err <= 4 * err_torch + 2e-6  # no worse than PyTorch's own f32 forward against an f64 reference
    # softmax priors within 1e-5 relative of the f64 network's
    pr, pr64 = torch.softmax(p.double(), 1), torch.softmax(ref64_p, 1)
    assert float(((pr - pr64).abs() / pr64).max()) <= 1e-5 * max(1.0, scale)


def test_f32_tower_packed_leaf_batch_equals_dense_planes():
    from alphazero_amd import Engine
    from alphazero_amd.fused import FusedTowerEvaluator
    from alphazero_amd.nets import GomokuNet

    torch.manual_seed(0)
    for board, G, ch in ((15, 1300, 64), (9, 700, 64), (15, 600, 128)):
        sims = 10
        ev = FusedTowerEvaluator(_randomised(GomokuNet(board, ch, 2)).eval().cuda(), "f32")
        eng = Engine(2, G, sims, board_size=board, nn_dtype=torch.float32, device="cuda:0")
        nn = eng.new_nn_input()
        A = board * board
        logits, values = eng.new(G, A), eng.new(G)
        it = torch.ones(G, device="cuda")
        for _ in range(4):
            eng.begin_search()
            for _ in range(sims):
                eng.select_gather(nn)
                eng.eval_hash(nn, logits, values, 6.0)
                eng.expand_backup(logits, values, None)
            eng.play_moves(it, torch.empty(G, A, device="cuda").exponential_(1))
        eng.begin_search()
        for _ in range(5):
            eng.select_gather(nn)
            eng.eval_hash(nn, logits, values, 6.0)
            eng.expand_backup(logits, values, None)
        eng.select_gather(nn)
        p1, v1 = ev(nn)
        p1, v1 = p1.clone(), v1.clone()
        ring, leaf, _, _, flg = eng.leaf_buffers()
        p2, v2 = ev(None, packed=(ring, leaf), n_boards=G)
        assert torch.equal(p1, p2) and torch.equal(v1, v2)
        for t in ev._buf[G]:
            t.fill_(-1.0)
        p3, v3 = ev(None, packed=(ring, leaf, None, None, flg), n_boards=G)
        touched = ev._buf[G][0][:, 0] >= 0
        assert int(touched.sum()) > 0
        assert torch.equal(p3[touched], p1[touched]) and torch.equal(v3[touched], v1[touched])
        eng.check()


def test_fused_tower_rejects_other_nets():
    from alphazero_amd.fused import FusedTowerEvaluator
    from alphazero_amd.nets import GomokuNet, TicTacToeNet

    assert not FusedTowerEvaluator.supports(TicTacToeNet())
    assert not FusedTowerEvaluator.supports(GomokuNet(15, 96, 2))
    assert FusedTowerEvaluator.supports(GomokuNet(15, 128, 2))


@pytest.mark.parametrize("board,ch", [(15, 64), (9, 64), (15, 128)])
def test_packed_leaf_batch_equals_dense_planes(board, ch):
    """The tower fed from the engine's packed leaf batch (bitboards) == fed from the dense bf16 planes."""
    from alphazero_amd import Engine
    from alphazero_amd.fused import FusedTowerEvaluator
    from alphazero_amd.nets import GomokuNet

    torch.manual_seed(0)
    G, sims = (1500 if (board, ch) == (15, 64) else 200), 12  # 1500: several tickets per workgroup (512 workgroups), flags included
    net = _randomised(GomokuNet(board, ch, 2)).eval().cuda()
    ev = FusedTowerEvaluator(net)
    eng = Engine(2, G, sims, board_size=board, nn_dtype=torch.bfloat16, device="cuda:0")
    nn = eng.new_nn_input()
    A = board * board
    logits, values = eng.new(G, A), eng.new(G)
    # play a few moves so that histories are non-trivial, then compare on a mid-search leaf batch
    it = torch.ones(G, device="cuda")
    for _ in range(5):
        eng.begin_search()
        for _ in range(sims):
            eng.select_gather(nn)
            eng.eval_hash(nn, logits, values, 6.0)
            eng.expand_backup(logits, values, None)
        eng.play_moves(it, torch.empty(G, A, device="cuda").exponential_(1))
    eng.begin_search()
    for _ in range(6):
        eng.select_gather(nn)
        eng.eval_hash(nn, logits, values, 6.0)
        eng.expand_backup(logits, values, None)
    before = eng.read_stats()
    eng.select_gather(nn)
    n_eval = eng.read_stats()["evals"] - before["evals"]
    p1, v1 = ev(nn)
    p1, v1 = p1.clone(), v1.clone()
    eng.compact_leaves()
    ring, leaf, lst, cnt, flg = eng.leaf_buffers()
    p2, v2 = ev(None, packed=(ring, leaf), n_boards=G)          # packed, every slot
    eng.check()
    assert torch.equal(p1, p2) and torch.equal(v1, v2)
    for t in ev._buf[G]:
        t.fill_(-1.0)  # relu outputs are >= 0, so -1 marks rows the kernel did not write
    p3, v3 = ev(None, packed=(ring, leaf, lst, cnt), n_boards=G)  # packed + compact list: only leaves that need the net
    pfeat = ev._buf[G][0]
    touched = pfeat[:, 0] >= 0
    assert 0 < n_eval <= G and int(touched.sum()) == n_eval
    assert torch.equal(p3[touched], p1[touched]) and torch.equal(v3[touched], v1[touched])
    for t in ev._buf[G]:
        t.fill_(-1.0)
    p4, v4 = ev(None, packed=(ring, leaf, None, None, flg), n_boards=G)  # packed + flags: the same set, no compaction launch
    touched4 = ev._buf[G][0][:, 0] >= 0
    assert torch.equal(touched4, touched)
    assert torch.equal(p4[touched], p1[touched]) and torch.equal(v4[touched], v1[touched])


@pytest.mark.parametrize("precision,board,G", [("bf16", 15, 1400), ("bf16", 9, 3400), ("f32", 15, 1400), ("f32", 9, 3400)])
def test_zero_block_tower_packed_flags_many_tickets(precision, board, G):
    """blocks = 0 (stem + heads only: no conv layer, hence no stage barrier, between the fetch of the next group's
    ticket / slots / input and their use) in packed + flags mode with more groups than workgroups, so every workgroup
    draws several tickets: == the dense-plane route row by row."""
    from alphazero_amd import Engine
    from alphazero_amd.fused import FusedTowerEvaluator
    from alphazero_amd.nets import GomokuNet

    torch.manual_seed(1)
    dt = torch.bfloat16 if precision == "bf16" else torch.float32
    ev = FusedTowerEvaluator(_randomised(GomokuNet(board, 64, 0)).eval().cuda(), precision)
    sims = 6
    eng = Engine(2, G, sims, board_size=board, nn_dtype=dt, device="cuda:0")
    nn = eng.new_nn_input()
    A = board * board
    logits, values = eng.new(G, A), eng.new(G)
    it = torch.ones(G, device="cuda")
    for _ in range(3):
        eng.begin_search()
        for _ in range(sims):
            eng.select_gather(nn)
            eng.eval_hash(nn, logits, values, 6.0)
            eng.expand_backup(logits, values, None)
        eng.play_moves(it, torch.empty(G, A, device="cuda").exponential_(1))
    eng.begin_search()
    for _ in range(4):
        eng.select_gather(nn)
        eng.eval_hash(nn, logits, values, 6.0)
        eng.expand_backup(logits, values, None)
    eng.select_gather(nn)
    p1, v1 = ev(nn)
    p1, v1 = p1.clone(), v1.clone()
    ring, leaf, _, _, flg = eng.leaf_buffers()
    for _ in range(3):  # several launches: a stale ticket or slot would show up as a wrong or missing row
        for t in ev._buf[G]:
            t.fill_(-1.0)
        p2, v2 = ev(None, packed=(ring, leaf, None, None, flg), n_boards=G)
        touched = ev._buf[G][0][:, 0] >= 0
        assert int(touched.sum()) == G  # no terminal leaves this early: every slot is flagged
        assert torch.equal(p2, p1) and torch.equal(v2, v1)
    eng.check()


@pytest.mark.parametrize("n,ch,blocks,G", [(15, 64, 2, 2100), (15, 64, 0, 2100), (15, 128, 1, 2100), (9, 64, 2, 5300), (9, 64, 0, 5300),
                                           (9, 128, 1, 1300)])
def test_f32_tower_lookahead_under_random_flags(n, ch, blocks, G):
    """The f32 tower (16-cell-tile kernel, 15x15 and 9x9) hands boards out by ticket and looks the next board up under the running one (ticket, flag byte and
    packed input by LDS-DMA into a second buffer).  Synthetic packed batches with every kind of flag pattern -- dense, sparse,
    long runs of terminal leaves, only the last slot on, nothing on -- must give, on the flagged rows, exactly the features
    of the dense route, and leave the other rows alone."""
    from alphazero_amd.fused import FusedTowerEvaluator
    from alphazero_amd.nets import GomokuNet

    torch.manual_seed(11)
    net = _randomised(GomokuNet(n, ch, blocks)).eval().cuda()
    ev = FusedTowerEvaluator(net, "f32")
    gen = torch.Generator().manual_seed(5)
    stones = torch.randint(-1, 2, (G, 8, n * n), generator=gen)          # 8 history states per slot
    player = torch.randint(0, 2, (G,), generator=gen) * 2 - 1
    # packed form: per state WC = ceil(n*n / 32) words of player-1 bits then WC words of player-2 bits; leaf record .w = player to move
    wc = (n * n + 31) // 32
    bits = torch.zeros(G, 8, 2, 32 * wc, dtype=torch.int64)
    bits[:, :, 0, :n * n] = (stones == 1).long()
    bits[:, :, 1, :n * n] = (stones == -1).long()
    w = (bits.view(G, 8, 2, wc, 32) << torch.arange(32)).sum(-1)       # (G, 8, 2, WC) 32-bit words
    ring = w.view(G, 16 * wc).to(torch.int64).cuda()
    ring = (ring & 0xFFFFFFFF).to(torch.int64)
    ring32 = torch.where(ring >= 2 ** 31, ring - 2 ** 32, ring).to(torch.int32).contiguous()
    leaf = torch.zeros(G, 4, dtype=torch.int32)
    leaf[:, 3] = player.int()
    leaf = leaf.cuda()
    x = torch.cat([stones.view(G, 8, n, n).float(), player.view(G, 1, 1, 1).float().expand(G, 1, n, n)], 1).cuda().contiguous()
    pd, vd = ev(x)
    pd, vd = pd.clone(), vd.clone().view(-1)
    patterns = {
        "all": torch.ones(G, dtype=torch.uint8),
        "half": (torch.rand(G, generator=gen) < 0.5).to(torch.uint8),
        "sparse": (torch.rand(G, generator=gen) < 0.03).to(torch.uint8),
        "runs": ((torch.arange(G) // 97) % 2).to(torch.uint8),
        "last_only": torch.zeros(G, dtype=torch.uint8).index_fill_(0, torch.tensor([G - 1]), 1),
        "first_only": torch.zeros(G, dtype=torch.uint8).index_fill_(0, torch.tensor([0]), 1),
        "none": torch.zeros(G, dtype=torch.uint8),
    }
    for name, fl in patterns.items():
        flg = fl.cuda()
        logits = torch.full((G, n * n), 77.0, device="cuda")
        ev._buf[G][2].fill_(55.0)
        _, v = ev(None, logits_out=logits, packed=(ring32.data_ptr(), leaf.data_ptr(), None, None, flg.data_ptr()), n_boards=G)
        torch.cuda.synchronize()
        on = flg.bool()
        assert torch.equal(logits[on], pd[on]), name
        assert torch.equal(v.view(-1)[on], vd[on]), name
        # (the heads kernel evaluates every row of the feature buffer: rows that are off keep stale FEATURES, so only the
        # flagged rows are specified; what must hold is that they equal the dense route bit for bit)


@pytest.mark.parametrize("board,ch,blocks,layers,batch", [(15, 128, 1, 5, 300), (15, 128, 4, 5, 270), (15, 64, 2, 3, 1100), (15, 64, 3, 1, 33),
                                                          (9, 64, 2, 5, 4200), (9, 64, 1, 2, 7), (15, 128, 2, 2, 5), (15, 64, 8, 5, 40),
                                                          (9, 128, 2, 3, 600)])
def test_quick_tower_matches_fp32_forward(board, ch, blocks, layers, batch):
    """The reference's QuickModel family (nn_models/quick_model.py, resnet_blocks.py:28-40; nets.QuickNet is pinned to the
    reference's own QuickModel by tests/golden/quick.npz) on the f32 tower's QuickModel instantiation + az_quick_heads_forward_f32.
    Same bar as the GomokuModel family: logits within 2e-5 of the logit scale of an f64 evaluation and no worse than 4x PyTorch's
    own f32 forward; softmax priors within 1e-5 relative."""
    from alphazero_amd.fused import FusedTowerEvaluator
    from alphazero_amd.nets import QuickNet

    torch.manual_seed(0)
    net = _randomised(QuickNet(board, 9, ch, blocks, layers)).eval().cuda()
    assert FusedTowerEvaluator.quick_shape(net) == (board, ch, blocks, layers) and FusedTowerEvaluator.supports(net)
    ev = FusedTowerEvaluator(net, "f32")
    x = torch.randint(-1, 2, (batch, 9, board, board), device="cuda").float()
    x[:, 8] = torch.where(torch.rand(batch, 1, 1, device="cuda") < 0.5, -1.0, 1.0)
    torch.backends.cudnn.allow_tf32 = False
    with torch.inference_mode():
        ref_p, ref_v = net(x)
        ref64_p, ref64_v = net.double()(x.double())
    net.float()
    p, v = ev(x.contiguous())
    assert p.shape == ref_p.shape and v.shape == ref_v.shape and p.dtype == torch.float32
    scale = float(ref_p.abs().max())
    err = float((p - ref64_p).abs().max())
    err_torch = float((ref_p - ref64_p).abs().max())
    assert err <= 2e-5 * scale + 2e-6, (err, scale)
    assert float((v - ref64_v).abs().max()) <= 2e-5
    assert err <= 4 * err_torch + 2e-6
    pr, pr64 = torch.softmax(p.double(), 1), torch.softmax(ref64_p, 1)
    assert float(((pr - pr64).abs() / pr64).max()) <= 1e-5 * max(1.0, scale)
    # deterministic per board: a second launch (other tickets, other workgroups, a stash full of the first launch's data) and a
    # permuted batch give the same rows bit for bit
    p1, v1 = p.clone(), v.clone()
    perm = torch.randperm(batch, device="cuda")
    p2, v2 = ev(x[perm].contiguous())
    assert torch.equal(p2, p1[perm]) and torch.equal(v2, v1[perm])


def test_quick_tower_rejects_what_it_does_not_implement():
    from alphazero_amd.engine import EngineError
    from alphazero_amd.fused import FusedTowerEvaluator
    from alphazero_amd.nets import QuickNet

    F = FusedTowerEvaluator
    assert F.quick_shape(QuickNet(15, 9, 96, 1, 5)) is None       # width
    assert F.quick_shape(QuickNet(3, 9, 64, 1, 5)) is None         # the reference's own tic-tac-toe instance
    assert F.quick_shape(QuickNet(15, 9, 64, 2, 5, features=32)) is None
    assert F.precision_for(QuickNet(9, 9, 128, 1, 5), torch.float32) == "f32"
    assert F.precision_for(QuickNet(9, 9, 128, 1, 5), torch.bfloat16) == "bf16"
    assert F.precision_for(QuickNet(15, 9, 64, 9, 5), torch.float32) is None     # f32: 45 conv layers
    assert F.precision_for(QuickNet(15, 9, 64, 9, 5), torch.bfloat16) == "bf16"
    assert F.precision_for(QuickNet(15, 9, 128, 4, 5), torch.float32) == "f32"
    assert F.precision_for(QuickNet(15, 9, 128, 4, 5), torch.float16) is None
    with pytest.raises(EngineError):
        F(QuickNet(15, 9, 64, 9, 5).eval().cuda(), "f32")


@pytest.mark.parametrize("board,ch,blocks,layers,batch", [(15, 128, 1, 5, 300), (15, 128, 4, 5, 270), (15, 64, 2, 3, 1100), (15, 64, 3, 1, 33),
                                                          (9, 64, 2, 5, 4200), (9, 128, 1, 2, 100), (9, 64, 1, 2, 7), (15, 64, 9, 5, 20)])
def test_quick_tower_bf16_matches_fp32_forward(board, ch, blocks, layers, batch):
    """QuickModel family on the bf16 tower's QuickModel instantiation + az_quick_heads_forward (throughput mode): same bf16-level
    bar as the GomokuModel family against the fp32 PyTorch forward, and no worse than PyTorch's own bf16 forward."""
    from alphazero_amd.fused import FusedTowerEvaluator
    from alphazero_amd.nets import QuickNet

    torch.manual_seed(0)
    net = _randomised(QuickNet(board, 9, ch, blocks, layers)).eval().cuda()
    ev = FusedTowerEvaluator(net, "bf16")
    x = torch.randint(-1, 2, (batch, 9, board, board), device="cuda").float()
    x[:, 8] = torch.where(torch.rand(batch, 1, 1, device="cuda") < 0.5, -1.0, 1.0)
    with torch.inference_mode():
        ref_p, ref_v = net(x)
    p, v = ev(x.bfloat16().contiguous())
    assert p.shape == ref_p.shape and v.shape == ref_v.shape and p.dtype == torch.float32
    netb = QuickNet(board, 9, ch, blocks, layers)
    netb.load_state_dict(net.state_dict())
    netb = netb.eval().cuda().bfloat16()
    with torch.inference_mode():
        bp, bv = netb(x.bfloat16())
    scale = float(ref_p.abs().max())
    depth = 1 + blocks * layers
    err, err_torch = float((p - ref_p).abs().mean()), float((bp.float() - ref_p).abs().mean())
    assert err <= 1.5 * err_torch + 1e-4, (err, err_torch)
    assert float((p - ref_p).abs().max()) <= (0.03 + 0.004 * depth) * scale + 0.02, (float((p - ref_p).abs().max()), scale)
    assert float((v - ref_v).abs().max()) <= max(0.03, 2.0 * float((bv.float() - ref_v).abs().max()))
    p1, v1 = p.clone(), v.clone()
    perm = torch.randperm(batch, device="cuda")
    p2, v2 = ev(x[perm].bfloat16().contiguous())
    assert torch.equal(p2, p1[perm]) and torch.equal(v2, v1[perm])


@pytest.mark.parametrize("precision", ["f32", "bf16"])
@pytest.mark.parametrize("n,ch,blocks,layers,G", [(15, 128, 1, 3, 1100), (15, 64, 2, 2, 2100), (9, 64, 1, 5, 5300)])
def test_quick_tower_packed_flags_equals_dense(n, ch, blocks, layers, G, precision):
    """QuickModel instantiation of the f32 tower on the engine's packed leaf batch with every kind of flag pattern (several
    tickets per workgroup, look-ahead of the next board under the running one) == the dense route, row by row"""
    from alphazero_amd.fused import FusedTowerEvaluator
    from alphazero_amd.nets import QuickNet

    torch.manual_seed(12)
    ev = FusedTowerEvaluator(_randomised(QuickNet(n, 9, ch, blocks, layers)).eval().cuda(), precision)
    gen = torch.Generator().manual_seed(6)
    stones = torch.randint(-1, 2, (G, 8, n * n), generator=gen)
    player = torch.randint(0, 2, (G,), generator=gen) * 2 - 1
    wc = (n * n + 31) // 32
    bits = torch.zeros(G, 8, 2, 32 * wc, dtype=torch.int64)
    bits[:, :, 0, :n * n] = (stones == 1).long()
    bits[:, :, 1, :n * n] = (stones == -1).long()
    w = (bits.view(G, 8, 2, wc, 32) << torch.arange(32)).sum(-1)
    ring = (w.view(G, 16 * wc).to(torch.int64).cuda() & 0xFFFFFFFF).to(torch.int64)
    ring32 = torch.where(ring >= 2 ** 31, ring - 2 ** 32, ring).to(torch.int32).contiguous()
    leaf = torch.zeros(G, 4, dtype=torch.int32)
    leaf[:, 3] = player.int()
    leaf = leaf.cuda()
    x = torch.cat([stones.view(G, 8, n, n).float(), player.view(G, 1, 1, 1).float().expand(G, 1, n, n)], 1).cuda().to(ev.in_dtype).contiguous()
    pd, vd = ev(x)
    pd, vd = pd.clone(), vd.clone().view(-1)
    patterns = {
        "all": torch.ones(G, dtype=torch.uint8),
        "half": (torch.rand(G, generator=gen) < 0.5).to(torch.uint8),
        "sparse": (torch.rand(G, generator=gen) < 0.03).to(torch.uint8),
        "runs": ((torch.arange(G) // 97) % 2).to(torch.uint8),
        "last_only": torch.zeros(G, dtype=torch.uint8).index_fill_(0, torch.tensor([G - 1]), 1),
        "none": torch.zeros(G, dtype=torch.uint8),
    }
    for name, fl in patterns.items():
        flg = fl.cuda()
        logits = torch.full((G, n * n), 77.0, device="cuda")
        _, v = ev(None, logits_out=logits, packed=(ring32.data_ptr(), leaf.data_ptr(), None, None, flg.data_ptr()), n_boards=G)
        torch.cuda.synchronize()
        on = flg.bool()
        assert torch.equal(logits[on], pd[on]), name
        assert torch.equal(v.view(-1)[on], vd[on]), name
