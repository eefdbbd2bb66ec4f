"""CPU tests of the host side: the games mirror against the reference's rule fixtures, the HashNet
helper against the oracle's hash evaluator, network contracts, sample ring, and the N>1 weight
publish over gloo (world_size 2)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from .conftest import GOLDEN, ROOT
from .helpers import HashNet

from alphazero_amd import games, nets
from alphazero_amd.arena import reference_tao, resolve_game
from alphazero_amd.samples import HostSampleRing, games_for_rank

MIRROR = [("tictactoe", games.TicTacToeGame), ("dab", games.DotsAndBoxesGame), ("gomoku15", games.GomokuGame),
          ("gomoku9", games.Gomoku9Game)]


@pytest.mark.parametrize("name,cls", MIRROR)
def test_games_mirror_matches_reference_rules(name, cls):
    z = np.load(os.path.join(GOLDEN, f"rules_{name}.npz"))
    n_games = int(z["n_games"])
    step = 1 if name != "gomoku15" else 2  # the O(N^2) winner scan is slow in torch; half the games suffice
    for g in range(0, n_games, step):
        s, player = cls.initial_state(), 1
        assert np.array_equal(cls.legal_action_mask(s).numpy().reshape(-1), z[f"g{g}_mask0"])
        for k, a in enumerate(z[f"g{g}_actions"]):
            s, player = cls.next_state(s, int(a), player)
            assert np.array_equal(s.numpy().reshape(-1).astype(np.int8), z[f"g{g}_states"][k])
            assert player == z[f"g{g}_next_players"][k]
            assert np.array_equal(cls.legal_action_mask(s).numpy().reshape(-1), z[f"g{g}_masks"][k])
            assert bool(cls.is_terminal(s)) == bool(z[f"g{g}_terminal"][k])
            w = cls._get_winner(s)
            assert (0 if w is None else w) == z[f"g{g}_winner"][k]
    s, _ = cls.next_state(cls.initial_state(), 0, 1)
    for a in (0, -1, 10 ** 6):
        with pytest.raises(ValueError):
            cls.next_state(s, a, -1)


def test_game_instance_history_and_player_channel():
    g = games.DotsAndBoxesGame()
    assert g.get_history().shape == (24, 4, 4) and g.get_player() == 1
    g.step(0)
    h = g.get_history()
    assert torch.equal(h[21:], g.get_state()) and float(h[:21].abs().sum()) == 0
    t = games.TicTacToeGame()
    assert t.get_history().shape == (8, 3, 3)
    pc = games.BaseGame.get_player_channel(t.get_state(), -1)
    assert pc.shape == (1, 3, 3) and pc.dtype == torch.int64 and int(pc[0, 0, 0]) == -1
    assert resolve_game(games.Gomoku9Game) == (2, 9) and resolve_game(games.TicTacToeGame) == (0, 3)


def test_hashnet_equals_oracle_hash(oracle):
    import ctypes as C

    rng = np.random.default_rng(0)
    x = rng.integers(-1, 2, size=(5, 9, 9, 9)).astype(np.float32)
    net = HashNet(81, 5.0)
    p, v = net(torch.from_numpy(x))
    fn, ctx, _keep = oracle.hash_eval(5.0)
    for b in range(5):
        lg = np.zeros(81, np.float32)
        val = C.c_float()
        row = np.ascontiguousarray(x[b].reshape(-1))
        fn(ctx, row.ctypes.data_as(C.POINTER(C.c_float)), row.size, lg.ctypes.data_as(C.POINTER(C.c_float)), 81,
           C.byref(val))
        assert np.array_equal(p[b].numpy(), lg) and float(v[b, 0]) == val.value


def test_net_contracts_and_sizes():
    # parameter counts of the reference's example nets
    cases = [(nets.TicTacToeNet(), (2, 9, 3, 3), 9, 154186), (nets.DotsAndBoxesNet(), (2, 25, 4, 4), 24, 53532),
             (nets.GomokuNet(), (2, 9, 15, 15), 225, 233257), (nets.GomokuNet(9), (2, 9, 9, 9), 81, None)]
    for net, shape, A, nparam in cases:
        net.eval()
        p, v = net(torch.zeros(shape))
        assert p.shape == (2, A) and v.shape == (2, 1) and float(v.abs().max()) <= 1.0
        if nparam:
            assert sum(q.numel() for q in net.parameters()) == nparam
    keys = set(nets.GomokuNet().state_dict())
    for k in ("_shared_layers.0.weight", "_shared_layers.3._conv1.weight", "_shared_layers.4._bn2.running_var",
              "_policy_head.4.weight", "_value_head.6.bias"):
        assert k in keys  # the reference's checkpoint keys (examples/gomoku/model.py)
    assert abs(nets.GomokuNet().flops_per_eval() - 68.9e6) < 0.2e6


def test_tao_schedule_and_sharding():
    assert reference_tao(0) == 1.0 and abs(reference_tao(10) - 0.997 ** 10) < 1e-12 and reference_tao(10 ** 4) == 0.1
    owned = [list(games_for_rank(10, r, 4)) for r in range(4)]
    assert sorted(sum(owned, [])) == list(range(10)) and owned[1] == [1, 5, 9]


def test_host_sample_ring_wraps_in_put_order():
    ring = HostSampleRing(5, (2, 1, 1), 3, pin=False)
    for start in (0, 4):
        n = 4
        st = torch.arange(start, start + n).float().view(n, 1).repeat(1, 2)
        ring.put_batch(st, torch.ones(n, 3) * st[:, :1], st[:, 0])
    assert ring.size() == 5 and ring.total == 8 and ring.write_idx == 3
    assert ring.values.view(-1).tolist() == [5.0, 6.0, 7.0, 3.0, 4.0]
    assert ring.states.shape == (5, 1, 2, 1, 1) and ring.priors.shape == (5, 1, 3)


_WORKER = r"""
import sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[3])
from alphazero_amd import nets
from alphazero_amd.samples import WeightPublisher
rank, port = int(sys.argv[1]), sys.argv[2]
dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=2)
torch.manual_seed(rank)                      # different weights on each rank before the publish
m = nets.DotsAndBoxesNet()
want = nets.DotsAndBoxesNet.__new__(nets.DotsAndBoxesNet)
torch.manual_seed(0); ref = nets.DotsAndBoxesNet()
n = WeightPublisher(m, src=0).publish()
same = all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), ref.state_dict().values()))
flag = torch.tensor([1.0 if same else 0.0]); dist.all_reduce(flag, op=dist.ReduceOp.MIN)
dist.barrier(); dist.destroy_process_group()
sys.exit(0 if (flag.item() == 1.0 and n == 4 * 53532) else 3)
"""


def test_weight_publish_world_size_2_gloo(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "w.py"
    script.write_text(_WORKER)
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(port), ROOT]) for r in range(2)]
    assert [p.wait(timeout=120) for p in procs] == [0, 0]


_DP_WORKER = r"""
import sys, torch, torch.distributed as dist
rank, port, root = int(sys.argv[1]), sys.argv[2], sys.argv[3]
sys.path.insert(0, root)
from alphazero_amd.learner import DataParallelLearner, calculate_loss
from alphazero_amd.nets import TicTacToeNet
dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=2)
torch.manual_seed(0)
model = TicTacToeNet()
ref = TicTacToeNet(); ref.load_state_dict(model.state_dict())
g = torch.Generator().manual_seed(7)
batches = []
for r in range(2):  # both ranks know both batches: rank r trains on batch r, the check needs both
    s = torch.randint(-1, 2, (16, 9, 3, 3), generator=g).float()
    p = torch.softmax(torch.randn(16, 9, generator=g), 1)
    z = torch.randint(-1, 2, (16, 1), generator=g).float()
    batches.append((s, p, z))
class Pool:
    def sample(self, n, timeout=None):
        return batches[rank]
opt = torch.optim.SGD(model.parameters(), lr=0.1)
dp = DataParallelLearner(model, opt, bucket_bytes=64 << 10)  # small buckets: several collectives, launched from hooks
assert len(dp.buckets) > 1
dp.step(Pool(), 16)
# what it must equal: the same step with the mean of the two ranks' gradients, computed here without any collective
grads = []
for r in range(2):
    m = TicTacToeNet(); m.load_state_dict(ref.state_dict()); m.train()
    loss, _, _ = calculate_loss(m, *batches[r]); loss.backward()
    grads.append([q.grad.clone() for q in m.parameters()])
ok = True
for q, ga, gb, cur in zip(ref.parameters(), grads[0], grads[1], model.parameters()):
    want = q.detach() - 0.1 * (ga + gb) / 2
    ok = ok and torch.allclose(cur.detach(), want, rtol=1e-5, atol=1e-6)
# and both ranks now hold the same weights
flat = torch.cat([q.detach().reshape(-1) for q in model.parameters()])
other = [torch.empty_like(flat) for _ in range(2)]
dist.all_gather(other, flat)
ok = ok and torch.equal(other[0], other[1])
dist.destroy_process_group()
sys.exit(0 if ok else 3)
"""


def test_data_parallel_learner_world_size_2_gloo(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "dp.py"
    script.write_text(_DP_WORKER)
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(port), ROOT]) for r in range(2)]
    assert [p.wait(timeout=180) for p in procs] == [0, 0]


_LOOP_WORKER = r"""
import sys, torch, torch.distributed as dist
rank, port, root = int(sys.argv[1]), sys.argv[2], sys.argv[3]
sys.path.insert(0, root)
from alphazero_amd.loop import SelfPlayTrainLoop
from alphazero_amd.nets import TicTacToeNet
dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=2)
torch.manual_seed(rank)  # different initial weights per rank: the first publish must make them equal
G = 8


class StubEngine:  # stands in for the HIP engine: rank 0's games end every move, rank 1's every fourth move
    def __init__(self):
        self.moves = 0
    def game_info(self):
        return torch.full((G,), 5, dtype=torch.int32), None, None, None
    def emit_samples(self, slots, plies):
        n = int(torch.as_tensor(plies).sum())
        g = torch.Generator().manual_seed(100 * rank + self.moves)
        return (torch.randint(-1, 2, (n, 81), generator=g).float(), torch.softmax(torch.randn(n, 9, generator=g), 1),
                torch.randint(-1, 2, (n,), generator=g).float(), None)
    def reset_games(self, mask):
        pass
    def check(self):
        pass


class StubRunner:
    fused = None
    def __init__(self, eng):
        self.e = eng
    def move_round(self, inv_tao):
        self.e.moves += 1
        done = torch.zeros(G, dtype=torch.uint8)
        if rank == 0 or self.e.moves % 4 == 0:
            done[: 2 + rank] = 1
        return None, done


class StubRing:
    def __init__(self):
        self.s, self.p, self.z = [], [], []
    def put_games(self, s, p, z):
        self.s.append(s); self.p.append(p); self.z.append(z)
    def size(self):
        return sum(t.shape[0] for t in self.s)
    def sample(self, n, timeout=None):
        s, p, z = torch.cat(self.s), torch.cat(self.p), torch.cat(self.z)
        i = torch.randint(0, s.shape[0], (n,))
        return s[i].view(-1, 9, 3, 3), p[i], z[i].view(-1, 1)


eng = StubEngine()
loop = SelfPlayTrainLoop(TicTacToeNet(), 0, 3, G, 4, torch.device("cpu"), batch_size=16, train_steps_per_move=2, min_samples=40,
                         nn_dtype=torch.float32, publish_every=3, engine=eng, runner=StubRunner(eng), ring=StubRing())
trained_at = []
for m in range(14):
    before = loop.train_steps
    loop.move()
    if loop.train_steps != before:
        trained_at.append(m)
# rank 0 alone reaches 40 samples after 4 moves (10 per move); rank 1 gets 15 every fourth move => 45 after move 12:
# nobody may train before that, and both must have run the same number of steps (= the same number of collectives)
steps = torch.tensor([loop.train_steps, trained_at[0] if trained_at else -1])
both = [torch.empty_like(steps) for _ in range(2)]
dist.all_gather(both, steps)
ok = torch.equal(both[0], both[1]) and loop.train_steps == 2 * len(trained_at) and trained_at and trained_at[0] == 11
loop.publisher.publish()
flat = torch.cat([q.detach().reshape(-1) for q in loop.master.parameters()])
other = [torch.empty_like(flat) for _ in range(2)]
dist.all_gather(other, flat)
ok = ok and torch.equal(other[0], other[1])
dist.destroy_process_group()
sys.exit(0 if ok else 3)
"""


def test_train_loop_control_flow_world_size_2_gloo(tmp_path):
    """SelfPlayTrainLoop.move() with ranks whose rings fill at different rates: the decision to train is collective, so
    nobody trains before every rank is ready, nobody hangs, and the weights agree after a publish."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "loop.py"
    script.write_text(_LOOP_WORKER)
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(port), ROOT]) for r in range(2)]
    assert [p.wait(timeout=180) for p in procs] == [0, 0]


def test_learner_reduces_buckets_whose_hooks_did_not_fire(tmp_path):
    """A frozen-in-place (unused) parameter: its bucket is still all-reduced once per step, and the hook path runs with
    one rank under force_collectives (what a one-GPU box can exercise)."""
    import torch
    import torch.distributed as dist

    from alphazero_amd.learner import DataParallelLearner

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        class Net(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.body = torch.nn.Linear(9 * 9, 9)
                self.unused = torch.nn.Linear(4, 4)  # never takes part in forward
                self.v = torch.nn.Linear(9 * 9, 1)

            def forward(self, x):
                x = x.reshape(x.shape[0], -1)
                return self.body(x), torch.tanh(self.v(x))

        torch.manual_seed(0)
        net = Net()
        dp = DataParallelLearner(net, torch.optim.SGD(net.parameters(), lr=0.1), bucket_bytes=64, force_collectives=True)
        assert dp.collective and len(dp.buckets) >= 3
        batch = (torch.randn(8, 9, 3, 3), torch.softmax(torch.randn(8, 9), 1), torch.randn(8, 1))

        class Pool:
            def sample(self, n, timeout=None):
                return batch

        w_unused = net.unused.weight.detach().clone()
        for k in range(2):
            dp.step(Pool(), 8)
            assert dp.collectives_issued == (k + 1) * len(dp.buckets)  # every bucket exactly once per step
        assert torch.equal(net.unused.weight, w_unused)  # zero gradient, untouched
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("tag", ["tictactoe", "dab"])
def test_learner_loss_and_step_match_reference_golden(tag):
    """learner.calculate_loss / train_step == the reference's Trainer._calculate_loss / Trainer.train
    (training/nn_trainer.py:49-94) recorded by tests/golden/make_golden.py on the reference's own model class: same
    weights (our nets keep its parameter names), same batch -> same loss, same parameters after one SGD step."""
    import numpy as np
    import torch

    from alphazero_amd.learner import calculate_loss, train_step
    from alphazero_amd.nets import DotsAndBoxesNet, TicTacToeNet

    z = np.load(os.path.join(ROOT, "tests", "golden", "loss.npz"))
    net = (TicTacToeNet if tag == "tictactoe" else DotsAndBoxesNet)()
    pre = f"{tag}__w__"
    state = {k[len(pre):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(pre)}
    assert set(state) == set(net.state_dict())  # checkpoint interchange: identical names and shapes
    net.load_state_dict(state)
    batch = (torch.from_numpy(z[f"{tag}__states"].astype(np.float32)), torch.from_numpy(z[f"{tag}__pis"]),
             torch.from_numpy(z[f"{tag}__z"]))
    loss, _, _ = calculate_loss(net, *batch)
    np.testing.assert_allclose(float(loss), float(z[f"{tag}__loss"]), rtol=1e-6)

    class Pool:
        def sample(self, n, timeout=None):
            return batch

    got = train_step(net, torch.optim.SGD(net.parameters(), lr=0.05), Pool(), 32)
    np.testing.assert_allclose(got[0], float(z[f"{tag}__loss"]), rtol=1e-6)
    for k, v in net.state_dict().items():
        a = v.detach().numpy().reshape(-1)
        np.testing.assert_allclose(a[:16], z[f"{tag}__after_head__{k}"], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(a.astype(np.float64).sum(), float(z[f"{tag}__after_sum__{k}"]), rtol=1e-5, atol=1e-5)


def test_quicknet_is_the_reference_quickmodel():
    """alphazero_amd.nets.QuickNet at the reference's default configuration (3x3 board, 128 channels, one residual block of
    five conv layers) == nn_models/quick_model.py QuickModel: same parameter names and shapes, same outputs on recorded
    inputs with formula weights (tests/golden/quick.npz, written by make_golden.py from the reference)."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("make_golden_formula", os.path.join(GOLDEN, "make_golden.py"))
    src = open(spec.origin).read()
    ns = {"torch": torch}
    start = src.index("def formula_weights")
    exec(src[start:src.index("def make_quick")], ns)  # the formula only (the rest of that script imports the reference)
    z = np.load(os.path.join(GOLDEN, "quick.npz"))
    net = nets.QuickNet(3, 9, 128, 1, 5, 9, 64).eval()
    sd = net.state_dict()
    assert list(sd.keys()) == [str(k) for k in z["names"]]
    assert [str(tuple(t.shape)) for t in sd.values()] == [str(s) for s in z["shapes"]]
    net.load_state_dict(ns["formula_weights"](sd))
    with torch.no_grad():
        p, v = net(torch.from_numpy(z["x"].astype(np.float32)))
    np.testing.assert_allclose(p.numpy(), z["logits"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(v.numpy(), z["value"], rtol=1e-5, atol=1e-6)
    big = nets.QuickNet(15, 9, 128, 4, 5)  # "scaled up" for gomoku 15x15 (BASELINE config 5)
    p, v = big.eval()(torch.zeros(2, 9, 15, 15))
    assert p.shape == (2, 225) and v.shape == (2, 1)


def test_fused_kernel_selection_is_decided_from_the_module_alone():
    """Which hand-written network kernel (if any) serves a module at an engine dtype -- the decision ChessArena(fused=None),
    SelfPlayTrainLoop and bench.py make -- needs no GPU: it reads the module's layer sequence."""
    import torch

    from alphazero_amd import nets
    from alphazero_amd.fused import FusedTowerEvaluator as F, SmallNetEvaluator as S, evaluator_available

    f32, bf16, f16 = torch.float32, torch.bfloat16, torch.float16
    assert F.precision_for(nets.GomokuNet(15, 64, 2), f32) == "f32" and F.precision_for(nets.GomokuNet(15, 64, 2), bf16) == "bf16"
    assert F.precision_for(nets.GomokuNet(9, 128, 3), f32) == "f32" and F.precision_for(nets.GomokuNet(15, 64, 2), f16) is None
    assert F.precision_for(nets.GomokuNet(15, 64, 30), f32) == "f32"      # 60 conv layers at 64 channels: the 32-cell-tile kernel
    assert F.precision_for(nets.GomokuNet(15, 128, 30), f32) is None and F.precision_for(nets.GomokuNet(15, 128, 30), bf16) == "bf16"
    assert F.precision_for(nets.GomokuNet(15, 96, 2), bf16) is None
    assert F.quick_shape(nets.QuickNet(15, 9, 128, 4, 5)) == (15, 128, 4, 5) and F.quick_shape(nets.GomokuNet(15, 64, 2)) is None
    assert F.precision_for(nets.QuickNet(9, 9, 64, 1, 5), f32) == "f32" and F.precision_for(nets.QuickNet(15, 9, 64, 9, 5), f32) is None
    assert F.precision_for(nets.TicTacToeNet(), f32) is None
    assert S.game_of(nets.TicTacToeNet()) == 0 and S.game_of(nets.DotsAndBoxesNet()) == 1 and S.game_of(nets.GomokuNet(9, 64, 1)) is None
    assert evaluator_available(0, nets.TicTacToeNet(), f32) and not evaluator_available(0, nets.TicTacToeNet(), bf16)
    assert evaluator_available(1, nets.DotsAndBoxesNet(), f32) and not evaluator_available(1, nets.TicTacToeNet(), f32)
    assert evaluator_available(2, nets.QuickNet(15, 9, 128, 1, 5), bf16) and not evaluator_available(2, nets.QuickNet(3, 9, 128, 1, 5, n_action=9), f32)


def test_winograd_weight_packing_reproduces_the_convolution():
    """fused._pack_winograd_f32 (the weights of csrc/az_tower_w.hip): unpacked through the kernel's own indexing -- stage, j, kq,
    channel tile, point row, channel, point column -- and run through a plain restatement of F(2x2, 3x3) on the 8 x 8 tiles of a
    15x15 board, U gives the same outputs as torch's conv2d on the original filters; the stem's padded input channels are zero;
    and which networks the Winograd tower takes is decided from the module alone."""
    import torch
    import torch.nn.functional as Fn

    from alphazero_amd import nets
    from alphazero_amd.fused import FusedTowerEvaluator as F, _pack_winograd_f32

    torch.manual_seed(5)
    for ci, pad in ((64, 64), (9, 16)):
        w = torch.randn(64, ci, 3, 3, dtype=torch.float32) * 0.1
        packed = _pack_winograd_f32(w, pad)
        assert tuple(packed.shape) == (pad // 8, 2, 4, 4, 4, 16, 4) and packed.dtype == torch.float32 and packed.is_contiguous()
        # U[co][ci][i][q]: ci = 8 stage + 2 kq + j, co = 16 tile + c
        u = packed.permute(3, 5, 0, 2, 1, 4, 6).reshape(64, pad, 4, 4).double()
        assert float(u[:, ci:].abs().max()) == 0.0 if ci < pad else True
        x = torch.randint(-1, 2, (2, ci, 15, 15)).double()
        xp = Fn.pad(x, (1, 2, 1, 2))  # rows / columns -1 .. 16
        bt = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
        at = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)
        y = torch.zeros(2, 64, 16, 16, dtype=torch.float64)
        for ty in range(8):
            for tx in range(8):
                d = xp[:, :, 2 * ty:2 * ty + 4, 2 * tx:2 * tx + 4]  # (B, ci, 4, 4)
                v = bt @ d @ bt.t()
                m = torch.einsum("ocij,bcij->boij", u[:, :ci], v)
                y[:, :, 2 * ty:2 * ty + 2, 2 * tx:2 * tx + 2] = at @ m @ at.t()
        ref = Fn.conv2d(x, w.double(), padding=1)
        assert float((y[:, :, :15, :15] - ref).abs().max()) < 1e-6
    assert F.winograd_supports(nets.GomokuNet(15, 64, 2)) and F.winograd_supports(nets.GomokuNet(15, 64, 18))
    assert not F.winograd_supports(nets.GomokuNet(15, 64, 19)) and F.winograd_supports(nets.GomokuNet(9, 64, 2))
    assert not F.winograd_supports(nets.GomokuNet(15, 128, 2)) and not F.winograd_supports(nets.QuickNet(15, 9, 64, 1, 2))


def test_bench_launcher_starts_one_rank_per_gpu_before_any_gpu_call():
    """`python bench.py --gpus 2` with no launcher around it becomes the launcher: two children of the same script with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, started before torch is imported (--launch-dry-run returns there).
    The scale-out it stands for: training/alphazero_trainer.py:395-456 (one spawned worker per unit of hardware)."""
    import json
    import subprocess
    import sys

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-dry-run"], env=env,
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    ranks = json.loads(out.stdout.strip().splitlines()[-1])
    assert [r["rank"] for r in ranks] == [0, 1] and [r["local_rank"] for r in ranks] == [0, 1]
    assert all(r["world"] == 2 and r["master_addr"] == "127.0.0.1" and not r["torch_imported"] for r in ranks)
    assert ranks[0]["master_port"] == ranks[1]["master_port"] and ranks[0]["pid"] != ranks[1]["pid"]
    # a world size that does not match --gpus is refused, loudly (the flag is not decoration)
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--launch-dry-run"],
                         env=dict(env, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2"), capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "WORLD_SIZE=2" in bad.stderr
