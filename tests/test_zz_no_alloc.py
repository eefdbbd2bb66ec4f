"""az_tower_heads_forward_static (the entry point for callers that cannot allocate: the feature rows between k_tower and
k_heads live in the library's static per-GPU array) against az_tower_forward + az_heads_forward with a caller-owned
buffer: bit-exact, on calls in a row that shrink and change the row layout (15x15 -> smaller 15x15 -> 9x9 -> 15x15),
through the dense and the packed + flags inputs; sizes beyond the array are refused with a clear error."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _net(board, ch, blocks, seed):
    from alphazero_amd.nets import GomokuNet

    from .test_gpu_tower import _randomised

    torch.manual_seed(seed)
    return _randomised(GomokuNet(board, ch, blocks)).eval().cuda().bfloat16()


def _batch(B, board, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.randint(-1, 2, (B, 9, board, board), device="cuda", generator=g).float()
    x[:, 8] = torch.where(torch.rand(B, 1, 1, device="cuda", generator=g) < 0.5, -1.0, 1.0)
    return x.bfloat16().contiguous()


def test_static_entry_point_matches_the_allocating_one():
    from alphazero_amd.engine import EngineError, load_library
    from alphazero_amd.fused import FusedTowerEvaluator

    lib = load_library()
    assert lib.az_static_max_boards(15) == (6 << 20) // (704 * 2) == 4468 and lib.az_static_max_boards(9) == (6 << 20) // (272 * 2)
    ev15, ev9, ev128 = FusedTowerEvaluator(_net(15, 64, 2, 0)), FusedTowerEvaluator(_net(9, 64, 3, 1)), FusedTowerEvaluator(_net(15, 128, 2, 2))
    # a row of calls on the one shared array: big, smaller (stale rows behind it), another row layout, back again
    for k, (ev, board, B) in enumerate([(ev15, 15, 1500), (ev15, 15, 137), (ev9, 9, 2000), (ev9, 9, 31), (ev128, 15, 300), (ev15, 15, 4468)]):
        x = _batch(B, board, 10 + k)
        want_p, want_v = ev(x)
        want_p, want_v = want_p.clone(), want_v.clone().view(-1)
        got_p = torch.full((B, board * board), float("nan"), device="cuda")
        got_v = torch.full((B,), float("nan"), device="cuda")
        before = torch.cuda.memory_allocated()
        ev.forward_static(x, got_p, got_v)
        assert torch.cuda.memory_allocated() == before  # (torch's view of it; the library itself has no allocation call)
        torch.cuda.synchronize()
        assert torch.equal(got_p, want_p) and torch.equal(got_v, want_v), (k, board, B)
    with pytest.raises(EngineError, match="exceeds the static scratch"):
        ev15.forward_static(_batch(4469, 15, 3), torch.empty(4469, 225, device="cuda"), torch.empty(4469, device="cuda"))


def test_static_entry_point_on_the_packed_leaf_batch():
    """the engine's packed leaf batch + flags, two searches in a row, the second with fewer slots"""
    from alphazero_amd import Engine
    from alphazero_amd.fused import FusedTowerEvaluator

    ev = FusedTowerEvaluator(_net(15, 64, 2, 0))
    for G in (700, 90):
        eng = Engine(2, G, 8, board_size=15, nn_dtype=torch.bfloat16, device="cuda:0")
        A = 225
        logits, values = eng.new(G, A), eng.new(G)
        it = torch.ones(G, device="cuda")
        for _ in range(3):  # a few plies so that histories differ
            eng.begin_search()
            for _ in range(8):
                eng.select_gather(None)
                eng.eval_hash(None, logits, values, 6.0)
                eng.expand_backup(logits, values, None)
            eng.play_moves(it, torch.empty(G, A, device="cuda").exponential_(1))
        eng.begin_search()
        eng.select_gather(None)
        ring, leaf, _, _, flags = eng.leaf_buffers()
        want_p, want_v = ev(None, packed=(ring, leaf, None, None, flags), n_boards=G)
        want_p, want_v = want_p.clone(), want_v.clone().view(-1)
        got_p, got_v = torch.zeros(G, A, device="cuda"), torch.zeros(G, device="cuda")
        ev.forward_static(None, got_p, got_v, packed=(ring, leaf, None, None, flags), n_boards=G)
        torch.cuda.synchronize()
        eng.check()
        assert torch.equal(got_p, want_p) and torch.equal(got_v, want_v)
