"""Fused evaluator for GomokuModel-family networks (examples/gomoku/model.py:41-90 and
alphazero_amd.nets.GomokuNet; trunk width 64 or 128, any number of residual blocks, board 9 or 15): the conv tower
runs as ONE hand-written bf16 MFMA kernel
(csrc/az_tower.hip, activations resident in LDS); eval-mode BatchNorm is folded into the conv weights
here; the heads' Linear layers run as one more small MFMA kernel (az_heads_forward).  Same contract as the module it wraps:
``(B, 9, N, N) -> (logits (B, N*N) f32, value (B, 1) f32)``.  Call ``refresh()`` after the module's
weights change (load_state_dict / weight publish).

The reference's QuickModel family (nn_models/quick_model.py with config/quick_model_config.py's default layout, any depth:
alphazero_amd.nets.QuickNet) -- blocks of ``layers`` x (conv3x3, BN, ReLU) summed with their input after the last ReLU,
3x3 head convs, two Linear layers per head -- runs on the QuickModel instantiations of the same two towers: precision="f32"
(az_quick_tower_forward_f32 + az_quick_heads_forward_f32, the reference's arithmetic) or "bf16" (az_quick_tower_forward +
az_quick_heads_forward, throughput mode)."""
from __future__ import annotations

import ctypes as C

import torch
from torch import nn

from .engine import EngineError, load_library


def _fold(conv: nn.Conv2d, bn: nn.BatchNorm2d):
    w = conv.weight.detach().float()
    s = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
    b = bn.bias.detach().float() - bn.running_mean.detach().float() * s
    if conv.bias is not None:
        b = b + conv.bias.detach().float() * s
    return w * s.view(-1, 1, 1, 1), b


def _pack3x3(w: torch.Tensor, cin_pad: int) -> torch.Tensor:
    """(co, ci, 3, 3) f32 -> [tap 9][k-step][k-half 2][co][8] bf16"""
    co, ci = w.shape[0], w.shape[1]
    if ci < cin_pad:
        w = torch.cat([w, w.new_zeros(co, cin_pad - ci, 3, 3)], 1)
    t = w.permute(2, 3, 1, 0).reshape(9, cin_pad // 16, 2, 8, co)  # tap, ks, half, j, co
    return t.permute(0, 1, 2, 4, 3).contiguous().to(torch.bfloat16)


def _pack_linear(w: torch.Tensor, n_pad: int, k_pad: int, k_off_src: int = 0) -> torch.Tensor:
    """(n, k) f32 Linear weight -> [n-tile][k-step][k-half 2][n 32][8 k] bf16, zero padded to (n_pad, k_pad)"""
    n, k = w.shape
    full = w.new_zeros(n_pad, k_pad)
    full[:n, :k] = w
    t = full.reshape(n_pad // 32, 32, k_pad // 16, 2, 8)  # nt, n, ks, half, j
    return t.permute(0, 2, 3, 1, 4).contiguous().to(torch.bfloat16)


def _pack3x3_f32(w: torch.Tensor, cin_pad: int) -> torch.Tensor:
    """(co, ci, 3, 3) f32 -> [tap 9][k-step ci/8][half 2][co][4] f32 (csrc/az_tower_f32.hip).  The stem (ci = 9 < cin_pad = 16)
    puts input channel c at position c + c // 3 (csrc/az_tower_f32.hip: stem_pos): every fourth position stays zero"""
    co, ci = w.shape[0], w.shape[1]
    if ci < cin_pad:
        wp = w.new_zeros(co, cin_pad, 3, 3)
        wp[:, [c + c // 3 for c in range(ci)]] = w
        w = wp
    t = w.permute(2, 3, 1, 0).reshape(9, cin_pad // 8, 2, 4, co)  # tap, ks, half, j, co
    return t.permute(0, 1, 2, 4, 3).contiguous().float()


def _pack_winograd_f32(w: torch.Tensor, cin_pad: int) -> torch.Tensor:
    """(co 64, ci, 3, 3) f32 -> U = G g G^T of every filter (Winograd F(2x2, 3x3); computed in f64, rounded to f32 once) laid out
    [stage ci/8][j 2][kq 4][co tile 4][i 4][co 16][q 4] f32 (csrc/az_tower_w.hip): input channel 8 stage + 2 kq + j, transform
    point (i, q), output channel 16 tile + co"""
    co, ci = w.shape[0], w.shape[1]
    g = torch.tensor([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]], dtype=torch.float64, device=w.device)
    u = (g @ w.double() @ g.t()).float()  # (co, ci, 4, 4)
    if ci < cin_pad:
        u = torch.cat([u, u.new_zeros(co, cin_pad - ci, 4, 4)], 1)
    t = u.reshape(4, 16, cin_pad // 8, 4, 2, 4, 4)  # ct, co16, stage, kq, j, i, q
    return t.permute(2, 4, 3, 0, 5, 1, 6).contiguous()


def _h2_scale(w: torch.Tensor) -> float:
    """the power of two SW with max |w| * SW in (2^13, 2^14] (csrc/az_tower_h2.hip: both f16 halves of w * SW stay in f16's
    normal range for weights down to 2^-14 of the largest), capped to [1, 2^20]"""
    m = float(w.abs().max())
    if not (m > 0.0):
        return 1.0
    import math

    return float(2.0 ** max(0, min(20, 14 - math.ceil(math.log2(m)))))


def _pack3x3_h2(w: torch.Tensor, cin_pad: int, scale: float) -> torch.Tensor:
    """(co 64, ci, 3, 3) f32 -> [tap 9][k-step cin_pad/32][co tile 4][half 2][lane 64][8] f16 (csrc/az_tower_h2.hip): w * scale
    split into two f16 halves, w1 = f16(w s), w2 = f16(w s - w1); lane = 16 kg + (co % 16) holds input channels
    32 ks + 8 kg .. + 7 of output channel 16 ct + (lane % 16)"""
    co, ci = w.shape[0], w.shape[1]
    if ci < cin_pad:
        w = torch.cat([w, w.new_zeros(co, cin_pad - ci, 3, 3)], 1)
    ws = w.float() * scale
    w1 = ws.half()
    w2 = (ws - w1.float()).half()
    parts = []
    for h in (w1, w2):
        t = h.permute(2, 3, 1, 0).reshape(9, cin_pad // 32, 4, 8, 4, 16)  # tap, ks, kg, j, ct, c
        parts.append(t.permute(0, 1, 4, 2, 5, 3))  # tap, ks, ct, kg, c, j
    return torch.stack(parts, 3).contiguous()  # tap, ks, ct, half, kg, c, j


def _pack3x3_h1(w: torch.Tensor, cin_pad: int, scale: float, dtype: torch.dtype) -> torch.Tensor:
    """(co 64, ci, 3, 3) f32 -> [tap 9][k-step cin_pad/32][co tile 4][lane 64][8] f16 | bf16 of w * scale (csrc/az_tower_h1.hip);
    lane = 16 kg + m holds input channels 32 ks + 8 kg .. + 7 of the output channel in row m of co tile ct (order below)"""
    co, ci = w.shape[0], w.shape[1]
    if ci < cin_pad:
        w = torch.cat([w, w.new_zeros(co, cin_pad - ci, 3, 3)], 1)
    # row m of co tile ct is output channel 32 (ct // 2) + 8 (m // 4) + 4 (ct % 2) + m % 4: a lane's results of a wave's two
    # tiles (rows 4 kg .. 4 kg + 3 of each) are then eight consecutive channels, one 16-byte store
    perm = [32 * (t // 2) + 8 * (m // 4) + 4 * (t % 2) + m % 4 for t in range(4) for m in range(16)]
    w = w[perm]
    t = (w.float() * scale).to(dtype).permute(2, 3, 1, 0).reshape(9, cin_pad // 32, 4, 8, 4, 16)  # tap, ks, kg, j, ct, c
    return t.permute(0, 1, 4, 2, 5, 3).contiguous()  # tap, ks, ct, kg, c, j


def _pack_stem_h1(w: torch.Tensor, scale: float, dtype: torch.dtype) -> torch.Tensor:
    """(co 64, 9, 3, 3) f32 -> [kernel row 3][co tile 4][lane 64][8] (csrc/az_tower_h1.hip): one k-step per kernel row, k = 9 tx +
    plane (27 of 32 used); output rows ordered as in _pack3x3_h1"""
    perm = [32 * (t // 2) + 8 * (m // 4) + 4 * (t % 2) + m % 4 for t in range(4) for m in range(16)]
    ws = (w.float() * scale)[perm]  # (64, 9, 3, 3)
    k = w.new_zeros(64, 3, 32)  # co, dy, k
    k[:, :, :27] = ws.permute(0, 2, 3, 1).reshape(64, 3, 27)  # k = 9 tx + plane
    t = k.to(dtype).reshape(4, 16, 3, 4, 8)  # ct, m, dy, kg, i
    return t.permute(2, 0, 3, 1, 4).contiguous()  # dy, ct, kg, m, i


def _pack_linear_f32(w: torch.Tensor, n_pad: int, k_pad: int) -> torch.Tensor:
    """(n, k) f32 Linear weight -> [n-tile][k-step k/8][half 2][n 32][4 k] f32, zero padded to (n_pad, k_pad)"""
    n, k = w.shape
    full = w.new_zeros(n_pad, k_pad)
    full[:n, :k] = w
    t = full.reshape(n_pad // 32, 32, k_pad // 8, 2, 4)  # nt, n, ks, half, j
    return t.permute(0, 2, 3, 1, 4).contiguous().float()


def _keep(obj, name: str, t: torch.Tensor, dev) -> None:
    """obj.<name> = t on ``dev`` -- IN PLACE when the attribute already holds a tensor of that shape and dtype.  A HIP graph
    captured by SelfPlayRunner(use_graph=True) has the packed weights' device addresses baked in: a refresh() after a weight
    publish must therefore rewrite the same allocations (shapes never change between refreshes of one module), not replace
    them -- a replay would go on reading the old, freed blocks."""
    old = getattr(obj, name, None)
    if isinstance(old, torch.Tensor) and old.shape == t.shape and old.dtype == t.dtype and old.device == torch.device(dev):
        old.copy_(t)
    else:
        setattr(obj, name, t.to(dev).contiguous())


class FusedTowerEvaluator:
    @staticmethod
    def quick_shape(model: nn.Module):
        """(board, channels, blocks, layers per block) when the module is a QuickModel-family network the f32 tower evaluates
        (nets.QuickNet / nn_models/quick_model.py defaults: BatchNorm on, one transition conv, 3x3 head convs with 2 / 1
        channels, two Linear layers of 64 features per head, N*N actions), else None"""
        try:
            sl, ph, vh = list(model._shared_layers), list(model._policy_head), list(model._value_head)
            stem, blocks = sl[0], sl[3:]
            ch = stem.out_channels
            conv3 = lambda c, i, o: isinstance(c, nn.Conv2d) and (c.in_channels, c.out_channels) == (i, o) and \
                c.kernel_size == (3, 3) and c.padding == (1, 1) and c.stride == (1, 1)
            if not (conv3(stem, 9, ch) and isinstance(sl[1], nn.BatchNorm2d) and isinstance(sl[2], nn.ReLU) and blocks):
                return None
            layers = None
            for b in blocks:
                seq = list(b.model)
                if len(seq) % 3 or not seq or (layers is not None and len(seq) != 3 * layers):
                    return None
                layers = len(seq) // 3
                for i in range(layers):
                    if not (conv3(seq[3 * i], ch, ch) and isinstance(seq[3 * i + 1], nn.BatchNorm2d) and isinstance(seq[3 * i + 2], nn.ReLU)):
                        return None
            lin = lambda m, i, o: isinstance(m, nn.Linear) and (m.in_features, m.out_features) == (i, o) and m.bias is not None
            if len(ph) != 7 or len(vh) != 7 or not isinstance(ph[6], nn.Linear):
                return None
            n2 = ph[6].out_features
            board = {81: 9, 225: 15}.get(n2)
            ok = board is not None and conv3(ph[0], ch, 2) and isinstance(ph[1], nn.BatchNorm2d) and isinstance(ph[2], nn.ReLU) and \
                isinstance(ph[3], nn.Flatten) and lin(ph[4], 2 * n2, 64) and isinstance(ph[5], nn.ReLU) and lin(ph[6], 64, n2) and \
                conv3(vh[0], ch, 1) and isinstance(vh[1], nn.BatchNorm2d) and isinstance(vh[2], nn.ReLU) and \
                isinstance(vh[3], nn.Flatten) and lin(vh[4], n2, 64) and isinstance(vh[5], nn.ReLU) and lin(vh[6], 64, 1)
            # (the family's tanh lives in the module's value_head method, nn_models/quick_model.py:102-104, not in the Sequential:
            # the kernel applies it)
            ok = ok and ch in (64, 128)
            return (board, ch, len(blocks), layers) if ok else None
        except Exception:
            return None

    @staticmethod
    def supports(model: nn.Module) -> bool:
        if FusedTowerEvaluator.quick_shape(model) is not None:
            return True
        try:
            sl = model._shared_layers
            stem, blocks = sl[0], [m for m in list(sl)[3:]]
            ch = stem.out_channels
            ok = isinstance(stem, nn.Conv2d) and stem.in_channels == 9 and ch in (64, 128) and \
                stem.kernel_size == (3, 3) and isinstance(sl[1], nn.BatchNorm2d)
            ok = ok and all(hasattr(b, "_conv1") and hasattr(b, "_bn2") and b._conv1.out_channels == ch and
                            b._conv1.in_channels == ch and b._conv2.out_channels == ch for b in blocks)
            ph, vh = model._policy_head, model._value_head
            ok = ok and ph[0].out_channels == 1 and vh[0].out_channels == 2 and isinstance(ph[4], nn.Linear) \
                and isinstance(vh[4], nn.Linear) and isinstance(vh[6], nn.Linear)
            n2 = ph[4].in_features
            return bool(ok and n2 in (81, 225))
        except Exception:
            return False

    @staticmethod
    def precision_for(model: nn.Module, dtype: torch.dtype):
        """"bf16" / "f32" when a hand-written tower evaluates this module at this engine dtype, else None (the module's own
        PyTorch forward runs): GomokuModel family and QuickModel family in bf16 and f32 (64 / 128 channels; f32 QuickModel and
        128-channel f32 towers up to 40 conv layers)"""
        if not FusedTowerEvaluator.supports(model):
            return None
        q = FusedTowerEvaluator.quick_shape(model)
        if q is not None:
            board, ch, nb, nl = q
            if dtype == torch.float32:  # the f32 tower: at most 40 conv layers after the transition conv
                return "f32" if nb * nl <= 40 else None
            return "bf16" if dtype == torch.bfloat16 else None
        if dtype == torch.bfloat16:
            return "bf16"
        if dtype != torch.float32:
            return None
        # (precision "f16x2" -- az_tower_h2.hip, f32-level accuracy on the f16 MFMA pipe, 64 channels -- is never picked by
        # default: ask for it by name)
        ch, n_layers = model._shared_layers[0].out_channels, 2 * (len(model._shared_layers) - 3)
        return "f32" if ch == 64 or n_layers <= 40 else None  # (128 channels: the 16-cell-tile kernel only, bias table of 40 layers)

    @classmethod
    def row_tile_supports(cls, model: nn.Module) -> bool:
        """the row-tile tower (csrc/az_tower_h1.hip; precisions "bf16r", "f16"): 64-channel GomokuModel family on 15x15"""
        return (cls.supports(model) and cls.quick_shape(model) is None and model._shared_layers[0].out_channels == 64
                and model._policy_head[4].in_features == 225)

    @classmethod
    def winograd_supports(cls, model: nn.Module) -> bool:
        """the Winograd f32 tower (csrc/az_tower_w.hip; precision "f32w"): 64-channel GomokuModel family on 9x9 or 15x15, <= 36 conv layers"""
        return (cls.supports(model) and cls.quick_shape(model) is None and model._shared_layers[0].out_channels == 64
                and 2 * (len(model._shared_layers) - 3) <= 36)

    def __init__(self, model: nn.Module, precision: str = "bf16"):
        """precision: "bf16" (az_tower.hip: bf16 operands, f32 accumulation -- the throughput mode), "f32"
        (az_tower_f32.hip: f32 MFMA, the reference's own arithmetic), "bf16r" / "f16" (az_tower_h1.hip, 15x15 / 64 channels:
        the faster throughput-mode tower, bf16 or f16 operands; dense input is f32) or "f16x2" (az_tower_h2.hip).  The
        weights are folded and packed from the module's parameters in f32 either way, whatever dtype the module itself holds."""
        if not self.supports(model):
            raise EngineError("FusedTowerEvaluator: not a GomokuModel- or QuickModel-family network (64- or 128-channel 3x3 residual tower)")
        self.lib = load_library()
        self.model = model
        self.device = next(model.parameters()).device
        if self.device.type != "cuda":
            raise EngineError("FusedTowerEvaluator needs the model on a GPU")
        if precision not in ("bf16", "f32", "f32w", "f16x2", "f16", "bf16r"):
            raise EngineError("FusedTowerEvaluator: precision must be 'bf16', 'f32', 'f32w', 'f16x2', 'f16' or 'bf16r'")
        self.quick = self.quick_shape(model)
        if precision == "f16x2" and (self.quick is not None or model._shared_layers[0].out_channels != 64):
            raise EngineError("FusedTowerEvaluator: the f16x2 tower covers 64-channel GomokuModel-family networks")
        if precision in ("f16", "bf16r") and (self.quick is not None or model._shared_layers[0].out_channels != 64
                                              or model._policy_head[4].in_features != 225):
            raise EngineError("FusedTowerEvaluator: the row-tile tower (f16 / bf16r) covers 64-channel GomokuModel-family networks on 15x15")
        if precision == "f32w" and not self.winograd_supports(model):
            raise EngineError("FusedTowerEvaluator: the Winograd f32 tower (f32w) covers 64-channel GomokuModel-family networks "
                              "(9x9 or 15x15), at most 18 residual blocks")
        if self.quick is not None:
            self.board, self.ch, self.q_blocks, self.q_layers = self.quick
            self.n2 = self.board * self.board
            self.precision, self.in_dtype = precision, (torch.float32 if precision == "f32" else torch.bfloat16)
            if self.precision_for(model, self.in_dtype) is None:
                raise EngineError("FusedTowerEvaluator: the f32 QuickModel tower takes at most 40 conv layers after the transition conv")
            self._buf = {}
            self.kernel_events = None
            self.refresh()
            return
        self.n2 = model._policy_head[4].in_features
        self.board = 15 if self.n2 == 225 else 9
        self.ch = model._shared_layers[0].out_channels
        if precision == "f32" and self.precision_for(model, torch.float32) is None:
            raise EngineError("FusedTowerEvaluator: the f32 tower takes 128-channel trunks of at most 40 conv layers")
        self.precision = precision
        self.in_dtype = torch.bfloat16 if precision == "bf16" else torch.float32  # (f16x2, f16, bf16r: f32 dense input)
        self.feat_dtype = torch.bfloat16 if precision in ("bf16", "bf16r") else torch.float32  # bf16 rows + bf16 heads, or f32 rows + f32 heads
        self._buf = {}
        self.kernel_events = None  # bench.py sets this to a list: (start, end) HIP events around the tower kernel alone
        self.refresh()

    generation = 0  # bumped when a refresh() changes something a captured HIP graph holds BY VALUE (a kernel argument)

    def _set_scalar(self, name: str, value: float) -> None:
        """the value head's last bias travels to the kernel as an argument: a graph captured before the refresh keeps the
        old number, so SelfPlayRunner re-captures when ``generation`` has moved (the packed tensors are rewritten in place
        and need no re-capture)"""
        if getattr(self, name, value) != value:
            self.generation += 1
        setattr(self, name, value)

    @torch.no_grad()
    def _refresh_quick(self) -> None:
        m, dev, n2 = self.model, self.device, self.n2
        sl = list(m._shared_layers)
        f32 = self.precision == "f32"
        pack3, packl = (_pack3x3_f32, _pack_linear_f32) if f32 else (_pack3x3, _pack_linear)
        w, b = _fold(sl[0], sl[1])
        _keep(self, "w0", pack3(w, 16), dev)
        biases, wl = [b], []
        for blk in sl[3:]:
            seq = list(blk.model)
            for i in range(self.q_layers):
                w, b = _fold(seq[3 * i], seq[3 * i + 1])
                wl.append(pack3(w, self.ch))
                biases.append(b)
        self.n_layers = len(wl)
        _keep(self, "wl", torch.stack(wl), dev)
        _keep(self, "bias", torch.stack(biases), dev)
        ph, vh = list(m._policy_head), list(m._value_head)
        pw, pb = _fold(ph[0], ph[1])  # (2, ch, 3, 3)
        vw, vb = _fold(vh[0], vh[1])  # (1, ch, 3, 3)
        # [3 head channels][9 taps][ch] f32: policy0, policy1, value
        _keep(self, "headw", torch.cat([pw, vw]).permute(0, 2, 3, 1).reshape(3, 9, self.ch), dev)
        _keep(self, "headb", torch.cat([pb, vb]), dev)
        f = lambda t: t.detach().float()
        self.kpq, self.kvq = (2 * n2 + 15) // 16 * 16, (n2 + 15) // 16 * 16
        self.feat_stride = int(self.lib.az_heads_feat_stride(self.board))
        assert self.feat_stride == self.kpq + self.kvq
        _keep(self, "w1p", packl(f(ph[4].weight), 64, self.kpq), dev)
        _keep(self, "b1p", f(ph[4].bias), dev)
        _keep(self, "w2p", packl(f(ph[6].weight), (n2 + 31) // 32 * 32, 64), dev)
        _keep(self, "b2p", f(ph[6].bias), dev)
        _keep(self, "w1v", packl(f(vh[4].weight), 64, self.kvq), dev)
        _keep(self, "b1v", f(vh[4].bias), dev)
        _keep(self, "w2v", f(vh[6].weight).reshape(-1), dev)
        self._set_scalar("b2v", float(vh[6].bias.detach().float().reshape(-1)[0]))
        if f32 and getattr(self, "stash", None) is None:  # block inputs in flight: one private region per workgroup of a full grid
            self.stash_bytes = int(self.lib.az_quick_stash_bytes(self.board, self.ch))
            if self.stash_bytes <= 0:
                raise EngineError("az_quick_stash_bytes: unsupported board / width")
            self.stash = torch.empty(self.stash_bytes // 4, dtype=torch.float32, device=dev)

    @torch.no_grad()
    def refresh(self) -> None:
        if self.quick is not None:
            return self._refresh_quick()
        m, dev = self.model, self.device
        sl = list(m._shared_layers)
        blocks = sl[3:]
        f32 = self.precision not in ("bf16", "bf16r")  # (f16x2 and f16 share the f32 tower's heads and their packing)
        h1 = self.precision in ("f16", "bf16r")  # the row-tile tower: one 16-bit product, f16 scaled like f16x2 or plain bf16
        h2 = self.precision == "f16x2" or h1
        h1_dtype = torch.bfloat16 if self.precision == "bf16r" else torch.float16
        # (the row-tile tower's f16 scales stop at 2^15 -- the next layer's scale rides on an f16 operand of the residual MFMA --
        # and are 1 unless a layer's weights are tiny (max |w| < 2^-6): one f16 product does not need its small operands lifted
        # the way the split halves of f16x2 do, and a layer at scale 1 skips the epilogue's multiplies)
        scale_of = ((lambda w: 1.0) if self.precision == "bf16r" else
                    (lambda w: 1.0 if float(w.abs().max()) >= 2.0 ** -6 else min(_h2_scale(w), 32768.0)) if h1 else _h2_scale)
        pack_h = (lambda w, cin, sc: _pack3x3_h1(w, cin, sc, h1_dtype)) if h1 else _pack3x3_h2
        pack3 = _pack3x3_f32 if self.precision == "f32" else _pack_winograd_f32 if self.precision == "f32w" else _pack3x3
        w, b = _fold(sl[0], sl[1])
        scales = [scale_of(w)] if h2 else None
        _keep(self, "w0", _pack_stem_h1(w, scales[0], h1_dtype) if h1 else pack_h(w, 32, scales[0]) if h2 else pack3(w, 16), dev)
        biases, wl = [b], []
        for blk in blocks:
            for conv, bn in ((blk._conv1, blk._bn1), (blk._conv2, blk._bn2)):
                w, b = _fold(conv, bn)
                if h2:
                    scales.append(scale_of(w))
                    wl.append(pack_h(w, 64, scales[-1]))
                else:
                    wl.append(pack3(w, self.ch))
                biases.append(b)
        self.n_layers = len(wl)
        wl_dtype = h1_dtype if h1 else torch.float16 if h2 else self.in_dtype
        _keep(self, "wl", torch.stack(wl) if wl else torch.zeros(1, dtype=wl_dtype), dev)
        if h2:  # accumulators run at (true value) x SW_layer x SA (SA = 16, bf16r: 1; csrc/az_tower_h2.hip): the bias rows are pre-scaled
            sw = torch.tensor(scales, dtype=torch.float32)
            _keep(self, "sw", sw, dev)
            _keep(self, "bias", torch.stack(biases).float().cpu() * sw.view(-1, 1) * (1.0 if self.precision == "bf16r" else 16.0), dev)
            if getattr(self, "overflow", None) is None:
                self.overflow = torch.zeros(1, dtype=torch.int32, device=dev)
        else:
            _keep(self, "bias", torch.stack(biases), dev)
        pw, pb = _fold(m._policy_head[0], m._policy_head[1])
        vw, vb = _fold(m._value_head[0], m._value_head[1])
        _keep(self, "headw", torch.cat([pw.view(1, self.ch), vw.view(2, self.ch)]), dev)
        _keep(self, "headb", torch.cat([pb.view(1), vb.view(2)]), dev)
        f = lambda t: t.detach().float()
        pl, v1, v2 = m._policy_head[4], m._value_head[4], m._value_head[6]
        if v1.out_features != 64 or v2.out_features != 1:
            raise EngineError("FusedTowerEvaluator: the value head must be Linear(2 N*N, 64) -> Linear(64, 1)")
        n2 = self.n2
        self.kp, self.kv = (n2 + 15) // 16 * 16, (2 * n2 + 15) // 16 * 16
        self.feat_stride = int(self.lib.az_heads_feat_stride(self.board))
        assert self.feat_stride == self.kp + self.kv
        packl = _pack_linear_f32 if f32 else _pack_linear
        _keep(self, "wp", packl(f(pl.weight), (n2 + 31) // 32 * 32, self.kp), dev)
        _keep(self, "bp", f(pl.bias), dev)
        _keep(self, "w1", packl(f(v1.weight), 64, self.kv), dev)
        _keep(self, "b1", f(v1.bias), dev)
        _keep(self, "w2", f(v2.weight).reshape(-1), dev)
        self._set_scalar("b2", float(v2.bias.detach().float().reshape(-1)[0]))

    def __call__(self, x: torch.Tensor | None, logits_out: torch.Tensor | None = None, packed=None, n_boards: int = 0):
        """x: contiguous (B,9,N,N) bf16 batch, or None with packed = (ring, leaf[, list, count[, flags]]) from
        Engine.leaf_buffers() and n_boards = the engine's slot count: the packed leaf batch is read straight from the
        engine; with list + count (after Engine.compact_leaves()) or, when those are None, with the flags, only the slots
        whose leaf needs the network are evaluated (other output rows keep old data)."""
        if x is not None:
            if x.dtype != self.in_dtype or not x.is_contiguous() or x.device != self.device or x.shape[1] != 9:
                raise EngineError(f"FusedTowerEvaluator expects a contiguous (B,9,N,N) {self.in_dtype} batch on the model's GPU")
            B = x.shape[0]
            ring_p = leaf_p = list_p = count_p = flags_p = None
        else:
            if packed is None or n_boards < 1:
                raise EngineError("FusedTowerEvaluator: give either x or packed=(ring, leaf) with n_boards")
            B = n_boards
            ring_p, leaf_p = C.c_void_p(packed[0]), C.c_void_p(packed[1])
            list_p = C.c_void_p(packed[2]) if len(packed) > 3 and packed[2] else None
            count_p = C.c_void_p(packed[3]) if len(packed) > 3 and packed[3] else None
            flags_p = C.c_void_p(packed[4]) if len(packed) > 4 and packed[4] else None
        if B not in self._buf:  # zeroed once: the kernels never write the pad columns, and the heads multiply them by zero weights
            self._buf[B] = (torch.zeros(B, self.feat_stride, dtype=getattr(self, "feat_dtype", self.in_dtype), device=self.device),
                            torch.empty(B, self.n2, device=self.device), torch.empty(B, device=self.device))
        feat, logits_own, value = self._buf[B]
        logits = logits_out if logits_out is not None else logits_own
        if logits.dtype != torch.float32 or not logits.is_contiguous() or tuple(logits.shape) != (B, self.n2):
            raise EngineError("FusedTowerEvaluator: logits_out must be a contiguous (B, N*N) float32 tensor")
        P = lambda t: C.c_void_p(t.data_ptr())
        with torch.cuda.device(self.device):
            return self._launch(x, P, B, ring_p, leaf_p, list_p, count_p, flags_p, feat, logits, value)

    def _launch(self, x, P, B, ring_p, leaf_p, list_p, count_p, flags_p, feat, logits, value):
        st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        if self.kernel_events is not None:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        if self.quick is not None:
            if list_p is not None:
                raise EngineError("the f32 tower takes the packed batch with flags (or all slots), not a compact list")
            if self.precision == "f32":
                rc = self.lib.az_quick_tower_forward_f32(self.board, self.ch, B, self.q_blocks, self.q_layers, P(x) if x is not None else None,
                                                         ring_p, leaf_p, flags_p, P(self.w0), P(self.wl), P(self.bias), P(self.headw),
                                                         P(self.headb), P(feat), self.feat_stride, P(self.stash), self.stash_bytes, st)
            else:
                rc = self.lib.az_quick_tower_forward(self.board, self.ch, B, self.q_blocks, self.q_layers, P(x) if x is not None else None,
                                                     ring_p, leaf_p, flags_p, P(self.w0), P(self.wl), P(self.bias), P(self.headw),
                                                     P(self.headb), P(feat), self.feat_stride, st)
            if rc != 0:
                raise EngineError(f"az_quick_tower_forward failed ({rc}): {self.lib.az_tower_last_error().decode()}")
            if self.kernel_events is not None:
                ev1.record()
                self.kernel_events.append((ev0, ev1))
            heads = self.lib.az_quick_heads_forward_f32 if self.precision == "f32" else self.lib.az_quick_heads_forward
            rc = heads(self.board, B, P(feat), self.feat_stride, P(self.w1p), P(self.b1p), P(self.w2p), P(self.b2p), P(self.w1v), P(self.b1v),
                       P(self.w2v), self.b2v, P(logits), P(value), st)
            if rc != 0:
                raise EngineError(f"az_quick_heads_forward failed ({rc}): {self.lib.az_tower_last_error().decode()}")
            return logits, value.view(B, 1)
        if self.precision in ("f16", "bf16r"):
            if list_p is not None:
                raise EngineError("the row-tile tower takes the packed batch with flags (or all slots), not a compact list")
            rc = self.lib.az_tower_forward_h1(self.board, self.ch, B, self.n_layers, 1 if self.precision == "bf16r" else 0,
                                              P(x) if x is not None else None, ring_p, leaf_p, flags_p, P(self.w0), P(self.wl),
                                              P(self.bias), P(self.sw), P(self.headw), P(self.headb), P(feat),
                                              1 if self.precision == "bf16r" else 0, self.feat_stride, P(self.overflow), st)
        elif self.precision == "f16x2":
            if list_p is not None:
                raise EngineError("the f16x2 tower takes the packed batch with flags (or all slots), not a compact list")
            rc = self.lib.az_tower_forward_h2(self.board, self.ch, B, self.n_layers, P(x) if x is not None else None, ring_p, leaf_p,
                                              flags_p, P(self.w0), P(self.wl), P(self.bias), P(self.sw), P(self.headw), P(self.headb),
                                              P(feat), self.feat_stride, P(self.overflow), st)
        elif self.precision == "f32w":
            if list_p is not None:
                raise EngineError("the Winograd f32 tower takes the packed batch with flags (or all slots), not a compact list")
            rc = self.lib.az_tower_forward_f32w(self.board, self.ch, B, self.n_layers, P(x) if x is not None else None, ring_p, leaf_p,
                                                flags_p, P(self.w0), P(self.wl), P(self.bias), P(self.headw), P(self.headb),
                                                P(feat), self.feat_stride, st)
        elif self.precision == "f32":
            if list_p is not None:
                raise EngineError("the f32 tower takes the packed batch with flags (or all slots), not a compact list")
            rc = self.lib.az_tower_forward_f32(self.board, self.ch, B, self.n_layers, P(x) if x is not None else None, ring_p, leaf_p,
                                               flags_p, P(self.w0), P(self.wl), P(self.bias), P(self.headw), P(self.headb),
                                               P(feat), self.feat_stride, st)
        else:
            rc = self.lib.az_tower_forward(self.board, self.ch, B, self.n_layers, P(x) if x is not None else None, ring_p, leaf_p,
                                           list_p, count_p, flags_p, P(self.w0), P(self.wl), P(self.bias),
                                           P(self.headw), P(self.headb), P(feat), self.feat_stride, st)
        if rc != 0:
            raise EngineError(f"az_tower_forward failed ({rc}): {self.lib.az_tower_last_error().decode()}")
        if self.kernel_events is not None:
            ev1.record()
            self.kernel_events.append((ev0, ev1))
        heads = self.lib.az_heads_forward if self.precision in ("bf16", "bf16r") else self.lib.az_heads_forward_f32
        rc = heads(self.board, B, P(feat), self.feat_stride, P(self.wp), P(self.bp), P(self.w1), P(self.b1),
                   P(self.w2), self.b2, P(logits), P(value), st)
        if rc != 0:
            raise EngineError(f"az_heads_forward failed ({rc}): {self.lib.az_tower_last_error().decode()}")
        value = value.view(B, 1)
        return logits, value

    def overflowed(self) -> bool:
        """f16x2 only: True when an activation left f16's range (|x| > 4,094) in any evaluation so far -- the results are then
        not to be trusted and the f32 tower must be used for this network.  Synchronises."""
        return self.precision in ("f16x2", "f16") and bool(int(self.overflow.item()))

    def forward_static(self, x: torch.Tensor | None, logits: torch.Tensor, values: torch.Tensor, packed=None, n_boards: int = 0):
        """The same evaluation through az_tower_heads_forward_static: nothing is allocated -- not here (the caller owns
        ``logits`` (B, N*N) and ``values`` (B,) f32) and not in the library (the feature rows between the two kernels live
        in its static per-GPU array).  B is limited to ``az_static_max_boards``; calls on one GPU must not overlap."""
        if self.precision != "bf16" or self.quick is not None:
            raise EngineError("forward_static exists for the bf16 tower of the GomokuModel family")
        if x is not None:
            if x.dtype != torch.bfloat16 or not x.is_contiguous() or x.device != self.device or x.shape[1] != 9:
                raise EngineError("forward_static expects a contiguous (B,9,N,N) bf16 batch on the model's GPU")
            B, ring_p, leaf_p, flags_p = x.shape[0], None, None, None
        else:
            if packed is None or n_boards < 1:
                raise EngineError("forward_static: give either x or packed=(ring, leaf[, list, count, flags]) with n_boards")
            B, ring_p, leaf_p = n_boards, C.c_void_p(packed[0]), C.c_void_p(packed[1])
            flags_p = C.c_void_p(packed[4]) if len(packed) > 4 and packed[4] else None
        for t, shape in ((logits, (B, self.n2)), (values, (B,))):
            if t.dtype != torch.float32 or not t.is_contiguous() or tuple(t.shape) != shape or t.device != self.device:
                raise EngineError(f"forward_static: outputs must be contiguous float32 {shape} tensors on the model's GPU")
        P = lambda t: C.c_void_p(t.data_ptr())
        with torch.cuda.device(self.device):
            st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            rc = self.lib.az_tower_heads_forward_static(self.board, self.ch, B, self.n_layers, P(x) if x is not None else None, ring_p,
                                                        leaf_p, flags_p, P(self.w0), P(self.wl), P(self.bias), P(self.headw),
                                                        P(self.headb), P(self.wp), P(self.bp), P(self.w1), P(self.b1), P(self.w2),
                                                        self.b2, P(logits), P(values), st)
        if rc != 0:
            raise EngineError(f"az_tower_heads_forward_static failed ({rc}): {self.lib.az_tower_last_error().decode()}")
        return logits, values


class SmallNetEvaluator:
    """The reference's two small example networks (examples/tictactoe/model.py, examples/dots_and_boxes/model.py;
    alphazero_amd.nets.TicTacToeNet / DotsAndBoxesNet) as ONE float32 MFMA kernel per evaluation
    (csrc/az_smallnet.hip: both convs, both heads, tanh; a workgroup per 7 / 4 boards, activations in LDS).  Same call
    contract as FusedTowerEvaluator: ``(B, C_in, H, W) f32`` or the engine's packed leaf batch ->
    ``(logits (B, A) f32, value (B, 1) f32)``; f32 operands and accumulation, the reference's precision."""

    generation = 0  # (as FusedTowerEvaluator.generation: nothing here travels to the kernel by value, so it never moves)

    @staticmethod
    def game_of(model: nn.Module):
        """GAME_TICTACTOE / GAME_DAB when the module has exactly the layer sequence the kernel implements, else None"""
        try:
            sl, ph, vh = list(model._shared_layers), list(model._policy_head), list(model._value_head)
            c1, c2 = sl[0], sl[2]
            ok = len(sl) == 4 and isinstance(c1, nn.Conv2d) and isinstance(c2, nn.Conv2d) and isinstance(sl[1], nn.ReLU) and \
                isinstance(sl[3], nn.ReLU) and c1.out_channels == 64 and (c2.in_channels, c2.out_channels) == (64, 64) and \
                all(c.kernel_size == (3, 3) and c.padding == (1, 1) and c.bias is not None for c in (c1, c2))
            if not ok:
                return None
            lin = lambda m, i, o: isinstance(m, nn.Linear) and (m.in_features, m.out_features) == (i, o) and m.bias is not None
            if c1.in_channels == 9 and len(ph) == 4 and len(vh) == 5 and isinstance(ph[0], nn.Flatten) and lin(ph[1], 576, 128) and \
                    isinstance(ph[2], nn.ReLU) and lin(ph[3], 128, 9) and isinstance(vh[0], nn.Flatten) and lin(vh[1], 576, 64) and \
                    isinstance(vh[2], nn.ReLU) and lin(vh[3], 64, 1) and isinstance(vh[4], nn.Tanh):
                return 0
            conv1 = lambda m, o: isinstance(m, nn.Conv2d) and (m.in_channels, m.out_channels) == (64, o) and \
                m.kernel_size == (1, 1) and m.bias is not None
            if c1.in_channels == 25 and len(ph) == 4 and len(vh) == 7 and conv1(ph[0], 2) and isinstance(ph[1], nn.ReLU) and \
                    isinstance(ph[2], nn.Flatten) and lin(ph[3], 32, 24) and conv1(vh[0], 1) and isinstance(vh[1], nn.ReLU) and \
                    isinstance(vh[2], nn.Flatten) and lin(vh[3], 16, 64) and isinstance(vh[4], nn.ReLU) and lin(vh[5], 64, 1) and \
                    isinstance(vh[6], nn.Tanh):
                return 1
        except Exception:
            pass
        return None

    @classmethod
    def supports(cls, model: nn.Module) -> bool:
        return cls.game_of(model) is not None

    def overflowed(self) -> bool:
        return False  # (f32 arithmetic: the FusedTowerEvaluator interface)

    def __init__(self, model: nn.Module):
        self.game = self.game_of(model)
        if self.game is None:
            raise EngineError("SmallNetEvaluator: not the reference's tic-tac-toe or dots-and-boxes network")
        self.lib = load_library()
        self.model = model
        self.device = next(model.parameters()).device
        if self.device.type != "cuda":
            raise EngineError("SmallNetEvaluator needs the model on a GPU")
        self.cin, self.cinp, self.hw, self.n_action = ((9, 16, 3, 9), (25, 32, 4, 24))[self.game]
        self.precision, self.in_dtype = "f32", torch.float32
        self._buf = {}
        self.kernel_events = None
        self.refresh()

    @staticmethod
    def _pack3x3(w: torch.Tensor, cin_pad: int) -> torch.Tensor:
        """(64, ci, 3, 3) -> [tap 9][ci chunk][co 64][4 ci] f32"""
        co, ci = w.shape[0], w.shape[1]
        if ci < cin_pad:
            w = torch.cat([w, w.new_zeros(co, cin_pad - ci, 3, 3)], 1)
        return w.permute(2, 3, 1, 0).reshape(9, cin_pad // 4, 4, co).permute(0, 1, 3, 2).contiguous()

    @torch.no_grad()
    def refresh(self) -> None:
        m, dev = self.model, self.device
        f = lambda t: t.detach().float().cpu()
        sl, ph, vh = list(m._shared_layers), list(m._policy_head), list(m._value_head)
        _keep(self, "w1", self._pack3x3(f(sl[0].weight), self.cinp), dev)
        _keep(self, "b1", f(sl[0].bias), dev)
        _keep(self, "w2", self._pack3x3(f(sl[2].weight), 64), dev)
        _keep(self, "b2", f(sl[2].bias), dev)
        if self.game == 0:
            w = torch.cat([f(ph[1].weight), f(vh[1].weight)])  # (192, 576), column c * 9 + pos (Flatten of (64, 3, 3))
            w = w.view(192, 64, 9).permute(2, 1, 0).reshape(36, 4, 4, 192)  # [pos * 4 + m][kq][j][n]: channel 16 m + 4 kq + j
            parts = [w.permute(0, 1, 3, 2), torch.cat([f(ph[1].bias), f(vh[1].bias)]), f(ph[3].weight), f(ph[3].bias),
                     f(vh[3].weight), f(vh[3].bias)]
        else:
            parts = [torch.cat([f(ph[0].weight).view(2, 64), f(vh[0].weight).view(1, 64)]), torch.cat([f(ph[0].bias), f(vh[0].bias)]),
                     f(ph[3].weight), f(ph[3].bias), f(vh[3].weight), f(vh[3].bias), f(vh[5].weight), f(vh[5].bias)]
        head = torch.cat([p.contiguous().reshape(-1) for p in parts])
        assert head.numel() == int(self.lib.az_smallnet_head_floats(self.game))
        _keep(self, "head", head, dev)

    def __call__(self, x: torch.Tensor | None, logits_out: torch.Tensor | None = None, packed=None, n_boards: int = 0):
        if x is not None:
            if x.dtype != torch.float32 or not x.is_contiguous() or x.device != self.device or \
                    tuple(x.shape[1:]) != (self.cin, self.hw, self.hw):
                raise EngineError(f"SmallNetEvaluator expects a contiguous (B,{self.cin},{self.hw},{self.hw}) float32 batch on the model's GPU")
            B, ring_p, leaf_p, flags_p = x.shape[0], None, None, None
        else:
            if packed is None or n_boards < 1:
                raise EngineError("SmallNetEvaluator: give either x or packed=(ring, leaf[, list, count, flags]) with n_boards")
            B, ring_p, leaf_p = n_boards, C.c_void_p(packed[0]), C.c_void_p(packed[1])
            flags_p = C.c_void_p(packed[4]) if len(packed) > 4 and packed[4] else None
        if B not in self._buf:
            self._buf[B] = (torch.empty(B, self.n_action, device=self.device), torch.empty(B, device=self.device))
        logits_own, value = self._buf[B]
        logits = logits_out if logits_out is not None else logits_own
        if logits.dtype != torch.float32 or not logits.is_contiguous() or tuple(logits.shape) != (B, self.n_action):
            raise EngineError("SmallNetEvaluator: logits_out must be a contiguous (B, n_action) float32 tensor")
        P = lambda t: C.c_void_p(t.data_ptr())
        with torch.cuda.device(self.device):
            st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            if self.kernel_events is not None:
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record()
            rc = self.lib.az_smallnet_forward_f32(self.game, B, P(x) if x is not None else None, ring_p, leaf_p, flags_p,
                                                  P(self.w1), P(self.b1), P(self.w2), P(self.b2), P(self.head), P(logits), P(value), st)
            if rc != 0:
                raise EngineError(f"az_smallnet_forward_f32 failed ({rc}): {self.lib.az_tower_last_error().decode()}")
            if self.kernel_events is not None:
                ev1.record()
                self.kernel_events.append((ev0, ev1))
        return logits, value.view(B, 1)


def evaluator_available(game_id: int, model: nn.Module, dtype: torch.dtype) -> bool:
    """True when one of the hand-written network kernels evaluates ``model`` for an engine of this game and nn_dtype
    (``SelfPlayRunner(..., fused=True)`` will then construct it): the GomokuModel / QuickModel towers for gomoku, the
    one-kernel f32 networks for tic-tac-toe (0) and dots-and-boxes (1)"""
    if game_id == 2:
        return FusedTowerEvaluator.precision_for(model, dtype) is not None
    return dtype == torch.float32 and SmallNetEvaluator.game_of(model) == game_id
