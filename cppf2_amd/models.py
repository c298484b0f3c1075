"""The tuple MLPs of CPPF++: plain torch modules (training, checkpoints, the reference's state-dict layout) whose inference
runs through the library's matrix-core kernels (fused_stack), with the tuple encode (prepare_tuple_inputs) and the bin draw
feeding / draining them in-kernel.

State-dict key layout is the reference's (train_shot.py:19-73, train_dino.py:21-89:
`shot_encoder.N.fc1.weight`, `tuple_encoder.N.fc0.bias`, `logit_encoder...`, `scale_encoder...`,
`desc_transform.*`, `desc_pair_transform.*`), so a Lightning checkpoint's `state_dict` loads with
load_reference_checkpoint().
"""
from __future__ import annotations

import os
from itertools import combinations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops


class ResLayer(nn.Module):
    """Residual 2-layer block, no norm, no dropout (train_shot.py:19-45 with bn=False, dropout=False)."""

    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.fc1 = nn.Linear(dim_in, dim_out)
        self.fc2 = nn.Linear(dim_out, dim_out)
        self.fc0 = nn.Linear(dim_in, dim_out) if dim_in != dim_out else None

    def forward(self, x):
        skip = x if self.fc0 is None else self.fc0(x)
        return self.fc2(F.relu(self.fc1(x))) + skip


def _stack(dims):
    return nn.Sequential(*[ResLayer(dims[i], dims[i + 1]) for i in range(len(dims) - 1)])


# Arithmetic of the inference MLPs on the GPU: "split" = every ResLayer as one kernel on the bf16 matrix cores with each
# float32 operand split exactly into three bf16 values (cppf_reslayer_split: error vs float64 within 3 x a float32 GEMM's, 2-3x the rate of
# the f32-input matrix instruction); "native" = f32-input matrix cores (library GEMMs + cppf_reslayer128).
# "split16" = the same kernels in f16x2 arithmetic (fp16 operand pairs, three products: half the matrix-core work, float32-GEMM
# level error against float64, fp16's operand range; ops.reslayer_split16) -- an option, not the default.
MLP_ARITH = os.environ.get("CPPF_MLP_ARITH", "split")


def _kernel_arith():
    """True when the ResLayers run as the library's matrix-core kernels (either operand split)."""
    return MLP_ARITH in ("split", "split16")


def split_bf16(w):
    """float32 tensor -> bfloat16 [3, ...]: w == hi + mid + lo exactly (round-to-nearest-even each, exact remainders)."""
    w = w.detach().float()
    hi = w.to(torch.bfloat16)
    r1 = w - hi.float()
    mid = r1.to(torch.bfloat16)
    lo = (r1 - mid.float()).to(torch.bfloat16)
    return torch.stack([hi, mid, lo])


def split_f16(w):
    """float32 tensor -> float16 [2, ...]: hi = RNE(w), lo = RNE(w - hi): w to 22-23 significant bits (f16x2 arithmetic)."""
    w = w.detach().float()
    hi = w.to(torch.float16)
    lo = (w - hi.float()).to(torch.float16)
    return torch.stack([hi, lo])


def f16_scale(*weights):
    """The power of two the weights of one f16x2 launch are multiplied by: the largest |w| lands in [2^12, 2^13)."""
    import math
    m = max(float(w.detach().abs().max()) for w in weights if w is not None)
    return 2.0 ** (13 - math.ceil(math.log2(m))) if m > 0 and math.isfinite(m) else 1.0


def pack_split(w1, w0, w2, k_in, chain=(), arith="bf16x3", scale=1.0):
    """The weight stream cppf_reslayer_split consumes for one ResLayer (w1 [N, K], w0 [N, K] or None, w2 [N, N] as
    nn.Linear stores them; k_in >= K = the columns of x the kernel reads, the extra ones get zero weights), optionally
    followed by the identity layers `chain` = [(w1_l [N, N], w2_l [N, N]), ...] of the same width.
    One operand fragment = 64 lanes x 8 bf16; lane = 32 g + i multiplies output feature 32 u + i of tile u; a tile is its
    (hi, mid, lo) fragments; a K step (16 input features) of a product is its tiles one after the other:
      first product, step s: lane half g holds input features 16 s + 8 g + j, j < 8; tiles = the N/32 tiles of W1 and, for
        a projection layer, the N/32 tiles of W0 behind them (one pass over x computes both);
      second product (W2), step (t, s') over the hidden features in the accumulator order of the first product:
        lane half g holds hidden features 32 t + 16 s' + 4 g + (j & 3) + 8 (j >> 2);
      then per chained layer W1_l and W2_l, both in that accumulator feature order (their input is the previous layer's
      output tiles in registers).
    arith="f16x2" (cppf_reslayer_split16): the same order with (hi, lo) fp16 fragments of scale x the weights."""
    n = w1.shape[0]
    nt = n // 32
    ks1 = (k_in + 15) // 16
    dev = w1.device
    if arith == "f16x2":
        pc, split = 2, (lambda w: split_f16(w * scale))
    else:
        pc, split = 3, split_bf16

    def pack_x(w, tiles):
        s = split(F.pad(w.detach().float(), (0, ks1 * 16 - w.shape[1])))
        return s.reshape(pc, tiles, 32, ks1, 2, 8).permute(3, 1, 0, 4, 2, 5).reshape(-1)      # [s, u, slice, g, i, j]

    def pack_h(w):
        t = torch.arange(nt, device=dev).view(nt, 1, 1, 1)
        sp = torch.arange(2, device=dev).view(1, 2, 1, 1)
        g = torch.arange(2, device=dev).view(1, 1, 2, 1)
        j = torch.arange(8, device=dev).view(1, 1, 1, 8)
        col = (32 * t + 16 * sp + 4 * g + (j & 3) + 8 * (j >> 2)).reshape(-1)
        s = split(w)[:, :, col]
        return s.reshape(pc, nt, 32, nt, 2, 2, 8).permute(3, 4, 1, 0, 5, 2, 6).reshape(-1)    # [t, s', u, slice, g, i, j]

    parts = [pack_x(w1, nt) if w0 is None else pack_x(torch.cat([w1, w0]), 2 * nt), pack_h(w2)]
    for w1_l, w2_l in chain:
        assert w1_l.shape == (n, n) and w2_l.shape == (n, n)
        parts += [pack_h(w1_l), pack_h(w2_l)]
    return torch.cat(parts).contiguous()


def pack_linear(w, k_in, arith="bf16x3", scale=1.0):
    """The weight stream cppf_linear_split consumes for out = x W^T (w [n_out, K] as nn.Linear stores it, n_out a multiple of
    256, k_in >= K the columns of x the kernel reads): per column group of 256 outputs the first-product segment of pack_split
    (K steps of 8 tiles, operand pieces per tile), the groups one after the other."""
    n, k = w.shape
    assert n % 256 == 0 and k_in >= k
    ks1 = (k_in + 15) // 16
    pc, split = (2, (lambda t: split_f16(t * scale))) if arith == "f16x2" else (3, split_bf16)
    s = split(F.pad(w.detach().float(), (0, ks1 * 16 - k)))                         # [pc, n, ks1 * 16]
    return s.reshape(pc, n // 256, 8, 32, ks1, 2, 8).permute(1, 4, 2, 0, 5, 3, 6).reshape(-1).contiguous()   # [grp, s, u, slice, g, i, j]


class _FoldedFirstLayer:
    """A tuple encoder's first ResLayer (a projection layer) on rows [heads | s] with s linear in per-point vectors p of the tuple's
    points, s = s0 + sum_i E_i p[idx_i]: the per-point parts of x W1^T and x W0^T are moved into per-point slot tables
        tables[n, i] = [A_i p_n | C_i p_n],   A_i = W1[:, s-columns] E_i,   C_i = W0[:, s-columns] E_i      (128 + 128 floats)
    (ops.linear_split with `wq_tab`), the constant parts into the biases, and the layer's own products shrink to the head columns
    (w1_heads / w0_heads).  Algebraically the same layer; the products are folded in float64 and rounded to float32 once."""

    def __init__(self, stamp, a_list, c_list, w1_heads, w0_heads, b1_add=None, b0_add=None):
        self.stamp = stamp
        self.slots = len(a_list)
        self.dp = a_list[0].shape[1]
        self.tab_w = torch.cat([torch.cat([a, c]) for a, c in zip(a_list, c_list)]).float().contiguous()     # [slots * 256, dp]
        self.w1_heads, self.w0_heads = w1_heads.float().contiguous(), w0_heads.float().contiguous()       # [128, head columns]
        self.b1_add, self.b0_add = b1_add, b0_add
        self._wq = {}

    def table_stream(self):
        """(packed stream of the table weights, weight scale or None) in the current arithmetic."""
        if MLP_ARITH not in self._wq:
            if MLP_ARITH == "split16":
                sc = f16_scale(self.tab_w)
                self._wq[MLP_ARITH] = (pack_linear(self.tab_w, self.dp, arith="f16x2", scale=sc), sc)
            else:
                self._wq[MLP_ARITH] = (pack_linear(self.tab_w, self.dp), None)
        return self._wq[MLP_ARITH]

    def tables(self, p):
        """[points, slots * 256] from the per-point vectors p [points, dp] (float32, device)."""
        wq, sc = self.table_stream()
        return ops.linear_split(p, wq, None, self.slots * 256, scale=sc)


def _fused_plan(seq):
    """Per-layer (W1^T, b1', W0^T | None, b0', W2^T) with the pending-offset algebra of fused_stack applied once; cached
    on the module and rebuilt when any parameter changed (version counters) or moved."""
    stamp = tuple((p.data_ptr(), p._version) for p in seq.parameters())
    cached = getattr(seq, "_fused_plan_cache", None)
    if cached is not None and cached[0] == stamp:
        return cached[1]
    plan, c = [], None
    with torch.no_grad():
        for layer in seq:
            w1, b1, w2, b2 = layer.fc1.weight, layer.fc1.bias, layer.fc2.weight, layer.fc2.bias
            if c is not None:
                b1 = torch.addmv(b1, w1, c)
            if layer.fc0 is not None:
                b0 = layer.fc0.bias + b2
                if c is not None:
                    b0 = torch.addmv(b0, layer.fc0.weight, c)
                plan.append([w1.t(), b1.clone(), layer.fc0.weight.t(), b0, w2.t(), None])
                c = None
            else:
                plan.append([w1.t(), b1.clone(), None, None, w2.t(), None])
                c = b2.clone() if c is None else c + b2
    seq._fused_plan_cache = (stamp, (plan, c))
    return plan, c


def _entry_cache(entry):
    """Per-layer cache of packed weight streams and chain lengths, one slot per launch shape (input width, chain | "decode"):
    alternating callers (with / without the fused bin draw, different input widths) do not evict each other's streams."""
    if entry[5] is None:
        entry[5] = {}
    return entry[5]


def _packed(cache, key, w1t, w0t, w2t, k_in, rest, b1, b0, stamp=None, add=None):
    """(weight stream, first biases of the launch's layers, skip bias, weight scale) of one launch in the current arithmetic,
    cached per launch shape: bf16 triples (scale 1) or fp16 pairs of scale x the weights with the biases scaled alike.
    stamp: weight versions of a SECOND stack the launch runs into (cross-stack launches); it is stored inside the slot and the
    slot is rebuilt when it changes, so that re-trained weights of the second stack replace their stream instead of adding one."""
    key = key + (MLP_ARITH,)
    hit = cache.get(key)
    if hit is None or hit[0] != stamp:
        if add is not None:                      # constant terms folded into the first layer's biases (_FoldedFirstLayer)
            b1 = b1 if add[0] is None else b1 + add[0]
            b0 = b0 if add[1] is None else b0 + add[1]
        chain = [(e[0].t(), e[4].t()) for e in rest]
        biases = torch.cat([b1] + [e[1] for e in rest])
        if MLP_ARITH == "split16":
            sc = f16_scale(w1t, w0t, w2t, *[w for pair in chain for w in pair])
            wq = pack_split(w1t.t(), None if w0t is None else w0t.t(), w2t.t(), k_in, chain=chain, arith="f16x2", scale=sc)
            hit = (stamp, (wq, (biases * sc).contiguous(), None if b0 is None else (b0 * sc).contiguous(), sc))
        else:
            wq = pack_split(w1t.t(), None if w0t is None else w0t.t(), w2t.t(), k_in, chain=chain)
            hit = (stamp, (wq, biases.contiguous(), b0, 1.0))
        cache[key] = hit
    return hit[1]


def _chain_len(cache, key, stamp, compute):
    """Chain length of a launch shape, cached like _packed (one slot per shape, the second stack's stamp inside it)."""
    hit = cache.get(key)
    if hit is None or hit[0] != stamp:
        hit = (stamp, compute())
        cache[key] = hit
    return hit[1]


def fused_stack(seq, x, keep_input=False, gather=None, decode=None, tail=None):
    """Inference-only execution of a stack of ResLayers (train_shot.py:19-45) on the matrix cores.
    MLP_ARITH == "split" (default): every layer whose shape the kernel covers -- widths 64 / 128 / 192 / 256, input columns
    a multiple of 8 -- runs as ONE cppf_reslayer_split launch together with the identity layers of the same width behind it
    (the activation stays in registers across the chain); split-bf16 arithmetic on exact float32 operands.
    MLP_ARITH == "native", and layers outside that coverage (the scale head's 64 -> 3 output layer): library GEMMs with the
    elementwise work folded into their epilogues --
      h   = relu(x W1^T + b1)            one GEMM, bias+ReLU epilogue (torch._addmm_activation)
      out = x W0^T + (b0 + b2)           one GEMM, bias epilogue          (layers with a projection skip)
      out += h W2^T                      one GEMM, beta = 1, in place
    -- and 128-wide identity layers as cppf_reslayer128 (f32-input matrix cores, h never leaves the registers).
    In both modes an identity layer's output bias b2 is not added to the activation but carried as a pending per-channel
    offset c (true activation = x + c) and folded into the biases of the next layer (b1 + W1 c, b0 + W0 c), which is
    algebraically the same network; every stack of the reference's models ends in a projection layer, which absorbs the
    pending offset.  The folded biases (and the packed weight streams of the split kernels) depend on the weights only and
    are computed once per weight version (_fused_plan).  An identity first layer overwrites `x` unless keep_input is set.
    gather = (heads [T, H], gidx int32 [T, k], table [points, F]) instead of x: the rows [heads | table[gidx[:, 0]] | ...] are
    read by the first layer's kernel itself (ops.reslayer_split_gather; split arithmetic, 128-wide projection first layer).
    decode = (uniforms [T, 6], prior [T, 6, 32] | None, bins int32 [T, 6] | None): the stack's last layer (the logit head's
    192-wide projection layer) draws the bins in its epilogue instead of writing its logits (ops.reslayer_split_decode); the
    return value is then the bins.  Use decode_supported(seq, x) first.
    seq = (stack_a, stack_b): the two stacks run as one, returning (output of stack_b, output of stack_a): stack_a's last layer
    (a projection layer) and the identity layers that open stack_b share one kernel whose first layer's result is written to a
    buffer of its own (ops.reslayer_split(tap=...)) -- the tuple encoder into the logit head, whose input also feeds the scale
    head (train_shot.py:112-114): the 256-wide features are written once and not read back by the logit head.
    tail = (rows int32 [n], counts int32 [n / per_group], per_group, out [T, n_out]): the stack's last layer, when it is a narrow
    one (n_out <= 8: ops.reslayer_tail), writes row i of its result to out[rows[i]], skipping the padded entries of each
    group (VotingPipeline.kept_rows32 / kept_count / max_kept); returns `out`.  Use tail_supported(seq) first."""
    tap_at, cross = None, None
    if isinstance(seq, (tuple, list)):
        # two stacks run as one: the first one's output is tapped (returned as well) while the identity layers that open the
        # second continue in the same kernel.  The first stack must end in a projection layer (nothing pending).
        seq_a, seq_b = seq
        plan_a, c_a = _fused_plan(seq_a)
        assert c_a is None, "the first stack must end in a projection layer"
        plan_b, c = _fused_plan(seq_b)
        plan = plan_a + plan_b
        tap_at = len(plan_a) - 1
        cross = seq_b._fused_plan_cache[0]          # weight versions of the second stack: stored inside every cross-stack cache slot
        seq = seq_b
    else:
        plan, c = _fused_plan(seq)
    tapped = None

    def cut(li_, chain_):
        """chain length of a launch starting at layer li_ so that it does not run past the tapped layer unless it starts there"""
        if tap_at is not None and li_ < tap_at <= li_ + chain_:
            return tap_at - li_
        return chain_
    li = 0
    if gather is not None:
        # gather = (heads, gidx, table): x-tile gather of [heads | table[gidx_0] | ...]; gather = (heads, gidx, tables, fold): the
        # per-point parts of the first products come from slot tables (fold: _FoldedFirstLayer) and are summed into the accumulators
        fold = gather[3] if len(gather) > 3 else None
        heads, gidx, table = gather[:3]
        source = None
        if isinstance(heads, ops.TupleSource):
            # the head columns are built inside the first launch (cppf_reslayer_split_encode: the SHOT model's 40 pair features;
            # cppf_reslayer_split_sumencode: the DINO model's coordinate block in front of its slot tables) where that kernel
            # exists; the other launch forms read them from the array the separate kernel writes
            if (MLP_ARITH == "split" and ops.reslayer_split_encode_supported(heads.k, 128)
                    and (fold is None) == (heads.nrm is not None)):
                source = heads
            else:
                heads, gidx = heads.heads()
        entry = plan[0]
        w1t, b1, w0t, b0, w2t = entry[:5]
        if fold is not None:
            k_in = heads.shape[1]
            assert _kernel_arith() and w0t is not None and w1t.shape[1] == 128 and k_in % 8 == 0, "table-fed first layer: see sum_supported"
            w1t, w0t = fold.w1_heads.t(), fold.w0_heads.t()
        else:
            k_in = heads.shape[1] + (source.k if source is not None else gidx.shape[1]) * table.shape[1]
            assert _kernel_arith() and w0t is not None and w1t.shape == (k_in, 128), "gathered first layer: see gather_supported"
        cache = _entry_cache(entry)

        def gather_chain():
            n = 0
            while (1 + n < len(plan) and plan[1 + n][2] is None and n < 15 and plan[1 + n][0].shape == (128, 128)):
                n += 1
            n = cut(0, n)
            return 0 if tap_at == 0 else n      # (the gathering launch has no second output)
        chain = _chain_len(cache, ("gather-chain", k_in, tap_at), cross, gather_chain)
        crossing = tap_at is not None and chain > tap_at
        stamp = (cross if crossing else None, None if fold is None else fold.stamp)
        wq, bb1, bb0, sc = _packed(cache, (k_in, chain, crossing, fold is not None), w1t, w0t, w2t, k_in, plan[1:1 + chain], b1, b0,
                                   stamp=stamp, add=None if fold is None else (fold.b1_add, fold.b0_add))
        if fold is not None and source is not None:
            x = ops.reslayer_split_sumencode(source, table, wq, bb1, bb0, 128, chain=chain)
        elif fold is not None:
            x = ops.reslayer_split_sumgather(heads, gidx, table, wq, bb1, bb0, 128, chain=chain,
                                             scale=sc if MLP_ARITH == "split16" else None)
        elif MLP_ARITH == "split16":
            x = ops.reslayer_split16(heads, wq, bb1, bb0, 128, sc, chain=chain, gather=(gidx, table))
        elif source is not None:
            x = ops.reslayer_split_encode(source, table, wq, bb1, bb0, 128, chain=chain)
        else:
            x = ops.reslayer_split_gather(heads, gidx, table, wq, bb1, bb0, 128, chain=chain)
        li = 1 + chain
        if tap_at is not None and li - 1 == tap_at:
            tapped = x
    while li < len(plan):
        entry = plan[li]
        w1t, b1, w0t, b0, w2t = entry[:5]
        n_out = w1t.shape[1]
        if (decode is not None and li == len(plan) - 1):
            assert c is None and decode_supported(seq, x), "fused bin draw: see decode_supported"
            cache = _entry_cache(entry)
            wq, bb1, bb0, sc = _packed(cache, (x.shape[1], "decode"), w1t, w0t, w2t, x.shape[1], [], b1, b0)
            if MLP_ARITH == "split16":
                bins = ops.reslayer_split16(x, wq, bb1, bb0, 192, sc, decode=decode)
            else:
                bins = ops.reslayer_split_decode(x, wq, bb1, bb0, decode[0], prior=decode[1], bins=decode[2])
            return bins if tap_at is None else (bins, tapped)
        if (_kernel_arith() and x.dtype == torch.float32 and x.stride(1) == 1 and x.stride(0) % 4 == 0
                and x.data_ptr() % 16 == 0 and x.shape[1] >= w1t.shape[0]
                and ops.reslayer_split_supported(x.shape[1], n_out, w0t is not None)):
            # the whole layer -- and the identity layers of the same width behind it, while they fit one kernel -- on the
            # bf16 matrix cores in split-float32 arithmetic; the activation stays in registers across the chain
            cache = _entry_cache(entry)
            ahead = tap_at is not None and tap_at >= li

            def plain_chain(li=li, n_out=n_out, w0t=w0t):  # (one library call per candidate layer: looked up once per shape)
                n = 0
                while (li + 1 + n < len(plan) and plan[li + 1 + n][2] is None and n < 15
                       and plan[li + 1 + n][0].shape == (n_out, n_out)
                       and ops.reslayer_split_supported(x.shape[1], n_out, w0t is not None, n + 1)):
                    n += 1
                return cut(li, n)
            chain = _chain_len(cache, ("chain", x.shape[1], tap_at if ahead else None), cross if ahead else None, plain_chain)
            crossing = tap_at is not None and li <= tap_at < li + chain           # the launch runs from one stack into the other
            wq, bb1, bb0, sc = _packed(cache, (x.shape[1], chain, crossing), w1t, w0t, w2t, x.shape[1],
                                       plan[li + 1:li + 1 + chain], b1, b0, stamp=cross if crossing else None)
            out = None
            if w0t is None and li == 0 and keep_input:
                out = torch.empty_like(x)
            if w0t is None and out is None and tapped is not None and x.data_ptr() == tapped.data_ptr():
                out = torch.empty_like(x)                 # an in-place identity layer must not overwrite the tapped activation
            tap_buf = None
            if tap_at == li and chain > 0:
                # the tapped layer opens this launch: its output goes to a buffer of its own, the chain's to another
                tap_buf = torch.empty((x.shape[0], n_out), dtype=torch.float32, device=x.device)
                if w0t is None and out is None:
                    out = torch.empty_like(x)
            if MLP_ARITH == "split16":
                x = ops.reslayer_split16(x, wq, bb1, bb0, n_out, sc, out=out, chain=chain, tap=tap_buf)
            else:
                x = ops.reslayer_split(x, wq, bb1, bb0, n_out, out=out, chain=chain, tap=tap_buf)
            if tap_at is not None and li <= tap_at <= li + chain:
                tapped = tap_buf if tap_buf is not None else x
            li += 1 + chain
            continue
        if (_kernel_arith() and n_out <= 8 and x.dtype == torch.float32 and x.stride(1) == 1 and x.stride(0) % 4 == 0
                and x.data_ptr() % 16 == 0 and w1t.shape[0] % 4 == 0 and x.shape[1] >= w1t.shape[0]
                and (w0t is not None or w1t.shape[0] == n_out) and (tail is not None or c is None or li + 1 < len(plan))):
            # a layer too narrow for a matrix-core tile (the scale head's 64 -> 3): plain float32, one thread per row, in the
            # library (cppf_reslayer_tail); tail = (rows, counts, per_group, out): its rows are scattered into `out` on the way
            last = li == len(plan) - 1
            t_ = tail if (tail is not None and last) else None
            x = ops.reslayer_tail(x, w1t.t(), b1, None if w0t is None else w0t.t(), b0, w2t.t(),
                                  out=None if t_ is None else t_[3], scatter_rows=None if t_ is None else t_[0],
                                  valid_count=None if t_ is None else t_[1], per_group=0 if t_ is None else t_[2])
            li += 1
            if t_ is not None:
                assert c is None
                return x
            continue
        li += 1
        if tap_at is not None and li - 2 == tap_at and tapped is None:
            tapped = x                                     # (library-GEMM path: layer by layer; the next layer must not overwrite it)
            x = x.clone() if plan[li - 1][2] is None else x
        if (w0t is None and w1t.shape == (128, 128) and x.shape[1] == 128 and x.dtype == torch.float32 and x.is_contiguous()
                and not (li == 1 and keep_input)):
            # 128-wide identity-skip layer: both GEMMs, the bias, the ReLU and the residual add in one matrix-core kernel
            # that reads and writes the activation once (cppf_reslayer128); w1t / w2t are transposed views of the weights
            x = ops.reslayer128_(x, w1t.t(), b1, w2t.t())
            continue
        if x.shape[1] > w1t.shape[0]:               # zero-padded input columns (see BeyondCPPFDino.prepare_tuple_inputs)
            x = x[:, :w1t.shape[0]]
        h = torch._addmm_activation(b1, x, w1t)
        if w0t is not None:
            x = torch.addmm(b0, x, w0t)
        elif li == 1 and keep_input:
            x = torch.addmm(x, h, w2t)              # same GEMM, written to a new tensor: the caller's x survives
            continue
        x = x.addmm_(h, w2t)
    if c is not None:
        x = x.add_(c)
    return x if tap_at is None else (x, tapped)


def tail_supported(seq):
    """True when fused_stack(seq, ..., tail=...) can scatter the last layer's rows itself: split arithmetic, inference, a
    projection layer with at most 8 outputs and a multiple of 4 inputs last."""
    last = seq[len(seq) - 1]
    return (_kernel_arith() and not torch.is_grad_enabled() and last.fc0 is not None and last.fc1.out_features <= 8
            and last.fc1.in_features % 4 == 0)


def decode_supported(seq, x):
    """True when fused_stack(seq, x, decode=...) can draw the bins inside the stack's last layer: split arithmetic, inference,
    a 192-wide projection layer (6 x 32 bins) last, preceded by a projection or nothing pending (no carried bias offset)."""
    last = seq[len(seq) - 1]
    plan, c = _fused_plan(seq)
    return (_kernel_arith() and not torch.is_grad_enabled() and x.is_cuda and last.fc0 is not None
            and last.fc1.out_features == 192 and last.fc1.in_features % 8 == 0 and c is None)


def _scale_head_rows(model, feat, rows, scatter=None):
    """model.scale_head(feat[rows]) without materialising feat[rows]: the scale head's first layer (a 128-wide projection
    layer) reads the selected rows of `feat` through the gathering x-tile fetch of cppf_reslayer_split_gather (a table
    gather with no pair-feature block).  rows: int32 (or int64) [n] tuple rows.
    scatter = (counts int32 [n / per_group], per_group, out [T, 3]): the result rows are written to out[rows[i]] by the
    head's last kernel (padded entries of each group skipped) and `out` is returned -- no index_put, no torch kernel."""
    first = model.scale_encoder[0]
    f = feat.shape[1]
    if not (_kernel_arith() and not torch.is_grad_enabled() and feat.is_cuda and feat.is_contiguous()
            and first.fc0 is not None and first.fc1.out_features == 128 and first.fc1.in_features == f
            and f >= 8 and f & (f - 1) == 0 and feat.shape[0] < 2 ** 31):
        vals = model.scale_head(feat[rows.long()])
        if scatter is None:
            return vals
        counts, per_group, out = scatter
        keep = (torch.arange(rows.numel(), device=rows.device) % per_group) < counts.repeat_interleave(per_group)
        out[rows.long()[keep]] = vals[keep]
        return out
    gidx = (rows if rows.dtype == torch.int32 else rows.to(torch.int32)).reshape(-1, 1)
    tail = None
    if scatter is not None and tail_supported(model.scale_encoder):
        tail = (gidx.reshape(-1), scatter[0], scatter[1], scatter[2])
    vals = fused_stack(model.scale_encoder, None, gather=(feat[:, :0], gidx, feat), tail=tail)
    if scatter is not None and tail is None:
        counts, per_group, out = scatter
        keep = (torch.arange(rows.numel(), device=rows.device) % per_group) < counts.repeat_interleave(per_group)
        out[rows.long()[keep]] = vals[keep]
        return out
    return vals


class _EncodeShot(torch.autograd.Function):
    """HIP tuple encode with a backward for the per-point feature table (the only differentiable input:
    points and normals are data).  d feat[n] = sum over (tuple, slot) with idx == n of d out[:, slot block]."""

    @staticmethod
    def forward(ctx, points, idx, feat, normal):
        out = ops.encode_tuples_shot(points, idx, feat.detach(), normal)
        ctx.save_for_backward(idx)
        ctx.feat_shape = feat.shape
        return out

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        n, f = ctx.feat_shape
        k = idx.shape[1]
        head = k * (k - 1) // 2 * 4
        gf = torch.zeros((n, f), dtype=g.dtype, device=g.device)
        for s in range(k):
            gf.index_add_(0, idx[:, s].long(), g[:, head + s * f: head + (s + 1) * f])
        return None, None, gf, None


class BeyondCPPFShot(nn.Module):
    """train_shot.py:48-122.  forward(points, point_idxs_all, shot_feat, normal) -> (preds_cls [T,6,32], preds_scale [T,3])."""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        k = cfg.num_more + 2
        self.shot_encoder = _stack([352] + [128] * 5 + [64])
        input_dim = len(list(combinations(range(k), 2))) * 4 + k * 64
        self.tuple_encoder = _stack([input_dim] + [128] * 5 + [256])
        self.logit_encoder = _stack([256, 256, 256, 64 * 3])
        self.scale_encoder = _stack([256, 128, 64, 3])

    def prepare_tuple_inputs(self, points, point_idxs_all, shot_feat, normal):
        idx = point_idxs_all.to(torch.int32)
        if shot_feat.requires_grad:
            return _EncodeShot.apply(points, idx, shot_feat, normal)
        return ops.encode_tuples_shot(points, idx, shot_feat, normal)

    def heads(self, inputs, lazy_scale=False):
        """(preds_cls, preds_scale).  lazy_scale=True (inference) returns (preds_cls, feat) instead: the scale head's
        output is only ever read for the pairs that survive the back-vote filter (eval.py:272, ~10 % of the tuples), so a
        caller can run scale_head() on just those rows of `feat` later -- same rows through the same layers."""
        if not torch.is_grad_enabled() and inputs.is_cuda:
            # tuple encoder and logit head as one chain of launches; `feat` (the tuple encoder's output) is tapped for the scale head
            preds_cls, feat = fused_stack((self.tuple_encoder, self.logit_encoder), inputs)
            if lazy_scale:
                return preds_cls.reshape(feat.shape[0], 6, -1), feat
            preds_scale = fused_stack(self.scale_encoder, feat)      # first layer projects: feat is left intact
            return preds_cls.reshape(feat.shape[0], 6, -1), preds_scale
        feat = self.tuple_encoder(inputs)
        preds_scale = self.scale_encoder(feat)
        preds_cls = self.logit_encoder(feat).reshape(feat.shape[0], 6, -1)
        return preds_cls, preds_scale

    def gather_supported(self, feat_dim, k):
        """True when heads_from_tuples can feed the tuple encoder without materialising its input rows."""
        first = self.tuple_encoder[0]
        return (_kernel_arith() and not torch.is_grad_enabled() and first.fc0 is not None and first.fc1.out_features == 128
                and feat_dim >= 8 and feat_dim & (feat_dim - 1) == 0 and (k * (k - 1) // 2 * 4) % 8 == 0 and k <= 8
                and first.fc1.in_features == k * (k - 1) // 2 * 4 + k * feat_dim)

    def sum_supported(self, feat_dim, k):
        """True when heads_from_tuples(sum_tables=True) can feed the tuple encoder from per-point slot tables."""
        first = self.tuple_encoder[0]
        head = k * (k - 1) // 2 * 4
        return (_kernel_arith() and not torch.is_grad_enabled() and first.fc0 is not None and first.fc1.out_features == 128
                and feat_dim % 8 == 0 and head % 8 == 0 and k <= 8 and first.fc1.in_features == head + k * feat_dim)

    def first_layer_fold(self, feat_dim, k):
        """_FoldedFirstLayer of tuple_encoder[0] for rows [pair features | feat[idx_0] | ... | feat[idx_{k-1}]]
        (train_shot.py:75-83): slot i's table weights are the columns of fc1 / fc0 that multiply feat[idx_i] (no product to
        fold: the same multiplications as the row form, summed per slot first)."""
        first = self.tuple_encoder[0]
        stamp = tuple((q.data_ptr(), q._version) for q in first.parameters()) + (feat_dim, k)
        cached = getattr(self, "_fold_cache", None)
        if cached is None or cached.stamp != stamp:
            head = k * (k - 1) // 2 * 4
            w1, w0 = first.fc1.weight.detach(), first.fc0.weight.detach()
            cached = _FoldedFirstLayer(stamp, [w1[:, head + i * feat_dim: head + (i + 1) * feat_dim] for i in range(k)],
                                       [w0[:, head + i * feat_dim: head + (i + 1) * feat_dim] for i in range(k)],
                                       w1[:, :head], w0[:, :head])
            self._fold_cache = cached
        return cached

    def heads_from_tuples(self, points, point_idxs_all, feat, normal, pt_off=None, tup_off=None, lazy_scale=False, decode=None,
                          sum_tables=False):
        """heads(prepare_tuple_inputs(...)) with the [T, 360] tuple rows never written: the first ResLayer's kernel computes the
        pair features (40 columns) of its rows from points / normals / the sampler's indices and reads the per-point descriptors
        `feat` itself (cppf_reslayer_split_encode; same values, same arithmetic, bit-identical logits).  Falls back to the materialised rows when the first layer has
        no gathering kernel (other widths, training, native arithmetic).
        sum_tables=True: the descriptor columns' products with the first layer's weights are evaluated once per point and slot
        (first_layer_fold) and summed into the kernel's accumulators: the same multiplications in another order (float32-level
        differences against the row form), 20 of the first product's 23 K steps fewer per tuple.
        decode = (uniforms [T, 6], prior [T, 6, 32] | None, bins int32 [T, 6]): the bins of eval.py:225-229 are drawn by the
        logit head's output layer into `bins` (e.g. VotingPipeline.bins) and None is returned in place of the logits -- only when
        decode_supported(); otherwise the logits are returned as usual and the caller decodes them."""
        idx = point_idxs_all.to(torch.int32)
        if not (points.is_cuda and self.gather_supported(feat.shape[1], idx.shape[1])):
            return self.heads(ops.encode_tuples_shot(points, idx, feat, normal, pt_off, tup_off), lazy_scale=lazy_scale)
        tuples = ops.TupleSource(points, idx, normal, pt_off, tup_off)
        draw = decode if (decode is not None and decode_supported(self.logit_encoder, feat)) else None
        if sum_tables and self.sum_supported(feat.shape[1], idx.shape[1]):
            # the descriptor columns' share of the first products, once per point and slot instead of once per tuple
            fold = self.first_layer_fold(feat.shape[1], idx.shape[1])
            heads, gidx = tuples.heads()
            src = (heads, gidx, fold.tables(feat.contiguous()), fold)
        else:
            # (round 5: the pair features are built by the first launch itself -- no per-tuple array between sampler and encoder)
            src = (tuples, None, feat.contiguous())
        preds_cls, feat = fused_stack((self.tuple_encoder, self.logit_encoder), None, gather=src, decode=draw)
        second = feat if lazy_scale else fused_stack(self.scale_encoder, feat)
        if draw is not None:
            return None, second
        return preds_cls.reshape(feat.shape[0], 6, -1), second

    def scale_head(self, feat_rows):
        """scale_encoder on a subset of tuple features (rows of the `feat` heads(lazy_scale=True) returned)."""
        if not torch.is_grad_enabled() and feat_rows.is_cuda:
            return fused_stack(self.scale_encoder, feat_rows)
        return self.scale_encoder(feat_rows)

    def scale_head_rows(self, feat, rows, scatter=None):
        """scale_head(feat[rows]) on the kept pairs only: see _scale_head_rows."""
        return _scale_head_rows(self, feat, rows, scatter)

    def encode_points(self, shot_feat):
        """shot_encoder over the per-point descriptors (train_shot.py:118)."""
        if not torch.is_grad_enabled() and shot_feat.is_cuda:
            return fused_stack(self.shot_encoder, shot_feat)
        return self.shot_encoder(shot_feat)

    def forward(self, points, point_idxs_all, shot_feat, normal):
        inputs = self.prepare_tuple_inputs(points, point_idxs_all, self.encode_points(shot_feat), normal)
        return self.heads(inputs)


class BeyondCPPFDino(nn.Module):
    """train_dino.py:58-133.  forward(points, point_descs, point_idxs_all).
    desc_transform is applied per point BEFORE the gather: same Linear on the same rows,
    1/5 of the FLOPs at T >> N and no [T,5,1024] temporary."""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        k = cfg.num_more + 2
        d = 256
        self.ncoord = len(list(combinations(range(k), 2))) * 3
        self.tuple_encoder = _stack([self.ncoord + d] + [128] * 5 + [256])
        self.logit_encoder = _stack([256, 256, 256, 64 * 3])
        self.scale_encoder = _stack([256, 128, 64, 3])
        self.desc_transform = nn.Linear(1024, d)
        self.desc_pair_transform = nn.Linear(d * k, d)

    def prepare_tuple_inputs(self, points, point_descs, point_idxs_all):
        idx = point_idxs_all.to(torch.int32)
        T, k = idx.shape
        per_point = self.desc_transform(point_descs)                                   # [N,256]
        d = per_point.shape[1]
        if not torch.is_grad_enabled() and per_point.is_cuda:
            # Linear over the concatenation = sum of k per-point products: k GEMMs over N rows + a gather-add kernel
            # instead of a [T, k*256] gather and a K = k*256 GEMM over T rows (T/N = 5x the FLOPs, 6.5 GB at bench size)
            w = self.desc_pair_transform.weight                                        # [256, k*256]
            tables = torch.stack([per_point @ w[:, i * d:(i + 1) * d].t() for i in range(k)], 1).contiguous()
            # rows padded with zero columns to a multiple of 8 floats (286 -> 288): 16-byte aligned rows for the MLP kernels,
            # which give the extra columns zero weights (fused_stack)
            width = (self.ncoord + d + 7) // 8 * 8
            out = torch.empty((T, width), dtype=torch.float32, device=per_point.device)
            if width > self.ncoord + d:
                out[:, self.ncoord + d:] = 0
            ops.encode_tuples_coord(points, idx, out=out)
            ops.encode_tuples_dino(tables, self.desc_pair_transform.bias, idx, out, self.ncoord)
            return out
        gathered = per_point[idx.long().reshape(-1)].reshape(T, k * d)                 # cat_i desc_transform(desc[idx_i])
        desc_part = self.desc_pair_transform(gathered)
        coord = ops.encode_tuples_coord(points, idx)
        return torch.cat([coord, desc_part], -1)

    def heads(self, inputs, decode=None, lazy_scale=False):
        """(preds_cls [T,6,32], preds_scale [T,3]) from the tuple inputs (train_dino.py:130-132); inference on the GPU
        runs the stacks as matrix-core kernels (fused_stack), like the SHOT model.  decode: see
        BeyondCPPFShot.heads_from_tuples (None is returned in place of the logits when the bins were drawn);
        lazy_scale=True returns the tuple features in place of the scale head's output (see BeyondCPPFShot.heads)."""
        if not torch.is_grad_enabled() and inputs.is_cuda:
            draw = decode if (decode is not None and decode_supported(self.logit_encoder, inputs)) else None
            preds_cls, feat = fused_stack((self.tuple_encoder, self.logit_encoder), inputs, decode=draw)
            second = feat if lazy_scale else fused_stack(self.scale_encoder, feat)      # first layer projects: feat is left intact
            if draw is not None:
                return None, second
            return preds_cls.reshape(feat.shape[0], 6, -1), second
        feat = self.tuple_encoder(inputs)
        preds_scale = self.scale_encoder(feat)
        preds_cls = self.logit_encoder(feat).reshape(feat.shape[0], 6, -1)
        return preds_cls, preds_scale

    def sum_supported(self, k):
        """True when heads_from_tuples can run: every Linear of prepare_tuple_inputs on the library's matrix-core kernels and the
        tuple rows never formed (split arithmetic, inference, the reference's layer widths)."""
        first = self.tuple_encoder[0]
        d = self.desc_transform.out_features
        return (_kernel_arith() and not torch.is_grad_enabled() and first.fc0 is not None and first.fc1.out_features == 128
                and d == 256 and self.desc_transform.in_features % 8 == 0 and k <= 8
                and self.desc_pair_transform.in_features == k * d and self.desc_pair_transform.out_features == d
                and first.fc1.in_features == k * (k - 1) // 2 * 3 + d)

    def first_layer_fold(self, k):
        """_FoldedFirstLayer of tuple_encoder[0] for rows [coords | desc_pair_transform(cat_i q[idx_i])], q = desc_transform(desc)
        (train_dino.py:91-97): the Linear over the concatenation is a sum of k per-point products W_i q[idx_i] (W = [W_0 | ... ]),
        and the first ResLayer's fc1 / fc0 are linear too, so slot i's table weights are fc1[:, desc columns] W_i and
        fc0[:, desc columns] W_i (folded in float64, rounded to float32 once), desc_pair_transform's bias goes through the same
        columns into the layer's biases, and the layer's own products keep the 30 coordinate columns only."""
        first = self.tuple_encoder[0]
        params = list(first.parameters()) + list(self.desc_pair_transform.parameters())
        stamp = tuple((q.data_ptr(), q._version) for q in params) + (k,)
        cached = getattr(self, "_fold_cache", None)
        if cached is None or cached.stamp != stamp:
            nc, d = self.ncoord, self.desc_transform.out_features
            w1, w0 = first.fc1.weight.detach().double(), first.fc0.weight.detach().double()
            wp, bp = self.desc_pair_transform.weight.detach().double(), self.desc_pair_transform.bias.detach().double()
            cached = _FoldedFirstLayer(stamp, [w1[:, nc:] @ wp[:, i * d:(i + 1) * d] for i in range(k)],
                                       [w0[:, nc:] @ wp[:, i * d:(i + 1) * d] for i in range(k)], w1[:, :nc], w0[:, :nc],
                                       b1_add=(w1[:, nc:] @ bp).float(), b0_add=(w0[:, nc:] @ bp).float())
            self._fold_cache = cached
        return cached

    def transform_points(self, point_descs):
        """desc_transform over the per-point descriptors (train_dino.py:95; applied per point, before the gather) on the library's
        matrix-core Linear kernel: [N, 1024] -> [N, 256]."""
        lin = self.desc_transform
        stamp = (lin.weight.data_ptr(), lin.weight._version, lin.bias.data_ptr(), lin.bias._version, MLP_ARITH)
        cached = getattr(self, "_desc_cache", None)
        if cached is None or cached[0] != stamp:
            if MLP_ARITH == "split16":
                sc = f16_scale(lin.weight)
                cached = (stamp, pack_linear(lin.weight, lin.in_features, arith="f16x2", scale=sc), (lin.bias.detach() * sc).contiguous(), sc)
            else:
                cached = (stamp, pack_linear(lin.weight, lin.in_features), lin.bias.detach().contiguous(), None)
            self._desc_cache = cached
        return ops.linear_split(point_descs, cached[1], cached[2], lin.out_features, scale=cached[3])

    def point_tables(self, point_descs, k):
        """Per-point slot tables [N, k * 256] of the folded first layer from the raw descriptors [N, 1024]: two library launches
        per batch (desc_transform, then the k folded slot products), no BLAS call, no [N, k, 256] stack copy."""
        return self.first_layer_fold(k).tables(self.transform_points(point_descs.contiguous()))

    def heads_from_tuples(self, points, point_descs, point_idxs_all, pt_off=None, tup_off=None, lazy_scale=False, decode=None,
                          tables=None):
        """heads(prepare_tuple_inputs(...)) with the [T, 286] tuple rows never written and every layer a kernel of the library:
        coordinate columns + global point indices (ops.encode_tuples_coord_heads), per-point slot tables (point_tables; pass
        `tables` to reuse them), and the first ResLayer summing its tuple's table rows into its accumulators
        (ops.reslayer_split_sumgather).  point_idxs_all are scene-local with pt_off / tup_off given (batches), global otherwise.
        lazy_scale / decode as BeyondCPPFShot.heads_from_tuples.  Falls back to heads(prepare_tuple_inputs(...)) when the
        kernels do not cover the configuration (training, native arithmetic, other widths)."""
        idx = point_idxs_all.to(torch.int32)
        k = idx.shape[1]
        if not (points.is_cuda and self.sum_supported(k)):
            if pt_off is not None:
                base = torch.repeat_interleave(pt_off[:-1], (tup_off[1:] - tup_off[:-1]).long()).to(torch.int32)
                idx = idx + base[:, None]
            return self.heads(self.prepare_tuple_inputs(points, point_descs, idx), decode=decode, lazy_scale=lazy_scale)
        fold = self.first_layer_fold(k)
        if tables is None:
            tables = fold.tables(self.transform_points(point_descs.contiguous()))
        # (round 5: the coordinate columns are built by the first launch itself -- no per-tuple array between sampler and encoder)
        tuples = ops.TupleSource(points, idx, None, pt_off, tup_off)
        draw = decode if (decode is not None and decode_supported(self.logit_encoder, tables)) else None
        preds_cls, feat = fused_stack((self.tuple_encoder, self.logit_encoder), None, gather=(tuples, None, tables, fold), decode=draw)
        second = feat if lazy_scale else fused_stack(self.scale_encoder, feat)
        if draw is not None:
            return None, second
        return preds_cls.reshape(feat.shape[0], 6, -1), second

    def scale_head(self, feat_rows):
        """scale_encoder on a subset of tuple features (rows of the `feat` heads(lazy_scale=True) returned)."""
        if not torch.is_grad_enabled() and feat_rows.is_cuda:
            return fused_stack(self.scale_encoder, feat_rows)
        return self.scale_encoder(feat_rows)

    def scale_head_rows(self, feat, rows, scatter=None):
        """scale_head(feat[rows]) on the kept pairs only (eval.py:272 reads nothing else; it is THIS model's scale that the
        reference keeps, eval.py:308-310): see _scale_head_rows."""
        return _scale_head_rows(self, feat, rows, scatter)

    def forward(self, points, point_descs, point_idxs_all):
        return self.heads(self.prepare_tuple_inputs(points, point_descs, point_idxs_all))


def load_reference_checkpoint(model, path):
    """Loads a Lightning checkpoint written by the reference's trainer (train_shot.py:139): weights live under
    ckpt['state_dict'] with the same key names as this module."""
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    sd = ckpt.get("state_dict", ckpt)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    if missing or unexpected:
        raise RuntimeError("checkpoint key mismatch: missing=%s unexpected=%s" % (missing, unexpected))
    return model
