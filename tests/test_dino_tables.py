"""The DINO branch of the per-instance loop (eval.py:219-224 -> train_dino.py:91-97, 128-133) as library kernels only:
cppf_linear_split (desc_transform, the folded slot tables), cppf_encode_tuples_coord_heads and cppf_reslayer_split_sumgather
(the tuple encoder's first ResLayer starting from bias + the tuple's table rows: the [T, 286] rows are never formed), and the
same table form for the SHOT model's first layer.  CPU part: the weight-stream layout and the fold's algebra in float64;
GPU part: kernels against float64 and against the row form."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")


def _cfg():
    from cppf2_amd.config import load_config
    return load_config("config", "config", ["category=bottle"])


def test_pack_linear_layout_and_stream_size():
    from cppf2_amd import models, ops
    g = torch.Generator().manual_seed(3)
    for n, k, k_in in ((256, 1024, 1024), (1280, 256, 256), (1280, 64, 64), (512, 40, 48)):
        w = torch.randn(n, k, generator=g)
        q = models.pack_linear(w, k_in)
        assert q.dtype == torch.bfloat16 and q.numel() * 2 == ops._L.cppf_linear_split_stream_bytes(k_in, n)
        ks1 = (k_in + 15) // 16
        c = q.float().numpy().reshape(n // 256, ks1, 8, 3, 2, 32, 8).sum(3)        # [group, step, tile, g, i, j], hi + mid + lo
        got = np.zeros((n, ks1 * 16), np.float32)
        for grp in range(n // 256):
            for s_ in range(ks1):
                for g_ in range(2):
                    got[256 * grp:256 * (grp + 1), 16 * s_ + 8 * g_:16 * s_ + 8 * g_ + 8] = c[grp, s_, :, g_].reshape(256, 8)
        assert np.array_equal(got[:, :k], w.numpy()) and not got[:, k:].any()
    assert ops._L.cppf_linear_split_stream_bytes(64, 128) == -1                     # column groups of 256
    assert ops._L.cppf_linear_split_stream_bytes(0, 256) == -1


def test_folded_first_layers_are_the_same_network_in_float64():
    """tables[idx_i][i] summed + head columns == the first layer on the materialised rows (float64 evaluation of both forms)."""
    from cppf2_amd import models
    torch.manual_seed(0)
    cfg = _cfg()
    k = cfg.num_more + 2
    N, T = 50, 200
    idx = torch.randint(0, N, (T, k))
    # DINO
    net = models.BeyondCPPFDino(cfg).double()
    fold = net.first_layer_fold(k)
    assert fold.slots == k and fold.dp == 256 and fold.tab_w.shape == (k * 256, 256) and fold.w1_heads.shape == (128, 30)
    pts, desc = torch.randn(N, 3, dtype=torch.float64), torch.randn(N, 1024, dtype=torch.float64)
    x = torch.cat([torch.cat([pts[idx[:, i]] - pts[idx[:, j]] for i in range(k) for j in range(i + 1, k)], -1),
                   net.desc_pair_transform(torch.cat([net.desc_transform(desc[idx[:, i]]) for i in range(k)], -1))], -1)
    first = net.tuple_encoder[0]
    want = torch.cat([first.fc1(x), first.fc0(x)], -1)                              # pre-activation | skip (without fc2's bias)
    tab = (net.desc_transform(desc) @ fold.tab_w.double().t()).reshape(N, k, 256)
    got = torch.cat([first.fc1.bias + fold.b1_add.double(), first.fc0.bias + fold.b0_add.double()])[None].repeat(T, 1)
    for i in range(k):
        got = got + tab[idx[:, i], i]
    got = got + torch.cat([x[:, :30] @ fold.w1_heads.double().t(), x[:, :30] @ fold.w0_heads.double().t()], -1)
    assert (got - want).abs().max().item() < 1e-6 * want.abs().max().item()         # float32 rounding of the folded weights
    # SHOT
    net = models.BeyondCPPFShot(cfg).double()
    fold = net.first_layer_fold(64, k)
    assert fold.dp == 64 and fold.tab_w.shape == (k * 256, 64) and fold.w1_heads.shape == (128, 40) and fold.b1_add is None
    feat, heads = torch.randn(N, 64, dtype=torch.float64), torch.randn(T, 40, dtype=torch.float64)
    x = torch.cat([heads] + [feat[idx[:, i]] for i in range(k)], -1)
    first = net.tuple_encoder[0]
    want = torch.cat([first.fc1(x), first.fc0(x)], -1)
    tab = (feat @ fold.tab_w.double().t()).reshape(N, k, 256)
    got = torch.cat([first.fc1.bias, first.fc0.bias])[None].repeat(T, 1)
    for i in range(k):
        got = got + tab[idx[:, i], i]
    got = got + torch.cat([heads @ fold.w1_heads.double().t(), heads @ fold.w0_heads.double().t()], -1)
    assert (got - want).abs().max().item() < 1e-12 * want.abs().max().item()        # a re-slicing: nothing rounded
    # the fold follows the weights
    s0 = net.first_layer_fold(64, k)
    with torch.no_grad():
        net.tuple_encoder[0].fc1.weight.mul_(2.0)
    assert net.first_layer_fold(64, k) is not s0


@pytest.mark.gpu
def test_linear_split_matches_float64_like_a_float32_gemm():
    from cppf2_amd import models, ops
    dev = torch.device("cuda:0")
    for n, k in ((256, 1024), (1280, 256), (1280, 64), (256, 8), (512, 72)):
        g = torch.Generator().manual_seed(n + k)
        w = (torch.randn(n, k, generator=g) / k ** 0.5).to(dev)
        b = (torch.randn(n, generator=g) * 0.1).to(dev)
        wq = models.pack_linear(w, k)
        # the stream's size is the library's (the packer and the kernel agree on the layout), and the loaded library is this ABI
        from cppf2_amd import _lib
        assert ops.linear_split_supported(k, n) and wq.numel() * wq.element_size() == _lib.load().cppf_linear_split_stream_bytes(k, n)
        assert _lib.load().cppf_version() == _lib.ABI_VERSION
        for rows in (1, 31, 257, 3001, 20000):
            x = torch.randn(rows, k, device=dev, generator=torch.Generator(device=dev).manual_seed(rows))
            want = x.double() @ w.double().t() + b.double()
            got = ops.linear_split(x, wq, b, n)
            nat = torch.addmm(b, x, w.t())
            scale = want.abs().max().item()
            e_split = (got.double() - want).abs().max().item() / scale
            e_nat = (nat.double() - want).abs().max().item() / scale
            assert e_split < 2e-6 and e_split < 3.0 * e_nat + 2e-7, (n, k, rows, e_split, e_nat)
            if rows == 257:                          # no bias; rows independent of the batch; output inside a wider buffer
                assert torch.equal(ops.linear_split(x, wq, None, n) , ops.linear_split(x, wq, torch.zeros_like(b), n))
                assert torch.equal(ops.linear_split(x[:100], wq, b, n), got[:100])
                buf = torch.full((rows, n + 8), 7.0, device=dev)
                ops.linear_split(x, wq, b, n, out=buf[:, 4:4 + n])
                assert torch.equal(buf[:, 4:4 + n], got) and bool((buf[:, :4] == 7.0).all()) and bool((buf[:, 4 + n:] == 7.0).all())
        # f16x2 arithmetic
        sc = models.f16_scale(w)
        wq16 = models.pack_linear(w, k, arith="f16x2", scale=sc)
        x = torch.randn(1000, k, device=dev)
        want = x.double() @ w.double().t() + b.double()
        got = ops.linear_split(x, wq16, b * sc, n, scale=sc)
        assert (got.double() - want).abs().max().item() < 3e-6 * want.abs().max().item()
    with pytest.raises(Exception):
        ops.linear_split(torch.randn(4, 12, device=dev), wq, None, 256)          # k_in % 8


@pytest.mark.gpu
def test_coord_heads_are_the_coordinate_columns_bit_for_bit():
    from cppf2_amd import ops
    dev = torch.device("cuda:0")
    torch.manual_seed(2)
    for k in (5, 4):
        Ns, Ts = [300, 77, 512], [1000, 33, 2049]
        pts = torch.randn(sum(Ns), 3, device=dev)
        idx = torch.cat([torch.randint(0, n, (t, k), device=dev, dtype=torch.int32) for n, t in zip(Ns, Ts)])
        pt_off, tup_off = ops._offsets(Ns, dev), ops._offsets(Ts, dev)
        want = ops.encode_tuples_coord(pts, idx, pt_off=pt_off, tup_off=tup_off)
        heads, gidx = ops.encode_tuples_coord_heads(pts, idx, pt_off, tup_off)
        nc = 3 * k * (k - 1) // 2
        assert heads.shape[1] == (nc + 7) // 8 * 8 and torch.equal(heads[:, :nc], want) and not heads[:, nc:].any()
        base = torch.repeat_interleave(pt_off[:-1].long(), torch.tensor(Ts, device=dev))
        assert torch.equal(gidx.long(), idx.long() + base[:, None])


@pytest.mark.gpu
def test_dino_heads_from_tuples_equals_the_row_form_and_float64():
    """BeyondCPPFDino.heads_from_tuples (library kernels only, rows never formed) against the plain module in float64, next to
    the row form's error (prepare_tuple_inputs -> heads: library GEMMs for the per-point transforms + the materialised rows)."""
    import copy
    from cppf2_amd import models, ops
    dev = torch.device("cuda:0")
    torch.manual_seed(7)
    net = models.BeyondCPPFDino(_cfg()).to(dev).eval()
    Ns, Ts = [300, 77, 512], [1000, 33, 2049]
    pts = torch.randn(sum(Ns), 3, device=dev) * 0.1
    desc = torch.nn.functional.normalize(torch.randn(sum(Ns), 1024, device=dev), dim=-1)
    idx = torch.cat([torch.randint(0, n, (t, 5), device=dev, dtype=torch.int32) for n, t in zip(Ns, Ts)])
    pt_off, tup_off = ops._offsets(Ns, dev), ops._offsets(Ts, dev)
    base = torch.repeat_interleave(pt_off[:-1], torch.tensor(Ts, device=dev)).to(torch.int32)
    gl = idx + base[:, None]
    net64 = copy.deepcopy(net).double()
    with torch.no_grad():
        d64 = net64.desc_transform(desc.double())
        x64 = torch.cat([torch.cat([pts.double()[gl[:, i].long()] - pts.double()[gl[:, j].long()] for i in range(5) for j in range(i + 1, 5)], -1),
                         net64.desc_pair_transform(torch.cat([d64[gl[:, i].long()] for i in range(5)], -1))], -1)
        f64 = net64.tuple_encoder(x64)
        want_cls, want_sc = net64.logit_encoder(f64).reshape(-1, 6, 32), net64.scale_encoder(f64)
    prev = models.MLP_ARITH
    try:
        for arith in ("split", "split16"):
            models.MLP_ARITH = arith
            with torch.no_grad():
                assert net.sum_supported(5)
                cls_t, sc_t = net.heads_from_tuples(pts, desc, idx, pt_off, tup_off)
                cls_r, sc_r = net.heads(net.prepare_tuple_inputs(pts, desc, gl))
                s = want_cls.abs().max().item()
                e_t = (cls_t.double() - want_cls).abs().max().item() / s
                e_r = (cls_r.double() - want_cls).abs().max().item() / s
                assert e_t < 5e-6 and e_t < 3.0 * e_r + 5e-7, (arith, e_t, e_r)
                assert (sc_t.double() - want_sc).abs().max().item() < 5e-6 * max(1.0, want_sc.abs().max().item())
                # lazy scale head on selected rows; the same tables reused; global indices without offsets
                cls_l, feat = net.heads_from_tuples(pts, desc, idx, pt_off, tup_off, lazy_scale=True)
                assert torch.equal(cls_l, cls_t) and torch.equal(net.scale_head(feat), sc_t)
                pick = torch.randint(0, feat.shape[0], (777,), device=dev)
                assert torch.equal(net.scale_head_rows(feat, pick), net.scale_head(feat[pick]))
                tab = net.point_tables(desc, 5)
                assert tab.shape == (sum(Ns), 5 * 256)
                cls_g, _ = net.heads_from_tuples(pts, desc, gl, tables=tab)
                assert torch.equal(cls_g, cls_t)
                # the bin draw in the output layer == decode of the logits
                u = torch.rand(cls_t.shape[0], 6, device=dev)
                bins = torch.empty((cls_t.shape[0], 6), dtype=torch.int32, device=dev)
                none, _ = net.heads_from_tuples(pts, desc, idx, pt_off, tup_off, decode=(u, None, bins), tables=tab)
                assert none is None
                want_bins = ops.decode_bins(cls_t.contiguous(), u, pts, idx, [0, 1, 0], [0, 0, 1], [1, 0, 0], pt_off, tup_off)["bins"]
                assert torch.equal(bins, want_bins)
        models.MLP_ARITH = "native"                      # no table kernel there: the row form
        with torch.no_grad():
            assert not net.sum_supported(5)
            cls_n, _ = net.heads_from_tuples(pts, desc, idx, pt_off, tup_off)
        assert (cls_n.double() - want_cls).abs().max().item() < 2e-5 * want_cls.abs().max().item()
    finally:
        models.MLP_ARITH = prev


@pytest.mark.gpu
def test_shot_first_layer_from_slot_tables_equals_the_gathered_form():
    from cppf2_amd import models, ops
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    Ns, Ts = [300, 77, 512], [1000, 33, 2049]
    pts = torch.randn(sum(Ns), 3, device=dev)
    nrm = torch.nn.functional.normalize(torch.randn(sum(Ns), 3, device=dev), dim=-1)
    feat = torch.randn(sum(Ns), 64, device=dev)
    idx = torch.cat([torch.randint(0, n, (t, 5), device=dev, dtype=torch.int32) for n, t in zip(Ns, Ts)])
    pt_off, tup_off = ops._offsets(Ns, dev), ops._offsets(Ts, dev)
    net = models.BeyondCPPFShot(_cfg()).to(dev).eval()
    prev = models.MLP_ARITH
    try:
        for arith in ("split", "split16"):
            models.MLP_ARITH = arith
            with torch.no_grad():
                assert net.sum_supported(64, 5)
                cls_a, sc_a = net.heads_from_tuples(pts, idx, feat, nrm, pt_off, tup_off)
                cls_b, sc_b = net.heads_from_tuples(pts, idx, feat, nrm, pt_off, tup_off, sum_tables=True)
            s = cls_a.abs().max().item()
            assert (cls_a - cls_b).abs().max().item() < 4e-6 * s and (sc_a - sc_b).abs().max().item() < 4e-6 * max(1.0, sc_a.abs().max().item())
    finally:
        models.MLP_ARITH = prev
