"""cppf_reslayer_split: the ResLayers of the tuple / point MLPs (train_shot.py:19-45) on the bf16 matrix cores in
split-float32 arithmetic (exact bf16 triples, six products).  CPU part: the exact three-way split and the documented weight-stream order;
GPU part: the kernel against a float64 evaluation, next to the library float32 path's error against the same."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

SHAPES = [(128, 128, False), (360, 128, True), (128, 256, True), (256, 256, False), (256, 192, True), (352, 128, True),
          (128, 64, True), (256, 128, True), (64, 64, False), (288, 128, True), (192, 192, False),
          (72, 64, True), (200, 128, True), (40, 256, True)]        # K steps not a multiple of the staged chunk


def _layer(k, n, proj, dev, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    w1 = (torch.randn(n, k, generator=g) / k ** 0.5).to(dev)
    w2 = (torch.randn(n, n, generator=g) / n ** 0.5).to(dev)
    w0 = (torch.randn(n, k, generator=g) / k ** 0.5).to(dev) if proj else None
    b1 = (torch.randn(n, generator=g) * 0.1).to(dev)
    b0 = (torch.randn(n, generator=g) * 0.1).to(dev) if proj else None
    return w1, b1, w0, b0, w2


def _ref64(x, w1, b1, w0, b0, w2):
    x = x.double()
    h = torch.relu(x @ w1.double().t() + b1.double())
    skip = x if w0 is None else x @ w0.double().t() + b0.double()
    return skip + h @ w2.double().t()


def test_split_is_exact_and_stream_order_is_the_documented_one():
    from cppf2_amd import models, ops
    g = torch.Generator().manual_seed(1)
    w = torch.randn(64, 40, generator=g) * torch.logspace(-6, 6, 40)[None]
    s = models.split_bf16(w)
    assert s.dtype == torch.bfloat16 and s.shape == (3, 64, 40)
    assert torch.equal((s[0].float() + s[1].float()) + s[2].float(), w)             # exact: 8 + 8 + 8 significand bits
    assert torch.equal(s[0], w.to(torch.bfloat16))
    for k, n, proj in SHAPES:
        w1, b1, w0, b0, w2 = _layer(k, n, proj, "cpu", seed=k + n)
        k_in = k + (8 if k == 288 else 0)                                           # x wider than dim_in: zero weights
        q = models.pack_split(w1, w0, w2, k_in)
        assert q.dtype == torch.bfloat16
        assert q.numel() * 2 == ops._L.cppf_reslayer_split_stream_bytes(k_in, n, int(proj), 0)
        # walk the stream the way pack_split's docstring describes it and rebuild the three matrices
        nt = n // 32
        ks1 = (k_in + 15) // 16
        f = q.float().numpy()
        pos = 0

        def take(tiles, steps):
            nonlocal pos
            c = f[pos:pos + steps * tiles * 3 * 512].reshape(steps, tiles, 3, 2, 32, 8)   # [step, tile, slice, g, i, j]
            pos += c.size
            return c.sum(2)                                                                # hi + mid + lo
        t0 = 2 * nt if proj else nt                   # projection layer: W0's tiles ride behind W1's
        c1 = take(t0, ks1)
        got1 = np.zeros((32 * t0, ks1 * 16), np.float32)
        for s_ in range(ks1):
            for g_ in range(2):
                got1[:, 16 * s_ + 8 * g_:16 * s_ + 8 * g_ + 8] = c1[s_, :, g_].reshape(32 * t0, 8)
        assert np.array_equal(got1[:n, :k], w1.numpy()) and not got1[:, k:].any()
        if proj:
            assert np.array_equal(got1[n:, :k], w0.numpy())
        c2 = take(nt, 2 * nt)
        got2 = np.zeros((n, n), np.float32)
        for t in range(nt):
            for sp in range(2):
                for g_ in range(2):
                    for j in range(8):
                        got2[:, 32 * t + 16 * sp + 4 * g_ + (j & 3) + 8 * (j >> 2)] = c2[2 * t + sp, :, g_, :, j].reshape(-1)
        assert pos == f.size and np.array_equal(got2, w2.numpy())
    assert ops._L.cppf_reslayer_split_stream_bytes(128, 96 + 1, 0, 0) == -1
    assert ops._L.cppf_reslayer_split_stream_bytes(128, 320, 0, 0) == -1
    assert ops._L.cppf_reslayer_split_stream_bytes(256, 192, 1, 16) == -1           # at most 15 chained layers
    # chained identity layers: their W1 and W2 follow, both in accumulator feature order
    w1, b1, w0, b0, w2 = _layer(72, 64, True, "cpu", seed=5)
    wa, _, _, _, wb = _layer(64, 64, False, "cpu", seed=6)
    q0 = models.pack_split(w1, w0, w2, 72)
    q = models.pack_split(w1, w0, w2, 72, chain=[(wa, wb)])
    assert q.numel() * 2 == ops._L.cppf_reslayer_split_stream_bytes(72, 64, 1, 1) and torch.equal(q[:q0.numel()], q0)
    tail = q[q0.numel():].float().numpy().reshape(2, 4, 2, 3, 2, 32, 8).sum(3)      # [matrix, step, tile, g, i, j]
    for m, want in enumerate((wa.numpy(), wb.numpy())):
        got = np.zeros((64, 64), np.float32)
        for st in range(4):
            for g_ in range(2):
                for j in range(8):
                    got[:, 32 * (st >> 1) + 16 * (st & 1) + 4 * g_ + (j & 3) + 8 * (j >> 2)] = tail[m, st, :, g_, :, j].reshape(-1)
        assert np.array_equal(got, want)


@pytest.mark.gpu
def test_reslayer_split_matches_float64_like_a_float32_gemm():
    from cppf2_amd import models, ops
    dev = torch.device("cuda:0")
    worst = 0.0
    for k, n, proj in SHAPES:
        w1, b1, w0, b0, w2 = _layer(k, n, proj, dev, seed=k * 7 + n)
        wq = models.pack_split(w1, w0, w2, k)
        for rows in (1, 31, 257, 3001):
            x = torch.randn(rows, k, device=dev, generator=torch.Generator(device=dev).manual_seed(rows))
            want = _ref64(x, w1, b1, w0, b0, w2)
            got = ops.reslayer_split(x.clone(), wq, b1, b0, n)
            h = torch._addmm_activation(b1, x, w1.t())
            nat = torch.addmm(x if w0 is None else torch.addmm(b0, x, w0.t()), h, w2.t())
            scale = want.abs().max().item()
            e_split = (got.double() - want).abs().max().item() / scale
            e_nat = (nat.double() - want).abs().max().item() / scale
            # float32-GEMM-level: a few units of 2^-24 of the largest output, and no worse than 3x what the library's
            # float32 GEMMs (f32-input matrix cores) leave against the same float64 evaluation
            assert e_split < 2e-6, (k, n, proj, rows, e_split)
            assert e_split < 3.0 * e_nat + 2e-7, (k, n, proj, rows, e_split, e_nat)
            worst = max(worst, e_split / max(e_nat, 1e-9))
    # a projection layer with identity layers chained behind it in the same kernel (the tuple / point encoders' shape)
    for k, n, proj, chain in ((360, 128, True, 4), (352, 128, True, 4), (128, 128, False, 3), (64, 64, False, 1), (128, 64, True, 2),
                              (128, 256, True, 2), (256, 256, False, 1), (256, 192, True, 1)):
        w1, b1, w0, b0, w2 = _layer(k, n, proj, dev, seed=11)
        rest = [_layer(n, n, False, dev, seed=20 + l) for l in range(chain)]
        wq = models.pack_split(w1, w0, w2, k, chain=[(e[0], e[4]) for e in rest])
        bias = torch.cat([b1] + [e[1] for e in rest])
        for rows in (5, 700):
            x = torch.randn(rows, k, device=dev)
            want = _ref64(x, w1, b1, w0, b0, w2)
            nat = torch.addmm(x if w0 is None else torch.addmm(b0, x, w0.t()), torch._addmm_activation(b1, x, w1.t()), w2.t())
            for e in rest:
                want = _ref64(want, e[0], e[1], None, None, e[4])
                nat = torch.addmm(nat, torch._addmm_activation(e[1], nat, e[0].t()), e[4].t())
            got = ops.reslayer_split(x.clone(), wq, bias, b0, n, chain=chain)
            scale = want.abs().max().item()
            e_split = (got.double() - want.double()).abs().max().item() / scale
            e_nat = (nat.double() - want.double()).abs().max().item() / scale
            assert e_split < 3e-6 and e_split < 3.0 * e_nat + 2e-7, (k, n, chain, rows, e_split, e_nat)
    # K at the edges of the supported range (one half-step; a long contraction with a ragged last step), output rows inside
    # a wider buffer (row stride > n_out) whose other columns stay untouched
    for k, n, proj in ((8, 64, True), (1032, 128, True), (520, 256, True)):
        w1, b1, w0, b0, w2 = _layer(k, n, proj, dev, seed=3)
        wq = models.pack_split(w1, w0, w2, k)
        x = torch.randn(777, k, device=dev)
        buf = torch.full((777, n + 12), 7.0, device=dev)
        got = ops.reslayer_split(x, wq, b1, b0, n, out=buf[:, 4:4 + n])
        want = _ref64(x, w1, b1, w0, b0, w2)
        assert (got.double() - want).abs().max().item() < 3e-6 * want.abs().max().item()
        assert bool((buf[:, :4] == 7.0).all()) and bool((buf[:, 4 + n:] == 7.0).all())
    # in place on a strided view (row stride > k), rows of the parent beyond the view untouched; NaN stays NaN through relu
    k = n = 128
    w1, b1, w0, b0, w2 = _layer(k, n, False, dev)
    wq = models.pack_split(w1, None, w2, k)
    buf = torch.randn(300, 160, device=dev)
    keep = buf.clone()
    view = buf[:290, 16:144]
    want = _ref64(view, w1, b1, None, None, w2)
    ops.reslayer_split(view, wq, b1, None, n)
    assert (view.double() - want).abs().max().item() < 2e-6 * want.abs().max().item()
    assert torch.equal(buf[290:], keep[290:]) and torch.equal(buf[:, :16], keep[:, :16]) and torch.equal(buf[:, 144:], keep[:, 144:])
    x = torch.randn(64, 128, device=dev)
    x[3, 5] = float("nan")
    got = ops.reslayer_split(x.clone(), wq, b1, None, n)
    assert torch.isnan(got[3]).all() and not torch.isnan(got[[0, 1, 2, 4]]).any()
    # out of place for an identity layer (fused_stack's keep_input) leaves x alone
    x = torch.randn(100, 128, device=dev)
    x0 = x.clone()
    out = ops.reslayer_split(x, wq, b1, None, n, out=torch.empty_like(x))
    assert torch.equal(x, x0) and torch.equal(out, ops.reslayer_split(x.clone(), wq, b1, None, n))
    # rows are independent of the batch they sit in (no split-K, no M-dependent tiling): bit-identical
    assert torch.equal(out[:37], ops.reslayer_split(x[:37].clone(), wq, b1, None, n))


@pytest.mark.gpu
def test_reslayer_split_rejects_bad_arguments():
    from cppf2_amd import models, ops, _lib
    dev = torch.device("cuda:0")
    w1, b1, w0, b0, w2 = _layer(128, 128, False, dev)
    wq = models.pack_split(w1, None, w2, 128)
    x = torch.randn(8, 128, device=dev)
    with pytest.raises(_lib.CppfError):
        ops.reslayer_split(x, wq[:-8], b1, None, 128)                       # stream size mismatch
    with pytest.raises(_lib.CppfError):
        ops.reslayer_split(x, wq, b1[:96], None, 96, out=torch.empty(8, 96, device=dev))   # unsupported width
    with pytest.raises(_lib.CppfError):
        ops.reslayer_split(torch.randn(8, 132, device=dev), wq, b1, None, 128, out=torch.empty(8, 128, device=dev))   # k_in % 8
    assert not ops.reslayer_split_supported(132, 128, False) and ops.reslayer_split_supported(360, 128, True, 4)
    assert ops.reslayer_split_supported(128, 256, True, 2) and not ops.reslayer_split_supported(192, 192, False, 16)


@pytest.mark.gpu
def test_models_split_arithmetic_matches_native_and_module_forward():
    """The inference stacks in split arithmetic against the f32-input path and the plain nn.Module forward
    (train_shot.py:100-122, train_dino.py:118-133): same network, float32-level differences only."""
    from cppf2_amd import models
    from cppf2_amd.config import load_config
    dev = torch.device("cuda:0")
    cfg = load_config("config", "config", ["category=bottle"])
    torch.manual_seed(3)
    shot = models.BeyondCPPFShot(cfg).to(dev).eval()
    dino = models.BeyondCPPFDino(cfg).to(dev).eval()
    x = torch.randn(5000, 360, device=dev)
    sf = torch.randn(3000, 352, device=dev)
    xd = torch.randn(5000, dino.ncoord + 256, device=dev)
    prev = models.MLP_ARITH
    try:
        outs = {}
        for mode in ("split", "native"):
            models.MLP_ARITH = mode
            with torch.no_grad():
                outs[mode] = (shot.heads(x.clone()), shot.heads(x.clone(), lazy_scale=True), shot.encode_points(sf.clone()),
                              dino.heads(xd.clone()), dino.heads(torch.nn.functional.pad(xd, (0, 2))))   # 286 -> 288 zero columns
        with torch.no_grad():
            feat = shot.tuple_encoder(x)
            ref = (shot.logit_encoder(feat).reshape(-1, 6, 32), shot.scale_encoder(feat), shot.shot_encoder(sf))
            featd = dino.tuple_encoder(xd)
            refd = (dino.logit_encoder(featd).reshape(-1, 6, 32), dino.scale_encoder(featd))
    finally:
        models.MLP_ARITH = prev
    for mode in ("split", "native"):
        (cls, sc), (cls_l, feat_l), pts, (dcls, dsc), (pcls, psc) = outs[mode]
        for got, want in ((cls, ref[0]), (sc, ref[1]), (cls_l, ref[0]), (pts, ref[2]), (dcls, refd[0]), (dsc, refd[1]),
                          (pcls, refd[0]), (psc, refd[1])):
            assert got.shape == want.shape
            assert (got - want).abs().max().item() < 2e-5 * max(1.0, want.abs().max().item()), mode
        assert (shot.scale_head(feat_l) - ref[1]).abs().max().item() < 2e-5


@pytest.mark.gpu
def test_gathered_first_layer_is_bit_identical_to_the_materialised_rows():
    """cppf_encode_tuples_shot_heads + cppf_reslayer_split_gather (the tuple rows of train_shot.py:75-83 never written)
    against cppf_encode_tuples_shot + cppf_reslayer_split: same logits bit for bit, on a ragged batch."""
    from cppf2_amd import models, ops
    from cppf2_amd.config import load_config
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    Ns, Ts = [300, 77, 512], [1000, 33, 2049]
    pts = torch.randn(sum(Ns), 3, device=dev)
    nrm = torch.nn.functional.normalize(torch.randn(sum(Ns), 3, device=dev), dim=-1)
    feat = torch.randn(sum(Ns), 64, device=dev)
    idx = torch.cat([torch.randint(0, n, (t, 5), device=dev, dtype=torch.int32) for n, t in zip(Ns, Ts)])
    pt_off, tup_off = ops._offsets(Ns, dev), ops._offsets(Ts, dev)
    rows = ops.encode_tuples_shot(pts, idx, feat, nrm, pt_off, tup_off)
    heads, gidx = ops.encode_tuples_shot_heads(pts, idx, nrm, pt_off, tup_off)
    assert torch.equal(heads, rows[:, :40])
    base = torch.repeat_interleave(pt_off[:-1].long(), torch.tensor(Ts, device=dev))
    assert torch.equal(gidx.long(), idx.long() + base[:, None])
    for j in range(5):
        assert torch.equal(feat[gidx[:, j].long()], rows[:, 40 + 64 * j:104 + 64 * j])
    net = models.BeyondCPPFShot(load_config("config", "config", ["category=bottle"])).to(dev).eval()
    prev = models.MLP_ARITH
    try:
        models.MLP_ARITH = "split"
        with torch.no_grad():
            assert net.gather_supported(64, 5)
            cls_a, sc_a = net.heads(rows.clone())
            cls_b, sc_b = net.heads_from_tuples(pts, idx, feat, nrm, pt_off, tup_off)
            assert torch.equal(cls_a, cls_b) and torch.equal(sc_a, sc_b)
            cls_c, feat_c = net.heads_from_tuples(pts, idx, feat, nrm, pt_off, tup_off, lazy_scale=True)
            assert torch.equal(cls_c, cls_a) and torch.equal(net.scale_head(feat_c), sc_a)
            # the scale head on selected rows, gathered by its first layer's kernel instead of feat[rows]
            pick = torch.randint(0, feat_c.shape[0], (1234,), device=dev)
            assert torch.equal(net.scale_head_rows(feat_c, pick), net.scale_head(feat_c[pick]))
            models.MLP_ARITH = "native"                      # no gathering kernel there: falls back to the materialised rows
            assert not net.gather_supported(64, 5)
            cls_n, _ = net.heads_from_tuples(pts, idx, feat, nrm, pt_off, tup_off)
            assert (cls_n - cls_a).abs().max().item() < 2e-5 * max(1.0, cls_a.abs().max().item())
    finally:
        models.MLP_ARITH = prev


@pytest.mark.gpu
def test_pair_features_built_inside_the_first_launch(small):
    """cppf_reslayer_split_encode (round 5: prepare_tuple_inputs, train_shot.py:75-83, and the tuple encoder's first launch in ONE
    kernel -- the 40 pair features are computed by the kernel's lanes and go straight into its x tiles) against the two-kernel form
    (cppf_encode_tuples_shot_heads + cppf_reslayer_split_gather) and against the materialised rows: the same outputs bit for bit.
    Cases: the reference's golden tuples (small_encode_shot: its rows ARE the reference's prepare_tuple_inputs output), a ragged
    batch with one-row and empty-ish scenes, and a batch large enough for every workgroup to walk several row blocks (XCD block map)."""
    from cppf2_amd import models, ops
    dev = torch.device("cuda:0")
    torch.manual_seed(11)
    g = torch.Generator(device="cpu").manual_seed(3)
    w1 = (torch.randn(128, 360, generator=g) / 360 ** 0.5).to(dev)
    w0 = (torch.randn(128, 360, generator=g) / 360 ** 0.5).to(dev)
    w2 = (torch.randn(128, 128, generator=g) / 128 ** 0.5).to(dev)
    chain = [((torch.randn(128, 128, generator=g) / 128 ** 0.5).to(dev), (torch.randn(128, 128, generator=g) / 128 ** 0.5).to(dev))
             for _ in range(2)]
    b1 = torch.randn(3 * 128, generator=g).to(dev)
    b0 = torch.randn(128, generator=g).to(dev)
    wq = models.pack_split(w1, w0, w2, 360, chain=chain)

    def both(pts, nrm, feat, idx, Ns, Ts):
        pt_off, tup_off = ops._offsets(Ns, dev), ops._offsets(Ts, dev)
        heads, gidx = ops.encode_tuples_shot_heads(pts, idx, nrm, pt_off, tup_off)
        want = ops.reslayer_split_gather(heads, gidx, feat, wq, b1, b0, 128, chain=2)
        src = ops.TupleSource(pts, idx, nrm, pt_off, tup_off)
        assert src.B == len(Ns) and src.shape == (idx.shape[0], 40)
        got = ops.reslayer_split_encode(src, feat, wq, b1, b0, 128, chain=2)
        assert got.shape == want.shape and torch.equal(got, want)
        return got, heads

    # the reference's own rows
    pts, nrm, feat = (torch.from_numpy(small[k]).to(dev) for k in ("small_pc", "small_normal", "small_feat"))
    idx = torch.from_numpy(small["small_idx"]).to(dev, torch.int32)
    got, heads = both(pts, nrm, feat, idx, [256], [512])
    assert np.array_equal(heads.cpu().numpy(), small["small_encode_shot"][:, :40])
    rows = torch.from_numpy(small["small_encode_shot"]).to(dev)
    assert torch.equal(got, ops.reslayer_split(rows, wq, b1, b0, 128, chain=2))
    # ragged: scenes of 1 row, 255 / 256 / 257 rows (row-block edges), one point
    Ns, Ts = [300, 1, 77, 512, 5], [255, 1, 256, 257, 33]
    pts = torch.randn(sum(Ns), 3, device=dev)
    nrm = torch.nn.functional.normalize(torch.randn(sum(Ns), 3, device=dev), dim=-1)
    nrm[7] = 0.0                                                       # a NaN-cleaned normal (eval.py:216)
    feat = torch.randn(sum(Ns), 64, device=dev)
    idx = torch.cat([torch.randint(0, n, (t, 5), device=dev, dtype=torch.int32) for n, t in zip(Ns, Ts)])
    both(pts, nrm, feat, idx, Ns, Ts)
    # many row blocks per workgroup: 40 scenes x 20 000 tuples = 3 125 row blocks over 256 workgroups
    Ns, Ts = [1024] * 40, [20000] * 40
    pts = torch.randn(sum(Ns), 3, device=dev)
    nrm = torch.nn.functional.normalize(torch.randn(sum(Ns), 3, device=dev), dim=-1)
    feat = torch.randn(sum(Ns), 64, device=dev)
    idx = torch.randint(0, 1024, (sum(Ts), 5), device=dev, dtype=torch.int32)
    both(pts, nrm, feat, idx, Ns, Ts)
    # unsupported shapes are refused, not approximated
    assert not ops.reslayer_split_encode_supported(4, 128) and not ops.reslayer_split_encode_supported(5, 256)


@pytest.mark.gpu
def test_coordinate_columns_built_inside_the_table_fed_launch():
    """cppf_reslayer_split_sumencode (the DINO model's first launch building its 30 coordinate differences itself, train_dino.py:92)
    against cppf_encode_tuples_coord_heads + cppf_reslayer_split_sumgather: the same outputs bit for bit, ragged batch and a batch
    with many row blocks per workgroup; through the model: heads_from_tuples == the row form's logits as before."""
    from cppf2_amd import models, ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(4)
    w1 = (torch.randn(128, 32, generator=g) / 32 ** 0.5).to(dev)
    w0 = (torch.randn(128, 32, generator=g) / 32 ** 0.5).to(dev)
    w2 = (torch.randn(128, 128, generator=g) / 128 ** 0.5).to(dev)
    chain = [((torch.randn(128, 128, generator=g) / 128 ** 0.5).to(dev), (torch.randn(128, 128, generator=g) / 128 ** 0.5).to(dev))]
    b1 = torch.randn(2 * 128, generator=g).to(dev)
    b0 = torch.randn(128, generator=g).to(dev)
    wq = models.pack_split(w1, w0, w2, 32, chain=chain)
    for Ns, Ts in (([300, 1, 77, 512], [255, 1, 256, 257]), ([1024] * 24, [20000] * 24)):
        pt_off, tup_off = ops._offsets(Ns, dev), ops._offsets(Ts, dev)
        pts = torch.randn(sum(Ns), 3, device=dev)
        tables = torch.randn(sum(Ns), 5 * 256, device=dev)
        idx = torch.cat([torch.randint(0, n, (t, 5), device=dev, dtype=torch.int32) for n, t in zip(Ns, Ts)])
        heads, gidx = ops.encode_tuples_coord_heads(pts, idx, pt_off, tup_off)
        want = ops.reslayer_split_sumgather(heads, gidx, tables, wq, b1, b0, 128, chain=1)
        src = ops.TupleSource(pts, idx, None, pt_off, tup_off)
        assert src.shape == (sum(Ts), 32) and src.nrm is None
        h2, g2 = src.heads()
        assert torch.equal(h2, heads) and torch.equal(g2, gidx)
        got = ops.reslayer_split_sumencode(src, tables, wq, b1, b0, 128, chain=1)
        assert torch.equal(got, want)


@pytest.mark.gpu
def test_bin_draw_fused_into_the_output_layer_equals_decode_bins():
    """cppf_reslayer_split_decode + cppf_decode_from_bins against cppf_reslayer_split + cppf_decode_bins (eval.py:225-240):
    the same bins and vote parameters bit for bit, with and without a logit prior, on a ragged batch."""
    from cppf2_amd import models, ops
    from cppf2_amd.config import load_config
    from cppf2_amd.pipeline import VotingPipeline
    dev = torch.device("cuda:0")
    torch.manual_seed(9)
    Ns, Ts = [200, 333], [1500, 777]
    T = sum(Ts)
    pts = torch.randn(sum(Ns), 3, device=dev) * 0.1
    idx = torch.cat([torch.randint(0, n, (t, 5), device=dev, dtype=torch.int32) for n, t in zip(Ns, Ts)])
    net = models.BeyondCPPFShot(load_config("config", "config", ["category=bottle"])).to(dev).eval()
    feat = torch.randn(T, 256, device=dev)
    u = torch.rand(T, 6, device=dev)
    prior = torch.randn(T, 6, 32, device=dev) * 3
    prev = models.MLP_ARITH
    try:
        models.MLP_ARITH = "split"
        with torch.no_grad():
            assert models.decode_supported(net.logit_encoder, feat)
            logits = models.fused_stack(net.logit_encoder, feat, keep_input=True).reshape(T, 6, 32)
            for pr in (None, prior):
                a = VotingPipeline(Ns, Ts, num_rots=36)
                a.decode(pts, idx, logits, u, prior=pr)
                b = VotingPipeline(Ns, Ts, num_rots=36)
                bins = models.fused_stack(net.logit_encoder, feat, keep_input=True, decode=(u, pr, b.bins))
                assert bins.data_ptr() == b.bins.data_ptr()
                b.decode_from_bins(pts, idx)
                for name in ("bins", "scaled", "scale", "tr", "rot"):
                    assert torch.equal(getattr(a, name), getattr(b, name)), name
            assert len(torch.unique(a.bins)) > 8                      # the draw is not degenerate
            models.MLP_ARITH = "native"
            assert not models.decode_supported(net.logit_encoder, feat)
    finally:
        models.MLP_ARITH = prev


@pytest.mark.gpu
def test_counted_wait_protocol_race_screen():
    """The split kernels keep LDS-DMA tiles in flight across raw s_barriers and wait on hand-computed vmcnt immediates: a slip
    there is a race that shows up only under memory pressure or at another grid size.  Every instantiation (widths 64 / 128 /
    192 / 256, identity and projection first layer, the gathering and the bin-drawing kernels; chains 0 / 1 / 4) is run 30 times
    beside a saturating copy stream with the number of persistent workgroups forced to 1, 7 and one per CU: every output must be
    bit-identical to the solo run at the default grid (rows do not depend on which workgroup computes them)."""
    import torch
    from cppf2_amd import _lib, models, ops
    L = _lib.load()
    dev = torch.device("cuda")
    hog_a = torch.empty(192 << 20, dtype=torch.uint8, device=dev)
    hog_b = torch.empty_like(hog_a)
    side = torch.cuda.Stream()
    g = torch.Generator(device="cpu").manual_seed(11)
    mk = lambda *s: torch.randn(*s, generator=g).to(dev)

    def layer(k, n, proj, chain):
        w1, w2 = mk(n, k) / k ** 0.5, mk(n, n) / n ** 0.5
        w0 = mk(n, k) / k ** 0.5 if proj else None
        rest = [(mk(n, n) / n ** 0.5, mk(n, n) / n ** 0.5) for _ in range(chain)]
        return models.pack_split(w1, w0, w2, k, chain=rest), mk((1 + chain) * n) * 0.1, (mk(n) * 0.1 if proj else None)

    def layer16(k, n, proj, chain):
        w1, w2 = mk(n, k) / k ** 0.5, mk(n, n) / n ** 0.5
        w0 = mk(n, k) / k ** 0.5 if proj else None
        rest = [(mk(n, n) / n ** 0.5, mk(n, n) / n ** 0.5) for _ in range(chain)]
        sc = models.f16_scale(w1, w0, w2, *[w for p in rest for w in p])
        return (models.pack_split(w1, w0, w2, k, chain=rest, arith="f16x2", scale=sc), mk((1 + chain) * n) * 0.1 * sc,
                (mk(n) * 0.1 * sc if proj else None), sc)

    cases = []
    for n in (64, 128, 192, 256):
        for proj in (False, True):
            for chain in (0, 1, 4):
                k = n if not proj else {64: 72, 128: 360, 192: 256, 256: 128}[n]
                wq, b1, b0 = layer(k, n, proj, chain)
                cases.append(("n%d proj%d chain%d" % (n, proj, chain),
                              lambda x, wq=wq, b1=b1, b0=b0, n=n, chain=chain: ops.reslayer_split(x.clone(), wq, b1, b0, n, chain=chain), k))
            chain = 2
            wq, b1, b0, sc = layer16(k, n, proj, chain)          # the f16x2 instantiations of the same kernels
            cases.append(("f16x2 n%d proj%d chain%d" % (n, proj, chain),
                          lambda x, wq=wq, b1=b1, b0=b0, n=n, chain=chain, sc=sc: ops.reslayer_split16(x.clone(), wq, b1, b0, n, sc, chain=chain), k))
    # the gathering first layer (heads 40 + 5 x 64 table columns) with its chain, and the output layer with the bin draw
    wq_g, b1_g, b0_g = layer(360, 128, True, 4)
    table = mk(5000, 64)
    wq_d, b1_d, b0_d = layer(256, 192, True, 0)
    try:
        for grid, rows, reps in ((1, 6001, 10), (7, 20011, 30), (0, 150001, 30)):
            gidx = torch.randint(0, 5000, (rows, 5), generator=g).to(torch.int32).to(dev)
            uni = torch.rand(rows, 6, generator=g).to(dev)
            prior = mk(rows, 192)
            extra = [("gather chain4", lambda x: ops.reslayer_split_gather(x, gidx, table, wq_g, b1_g, b0_g, 128, chain=4), 40),
                     ("decode", lambda x: ops.reslayer_split_decode(x, wq_d, b1_d, b0_d, uni, prior=prior), 256)]
            for name, fn, k in cases + extra:
                x = mk(rows, k)
                _lib.check(L.cppf_reslayer_split_debug_grid(0), "grid")
                ref = fn(x)
                torch.cuda.synchronize()
                _lib.check(L.cppf_reslayer_split_debug_grid(grid), "grid")
                for rep in range(reps):
                    with torch.cuda.stream(side):
                        hog_b.copy_(hog_a)
                        hog_a.copy_(hog_b)
                    got = fn(x)
                    assert torch.equal(got, ref), (name, "workgroups", grid or "one per CU", "repetition", rep)
                torch.cuda.synchronize()
    finally:
        L.cppf_reslayer_split_debug_grid(0)


@pytest.mark.gpu
@pytest.mark.parametrize("arith", ["split", "split16"])
def test_generated_bin_prior_equals_its_array(arith):
    """ops.BinPrior (cppf_reslayer_split_decode's prior_pos / prior_inv_sigma): the Gaussian logit prior generated in the bin draw's
    epilogue draws the bins of the same prior passed as a [T, 6, 32] array, bit for bit -- positions inside, at the edges of and
    outside the bin range, several widths -- and of decode_bins on the written logits + that array; both at once are refused."""
    from cppf2_amd import _lib, models, ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(13)
    mk = lambda *s: torch.randn(*s, generator=g).to(dev)
    rows, k = 50001, 256
    w1, w0, w2 = mk(192, k) / k ** 0.5, mk(192, k) / k ** 0.5, mk(192, 192) / 192 ** 0.5
    b1, b0 = mk(192) * 0.1, mk(192) * 0.1
    x = mk(rows, k)
    uni = torch.rand(rows, 6, generator=g).to(dev)
    pos = (torch.rand(rows, 6, generator=g) * 37.0 - 3.0).to(dev)          # [-3, 34): also beyond bins 0 .. 31
    pos[:100] = torch.round(pos[:100])                                     # exactly on a bin
    for inv_sigma in (1.0 / 0.6, 0.25, 3.0):
        prior = ops.BinPrior(pos, inv_sigma)
        dense = prior.dense(32)
        assert dense.shape == (rows, 6, 32) and float(dense.max()) <= 0.0
        if arith == "split":
            wq = models.pack_split(w1, w0, w2, k)
            got = ops.reslayer_split_decode(x, wq, b1, b0, uni, prior=prior)
            want = ops.reslayer_split_decode(x, wq, b1, b0, uni, prior=dense)
        else:
            sc = models.f16_scale(w1, w0, w2)
            wq = models.pack_split(w1, w0, w2, k, arith="f16x2", scale=sc)
            got = ops.reslayer_split16(x, wq, b1 * sc, b0 * sc, 192, sc, decode=(uni, prior, None))
            want = ops.reslayer_split16(x, wq, b1 * sc, b0 * sc, 192, sc, decode=(uni, dense, None))
        assert torch.equal(got, want), inv_sigma
        assert not torch.equal(got, (ops.reslayer_split_decode(x, wq, b1, b0, uni) if arith == "split" else got * 0 - 1))   # the prior matters
    if arith == "split":
        L = _lib.load()
        bins = torch.empty((rows, 6), dtype=torch.int32, device=dev)
        rc = L.cppf_reslayer_split_decode_prior(ops._p(x), x.stride(0), k, rows, ops._p(wq), wq.numel() * wq.element_size(), ops._p(b1),
                                                ops._p(b0), ops._p(dense), ops._p(pos), 1.0, ops._p(uni), ops._p(bins), None, ops._stream())
        assert rc != 0                                                     # an array AND a generator: refused


@pytest.mark.gpu
def test_dynamic_row_blocks_equal_the_static_split():
    """The `sched` argument (ABI 10): persistent workgroups claiming their row blocks from a counter compute the bits of the
    fixed-share launch -- plain, chained, gathered, encoded, with the bin draw; at 1, 7 and all CUs; beside a saturating copy stream --
    and every launch leaves its two counter words zero, so consecutive launches of a stream share them."""
    from cppf2_amd import _lib, models, ops
    L = _lib.load()
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(9)
    mk = lambda *s: torch.randn(*s, generator=g).to(dev)

    def layer(k, n, proj, chain):
        w1, w2 = mk(n, k) / k ** 0.5, mk(n, n) / n ** 0.5
        w0 = mk(n, k) / k ** 0.5 if proj else None
        rest = [(mk(n, n) / n ** 0.5, mk(n, n) / n ** 0.5) for _ in range(chain)]
        return models.pack_split(w1, w0, w2, k, chain=rest), mk((1 + chain) * n) * 0.1, (mk(n) * 0.1 if proj else None)

    rows = 90001
    table = mk(5000, 64)
    gidx = torch.randint(0, 5000, (rows, 5), generator=g).to(torch.int32).to(dev)
    uni = torch.rand(rows, 6, generator=g).to(dev)
    cases = []
    for n, k, proj, chain in ((64, 64, False, 1), (128, 360, True, 4), (192, 256, True, 0), (256, 128, True, 2), (256, 256, False, 1)):
        wq, b1, b0 = layer(k, n, proj, chain)
        cases.append(("n%d k%d chain%d" % (n, k, chain), lambda x, wq=wq, b1=b1, b0=b0, n=n, chain=chain: ops.reslayer_split(x.clone(), wq, b1, b0, n, chain=chain), k))
    wq_g, b1_g, b0_g = layer(360, 128, True, 4)
    cases.append(("gather", lambda x: ops.reslayer_split_gather(x, gidx, table, wq_g, b1_g, b0_g, 128, chain=4), 40))
    wq_d, b1_d, b0_d = layer(256, 192, True, 0)
    cases.append(("decode", lambda x: ops.reslayer_split_decode(x, wq_d, b1_d, b0_d, uni), 256))
    side = torch.cuda.Stream()
    hog_a, hog_b = torch.empty(1 << 27, device=dev, dtype=torch.uint8), torch.empty(1 << 27, device=dev, dtype=torch.uint8)
    assert ops.DYNAMIC_BLOCKS
    try:
        for name, fn, k in cases:
            x = mk(rows, k)
            ops.DYNAMIC_BLOCKS = False
            ref = fn(x)
            ops.DYNAMIC_BLOCKS = True
            for grid in (0, 1, 7):
                _lib.check(L.cppf_reslayer_split_debug_grid(grid), "grid")
                for rep in range(4):
                    with torch.cuda.stream(side):
                        hog_b.copy_(hog_a)
                    got = fn(x)
                    assert torch.equal(got, ref), (name, grid, rep)
                torch.cuda.synchronize()
                for buf in ops._SCHED.values():
                    assert int(buf.abs().sum()) == 0, (name, grid)
    finally:
        ops.DYNAMIC_BLOCKS = True
        L.cppf_reslayer_split_debug_grid(0)


@pytest.mark.gpu
def test_reserved_cus_change_the_grid_not_the_results():
    """cppf_mlp_reserve_cus (the batch mode's knob): launches that leave CUs to other streams -- 1, one per shader engine, half
    the chip, more than the library accepts to give away -- return the bits of the default launch; the context manager restores
    one workgroup per CU; a negative count is refused."""
    from cppf2_amd import _lib, models, ops
    L = _lib.load()
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(5)
    mk = lambda *s: torch.randn(*s, generator=g).to(dev)
    n, k, rows = 128, 360, 70001
    rest = [(mk(n, n) / n ** 0.5, mk(n, n) / n ** 0.5) for _ in range(2)]
    wq = models.pack_split(mk(n, k) / k ** 0.5, mk(n, k) / k ** 0.5, mk(n, n) / n ** 0.5, k, chain=rest)
    b1, b0 = mk(3 * n) * 0.1, mk(n) * 0.1
    x = mk(rows, k)
    ref = ops.reslayer_split(x.clone(), wq, b1, b0, n, chain=2)
    assert ops.batch_mode_reserved_cus(dev) == torch.cuda.get_device_properties(dev).multi_processor_count // 8
    try:
        for cus in (1, ops.batch_mode_reserved_cus(dev), 128, 1000):
            ops.mlp_reserve_cus(cus)
            assert torch.equal(ops.reslayer_split(x.clone(), wq, b1, b0, n, chain=2), ref), cus
        # ABI 11: the call returns the previous reservation, a negative argument only queries, and the context managers restore
        # the enclosing block's value instead of zeroing it
        assert ops.mlp_reserve_cus(7) == 1000 and ops.mlp_reserve_cus(-1) == 7 and L.cppf_mlp_reserve_cus(-3) == 7
        with ops.mlp_cus_reserved():
            assert ops.mlp_reserve_cus(-1) == ops.batch_mode_reserved_cus(dev)
            assert torch.equal(ops.reslayer_split(x.clone(), wq, b1, b0, n, chain=2), ref)
            from cppf2_amd.pipeline import BatchMode
            with BatchMode([object(), object()], device=dev, reserve_cus=5):
                assert ops.mlp_reserve_cus(-1) == 5
            assert ops.mlp_reserve_cus(-1) == ops.batch_mode_reserved_cus(dev)
        assert ops.mlp_reserve_cus(-1) == 7
        assert L.cppf_mlp_reserve_cus(5000) < 0 and ops.mlp_reserve_cus(-1) == 7          # out of range: refused, unchanged
    finally:
        ops.mlp_reserve_cus(0)


@pytest.mark.gpu
def test_tapped_first_layer_equals_the_separate_launches():
    """cppf_reslayer_split_tap: the first layer's output (the tuple features) is bit-identical to what that layer's own launch
    writes; the chain's output equals the separate launches' (128 -> 256 projection, then two 256-wide identity layers) up to
    the summation order -- a layer fed from memory contracts its K steps in another feature order than one fed from the
    accumulators --, i.e. to float32 rounding; the models' two-stack forward (tuple encoder into logit head) returns both."""
    from cppf2_amd import models, ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(21)
    mk = lambda *s: torch.randn(*s, generator=g).to(dev)
    for k, n, chain in ((128, 256, 2), (360, 128, 4), (72, 64, 1)):
        w1, w0, w2 = mk(n, k) / k ** 0.5, mk(n, k) / k ** 0.5, mk(n, n) / n ** 0.5
        rest = [(mk(n, n) / n ** 0.5, mk(n, n) / n ** 0.5) for _ in range(chain)]
        b1, b0 = mk((1 + chain) * n) * 0.1, mk(n) * 0.1
        x = mk(3001, k)
        first = ops.reslayer_split(x, models.pack_split(w1, w0, w2, k), b1[:n].contiguous(), b0, n)
        if chain:
            wq_rest = models.pack_split(rest[0][0], None, rest[0][1], n, chain=rest[1:])
            want = ops.reslayer_split(first.clone(), wq_rest, b1[n:].contiguous(), None, n, chain=chain - 1)
        tap = torch.full((3001, n), float("nan"), device=dev)
        got = ops.reslayer_split(x, models.pack_split(w1, w0, w2, k, chain=rest), b1, b0, n, chain=chain, tap=tap)
        assert torch.equal(tap, first), (k, n, chain)
        assert (got - want).abs().max().item() < 4e-6 * max(1.0, want.abs().max().item()), (k, n, chain)
        again = ops.reslayer_split(x, models.pack_split(w1, w0, w2, k, chain=rest), b1, b0, n, chain=chain)     # no tap: same chain
        assert torch.equal(got, again)
    with pytest.raises(Exception):
        ops.reslayer_split(x, models.pack_split(w1, w0, w2, k, chain=rest), b1, b0, n, out=tap, chain=chain, tap=tap)
    # the models: heads() through the two-stack form == the stacks one after the other
    class Cfg:
        num_more = 3
    torch.manual_seed(2)
    net = models.BeyondCPPFShot(Cfg()).to(dev).eval()
    xin = mk(5000, 360) * 0.3
    with torch.no_grad():
        cls, feat = net.heads(xin.clone(), lazy_scale=True)
        feat_want = models.fused_stack(net.tuple_encoder, xin.clone())
        cls_want = models.fused_stack(net.logit_encoder, feat_want.clone())
        assert torch.equal(feat, feat_want)
        assert (cls.reshape(5000, -1) - cls_want).abs().max().item() < 4e-6 * max(1.0, cls_want.abs().max().item())
        # the logit head's weights change: the cross-stack weight stream is rebuilt (cache keyed by both stacks' versions)
        net.logit_encoder[0].fc1.weight.mul_(1.5)
        cls2, _ = net.heads(xin.clone(), lazy_scale=True)
        cls2_want = models.fused_stack(net.logit_encoder, feat_want.clone())
        assert (cls2.reshape(5000, -1) - cls2_want).abs().max().item() < 4e-6 * max(1.0, cls2_want.abs().max().item())
        assert (cls2 - cls).abs().max().item() > 1e-3


def test_f16_pairs_carry_22_bits_and_the_host_packing_has_the_documented_layout():
    """models.split_f16 / pack_split(arith="f16x2"): hi + lo reproduces scale x w to 2^-22 relative (2^-25 absolute where lo is
    subnormal), both pieces are finite for |scale x w| < 2^14, and the stream has two fragments per tile in the bf16 order."""
    from cppf2_amd import models
    g = torch.Generator().manual_seed(2)
    w = torch.randn(256, 256, generator=g) / 16
    sc = models.f16_scale(w)
    assert 2 ** 12 <= float((w * sc).abs().max()) <= 2 ** 13 and sc == 2.0 ** round(float(torch.log2(torch.tensor(sc))))
    pieces = models.split_f16(w * sc)
    assert pieces.dtype == torch.float16 and torch.isfinite(pieces).all()
    back = pieces[0].double() + pieces[1].double()
    err = (back - (w * sc).double()).abs()
    assert float((err / (w * sc).double().abs().clamp_min(2.0 ** -2)).max()) < 2.0 ** -21.9
    a3 = models.pack_split(w[:128, :128].contiguous(), None, w[128:, 128:].contiguous(), 128)
    a2 = models.pack_split(w[:128, :128].contiguous(), None, w[128:, 128:].contiguous(), 128, arith="f16x2", scale=sc)
    assert a2.dtype == torch.float16 and a3.dtype == torch.bfloat16 and a2.numel() * 3 == a3.numel() * 2
    # first fragment of the stream = hi pieces of W1 rows 0..31 (lane i) at input features 8 g + j of K step 0
    frag = a2[:512].reshape(2, 32, 8).float()
    want = pieces[0][:128, :128][:32, :16].reshape(32, 2, 8).permute(1, 0, 2).float()
    assert torch.equal(frag, want)


@pytest.mark.gpu
def test_reslayer_split16_matches_float64_like_a_float32_gemm():
    """The f16x2 kernels (cppf_reslayer_split16) on every supported shape, plain / chained / tapped / gathered / with the bin
    draw: error against float64 under the same bound as the exact bf16-triple kernels (a few 2^-24 of the output scale, no worse
    than 3x the library float32 GEMMs'), and in practice below the bf16-triple kernels' own error (fewer accumulator roundings)."""
    from cppf2_amd import models, ops
    dev = torch.device("cuda:0")
    ratio16, ratio3 = [], []
    for k, n, proj, chain in [(k_, n_, p_, 0) for k_, n_, p_ in SHAPES] + [(360, 128, True, 4), (352, 128, True, 4), (128, 128, False, 3),
                                                                             (64, 64, False, 1), (128, 64, True, 2), (128, 256, True, 2),
                                                                             (256, 256, False, 1), (256, 192, True, 1)]:
        w1, b1, w0, b0, w2 = _layer(k, n, proj, dev, seed=k * 7 + n)
        rest = [_layer(n, n, False, dev, seed=20 + l) for l in range(chain)]
        pairs = [(e[0], e[4]) for e in rest]
        sc = models.f16_scale(w1, w0, w2, *[w for p in pairs for w in p])
        wq16 = models.pack_split(w1, w0, w2, k, chain=pairs, arith="f16x2", scale=sc)
        wq3 = models.pack_split(w1, w0, w2, k, chain=pairs)
        bias = torch.cat([b1] + [e[1] for e in rest])
        for rows in (1, 257, 3001):
            x = torch.randn(rows, k, device=dev, generator=torch.Generator(device=dev).manual_seed(rows))
            want = _ref64(x, w1, b1, w0, b0, w2)
            nat = torch.addmm(x if w0 is None else torch.addmm(b0, x, w0.t()), torch._addmm_activation(b1, x, w1.t()), w2.t())
            for e in rest:
                want = _ref64(want, e[0], e[1], None, None, e[4])
                nat = torch.addmm(nat, torch._addmm_activation(e[1], nat, e[0].t()), e[4].t())
            got = ops.reslayer_split16(x.clone(), wq16, bias * sc, None if b0 is None else b0 * sc, n, sc, chain=chain)
            got3 = ops.reslayer_split(x.clone(), wq3, bias, b0, n, chain=chain)
            scale = want.abs().max().item()
            e16 = (got.double() - want.double()).abs().max().item() / scale
            e3 = (got3.double() - want.double()).abs().max().item() / scale
            e_nat = (nat.double() - want.double()).abs().max().item() / scale
            assert e16 < 3e-6 and e16 < 3.0 * e_nat + 2e-7, (k, n, proj, chain, rows, e16, e_nat)
            if rows == 3001:
                r16 = ((got.double() - want).pow(2).mean().sqrt() / want.pow(2).mean().sqrt()).item()
                r3 = ((got3.double() - want).pow(2).mean().sqrt() / want.pow(2).mean().sqrt()).item()
                rn = ((nat.double() - want).pow(2).mean().sqrt() / want.pow(2).mean().sqrt()).item()
                ratio16.append(r16 / rn)
                ratio3.append(r3 / rn)
            # rows do not depend on the batch they sit in
            if rows == 257:
                assert torch.equal(got[:31], ops.reslayer_split16(x[:31].clone(), wq16, bias * sc, None if b0 is None else b0 * sc, n, sc, chain=chain))
    # rms error relative to the library float32 GEMMs': f16x2 stays within 1.6x on every shape and is on average no worse than
    # the exact bf16 triples
    assert max(ratio16) < 1.6 and sum(ratio16) / len(ratio16) <= sum(ratio3) / len(ratio3) + 0.05, (ratio16, ratio3)
    # tap: first layer's output and the chain's, both equal to the untapped launches of the same arithmetic
    k, n, chain = 128, 256, 2
    w1, b1, w0, b0, w2 = _layer(k, n, True, dev, seed=5)
    rest = [_layer(n, n, False, dev, seed=30 + l) for l in range(chain)]
    pairs = [(e[0], e[4]) for e in rest]
    sc = models.f16_scale(w1, w0, w2, *[w for p in pairs for w in p])
    wq = models.pack_split(w1, w0, w2, k, chain=pairs, arith="f16x2", scale=sc)
    bias = torch.cat([b1] + [e[1] for e in rest]) * sc
    x = torch.randn(2000, k, device=dev)
    tap = torch.empty(2000, n, device=dev)
    out = ops.reslayer_split16(x, wq, bias, b0 * sc, n, sc, chain=chain, tap=tap)
    assert torch.equal(out, ops.reslayer_split16(x, wq, bias, b0 * sc, n, sc, chain=chain))
    first_want = _ref64(x, w1, b1, w0, b0, w2)
    assert (tap.double() - first_want).abs().max().item() < 2e-6 * first_want.abs().max().item()
    # NaN stays NaN; an activation beyond fp16's range turns its row non-finite instead of silently wrong
    xx = torch.randn(64, k, device=dev)
    xx[3, 5] = float("nan")
    xx[7, 1] = 1e6
    got = ops.reslayer_split16(xx, wq, bias, b0 * sc, n, sc, chain=chain)
    assert torch.isnan(got[3]).all() and not torch.isfinite(got[7]).any() and torch.isfinite(got[[0, 1, 2, 4, 5, 6, 8]]).all()
    # bad arguments
    from cppf2_amd import _lib
    with pytest.raises(_lib.CppfError):
        ops.reslayer_split16(x, wq, bias, b0 * sc, n, 3.0, chain=chain)            # scale not a power of two
    with pytest.raises(_lib.CppfError):
        ops.reslayer_split16(x, wq[:-8], bias, b0 * sc, n, sc, chain=chain)        # stream size


@pytest.mark.gpu
def test_models_in_f16x2_arithmetic_match_the_modules_and_their_own_unfused_forms():
    """MLP_ARITH = "split16" through fused_stack: the SHOT model's forward against the plain float32 modules (3e-5), the gathered
    first layer bit-identical to the materialised rows, the fused bin draw bit-identical to decode_bins on the same arithmetic's
    logits, and the logits within 2e-5 of the exact bf16-triple arithmetic's."""
    import os
    from cppf2_amd import models, ops
    from cppf2_amd.config import load_config
    dev = torch.device("cuda:0")
    cfg = load_config(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "config"), "config", ["category=bottle"])
    torch.manual_seed(4)
    net = models.BeyondCPPFShot(cfg).to(dev).eval()
    N, T = 1500, 6000
    pts = torch.randn(N, 3, device=dev) * 0.05
    nrm = torch.nn.functional.normalize(torch.randn(N, 3, device=dev), dim=-1)
    sfeat = torch.rand(N, 352, device=dev)
    idx = torch.randint(0, N, (T, 5), device=dev, dtype=torch.int32)
    u = torch.rand(T, 6, device=dev)
    prev = models.MLP_ARITH
    try:
        with torch.no_grad():
            want_cls, want_scale = net(pts, idx.long(), sfeat, nrm)            # plain modules (library float32)
            models.MLP_ARITH = "split16"
            feat = net.encode_points(sfeat)
            assert (feat - net.shot_encoder(sfeat)).abs().max().item() < 2e-5 * max(1.0, feat.abs().max().item())
            x = ops.encode_tuples_shot(pts, idx, feat, nrm)
            cls_m, sc_m = net.heads(x.clone())
            assert (cls_m - want_cls).abs().max().item() < 3e-5 * max(1.0, want_cls.abs().max().item())
            assert (sc_m - want_scale).abs().max().item() < 3e-5 * max(1.0, want_scale.abs().max().item())
            cls_g, sc_g = net.heads_from_tuples(pts, idx, feat, nrm)           # gathered first layer: same arithmetic, same bits
            assert torch.equal(cls_g, cls_m) and torch.equal(sc_g, sc_m)
            bins = torch.empty(T, 6, dtype=torch.int32, device=dev)
            none, _ = net.heads_from_tuples(pts, idx, feat, nrm, decode=(u, None, bins))
            assert none is None
            want_bins = ops.decode_bins(cls_m.contiguous(), u, pts, idx, cfg.up, cfg.front, cfg.right)["bins"]
            assert torch.equal(bins, want_bins)
            models.MLP_ARITH = "split"
            cls_3, _ = net.heads(x.clone())
            assert (cls_3 - cls_m).abs().max().item() < 2e-5 * max(1.0, want_cls.abs().max().item())
    finally:
        models.MLP_ARITH = prev
