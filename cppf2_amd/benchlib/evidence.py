"""Untimed accuracy evidence bench.py attaches to the headline line, measured on the bench's own tuples after the timed loops."""
import torch

from .workloads import Cfg


def arithmetic_evidence(step, rows_err=4096, scenes_flip=10):
    """Untimed evidence for the line's `dtype` and parity claims, measured on the bench's own tuples (rank 0, after timing):
    * mlp_error_vs_f64: the tuple MLP's logits (tuple_encoder + logit_encoder, train_shot.py:56-66) of the first `rows_err`
      tuples in split arithmetic (what the step runs) and on PyTorch's float32 library GEMMs, each against a float64 evaluation
      of the same weights and rows; errors relative to the largest |logit|;
    * bin_flip_rate_vs_expf: the bins the decode kernel draws (softmax_exp: hardware exp2, ~1.5 ulp) against the same draw
      with torch.exp (libm-accurate expf) in the same float32 running-sum order, over `scenes_flip` scenes x T tuples x 6."""
    import copy
    from cppf2_amd import models as M
    ops, pipe, a, dev = step.ops, step.pipe, step.args, step.dev
    N, T = step.N, step.T
    sc = min(scenes_flip, step.B)
    ids = tuple(range(step.scene0, step.scene0 + sc))
    idx = ops.sample_tuples(N, T, 5, a.seed, ids, dev)
    pt_off, tup_off = ops._uniform_offsets(N, sc, dev), ops._uniform_offsets(T, sc, dev)
    feat = step.model.encode_points(step.shot[:sc * N])
    x = ops.encode_tuples_shot(step.pts[:sc * N], idx, feat, step.normal[:sc * N], pt_off, tup_off)
    out = {}
    if M.MLP_ARITH in ("split", "split16"):
        arith0 = M.MLP_ARITH
        xe = x[:rows_err].contiguous()
        m64 = copy.deepcopy(step.model).double()
        want = m64.logit_encoder(m64.tuple_encoder(xe.double()))
        scale = want.abs().max().item()
        M.MLP_ARITH = "split"
        split = M.fused_stack(step.model.logit_encoder, M.fused_stack(step.model.tuple_encoder, xe.clone()))
        M.MLP_ARITH = "split16"
        split16 = M.fused_stack(step.model.logit_encoder, M.fused_stack(step.model.tuple_encoder, xe.clone()))
        M.MLP_ARITH = arith0
        native = step.model.logit_encoder(step.model.tuple_encoder(xe))              # plain nn.Linear: library f32 GEMMs

        def err(t):
            d = t.double() - want
            return {"max": d.abs().max().item() / scale, "rms": d.pow(2).mean().sqrt().item() / scale}
        out["mlp_error_vs_f64"] = {"rows": int(xe.shape[0]), "logit_scale": scale, "split_bf16x3": err(split),
                                   "split_f16x2": err(split16), "library_f32_gemm": err(native),
                                   "note": "relative to max |logit|; the timed step runs " + ("split_bf16x3" if arith0 == "split" else "split_f16x2")}
    logits = step.model.heads(x, lazy_scale=True)[0].contiguous()                  # [sc*T, 6, 32]
    u = ops.philox_uniform(T, 6, a.seed, 1, ids, dev)
    prior = step.prior_dense[:sc * T]
    got = ops.decode_bins(logits, u, step.pts[:sc * N], idx, Cfg.up, Cfg.front, Cfg.right, pt_off, tup_off, prior=prior)["bins"]
    e = logits + prior
    p = torch.exp(e - e.max(-1, keepdim=True).values)
    cdf = torch.empty_like(p)
    run = torch.zeros_like(p[..., 0])
    for j in range(p.shape[-1]):                                                   # the kernel's float32 running sum, in bin order
        run = run + p[..., j]
        cdf[..., j] = run
    target = u.reshape(-1, 6) * run
    ref = (cdf <= target[..., None]).sum(-1).clamp(max=p.shape[-1] - 1).to(torch.int32)
    flips = int((ref != got).sum().item())
    out["bin_flip_rate_vs_expf"] = {"draws": int(ref.numel()), "flips": flips, "rate": flips / ref.numel(),
                                    "max_bin_distance": int((ref - got).abs().max().item())}
    return out


MLP_LAUNCH_OPS = ("reslayer_split_encode", "reslayer_split_gather", "reslayer_split", "reslayer_split_decode", "reslayer_split16")


@torch.no_grad()
def mlp_launch_loops(step, tel, seconds=1.5):
    """Each launch of the tuple MLP (the step's three cppf_reslayer_split* calls: gathered 360 -> 128 chain | 128 -> 256 tapped +
    the logit head's 256-wide layers | 256 -> 192 + bin draw) run BACK TO BACK ON ITS OWN for ~`seconds`, under telemetry windows
    `mlp_launch_<i>`: socket power and shader clock while nothing but that launch form runs -- the measurement behind (or against)
    "the MLP kernels sit at the power limit".  The calls are recorded from one real step (same tensors, same arguments) and
    replayed; returns [{launch, op, ms (HIP events over the loop), window}]."""
    from cppf2_amd import models as M
    from cppf2_amd import shot as shotmod
    ops, pipe, a = step.ops, step.pipe, step.args
    calls = []
    saved = {n: getattr(ops, n) for n in MLP_LAUNCH_OPS}

    def recorder(name):
        def f(*args, **kw):
            calls.append((name, args, kw))
            return saved[name](*args, **kw)
        return f
    ids = tuple(range(step.scene0, step.scene0 + step.B))
    idx = ops.sample_tuples(step.N, step.T, 5, a.seed, ids, step.dev)
    shotmod.prepare_device(step.pts, pipe.pt_off, Cfg.res * 10, Cfg.res * 10, step.normal)
    shot = shotmod.describe_device(step.pts, pipe.pt_off, step.normal, Cfg.res * 10, out=step.shot, nan_to_zero=True)
    normal = ops.nan_to_zero_(step.normal)
    feat = step.model.encode_points(shot)
    u = ops.philox_uniform(step.T, 6, a.seed, 1, ids, step.dev)
    src = ops.TupleSource(step.pts, idx, normal, pipe.pt_off, pipe.tup_off)
    for n in MLP_LAUNCH_OPS:
        setattr(ops, n, recorder(n))
    try:
        M.fused_stack((step.model.tuple_encoder, step.model.logit_encoder), None, gather=(src, None, feat),
                      decode=(u, step.prior, pipe.bins) if M.decode_supported(step.model.logit_encoder, feat) else None)
    finally:
        for n, f in saved.items():
            setattr(ops, n, f)
    torch.cuda.synchronize()
    out = []
    for i, (name, args, kw) in enumerate(calls):
        fn = saved[name]
        for _ in range(3):
            fn(*args, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(8):
            fn(*args, **kw)
        e1.record()
        torch.cuda.synchronize()
        reps = max(16, int(seconds * 1e3 / max(e0.elapsed_time(e1) / 8, 1e-3)))
        wname = "mlp_launch_%d" % i
        with tel.window(wname, torch.cuda.synchronize):
            e0.record()
            for _ in range(reps):
                fn(*args, **kw)
            e1.record()
        out.append({"launch": i, "op": name, "chain": kw.get("chain"), "reps": reps, "ms": round(e0.elapsed_time(e1) / reps, 4),
                    "window": wname})
    return out


@torch.no_grad()
def library_bf16_gemm_loop(dev, tel, seconds=1.5, n=8192):
    """What the bf16 matrix pipe delivers on THIS chip under THIS power cap when nothing but a library GEMM runs: torch.matmul of two
    n x n bf16 matrices (hipBLASLt), back to back for ~`seconds`, under the telemetry window `library_bf16_gemm`.  The yardstick
    beside roofline.frac_executed: the MLP kernels issue bf16 MFMA work at frac_executed x the 2.5 PFLOP/s nameplate; the nameplate
    assumes 2.4 GHz, which no kernel that keeps the matrix pipe busy holds at 1 400 W."""
    g = torch.Generator(device=dev).manual_seed(0)
    a = torch.randn((n, n), device=dev, generator=g).to(torch.bfloat16)
    b = torch.randn((n, n), device=dev, generator=g).to(torch.bfloat16)
    c = torch.empty((n, n), device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        torch.matmul(a, b, out=c)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(4):
        torch.matmul(a, b, out=c)
    e1.record()
    torch.cuda.synchronize()
    reps = max(8, int(seconds * 1e3 / max(e0.elapsed_time(e1) / 4, 1e-3)))
    with tel.window("library_bf16_gemm", torch.cuda.synchronize):
        e0.record()
        for _ in range(reps):
            torch.matmul(a, b, out=c)
        e1.record()
    ms = e0.elapsed_time(e1) / reps
    return {"op": "torch.matmul (hipBLASLt), bf16 x bf16 -> bf16, %d^3, random normal operands" % n, "reps": reps, "ms": round(ms, 4),
            "tflops": round(2.0 * n ** 3 / 1e12 / (ms / 1e3), 1), "window": "library_bf16_gemm"}
