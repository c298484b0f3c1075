"""Race screen for cppf_reslayer_split's counted waits: every launch shape of the bench (and the chains) repeated with a
bandwidth hog running beside it on a second stream; all repetitions must be bit-identical (and equal to a run alone)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cppf2_amd import models, ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 320000
hog_a = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
hog_b = torch.empty_like(hog_a)
side = torch.cuda.Stream()
bad = 0
for k, n, proj, chain in ((360, 128, True, 4), (128, 256, True, 0), (256, 256, False, 1), (256, 192, True, 0), (352, 128, True, 4),
                          (128, 64, True, 0), (256, 128, True, 0), (72, 64, True, 1)):
    g = torch.Generator(device="cpu").manual_seed(k + n)
    mk = lambda *s: torch.randn(*s, generator=g).to(dev)
    w1, w2 = mk(n, k) / k ** 0.5, mk(n, n) / n ** 0.5
    w0 = mk(n, k) / k ** 0.5 if proj else None
    rest = [(mk(n, n) / n ** 0.5, mk(n, n) / n ** 0.5) for _ in range(chain)]
    wq = models.pack_split(w1, w0, w2, k, chain=rest)
    b1 = mk((1 + chain) * n) * 0.1
    b0 = mk(n) * 0.1 if proj else None
    x = mk(rows, k)
    ref = ops.reslayer_split(x.clone(), wq, b1, b0, n, chain=chain)
    torch.cuda.synchronize()
    for rep in range(12):
        with torch.cuda.stream(side):
            for _ in range(3):
                hog_b.copy_(hog_a)
        got = ops.reslayer_split(x.clone(), wq, b1, b0, n, chain=chain)
        torch.cuda.synchronize()
        if not torch.equal(got, ref):
            bad += 1
            print("MISMATCH", k, n, proj, chain, "rep", rep, (got - ref).abs().max().item())
    print("shape K=%d N=%d proj=%d chain=%d: 12 repetitions beside a copy stream identical to the solo run" % (k, n, proj, chain), flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
