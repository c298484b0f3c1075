import torch, time
dev = torch.device("cuda")
n = 1280000 * 360
x = torch.empty(n, dtype=torch.float32, device=dev)
y = torch.empty(n, dtype=torch.float32, device=dev)
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ms = t(lambda: x.fill_(1.0)); print("fill  %.3f ms  %.2f TB/s write" % (ms, n * 4 / ms / 1e9))
ms = t(lambda: y.copy_(x)); print("copy  %.3f ms  %.2f TB/s r+w" % (ms, 2 * n * 4 / ms / 1e9))
ms = t(lambda: x.sum()); print("sum   %.3f ms  %.2f TB/s read" % (ms, n * 4 / ms / 1e9))
