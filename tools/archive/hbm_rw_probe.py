"""What the memory system gives a pure write / pure read / copy stream (torch elementwise kernels, 4.3 GB tensors)."""
import torch
n = 843264 * 2560
x = torch.empty(n, dtype=torch.float16, device="cuda")
y = torch.empty(n, dtype=torch.float16, device="cuda")
def t(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
gb = n * 2 / 1e9
ms = t(lambda: x.fill_(1.0)); print("fill  (write only) %.3f ms  %.2f TB/s" % (ms, gb / ms))
ms = t(lambda: x.sum());      print("sum   (read only)  %.3f ms  %.2f TB/s" % (ms, gb / ms))
ms = t(lambda: y.copy_(x));   print("copy  (read+write) %.3f ms  %.2f TB/s total" % (ms, 2 * gb / ms))
ms = t(lambda: torch.mul(x, 2.0, out=y)); print("scale (read+write) %.3f ms  %.2f TB/s total" % (ms, 2 * gb / ms))
