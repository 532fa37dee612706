"""What a 2-read : 1-write elementwise stream reaches on this GPU (torch's own vectorised add / copy kernels as the yardstick for
bn_apply_res, which reads the raw conv output and the identity and writes the block output + sign bytes)."""
import torch
dev = torch.device("cuda:0")
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for n in (1056 * 56 * 56 * 256, 1056 * 28 * 28 * 512, 1056 * 14 * 14 * 1024):
    a = torch.randn(n, device=dev); b = torch.randn(n, device=dev); c = torch.empty_like(a)
    t_add = timeit(lambda: torch.add(a, b, out=c))
    t_cpy = timeit(lambda: c.copy_(a))
    t_relu = timeit(lambda: torch.relu_(c))
    print(f"{n*4/1e9:.2f} GB tensors: add (2R:1W) {3*n*4/t_add/1e9:.0f} GB/s | copy (1R:1W) {2*n*4/t_cpy/1e9:.0f} GB/s | relu_ in place (1R:1W same lines) {2*n*4/t_relu/1e9:.0f} GB/s")
