"""K-split of the tail tiles of the fp32 gather (EDRL_GATHER_TAIL_SPLIT) on/off, forward and data gradient: time and the largest
difference relative to max |ref| (the split associates the K sum differently on the tail tiles) (diagnostic, GPU)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import edrl_amd
ops = edrl_amd.ops
dev = torch.device("cuda:0")


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, C, H, k, Co in (("l2 3x3 128", 128, 28, 3, 128), ("l3 3x3 256", 256, 14, 3, 256), ("l4 3x3 512", 512, 7, 3, 512), ("l4 1x1 2048-512", 2048, 7, 1, 512),
                          ("l4 1x1 512-2048", 512, 7, 1, 2048), ("l3 1x1 1024-256", 1024, 14, 1, 256), ("l3 1x1 256-1024", 256, 14, 1, 1024)):
    w = torch.randn(Co, k, k, C, device=dev) * 0.05
    wt = ops.permute_weight(w)
    for N in (1024,):
        x = torch.randn(N, H, H, C, device=dev)
        dy = torch.randn(N, H, H, Co, device=dev)
        fl = 2.0 * N * H * H * C * k * k * Co
        res = {}
        for sp in ("0", "1"):
            edrl_amd._lib.set_switches(EDRL_GATHER_TAIL_SPLIT=sp)
            y = ops.conv2d_fwd(x, w, stride=1, pad=k // 2)
            dx = ops.conv2d_dgrad(dy, wt, tuple(x.shape), 1, k // 2)
            res[sp] = (y.clone(), dx.clone())
            tf = min(timeit(lambda: ops.conv2d_fwd(x, w, stride=1, pad=k // 2)) for _ in range(3))
            td = min(timeit(lambda: ops.conv2d_dgrad(dy, wt, tuple(x.shape), 1, k // 2)) for _ in range(3))
            print(f"{name}  N {N:5d}  split {sp}  fwd {tf:7.3f} ms {fl / tf / 1e9:6.1f} TF | dgrad {td:7.3f} ms {fl / td / 1e9:6.1f} TF", flush=True)
        rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
        print(f"   max rel diff split vs unsplit: fwd {rel(res['1'][0], res['0'][0]):.2e}  dgrad {rel(res['1'][1], res['0'][1]):.2e}")
