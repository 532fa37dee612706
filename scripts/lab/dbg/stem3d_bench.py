"""The 3-D stem (7x7x7 / stride 2 on one channel): plain unfolded path against the space-to-depth path (diagnostic, GPU)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import edrl_amd
from edrl_amd_pkg.encoders3d import Conv3dFn, depth_unfold
ops = edrl_amd.ops
dev = torch.device("cuda:0")


def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


B = 32
x = torch.randn(B, 32, 224, 224, 1, device=dev)
w = torch.randn(64, 7, 7, 8, device=dev) * 0.05
w[..., 7:] = 0
xu = depth_unfold(x, 7, 2, 3, 8)
xu4 = xu.view(B * 16, 224, 224, 8)
fl = 2.0 * B * 16 * 112 * 112 * 64 * 343
print("unfold            %.3f ms" % timeit(lambda: depth_unfold(x, 7, 2, 3, 8)))
t = timeit(lambda: ops.conv2d_fwd(xu4, w, stride=2, pad=3)); print("plain 7x7/s2 K=392 %.3f ms  %.1f TF (algorithmic)" % (t, fl / t / 1e9))
t = timeit(lambda: ops.stem_conv_fwd(xu4, w)); print("s2d path (s2d + fold + conv) %.3f ms  %.1f TF" % (t, fl / t / 1e9))
xs = ops.space_to_depth2(xu4); wf = ops.stem_weight_fold(w)
print("  s2d alone       %.3f ms" % timeit(lambda: ops.space_to_depth2(xu4)))
t = timeit(lambda: ops.conv2d_fwd(xs, wf, stride=1, pad=2, out_hw=(112, 112))); print("  4x4/s1 K=512    %.3f ms  %.1f TF" % (t, fl / t / 1e9))
dy = torch.randn(B * 16, 112, 112, 64, device=dev)
t = timeit(lambda: ops.conv2d_wgrad(dy, xu4, (64, 7, 7, 8), 2, 3)); print("wgrad plain       %.3f ms  %.1f TF" % (t, fl / t / 1e9))
t = timeit(lambda: ops.stem_conv_wgrad(dy, xs, (64, 7, 7, 8), True)); print("wgrad s2d         %.3f ms  %.1f TF" % (t, fl / t / 1e9))
