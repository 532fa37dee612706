"""debug: GradSync + trunk gradient stash on one GPU without a process group"""
import sys, os, types, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, edrl_amd
dev = torch.device("cuda:0")
for overlap in (False, True):
    edrl_amd.set_view_overlap(overlap)
    args = types.SimpleNamespace(mode="train", batch_size=2, encoder_depth=18)
    torch.manual_seed(0)
    model = edrl_amd.MedFusion(2, 2, None, args).to(dev).train()
    sync = edrl_amd.GradSync(model, bucket_mb=8)
    names = {p: n for n, p in model.named_parameters()}
    orig = sync._on_grad
    log = []
    def on_grad(p, orig=orig, log=log):
        b = sync.buckets[sync.index[p]]
        inside = any("params_ready" in f.name for f in traceback.extract_stack())
        log.append((names[p], b["index"], b["launched"], "sink" if inside else "hook"))
        if b["launched"]:
            print("ALREADY LAUNCHED:", names[p], "bucket", b["index"], "via", "sink" if inside else "hook")
            for e in log[-12:]: print("   ", e)
            print("   bucket params:", [names[q] for q in b["params"]][:8], "...", len(b["params"]))
        return orig(p)
    sync._on_grad = on_grad
    opt = edrl_amd.FusedAdam(model.parameters(), lr=1e-3)
    data, y = edrl_amd.synthetic_batch(2, 64, 64, 4, device=dev)
    try:
        edrl_amd.train_step(model, opt, data, y, grad_sync=sync)
        torch.cuda.synchronize()
        print("overlap", overlap, "ok;", len(log), "reports;", sum(1 for e in log if e[3] == "hook"), "by hook;", sync.hook_calls_ignored, "hook calls ignored")
    except Exception as e:
        print("overlap", overlap, "FAILED:", repr(e)[:200])
