import torch


def relerr(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def check(name, got, ref, tol):
    assert got.shape == ref.shape, f"{name}: shape {tuple(got.shape)} vs {tuple(ref.shape)}"
    e = relerr(got, ref)
    print(f"[parity] {name}: max-rel-err {e:.3e} (tol {tol:.1e})")
    assert e == e and e <= tol, f"{name}: rel err {e:.3e} > {tol:.1e}"
