"""Per-tensor gradient error at the full Lyft grid: GPU vs fp64 oracle next to the fp32 oracle vs fp64 oracle
(what tests/test_gpu_network.py bounds).  python tools/grad_conditioning.py [u20k|dense] [mse|smoothl1_ce]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_network as T  # noqa: E402

cloud = sys.argv[1] if len(sys.argv) > 1 else "u20k"
loss = sys.argv[2] if len(sys.argv) > 2 else "mse"
pts = T.u20k(5) if cloud == "u20k" else T.dense_sweep(6)
out, rows = T.full_grid_gradient_report(pts, loss, seed=5 if cloud == "u20k" else 6)
print(f"{cloud} {loss}: loss {out['loss']:.8g} ref {out['loss_ref']:.8g}")
for n, sc, e, o in rows:
    flag = "" if out["l2"][n][0] <= max(T.FLAT, T.SPREAD * out["l2"][n][1]) else "  <-- beyond max(FLAT, SPREAD x own)"
    a, b, c = out["l2"][n]
    print(f"{n:20s} max|ref| {sc:8.2e} max-norm: gpu {e:8.2e} o32 {o:8.2e} r {e / max(o, 1e-30):5.2f} | "
          f"L2: gpu {a:8.2e} o32 {b:8.2e} r {a / max(b, 1e-30):5.2f} gpu-vs-o32 {c:8.2e}{flag}")
print("worst gpu", max(r[2] for r in rows), "worst fp32 oracle", max(r[3] for r in rows))
