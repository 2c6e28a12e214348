"""Where a reference-protocol step through the drop-in operator spends its time (every section synchronised, then free-running):
activations + torch.ones, rasterization forward, loss, backward, absgrad, the four torch.optim.Adam steps.
usage: python tools/operator_profile.py [config2] [native]   (native: edgegaussians_amd.optim.Adam instead of torch's)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from edgegaussians_amd import synth
from gsplat import rasterization
name = sys.argv[1] if len(sys.argv) > 1 else "config2"
n, V, w, h = bench.CONFIGS[name]
dev = "cuda:0"
sc = synth.make_scene(n, V, w, h, seed=0, anisotropy=5.0, cameras_npz=bench.REAL_POSES)
P = {"means": torch.nn.Parameter(sc.means.to(dev)), "scales": torch.nn.Parameter(sc.log_scales.to(dev)),
     "quats": torch.nn.Parameter(sc.quats.to(dev)), "opacities": torch.nn.Parameter(sc.logit_opacities.to(dev))}
lrs = {"means": 2e-6, "scales": 1e-7, "quats": 1e-6, "opacities": 3e-5}
if "native" in sys.argv[2:]:
    from edgegaussians_amd.optim import Adam as adam_cls
else:
    adam_cls = torch.optim.Adam
opts = [adam_cls([P[k]], lr=lrs[k]) for k in P]
absgrads = torch.zeros(n, device=dev)
vms, Ks, gt = sc.viewmats.to(dev), sc.Ks.to(dev), sc.gt.to(dev)
whole = synth.weight_map("whole", sc.gt[0]).to(dev)
# host-only time of the operator's two autograd hooks (no synchronisation added: what the host spends inside them)
from edgegaussians_amd import rasterizer as _R
H = {"fwd": 0.0, "bwd": 0.0, "n": 0}
_f0, _b0 = _R._UnitRasterization.forward, _R._UnitRasterization.backward
def _fw(ctx, *a):
    t = time.perf_counter(); r = _f0(ctx, *a); H["fwd"] += time.perf_counter() - t; H["n"] += 1; return r
def _bw(ctx, *a):
    t = time.perf_counter(); r = _b0(ctx, *a); H["bwd"] += time.perf_counter() - t; return r
_R._UnitRasterization.forward, _R._UnitRasterization.backward = staticmethod(_fw), staticmethod(_bw)
T = {}
def tick(k, t0):
    torch.cuda.synchronize(); T[k] = T.get(k, 0.0) + time.perf_counter() - t0; return time.perf_counter()
def step(s, timed):
    v = s % V
    t = time.perf_counter()
    colors = torch.ones(n, 3, device=dev)
    sc_, op_ = torch.exp(P["scales"]), torch.sigmoid(P["opacities"]).squeeze(-1)
    if timed: t = tick("activations+ones", t)
    render, alpha, info = rasterization(means=P["means"], quats=P["quats"], scales=sc_, opacities=op_, colors=colors,
                                        viewmats=vms[v:v + 1], Ks=Ks[v:v + 1], width=w, height=h, tile_size=16, packed=False,
                                        near_plane=0.01, far_plane=1e10, render_mode="RGB", sparse_grad=False, absgrad=True,
                                        rasterize_mode="antialiased")
    info["means2d"].retain_grad()
    if timed: t = tick("rasterization fwd", t)
    rgb = torch.clamp(render[0, ..., :3], 0.0, 1.0)
    loss = (whole * (rgb[:, :, 0] - gt[v]).abs()).sum()
    if timed: t = tick("loss", t)
    loss.backward()
    if timed: t = tick("backward", t)
    absgrads.add_(info["means2d"].absgrad[0].norm(dim=-1))
    if timed: t = tick("absgrad", t)
    for o in opts:
        o.step(); o.zero_grad()
    if timed: t = tick("4 x Adam", t)
for s in range(30): step(s, False)
torch.cuda.synchronize()
K = 100
for s in range(K): step(s, True)
tot = sum(T.values())
for k, v in T.items(): print(f"{k:22s} {1e6 * v / K:8.1f} us")
print(f"{'sum (every section synchronised)':22s} {1e6 * tot / K:8.1f} us")
H.update(fwd=0.0, bwd=0.0, n=0)
t0 = time.perf_counter()
for s in range(K): step(s, False)
torch.cuda.synchronize()
print(f"free-running              {1e6 * (time.perf_counter() - t0) / K:8.1f} us/step")
print(f"host time inside the operator's hooks, free-running: forward {1e6 * H['fwd'] / max(H['n'], 1):6.1f} us, "
      f"backward {1e6 * H['bwd'] / max(H['n'], 1):6.1f} us per call (EG_OPERATOR_DEFER={os.environ.get('EG_OPERATOR_DEFER', '1')})")
