import sys, time, torch
sys.path.insert(0, '.')
import bench
from edgegaussians_amd import _lib
from edgegaussians_amd._lib import ptr, stream
import ctypes as C
tr, sc, whole, ratio = bench.build_trainer(bench.CONFIGS["config2"], 0, "cuda:0")
tr.ensure_capacity()
for s in range(30):
    tr.train_step(s % 50, whole)
torch.cuda.synchronize()
lib = _lib.load()
st = stream()
W, H, N = tr.width, tr.height, tr.N
view = 1
def bind(name, *args):
    fn = getattr(lib, name)
    def f():
        rc = fn(*args); assert rc == 0
    return f
fwd = bind("eg_composite_fwd", ptr(tr.splat), None, 1, ptr(tr.offsets), ptr(tr.flatten_ids), W, H, ptr(tr.render), ptr(tr.alphas), ptr(tr.last_ids), ptr(tr.gt[view]), ptr(whole), 1.0, ptr(tr.vpix), ptr(tr.loss_acc), ptr(tr.item_offsets), ptr(tr.total), tr.max_items, ptr(tr.workspace), ptr(tr.gtstop), st)
fp = bind("eg_composite_bwd_footprint", ptr(tr.splat), N, W, H, ptr(tr.gtstop), ptr(tr.g2d), st)
srt = bind("eg_sort_pairs", ptr(tr.keys), ptr(tr.offsets), tr.T, tr.capacity, ptr(tr.flatten_ids), None, 0, st)
def host_time(f, reps):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return 1e6 * (time.perf_counter() - t0) / reps
def event_time(f, reps):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize(); return 1e3 * e0.elapsed_time(e1) / reps
for name, f in (("composite_fwd", fwd), ("footprint", fp), ("sort", srt)):
    print(name, "host", [round(host_time(f, r), 1) for r in (20, 200, 1000)], "event", [round(event_time(f, r), 1) for r in (20, 200, 1000)])
# fused step host timing
def step(): tr.train_step(3, whole)
print("fused step host", [round(host_time(step, r), 1) for r in (50, 400)])
