import importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
bench = importlib.import_module("bench")
PKG = "sfm-learner-chainer_amd"
ops = importlib.import_module(PKG + ".ops"); synth = importlib.import_module(PKG + ".synth")
links = importlib.import_module(PKG + ".links"); cs = importlib.import_module(PKG + ".chainer_surface")
dev = torch.device("cuda", 0)
r = bench.Runner(torch, np, ops, synth, dev, "ref_b4", "hwc", "fused")
model = links.SFMLearnerLoss(dict(seq_len=3, smooth_reg=0.0, exp_reg=0.0, ssim_rate=0.0))
K, disps, poses = r.common
vd, vp = [cs.Variable(a) for a in disps], [cs.Variable(a) for a in poses]
tgt, src = r.full
def step():
    for v in vd + vp: v.cleargrad()
    loss = model(tgt, src, K, None, vd, vp); loss.backward(); return loss
for _ in range(100): step()
torch.cuda.synchronize()
N = 3000
def timeit(f, n=N):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    dt = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize(); return dt
print("whole step            %.1f us" % timeit(step))
def only_call():
    return model(tgt, src, K, None, vd, vp)
print("model() only          %.1f us" % timeit(only_call))
st = model._repeat.st
fused = st.fused
out = torch.empty((5,), dtype=torch.float32, device=dev)
print("step_from_frames      %.1f us" % timeit(lambda: fused.step_from_frames(model._repeat.tgt, model._repeat.stacked, True, out)))
print("torch.empty((5,))     %.1f us" % timeit(lambda: torch.empty((5,), dtype=torch.float32, device=dev)))
print("out[0:1].reshape(())  %.1f us" % timeit(lambda: out[0:1].reshape(())))
print("out.unbind(0)         %.1f us" % timeit(lambda: out.unbind(0)))
print("ones_like scalar      %.1f us" % timeit(lambda: torch.ones_like(out[0])))
print("Variable(out)         %.1f us" % timeit(lambda: cs.Variable(out)))
inputs = vd + vp
print("any(requires_grad)    %.1f us" % timeit(lambda: cs.config.enable_backprop and any(isinstance(v, cs.Variable) and v.requires_grad for v in inputs)))
loss = only_call()
def bw():
    for v in inputs: v.grad = None
    loss.grad = None
    loss.backward()
print("backward only         %.1f us" % timeit(bw))
print("5 reports             %.1f us" % timeit(lambda: [cs.report({'a': 1}, model) for _ in range(5)]))
import ctypes as C
print("lib.sfm_abi_version   %.2f us" % timeit(lambda: ops.lib.sfm_abi_version()))
