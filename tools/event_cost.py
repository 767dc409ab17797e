import os, sys, time, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
from richsem_amd import _lib, workload as W
from richsem_amd import MultiScaleDeformableAttention as MSDA
_lib.load()
layers = []
for ci, (c, reps) in enumerate([(W.call_E(2), 6), (W.call_Dd(2), 6)]):
    for layer in range(reps):
        t = W.make_inputs(c, "init", seed=1000 * ci + layer, device="cuda")
        layers.append((c, t))
def step():
    for c, t in layers:
        MSDA.ms_deform_attn_forward(t["value"], t["shapes"], t["lsi"], t["loc"], t["aw"], 64)
    for c, t in reversed(layers):
        MSDA.ms_deform_attn_backward(t["value"], t["shapes"], t["lsi"], t["loc"], t["aw"], t["grad_out"], 64)
def timed(profile, steps=20):
    for _ in range(5): step()
    torch.cuda.synchronize()
    if profile: _lib.profile_enable(24 * steps)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    if profile: _lib.profile_collect(); _lib.profile_enable(0)
    return el / steps * 1e3
for r in range(3):
    print("profile on %.4f ms   off %.4f ms" % (timed(True), timed(False)))
