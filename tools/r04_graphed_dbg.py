import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_step
dev = torch.device("cuda", 0)
model = bench_step.Step(n_img=2, dev=dev)
model.timing = False
images, mask, targets = model.batch()
model.prepare(mask, targets)
model._mask = mask
s = torch.cuda.Stream()
torch.cuda.graph.default_capture_stream = s
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    torch.manual_seed(1)
    with torch.no_grad():
        outs = model.model_part(images, mask)
        idx = model.pack_indices(model.match(*outs[:4], targets), targets)
        print("eager loss", float(model.loss_part(*outs, *idx)))
    model.loss_part(*model.model_part(images, mask), *idx).backward()
    for p in model.parameters():
        p.grad = None
    torch.cuda.synchronize()
    part_a = bench_step._ModelPart(model)
    ga = torch.cuda.make_graphed_callables(part_a, (images,), num_warmup_iters=3, allow_unused_input=True)
    for rep in range(3):
        torch.manual_seed(1)
        g_outs = ga(images)
        torch.cuda.synchronize()
        print("replay", rep, [f"{float((a.float() - b.float()).abs().mean()):.3g}/{float(b.float().abs().mean()):.3g}" for a, b in zip(g_outs, outs)],
              "finite", [bool(torch.isfinite(a).all()) for a in g_outs], flush=True)
    with torch.no_grad():
        print("loss on graphed outputs", float(model.loss_part(*[o.detach() for o in g_outs], *idx)))
