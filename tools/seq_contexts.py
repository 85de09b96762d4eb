"""GPU box, diagnosis: what makes contexts created LATER in a process run slower?  python tools/seq_contexts.py <variant>"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rnb_neus2_amd as rnb
from rnb_neus2_amd import synthetic

variant = sys.argv[1] if len(sys.argv) > 1 else "det+state"
scene = synthetic.make_scene(64, 800)
KW = dict(apply_no_albedo=1, mask_loss_weight=1.0, overlap=1)


def timed(c, n=200):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        c.train_step()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


def fresh(**kw):
    c = rnb.Context(**dict(KW, **kw))
    c.init_params()
    c.set_dataset(*scene)
    return c


first = fresh(deterministic=1 if "det" in variant else 0)
st = None
for _ in range(985):
    st = first.train_step()
print(variant, "| first context: %.4f ms/step" % timed(first), flush=True)
state = first.training_state(st) if "state" in variant else None
if "noclose" not in variant:
    first.close()
second = fresh()
if state is not None and "load" in variant:
    second.load_training_state(state)
else:
    for _ in range(985):
        second.train_step()
print(variant, "| second context: %.4f ms/step" % timed(second), flush=True)
second.close()
third = fresh()
for _ in range(985):
    third.train_step()
print(variant, "| third context: %.4f ms/step" % timed(third), flush=True)
if variant == "two-alive":
    third.close()
    a = fresh()
    b = fresh(deterministic=1)
    for _ in range(300):
        b.train_step()
    b.close()
    for _ in range(985):
        a.train_step()
    print(variant, "| context A (created first, B created and destroyed while A was alive): %.4f ms/step" % timed(a), flush=True)
    a.close()
    c = fresh()
    for _ in range(985):
        c.train_step()
    print(variant, "| context C (after both): %.4f ms/step" % timed(c), flush=True)
    c.close()
    d = fresh(target_batch_size=1 << 15, max_rays_per_batch=1 << 15, initial_rays_per_batch=512)
    for _ in range(985):
        d.train_step()
    print(variant, "| context D (after both, 2^15 samples): %.4f ms/step" % timed(d), flush=True)
