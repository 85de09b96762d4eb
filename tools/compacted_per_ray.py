import sys; sys.path.insert(0,".")
import numpy as np, rnb_neus2_amd as rnb
from rnb_neus2_amd import synthetic
ctx = rnb.Context(apply_no_albedo=1, mask_loss_weight=1.0); ctx.init_params(); ctx.set_dataset(*synthetic.make_scene(64, 800))
for target in (1001, 2001, 6001):
    while ctx.training_step < target: st = ctx.train_step()
    kept = int(ctx.get("COUNTERS")[2])
    ns = ctx.get("NUMSTEPS", kept*2).reshape(-1,2)[:,0]
    nz = ns[ns>0]
    print("step", target, "rays kept", kept, "with compacted", nz.size, "mean %.1f"%nz.mean(), "p50 %d p90 %d p99 %d max %d"%tuple(np.percentile(nz,[50,90,99,100])), "rays > 64: %d, > 128: %d, > 256: %d"%((nz>64).sum(),(nz>128).sum(),(nz>256).sum()))
ctx.close()
