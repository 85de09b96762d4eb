#!/bin/bash
# GPU box: kernel statistics and one step's timeline of the half mode (accumulate = RNB_ACCUM_HALF) around training step 1000 -> gpurun_out/<tag>/{stats.txt,results.db}
#   bash tools/half_mode_profile.sh <tag> [ENV=V ...]
tag=$1; shift
mkdir -p gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
env "$@" timeout 600 rocprofv3 --kernel-trace -d gpurun_out/$tag -o half -- python bench.py --accumulate half --steps 100 --no-cpu-baseline --profile-steps 0 --late-steps 0 --fixed-cost-steps 0 --parity-mode-steps 0 --window-end 0 --no-live-pmc > gpurun_out/$tag/run.log 2>&1
grep "^{" gpurun_out/$tag/run.log | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('ms_per_step under rocprofv3', d['ms_per_step'])"
python - gpurun_out/$tag/half_results.db <<'PY' | tee gpurun_out/$tag/stats.txt
import sqlite3, sys
from collections import defaultdict
rows = list(sqlite3.connect(sys.argv[1]).cursor().execute("select name, start, end, queue_id from kernels order by start"))
sub = rows[int(len(rows) * 0.6):]
d = defaultdict(list)
for name, s, e, q in sub: d[name[:64]].append((e - s) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:14]: print("%-66s n=%5d avg %8.1f us" % (k, len(v), sum(v) / len(v)))
idx = [i for i, r in enumerate(rows) if 'k_loss_pass2_rays' in r[0]]
a, b = idx[-3], idx[-2]
t0 = rows[a][1]
for r in rows[a:b + 1]: print("%8.1f %8.1f %7.1f q%s %s" % ((r[1] - t0) / 1e3, (r[2] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3], r[0][:56]))
PY
