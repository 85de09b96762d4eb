#!/bin/bash
# GPU box, measurement aid (nothing is shipped): which kernels' packed fp32 instructions break the determinism guards?
# Builds the library from a rewritten copy of csrc/ in which every __global__ kernel carries a macro of its group (RAY LOSS OCC FWD FBS DW SC ADAM MISC), with
# __attribute__((target("packed-fp32-ops"))) on ONE group at a time (the shipped flags keep the feature off everywhere else), and runs the overlapped-backward /
# overlapped-march guards on each build.   bash tools/packed_fp32_bisect.sh [groups...]  -> gpurun_out/pkbisect/result.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/pkbisect; mkdir -p $O; : > $O/result.txt
L=rnb-neus2_amd/librnb_neus2_hip.so
cp $L /tmp/lib_shipped.so
rm -rf /tmp/pkb && mkdir -p /tmp/pkb/rnb-neus2_amd && cp -r rnb-neus2_amd/csrc rnb-neus2_amd/host /tmp/pkb/rnb-neus2_amd/ && cp -r include /tmp/pkb/include
python - <<'PY'
import re, glob
groups = [("RAY", r"k_march|k_scan_rays|k_ray_constants|k_coarse_bitfield"), ("LOSS", r"k_loss|k_scan_compact|k_reduce_losses"), ("OCC", r"k_grid_samples|k_ema_mean|k_bitfield|k_pool|k_scan_blocks|k_scan_add"),
          ("FWD", r"k_forward|k_point_query"), ("FBS", r"k_fwd_bwd|k_rgb_fwd_bwd"), ("DW", r"k_dw"), ("SC", r"k_grid_scatter"), ("ADAM", r"k_adam")]
count = {}
for f in glob.glob("/tmp/pkb/rnb-neus2_amd/csrc/*"):
    s = open(f).read()
    def rep(m):
        name = m.group(2)
        g = next((g for g, pat in groups if re.match(pat, name)), "MISC")
        count[g] = count.get(g, 0) + 1
        return "PK_%s __global__%s" % (g, m.group(1))
    s2 = re.sub(r"__global__((?:(?!__global__)[^;{])*?void\s+(k_\w+)\s*\()", rep, s)
    if f.endswith("common.cuh"):
        s2 = "".join("#ifndef PK_%s\n#define PK_%s\n#endif\n" % (g, g) for g in [g for g, _ in groups] + ["MISC"]) + s2
    open(f, "w").write(s2)
print("kernels per group:", count)
PY
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -Xclang -target-feature -Xclang -packed-fp32-ops"
# a group name with the suffix +W0 adds -mllvm -amdgpu-waitcnt-forcezero=1 (an s_waitcnt 0 in front of every instruction: does a missing wait explain it?)
for g in ${@:-NONE RAY LOSS OCC FWD FBS DW SC ADAM MISC}; do
  gg=${g%+W0}; extra=""; [ "$gg" != "$g" ] && extra="-mllvm -amdgpu-waitcnt-forcezero=1"
  def=""; [ $gg != NONE ] && def="-DPK_$gg=__attribute__((target(\"packed-fp32-ops\")))"
  hipcc $FLAGS $extra "$def" -o /tmp/lib_$g.so /tmp/pkb/rnb-neus2_amd/csrc/rnb_neus2_hip.hip 2>/tmp/pkb/build_$g.log || { echo "$g: build failed"; grep -v "not a recognized" /tmp/pkb/build_$g.log | head -5; continue; } 
  cp /tmp/lib_$g.so $L
  n=$(tools/kernel_resources.sh --check-no-pk-f32 2>/dev/null | grep -o "[0-9]*$")
  timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -k "overlapped_backward_equals_serial or overlapped_march" > /tmp/pkb/test_$g.log 2>&1
  r=$(grep -E "passed|failed" /tmp/pkb/test_$g.log | tail -1)
  f=$(grep "^FAILED" /tmp/pkb/test_$g.log | sed 's/.*:://' | cut -d' ' -f1 | tr '\n' ' ')
  echo "packed fp32 in $g only: $n v_pk_*_f32 | $r | failed: ${f:-none}" | tee -a $O/result.txt
  [ -n "$PK_DIFF" ] && timeout 900 python tools/packed_fp32_diff.py $PK_DIFF > $O/diff_$g.txt 2>&1
done
cp /tmp/lib_shipped.so $L
