#!/bin/bash
# GPU box, measurement aid (nothing is shipped): is the packed-fp32 deviation of the overlapped schedule an UNINITIALISED LDS READ?
#  1. the shipped flags + -DRNB_POISON_LDS (kernels_net.cuh: poison_lds fills the MFMA kernels' dynamic LDS with NaN patterns at entry): the stage-parity files and the overlap guards
#     must pass unchanged -- a kernel that reads a tile / padding column / weight slot it never wrote would show NaNs or other values
#  2. packed fp32 in the evaluation kernels (the build of tools/packed_fp32_bisect.sh, group FWD) with and without the poison: tools/packed_fp32_diff.py -- if the deviation of
#     d sdf / dy changes with the poison or NaNs appear, it reads LDS it did not write; if it is the same rows by the same amounts, it does not
#   bash tools/lds_poison_check.sh  ->  gpurun_out/r06_poison/result.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_poison; mkdir -p $O; : > $O/result.txt
L=rnb-neus2_amd/librnb_neus2_hip.so
cp $L /tmp/lib_shipped.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -Xclang -target-feature -Xclang -packed-fp32-ops"
echo "== 1. shipped flags + poison" | tee -a $O/result.txt
hipcc $FLAGS -DRNB_POISON_LDS -o /tmp/lib_poison.so rnb-neus2_amd/csrc/rnb_neus2_hip.hip 2>/tmp/poison_build.log || { echo "build failed"; grep -v "not a recognized" /tmp/poison_build.log | head; exit 1; }
cp /tmp/lib_poison.so $L
RNB_GPU_ISOLATE=0 timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_half_mode.py tests/test_gpu_deterministic.py -m gpu -q -p no:cacheprovider > $O/poison_stage_tests.log 2>&1
echo "stage parity files on the poisoned build: $(grep -aE ' passed| failed' $O/poison_stage_tests.log | tail -1)" | tee -a $O/result.txt
RNB_GPU_ISOLATE=0 timeout 1200 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -p no:cacheprovider -k "overlapped_backward_equals_serial or overlapped_march or full_size_step or pinned" > $O/poison_fullsize_tests.log 2>&1
echo "full-size guards + oracle comparisons + pinned-state hashes on the poisoned build: $(grep -aE ' passed| failed' $O/poison_fullsize_tests.log | tail -1)" | tee -a $O/result.txt
cp /tmp/lib_shipped.so $L
echo "== 2. packed fp32 in the evaluation kernels, without / with the poison" | tee -a $O/result.txt
rm -rf /tmp/pkb && mkdir -p /tmp/pkb/rnb-neus2_amd && cp -r rnb-neus2_amd/csrc rnb-neus2_amd/host /tmp/pkb/rnb-neus2_amd/ && cp -r include /tmp/pkb/include
python - <<'PY'
import re, glob
groups = [("FWD", r"k_forward|k_point_query")]
for f in glob.glob("/tmp/pkb/rnb-neus2_amd/csrc/*"):
    s = open(f).read()
    def rep(m):
        name = m.group(2)
        g = next((g for g, pat in groups if re.match(pat, name)), "MISC")
        return "PK_%s __global__%s" % (g, m.group(1))
    s2 = re.sub(r"__global__((?:(?!__global__)[^;{])*?void\s+(k_\w+)\s*\()", rep, s)
    if f.endswith("common.cuh"):
        s2 = "#ifndef PK_FWD\n#define PK_FWD\n#endif\n#ifndef PK_MISC\n#define PK_MISC\n#endif\n" + s2
    open(f, "w").write(s2)
PY
for v in plain poison; do
  extra=""; [ $v = poison ] && extra="-DRNB_POISON_LDS"
  hipcc $FLAGS $extra '-DPK_FWD=__attribute__((target("packed-fp32-ops")))' -o /tmp/lib_pk_$v.so /tmp/pkb/rnb-neus2_amd/csrc/rnb_neus2_hip.hip 2>/tmp/pkb/build_$v.log || { echo "$v: build failed"; grep -v "not a recognized" /tmp/pkb/build_$v.log | head -5; continue; }
  cp /tmp/lib_pk_$v.so $L
  n=$(tools/kernel_resources.sh --check-no-pk-f32 2>/dev/null | grep -o "[0-9]*$")
  timeout 900 python tools/packed_fp32_diff.py 8 > $O/diff_$v.txt 2>&1
  echo "packed evaluation kernels ($n v_pk_*_f32), $v:" | tee -a $O/result.txt
  grep -a "^rep\|^serial" $O/diff_$v.txt | cut -c1-330 | tee -a $O/result.txt
done
cp /tmp/lib_shipped.so $L
