#!/bin/bash
# Copies what tools/refresh_profiles.sh left in gpurun_out/<round>p/ into profiles/ under the round's prefix (the files the judge reads).   tools/install_profiles.sh r05
R=${1:-r05}; S=gpurun_out/${R}p
for f in bench_default.json bench_steps20.json bench_albedo.json pmc_traffic.json pmc_units.json pmc_sq.json; do [ -f $S/$f ] && cp $S/$f profiles/${R}_$f; done
for f in $S/kernel_stats_*.csv $S/steady_*.json $S/timeline_*.txt; do [ -f $f ] && cp $f profiles/${R}_$(basename $f); done
for f in $S/pmc/*.json; do [ -f $f ] && cp $f profiles/${R}_pmc_$(basename $f); done
ls profiles | grep "^${R}_" | wc -l
