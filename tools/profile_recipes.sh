#!/bin/bash
# Kernel traces of the other recipes on the GPU box (through gpurun, from the repo root):
#   tools/profile_recipes.sh r03_v1 [cascaded_plus hybrid_plus_large trainable]
# -> gpurun_out/profiles/<tag>_<recipe>_{kernel_stats.csv,section.txt,run.json}; the rocpd databases stay in /tmp on the box.
set -e -o pipefail
tag=$1; shift
recipes=${*:-cascaded_plus hybrid_plus_large trainable}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/profiles; mkdir -p $out
for r in $recipes; do
    case $r in
        trainable) args="--model base --trainable" ;;
        *) args="--model $r" ;;
    esac
    rm -rf /tmp/p_$r
    rocprofv3 --kernel-trace --stats -d /tmp/p_$r -o t -- python3 bench.py --steps 5 --warmup 2 --cpu-utts 0 --no-recall $args \
        > $out/${tag}_${r}_run.json 2> /tmp/p_$r.err
    db=$(find /tmp/p_$r -name "*.db" | head -1)
    python3 tools/rocpd_section.py "$db" --step 4 --top 60 > $out/${tag}_${r}_section.txt
    python3 tools/rocpd_timeline.py "$db" --step 4 --csv $out/${tag}_${r}_kernel_stats.csv > $out/${tag}_${r}_tail_timeline.txt
    head -1 $out/${tag}_${r}_section.txt
done
