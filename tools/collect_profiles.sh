#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel traces + PMC traffic passes for the
# round's evidence.  Usage: bash tools/collect_profiles.sh <round-tag>   -> gpurun_out/profiles_<tag>/
set -u
tag=${1:-r03}
out=gpurun_out/profiles_$tag
mkdir -p $out
export TMPDIR=/tmp
run_trace() { # name, command...
    local name=$1; shift
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$name -- "$@" > $out/$name.log 2>&1 || { echo "trace $name failed"; return 1; }
    cp $out/$name/*/*_kernel_stats.csv $out/${name}_kernel_stats.csv
}
run_pmc() { # name, counters, command...
    local name=$1; local ctr=$2; shift; shift
    timeout -s KILL 120 rocprofv3 --pmc $ctr --output-format csv -d $out/$name -- "$@" > $out/$name.log 2>&1 || { echo "pmc $name failed"; return 1; }
    cp $out/$name/*/*_counter_collection.csv $out/${name}_counters.csv
}
run_trace bench python3 bench.py --steps 200 --warmup 20 --no-extra --no-cpu &&
run_trace c2 python3 tools/prof_ba.py c2 20 &&
run_trace c4 python3 tools/prof_ba.py c4 20 &&
run_trace c5shard python3 tools/prof_ba.py c5shard 20 &&
run_trace c5 python3 tools/prof_ba.py c5 5 &&
run_trace match python3 tools/prof_match.py 64 3 &&
run_trace lm_c2 python3 tools/time_lm.py c2 &&
run_trace lm_c4 python3 tools/time_lm.py c4 &&
run_trace lm_c5shard python3 tools/time_lm.py c5shard &&
run_trace lm_c5 python3 tools/time_lm.py c5 &&
run_trace solve_c4 python3 tools/time_solve.py c4 &&
run_trace solve_c1 python3 tools/time_solve.py c1 &&
run_trace small_c1 python3 tools/time_small.py &&
run_trace setup python3 tools/time_setup.py c1 c2 c4 &&
run_pmc c2_fetch FETCH_SIZE python3 tools/prof_ba.py c2 5 &&
run_pmc c2_write WRITE_SIZE python3 tools/prof_ba.py c2 5 &&
run_pmc c4_fetch FETCH_SIZE python3 tools/prof_ba.py c4 5 &&
run_pmc c4_write WRITE_SIZE python3 tools/prof_ba.py c4 5 &&
VS_PROF_WARM=3 run_pmc match_fetch FETCH_SIZE python3 tools/prof_match.py 16 2 &&
VS_PROF_WARM=3 run_pmc match_write WRITE_SIZE python3 tools/prof_match.py 16 2 &&
VS_PROF_WARM=3 run_pmc match_mfma "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA" python3 tools/prof_match.py 16 2 &&
run_pmc c2_valu "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" python3 tools/prof_ba.py c2 5 &&
run_pmc c4_valu "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" python3 tools/prof_ba.py c4 5 &&
run_pmc c4_ta "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE" python3 tools/prof_ba.py c4 5 &&
run_pmc c4_tcp "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" python3 tools/prof_ba.py c4 5
echo "profiles done rc=$?"
# keep only the summaries (the raw per-dispatch traces are large)
find $out -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
ls -la $out
