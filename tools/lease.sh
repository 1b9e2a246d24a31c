#!/bin/bash
# tools/lease.sh -- everything that runs on the GPU box, ONE script for every lease (gpurun call).
#
#   gpurun --timeout 900 -- 'bash tools/lease.sh <tag> <job> [<job> ...]'
#
# Results go to gpurun_out/<tag>/ (merged back into the build container); what is kept is copied to profiles/<round>/ by hand
# or by tools/summarize_profile.py.  A job is a function name below, optionally with arguments after colons
# (ab:render1m  bench:C2).  Jobs run in the order given; a failing job does not stop the lease.
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
export TMPDIR=/tmp
B="python $REPO/bench.py --steps 20 --warmup 5"
export BBAI_BENCH_LINE=full     # every job parses the FULL record from stdout; `judged` / `rejudged` run the driver's command as the driver does

suite() {            # the GPU test suite (optionally: -k expression)
    cd $REPO
    if [ -n "$1" ]; then timeout 1500 python -m pytest tests -m gpu -x -q -k "$1" > $OUT/gpu_tests.log 2>&1
    else timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1; fi
    echo "pytest rc=$?" >> $OUT/gpu_tests.log
    tail -4 $OUT/gpu_tests.log
}
suiteenv() {         # the GPU suite under an environment setting, all failures listed (no -x): suiteenv:BBAI_INPLACE=1[:-k expression]
    cd $REPO
    env "$1" timeout 1500 python -m pytest tests -m gpu -q ${2:+-k "$2"} > $OUT/gpu_tests_$1.log 2>&1
    echo "pytest rc=$?" >> $OUT/gpu_tests_$1.log
    grep -E "^FAILED|^ERROR" $OUT/gpu_tests_$1.log | head -40
    tail -3 $OUT/gpu_tests_$1.log
}
build() { cd $REPO && python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -3; }
judged() {           # the driver's command
    cd /tmp && env -u BBAI_BENCH_LINE timeout 600 $B --full-out $OUT/bench_boss_pixel_1M.json > $OUT/bench_boss_pixel_1M.line.json 2> $OUT/bench_boss_pixel_1M.err
    echo "judged line: $(wc -c < $OUT/bench_boss_pixel_1M.line.json) bytes, $(wc -l < $OUT/bench_boss_pixel_1M.line.json) line(s)"
    python $REPO/tools/summarize_profile.py --line $OUT/bench_boss_pixel_1M.json
}
rejudged() {         # after trace + pmc: condense them ON THE BOX (profiles/pmc_latest.json of this copy) and run the driver's command once
                     # more, so that the line quotes the counter passes of its own sources
    cd $REPO && python tools/summarize_profile.py $TAG > $OUT/summarize.log 2>&1
    mkdir -p $OUT/condensed && cp profiles/$TAG/* profiles/pmc_latest.json $OUT/condensed/ 2>/dev/null
    rm -rf $OUT/stats $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq
    cd /tmp && env -u BBAI_BENCH_LINE timeout 600 $B --full-out $OUT/bench_boss_pixel_1M_with_traffic.json > $OUT/bench_boss_pixel_1M_with_traffic.line.json 2> $OUT/bench_boss_pixel_1M_with_traffic.err
    echo "judged line: $(wc -c < $OUT/bench_boss_pixel_1M_with_traffic.line.json) bytes"
    python $REPO/tools/summarize_profile.py --line $OUT/bench_boss_pixel_1M_with_traffic.json | head -1
}
trace() {            # rocprofv3 --kernel-trace --stats of the same command
    cd /tmp && rm -rf $OUT/stats
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o boss -- $B --no-cpu-baseline --parity-envs 0 --no-extra-configs \
        > $OUT/bench_boss_pixel_1M_under_rocprof.json 2> $OUT/rocprof_stats.log
    find $OUT -name "*kernel_trace.csv" -size +20M -delete
}
pmc() {              # FETCH_SIZE / WRITE_SIZE, one pass each, no other tracing (gpurun refuses the combination)
    cd /tmp && rm -rf $OUT/pmc_fetch $OUT/pmc_write
    timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o boss -- $B --no-cpu-baseline --parity-envs 0 --min-seconds 0 --no-extra-configs > $OUT/rocprof_fetch.log 2>&1
    timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o boss -- $B --no-cpu-baseline --parity-envs 0 --min-seconds 0 --no-extra-configs > $OUT/rocprof_write.log 2>&1
    find $OUT -name "*kernel_trace.csv" -size +20M -delete
}
bench() {            # one BASELINE config as its own line: bench:C2 [extra args]
    cd /tmp && timeout 600 python $REPO/bench.py --config $1 --no-extra-configs ${@:2} > $OUT/bench_$1.json 2>> $OUT/bench.err
    python $REPO/tools/summarize_profile.py --line $OUT/bench_$1.json
}
benchenv() {         # one BASELINE config under an environment setting: benchenv:BBAI_GATE_STRICT=1:C4[:extra bench args]
    local kv=$1 cfg=$2; shift 2
    cd /tmp && env ${kv//,/ } timeout 600 python $REPO/bench.py --config $cfg --no-extra-configs --no-cpu-baseline ${@} > $OUT/bench_${cfg}_$kv.json 2>> $OUT/bench.err      # (several settings: A=1,B=2)
    python - <<PY
import json
d = json.loads(open("$OUT/bench_${cfg}_$kv.json").read().strip().splitlines()[-1])
t = d["timing"]
r = d["roofline"]
print("$kv $cfg ms/step mean %.4f median %.4f max/med %.2f k_step per step %.4f parity %s gate_timeouts %s" % (d["ms_per_step"], t["block_ms"]["median"] / d["steps"], t["max_over_median"],
      r["kernel_avg_ms"].get("k_step", 0) / (r.get("steps_per_launch") or 1.0), (d.get("parity") or {}).get("mismatches_all_ranks"), d.get("gate_timeouts")))
PY
}
ab() {               # tools/ab.py presets
    cd /tmp
    case $1 in
    render)     # ab:render:<envs>:<steps per block>  -- the render queue shapes of render_launch against the one-shot shape, in the step loop
                timeout 600 python $REPO/tools/ab.py --tag $TAG --level BossLevel --envs ${2:-1048576} --pixel --steps ${3:-20} --blocks 8 --reps 3 \
                    --base render_queue_bpc=0,render_queue_blocks=0 --settings render_queue=0 render_queue=1 render_queue=2 render_queue=3 render_queue=4 render_queue=5 \
                    render_queue=6 render_queue=7 render_queue=8 render_queue=10 render_queue=11 render_queue=1,render_queue_bpc=2 render_queue=1,render_queue_blocks=224 \
                    > $OUT/render_queue_ab_${2:-1048576}.jsonl 2>> $OUT/ab.err; tail -1 $OUT/render_queue_ab_${2:-1048576}.jsonl ;;
    pace)       # ab:pace[:<envs>:<steps>] -- time-gated render tickets (option render_pace, 1/16 ns per ticket, two counters) against the shipped one-counter shape
                timeout 600 python $REPO/tools/ab.py --tag $TAG --level BossLevel --envs ${2:-1048576} --pixel --steps ${3:-20} --blocks 8 --reps 3 \
                    --base render_queue=-1,render_queue_bpc=0,render_queue_blocks=0,render_pace=0 --settings render_pace=0 render_queue=3,render_pace=0 \
                    render_pace=183 render_pace=180 render_pace=178 render_pace=176 render_pace=174 \
                    > $OUT/render_pace_ab_${2:-1048576}.jsonl 2>> $OUT/ab.err; tail -1 $OUT/render_pace_ab_${2:-1048576}.jsonl ;;
    *)          # ab:<name>:<level>:<envs>:<steps>:<pixel 0|1>:<setting>:<setting>...   (settings use '/' for ',')
                local name=$1 level=$2 envs=$3 steps=$4 pix=$5; shift 5
                local sets=(); for s in "$@"; do sets+=("${s//\//,}"); done
                timeout 600 python $REPO/tools/ab.py --tag $TAG --level $level --envs $envs $([ "$pix" = 1 ] && echo --pixel) --steps $steps --blocks 8 --reps 3 --base "$AB_BASE" \
                    --settings "${sets[@]}" > $OUT/ab_$name$AB_SUFFIX.jsonl 2>> $OUT/ab.err; tail -1 $OUT/ab_$name$AB_SUFFIX.jsonl ;;
    esac
}
sq() {               # SQ counter pass over the headline bench: where the waves of k_render / k_step spend their cycles
    cd /tmp && rm -rf $OUT/pmc_sq
    timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU \
        --kernel-trace --output-format csv -d $OUT/pmc_sq -o boss -- python $REPO/bench.py --steps 8 --warmup 2 --no-cpu-baseline --parity-envs 0 --min-seconds 0 --no-extra-configs ${@} > $OUT/rocprof_sq.log 2>&1
    python - <<PY | tee $OUT/sq_counters.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob("$OUT/pmc_sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if k.startswith(("k_render", "k_step", "k_consume")):
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: round(sorted(v)[len(v) // 2]) for c, v in d.items()}, "launches", len(next(iter(d.values()))))
PY
    find $OUT -name "*.csv" -size +20M -delete
}
botbench() {         # expert-driven env-steps/s per config (tools/bot_bench.py)
    cd /tmp
    for cfg in "GoToLocal 65536 200" "PickupLoc 262144 100" "GoTo 131072 100" "BossLevel 262144 60" "BossLevel 1048576 30"; do
        timeout 300 python $REPO/tools/bot_bench.py $cfg 2>> $OUT/bot_bench.err | tail -1 | tee -a $OUT/bot_bench.jsonl
    done
}
botab() {            # the expert's kernels side by side on one box: lane = env (k_bot) against lane groups (k_botg), same episodes expected
    cd /tmp
    for cfg in "BossLevel 1048576 30" "GoToLocal 65536 200" "PickupLoc 262144 100" "GoTo 131072 100"; do
        for mode in ${BOTAB_MODES:-0 16}; do       # lanes per env (0: lane = env)
            BBAI_BOT_GROUP=$mode timeout 300 python $REPO/tools/bot_bench.py $cfg 2>> $OUT/bot_ab.err | tail -1 | tee -a $OUT/bot_ab.jsonl
        done
    done
}
botpmc() {           # where the expert's waves spend their cycles: SQ + instruction-cache counters of k_bot / k_botg (BossLevel 262 144, 10 steps)
    cd /tmp
    rocprofv3 -L 2>/dev/null | grep -o -i -E "\b(SQC?_[A-Z_]*(ICACHE|IFETCH)[A-Z_]*|SQ_INSTS_[A-Z_]+)\b" | sort -u > $OUT/counters_avail.txt
    for mode in ${BOTAB_MODES:-0 16}; do
        n=0
        for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES" \
                   "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
            n=$((n + 1))
            BBAI_BOT_GROUP=$mode timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/botpmc_${mode}_$n -o bot \
                -- python $REPO/tools/bot_bench.py BossLevel 262144 10 > $OUT/botpmc_${mode}_$n.log 2>&1
        done
    done
    python - <<PY | tee $OUT/botpmc.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob("$OUT/botpmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if k.startswith("k_bot"):
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items()):
    print(k, {c: round(sorted(v)[len(v) // 2]) for c, v in sorted(d.items())}, "launches", len(next(iter(d.values()))))
PY
    for f in $OUT/botpmc_*_2.log; do tail -n 1 $f; done
    rm -rf $OUT/botpmc_*_[12]
}
botprof() {          # phase timers of the expert (tools/bot_prof.py) on the -DBBAI_BOT_PROF build that travelled in tools/_prof/
    cd /tmp
    for cfg in "BossLevel 262144 20" "GoToLocal 65536 60"; do
        for mode in ${BOTAB_MODES:-0 16}; do
            BBAI_BOT_GROUP=$mode BBAI_ENGINE_LIB=$REPO/tools/_prof/libbbai_botprof.so timeout 300 python $REPO/tools/bot_prof.py $cfg 2>> $OUT/bot_prof.err | tail -1 | tee -a $OUT/bot_prof.jsonl
        done
    done
}
soak() {             # scattered envs of large batches vs the oracle over many steps (tools/gpu_soak.py)
    cd $REPO && timeout 900 python - > $OUT/soak_random.txt 2>&1 <<PY
import sys
sys.path.insert(0, "$REPO/tools"); sys.path.insert(0, "$REPO")
import gpu_soak
bad = 0
for level, n, T in (("BossLevel", 1048576, 200), ("GoToLocal", 65536, 400), ("PickupLoc", 262144, 300), ("GoTo", 131072, 300), ("PutNextS5N2Carrying", 65536, 300), ("KeyInBox", 65536, 300), ("SynthSeq", 131072, 200)):
    bad += gpu_soak.soak(level, n, T, 48, 12345)
print("soak mismatches:", bad)
PY
    tail -4 $OUT/soak_random.txt
}
soakbot() {          # expert-driven soak: scattered envs against the host build incl. the expert, very high reset rates
    cd $REPO && timeout 900 python tools/gpu_soak.py bot > $OUT/soak_bot.txt 2>&1; tail -7 $OUT/soak_bot.txt
}
demobench() {        # generate_demos: device rollout vs the step-by-step host loop, equal digests (tools/demo_bench.py)
    cd /tmp
    for cfg in "GoToLocal 65536 65536" "BossLevel 32768 32768" "BossLevel 131072 131072"; do
        timeout 600 python $REPO/tools/demo_bench.py $cfg 2>> $OUT/demo_bench.err | tail -1 | tee -a $OUT/demo_bench.jsonl
    done
}
listatomic() {       # one list atomic per stepping wave, sub-lists, and a reader next to it (tools/ubench_listatomic.hip): listatomic[:waves]
    cd $REPO && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_listatomic tools/ubench_listatomic.hip 2>/dev/null
    timeout 200 /tmp/ubench_listatomic ${1:-4096} | tee $OUT/ubench_listatomic_${1:-4096}.jsonl | tail -40
}
genrate() {          # bulk level-generation rate (bbai_seed's first fill) per generator shape: genrate[:<groups ...>]  (tools/gen_rate.py)
    cd /tmp
    for g in ${@:-32 1}; do
        BBAI_PREGEN_GROUP=$g timeout 300 python $REPO/tools/gen_rate.py 2>> $OUT/gen_rate.err | tee -a $OUT/gen_rate_by_group_width.jsonl
    done
}
genlane() {          # bulk level-generation rate, lane-group kernel (BBAI_PREGEN_LANE=0) against lane = level (=1): genlane[:<gen_rate.py args>]
    cd /tmp
    for l in 0 1; do
        BBAI_PREGEN_LANE=$l timeout 300 python $REPO/tools/gen_rate.py ${@} 2>> $OUT/gen_rate.err | tee -a $OUT/gen_rate_lane_vs_group.jsonl
    done
}
genpmc() {           # instruction counters of the level generator's bulk fill (tools/gen_rate.py): VALU / SALU / LDS instructions and cycles per level
    cd /tmp
    for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
        n=$(echo $pass | cut -d' ' -f1)
        rm -rf $OUT/genpmc_$n
        timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/genpmc_$n -o g -- python $REPO/tools/gen_rate.py PickupLoc 262144 8 GoToLocal 65536 16 GoTo 131072 8 BossLevel 262144 4 > $OUT/genpmc_$n.log 2>&1
    done
    python - <<PY | tee $OUT/genpmc.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob("$OUT/genpmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if k.startswith("k_pregen"):
            agg[(k, r.get("Grid_Size", r.get("Grid_Size_X", "?")))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (k, grid), d in sorted(agg.items()):
    print(k, "grid", grid, {c: (len(v), sum(v) / len(v)) for c, v in sorted(d.items())})
PY
    rm -rf $OUT/genpmc_SQ_*
}
genprof() {          # phase profile of the level generator (tools/genprof.hip): genprof[:<level> ...]
    cd $REPO && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/genprof tools/genprof.hip 2>/dev/null
    for l in ${@:-GoToLocal PickupLoc GoTo BossLevel}; do timeout 120 /tmp/genprof $l | tee -a $OUT/genprof.txt; done
}
saluvalu() {         # do scalar-bound and vector-bound waves of a CU add up? (tools/ubench_salu_valu.hip)
    cd $REPO && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_salu_valu tools/ubench_salu_valu.hip 2>/dev/null
    timeout 120 /tmp/ubench_salu_valu | tee $OUT/ubench_salu_valu.jsonl
}
ubench() {           # the microbenchmarks DESIGN.md quotes (built here: hipcc is on the box)
    cd $REPO && for u in gather render fetchcal; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_$u tools/ubench_$u.hip 2>/dev/null; done
    timeout 120 python tools/membw.py > $OUT/membw.json 2> $OUT/membw.err
    timeout 300 /tmp/ubench_render ${1:-1048576} ${2:-queue} > $OUT/ubench_render_${2:-queue}.jsonl 2> $OUT/ubench_render.err
    cat $OUT/membw.json; tail -30 $OUT/ubench_render_${2:-queue}.jsonl
}
halfline() {         # 64-byte half-line gathers against 128-byte line gathers at k_step's shape (tools/ubench_halfline.hip), + the FETCH_SIZE of each
    cd $REPO && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_halfline tools/ubench_halfline.hip 2>/dev/null
    timeout 120 /tmp/ubench_halfline | tee $OUT/ubench_halfline.jsonl
    cd /tmp && rm -rf $OUT/halfline_fetch && timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/halfline_fetch -o h -- /tmp/ubench_halfline > /dev/null 2> $OUT/halfline_fetch.log
    python - <<PY | tee $OUT/ubench_halfline_fetch.txt
import csv, glob, collections
acc = collections.defaultdict(list)
for path in glob.glob("$OUT/halfline_fetch/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") == "FETCH_SIZE":
            acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    v.sort()
    print("%-28s launches=%d FETCH_SIZE median=%.0f KB (x 2 on gfx950 = %.1f MB = %.1f B per lane)" % (k, len(v), v[len(v) // 2], 2 * v[len(v) // 2] / 1024, 2 * v[len(v) // 2] * 1024 / 1048576))
PY
    rm -rf $OUT/halfline_fetch
}
lib() {              # the same ab spec under two engine builds, alternated twice: lib:<other.so>:<ab args as for ab:>
    local other=$1; shift
    for rep in 1 2; do
        echo "--- default build"; AB_SUFFIX=_default_$rep ab "$@"
        echo "--- $other"; BBAI_ENGINE_LIB=$REPO/$other AB_SUFFIX=_other_$rep ab "$@"
    done
}
loopab() {           # bench.py's block loop: one bbai_rollout call per block against per-step calls from Python, per config, alternated twice
    cd /tmp
    for rep in 1 2; do for cfg in C2 C3 C4-shard C4 C5; do for mode in "" "--rollout-entry"; do
        timeout 300 python $REPO/bench.py --config $cfg --no-extra-configs --no-cpu-baseline --parity-envs 64 --min-seconds 0.4 $mode 2>> $OUT/loopab.err | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'config': '$cfg', 'loop': 'bbai_rollout' if '$mode' else 'python', 'ms_per_step': round(d['ms_per_step'], 5), 'value': d['value'], 'kernels': d['roofline']['kernel_avg_ms'], 'parity': (d['parity'] or {}).get('mismatches_all_ranks')}))" | tee -a $OUT/loop_ab.jsonl
    done; done; done
}
profcfg() {          # profcfg:<config>[:<bench args>] -- the evidence set of ONE BASELINE workload on the shipped sources: rocprofv3 --kernel-trace --stats,
                     # FETCH_SIZE and WRITE_SIZE passes (each on its own: gpurun refuses counter + trace-domain mixes), and the SQ pass; condensed by
                     # tools/summarize_profile.py into profiles/<tag>/ and profiles/pmc_latest.json (one entry per workload)
    local cfg=$1; shift
    local B1="python $REPO/bench.py --config $cfg --no-extra-configs --no-cpu-baseline --parity-envs 0 --min-seconds 0.25 --max-blocks 8 --steps 192 $@"      # (192: whole look-ahead windows of 16 / 64 / 96 ticks -- every k_step_ticks launch of a block takes the same number of steps)
    cd /tmp && rm -rf $OUT/stats_$cfg $OUT/pmc_fetch_$cfg $OUT/pmc_write_$cfg $OUT/pmc_sq_$cfg
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$cfg -o t -- $B1 > $OUT/bench_${cfg}_under_rocprof.json 2> $OUT/rocprof_stats_$cfg.log
    timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_$cfg -o t -- $B1 > /dev/null 2> $OUT/rocprof_fetch_$cfg.log
    timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_$cfg -o t -- $B1 > /dev/null 2> $OUT/rocprof_write_$cfg.log
    timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_ACTIVE_INST_VALU \
        --kernel-trace --output-format csv -d $OUT/pmc_sq_$cfg -o t -- $B1 > /dev/null 2> $OUT/rocprof_sq_$cfg.log
    cd $REPO && python tools/summarize_profile.py $TAG --config $cfg 2>&1 | tail -12
    mkdir -p $OUT/condensed && cp profiles/$TAG/* profiles/pmc_latest.json $OUT/condensed/ 2>/dev/null     # (the box's profiles/ does not travel back: gpurun_out/ does,
    rm -rf $OUT/stats_$cfg $OUT/pmc_fetch_$cfg $OUT/pmc_write_$cfg $OUT/pmc_sq_$cfg                          #  up to 64 MiB: the raw passes stay on the box)
}
benchlib() {         # one BASELINE config on another engine build: benchlib:<name>:<path to .so relative to the repo>:<config>[:bench args] -> bench_<config>_<name>.json
    local name=$1 so=$2 cfg=$3; shift 3
    cd /tmp && BBAI_ENGINE_LIB=$REPO/$so timeout 600 python $REPO/bench.py --config $cfg --no-extra-configs --no-cpu-baseline ${@} > $OUT/bench_${cfg}_$name.json 2>> $OUT/bench.err
    python - <<PY
import json
d = json.loads(open("$OUT/bench_${cfg}_$name.json").read().strip().splitlines()[-1])
t = d["timing"]
print("$name $cfg ms/step mean %.4f median %.4f max/med %.2f kernels %s parity %s" % (d["ms_per_step"], t["block_ms"]["median"] / d["steps"], t["max_over_median"],
      d["roofline"]["kernel_avg_ms"], (d.get("parity") or {}).get("mismatches_all_ranks")))
PY
}
tracecfg() {         # rocprofv3 kernel trace of one BASELINE config (tracecfg:C2): launch gaps on the small shards
    cd /tmp && rm -rf $OUT/trace_$1
    timeout 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_$1 -o t -- python $REPO/bench.py --config $1 --no-extra-configs --no-cpu-baseline --parity-envs 0 --min-seconds 0.2 --steps 256 --warmup 16 > $OUT/trace_$1.json 2> $OUT/trace_$1.log
    python - <<PY | tee $OUT/trace_$1_gaps.txt
import csv, glob
rows = []
for path in glob.glob("$OUT/trace_$1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")))
rows.sort()
steady = [r for r in rows if r[2].startswith(("k_step", "k_consume", "k_tap", "k_pregen"))]
main = [r for r in steady if not r[2].startswith("k_pregen")]
main = main[len(main) // 2:]                       # the second half of the run: steady state
import collections
dur, gap = collections.defaultdict(list), collections.defaultdict(list)
for a, b in zip(main, main[1:]):
    dur[a[2]].append((a[1] - a[0]) / 1e3)
    gap[a[2] + " -> " + b[2]].append((b[0] - a[1]) / 1e3)
med = lambda v: sorted(v)[len(v) // 2]
for k, v in dur.items(): print("kernel %-28s n=%5d median %.2f us" % (k, len(v), med(v)))
for k, v in gap.items(): print("gap    %-44s n=%5d median %.2f us" % (k, len(v), med(v)))
steps = [r for r in main if r[2].startswith("k_step")]
if len(steps) > 2: print("step period median %.2f us" % med([(b[0] - a[0]) / 1e3 for a, b in zip(steps, steps[1:])]))
PY
    find $OUT -name "*.csv" -size +20M -delete
}
timeline() {         # timeline:<config>:<A=1,B=2 settings or ->[:<bench args>] -- every kernel of ~3 windows in the middle of the run, start / end in us, with its queue
    local cfg=$1 kv=$2; shift 2
    local name=timeline_${cfg}_${kv}
    cd /tmp && rm -rf $OUT/$name
    env ${kv//,/ } DUMMY=1 timeout 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/$name -o t -- python $REPO/bench.py --config $cfg --no-extra-configs --no-cpu-baseline --parity-envs 0 --min-seconds 0.2 --max-blocks 6 $@ > $OUT/$name.json 2> $OUT/$name.log
    python - <<PY | tee $OUT/$name.txt
import csv, glob
rows = []
for path in glob.glob("$OUT/$name/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""), r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
steps = [i for i, r in enumerate(rows) if r[2].startswith("k_step")]
if steps:
    mid = steps[len(steps) * 2 // 3]
    t0 = rows[mid][0]
    sel = [r for r in rows if t0 <= r[0] <= t0 + 2500000][:140]
    for a, b, k, q, st in sel:
        print("%9.1f %9.1f  dur %8.1f us  q%-3s s%-3s %s" % ((a - t0) / 1e3, (b - t0) / 1e3, (b - a) / 1e3, q, st, k))
PY
    rm -rf $OUT/$name
}
run() { cd $REPO && timeout 900 "$@"; }          # run:python:tools/x.py:arg ...

for job in "$@"; do
    IFS=':' read -r -a parts <<< "$job"
    echo "=== $job ($(date +%T))"
    "${parts[@]}"
done
echo "=== done ($(date +%T))"
