#!/bin/bash
# Round 3, lease E: window plane on / off, alternated four times on the headline and on the small reset-heavy shard (after
# k_consume's LDS-staged plane build); HBM counters of the step kernels on the encoded 1M workload, both ways.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03e
mkdir -p $OUT
cd $REPO
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden_trace or every_registered_level or odd_batch or autoreset or checkpoint or per_env_reset or start or Carrying or manyenvs or render_current" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict($1, ms_per_step=d['ms_per_step'], value=d['value'], parity=(d['parity'] or {}).get('mismatches_all_ranks'), kernels=d['roofline']['kernel_avg_ms'], frac=d['roofline']['frac'], fill_GBs=d['roofline']['achievable']['fill_GBs'])))"; }
for rep in 1 2 3 4; do
  for vp in 1 0; do
    BBAI_VPLANE=$vp timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --parity-envs 128 --min-seconds 1.0 2>>$OUT/ab.err | line "vplane=$vp, config='boss_pixel_1M'" >> $OUT/vplane_ab2.jsonl
    BBAI_VPLANE=$vp timeout 300 python bench.py --config C2 --steps 256 --warmup 16 --no-cpu-baseline --parity-envs 128 --min-seconds 0.5 2>>$OUT/ab.err | line "vplane=$vp, config='C2'" >> $OUT/vplane_ab2.jsonl
    BBAI_VPLANE=$vp timeout 300 python bench.py --config C5-shard --steps 64 --warmup 8 --no-cpu-baseline --parity-envs 128 --min-seconds 0.5 2>>$OUT/ab.err | line "vplane=$vp, config='C5-shard'" >> $OUT/vplane_ab2.jsonl
  done
done
cat $OUT/vplane_ab2.jsonl
cd /tmp && export TMPDIR=/tmp
for vp in 1 0; do
  for c in FETCH_SIZE WRITE_SIZE; do
    BBAI_VPLANE=$vp timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_vp${vp}_$c -o enc -- python $REPO/bench.py --no-pixel --steps 16 --warmup 4 --no-cpu-baseline --parity-envs 0 --min-seconds 0 > $OUT/pmc_vp${vp}_$c.log 2>&1
  done
done
cd $REPO
python - <<PY
import csv, glob, collections, json
res = {}
for vp in (1, 0):
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        rows = collections.defaultdict(list)
        for f in glob.glob("$OUT/pmc_vp%d_%s/**/*counter_collection.csv" % (vp, c), recursive=True):
            for r in csv.DictReader(open(f)):
                if r.get("Counter_Name") == c:
                    rows[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
        for k, v in rows.items():
            if k.startswith("void k_step") or k.startswith("k_consume") or k.startswith("void k_pregen"):
                res.setdefault("vplane=%d %s" % (vp, k), {})[c + "_KB_median"] = sorted(v)[len(v) // 2]
                res["vplane=%d %s" % (vp, k)]["launches"] = len(v)
json.dump(res, open("$OUT/step_counters_boss_encoded_1M.json", "w"), indent=1)
for k, v in res.items():
    print(k[:90], v)
PY
find $OUT -name "*.csv" -size +5M -delete
