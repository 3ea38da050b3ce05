#!/bin/bash
# Runs ON the GPU box (through gpurun): instruction-mix PMC pass + env-only throughput of the fused self-play kernel.
# usage: tools/env_pmc.sh TAG   -> gpurun_out/envpmc_TAG/{pmc/..., env_only_*.json}
set -e
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/envpmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc \
    -- python3 $ROOT/bench.py --env-only --games 8192 --steps 40 --warmup 200 > $OUT/env_only_8192_under_pmc.json 2> $OUT/pmc.err
for g in 8192 131072 1048576; do
    python3 $ROOT/bench.py --env-only --games $g --steps 200 --warmup 300 > $OUT/env_only_$g.json 2> $OUT/env_only_$g.err
done
python3 - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(list)
for f in glob.glob("$OUT/pmc/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "env_kernel<2>" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in agg.items()}
m["launches_sampled"] = len(next(iter(agg.values()))) if agg else 0
if "SQ_WAVES" in m and m["SQ_WAVES"]:
    for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM"):
        m[k + "_per_board"] = m.get(k, 0) / m["SQ_WAVES"]
    m["wave_instructions_per_board"] = sum(m.get(k, 0) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM")) / m["SQ_WAVES"]
json.dump(m, open("$OUT/env_kernel_instruction_mix.json", "w"), indent=1, sort_keys=True)
print(json.dumps(m, indent=1, sort_keys=True))
for g in (8192, 131072, 1048576):
    print(open("$OUT/env_only_%d.json" % g).read().strip())
PY
