#!/bin/bash
# Runs ON the GPU box (through gpurun): the round's closing measurement of ONE configuration with the tree as it is —
# rocprofv3 kernel stats + PMC passes + bench lines (tools/collect_profiles*.sh), a step timeline and the duration percentiles of a traced run,
# and for the headline configuration the driver's own bench command and a sustained run.
# usage: tools/collect_round.sh TAG CONFIG      -> gpurun_out/prof_TAG[_configK]/..., gpurun_out/timeline_TAG[_configK].txt, ...
TAG=$1; CFG=${2:-2}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
SUF=""; [ "$CFG" != "2" ] && SUF="_config$CFG"
if [ "$CFG" = "2" ]; then tools/collect_profiles.sh ${TAG} > gpurun_out/${TAG}_collect.log 2>&1
else tools/collect_profiles_cfg.sh ${TAG}${SUF} $CFG > gpurun_out/${TAG}${SUF}_collect.log 2>&1; fi
# summarise ON the box into gpurun_out/summaries/ and drop the raw counter CSVs (~45 MB per configuration; gpurun carries back 64 MiB per call)
python3 tools/summarize_profiles.py ${TAG}${SUF} $ROOT/gpurun_out/summaries > gpurun_out/${TAG}${SUF}_summarize.log 2>&1
rm -rf gpurun_out/prof_${TAG}${SUF}/fetch gpurun_out/prof_${TAG}${SUF}/write gpurun_out/prof_${TAG}${SUF}/sq gpurun_out/prof_${TAG}${SUF}/sq2 gpurun_out/prof_${TAG}${SUF}/sq3 gpurun_out/prof_${TAG}${SUF}/trace
echo "profiles done: $(date +%T)"
# timeline of one steady-state step + percentiles over the traced run
OUT=$ROOT/gpurun_out/tl_${TAG}${SUF}; mkdir -p $OUT
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/bench.py --config $CFG --steps 30 --warmup 10 --no-cpu-baseline --no-variants --no-facade --repeats 1 > $OUT/bench.json 2> $OUT/err.txt)
python3 tools/step_timeline.py $OUT $([ "$CFG" = "2" ] && echo 200 || echo 60) 1 > gpurun_out/${TAG}_step_timeline_config${CFG}.txt 2>> $OUT/err.txt
python3 tools/kernel_percentiles.py $OUT 20 > gpurun_out/${TAG}_kernel_duration_percentiles_config${CFG}.txt 2>> $OUT/err.txt
find $OUT -name "*_kernel_trace.csv" -delete
echo "timeline done: $(date +%T)"
if [ "$CFG" = "2" ]; then
  python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver_command.json 2> gpurun_out/${TAG}_bench_driver_command.err
  python3 bench.py --sustain 20000 --no-cpu-baseline --no-facade > gpurun_out/${TAG}_sustain_20000_updates.json 2> gpurun_out/${TAG}_sustain.err
  echo "driver + sustain done: $(date +%T)"
fi
tail -c 300 gpurun_out/${TAG}_step_timeline_config${CFG}.txt
