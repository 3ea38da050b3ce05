#!/bin/bash
# Runs ON the GPU box: the C++ entry point under rocprofv3 --kernel-trace, several processes; per process the time per update, the hardware
# queue of each kernel family and the mean duration of the step's kernels over the second half of the run (what differs between a 0.173-ms and a
# 0.207-ms process?)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
N=${1:-6}
cd /tmp && export TMPDIR=/tmp
for i in $(seq 1 $N); do
  OUT=$ROOT/gpurun_out/fq_$i; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --output-format csv -d $OUT -- $ROOT/examples/_build/train_selfplay 400000 /tmp/model.bin 8192 --replay 1048576 --minibatch 8192 --hidden 256,256 --save-every 0 --prefill 300 --seed 0x5EED --json --derive > $OUT/out.json 2> $OUT/err.txt
  python3 - <<PY
import csv, glob, json, collections
d=[json.loads(l) for l in open("$OUT/out.json") if l.startswith("{")][-1]
f=sorted(glob.glob("$OUT/**/*_kernel_trace.csv", recursive=True))[-1]
rows=list(csv.DictReader(open(f)))
rows=rows[len(rows)//2:]
q=collections.defaultdict(set); dur=collections.defaultdict(list)
for r in rows:
    n=r["Kernel_Name"]
    key=("env" if "env_kernel" in n else "tail39" if "td_tail_kernel<39" in n else "tail30" if "td_tail_kernel<30" in n else "screen" if "screen_top2" in n else
         "sel_gemm" if "gemm_f32_kernel" in n else "fwd" if "gemm_fwd_persistent" in n else "refine" if "qmax_refine2" in n else "sgd" if "sgd_segments" in n else
         "l0" if "l0_forward" in n else None)
    if key:
        q[key].add(r["Queue_Id"]); dur[key].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
sg=[int(r["End_Timestamp"]) for r in rows if "sgd_segments" in r["Kernel_Name"]]
step=(sg[-1]-sg[0])/1e3/(len(sg)-1)
print("run $i: %.4f ms per update untraced-clock (%d updates); traced step %.1f us;" % (1e3*d["loop_seconds"]/d["updates"], d["updates"], step),
      " ".join("%s q%s %.1f" % (k, "/".join(sorted(q[k])), sum(v)/len(v)) for k,v in sorted(dur.items())))
PY
  python3 $ROOT/tools/step_timeline.py $OUT 3000 1 2>/dev/null | cut -c1-110
  find $OUT -name "*_kernel_trace.csv" -delete
done
