# Dev: same-box A/B of an environment switch: kernel census of one graph-replayed step for each value of $1 in "${@:2}"
export TMPDIR=/tmp; cd /root/repo
for V in "${@:2}"; do
  env $1=$V rocprofv3 --kernel-trace -d /tmp/abe$V -o t -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train-mode --no-decode > /tmp/g.log 2>&1
  echo "== $1=$V"; python tools/dev/step_segment.py /tmp/abe$V/t_results.db 8 2>&1 | cut -c1-120 | head -${TOPN:-45}
done
