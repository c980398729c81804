#!/bin/bash
# Dev: what the language-level agent-scope release / acquire fences cost in the last-arriver merges (csrc/st_common.cuh,
# build switch ST_MERGE_FENCE=1): the step as shipped, then the same box with the fenced build (the stress test runs on both).
export TMPDIR=/tmp; cd /root/repo
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-train-mode --no-decode --no-dp-probe"
echo "== as shipped"; $B 2>/dev/null | python -c "import sys, json; j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', j['ms_per_step'], {k: v['ms_per_step'] for k, v in j['kernels'].items() if k in ('row_chain_bwd_dec', 'row_chain_dec', 'st_grad_norm', 'gemm_dgrad')})"
export ST_MERGE_FENCE=1
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
echo "== ST_MERGE_FENCE=1"; $B 2>/dev/null | python -c "import sys, json; j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', j['ms_per_step'], {k: v['ms_per_step'] for k, v in j['kernels'].items() if k in ('row_chain_bwd_dec', 'row_chain_dec', 'st_grad_norm', 'gemm_dgrad')})"
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "last_arriver or split or grad_norm or beam" 2>&1 | tail -2
