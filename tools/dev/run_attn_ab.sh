cd /root/repo
for w in 0 1 2 0 1 2; do echo "== split=$w"; ST_ATTN_BWD_SPLIT=$w python tools/bench_kernels.py attn 2>&1 | grep "bwd all enc self"; done
