# Dev: same-box comparison of generator variants of the hand-scheduled attention backward.
# usage (on the GPU box): bash tools/dev/attn_bwd64_sweep.sh "" "BWD64_PKMUL=1" "BWD64_WAIT_GROUP=2 BWD64_VALU_FIRST=1" ...
# every argument is one variant (a list of BWD64_* switches of tools/gen_attn_bwd64.py; "" = the committed defaults); each is
# generated, built (only st_attn_bwd64.hip recompiles), checked (CHECK=1) and timed; the defaults are restored at the end.
export TMPDIR=/tmp; cd /root/repo
for V in "$@"; do
  echo "== variant: [$V]"
  env $V python tools/gen_attn_bwd64.py > /tmp/gen.log 2>&1 || { echo "GENERATOR FAILED"; tail -3 /tmp/gen.log; continue; }
  python -c "import __graft_entry__ as g; g.build()" > /tmp/build.log 2>&1 || { echo "BUILD FAILED"; tail -5 /tmp/build.log; continue; }
  if [ -n "$CHECK" ]; then timeout 300 python tools/dev/attn_bwd64_check.py 2>&1 | grep -E "ALL OK|MISMATCH"; fi
  timeout 300 python tools/dev/attn_bwd64_time.py 2>&1 | grep -v amdgpu.ids
done
python tools/gen_attn_bwd64.py > /dev/null 2>&1
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
