# Dev: same-box A/B of a SET of source-file variants under an arbitrary command.  usage: ab_cmd.sh <rounds> "<command>" <csrc name>=<variant path> ...
# A = the tree as it is, B = every named file replaced by its variant; alternating; the command's stdout is printed behind the tag.
export TMPDIR=/tmp; cd /root/repo
ROUNDS=$1; CMD=$2; shift 2
C=speech-tranformer-pytorch_amd/csrc
for kv in "$@"; do f=${kv%%=*}; cp $C/$f /tmp/_orig_$f; done
run() { python -c "import __graft_entry__ as g; g.build()" > /tmp/build.log 2>&1 || { echo BUILD FAILED; tail -5 /tmp/build.log; return; }
  echo "== $1"; bash -c "$CMD" 2>&1 | grep -v "amdgpu.ids"; }
for r in $(seq 1 $ROUNDS); do
  for kv in "$@"; do f=${kv%%=*}; cp /tmp/_orig_$f $C/$f; done; run A
  for kv in "$@"; do f=${kv%%=*}; cp ${kv#*=} $C/$f; done; run B
done
for kv in "$@"; do f=${kv%%=*}; cp /tmp/_orig_$f $C/$f; done
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
