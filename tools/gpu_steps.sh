# Runs GPU steps one after another on the GPU box: bash tools/gpu_steps.sh <outdir> "<timeout> <name> <command...>" ...
# A step that fails ordinarily (assertion, non-zero exit) does not stop the following ones; a step that TIMES OUT or is
# killed does (nothing else is started on a GPU that may be wedged).  stdout/stderr of every step go to <outdir>/<name>.log.
OUT=$1; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for spec in "$@"; do
    set -- $spec
    T=$1; NAME=$2; shift 2
    echo "== $NAME (limit ${T}s): $*"
    timeout -k 10 "$T" bash -c "$*" > "$OUT/$NAME.log" 2>&1
    rc=$?
    echo "== $NAME rc=$rc"; tail -3 "$OUT/$NAME.log"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "== $NAME timed out or was killed: stopping here"; exit $rc; fi
done
exit 0
