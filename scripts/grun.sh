#!/bin/bash
# grun.sh TAG TIMEOUT 'command' -- runs `command` on a GPU box from a frozen copy of the tree.
# gpurun snapshots /root/repo when the box is acquired (minutes after the call starts, while it queues), so
# edits made meanwhile would travel half-done.  This copies the tree to .snap/TAG first (git-ignored, shipped)
# and runs the command from there; $OUT is the repo's gpurun_out (merged back by gpurun).
set -e
TAG=$1; TMO=$2; CMD=$3
ROOT=$(cd "$(dirname "$0")/.." && pwd)
rm -rf "$ROOT/.snap/$TAG"
mkdir -p "$ROOT/.snap/$TAG" "$ROOT/gpurun_out"
# (.gpurunignore: CPU-only files gpurun's gate must not see -- the host-sanitizer script and tests)
(cd "$ROOT" && tar --exclude=./.git --exclude=./gpurun_out --exclude=./.snap --exclude=__pycache__ \
      --exclude=.pytest_cache $(sed 's|^|--exclude=./|' .gpurunignore 2>/dev/null) -cf - .) | tar -xf - -C "$ROOT/.snap/$TAG"
exec /usr/local/graft/bin/gpurun --timeout "$TMO" -- "export OUT=\$GRAFT_REPO_ROOT/gpurun_out; mkdir -p \$OUT; cd .snap/$TAG && $CMD"
