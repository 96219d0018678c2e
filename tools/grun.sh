#!/bin/bash
# Build the in-tree library (stale sources -> stale .so on the GPU box otherwise), then hand the command to gpurun.
#   tools/grun.sh <timeout_s> '<command>'
set -e
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()"
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
