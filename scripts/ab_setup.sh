#!/bin/sh
# Same-box A/B harness: export HEAD (with its own build) to .ab/old so that one gpurun call can time both trees back to back:
#   sh scripts/ab_setup.sh && gpurun -- '(cd .ab/old && python bench.py ...); python bench.py ...'
# Box-to-box spread of the step time is +-0.5 ms; deltas below that are only visible this way. .ab/ is git-ignored.
set -e
cd "$(dirname "$0")/.."
rm -rf .ab/old && mkdir -p .ab/old
git archive HEAD | tar -x -C .ab/old
(cd .ab/old && python -c "import __graft_entry__ as g; g.build()")
