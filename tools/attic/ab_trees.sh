#!/bin/bash
# usage (GPU box, repo root): tools/attic/ab_trees.sh <other tree> [rounds] [bench flags]   -- alternating default bench runs of this
# tree and of another built copy of the repo (e.g. tools/var/r03_tree: `git archive <rev> | tar -x`, then its own build): same
# box, same minute -- the only way step times of two revisions compare on this pool (boxes differ by 5 %).
OTHER=${1:?tree}; ROUNDS=${2:-3}; shift 2
one() { local dir=$1 tag=$2; shift 2; (cd $dir && python bench.py --no-cpu-baseline --no-other-configs "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; g=d.get('roofline_gather',{}); print('$tag', 'ms/step', round(d['ms_per_step'],4), 'dense us', round(1e3*r['avg_launch_ms'],1), 'edge us', round(1e3*g.get('avg_launch_ms',0),1))"); }
for i in $(seq $ROUNDS); do one . new "$@"; one $OTHER other "$@"; done
