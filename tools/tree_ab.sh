#!/bin/bash
# ON THE GPU BOX: the hot-path bench of TWO TREES alternated on one box (A B A B): this tree against a frozen copy of an earlier one --
#   (here)  mkdir .r05_tree && git archive <commit> -- mucon_amd bench.py oracle include | tar -x -C .r05_tree && (cd .r05_tree && python -m mucon_amd.build)
#   (box)   bash tools/tree_ab.sh .r05_tree [bench args]
# (.r05_tree is git-ignored; its built library travels with the snapshot.)  Prints ms per step and the two event-timed launches per run.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
other=$1; shift
legs="--no-viterbi"
[ "${WITH_E2E:-0}" = "1" ] && legs=""     # WITH_E2E=1: the end-to-end / evaluation legs too
for rnd in 1 2; do
  for t in "$other" .; do
    (cd "$t" && python3 bench.py --steps 100 --warmup 10 --repeats 3 $legs --no-cpu-baseline --no-traffic "$@" 2>/dev/null) | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
e = (d.get('end_to_end', {}).get('ms_per_video'), d.get('evaluation', {}).get('ms_per_video'), (d.get('dense_batch1') or {}).get('B1_T2000', {}).get('ms_per_step'))
print('tree [%s]: ms_per_step %.4f  repeats %s  wgrad %.4f ms  first_conv %.4f ms' % ('$t', d['ms_per_step'], d['ms_per_step_repeats'], d['roofline']['avg_launch_ms'], d['roofline_first_conv_fwd']['avg_launch_ms']) + ('  e2e / eval / dense B=1 T=2000 ms: %s' % (e,) if e[0] is not None else ''))"
  done
done
