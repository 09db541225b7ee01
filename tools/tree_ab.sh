#!/bin/bash
# ON THE GPU BOX: the hot-path bench of TWO TREES alternated on one box (A B A B): this tree against a frozen copy of an earlier one --
#   (here)  mkdir .r05_tree && git archive <commit> -- mucon_amd bench.py oracle include | tar -x -C .r05_tree && (cd .r05_tree && python -m mucon_amd.build)
#   (box)   bash tools/tree_ab.sh .r05_tree [bench args]
# (.r05_tree is git-ignored; its built library travels with the snapshot.)  Prints ms per step and the two event-timed launches per run.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
other=$1; shift
for rnd in 1 2; do
  for t in "$other" .; do
    (cd "$t" && python3 bench.py --steps 100 --warmup 10 --repeats 3 --no-viterbi --no-cpu-baseline --no-traffic "$@" 2>/dev/null) | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('tree [%s]: ms_per_step %.4f  repeats %s  wgrad %.4f ms  first_conv %.4f ms' % ('$t', d['ms_per_step'], d['ms_per_step_repeats'], d['roofline']['avg_launch_ms'], d['roofline_first_conv_fwd']['avg_launch_ms']))"
  done
done
