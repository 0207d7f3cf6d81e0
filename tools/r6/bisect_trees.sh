#!/bin/bash
# round 6: how the same-box comparison of older trees behind profiles/r6_inflight_scratch_ab.json was run.
# gpurun snapshots /root/repo, so an older tree is exported NEXT TO the current one (git-ignored, not gpurun-ignored), built in
# place, and its own bench.py is run from its own directory inside ONE gpurun call together with the current tree:
#
#   git archive 379b14d | tar -x -C _r4tree      # end of round 4
#   git archive 0e00ebc | tar -x -C _r5a         # the next commit (own scratch, routing scopes)
#   (cd _r4tree && python -m hallo_amd.build); (cd _r5a && python -m hallo_amd.build)
#   gpurun -- 'cd _r4tree && python bench.py --no-cpu-baseline --no-profile --steps 12 --warmup 3 --no-serial-leg > ../gpurun_out/r4tree.json; cd ../_r5a && ...'
#
# Hybrids (one file of one tree copied over the other's: ops.py, face_animate.py, bench.py, then single functions of ops.py)
# narrowed the 10 % down to groupnorm()'s scratch selection, and a probe appended to the old bench.py (nan / inf fraction, mean and
# share of exact 0 / 1 values of the last timed clips' frames) showed what the fast executions compute: all-zero frames.
# The trees are not kept in the repository; the commands above recreate them.
echo "see the comment block: the trees are recreated with git archive" >&2
