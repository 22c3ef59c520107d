cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests -m gpu -x -q -k "tree or forest or ensemble" 2>&1 | tail -3
for i in 1 2; do timeout 200 python tools/treebench.py --reps 5 2>&1 | tail -1 | cut -c1-330; done
FR_TREE_NOSTAGE=1 timeout 200 python tools/treebench.py --reps 5 --check 0 2>&1 | tail -1 | cut -c1-260
FR_TREE_NOWALK=1 timeout 200 python tools/treebench.py --reps 5 --check 0 2>&1 | tail -1 | cut -c1-260
