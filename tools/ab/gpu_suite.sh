#!/bin/bash
# the whole GPU suite, then the pipelined full-ranking training rates
python -m pytest tests -x -q -m gpu 2>&1 | tail -6
for m in ndcg map; do python tools/train_e2e.py --measure $m --shape 30k --restarts 32 --max-ticks 136 2>&1 | tail -1 | cut -c1-330; done
