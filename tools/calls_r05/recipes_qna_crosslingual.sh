#!/bin/bash
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r05r; mkdir -p $o
timeout 200 python tools/train_recipe_from_ids.py qna --epochs 250 --lr 0.005 --eval-every 50 > $o/recipe_qna_lr0.005_250ep.txt 2>&1; echo "rc=$?"; tail -3 $o/recipe_qna_lr0.005_250ep.txt
timeout 200 python tools/train_recipe_from_ids.py qna --epochs 200 --lr 0.9 --eval-every 50 > $o/recipe_qna_lr0.9_200ep.txt 2>&1; echo "rc=$?"; tail -3 $o/recipe_qna_lr0.9_200ep.txt
timeout 200 python tools/train_recipe_from_ids.py crosslingual --epochs 40 --lr 0.005 --eval-every 20 > $o/recipe_crosslingual_lr0.005_40ep.txt 2>&1; echo "rc=$?"; tail -4 $o/recipe_crosslingual_lr0.005_40ep.txt
