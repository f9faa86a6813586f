#!/bin/bash
# the same pytest selection in four processes side by side (what pytest -n 4 does to the GPU); prints every failure's assertion
for k in 1 2 3; do (timeout 200 python -m pytest tests -m gpu -q -x -k "$1" > /tmp/r4_$k.txt 2>&1 &); done
timeout 200 python -m pytest tests -m gpu -q -x -k "$1" > /tmp/r4_0.txt 2>&1
sleep 6
for k in 0 1 2 3; do tail -n 1 /tmp/r4_$k.txt; grep -h "AssertionError\|^E  " /tmp/r4_$k.txt | head -6; done
