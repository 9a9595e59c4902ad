#!/bin/bash
# round 3: blind whole-line first touches vs read-modify-writes (tools/line_wall.hip); run from the repo root on the GPU box
set -e
cd "$(dirname "$0")"
for wpc in 6 4; do
  for blind in 0 25 50 60 75 100; do
    ./line_wall $blind 8 $wpc 8
  done
done
./line_wall 0 8 6 0
./line_wall 60 8 6 0
./line_wall 0 1 6 8
./line_wall 60 1 6 8
./line_wall 0 8 8 8
./line_wall 60 8 8 8
./line_wall 60 8 12 8
