#!/bin/bash
# round 3: what does a blind whole-line write cost next to loads?  shapes: 0 = one lane, four 16-byte stores; 1 = a quad per line
set -e
cd "$(dirname "$0")"
for shape in 0 1; do
  ./line_wall 100 8 6 0 512 $shape 1
  ./line_wall 100 8 6 0 512 $shape 0
  ./line_wall 100 8 6 8 512 $shape 0
  ./line_wall 100 8 12 8 512 $shape 0
  ./line_wall 75 8 6 8 512 $shape 0
  ./line_wall 75 8 12 8 512 $shape 0
  ./line_wall 50 8 6 8 512 $shape 0
done
./line_wall 0 8 6 8 512 0 0
./line_wall 0 8 12 8 512 0 0
