#!/bin/bash
# what bounds the output pass: N threads writing one new 5 GiB file, by method (tools/outbench.cpp)
cd $GRAFT_REPO_ROOT
df -T /tmp . /dev/shm | cat
for TH in 1 16 64; do tools/outbench /tmp/outbench.dat 5 $TH; done
echo "-- /dev/shm"
tools/outbench /dev/shm/outbench.dat 5 16
