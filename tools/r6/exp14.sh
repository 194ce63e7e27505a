#!/bin/bash
# round 6, GPU call 14: fuse kernel with non-temporal loads (and stores), same-box A/B
tools/ab.sh gpurun_out/r6_exp14 default fusent fusents default fusent fusents
