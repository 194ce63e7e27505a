#!/bin/bash
tools/ab.sh gpurun_out/r6_exp15 default fusent default fusent default fusent default fusent
