#!/bin/bash
# round 3, session ZA: the flat match kernel's table traffic in isolation -- loads + stores against atomic exchanges, nontemporal forms, 2-byte cells (tests/ubench/tablebench.hip)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r03za && O=gpurun_out/r03za
( ./tests/ubench/tablebench 65536 384 2000; ./tests/ubench/tablebench 16384 384 4000 ) > $O/tablebench.txt 2>&1; cat $O/tablebench.txt
