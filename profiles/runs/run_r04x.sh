#!/bin/bash
# round 4, GPU session X: tests/ubench/allocbench -- the table traffic's rate against the way the 25 GiB were allocated (one process per way)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r04x && O=gpurun_out/r04x
A=tests/ubench/allocbench
{
for m in "fresh" "prefrag" "contig" "vmm 2" "vmm 64" "vmm 1024" "prefrag+vmm 2" "prefrag+vmm 64" "prefrag+vmm 1024" "prefrag+contig" "fresh" "prefrag"; do
    timeout 120 $A $m 2>&1
done
} | tee $O/allocbench.txt
