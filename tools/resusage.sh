#!/bin/bash
# summarise -Rpass-analysis=kernel-resource-usage output: one line per kernel
# usage: tools/resusage.sh file.hip [extra hipcc flags]
f=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -I include -c "$f" -o /tmp/resusage.o -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | \
awk '/Function Name:/ {name=$0; sub(/.*Function Name: /,"",name); sub(/ \[-Rpass.*/,"",name)}
     /    VGPRs:/ {v=$(NF-1)} /AGPRs:/ {a=$(NF-1)} /VGPRs Spill/ {sp=$(NF-1)} /ScratchSize/ {sc=$(NF-1)} /Occupancy/ {oc=$(NF-1)}
     /LDS Size/ {print v, "vgpr", a, "agpr", sp, "spill", sc, "scratch", oc, "occ", name}
     /error/ {print}'
