#!/bin/bash
# phase stamps of the timing builds under build/variants (t*.so): bash profiles/run_phases.sh   (on the GPU box)
for so in build/variants/t*.so; do
  name=$(basename $so .so)
  echo "== $name"
  LFD_DENSE_TIMING=/tmp/stamps_$name.bin LFD_DENSIFY_LIB=$so python bench.py --cpu-sample-refs 0 --steps 2 --warmup 1 --light > /dev/null 2>&1
  python profiles/dense_phases.py /tmp/stamps_$name.bin | head -28
  python profiles/phases_by_generation.py /tmp/stamps_$name.bin
done
