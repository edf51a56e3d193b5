#!/bin/bash
for a in 256 257 256 257; do MMIF_ABLATE=conv=$a python tools/scratch/conv_clock.py 2>&1 | grep -v "amdgpu.ids\|WARNING"; done
