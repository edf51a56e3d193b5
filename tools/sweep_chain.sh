#!/bin/bash
for a in 0 1 2 4 3 7 0; do MMIF_ABLATE=ec=$a python3 tools/bench_chain_stream.py 2>/dev/null; done
