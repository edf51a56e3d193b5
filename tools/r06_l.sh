#!/bin/bash
for i in 1 2 3; do python tools/diag_stall.py gc 2>/dev/null; python tools/diag_stall.py nogc 2>/dev/null; done
