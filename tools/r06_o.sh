#!/bin/bash
for i in 1 2 3 4; do python tools/diag_stall2.py sample 2>/dev/null; python tools/diag_stall2.py plain 2>/dev/null; done
